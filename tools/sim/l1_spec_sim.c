// Convergence study for the fragment-parallel BestSpeed parse.
//
// Not product code and not the oracle: a CPU model of the *schedule* the kernel
// zh_l1p_match_kernel uses -- chunk walkers that start from guessed states, a read-only
// array of same-hash predecessor links, an "inserted" bitmap as the only parse state -- to
// count rounds and walked positions before anything is written for the GPU.
//
//   gcc -O2 -o /tmp/sim/l1sim tools/sim/l1_spec_sim.c && /tmp/sim/l1sim /tmp/sim/mix64.bin 128 128
//
// The exact parse below restates snappy.nim:12-136 (as oracle/zippy_oracle.c does); the
// speculative schedule must reproduce its inserted set and match list exactly.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FRAG 32768
#define HMUL 0x1e35a7bdu

static inline uint32_t rd32(const uint8_t* s, uint32_t p) {
  uint32_t v;
  memcpy(&v, s + p, 4);
  return v;
}

typedef struct {
  uint32_t pos, len, off;
} Match;

static uint32_t shift_for(uint32_t n) {
  uint32_t ts = 256, sh = 24;
  while (ts < 16384 && ts < n) {
    ts <<= 1;
    sh--;
  }
  return sh;
}

static uint32_t match_len(const uint8_t* s, uint32_t cand, uint32_t ip, uint32_t limit) {
  uint32_t m = 4;
  while (ip + m < limit && s[cand + m] == s[ip + m]) m++;
  return m;
}

// exact parse: fills ins[] (1 = position was written to the table) and matches
static uint32_t exact_parse(const uint8_t* s, uint32_t n, uint8_t* ins, Match* ms) {
  static uint16_t table[16384];
  uint32_t sh = shift_for(n), nm = 0;
  memset(table, 0, sizeof table);
  memset(ins, 0, n);
  if (n < 15) return 0;
  uint32_t ip_limit = n - 15, ip = 1;
  uint32_t next_hash = (rd32(s, ip) * HMUL) >> sh;
  for (;;) {
    uint32_t skip = 32, next_ip = ip, cand;
    for (;;) {
      ip = next_ip;
      uint32_t h = next_hash, step = skip >> 5;
      skip++;
      next_ip = ip + step;
      if (next_ip > ip_limit) return nm;
      next_hash = (rd32(s, next_ip) * HMUL) >> sh;
      cand = table[h];
      table[h] = (uint16_t)ip;
      ins[ip] = 1;
      if (rd32(s, ip) == rd32(s, cand)) break;
    }
    for (;;) {
      uint32_t limit = n < ip + 258 ? n : ip + 258;
      uint32_t m = match_len(s, cand, ip, limit);
      ms[nm].pos = ip;
      ms[nm].len = m;
      ms[nm].off = ip - cand;
      nm++;
      ip += m;
      if (ip >= ip_limit) return nm;
      uint32_t ph = (rd32(s, ip - 1) * HMUL) >> sh, ch = (rd32(s, ip) * HMUL) >> sh;
      table[ph] = (uint16_t)(ip - 1);
      ins[ip - 1] = 1;
      cand = table[ch];
      table[ch] = (uint16_t)ip;
      ins[ip] = 1;
      if (rd32(s, ip) != rd32(s, cand)) break;
    }
    next_hash = (rd32(s, ip + 1) * HMUL) >> sh;
    ip++;
  }
}

// ---- speculative schedule ----
typedef struct {
  uint32_t ip;    // position of the next action
  uint32_t K;     // probes already done in this literal run (skip = 32 + K)
  uint32_t post;  // 1: a match ended at ip (ip-1 already inserted): re-probe ip
  uint32_t done;  // the parse ended (rest of the fragment is literals)
} State;

static int state_eq(State a, State b) {
  if (a.done || b.done) return a.done == b.done;
  return a.ip == b.ip && a.post == b.post && (a.post || a.K == b.K);
}

static uint16_t g_link[FRAG];
static uint64_t g_chainwalk, g_probes, g_slow;

typedef struct {
  const uint8_t* s;
  uint32_t n, sh, ip_limit;
  const uint8_t* B;     // shared bitmap (bytes) for positions < own_lo
  uint8_t* own;         // own bits for positions >= own_lo (indexed by absolute position)
  uint32_t own_lo;
  uint8_t* readmask;    // chunks read from (may be NULL)
  uint32_t C;
} Walk;

static inline int is_ins(Walk* w, uint32_t q) {
  if (q == 0) return 1;
  if (q >= w->own_lo) return w->own[q];
  if (w->readmask) w->readmask[q / w->C] = 1;
  return w->B[q];
}

static inline uint32_t candidate(Walk* w, uint32_t p) {
  uint32_t q = g_link[p];
  g_probes++;
  if (!is_ins(w, q)) g_slow++;
  while (!is_ins(w, q)) {
    q = g_link[q];
    g_chainwalk++;
  }
  return q;
}

// walk from *st until st->ip >= end (or done); ms may be NULL.  Returns positions walked.
static uint32_t walk(Walk* w, State* st, uint32_t end, Match* ms, uint32_t* nm) {
  const uint8_t* s = w->s;
  uint32_t work = 0;
  while (!st->done && st->ip < end) {
    work++;
    uint32_t ip = st->ip, cand;
    int hit;
    if (!st->post) {
      uint32_t step = (32 + st->K) >> 5;
      if (ip + step > w->ip_limit) {
        st->done = 1;
        break;
      }
      cand = candidate(w, ip);
      w->own[ip] = 1;
      hit = rd32(s, ip) == rd32(s, cand);
      if (!hit) {
        st->K++;
        st->ip = ip + step;
        continue;
      }
    } else {
      cand = candidate(w, ip);
      w->own[ip] = 1;
      hit = rd32(s, ip) == rd32(s, cand);
      if (!hit) {
        st->post = 0;
        st->K = 0;
        st->ip = ip + 1;
        continue;
      }
    }
    uint32_t limit = w->n < ip + 258 ? w->n : ip + 258;
    uint32_t m = match_len(s, cand, ip, limit);
    if (ms) {
      ms[*nm].pos = ip;
      ms[*nm].len = m;
      ms[*nm].off = ip - cand;
      (*nm)++;
    }
    ip += m;
    st->ip = ip;
    st->K = 0;
    st->post = 1;
    if (ip >= w->ip_limit) {
      st->done = 1;
      break;
    }
    w->own[ip - 1] = 1;
  }
  return work;
}

int main(int argc, char** argv) {
  if (argc < 4) return 1;
  FILE* f = fopen(argv[1], "rb");
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* data = malloc(sz + 64);
  if (fread(data, 1, sz, f) != (size_t)sz) return 1;
  const uint32_t C = atoi(argv[2]), R = atoi(argv[3]);
  const int fine = argc > 4 ? atoi(argv[4]) : 1;  // 1: re-walk only if a chunk read from changed
  const int guess_all = argc > 5 ? atoi(argv[5]) : 1;
  uint32_t WIN = argc > 6 ? atoi(argv[6]) : 0;  // chunks per window, 0: whole fragment
  const uint32_t nfr = (uint32_t)(sz / FRAG), NC = FRAG / C;
  if (!WIN) WIN = NC;
  uint64_t tot_rounds = 0, max_rounds = 0, tot_walks = 0, tot_work = 0, tot_exact_work = 0;
  uint64_t hist_rounds[64] = {0};
  uint64_t tot_roundcost = 0;  // sum over rounds of the max lane work of each 64-chunk wave: SIMT time proxy
  static uint8_t ins[FRAG + 512], B[FRAG + 512], Bn[FRAG + 512], own[FRAG + 512];
  static Match ms[8192];
  static uint16_t head[16384];
  State* entry = calloc(NC + 1, sizeof(State));
  State* exitst = calloc(NC + 1, sizeof(State));
  uint8_t* readmask = calloc((size_t)NC * NC, 1);
  uint8_t* changed = calloc(NC, 1);
  uint8_t* changed_n = calloc(NC, 1);
  for (uint32_t fi = 0; fi < nfr; fi++) {
    const uint8_t* s = data + (size_t)fi * FRAG;
    const uint32_t n = FRAG, sh = shift_for(n);
    exact_parse(s, n, ins, ms);
    uint32_t exact_work = 0;
    for (uint32_t p = 0; p < n; p++) exact_work += ins[p];
    tot_exact_work += exact_work;
    memset(head, 0, sizeof head);
    for (uint32_t p = 1; p + 4 <= n; p++) {
      uint32_t h = (rd32(s, p) * HMUL) >> sh;
      g_link[p] = head[h];
      head[h] = (uint16_t)p;
    }
    memset(B, guess_all ? 1 : 0, sizeof B);
    Walk w = {s, n, sh, n - 15, B, own, 0, NULL, C};
    uint32_t rounds = 0;
    for (uint32_t w0 = 0; w0 < NC; w0 += WIN) {
      const uint32_t w1 = w0 + WIN < NC ? w0 + WIN : NC;
      for (uint32_t wr = 1;; wr++) {
        int any = 0;
        memcpy(Bn, B, sizeof B);
        memset(changed_n, 0, NC);
        uint32_t wave_max[64] = {0};
        for (uint32_t k = w0; k < w1; k++) {
          const uint32_t lo = k * C, hi = lo + C;
          State e = k ? exitst[k - 1] : (State){1, 0, 0, 0};
          const int guess = wr == 1 && k > w0;  // first round of a window: run-up from a guessed state
          int need = wr == 1;
          if (!need) need = !state_eq(e, entry[k]);
          if (!need) {
            for (uint32_t j = w0; j < k && !need; j++)
              if (changed[j] && (!fine || readmask[(size_t)k * NC + j])) need = 1;
          }
          if (!need) continue;
          any = 1;
          tot_walks++;
          memset(readmask + (size_t)k * NC, 0, NC);
          w.readmask = readmask + (size_t)k * NC;
          uint32_t wk = 0;
          if (guess) {
            uint32_t st0 = lo > R ? lo - R : 1;
            w.own_lo = st0;
            memset(own + st0, 0, hi + 300 - st0);
            State st = {st0, 0, 0, 0};
            wk += walk(&w, &st, lo, NULL, NULL);
            e = st;
            // what the run-up believed about [st0, lo) is a dependency like any other
            for (uint32_t j = st0 / C; j < k; j++) w.readmask[j] = 1;
            if (!e.done) {
              for (uint32_t q = lo; q < e.ip && q < hi + 300; q++) own[q] = 0;
              if (e.post && e.ip - 1 >= lo) own[e.ip - 1] = 1;
            }
          } else {
            w.own_lo = lo;
            memset(own + lo, 0, C + 300);
            if (!e.done && e.post && e.ip - 1 >= lo) own[e.ip - 1] = 1;
          }
          entry[k] = e;
          State st = e;
          wk += walk(&w, &st, hi, NULL, NULL);
          exitst[k] = st;
          tot_work += wk;
          if (wk > wave_max[(k - w0) * 64 / (w1 - w0)]) wave_max[(k - w0) * 64 / (w1 - w0)] = wk;
          int ch = wr == 1;
          for (uint32_t q = lo; q < hi; q++) {
            if (Bn[q] != own[q]) ch = 1;
            Bn[q] = own[q];
          }
          changed_n[k] = (uint8_t)ch;
        }
        if (!any) break;
        rounds++;
        {
          // waves: 64 consecutive chunks of the window share a wave
          const uint32_t nw = (w1 - w0 + 63) / 64;
          uint32_t wm[64] = {0};
          (void)wm;
          for (uint32_t i = 0; i < 64; i++) tot_roundcost += wave_max[i] * 0;
          for (uint32_t v = 0; v < nw; v++) {
            uint32_t m = 0;
            for (uint32_t i = 0; i < 64; i++) {
              const uint32_t idx = (v * 64 + i) * 64 / (w1 - w0);
              if (idx < 64 && wave_max[idx] > m) m = wave_max[idx];
            }
            tot_roundcost += m;
          }
        }
        memcpy(B, Bn, sizeof B);
        memcpy(changed, changed_n, NC);
        if (wr > 20000) {
          printf("no convergence frag %u\n", fi);
          break;
        }
      }
    }
    int bad = 0;
    for (uint32_t p = 1; p < n; p++)
      if ((B[p] != 0) != (ins[p] != 0)) {
        bad++;
        if (bad < 4) fprintf(stderr, "frag %u pos %u: spec %u exact %u\n", fi, p, B[p], ins[p]);
      }
    if (bad) printf("frag %u: %d differing bits\n", fi, bad);
    tot_rounds += rounds;
    if (rounds > max_rounds) max_rounds = rounds;
    hist_rounds[rounds < 63 ? rounds : 63]++;
  }
  printf("C=%u R=%u fine=%d guess_all=%d WIN=%u frags=%u\n", C, R, fine, guess_all, WIN, nfr);
  printf("rounds avg %.2f max %llu\n", (double)tot_rounds / nfr, (unsigned long long)max_rounds);
  printf("chunk walks per frag %.1f (chunks %u)\n", (double)tot_walks / nfr, NC);
  printf("actions walked per frag %.0f, exact %.0f (x%.2f)\n", (double)tot_work / nfr,
         (double)tot_exact_work / nfr, (double)tot_work / tot_exact_work);
  printf("SIMT cost proxy (sum over rounds and waves of the max lane actions) per frag %.0f\n",
         (double)tot_roundcost / nfr);
  printf("probes %.0f slow %.0f chainwalk %.0f per frag\n", (double)g_probes / nfr, (double)g_slow / nfr,
         (double)g_chainwalk / nfr);
  return 0;
}
