// BestSpeed (level 1) matcher: one 64-lane wave parses one <= 32 KiB fragment.
//
// Replaces snappy.nim:12-136 encodeFragment (+ the addLiteral/addCopy
// bookkeeping of snappy.nim:33-64) and deflate.nim:153-177 encodeAllLiterals
// (level -2, `huffman_only`).  The parse is the reference's greedy single-probe
// parse, reproduced decision for decision (same hash, same table size rule,
// same skip-ahead schedule, same "no match starts in the last 15 bytes" rule),
// so the match list equals the reference's token stream fragment by fragment
// (tests compare them token for token through zh_debug_tokens).
//
// LDS per wave: the fragment's bytes (32 KiB + pad), the u16 hash table
// (32 KiB, snappy.nim:7), the symbol histograms and a coverage bitmap.
// Output per fragment (HBM scratch): the match list (start, length, offset as
// u16 SoA), the litlen/distance histograms (u16 x 320), literal count and the
// sum of extra bits -- everything the Huffman and emission kernels need.
#include <cstdlib>

#include "zh_common.h"
#include "zh_tables.h"

namespace {
__constant__ zh::LenTables c_len = zh::make_len_tables();
__constant__ zh::DistTables c_dist = zh::make_dist_tables();
constexpr uint32_t kHashMul = 0x1e35a7bdu;  // snappy.nim:70-71
}  // namespace

__global__ __launch_bounds__(64) void zh_l1_match_kernel(const uint8_t* __restrict__ d_src,
                                                         ZhCompressArgs a, int huffman_only,
                                                         int serial_parse) {
  __shared__ __attribute__((aligned(16))) uint32_t s_src[ZH_FRAG_SIZE / 4 + 8];
  __shared__ uint16_t s_table[16384];
  __shared__ uint32_t s_hist[ZH_HIST_STRIDE];
  __shared__ uint32_t s_cover[ZH_FRAG_SIZE / 32];  // bit p set: byte p lies inside a match
  __shared__ uint32_t s_nmatch;
  __shared__ uint32_t s_scr[1024];  // per-step hash collision counters

  const unsigned lane = zh_lane();
  const uint32_t f = blockIdx.x;
  const ZhFragDesc fd = a.frags[f];
  const uint32_t n = fd.len;
  const uint8_t* src = d_src + fd.src_off;
  uint16_t* m_pos = a.m_pos + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  uint16_t* m_len = a.m_len + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  uint16_t* m_off = a.m_off + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;

  // ---- stage the fragment into LDS (coalesced), clear state ----
  {
    const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t* asrc = reinterpret_cast<const uint32_t*>(src - mis);
    const uint32_t nwords = (n + 3) / 4;
    const uint32_t nalign = (n + mis + 3) / 4;  // aligned dwords that hold fragment bytes
    for (uint32_t w = lane; w < nwords + 2; w += 64) {
      uint32_t v = 0;
      if (w < nwords) {
        // aligned dword pair -> the 4 bytes at fragment offset 4w (never reads past the
        // aligned dword that holds the fragment's last byte)
        const uint32_t lo = asrc[w];
        const uint32_t hi = (mis && w + 1 < nalign) ? asrc[w + 1] : 0u;
        v = __builtin_amdgcn_alignbyte(hi, lo, mis);
        const uint32_t valid = n - 4 * w;  // bytes of this word inside the fragment
        if (valid < 4) v &= (1u << (8 * valid)) - 1u;
      }
      s_src[w] = v;
    }
    for (uint32_t i = lane; i < ZH_HIST_STRIDE; i += 64) s_hist[i] = 0;
    for (uint32_t i = lane; i < ZH_FRAG_SIZE / 32; i += 64) s_cover[i] = 0;
    for (uint32_t i = lane; i < 1024; i += 64) s_scr[i] = 0;
  }

  uint32_t table_size = 256, shift = 24;  // snappy.nim:24-29
  while (table_size < 16384u && table_size < n) {
    table_size <<= 1;
    shift--;
  }
  if (!huffman_only)
    for (uint32_t i = lane; i < table_size / 2; i += 64) reinterpret_cast<uint32_t*>(s_table)[i] = 0;
  zh_wave_sync();

  // ---- greedy parse, wave-parallel (default) ----
  // The reference probes one position at a time: h = hash(load32(p)); cand = table[h];
  // table[h] = p; hit iff load32(p) == load32(cand) (snappy.nim:86-101).  Here the 64
  // lanes take the next 64 probe positions of the skip-ahead schedule at once and the
  // first hit in probe order is found with ballots.  A lane's candidate must reflect
  // the table inserts of the EARLIER probes of the same step: lanes whose hash equals
  // an earlier lane's are resolved one by one in probe order, everything else in
  // parallel.  Only the probes up to the first hit insert into the table, so the table
  // evolves exactly as in the serial walk and the parse is identical.
  //
  // A wave alone on its SIMD issues ~1 instruction per 4-5 cycles and an LDS round trip
  // costs ~125 cycles, so the step is built to need three dependent round trips (source
  // bytes, table, candidate bytes): the insert of ip-1 that follows a match rides along
  // as lane 0 of the next step, and every lane compares 16 bytes against its candidate so
  // that matches shorter than 16 need no further read.
  if (!serial_parse) {
    uint32_t nm = 0;
    uint32_t rp = 0, rl = 0, ro = 0;  // match list staging: lane (nm & 63) holds match nm
    if (!huffman_only && n >= 15) {
      const uint32_t ip_limit = n - 15;
      // lane roles of a step that follows a match (snappy.nim:116-131): lane 0 inserts
      // ip-1, lane 1 re-probes ip, lane j >= 2 is probe j-2 of the run that starts at ip+1
      uint32_t post_off, post_step;
      {
        const uint32_t j = lane - 2;
        post_off = lane == 0 ? 0xffffffffu : lane == 1 ? 0u : 1u + j + (j > 32u ? j - 32u : 0u);
        post_step = lane < 2 ? 0u : (32u + j) >> 5;
      }
      uint32_t ip = 1;     // position of the next probe (post: position right after a match)
      uint32_t K = 0;      // probes already done in this literal run (skip = 32 + K)
      bool post = false;
      for (;;) {
        uint32_t pos, step;
        if (post) {
          pos = ip + post_off;
          step = post_step;
        } else {  // closed form of the skip>>5 schedule, snappy.nim:88-92
          const uint32_t s = 32 + K, q = s >> 5, r = s & 31u, j = lane;
          uint32_t off = q * j;
          if (j > 32u - r) off += j - (32u - r);
          if (j > 64u - r) off += j - (64u - r);
          pos = ip + off;
          step = (s + j) >> 5;
        }
        const bool valid = pos + step <= ip_limit;  // lanes 0/1 of a post step: step 0, ip < ip_limit
        if (!valid) pos = 1;  // keep LDS reads in range; the lane is ignored
        // round trip 1: 16 source bytes at pos (five aligned dwords)
        const uint32_t pw = pos >> 2;
        const uint32_t p0 = s_src[pw], p1 = s_src[pw + 1], p2 = s_src[pw + 2], p3 = s_src[pw + 3],
                       p4 = s_src[pw + 4];
        const uint32_t a0 = __builtin_amdgcn_alignbyte(p1, p0, pos);
        const uint32_t h = (a0 * kHashMul) >> shift;
        // round trip 2: the table; the first 16 probes also tick a duplicate-hash counter
        // (answer read back together with the candidate bytes, so no extra round trip)
        const bool counted = valid && lane < 16;
        const uint32_t old = s_table[h];
        if (counted) atomicAdd(&s_scr[h & 1023u], 1u);
        zh_wave_sync();  // (orders the counter traffic between lanes; emits nothing)
        const uint32_t a1 = __builtin_amdgcn_alignbyte(p2, p1, pos);
        const uint32_t a2 = __builtin_amdgcn_alignbyte(p3, p2, pos);
        const uint32_t a3 = __builtin_amdgcn_alignbyte(p4, p3, pos);
        // round trip 3: 16 bytes at the candidate the table held when the step began
        const uint32_t ow = old >> 2;
        const uint32_t q0 = s_src[ow], q1 = s_src[ow + 1], q2 = s_src[ow + 2], q3 = s_src[ow + 3],
                       q4 = s_src[ow + 4];
        const uint32_t cnt16 = counted ? s_scr[h & 1023u] : 0u;
        zh_wave_sync();
        if (counted) s_scr[h & 1023u] = 0;
        const uint32_t x0 = a0 ^ __builtin_amdgcn_alignbyte(q1, q0, old);
        const uint32_t x1 = a1 ^ __builtin_amdgcn_alignbyte(q2, q1, old);
        const uint32_t x2 = a2 ^ __builtin_amdgcn_alignbyte(q3, q2, old);
        const uint32_t x3 = a3 ^ __builtin_amdgcn_alignbyte(q4, q3, old);
        // equal leading bytes 0..16, branch-free: ffs(0) - 1 = 0xffffffff -> min(.., 4) = 4
        const uint32_t c0 = min(((uint32_t)__ffs((int)x0) - 1u) >> 3, 4u);
        const uint32_t c1 = min(((uint32_t)__ffs((int)x1) - 1u) >> 3, 4u);
        const uint32_t c2 = min(((uint32_t)__ffs((int)x2) - 1u) >> 3, 4u);
        const uint32_t c3 = min(((uint32_t)__ffs((int)x3) - 1u) >> 3, 4u);
        const uint32_t c23 = c2 + (c2 == 4u ? c3 : 0u);
        const uint32_t c123 = c1 + (c1 == 4u ? c23 : 0u);
        const uint32_t eqlen = c0 + (c0 == 4u ? c123 : 0u);
        const bool hit_old = valid && x0 == 0 && !(post && lane == 0);

        const uint64_t V = __ballot(valid);
        const uint64_t H = __ballot(hit_old);
        const uint32_t t = V == ~0ull ? 64u : (uint32_t)__ffsll((long long)~V) - 1u;  // first probe past ip_limit
        const uint32_t g0 = H ? (uint32_t)__ffsll((long long)H) - 1u : 64u;
        // lanes whose inserts can happen in this step: up to the first hit, or up to the limit
        const uint32_t span = g0 < t ? g0 + 1 : t;
        // C: superset of the lanes (below span) that share their hash with another lane
        uint64_t C;
        bool c_full = false;  // C covers every valid lane, not only those below span
        if (span <= 16) {
          C = __ballot(cnt16 > 1u) & ((1ull << span) - 1ull);
        } else {
          zh_wave_sync();
          if (valid) atomicAdd(&s_scr[h & 1023u], 1u);
          zh_wave_sync();
          const bool coll = valid && s_scr[h & 1023u] > 1u;
          zh_wave_sync();
          if (valid) s_scr[h & 1023u] = 0;
          C = __ballot(coll);
          c_full = true;
        }
        uint32_t f = g0 < t ? g0 : 64u, cand_lane_old = f;  // first hit in probe order
        uint32_t cand = 0, flen = 0;
        bool cand_in_step = false;
        if (C) {
          if (!c_full && g0 < 64 && ((C >> g0) & 1ull)) {
            // the first apparent hit may be void (an earlier probe of the step re-used its
            // table slot): later lanes come into play, so classify all of them
            zh_wave_sync();
            if (valid) atomicAdd(&s_scr[h & 1023u], 1u);
            zh_wave_sync();
            const bool coll = valid && s_scr[h & 1023u] > 1u;
            zh_wave_sync();
            if (valid) s_scr[h & 1023u] = 0;
            C = __ballot(coll);
          }
          // exact resolution of the colliding probes before the first clean event
          const uint64_t clean_hits = H & ~C;
          const uint32_t g = clean_hits ? (uint32_t)__ffsll((long long)clean_hits) - 1u : 64u;
          const uint32_t bound = g < t ? g : t;
          uint64_t cb = bound >= 64 ? C : (C & ((1ull << bound) - 1ull));
          f = 64;
          while (cb) {
            const uint32_t jx = (uint32_t)__ffsll((long long)cb) - 1u;
            cb &= cb - 1;
            if (post && jx == 0) continue;  // the insert-only lane never hits
            const uint32_t hj = __builtin_amdgcn_readlane(h, jx);
            const uint64_t same = __ballot(valid && h == hj) & ((1ull << jx) - 1ull);
            if (same) {  // an earlier probe of this step inserted this hash last
              const uint32_t i = 63u - (uint32_t)__clzll((long long)same);
              if (__builtin_amdgcn_readlane(a0, i) == __builtin_amdgcn_readlane(a0, jx)) {
                f = jx;
                cand = __builtin_amdgcn_readlane(pos, i);
                cand_in_step = true;
                break;
              }
            } else if ((H >> jx) & 1ull) {
              f = jx;
              cand_lane_old = jx;
              break;
            }
          }
          if (f == 64 && g < t) {
            f = g;
            cand_lane_old = g;
          }
        }
        // table inserts of the probes that really happened
        const uint32_t last_plus1 = f < 64 ? f + 1 : t;
        if (lane < last_plus1 && !((C >> lane) & 1ull)) s_table[h] = (uint16_t)pos;
        if (C) {
          uint64_t cc = last_plus1 >= 64 ? C : (C & ((1ull << last_plus1) - 1ull));
          while (cc) {  // same-hash probes write in probe order (the later one wins)
            const uint32_t jx = (uint32_t)__ffsll((long long)cc) - 1u;
            cc &= cc - 1;
            if (lane == jx) s_table[h] = (uint16_t)pos;
          }
        }
        if (f < 64) {
          const uint32_t mp = __builtin_amdgcn_readlane(pos, f);
          const uint32_t limit = n < mp + 258u ? n : mp + 258u;
          uint32_t matched;
          if (!cand_in_step) {
            cand = __builtin_amdgcn_readlane(old, cand_lane_old);
            flen = __builtin_amdgcn_readlane(eqlen, f);
          } else {
            flen = 4;  // only the first four bytes are known to match
          }
          if (flen < 16u && !cand_in_step) {
            matched = flen;
          } else {
            // 4 + determineMatchLength(cand + 4, mp + 4, limit), internal.nim:251-270:
            // lane l compares bytes flen+4l .. flen+3+4l against the candidate
            zh_wave_sync();
            const uint32_t o = flen + 4u * lane;
            uint32_t avail = 0;
            if (mp + o < limit) avail = limit - (mp + o) < 4u ? limit - (mp + o) : 4u;
            const uint32_t x = zh_ld32(s_src, mp + o) ^ zh_ld32(s_src, cand + o);
            uint32_t eq = min(((uint32_t)__ffs((int)x) - 1u) >> 3, 4u);
            if (eq > avail) eq = avail;
            const uint64_t stop = __ballot(eq < 4u);  // some lane always stops (limit <= mp + 258)
            const uint32_t fl = (uint32_t)__ffsll((long long)stop) - 1u;
            matched = flen + 4u * fl + __builtin_amdgcn_readlane(eq, fl);
          }
          if (matched > limit - mp) matched = limit - mp;
          if (lane == (nm & 63u)) {
            rp = mp;
            rl = matched;
            ro = mp - cand;
          }
          nm++;
          if ((nm & 63u) == 0) {  // 64 staged matches -> one coalesced store per array
            m_pos[nm - 64 + lane] = (uint16_t)rp;
            m_len[nm - 64 + lane] = (uint16_t)rl;
            m_off[nm - 64 + lane] = (uint16_t)ro;
          }
          ip = mp + matched;
          if (ip >= ip_limit) break;  // snappy.nim:118-120
          post = true;
          K = 0;
          continue;
        }
        if (t < 64) break;  // snappy.nim:93-95: the rest of the fragment is literals
        // 64 probes without a hit: continue the same literal run
        const uint32_t p63 = __builtin_amdgcn_readlane(pos, 63), s63 = __builtin_amdgcn_readlane(step, 63);
        K = post ? 62 : K + 64;
        post = false;
        ip = p63 + s63;
      }
    }
    if (lane < (nm & 63u)) {
      const uint32_t b = nm & ~63u;
      m_pos[b + lane] = (uint16_t)rp;
      m_len[b + lane] = (uint16_t)rl;
      m_off[b + lane] = (uint16_t)ro;
    }
    if (lane == 0) s_nmatch = nm;
  }
  // ---- greedy parse, one lane walks the fragment (snappy.nim:76-136 verbatim order;
  // kept as the A/B reference for the wave-parallel parse above) ----
  if (serial_parse && lane == 0) {
    uint32_t nm = 0;
    if (!huffman_only && n >= 15) {
      const uint32_t ip_limit = n - 15;
      uint32_t ip = 1;
      uint32_t next_hash = (zh_ld32(s_src, ip) * kHashMul) >> shift;
      bool done = false;
      while (!done) {
        uint32_t skip = 32, next_ip = ip, cand = 0;
        for (;;) {  // probe loop, snappy.nim:86-101
          ip = next_ip;
          const uint32_t h = next_hash;
          const uint32_t step = skip >> 5;
          skip++;
          next_ip = ip + step;
          if (next_ip > ip_limit) { done = true; break; }
          next_hash = (zh_ld32(s_src, next_ip) * kHashMul) >> shift;
          cand = s_table[h];
          s_table[h] = (uint16_t)ip;
          if (zh_ld32(s_src, ip) == zh_ld32(s_src, cand)) break;
        }
        if (done) break;
        for (;;) {  // match + immediate re-probe, snappy.nim:108-131
          const uint32_t limit = n < ip + 258u ? n : ip + 258u;
          uint32_t s1 = cand + 4, s2 = ip + 4;  // internal.nim:251-270
          while (s2 + 4 <= limit && zh_ld32(s_src, s1) == zh_ld32(s_src, s2)) { s1 += 4; s2 += 4; }
          while (s2 < limit && zh_ld8(s_src, s1) == zh_ld8(s_src, s2)) { s1++; s2++; }
          const uint32_t matched = s2 - ip;
          m_pos[nm] = (uint16_t)ip;
          m_len[nm] = (uint16_t)matched;
          m_off[nm] = (uint16_t)(ip - cand);
          nm++;
          ip += matched;
          if (ip >= ip_limit) { done = true; break; }
          const uint64_t input = zh_ld64(s_src, ip - 1);
          const uint32_t prev_hash = ((uint32_t)input * kHashMul) >> shift;
          const uint32_t cur = (uint32_t)(input >> 8);
          const uint32_t cur_hash = (cur * kHashMul) >> shift;
          s_table[prev_hash] = (uint16_t)(ip - 1);
          cand = s_table[cur_hash];
          s_table[cur_hash] = (uint16_t)ip;
          if (cur != zh_ld32(s_src, cand)) {
            next_hash = ((uint32_t)(input >> 16) * kHashMul) >> shift;
            ip++;
            break;
          }
        }
      }
    }
    s_nmatch = nm;
  }
  __threadfence_block();
  zh_wave_sync();
  const uint32_t nmatch = s_nmatch;

  // ---- coverage bitmap + match histograms (lanes over matches) ----
  uint32_t extra_bits = 0, covered = 0;
  for (uint32_t m = lane; m < nmatch; m += 64) {
    const uint32_t p = m_pos[m], l = m_len[m], o = m_off[m];
    const uint32_t li = c_len.index_of[l - 3], di = zh_dist_code(o);
    atomicAdd(&s_hist[257 + li], 1u);
    atomicAdd(&s_hist[ZH_NUM_LITLEN + di], 1u);
    extra_bits += c_len.extra[li] + c_dist.extra[di];
    covered += l;
    const uint32_t e = p + l;  // set bits [p, e)
    for (uint32_t w = p >> 5; w <= (e - 1) >> 5; w++) {
      const uint32_t lo = w == (p >> 5) ? (p & 31u) : 0u;
      const uint32_t hi = w == ((e - 1) >> 5) ? ((e - 1) & 31u) : 31u;
      const uint32_t mask = (0xffffffffu >> (31u - hi)) & (0xffffffffu << lo);
      atomicOr(&s_cover[w], mask);
    }
  }
  zh_wave_sync();
  // ---- literal histogram (lanes over positions) ----
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t p = base + lane;
    if (p < n && !((s_cover[p >> 5] >> (p & 31u)) & 1u)) atomicAdd(&s_hist[zh_ld8(s_src, p)], 1u);
  }
  extra_bits = zh_wave_sum(extra_bits);
  covered = zh_wave_sum(covered);
  zh_wave_sync();

  uint16_t* hist_out = a.f_hist + (size_t)f * ZH_HIST_STRIDE;
  for (uint32_t i = lane; i < ZH_HIST_STRIDE; i += 64) hist_out[i] = (uint16_t)s_hist[i];
  if (lane == 0) {
    a.f_nmatch[f] = nmatch;
    a.f_spill[f] = 0;
    a.f_nlit[f] = n - covered;
    a.f_extra_bits[f] = extra_bits;
  }
}

extern "C" void zh_launch_l1_match(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a,
                                   int huffman_only) {
  if (!a.nfrags) return;
  static const int serial_parse = getenv("ZH_L1_SERIAL") != nullptr;  // A/B switch, see kernel
  hipLaunchKernelGGL(zh_l1_match_kernel, dim3(a.nfrags), dim3(64), 0, stream, d_src, a,
                     huffman_only, serial_parse);
}
