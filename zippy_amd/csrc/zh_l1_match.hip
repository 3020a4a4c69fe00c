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
  __shared__ uint32_t s_scr[128];  // per-step hash collision counters

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
    for (uint32_t i = lane; i < 128; i += 64) s_scr[i] = 0;
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
  // the table inserts of the EARLIER probes of the same step; lanes whose hash
  // collides with another lane of the step (detected with 128 LDS counters) are
  // resolved one by one in probe order, everything else in parallel.  Only the probes
  // up to the first hit insert into the table, so the table evolves exactly as in the
  // serial walk and the parse is identical.
  if (!serial_parse) {
    uint32_t nm = 0;
    if (!huffman_only && n >= 15) {
      const uint32_t ip_limit = n - 15;
      uint32_t ip = 1;     // position of the next probe (post: position right after a match)
      uint32_t K = 0;      // probes already done in this literal run (skip = 32 + K)
      bool post = false;   // the step starts with the re-probe that follows a match
      for (;;) {
        // probe positions of this step (closed form of skip>>5 steps, snappy.nim:88-92)
        uint32_t j = lane, s = 32 + K, base = ip;
        bool valid = true;
        if (post) {  // snappy.nim:116-131: insert ip-1, then probe ip, then the normal run from ip+1
          if (lane == 0) s_table[(zh_ld32(s_src, ip - 1) * kHashMul) >> shift] = (uint16_t)(ip - 1);
          s = 32;
          base = ip + 1;
          j = lane - 1;  // lane 0 is the re-probe at ip itself
        }
        uint32_t pos, step;
        if (post && lane == 0) {
          pos = ip;
          step = 1;
        } else {
          const uint32_t q = s >> 5, r = s & 31u;
          uint32_t off = q * j;
          if (j > 32u - r) off += j - (32u - r);
          if (j > 64u - r) off += j - (64u - r);
          pos = base + off;
          step = (s + j) >> 5;
          valid = pos + step <= ip_limit;
        }
        if (!valid) pos = 1;  // keep LDS reads in range; the lane is ignored
        zh_wave_sync();
        const uint32_t v = zh_ld32(s_src, pos);
        const uint32_t h = (v * kHashMul) >> shift;
        const uint32_t old = s_table[h];
        const bool hit_old = valid && zh_ld32(s_src, old) == v;
        if (valid) atomicAdd(&s_scr[h & 127u], 1u);
        zh_wave_sync();
        const bool collide = valid && s_scr[h & 127u] > 1u;
        zh_wave_sync();
        if (valid) s_scr[h & 127u] = 0;
        const uint64_t V = __ballot(valid);
        const uint64_t H = __ballot(hit_old);
        const uint64_t C = __ballot(collide);
        const uint32_t t = V == ~0ull ? 64u : (uint32_t)__ffsll((long long)~V) - 1u;  // first probe past ip_limit
        const uint64_t clean_hits = H & ~C;
        const uint32_t g = clean_hits ? (uint32_t)__ffsll((long long)clean_hits) - 1u : 64u;
        const uint32_t bound = g < t ? g : t;
        uint32_t f = 64, cand = 0;  // first hit in probe order and its candidate
        {
          uint64_t cb = bound >= 64 ? C : (C & ((1ull << bound) - 1ull));
          while (cb) {  // colliding probes before the first clean event, in order
            const uint32_t jx = (uint32_t)__ffsll((long long)cb) - 1u;
            cb &= cb - 1;
            const uint32_t hj = __builtin_amdgcn_readlane(h, jx);
            const uint64_t same = __ballot(valid && h == hj) & ((1ull << jx) - 1ull);
            bool hitj;
            uint32_t candj;
            if (same) {  // an earlier probe of this step inserted this hash last
              const uint32_t i = 63u - (uint32_t)__clzll((long long)same);
              hitj = __builtin_amdgcn_readlane(v, i) == __builtin_amdgcn_readlane(v, jx);
              candj = __builtin_amdgcn_readlane(pos, i);
            } else {
              hitj = (H >> jx) & 1ull;
              candj = __builtin_amdgcn_readlane(old, jx);
            }
            if (hitj) {
              f = jx;
              cand = candj;
              break;
            }
          }
          if (f == 64 && g < t) {
            f = g;
            cand = __builtin_amdgcn_readlane(old, g);
          }
        }
        // table inserts of the probes that really happened: up to the hit, or up to the limit
        const uint32_t last_plus1 = f < 64 ? f + 1 : t;
        const bool commit = lane < last_plus1;
        if (commit && !collide) s_table[h] = (uint16_t)pos;
        {
          uint64_t cc = last_plus1 >= 64 ? C : (C & ((1ull << last_plus1) - 1ull));
          while (cc) {  // same-bucket probes write in probe order (the later one wins)
            const uint32_t jx = (uint32_t)__ffsll((long long)cc) - 1u;
            cc &= cc - 1;
            if (lane == jx) s_table[h] = (uint16_t)pos;
          }
        }
        if (f < 64) {
          const uint32_t mp = __builtin_amdgcn_readlane(pos, f);
          // 4 + determineMatchLength(cand + 4, mp + 4, limit), internal.nim:251-270:
          // lane l compares bytes 4+4l .. 7+4l (64 lanes cover the 258-byte maximum)
          const uint32_t limit = n < mp + 258u ? n : mp + 258u;
          const uint32_t o = 4u + 4u * lane;
          uint32_t avail = 0;
          if (mp + o < limit) avail = limit - (mp + o) < 4u ? limit - (mp + o) : 4u;
          const uint32_t x = zh_ld32(s_src, mp + o) ^ zh_ld32(s_src, cand + o);
          uint32_t eq = x ? ((uint32_t)__ffs((int)x) - 1u) >> 3 : 4u;
          if (eq > avail) eq = avail;
          const uint64_t stop = __ballot(eq < 4u);  // lane 63 always stops (limit <= mp + 258)
          const uint32_t fl = (uint32_t)__ffsll((long long)stop) - 1u;
          const uint32_t matched = 4u + 4u * fl + __builtin_amdgcn_readlane(eq, fl);
          if (lane == 0) {
            m_pos[nm] = (uint16_t)mp;
            m_len[nm] = (uint16_t)matched;
            m_off[nm] = (uint16_t)(mp - cand);
          }
          nm++;
          ip = mp + matched;
          if (ip >= ip_limit) break;  // snappy.nim:118-120
          post = true;
          K = 0;
          continue;
        }
        if (t < 64) break;  // snappy.nim:93-95: the rest of the fragment is literals
        // 64 probes without a hit: continue the same literal run
        const uint32_t p63 = __builtin_amdgcn_readlane(pos, 63), s63 = __builtin_amdgcn_readlane(step, 63);
        K = post ? 63 : K + 64;
        post = false;
        ip = p63 + s63;
      }
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
