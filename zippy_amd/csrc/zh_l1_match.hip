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
#include "zh_common.h"
#include "zh_tables.h"

namespace {
__constant__ zh::LenTables c_len = zh::make_len_tables();
__constant__ zh::DistTables c_dist = zh::make_dist_tables();
constexpr uint32_t kHashMul = 0x1e35a7bdu;  // snappy.nim:70-71
}  // namespace

__global__ __launch_bounds__(64) void zh_l1_match_kernel(const uint8_t* __restrict__ d_src,
                                                         ZhCompressArgs a, int huffman_only) {
  __shared__ __attribute__((aligned(16))) uint32_t s_src[ZH_FRAG_SIZE / 4 + 8];
  __shared__ uint16_t s_table[16384];
  __shared__ uint32_t s_hist[ZH_HIST_STRIDE];
  __shared__ uint32_t s_cover[ZH_FRAG_SIZE / 32];  // bit p set: byte p lies inside a match
  __shared__ uint32_t s_nmatch;

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
  }

  uint32_t table_size = 256, shift = 24;  // snappy.nim:24-29
  while (table_size < 16384u && table_size < n) {
    table_size <<= 1;
    shift--;
  }
  if (!huffman_only)
    for (uint32_t i = lane; i < table_size / 2; i += 64) reinterpret_cast<uint32_t*>(s_table)[i] = 0;
  zh_wave_sync();

  // ---- greedy parse (snappy.nim:76-136), one lane walks the fragment ----
  if (lane == 0) {
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
  hipLaunchKernelGGL(zh_l1_match_kernel, dim3(a.nfrags), dim3(64), 0, stream, d_src, a,
                     huffman_only);
}
