// Hash-chain LZ77 matcher for levels -1 (= 6) and 2..9: one wave per <= 4 MiB
// block (the reference resets the LZ history per block, lz77.nim:63-64, so the
// block is the independent unit; there is no 32 KiB fragment independence here).
//
// Replaces lz77.nim:10-130 encodeLz77, decision for decision: 17-bit hash of 4
// bytes into `head`, 32 K-entry `chain` of window positions, window position 0
// as the "empty" sentinel, bounded chain walk with the good/nice/chain
// parameters of internal.nim:177-189, the decreasing-offset wrap test, greedy
// acceptance of matches longer than 4, hash insertion for the bytes skipped by
// a match.  The walk state is wave-uniform; the 64 lanes split each candidate
// comparison (internal.nim:251-270 determineMatchLength) 4 bytes per lane.
// `chain` (64 KiB) lives in LDS, `head` (256 KiB > LDS) in an HBM scratch slice
// that stays L2-resident.
//
// zh_frag_stats_kernel then derives the per-fragment histograms from the match
// list (the addLiteral/addCopy bookkeeping of lz77.nim:19-50) for the matchers
// that do not fuse it.
#include "zh_common.h"
#include "zh_tables.h"

namespace {
__constant__ zh::LenTables c_len = zh::make_len_tables();
__constant__ zh::DistTables c_dist = zh::make_dist_tables();
constexpr uint32_t kHashMul = 0x1e35a7bdu;
constexpr uint32_t kHashBits = 17;

__device__ inline uint32_t load32(const uint8_t* p) {
  return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
}
}  // namespace

__global__ __launch_bounds__(64) void zh_chain_match_kernel(const uint8_t* __restrict__ d_src,
                                                            ZhCompressArgs a, int good, int nice,
                                                            int max_chain,
                                                            uint16_t* __restrict__ head_scratch) {
  __shared__ uint16_t s_chain[32768];
  const unsigned lane = zh_lane();
  const uint32_t b = blockIdx.x;
  const ZhBlockDesc bd = a.blocks[b];
  const uint8_t* src = d_src + bd.src_off;  // block-relative addressing below
  const uint32_t block_len = (uint32_t)bd.len;
  uint16_t* head = head_scratch + ((size_t)b << kHashBits);  // zeroed by the host before launch

  for (uint32_t i = lane; i < 32768; i += 64) s_chain[i] = 0;
  zh_wave_sync();

  uint32_t cur_frag = 0, frag_matches = 0;  // fragment currently receiving matches
  auto close_frags_until = [&](uint32_t frag) {  // publish counts of fragments [cur_frag, frag)
    while (cur_frag < frag) {
      if (lane == 0) a.f_nmatch[bd.first_frag + cur_frag] = frag_matches;
      frag_matches = 0;
      cur_frag++;
    }
  };
  if (lane == 0)
    for (uint32_t k = 0; k < bd.nfrag; k++) a.f_spill[bd.first_frag + k] = 0;

  uint32_t pos = 0;
  if (block_len > 4) {  // lz77.nim:54-56: blocks of <= 4 bytes are all literals
    while (pos < block_len) {
      if (pos + 4 >= block_len) break;  // lz77.nim:74-76: the tail is literals
      const uint32_t window_pos = pos & 32767u;
      const uint32_t hash = (load32(src + pos) * kHashMul) >> (32 - kHashBits);
      // updateChain (lz77.nim:69-71)
      uint32_t hash_pos = zh_bcast(head[hash]);
      zh_wave_sync();
      if (lane == 0) {
        s_chain[window_pos] = (uint16_t)hash_pos;
        head[hash] = (uint16_t)window_pos;
      }
      zh_wave_sync();

      const uint32_t limit = block_len < pos + 258u ? block_len : pos + 258u;
      int tries = max_chain;
      int prev_offset = 0, longest_offset = 0, longest_len = 0;
      while (tries > 0 && hash_pos != 0) {  // lz77.nim:88-112
        tries--;
        const int offset = hash_pos <= window_pos ? (int)(window_pos - hash_pos)
                                                  : (int)(window_pos - hash_pos + 32768u);
        if (offset <= 0 || offset < prev_offset) break;
        prev_offset = offset;
        // determineMatchLength(src, pos - offset, pos, limit), 4 bytes per lane
        int match_len = 0;
        bool stopped = false;
        for (uint32_t base = 0; pos + base < limit; base += 256) {
          const uint32_t o = base + lane * 4;
          uint32_t avail = 0;
          if (pos + o < limit) avail = limit - (pos + o) < 4 ? limit - (pos + o) : 4;
          uint32_t eq = 0;
          while (eq < avail && src[pos + o + eq] == src[pos - offset + o + eq]) eq++;
          const uint64_t stop = __ballot(eq < 4);  // mismatch, or this lane hit the limit
          if (stop) {
            const uint32_t fl = (uint32_t)__ffsll((long long)stop) - 1;
            match_len = (int)(base + fl * 4 + __shfl(eq, fl, 64));
            stopped = true;
            break;
          }
        }
        if (!stopped) match_len = (int)(limit - pos);
        if (match_len > longest_len) {
          if (match_len >= good) tries >>= 2;
          longest_len = match_len;
          longest_offset = offset;
        }
        const uint32_t nxt = zh_bcast(s_chain[hash_pos]);
        if (longest_len >= nice || hash_pos == nxt) break;
        hash_pos = nxt;
      }

      if (longest_len > 4) {  // lz77.nim:114
        const uint32_t frag = pos >> 15;
        close_frags_until(frag);
        if (lane == 0) {
          const size_t slot = (size_t)(bd.first_frag + frag) * ZH_MAX_MATCHES_PER_FRAG + frag_matches;
          a.m_pos[slot] = (uint16_t)(pos & 32767u);
          a.m_len[slot] = (uint16_t)longest_len;
          a.m_off[slot] = (uint16_t)longest_offset;
          const uint32_t end = pos + (uint32_t)longest_len;
          if ((end - 1) >> 15 != frag) a.f_spill[bd.first_frag + frag + 1] = end & 32767u;
        }
        frag_matches++;
        for (int i = 1; i < longest_len; i++) {  // lz77.nim:121-126
          pos++;
          if (pos + 4 < block_len) {
            const uint32_t wp = pos & 32767u;
            const uint32_t h = (load32(src + pos) * kHashMul) >> (32 - kHashBits);
            const uint32_t old = zh_bcast(head[h]);
            zh_wave_sync();
            if (lane == 0) {
              s_chain[wp] = (uint16_t)old;
              head[h] = (uint16_t)wp;
            }
            zh_wave_sync();
          }
        }
      }
      pos++;
    }
  }
  close_frags_until(bd.nfrag);
}

// Per-fragment statistics from a match list: litlen/distance histograms,
// literal count, sum of extra bits (lz77.nim:19-50 / snappy.nim:33-64).
__global__ __launch_bounds__(64) void zh_frag_stats_kernel(const uint8_t* __restrict__ d_src,
                                                           ZhCompressArgs a) {
  __shared__ uint32_t s_hist[ZH_HIST_STRIDE];
  __shared__ uint32_t s_cover[ZH_FRAG_SIZE / 32];
  const unsigned lane = zh_lane();
  const uint32_t f = blockIdx.x;
  const ZhFragDesc fd = a.frags[f];
  const uint32_t n = fd.len;
  const uint8_t* src = d_src + fd.src_off;
  for (uint32_t i = lane; i < ZH_HIST_STRIDE; i += 64) s_hist[i] = 0;
  for (uint32_t i = lane; i < ZH_FRAG_SIZE / 32; i += 64) s_cover[i] = 0;
  zh_wave_sync();
  const uint32_t nmatch = a.f_nmatch[f], spill = a.f_spill[f];
  const uint16_t* m_pos = a.m_pos + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  const uint16_t* m_len = a.m_len + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  const uint16_t* m_off = a.m_off + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  uint32_t extra_bits = 0;
  for (uint32_t m = lane; m < nmatch + 1; m += 64) {
    uint32_t p, e;
    if (m < nmatch) {
      p = m_pos[m];
      const uint32_t l = m_len[m], o = m_off[m];
      const uint32_t li = c_len.index_of[l - 3], di = zh_dist_code(o);
      atomicAdd(&s_hist[257 + li], 1u);
      atomicAdd(&s_hist[ZH_NUM_LITLEN + di], 1u);
      extra_bits += c_len.extra[li] + c_dist.extra[di];
      e = p + l;
      if (e > n) e = n;
    } else {
      p = 0;
      e = spill < n ? spill : n;
    }
    if (p < e) {
      for (uint32_t w = p >> 5; w <= (e - 1) >> 5; w++) {
        const uint32_t lo = w == (p >> 5) ? (p & 31u) : 0u;
        const uint32_t hi = w == ((e - 1) >> 5) ? ((e - 1) & 31u) : 31u;
        atomicOr(&s_cover[w], (0xffffffffu >> (31u - hi)) & (0xffffffffu << lo));
      }
    }
  }
  zh_wave_sync();
  uint32_t nlit = 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t p = base + lane;
    if (p < n && !((s_cover[p >> 5] >> (p & 31u)) & 1u)) {
      atomicAdd(&s_hist[src[p]], 1u);
      nlit++;
    }
  }
  extra_bits = zh_wave_sum(extra_bits);
  nlit = zh_wave_sum(nlit);
  zh_wave_sync();
  uint16_t* hist_out = a.f_hist + (size_t)f * ZH_HIST_STRIDE;
  for (uint32_t i = lane; i < ZH_HIST_STRIDE; i += 64) hist_out[i] = (uint16_t)s_hist[i];
  if (lane == 0) {
    a.f_nlit[f] = nlit;
    a.f_extra_bits[f] = extra_bits;
  }
}

extern "C" void zh_launch_chain_match(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a,
                                      int good, int nice, int max_chain, uint16_t* head_scratch) {
  if (!a.nblocks) return;
  hipLaunchKernelGGL(zh_chain_match_kernel, dim3(a.nblocks), dim3(64), 0, stream, d_src, a, good,
                     nice, max_chain, head_scratch);
}
extern "C" void zh_launch_frag_stats(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a) {
  if (!a.nfrags) return;
  hipLaunchKernelGGL(zh_frag_stats_kernel, dim3(a.nfrags), dim3(64), 0, stream, d_src, a);
}
