// Token -> bit emission: one 64-lane wave per fragment.
//
// Replaces deflate.nim:403-466 (the literal/length/distance bit packing loop)
// and the data copy of addNoCompressionBlock (deflate.nim:200-205).  The
// reference packs serially through BitStreamWriter.addBits; here every source
// position of the fragment is a lane-item: a literal contributes its code, a
// match start contributes code + length extra + distance code + distance extra
// (<= 48 bits, assembled exactly as deflate.nim:417-433), bytes inside a match
// contribute nothing.  A wave prefix sum of the bit lengths gives each item its
// bit offset; items are OR-ed into an LDS staging window that is written back to
// HBM as whole, coalesced 32-bit words.  Only the first and last word of a
// fragment can be shared with a neighbour and go out as atomic ORs.
// Algorithmic traffic: source bytes read once, compressed bytes written once
// (+ the match list produced by the matcher).
#include "zh_common.h"
#include "zh_tables.h"

namespace {
__constant__ zh::LenTables c_len = zh::make_len_tables();
__constant__ zh::DistTables c_dist = zh::make_dist_tables();
constexpr uint32_t kStageWords = 1024;                 // 4 KiB staging window
constexpr uint32_t kFlushBits = (kStageWords - 128) * 32;  // flush threshold
}  // namespace

__global__ __launch_bounds__(64) void zh_emit_kernel(const uint8_t* __restrict__ d_src,
                                                     uint8_t* __restrict__ d_dst, ZhCompressArgs a) {
  __shared__ uint32_t s_lit[288];
  __shared__ uint32_t s_dist[32];
  __shared__ uint32_t s_start[ZH_FRAG_SIZE / 32];  // bit p: a match starts at p
  __shared__ uint32_t s_cover[ZH_FRAG_SIZE / 32];  // bit p: p is inside a match (not its start)
  __shared__ uint32_t s_stage[kStageWords + 4];

  const unsigned lane = zh_lane();
  const uint32_t f = blockIdx.x;
  const ZhFragDesc fd = a.frags[f];
  const uint32_t n = fd.len;
  const uint32_t b = fd.block;
  const ZhBlockDesc blk = a.blocks[b];
  if (a.status[blk.buf] != ZH_OK) return;  // output slot too small: nothing may be written
  const uint8_t* src = d_src + fd.src_off;
  const uint32_t mode = a.b_mode[b];

  if (mode == ZH_MODE_STORED) {
    // deflate.nim:200-205: block byte o lands at d0 + o + 5 * (o / 65535)
    const uint64_t d0 = a.b_stored_d0[b];
    const uint64_t o0 = fd.src_off - blk.src_off;
    for (uint32_t i = lane * 4; i < n; i += 256) {
      const uint32_t cnt = n - i < 4 ? n - i : 4;
      for (uint32_t j = 0; j < cnt; j++) {
        const uint64_t o = o0 + i + j;
        d_dst[d0 + o + 5 * (o / ZH_STORED_MAX)] = src[i + j];
      }
    }
    return;
  }

  // ---- code tables and match bitmaps into LDS ----
  for (uint32_t i = lane; i < 288; i += 64) s_lit[i] = a.b_litcode[(size_t)b * 288 + i];
  if (lane < 32) s_dist[lane] = a.b_distcode[(size_t)b * 32 + lane];
  for (uint32_t i = lane; i < ZH_FRAG_SIZE / 32; i += 64) {
    s_start[i] = 0;
    s_cover[i] = 0;
  }
  for (uint32_t i = lane; i < kStageWords + 4; i += 64) s_stage[i] = 0;
  zh_wave_sync();

  const uint16_t* m_pos = a.m_pos + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  const uint16_t* m_len = a.m_len + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  const uint16_t* m_off = a.m_off + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  const uint32_t nmatch = a.f_nmatch[f];
  const uint32_t spill = a.f_spill[f];  // bytes covered by a match begun in the previous fragment
  for (uint32_t m = lane; m < nmatch + 1; m += 64) {
    uint32_t p, e;  // cover [p, e)
    if (m < nmatch) {
      const uint32_t s = m_pos[m];
      atomicOr(&s_start[s >> 5], 1u << (s & 31u));
      p = s + 1;
      e = s + m_len[m];
      if (e > n) e = n;  // chain levels: a match may run into the next fragment
    } else {
      p = 0;
      e = spill;
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

  // ---- emission ----
  const uint64_t bit0 = a.f_bit_start[f];              // absolute bit position in d_dst
  uint32_t* gwords = reinterpret_cast<uint32_t*>(d_dst) + (bit0 >> 5);  // d_dst is 4-byte aligned
  uint32_t stage_bits = (uint32_t)(bit0 & 31u);        // bits in use in the staging window
  bool first_word_pending = true;                       // gwords[0] may be shared with the previous writer
  uint32_t mbase = 0;                                   // matches before the current batch

  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t p = base + lane;
    const bool in = p < n;
    const bool is_start = in && ((s_start[p >> 5] >> (p & 31u)) & 1u);
    const bool is_cov = in && ((s_cover[p >> 5] >> (p & 31u)) & 1u);
    const uint64_t start_mask = __ballot(is_start);
    uint64_t val = 0;
    uint32_t nbits = 0;
    if (is_start) {
      const uint32_t m = mbase + (uint32_t)__popcll(start_mask & zh_lanemask_lt());
      const uint32_t length = m_len[m], offset = m_off[m];
      const uint32_t li = c_len.index_of[length - 3], di = zh_dist_code(offset);
      const uint32_t lc = s_lit[257 + li], dc = s_dist[di];
      // deflate.nim:417-433
      val = lc & 0xffffu;
      nbits = lc >> 16;
      val |= (uint64_t)(length - c_len.base[li]) << nbits;
      nbits += c_len.extra[li];
      val |= (uint64_t)(dc & 0xffffu) << nbits;
      nbits += dc >> 16;
      val |= (uint64_t)(offset - c_dist.base[di]) << nbits;
      nbits += c_dist.extra[di];
    } else if (in && !is_cov) {
      const uint32_t lc = s_lit[src[p]];
      val = lc & 0xffffu;
      nbits = lc >> 16;
    }
    mbase += (uint32_t)__popcll(start_mask);

    const uint32_t incl = zh_wave_scan(nbits);
    const uint32_t total = __shfl(incl, 63, 64);
    if (nbits) {
      const uint32_t bp = stage_bits + incl - nbits;
      const uint32_t w = bp >> 5, s = bp & 31u;
      const uint64_t lo64 = val << s;  // nbits <= 48, so only s + nbits > 64 loses bits here
      atomicOr(&s_stage[w], (uint32_t)lo64);
      if (s + nbits > 32) atomicOr(&s_stage[w + 1], (uint32_t)(lo64 >> 32));
      if (s + nbits > 64) atomicOr(&s_stage[w + 2], (uint32_t)(val >> (64 - s)));
    }
    stage_bits += total;

    const bool last = base + 64 >= n;
    if (stage_bits >= kFlushBits || last) {
      zh_wave_sync();
      const uint32_t full = stage_bits >> 5;           // complete words
      const uint32_t rem = stage_bits & 31u;
      for (uint32_t w = lane; w < full; w += 64) {
        const uint32_t v = s_stage[w];
        if (w == 0 && first_word_pending) atomicOr(&gwords[0], v);
        else gwords[w] = v;
      }
      zh_wave_sync();
      if (last) {
        if (rem && lane == 0) atomicOr(&gwords[full], s_stage[full]);  // shared with the next writer
      } else {
        const uint32_t carry = s_stage[full];
        zh_wave_sync();
        for (uint32_t w = lane; w <= full + 2 && w < kStageWords + 4; w += 64) s_stage[w] = 0;
        zh_wave_sync();
        if (lane == 0) s_stage[0] = carry;
        if (full) first_word_pending = false;
        gwords += full;
        stage_bits = rem;
        zh_wave_sync();
      }
    }
  }
}

extern "C" void zh_launch_emit(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst,
                               ZhCompressArgs a) {
  if (!a.nfrags) return;
  hipLaunchKernelGGL(zh_emit_kernel, dim3(a.nfrags), dim3(64), 0, stream, d_src, d_dst, a);
}
