// Token -> bit emission: one 64-lane wave per fragment.
//
// Replaces deflate.nim:403-466 (the literal/length/distance bit packing loop)
// and the data copy of addNoCompressionBlock (deflate.nim:200-205).  The
// reference packs serially through BitStreamWriter.addBits; here every source
// position of the fragment is an item (kPos = 8 per lane and pass: the loads of a pass -- the lane's source
// bytes and the match fields -- are in flight together, and what a pass costs whatever its width -- two wave
// scans, the match coding, the bitmap reads, the loop -- is paid once per 512 positions):
// a literal contributes its code, a
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
#include "zh_kprof.h"

namespace {
#ifndef ZH_EMIT_POS
#define ZH_EMIT_POS 8
#endif
constexpr uint32_t kPos = ZH_EMIT_POS;                 // positions per lane and pass (4 or 8: one bitmap word a lane)
constexpr uint32_t kPass = 64 * kPos;                  // positions per pass
constexpr uint32_t kPosMask = (1u << kPos) - 1u;
#ifndef ZH_EMIT_CHUNK
#define ZH_EMIT_CHUNK 2048  // (round 6, ms for 4096 x 1 MiB: 8192: 5.11, 4096: 5.27, 2048: 5.03, 1024: 5.26, 512: 5.82)
#endif
constexpr uint32_t kChunk = ZH_EMIT_CHUNK;             // positions per match-bitmap chunk
#ifndef ZH_EMIT_STAGE
#define ZH_EMIT_STAGE 512
#endif
constexpr uint32_t kStageWords = ZH_EMIT_STAGE;        // 2 KiB staging window
// flush threshold: a position adds at most 16 bits (a 15-bit literal, or 48 bits for a match of >= 3)
constexpr uint32_t kFlushBits = (kStageWords - kPass / 2 - 16) * 32;
constexpr uint32_t kPassMatches = kPass / 3 + 3;       // a pass starts at most kPass / 3 + 1 matches
constexpr uint32_t kAhead = kPos / 4;                  // match fields asked for a pass ahead, per lane
static_assert(kPos == 4 || kPos == 8, "one bitmap word and whole source dwords per lane");
}  // namespace

// kCoverIn: the matcher has left the fragment's coverage bitmap in a.f_cover (zh_l1_match_kernel: bit p = byte p is
// inside a match, behind its first byte), a chunk's 64 words are one load a lane -- asked for two chunks ahead -- and a
// match's first byte is the clear bit in front of a set one.  Otherwise (parallel parse, chain levels) the bitmaps are
// made here from the match list, chunk by chunk.  (Round 6, ms for 4096 x 1 MiB: 4.99 -> 4.38, the matcher's 64.8 unchanged;
// the parallel matcher leaving the same bitmap -- two LDS flips a match in its output walk, a workgroup-wide parity -- cost it
// 23.7 -> 24.7 ms for 5.2 -> 4.55 here: not adopted.  profiles/r06_ab_*, r06_ad_*)
template <bool kCoverIn>
__global__ __launch_bounds__(64) void zh_emit_kernel(const uint8_t* __restrict__ d_src,
                                                     uint8_t* __restrict__ d_dst, ZhCompressArgs a) {
  __shared__ uint32_t s_lit[288];
  __shared__ uint32_t s_dist[32];
  // match bitmaps of the current 4 KiB chunk of the fragment (whole-fragment bitmaps would cost
  // 8 KiB of LDS and leave 13 waves per CU instead of 32)
  __shared__ uint32_t s_start[kChunk / 32];  // bit p - c0: a match starts at p
  __shared__ uint32_t s_cover[kChunk / 32];  // bit p - c0: p is inside a match (not its start)
  __shared__ uint32_t s_stage[kStageWords + 4];
  // the pass's matches, coded once each by the first lanes
  __shared__ uint64_t s_mval[kPassMatches];
  __shared__ uint32_t s_mbits[kPassMatches];

  const unsigned lane = zh_lane();
  KPROF_DECL(8);  // 0 flush + loop, 1 chunk bitmaps, 2 bitmaps/source/scan/match fields, 3 codes, 4 scan + LDS ORs, 5 last flush, 6 waves
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
  for (uint32_t i = lane; i < kStageWords + 4; i += 64) s_stage[i] = 0;
  zh_wave_sync();

  const uint16_t* m_pos = a.m_pos + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  const uint16_t* m_len = a.m_len + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  const uint16_t* m_off = a.m_off + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  const uint32_t nmatch = a.f_nmatch[f];
  const uint32_t spill = a.f_spill[f];  // bytes covered by a match begun in the previous fragment
  uint32_t mnext = 0;                   // first match that starts at or behind the current chunk
  uint32_t prev_s = 0, prev_l = 0;      // the match before it (start, length): what reaches into the chunk from before
  // [p, e) is inside a match: noted as the two places where "inside" flips, clipped to the chunk [c0, c1), as bits
  // relative to c0 (matches do not overlap and a match covers at least two bytes behind its start: no two flips
  // share a bit); build_chunk turns the flips into the bitmap with a prefix parity
  auto cover = [&](uint32_t p, uint32_t e, uint32_t c0, uint32_t c1) {
    if (p < c0) p = c0;
    if (p >= e || p >= c1) return;
    atomicOr(&s_cover[(p - c0) >> 5], 1u << ((p - c0) & 31u));
    if (e < c1) atomicOr(&s_cover[(e - c0) >> 5], 1u << ((e - c0) & 31u));
  };
  // (kCoverIn) the lane's word of the current chunk's bitmap and of the next chunk's; a chunk that starts behind the
  // fragment reads as zero (the matcher writes whole chunks' worth of words)
  const uint32_t* fcov = kCoverIn ? a.f_cover + (size_t)f * (ZH_FRAG_SIZE / 32u) : nullptr;
  auto cover_words = [&](uint32_t c0) -> uint32_t { return c0 < n ? fcov[(c0 >> 5) + lane] : 0u; };
  uint32_t cov_a = 0, cov_b = 0;
  if (kCoverIn) {
    static_assert(!kCoverIn || kChunk == 2048u, "a bitmap word a lane and chunk");
    cov_a = cover_words(0);
    cov_b = cover_words(kChunk);
  }
  auto build_chunk = [&](uint32_t c0) {
    if (kCoverIn) {
      const uint32_t cw = cov_a;
      // the bit behind the word's last: the next lane's first, the next chunk's for lane 63
      const uint32_t nb0 = (uint32_t)__builtin_amdgcn_readlane((int)cov_b, 0);
      uint32_t nx = (uint32_t)__shfl_down((int)cw, 1, 64);
      if (lane == 63u) nx = nb0;
      zh_wave_sync();
      s_cover[lane] = cw;
      s_start[lane] = ~cw & ((cw >> 1) | (nx << 31));
      cov_a = cov_b;
      cov_b = cover_words(c0 + 2u * kChunk);
      zh_wave_sync();
      return;
    }
    const uint32_t c1 = c0 + kChunk < n ? c0 + kChunk : n;
    zh_wave_sync();
    for (uint32_t i = lane; i < kChunk / 32; i += 64) {
      s_start[i] = 0;
      s_cover[i] = 0;
    }
    zh_wave_sync();
    // what reaches in from before the chunk: the previous match (matches do not overlap), or
    // for the first chunk the match begun in the previous fragment
    // (kept from the chunk before: two dependent global loads by one lane at the head of every chunk otherwise)
    if (lane == 0) {
      if (mnext) cover(prev_s + 1u, prev_s + prev_l, c0, c1);
      else cover(0, spill, c0, c1);
    }
    // (the next 64 matches' fields are asked for before these are filed: unconditional loads at clamped
    // indices; what was asked for in vain when the chunk ends is asked for again by the next chunk)
    const uint32_t mlast = nmatch ? nmatch - 1u : 0u;
    auto at = [&](uint32_t m) { return m < nmatch ? m : mlast; };
    uint32_t sq = m_pos[at(mnext + lane)], lq = m_len[at(mnext + lane)];
    for (;;) {
      const uint32_t m = mnext + lane;
      const uint32_t s = m < nmatch ? sq : 0xffffffffu;
      const uint32_t l = lq;
      sq = m_pos[at(m + 64u)];
      lq = m_len[at(m + 64u)];
      const bool here = s < c1;
      if (here) {
        atomicOr(&s_start[(s - c0) >> 5], 1u << ((s - c0) & 31u));
        cover(s + 1u, s + l, c0, c1);  // (chain levels: a match may run past the fragment)
      }
      const uint32_t cnt = (uint32_t)__popcll(__ballot(here));
      if (cnt) {  // (the lanes that are `here` are a prefix: matches come in position order)
        prev_s = (uint32_t)__builtin_amdgcn_readlane((int)s, (int)(cnt - 1u));
        prev_l = (uint32_t)__builtin_amdgcn_readlane((int)l, (int)(cnt - 1u));
      }
      mnext += cnt;
      if (cnt < 64u) break;
    }
    zh_wave_sync();
    // flips -> "inside": bit i = parity of the flips at or before i; a lane the words `lane`, `lane + 64`, ...
    constexpr uint32_t kWords = kChunk / 32, kWordsLane = (kWords + 63u) / 64u;
    static_assert(kChunk % kPass == 0 && kChunk >= kPass, "whole passes a chunk");
    uint32_t carry_odd = 0;
#pragma unroll
    for (uint32_t k = 0; k < kWordsLane; k++) {
      const bool have = lane + 64u * k < kWords;
      uint32_t x = have ? s_cover[lane + 64u * k] : 0u;
#pragma unroll
      for (uint32_t sh = 1; sh < 32; sh <<= 1) x ^= x << sh;
      const uint64_t odd = __ballot((x >> 31) != 0u);
      if ((carry_odd + (uint32_t)__popcll(odd & zh_lanemask_lt())) & 1u) x = ~x;
      if (have) s_cover[lane + 64u * k] = x;
      carry_odd += (uint32_t)__popcll(odd);
    }
    zh_wave_sync();
  };

  // ---- emission ----
  const uint64_t bit0 = a.f_bit_start[f];              // absolute bit position in d_dst
  uint32_t* gwords = reinterpret_cast<uint32_t*>(d_dst) + (bit0 >> 5);  // d_dst is 4-byte aligned
  uint32_t stage_bits = (uint32_t)(bit0 & 31u);        // bits in use in the staging window
  bool first_word_pending = true;                       // gwords[0] may be shared with the previous writer
  uint32_t mbase = 0;                                   // matches before the current batch

  // the fragment's bytes: aligned dwords of the stream below `src`, never past the dword that
  // holds its last byte
  const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
  const uint32_t* asrc = reinterpret_cast<const uint32_t*>(src - mis);
  const uint32_t last_dw = (n + mis - 1u) >> 2;  // n >= 1 here
  auto dw = [&](uint32_t i) -> uint32_t { return asrc[i < last_dw ? i : last_dw]; };

  // a pass ahead: the lane's source bytes and the next 64 * kAhead matches' fields (clamped, unconditional loads)
  const uint32_t mlast = nmatch ? nmatch - 1u : 0u;
  uint32_t w_next[kPos / 4];
  auto words_at = [&](uint32_t p0) {
    const uint32_t q = p0 + mis;
    uint32_t d = dw(q >> 2);
#pragma unroll
    for (uint32_t i = 0; i < kPos / 4; i++) {
      const uint32_t e = dw((q >> 2) + i + 1u);
      w_next[i] = __builtin_amdgcn_alignbyte(e, d, q);
      d = e;
    }
  };
  words_at(kPos * lane);
  uint32_t pl[kAhead], po[kAhead];
  auto fields_at = [&](uint32_t m0) {
#pragma unroll
    for (uint32_t i = 0; i < kAhead; i++) {
      const uint32_t mi = m0 + 64u * i + lane < nmatch ? m0 + 64u * i + lane : mlast;
      pl[i] = m_len[mi];
      po[i] = m_off[mi];
    }
  };
  fields_at(0);
  for (uint32_t base = 0; base < n; base += kPass) {
    KPROF_MARK(0);
    if ((base & (kChunk - 1u)) == 0) build_chunk(base);
    KPROF_MARK(1);
    const uint32_t p0 = base + kPos * lane;  // this lane's positions p0 .. p0 + kPos - 1
    const bool in = p0 < n;
    const uint32_t bw = in ? (p0 & (kChunk - 1u)) >> 5 : 0u, bs = p0 & 31u;  // p0 is a multiple of kPos: one bitmap word
    const uint32_t sw = s_start[bw], cw = s_cover[bw];  // (unconditional: bw is 0 for a lane past the fragment)
    const uint32_t st4 = in ? (sw >> bs) & kPosMask : 0u;
    uint32_t skip4 = in ? (cw >> bs) & kPosMask : kPosMask;
    if (in && n - p0 < kPos) skip4 |= (kPosMask << (n - p0)) & kPosMask;  // positions past the fragment
    uint32_t w[kPos / 4];
#pragma unroll
    for (uint32_t i = 0; i < kPos / 4; i++) w[i] = in ? w_next[i] : 0u;
    words_at(p0 + kPass);  // (dw() clamps behind the fragment's last byte)
    // index (in the pass) of this lane's first match start
    const uint32_t nst = (uint32_t)__popc(st4);
    const uint32_t incl_st = zh_wave_scan(nst);
    uint32_t mj = incl_st - nst;
    const uint32_t npass = (uint32_t)__builtin_amdgcn_readlane(incl_st, 63);
    // ---- the pass's matches: length code + extra + distance code + extra (deflate.nim:417-433), one
    // match a lane (their fields are neighbours in the match list), parked for the lanes that own the
    // positions ----
#pragma unroll
    for (uint32_t i = 0; i < (kPassMatches + 63u) / 64u; i++) {
      const uint32_t j = lane + 64u * i;
      if (64u * i >= npass) break;
      if (j >= npass) continue;
      const uint32_t length = i < kAhead ? pl[i < kAhead ? i : 0] : m_len[mbase + j];
      const uint32_t offset = i < kAhead ? po[i < kAhead ? i : 0] : m_off[mbase + j];
      const uint32_t li = zh_len_code(length), di = zh_dist_code(offset);
      const uint32_t lc = s_lit[257 + li], dc = s_dist[di];
      uint64_t v = lc & 0xffffu;
      uint32_t nbits = lc >> 16;
      v |= (uint64_t)(length - zh_len_base(li)) << nbits;
      nbits += zh_len_extra_bits(li);
      v |= (uint64_t)(dc & 0xffffu) << nbits;
      nbits += dc >> 16;
      v |= (uint64_t)(offset - zh_dist_base(di)) << nbits;
      nbits += zh_dist_extra_bits(di);
      s_mval[j] = v;
      s_mbits[j] = nbits;
    }
    mbase += npass;
    fields_at(mbase);
    zh_wave_sync();
    KPROF_MARK(2);
    uint64_t val[kPos];
    uint32_t nb[kPos], lane_bits = 0;
    // the eight literal codes asked for together, whatever the positions turn out to be: under a position's own condition
    // the compiler waits for each look-up where it is issued -- eight LDS trips one behind the other (round 6: 5.60 -> 5.34
    // ms; the lane's matches hoisted the same way: nothing more; every position's ORs without a branch: 8.0 ms,
    // profiles/r06_n_*)
    uint32_t lcs[kPos];
#pragma unroll
    for (uint32_t k = 0; k < kPos; k++) lcs[k] = s_lit[(w[k / 4] >> (8u * (k & 3u))) & 255u];
#pragma unroll
    for (uint32_t k = 0; k < kPos; k++) {
      uint64_t v = 0;
      uint32_t nbits = 0;
      if ((st4 >> k) & 1u) {
        v = s_mval[mj];
        nbits = s_mbits[mj];
        mj++;
      } else if (!((skip4 >> k) & 1u)) {
        const uint32_t lc = lcs[k];
        v = lc & 0xffffu;
        nbits = lc >> 16;
      }
      val[k] = v;
      nb[k] = nbits;
      lane_bits += nbits;
    }

#ifdef ZH_KPROF
    asm volatile("" ::"v"(lane_bits));
#endif
    KPROF_MARK(3);
    const uint32_t incl = zh_wave_scan(lane_bits);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(incl, 63);
    uint32_t bp = stage_bits + incl - lane_bits;
#pragma unroll
    for (uint32_t k = 0; k < kPos; k++) {
      if (nb[k]) {
        const uint32_t wd = bp >> 5, sft = bp & 31u;
        const uint64_t lo64 = val[k] << sft;  // nbits <= 48, so only sft + nbits > 64 loses bits here
        atomicOr(&s_stage[wd], (uint32_t)lo64);
        if (sft + nb[k] > 32) atomicOr(&s_stage[wd + 1], (uint32_t)(lo64 >> 32));
        if (sft + nb[k] > 64) atomicOr(&s_stage[wd + 2], (uint32_t)(val[k] >> (64 - sft)));
        bp += nb[k];
      }
    }
    stage_bits += total;
#ifdef ZH_KPROF
    zh_wave_sync();
#endif
    KPROF_MARK(4);

    const bool last = base + kPass >= n;
    if (stage_bits >= kFlushBits || last) {
      zh_wave_sync();
      const uint32_t full = stage_bits >> 5;           // complete words
      const uint32_t rem = stage_bits & 31u;
      for (uint32_t wv = lane; wv < full; wv += 64) {
        const uint32_t v = s_stage[wv];
        if (wv == 0 && first_word_pending) atomicOr(&gwords[0], v);
        else gwords[wv] = v;
      }
      zh_wave_sync();
      if (last) {
        if (rem && lane == 0) atomicOr(&gwords[full], s_stage[full]);  // shared with the next writer
      } else {
        const uint32_t carry = s_stage[full];
        zh_wave_sync();
        for (uint32_t wv = lane; wv <= full + 2 && wv < kStageWords + 4; wv += 64) s_stage[wv] = 0;
        zh_wave_sync();
        if (lane == 0) s_stage[0] = carry;
        if (full) first_word_pending = false;
        gwords += full;
        stage_bits = rem;
        zh_wave_sync();
      }
    }
  }
  KPROF_MARK(5);
  KPROF_COUNT(6, 1);
  KPROF_FLUSH(32, 8);
}

// cover_in: a.f_cover holds this run's bitmaps (the matcher was zh_l1_match_kernel)
extern "C" void zh_launch_emit(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst,
                               ZhCompressArgs a, int cover_in) {
  if (!a.nfrags) return;
  if (cover_in && a.f_cover)
    hipLaunchKernelGGL(zh_emit_kernel<true>, dim3(a.nfrags), dim3(64), 0, stream, d_src, d_dst, a);
  else
    hipLaunchKernelGGL(zh_emit_kernel<false>, dim3(a.nfrags), dim3(64), 0, stream, d_src, d_dst, a);
}
