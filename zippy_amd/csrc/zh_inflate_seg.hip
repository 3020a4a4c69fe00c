// One LARGE deflate stream on many workgroups: the kernels around the segment forms of
// zh_inflate_tokens_kernel / zh_inflate_write_kernel (zh_inflate_split.hip).
//
// inflate.nim:268-289 walks a stream block by block; where a block starts is only known once the
// one before it has been decoded, which ties a stream to one workgroup however large it is (a
// tar.gz, a 128 MiB zlib stream).  This file removes that chain the way parallel gzip readers do:
//
//   zh_seg_find_kernel     the compressed bytes are cut into segments of equal length, and every
//   zh_seg_check_kernel    bit position is asked whether it reads as the header of a dynamic-
//                          Huffman block (inflate.nim:115-171).  The find kernel takes the cheap
//                          questions, a thread per 64 positions: BFINAL = 0 (either value in the last
//                          4.5 MiB of the stream, where its last block starts), BTYPE = 2, HLIT and
//                          HDIST in range (all 64 at once, bitwise), then a complete code-length
//                          code (one position in 4 500 of random bits survives: they are queued);
//                          the check kernel, a thread per survivor, decodes the code lengths: exactly
//                          HLIT + HDIST of them, a complete literal/length code with an end-of-block
//                          symbol, a usable distance code.  The lowest position of a segment that
//                          passes is the segment's start: almost certainly a block start, but only
//                          a GUESS;
//   tokens kernel phase 0   sub-starts: a segment inside a long block (this library's own streams are
//   zh_seg_decide_kernel    ONE block per 4 MiB of input) gets a token boundary of that block as its
//                          start -- guessed by decoding towards the segment from 4096 bits before it
//                          with the tables of the nearest header found before it (Huffman decoding
//                          falls in step by itself) -- together with that header's position; the
//                          decide kernel merges them and sends streams with fewer than four starts
//                          to the ordinary kernels (zh_inflate_split.hip has the details);
//   tokens (segment form)  all segments at once, each from its found start.  A decoder stops at the
//                          first block boundary that IS a found start (or at the end of the stream):
//                          a found start it runs past was a wrong guess -- bits inside a stored block
//                          that read like a header, which a compressed file inside an archive is
//                          full of -- and is ignored;
//   zh_seg_chain_kernel    the proof: the stream's first segment starts at the stream's first block,
//                          which is exact; the next link is the segment whose found start is exactly
//                          where the last link's decoder stopped, so by induction every start on
//                          the chain is a real block start and the concatenated tokens are the serial
//                          decoder's.  Prefix-sums the output bytes.  Anything else -- an error in a
//                          chain segment, a token region that overflowed, a chain that does not
//                          reach the last block -- clears the stream's flag and the ordinary
//                          one-workgroup kernels decode it (they are launched right behind and return
//                          at once for flagged streams).  A chain that needs more room than the slot
//                          has is ZH_ERR_DST_TOO_SMALL at once; a sizing pass ends here;
//   writer (segment form)  every chain segment's bytes as 16-bit symbols: a copy that reaches into
//                          the 32 KiB before the segment yields "window byte k" instead of a value;
//   zh_seg_windows_*       what those windows are: see the kernels;
//   zh_seg_finish_kernel   symbols -> bytes in the caller's slot, all segments at once.
//
// Accept/reject, status and bytes are those of the ordinary kernels (same decoder, same checks; the
// tests run both against the oracle, damaged streams included).  Extra traffic: 4 bytes per token
// and 2 + 2 bytes per output byte for the symbols.
#include <cstdlib>
#include <cstring>

#include "zh_common.h"
#include "zh_kprof.h"
#include "zh_tables.h"
#include "zh_inflate_tables.h"

namespace {

constexpr uint32_t kFindThreads = 256;
constexpr uint32_t kFindBatch = 65536;  // bit positions a workgroup of the search takes
constexpr uint32_t kFindSlots = kSegFindSlots;  // candidates it can hand on (measured on 1 GiB streams of this library: 32-39 a batch on
                                        // average, 700 at most; with 64 slots a batch in six of a stream of literals dropped some --
                                        // among them block starts that every segment of a 4 MiB block depends on)
constexpr uint32_t kWin = 32768;
constexpr uint32_t kSegCandStored = 0x80000000u;  // a queued candidate that reads like a stored block's header

// A batch of the search: 65536 bit positions and the 74 bits of header behind the last of them,
// staged in LDS as dwords.
constexpr uint32_t kFindStage = kFindBatch / 32u + 8u;  // (a thread reads five dwords from its 64 positions' first)

// the 32 bits at bit `rel` of the staged bytes
__device__ __forceinline__ uint32_t seg_peek(const uint32_t* s_buf, uint32_t rel) {
  const uint32_t i = rel >> 5;
  return zh_alignbit(s_buf[i + 1u], s_buf[i], rel);
}
// the 32 bits at bit p of the stream itself (bits behind its end read as zero)
__device__ __forceinline__ uint32_t seg_peek_stream(const uint8_t* src, uint64_t len, uint64_t p) {
  const uint64_t b = p >> 3;
  uint64_t v = 0;
  if (b + 8 <= len) {
    struct __attribute__((packed)) U64 { uint64_t v; };
    v = reinterpret_cast<const U64*>(src + b)->v;
  } else {
    for (uint32_t i = 0; i < 8; i++)
      if (b + i < len) v |= (uint64_t)src[b + i] << (8 * i);
  }
  return (uint32_t)(v >> ((uint32_t)p & 7u));
}

// The code-length code of a dynamic header at staged bit p (whose first 13 bits have passed): 3
// bits per entry, HCLEN + 4 entries; it must be complete.  Branch-free: most candidates end here.
__device__ __forceinline__ bool seg_precode_complete(const uint32_t* s_buf, uint32_t p) {
  const uint32_t hclen = ((seg_peek(s_buf, p + 13u) & 15u) + 4u) * 3u;  // bits
  uint32_t w0 = seg_peek(s_buf, p + 17u), w1 = seg_peek(s_buf, p + 47u);  // entries 0-9, 10-18
  w0 &= hclen >= 30u ? 0x3fffffffu : (1u << hclen) - 1u;
  w1 &= hclen > 30u ? (1u << (hclen - 30u)) - 1u : 0u;
  uint32_t kraft = 0;
#pragma unroll
  for (uint32_t i = 0; i < 10; i++) {
    const uint32_t v = (w0 >> (3u * i)) & 7u;
    kraft += (128u >> v) & (v ? 0xffu : 0u);
  }
#pragma unroll
  for (uint32_t i = 0; i < 9; i++) {
    const uint32_t v = (w1 >> (3u * i)) & 7u;
    kraft += (128u >> v) & (v ? 0xffu : 0u);
  }
  return kraft == 128u;
}

// The rest of a dynamic header at bit p of the stream whose code-length code is complete.  The
// bits come through a 128-bit window; the load of the next 64 is in flight while these are used.
__device__ bool seg_lengths_ok(const uint8_t* src, uint64_t len, uint64_t p) {
  auto load64 = [&](uint64_t b) -> uint64_t {  // stream bytes b .. b + 7 (zeros behind the end)
    uint64_t v = 0;
    if (b + 8 <= len) {
      struct __attribute__((packed)) U64 { uint64_t v; };
      v = reinterpret_cast<const U64*>(src + b)->v;
    } else {
      for (uint32_t i = 0; i < 8; i++)
        if (b + i < len) v |= (uint64_t)src[b + i] << (8 * i);
    }
    return v;
  };
  uint64_t nb = p >> 3;  // next byte to load
  uint64_t lo = load64(nb), hi = load64(nb + 8);
  nb += 16;
  uint32_t used = (uint32_t)p & 7u;  // bits of `lo` already taken
  uint64_t q = p;                    // stream position of the window's next bit
  auto peek = [&]() -> uint32_t { return (uint32_t)(used ? (lo >> used) | (hi << (64u - used)) : lo); };
  auto take = [&](uint32_t n) {
    used += n;
    q += n;
    if (used >= 64u) {
      lo = hi;
      hi = load64(nb);
      nb += 8;
      used -= 64u;
    }
  };
  const uint32_t h = peek();
  const uint32_t hlit = ((h >> 3) & 31u) + 257u, hdist = ((h >> 8) & 31u) + 1u, hclen = ((h >> 13) & 15u) + 4u;
  take(17);
  uint64_t cl = 0;  // 3 bits per symbol 0..18
  for (uint32_t i = 0; i < hclen; i++) {
    cl |= (uint64_t)(peek() & 7u) << (3u * c_clcl_order[i]);
    take(3);
  }
  // canonical code (inflate.nim:29-65 in miniature): counts per length (4 bits each), first slot
  // per length, symbols in code order (5 bits each)
  uint32_t counts = 0;
  for (uint32_t s = 0; s < 19; s++) counts += 1u << (4u * ((uint32_t)(cl >> (3u * s)) & 7u));  // (a count is < 16: two lengths at least)
  uint64_t offs = 0;  // 8 bits per length
  {
    uint32_t run = 0;
    for (uint32_t l = 1; l <= 7; l++) {
      offs |= (uint64_t)run << (8u * l);
      run += (counts >> (4u * l)) & 15u;
    }
  }
  uint64_t sorted_lo = 0, sorted_hi = 0;  // 12 slots a word
  for (uint32_t s = 0; s < 19; s++) {
    const uint32_t l = (uint32_t)(cl >> (3u * s)) & 7u;
    if (!l) continue;
    const uint32_t at = (uint32_t)(offs >> (8u * l)) & 255u;
    offs += 1ull << (8u * l);
    if (at < 12u) sorted_lo |= (uint64_t)s << (5u * at);
    else sorted_hi |= (uint64_t)s << (5u * (at - 12u));
  }
  // the HLIT + HDIST code lengths (inflate.nim:131-165); both codes are checked as they come
  const uint32_t total = hlit + hdist;
  uint32_t i = 0, prev = 0, lit_kraft = 0, dist_kraft = 0, dist_used = 0;
  bool eob = false;
  while (i < total) {
    uint32_t w = peek();
    uint32_t code = 0, first = 0, index = 0, sym = 0xffu, nbits = 0;
    for (uint32_t l = 1; l <= 7; l++) {
      code |= w & 1u;
      w >>= 1;
      const uint32_t c = (counts >> (4u * l)) & 15u;
      if (code < first + c) {  // (code >= first always holds: the code is complete)
        const uint32_t at = index + (code - first);
        sym = at < 12u ? (uint32_t)(sorted_lo >> (5u * at)) & 31u : (uint32_t)(sorted_hi >> (5u * (at - 12u))) & 31u;
        nbits = l;
        break;
      }
      index += c;
      first = (first + c) << 1;
      code <<= 1;
    }
    if (sym == 0xffu) return false;
    uint32_t rep = 1, val = sym;
    if (sym == 16) {
      if (i == 0) return false;
      rep = (w & 3u) + 3u;
      nbits += 2;
      val = prev;
    } else if (sym == 17) {
      rep = (w & 7u) + 3u;
      nbits += 3;
      val = 0;
    } else if (sym == 18) {
      rep = (w & 127u) + 11u;
      nbits += 7;
      val = 0;
    }
    take(nbits);
    if (i + rep > total) return false;
    if (val) {
      for (uint32_t k = 0; k < rep; k++) {
        const uint32_t at = i + k;
        if (at < hlit) {
          lit_kraft += 32768u >> val;
          if (at == 256u) eob = true;
        } else {
          dist_kraft += 32768u >> val;
          dist_used++;
        }
      }
      if (lit_kraft > 32768u || dist_kraft > 32768u) return false;
    }
    prev = val;
    i += rep;
  }
  if (q > len * 8) return false;
  if (!eob || lit_kraft != 32768u) return false;
  return dist_kraft == 32768u || dist_used <= 1u;
}

}  // namespace

// A workgroup per batch of 65536 bit positions of a segment's search range (g.find_seg /
// g.find_batch name it): staged once, then a thread takes 64 consecutive positions at a time.
__global__ __launch_bounds__(kFindThreads) void zh_seg_find_kernel(const uint8_t* __restrict__ d_src, ZhInflateArgs a,
                                                                   ZhSegArgs g) {
  __shared__ uint32_t s_buf[kFindStage];
  __shared__ uint32_t s_cand[kFindSlots], s_ncand;
  const uint32_t sid = g.find_seg[blockIdx.x], batch = g.find_batch[blockIdx.x] & 0x7fffffffu, tid = threadIdx.x;
  const bool tail = (g.find_batch[blockIdx.x] >> 31) != 0u;  // the stream's last block may start in this batch
  if (tid == 0) g.cand_n[blockIdx.x] = 0;
  const uint32_t bid = g.parent[sid];
  const bool first = sid == g.first_seg[bid];
  const bool live = a.status[bid] == ZH_OK;
  if (batch == 0 && tid == 0) {  // the stream's first block: behind the container header, exact
    g.start_bit[sid] = first && live ? (uint64_t)a.body_pos[bid] * 8 : kSegNone;
    g.start2_bit[sid] = kSegNone;
    g.stored_bit[sid] = kSegNone;
  }
  if (first || !live) return;
  const ZhBufDesc bd = a.bufs[bid];
  const uint8_t* src = d_src + bd.src_off;
  const uint64_t len = a.src_len_dev ? a.src_len_dev[bid] : bd.src_len;
  const uint64_t base = g.nominal_bit[sid] + (uint64_t)batch * kFindBatch;
  uint64_t hi = g.nominal_bit[sid] + g.search_bits[sid];
  if (hi > len * 8) hi = len * 8;
  if (base >= hi) return;
  KPROF_DECL(8);  // cycles: 0 staging, 1 first 13 bits, 2 code-length codes; counts: 5 candidates, 6 queued, 7 waves
  KPROF_COUNT(7, 1);
  // stage the batch: dword i = stream bytes sb + 4 i .. (zeros behind the end)
  const uint64_t sb = (base >> 3) & ~(uint64_t)3;
  constexpr uint32_t kLoads = (kFindStage + kFindThreads - 1u) / kFindThreads;
  if (sb + 4ull * kLoads * kFindThreads <= len) {  // all of it inside the stream: the loads go out together
    struct __attribute__((packed)) U32 { uint32_t v; };
    uint32_t v[kLoads];
#pragma unroll
    for (uint32_t k = 0; k < kLoads; k++) v[k] = reinterpret_cast<const U32*>(src + sb + 4ull * (tid + k * kFindThreads))->v;
#pragma unroll
    for (uint32_t k = 0; k < kLoads; k++)
      if (tid + k * kFindThreads < kFindStage) s_buf[tid + k * kFindThreads] = v[k];
  } else {
    for (uint32_t i = tid; i < kFindStage; i += kFindThreads) {
      const uint64_t b = sb + 4ull * i;
      uint32_t v = 0;
      for (uint32_t k = 0; k < 4; k++)
        if (b + k < len) v |= (uint32_t)src[b + k] << (8u * k);
      s_buf[i] = v;
    }
  }
  if (tid == 0) s_ncand = 0;
  __syncthreads();
  KPROF_MARK(0);
  const uint32_t rel0 = (uint32_t)(base - sb * 8);  // < 32
  const uint64_t left = hi - base;  // positions of this batch inside the search range
  for (uint32_t part = 0; part < kFindBatch / (kFindThreads * 64u); part++) {
    // The header's first 13 bits, 64 positions at once: BFINAL = 0 (unless the stream's last block
    // may start here), BTYPE = 2 (bits 1-2 = 0, 1), HLIT < 30 (not all of bits 4-7 set),
    // HDIST < 30 (not all of bits 9-12 set).
    const uint32_t mine = (part * kFindThreads + tid) * 64u;
    if (mine >= left) break;
    const uint32_t r = rel0 + mine, bi = r >> 5;
    const uint32_t d0 = s_buf[bi], d1 = s_buf[bi + 1u], d2 = s_buf[bi + 2u], d3 = s_buf[bi + 3u], d4 = s_buf[bi + 4u];
    const uint32_t e0 = zh_alignbit(d1, d0, r), e1 = zh_alignbit(d2, d1, r), e2 = zh_alignbit(d3, d2, r);
    auto x = [&](uint32_t k) -> uint64_t {  // bit j = stream bit r + j + k
      return (uint64_t)zh_alignbit(e1, e0, k) | ((uint64_t)zh_alignbit(e2, e1, k) << 32);
    };
    uint64_t m = ~x(1) & x(2) & ~(x(4) & x(5) & x(6) & x(7)) & ~(x(9) & x(10) & x(11) & x(12));
    if (!tail) m &= ~x(0);  // BFINAL = 0, except near the end of the stream
    if (left - mine < 64u) m &= (1ull << (left - mine)) - 1ull;
    KPROF_MARK(1);
    while (m) {
      const uint32_t j = (uint32_t)__ffsll((long long)m) - 1u;
      m &= m - 1ull;
      const uint32_t off = mine + j;
      KPROF_COUNT(5, 1);
      if (seg_precode_complete(s_buf, rel0 + off)) {
        KPROF_COUNT(6, 1);
        const uint32_t at = atomicAdd(&s_ncand, 1u);
        if (at < kFindSlots) s_cand[at] = off;  // (a full queue drops candidates: a later block start will do)
      }
    }
    // Stored blocks (inflate.nim:252-266; incompressible stretches are chains of them, 16 K a GiB): a header at a byte
    // boundary -- BFINAL = 0, BTYPE = 0 in a byte's low bits, then LEN and its complement -- is a candidate too (the
    // check kernel wants another one behind its bytes).  The eight byte boundaries among the 64 positions out of the
    // 104 bits already in registers.
    {
      const uint32_t skip = (8u - (r & 7u)) & 7u;  // bits to the first byte boundary
      const uint32_t e3 = zh_alignbit(d4, d3, r);
      const uint32_t y[4] = {zh_alignbit(e1, e0, skip), zh_alignbit(e2, e1, skip), zh_alignbit(e3, e2, skip), e3 >> skip};
      auto byte_at = [&](uint32_t k) -> uint32_t { return (y[k >> 2] >> (8u * (k & 3u))) & 255u; };
#pragma unroll
      for (uint32_t t = 0; t < 8; t++) {
        const uint32_t lenv = byte_at(t + 1u) | (byte_at(t + 2u) << 8), nlen = byte_at(t + 3u) | (byte_at(t + 4u) << 8);
        if ((byte_at(t) & 7u) == 0u && (lenv ^ nlen) == 0xffffu && mine + skip + 8u * t < left) {
          const uint32_t at = atomicAdd(&s_ncand, 1u);
          if (at < kFindSlots) s_cand[at] = (mine + skip + 8u * t) | kSegCandStored;
        }
      }
    }
    KPROF_MARK(2);
  }
  __syncthreads();
  // the survivors go to this workgroup's slots of the queue (one position in 4 500 of random bits:
  // fifteen a batch)
  const uint32_t nc = s_ncand < kFindSlots ? s_ncand : kFindSlots;
  for (uint32_t k = tid; k < nc; k += kFindThreads) g.cand_off[(size_t)blockIdx.x * kFindSlots + k] = s_cand[k];
  if (tid == 0) g.cand_n[blockIdx.x] = s_ncand;  // (all of them: who reads the queue clamps; ZH_TRACE_SEG counts the overflows)
  KPROF_FLUSH(56, 8);
}

// where code-length symbol s sits in a header (the inverse of c_clcl_order)
__constant__ uint8_t c_clcl_place[19] = {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2};

// The code lengths of a candidate header, by a whole wave (the serial form above, seg_lengths_ok,
// is what it has to agree with: ZH_SEG_CHECK=serial runs that one).  The code-length code goes into
// a 128-entry table in LDS; then the 64 lanes decode the symbols that start at 64 consecutive bit
// positions, a wave-uniform walk (one v_readlane a symbol) picks the ones really in the sequence,
// and repeat counts, "previous length" and the two codes' Kraft sums are prefix sums and reductions.
// The header (at most 17 + 57 + 316 * 14 bits) is staged in LDS first: one trip to memory a candidate.
constexpr uint32_t kHdrWords = 144;  // 4608 bits
__device__ bool seg_lengths_ok_wave(const uint8_t* src, uint64_t len, uint64_t p, uint8_t* s_lut, uint32_t* s_hdr) {
  const unsigned lane = zh_lane();
  const uint64_t hb = p >> 3;  // first staged byte
  zh_wave_sync();              // (the candidate before is done with the staged bytes)
  for (uint32_t i = lane; i < kHdrWords + 2u; i += 64u) {
    const uint64_t b = hb + 4ull * i;
    uint32_t v = 0;
    if (b + 4 <= len) {
      struct __attribute__((packed)) U32 { uint32_t v; };
      v = reinterpret_cast<const U32*>(src + b)->v;
    } else {
      for (uint32_t k = 0; k < 4; k++)
        if (b + k < len) v |= (uint32_t)src[b + k] << (8 * k);
    }
    s_hdr[i] = v;
  }
  zh_wave_sync();
  auto peek = [&](uint64_t at) -> uint32_t {  // the 32 bits at stream bit `at` (inside the staged header)
    const uint32_t rel = (uint32_t)(at - hb * 8), i = rel >> 5;
    return zh_alignbit(s_hdr[i + 1u], s_hdr[i], rel);
  };
  const uint32_t h = zh_bcast(peek(p));
  const uint32_t hlit = ((h >> 3) & 31u) + 257u, hdist = ((h >> 8) & 31u) + 1u, hclen = ((h >> 13) & 15u) + 4u;
  // lane s < 19: the length of code-length symbol s (its place in the header: the inverse of c_clcl_order)
  const uint32_t place = lane < 19u ? c_clcl_place[lane] : 99u;
  const uint32_t l = place < hclen ? peek(p + 17u + 3u * place) & 7u : 0u;
  // canonical codes (inflate.nim:29-65): symbols of one length in symbol order
  uint32_t code = 0, next = 0;
#pragma unroll
  for (uint32_t k = 1; k <= 7; k++) {
    const uint64_t m = __ballot(l == k);
    if (l == k) code = next + (uint32_t)__popcll(m & zh_lanemask_lt());
    next = (next + (uint32_t)__popcll(m)) << 1;
  }
  zh_wave_sync();  // (the table of the candidate before has been read)
  if (l) {
    const uint32_t rev = __brev(code) >> (32u - l);  // the stream carries codes first bit first
    for (uint32_t e = rev; e < 128u; e += 1u << l) s_lut[e] = (uint8_t)(lane | (l << 5));
  }
  zh_wave_sync();
  uint64_t q = p + 17u + 3u * hclen;
  const uint32_t total = hlit + hdist;
  uint32_t i = 0, prev = 0, lit_kraft = 0, dist_kraft = 0, dist_used = 0;
  bool eob = false;
  while (i < total) {
    // the symbol that starts at bit q + lane
    const uint32_t w = peek(q + lane);
    const uint32_t e = s_lut[w & 127u], sym = e & 31u, cl = e >> 5;
    const uint32_t x = w >> cl;
    uint32_t rep = 1, val = sym, nb = cl;
    if (sym == 16u) {
      rep = (x & 3u) + 3u;
      nb += 2u;
    } else if (sym == 17u) {
      rep = (x & 7u) + 3u;
      nb += 3u;
      val = 0;
    } else if (sym == 18u) {
      rep = (x & 127u) + 11u;
      nb += 7u;
      val = 0;
    }
    if (__ballot(cl == 0u)) return false;  // (cannot happen with a complete code: keeps the walk below moving whatever the bits)
    // the walk: which of the 64 positions start a symbol of the sequence
    uint64_t on = 0;
    uint32_t pos = 0;
    while (pos < 64u) {
      on |= 1ull << pos;
      pos += (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)pos);
    }
    bool mine = (on >> lane) & 1ull;
    // entries before mine in this window; symbols past the end of the sequence do not count
    const uint32_t incl = zh_wave_scan(mine ? rep : 0u);
    const uint32_t at = i + incl - (mine ? rep : 0u);  // index of my first entry
    if (mine && at >= total) mine = false;
    const uint64_t ON = __ballot(mine);
    if (__ballot(mine && at + rep > total)) return false;  // inflate.nim:161-162: runs past the end
    // "previous length" for symbol 16: the nearest symbol before that is not a 16
    const uint64_t plain = __ballot(mine && sym != 16u) & zh_lanemask_lt();
    const uint32_t from = plain ? 63u - (uint32_t)__clzll((long long)plain) : 0u;
    const uint32_t pv = (uint32_t)__shfl((int)val, (int)from, 64);
    if (sym == 16u) val = plain ? pv : prev;
    if (i == 0 && (ON & 1ull) && (uint32_t)__builtin_amdgcn_readlane((int)sym, 0) == 16u) return false;  // nothing to repeat
    uint32_t lk = 0, dk = 0, du = 0;
    bool eb = false;
    if (mine && val) {
      const uint32_t nl = at >= hlit ? 0u : (at + rep <= hlit ? rep : hlit - at), nd = rep - nl;
      lk = nl * (32768u >> val);
      dk = nd * (32768u >> val);
      du = nd;
      eb = at <= 256u && 256u < at + nl;
    }
    lit_kraft += zh_wave_sum(lk);
    dist_kraft += zh_wave_sum(dk);
    dist_used += zh_wave_sum(du);
    eob = eob || __ballot(eb) != 0;
    if (lit_kraft > 32768u || dist_kraft > 32768u) return false;
    // behind the last symbol taken
    const uint32_t lastl = 63u - (uint32_t)__clzll((long long)ON);  // (ON has bit 0: at = i < total)
    i = (uint32_t)__builtin_amdgcn_readlane((int)(at + rep), (int)lastl);
    prev = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)lastl);
    q += lastl + (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)lastl);
  }
  if (q > len * 8) return false;
  if (!eob || lit_kraft != 32768u) return false;
  return dist_kraft == 32768u || dist_used <= 1u;
}

// A wave per batch of the search: its queued candidates, one after the other; the segment keeps its
// lowest position that passes.
// A segment keeps the TWO lowest positions that read like a block's start: what reads like one and is none (bits of
// a payload: about one a GiB, many more in streams of literals only) would otherwise hide the block start behind it
// in the same segment -- and with blocks of many segments that start is the one every segment of its block needs.
// A stored block's header at bit p (a byte boundary) with another stored block's header behind its bytes: LEN and
// its complement twice over, 2^-36 of random positions.  (The last stored block of a chain, with a compressed block
// or the stream's end behind it, is not taken: whoever decodes the chain gets there.)
__device__ inline bool seg_stored_chain_ok(const uint8_t* src, uint64_t len, uint64_t p) {
  const uint64_t q = p >> 3;
  if (q + 10u > len) return false;
  const uint32_t lenv = src[q + 1] | ((uint32_t)src[q + 2] << 8), nlen = src[q + 3] | ((uint32_t)src[q + 4] << 8);
  if ((src[q] & 7u) != 0u || (lenv ^ nlen) != 0xffffu) return false;
  const uint64_t q2 = q + 5u + lenv;
  if (q2 + 5u > len) return false;
  const uint32_t len2 = src[q2 + 1] | ((uint32_t)src[q2 + 2] << 8), nlen2 = src[q2 + 3] | ((uint32_t)src[q2 + 4] << 8);
  return (src[q2] & 6u) == 0u && (len2 ^ nlen2) == 0xffffu;
}
__device__ __forceinline__ void seg_note_start(const ZhSegArgs& g, uint32_t sid, uint64_t p) {
  const uint64_t old = atomicMin((unsigned long long*)&g.start_bit[sid], (unsigned long long)p);
  if (old != p) atomicMin((unsigned long long*)&g.start2_bit[sid], (unsigned long long)(old > p ? old : p));
}
__global__ __launch_bounds__(64) void zh_seg_check_kernel(const uint8_t* __restrict__ d_src, ZhInflateArgs a,
                                                                  ZhSegArgs g, int serial) {
  __shared__ uint8_t s_lut[128];
  __shared__ uint32_t s_hdr[kHdrWords + 2];
  const uint32_t w = blockIdx.x, lane = threadIdx.x;
  const uint32_t nc = g.cand_n[w] < kFindSlots ? g.cand_n[w] : kFindSlots;
  if (!nc) return;
  const uint32_t sid = g.find_seg[w];
  const uint64_t base = g.nominal_bit[sid] + (uint64_t)(g.find_batch[w] & 0x7fffffffu) * kFindBatch;
  const uint32_t bid = g.parent[sid];
  const ZhBufDesc bd = a.bufs[bid];
  const uint64_t len = a.src_len_dev ? a.src_len_dev[bid] : bd.src_len;
#ifdef ZH_XCHECK
  if (serial) {  // a thread per candidate (the test build's cross-check)
    for (uint32_t c = lane; c < nc; c += 64u) {
      const uint32_t cand = g.cand_off[(size_t)w * kFindSlots + c];
      const uint64_t p = base + (cand & ~kSegCandStored);
      if (cand & kSegCandStored) {
        if (seg_stored_chain_ok(d_src + bd.src_off, len, p)) atomicMin((unsigned long long*)&g.stored_bit[sid], (unsigned long long)p);
        continue;
      }
      if (p >= *(volatile uint64_t*)&g.start2_bit[sid]) continue;  // (two lower positions have passed already)
      if (seg_lengths_ok(d_src + bd.src_off, len, p)) seg_note_start(g, sid, p);
    }
    return;
  }
#endif
  KPROF_DECL(8);  // cycles: 0 set-up, 1 candidates; counts: 3 candidates, 4 passed, 5 waves
  KPROF_COUNT(5, 1);
  KPROF_MARK(0);
  for (uint32_t c0 = 0; c0 < nc; c0 += 64u) {
    const uint32_t mine = c0 + lane < nc ? g.cand_off[(size_t)w * kFindSlots + c0 + lane] : 0u;
    // (stored-block candidates: a lane each)
    if (c0 + lane < nc && (mine & kSegCandStored)) {
      const uint64_t p = base + (mine & ~kSegCandStored);
      if (seg_stored_chain_ok(d_src + bd.src_off, len, p)) atomicMin((unsigned long long*)&g.stored_bit[sid], (unsigned long long)p);
    }
    for (uint32_t c = 0; c < 64u && c0 + c < nc; c++) {
      const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)c);
      if (cand & kSegCandStored) continue;
      const uint64_t p = base + cand;
      if (p >= zh_bcast64(*(volatile uint64_t*)&g.start2_bit[sid])) continue;  // (two lower positions have passed already)
      KPROF_COUNT(3, 1);
      if (seg_lengths_ok_wave(d_src + bd.src_off, len, p, s_lut, s_hdr)) {
        KPROF_COUNT(4, 1);
        if (lane == 0) seg_note_start(g, sid, p);
      }
    }
  }
  KPROF_MARK(1);
  KPROF_FLUSH(16, 8);
}

// A segment with no compressed block's start takes its first chained stored block's: inside a chain of stored blocks
// (incompressible data) every segment then has a decoder and a writer of its own instead of one workgroup copying the
// whole chain.  Compressed blocks' starts come first: the segments behind a long block take their sub-starts from its
// tables.
__global__ __launch_bounds__(256) void zh_seg_stored_starts_kernel(ZhSegArgs g) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k < g.nsegs && g.start_bit[k] == kSegNone) g.start_bit[k] = g.stored_bit[k];
}

// One wave per stream: is it worth it, and with how many decoders?  Fewer than four found starts
// (one huge block, fixed-code blocks: this library's own streams of up to 4 MiB are a single block)
// send the stream to the ordinary kernels right away.  Where starts are rare -- blocks much longer
// than a segment -- the segments are taken in groups of f (a power of two, about half the spacing of
// the starts): only a group's first found start keeps its decoder, and that decoder gets the token
// regions of the whole group, which lie back to back -- so that a region holds the tokens of a
// block or two whatever the block size.
__global__ __launch_bounds__(64) void zh_seg_decide_kernel(ZhInflateArgs a, ZhSegArgs g, int rerun) {
  const uint32_t bid = blockIdx.x;
  const unsigned lane = zh_lane();
  if (rerun && !g.repair[bid]) return;
  const uint32_t first = g.first_seg[bid], last = g.first_seg[bid + 1u], n = last - first;
  // segments inside long blocks take the sub-starts phase 0 of the tokens kernel guessed for them
  uint32_t found = 0;
  for (uint32_t k = first + lane; k < last; k += 64u) {
    uint64_t sk = g.start_bit[k];
    uint32_t sub = 0;
    if (sk == kSegNone && k != first && g.sub_start[k] != kSegNone) {
      sk = g.sub_start[k];
      g.start_bit[k] = sk;
      sub = 1;
    }
    g.is_sub[k] = sub;
    if (!rerun) g.held_start[k] = kSegNone;
    found += sk != kSegNone;
  }
  found = zh_wave_sum(found);
  zh_wave_sync();
  const bool go = a.status[bid] == ZH_OK && found >= 4u;
  // (half the spacing of the starts: a region then takes three tokens per compressed byte of the
  // stretch its decoder covers, and no decoder is given up while starts are merely not everywhere)
  uint32_t f = 1;
  while (go && f * found * 2u < n) f <<= 1;
  // lane l takes groups l, l + 64, ...
  for (uint32_t grp = lane; go && grp * f < n; grp += 64u) {
    const uint32_t k0 = first + grp * f, k1 = k0 + f < last ? k0 + f : last;
    const uint64_t base = g.tok_off[k0], room = g.tok_off[k1 - 1u] + g.tok_cap[k1 - 1u] - base;
    bool taken = false;
    for (uint32_t k = k0; k < k1; k++) {
      if (g.start_bit[k] == kSegNone) continue;
      if (taken) {  // (set aside, not forgotten: should the group's first start be none, a repair round asks this one)
        if (!g.is_sub[k]) g.held_start[k] = g.start_bit[k];
        g.start_bit[k] = kSegNone;
        continue;
      }
      taken = true;
      g.eff_tok_off[k] = base;
      g.eff_tok_cap[k] = room;
    }
  }
  // A decoder carries on through the segments behind it that have no start of their own (a stretch without
  // block starts and without sub-starts: a stored block; a block whose sub-starts a wrong guess before them has
  // spoilt), and their token regions, which lie right behind its own, are nobody's: they are its room, too.
  __threadfence();
  zh_wave_sync();
  auto start_of = [&](uint32_t k) -> uint64_t {
    return __hip_atomic_load(&g.start_bit[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (uint32_t k = first + lane; go && k < last; k += 64u) {
    if (start_of(k) == kSegNone) continue;
    uint32_t j = k + 1u;
    while (j < last && start_of(j) == kSegNone) j++;
    const uint64_t off = __hip_atomic_load(&g.eff_tok_off[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t room_end = j < last ? __hip_atomic_load(&g.eff_tok_off[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : g.tok_off[last - 1u] + g.tok_cap[last - 1u];
    if (room_end > off) g.eff_tok_cap[k] = room_end - off;
  }
  if (lane == 0) g.go[bid] = go ? 1u : 0u;
}

// One wave per stream, behind the chain kernel: a chain that does not hold, ONCE MORE without the found starts that
// were none.  Bits of a payload that read like a block header turn up about once a GiB (in streams of literals
// only: dozens), and where blocks are many segments long such a guess costs more than its own decoder: the
// segments behind it take their sub-starts from its "tables" and find none or wrong ones, it may hide the block's
// real start behind it in the same segment, and the decoder before is left alone with the rest of its block until
// its token room runs out -- the chain breaks and the whole stream falls back to one workgroup, a hundred times
// slower.  What is none gives itself away: its decoder fails within a few hundred bytes (an invalid symbol, a
// distance before the start of the data, an "end of block" with no header behind it), which in a sound stream no
// real block's decoder does.  So: every found start whose decoder failed makes room for its segment's second
// candidate (g.start2_bit), every sub-start is forgotten, and phase 0, the decision, the tokens and the chain run
// again for this stream (g.repair).  A damaged stream fails again and goes to the ordinary kernels, which report
// what is wrong with it.
__global__ __launch_bounds__(64) void zh_seg_repair_kernel(ZhInflateArgs a, ZhSegArgs g) {
  const uint32_t bid = blockIdx.x;
  const unsigned lane = zh_lane();
  const uint32_t first = g.first_seg[bid], last = g.first_seg[bid + 1u];
  const bool broken = last > first && a.status[bid] == ZH_OK && g.go[bid] != 0u && g.stream_ok[bid] == 0u;
  auto failed = [&](uint32_t k) -> bool {  // a found start whose decoder gave itself away
    return g.start_bit[k] != kSegNone && !g.is_sub[k] && g.seg_status[k] != ZH_OK && g.seg_status[k] != ZH_ERR_DST_TOO_SMALL;
  };
  uint32_t dropped = 0;
  for (uint32_t k = first + 1u + lane; broken && k < last; k += 64u) dropped += failed(k) ? 1u : 0u;  // (the stream's first block is where it is)
  dropped = zh_wave_sum(dropped);
  // nothing to take out: nothing is touched (the stream goes to the ordinary kernels as it stands)
  for (uint32_t k = first + 1u + lane; dropped && k < last; k += 64u) {
    if (g.start_bit[k] == kSegNone) {
      // a found start the grouping had set aside: the group's first one may be what is taken out now, and the
      // decision of the round to come sees every found start again
      if (g.held_start[k] != kSegNone) {
        g.start_bit[k] = g.held_start[k];
        g.held_start[k] = kSegNone;
      }
    } else if (g.is_sub[k]) {
      g.start_bit[k] = kSegNone;
    } else if (failed(k)) {
      g.start_bit[k] = g.start2_bit[k];
      g.start2_bit[k] = kSegNone;
    }
  }
  if (lane == 0) g.repair[bid] = dropped ? 1u : 0u;
}

// One wave per stream: the chain of segments.  It starts with the stream's first segment; the next
// link is the segment whose found start is exactly where the decoder of the last one stopped (it
// stops nowhere else, unless the stream ends or fails); found starts in between were wrong guesses.
__global__ __launch_bounds__(64) void zh_seg_chain_kernel(ZhInflateArgs a, ZhSegArgs g, int rerun) {
  const uint32_t bid = blockIdx.x;
  const unsigned lane = zh_lane();
  if (rerun && !g.repair[bid]) return;
  const uint32_t first = g.first_seg[bid], last = g.first_seg[bid + 1u];
  bool ok = a.status[bid] == ZH_OK && g.go[bid] != 0u;
  bool done = false;
  uint64_t total = 0;
  uint32_t nchain = 0, before = 0xffffffffu, c = first;
  auto shfl64 = [&](uint64_t v, uint32_t l) -> uint64_t {
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)l, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, (int)l, 64);
  };
  // a chunk of 64 segments in registers
  uint64_t start = kSegNone, endb = 0, outb = 0;
  int32_t sst = 0;
  uint32_t fin = 0;
  auto load_chunk = [&]() {
    const uint32_t k = c + lane;
    start = k < last ? g.start_bit[k] : kSegNone;
    const bool found = start != kSegNone;
    sst = found ? g.seg_status[k] : 0;
    fin = found ? g.final_block[k] : 0u;
    endb = found ? g.end_bit[k] : 0ull;
    outb = found ? g.seg_out[k] : 0ull;
    if (k < last) g.valid[k] = 0;
  };
  if (ok) load_chunk();
  uint64_t want = ok ? shfl64(start, 0) : kSegNone;  // (the first segment's start is exact)
  ok = ok && want != kSegNone;
  while (ok && !done) {
    const bool found = start != kSegNone;
    const uint64_t E = __ballot(found && start == want);
    if (!E) {
      // not in this chunk: behind it, unless a found start lies past the wanted position already
      if (__ballot(found && start > want) || c + 64u >= last) {
        ok = false;
        break;
      }
      c += 64u;
      load_chunk();
      continue;
    }
    // every lane's link inside the chunk: the lane whose start is where this lane's decoder stopped
    // (starts grow with the lane, so there is at most one)
    uint32_t nxt = 64u;
    for (int d = 1; d < 64; d++) {
      const uint64_t s2 = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(start >> 32), d, 64) << 32) |
                          (uint32_t)__shfl_down((int)(uint32_t)start, d, 64);
      if (found && lane + (unsigned)d < 64u && s2 == endb && s2 != kSegNone) nxt = lane + (unsigned)d;
    }
    const uint64_t BAD = __ballot(found && sst != ZH_OK), FIN = __ballot(found && fin != 0u);
    // the walk through the chunk: wave-uniform, one v_readlane per link
    uint32_t j = (uint32_t)__ffsll((long long)E) - 1u, jl = j;
    uint64_t ON = 0;
    for (;;) {
      ON |= 1ull << j;
      jl = j;
      if ((BAD >> j) & 1ull) {
        ok = false;
        break;
      }
      if ((FIN >> j) & 1ull) {
        done = true;
        break;
      }
      const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)j);
      if (nj >= 64u) break;
      j = nj;
    }
    if (!ok) break;
    const bool on = (ON >> lane) & 1ull;
    const uint64_t below = ON & zh_lanemask_lt();
    uint32_t slo = on ? (uint32_t)outb : 0u, shi = on ? (uint32_t)(outb >> 32) : 0u;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t tl = (uint32_t)__shfl_up((int)slo, o, 64), th = (uint32_t)__shfl_up((int)shi, o, 64);
      if (lane >= (unsigned)o) {
        const uint64_t sum = (((uint64_t)shi << 32) | slo) + (((uint64_t)th << 32) | tl);
        slo = (uint32_t)sum;
        shi = (uint32_t)(sum >> 32);
      }
    }
    const uint64_t incl = ((uint64_t)shi << 32) | slo;
    if (on) {
      const uint32_t k = c + lane, ord = nchain + (uint32_t)__popcll(below);
      g.valid[k] = 1;
      g.prev[k] = below ? c + 63u - (uint32_t)__clzll((long long)below) : before;
      g.out_start[k] = total + incl - outb;
      g.order[first + ord] = k;
      g.ordinal[k] = ord;
    }
    nchain += (uint32_t)__popcll(ON);
    total += shfl64(incl, 63);
    before = c + jl;
    want = shfl64(endb, jl);
    if (done) break;
    if (c + 64u >= last) {  // (the chain leaves the last chunk without having met the last block)
      ok = false;
      break;
    }
    c += 64u;
    load_chunk();
  }
  // (a chain of a few links -- a stream of stored blocks: no starts to land on, the first decoder has taken it all --
  // is one workgroup's work whatever it is called, and the ordinary kernels are the faster way to do that)
  ok = ok && done && nchain >= 4u;
  bool write = ok;
  if (ok && a.count_only) {  // a sizing pass: `total` is the answer
    write = false;
  } else if (ok && total > a.bufs[bid].dst_cap) {
    // more output than the slot holds: nothing is written (0 bytes of the slot are valid)
    write = false;
    total = 0;
    if (lane == 0) a.status[bid] = ZH_ERR_DST_TOO_SMALL;
  }
  if (!write)
    for (uint32_t k = first + lane; k < last; k += 64u) g.valid[k] = 0;
  else
    for (uint32_t k = c + 64u + lane; k < last; k += 64u) g.valid[k] = 0;
  if (lane == 0) {
    g.stream_ok[bid] = ok ? 1u : 0u;
    g.nchain[bid] = write ? nchain : 0u;
    if (ok) a.out_len[bid] = total;
#ifdef ZH_EMU
    if (getenv("ZH_DBG_SEG"))
      fprintf(stderr, "stream %u: %u segments, %u on the chain, ok %d, written %d, %llu bytes\n", bid, last - first,
              nchain, (int)ok, (int)write, (unsigned long long)total);
    if (getenv("ZH_DBG_SEG") && atoi(getenv("ZH_DBG_SEG")) > 1 && !ok)
      for (uint32_t k = first; k < last; k++)
        fprintf(stderr, "  seg %u: nominal %llu start %lld end %llu status %d final %u out %llu\n", k - first,
                (unsigned long long)g.nominal_bit[k], (long long)g.start_bit[k], (unsigned long long)g.end_bit[k],
                g.seg_status[k], g.final_block[k], (unsigned long long)g.seg_out[k]);
#endif
  }
}

// The windows.  What a chain segment's last 32 KiB are, given the 32 KiB before the segment, is a
// map (g.winsym: a byte, or "byte k of the window before"), and maps compose: the chain is cut into
// groups of kWinGroup segments,
//   zh_seg_windows_group_kernel  a workgroup per group composes its segments' maps in order, so that
//                                each refers to the window before the GROUP (in place, window in LDS);
//   zh_seg_windows_chain_kernel  a workgroup per stream turns the maps of the groups' last segments
//                                into bytes, group by group (g.windows of those segments);
// and a byte of any segment's window is at most two look-ups away (zh_seg_finish_kernel).  The
// serial depth is kWinGroup + groups steps of 32 KiB instead of one step per segment.
constexpr uint32_t kWinGroup = 32, kWinMaxGroups = 4096 / kWinGroup;

__global__ __launch_bounds__(1024) void zh_seg_windows_group_kernel(ZhInflateArgs a, ZhSegArgs g) {
  __shared__ uint16_t s_win[kWin];
  const uint32_t bid = blockIdx.x / kWinMaxGroups, grp = blockIdx.x % kWinMaxGroups, tid = threadIdx.x;
  if (!g.stream_ok[bid] || a.status[bid] != ZH_OK) return;
  const uint32_t n = g.nchain[bid], first = g.first_seg[bid];
  const uint32_t c0 = grp * kWinGroup, c1 = c0 + kWinGroup < n ? c0 + kWinGroup : n;
  if (c0 >= n) return;
  constexpr uint32_t kPer = kWin / 2u / 1024u;  // dwords (symbol pairs) per thread: 16
  auto load = [&](uint32_t k, uint32_t* w) {
    const uint32_t* ws = reinterpret_cast<const uint32_t*>(g.winsym + (size_t)k * kWin);
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) w[i] = ws[tid + 1024u * i];
  };
  uint32_t cur[kPer], nxt[kPer];
  uint32_t k = g.order[first + c0], kn = c0 + 1u < c1 ? g.order[first + c0 + 1u] : k;
  load(k, cur);
  for (uint32_t c = c0; c < c1; c++) {
    const uint32_t knn = c + 2u < c1 ? g.order[first + c + 2u] : kn;  // (two ahead: its address is there in time)
    if (c + 1u < c1) load(kn, nxt);
    const bool head = c == c0;
    uint32_t v[kPer];
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) {
      uint32_t s0 = cur[i] & 0xffffu, s1 = cur[i] >> 16;
      if (head) {
        if (c == 0) {  // nothing lies before the stream (a copy from there has been refused already)
          s0 &= s0 & 0x8000u ? 0u : 0xffu;
          s1 &= s1 & 0x8000u ? 0u : 0xffu;
        }
      } else {
        if (s0 & 0x8000u) s0 = s_win[s0 & 0x7fffu];
        if (s1 & 0x8000u) s1 = s_win[s1 & 0x7fffu];
      }
      v[i] = s0 | (s1 << 16);
    }
    __syncthreads();  // (everybody has read the window before it changes)
    uint32_t* gw = reinterpret_cast<uint32_t*>(g.winsym + (size_t)k * kWin);
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) {
      const uint32_t j = tid + 1024u * i;  // pair index
      reinterpret_cast<uint32_t*>(s_win)[j] = v[i];
      if (!head || c == 0) gw[j] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) cur[i] = nxt[i];
    k = kn;
    kn = knn;
  }
}

__global__ __launch_bounds__(1024) void zh_seg_windows_chain_kernel(ZhInflateArgs a, ZhSegArgs g) {
  __shared__ uint8_t s_win[kWin];
  __shared__ uint32_t s_failed;
  const uint32_t bid = blockIdx.x, tid = threadIdx.x;
  if (!g.stream_ok[bid] || a.status[bid] != ZH_OK) return;
  const uint32_t n = g.nchain[bid], first = g.first_seg[bid];
  // a writer that refused a copy (inflate.nim:224-225) fails the stream where it stopped
  if (tid == 0) s_failed = 0xffffffffu;
  __syncthreads();
  for (uint32_t c = tid; c < n; c += 1024u)
    if (g.seg_status[g.order[first + c]] != ZH_OK) atomicMin(&s_failed, c);
  __syncthreads();
  if (s_failed != 0xffffffffu) {
    if (tid == 0) {
      const uint32_t k = g.order[first + s_failed];
      a.status[bid] = g.seg_status[k];
      a.out_len[bid] = g.out_start[k] + g.wr_len[k];
    }
    return;
  }
  const uint32_t groups = (n + kWinGroup - 1u) / kWinGroup;
  constexpr uint32_t kPer = kWin / 2u / 1024u;
  auto last_of = [&](uint32_t grp) -> uint32_t {
    const uint32_t c = (grp + 1u) * kWinGroup < n ? (grp + 1u) * kWinGroup : n;
    return g.order[first + c - 1u];
  };
  auto load = [&](uint32_t k, uint32_t* w) {
    const uint32_t* ws = reinterpret_cast<const uint32_t*>(g.winsym + (size_t)k * kWin);
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) w[i] = ws[tid + 1024u * i];
  };
  uint32_t cur[kPer], nxt[kPer];
  uint32_t k = groups ? last_of(0) : 0u, kn = groups > 1u ? last_of(1) : k;
  if (groups) load(k, cur);
  for (uint32_t grp = 0; grp < groups; grp++) {
    const uint32_t knn = grp + 2u < groups ? last_of(grp + 2u) : kn;
    if (grp + 1u < groups) load(kn, nxt);
    uint16_t v[kPer];
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) {
      const uint32_t s0 = cur[i] & 0xffffu, s1 = cur[i] >> 16;
      uint32_t v0 = s0 & 0xffu, v1 = s1 & 0xffu;
      if (s0 & 0x8000u) v0 = grp ? s_win[s0 & 0x7fffu] : 0u;
      if (s1 & 0x8000u) v1 = grp ? s_win[s1 & 0x7fffu] : 0u;
      v[i] = (uint16_t)(v0 | (v1 << 8));
    }
    __syncthreads();
    uint16_t* gw = reinterpret_cast<uint16_t*>(g.windows + (size_t)k * kWin);
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) {
      const uint32_t j = tid + 1024u * i;
      reinterpret_cast<uint16_t*>(s_win)[j] = v[i];
      gw[j] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) cur[i] = nxt[i];
    k = kn;
    kn = knn;
  }
}

// Symbols -> bytes; `parts` workgroups share a segment.  A marker is byte k of the window before the
// segment: the map of the chain segment before it (relative to its group), then the byte window
// that ends the group before that one.
__global__ __launch_bounds__(256) void zh_seg_finish_kernel(uint8_t* __restrict__ d_dst, ZhInflateArgs a, ZhSegArgs g,
                                                            uint32_t parts) {
  const uint32_t k = blockIdx.x / parts, part = blockIdx.x % parts;
  if (!g.valid[k]) return;
  const uint32_t bid = g.parent[k];
  if (a.status[bid] != ZH_OK) return;
  const uint64_t start = g.out_start[k], n = g.wr_len[k];
  const uint16_t* sym = g.sym + g.sym_base[bid] + start;
  uint8_t* dst = d_dst + a.bufs[bid].dst_off + start;
  const uint32_t pk = g.prev[k];
  const uint16_t* pmap = nullptr;
  const uint8_t* pwin = nullptr;
  if (pk != 0xffffffffu) {
    pmap = g.winsym + (size_t)pk * kWin;
    const uint32_t grp = g.ordinal[pk] / kWinGroup;
    if (grp) pwin = g.windows + (size_t)g.order[g.first_seg[bid] + grp * kWinGroup - 1u] * kWin;
  }
  for (uint64_t i = (uint64_t)part * 256u + threadIdx.x; i < n; i += (uint64_t)parts * 256u) {
    uint32_t s = sym[i];
    if (s & 0x8000u) {
      s = pmap ? pmap[s & 0x7fffu] : 0u;
      if (s & 0x8000u) s = pwin ? pwin[s & 0x7fffu] : 0u;
    }
    dst[i] = (uint8_t)s;
  }
}

extern "C" void zh_launch_seg_find(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nsegs || !g.nfind) return;
  hipLaunchKernelGGL(zh_seg_find_kernel, dim3(g.nfind), dim3(kFindThreads), 0, stream, d_src, a, g);
}
extern "C" void zh_launch_seg_check(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nsegs || !g.nfind) return;
#ifdef ZH_XCHECK  // (the test build: a thread instead of a wave per candidate, ZH_SEG_CHECK=serial)
  static const int serial = [] {
    const char* e = getenv("ZH_SEG_CHECK");
    return e && strcmp(e, "serial") == 0 ? 1 : 0;
  }();
#else
  const int serial = 0;
#endif
  hipLaunchKernelGGL(zh_seg_check_kernel, dim3(g.nfind), dim3(64), 0, stream, d_src, a, g, serial);
  hipLaunchKernelGGL(zh_seg_stored_starts_kernel, dim3((g.nsegs + 255u) / 256u), dim3(256), 0, stream, g);
}
// ZH_SEG_FAKE_START=<bit> (tests): bits that read like a block header and are none happen in any long stream's
// payload, about once a GiB -- too rare for a test to wait for.  This plants one: the segment whose search range
// holds stream bit `bit` reports a block start there (unless it found an earlier one), the way zh_seg_check_kernel
// would have.
__global__ __launch_bounds__(64) void zh_seg_fake_start_kernel(ZhSegArgs g, uint64_t bit) {
  const uint32_t k = blockIdx.x * 64u + threadIdx.x;
  if (k >= g.nsegs || k == g.first_seg[g.parent[k]]) return;
  if (g.nominal_bit[k] <= bit && bit < g.nominal_bit[k] + g.search_bits[k]) {
    if (bit < g.start_bit[k]) {
      g.start2_bit[k] = g.start_bit[k];
      g.start_bit[k] = bit;
    } else if (bit > g.start_bit[k] && bit < g.start2_bit[k]) {
      g.start2_bit[k] = bit;
    }
  }
}
extern "C" void zh_launch_seg_fake_start(hipStream_t stream, ZhSegArgs g, uint64_t bit) {
  if (!g.nsegs) return;
  hipLaunchKernelGGL(zh_seg_fake_start_kernel, dim3((g.nsegs + 63u) / 64u), dim3(64), 0, stream, g, bit);
}
extern "C" void zh_launch_seg_decide(hipStream_t stream, ZhInflateArgs a, ZhSegArgs g, int rerun) {
  if (!g.nsegs) return;
  hipLaunchKernelGGL(zh_seg_decide_kernel, dim3(g.nstreams), dim3(64), 0, stream, a, g, rerun);
}
extern "C" void zh_launch_seg_chain(hipStream_t stream, ZhInflateArgs a, ZhSegArgs g, int rerun) {
  if (!g.nstreams) return;
  hipLaunchKernelGGL(zh_seg_chain_kernel, dim3(g.nstreams), dim3(64), 0, stream, a, g, rerun);
}
// zh_debug_segment_stats: streams that were cut into segments / whose chain held, counted where the answer is
// (two device counters of the context; nothing of it on the host side of a run or of a result read)
__global__ __launch_bounds__(64) void zh_seg_stats_kernel(ZhSegArgs g, unsigned long long* stats) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  if (i >= g.nstreams || g.first_seg[i + 1u] <= g.first_seg[i]) return;
  atomicAdd(&stats[0], 1ull);
  if (g.stream_ok[i]) atomicAdd(&stats[1], 1ull);
}
extern "C" void zh_launch_seg_stats(hipStream_t stream, ZhSegArgs g, uint64_t* stats) {
  if (!g.nstreams || !stats) return;
  hipLaunchKernelGGL(zh_seg_stats_kernel, dim3((g.nstreams + 63u) / 64u), dim3(64), 0, stream, g,
                     reinterpret_cast<unsigned long long*>(stats));
}
extern "C" void zh_launch_seg_repair(hipStream_t stream, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nstreams) return;
  hipLaunchKernelGGL(zh_seg_repair_kernel, dim3(g.nstreams), dim3(64), 0, stream, a, g);
}
extern "C" void zh_launch_seg_windows(hipStream_t stream, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nstreams) return;
  hipLaunchKernelGGL(zh_seg_windows_group_kernel, dim3(g.nstreams * kWinMaxGroups), dim3(1024), 0, stream, a, g);
  hipLaunchKernelGGL(zh_seg_windows_chain_kernel, dim3(g.nstreams), dim3(1024), 0, stream, a, g);
}
extern "C" void zh_launch_seg_finish(hipStream_t stream, uint8_t* d_dst, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nsegs) return;
  const uint32_t parts = g.nsegs >= 2048u ? 2u : g.nsegs >= 512u ? 8u : 32u;
  hipLaunchKernelGGL(zh_seg_finish_kernel, dim3(g.nsegs * parts), dim3(256), 0, stream, d_dst, a, g, parts);
}
