// One LARGE deflate stream on many workgroups: the kernels around the segment forms of
// zh_inflate_tokens_kernel / zh_inflate_write_kernel (zh_inflate_split.hip).
//
// inflate.nim:268-289 walks a stream block by block; where a block starts is only known once the
// one before it has been decoded, which ties a stream to one workgroup however large it is (a
// tar.gz, a 128 MiB zlib stream).  This file removes that chain the way parallel gzip readers do:
//
//   zh_seg_find_kernel     the compressed bytes are cut into segments of equal length; a workgroup
//                          per segment looks for the first bit position at or behind the segment's
//                          nominal start that reads as the header of a dynamic-Huffman block
//                          (inflate.nim:115-171): BTYPE = 2, HLIT/HDIST in range, a complete
//                          code-length code, code lengths that decode to exactly HLIT + HDIST
//                          entries and describe a complete literal/length code with an
//                          end-of-block symbol and a usable distance code.  A position that passes
//                          is almost certainly a block start, but it is only a GUESS;
//   tokens (segment form)  decode from the found start to the first block boundary at or behind the
//                          next found start, all segments at once;
//   zh_seg_chain_kernel    the proof: walking the stream's segments in order, every decoder must
//                          have stopped exactly where the next one started (the first one starts
//                          at the stream's first block, which is exact, so by induction every start
//                          on the chain then is a real block start and the concatenated tokens are
//                          the serial decoder's).  Prefix-sums the output bytes.  Anything else --
//                          a wrong guess, an error in a chain segment, a token region that
//                          overflowed, more output than the slot holds -- clears the stream's flag
//                          and the ordinary one-workgroup kernels decode it (they are launched
//                          right behind and return at once for flagged streams);
//   writer (segment form)  every chain segment's bytes as 16-bit symbols: a copy that reaches into
//                          the 32 KiB before the segment yields "window byte k" instead of a value;
//   zh_seg_windows_kernel  chain segments in order, one workgroup per stream: the last 32 KiB at
//                          the end of a segment, resolved through the window before it (LDS);
//   zh_seg_finish_kernel   symbols -> bytes in the caller's slot, all segments at once.
//
// Accept/reject, status and bytes are those of the ordinary kernels (same decoder, same checks; the
// tests run both against the oracle).  Extra traffic: 4 bytes per token and 2 + 2 bytes per output
// byte for the symbols.
#include <cstdlib>

#include "zh_common.h"
#include "zh_tables.h"
#include "zh_inflate_tables.h"

namespace {

constexpr uint32_t kFindThreads = 1024;
constexpr uint32_t kFindPer = 64;                         // bit positions per thread and batch
constexpr uint32_t kFindBatch = kFindThreads * kFindPer;  // 65536 positions
constexpr uint32_t kWin = 32768;

// the 32 bits at bit position p of the stream (bits past the end read as zero)
__device__ __forceinline__ uint32_t seg_peek(const uint8_t* src, uint64_t len, uint64_t p) {
  const uint64_t b = p >> 3;
  const uint32_t sh = (uint32_t)p & 7u;
  uint64_t v = 0;
  if (b + 8 <= len) {
    struct __attribute__((packed)) U64 { uint64_t v; };
    v = reinterpret_cast<const U64*>(src + b)->v;
  } else {
    for (uint32_t i = 0; i < 8; i++)
      if (b + i < len) v |= (uint64_t)src[b + i] << (8 * i);
  }
  return (uint32_t)(v >> sh);
}

// Does a dynamic-Huffman block header start at bit p?  (The cheap tests first.)
__device__ bool seg_header_at(const uint8_t* src, uint64_t len, uint64_t p) {
  const uint32_t h = seg_peek(src, len, p);
  if ((h & 7u) != 4u) return false;  // BFINAL = 0, BTYPE = 2 (the last block is left to the decoder before it)
  const uint32_t hlit = ((h >> 3) & 31u) + 257u, hdist = ((h >> 8) & 31u) + 1u, hclen = ((h >> 13) & 15u) + 4u;
  if (hlit > 286u || hdist > 30u) return false;
  // the code-length code: 3 bits per entry in c_clcl_order; it must be complete
  uint64_t cl = 0;  // 3 bits per symbol 0..18
  uint32_t kraft = 0;
  {
    const uint32_t w0 = seg_peek(src, len, p + 17), w1 = seg_peek(src, len, p + 47);  // entries 0-9, 10-18
    for (uint32_t i = 0; i < hclen; i++) {
      const uint32_t v = (i < 10u ? w0 >> (3u * i) : w1 >> (3u * (i - 10u))) & 7u;
      cl |= (uint64_t)v << (3u * c_clcl_order[i]);
      if (v) kraft += 128u >> v;
    }
  }
  if (kraft != 128u) return false;
  // canonical code (inflate.nim:29-65 in miniature): counts per length, symbols in code order
  uint32_t count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t s = 0; s < 19; s++) count[(cl >> (3u * s)) & 7u]++;
  uint32_t offs[8];
  offs[1] = 0;
  for (uint32_t l = 1; l < 7; l++) offs[l + 1] = offs[l] + count[l];
  uint64_t sorted_lo = 0, sorted_hi = 0;  // 5 bits per slot, 12 slots a word
  for (uint32_t s = 0; s < 19; s++) {
    const uint32_t l = (uint32_t)(cl >> (3u * s)) & 7u;
    if (!l) continue;
    const uint32_t at = offs[l]++;
    if (at < 12u) sorted_lo |= (uint64_t)s << (5u * at);
    else sorted_hi |= (uint64_t)s << (5u * (at - 12u));
  }
  // the HLIT + HDIST code lengths (inflate.nim:131-165); both codes are checked as they come
  uint64_t q = p + 17 + 3 * hclen;
  const uint32_t total = hlit + hdist;
  uint32_t i = 0, prev = 0, lit_kraft = 0, dist_kraft = 0, dist_used = 0;
  bool eob = false;
  while (i < total) {
    uint32_t w = seg_peek(src, len, q);
    uint32_t code = 0, first = 0, index = 0, sym = 0xffu, nb = 0;
    for (uint32_t l = 1; l <= 7; l++) {
      code |= w & 1u;
      w >>= 1;
      const uint32_t c = count[l];
      if (code < first + c) {  // (code >= first always holds: the code is complete)
        const uint32_t at = index + (code - first);
        sym = at < 12u ? (uint32_t)(sorted_lo >> (5u * at)) & 31u : (uint32_t)(sorted_hi >> (5u * (at - 12u))) & 31u;
        nb = l;
        break;
      }
      index += c;
      first = (first + c) << 1;
      code <<= 1;
    }
    if (sym == 0xffu) return false;
    q += nb;
    uint32_t rep = 1, val = sym;
    if (sym == 16) {
      if (i == 0) return false;
      rep = (w & 3u) + 3u;
      q += 2;
      val = prev;
    } else if (sym == 17) {
      rep = (w & 7u) + 3u;
      q += 3;
      val = 0;
    } else if (sym == 18) {
      rep = (w & 127u) + 11u;
      q += 7;
      val = 0;
    }
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

__global__ __launch_bounds__(kFindThreads) void zh_seg_find_kernel(const uint8_t* __restrict__ d_src, ZhInflateArgs a,
                                                                   ZhSegArgs g) {
  __shared__ uint32_t s_best;
  const uint32_t sid = blockIdx.x, tid = threadIdx.x;
  const uint32_t bid = g.parent[sid];
  if (a.status[bid] != ZH_OK) {
    if (tid == 0) g.start_bit[sid] = kSegNone;
    return;
  }
  if (sid == g.first_seg[bid]) {  // the stream's first block: behind the container header, exact
    if (tid == 0) g.start_bit[sid] = (uint64_t)a.body_pos[bid] * 8;
    return;
  }
  const ZhBufDesc bd = a.bufs[bid];
  const uint8_t* src = d_src + bd.src_off;
  const uint64_t len = a.src_len_dev ? a.src_len_dev[bid] : bd.src_len;
  const uint64_t lo = g.nominal_bit[sid];
  uint64_t hi = lo + g.search_bits[sid];
  if (hi > len * 8) hi = len * 8;
  for (uint64_t base = lo; base < hi; base += kFindBatch) {
    if (tid == 0) s_best = 0xffffffffu;
    __syncthreads();
    // interleaved: in step j the threads test kFindThreads consecutive positions
    for (uint32_t j = 0; j < kFindPer; j++) {
      const uint64_t p = base + (uint64_t)j * kFindThreads + tid;
      if (p >= hi) break;
      if ((uint32_t)(p - base) > *(volatile uint32_t*)&s_best) break;  // (a lower position has passed already)
      if (seg_header_at(src, len, p)) {
        atomicMin(&s_best, (uint32_t)(p - base));
        break;
      }
    }
    __syncthreads();
    const uint32_t best = s_best;
    if (best != 0xffffffffu) {
      if (tid == 0) g.start_bit[sid] = base + best;
      return;
    }
    __syncthreads();
  }
  if (tid == 0) g.start_bit[sid] = kSegNone;
}

// One thread per stream: does the chain of segments hold?
__global__ __launch_bounds__(64) void zh_seg_chain_kernel(ZhInflateArgs a, ZhSegArgs g) {
  const uint32_t bid = blockIdx.x * 64u + threadIdx.x;
  if (bid >= g.nstreams) return;
  const uint32_t first = g.first_seg[bid], last = g.first_seg[bid + 1u];
  for (uint32_t k = first; k < last; k++) g.valid[k] = 0;
  bool ok = a.status[bid] == ZH_OK && g.start_bit[first] != kSegNone;
  uint64_t total = 0;
  if (ok) {
    uint32_t cur = first, before = 0xffffffffu;
    for (;;) {
      if (g.seg_status[cur] != ZH_OK) {
        ok = false;
        break;
      }
      g.valid[cur] = 1;
      g.prev[cur] = before;
      g.out_start[cur] = total;
      total += g.seg_out[cur];
      before = cur;
      if (g.final_block[cur]) break;
      uint32_t t = cur + 1u;
      while (t < last && g.start_bit[t] == kSegNone) t++;
      if (t == last || g.end_bit[cur] != g.start_bit[t]) {  // the guess behind `cur` was wrong
        ok = false;
        break;
      }
      cur = t;
    }
  }
  if (ok && a.count_only) {  // a sizing pass: this is the answer
    for (uint32_t k = first; k < last; k++) g.valid[k] = 0;
  } else if (ok && total > a.bufs[bid].dst_cap) {
    // more output than the slot holds: nothing is written (out_len 0 bytes of it are valid)
    for (uint32_t k = first; k < last; k++) g.valid[k] = 0;
    a.status[bid] = ZH_ERR_DST_TOO_SMALL;
    total = 0;
  }
  if (!ok)
    for (uint32_t k = first; k < last; k++) g.valid[k] = 0;
  g.stream_ok[bid] = ok ? 1u : 0u;
  if (ok) a.out_len[bid] = total;
#ifdef ZH_EMU
  if (getenv("ZH_DBG_SEG")) {
    uint32_t found = 0, onchain = 0;
    for (uint32_t k = first; k < last; k++) {
      found += g.start_bit[k] != kSegNone;
      onchain += g.valid[k];
    }
    fprintf(stderr, "stream %u: %u segments, %u starts found, %u on the chain, ok %d, %llu bytes\n", bid, last - first,
            found, onchain, (int)ok, (unsigned long long)total);
  }
#endif
}

// One workgroup per stream: the 32 KiB of output that end each chain segment, as bytes.
__global__ __launch_bounds__(1024) void zh_seg_windows_kernel(ZhInflateArgs a, ZhSegArgs g) {
  __shared__ uint8_t s_win[2][kWin];
  const uint32_t bid = blockIdx.x, tid = threadIdx.x;
  if (!g.stream_ok[bid] || a.status[bid] != ZH_OK) return;
  const uint32_t first = g.first_seg[bid], last = g.first_seg[bid + 1u];
  const uint16_t* sym = g.sym + g.sym_base[bid];
  uint32_t par = 0;
  bool have_prev = false;
  int st = ZH_OK;
  uint64_t fail_len = 0;
  for (uint32_t k = first; k < last; k++) {
    if (!g.valid[k]) continue;
    if (g.seg_status[k] != ZH_OK) {  // the writer refused a copy (inflate.nim:224-225)
      st = g.seg_status[k];
      fail_len = g.out_start[k] + g.wr_len[k];
      break;
    }
    const uint64_t start = g.out_start[k], n = g.seg_out[k];
    const uint8_t* pw = s_win[par];
    uint8_t* nw = s_win[par ^ 1u];
    uint8_t* gw = g.windows + (size_t)k * kWin;
    for (uint32_t j = tid; j < kWin; j += 1024u) {
      // window byte j is output byte start + n - 32768 + j
      uint32_t v = 0;
      if (n + j >= kWin) {  // inside this segment
        const uint32_t s = sym[start + n + j - kWin];
        v = s & 0x8000u ? (have_prev ? pw[s & 0x7fffu] : 0u) : s;
      } else if (have_prev) {
        v = pw[j + n];
      }
      nw[j] = (uint8_t)v;
      gw[j] = (uint8_t)v;
    }
    __syncthreads();
    par ^= 1u;
    have_prev = true;
  }
  if (tid == 0 && st != ZH_OK) {
    a.status[bid] = st;
    a.out_len[bid] = fail_len;
  }
}

// Symbols -> bytes; `parts` workgroups share a segment.
__global__ __launch_bounds__(256) void zh_seg_finish_kernel(uint8_t* __restrict__ d_dst, ZhInflateArgs a, ZhSegArgs g,
                                                            uint32_t parts) {
  const uint32_t k = blockIdx.x / parts, part = blockIdx.x % parts;
  if (!g.valid[k]) return;
  const uint32_t bid = g.parent[k];
  const uint64_t start = g.out_start[k], n = g.wr_len[k];
  const uint16_t* sym = g.sym + g.sym_base[bid] + start;
  uint8_t* dst = d_dst + a.bufs[bid].dst_off + start;
  const uint32_t pk = g.prev[k];
  const uint8_t* pw = pk == 0xffffffffu ? nullptr : g.windows + (size_t)pk * kWin;
  for (uint64_t i = (uint64_t)part * 256u + threadIdx.x; i < n; i += (uint64_t)parts * 256u) {
    const uint32_t s = sym[i];
    dst[i] = (uint8_t)(s & 0x8000u ? (pw ? pw[s & 0x7fffu] : 0u) : s);
  }
}

extern "C" void zh_launch_seg_find(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nsegs) return;
  hipLaunchKernelGGL(zh_seg_find_kernel, dim3(g.nsegs), dim3(kFindThreads), 0, stream, d_src, a, g);
}
extern "C" void zh_launch_seg_chain(hipStream_t stream, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nstreams) return;
  hipLaunchKernelGGL(zh_seg_chain_kernel, dim3((g.nstreams + 63u) / 64u), dim3(64), 0, stream, a, g);
}
extern "C" void zh_launch_seg_windows(hipStream_t stream, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nstreams) return;
  hipLaunchKernelGGL(zh_seg_windows_kernel, dim3(g.nstreams), dim3(1024), 0, stream, a, g);
}
extern "C" void zh_launch_seg_finish(hipStream_t stream, uint8_t* d_dst, ZhInflateArgs a, ZhSegArgs g) {
  if (!g.nsegs) return;
  const uint32_t parts = g.nsegs >= 2048u ? 2u : g.nsegs >= 512u ? 8u : 32u;
  hipLaunchKernelGGL(zh_seg_finish_kernel, dim3(g.nsegs * parts), dim3(256), 0, stream, d_dst, a, g, parts);
}
