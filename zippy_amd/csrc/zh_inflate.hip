// Batched inflate: container unwrap + RFC 1951 decode, one 64-lane wave per
// stream (a foreign deflate stream has no block index, so the only parallelism
// across a stream is lane-level; throughput comes from thousands of streams).
//
// Replaces src/zippy.nim:100-165 (format detect / zlib header), gzip.nim:3-88
// (gzip header + trailer), inflate.nim:24-291 (Huffman tables, decode loop,
// stored blocks) and the BitStreamReader of bitstreams.nim:22-82.
//
// Per wave, in LDS (22 KiB, 7 waves per CU -- a wave alone on its SIMD issues one
// instruction per ~4.4 cycles, so resident waves are what buys throughput): the last
// 16 KiB of output as a ring (LZ copies that reach further back, a few percent, re-read
// the already written-back output through L2), a 10-bit literal/length LUT and an 8-bit distance
// LUT of self-describing 32-bit entries (inflate.nim's 9-bit `fast` table,
// re-shaped), and the canonical slow-path arrays (firstCode / firstSymbol /
// maxCodes / values, inflate.nim:14-19).  The compressed stream is held 512
// bytes at a time in two VGPRs per lane and fed to a wave-uniform 64-bit bit
// buffer with v_readlane, so the per-symbol chain is one LDS lookup.  The decode
// state lives in scalar registers; the 64 lanes cooperate on LZ copies, stored
// block copies, table construction and the coalesced write-back of the window.
// Algorithmic traffic: compressed bytes read once, output written once.
#include "zh_common.h"
#include "zh_tables.h"


namespace {

__constant__ zh::LenTables c_len = zh::make_len_tables();
__constant__ zh::DistTables c_dist = zh::make_dist_tables();
__constant__ uint8_t c_clcl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

constexpr uint32_t kLitBits = 10, kDistBits = 8;
constexpr uint32_t kRing = 16384;       // bytes of recent output kept in LDS
constexpr uint32_t kFlushChunk = 4096;  // write-back granularity

}  // namespace

// ---------------------------------------------------------------------------
// Container unwrap: one thread per stream.
// ---------------------------------------------------------------------------
__global__ void zh_unwrap_kernel(const uint8_t* __restrict__ d_src, ZhInflateArgs a) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nbufs) return;
  const ZhBufDesc b = a.bufs[i];
  const uint8_t* src = d_src + b.src_off;
  const uint64_t len = a.src_len_dev ? a.src_len_dev[i] : b.src_len;
  int fmt = a.data_format;
  int st = ZH_OK;
  uint32_t pos = 0, sum = 0, isize = 0;

  if (fmt == ZH_DF_DETECT) {  // zippy.nim:108-125
    if (len > 18 && src[0] == 31 && src[1] == 139 && src[2] == 8 && (src[3] & 0xe0) == 0)
      fmt = ZH_DF_GZIP;
    else if (len > 6 && (src[0] & 0x0f) == 8 && (src[0] >> 4) <= 7 &&
             (((uint32_t)src[0] * 256u) + src[1]) % 31u == 0)
      fmt = ZH_DF_ZLIB;
    else
      st = ZH_ERR_DETECT;
  }
  if (st == ZH_OK && fmt == ZH_DF_GZIP) {  // gzip.nim:9-66
    if (len < 18) {
      st = ZH_ERR_INVALID_BUFFER;
    } else {
      const uint8_t flg = src[3];
      if (src[0] != 31 || src[1] != 139) st = ZH_ERR_GZIP_ID;
      else if (src[2] != 8) st = ZH_ERR_UNSUPPORTED_METHOD;
      else if (flg & 0xe0) st = ZH_ERR_RESERVED_FLAGS;
      else if (flg & 4) st = ZH_ERR_UNSUPPORTED_FLAGS;  // FEXTRA
      uint64_t p = 10;
      for (int field = 0; field < 2 && st == ZH_OK; field++) {  // FNAME, FCOMMENT
        if (flg & (field == 0 ? 8 : 16)) {
          while (p < len && src[p] != 0) p++;
          if (p >= len) st = ZH_ERR_INVALID_BUFFER;
          p++;
        }
      }
      if (st == ZH_OK && (flg & 2)) {  // FHCRC: skipped, not verified (gzip.nim:55-59)
        if (p + 2 >= len) st = ZH_ERR_INVALID_BUFFER;
        p += 2;
      }
      if (st == ZH_OK && p + 8 >= len) st = ZH_ERR_INVALID_BUFFER;
      if (st == ZH_OK) {
        pos = (uint32_t)p;
        const uint8_t* t = src + len - 8;
        sum = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
      }
    }
  } else if (st == ZH_OK && fmt == ZH_DF_ZLIB) {  // zippy.nim:130-150
    if (len < 6) {
      st = ZH_ERR_INVALID_BUFFER;
    } else {
      const uint8_t cmf = src[0], flg = src[1];
      if ((cmf & 0x0f) != 8) st = ZH_ERR_UNSUPPORTED_METHOD;
      else if ((cmf >> 4) > 7) st = ZH_ERR_COMPRESSION_INFO;
      else if ((((uint32_t)cmf * 256u) + flg) % 31u != 0) st = ZH_ERR_INVALID_HEADER;
      else if (flg & 0x20) st = ZH_ERR_PRESET_DICT;
      pos = 2;
      const uint8_t* t = src + len - 4;
      sum = ((uint32_t)t[0] << 24) | (t[1] << 16) | (t[2] << 8) | t[3];
    }
  } else if (st == ZH_OK && fmt != ZH_DF_DEFLATE) {
    st = ZH_ERR_INVALID_FORMAT;
  }
  a.body_pos[i] = pos;
  a.fmt[i] = (uint32_t)fmt;
  a.expect_sum[i] = sum;
  a.expect_isize[i] = isize;
  a.status[i] = st;
  a.out_len[i] = 0;
}


// ---------------------------------------------------------------------------
// Decode tables (inflate.nim:24-65 initHuffman), built by the whole wave.
//
// LUT entries are self-describing 32-bit words so that the decode loop needs
// one LDS lookup per code and no second table for base/extra values:
//   bits 0-3   code length in bits (0 = not in this table: take the slow path)
//   bits 4-7   number of extra bits that follow the code
//   bits 8-9   kind: 0 literal, 1 length (or any distance), 2 end of block, 3 invalid symbol
//   bit  15    set for literals (single-bit test on the hot path)
//   bits 16-31 literal byte / base length / base distance
// ---------------------------------------------------------------------------
namespace {

enum { kKindLit = 0, kKindBase = 1, kKindEob = 2, kKindBad = 3 };

struct HuffTab {
  uint16_t first_code[16];
  uint16_t first_symbol[16];
  uint32_t max_codes[17];
};

__device__ __forceinline__ uint32_t litlen_entry(uint32_t sym, uint32_t len) {
  if (sym < 256) return len | (kKindLit << 8) | 0x8000u | (sym << 16);
  if (sym == 256) return len | (kKindEob << 8);
  if (sym < 286) {  // inflate.nim:199-209
    const uint32_t li = sym - 257;
    return len | ((uint32_t)c_len.extra[li] << 4) | (kKindBase << 8) | ((uint32_t)c_len.base[li] << 16);
  }
  return len | (kKindBad << 8);  // 286, 287 and the 0xffff "unassigned code" marker
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len) {
  if (sym < 30)  // inflate.nim:210-222
    return len | ((uint32_t)c_dist.extra[sym] << 4) | (kKindBase << 8) | ((uint32_t)c_dist.base[sym] << 16);
  return len | (kKindBad << 8);
}
__device__ __forceinline__ uint32_t cl_entry(uint32_t sym, uint32_t len) { return len | (sym << 16); }

// lens[0..n): code lengths in LDS.  kind selects the entry encoder (0 litlen, 1 distance,
// 2 code-length alphabet).  Returns ZH_OK or ZH_ERR_INVALID_BUFFER (over-subscribed;
// incomplete codes are accepted like the reference).
__device__ int build_table(const uint8_t* lens, uint32_t n, uint32_t* lut, uint32_t lut_bits,
                           int kind, HuffTab* tab, uint16_t* values, uint32_t* s_cnt) {
  const unsigned lane = zh_lane();
  zh_wave_sync();
  if (lane < 16) s_cnt[lane] = 0;
  for (uint32_t k = lane; k < (1u << lut_bits); k += 64) lut[k] = 0;
  zh_wave_sync();
  for (uint32_t s = lane; s < n; s += 64) {
    uint32_t l = lens[s];
    if (l) atomicAdd(&s_cnt[l], 1u);
  }
  zh_wave_sync();
  // inflate.nim:32-51 (uniform; every lane computes the same values)
  uint32_t next_code[16];
  uint32_t code = 0, k = 0;
  int bad = 0;
  next_code[0] = 0;
#pragma unroll
  for (int i = 1; i < 16; i++) {
    uint32_t h = s_cnt[i];
    if (h > (1u << i)) bad = 1;
    next_code[i] = code;
    if (lane == 0) {
      tab->first_code[i] = (uint16_t)code;
      tab->first_symbol[i] = (uint16_t)k;
    }
    code += h;
    if (h > 0 && code - 1 >= (1u << i)) bad = 1;
    if (lane == 0) tab->max_codes[i] = code << (16 - i);
    code <<= 1;
    k += h;
  }
  if (lane == 0) tab->max_codes[16] = 1u << 16;
  if (bad) return ZH_ERR_INVALID_BUFFER;
  zh_wave_sync();

  // canonical code of each symbol = first code of its length + rank among the
  // symbols of that length in symbol order (ballot + popcount instead of the
  // reference's serial nextCode[len]++ walk, inflate.nim:53-65)
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t s = base + lane;
    const uint32_t l = s < n ? lens[s] : 0;
    uint32_t my_code = 0;
#pragma unroll
    for (int L = 1; L < 16; L++) {
      const uint64_t m = __ballot(l == (uint32_t)L);
      if (l == (uint32_t)L) my_code = next_code[L] + (uint32_t)__popcll(m & zh_lanemask_lt());
      next_code[L] += (uint32_t)__popcll(m);
    }
    if (l) {
      values[my_code - tab->first_code[l] + tab->first_symbol[l]] = (uint16_t)s;
      if (l <= lut_bits) {
        const uint32_t entry = kind == 0 ? litlen_entry(s, l) : kind == 1 ? dist_entry(s, l) : cl_entry(s, l);
        for (uint32_t kk = __brev(my_code) >> (32 - l); kk < (1u << lut_bits); kk += 1u << l)
          lut[kk] = entry;
      }
    }
  }
  zh_wave_sync();
  return ZH_OK;
}

}  // namespace

__global__ __launch_bounds__(64) void zh_inflate_kernel(const uint8_t* __restrict__ d_src,
                                                        uint8_t* __restrict__ d_dst,
                                                        ZhInflateArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kRing];
  __shared__ uint32_t s_lit[1u << kLitBits];
  __shared__ uint32_t s_dst[1u << kDistBits];  // also hosts the 7-bit code-length table
  __shared__ HuffTab s_tab_lit, s_tab_dist, s_tab_cl;
  __shared__ uint16_t s_val_lit[288], s_val_dist[32], s_val_cl[20];
  __shared__ uint8_t s_lens[320 + 16];
  __shared__ uint32_t s_cnt[16];

  const unsigned lane = zh_lane();
  const uint32_t sid = blockIdx.x;
  if (a.status[sid] != ZH_OK) return;  // unwrap already failed this stream

  const ZhBufDesc bd = a.bufs[sid];
  const uint8_t* src = d_src + bd.src_off;
  const uint64_t src_len = a.src_len_dev ? a.src_len_dev[sid] : bd.src_len;
  uint8_t* dst = d_dst + bd.dst_off;
  const uint64_t cap = bd.dst_cap;
  const int count_only = a.count_only;
  const bool dst_al16 = (((uintptr_t)dst) & 15u) == 0;

  // ---- input: two 256-byte windows of the stream held in registers (one dword per
  // lane), addressed relative to the 4-byte aligned base below `src`; the bit buffer
  // is topped up 32 aligned bits at a time with v_readlane (bitstreams.nim:22-49) ----
  const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
  const uint32_t* asrc = reinterpret_cast<const uint32_t*>(src - mis);
  const uint64_t end = mis + src_len;  // first byte offset (from asrc) past the stream
  auto load_dword = [&](uint64_t off) -> uint32_t {  // bytes past the end read as zero
    if (off >= end) return 0u;
    uint32_t v = asrc[off >> 2];
    if (off + 4 > end) v &= (1u << (8 * (uint32_t)(end - off))) - 1u;
    return v;
  };
  uint64_t buf = 0;   // bit buffer (LSB first)
  int32_t cnt = 0;    // valid bits in buf
  uint64_t wbase = 0; // byte offset (from asrc) of lane 0 of wcur
  uint32_t widx = 0;  // next dword of wcur to consume
  uint32_t wcur = 0, wnxt = 0;
  auto seek = [&](uint64_t sb) {  // restart the bit reader at byte offset sb (from asrc)
    wbase = sb & ~(uint64_t)255;
    wcur = load_dword(wbase + 4 * lane);
    wnxt = load_dword(wbase + 256 + 4 * lane);
    widx = (uint32_t)(sb - wbase) >> 2;
    const uint32_t w = __builtin_amdgcn_readlane(wcur, widx);
    widx++;
    buf = (uint64_t)(w >> (8 * ((uint32_t)sb & 3u)));
    cnt = 32 - 8 * (int32_t)((uint32_t)sb & 3u);
  };
  auto refill = [&]() {  // guarantees cnt > 32
    if (cnt <= 32) {
      if (widx == 64) {
        wcur = wnxt;
        wbase += 256;
        wnxt = load_dword(wbase + 256 + 4 * lane);
        widx = 0;
      }
      const uint32_t w = __builtin_amdgcn_readlane(wcur, widx);
      widx++;
      buf |= (uint64_t)w << cnt;
      cnt += 32;
    }
  };
  // bits consumed so far (from asrc); the role of `bitsBuffered < 0`
  auto past_end = [&]() -> bool { return (wbase + 4ull * widx) * 8 - (uint64_t)cnt > end * 8; };
  auto take = [&](uint32_t nbits) -> uint32_t {
    const uint32_t v = (uint32_t)buf & ((1u << nbits) - 1u);
    buf >>= nbits;
    cnt -= (int32_t)nbits;
    return v;
  };
  // inflate.nim:67-91 decodeSymbolSlow for codes longer than the LUT; returns the
  // symbol (0xffff = unassigned code) and consumes its bits
  auto decode_slow = [&](uint32_t lut_bits, const HuffTab* tab, const uint16_t* values) -> uint32_t {
    const uint32_t k = __brev((uint32_t)buf) >> 16;
    uint32_t cl = lut_bits + 1;
    while (cl < 16 && k >= zh_bcast(tab->max_codes[cl])) cl++;
    if (cl >= 16) return 0xffffu;
    const uint32_t id = ((k >> (16 - cl)) - zh_bcast(tab->first_code[cl]) +
                         zh_bcast(tab->first_symbol[cl])) & 0xffffu;
    take(cl);
    return zh_bcast(values[id]);
  };

  seek((uint64_t)mis + a.body_pos[sid]);

  uint64_t op = 0;          // bytes produced (including the pending literals' predecessors)
  uint32_t unflushed = 0;   // op - (bytes already written back to HBM)
  int st = ZH_OK;
  uint64_t pend = 0;        // up to 8 decoded literals not yet stored in the ring
  uint32_t psh = 0;         // 8 * number of pending literals

  // write the oldest `nbytes` unflushed ring bytes back to HBM (nbytes <= unflushed <= kRing)
  auto flush = [&](uint32_t nbytes) {
    zh_wave_sync();
    if (!count_only) {
      const uint64_t from = op - unflushed;
      const uint64_t upto = from + nbytes;
      uint64_t p = from;
      if (dst_al16) {  // `from` is a multiple of kFlushChunk here (chunks go out whole)
        for (; p + 1024 <= upto; p += 1024) {
          const uint64_t q = p + lane * 16u;
          *reinterpret_cast<uint4*>(dst + q) = *reinterpret_cast<const uint4*>(&s_win[q & (kRing - 1u)]);
        }
      }
      for (uint64_t q = p + lane; q < upto; q += 64) dst[q] = s_win[q & (kRing - 1u)];
    }
    unflushed -= nbytes;
    zh_wave_sync();
  };
  // store the pending literals in the ring; checks that are only needed now and then
  // (end of input, slot capacity, write-back) ride along here
  auto flush_pend = [&]() {
    const uint32_t npend = psh >> 3;
    if (npend) {
      if (!count_only && lane < npend) s_win[(op + lane) & (kRing - 1u)] = (uint8_t)(pend >> (8 * lane));
      op += npend;
      unflushed += npend;
      psh = 0;
      pend = 0;
    }
    if (unflushed >= kFlushChunk + 2048) {
      if (past_end()) st = ZH_ERR_END_OF_BUFFER;
      else if (op > cap && !count_only) st = ZH_ERR_DST_TOO_SMALL;
      else flush(kFlushChunk);
    }
  };

  bool final_block = false;
  while (!final_block && st == ZH_OK) {  // inflate.nim:273-289
    refill();
    const uint32_t bfinal = take(1), btype = take(2);
    if (bfinal) final_block = true;

    if (btype == 0) {  // inflate.nim:252-266 inflateNoCompression
      take((uint32_t)cnt & 7u);
      refill();
      const uint32_t len = take(16), nlen = take(16);
      if (len + nlen != 65535u) { st = ZH_ERR_INVALID_BUFFER; break; }
      const uint64_t byte_pos = wbase + 4ull * widx - (uint64_t)(cnt >> 3);  // from asrc
      if (byte_pos + len > end) { st = ZH_ERR_END_OF_BUFFER; break; }
      if (op + len > cap && !count_only) { st = ZH_ERR_DST_TOO_SMALL; break; }
      const uint8_t* raw = reinterpret_cast<const uint8_t*>(asrc) + byte_pos;
      for (uint32_t done = 0; done < len;) {
        const uint32_t n = len - done < kFlushChunk ? len - done : kFlushChunk;
        zh_wave_sync();
        if (!count_only)
          for (uint32_t i = lane; i < n; i += 64) s_win[(op + i) & (kRing - 1u)] = raw[done + i];
        op += n;
        unflushed += n;
        done += n;
        if (unflushed >= kFlushChunk + 2048) flush(kFlushChunk);
      }
      seek(byte_pos + len);
      continue;
    }
    if (btype == 3) { st = ZH_ERR_BLOCK_HEADER; break; }

    uint32_t hlit = 288, hdist = 30;
    if (btype == 1) {  // fixed codes, inflate.nim:111-113 (rebuilt per block like the reference)
      zh_wave_sync();
      for (uint32_t s = lane; s < 288; s += 64) s_lens[s] = (uint8_t)(s <= 143 ? 8 : s <= 255 ? 9 : s <= 279 ? 7 : 8);
      if (lane < 30) s_lens[288 + lane] = 5;
    } else {  // dynamic header, inflate.nim:115-171
      hlit = take(5) + 257;
      hdist = take(5) + 1;
      const uint32_t hclen = take(4) + 4;
      if (hlit > 286 || hdist > 30) { st = ZH_ERR_INVALID_BUFFER; break; }
      zh_wave_sync();
      if (lane < 20) s_lens[lane] = 0;
      zh_wave_sync();
      for (uint32_t i = 0; i < hclen; i++) {
        refill();
        const uint32_t v = take(3);
        if (lane == 0) s_lens[c_clcl_order[i]] = (uint8_t)v;
      }
      st = build_table(s_lens, 19, s_dst, 7, 2, &s_tab_cl, s_val_cl, s_cnt);
      if (st != ZH_OK) break;
      uint32_t i = 0;
      const uint32_t total = hlit + hdist;
      uint32_t prev = 0;
      while (i != total) {
        refill();
        uint32_t sym;
        const uint32_t e = zh_bcast(s_dst[(uint32_t)buf & 127u]);
        if (e) {
          take(e & 15u);
          sym = e >> 16;
        } else {
          sym = decode_slow(7, &s_tab_cl, s_val_cl);
        }
        if (past_end()) { st = ZH_ERR_END_OF_BUFFER; break; }
        if (sym <= 15) {
          if (lane == 0) s_lens[i] = (uint8_t)sym;
          prev = sym;
          i++;
        } else if (sym == 16) {
          if (i == 0) { st = ZH_ERR_INVALID_BUFFER; break; }
          const uint32_t rep = take(2) + 3;
          if (i + rep > 320) { st = ZH_ERR_INVALID_BUFFER; break; }
          if (lane < rep) s_lens[i + lane] = (uint8_t)prev;
          i += rep;
        } else if (sym == 17 || sym == 18) {
          const uint32_t rep = sym == 17 ? take(3) + 3 : take(7) + 11;
          for (uint32_t j = lane; j < rep && i + j < 320 + 16; j += 64) s_lens[i + j] = 0;
          prev = 0;
          i += rep;
        } else {
          st = ZH_ERR_INVALID_SYMBOL;
          break;
        }
        if (i > total) { st = ZH_ERR_INVALID_BUFFER; break; }
      }
      if (st != ZH_OK) break;
    }
    {
      const uint32_t dist_at = btype == 1 ? 288u : hlit;
      st = build_table(s_lens, hlit, s_lit, kLitBits, 0, &s_tab_lit, s_val_lit, s_cnt);
      if (st != ZH_OK) break;
      st = build_table(s_lens + dist_at, hdist, s_dst, kDistBits, 1, &s_tab_dist, s_val_dist, s_cnt);
      if (st != ZH_OK) break;
    }

    for (;;) {  // inflate.nim:173-250
      refill();
      uint32_t e = zh_bcast(s_lit[(uint32_t)buf & ((1u << kLitBits) - 1u)]);
      if (e & 0x8000u) {  // literal straight out of the LUT: the hot path
        const uint32_t nb = e & 15u;
        buf >>= nb;
        cnt -= (int32_t)nb;
        pend |= (uint64_t)(e >> 16) << psh;
        psh += 8;
        if (psh == 64) {
          flush_pend();
          if (st != ZH_OK) break;
        }
        continue;
      }
      if (e == 0) {  // longer than the LUT, or unassigned
        const uint32_t sym = decode_slow(kLitBits, &s_tab_lit, s_val_lit);
        e = litlen_entry(sym, 0);
        if (e & 0x8000u) {
          pend |= (uint64_t)(e >> 16) << psh;
          psh += 8;
          if (psh == 64) {
            flush_pend();
            if (st != ZH_OK) break;
          }
          continue;
        }
      } else {
        take(e & 15u);
      }
      // literals decoded from beyond the end of the input are caught here at the latest
      if (past_end()) { st = ZH_ERR_END_OF_BUFFER; break; }
      flush_pend();
      if (st != ZH_OK) break;
      const uint32_t kind = (e >> 8) & 3u;
      if (kind == kKindEob) break;
      if (kind == kKindBad) { st = ZH_ERR_INVALID_BUFFER; break; }  // inflate.nim:202-204
      const uint32_t length = (e >> 16) + take((e >> 4) & 15u);
      refill();
      uint32_t de = zh_bcast(s_dst[(uint32_t)buf & ((1u << kDistBits) - 1u)]);
      if (de == 0) {
        const uint32_t dsym = decode_slow(kDistBits, &s_tab_dist, s_val_dist);
        de = dist_entry(dsym, 0);
      } else {
        take(de & 15u);
      }
      if (((de >> 8) & 3u) == kKindBad) { st = ZH_ERR_INVALID_BUFFER; break; }  // inflate.nim:211-213
      const uint32_t dist = (de >> 16) + take((de >> 4) & 15u);
      if (dist > op) { st = ZH_ERR_INVALID_BUFFER; break; }  // inflate.nim:224-225
      if (!count_only) {
        // inflate.nim:227-250: byte-sequential LZ77 copy semantics; an overlapping copy
        // (dist < length) repeats the dist-byte pattern, so every lane reads its source
        // from the region that is already written.
        zh_wave_sync();
        if (dist > kRing) {
          // source older than the ring: it was written back at least kRing - kFlushChunk -
          // 2048 - 258 bytes ago.  Wait for those stores, then read through L2 (this CU's
          // L1 may hold a stale copy of a partially written line).
          if (op + length > cap) { st = ZH_ERR_DST_TOO_SMALL; break; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          for (uint32_t i = lane; i < length; i += 64)
            s_win[(op + i) & (kRing - 1u)] =
                __hip_atomic_load(dst + (op - dist + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (dist + length > kRing) {
          // the destination wraps onto the source's ring slots: each 64-byte group must be
          // read before the next group is written (program order on the GPU; the explicit
          // syncs keep the CPU emulator, whose lanes are not in lockstep, honest)
          for (uint32_t base = 0; base < length; base += 64) {
            const uint32_t i = base + lane;
            const uint8_t v = s_win[(op - dist + i) & (kRing - 1u)];
            zh_wave_sync();
            if (i < length) s_win[(op + i) & (kRing - 1u)] = v;
            zh_wave_sync();
          }
        } else if (dist >= length) {
          for (uint32_t i = lane; i < length; i += 64)
            s_win[(op + i) & (kRing - 1u)] = s_win[(op - dist + i) & (kRing - 1u)];
        } else if (dist == 1) {
          const uint8_t v = s_win[(op - 1) & (kRing - 1u)];
          for (uint32_t i = lane; i < length; i += 64) s_win[(op + i) & (kRing - 1u)] = v;
        } else {
          for (uint32_t i = lane; i < length; i += 64)
            s_win[(op + i) & (kRing - 1u)] = s_win[(op - dist + i % dist) & (kRing - 1u)];
        }
      }
      op += length;
      unflushed += length;
    }
  }

  if (st == ZH_OK) {
    if (op > cap && !count_only) st = ZH_ERR_DST_TOO_SMALL;
    else flush(unflushed);
  }
  if (lane == 0) {
    a.out_len[sid] = op;
    a.status[sid] = st;
  }
}

// Final check of each stream against its trailer (gzip.nim:80-88, zippy.nim:152-162).
__global__ void zh_verify_kernel(ZhInflateArgs a, const uint32_t* __restrict__ buf_crc,
                                 const uint32_t* __restrict__ buf_adler) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nbufs) return;
  if (a.status[i] != ZH_OK) return;
  const uint32_t fmt = a.fmt[i];
  if (fmt == ZH_DF_GZIP) {
    if (a.expect_sum[i] != buf_crc[i]) a.status[i] = ZH_ERR_CHECKSUM;
    else if (a.expect_isize[i] != (uint32_t)(a.out_len[i] & 0xffffffffu)) a.status[i] = ZH_ERR_SIZE;
  } else if (fmt == ZH_DF_ZLIB) {
    if (a.expect_sum[i] != buf_adler[i]) a.status[i] = ZH_ERR_CHECKSUM;
  }
}

extern "C" void zh_launch_unwrap(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a) {
  if (!a.nbufs) return;
  hipLaunchKernelGGL(zh_unwrap_kernel, dim3((a.nbufs + 63) / 64), dim3(64), 0, stream, d_src, a);
}
extern "C" void zh_launch_inflate(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst,
                                  ZhInflateArgs a) {
  if (!a.nbufs) return;
  hipLaunchKernelGGL(zh_inflate_kernel, dim3(a.nbufs), dim3(64), 0, stream, d_src, d_dst, a);
}
extern "C" void zh_launch_verify(hipStream_t stream, ZhInflateArgs a, const uint32_t* buf_crc,
                                 const uint32_t* buf_adler) {
  if (!a.nbufs) return;
  hipLaunchKernelGGL(zh_verify_kernel, dim3((a.nbufs + 63) / 64), dim3(64), 0, stream, a, buf_crc,
                     buf_adler);
}
