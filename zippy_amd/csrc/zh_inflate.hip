// Batched inflate: container unwrap + RFC 1951 decode, one 64-lane wave per
// stream (a foreign deflate stream has no block index, so the only parallelism
// across a stream is lane-level; throughput comes from thousands of streams).
//
// Replaces src/zippy.nim:100-165 (format detect / zlib header), gzip.nim:3-88
// (gzip header + trailer), inflate.nim:24-291 (Huffman tables, decode loop,
// stored blocks) and the BitStreamReader of bitstreams.nim:22-82.
//
// Per wave, in LDS: the 32 KiB output window as a ring (LZ copies never touch
// HBM), a 4 KiB sliding window of the input stream, a 10-bit literal/length
// LUT and a 9-bit distance LUT (inflate.nim's 9-bit `fast` table widened), and
// the canonical slow-path arrays (firstCode / firstSymbol / maxCodes / values,
// inflate.nim:14-19).  The decode state (bit buffer, positions) is wave-uniform
// and lives in scalar registers; the 64 lanes cooperate on LZ copies, stored
// block copies, table construction and the coalesced write-back of the window.
// Algorithmic traffic: compressed bytes read once, output written once.
#include "zh_common.h"
#include "zh_tables.h"


namespace {

__constant__ zh::LenTables c_len = zh::make_len_tables();
__constant__ zh::DistTables c_dist = zh::make_dist_tables();
__constant__ uint8_t c_clcl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

constexpr uint32_t kLitBits = 10, kDistBits = 9;
constexpr uint32_t kInWin = 4096;  // bytes of input staged in LDS

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
// ---------------------------------------------------------------------------
namespace {

struct HuffTab {
  uint16_t first_code[16];
  uint16_t first_symbol[16];
  uint32_t max_codes[17];
};

// lens[0..n): code lengths in LDS.  lut: 1 << lut_bits entries of len << 12 | symbol.
// Returns ZH_OK or ZH_ERR_INVALID_BUFFER (over-subscribed; incomplete codes are
// accepted like the reference).
__device__ int build_table(const uint8_t* lens, uint32_t n, uint16_t* lut, uint32_t lut_bits,
                           HuffTab* tab, uint16_t* values, uint32_t* s_cnt) {
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
        const uint16_t entry = (uint16_t)((l << 12) | s);
        for (uint32_t kk = __brev(my_code) >> (32 - l); kk < (1u << lut_bits); kk += 1u << l)
          lut[kk] = entry;
      }
    }
  }
  zh_wave_sync();
  return ZH_OK;
}

// Wave-uniform bit reader over the LDS input window (bitstreams.nim:22-62).
struct BitReader {
  uint64_t buf;
  int32_t cnt;        // valid bits in buf
  uint64_t in_pos;    // stream byte offset of the next byte to load
  uint64_t win_base;  // stream byte offset of s_in[0]
};

}  // namespace

__global__ __launch_bounds__(64) void zh_inflate_kernel(const uint8_t* __restrict__ d_src,
                                                        uint8_t* __restrict__ d_dst,
                                                        ZhInflateArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[32768];
  __shared__ __attribute__((aligned(16))) uint32_t s_in[kInWin / 4 + 4];
  __shared__ uint16_t s_lit[1u << kLitBits];
  __shared__ uint16_t s_dst[1u << kDistBits];
  __shared__ uint16_t s_clt[128];
  __shared__ HuffTab s_tab_lit, s_tab_dist, s_tab_cl;
  __shared__ uint16_t s_val_lit[288], s_val_dist[32], s_val_cl[20];
  __shared__ uint8_t s_lens[320 + 16];
  __shared__ uint32_t s_cnt[16];
  __shared__ uint32_t s_lenbase[32], s_distbase[32];

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

  if (lane < 29) s_lenbase[lane] = c_len.base[lane] | ((uint32_t)c_len.extra[lane] << 16);
  if (lane < 30) s_distbase[lane] = c_dist.base[lane] | ((uint32_t)c_dist.extra[lane] << 16);

  BitReader br;
  br.buf = 0;
  br.cnt = 0;
  br.in_pos = a.body_pos[sid];
  br.win_base = br.in_pos & ~(uint64_t)3;

  // (re)load the whole LDS input window starting at br.win_base (zero past the end)
  auto load_window = [&](uint32_t from_word) {
    zh_wave_sync();
    for (uint32_t w = from_word + lane; w < kInWin / 4 + 4; w += 64) {
      const uint64_t p = br.win_base + (uint64_t)w * 4;
      uint32_t v = 0;
      if (p + 4 <= src_len) {
        v = src[p] | (src[p + 1] << 8) | (src[p + 2] << 16) | ((uint32_t)src[p + 3] << 24);
      } else {
        for (int j = 0; j < 4; j++)
          if (p + j < src_len) v |= (uint32_t)src[p + j] << (8 * j);
      }
      s_in[w] = v;
    }
    zh_wave_sync();
  };
  load_window(0);

  auto refill = [&]() {  // guarantees cnt > 32
    if (br.cnt <= 32) {
      uint32_t off = (uint32_t)(br.in_pos - br.win_base);
      if (off + 4 > kInWin) {  // slide: keep the upper half, fetch 2 KiB more
        zh_wave_sync();
        uint32_t keep[8];
        for (int j = 0; j < 8; j++) keep[j] = s_in[kInWin / 8 + j * 64 + lane];
        zh_wave_sync();
        for (int j = 0; j < 8; j++) s_in[j * 64 + lane] = keep[j];
        br.win_base += kInWin / 2;
        load_window(kInWin / 8);
        off -= kInWin / 2;
      }
      const uint32_t w = zh_bcast(zh_ld32(s_in, off));
      br.buf |= (uint64_t)w << br.cnt;
      br.cnt += 32;
      br.in_pos += 4;
    }
  };
  auto consumed_past_end = [&]() -> bool {  // the role of `bitsBuffered < 0`
    return br.in_pos * 8 - (uint64_t)br.cnt > src_len * 8;
  };
  auto take = [&](uint32_t nbits) -> uint32_t {
    uint32_t v = (uint32_t)br.buf & ((1u << nbits) - 1u);
    br.buf >>= nbits;
    br.cnt -= (int32_t)nbits;
    return v;
  };
  // inflate.nim:67-102 decodeSymbol / decodeSymbolSlow; 0xffff = unassigned code
  auto decode = [&](const uint16_t* lut, uint32_t lut_bits, const HuffTab* tab,
                    const uint16_t* values) -> uint32_t {
    const uint32_t e = zh_bcast(lut[(uint32_t)br.buf & ((1u << lut_bits) - 1u)]);
    if (e) {
      take(e >> 12);
      return e & 0xfffu;
    }
    const uint32_t k = __brev((uint32_t)br.buf) >> 16;
    uint32_t cl = lut_bits + 1;
    while (cl < 16 && k >= zh_bcast(tab->max_codes[cl])) cl++;
    if (cl >= 16) return 0xffffu;
    const uint32_t id = ((k >> (16 - cl)) - zh_bcast(tab->first_code[cl]) +
                         zh_bcast(tab->first_symbol[cl])) & 0xffffu;
    take(cl);
    return zh_bcast(values[id]);
  };

  uint64_t op = 0, flushed = 0;
  int st = ZH_OK;

  // write ring bytes [flushed, upto) back to HBM; upto - flushed <= 32768
  auto flush = [&](uint64_t upto) {
    zh_wave_sync();
    if (!count_only) {
      uint64_t p = flushed;
      if (dst_al16) {
        for (; p + 1024 <= upto; p += 1024) {
          const uint64_t q = p + lane * 16u;  // flushed is always a multiple of 16 here
          *reinterpret_cast<uint4*>(dst + q) = *reinterpret_cast<const uint4*>(&s_win[q & 32767u]);
        }
      }
      for (uint64_t q = p + lane; q < upto; q += 64) dst[q] = s_win[q & 32767u];
    }
    flushed = upto;
    zh_wave_sync();
  };

  bool final_block = false;
  while (!final_block && st == ZH_OK) {  // inflate.nim:273-289
    refill();
    const uint32_t bfinal = take(1), btype = take(2);
    if (bfinal) final_block = true;

    if (btype == 0) {  // inflate.nim:252-266 inflateNoCompression
      take((uint32_t)br.cnt & 7u);
      refill();
      const uint32_t len = take(16), nlen = take(16);
      if (len + nlen != 65535u) { st = ZH_ERR_INVALID_BUFFER; break; }
      const uint64_t byte_pos = br.in_pos - (uint64_t)(br.cnt >> 3);
      if (byte_pos + len > src_len) { st = ZH_ERR_END_OF_BUFFER; break; }
      if (op + len > cap && !count_only) { st = ZH_ERR_DST_TOO_SMALL; break; }
      for (uint32_t done = 0; done < len;) {
        uint32_t n = len - done < 8192u ? len - done : 8192u;
        zh_wave_sync();
        if (!count_only)
          for (uint32_t i = lane; i < n; i += 64) s_win[(op + i) & 32767u] = src[byte_pos + done + i];
        op += n;
        done += n;
        if (op - flushed >= 16384) flush(flushed + 16384);
      }
      br.buf = 0;
      br.cnt = 0;
      br.in_pos = byte_pos + len;
      br.win_base = br.in_pos & ~(uint64_t)3;
      load_window(0);
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
      st = build_table(s_lens, 19, s_clt, 7, &s_tab_cl, s_val_cl, s_cnt);
      if (st != ZH_OK) break;
      // the cl lengths sit in s_lens[0..19); unpack litlen+dist lengths after them
      uint8_t* unpacked = s_lens;  // overwritten below only after the cl table is built
      uint32_t i = 0;
      const uint32_t total = hlit + hdist;
      uint32_t prev = 0;
      while (i != total) {
        refill();
        const uint32_t sym = decode(s_clt, 7, &s_tab_cl, s_val_cl);
        if (consumed_past_end()) { st = ZH_ERR_END_OF_BUFFER; break; }
        if (sym <= 15) {
          if (lane == 0) unpacked[i] = (uint8_t)sym;
          prev = sym;
          i++;
        } else if (sym == 16) {
          if (i == 0) { st = ZH_ERR_INVALID_BUFFER; break; }
          const uint32_t rep = take(2) + 3;
          if (i + rep > 320) { st = ZH_ERR_INVALID_BUFFER; break; }
          if (lane < rep) unpacked[i + lane] = (uint8_t)prev;
          i += rep;
        } else if (sym == 17 || sym == 18) {
          const uint32_t rep = sym == 17 ? take(3) + 3 : take(7) + 11;
          for (uint32_t j = lane; j < rep && i + j < 320 + 16; j += 64) unpacked[i + j] = 0;
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
      st = build_table(s_lens, hlit, s_lit, kLitBits, &s_tab_lit, s_val_lit, s_cnt);
      if (st != ZH_OK) break;
      st = build_table(s_lens + dist_at, hdist, s_dst, kDistBits, &s_tab_dist, s_val_dist, s_cnt);
      if (st != ZH_OK) break;
    }

    for (;;) {  // inflate.nim:173-250
      refill();
      const uint32_t sym = decode(s_lit, kLitBits, &s_tab_lit, s_val_lit);
      if (consumed_past_end()) { st = ZH_ERR_END_OF_BUFFER; break; }
      if (sym <= 255) {
        if (op >= cap && !count_only) { st = ZH_ERR_DST_TOO_SMALL; break; }
        if (!count_only && lane == 0) s_win[op & 32767u] = (uint8_t)sym;
        op++;
      } else if (sym == 256) {
        break;
      } else {
        const uint32_t li = sym - 257;
        if (li >= 29) { st = ZH_ERR_INVALID_BUFFER; break; }  // inflate.nim:202-204 (and 0xffff)
        refill();
        const uint32_t lb = zh_bcast(s_lenbase[li]);
        const uint32_t length = (lb & 0xffffu) + take(lb >> 16);
        const uint32_t dsym = decode(s_dst, kDistBits, &s_tab_dist, s_val_dist);
        if (dsym >= 30) { st = ZH_ERR_INVALID_BUFFER; break; }  // inflate.nim:211-213
        const uint32_t db = zh_bcast(s_distbase[dsym]);
        const uint32_t dist = (db & 0xffffu) + take(db >> 16);
        if (dist > op) { st = ZH_ERR_INVALID_BUFFER; break; }  // inflate.nim:224-225
        if (op + length > cap && !count_only) { st = ZH_ERR_DST_TOO_SMALL; break; }
        if (!count_only) {
          // inflate.nim:227-250: byte-sequential LZ77 copy semantics; an overlapping
          // copy (dist < length) repeats the dist-byte pattern, so every lane can
          // read its source from the already written region.
          zh_wave_sync();
          const bool overlap = dist < length;
          for (uint32_t i = lane; i < length; i += 64) {
            const uint32_t si = overlap ? i % dist : i;
            const uint8_t v = s_win[(op - dist + si) & 32767u];
            s_win[(op + i) & 32767u] = v;
          }
        }
        op += length;
      }
      if (op - flushed >= 16384 + 512) flush(flushed + 16384);
    }
  }

  if (st == ZH_OK) flush(op);
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
