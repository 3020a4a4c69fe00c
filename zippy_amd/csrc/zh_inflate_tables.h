// Huffman decode tables of the inflate kernels (shared by zh_inflate.hip and zh_inflate_split.hip):
// inflate.nim:24-65 initHuffman re-shaped into self-describing LUT entries, built by one wave.
#pragma once
#include "zh_common.h"
#include "zh_tables.h"

namespace {

// (table entries are built once per block and in the rare lone-token path: tables are fine here,
// and the arithmetic forms of zh_tables.h cost the round loop registers)
__constant__ zh::LenTables c_len = zh::make_len_tables();
__constant__ zh::DistTables c_dist = zh::make_dist_tables();
__constant__ uint8_t c_clcl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

constexpr uint32_t kLitBits = 10, kDistBits = 8;
// second-level tables behind the litlen root table for codes longer than kLitBits (a complete
// 286-symbol code of maximum length 15 needs at most 308 entries with a 10-bit root)
constexpr uint32_t kLitSub = 320;
constexpr uint32_t kInWords = 128;   // staging ring of the compressed stream (dwords, power of two)

}  // namespace

// ---------------------------------------------------------------------------
// Decode tables (inflate.nim:24-65 initHuffman), built by the whole wave.
//
// LUT entries are self-describing 32-bit words so that the decode loop needs
// one LDS lookup per code and no second table for base/extra values:
//   bits 0-3   code length in bits (0 = not in this table: take the slow path)
//   bits 4-7   number of extra bits that follow the code
//   bits 8-9   kind: 0 literal, 1 length (or any distance), 2 end of block, 3 invalid symbol
//   bit  10    link: the code is longer than the root table; bits 0-3 = index bits of its
//              second-level table, bits 16-31 = where that table starts (entries of a
//              second-level table carry the code's full length)
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
                           int kind, HuffTab* tab, uint16_t* values, uint32_t* s_cnt, uint32_t sub_cap = 0) {
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
  // Codes longer than the root table: one second-level table per root prefix, like zlib's
  // inflate_table.  Canonical codes are sorted, so the codes below one prefix are consecutive in
  // `values` order; once per block and a few hundred steps at most, so one lane does it.  Patterns
  // no code claims stay 0 (= "decode alone on the canonical path", which also finds the errors).
  if (sub_cap && lane == 0) {
    const uint32_t R = lut_bits;
    auto len_of = [&](uint32_t t, uint32_t l) -> uint32_t {  // code length of canonical index t (l: a lower bound)
      while (l < 15u && t >= (uint32_t)tab->first_symbol[l] + s_cnt[l]) l++;
      return l;
    };
    auto code_of = [&](uint32_t t, uint32_t l) -> uint32_t { return (uint32_t)tab->first_code[l] + (t - tab->first_symbol[l]); };
    uint32_t next = 1u << R;
    uint32_t t = tab->first_symbol[R + 1u], tl = R + 1u;
    while (t < k) {
      tl = len_of(t, tl);
      const uint32_t p = code_of(t, tl) >> (tl - R);
      uint32_t j = t, jl = tl;  // last code below prefix p (lengths do not decrease)
      while (j + 1u < k) {
        const uint32_t l2 = len_of(j + 1u, jl);
        if ((code_of(j + 1u, l2) >> (l2 - R)) != p) break;
        j++;
        jl = l2;
      }
      const uint32_t sb = jl - R, size = 1u << sb;
      if (next + size > (1u << R) + sub_cap) break;  // no room: the rest keeps taking the slow path
      for (uint32_t q = 0; q < size; q++) lut[next + q] = 0;
      uint32_t ul = tl;
      for (uint32_t u = t; u <= j; u++) {
        ul = len_of(u, ul);
        const uint32_t low = code_of(u, ul) & ((1u << (ul - R)) - 1u);
        const uint32_t entry = kind == 0 ? litlen_entry(values[u], ul) : dist_entry(values[u], ul);
        for (uint32_t q = __brev(low) >> (32u - (ul - R)); q < size; q += 1u << (ul - R)) lut[next + q] = entry;
      }
      lut[__brev(p) >> (32u - R)] = sb | 0x400u | (next << 16);
      next += size;
      t = j + 1u;
    }
  }
  zh_wave_sync();
  return ZH_OK;
}

}  // namespace

