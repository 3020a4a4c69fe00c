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
//   bit  11    plain: a literal, a length or a distance, decoded by this entry alone -- the ONE test of the split
//              decoder's hot loop; everything rare (a link, an empty entry, end of block, an invalid symbol) has it clear
//   bit  15    set for literals (single-bit test on the hot path)
//   bits 16-31 literal byte / base length / base distance
// ---------------------------------------------------------------------------
namespace {

enum { kKindLit = 0, kKindBase = 1, kKindEob = 2, kKindBad = 3 };
constexpr uint32_t kEntryPlain = 0x800u;

struct HuffTab {
  uint16_t first_code[16];
  uint16_t first_symbol[16];
  uint32_t max_codes[17];
};

__device__ __forceinline__ uint32_t litlen_entry(uint32_t sym, uint32_t len) {
  if (sym < 256) return len | (kKindLit << 8) | kEntryPlain | 0x8000u | (sym << 16);
  if (sym == 256) return len | (kKindEob << 8);
  if (sym < 286) {  // inflate.nim:199-209
    const uint32_t li = sym - 257;
    return len | ((uint32_t)c_len.extra[li] << 4) | (kKindBase << 8) | kEntryPlain | ((uint32_t)c_len.base[li] << 16);
  }
  return len | (kKindBad << 8);  // 286, 287 and the 0xffff "unassigned code" marker
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len) {
  if (sym < 30)  // inflate.nim:210-222
    return len | ((uint32_t)c_dist.extra[sym] << 4) | (kKindBase << 8) | kEntryPlain | ((uint32_t)c_dist.base[sym] << 16);
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


// ---------------------------------------------------------------------------
// The same tables built by a whole WORKGROUP (zh_inflate_split.hip: a foreign stream of short blocks --
// system zlib's are ~ 40 KiB of input each -- spent 38 % of the tokens kernel in block headers, one wave
// at work and the others waiting at a barrier).  Not the same stores in the same order as build_table(),
// but the same function of the code lengths: a root entry per R-bit prefix -- the code of at most R bits
// that the prefix starts with, or a link to the second-level table of the longer codes below it, or 0
// ("decode alone on the canonical path", which also finds the errors) -- second-level tables in prefix
// order as long as they fit, the canonical arrays for the slow path, and ZH_ERR_INVALID_BUFFER for an
// over-subscribed code (inflate.nim:32-51).  Every root entry is written (no clearing pass), every thread
// resolves its own prefixes with the counts in registers (no serial walk over the symbols).
// ---------------------------------------------------------------------------
// entry encoders without the constant tables (a per-lane lookup there is a trip to memory)
__device__ __forceinline__ uint32_t litlen_entry_a(uint32_t sym, uint32_t len) {
  if (sym < 256) return len | (kKindLit << 8) | kEntryPlain | 0x8000u | (sym << 16);
  if (sym == 256) return len | (kKindEob << 8);
  if (sym < 286) {
    const uint32_t li = sym - 257;
    return len | (zh_len_extra_bits(li) << 4) | (kKindBase << 8) | kEntryPlain | (zh_len_base(li) << 16);
  }
  return len | (kKindBad << 8);
}
__device__ __forceinline__ uint32_t dist_entry_a(uint32_t sym, uint32_t len) {
  if (sym < 30) return len | (zh_dist_extra_bits(sym) << 4) | (kKindBase << 8) | kEntryPlain | (zh_dist_base(sym) << 16);
  return len | (kKindBad << 8);
}

constexpr uint32_t kWgScratchWords = 256u;  // build_tables_wg's scratch (dwords): 128 a table

// One table of build_tables_wg: its share of every step (the steps of both tables run between the same barriers).
// R-bit root table, SUBCAP second-level entries behind it; KIND 0 litlen (up to 288 symbols), 1 distance (up to 32).
// A canonical code is an ordered partition of the 15-bit values: the codes of l bits, left-justified, are the
// values [lim[l - 1], lim[l]) with lim[l] = (first code of l bits + their count) << (15 - l).  So the length of the
// code a bit pattern starts with is a count of comparisons, not a search -- for a root prefix (its first R bits) and
// for a second-level entry (prefix + its index bits) alike.
template <uint32_t T, uint32_t R, uint32_t SUBCAP, int KIND>
struct WgTable {
  static_assert(T % 64u == 0 && R <= 10u, "");
  static constexpr uint32_t kGroups = KIND == 0 ? 5u : 1u;          // symbols in groups of 64
  static constexpr uint32_t kSlots = (kGroups * 64u + T - 1u) / T;  // symbol slots a thread
  static constexpr uint32_t kPP = ((1u << R) + T - 1u) / T;         // prefixes a thread
  const uint8_t* lens;  // [n] code lengths (LDS)
  uint32_t n;
  uint32_t* lut;
  HuffTab* tab;
  uint16_t* values;
  uint32_t* scratch;  // 128 dwords: [0, 80) symbols of a length in a group, [80, 96) lim[] ([80]: over-subscribed), [96, ..) wave sums
  uint32_t sl[kSlots], srank[kSlots], lim[16], pl[kPP], size_sum, incl;

  // 1. every symbol's rank among the symbols of its length in its group; `vt`: the thread's number among the
  // threads that take this table's symbols (>= kGroups * 64: none)
  __device__ __forceinline__ void ranks(uint32_t vt) {
    const unsigned lane = zh_lane();
#pragma unroll
    for (uint32_t j = 0; j < kSlots; j++) {
      const uint32_t s = vt + j * T;
      sl[j] = 0;
      srank[j] = 0;
      if (s < kGroups * 64u) {  // (wave-uniform)
        const uint32_t l = s < n ? lens[s] : 0u;
        sl[j] = l;
#pragma unroll
        for (uint32_t L = 1; L < 16; L++) {
          const uint64_t m = __ballot(l == L);
          if (l == L) srank[j] = (uint32_t)__popcll(m & zh_lanemask_lt());
          if (lane == L) scratch[(s >> 6) * 16u + L] = (uint32_t)__popcll(m);
        }
      }
    }
  }
  // 2. inflate.nim:32-51: counts, first codes, first canonical indices: lane l of ONE wave for length l
  __device__ __forceinline__ void counts() {
    const unsigned lane = zh_lane();
    uint32_t h = 0;
    if (lane >= 1 && lane < 16) {
#pragma unroll
      for (uint32_t g = 0; g < kGroups; g++) h += scratch[g * 16u + lane];
    }
    // first code of length l = sum over shorter lengths l' of count[l'] << (l - l'), i.e. lim[l - 1] >> (15 - l);
    // lim[l] = sum over l' <= l of count[l'] << (15 - l')
    const uint32_t lm = zh_wave_scan(lane >= 1 && lane < 16 ? h << (15u - lane) : 0u);
    const uint32_t first_sym = zh_wave_scan(h) - h;
    const uint32_t first_code = lane >= 1 && lane < 16 ? (lm - (h << (15u - lane))) >> (15u - lane) : 0u;
    // over-subscribed (inflate.nim:41-46): more codes of a length than that length has, or past its last value
    const bool bad = lane >= 1 && lane < 16 && (h > (1u << lane) || (h > 0 && first_code + h - 1u >= (1u << lane)));
    const uint64_t anybad = __ballot(bad);
    if (lane >= 1 && lane < 16) {
      tab->first_code[lane] = (uint16_t)first_code;
      tab->first_symbol[lane] = (uint16_t)first_sym;
      tab->max_codes[lane] = (first_code + h) << (16u - lane);
      scratch[80u + lane] = lm;
    }
    if (lane == 0) {
      tab->max_codes[16] = 1u << 16;
      scratch[80] = anybad ? 1u : 0u;
    }
  }
  __device__ __forceinline__ bool oversubscribed() const { return scratch[80] != 0u; }
  // 3. symbols in canonical order (and the partition's limits into registers)
  __device__ __forceinline__ void canonical(uint32_t vt) {
#pragma unroll
    for (uint32_t i = 1; i < 16; i++) lim[i] = scratch[80u + i];
#pragma unroll
    for (uint32_t j = 0; j < kSlots; j++) {
      const uint32_t s = vt + j * T;
      if (s < kGroups * 64u && sl[j]) {
        uint32_t before = tab->first_symbol[sl[j]];
        for (uint32_t g = 0; g < (s >> 6); g++) before += scratch[g * 16u + sl[j]];
        values[before + srank[j]] = (uint16_t)s;
      }
    }
  }
  // the code that the 15-bit value v starts with: its length (16: none, the code is incomplete there)
  __device__ __forceinline__ uint32_t len_of(uint32_t v) const {
    uint32_t l = 1;
#pragma unroll
    for (uint32_t i = 1; i < 16; i++) l += v >= lim[i] ? 1u : 0u;
    return l;
  }
  __device__ __forceinline__ uint32_t entry_of(uint32_t v, uint32_t l) const {
    const uint32_t sym = values[(uint32_t)tab->first_symbol[l] + (v >> (15u - l)) - (uint32_t)tab->first_code[l]];
    return KIND == 0 ? litlen_entry_a(sym, l) : dist_entry_a(sym, l);
  }
  // 4a. every root prefix (prefix p = the first R bits of the stream as a number, first bit on top): the length of
  // the code it starts with, or (R + index bits of its second-level table) | 32; the tables' sizes, summed a wave
  __device__ __forceinline__ void classify() {
    const uint32_t tid = threadIdx.x;
    size_sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPP; k++) {
      const uint32_t p = tid * kPP + k;
      pl[k] = 0;
      if (p < (1u << R)) {
        const uint32_t v0 = p << (15u - R);
        const uint32_t l = len_of(v0);
        if (l <= R) {
          pl[k] = l;
        } else if (v0 < lim[15]) {  // codes below the prefix: the longest is the one its last assigned value starts with
          const uint32_t top = (v0 + (1u << (15u - R)) < lim[15] ? v0 + (1u << (15u - R)) : lim[15]) - 1u;
          const uint32_t jl = len_of(top);
          pl[k] = jl | 32u;
          size_sum += 1u << (jl - R);
        }
      }
    }
    incl = zh_wave_scan(size_sum);
    if (zh_lane() == 63) scratch[96u + (tid >> 6)] = incl;
  }
  // 4b. the root entries, and the second-level tables in prefix order (= canonical order) as long as they fit
  __device__ __forceinline__ void fill() {
    const uint32_t tid = threadIdx.x, wv = tid >> 6;
    uint32_t at = incl - size_sum;
    for (uint32_t w = 0; w < wv; w++) at += scratch[96u + w];
#pragma unroll
    for (uint32_t k = 0; k < kPP; k++) {
      const uint32_t p = tid * kPP + k;
      if (p < (1u << R)) {
        const uint32_t x = __brev(p) >> (32u - R);  // where the decoder looks: the stream carries codes first bit first
        const uint32_t v0 = p << (15u - R);
        uint32_t e = 0;
        if (pl[k] & 32u) {
          const uint32_t sb = (pl[k] & 31u) - R, size = 1u << sb;
          if (at + size <= SUBCAP) {  // (the sums grow: behind the first table that does not fit none does)
            const uint32_t base = (1u << R) + at;
            e = sb | 0x400u | (base << 16);
            for (uint32_t q = 0; q < size; q++) {
              const uint32_t v = v0 | ((__brev(q) >> (32u - sb)) << (15u - R - sb));
              const uint32_t l = len_of(v);
              lut[base + q] = l <= R + sb && v < lim[15] ? entry_of(v, l) : 0u;
            }
          }
          at += size;
        } else if (pl[k]) {
          e = entry_of(v0, pl[k]);
        }
        lut[x] = e;
      }
    }
  }
};

// Both decode tables of a block by the whole workgroup (T threads, 256 or 1024, all of them), the two tables' steps
// between the same five barriers: the literal / length code's symbols on every wave, the distance code's on wave 1;
// the counts of the one by wave 0, of the other by wave 1; the prefixes of both on every thread.  lens in LDS,
// `scratch`: kWgScratchWords dwords of LDS nobody else uses meanwhile.  Ends with a barrier (the tables are ready
// for every thread); the returned status is the same in every thread.
template <uint32_t T>
__device__ __noinline__ int build_tables_wg(const uint8_t* lit_lens, uint32_t hlit, uint32_t* lit_lut, HuffTab* lit_tab,
                                            uint16_t* lit_values, const uint8_t* dist_lens, uint32_t hdist,
                                            uint32_t* dist_lut, HuffTab* dist_tab, uint16_t* dist_values,
                                            uint32_t* scratch) {
  const uint32_t tid = threadIdx.x;
  WgTable<T, kLitBits, kLitSub, 0> lit;
  lit.lens = lit_lens; lit.n = hlit; lit.lut = lit_lut; lit.tab = lit_tab; lit.values = lit_values; lit.scratch = scratch;
  WgTable<T, kDistBits, 256u, 1> dst;
  dst.lens = dist_lens; dst.n = hdist; dst.lut = dist_lut; dst.tab = dist_tab; dst.values = dist_values; dst.scratch = scratch + 128;
  const uint32_t dvt = tid - 64u;  // the distance code's symbols on wave 1 (other threads: a number past its symbols)
  lit.ranks(tid);
  dst.ranks(dvt);
  __syncthreads();
  if (tid < 64u) lit.counts();
  else if (tid < 128u) dst.counts();
  __syncthreads();
  if (lit.oversubscribed() || dst.oversubscribed()) {
    __syncthreads();  // (everybody has read the flags before the scratch is used again)
    return ZH_ERR_INVALID_BUFFER;
  }
  lit.canonical(tid);
  dst.canonical(dvt);
  __syncthreads();
  lit.classify();
  dst.classify();
  __syncthreads();
  lit.fill();
  dst.fill();
  __syncthreads();
  return ZH_OK;
}

// The code lengths of a dynamic header (inflate.nim:115-171) by ONE WAVE instead of one lane's scalar chain:
// the 64 lanes decode the code-length symbols that start at 64 consecutive bit positions, a wave-uniform walk
// (one v_readlane a symbol) picks the ones really in the sequence, repeat counts become places by a prefix sum.
// `hdr`: the header staged in LDS (dwords), bit 0 of hdr[0] = bit 0 of the staged range; `q`: the first bit
// behind HCLEN; `end_bits`: the input's end in the same bit coordinates.  Fills lens[0 .. hlit + hdist) and
// returns the bit behind the last symbol -- or 0 where the header is anything but clean (a code-length code
// that is over-subscribed or does not cover what it is asked, a repeat with nothing before it or past the
// end, input that ends inside): the caller then runs the serial reader, which knows the reference's status
// for every such case.  `lut8`: 128 bytes of LDS.
__device__ __noinline__ uint32_t code_lengths_wave(const uint32_t* hdr, uint32_t q, uint32_t hlit, uint32_t hdist, uint32_t hclen,
                                      uint64_t end_bits, uint8_t* lens, uint8_t* lut8) {
  const unsigned lane = zh_lane();
  auto peek = [&](uint32_t at) -> uint32_t { return zh_alignbit(hdr[(at >> 5) + 1u], hdr[at >> 5], at); };
  // lane s < 19: the length of code-length symbol s; its place in the header is the inverse of c_clcl_order
  // {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2}, five bits a symbol in two constants
  constexpr uint64_t kPlaceLo = 3ull | 17ull << 5 | 15ull << 10 | 13ull << 15 | 11ull << 20 | 9ull << 25 | 7ull << 30 |
                                5ull << 35 | 4ull << 40 | 6ull << 45 | 8ull << 50 | 10ull << 55;
  constexpr uint64_t kPlaceHi = 12ull | 14ull << 5 | 16ull << 10 | 18ull << 15 | 0ull << 20 | 1ull << 25 | 2ull << 30;
  const uint32_t place = lane < 12u ? (uint32_t)(kPlaceLo >> (5u * lane)) & 31u
                                    : lane < 19u ? (uint32_t)(kPlaceHi >> (5u * (lane - 12u))) & 31u : 99u;
  const uint32_t l = place < hclen ? peek(q + 3u * place) & 7u : 0u;
  q += 3u * hclen;
  // canonical codes (inflate.nim:29-65): symbols of one length in symbol order; Kraft sum in 1/128ths
  uint32_t code = 0, next = 0;
#pragma unroll
  for (uint32_t k = 1; k <= 7; k++) {
    const uint64_t m = __ballot(l == k);
    if (l == k) code = next + (uint32_t)__popcll(m & zh_lanemask_lt());
    next = (next + (uint32_t)__popcll(m)) << 1;
  }
  if (zh_wave_sum(l ? 128u >> l : 0u) > 128u) return 0;  // over-subscribed
  zh_wave_sync();
  if (lane < 32u) reinterpret_cast<uint32_t*>(lut8)[lane] = 0;
  for (uint32_t i = lane; i < (320u + 16u) / 4u; i += 64u) reinterpret_cast<uint32_t*>(lens)[i] = 0;
  zh_wave_sync();
  if (l) {
    const uint32_t rev = __brev(code) >> (32u - l);
    for (uint32_t e = rev; e < 128u; e += 1u << l) lut8[e] = (uint8_t)(lane | (l << 5));
  }
  zh_wave_sync();
  const uint32_t total = hlit + hdist;
  uint32_t i = 0, prev = 0;
  while (i < total) {
    const uint32_t w = peek(q + lane);  // the symbol that starts at bit q + lane
    const uint32_t e = lut8[w & 127u], sym = e & 31u, cl = e >> 5;
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
    if (cl == 0u) nb = 64u;  // no code here: a walk that comes by ends the fast path
    uint64_t on = 0;
    uint32_t pos = 0;
    while (pos < 64u) {
      on |= 1ull << pos;
      pos += (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)pos);
    }
    bool mine = (on >> lane) & 1ull;
    const uint32_t incl = zh_wave_scan(mine ? rep : 0u);
    const uint32_t at = i + incl - (mine ? rep : 0u);  // index of my first entry
    if (mine && at >= total) mine = false;             // behind the last length: not part of the header
    const uint64_t ON = __ballot(mine);                // (bit 0 is set: at = i < total)
    if (__ballot(mine && (cl == 0u || at + rep > total))) return 0;
    if (i == 0 && (uint32_t)__builtin_amdgcn_readlane((int)sym, 0) == 16u) return 0;  // nothing to repeat
    // "previous length" for symbol 16: the nearest symbol before that is not a 16 (17 / 18 leave zero)
    const uint64_t plain = __ballot(mine && sym != 16u) & zh_lanemask_lt();
    const uint32_t from = plain ? 63u - (uint32_t)__clzll((long long)plain) : 0u;
    const uint32_t pv = (uint32_t)__shfl((int)val, (int)from, 64);
    if (sym == 16u) val = plain ? pv : prev;
    if (mine && val)  // (zeros are there already; what is left repeats six times at most)
      for (uint32_t j = 0; j < rep; j++) lens[at + j] = (uint8_t)val;
    const uint32_t lastl = 63u - (uint32_t)__clzll((long long)ON);
    i = (uint32_t)__builtin_amdgcn_readlane((int)(at + rep), (int)lastl);
    prev = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)lastl);
    q += lastl + (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)lastl);
  }
  zh_wave_sync();
  if ((uint64_t)q > end_bits) return 0;
  return q;
}

}  // namespace

