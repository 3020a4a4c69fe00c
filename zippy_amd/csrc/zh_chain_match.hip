// Hash-chain LZ77 matcher for levels -1 (= 6) and 2..9: lz77.nim:10-130 encodeLz77,
// re-cut so that almost all of it is position-parallel.
//
// The reference walks a block once: at every position it inserts the position into
// `head`/`chain` (also for the bytes a match skips, lz77.nim:121-126), so the chain state
// a position sees does NOT depend on the parse: it is "every earlier position of the
// block, newest first, linked by equal 17-bit hash".  Hence
//   1. the links: prevw[P] = the value the reference's `chain[windowPos]` receives when P is
//      inserted = window position of the latest earlier position with the same hash, 0
//      when there is none (the reference's "empty" sentinel, lz77.nim:88) -- stale
//      entries older than the window included, exactly like its never-cleared `head`.
//      zh_chain_class_* (1c, the default): a block's positions sorted into 32 hash classes,
//      each class linked in order by a wave of its own; zh_chain_prev_kernel / _ldst_kernel
//      (1 / 1b, ZH_CHAIN_PREV=serial): one wave per <= 4 MiB block, 64 positions per turn;
//   2. zh_chain_walk_kernel (a thread per 32-position chunk, only the positions a greedy walk
//      visits; zh_chain_search_kernel, one THREAD per position, is the cross-check): the reference's bounded chain
//      walk (good / nice / chain of internal.nim:177-189, the decreasing-offset wrap test,
//      the self-loop test, determineMatchLength) with `chain[w]` read as prevw[] of the
//      position that owns window slot w at that time -> best (length, offset) per position;
//   3. zh_chain_select_par_kernel (a workgroup per block, chunk walkers handing their ends on;
//      zh_chain_select_kernel, one wave per block, is the cross-check): the greedy parse itself
//      (lz77.nim:73-130): accept matches longer than 4, hop over them, literals otherwise;
//      what no walk of step 2 came by it works out itself.
// The match list / fragment bookkeeping handed to the Huffman and emission kernels is
// the same as the BestSpeed matcher's; output is byte-identical to the serial walk.
//
// zh_frag_stats_kernel then derives the per-fragment histograms from the match
// list (the addLiteral/addCopy bookkeeping of lz77.nim:19-50).
#include <cstdlib>
#include <cstring>

#include <atomic>
#include <type_traits>
#include "zh_common.h"
#include "zh_tables.h"

namespace {
constexpr uint32_t kHashMul = 0x1e35a7bdu;
constexpr uint32_t kHashBits = 17;

__device__ inline uint64_t load64u(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ inline uint32_t load32u(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
// NARROW link records (round 5): 32 bits a position -- the 16-bit link and a 16-bit tag of the position's first FIVE
// bytes -- instead of 64 (the link and six bytes).  The searches are bound by the lines their gathers pull through
// the fabric (128 bytes a miss for one record: profiles/r05_*_pmc_sq.txt), and a line of narrow records holds 32
// positions instead of 16.  What a tag can decide is "this candidate matches fewer than five bytes" (tags differ),
// which is all the search needs to know of such a candidate where `good` >= 5 (levels 5-9 and -1: a match of up to
// four bytes neither cuts `tries`, nor reaches `nice`, nor is returned, lz77.nim:104-114; it only raises a
// longest_len that stays below every length that matters); a candidate whose tag agrees is compared byte for byte
// against the source, as a candidate whose six bytes agree always was.  Levels 2-4 (`good` = 4: a match of exactly
// four bytes cuts `tries`) and the in-order link kernels keep the wide records.
__device__ __forceinline__ uint32_t zh_chain_tag(uint64_t first8) {  // (32-bit arithmetic: two multiplies, not a 64-bit one)
  return ((uint32_t)first8 * 0x9e3779b1u + ((uint32_t)(first8 >> 32) & 0xffu) * 0x85ebca6bu) >> 16;
}
template <bool kNarrow>
struct ChainRec {
  typedef typename std::conditional<kNarrow, uint32_t, uint64_t>::type T;
};
}  // namespace

// ---- 1. previous same-hash window position of every inserted position ----
// One wave a block, 64 positions a step, in order.  `head` (a dword a slot, in L2) is only ever
// EXCHANGED -- by the last lane of a step that has a hash; lanes that share one are sorted out
// inside the wave -- and nothing waits for what an exchange returns: exchanges of one wave on one
// address are served in program order, so the steps of a block of sixteen are issued back to back
// and their results picked up afterwards.  (The load / compare / store form this replaces paid
// the L2's latency 16 384 times a MiB.)
template <bool kTiny>  // kTiny: a block of fewer than eight bytes (its bytes are fetched one by one)
__device__ __forceinline__ void zh_chain_prev_block(const uint8_t* __restrict__ d_src, const ZhCompressArgs& a,
                                                    uint32_t* __restrict__ head,
                                                    uint64_t* __restrict__ prevw, uint32_t* s_cnt,
                                                    const ZhBlockDesc& bd) {
  constexpr uint32_t kDepth = 16;  // steps in flight (an exchange takes ~4 500 cycles to come back, a step ~900)
  const unsigned lane = zh_lane();
  const uint8_t* src = d_src + bd.src_off;
  const uint32_t block_len = (uint32_t)bd.len;
  uint64_t* pw = prevw + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  const uint32_t nins = block_len > 4u ? block_len - 4u : 0u;  // positions with pos + 4 < block end
  for (uint32_t i = lane; i < 1024; i += 64) s_cnt[i] = 0;
  zh_wave_sync();
  // the eight bytes at a position (zeros behind the block's end), a block of steps ahead.  Every
  // memory operation of the loop is unconditional -- addresses are clamped -- so that the compiler
  // counts what is in flight instead of waiting for all of it.
  auto fetch = [&](uint32_t base) -> uint64_t {
    const uint32_t P = base + lane;
    const uint32_t Pc = P < nins ? P : nins - 1u;
    if (kTiny) {
      uint64_t v = 0;
      for (uint32_t j = 0; j < 8u && Pc + j < block_len; j++) v |= (uint64_t)src[Pc + j] << (8u * j);
      return v;
    }
    const uint32_t at = Pc + 8u <= block_len ? Pc : block_len - 8u;
    return load64u(src + at) >> (8u * (Pc - at));
  };
  uint64_t wq[kDepth];
#pragma unroll
  for (uint32_t k = 0; k < kDepth; k++) wq[k] = nins ? fetch(64u * k) : 0ull;
  for (uint32_t base0 = 0; base0 < nins; base0 += 64 * kDepth) {
    uint32_t ret[kDepth], link_in[kDepth], from[kDepth];
    uint64_t six[kDepth];
#pragma unroll
    for (uint32_t k = 0; k < kDepth; k++) {
      const uint32_t base = base0 + 64u * k;
      const uint32_t P = base + lane;
      const bool valid = P < nins;
      const uint64_t w8 = wq[k];
      wq[k] = fetch(base + 64u * kDepth);
      const uint32_t h = ((uint32_t)w8 * kHashMul) >> (32 - kHashBits);
      const uint32_t ck = (h & 4095u) >> 2, cs = (h & 3u) * 8u;
      if (valid) atomicAdd(&s_cnt[ck], 1u << cs);
      zh_wave_sync();
      const uint32_t cnt = valid ? (s_cnt[ck] >> cs) & 255u : 0u;
      zh_wave_sync();
      if (valid) s_cnt[ck] = 0;
      bool last = true;             // last position of the step with this hash: it ends up in `head`
      uint32_t in_step = 0xffffffffu;  // the link, if an earlier lane of the step has the hash
      uint32_t last_lane = lane;       // ... else what `head` held: the last lane of the group gets it back
      uint64_t cc = __ballot(cnt > 1u);
      while (cc) {  // groups of lanes that may share a hash: resolve exactly, in position order
        const uint32_t jx = (uint32_t)__ffsll((long long)cc) - 1u;
        const uint32_t hj = __builtin_amdgcn_readlane(h, jx);
        const uint64_t same = __ballot(valid && h == hj);
        if ((same >> lane) & 1ull) {
          const uint64_t below = same & zh_lanemask_lt();
          if (below) in_step = (base + 63u - (uint32_t)__clzll((long long)below)) & 32767u;
          last = (same >> lane) >> 1 == 0;
          last_lane = 63u - (uint32_t)__clzll((long long)same);
        }
        cc &= ~same;
      }
      ret[k] = __hip_atomic_exchange(head + (valid && last ? h : (1u << kHashBits) + lane), P & 32767u,
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      link_in[k] = in_step;
      from[k] = last_lane;
      // the position's first six bytes travel with its link: the search reads both with one gather
      six[k] = w8 & 0xffffffffffffull;
    }
#pragma unroll
    for (uint32_t k = 0; k < kDepth; k++) {
      const uint32_t P = base0 + 64u * k + lane;
      const uint32_t got = (uint32_t)__shfl((int)ret[k], (int)from[k], 64);
      const uint32_t old = link_in[k] != 0xffffffffu ? link_in[k] : got;
      // (positions behind the last inserted one: a slot nobody reads)
      pw[P < nins ? P : nins] = (uint64_t)(old & 0xffffu) | (six[k] << 16);
    }
  }
}

__global__ __launch_bounds__(64) void zh_chain_prev_kernel(const uint8_t* __restrict__ d_src,
                                                           ZhCompressArgs a,
                                                           uint32_t* __restrict__ head_scratch,
                                                           uint64_t* __restrict__ prevw, uint32_t first_block) {
  __shared__ uint32_t s_cnt[1024];  // byte-wide counters of the group's hashes (12 bits)
  const ZhBlockDesc bd = a.blocks[first_block + blockIdx.x];
  uint32_t* head = head_scratch + (size_t)blockIdx.x * ZH_CHAIN_HEAD_WORDS;  // zeroed before the launch
  if (bd.len < 8u) zh_chain_prev_block<true>(d_src, a, head, prevw, s_cnt, bd);
  else zh_chain_prev_block<false>(d_src, a, head, prevw, s_cnt, bd);
}

// ---- 1b. the same links by load / compare / store on a 16-bit `head` ----
// For batches of more than zh_chain_prev_slice() blocks: every lane of every step of the exchanging
// form above pulls a line of its block's 512 KiB table through the fabric, and with thousands of
// blocks in flight that traffic, not the latency, is what it waits for (4096 blocks: 204 ms against
// 123 ms for this form, whose tables are half as large; 512 blocks: 21 ms against 31 ms).
__global__ __launch_bounds__(64) void zh_chain_prev_ldst_kernel(const uint8_t* __restrict__ d_src,
                                                                ZhCompressArgs a,
                                                                uint16_t* __restrict__ head_scratch,
                                                                uint64_t* __restrict__ prevw, uint32_t first_block) {
  __shared__ uint32_t s_cnt[1024];  // byte-wide counters of the group's hashes (12 bits)
  const unsigned lane = zh_lane();
  const ZhBlockDesc bd = a.blocks[first_block + blockIdx.x];
  const uint8_t* src = d_src + bd.src_off;
  const uint32_t block_len = (uint32_t)bd.len;
  uint16_t* head = head_scratch + ((size_t)blockIdx.x << kHashBits);  // zeroed before the launch
  uint64_t* pw = prevw + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  const uint32_t nins = block_len > 4u ? block_len - 4u : 0u;  // positions with pos + 4 < block end
  for (uint32_t i = lane; i < 1024; i += 64) s_cnt[i] = 0;
  zh_wave_sync();
  for (uint32_t base = 0; base < nins; base += 64) {
    const uint32_t P = base + lane;
    const bool valid = P < nins;
    const uint32_t h = valid ? (load32u(src + P) * kHashMul) >> (32 - kHashBits) : 0u;
    // `head` lives in HBM/L2; read and written past this CU's L1 so that the next turn sees it
    uint32_t old = valid ? __hip_atomic_load(head + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const uint32_t ck = (h & 4095u) >> 2, cs = (h & 3u) * 8u;
    if (valid) atomicAdd(&s_cnt[ck], 1u << cs);
    zh_wave_sync();
    const uint32_t cnt = valid ? (s_cnt[ck] >> cs) & 255u : 0u;
    zh_wave_sync();
    if (valid) s_cnt[ck] = 0;
    bool last = true;  // last position of the group with this hash: it ends up in `head`
    uint64_t cc = __ballot(cnt > 1u);
    while (cc) {  // groups of lanes that may share a hash: resolve exactly, in position order
      const uint32_t jx = (uint32_t)__ffsll((long long)cc) - 1u;
      const uint32_t hj = __builtin_amdgcn_readlane(h, jx);
      const uint64_t same = __ballot(valid && h == hj);
      if ((same >> lane) & 1ull) {
        const uint64_t below = same & zh_lanemask_lt();
        if (below) old = (base + 63u - (uint32_t)__clzll((long long)below)) & 32767u;
        last = (same >> lane) >> 1 == 0;
      }
      cc &= ~same;
    }
    if (valid) {
      uint64_t six = 0;
      if (P + 8u <= block_len) {
        six = load64u(src + P) & 0xffffffffffffull;
      } else {
        for (uint32_t k = 0; k < 6u && P + k < block_len; k++) six |= (uint64_t)src[P + k] << (8u * k);
      }
      pw[P] = (uint64_t)(old & 0xffffu) | (six << 16);
      if (last) __hip_atomic_store(head + h, (uint16_t)(P & 32767u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    zh_wave_sync();
  }
}

// ---- 1c. the same links, a block on many waves: positions sorted into hash classes first ----
// The in-order kernels above walk a block with one wave: 16 384 steps a MiB whatever the batch, and a `head`
// table of 256-512 KiB a block that lives in L2/MALL (a line through the fabric per lane and step).  A link
// only ever connects positions of equal hash, so the positions of a block are first sorted -- stably: in
// position order -- into kClasses classes by the low bits of their hash (a counting pass, a scan, a scatter,
// all position-parallel), and each class is then linked by a wave of its own, in order, against its slots
// of `head` in LDS (16-bit entries).  Same values as kernels 1 / 1b: the previous position of equal hash
// (its low 15 bits), stale entries and all.
// Scratch of a block: `cls` = per unit of kUnit positions the class counts (then: where the unit's
// positions of a class go), followed by the classes' starts and sizes; the sorted positions themselves
// borrow the block's part of best[] (cleared afterwards, before the walks, anyway).
#ifndef ZH_CHAIN_LINK_TURNS
#define ZH_CHAIN_LINK_TURNS 3
#endif
// classes of a block that one workgroup links in step, and the positions a tile of that step (zh_chain_class_links_kernel;
// 512 x 1 MiB at level -1, the three link kernels together, ms: a wave a class at its own pace 11.9-12.7; 4 classes a
// workgroup 12.1; 8: 10.2-11.1 at 8192 positions a tile, 11.3 at 4096, 10.4-11.0 at 16 384, 10.6-11.1 at 32 768, 10.9-11.6
// at 65 536; 64 classes, 16 / 8 a workgroup: 11.8 / 13.0.  16 of 32 do not fit the LDS.  profiles/r06_an_*, r06_ao_*)
#ifndef ZH_CHAIN_LINKS_GROUP
#define ZH_CHAIN_LINKS_GROUP 8
#endif
#ifndef ZH_CHAIN_LINKS_TILE
#define ZH_CHAIN_LINKS_TILE 16384
#endif
#ifndef ZH_CHAIN_CHUNK
#define ZH_CHAIN_CHUNK 32  // positions a walk starts at the first of
#endif
#ifndef ZH_CHAIN_CLASS_BITS
#define ZH_CHAIN_CLASS_BITS 5
#endif
namespace {
constexpr uint32_t kClassBits = ZH_CHAIN_CLASS_BITS;
constexpr uint32_t kClasses = 1u << kClassBits;
constexpr uint32_t kClsStride = 1u << 16;   // words of scratch a block: (4 MiB / kUnit) units x kClasses, + 2 x kClasses
constexpr uint32_t kClsInfo = 1u << 15;     // ... of which [kClsInfo, kClsInfo + 2 kClasses): class starts and sizes
constexpr uint32_t kUnit = 128u * kClasses; // positions a wave counts / scatters
constexpr uint32_t kUnitsPerFrag = ZH_FRAG_SIZE / kUnit;
constexpr uint32_t kClsWaves = kUnitsPerFrag < 4u ? kUnitsPerFrag : 4u;  // waves of a counting workgroup
__device__ __forceinline__ uint32_t zh_chain_hash(uint32_t four) { return (four * kHashMul) >> (32 - kHashBits); }
}  // namespace
template <bool kScatter>
__global__ __launch_bounds__(64 * kClsWaves) void zh_chain_class_kernel(const uint8_t* __restrict__ d_src, ZhCompressArgs a,
                                                                        uint32_t* __restrict__ cls_scratch,
                                                                        uint32_t* __restrict__ lists) {
  __shared__ uint32_t s_c[kClsWaves][kClasses];  // count / next free place of a class, a row a wave
  const unsigned lane = zh_lane();
  const uint32_t wv = threadIdx.x >> 6;
  constexpr uint32_t kGroups = kUnitsPerFrag / kClsWaves;  // workgroups a fragment
  const uint32_t f = a.first_frag + blockIdx.x / kGroups;
  const uint32_t unit_in_frag = (blockIdx.x % kGroups) * kClsWaves + wv;
  const ZhFragDesc fd = a.frags[f];
  if (unit_in_frag * kUnit >= fd.len) return;
  const ZhBlockDesc bd = a.blocks[fd.block];
  const uint8_t* src = d_src + bd.src_off;
  const uint32_t block_len = (uint32_t)bd.len;
  const uint32_t nins = block_len > 4u ? block_len - 4u : 0u;  // positions with pos + 4 < block end
  const uint32_t unit = (f - bd.first_frag) * kUnitsPerFrag + unit_in_frag;  // of the block
  uint32_t* cls = cls_scratch + (size_t)(fd.block - a.first_block) * kClsStride + (size_t)unit * kClasses;
  uint32_t* list = lists + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  if (lane < kClasses) s_c[wv][lane] = kScatter ? cls[lane] : 0u;
  zh_wave_sync();
  for (uint32_t r = 0; r < kUnit / 64u; r++) {
    const uint32_t P = unit * kUnit + r * 64u + lane;
    if (P < nins) {
      const uint32_t c = zh_chain_hash(load32u(src + P)) & (kClasses - 1u);
      // (the lanes of one LDS atomic are served in ascending order: places in position order)
      const uint32_t dest = atomicAdd(&s_c[wv][c], 1u);
      if (kScatter) list[dest] = P;
    }
    zh_wave_sync();
  }
  if (!kScatter && lane < kClasses) cls[lane] = s_c[wv][lane];
}
// counts -> places: class c's positions start at the classes' sizes before it, unit u's at its class's start
// plus the counts of the units before it.  One workgroup a block, a wave takes kClasses / 8 classes.
__global__ __launch_bounds__(512) void zh_chain_class_scan_kernel(ZhCompressArgs a, uint32_t* __restrict__ cls_scratch) {
  __shared__ uint32_t s_tot[kClasses];
  const unsigned lane = zh_lane();
  const uint32_t wv = threadIdx.x >> 6;
  const ZhBlockDesc bd = a.blocks[a.first_block + blockIdx.x];
  const uint32_t block_len = (uint32_t)bd.len;
  const uint32_t nins = block_len > 4u ? block_len - 4u : 0u;
  const uint32_t units = (nins + kUnit - 1u) / kUnit;
  uint32_t* cls = cls_scratch + (size_t)blockIdx.x * kClsStride;
  for (uint32_t c = wv; c < kClasses; c += 8u) {
    uint32_t tot = 0;
    for (uint32_t u0 = 0; u0 < units; u0 += 64u) {
      const uint32_t u = u0 + lane;
      tot += u < units ? cls[(size_t)u * kClasses + c] : 0u;
    }
    tot = zh_wave_sum(tot);
    if (lane == 0) s_tot[c] = tot;
  }
  __syncthreads();
  for (uint32_t c = wv; c < kClasses; c += 8u) {
    uint32_t start = 0;
    for (uint32_t k = 0; k < c; k++) start += s_tot[k];
    if (lane == 0) {
      cls[kClsInfo + c] = start;
      cls[kClsInfo + kClasses + c] = s_tot[c];
    }
    uint32_t run = start;
    for (uint32_t u0 = 0; u0 < units; u0 += 64u) {
      const uint32_t u = u0 + lane;
      const uint32_t n = u < units ? cls[(size_t)u * kClasses + c] : 0u;
      const uint32_t incl = zh_wave_scan(n);
      if (u < units) cls[(size_t)u * kClasses + c] = run + incl - n;
      run += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
    }
  }
}
// a class of a block, in order: 64 of its positions a step.  `head` holds position + 1 (zero: empty), and a step
// is ONE LDS atomic: positions grow along the list and the lanes of an LDS atomic are served in ascending order,
// so atomicMax hands every lane what the slot held just before it -- the previous position of equal hash, be
// it an earlier lane of the step or an earlier step -- and leaves the step's last one there.
template <bool kNarrow, uint32_t kGroup>
__global__ __launch_bounds__(64 * kGroup) void zh_chain_class_links_kernel(const uint8_t* __restrict__ d_src, ZhCompressArgs a,
                                                                           const uint32_t* __restrict__ cls_scratch,
                                                                           uint32_t* __restrict__ lists,
                                                                           uint64_t* __restrict__ prevw, uint32_t ngroups) {
  constexpr uint32_t kAhead = 4;                 // steps whose positions and bytes are on their way
  constexpr uint32_t kSlots = 1u << (kHashBits - kClassBits);
  __shared__ uint32_t s_head_all[kGroup][kSlots];  // lz77.nim:5-6 `head`, a class's slots a wave
  const unsigned lane = zh_lane(), wave = threadIdx.x >> 6;
  uint32_t* const s_head = s_head_all[wave];
  // (the classes of a block on ONE XCD: their stores fill the same lines of prevw)
  // kGroup > 1 (round 6): kGroup classes of a block are the waves of one workgroup and cross the block tile by tile, a
  // barrier a tile.  A 128-byte line of prevw holds 32 positions of as many classes; with a wave a class, each at its own
  // pace, 92 % of the 4-byte stores went on to the fabric alone (396 M writes for 512 MiB: what this kernel's time was);
  // what a workgroup's classes write into a line now arrives while the L2 still holds it, and leaves as one write.
  static_assert(kClasses % kGroup == 0, "a workgroup's classes belong to one block");
  const uint32_t wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t bid = wg * kGroup + wave;
  if (wg * kGroup >= ngroups) return;
  const uint32_t b = bid >> kClassBits, c = bid & (kClasses - 1u);  // (b: of the range)
  const ZhBlockDesc bd = a.blocks[a.first_block + b];
  const uint8_t* src = d_src + bd.src_off;
  const uint32_t block_len = (uint32_t)bd.len;
  const uint32_t* cls = cls_scratch + (size_t)b * kClsStride;
  const uint32_t n = cls[kClsInfo + kClasses + c];
  if (kGroup == 1 && !n) return;
  uint32_t* list_rw = lists + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE + cls[kClsInfo + c];
  const uint32_t* list = list_rw;
  typedef typename ChainRec<kNarrow>::T Rec;
  Rec* pw = reinterpret_cast<Rec*>(prevw) + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  for (uint32_t i = lane; i < kSlots; i += 64) s_head[i] = 0;
  zh_wave_sync();
  // the eight bytes at a listed position (zeros behind the block's end); every load is unconditional
  auto pos_at = [&](uint32_t i) -> uint32_t { return n ? list[i < n ? i : n - 1u] : 0u; };
  auto bytes_at = [&](uint32_t P) -> uint64_t {
    if (block_len < 8u) {
      uint64_t v = 0;
      for (uint32_t j = 0; j < 8u && P + j < block_len; j++) v |= (uint64_t)src[P + j] << (8u * j);
      return v;
    }
    const uint32_t at = P + 8u <= block_len ? P : block_len - 8u;
    return load64u(src + at) >> (8u * (P - at));
  };
  // a step's positions are asked for two rounds of kAhead steps ahead, its bytes (whose address they are) one
  uint32_t pq[kAhead], pn[kAhead];
  uint64_t wq[kAhead];
#pragma unroll
  for (uint32_t k = 0; k < kAhead; k++) {
    pq[k] = pos_at(64u * k + lane);
    pn[k] = pos_at(64u * (kAhead + k) + lane);
  }
#pragma unroll
  for (uint32_t k = 0; k < kAhead; k++) wq[k] = bytes_at(pq[k]);
  constexpr uint32_t kTile = ZH_CHAIN_LINKS_TILE;  // positions a tile (kGroup > 1)
  uint32_t base0 = 0;
  for (uint32_t tile_end = kGroup == 1 ? 0xffffffffu : kTile;; tile_end += kTile) {
  for (; base0 < n && (kGroup == 1 || (uint32_t)__builtin_amdgcn_readfirstlane((int)pq[0]) < tile_end); base0 += 64u * kAhead) {
    uint32_t P[kAhead], old[kAhead];
    uint64_t w8[kAhead];
#pragma unroll
    for (uint32_t k = 0; k < kAhead; k++) {  // (the steps' atomics in order, one behind the other)
      const uint32_t i = base0 + 64u * k + lane;
      P[k] = pq[k];
      w8[k] = wq[k];
      pq[k] = pn[k];
      wq[k] = bytes_at(pq[k]);
      pn[k] = pos_at(i + 2u * 64u * kAhead);
      const uint32_t h = zh_chain_hash((uint32_t)w8[k]) >> kClassBits;  // the slot inside the class
      old[k] = i < n ? atomicMax(&s_head[h], P[k] + 1u) : 0u;
      zh_wave_sync();
    }
#pragma unroll
    for (uint32_t k = 0; k < kAhead; k++) {
      const uint32_t i = base0 + 64u * k + lane;
      const uint32_t link = old[k] ? (old[k] - 1u) & 32767u : 0u;
      if (i < n) {
        if (kNarrow) pw[P[k]] = (Rec)(link | (zh_chain_tag(w8[k]) << 16));
        else pw[P[k]] = (Rec)((uint64_t)link | ((w8[k] & 0xffffffffffffull) << 16));
        list_rw[i] = 0;  // best[] goes back the way the walks expect it: nothing worked out
      }
    }
  }
  if (kGroup == 1 || tile_end >= block_len) break;
  __syncthreads();
  }
}

// ---- 2. best match of a position (lz77.nim:83-112) ----
// `pos` is block-relative, `pw` the block's links (kernel 1).  Returns length | offset << 16, or 0
// when the longest match is not longer than 4 (lz77.nim:114).
template <bool kNarrow>
__device__ __forceinline__ uint32_t zh_chain_search_one(const uint8_t* __restrict__ src,
                                                        const typename ChainRec<kNarrow>::T* __restrict__ pw, uint32_t pos,
                                                        uint32_t block_len, int good, int nice, int max_chain) {
  if (pos + 4u >= block_len) return 0;
  const uint32_t window_pos = pos & 32767u;
  const uint32_t limit = block_len < pos + 258u ? block_len : pos + 258u;
  const uint64_t own_rec = pw[pos];
  uint32_t hash_pos = (uint32_t)own_rec & 0xffffu;
  // the position's own first bytes: every candidate is compared against them, first through the
  // six bytes that travel with the candidate's chain link (one gather decides most candidates);
  // narrow records: through the tag of its first five
  const bool wide = pos + 8u <= limit;
  const uint64_t own6 = wide ? load64u(src + pos) & 0xffffffffffffull : 0ull;
  const uint32_t own_tag = (uint32_t)own_rec >> 16;
  int tries = max_chain;
  int prev_offset = 0, longest_offset = 0, longest_len = 0;
  while (tries > 0 && hash_pos != 0) {
    tries--;
    const int offset = hash_pos <= window_pos ? (int)(window_pos - hash_pos)
                                              : (int)(window_pos - hash_pos + 32768u);
    if (offset <= 0 || offset < prev_offset) break;
    prev_offset = offset;
    // determineMatchLength(src, pos - offset, pos, limit), internal.nim:251-270
    const uint64_t entry = pw[pos - (uint32_t)offset];  // chain[hashPos] | the candidate's six bytes (or its tag) << 16
    const uint8_t* s1 = src + (pos - (uint32_t)offset);
    uint32_t s2 = pos;
    int match_len = 0;
    bool done = false;
    if (kNarrow) {
      done = ((uint32_t)entry >> 16) != own_tag;  // fewer than five bytes: as good as none (see ChainRec)
    } else if (wide) {
      const uint64_t x6 = (entry >> 16) ^ own6;
      if (x6 != 0) {
        match_len = (int)((uint32_t)__builtin_ctzll(x6) >> 3);
        done = true;
      }
    }
    while (!done && s2 + 8u <= limit) {
      const uint64_t x = load64u(src + s2) ^ load64u(s1 + match_len);
      if (x != 0) {
        match_len += (int)((uint32_t)__builtin_ctzll(x) >> 3);
        done = true;
        break;
      }
      s2 += 8;
      match_len += 8;
    }
    if (!done)
      while (s2 < limit && src[s2] == s1[match_len]) {
        s2++;
        match_len++;
      }
    if (match_len > longest_len) {
      if (match_len >= good) tries >>= 2;
      longest_len = match_len;
      longest_offset = offset;
    }
    // chain[hashPos]: the value stored when the position in that window slot was inserted
    const uint32_t nxt = (uint32_t)entry & 0xffffu;
    if (longest_len >= nice || hash_pos == nxt) break;
    hash_pos = nxt;
  }
  return longest_len > 4 ? (uint32_t)longest_len | ((uint32_t)longest_offset << 16) : 0u;
}

// best[] entries: bit 31 = "worked out" (length in bits 0-15, offset in bits 16-30)
constexpr uint32_t kBestKnown = 0x80000000u;

#ifdef ZH_XCHECK
// ---- 2a. every position (the test build's ZH_CHAIN_SEARCH=dense: cross-check) ----
__global__ __launch_bounds__(256) void zh_chain_search_kernel(const uint8_t* __restrict__ d_src,
                                                              ZhCompressArgs a, int good, int nice,
                                                              int max_chain,
                                                              const uint64_t* __restrict__ prevw,
                                                              uint32_t* __restrict__ best,
                                                              uint32_t first_frag) {
  const uint32_t f = first_frag + blockIdx.x / (ZH_FRAG_SIZE / 256u);
  const uint32_t local = (blockIdx.x % (ZH_FRAG_SIZE / 256u)) * 256u + threadIdx.x;
  const ZhFragDesc fd = a.frags[f];
  if (local >= fd.len) return;
  const ZhBlockDesc bd = a.blocks[fd.block];
  const uint32_t pos = (f - bd.first_frag) * ZH_FRAG_SIZE + local;  // block-relative
  const uint64_t* pw = prevw + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  best[(size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE + pos] =
      kBestKnown | zh_chain_search_one<false>(d_src + bd.src_off, pw, pos, (uint32_t)bd.len, good, nice, max_chain);
}
#endif  // ZH_XCHECK

// ---- 2b. the positions a greedy walk comes by ----
// The parse (kernel 3) only ever asks for the positions it visits -- about a third of them -- and
// a walk p -> p + (length ? length : 1) falls in step with the true one quickly wherever it
// starts.  So a thread walks a 32-position chunk from its first byte and works out the best match
// of what it visits, on to the first position behind its chunk that the next chunk's walk has
// already done (there the two walks have met) or 16 384 positions at most.  Whatever the true walk
// visits and no walk here did is worked out by kernel 3 when it gets there: the values are those of
// kernel 2a either way, only fewer.
namespace {
constexpr uint32_t kWalkThreads = 256u;
}  // namespace
// kChunk: positions a walk starts at the first of (512 x 1 MiB, this kernel in ms: 16: 40.3, 32: 36.8, 64: 40.7, 128: 47.4)
// (Round 6: a WAVE's lanes taking the wave's chunks first come, first served -- a ballot and a count when a lane is done, the
// wave's positions and so an XCD's footprint unchanged -- to keep more than two lanes in five alive: byte-identical, and
// slower in every shape.  Walk + parse, ms, against 29.1 + 3.0 with one 32-position chunk a lane: two 16-position chunks a
// lane 31.8 + 10.9 with walks ending 64 positions behind their chunk, 33.4 + 3.0 with the usual 16 384; four of 8: 40.4 + 10.3;
// two of 32 / four of 16 / four of 32 (a wave over 4096 / 8192 positions): 40.1 / 47.2 / 51.1.  Shorter chunks pay more
// walks that are not yet in step, wider waves pay locality -- what round 3 found for the workgroup-wide form.
// profiles/r06_e_chain_walk_dynamic_chunks_ab.log)
template <uint32_t kChunk, bool kNarrow>
__global__ __launch_bounds__(kWalkThreads) void zh_chain_walk_kernel(const uint8_t* __restrict__ d_src, ZhCompressArgs a,
                                                            int good, int nice, int max_chain,
                                                            const uint64_t* __restrict__ prevw,
                                                            uint32_t* __restrict__ best, uint32_t first_frag,
                                                            uint32_t ngroups) {
  constexpr uint32_t kGroups = ZH_FRAG_SIZE / kChunk / kWalkThreads;  // workgroups a fragment
  // Workgroups go to the eight XCDs round robin.  Each XCD takes a contiguous eighth of the launch, so that
  // the workgroups of a fragment -- and of its neighbours, whose windows overlap -- gather through ONE L2
  // instead of eight (512 x 1 MiB: 56.0 -> 51.7 ms, 2048 x 1 MiB: 236.5 -> 204.1 ms).  gridDim.x is a multiple of 8.
  const uint32_t bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (bid >= ngroups) return;
  const uint32_t f = first_frag + bid / kGroups;
  const uint32_t local = ((bid % kGroups) * kWalkThreads + threadIdx.x) * kChunk;
  const ZhFragDesc fd = a.frags[f];
  if (local >= fd.len) return;
  const ZhBlockDesc bd = a.blocks[fd.block];
  const uint8_t* src = d_src + bd.src_off;
  const uint32_t block_len = (uint32_t)bd.len;
  typedef typename ChainRec<kNarrow>::T Rec;
  const Rec* pw = reinterpret_cast<const Rec*>(prevw) + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  uint32_t* bst = best + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  const uint32_t nmain = block_len > 4u ? block_len - 4u : 0u;  // lz77.nim:74-76: behind it only literals
  uint32_t pos = (f - bd.first_frag) * ZH_FRAG_SIZE + local;    // block-relative
  // a walk goes on behind its chunk until it meets a position that is worked out -- or 16 384 positions at
  // most (data whose walks never meet: runs).  What the parse then misses it works out itself, one search
  // at a time, so the allowance is generous (512 x 1 MiB, walk + parse: 1024: 36.8 + 6.5 ms, 4096: 36.9 + 4.6,
  // 16 384: 37.1 + 4.3, 32 768: 37.0 + 4.2; all zeros, 256 x 1 MiB: 5.1 + 18.9 -> 14.2 + 3.8)
  const uint32_t end = pos + kChunk, stop = end + 16384u;
  // The walk as ONE loop whose turn is one round trip to memory for every lane, whatever the lane is
  // doing: looking at a position (T: its entry of best[] and of pw[]), following a chain link (C: the
  // candidate's entry), comparing on (E: eight bytes of either side).  Chain lengths are very uneven
  // (1 MiB of the bench data: 15.8 links a visited position on average, two thirds of the positions
  // under eight, one in twenty-three all 128): with a loop per search inside a loop per position a wave
  // stayed in every search as long as its longest lane -- almost always 128 links -- at an eighth of its
  // lanes busy.  Here a lane that is done with a search goes on to its next position while the others
  // follow their chains; the values are zh_chain_search_one's, decision for decision.
  enum : uint32_t { kT = 0, kC = 1, kE = 2, kF = 3, kDone = 4 };
  constexpr uint32_t kLinkTurns = ZH_CHAIN_LINK_TURNS;  // (512 x 1 MiB, this kernel: 0: 37.0 ms, 1: 34.1, 3: 33.7, 7: 34.5)
  uint32_t st = kT;
  uint32_t hash_pos = 0, nxt = 0, limit = 0, window_pos = 0;
  int tries = 0, prev_offset = 0, longest_len = 0, longest_offset = 0, offset = 0, m = 0;
  uint64_t own6 = 0;
  bool wide = false;
  const uint64_t* bst2 = reinterpret_cast<const uint64_t*>(bst);  // (the block's part of best[] starts 8-byte aligned)
  while (st != kDone) {
    // ---- what ends without a load: a chain's end, a finished search, the walk's end ----
    if (st == kC) {
      if (!(tries > 0 && hash_pos != 0)) {
        st = kF;
      } else {
        tries--;
        offset = hash_pos <= window_pos ? (int)(window_pos - hash_pos) : (int)(window_pos - hash_pos + 32768u);
        if (offset <= 0 || offset < prev_offset) st = kF;
        else prev_offset = offset;
      }
    }
    if (st == kF) {
      const uint32_t r = longest_len > 4 ? (uint32_t)longest_len | ((uint32_t)longest_offset << 16) : 0u;
      __hip_atomic_store(bst + pos, r | kBestKnown, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pos += longest_len > 4 ? (uint32_t)longest_len : 1u;
      st = kT;
    }
    if (st == kT && !(pos < nmain && pos < stop)) st = kDone;
    // ---- the turn's loads (a lane that has just finished reads its block's first entry) ----
    const uint32_t cand = pos - (uint32_t)offset;
    const uint8_t* a1 = reinterpret_cast<const uint8_t*>(pw);
    const uint8_t* a2 = a1;
    if (st == kT) {
      a1 = reinterpret_cast<const uint8_t*>(pw + pos);
      a2 = reinterpret_cast<const uint8_t*>(bst2 + (pos >> 1));
    } else if (st == kC) {
      a1 = reinterpret_cast<const uint8_t*>(pw + cand);
    } else if (st == kE && pos + (uint32_t)m + 8u <= limit) {  // (the block's last bytes are read one by one below)
      a1 = src + pos + (uint32_t)m;
      a2 = src + cand + (uint32_t)m;
    }
    // (a narrow record is four bytes: its load must not reach behind the array; the other states' are eight)
    const uint64_t v1 = kNarrow && st != kE ? (uint64_t)load32u(a1) : load64u(a1);
    // (past the L1: best[] entries come from other workgroups, too; eight bytes at any address)
    const uint64_t v2 = __hip_atomic_load(reinterpret_cast<const uint64_t*>(a2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- what they mean ----
    bool decided = false;  // a candidate's match length `m` is known
    if (st == kT) {
      const uint32_t r = (uint32_t)(v2 >> ((pos & 1u) * 32u));
      if (r & kBestKnown) {
        if (pos >= end) {
          st = kDone;  // the next chunk's walk has been here: the two have met
        } else {
          const uint32_t len = r & 0xffffu;
          pos += len ? len : 1u;
        }
      } else {  // zh_chain_search_one<kNarrow>(src, pw, pos, ...) starts (pos < nmain: pos + 4 < block_len)
        window_pos = pos & 32767u;
        limit = block_len < pos + 258u ? block_len : pos + 258u;
        wide = pos + 8u <= limit;
        own6 = kNarrow ? (v1 >> 16) & 0xffffull : wide ? v1 >> 16 : 0ull;  // (the position's own six bytes -- or its tag -- travel with its link, too)
        hash_pos = (uint32_t)v1 & 0xffffu;
        tries = max_chain;
        prev_offset = 0;
        longest_len = 0;
        longest_offset = 0;
        st = kC;
      }
    } else if (st == kC) {
      nxt = (uint32_t)v1 & 0xffffu;
      const uint64_t x6 = ((v1 >> 16) & (kNarrow ? 0xffffull : ~0ull)) ^ own6;
      m = 0;
      if (kNarrow ? x6 != 0 : wide && x6 != 0) {  // (narrow: the tags differ -- fewer than five bytes, as good as none)
        m = kNarrow ? 0 : (int)((uint32_t)__builtin_ctzll(x6) >> 3);
        decided = true;
      } else {
        st = kE;
      }
    } else if (st == kE) {
      if (pos + (uint32_t)m + 8u <= limit) {
        const uint64_t x = v1 ^ v2;
        if (x != 0) {
          m += (int)((uint32_t)__builtin_ctzll(x) >> 3);
          decided = true;
        } else {
          m += 8;
        }
      } else {  // the block's last bytes, one by one
        uint32_t s2 = pos + (uint32_t)m;
        while (s2 < limit && src[s2] == src[cand + (uint32_t)m]) {
          s2++;
          m++;
        }
        decided = true;
      }
    }
    if (decided) {  // lz77.nim:104-112
      st = kC;
      if (m > longest_len) {
        if (m >= good) tries >>= 2;
        longest_len = m;
        longest_offset = offset;
      }
      if (longest_len >= nice || hash_pos == nxt) st = kF;
      else hash_pos = nxt;
    }
    // ---- kLinkTurns turns for the lanes that follow a chain, and nothing else: most of a walk's turns are
    // links, and a turn that only has to do that costs half the instructions of the one above.  (A lane that
    // ends its search here waits for the next full turn.)
#pragma unroll 1
    for (uint32_t r = 0; r < kLinkTurns; r++) {
      if (st == kC) {
        if (!(tries > 0 && hash_pos != 0)) {
          st = kF;
        } else {
          tries--;
          offset = hash_pos <= window_pos ? (int)(window_pos - hash_pos) : (int)(window_pos - hash_pos + 32768u);
          if (offset <= 0 || offset < prev_offset) {
            st = kF;
          } else {
            prev_offset = offset;
            const uint64_t e = pw[pos - (uint32_t)offset];
            nxt = (uint32_t)e & 0xffffu;
            const uint64_t x6 = (e >> 16) ^ own6;
            if (kNarrow ? x6 != 0 : wide && x6 != 0) {
              m = kNarrow ? 0 : (int)((uint32_t)__builtin_ctzll(x6) >> 3);
              if (m > longest_len) {
                if (m >= good) tries >>= 2;
                longest_len = m;
                longest_offset = offset;
              }
              if (longest_len >= nice || hash_pos == nxt) st = kF;
              else hash_pos = nxt;
            } else {
              m = 0;
              st = kE;  // (compared on in the full turns)
            }
          }
        }
      }
    }
  }
}

#ifdef ZH_XCHECK
// ---- 3. the greedy parse (lz77.nim:73-130) over the per-position results, a wave a block (the test build's
// ZH_CHAIN_SELECT=serial: cross-check) ----
__global__ __launch_bounds__(64) void zh_chain_select_kernel(ZhCompressArgs a,
                                                             const uint32_t* __restrict__ best) {
  const unsigned lane = zh_lane();
  const uint32_t b = a.first_block + blockIdx.x;
  const ZhBlockDesc bd = a.blocks[b];
  const uint32_t block_len = (uint32_t)bd.len;
  const uint32_t* bst = best + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;

  uint32_t cur_frag = 0, frag_matches = 0;  // fragment currently receiving matches
  auto close_frags_until = [&](uint32_t frag) {  // publish counts of fragments [cur_frag, frag)
    while (cur_frag < frag) {
      if (lane == 0) a.f_nmatch[bd.first_frag + cur_frag] = frag_matches;
      frag_matches = 0;
      cur_frag++;
    }
  };
  for (uint32_t k = lane; k < bd.nfrag; k += 64) a.f_spill[bd.first_frag + k] = 0;
  zh_wave_sync();

  uint32_t pos = 0;
  // lz77.nim:54-56,74-76: the last four positions (and blocks of <= 4 bytes) are literals
  const uint32_t nmain = block_len > 4u ? block_len - 4u : 0u;
  while (pos < nmain) {
    const uint32_t base = pos & ~63u;
    const uint32_t v = base + lane < nmain ? bst[base + lane] : 0u;
    uint32_t cur = pos - base;
    while (cur < 64u && base + cur < nmain) {
      const uint32_t r = __builtin_amdgcn_readlane(v, cur) & ~kBestKnown;  // (every position is worked out: dense search)
      const uint32_t len = r & 0xffffu;
      if (len) {
        const uint32_t p = base + cur, frag = p >> 15;
        close_frags_until(frag);
        if (lane == 0) {
          const size_t slot = (size_t)(bd.first_frag + frag) * ZH_MAX_MATCHES_PER_FRAG + frag_matches;
          a.m_pos[slot] = (uint16_t)(p & 32767u);
          a.m_len[slot] = (uint16_t)len;
          a.m_off[slot] = (uint16_t)(r >> 16);
          const uint32_t end = p + len;
          if ((end - 1) >> 15 != frag) a.f_spill[bd.first_frag + frag + 1] = end & 32767u;
        }
        frag_matches++;
        cur += len;
      } else {
        cur++;
      }
    }
    pos = base + cur;
  }
  close_frags_until(bd.nfrag);
}
#endif  // ZH_XCHECK

// ---- 3b. the same greedy parse, parallel inside a block ----
// The per-position results are static, so the parse is a walk p -> p + (len ? len : 1) over fixed
// data, and such walks fall in step with each other quickly whatever position they start from
// (the same property the inflate tokens kernel uses, zh_inflate_split.hip).  One workgroup per
// block takes a fragment (32768 positions, their match lengths staged in LDS as bytes) at a time, a
// thread per 128-position chunk: every thread walks its chunk from a guessed start (the chunk's
// first position) to the first visited position behind it, ends are handed on as the next chunk's
// start, and threads whose start changed walk again until no start changes -- thread 0's start
// (where the walk entered the fragment) is exact, so by induction all of them then are.  Data
// whose walks never meet (a run parsed into back-to-back maximal matches) makes the final prefix
// grow one chunk a turn: thread 0 then simply finishes the walk through LDS alone.  Counts are
// prefix-summed and a last walk files the matches.  Same match list as zh_chain_select_kernel.
template <bool kNarrow>
__global__ __launch_bounds__(256) void zh_chain_select_par_kernel(const uint8_t* __restrict__ d_src, ZhCompressArgs a,
                                                                  int good, int nice, int max_chain,
                                                                  const uint64_t* __restrict__ prevw,
                                                                  uint32_t* __restrict__ best) {
  constexpr uint32_t kT = 256, kChunk = ZH_FRAG_SIZE / kT;  // 128 positions per thread
  __shared__ uint8_t s_len[ZH_FRAG_SIZE];  // 0: literal, else match length - 4 (5..258 -> 1..254)
  __shared__ uint32_t s_end[kT];
  __shared__ uint32_t s_first_dirty[2];
  __shared__ uint32_t s_wsum[kT / 64];
  const uint32_t tid = threadIdx.x;
  const unsigned lane = zh_lane();
  const uint32_t b = a.first_block + blockIdx.x;
  const ZhBlockDesc bd = a.blocks[b];
  const uint32_t block_len = (uint32_t)bd.len;
  uint32_t* bst = best + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  const uint8_t* src = d_src + bd.src_off;
  typedef typename ChainRec<kNarrow>::T Rec;
  const Rec* pw = reinterpret_cast<const Rec*>(prevw) + (size_t)(bd.first_frag - a.first_frag) * ZH_FRAG_SIZE;
  // lz77.nim:54-56,74-76: the last four positions (and blocks of <= 4 bytes) are literals
  const uint32_t nmain = block_len > 4u ? block_len - 4u : 0u;
  uint32_t entry = 0;  // block-relative position at which the walk enters the next fragment

  for (uint32_t frag = 0; frag < bd.nfrag; frag++) {
    const uint32_t base = frag * ZH_FRAG_SIZE;
    const uint32_t npos = nmain > base ? (nmain - base < ZH_FRAG_SIZE ? nmain - base : ZH_FRAG_SIZE) : 0u;
    __syncthreads();  // (the previous fragment's walks are done with s_len)
    for (uint32_t i = tid * 4u; i < ZH_FRAG_SIZE; i += kT * 4u) {  // four entries a load, four bytes a store
      uint32_t v[4] = {kBestKnown, kBestKnown, kBestKnown, kBestKnown};
      if (i + 4u <= npos) {
        const uint4 q = *reinterpret_cast<const uint4*>(bst + base + i);
        v[0] = q.x;
        v[1] = q.y;
        v[2] = q.z;
        v[3] = q.w;
      } else {
        for (uint32_t k = 0; k < 4u; k++)
          if (i + k < npos) v[k] = bst[base + i + k];
      }
      uint32_t packed = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t len = v[k] & 0xffffu;
        packed |= (v[k] & kBestKnown ? (len ? len - 4u : 0u) : 255u) << (8u * k);  // 255: nobody came by here yet
      }
      *reinterpret_cast<uint32_t*>(&s_len[i]) = packed;
    }
    if (tid == 0) s_first_dirty[0] = s_first_dirty[1] = kT;
    __syncthreads();
    // walks are in fragment-relative positions; one that enters behind the fragment passes through
    const uint32_t rel0 = entry > base ? entry - base : 0u;
    const uint32_t limit = (tid + 1u) * kChunk;
    // match length - 4 at fragment position p (0: literal); a position no walk of kernel 2b visited
    // is worked out here and now (and kept: the other walks of this kernel may come by as well)
    auto len_at = [&](uint32_t p) -> uint32_t {
      uint32_t l = s_len[p];
      if (l == 255u) {
        const uint32_t r = zh_chain_search_one<kNarrow>(src, pw, base + p, block_len, good, nice, max_chain);
        __hip_atomic_store(bst + base + p, r | kBestKnown, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        l = (r & 0xffffu) ? (r & 0xffffu) - 4u : 0u;
        s_len[p] = (uint8_t)l;
      }
      return l;
    };
    auto walk = [&](uint32_t p, uint32_t* nmatch) -> uint32_t {
      uint32_t n = 0;
      while (p < limit && p < npos) {
        const uint32_t l = len_at(p);
        n += l != 0u;
        p += l ? l + 4u : 1u;
      }
      *nmatch = n;
      return p < limit && p >= npos ? limit : p;  // (behind the last walkable position: literals to the end)
    };
    uint32_t my_start = tid == 0 ? rel0 : tid * kChunk, my_end = 0, my_n = 0, prev_fd = 0;
    bool dirty = true;
    for (uint32_t turn = 1;; turn++) {
      const uint32_t par = turn & 1u;
      if (dirty) {
        my_end = my_start < limit ? walk(my_start, &my_n) : (my_n = 0, my_start);
        s_end[tid] = my_end;
      }
      __syncthreads();
      const uint32_t ns = tid == 0 ? rel0 : s_end[tid - 1u];
      dirty = ns != my_start;
      my_start = ns;
      if (dirty) atomicMin(&s_first_dirty[par], tid);
      if (tid == 0) s_first_dirty[par ^ 1u] = kT;
      __syncthreads();
      const uint32_t fd = s_first_dirty[par];
      if (fd == kT) break;
      const uint32_t gain = fd - prev_fd;
      prev_fd = fd;
      if (turn < 3u || gain >= 4u) continue;
      // the walks do not meet here: thread 0 carries the exact walk on from the last final end through
      // the next 16 chunks alone; what lies behind them may well be in step already
      const uint32_t upto = fd + 16u < kT ? fd + 16u : kT;
      if (tid == 0) {
        uint32_t p = s_end[fd - 1u];  // (thread 0 is never dirty: fd >= 1)
        for (uint32_t t = fd; t < upto; t++) {
          const uint32_t lim = (t + 1u) * kChunk;
          while (p < lim && p < npos) {
            const uint32_t l = len_at(p);
            p += l ? l + 4u : 1u;
          }
          if (p < lim) p = lim;  // (behind the last walkable position)
          s_end[t] = p;
        }
      }
      __syncthreads();
      const uint32_t ns2 = tid == 0 ? rel0 : s_end[tid - 1u];
      if (ns2 != my_start) {
        my_start = ns2;
        dirty = true;
      }
      prev_fd = upto - 1u;
    }
    // every start is exact: file the matches of this fragment in order
    const uint32_t incl = zh_wave_scan(my_n);
    if (lane == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - my_n, total = 0;
    for (uint32_t w = 0; w < kT / 64u; w++) {
      const uint32_t ws = s_wsum[w];
      if (w < (tid >> 6)) before += ws;
      total += ws;
    }
    const uint32_t f = bd.first_frag + frag;
    if (tid == 0) a.f_nmatch[f] = total;
    {
      const size_t slot0 = (size_t)f * ZH_MAX_MATCHES_PER_FRAG + before;
      uint32_t k = 0, p = my_start;
      while (p < limit && p < npos) {
        const uint32_t l = s_len[p];  // (the last walk came by here: it is worked out)
        if (l) {
          const uint32_t len = l + 4u;
          a.m_pos[slot0 + k] = (uint16_t)p;
          a.m_len[slot0 + k] = (uint16_t)len;
          a.m_off[slot0 + k] = (uint16_t)((__hip_atomic_load(bst + base + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16) & 0x7fffu);
          k++;
          const uint32_t end = p + len;  // a match that reaches into the next fragment (lz77.nim's
          if (end > ZH_FRAG_SIZE && frag + 1u < bd.nfrag) a.f_spill[f + 1u] = end - ZH_FRAG_SIZE;  // spill)
          p = end;
        } else {
          p++;
        }
      }
    }
    entry = base + s_end[kT - 1u];
    if (frag == 0 && tid == 0) a.f_spill[f] = 0;
    if (tid == 1 && frag + 1u < bd.nfrag && entry <= base + ZH_FRAG_SIZE) a.f_spill[f + 1u] = 0;
  }
}

// Per-fragment statistics from a match list: litlen/distance histograms,
// literal count, sum of extra bits (lz77.nim:19-50 / snappy.nim:33-64).
__global__ __launch_bounds__(64) void zh_frag_stats_kernel(const uint8_t* __restrict__ d_src,
                                                           ZhCompressArgs a) {
  __shared__ uint32_t s_hist[ZH_HIST_STRIDE];
  __shared__ uint32_t s_cover[ZH_FRAG_SIZE / 32];
  const unsigned lane = zh_lane();
  const uint32_t f = a.first_frag + blockIdx.x;
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
      const uint32_t li = zh_len_code(l), di = zh_dist_code(o);
      atomicAdd(&s_hist[257 + li], 1u);
      atomicAdd(&s_hist[ZH_NUM_LITLEN + di], 1u);
      extra_bits += zh_len_extra_bits(li) + zh_dist_extra_bits(di);
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
  for (uint32_t base = 0; base < n; base += 256) {  // four positions a lane: one load, one word of the bitmap
    const uint32_t p0 = base + 4u * lane;
    if (p0 < n) {
      uint32_t w = 0, live = 15u;
      if (p0 + 4u <= n) {
        w = load32u(src + p0);
      } else {
        live = (1u << (n - p0)) - 1u;
        for (uint32_t k = 0; k < n - p0; k++) w |= (uint32_t)src[p0 + k] << (8u * k);
      }
      live &= ~(s_cover[p0 >> 5] >> (p0 & 31u));
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++)
        if ((live >> k) & 1u) atomicAdd(&s_hist[(w >> (8u * k)) & 255u], 1u);
      nlit += (uint32_t)__popc(live);
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

// Up to this many blocks the exchanging kernel with its 512 KiB tables, beyond it the load / store
// kernel with 256 KiB ones (the plan's scratch is sized for either).
extern "C" uint32_t zh_chain_prev_slice(void) { return 1024u; }
// ---- the hardware property kernels 1c take their ORDER from, asked of the device itself ----
// "The lanes of one returning LDS atomic are served in ascending lane order" is what makes an atomicAdd a
// stable rank and an atomicMax "the previous occupant of my slot, in position order" (zh_chain_class_kernel,
// zh_chain_class_links_kernel).  The ISA manual does not promise it; a device that did otherwise would still
// produce valid deflate (candidates are compared byte by byte), only not the reference's bytes.  So every
// device answers a known-answer probe once, when the first context on it is made (zh_create): 64 rounds of
// both atomics on few and many slots, with all and with some lanes active, each lane's answer compared with
// the one the order implies.  A device that fails it gets the in-order kernels (ZH_CHAIN_PREV=serial's).
__global__ __launch_bounds__(64) void zh_lds_order_probe_kernel(uint32_t* __restrict__ out) {
  __shared__ uint32_t s_cnt[64], s_max[64];
  const unsigned lane = zh_lane();
  uint32_t bad = 0;
  for (uint32_t round = 0; round < 64u; round++) {
    s_cnt[lane] = 0;
    if (round == 0) s_max[lane] = 0;
    zh_wave_sync();
    const uint32_t nslots = 1u + ((round * 11u) & 63u);                       // 1 .. 64 slots in use
    const uint32_t h = ((lane * 2654435761u + round * 40503u) >> 7) % nslots;  // this lane's slot
    const bool active = (round & 3u) != 3u || ((lane * 7u + round) & 3u) != 0u;  // every fourth round: three lanes in four
    // what the order implies: the active lower lanes of my slot -- how many, and the highest
    uint32_t below = 0, last = 64;
    for (uint32_t j = 0; j < 64u; j++) {
      const uint32_t hj = (uint32_t)__shfl((int)h, (int)j, 64);
      const bool aj = __shfl((int)active, (int)j, 64) != 0;
      if (aj && hj == h && j < lane) {
        below++;
        last = j;
      }
    }
    const uint32_t before = s_max[h];  // (what earlier rounds left in my slot)
    zh_wave_sync();
    if (active) {
      const uint32_t rank = atomicAdd(&s_cnt[h], 1u);
      const uint32_t prev = atomicMax(&s_max[h], round * 64u + lane + 1u);
      bad |= rank != below;
      bad |= prev != (last < 64u ? round * 64u + last + 1u : before);
    }
    zh_wave_sync();
  }
  const uint64_t anybad = __ballot(bad != 0);
  if (lane == 0) out[0] = anybad ? 2u : 1u;
}
static std::atomic<int> g_lds_order[64];  // per device: 0 not asked, 1 in lane order, 2 not
// -> 1: this device serves the lanes of an LDS atomic in ascending order (kernels 1c are exact), 0: it does not
// (or could not be asked): the in-order kernels run.  Synchronises `stream` the first time a device is asked.
extern "C" int zh_chain_lds_order_ok(int device, hipStream_t stream) {
  if (device < 0 || device >= 64) return 0;
  int v = g_lds_order[device].load();
  if (v == 0) {
    uint32_t* d = nullptr;
    uint32_t h = 0;
    if (hipMalloc(&d, 4) == hipSuccess) {
      if (hipMemsetAsync(d, 0, 4, stream) == hipSuccess) {
        hipLaunchKernelGGL(zh_lds_order_probe_kernel, dim3(1), dim3(64), 0, stream, d);
        if (hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
          h = 0;
      }
      (void)hipFree(d);
    }
    v = h == 1u ? 1 : 2;
    const char* e = getenv("ZH_LDS_ORDER_PROBE");  // test aid: "fail" pretends the device answered otherwise
    if (e && strcmp(e, "fail") == 0) v = 2;
    g_lds_order[device].store(v);
  }
  return v == 1;
}
// `links_serial` (zh_ctx::chain_links_serial): the in-order kernels 1 / 1b (a wave a block) instead of 1c -- what a device
// gets that failed the probe above (the decision is the context's, made once at zh_create for ITS device; the test
// build -DZH_XCHECK also takes it from ZH_CHAIN_PREV=serial, as a cross-check).
// `lists`: 4 bytes a position of scratch (the plan lends best[], which the walks clear before they use it)
// The cross-checks of the test build (-DZH_XCHECK; the product library has neither the switches nor the kernels):
// ZH_CHAIN_SEARCH=dense: the best match of EVERY position (kernel 2a) instead of the walks of kernel 2b;
// ZH_CHAIN_SELECT=serial (one wave per block, needs the dense search).
#ifdef ZH_XCHECK
static bool chain_select_serial() {
  static const bool on = [] {
    const char* e = getenv("ZH_CHAIN_SELECT");
    return e && strcmp(e, "serial") == 0;
  }();
  return on;
}
static bool chain_search_dense() {
  static const bool on = [] {
    const char* e = getenv("ZH_CHAIN_SEARCH");
    return e && strcmp(e, "dense") == 0;
  }();
  return on || chain_select_serial();
}
#else
static constexpr bool chain_select_serial() { return false; }
static constexpr bool chain_search_dense() { return false; }
#endif
// narrow records (ChainRec): where the links come from the class kernels and the level's `good` is at least five, and
// no cross-check of the test build asks for the every-position search (which reads wide records)
static bool chain_narrow(int links_serial, int good) { return !links_serial && good >= 5 && !chain_search_dense(); }
extern "C" void zh_launch_chain_prev(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a,
                                     uint32_t* head_scratch, uint64_t* prevw, uint32_t* lists, int links_serial, int good) {
  if (!a.nblocks) return;
  if (!links_serial) {
    // (head_scratch holds at least 256 KiB a block: kClsStride words)
    const uint32_t ngrid = a.nfrags * (kUnitsPerFrag / kClsWaves);
    hipLaunchKernelGGL(zh_chain_class_kernel<false>, dim3(ngrid), dim3(64 * kClsWaves), 0, stream, d_src, a, head_scratch, lists);
    hipLaunchKernelGGL(zh_chain_class_scan_kernel, dim3(a.nblocks), dim3(512), 0, stream, a, head_scratch);
    hipLaunchKernelGGL(zh_chain_class_kernel<true>, dim3(ngrid), dim3(64 * kClsWaves), 0, stream, d_src, a, head_scratch, lists);
    const uint32_t ng = a.nblocks * kClasses;
    constexpr uint32_t kGroup = ZH_CHAIN_LINKS_GROUP;
    const uint32_t nwg = ng / kGroup;
    if (chain_narrow(links_serial, good))
      hipLaunchKernelGGL((zh_chain_class_links_kernel<true, kGroup>), dim3((nwg + 7u) & ~7u), dim3(64 * kGroup), 0, stream, d_src, a,
                         head_scratch, lists, prevw, ng);
    else
      hipLaunchKernelGGL((zh_chain_class_links_kernel<false, kGroup>), dim3((nwg + 7u) & ~7u), dim3(64 * kGroup), 0, stream, d_src, a,
                         head_scratch, lists, prevw, ng);
    return;
  }
  const uint32_t slice = zh_chain_prev_slice();
  if (a.nblocks <= slice) {
    (void)hipMemsetAsync(head_scratch, 0, (size_t)a.nblocks * ZH_CHAIN_HEAD_WORDS * 4u, stream);
    hipLaunchKernelGGL(zh_chain_prev_kernel, dim3(a.nblocks), dim3(64), 0, stream, d_src, a, head_scratch, prevw, a.first_block);
    return;
  }
  // (all blocks at once: this form lives on the number of loads in flight)
  (void)hipMemsetAsync(head_scratch, 0, (size_t)a.nblocks << (kHashBits + 1), stream);
  hipLaunchKernelGGL(zh_chain_prev_ldst_kernel, dim3(a.nblocks), dim3(64), 0, stream, d_src, a,
                     reinterpret_cast<uint16_t*>(head_scratch), prevw, a.first_block);
}
extern "C" void zh_launch_chain_search(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a,
                                       int good, int nice, int max_chain, const uint64_t* prevw,
                                       uint32_t* best, int links_serial) {
  // a launch takes at most 2^30 positions (a grid of 2^32 threads or more is refused), so batches
  // of 1 GiB and more go in slices
  constexpr uint32_t kSlice = 32768;  // fragments per launch
  for (uint32_t f0 = 0; f0 < a.nfrags; f0 += kSlice) {
    const uint32_t nf = a.nfrags - f0 < kSlice ? a.nfrags - f0 : kSlice;
#ifdef ZH_XCHECK
    if (chain_search_dense()) {
      hipLaunchKernelGGL(zh_chain_search_kernel, dim3(nf * (ZH_FRAG_SIZE / 256u)), dim3(256), 0, stream,
                         d_src, a, good, nice, max_chain, prevw, best, a.first_frag + f0);
    } else
#endif
    {
      // (nothing is worked out yet -- the walks of one launch may look at the next launch's entries --: the
      // class-sorted links have left best[] cleared; after the in-order kernels it still holds the last run's)
      if (f0 == 0 && links_serial) (void)hipMemsetAsync(best, 0, (size_t)a.nfrags * ZH_FRAG_SIZE * 4u, stream);
      constexpr uint32_t kChunk = ZH_CHAIN_CHUNK;
      const uint32_t ng = nf * (ZH_FRAG_SIZE / kChunk / kWalkThreads);
      if (chain_narrow(links_serial, good))
        hipLaunchKernelGGL((zh_chain_walk_kernel<kChunk, true>), dim3((ng + 7u) & ~7u), dim3(kWalkThreads), 0, stream, d_src, a,
                           good, nice, max_chain, prevw, best, a.first_frag + f0, ng);
      else
        hipLaunchKernelGGL((zh_chain_walk_kernel<kChunk, false>), dim3((ng + 7u) & ~7u), dim3(kWalkThreads), 0, stream, d_src, a,
                           good, nice, max_chain, prevw, best, a.first_frag + f0, ng);
    }
  }
}
extern "C" void zh_launch_chain_select(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a, int good,
                                       int nice, int max_chain, const uint64_t* prevw, uint32_t* best, int links_serial) {
  if (!a.nblocks) return;
#ifdef ZH_XCHECK
  if (chain_select_serial()) {
    hipLaunchKernelGGL(zh_chain_select_kernel, dim3(a.nblocks), dim3(64), 0, stream, a, best);
    return;
  }
#endif
  if (chain_narrow(links_serial, good))
    hipLaunchKernelGGL(zh_chain_select_par_kernel<true>, dim3(a.nblocks), dim3(256), 0, stream, d_src, a, good, nice,
                       max_chain, prevw, best);
  else
    hipLaunchKernelGGL(zh_chain_select_par_kernel<false>, dim3(a.nblocks), dim3(256), 0, stream, d_src, a, good, nice,
                       max_chain, prevw, best);
}
extern "C" void zh_launch_frag_stats(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a) {
  if (!a.nfrags) return;
  hipLaunchKernelGGL(zh_frag_stats_kernel, dim3(a.nfrags), dim3(64), 0, stream, d_src, a);
}
