// BestSpeed (level 1) matcher: one 64-lane wave parses one <= 32 KiB fragment.
//
// Replaces snappy.nim:12-136 encodeFragment (+ the addLiteral/addCopy
// bookkeeping of snappy.nim:33-64) and deflate.nim:153-177 encodeAllLiterals
// (level -2, `huffman_only`).  The parse is the reference's greedy single-probe
// parse, reproduced decision for decision (same hash, same table size rule,
// same skip-ahead schedule, same "no match starts in the last 15 bytes" rule),
// so the match list equals the reference's token stream fragment by fragment
// (tests compare them token for token through zh_debug_tokens).
//
// Each wave keeps its u16 hash table (32 KiB, snappy.nim:7) in a pooled, L2/MALL-resident slot
// of HBM scratch and only 7.4 KiB in LDS (per-step slot counters, reused as the coverage bitmap
// afterwards; the slot-written bitmap; the symbol histograms): 20 waves per CU.  A lone wave runs
// at well under a tenth of its SIMD's issue rate (every step is a chain of dependent round
// trips), so waves per CU, not latency per access, is what buys throughput.  The fragment's own
// bytes and the candidates' are read through L1/L2 with unaligned 128-bit loads.
// Output per fragment (HBM scratch): the match list (start, length, offset as
// u16 SoA), the litlen/distance histograms (u16 x 320), literal count and the
// sum of extra bits -- everything the Huffman and emission kernels need.
#include <cstdlib>
#include <cstring>

#include "zh_common.h"
#include "zh_kprof.h"
#include "zh_tables.h"

namespace {
constexpr uint32_t kHashMul = 0x1e35a7bdu;  // snappy.nim:70-71
// byte-wide per-step slot counters, four to a dword, keyed by the low hash bits (also the size
// of the coverage bitmap that reuses the array: 1024 words = 32768 bits)
constexpr uint32_t kCntWords = 1024;
// 16 bytes at any byte address (gfx950 global loads need no alignment)
struct __attribute__((packed)) Bytes16 {
  uint32_t x, y, z, w;
};

}  // namespace

// kLdsTable: the wave's hash table in LDS (32 KiB: four waves per CU, no table traffic at all)
// instead of the HBM pool.  A measurement aid for the trade DESIGN.md 4.1 describes -- with one
// wave per fragment the pooled form is the faster one by far -- selected with ZH_L1_TABLE=lds.
template <bool kLdsTable>
__global__ __launch_bounds__(64) void zh_l1_match_kernel(const uint8_t* __restrict__ d_src,
                                                         ZhCompressArgs a, int huffman_only,
                                                         uint16_t* __restrict__ table_pool,
                                                         uint32_t* __restrict__ next_frag,
                                                         const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ cost) {
  __shared__ uint32_t s_hist[ZH_HIST_STRIDE];
  // parse: 4096 byte-wide counters (4 per dword) of the probes per table slot in one step,
  // all zero between steps; afterwards the first 4 KiB are the coverage bitmap (bit p set:
  // byte p lies inside a match, behind its first byte)
  __shared__ uint32_t s_scr[kCntWords];
  // one bit per table slot: written while parsing this fragment.  A slot that was not holds the
  // reference's initial zero, which needs no load -- and the pooled table needs no clearing.
  __shared__ uint32_t s_bits[512];
  __shared__ uint32_t s_nmatch;
  uint32_t* const s_cover = s_scr;

  const unsigned lane = zh_lane();
  // this wave's hash table (u16 x 16384, snappy.nim:7) in the L2/MALL-resident pool: read with
  // L1-bypassing loads, written through, re-zeroed for every fragment the wave takes
  __shared__ uint16_t s_table_lds[kLdsTable ? 16384 : 2];
  uint16_t* const s_table = kLdsTable ? s_table_lds : table_pool + (size_t)blockIdx.x * 16384u;
  // fragments are handed out first come, first served (`next_frag` starts at gridDim.x): they cost
  // very different amounts of time, and a fixed share per wave leaves the last ones running alone
  // ... and from a plan's second run on the cheapest ones last: `order` (null: as numbered) is the fragments as numbered
  // but for the cheapest two rounds' worth, which go to the end -- by what they cost the run before (`cost`: this
  // run's, for the next; zh_launch_l1_match) --, so that what is still running when waves fall idle is short.  Which
  // wave takes which fragment when decides nothing of what comes out.
  for (uint32_t ticket = blockIdx.x; ticket < a.nfrags;) {
  const uint32_t f = order ? order[ticket] : ticket;
  const uint64_t t_start = zh_clock();
  KPROF_DECL(16);  // cycles: 0 stage-in, 1 vector part, 2 fast walk, 3 slow walk, 4 inserts, 5 stats; counts: 6..12
  const ZhFragDesc fd = a.frags[f];
  const uint32_t n = fd.len;
  const uint8_t* src = d_src + fd.src_off;
  uint16_t* m_pos = a.m_pos + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  uint16_t* m_len = a.m_len + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
  uint16_t* m_off = a.m_off + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;

  // ---- the fragment's bytes: aligned dwords of the stream below `src`, never past the
  // dword that holds its last byte ----
  const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
  const uint32_t* asrc = reinterpret_cast<const uint32_t*>(src - mis);
  const uint32_t last_dw = n ? (n + mis - 1u) >> 2 : 0u;
  auto dw = [&](uint32_t i) -> uint32_t { return asrc[i < last_dw ? i : last_dw]; };
  auto ld32 = [&](uint32_t p) -> uint32_t {  // the 4 bytes at fragment offset p (p + 4 <= n + 3)
    const uint32_t q = p + mis, i = q >> 2;
    return __builtin_amdgcn_alignbyte(dw(i + 1), dw(i), q);
  };
  for (uint32_t i = lane; i < ZH_HIST_STRIDE; i += 64) s_hist[i] = 0;
  for (uint32_t i = lane; i < kCntWords; i += 64) s_scr[i] = 0;

  uint32_t table_size = 256, shift = 24;  // snappy.nim:24-29
  while (table_size < 16384u && table_size < n) {
    table_size <<= 1;
    shift--;
  }
  // (Round 6, two probes of what this kernel waits for, both byte-identical, both measured and taken out again.
  // READS: a SECOND tag bit a slot kept in LDS (2 KiB more a wave, 17 waves a CU) and looked at BEFORE the table, so that a
  // probe whose bit differs skips the 128-byte read of an entry that could only say "no": the fabric's read requests fell
  // 645 M -> 529 M a GiB of input (- 18 %) and the kernel took 62.9-66.1 ms against 64.1-64.5.  WRITES: match records
  // gathered in an LDS ring and written 64 at a time as whole lines instead of a partial write a batch: write requests
  // 455 M -> 365 M (- 20 %), 62.0 / 62.1 / 64.7 ms against 63.5 / 63.6 / 63.4, and one GPU's share 8.9 -> 9.8 ms (the flush
  // sits in the walk).  Neither the reads nor the writes are what a step waits for: it is the three DEPENDENT trips through
  // a CU's in-order vector memory pipeline, each as long as the pipeline's oldest miss, and the walk behind them.
  // ... and the first of the three is not what it seems either: the NEXT step's source bytes asked for before this step's
  // walk -- 1 KiB a wave in registers, wherever the walk ends the step behind it reads inside it, the lanes' 16 bytes cut
  // out by eight ds_bpermute -- took the trip off the chain and made the kernel SLOWER, 64.4 -> 68.95 ms (a share 9.07 ->
  // 9.49, config 2 2.34 -> 2.49): the L1 serves that load quickly enough, as round 1's LDS ring had already said.
  // profiles/r06_h_*, r06_i_*, r06_j_*.)
  // A table entry is position | tag << 15: the tag is one more bit of the hash product of the
  // position's four bytes, so a probe whose tag differs cannot match and need not fetch the
  // candidate's bytes (half of the non-matching gathers, which are what this kernel's memory
  // traffic is made of).  "Empty" is the reference's zero = position 0, with position 0's tag (e0).
  const uint32_t e0 = (!huffman_only && n >= 4) ? (((ld32(0) * kHashMul) >> 17) & 1u) << 15 : 0u;
  for (uint32_t i = lane; i < 512; i += 64) s_bits[i] = 0;
  zh_wave_sync();
  KPROF_MARK(0);

  // ---- greedy parse, wave-parallel (default) ----
  // The reference probes one position at a time: h = hash(load32(p)); cand = table[h];
  // table[h] = p; hit iff load32(p) == load32(cand) (snappy.nim:86-101), and after a
  // match it inserts ip-1, re-probes ip and restarts the skip schedule (snappy.nim:108-131).
  //
  // A step lays the 64 lanes over upcoming probe positions and does the parse-independent
  // work for all of them at once in three dependent LDS round trips: 16 source bytes and
  // the hash, the table slot (`old`, the candidate if no probe of this step re-uses the
  // slot first), the 16 candidate bytes and their common prefix length.  A wave-uniform
  // walk (scalar registers: ballots, s_ff1, v_readlane) then replays the reference's
  // decisions over those lanes in probe order.
  //   dense step  (fewer than 32 probes into the literal run, the usual case): the lanes
  //     are 64 CONSECUTIVE positions, so the walk can carry on behind a match -- mark
  //     ip-1 inserted, re-probe ip, restart the run -- and typically retires several
  //     matches per step;
  //   sparse step (>= 32 misses in a row: the reference probes every 2nd, 3rd ... byte,
  //     snappy.nim:88-92): the lanes are the next 64 positions of that schedule in closed
  //     form; the step ends at its first match.
  // A lane's true candidate is the last position inserted before it with the same hash:
  // a lane of this step if one shares its hash and was really inserted (tracked in the
  // `ins` mask; lanes that share a table slot are flagged through an LDS counter and
  // resolved exactly, one by one), else `old`.  Only the lanes the reference would have
  // probed or inserted write the table, in probe order, so the table evolves exactly as
  // in the serial walk and the match list equals the reference's token stream.
  {
    uint32_t nm = 0;
    if (!huffman_only && n >= 15) {
      const uint32_t ip_limit = n - 15;
      uint32_t ip = 1;     // position of the next probe (post: position right after a match)
      uint32_t K = 0;      // probes already done in this literal run (skip = 32 + K)
      bool post = false;   // a match has just ended at ip: insert ip-1, then re-probe ip
      bool finished = false;
      while (!finished) {
        const bool dense = post || K < 32u;
        uint32_t pos, step, W0 = 0;
        if (dense) {
          W0 = post ? ip - 1u : ip;
          pos = W0 + lane;
          step = 1;
        } else {  // closed form of the skip>>5 schedule, snappy.nim:88-92
          const uint32_t s = 32 + K, q = s >> 5, r = s & 31u, j = lane;
          uint32_t off = q * j;
          if (j > 32u - r) off += j - (32u - r);
          if (j > 64u - r) off += j - (64u - r);
          pos = ip + off;
          step = (s + j) >> 5;
        }
        const bool valid = pos + step <= ip_limit;  // the reference's `nextIp > ipLimit` test
        if (!valid) pos = 1;  // keep LDS reads in range; the lane is never walked
        // round trip 1: the 16 source bytes at pos, one unaligned 128-bit load (a valid lane has
        // pos + 16 <= n, and so has every candidate, which lies before it)
        Bytes16 av = {0, 0, 0, 0};
        if (valid) av = *reinterpret_cast<const Bytes16*>(src + pos);
        const uint32_t a0 = av.x, a1 = av.y, a2 = av.z, a3 = av.w;
        const uint32_t hp = a0 * kHashMul;
        const uint32_t h = hp >> shift, tag = (hp >> 17) & 1u;
        // round trip 2: the table slot; every lane also ticks a counter of its (folded) hash,
        // read back together with the candidate bytes
        uint32_t oldw = e0;
        if (valid && ((s_bits[h >> 5] >> (h & 31u)) & 1u))
          oldw = kLdsTable ? (uint32_t)s_table[h]
                           : (uint32_t)__hip_atomic_load(s_table + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t old = oldw & 0x7fffu;
        const bool fetch = valid && (oldw >> 15) == tag;  // equal bytes have equal tags
        const uint32_t ck = (h & (kCntWords * 4u - 1u)) >> 2, cs = (h & 3u) * 8u;
        if (valid) atomicAdd(&s_scr[ck], 1u << cs);
        zh_wave_sync();  // (orders the counter traffic between lanes; emits nothing)
        // round trip 3: 16 bytes at the candidate the table held when the step began -- one
        // scattered 128-bit load per lane (one L1 tag lookup instead of five)
        Bytes16 qv = {0, 0, 0, 0};
        if (fetch) qv = *reinterpret_cast<const Bytes16*>(src + old);
        const uint32_t cnt = valid ? (s_scr[ck] >> cs) & 255u : 0u;
        zh_wave_sync();
        if (valid) s_scr[ck] = 0;
        const uint32_t x0 = fetch ? a0 ^ qv.x : 1u;
        const uint32_t x1 = a1 ^ qv.y;
        const uint32_t x2 = a2 ^ qv.z;
        const uint32_t x3 = a3 ^ qv.w;
        // equal leading bytes 0..16, branch-free: ffs(0) - 1 = 0xffffffff -> min(.., 4) = 4
        const uint32_t c0 = min(((uint32_t)__ffs((int)x0) - 1u) >> 3, 4u);
        const uint32_t c1 = min(((uint32_t)__ffs((int)x1) - 1u) >> 3, 4u);
        const uint32_t c2 = min(((uint32_t)__ffs((int)x2) - 1u) >> 3, 4u);
        const uint32_t c3 = min(((uint32_t)__ffs((int)x3) - 1u) >> 3, 4u);
        const uint32_t c23 = c2 + (c2 == 4u ? c3 : 0u);
        const uint32_t c123 = c1 + (c1 == 4u ? c23 : 0u);
        const uint32_t eqlen = c0 + (c0 == 4u ? c123 : 0u);

        const uint64_t V = __ballot(valid);  // a prefix of the lanes (pos + step is monotone)
        const uint64_t H = __ballot(valid && x0 == 0);
        // lanes that may share their table slot with another lane of this step (a superset:
        // the counters see 12 of the 14 hash bits); the general walk below sorts them out exactly
        const uint64_t C = __ballot(cnt > 1u);
        // Cw: the flagged lanes that really have an earlier lane with their hash in this step --
        // only those can have a candidate other than `old` (the first lane of a group, and lanes
        // flagged by the folded counter alone, walk like any other lane)
        uint64_t Cw = C;
        if (C && __popcll(C) <= 16) {
          Cw = 0;
          for (uint64_t cm = C; cm; cm &= cm - 1) {
            const uint32_t g = (uint32_t)__ffsll((long long)cm) - 1u;
            const uint32_t hg = __builtin_amdgcn_readlane(h, g);
            const bool same_h = valid && h == hg;
            if (__ballot(same_h) & ((1ull << g) - 1ull)) Cw |= 1ull << g;
          }
        }
        const uint32_t t = V == ~0ull ? 64u : (uint32_t)__ffsll((long long)~V) - 1u;  // first probe past ip_limit

        KPROF_MARK(1);
        KPROF_COUNT(6, 1);
#ifndef ZH_EMU
        __builtin_amdgcn_s_setprio(1);  // the walk is a dependent scalar chain: first pick of the issue slots
#endif
        // ---- the walk: wave-uniform replay of the reference's decisions ----
        uint64_t ins = post ? 1ull : 0ull;  // lanes whose position was inserted, in probe order
        uint32_t i = post ? 1u : 0u;        // next lane to probe
        bool reprobe = post;                // the probe at lane i is the re-probe that follows a match
        // Hop table (dense steps): a probe that arrives at lane j runs into the first lane
        // g >= j that is a hit or shares a table slot; if g is a plain hit (candidate `old`,
        // all of the match inside the 16 compared bytes) and the run cannot reach its 32nd
        // probe on the way, the walk resumes at T = g + length.  Every lane works out its
        // own (g, T) -- one cross-lane fetch -- so the wave-uniform walk is a chain of
        // v_readlane hops, one per match, in a loop small enough to stay in the instruction
        // buffer (a taken branch to other code costs a lone wave ~50 cycles, the loop edge
        // next to nothing); anything else stops the chain for one turn of the general walk.
        // hop word: bits 0-5 lane g of the match (or 63), bit 6 "there is a match at g",
        // bits 8-14 the lane the walk resumes at, bit 15 "not for the chain".  Besides matches
        // the chain can take "nothing but misses up to lane 63": the literal run then goes on
        // in the next step (only from lanes >= 32, so that it cannot reach its 32nd probe here).
        uint32_t hop = 0x8000u;
        if (dense) {
          const uint64_t ev = (H | Cw) >> lane;
          const uint32_t d = ev ? (uint32_t)__ffsll((long long)ev) - 1u : 64u;
          const uint32_t info = eqlen | ((uint32_t)((Cw >> lane) & 1ull) << 5);
          const uint32_t gi = (uint32_t)__shfl((int)info, (int)((lane + d) & 63u), 64);
          if (d < 32u && gi < 16u) {
            // (the walk's bookkeeping, worked out here once per lane instead of once per hop:
            // bits 16-21 = number of probes d + 1, bits 22-27 = the lane of ip-1 behind the match
            // if it is in this step, else the match lane again)
            const uint32_t g0 = lane + d, nx = g0 + gi;
            hop = g0 | 0x40u | (nx << 8) | ((d + 1u) << 16) | ((nx <= 64u ? nx - 1u : g0) << 22);
          } else if (!ev && lane >= 32u) {
            hop = 63u | (64u << 8);
          }
        }
        const uint32_t tt = ip_limit > W0 ? ip_limit - W0 : 0u;  // dense: first lane past ip_limit
        for (;;) {
          if (dense && i < 64u) {
            uint64_t sel = 0;  // match lanes of this chain
            uint32_t cur = zh_bcast(i), tv = 0x8000u;
            KPROF_MARK(3);
            // a run that continues from the previous step may only hop if it finds its match
            // before its 32nd probe (hops behind a match need d < 32 only, which the table has)
            bool go = true;
            if (!reprobe && K) {
              tv = __builtin_amdgcn_readlane(hop, cur);
              go = !(tv & 0x8000u) && (tv & 0x40u) && K + ((tv & 63u) - cur) <= 31u;
            }
            const uint32_t lim = zh_bcast(tt < 64u ? tt : 64u);
            if (go && cur < lim) {
#ifdef ZH_EMU
              do {
                tv = __builtin_amdgcn_readlane(hop, cur);
                if ((tv & 0x8040u) != 0x40u) break;  // not a match hop
                const uint32_t g = tv & 63u, nxt = (tv >> 8) & 127u;
                sel |= 1ull << g;
                ins |= (~0ull << cur) & (~0ull >> (63u - g));  // probes cur..g (table[h] = ip)
                ins |= 1ull << (nxt <= 64u ? nxt - 1u : g);     // ip-1 behind the match (snappy.nim:126)
                cur = nxt;
              } while (cur < lim);
#else
              // the same loop in 12 instructions with one taken branch per match
              uint32_t t0, g, nxt;
              uint64_t m1;
              asm volatile(
                  "1:\n\t"
                  "v_readlane_b32 %[tv], %[hop], %[cur]\n\t"
                  "s_bitcmp1_b32 %[tv], 6\n\t"
                  "s_cbranch_scc0 2f\n\t"
                  "s_and_b32 %[g], %[tv], 63\n\t"
                  "s_bitset1_b64 %[sel], %[g]\n\t"
                  "s_bfe_u32 %[t0], %[tv], 0x60010\n\t"
                  "s_bfm_b64 %[m1], %[t0], %[cur]\n\t"
                  "s_or_b64 %[ins], %[ins], %[m1]\n\t"
                  "s_bfe_u32 %[t0], %[tv], 0x60016\n\t"
                  "s_bitset1_b64 %[ins], %[t0]\n\t"
                  "s_bfe_u32 %[cur], %[tv], 0x70008\n\t"
                  "s_cmp_lt_u32 %[cur], %[lim]\n\t"
                  "s_cbranch_scc1 1b\n\t"
                  "2:"
                  : [tv] "=&s"(tv), [cur] "+s"(cur), [sel] "+s"(sel), [ins] "+s"(ins), [t0] "=&s"(t0),
                    [g] "=&s"(g), [nxt] "=&s"(nxt), [m1] "=&s"(m1)
                  : [hop] "v"(hop), [lim] "s"(lim)
                  : "scc");
#endif
            }
            const bool chained = cur != i;  // at least one match: the probe at cur is a re-probe
            // "nothing but misses up to lane 63" (a hop word without match and stop flags):
            // the literal run goes on in the next step
            bool run_out = false;
            if (go && cur < lim && !(tv & 0x8040u) && tt >= 64u) {
              K = (chained || reprobe ? 63u : K + 64u) - cur;  // probes cur+1..63, or 64-cur more
              ins |= ~0ull << cur;
              run_out = true;
            }
            KPROF_MARK(2);
            if (sel) {
              KPROF_COUNT(8, __popcll(sel));
              if ((sel >> lane) & 1ull) {  // every match lane files its own record
                const uint32_t k = nm + (uint32_t)__popcll(sel & zh_lanemask_lt());
                zh_store_nt(m_pos + k, (uint16_t)pos);  // (streaming: keep the records out of the tables' way)
                zh_store_nt(m_len + k, (uint16_t)eqlen);
                zh_store_nt(m_off + k, (uint16_t)(pos - old));
              }
              nm += (uint32_t)__popcll(sel);
            }
            if (run_out) {
              post = false;
              ip = W0 + 64u;
              break;
            }
            if (chained) {
              K = 0;
              if (cur >= tt) {  // snappy.nim:118-120
                finished = true;
                break;
              }
              if (cur >= 64u) {
                post = true;
                ip = W0 + cur;
                break;
              }
              i = cur;
              reprobe = true;
            }
          }
          KPROF_COUNT(9, 1);
          if (i >= t) {  // snappy.nim:93-95 / 118-120: the rest of the fragment is literals
            finished = true;
            break;
          }
          uint32_t g = i;
          if (!reprobe) {
            // the run's next probes are lanes i, i+1, ...: find the first that can be a hit
            const uint32_t last_dense = dense ? i + (31u - K) : 63u;  // lane of probe #31 of the run
            uint32_t lim = last_dense < 63u ? last_dense : 63u;
            if (t - 1u < lim) lim = t - 1u;
            const uint64_t E = (H | Cw) & (~0ull << i);
            g = E ? (uint32_t)__ffsll((long long)E) - 1u : 64u;
            if (g > lim) {  // lanes i..lim all miss
              KPROF_COUNT(12, 1);
              ins |= (~0ull << i) & (~0ull >> (63u - lim));
              if (dense && lim == last_dense) {  // probe #32 on: the sparse schedule takes over
                K = 32;
                ip = W0 + lim + 1u;
                post = false;
              } else if (lim + 1u == t && t < 64u) {  // the next probe is past ip_limit
                finished = true;
              } else {  // lim == 63: the run continues in the next step
                K += 64u - i;
                post = false;
                if (dense) {
                  ip = W0 + 64u;
                } else {
                  ip = __builtin_amdgcn_readlane(pos, 63) + __builtin_amdgcn_readlane(step, 63);
                }
              }
              break;
            }
            if (g > i) ins |= (~0ull << i) & ((1ull << g) - 1ull);
            K += g - i;
          }
          // ---- probe lane g: its candidate is the last same-hash insert before it ----
          uint32_t cand, flen;
          bool hit;
          if ((Cw >> g) & 1ull) {
            KPROF_COUNT(14, 1);
            const uint32_t hg = __builtin_amdgcn_readlane(h, g);
            const uint64_t same = __ballot(h == hg) & ins;  // ins holds only lanes probed before g
            if (same) {
              const uint32_t k = 63u - (uint32_t)__clzll((long long)same);
              cand = __builtin_amdgcn_readlane(pos, k);
              hit = __builtin_amdgcn_readlane(a0, k) == __builtin_amdgcn_readlane(a0, g);
              flen = 4;
              if (hit) {  // both lanes hold their 16 bytes: the common prefix without a memory trip
                const uint32_t y1 = (uint32_t)__builtin_amdgcn_readlane(a1, k) ^ (uint32_t)__builtin_amdgcn_readlane(a1, g);
                if (y1) {
                  flen = 4u + (((uint32_t)__ffs((int)y1) - 1u) >> 3);
                } else {
                  const uint32_t y2 = (uint32_t)__builtin_amdgcn_readlane(a2, k) ^ (uint32_t)__builtin_amdgcn_readlane(a2, g);
                  if (y2) {
                    flen = 8u + (((uint32_t)__ffs((int)y2) - 1u) >> 3);
                  } else {
                    const uint32_t y3 = (uint32_t)__builtin_amdgcn_readlane(a3, k) ^ (uint32_t)__builtin_amdgcn_readlane(a3, g);
                    flen = y3 ? 12u + (((uint32_t)__ffs((int)y3) - 1u) >> 3) : 16u;
                  }
                }
              }
            } else {
              cand = __builtin_amdgcn_readlane(old, g);
              hit = (H >> g) & 1ull;
              flen = __builtin_amdgcn_readlane(eqlen, g);
            }
          } else {
            cand = __builtin_amdgcn_readlane(old, g);
            hit = (H >> g) & 1ull;
            flen = __builtin_amdgcn_readlane(eqlen, g);
          }
          ins |= 1ull << g;  // table[h] = ip precedes the comparison (snappy.nim:97-98,128)
          if (!hit) {
            if (reprobe) {  // snappy.nim:130-134: a fresh run starts behind the re-probe
              reprobe = false;
              K = 0;
            } else {
              K++;
            }
            i = g + 1u;
            if (dense && K == 32u) {
              ip = W0 + i;
              post = false;
              break;
            }
            if (i >= 64u) {
              post = false;
              ip = dense ? W0 + 64u
                         : __builtin_amdgcn_readlane(pos, 63) + __builtin_amdgcn_readlane(step, 63);
              break;
            }
            continue;
          }
          // ---- match at lane g (snappy.nim:103-114) ----
          const uint32_t mp = __builtin_amdgcn_readlane(pos, g);
          const uint32_t limit = n < mp + 258u ? n : mp + 258u;
          uint32_t matched;
          if (flen < 16u) {
            KPROF_COUNT(15, 1);
            matched = flen;
          } else {
            // 4 + determineMatchLength(cand + 4, mp + 4, limit), internal.nim:251-270:
            // lane l compares bytes flen+4l .. flen+3+4l against the candidate
            KPROF_COUNT(10, 1);
            zh_wave_sync();
            const uint32_t o = flen + 4u * lane;
            uint32_t avail = 0;
            if (mp + o < limit) avail = limit - (mp + o) < 4u ? limit - (mp + o) : 4u;
            const uint32_t x = ld32(mp + o) ^ ld32(cand + o);
            uint32_t eq = min(((uint32_t)__ffs((int)x) - 1u) >> 3, 4u);
            if (eq > avail) eq = avail;
            const uint64_t stop = __ballot(eq < 4u);  // some lane always stops (limit <= mp + 258)
            const uint32_t fl = (uint32_t)__ffsll((long long)stop) - 1u;
            matched = flen + 4u * fl + __builtin_amdgcn_readlane(eq, fl);
          }
          if (matched > limit - mp) matched = limit - mp;
          if (lane == 0) {
            zh_store_nt(m_pos + nm, (uint16_t)mp);
            zh_store_nt(m_len + nm, (uint16_t)matched);
            zh_store_nt(m_off + nm, (uint16_t)(mp - cand));
          }
          nm++;
          ip = mp + matched;
          K = 0;
          post = true;
          if (ip >= ip_limit) {  // snappy.nim:118-120
            finished = true;
            break;
          }
          const uint32_t ni = g + matched;  // dense: lane of the new ip
          if (!dense || ni >= 64u) break;   // the step that follows inserts ip-1 as its lane 0
          ins |= 1ull << (ni - 1u);         // table[hash(ip-1)] = ip-1 (snappy.nim:126)
          i = ni;
          reprobe = true;
        }
        KPROF_MARK(3);
#ifndef ZH_EMU
        __builtin_amdgcn_s_setprio(0);
#endif
        if (finished) break;  // nothing reads the table any more
        // ---- table inserts of the probes that really happened, in probe order ----
        const bool mine = (ins >> lane) & 1ull;
        if (mine) atomicOr(&s_bits[h >> 5], 1u << (h & 31u));
        if (mine && !((C >> lane) & 1ull)) s_table[h] = (uint16_t)(pos | (tag << 15));
        uint64_t cc = ins & C;
        while (cc) {  // probes that share a slot write one by one (the later one wins)
          const uint32_t jx = (uint32_t)__ffsll((long long)cc) - 1u;
          cc &= cc - 1;
          if (lane == jx) s_table[h] = (uint16_t)(pos | (tag << 15));
        }
        zh_wave_sync();
        KPROF_MARK(4);
      }
    }
    if (lane == 0) s_nmatch = nm;
  }
  __threadfence_block();
  zh_wave_sync();
  KPROF_MARK(3);
  const uint32_t nmatch = s_nmatch;

  // ---- match histograms, and the coverage bitmap as the places where "inside a match, behind its first byte" flips:
  // the match's second byte and the byte behind its last (a match is four bytes at least: no two flips share a bit).
  // A match's first byte is then the clear bit in front of a set one -- which is how the emission reads the same bitmap
  // (zh_emit_kernel: it gets the bitmap, a.f_cover, instead of filing the match list a second time).
  // The next 64 matches' fields are asked for before these are filed (clamped, unconditional loads) ----
  uint32_t extra_bits = 0, covered = 0;
  {
    const uint32_t mlast = nmatch ? nmatch - 1u : 0u;
    auto at = [&](uint32_t m) { return m < nmatch ? m : mlast; };
    uint32_t pq = m_pos[at(lane)], lq = m_len[at(lane)], oq = m_off[at(lane)];
    for (uint32_t m = lane; m - lane < nmatch; m += 64) {
      const uint32_t p = pq, l = lq, o = oq;
      pq = m_pos[at(m + 64u)];
      lq = m_len[at(m + 64u)];
      oq = m_off[at(m + 64u)];
      if (m < nmatch) {
        const uint32_t li = zh_len_code(l), di = zh_dist_code(o);
        atomicAdd(&s_hist[257 + li], 1u);
        atomicAdd(&s_hist[ZH_NUM_LITLEN + di], 1u);
        extra_bits += zh_len_extra_bits(li) + zh_dist_extra_bits(di);
        covered += l;
        atomicXor(&s_cover[(p + 1u) >> 5], 1u << ((p + 1u) & 31u));
        if (p + l < kCntWords * 32u) atomicXor(&s_cover[(p + l) >> 5], 1u << ((p + l) & 31u));
      }
    }
  }
  zh_wave_sync();
  // flips -> "inside": bit i = parity of the flips at or before i; 64 words a turn, a word a lane
  {
    uint32_t carry = 0;  // parity of all flips before this turn's words
    for (uint32_t w0 = 0; w0 < kCntWords && w0 * 32u < n; w0 += 64) {
      uint32_t x = s_cover[w0 + lane];
#pragma unroll
      for (uint32_t sh = 1; sh < 32; sh <<= 1) x ^= x << sh;
      const uint64_t odd = __ballot((x >> 31) != 0u);
      if ((carry + (uint32_t)__popcll(odd & zh_lanemask_lt())) & 1u) x = ~x;
      s_cover[w0 + lane] = x;
      carry += (uint32_t)__popcll(odd);
    }
  }
  zh_wave_sync();
  // the bitmap for the emission: whole groups of 64 words (2048 positions: its chunk), zero behind the fragment
  if (a.f_cover) {
    uint32_t* cov_out = a.f_cover + (size_t)f * kCntWords;
    for (uint32_t w0 = 0; w0 * 32u < n; w0 += 64) cov_out[w0 + lane] = s_cover[w0 + lane];
  }
  // ---- literal histogram (lanes over positions): four positions a lane and pass, four passes' source words in
  // flight together.  A literal is a byte that is neither behind a match's first byte nor one itself: its bit and
  // the next are clear ----
  for (uint32_t base = 0; base < n; base += 1024) {
    uint32_t wq[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
      const uint32_t p = base + 256u * u + 4u * lane;
      wq[u] = p < n ? ld32(p) : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
      const uint32_t p = base + 256u * u + 4u * lane;
      if (p < n) {
        // p is a multiple of 4: one word for all four, and the next word's first bit behind position 31
        const uint32_t wi = p >> 5;
        const uint32_t c_lo = s_cover[wi], c_hi = wi + 1u < kCntWords ? s_cover[wi + 1u] : 0u;
        const uint32_t cov = (uint32_t)((((uint64_t)c_hi << 32) | c_lo) >> (p & 31u));
        const uint32_t notlit = cov | (cov >> 1);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
          if (p + k < n && !((notlit >> k) & 1u)) atomicAdd(&s_hist[(wq[u] >> (8u * k)) & 255u], 1u);
      }
    }
  }
  extra_bits = zh_wave_sum(extra_bits);
  covered = zh_wave_sum(covered);
  zh_wave_sync();

  uint16_t* hist_out = a.f_hist + (size_t)f * ZH_HIST_STRIDE;
  for (uint32_t i = lane; i < ZH_HIST_STRIDE; i += 64) hist_out[i] = (uint16_t)s_hist[i];
  if (lane == 0) {
    a.f_nmatch[f] = nmatch;
    a.f_spill[f] = 0;
    a.f_nlit[f] = n - covered;
    a.f_extra_bits[f] = extra_bits;
  }
  KPROF_MARK(5);
  KPROF_COUNT(11, 1);
  KPROF_FLUSH(0, 16);
  zh_wave_sync();
  {
    uint32_t nf = 0;
    if (lane == 0) {
      if (cost) {
        const uint64_t dt = zh_clock() - t_start;
        cost[f] = dt < 0xffffffffull ? (uint32_t)dt : 0xffffffffu;
      }
      nf = atomicAdd(next_frag, 1u);
    }
    ticket = zh_bcast(nf);
  }
  }  // next fragment of this wave
}

// ---- the cheap ones last: from a plan's second run on the fragments are handed out as numbered -- the mix of kinds
// the batch came with keeps the fabric evenly loaded: all of them sorted by cost, dearest first, measured 2.7 %
// SLOWER on 4096 x 1 MiB (the dear ones are the ones with the most table traffic, and they would all run together)
// -- EXCEPT that the cheapest ~ 2 rounds' worth go to the end, so that what runs last, with waves already idle,
// is short.  Costs: the cycles a fragment took the run before, in 64 classes (four an octave); the cheapest classes
// that together hold at most `tail` fragments form the tail; both parts keep their order (a stable partition: flags,
// a two-level scan, a scatter -- four small launches in front of the matcher). ----
namespace {
constexpr uint32_t kCostClasses = 64;
__device__ __forceinline__ uint32_t cost_class(uint32_t c) {  // 2^10 .. 2^26 cycles, a quarter octave a class
  if (c < 1024u) return 0u;
  const uint32_t lg = 31u - (uint32_t)__clz((int)c);
  const uint32_t k = (lg - 10u) * 4u + ((c >> (lg - 2u)) & 3u);
  return k < kCostClasses ? k : kCostClasses - 1u;
}
}  // namespace
// hist[0 .. 64): fragments a class
__global__ __launch_bounds__(256) void zh_l1_cost_hist_kernel(const uint32_t* __restrict__ cost, uint32_t n,
                                                              uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_h[kCostClasses];
  if (threadIdx.x < kCostClasses) s_h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) atomicAdd(&s_h[cost_class(cost[i])], 1u);
  __syncthreads();
  if (threadIdx.x < kCostClasses && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}
// few rounds: ALL fragments by class, the dearest class first (inside a class as they come): hist[k] -> hist[64 + k], the
// first place of class k; then the scatter
__global__ __launch_bounds__(64) void zh_l1_cost_places_kernel(uint32_t* __restrict__ hist) {
  const unsigned lane = zh_lane();
  const uint32_t mine = hist[kCostClasses - 1u - lane];  // (lane 0: the dearest class)
  const uint32_t incl = zh_wave_scan(mine);
  hist[kCostClasses + (kCostClasses - 1u - lane)] = incl - mine;
}
__global__ __launch_bounds__(256) void zh_l1_cost_sort_kernel(const uint32_t* __restrict__ cost, uint32_t n,
                                                              uint32_t* __restrict__ hist, uint32_t* __restrict__ order) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) order[atomicAdd(&hist[kCostClasses + cost_class(cost[i])], 1u)] = i;
}
// -> hist[64]: the highest class of the tail (classes 0 .. hist[64] hold at most `tail` fragments; 0xffffffff: none),
// hist[65]: how many fragments that is.  Then the tail flags a block of 256 fragments, counted: blk[b].
__global__ __launch_bounds__(64) void zh_l1_cost_cut_kernel(uint32_t* __restrict__ hist, uint32_t tail) {
  const unsigned lane = zh_lane();
  const uint32_t incl = zh_wave_scan(hist[lane]);
  const uint64_t fits = __ballot(incl <= tail);  // (a prefix of the lanes: the sums grow)
  const uint32_t k = (uint32_t)__popcll(fits);
  if (lane == 0) hist[kCostClasses] = k ? k - 1u : 0xffffffffu;
  if (lane == (k ? k - 1u : 0u)) hist[kCostClasses + 1u] = k ? incl : 0u;
}
__global__ __launch_bounds__(256) void zh_l1_cost_count_kernel(const uint32_t* __restrict__ cost, uint32_t n,
                                                               const uint32_t* __restrict__ hist,
                                                               uint32_t* __restrict__ blk) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x, cut = hist[kCostClasses];
  const bool last = i < n && cut != 0xffffffffu && cost_class(cost[i]) <= cut;
  const uint32_t c = (uint32_t)__popcll(__ballot(last));
  __shared__ uint32_t s_c[4];
  if (zh_lane() == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
// blk[b] -> the tail fragments before block b (one workgroup, 1024 blocks a turn)
__global__ __launch_bounds__(1024) void zh_l1_cost_scan_kernel(uint32_t* __restrict__ blk, uint32_t nblk) {
  __shared__ uint32_t s_w[16], s_carry;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblk; base += 1024u) {
    const uint32_t i = base + tid, v = i < nblk ? blk[i] : 0u;
    const uint32_t incl = zh_wave_scan(v);
    if (zh_lane() == 63) s_w[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = s_carry;
    for (uint32_t w = 0; w < (tid >> 6); w++) before += s_w[w];
    if (i < nblk) blk[i] = before + incl - v;
    __syncthreads();
    if (tid == 1023) s_carry = before + incl;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void zh_l1_cost_scatter_kernel(const uint32_t* __restrict__ cost, uint32_t n,
                                                                 const uint32_t* __restrict__ hist,
                                                                 const uint32_t* __restrict__ blk,
                                                                 uint32_t* __restrict__ order) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x, cut = hist[kCostClasses], ntail = hist[kCostClasses + 1u];
  const bool last = i < n && cut != 0xffffffffu && cost_class(cost[i]) <= cut;
  const uint64_t m = __ballot(last);
  __shared__ uint32_t s_c[4];
  if (zh_lane() == 0) s_c[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t before = blk[blockIdx.x] + (uint32_t)__popcll(m & zh_lanemask_lt());  // tail fragments before this one
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += s_c[w];
  if (i < n) order[last ? (n - ntail) + before : i - before] = i;
}

// waves that share the table pool: 19 per CU on 256 CUs, LDS 7.4 KiB each (ZH_L1_SLOTS: tuning override).  The kernel's
// throughput is the fabric's, not its waves': 4096 .. 5376 of them are within 2 %; three rounds on one box,
// ms for 4096 x 1 MiB: 4608 waves 63.7, 4864 63.5, 5120 (rounds 2-5) 64.4; one GPU's share (16 384 fragments) the same
extern "C" uint32_t zh_l1_table_slots(void) {
  static const uint32_t slots = [] {
    const char* e = getenv("ZH_L1_SLOTS");
    const long v = e ? atol(e) : 0;
    return v >= 64 && v <= 16384 ? (uint32_t)v : 4864u;
  }();
  return slots;
}

__global__ void zh_l1_set_counter_kernel(uint32_t* next_frag, uint32_t v) { *next_frag = v; }

// `cost` / `order` / `hist` (each may be null): this run's cycles a fragment; the order made of LAST run's (built here
// when `use_order`: the plan has run before); 128 words + a word a 256 fragments of scratch for that
extern "C" void zh_launch_l1_match(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a,
                                   int huffman_only, uint16_t* table_pool, uint32_t* next_frag, uint32_t* cost,
                                   uint32_t* order, uint32_t* hist, int use_order) {
  if (!a.nfrags) return;
#ifdef ZH_XCHECK  // the test / measurement build only: the same kernel with its table in LDS (ZH_L1_TABLE=lds)
  static const bool lds = [] {
    const char* e = getenv("ZH_L1_TABLE");
    return e && strcmp(e, "lds") == 0;
  }();
  if (lds) {
    const uint32_t grid = a.nfrags < 1024u ? a.nfrags : 1024u;  // (39.4 KiB of LDS: four waves per CU)
    hipLaunchKernelGGL(zh_l1_set_counter_kernel, dim3(1), dim3(1), 0, stream, next_frag, grid);
    hipLaunchKernelGGL(zh_l1_match_kernel<true>, dim3(grid), dim3(64), 0, stream, d_src, a, huffman_only,
                       table_pool, next_frag, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    return;
  }
#endif
  const uint32_t slots = zh_l1_table_slots();
  const uint32_t grid = a.nfrags < slots ? a.nfrags : slots;
  const bool sorted = use_order && cost && order && hist && a.nfrags > grid;  // (one round of the machine: nothing to order)
  if (sorted) {
    const uint32_t g = (a.nfrags + 255u) / 256u;
    static const uint32_t rounds = [] {  // (ZH_L1_TAIL_ROUNDS: measurement)
      const char* e = getenv("ZH_L1_TAIL_ROUNDS");
      const long v = e ? atol(e) : 0;
      return v >= 1 && v <= 64 ? (uint32_t)v : 2u;
    }();
    const uint32_t tail = a.nfrags / 2u < rounds * grid ? a.nfrags / 2u : rounds * grid;  // two rounds' worth, half of all at most
    uint32_t* blk = hist + 128;  // (hist: 128 words + a word a block of 256 fragments)
    (void)hipMemsetAsync(hist, 0, 128u * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(zh_l1_cost_hist_kernel, dim3(g), dim3(256), 0, stream, cost, a.nfrags, hist);
    if (a.nfrags < 8u * grid) {
      // a few rounds of the machine (one GPU's share of eight: 3.2): the dearest first -- what starts last then is
      // short (512 x 1 MiB: 9.41 -> 8.97 ms, against 9.33 with only the cheapest moved)
      hipLaunchKernelGGL(zh_l1_cost_places_kernel, dim3(1), dim3(64), 0, stream, hist);
      hipLaunchKernelGGL(zh_l1_cost_sort_kernel, dim3(g), dim3(256), 0, stream, cost, a.nfrags, hist, order);
    } else {
      // many rounds (4096 x 1 MiB: 25.6): the order the batch came in, the cheapest two rounds' worth last (65.57 ->
      // 65.22 ms; all of them sorted: 66.4, see above)
      hipLaunchKernelGGL(zh_l1_cost_cut_kernel, dim3(1), dim3(64), 0, stream, hist, tail);
      hipLaunchKernelGGL(zh_l1_cost_count_kernel, dim3(g), dim3(256), 0, stream, cost, a.nfrags, hist, blk);
      hipLaunchKernelGGL(zh_l1_cost_scan_kernel, dim3(1), dim3(1024), 0, stream, blk, g);
      hipLaunchKernelGGL(zh_l1_cost_scatter_kernel, dim3(g), dim3(256), 0, stream, cost, a.nfrags, hist, blk, order);
    }
  }
  hipLaunchKernelGGL(zh_l1_set_counter_kernel, dim3(1), dim3(1), 0, stream, next_frag, grid);
  hipLaunchKernelGGL(zh_l1_match_kernel<false>, dim3(grid), dim3(64), 0, stream, d_src, a, huffman_only,
                     table_pool, next_frag, sorted ? order : (const uint32_t*)nullptr, cost);
}
