// BestSpeed matcher, PARALLEL parse (opt-in: zh_set_l1_parse(ctx, 1) / ZH_L1_PARSE=parallel).
//
// NOT the reference's parse.  zh_l1_match_kernel replays snappy.nim:12-136 decision for
// decision, which ties a fragment to one serial walk; this kernel produces a
// *different*, equally valid token stream under the contract BASELINE.json's north star
// states for the encoder: an RFC 1951 stream that zippy's uncompress() decodes to the input
// bit for bit, with a compressed size within a stated margin of zippy's at the same level
// (tests: every buffer through oracle.uncompress and zlib, aggregate size <= 1.02 x oracle).
// Everything behind the match list -- statistics, Huffman codes, layout, emission -- is the
// exact path's.  Same fragment rule as the reference (snappy.nim:150-163): matches never
// leave their 32 KiB fragment, so fragments stay independent.
//
// One workgroup (1024 threads) per fragment, two workgroups a CU:
//   P1  links    one wave walks the fragment in order, 64 positions a step, through a
//                16384-entry hash table in LDS (same hash as snappy.nim:70-71), one returning
//                atomicMax a position: what comes back is the latest earlier position with the
//                slot's hash -- EVERY position is inserted, not only the ones a serial parse
//                visits -- and goes to the workgroup's slot of scratch as it is;
//   P2+P3 parse  greedy left to right (take the match at p if there is one, else a literal), the
//                fragment's bytes now in LDS where the table was.  A thread per 32 positions walks from
//                a guessed entry and works out the match length of what it VISITS (the candidate from
//                the table's answer -- or the position before, in a run --, the common prefix, 0 or
//                4..258: internal.nim:251-270 determineMatchLength, limit snappy.nim:110); exits are
//                handed on and threads whose entry changed walk again; thread 0's entry is exact, so
//                at the fixed point every entry is the serial walk's by induction.  (Round 3 worked out
//                the length of every position first, a thread a position: 47 % of the kernel for values of
//                which the parse reads a third.  Same lengths, same parse, same bytes.)
//   P4  output   match list (same SoA records the exact matcher writes), litlen / distance
//                histograms, literal count, extra-bit sum.
// Algorithmic bytes: N read.  The table's answers (128 KiB a fragment) go through a per-workgroup slot of
// HBM scratch (persistent workgroups).
#include <cstdlib>
#include <cstring>

#include "zh_common.h"
#include "zh_kprof.h"
#include "zh_tables.h"

namespace {
constexpr uint32_t kHashMul = 0x1e35a7bdu;  // snappy.nim:70-71
constexpr uint32_t kT = 1024;               // threads per workgroup
constexpr uint32_t kChunk = 32;             // positions per parse thread (kT * kChunk = 32768)
constexpr uint32_t kSrcWords = 8192 + 72;   // fragment + slack for compares that run past its end
constexpr uint32_t kShift = 18;             // 14 hash bits
// 16 bytes at any 4-byte address (gfx950 global loads need no alignment)
struct __attribute__((packed, aligned(4))) Bytes16 {
  uint32_t x, y, z, w;
};
// 4 bytes at any byte address
struct __attribute__((packed)) Word32 {
  uint32_t v;
  __device__ operator uint32_t() const { return v; }
};
}  // namespace

__global__ __launch_bounds__(kT, 8) void zh_l1p_match_kernel(const uint8_t* __restrict__ d_src, ZhCompressArgs a,
                                                          uint16_t* __restrict__ link_pool,
                                                          uint32_t* __restrict__ next_frag) {
  // P1: the hash table, a dword a slot (atomicMax); afterwards the fragment's bytes (s_src) and the
  // match lengths, a byte a position (s_mlen)
  __shared__ uint32_t s_big[kSrcWords + 8192];
  __shared__ uint32_t s_hist[ZH_HIST_STRIDE];
  __shared__ uint32_t s_exit[kT];
  __shared__ uint32_t s_ring[kT];    // P1: a block of wave 0's results on its way out (the other block: s_exit)
  __shared__ uint32_t s_wsum[kT / 64][3];
  __shared__ uint32_t s_misc[4];     // 0: next fragment, 1: "some entry changed"
  uint32_t* const s_tab = s_big;
  uint32_t* const s_src = s_big;
  uint8_t* const s_mlen = reinterpret_cast<uint8_t*>(s_big + kSrcWords);

  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  // a workgroup's slot of the pool: what the table returned for every position (a dword each, P1 -> P2 / P4)
  uint32_t* const raws = reinterpret_cast<uint32_t*>(link_pool + (size_t)blockIdx.x * (2u * ZH_FRAG_SIZE));

  for (uint32_t f = blockIdx.x; f < a.nfrags;) {
    KPROF_DECL(8);  // cycles of thread 0: 0 stage-in, 1 links, 2 first walks (lengths on demand), 3 turns, 4 output; counts: 5 turns, 6 fragments
    const ZhFragDesc fd = a.frags[f];
    const uint32_t n = fd.len;
    const uint8_t* src = d_src + fd.src_off;
    const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t* asrc = reinterpret_cast<const uint32_t*>(src - mis);
    const uint32_t ndw = (n + mis + 3u) >> 2;
    for (uint32_t i = t; i < 4096; i += kT) reinterpret_cast<uint4*>(s_tab)[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = t; i < ZH_HIST_STRIDE; i += kT) s_hist[i] = 0;
    __syncthreads();
    KPROF_MARK(0);
    // ---- P1: candidate links (wave 0; the other waves carry its results out).  A step is 64
    // consecutive positions and ONE returning atomicMax a lane on the table: what comes back is the
    // latest position with the lane's hash that entered before it -- in an earlier step, or (the LDS
    // unit serves a wave's lanes in ascending order) in a lower lane of this one.  Anything else a
    // lane might get back is thrown away below (a candidate must lie before its position), so the
    // stream is valid whatever the order; the four bytes at a position come straight from the
    // stream (coalesced, any byte address), fetched 16 steps ahead ----
    // (Round 6: the wave's results used to go out to the pool by its own stores -- and with loads AND stores in flight the
    // compiler can only wait for all of them (one counter, two kinds of operation: `s_waitcnt vmcnt(0)` at the head of every
    // block of steps), so the loads fetched "ahead" were waited for as soon as they were asked, and the stores with them.
    // Now wave 0 leaves a block's results in LDS and the waves that wait for it anyway carry them out: wave 0 has loads only,
    // waited for one by one: 23.65 -> 23.05 ms for 4096 x 1 MiB, the same streams -- the rest of a step's 150 cycles is the LDS
    // pipe the CU's other workgroup's walks share, as round 4 found.  profiles/r06_al_*)
    if (n >= 16u) {
      const uint32_t steps = (n + 63u) >> 6;
      constexpr uint32_t kAhead = 16;  // source words are fetched this many steps ahead; steps a block
      static_assert(kAhead * 64u == kT, "a block of results is the size of s_exit");
      auto fetch = [&](uint32_t step) -> uint32_t {
        const uint32_t p = step * 64u + lane;
        return (uint32_t) * reinterpret_cast<const Word32*>(src + (p + 4u <= n ? p : n - 4u));
      };
      uint32_t wq[kAhead];
      if (wave == 0) {
#ifndef ZH_EMU
        __builtin_amdgcn_s_setprio(3);  // the one serial phase: first pick of the SIMD's issue slots
#endif
#pragma unroll
        for (uint32_t k = 0; k < kAhead; k++) wq[k] = fetch(k);
      }
      for (uint32_t s0 = 0; s0 < steps; s0 += kAhead) {
        uint32_t* const ring = ((s0 / kAhead) & 1u) ? s_exit : s_ring;  // (two blocks: one being carried out, one being made)
        if (wave == 0) {
          // No lane is masked: position 0's atomicMax changes nothing (and a candidate must lie before its position: it
          // gets none), positions behind n - 4 hash bytes of the last word and are never looked at by P2 (no match starts
          // in the last 15 bytes), positions behind n enter the table when nobody reads it any more.  Only the loads are
          // kept inside the fragment.
          // (the atomics of half a block are issued back to back, their results left in the ring afterwards)
#pragma unroll
          for (uint32_t half = 0; half < 2; half++) {
            uint32_t raw[kAhead / 2];
#pragma unroll
            for (uint32_t k = 0; k < kAhead / 2; k++) {
              const uint32_t j = half * (kAhead / 2) + k, step = s0 + j;
              const uint32_t p = step * 64u + lane;
              // a table entry is position << 16 | the 16 bits of the hash product below the slot's 14:
              // the maximum is still the latest position, and a candidate whose 30 bits agree has the
              // position's four bytes but for one case in 65 536 -- which P2's compare settles
              const uint32_t h = wq[j] * kHashMul;
              wq[j] = fetch(step + kAhead);
              raw[k] = atomicMax(&s_tab[h >> kShift], (p << 16) | ((h >> 2) & 0xffffu));
#ifdef ZH_EMU
              zh_wave_sync();  // (the emulator runs a lane at a time: keep the steps in step)
#endif
            }
#pragma unroll
            for (uint32_t k = 0; k < kAhead / 2; k++) ring[(half * (kAhead / 2) + k) * 64u + lane] = raw[k];
          }
        }
        __syncthreads();
        if (wave != 0)
          for (uint32_t i = t - 64u; i < kAhead * 64u; i += kT - 64u)
            raws[(s0 * 64u + i) & (ZH_FRAG_SIZE - 1u)] = ring[i];
      }
#ifndef ZH_EMU
      if (wave == 0) __builtin_amdgcn_s_setprio(0);
#endif
    }
    __syncthreads();
    // ---- the fragment's bytes into LDS, over the table (aligned dwords of the stream below `src`,
    // never past the dword that holds its last byte); LDS byte q = fragment byte q - mis.  All of
    // a thread's loads are in flight together: 16 bytes each, at any 4-byte address ----
    {
      Bytes16 v[8192 / 4 / kT];
#pragma unroll
      for (uint32_t k = 0; k < 8192 / 4 / kT; k++) {
        const uint32_t i = (k * kT + t) * 4u;
        v[k] = Bytes16{0, 0, 0, 0};
        if (i + 4u <= ndw) {
          v[k] = *reinterpret_cast<const Bytes16*>(asrc + i);
        } else if (i < ndw) {
          v[k].x = asrc[i];
          if (i + 1u < ndw) v[k].y = asrc[i + 1];
          if (i + 2u < ndw) v[k].z = asrc[i + 2];
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < 8192 / 4 / kT; k++)
        *reinterpret_cast<uint4*>(s_src + (k * kT + t) * 4u) = make_uint4(v[k].x, v[k].y, v[k].z, v[k].w);
    }
    for (uint32_t i = 8192 + t; i < kSrcWords; i += kT) s_src[i] = 0;
    __syncthreads();
    // (unaligned ds_read_b32 / b64 work on gfx950 but cost this kernel a factor of two: measured)
    auto ld32 = [&](uint32_t p) -> uint32_t { return zh_ld32(s_src, p + mis); };
    auto ld64 = [&](uint32_t p) -> uint64_t { return zh_ld64(s_src, p + mis); };

    KPROF_MARK(1);
    // ---- P2 + P3: the greedy parse (take the match at p if there is one, else a literal), and match lengths
    // only where it comes by.  The parse visits a third of the positions; working out the length of EVERY
    // position first (round 3: a thread a position, 32 turns) was where the kernel spent its instructions.
    // A thread walks its 32-position chunk from a guessed entry -- the chunk's first byte -- and works out what
    // it visits: the candidate from the table's answer (or the position before, in a run), the common prefix,
    // 0 or 4..258 (internal.nim:251-270 determineMatchLength, limit snappy.nim:110); results stay in s_mlen and
    // a bit a position in a register says which are there (a thread only ever works inside its own chunk).
    // Exits are handed on, threads whose entry changed walk again -- what they meet a second time they look up
    // --; thread 0's entry is exact, so at the fixed point every entry is the serial walk's by induction. ----
    const uint32_t lo = t * kChunk, hi = lo + kChunk < n ? lo + kChunk : n;
    // the candidate of position p (0: none): the latest earlier position with the hash -- its 30 hash bits must be
    // the position's -- or, in a run, the position right before: the nearest candidate there is
    auto candidate = [&](uint32_t p, uint32_t raw) -> uint32_t {
      const uint64_t w5 = zh_ld64(s_src, p + mis - (p ? 1u : 0u));  // the byte before p and p's four
      const uint32_t h = (uint32_t)(p ? w5 >> 8 : w5) * kHashMul;
      const uint32_t h_prev = p ? (uint32_t)w5 * kHashMul : ~h;
      uint32_t c = raw >> 16;
      bool same = c < p && ((raw ^ (h >> 2)) & 0xffffu) == 0u;
      if ((h >> 2) == (h_prev >> 2) && p >= 2u) {
        c = p - 1u;
        same = true;
      }
      return same && c < p ? c : 0u;
    };
    uint32_t known = 0;  // bit i: the length of position lo + i is in s_mlen
    // the table's answers come four positions a load: two tokens in three are literals, and the position behind
    // a literal then has its answer in a register already instead of a trip to L2 away
    uint32_t wbase = 0x80000000u;
    Bytes16 wr = {0, 0, 0, 0};
    // The walk as ONE loop (a loop per compare inside a loop per position makes a wave wait for its longest
    // match at every position): a turn takes a lane that stands at a new position through its candidate -- or the
    // length it already knows -- and a lane that is comparing through 16 more bytes; a lane whose match runs on
    // compares on while its neighbours walk.  Match length: 0 or 4..258 (internal.nim:251-270, limit
    // snappy.nim:110); no match starts in the last 15 bytes (the reference's ip_limit).
    auto walk = [&](uint32_t p, uint32_t end) -> uint32_t {
      uint32_t c = 0, m = 0, lim = 0;  // comparing: candidate, bytes equal so far, limit (c == 0: standing at p)
      while (p < end) {
        uint32_t m8 = 0;
        bool done = false;  // the length at p is known this turn
        if (c == 0u) {
          const uint32_t bit = 1u << (p - lo);
          if (known & bit) {
            m8 = s_mlen[p];
            done = true;
          } else {
            known |= bit;
            uint32_t d = p - wbase;
            if (d >= 4u) {
              wr = *reinterpret_cast<const Bytes16*>(raws + p);  // (past the fragment's last position: the pool's next bytes, unused)
              wbase = p;
              d = 0;
            }
            if (p + 16u <= n) c = candidate(p, d == 0u ? wr.x : d == 1u ? wr.y : d == 2u ? wr.z : wr.w);
            if (c == 0u) {
              s_mlen[p] = 0;
              done = true;
            } else {
              m = 0;
              lim = n - p < 258u ? n - p : 258u;  // (>= 16 here)
            }
          }
        }
        if (c != 0u) {  // 16 bytes: both streams as aligned dwords shifted into place
          const uint32_t ab = p + m + mis, bb = c + m + mis, as = ab & 3u, bs = bb & 3u;
          const uint32_t* ap = s_src + (ab >> 2);
          const uint32_t* bp = s_src + (bb >> 2);
          uint32_t x[4];
#pragma unroll
          for (uint32_t i = 0; i < 4; i++)
            x[i] = __builtin_amdgcn_alignbyte(ap[i + 1], ap[i], as) ^ __builtin_amdgcn_alignbyte(bp[i + 1], bp[i], bs);
          uint32_t d = 16;
#pragma unroll
          for (uint32_t i = 4; i-- > 0;)
            if (x[i]) d = 4u * i + (((uint32_t)__ffs((int)x[i]) - 1u) >> 3);
          m += d;
          if (d < 16u || m >= lim) {
            if (m > lim) m = lim;
            m8 = m < 4u ? 0u : m - 3u;
            s_mlen[p] = (uint8_t)m8;
            c = 0;
            done = true;
          }
        }
        if (done) p += m8 ? m8 + 3u : 1u;
      }
      return p;
    };
    uint32_t entry = lo, ex = 0;
    if (lo < n) ex = walk(entry, hi);
    if (lane == 63) s_exit[t] = ex;
    __syncthreads();
    KPROF_MARK(2);
    // Exits are handed on inside a wave first (lane to lane, no barrier: a wave's 64 chunks settle among
    // themselves), then from wave to wave through s_exit; a round of the outer loop is one such hand-over.
    for (;;) {
      uint32_t from_before = entry;  // what the chunk before hands this one: lane 0 gets it from the wave before
      if (lane == 0 && t && lo < n) from_before = s_exit[t - 1];
      bool any = false;
      for (;;) {
        uint32_t want = (uint32_t)__shfl_up((int)ex, 1, 64);
        if (lane == 0) want = from_before;
        if (!(t && lo < n)) want = entry;
        const bool changed = want != entry;
        if (changed) {
          entry = want;
          ex = walk(entry, hi);
        }
        KPROF_COUNT(5, 1);
        if (!__ballot(changed)) break;
        any = true;
      }
      __syncthreads();  // (everybody has read the exits of the round before)
      if (t == 0) s_misc[1] = 0;
      if (lane == 63) s_exit[t] = ex;
      __syncthreads();
      // a wave whose first lane would now be handed something else goes on
      const bool more = lane == 0 && t && lo < n && s_exit[t - 1] != entry;
      if (more) s_misc[1] = 1;
      __syncthreads();
      if (!s_misc[1]) break;
      (void)any;
    }
    KPROF_MARK(3);

    // ---- P4: records and statistics ----
    uint32_t nm = 0, nl = 0;
    if (lo < n) {
      uint32_t p = entry;
      while (p < hi) {
        const uint32_t m = s_mlen[p];
        if (m) {
          nm++;
          p += m + 3u;
        } else {
          nl++;
          p++;
        }
      }
    }
    // exclusive prefix of the match counts over the workgroup
    const uint32_t incl = zh_wave_scan(nm);
    if (lane == 63) s_wsum[wave][0] = incl;
    __syncthreads();
    uint32_t mbase = incl - nm, total_m = 0;
    for (uint32_t w = 0; w < kT / 64; w++) {
      const uint32_t c = s_wsum[w][0];
      if (w < wave) mbase += c;
      total_m += c;
    }
    uint16_t* m_pos = a.m_pos + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
    uint16_t* m_len = a.m_len + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
    uint16_t* m_off = a.m_off + (size_t)f * ZH_MAX_MATCHES_PER_FRAG;
    if (lo < n) {
      uint32_t p = entry, k = mbase;
      while (p < hi) {
        const uint32_t m = s_mlen[p];
        if (m) {
          const uint32_t len = m + 3u;
          m_pos[k] = (uint16_t)p;
          m_len[k] = (uint16_t)len;
          k++;
          atomicAdd(&s_hist[257 + zh_len_code(len)], 1u);
          p += len;
        } else {
          atomicAdd(&s_hist[zh_ld8(s_src, p + mis)], 1u);
          p++;
        }
      }
    }
    __syncthreads();  // (the records were written by this workgroup: its CU's L1 has them)
    // distances: a thread per match (its candidate once more: the table's answers are all in flight at once)
    uint32_t extra_bits = 0;
    for (uint32_t k = t; k < total_m; k += kT) {
      const uint32_t p = m_pos[k], len = m_len[k];
      const uint32_t off = p - candidate(p, raws[p]);
      m_off[k] = (uint16_t)off;
      const uint32_t di = zh_dist_code(off);
      atomicAdd(&s_hist[ZH_NUM_LITLEN + di], 1u);
      extra_bits += zh_len_extra_bits(zh_len_code(len)) + zh_dist_extra_bits(di);
    }
    extra_bits = zh_wave_sum(extra_bits);
    nl = zh_wave_sum(nl);
    if (lane == 0) {
      s_wsum[wave][1] = extra_bits;
      s_wsum[wave][2] = nl;
    }
    __syncthreads();
    uint16_t* hist_out = a.f_hist + (size_t)f * ZH_HIST_STRIDE;
    for (uint32_t i = t; i < ZH_HIST_STRIDE; i += kT) hist_out[i] = (uint16_t)s_hist[i];
    if (t == 0) {
      uint32_t eb = 0, lits = 0;
      for (uint32_t w = 0; w < kT / 64; w++) {
        eb += s_wsum[w][1];
        lits += s_wsum[w][2];
      }
      a.f_nmatch[f] = total_m;
      a.f_spill[f] = 0;
      a.f_nlit[f] = lits;
      a.f_extra_bits[f] = eb;
      s_misc[0] = atomicAdd(next_frag, 1u);
    }
    __syncthreads();
    KPROF_MARK(4);
    KPROF_COUNT(6, 1);
    if (t == 0) KPROF_FLUSH(0, 8);
    f = s_misc[0];
    __syncthreads();
  }
}

// workgroups that share the pool: two per CU (71 KiB of LDS each)
extern "C" uint32_t zh_l1p_slots(void) {
  static const uint32_t slots = [] {
    const char* e = getenv("ZH_L1P_SLOTS");
    const long v = e ? atol(e) : 0;
    return v >= 1 && v <= 4096 ? (uint32_t)v : 512u;
  }();
  return slots;
}

__global__ void zh_l1p_set_counter_kernel(uint32_t* next_frag, uint32_t v) { *next_frag = v; }

extern "C" void zh_launch_l1p_match(hipStream_t stream, const uint8_t* d_src, ZhCompressArgs a,
                                    uint16_t* link_pool, uint32_t* next_frag) {
  if (!a.nfrags) return;
  const uint32_t slots = zh_l1p_slots();
  const uint32_t grid = a.nfrags < slots ? a.nfrags : slots;
  hipLaunchKernelGGL(zh_l1p_set_counter_kernel, dim3(1), dim3(1), 0, stream, next_frag, grid);
  hipLaunchKernelGGL(zh_l1p_match_kernel, dim3(grid), dim3(kT), 0, stream, d_src, a, link_pool, next_frag);
}
