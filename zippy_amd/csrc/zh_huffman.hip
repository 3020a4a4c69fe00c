// Per-block Huffman construction and per-buffer output layout.
//
// zh_huffman_kernel (one wave per deflate block) replaces deflate.nim:274-394:
//   * sums the fragment histograms into the block's BlockMetadata
//     (internal.nim:128-131), applies the "almost all literals -> stored" test
//     (deflate.nim:274-277, float32 multiply + truncation) and the fixed-code rule
//     (deflate.nim:279-294);
//   * builds the three length-limited Huffman codes exactly like
//     deflate.nim:13-151 `huffmanCodes`: binary-heap Huffman with the reference's
//     tie-breaking (Nim std/heapqueue == CPython heapq sift rules, compared on
//     frequency only), the histogram rebalancing loop, the unstable quicksort and
//     shortest-first reassignment when a code exceeds the limit, canonical
//     bit-reversed codes.  This part is inherently serial (<= 286 symbols) and
//     runs on lane 0 out of LDS; thousands of blocks run concurrently instead;
//   * run-length encodes the code lengths and assembles the dynamic block
//     header bit string (deflate.nim:296-394);
//   * computes every fragment's encoded bit length as a dot product of its
//     histogram with the code lengths (all 64 lanes).
//
// zh_layout_kernel (one wave per buffer) turns block bit lengths into absolute bit
// positions (the job of BitStreamWriter.pos/bitPos, bitstreams.nim:84-123) and writes the
// container header (zippy.nim:22-42,61-69), the final padding and the trailer
// (zippy.nim:47-58,71-78); zh_block_layout_kernel (one wave per block) follows with the
// block header, the bit position of each fragment, the end-of-block code and, for stored
// blocks, the chunk headers (deflate.nim:179-205).  Everything that is not fragment payload
// comes from these two; all of it is OR-ed into a zeroed output.
#include "zh_common.h"
#include "zh_tables.h"
#include "zh_kprof.h"

namespace {

__constant__ uint8_t c_clcl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

constexpr int kMaxSyms = 288;

// 5.8 KiB, so that the kernel's LDS stays under 10 KiB and sixteen blocks share a CU (4096 blocks: one
// round of the machine): children only for the internal nodes, and the quicksort stack and the depth
// histogram of the length limiting live where the heap was.
struct HuffWork {
  uint64_t hkey[kMaxSyms + 2];     // the heap: node frequency << 32 | node
  uint16_t left[kMaxSyms];         // left and right together: node -> parent (2n - 1 nodes)
  uint16_t right[kMaxSyms];
  uint16_t depth[2 * kMaxSyms];    // node -> depth (while the heap runs: the leaves' frequencies)
  int16_t symbol[kMaxSyms];        // leaf -> symbol
  uint16_t order[kMaxSyms];        // leaves, sorted by depth when limiting
  // behind the heap phase, in hkey's bytes:
  __device__ uint16_t* stack() { return reinterpret_cast<uint16_t*>(hkey); }                         // [2 * kMaxSyms + 4]
  __device__ int32_t* histogram() { return reinterpret_cast<int32_t*>(hkey) + (kMaxSyms + 2); }     // [kMaxSyms + 2] (depths up to n - 1)
};
static_assert((2 * kMaxSyms + 4) * 2 <= (kMaxSyms + 2) * 4 && (kMaxSyms + 2) * 8 <= sizeof(uint64_t) * (kMaxSyms + 2), "");

__device__ inline uint32_t rev16(uint32_t v) { return __brev(v) >> 16; }

// deflate.nim:136-149: canonical codes, bit-reversed, symbols of one length in symbol order (wide counters, see
// SURVEY.md 9.5) -- by the whole wave: a symbol's code is its length's first code plus the symbols of that length
// before it (ballots).  lens / codes in LDS.
__device__ inline void canonical_codes(const uint8_t* lens, uint16_t* codes, int num_codes) {
  const unsigned lane = zh_lane();
  constexpr int kPer = (kMaxSyms + 63) / 64;  // symbols a lane: lane, lane + 64, ...
  uint32_t l[kPer], hist[16];
#pragma unroll
  for (int L = 0; L < 16; L++) hist[L] = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int sidx = (int)lane + 64 * k;
    l[k] = sidx < num_codes ? lens[sidx] : 0u;
#pragma unroll
    for (int L = 1; L < 16; L++) hist[L] += (uint32_t)__popcll(__ballot(l[k] == (uint32_t)L));
  }
  uint32_t next_code[16];
  next_code[0] = 0;
  hist[0] = 0;
#pragma unroll
  for (int L = 1; L < 16; L++) next_code[L] = (next_code[L - 1] + hist[L - 1]) << 1;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int sidx = (int)lane + 64 * k;
    uint32_t c = 0;
#pragma unroll
    for (int L = 1; L < 16; L++) {
      const uint64_t m = __ballot(l[k] == (uint32_t)L);
      if (l[k] == (uint32_t)L) c = next_code[L] + (uint32_t)__popcll(m & zh_lanemask_lt());
      next_code[L] += (uint32_t)__popcll(m);
    }
    if (l[k]) codes[sidx] = (uint16_t)(rev16(c & 0xffffu) >> (16 - l[k]));
  }
  zh_wave_sync();
}

// ---- Nim std/heapqueue (CPython heapq) on nodes, `<` on frequency only: the reference's tie-breaks ARE the heap's
// mechanics, so the heap is replayed move for move.  A key is frequency << 32 | node (a frequency is a sum over one
// block, <= 4 MiB of symbols: the high dword compares).  One lane replaying it is a chain of ~ 240 instructions a pop
// at ~ 10 cycles each (measured: three heap levels an LDS trip with scalar decisions cost what a level a trip had,
// 5200 cycles a merge, 80 % of this kernel), so the WAVE does a pop: the 62 nodes of the five levels below the hole a
// lane each.  A lane reads its node and its right neighbour, two ballots say where the right child is the one to
// take and which nodes are larger than the item, the scalar unit walks those bits down the smaller children (the
// path only depends on what is in the heap), and the lanes on the path write their nodes a level up -- all at once.
// heappop's _siftup sinks the last item to a leaf and lets it climb back while it is smaller than its parent; along
// that path frequencies do not decrease, so it ends right above the first child that is larger than the item, with
// everything below back in place: the descent stops there instead.  A push is the mirror: a lane an ancestor. ----
struct HeapLane {  // a lane's node in the tree below a hole: the hole is node 0, node t's children 2t + 1 and 2t + 2
  int shift;       // its level, 1..5
  int offset;      // heap index = ((hole + 1) << shift) + offset
  bool in_tree;    // lanes 1..62
};
__device__ inline HeapLane heap_lane() {
  const int t = (int)zh_lane();
  HeapLane h;
  h.shift = 31 - __clz(t + 1);
  h.offset = t - (1 << h.shift);
  h.in_tree = t >= 1 && t <= 62;
  return h;
}
// heappush: the item goes in at `pos` (the heap's length before) and climbs while it is smaller than its parent.
// Lane k holds ancestor k + 1 levels up; every lane of the wave calls, pos and item are the same in all.
__device__ inline void heap_push(HuffWork& w, int pos, uint2 item) {
  uint2* const hk = reinterpret_cast<uint2*>(w.hkey);
  const int k = (int)zh_lane();
  const int up = k < 9 ? (pos + 1) >> (k + 1) : 0;  // pos < 512
  const uint2 key = hk[up ? up - 1 : 0];
  const uint64_t smaller = __ballot(up != 0 && item.y < key.y);
  const int climb = __ffsll((long long)~smaller) - 1;  // (frequencies do not increase towards the root: a run of ones)
  if (k < climb) hk[((pos + 1) >> k) - 1] = key;        // the ancestors it passes, each a level down
  if (k == 0) hk[((pos + 1) >> climb) - 1] = item;
}
// heappop; every lane of the wave calls.
__device__ inline uint2 heap_pop(HuffWork& w, int& len, const HeapLane& hl) {
  uint2* const hk = reinterpret_cast<uint2*>(w.hkey);
  const unsigned lane = zh_lane();
  --len;
  const uint2 last = hk[len];
  if (len == 0) return last;
  const uint2 result = hk[0];
  const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)last.y);
  int pos = 0;
  while (2 * pos + 1 < len) {
    const int at = ((pos + 1) << hl.shift) + hl.offset;
    const bool have = hl.in_tree && at < len, have_next = hl.in_tree && at + 1 < len;
    const uint2 key = hk[have ? at : 0], next = hk[have_next ? at + 1 : 0];
    const uint32_t fk = have ? key.y : 0xffffffffu, fn = have_next ? next.y : 0xffffffffu;
    // bit t, t odd (a left child): its right neighbour is the one to take (heapqueue's `not (heap[childpos] <
    // heap[rightpos])`); bit t: node t is larger than the item (so are the nodes that are not there)
    const uint64_t right = __ballot(!(fk < fn)), larger = __ballot(fk > f);
    // the smaller children all the way down (no decisions: five steps of bit arithmetic), then where the item stops: the
    // first node of the path that is larger -- node numbers grow along a path --; what lies before it moves up
    uint64_t path = 0;
    int t = 0;
#pragma unroll
    for (int step = 0; step < 5; step++) {
      t = 2 * t + 1 + (int)((right >> (2 * t + 1)) & 1ull);
      path |= 1ull << t;
    }
    const uint64_t stop = path & larger;
    const bool done = stop != 0;
    if (done) path &= (stop & (0ull - stop)) - 1ull;
    t = path ? 63 - __clzll((long long)path) : 0;  // the new hole
    if ((path >> lane) & 1ull) hk[(at - 1) >> 1] = key;
    zh_wave_sync();
    if (t) pos = __builtin_amdgcn_readlane(at, t);
    if (done) break;
  }
  if (lane == 0) hk[pos] = last;
  return result;
}

// deflate.nim:13-151 huffmanCodes, the reference's code symbol for symbol.  Every lane of the wave: what is a chain of
// decisions (the heap, the length limiting's quicksort) runs on lane 0, everything around it -- which symbols are
// used, the leaves' depths (pointer doubling over the parent links), lengths and canonical codes -- on the wave.
// freq / codes / lens in LDS.  Returns the number of codes.
#ifdef ZH_KPROF
#define HPROF_MARK(i)                                               \
  do {                                                              \
    const unsigned long long hp_now = __builtin_readcyclecounter(); \
    hp_acc[i] += hp_now - hp_t;                                     \
    hp_t = hp_now;                                                  \
  } while (0)
#else
#define HPROF_MARK(i) ((void)0)
#endif
__device__ int huffman_codes(const uint32_t* freq, int num_freq, int min_codes, int limit,
                             uint16_t* codes, uint8_t* lens, HuffWork& w, bool prof = false) {
  const unsigned lane = zh_lane();
#ifdef ZH_KPROF
  // tuning builds, the literal / length code: 0 used symbols, 1 leaves pushed, 2 merges, 3 depths, 4 the limit's
  // histogram, 5 its sort, 6 its lengths, 7 lengths and codes out (slots 56..63)
  unsigned long long hp_acc[8] = {}, hp_t = __builtin_readcyclecounter();
#endif
  constexpr int kPer = (kMaxSyms + 63) / 64;
  // while the heap runs: the used symbols' frequencies, in symbol order, where the depths will be; the nodes' parents
  // where round 4 kept the children (left and right are neighbours)
  uint32_t* const leaf_f = reinterpret_cast<uint32_t*>(w.depth);
  uint16_t* const par = w.left;
  static_assert(2 * kMaxSyms * sizeof(uint16_t) >= kMaxSyms * sizeof(uint32_t) &&
                    offsetof(HuffWork, right) == offsetof(HuffWork, left) + kMaxSyms * sizeof(uint16_t), "");
  uint32_t f[kPer];
  int leaf[kPer];  // the symbol's place among the used ones
  int highest = 0, used = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int sidx = (int)lane + 64 * k;
    f[k] = sidx < num_freq ? freq[sidx] : 0u;
    const uint64_t m = __ballot(f[k] != 0u);
    leaf[k] = used + __popcll(m & zh_lanemask_lt());
    used += __popcll(m);
    if (m) highest = 64 * k + 63 - __clzll((long long)m);
  }
  const int num_codes = (highest > min_codes ? highest : min_codes) + 1;
  for (int i = (int)lane; i < num_codes; i += 64) {
    codes[i] = 0;
    lens[i] = 0;
  }
  zh_wave_sync();
  if (used <= 1) {  // :34-45
    if (lane == 0) {
      if (used == 0) {
        lens[0] = 1;
        lens[1] = 1;
      } else {
        lens[highest] = 1;
        lens[highest == 0 ? 1 : 0] = 1;
      }
    }
  } else {
    const int n = used;
#pragma unroll
    for (int k = 0; k < kPer; k++)
      if (f[k]) {
        w.symbol[leaf[k]] = (int16_t)((int)lane + 64 * k);
        w.order[leaf[k]] = (uint16_t)leaf[k];
        leaf_f[leaf[k]] = f[k];
      }
    zh_wave_sync();
    HPROF_MARK(0);
    {
      const HeapLane hl = heap_lane();
      for (int i = 0; i < n; i++) {  // :54-55 push the leaves, in symbol order
        heap_push(w, i, make_uint2((uint32_t)i, leaf_f[i]));
        zh_wave_sync();
      }
      int hlen = n, total = n;
      HPROF_MARK(1);
      while (hlen >= 2) {  // :57-63
        const uint2 l = heap_pop(w, hlen, hl);
        zh_wave_sync();
        const uint2 r = heap_pop(w, hlen, hl);
        zh_wave_sync();
        if (lane == 0) {
          par[l.x] = (uint16_t)total;
          par[r.x] = (uint16_t)total;
        }
        heap_push(w, hlen++, make_uint2((uint32_t)total, l.y + r.y));
        zh_wave_sync();
        total++;
      }
      if (lane == 0) par[total - 1] = (uint16_t)(total - 1);  // the root
    }
    zh_wave_sync();
    HPROF_MARK(2);
    // :65-75 leaf depths: dep += dep[par], par = par[par] until every node hangs on the root
    const int nodes = 2 * n - 1;
    for (int v = (int)lane; v < nodes; v += 64) w.depth[v] = v == nodes - 1 ? 0 : 1;
    zh_wave_sync();
    constexpr int kNodesPer = (2 * kMaxSyms + 63) / 64;
    for (;;) {
      uint32_t np[kNodesPer], nd[kNodesPer], upd = 0;
#pragma unroll
      for (int k = 0; k < kNodesPer; k++) {
        const int v = (int)lane + 64 * k;
        if (v < nodes) {
          const uint32_t p1 = par[v];
          if (p1 != (uint32_t)(nodes - 1) && p1 != (uint32_t)v) {
            np[k] = par[p1];
            nd[k] = (uint32_t)w.depth[v] + w.depth[p1];
            upd |= 1u << k;
          }
        }
      }
      zh_wave_sync();  // (every lane has read before any lane writes)
#pragma unroll
      for (int k = 0; k < kNodesPer; k++) {
        const int v = (int)lane + 64 * k;
        if ((upd >> k) & 1u) {
          w.depth[v] = (uint16_t)nd[k];
          par[v] = (uint16_t)np[k];
        }
      }
      zh_wave_sync();
      if (!__ballot(upd != 0u)) break;
    }
    HPROF_MARK(3);
    uint32_t deepest = 0;
    for (int v = (int)lane; v < n; v += 64) deepest = w.depth[v] > deepest ? w.depth[v] : deepest;
    const int longest = __builtin_amdgcn_readlane((int)zh_wave_scan_max(deepest), 63);
    if (longest > limit && lane == 0) {  // :78-131 (a decision a step: one lane)
      uint16_t* const stack_ = w.stack();
      int32_t* const hist_ = w.histogram();
      for (int i = 0; i <= longest; i++) hist_[i] = 0;
      for (int i = 0; i < n; i++) hist_[w.depth[i]]++;
      int i = longest;
      while (i > limit) {
        if (hist_[i] == 0) {
          i--;
          continue;
        }
        int j = i - 2;
        while (j > 0 && hist_[j] == 0) j--;
        hist_[i] -= 2;
        hist_[i - 1]++;
        hist_[j + 1] += 2;
        hist_[j]--;
      }
      // :103-123 quickSort(nodes by depth), explicit stack instead of recursion
      HPROF_MARK(4);
      int sp = 0;
      stack_[sp++] = 0;
      stack_[sp++] = (uint16_t)(n - 1);
      while (sp > 0) {
        const int inr = (int16_t)stack_[--sp];
        const int inl = (int16_t)stack_[--sp];
        int r = inr, l = inl;
        const int cnt = r - l + 1;
        if (cnt < 2) continue;
        const uint16_t p = w.depth[w.order[l + 3 * cnt / 4]];
        while (l <= r) {
          if (w.depth[w.order[l]] < p) {
            l++;
          } else if (w.depth[w.order[r]] > p) {
            r--;
          } else {
            const uint16_t t = w.order[l];
            w.order[l] = w.order[r];
            w.order[r] = t;
            l++;
            r--;
          }
        }
        stack_[sp++] = (uint16_t)inl;
        stack_[sp++] = (uint16_t)(int16_t)r;
        stack_[sp++] = (uint16_t)l;
        stack_[sp++] = (uint16_t)inr;
      }
      HPROF_MARK(5);
      int code_len = 1;
      for (int k = 0; k < n; k++) {  // :125-131
        while (hist_[code_len] == 0) code_len++;
        w.depth[w.order[k]] = (uint16_t)code_len;
        hist_[code_len]--;
      }
    }
    zh_wave_sync();
    HPROF_MARK(6);
    for (int i = (int)lane; i < n; i += 64) lens[w.symbol[i]] = (uint8_t)w.depth[i];
  }
  zh_wave_sync();
  canonical_codes(lens, codes, num_codes);
#ifdef ZH_KPROF
  HPROF_MARK(7);
  if (prof && lane == 0)
    for (int i = 0; i < 8; i++) atomicAdd(&zh_kprof_slots[56 + i], hp_acc[i]);
#endif
  return num_codes;
}


// ---------------------------------------------------------------------------
// Contract mode (zh_set_l1_parse(ctx, 1): valid streams of about the reference's size, not its bytes): an
// optimal length-limited prefix code WITHOUT the replay of Nim's heapqueue (above: ~ 70 dependent LDS trips a
// symbol for the sake of its tie-breaks, 1.2 ms a block on one lane).  Any optimal code costs the same payload
// bits, so: the used symbols are ranked by (frequency, symbol) by the whole wave, one lane merges them with the
// two-queue method (leaves and internal nodes both come in ascending order: no heap), depths come from pointer
// doubling over the parent links, codes deeper than the limit are repaired on the histogram of lengths the way
// miniz / zlib do (take a code from the deepest level, split the deepest shorter one) and the lengths are
// handed out by rank -- the longest to the rarest --, canonical codes by ballots.  RFC 1951 3.2.2.
// Same special cases as deflate.nim:34-45 (no symbol / one symbol used), same `numCodes` rule (:24-32).
// ---------------------------------------------------------------------------
struct FastWork {  // overlays HuffWork (5.8 KiB)
  uint32_t lf[kMaxSyms];       // leaf frequencies, ascending
  uint32_t nf[kMaxSyms];       // internal nodes' frequencies, in creation (= ascending) order
  uint16_t sym_of[kMaxSyms];   // rank -> symbol
  uint16_t par[2 * kMaxSyms];  // node -> parent (leaves 0 .. n-1 by rank, internal nodes n .. 2n-2)
  uint16_t dep[2 * kMaxSyms];  // node -> depth
};
static_assert(sizeof(FastWork) <= sizeof(HuffWork) + 0, "FastWork lives in HuffWork's bytes");

// every lane of the wave; freq / codes / lens in LDS.  Returns the number of codes.
__device__ int huffman_codes_fast(const uint32_t* freq, int num_freq, int min_codes, int limit, uint16_t* codes,
                                  uint8_t* lens, FastWork& w, uint32_t* s_num) {
  const unsigned lane = zh_lane();
  constexpr int kPer = (kMaxSyms + 63) / 64;  // symbols a lane: lane, lane + 64, ...
  // ---- used symbols, the highest of them ----
  uint32_t f[kPer];
  int highest = 0, used = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int sidx = (int)lane + 64 * k;
    f[k] = sidx < num_freq ? freq[sidx] : 0u;
    const uint64_t m = __ballot(f[k] != 0u);
    used += __popcll(m);
    if (m) highest = 64 * k + 63 - __clzll((long long)m);
  }
  const int num_codes = (highest > min_codes ? highest : min_codes) + 1;
  for (int i = (int)lane; i < num_codes; i += 64) {
    codes[i] = 0;
    lens[i] = 0;
  }
  zh_wave_sync();
  if (used <= 1) {  // deflate.nim:34-45
    if (lane == 0) {
      if (used == 0) {
        lens[0] = 1;
        lens[1] = 1;
      } else {
        lens[highest] = 1;
        lens[highest == 0 ? 1 : 0] = 1;
      }
    }
  } else {
    const int n = used;
    // ---- rank by (frequency, symbol): how many used symbols come before mine ----
    uint32_t rk[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) rk[k] = 0;
    for (int j = 0; j < num_freq; j++) {
      const uint32_t fj = freq[j];  // (one address for the wave: a broadcast)
      if (fj == 0u) continue;
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        const int sidx = (int)lane + 64 * k;
        rk[k] += (fj < f[k] || (fj == f[k] && j < sidx)) ? 1u : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int sidx = (int)lane + 64 * k;
      if (f[k]) {
        w.lf[rk[k]] = f[k];
        w.sym_of[rk[k]] = (uint16_t)sidx;
      }
    }
    zh_wave_sync();
    // ---- two-queue merge (one lane): the two smallest of {next leaves, next internal nodes}, n - 1 times ----
    if (lane == 0) {
      // the heads of both queues in registers (what a pick needs is there; what replaces it is asked for a pick or
      // two before it is compared): a merge is ~ 30 instructions, not a chain of LDS trips
      constexpr uint32_t kInf = 0xffffffffu;
      int i = 0, j = 0;  // next leaf, next internal node
      uint32_t la = w.lf[0], lb = w.lf[1], na = kInf, nb = kInf;  // (n >= 2)
      for (int m = 0; m < n - 1; m++) {
        uint32_t sum = 0;
#pragma unroll
        for (int t = 0; t < 2; t++) {
          if (la <= na) {  // (ties: the leaf -- the flatter tree)
            sum += la;
            w.par[i] = (uint16_t)(n + m);
            i++;
            la = lb;
            lb = i + 1 < n ? w.lf[i + 1] : kInf;
          } else {
            sum += na;
            w.par[n + j] = (uint16_t)(n + m);
            j++;
            na = nb;
            nb = j + 1 < m ? w.nf[j + 1] : kInf;  // (node m is not made yet)
          }
        }
        w.nf[m] = sum;
        // node m joins its queue: as the head, as the one behind it, or further back (then it is read when its turn comes)
        if (j == m) na = sum;
        else if (j + 1 == m) nb = sum;
      }
      w.par[2 * n - 2] = (uint16_t)(2 * n - 2);  // the root
    }
    zh_wave_sync();
    // ---- depths: dep += dep[par], par = par[par] until every node hangs on the root ----
    const int nodes = 2 * n - 1;
    for (int v = (int)lane; v < nodes; v += 64) w.dep[v] = v == nodes - 1 ? 0 : 1;
    zh_wave_sync();
    constexpr int kNodesPer = (2 * kMaxSyms + 63) / 64;
    for (;;) {
      uint32_t np[kNodesPer], nd[kNodesPer], upd = 0;
#pragma unroll
      for (int k = 0; k < kNodesPer; k++) {
        const int v = (int)lane + 64 * k;
        if (v < nodes) {
          const uint32_t p1 = w.par[v];
          if (p1 != (uint32_t)(nodes - 1) && p1 != (uint32_t)v) {
            np[k] = w.par[p1];
            nd[k] = (uint32_t)w.dep[v] + w.dep[p1];
            upd |= 1u << k;
          }
        }
      }
      zh_wave_sync();  // (every lane has read before any lane writes)
#pragma unroll
      for (int k = 0; k < kNodesPer; k++) {
        const int v = (int)lane + 64 * k;
        if ((upd >> k) & 1u) {
          w.dep[v] = (uint16_t)nd[k];
          w.par[v] = (uint16_t)np[k];
        }
      }
      zh_wave_sync();
      if (!__ballot(upd != 0u)) break;
    }
    // ---- histogram of the leaves' depths, everything deeper than the limit counted at the limit ----
    if (lane < 32) s_num[lane] = 0;
    zh_wave_sync();
    for (int v = (int)lane; v < n; v += 64) {
      const uint32_t d = w.dep[v];
      atomicAdd(&s_num[d < (uint32_t)limit ? d : (uint32_t)limit], 1u);
    }
    zh_wave_sync();
    // lane L holds num[L]; Kraft sum in units of 2^-limit
    uint32_t num = lane >= 1 && (int)lane <= limit ? s_num[lane] : 0u;
    uint32_t total = zh_wave_sum(lane >= 1 && (int)lane <= limit ? num << (limit - (int)lane) : 0u);
    while (total > (1u << limit)) {  // over-subscribed: a code leaves the deepest level, the deepest shorter code splits
      const uint64_t have = __ballot(num != 0u) & ((1ull << limit) - 2ull);  // levels 1 .. limit - 1 in use
      const int i = 63 - __clzll((long long)have);                           // (there is one: total > 2^limit)
      if ((int)lane == limit) num -= 1u;
      if ((int)lane == i) num -= 1u;
      if ((int)lane == i + 1) num += 2u;
      total--;
    }
    // ---- lengths by rank: the num[1] most frequent symbols get one bit, the next num[2] two, ... ----
    const uint32_t incl = zh_wave_scan(num);  // lane L: symbols with at most L bits
    uint32_t le[16];
#pragma unroll
    for (int L = 1; L < 16; L++) le[L] = (uint32_t)__builtin_amdgcn_readlane((int)incl, L);
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int sidx = (int)lane + 64 * k;
      if (f[k]) {
        const uint32_t desc = (uint32_t)(n - 1) - rk[k];  // 0: the most frequent symbol
        uint32_t L = 1;
#pragma unroll
        for (int q = 1; q < 15; q++) L += desc >= le[q] ? 1u : 0u;
        lens[sidx] = (uint8_t)L;
      }
    }
  }
  zh_wave_sync();
  canonical_codes(lens, codes, num_codes);
  return num_codes;
}

struct HdrWriter {
  uint32_t* words;
  uint32_t bits;
};
__device__ inline void hdr_add(HdrWriter& h, uint32_t value, uint32_t n) {
  if (!n) return;
  const uint32_t w = h.bits >> 5, s = h.bits & 31u;
  h.words[w] |= value << s;
  if (s + n > 32) h.words[w + 1] |= value >> (32 - s);
  h.bits += n;
}

// the same into LDS words that other lanes are writing at the same time
__device__ inline void hdr_add_shared(HdrWriter& h, uint32_t value, uint32_t n) {
  if (!n) return;
  const uint32_t w = h.bits >> 5, s = h.bits & 31u;
  atomicOr(&h.words[w], value << s);
  if (s + n > 32) atomicOr(&h.words[w + 1], value >> (32 - s));
  h.bits += n;
}

// OR `nbits` (<= 32) of value at absolute bit position `bit` of the byte stream at base.
__device__ inline void or_bits(uint8_t* base, uint64_t bit, uint32_t value, uint32_t nbits) {
  if (!nbits) return;
  if (nbits < 32) value &= (1u << nbits) - 1u;
  const uintptr_t addr = (uintptr_t)base + (bit >> 3);
  uint32_t* wp = reinterpret_cast<uint32_t*>(addr & ~(uintptr_t)3);
  const uint32_t s = (uint32_t)((addr & 3u) * 8u + (bit & 7u));  // 0..31
  atomicOr(wp, value << s);
  if (s + nbits > 32) atomicOr(wp + 1, value >> (32 - s));
}
__device__ inline void or_byte(uint8_t* base, uint64_t byte_pos, uint32_t v) {
  or_bits(base, byte_pos * 8, v & 0xffu, 8);
}

}  // namespace

// contract = true: the block's three codes by huffman_codes_fast (optimal codes, not the reference's tie-breaks:
// zh_set_l1_parse(ctx, 1)); false: by the replay of deflate.nim:13-151, byte-identical.  (Two kernels: the fast
// builder's registers would cost the replay, which needs few, a quarter of its waves.)
template <bool contract>
__global__ __launch_bounds__(64) void zh_huffman_kernel(ZhCompressArgs a) {
  __shared__ uint32_t s_freq[ZH_HIST_STRIDE];
  __shared__ HuffWork s_work;
  __shared__ uint16_t s_codes[ZH_HIST_STRIDE];  // litlen at 0, distance at 288
  __shared__ uint8_t s_lens[ZH_HIST_STRIDE];
  __shared__ __attribute__((aligned(16))) uint8_t s_cl_all[ZH_HIST_STRIDE];
  __shared__ uint32_t s_num[32];  // the fast builder's length counts
  __shared__ uint32_t s_hdr[ZH_HDR_WORDS];
  __shared__ uint32_t s_mode, s_hdr_bits;
  // (the kernel's LDS stays under 10 KiB -- sixteen blocks a CU, 4096 blocks one round of the machine --, so the
  // small arrays of the later stages live in bytes that are dead by then)
  uint32_t* const s_clfreq = s_freq;        // [32] the code-length alphabet's histogram: the block's own is done with
  uint16_t* const s_clcodes = reinterpret_cast<uint16_t*>(s_freq + 32);  // [20]
  uint8_t* const s_cllens = reinterpret_cast<uint8_t*>(s_freq + 48);     // [20]
  int* const s_n = reinterpret_cast<int*>(s_freq + 64);  // [2] litlen codes, distance codes (behind the litlen build)

  const unsigned lane = zh_lane();
  KPROF_DECL(8);
  const uint32_t b = blockIdx.x;
  const ZhBlockDesc bd = a.blocks[b];
  const int level = a.level;
  FastWork& fwork = *reinterpret_cast<FastWork*>(&s_work);

  // ---- block histogram = sum of the fragment histograms (BlockMetadata) ----
  uint32_t nlit = 0;
  for (uint32_t i = lane; i < ZH_HIST_STRIDE; i += 64) {
    uint32_t acc = 0;
#pragma unroll 8
    for (uint32_t k = 0; k < bd.nfrag; k++) acc += a.f_hist[(size_t)(bd.first_frag + k) * ZH_HIST_STRIDE + i];  // (eight loads in flight)
    if (i == 256) acc = 1;  // always one end-of-block symbol (snappy.nim:145, lz77.nim:52)
    s_freq[i] = acc;
  }
  for (uint32_t k = lane; k < bd.nfrag; k += 64) nlit += a.f_nlit[bd.first_frag + k];
  nlit = zh_wave_sum(nlit);
  for (uint32_t i = lane; i < ZH_HDR_WORDS; i += 64) s_hdr[i] = 0;
  for (uint32_t i = lane; i < ZH_HIST_STRIDE; i += 64) {
    s_codes[i] = 0;
    s_lens[i] = 0;
  }
  zh_wave_sync();
  KPROF_MARK(0);

  HdrWriter h{s_hdr, 0};
  if (lane == 0) {
    uint32_t mode;
    if (level == 0 ||
        (level != -2 && (int64_t)nlit >= (int64_t)((float)bd.len * 0.98f))) {  // deflate.nim:274-277
      mode = ZH_MODE_STORED;
    } else if (level <= 6 && bd.len <= 2048) {  // deflate.nim:280
      mode = ZH_MODE_FIXED;
    } else {
      mode = ZH_MODE_DYNAMIC;
    }
    s_mode = mode;
    if (mode == ZH_MODE_FIXED) {  // internal.nim:151-175
      for (int i = 0; i < 288; i++) s_lens[i] = (uint8_t)(i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8);
      for (int i = 0; i < 30; i++) s_lens[288 + i] = 5;
      uint32_t next8 = 0x30, next9 = 0x190, next7 = 0;  // canonical first codes of lengths 8, 9, 7
      for (int i = 0; i < 288; i++) {
        const uint32_t l = s_lens[i];
        uint32_t c = l == 8 ? next8++ : l == 9 ? next9++ : next7++;
        s_codes[i] = (uint16_t)(rev16(c) >> (16 - l));
      }
      for (int i = 0; i < 30; i++) s_codes[288 + i] = (uint16_t)(rev16((uint32_t)i) >> 11);
      hdr_add(h, bd.is_final ? 1 : 0, 1);  // deflate.nim:296-298
      hdr_add(h, 1, 2);
    }
  }
  zh_wave_sync();
  if (s_mode == ZH_MODE_DYNAMIC) {
    // ---- the literal / length and the distance code ----
    if (contract) {
      const int nl = huffman_codes_fast(s_freq, ZH_NUM_LITLEN, 257, 15, s_codes, s_lens, fwork, s_num);
      KPROF_MARK(1);
      const int nd = huffman_codes_fast(s_freq + ZH_NUM_LITLEN, ZH_NUM_DIST, 2, 15, s_codes + 288, s_lens + 288, fwork, s_num);
      KPROF_MARK(2);
      if (lane == 0) {
        s_n[0] = nl;
        s_n[1] = nd;
      }
    } else {
      const int nl = huffman_codes(s_freq, ZH_NUM_LITLEN, 257, 15, s_codes, s_lens, s_work, true);
      KPROF_MARK(1);
      const int nd = huffman_codes(s_freq + ZH_NUM_LITLEN, ZH_NUM_DIST, 2, 15, s_codes + 288, s_lens + 288, s_work);
      KPROF_MARK(2);
      if (lane == 0) {
        s_n[0] = nl;
        s_n[1] = nd;
      }
    }
    zh_wave_sync();
    const int n_litlen = s_n[0], n_dist = s_n[1];
    // ---- deflate.nim:313-350: the code lengths, run-length coded -- by the whole wave.  The reference walks the
    // lengths one by one; what it emits for a maximal run of `run` equal lengths v is a closed form of (v, run):
    //   zeros, run >= 3:  a symbol 18 (138 zeros) per full 138, then for the rest r: >= 11 one 18, >= 3 one 17, else r zeros
    //   v != 0, run >= 4: v itself, a symbol 16 (repeat 6) per full six of the run - 1 behind it, then for the rest
    //                     r: >= 3 one 16, else r times v
    //   otherwise:        v, run times
    // so every run's first lane knows its items, their count a symbol (the code-length alphabet's histogram,
    // deflate.nim:352-360) and, once that alphabet has its code, their bits and where they go. ----
    const int num_codes = n_litlen + n_dist;
    constexpr int kG = 5;  // 316 lengths at most
    for (int i = (int)lane; i < n_litlen; i += 64) s_cl_all[i] = s_lens[i];
    for (int i = (int)lane; i < n_dist; i += 64) s_cl_all[n_litlen + i] = s_lens[288 + i];
    if (lane < 19) s_clfreq[lane] = 0;
    zh_wave_sync();
    uint32_t rv[kG], rrun[kG];  // a run's first lane: its value and length (0: this lane starts none)
    {
      uint64_t H[kG];
#pragma unroll
      for (int g = 0; g < kG; g++) {
        const int pos = 64 * g + (int)lane;
        rv[g] = pos < num_codes ? s_cl_all[pos] : 0xffu;
        const bool head = pos < num_codes && (pos == 0 || s_cl_all[pos - 1] != rv[g]);
        H[g] = __ballot(head);
      }
#pragma unroll
      for (int g = 0; g < kG; g++) {
        const int pos = 64 * g + (int)lane;
        rrun[g] = 0;
        if ((H[g] >> lane) & 1ull) {
          int next = num_codes;
          bool found = false;
          const uint64_t above = lane == 63 ? 0ull : H[g] & ~((2ull << lane) - 1ull);
          if (above) {
            next = 64 * g + __ffsll((long long)above) - 1;
            found = true;
          }
#pragma unroll
          for (int g2 = 0; g2 < kG; g2++)
            if (g2 > g && !found && H[g2]) {
              next = 64 * g2 + __ffsll((long long)H[g2]) - 1;
              found = true;
            }
          rrun[g] = (uint32_t)(next - pos);
        }
      }
    }
#pragma unroll
    for (int g = 0; g < kG; g++) {
      const uint32_t v = rv[g], run = rrun[g];
      if (!run) continue;
      if (v == 0 && run >= 3) {
        const uint32_t n18 = run / 138u, r = run % 138u;
        if (n18 + (r >= 11 ? 1u : 0u)) atomicAdd(&s_clfreq[18], n18 + (r >= 11 ? 1u : 0u));
        if (r >= 3 && r < 11) atomicAdd(&s_clfreq[17], 1u);
        if (r < 3 && r) atomicAdd(&s_clfreq[0], r);
      } else if (run >= 4) {
        const uint32_t q = (run - 1u) / 6u, r = (run - 1u) % 6u;
        atomicAdd(&s_clfreq[v], 1u + (r < 3 ? r : 0u));
        atomicAdd(&s_clfreq[16], q + (r >= 3 ? 1u : 0u));
      } else {
        atomicAdd(&s_clfreq[v], run);
      }
    }
    zh_wave_sync();
    KPROF_MARK(3);
    // ---- the code of the code lengths (deflate.nim:362) ----
    if (contract) {
      huffman_codes_fast(s_clfreq, 19, 19, 7, s_clcodes, s_cllens, fwork, s_num);
    } else {
      huffman_codes(s_clfreq, 19, 19, 7, s_clcodes, s_cllens, s_work);
    }
    zh_wave_sync();
    uint32_t hclen4 = 0;  // HCLEN + 4: the code-length code's lengths that go into the header
    {
      const uint32_t ordered = lane < 19 ? s_cllens[c_clcl_order[lane]] : 0u;
      const uint64_t nz = __ballot(ordered != 0u);
      hclen4 = nz ? 64u - (uint32_t)__clzll((long long)nz) : 0u;  // (one symbol is always used: the last nonzero is there)
      KPROF_MARK(4);
      // deflate.nim:376-386: BFINAL, BTYPE, HLIT, HDIST, HCLEN, the 3-bit lengths: 17 + 3 (HCLEN + 4) bits, lane 0 the
      // fixed fields, lane i the i-th length
      if (lane == 0) atomicOr(&s_hdr[0], (bd.is_final ? 1u : 0u) | (2u << 1) | ((uint32_t)(n_litlen - 257) << 3) |
                                             ((uint32_t)(n_dist - 1) << 8) | ((hclen4 - 4u) << 13));
      if (lane < hclen4) {
        const uint32_t at = 17u + 3u * lane;
        atomicOr(&s_hdr[at >> 5], ordered << (at & 31u));
        if ((at & 31u) > 29u) atomicOr(&s_hdr[(at >> 5) + 1u], ordered >> (32u - (at & 31u)));
      }
    }
    // ---- deflate.nim:388-401: the items' bits ----
    auto item_bits = [&](uint32_t sym) -> uint32_t {
      return (uint32_t)s_cllens[sym] + (sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u);
    };
    uint32_t cursor = 17u + 3u * hclen4;
#pragma unroll
    for (int g = 0; g < kG; g++) {
      const uint32_t v = rv[g], run = rrun[g];
      uint32_t bits = 0;
      if (run) {
        if (v == 0 && run >= 3) {
          const uint32_t n18 = run / 138u, r = run % 138u;
          bits = n18 * item_bits(18) + (r >= 11 ? item_bits(18) : r >= 3 ? item_bits(17) : r * item_bits(0));
        } else if (run >= 4) {
          const uint32_t q = (run - 1u) / 6u, r = (run - 1u) % 6u;
          bits = item_bits(v) * (1u + (r < 3 ? r : 0u)) + item_bits(16) * (q + (r >= 3 ? 1u : 0u));
        } else {
          bits = run * item_bits(v);
        }
      }
      const uint32_t incl = zh_wave_scan(bits);
      HdrWriter hw{s_hdr, cursor + incl - bits};
      auto put = [&](uint32_t sym, uint32_t extra) {  // the symbol's code, then its extra bits
        hdr_add_shared(hw, s_clcodes[sym], s_cllens[sym]);
        if (sym == 16u) hdr_add_shared(hw, extra, 2);
        else if (sym == 17u) hdr_add_shared(hw, extra, 3);
        else if (sym == 18u) hdr_add_shared(hw, extra, 7);
      };
      if (run) {
        if (v == 0 && run >= 3) {
          const uint32_t n18 = run / 138u, r = run % 138u;
          for (uint32_t k = 0; k < n18; k++) put(18, 138u - 11u);
          if (r >= 11) put(18, r - 11u);
          else if (r >= 3) put(17, r - 3u);
          else
            for (uint32_t k = 0; k < r; k++) put(0, 0);
        } else if (run >= 4) {
          const uint32_t q = (run - 1u) / 6u, r = (run - 1u) % 6u;
          put(v, 0);
          for (uint32_t k = 0; k < q; k++) put(16, 3);
          if (r >= 3) put(16, r - 3u);
          else
            for (uint32_t k = 0; k < r; k++) put(v, 0);
        } else {
          for (uint32_t k = 0; k < run; k++) put(v, 0);
        }
      }
      cursor += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    h.bits = cursor;
  }
  if (lane == 0) s_hdr_bits = h.bits;
  zh_wave_sync();
  KPROF_MARK(5);

  const uint32_t mode = s_mode;
  if (lane == 0) {
    a.b_mode[b] = mode;
    a.b_hdr_bits[b] = s_hdr_bits;
  }
  if (mode == ZH_MODE_STORED) return;

  for (uint32_t i = lane; i < 288; i += 64) a.b_litcode[(size_t)b * 288 + i] = s_codes[i] | ((uint32_t)s_lens[i] << 16);
  if (lane < 32) a.b_distcode[(size_t)b * 32 + lane] = lane < 30 ? (s_codes[288 + lane] | ((uint32_t)s_lens[288 + lane] << 16)) : 0u;
  for (uint32_t i = lane; i < ZH_HDR_WORDS; i += 64) a.b_hdr[(size_t)b * ZH_HDR_WORDS + i] = s_hdr[i];

  // ---- encoded size of every fragment under these codes: histogram x code lengths, four fragments' loads in
  // flight together (one after the other was a tenth of the kernel) ----
  uint64_t total = s_hdr_bits + s_lens[256];
  constexpr uint32_t kSymPer = (ZH_NUM_LITLEN + ZH_NUM_DIST + 63) / 64, kFr = 4;
  uint32_t ll[kSymPer];  // this lane's symbols' lengths
#pragma unroll
  for (uint32_t j = 0; j < kSymPer; j++) {
    const uint32_t i = lane + 64u * j;
    ll[j] = i < ZH_NUM_LITLEN ? s_lens[i] : i < ZH_NUM_LITLEN + ZH_NUM_DIST ? s_lens[288 + (i - ZH_NUM_LITLEN)] : 0u;
  }
  for (uint32_t k0 = 0; k0 < bd.nfrag; k0 += kFr) {
    uint32_t hv[kFr][kSymPer], extra[kFr];
#pragma unroll
    for (uint32_t u = 0; u < kFr; u++) {
      const uint32_t f = bd.first_frag + (k0 + u < bd.nfrag ? k0 + u : bd.nfrag - 1u);
      const uint16_t* hist = a.f_hist + (size_t)f * ZH_HIST_STRIDE;
#pragma unroll
      for (uint32_t j = 0; j < kSymPer; j++) {
        const uint32_t i = lane + 64u * j;
        hv[u][j] = hist[i < ZH_NUM_LITLEN + ZH_NUM_DIST ? i : 0u];
      }
      extra[u] = a.f_extra_bits[f];
    }
#pragma unroll
    for (uint32_t u = 0; u < kFr; u++) {
      if (k0 + u >= bd.nfrag) break;
      uint32_t acc = 0;
#pragma unroll
      for (uint32_t j = 0; j < kSymPer; j++) acc += hv[u][j] * ll[j];
      acc = zh_wave_sum(acc) + extra[u];
      if (lane == 0) a.f_bits[bd.first_frag + k0 + u] = acc;
      total += acc;
    }
  }
  if (lane == 0) a.b_bits[b] = total;
  KPROF_MARK(6);
  KPROF_COUNT(7, 1);
  KPROF_FLUSH(40, 8);
}

// Output slots are NOT cleared beforehand (round 6; until then a memset of every slot led each compress run).  Whole
// bytes -- container header, trailer, stored chunks' LEN / NLEN -- are plain stores; bits that share a word with another
// writer are OR-ed in (block headers, end-of-block codes, a fragment's first and last word: zh_emit.hip), and every word
// that is OR-ed into is zeroed first by a kernel -- or by the same wave -- that comes before all its writers:
//   zh_layout_kernel (a wave a buffer, before everything else): the word of every block's first bit and of the bit two
//     behind it (a stored block's three header bits and its padding), and the word of the body's last bit -- all clipped
//     to the body's own bytes: the neighbouring bytes of such a word may be the container header's, the trailer's or
//     another slot's;
//   zh_block_layout_kernel (a wave a block): the words strictly between its block's first word and the next block's
//     (the last block: the word of the body's last bit) that hold a header bit, a fragment's first bit or the end-of-block
//     code -- nobody else touches those before the emission --, then a fence, then its ORs.
// Everything else of a stream is whole words or bytes stored by their one owner.
__device__ inline void zero_word_clipped(uint8_t* d_dst, uint64_t abs_bit, uint64_t lo_byte, uint64_t hi_byte) {
  const uint64_t w0 = (abs_bit >> 5) << 2;
  const uint64_t lo = w0 > lo_byte ? w0 : lo_byte, hi = w0 + 4 < hi_byte ? w0 + 4 : hi_byte;
  if (lo >= hi) return;
  if (hi - lo == 4) {
    *reinterpret_cast<uint32_t*>(d_dst + w0) = 0u;
  } else {
    for (uint64_t x = lo; x < hi; x++) d_dst[x] = 0;
  }
}

// The trailer (after padding to a byte, deflate.nim:473): the source's checksum and, for gzip, its length.
__device__ inline void write_trailer(uint8_t* out, uint64_t tpos, int fmt, uint32_t crc, uint32_t adler, uint64_t src_len,
                                     unsigned lane) {
  if (fmt == ZH_DF_GZIP) {  // zippy.nim:47-58
    const uint32_t isize = (uint32_t)(src_len & 0xffffffffu);
    if (lane < 4) out[tpos + lane] = (uint8_t)(crc >> (8 * lane));
    else if (lane < 8) out[tpos + lane] = (uint8_t)(isize >> (8 * (lane - 4)));
  } else if (fmt == ZH_DF_ZLIB) {  // zippy.nim:71-78 (big-endian)
    if (lane < 4) out[tpos + lane] = (uint8_t)(adler >> (8 * (3 - lane)));
  }
}

// with_trailer = 0: the trailer comes later, from zh_trailer_kernel (the checksum is still on its way: it runs beside
// the code builder AND the emission, zh_plan_run.hip)
__global__ __launch_bounds__(64) void zh_layout_kernel(uint8_t* __restrict__ d_dst, ZhCompressArgs a,
                                                       const uint32_t* __restrict__ buf_crc,
                                                       const uint32_t* __restrict__ buf_adler, int with_trailer) {
  const unsigned lane = zh_lane();
  const uint32_t bi = blockIdx.x;
  const ZhBufDesc bd = a.bufs[bi];
  const int fmt = a.data_format;
  uint8_t* out = d_dst + bd.dst_off;

  const uint32_t hdr_len = fmt == ZH_DF_GZIP ? 10 + bd.fname_len + 1 : fmt == ZH_DF_ZLIB ? 2 : 0;
  const uint32_t trailer_len = fmt == ZH_DF_GZIP ? 8 : fmt == ZH_DF_ZLIB ? 4 : 0;

  // ---- where every block starts (bits from the start of the deflate body), 64 blocks at a time ----
  auto lane64 = [](uint64_t v, uint32_t j) -> uint64_t {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)j) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)j) << 32);
  };
  uint64_t cursor = 0;
  for (uint32_t k0 = 0; k0 < bd.nblocks; k0 += 64) {
    const uint32_t k = k0 + lane;
    const bool have = k < bd.nblocks;
    const uint32_t b = bd.first_block + k;
    const uint32_t mode = have ? a.b_mode[b] : (uint32_t)ZH_MODE_DYNAMIC;
    const uint64_t bits = have && mode != ZH_MODE_STORED ? a.b_bits[b] : 0ull;
    const uint64_t blen = have ? a.blocks[b].len : 0ull;
    uint64_t start = 0;
    if (!__ballot(have && mode == ZH_MODE_STORED)) {
      // 64 compressed blocks sum to < 2^32 bits (64 * 4 MiB * 15 bits + headers)
      const uint32_t incl = zh_wave_scan((uint32_t)bits);
      start = cursor + (incl - (uint32_t)bits);
      cursor += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    } else {
      // a stored block pads to a byte boundary (deflate.nim:179-205): not additive, walk the batch
      const uint32_t cnt = bd.nblocks - k0 < 64u ? bd.nblocks - k0 : 64u;
      for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)mode, (int)j);
        if (lane == j) start = cursor;
        if (mj == ZH_MODE_STORED) {
          const uint64_t lj = lane64(blen, j);
          uint64_t chunks = (lj + ZH_STORED_MAX - 1) / ZH_STORED_MAX;
          if (chunks < 1) chunks = 1;
          const uint64_t first_len_byte = (cursor + 3 + 7) >> 3;
          cursor = (first_len_byte + 4 + lj + 5 * (chunks - 1)) * 8;
        } else {
          cursor += lane64(bits, j);
        }
      }
    }
    if (have) a.b_start[b] = start;
  }
  if (lane == 0) a.b_start[a.nblocks + bi] = cursor;  // end of the last block (closing index entry)
  const uint64_t body_bytes = (cursor + 7) >> 3;
  const uint64_t total_len = hdr_len + body_bytes + trailer_len;
  if (total_len > bd.dst_cap) {  // the block layout kernel skips this buffer
    if (lane == 0) {
      a.out_len[bi] = total_len;
      a.status[bi] = ZH_ERR_DST_TOO_SMALL;
    }
    return;
  }

  // ---- the seams between blocks, zeroed for the kernels that OR into them (see above) ----
  {
    const uint64_t body0 = bd.dst_off + hdr_len, body1 = body0 + body_bytes;  // the body's bytes in d_dst
    for (uint32_t k = lane; k < bd.nblocks; k += 64) {
      const uint64_t s0 = body0 * 8 + a.b_start[bd.first_block + k];  // (this lane's own store above)
      zero_word_clipped(d_dst, s0, body0, body1);
      zero_word_clipped(d_dst, s0 + 2, body0, body1);
    }
    if (lane == 0 && cursor) zero_word_clipped(d_dst, body0 * 8 + cursor - 1, body0, body1);
  }
  // ---- container header ----
  if (fmt == ZH_DF_GZIP) {  // zippy.nim:22-42: 1f 8b 08 08 00 x 6, the FNAME letters, its NUL
    if (lane < 10) out[lane] = lane == 0 ? 31 : lane == 1 ? 139 : lane == 2 ? 8 : lane == 3 ? (1u << 3) : 0;
    if (lane < bd.fname_len) out[10 + lane] = (uint8_t)(97 + lane);
    if (lane == 0) out[10 + bd.fname_len] = 0;
  } else if (fmt == ZH_DF_ZLIB) {  // zippy.nim:61-69
    const uint32_t cmf = (7u << 4) | 8u;
    if (lane == 0) out[0] = (uint8_t)cmf;
    if (lane == 1) out[1] = (uint8_t)(31u - (cmf * 256u) % 31u);
  }

  if (with_trailer)
    write_trailer(out, hdr_len + body_bytes, fmt, fmt == ZH_DF_GZIP ? buf_crc[bi] : 0u, fmt == ZH_DF_ZLIB ? buf_adler[bi] : 0u,
                  bd.src_len, lane);
  if (lane == 0) {
    a.out_len[bi] = total_len;
    a.status[bi] = ZH_OK;
  }
}

// One wave per buffer, behind everything else of a compress run: the trailer of a buffer whose layout went without.
__global__ __launch_bounds__(64) void zh_trailer_kernel(uint8_t* __restrict__ d_dst, ZhCompressArgs a,
                                                        const uint32_t* __restrict__ buf_crc,
                                                        const uint32_t* __restrict__ buf_adler) {
  const unsigned lane = zh_lane();
  const uint32_t bi = blockIdx.x;
  if (a.status[bi] != ZH_OK) return;
  const ZhBufDesc bd = a.bufs[bi];
  const int fmt = a.data_format;
  const uint32_t hdr_len = fmt == ZH_DF_GZIP ? 10 + bd.fname_len + 1 : fmt == ZH_DF_ZLIB ? 2 : 0;
  const uint64_t body_bytes = (a.b_start[a.nblocks + bi] + 7) >> 3;  // (the layout's closing entry: the body's bits)
  write_trailer(d_dst + bd.dst_off, hdr_len + body_bytes, fmt, fmt == ZH_DF_GZIP ? buf_crc[bi] : 0u,
                fmt == ZH_DF_ZLIB ? buf_adler[bi] : 0u, bd.src_len, lane);
}

// One wave per block, after zh_layout_kernel: the block's header, the bit position of each of its
// fragments, its end-of-block code; for a stored block the chunk headers.
__global__ __launch_bounds__(64) void zh_block_layout_kernel(uint8_t* __restrict__ d_dst, ZhCompressArgs a) {
  const unsigned lane = zh_lane();
  const uint32_t b = blockIdx.x;
  const ZhBlockDesc blk = a.blocks[b];
  const ZhBufDesc bd = a.bufs[blk.buf];
  if (a.status[blk.buf] != ZH_OK) {  // poison the fragments so that the emission kernel skips them
    for (uint32_t j = lane; j < blk.nfrag; j += 64) a.f_bit_start[blk.first_frag + j] = ~0ull;
    return;
  }
  const int fmt = a.data_format;
  uint8_t* out = d_dst + bd.dst_off;
  const uint32_t hdr_len = fmt == ZH_DF_GZIP ? 10 + bd.fname_len + 1 : fmt == ZH_DF_ZLIB ? 2 : 0;
  const uint64_t body_bit0 = (uint64_t)hdr_len * 8;  // relative to `out`
  uint64_t cursor = a.b_start[b];
  if (a.b_mode[b] == ZH_MODE_STORED) {
    uint64_t chunks = (blk.len + ZH_STORED_MAX - 1) / ZH_STORED_MAX;
    if (chunks < 1) chunks = 1;
    const uint64_t first_len_byte = (body_bit0 + cursor + 3 + 7) >> 3;  // relative to out
    const uint64_t d0 = first_len_byte + 4;  // block byte o lands at d0 + o + 5 * (o / 65535)
    for (uint64_t c = lane; c < chunks; c += 64) {
      const uint32_t fin = (blk.is_final && c == chunks - 1) ? 1u : 0u;
      const uint64_t clen = (c == chunks - 1) ? blk.len - c * ZH_STORED_MAX : ZH_STORED_MAX;
      const uint64_t len_byte = d0 - 4 + c * (ZH_STORED_MAX + 5ull);
      // BFINAL + BTYPE = 00, then padding: the first chunk's three bits share their bytes with the block before (zeroed
      // by zh_layout_kernel), a later chunk's byte is its own; LEN and NLEN are whole bytes
      if (c == 0) or_bits(out, body_bit0 + cursor, fin, 3);
      else out[len_byte - 1] = (uint8_t)fin;
      const uint32_t ln = (uint32_t)clen | ((uint32_t)(ZH_STORED_MAX - clen) << 16);
      for (uint32_t i = 0; i < 4; i++) out[len_byte + i] = (uint8_t)(ln >> (8u * i));
    }
    if (lane == 0) a.b_stored_d0[b] = bd.dst_off + d0;
    return;
  }
  const uint32_t hbits = a.b_hdr_bits[b];
  const uint32_t* hdr = a.b_hdr + (size_t)b * ZH_HDR_WORDS;
  const uint32_t eob = a.b_litcode[(size_t)b * 288 + 256];
  // words of d_dst this wave may zero: strictly between its block's first word and the next block's first word (the last
  // block of a buffer: the word of the body's last bit); those two are zh_layout_kernel's
  const uint64_t abs0 = (bd.dst_off + hdr_len) * 8;  // the body's first bit in d_dst
  const uint64_t w_lo = (abs0 + cursor) >> 5;
  const uint64_t w_hi = blk.is_final ? (abs0 + a.b_start[a.nblocks + blk.buf] - 1) >> 5 : (abs0 + a.b_start[b + 1]) >> 5;
  uint32_t* const dwords = reinterpret_cast<uint32_t*>(d_dst);
  auto zero_inner = [&](uint64_t abs_bit) {
    const uint64_t w = abs_bit >> 5;
    if (w > w_lo && w < w_hi) dwords[w] = 0u;
  };
  for (uint32_t w = lane; w * 32 < hbits; w += 64) zero_inner(abs0 + cursor + (uint64_t)w * 32);
  if (lane == 0 && hbits) zero_inner(abs0 + cursor + hbits - 1);
  uint64_t fcur = cursor + hbits;
  for (uint32_t base = 0; base < blk.nfrag; base += 64) {
    const uint32_t j = base + lane;
    const uint32_t fb = j < blk.nfrag ? a.f_bits[blk.first_frag + j] : 0u;
    // fragment sums fit in 32 bits: 64 fragments * 32 KiB * 15 bits < 2^32
    const uint32_t incl = zh_wave_scan(fb);
    if (j < blk.nfrag) {
      const uint64_t fs = abs0 + fcur + (incl - fb);
      a.f_bit_start[blk.first_frag + j] = fs;
      zero_inner(fs);  // the fragment's first word = the last word of what lies before it (zh_emit.hip ORs into both)
    }
    fcur += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }
  if (lane == 0) {  // the end-of-block code: behind the last fragment, 15 bits at most
    zero_inner(abs0 + fcur);
    zero_inner(abs0 + fcur + (eob >> 16) - 1);
  }
  // (this wave's zeroes before this wave's ORs: stores and atomics of one wave are not ordered by themselves)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  zh_wave_sync();
  for (uint32_t w = lane; w * 32 < hbits; w += 64) {
    const uint32_t nb = hbits - w * 32 < 32 ? hbits - w * 32 : 32;
    or_bits(out, body_bit0 + cursor + (uint64_t)w * 32, hdr[w], nb);
  }
  if (lane == 0) or_bits(out, body_bit0 + fcur, eob & 0xffffu, eob >> 16);  // deflate.nim:471
}

// zh_debug_huffman: one code from a histogram, by either builder (the tests hold the exact one against the oracle's
// huffmanCodes symbol for symbol, the fast one against the optimum's cost and the prefix-code conditions)
__global__ __launch_bounds__(64) void zh_huffman_probe_kernel(const uint32_t* __restrict__ freq, int num_freq, int min_codes,
                                                              int limit, int contract, uint16_t* __restrict__ codes,
                                                              uint8_t* __restrict__ lens, int* __restrict__ n_out) {
  __shared__ uint32_t s_freq[ZH_HIST_STRIDE];
  __shared__ HuffWork s_work;
  __shared__ uint16_t s_codes[ZH_HIST_STRIDE];
  __shared__ uint8_t s_lens[ZH_HIST_STRIDE];
  __shared__ uint32_t s_num[32];
  __shared__ int s_n;
  const unsigned lane = zh_lane();
  for (int i = (int)lane; i < (int)ZH_HIST_STRIDE; i += 64) {
    s_freq[i] = i < num_freq ? freq[i] : 0u;
    s_codes[i] = 0;
    s_lens[i] = 0;
  }
  zh_wave_sync();
  if (contract) {
    const int n = huffman_codes_fast(s_freq, num_freq, min_codes, limit, s_codes, s_lens, *reinterpret_cast<FastWork*>(&s_work), s_num);
    if (lane == 0) s_n = n;
  } else {
    const int n = huffman_codes(s_freq, num_freq, min_codes, limit, s_codes, s_lens, s_work);
    if (lane == 0) s_n = n;
  }
  zh_wave_sync();
  const int n = s_n;
  for (int i = (int)lane; i < n; i += 64) {
    codes[i] = s_codes[i];
    lens[i] = s_lens[i];
  }
  if (lane == 0) *n_out = n;
}
extern "C" void zh_launch_huffman_probe(hipStream_t stream, const uint32_t* freq, int num_freq, int min_codes, int limit,
                                        int contract, uint16_t* codes, uint8_t* lens, int* n_out) {
  hipLaunchKernelGGL(zh_huffman_probe_kernel, dim3(1), dim3(64), 0, stream, freq, num_freq, min_codes, limit, contract, codes,
                     lens, n_out);
}
extern "C" void zh_launch_huffman(hipStream_t stream, ZhCompressArgs a, int contract) {
  if (!a.nblocks) return;
  if (contract)
    hipLaunchKernelGGL(zh_huffman_kernel<true>, dim3(a.nblocks), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL(zh_huffman_kernel<false>, dim3(a.nblocks), dim3(64), 0, stream, a);
}
extern "C" void zh_launch_trailer(hipStream_t stream, uint8_t* d_dst, ZhCompressArgs a, const uint32_t* buf_crc,
                                  const uint32_t* buf_adler) {
  if (!a.nbufs || a.data_format == ZH_DF_DEFLATE) return;
  hipLaunchKernelGGL(zh_trailer_kernel, dim3(a.nbufs), dim3(64), 0, stream, d_dst, a, buf_crc, buf_adler);
}
extern "C" void zh_launch_layout(hipStream_t stream, uint8_t* d_dst, ZhCompressArgs a,
                                 const uint32_t* buf_crc, const uint32_t* buf_adler, int with_trailer) {
  if (!a.nbufs) return;
  hipLaunchKernelGGL(zh_layout_kernel, dim3(a.nbufs), dim3(64), 0, stream, d_dst, a, buf_crc,
                     buf_adler, with_trailer);
  hipLaunchKernelGGL(zh_block_layout_kernel, dim3(a.nblocks), dim3(64), 0, stream, d_dst, a);
}
