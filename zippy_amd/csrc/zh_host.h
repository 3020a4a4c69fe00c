// Host side of the C ABI, shared declarations (not installed: include/zippy_hip.h is the public header).
// The host side lives in five files -- zh_context.hip (contexts, the device block cache, bounds),
// zh_plan_compress.hip / zh_plan_uncompress.hip (device-resident plans: descriptors and scratch),
// zh_plan_run.hip (kernel sequencing, switches, results), zh_host_batch.hip (host-buffer batches: staging,
// pipelined groups, sharding over contexts) and zh_host_calls.hip (single-buffer calls, the block-parallel
// form, checksums, debug hooks).  No compute happens in any of them.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "zh_common.h"
#include "zh_kprof.h"
#include "zh_tables.h"

#define ZH_INTERNAL __attribute__((visibility("hidden")))

// ---- kernel launchers (defined next to their kernels) ----
extern "C" {
const void* zh_checksum_tables(int device);
void zh_launch_checksum_pieces(hipStream_t, const void* tabs, const uint8_t* d_data,
                               const ZhPieceDesc* pieces, uint32_t npieces, const uint64_t* dyn_len,
                               int want_crc, int want_adler, uint32_t* out_crc, uint32_t* out_adler,
                               uint32_t* out_len);
void zh_launch_checksum_combine(hipStream_t, const void* tabs, const ZhBufDesc* bufs, uint32_t nbufs,
                                const uint32_t* piece_crc, const uint32_t* piece_adler,
                                const uint32_t* piece_len, int want_crc, int want_adler,
                                uint32_t* buf_crc, uint32_t* buf_adler);
void zh_launch_unwrap(hipStream_t, const uint8_t* d_src, ZhInflateArgs a);
void zh_launch_inflate(hipStream_t, const uint8_t* d_src, uint8_t* d_dst, ZhInflateArgs a);
void zh_launch_verify(hipStream_t, ZhInflateArgs a, const uint32_t* buf_crc, const uint32_t* buf_adler);
void zh_launch_inflate_tokens(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, uint32_t* tok_pool,
                              const uint64_t* tok_off, const uint64_t* tok_cap);
void zh_launch_inflate_write(hipStream_t, const uint8_t* d_src, uint8_t* d_dst, ZhInflateArgs a,
                             const uint32_t* tok_pool, const uint64_t* tok_off);
void zh_launch_inflate_count(hipStream_t, const uint8_t* d_src, ZhInflateArgs a);
void zh_launch_segments_reduce(hipStream_t, ZhInflateArgs seg, ZhInflateArgs whole);
void zh_launch_seg_find(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_fake_start(hipStream_t, ZhSegArgs g, uint64_t bit);
void zh_launch_seg_check(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_tokens(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, uint32_t* tok_pool, ZhSegArgs g, int phase);
void zh_launch_seg_decide(hipStream_t, ZhInflateArgs a, ZhSegArgs g, int rerun);
void zh_launch_seg_chain(hipStream_t, ZhInflateArgs a, ZhSegArgs g, int rerun);
void zh_launch_seg_repair(hipStream_t, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_stats(hipStream_t, ZhSegArgs g, uint64_t* stats);
void zh_launch_seg_write(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, const uint32_t* tok_pool, ZhSegArgs g);
void zh_launch_seg_windows(hipStream_t, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_finish(hipStream_t, uint8_t* d_dst, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_l1_match(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, int huffman_only,
                        uint16_t* table_pool, uint32_t* next_frag, uint32_t* cost, uint32_t* order, uint32_t* hist,
                        int use_order);
uint32_t zh_l1_table_slots(void);
void zh_launch_l1p_match(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, uint16_t* link_pool,
                         uint32_t* next_frag);
uint32_t zh_l1p_slots(void);
void zh_launch_chain_prev(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, uint32_t* head_scratch,
                          uint64_t* prevw, uint32_t* lists, int links_serial, int good);
uint32_t zh_chain_prev_slice(void);
int zh_chain_lds_order_ok(int device, hipStream_t stream);
void zh_launch_chain_search(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, int good, int nice,
                            int max_chain, const uint64_t* prevw, uint32_t* best, int links_serial);
void zh_launch_chain_select(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, int good, int nice,
                            int max_chain, const uint64_t* prevw, uint32_t* best, int links_serial);
void zh_launch_frag_stats(hipStream_t, const uint8_t* d_src, ZhCompressArgs a);
void zh_launch_huffman(hipStream_t, ZhCompressArgs a, int contract);
void zh_launch_huffman_probe(hipStream_t, const uint32_t* freq, int num_freq, int min_codes, int limit, int contract,
                             uint16_t* codes, uint8_t* lens, int* n_out);
void zh_launch_layout(hipStream_t, uint8_t* d_dst, ZhCompressArgs a, const uint32_t* buf_crc,
                      const uint32_t* buf_adler, int with_trailer);
void zh_launch_trailer(hipStream_t, uint8_t* d_dst, ZhCompressArgs a, const uint32_t* buf_crc, const uint32_t* buf_adler);
void zh_launch_emit(hipStream_t, const uint8_t* d_src, uint8_t* d_dst, ZhCompressArgs a, int cover_in);
}

// internal.nim:177-189 configurationTable (good, nice, chain); `lazy` is unused by the reference
static const int kChainConfig[10][3] = {{0, 0, 0},     {4, 8, 4},      {4, 16, 8},    {4, 32, 32},
                                        {4, 16, 16},   {8, 32, 32},    {8, 128, 128}, {8, 256, 256},
                                        {32, 258, 1024}, {32, 258, 4096}};

struct zh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int fname_len = -1;
  int inflate_mode = -1;  // -1: ZH_INFLATE or the default (split), 0 split, 1 serial
  int l1_parse = -1;      // -1: ZH_L1_PARSE or the default (exact), 0 exact (the reference's parse), 1 parallel
  // chain levels: THIS device failed zh_create's probe of the LDS atomic lane order (zh_chain_match.hip): its links
  // come from the in-order kernels
  bool chain_links_serial = false;
  std::string last_error;
  uint64_t* d_seg_stats = nullptr;  // zh_debug_segment_stats: two device counters (streams cut into segments / whose chain held)
  const void* cktabs = nullptr;
  std::mt19937 rng{std::random_device{}()};
  // host-buffer calls: two pinned staging chunks between the caller's pageable memory and HBM
  // (allocated on the first such call)
  uint8_t* pin[2] = {nullptr, nullptr};
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool pin_busy[2] = {false, false};
  hipStream_t copy_stream = nullptr;  // transfers of a pipelined batch, next to `stream`'s kernels
  // a compress run's checksum kernels beside its code builder (zh_plan_run.hip): forked off `stream` behind the match
  // finder, joined in front of the layout; the events are the context's (runs of one context follow each other)
  hipStream_t aux_stream = nullptr;
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  // zh_*_batch_into: the caller's output buffers and their sizes for the call in progress
  // (into_base: the dsts array the batch functions were handed, to find a buffer's index again)
  void* const* into_ptrs = nullptr;
  const size_t* into_caps = nullptr;
  void** into_base = nullptr;
  uint64_t pipe_min = 0, pipe_group = 0;  // zh_set_host_pipeline (0: ZH_PIPE_MIN / ZH_PIPE_GROUP / default)
  // device memory the context has freed, kept for its next call (ctx_malloc / ctx_free)
  struct DevBlock {
    void* p;
    size_t size;
    bool used;
    uint64_t stamp;
  };
  std::vector<DevBlock> dev_blocks;
  size_t dev_cached = 0, dev_cache_max = 0;
  uint64_t dev_stamp = 0;
  bool dev_poison = false;
};

// (zh_context.hip)
ZH_INTERNAL hipError_t ctx_malloc(zh_ctx* ctx, void** out, size_t bytes);
ZH_INTERNAL void ctx_free(zh_ctx* ctx, void* p);

#define ZH_HIP(ctx, call)                                                            \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);         \
      return ZH_ERR_DEVICE;                                                          \
    }                                                                                \
  } while (0)

static inline size_t container_overhead(int fmt) {
  return fmt == ZH_DF_GZIP ? 10 + 26 + 8 : fmt == ZH_DF_ZLIB ? 6 : 0;
}
static inline size_t typical_cap(size_t len, int fmt) {
  size_t nblocks = (len + ZH_BLOCK_SIZE - 1) / ZH_BLOCK_SIZE;
  if (!nblocks) nblocks = 1;
  return len + len / 8 + 1024 * nblocks + 5 * (len / ZH_STORED_MAX + 1) + container_overhead(fmt) + 64;
}

// ---------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------
struct Arena {
  size_t size = 0;
  uint8_t* base = nullptr;
  size_t reserve(size_t bytes) {
    size_t off = (size + 255) & ~(size_t)255;
    size = off + bytes;
    return off;
  }
};

// Device-resident plans keep their per-position scratch -- the chain levels' links and best matches (12 bytes a
// position), the split decoder's token records (4 bytes an output byte) -- for at most this many bytes at a time;
// a batch that needs more runs those kernels over ranges of its blocks / streams, one range after the other
// through the same scratch (same bytes out).  ZH_SCRATCH_MB (default 32768: BASELINE's 4096 x 1 MiB batch decodes
// as one group -- its records take 16.8 GiB, and two launches of 2048 streams measured 12 % slower than one of
// 4096 --; the tests force it low).
static inline uint64_t scratch_budget() {
  const char* e = getenv("ZH_SCRATCH_MB");
  const uint64_t mb = e ? strtoull(e, nullptr, 10) : 32768ull;
  return (mb ? mb : 1ull) << 20;
}

struct ZhPlanRange {
  uint32_t b0, nb, f0, nf;
};
struct zh_plan {
  zh_ctx* ctx = nullptr;
  bool is_compress = true;
  size_t n = 0;
  int level = 0, fmt = 0;
  int count_only = 0;
  uint8_t* arena = nullptr;
  ZhCompressArgs ca{};
  ZhInflateArgs ia{};
  ZhBufDesc* d_bufs = nullptr;
  ZhPieceDesc* d_pieces = nullptr;
  uint32_t npieces = 0;
  uint32_t *piece_crc = nullptr, *piece_adler = nullptr, *piece_len = nullptr;
  uint32_t *buf_crc = nullptr, *buf_adler = nullptr;
  uint32_t* head_scratch = nullptr;  // chain levels: `head` per block, previous-position links and
  size_t head_bytes = 0;             // best match per position (zh_chain_match.hip)
  uint16_t* l1_tables = nullptr;  // BestSpeed: pool of per-wave hash tables (zh_l1_match.hip)
  void* l1_pool_own = nullptr;    // ... when it is an allocation of its own (ZH_L1_POOL: uncached / fine-grained memory)
  uint32_t* l1_counter = nullptr; // ... and the counter its waves draw fragments from
  // longest first (zh_l1_match.hip): a fragment's cycles in the last run, the order made of them, scratch; l1_runs: runs so far
  uint32_t *l1_cost = nullptr, *l1_order = nullptr, *l1_hist = nullptr;
  uint32_t l1_runs = 0;
  uint64_t* chain_prev = nullptr;
  uint32_t* chain_best = nullptr;
  // best[] is cleared when the plan is made and handed back cleared by every run's link kernels; a run that
  // did not get as far (a launch that failed between the scatter and the links) leaves it dirty, and the next
  // run clears it before anything reads it
  bool chain_best_dirty = false;
  // chain levels: the ranges of blocks (first block, blocks, first fragment, fragments) that share the scratch in turn
  struct ChainRange {
    uint32_t b0, nb, f0, nf;
  };
  std::vector<ChainRange> chain_ranges;
  size_t chain_scratch_frags = 0;  // fragments the scratch holds (the largest range)
  // split inflate: the groups of streams (first, count) that share the token pool in turn
  std::vector<std::pair<uint32_t, uint32_t>> tok_groups;
  uint64_t dst_max_cap = 0;
  uint64_t* out_len = nullptr;
  int32_t* status = nullptr;
  const uint64_t* src_len_dev = nullptr;
  uint64_t src_max_len = 0;         // uncompress plans: the longest source slot (zh_plan_unpack's grid)
  uint64_t* unpack_lens = nullptr;  // ... and the device-side lengths zh_plan_unpack leaves (allocated by its first call)
  // block index (zh_plan_block_index): host copies of the geometry
  std::vector<ZhBufDesc> h_bufs;
  uint32_t half_piece = 0;  // uncompress plans: the first checksum piece of buffer n / 2 (the batch as two halves, zh_plan_run)
  std::vector<ZhBlockDesc> h_blocks;
  // block-parallel decode (zh_plan_uncompress_indexed): `ia` describes the one stream, `seg` its blocks
  bool force_crc = false;  // CRC-32 of the uncompressed side whatever the container (ZIP entries)
  bool indexed = false;
  // split inflate (zh_inflate_split.hip): per-stream token buffers, allocated on the first run
  uint32_t* tok_pool = nullptr;
  uint64_t tok_words = 0;
  const uint64_t *tok_off = nullptr, *tok_cap = nullptr;
  bool tok_failed = false;
  bool tok_borrowed = false;  // the pool belongs to the caller (pipelined groups share one)
  ZhInflateArgs seg{};
  uint8_t* seg_arena = nullptr;
  // large streams decoded segment-wise (zh_inflate_seg.hip); the symbol and window buffers come
  // with the token pool
  bool segmented = false;
  // test aids read when the plan is made (not on the run path): ZH_SEG_FAKE_START (a stream bit, ~0: none), ZH_TRACE_SEG
  uint64_t sg_fake_start = ~0ull;
  bool sg_trace = false;
  int sg_repair_rounds = 1;  // zh_seg_repair_kernel: twice where there is a lot to go wrong (plan_segments)
  ZhSegArgs sg{};
  uint8_t* sg_arena = nullptr;
  uint16_t* sg_sym = nullptr;
  uint8_t* sg_windows = nullptr;
  uint16_t* sg_winsym = nullptr;
  uint64_t sg_sym_count = 0;
  // profiling
  bool profiling = false;
  bool trailer_late = false;
  uint32_t halves_min = 0;    // uncompress: batches of at least this many streams go as two halves on two streams (0: never; zh_plan_run.hip)  // compress: the checksum joins behind the emission (large batches; zh_plan_run.hip)
  std::vector<const char*> k_names;
  std::vector<hipEvent_t> k_events;
  std::vector<float> k_ms;
  // kernels on the context's second stream (the checksum of a compress run): start / between / end
  hipEvent_t k_aux[3] = {nullptr, nullptr, nullptr};
  bool k_aux_used = false;
  hipEvent_t k_half[3] = {nullptr, nullptr, nullptr};  // ... of the second half's tokens kernel and writer (uncompress plans)
  bool k_half_used = false;
};

template <class T>
static inline T* carve(uint8_t* base, size_t off) {
  return reinterpret_cast<T*>(base + off);
}

static inline bool valid_block_bytes(size_t bb) {
  return bb >= ZH_FRAG_SIZE && bb <= ZH_BLOCK_SIZE && bb % ZH_FRAG_SIZE == 0;
}

// (zh_plan_run.hip)
ZH_INTERNAL bool inflate_split_enabled(const zh_ctx* ctx);
ZH_INTERNAL bool l1_parallel(const zh_ctx* ctx);
ZH_INTERNAL bool plan_token_pool(zh_plan* p);
ZH_INTERNAL void plan_lend_token_pool(zh_plan* p, uint32_t* pool, uint64_t words);
static inline void plan_set_count_only(zh_plan* plan, int on) { plan->ia.count_only = on; }

// ---------------------------------------------------------------------------
// host-buffer API: device buffers of a call, plans that go away with their scope, phase timing
// ---------------------------------------------------------------------------
struct DevBuf {
  uint8_t* p = nullptr;
  zh_ctx* ctx = nullptr;
  ~DevBuf() {
    if (p) ctx_free(ctx, p);
  }
};
static inline hipError_t dev_alloc(zh_ctx* ctx, DevBuf& b, size_t bytes) {
  b.ctx = ctx;
  return ctx_malloc(ctx, (void**)&b.p, bytes);
}
struct PlanGuard {
  zh_plan* p = nullptr;
  ~PlanGuard() { zh_plan_destroy(p); }
};

// ZH_TRACE=1: wall-clock of the host-buffer calls' phases on stderr (tuning aid; syncs the stream)
struct Trace {
  bool on = getenv("ZH_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(zh_ctx* ctx, const char* what) {
    if (!on) return;
    (void)hipStreamSynchronize(ctx->stream);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[zh] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};


// (zh_host_batch.hip) pack host buffers into one device allocation / results into fresh host buffers
ZH_INTERNAL int zhh_upload(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n, DevBuf& dev,
                           std::vector<uint64_t>& off, std::vector<uint64_t>& len64);
ZH_INTERNAL int zhh_download(zh_ctx* ctx, const uint8_t* d_dst, size_t n, const std::vector<uint64_t>& doff,
                             const std::vector<uint64_t>& olen, const std::vector<char>& take, void** dsts,
                             size_t* dst_lens, int32_t* statuses);
