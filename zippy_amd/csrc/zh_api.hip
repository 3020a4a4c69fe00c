// Host side of the C ABI (include/zippy_hip.h): contexts, plans (device
// descriptors + scratch), kernel sequencing on one HIP stream, and the
// host-buffer batch entry points.  No compute happens here.
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

// ---- kernel launchers (defined next to their kernels) ----
extern "C" {
const void* zh_checksum_tables(int device);
void zh_launch_checksum_pieces(hipStream_t, const void* tabs, const uint8_t* d_data,
                               const ZhPieceDesc* pieces, uint32_t npieces, const uint64_t* dyn_len,
                               int want_crc, int want_adler, uint32_t* out_crc, uint32_t* out_adler,
                               uint32_t* out_len);
void zh_launch_checksum_combine(hipStream_t, const ZhBufDesc* bufs, uint32_t nbufs,
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
void zh_launch_segments_reduce(hipStream_t, ZhInflateArgs seg, ZhInflateArgs whole);
void zh_launch_seg_find(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_check(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_tokens(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, uint32_t* tok_pool, ZhSegArgs g, int phase);
void zh_launch_seg_decide(hipStream_t, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_chain(hipStream_t, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_write(hipStream_t, const uint8_t* d_src, ZhInflateArgs a, const uint32_t* tok_pool, ZhSegArgs g);
void zh_launch_seg_windows(hipStream_t, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_seg_finish(hipStream_t, uint8_t* d_dst, ZhInflateArgs a, ZhSegArgs g);
void zh_launch_l1_match(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, int huffman_only,
                        uint16_t* table_pool, uint32_t* next_frag);
uint32_t zh_l1_table_slots(void);
void zh_launch_l1p_match(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, uint16_t* link_pool,
                         uint32_t* next_frag);
uint32_t zh_l1p_slots(void);
void zh_launch_chain_prev(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, uint32_t* head_scratch,
                          uint64_t* prevw, uint32_t* lists);
uint32_t zh_chain_prev_slice(void);
int zh_chain_lds_order_ok(int device, hipStream_t stream);
int zh_chain_prev_is_serial(void);
void zh_launch_chain_search(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, int good, int nice,
                            int max_chain, const uint64_t* prevw, uint32_t* best);
void zh_launch_chain_select(hipStream_t, const uint8_t* d_src, ZhCompressArgs a, int good, int nice,
                            int max_chain, const uint64_t* prevw, uint32_t* best);
void zh_launch_frag_stats(hipStream_t, const uint8_t* d_src, ZhCompressArgs a);
void zh_launch_huffman(hipStream_t, ZhCompressArgs a, int contract);
void zh_launch_huffman_probe(hipStream_t, const uint32_t* freq, int num_freq, int min_codes, int limit, int contract,
                             uint16_t* codes, uint8_t* lens, int* n_out);
void zh_launch_layout(hipStream_t, uint8_t* d_dst, ZhCompressArgs a, const uint32_t* buf_crc,
                      const uint32_t* buf_adler);
void zh_launch_emit(hipStream_t, const uint8_t* d_src, uint8_t* d_dst, ZhCompressArgs a);
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
  std::string last_error;
  const void* cktabs = nullptr;
  std::mt19937 rng{std::random_device{}()};
  // host-buffer calls: two pinned staging chunks between the caller's pageable memory and HBM
  // (allocated on the first such call)
  uint8_t* pin[2] = {nullptr, nullptr};
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool pin_busy[2] = {false, false};
  hipStream_t copy_stream = nullptr;  // transfers of a pipelined batch, next to `stream`'s kernels
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

// Device allocations of a context.  hipMalloc / hipFree cost a call of one small buffer more than
// its kernels (hipFree also waits for the device), so freed blocks are kept -- up to ZH_DEV_CACHE_MB
// (default 2048; 0: none), blocks of up to half of that -- and handed out again to requests they fit
// without wasting more than half.  Everything a context does is ordered on its stream, so a block
// may be reused as soon as it has been given back.  ZH_DEV_CACHE_POISON=1 fills every block handed
// out (test aid: nothing may rely on fresh memory being zero).
static hipError_t ctx_malloc(zh_ctx* ctx, void** out, size_t bytes) {
  const size_t want = bytes < 4096 ? 4096 : bytes;
  int best = -1;
  for (size_t i = 0; i < ctx->dev_blocks.size(); i++) {
    const zh_ctx::DevBlock& b = ctx->dev_blocks[i];
    if (b.used || b.size < want || b.size > 2 * want + (1u << 20)) continue;
    if (best < 0 || b.size < ctx->dev_blocks[best].size) best = (int)i;
  }
  hipError_t e = hipSuccess;
  if (best >= 0) {
    ctx->dev_blocks[best].used = true;
    ctx->dev_cached -= ctx->dev_blocks[best].size;
    *out = ctx->dev_blocks[best].p;
  } else {
    size_t alloc = want;
    if (want <= (1u << 20)) {
      alloc = 4096;
      while (alloc < want) alloc <<= 1;
    } else {
      alloc = (want + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    }
    e = hipMalloc(out, alloc);
    if (e != hipSuccess) {  // give the kept blocks back and try once more
      (void)hipGetLastError();
      for (size_t i = ctx->dev_blocks.size(); i-- > 0;)
        if (!ctx->dev_blocks[i].used) {
          (void)hipFree(ctx->dev_blocks[i].p);
          ctx->dev_cached -= ctx->dev_blocks[i].size;
          ctx->dev_blocks.erase(ctx->dev_blocks.begin() + i);
        }
      e = hipMalloc(out, alloc);
    }
    if (e != hipSuccess) return e;
    ctx->dev_blocks.push_back({*out, alloc, true, 0});
  }
  if (ctx->dev_poison) (void)hipMemsetAsync(*out, 0xa5, bytes, ctx->stream);
  return hipSuccess;
}
static void ctx_free(zh_ctx* ctx, void* p) {
  if (!p) return;
  // A block given back may be handed out again at once: safe for what is ordered on ctx->stream,
  // not for transfers still queued on the copy stream of the pipelined host calls -- wait for those.
  if (ctx->copy_stream && hipStreamQuery(ctx->copy_stream) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(ctx->copy_stream);
  }
  for (size_t i = 0; i < ctx->dev_blocks.size(); i++) {
    zh_ctx::DevBlock& b = ctx->dev_blocks[i];
    if (b.p != p) continue;
    if (b.size > ctx->dev_cache_max / 2) {
      (void)hipFree(p);
      ctx->dev_blocks.erase(ctx->dev_blocks.begin() + i);
      return;
    }
    b.used = false;
    b.stamp = ++ctx->dev_stamp;
    ctx->dev_cached += b.size;
    while (ctx->dev_cached > ctx->dev_cache_max) {  // the longest unused goes first
      int old = -1;
      for (size_t k = 0; k < ctx->dev_blocks.size(); k++)
        if (!ctx->dev_blocks[k].used && (old < 0 || ctx->dev_blocks[k].stamp < ctx->dev_blocks[old].stamp)) old = (int)k;
      if (old < 0) break;
      (void)hipFree(ctx->dev_blocks[old].p);
      ctx->dev_cached -= ctx->dev_blocks[old].size;
      ctx->dev_blocks.erase(ctx->dev_blocks.begin() + old);
    }
    return;
  }
  (void)hipFree(p);  // (not ours)
}

#define ZH_HIP(ctx, call)                                                            \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);         \
      return ZH_ERR_DEVICE;                                                          \
    }                                                                                \
  } while (0)

extern "C" const char* zh_strerror(int status) {
  switch (status) {
    case ZH_OK: return "ok";
    case ZH_ERR_INVALID_LEVEL: return "Invalid compression level";
    case ZH_ERR_INVALID_FORMAT: return "Invalid data format";
    case ZH_ERR_DETECT: return "Unable to detect compressed data format";
    case ZH_ERR_UNSUPPORTED_METHOD: return "Unsupported compression method";
    case ZH_ERR_COMPRESSION_INFO: return "Invalid compression info";
    case ZH_ERR_INVALID_HEADER: return "Invalid header";
    case ZH_ERR_PRESET_DICT: return "Preset dictionary is not yet supported";
    case ZH_ERR_CHECKSUM: return "Checksum verification failed";
    case ZH_ERR_SIZE: return "Size verification failed";
    case ZH_ERR_GZIP_ID: return "Failed gzip identification values check";
    case ZH_ERR_RESERVED_FLAGS: return "Reserved flag bits set";
    case ZH_ERR_UNSUPPORTED_FLAGS: return "Currently unsupported flags are set";
    case ZH_ERR_INVALID_BUFFER: return "Invalid buffer, unable to uncompress";
    case ZH_ERR_COMPRESS_INTERNAL: return "Unexpected error while compressing";
    case ZH_ERR_END_OF_BUFFER: return "Cannot read further, at end of buffer";
    case ZH_ERR_BYTE_BOUNDARY: return "Must be at a byte boundary";
    case ZH_ERR_BLOCK_HEADER: return "Invalid block header";
    case ZH_ERR_INVALID_SYMBOL: return "Invalid symbol";
    case ZH_ERR_NOMEM: return "Out of memory";
    case ZH_ERR_DEVICE: return "GPU/HIP error";
    case ZH_ERR_DST_TOO_SMALL: return "Output slot too small";
    case ZH_ERR_ARGUMENT: return "Invalid argument";
    case ZH_ERR_ARCHIVE_EOF: return "Unexpected EOF, invalid archive?";
    case ZH_ERR_ZIP_FILE_HEADER: return "Invalid file header";
    case ZH_ERR_ZIP_METHOD: return "Unsupported archive, compression method";
    case ZH_ERR_ZIP_NO_RECORD: return "No file record found";
    case ZH_ERR_ZIP_CRC: return "Verifying crc32 failed";
    case ZH_ERR_ZIP_UNSUPPORTED: return "Unsupported archive, disk or record number";
    case ZH_ERR_ZIP_CENTRAL_HEADER: return "Invalid central directory file header";
    case ZH_ERR_ZIP_DISK_NUMBER: return "Invalid file disk number";
    case ZH_ERR_ZIP_DUPLICATE: return "Unsupported archive, duplicate entry";
    case ZH_ERR_ZIP_CENTRAL_SIZE: return "Invalid central directory size";
    case ZH_ERR_ZIP_NAME: return "Invalid file name (empty, absolute or longer than uint16.high)";
    case ZH_ERR_TAR_HEADER_TYPE: return "Unsupported header type";
    case ZH_ERR_UNSAFE_PATH: return "Path not allowed (absolute or containing ../)";
    case ZH_ERR_TAR_NUMBER: return "Invalid octal number in tar header";
    default: return "Unknown status";
  }
}

extern "C" int zh_create(int device, void* stream, zh_ctx** out) {
  if (!out) return ZH_ERR_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZH_ERR_DEVICE;
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) return ZH_ERR_DEVICE;
  }
  if (device >= count) return ZH_ERR_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return ZH_ERR_DEVICE;
  zh_ctx* c = new zh_ctx;
  c->device = device;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreate(&c->stream) != hipSuccess) {
      delete c;
      return ZH_ERR_DEVICE;
    }
    c->own_stream = true;
  }
  {
    const char* e = getenv("ZH_DEV_CACHE_MB");
    c->dev_cache_max = (size_t)(e ? strtoull(e, nullptr, 10) : 2048) << 20;
    c->dev_poison = getenv("ZH_DEV_CACHE_POISON") != nullptr;
  }
  c->cktabs = zh_checksum_tables(device);
  if (!c->cktabs) {
    delete c;
    return ZH_ERR_DEVICE;
  }
  // the chain levels' parallel link kernels rest on the order in which the LDS unit serves the lanes of an
  // atomic: the device is asked once (zh_chain_match.hip); one that answers otherwise runs the in-order kernels
  if (!zh_chain_lds_order_ok(device, c->stream))
    c->last_error = "this device does not serve the lanes of an LDS atomic in ascending order (or could not be asked): "
                    "levels -1, 2..9 build their chain links with the in-order kernels (ZH_CHAIN_PREV=serial)";
  *out = c;
  return ZH_OK;
}
extern "C" int zh_chain_links_parallel(zh_ctx* ctx) {
  if (!ctx || hipSetDevice(ctx->device) != hipSuccess) return 0;
  return zh_chain_prev_is_serial() ? 0 : 1;
}

extern "C" void zh_destroy(zh_ctx* ctx) {
  if (!ctx) return;
  for (int k = 0; k < 2; k++) {
    if (ctx->pin_ev[k]) (void)hipEventDestroy(ctx->pin_ev[k]);
    if (ctx->pin[k]) (void)hipHostFree(ctx->pin[k]);
  }
  for (auto& b : ctx->dev_blocks) (void)hipFree(b.p);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}
extern "C" const char* zh_last_error(zh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
extern "C" void* zh_stream(zh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" void zh_set_gzip_fname_len(zh_ctx* ctx, int k) {
  if (ctx) ctx->fname_len = k > 25 ? 25 : k;
}
extern "C" void zh_free(void* p) { free(p); }
extern "C" void zh_set_host_pipeline(zh_ctx* ctx, size_t min_batch_bytes, size_t group_bytes) {
  if (!ctx) return;
  ctx->pipe_min = min_batch_bytes;
  ctx->pipe_group = group_bytes;
}

static size_t container_overhead(int fmt) {
  return fmt == ZH_DF_GZIP ? 10 + 26 + 8 : fmt == ZH_DF_ZLIB ? 6 : 0;
}
// Worst case of the reference's encoder: it has no "stored if larger" fallback, so
// a block that escapes the 98 % literal test can still use up to 15 bits per
// literal; every block adds a <= 1 KiB header.
extern "C" size_t zh_compress_bound(size_t len, int data_format) {
  size_t nblocks = (len + ZH_BLOCK_SIZE - 1) / ZH_BLOCK_SIZE;
  if (!nblocks) nblocks = 1;
  return len * 2 + 1024 * nblocks + 5 * (len / ZH_STORED_MAX + 1) + container_overhead(data_format) + 64;
}
static size_t typical_cap(size_t len, int fmt) {
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
// through the same scratch (same bytes out).  ZH_SCRATCH_MB (default 16384; the tests force it low).
static uint64_t scratch_budget() {
  const char* e = getenv("ZH_SCRATCH_MB");
  const uint64_t mb = e ? strtoull(e, nullptr, 10) : 16384ull;
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
  uint32_t* l1_counter = nullptr; // ... and the counter its waves draw fragments from
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
  uint64_t dst_lo = 0, dst_hi = 0;  // byte range of d_dst covered by the slots
  bool dst_dense = true;            // the slots tile [dst_lo, dst_hi) without gaps
  uint64_t dst_max_cap = 0;
  uint64_t* out_len = nullptr;
  int32_t* status = nullptr;
  const uint64_t* src_len_dev = nullptr;
  // block index (zh_plan_block_index): host copies of the geometry
  std::vector<ZhBufDesc> h_bufs;
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
  ZhSegArgs sg{};
  uint8_t* sg_arena = nullptr;
  uint16_t* sg_sym = nullptr;
  uint8_t* sg_windows = nullptr;
  uint16_t* sg_winsym = nullptr;
  uint64_t sg_sym_count = 0;
  // profiling
  bool profiling = false;
  std::vector<const char*> k_names;
  std::vector<hipEvent_t> k_events;
  std::vector<float> k_ms;
};

static void prof_mark(zh_plan* p, const char* name) {
  if (!p->profiling) return;
  size_t i = p->k_names.size();
  if (p->k_events.size() <= i) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {  // no timing rather than a failed run
      p->profiling = false;
      return;
    }
    p->k_events.push_back(e);
  }
  if (hipEventRecord(p->k_events[i], p->ctx->stream) != hipSuccess) {
    p->profiling = false;
    return;
  }
  p->k_names.push_back(name);
}

extern "C" void zh_plan_set_profiling(zh_plan* plan, int on) {
  if (plan) plan->profiling = on != 0;
}

extern "C" int zh_plan_kernel_times(zh_plan* p, const char** names, float* ms, int max_entries) {
  if (!p || !p->profiling || p->k_names.size() < 2) return 0;
  if (hipStreamSynchronize(p->ctx->stream) != hipSuccess) return 0;
  int cnt = 0;
  for (size_t i = 0; i + 1 < p->k_names.size() && cnt < max_entries; i++) {
    float t = 0;
    if (hipEventElapsedTime(&t, p->k_events[i], p->k_events[i + 1]) != hipSuccess) break;
    names[cnt] = p->k_names[i];  // the marker recorded BEFORE a launch carries its name
    ms[cnt] = t;
    cnt++;
  }
  return cnt;
}

extern "C" void zh_plan_destroy(zh_plan* p) {
  if (!p) return;
  // (nothing useful can be done about a failure while tearing down)
  (void)hipStreamSynchronize(p->ctx->stream);
  if (p->arena) ctx_free(p->ctx, p->arena);
  if (p->seg_arena) ctx_free(p->ctx, p->seg_arena);
  if (p->tok_pool && !p->tok_borrowed) ctx_free(p->ctx, p->tok_pool);
  if (p->sg_arena) ctx_free(p->ctx, p->sg_arena);
  if (p->sg_sym) ctx_free(p->ctx, p->sg_sym);
  if (p->sg_windows) ctx_free(p->ctx, p->sg_windows);
  if (p->sg_winsym) ctx_free(p->ctx, p->sg_winsym);
  for (auto e : p->k_events) (void)hipEventDestroy(e);
  delete p;
}

template <class T>
static T* carve(uint8_t* base, size_t off) {
  return reinterpret_cast<T*>(base + off);
}

static bool valid_block_bytes(size_t bb) {
  return bb >= ZH_FRAG_SIZE && bb <= ZH_BLOCK_SIZE && bb % ZH_FRAG_SIZE == 0;
}

extern "C" int zh_plan_compress_blocks(zh_ctx* ctx, size_t n, const uint64_t* src_off,
                                       const uint64_t* src_len, const uint64_t* dst_off,
                                       const uint64_t* dst_cap, int level, int data_format,
                                       size_t block_bytes, zh_plan** out) {
  if (!ctx || !out || (n && (!src_off || !src_len || !dst_off || !dst_cap))) return ZH_ERR_ARGUMENT;
  *out = nullptr;
  // whole fragments per block keep lz77.nim:78's block-relative window position equal to
  // lz77.nim:123's absolute one
  if (!valid_block_bytes(block_bytes)) return ZH_ERR_ARGUMENT;
  if (level < -2 || level > 9) return ZH_ERR_INVALID_LEVEL;  // deflate.nim:208-209
  if (data_format != ZH_DF_GZIP && data_format != ZH_DF_ZLIB && data_format != ZH_DF_DEFLATE)
    return ZH_ERR_INVALID_FORMAT;  // zippy.nim:83-84
  ZH_HIP(ctx, hipSetDevice(ctx->device));

  std::vector<ZhBufDesc> bufs(n);
  std::vector<ZhBlockDesc> blocks;
  std::vector<ZhFragDesc> frags;
  std::vector<ZhPieceDesc> pieces;
  uint64_t lo = ~0ull, hi = 0, cap_sum = 0, cap_max = 0;
  for (size_t i = 0; i < n; i++) {
    ZhBufDesc& b = bufs[i];
    b.src_off = src_off[i];
    b.src_len = src_len[i];
    b.dst_off = dst_off[i];
    b.dst_cap = dst_cap[i];
    b.first_block = (uint32_t)blocks.size();
    b.first_piece = (uint32_t)frags.size();
    int k = ctx->fname_len;
    if (k < 0) k = (int)(ctx->rng() % 26);  // zippy.nim:28-38
    b.fname_len = (uint32_t)k;
    b.pad = 0;
    // deflate.nim:228: blocks of <= 4 MiB; level 0 uses one run of stored chunks over the
    // whole buffer (deflate.nim:214-226)
    const uint64_t bsize = level == 0 ? (b.src_len ? b.src_len : 1) : block_bytes;
    uint64_t nb = (b.src_len + bsize - 1) / bsize;
    if (nb < 1) nb = 1;
    for (uint64_t j = 0; j < nb; j++) {
      ZhBlockDesc blk;
      const uint64_t bstart = j * bsize;
      blk.src_off = b.src_off + bstart;
      blk.len = std::min<uint64_t>(b.src_len - bstart, bsize);
      blk.buf = (uint32_t)i;
      blk.first_frag = (uint32_t)frags.size();
      blk.is_final = j == nb - 1;
      for (uint64_t o = 0; o < blk.len; o += ZH_FRAG_SIZE) {
        ZhFragDesc f;
        f.src_off = blk.src_off + o;
        f.len = (uint32_t)std::min<uint64_t>(blk.len - o, ZH_FRAG_SIZE);
        f.block = (uint32_t)blocks.size();
        frags.push_back(f);
        pieces.push_back(ZhPieceDesc{f.src_off, f.len, (uint32_t)i, bstart + o});
      }
      blk.nfrag = (uint32_t)frags.size() - blk.first_frag;
      blocks.push_back(blk);
    }
    b.nblocks = (uint32_t)blocks.size() - b.first_block;
    b.npieces = (uint32_t)frags.size() - b.first_piece;
    lo = std::min(lo, b.dst_off);
    hi = std::max(hi, b.dst_off + b.dst_cap);
    cap_sum += b.dst_cap;
    cap_max = std::max(cap_max, b.dst_cap);
  }
  if (blocks.size() >= 0xffffffffull || frags.size() >= 0xffffffffull) return ZH_ERR_ARGUMENT;

  zh_plan* p = new zh_plan;
  p->ctx = ctx;
  p->is_compress = true;
  p->n = n;
  p->level = level;
  p->fmt = data_format;
  p->dst_lo = n ? lo : 0;
  p->dst_hi = n ? hi : 0;
  p->dst_dense = !n || cap_sum >= hi - lo;  // (overlapping slots are the caller's error either way)
  p->dst_max_cap = cap_max;
  const size_t nf = frags.size(), nb = blocks.size();
  const bool chain = level == -1 || level >= 2;
  const bool need_matches = level != 0;

  Arena ar;
  const size_t o_bufs = ar.reserve(n * sizeof(ZhBufDesc));
  const size_t o_blocks = ar.reserve(nb * sizeof(ZhBlockDesc));
  const size_t o_frags = ar.reserve(nf * sizeof(ZhFragDesc));
  const size_t o_pieces = ar.reserve(nf * sizeof(ZhPieceDesc));
  const size_t mslots = need_matches ? nf * ZH_MAX_MATCHES_PER_FRAG : 0;
  const size_t o_mpos = ar.reserve(mslots * 2), o_mlen = ar.reserve(mslots * 2), o_moff = ar.reserve(mslots * 2);
  const size_t o_fnm = ar.reserve(nf * 4), o_fsp = ar.reserve(nf * 4), o_fnl = ar.reserve(nf * 4),
               o_fex = ar.reserve(nf * 4), o_fhist = ar.reserve(nf * ZH_HIST_STRIDE * 2),
               o_fbits = ar.reserve(nf * 4), o_fstart = ar.reserve(nf * 8);
  const size_t o_pcrc = ar.reserve(nf * 4), o_pad = ar.reserve(nf * 4), o_plen = ar.reserve(nf * 4);
  const size_t o_bmode = ar.reserve(nb * 4), o_blit = ar.reserve(nb * 288 * 4),
               o_bdist = ar.reserve(nb * 32 * 4), o_bhdr = ar.reserve(nb * ZH_HDR_WORDS * 4),
               o_bhb = ar.reserve(nb * 4), o_bbits = ar.reserve(nb * 8), o_bd0 = ar.reserve(nb * 8),
               o_bst = ar.reserve((nb + n) * 8);
  const size_t o_bcrc = ar.reserve(n * 4), o_bad = ar.reserve(n * 4), o_olen = ar.reserve(n * 8),
               o_st = ar.reserve(n * 4);
  // chain levels: ranges of whole blocks whose scratch (12 bytes a position + the links' tables) fits the budget
  size_t range_blocks = 0, range_frags = 0;
  if (chain && nb) {
    const uint64_t per_frag = (uint64_t)ZH_FRAG_SIZE * 12u, per_block = (uint64_t)ZH_CHAIN_HEAD_WORDS * 4u;
    const uint64_t budget = scratch_budget();
    ZhPlanRange cur{0, 0, 0, 0};
    uint64_t cur_bytes = 0;
    for (size_t b = 0; b < nb; b++) {
      const uint64_t need = blocks[b].nfrag * per_frag + per_block;
      if (cur.nb && cur_bytes + need > budget) {
        p->chain_ranges.push_back({cur.b0, cur.nb, cur.f0, cur.nf});
        cur = ZhPlanRange{(uint32_t)b, 0, blocks[b].first_frag, 0};
        cur_bytes = 0;
      }
      cur.nb++;
      cur.nf += blocks[b].nfrag;
      cur_bytes += need;
    }
    p->chain_ranges.push_back({cur.b0, cur.nb, cur.f0, cur.nf});
    for (const auto& r : p->chain_ranges) {
      range_blocks = std::max<size_t>(range_blocks, r.nb);
      range_frags = std::max<size_t>(range_frags, r.nf);
    }
  }
  p->chain_scratch_frags = range_frags;
  if (p->chain_ranges.size() > 1 && getenv("ZH_TRACE"))
    fprintf(stderr, "zippy_hip: chain scratch for %zu of %zu fragments: %zu ranges of blocks\n", range_frags, nf, p->chain_ranges.size());
  p->head_bytes = !chain ? 0
                  : range_blocks <= zh_chain_prev_slice() ? range_blocks * ((size_t)ZH_CHAIN_HEAD_WORDS * 4)
                                                          : range_blocks * ((size_t)2 << 17);  // (zh_launch_chain_prev)
  const size_t o_head = ar.reserve(p->head_bytes);
  // one 32 KiB hash table per persistent matcher wave (zh_launch_l1_match: min(fragments, slots) waves)
  // (the parallel parse, zh_launch_l1p_match, keeps 128 KiB of table results per workgroup there instead)
  const size_t o_l1tab = ar.reserve(level == 1 ? std::max(std::min<size_t>(nf, zh_l1_table_slots()) * 32768,
                                                          std::min<size_t>(nf, zh_l1p_slots()) * 131072)
                                               : 0);
  const size_t o_l1ctr = ar.reserve(256);
  const size_t o_cprev = ar.reserve(chain ? range_frags * (size_t)ZH_FRAG_SIZE * 8 : 0);
  const size_t o_cbest = ar.reserve(chain ? range_frags * (size_t)ZH_FRAG_SIZE * 4 : 0);
  ar.reserve(256);

  if (ctx_malloc(p->ctx, (void**)&p->arena, ar.size) != hipSuccess) {
    ctx->last_error = "hipMalloc(plan arena, " + std::to_string(ar.size) + " bytes)";
    delete p;
    return ZH_ERR_NOMEM;
  }
  uint8_t* base = p->arena;
  hipStream_t s = ctx->stream;
  hipError_t up = hipSuccess;
  auto chk = [&](hipError_t e) {
    if (up == hipSuccess) up = e;
  };
  chk(hipMemcpyAsync(base + o_bufs, bufs.data(), n * sizeof(ZhBufDesc), hipMemcpyHostToDevice, s));
  chk(hipMemcpyAsync(base + o_blocks, blocks.data(), nb * sizeof(ZhBlockDesc), hipMemcpyHostToDevice, s));
  chk(hipMemcpyAsync(base + o_frags, frags.data(), nf * sizeof(ZhFragDesc), hipMemcpyHostToDevice, s));
  chk(hipMemcpyAsync(base + o_pieces, pieces.data(), nf * sizeof(ZhPieceDesc), hipMemcpyHostToDevice, s));
  chk(hipMemsetAsync(base + o_fnm, 0, nf * 4, s));
  chk(hipMemsetAsync(base + o_fsp, 0, nf * 4, s));
  chk(hipMemsetAsync(base + o_fhist, 0, nf * ZH_HIST_STRIDE * 2, s));
  chk(hipMemsetAsync(base + o_fnl, 0, nf * 4, s));
  chk(hipMemsetAsync(base + o_fex, 0, nf * 4, s));
  // best[] starts out all "not worked out" -- once: every run leaves it that way again (the links kernel
  // clears the sorted positions it has borrowed the array for, zh_chain_class_links_kernel)
  if (chain && nf) chk(hipMemsetAsync(base + o_cbest, 0, range_frags * (size_t)ZH_FRAG_SIZE * 4, s));
  chk(hipStreamSynchronize(s));  // host vectors go out of scope
  if (up != hipSuccess) {
    ctx->last_error = std::string("plan upload: ") + hipGetErrorString(up);
    zh_plan_destroy(p);
    return ZH_ERR_DEVICE;
  }

  ZhCompressArgs& a = p->ca;
  a.bufs = p->d_bufs = carve<ZhBufDesc>(base, o_bufs);
  a.blocks = carve<ZhBlockDesc>(base, o_blocks);
  a.frags = carve<ZhFragDesc>(base, o_frags);
  p->d_pieces = carve<ZhPieceDesc>(base, o_pieces);
  p->npieces = (uint32_t)nf;
  a.nfrags = (uint32_t)nf;
  a.nblocks = (uint32_t)nb;
  a.nbufs = (uint32_t)n;
  a.level = level;
  a.data_format = data_format;
  a.m_pos = carve<uint16_t>(base, o_mpos);
  a.m_len = carve<uint16_t>(base, o_mlen);
  a.m_off = carve<uint16_t>(base, o_moff);
  a.f_nmatch = carve<uint32_t>(base, o_fnm);
  a.f_spill = carve<uint32_t>(base, o_fsp);
  a.f_nlit = carve<uint32_t>(base, o_fnl);
  a.f_extra_bits = carve<uint32_t>(base, o_fex);
  a.f_hist = carve<uint16_t>(base, o_fhist);
  a.f_crc = p->piece_crc = carve<uint32_t>(base, o_pcrc);
  a.f_adler = p->piece_adler = carve<uint32_t>(base, o_pad);
  p->piece_len = carve<uint32_t>(base, o_plen);
  a.f_bits = carve<uint32_t>(base, o_fbits);
  a.f_bit_start = carve<uint64_t>(base, o_fstart);
  a.b_mode = carve<uint32_t>(base, o_bmode);
  a.b_litcode = carve<uint32_t>(base, o_blit);
  a.b_distcode = carve<uint32_t>(base, o_bdist);
  a.b_hdr = carve<uint32_t>(base, o_bhdr);
  a.b_hdr_bits = carve<uint32_t>(base, o_bhb);
  a.b_bits = carve<uint64_t>(base, o_bbits);
  a.b_stored_d0 = carve<uint64_t>(base, o_bd0);
  a.b_start = carve<uint64_t>(base, o_bst);
  p->buf_crc = carve<uint32_t>(base, o_bcrc);
  p->buf_adler = carve<uint32_t>(base, o_bad);
  a.out_len = p->out_len = carve<uint64_t>(base, o_olen);
  a.status = p->status = carve<int32_t>(base, o_st);
  p->head_scratch = carve<uint32_t>(base, o_head);
  p->l1_tables = carve<uint16_t>(base, o_l1tab);
  p->l1_counter = carve<uint32_t>(base, o_l1ctr);
  p->chain_prev = carve<uint64_t>(base, o_cprev);
  p->chain_best = carve<uint32_t>(base, o_cbest);
  p->h_bufs.swap(bufs);
  p->h_blocks.swap(blocks);
  *out = p;
  return ZH_OK;
}

extern "C" int zh_plan_compress(zh_ctx* ctx, size_t n, const uint64_t* src_off,
                                const uint64_t* src_len, const uint64_t* dst_off,
                                const uint64_t* dst_cap, int level, int data_format, zh_plan** out) {
  return zh_plan_compress_blocks(ctx, n, src_off, src_len, dst_off, dst_cap, level, data_format,
                                 ZH_BLOCK_SIZE, out);  // deflate.nim:228
}

// Where every deflate block of buffer `buf` begins, from the layout kernel's positions.
extern "C" int zh_plan_block_index(zh_plan* p, size_t buf, zh_block_entry** index, size_t* n_entries) {
  if (!p || !p->is_compress || buf >= p->n || !index || !n_entries) return ZH_ERR_ARGUMENT;
  zh_ctx* ctx = p->ctx;
  *index = nullptr;
  *n_entries = 0;
  const ZhBufDesc& b = p->h_bufs[buf];
  const size_t nb = b.nblocks, nb_all = p->h_blocks.size();
  std::vector<uint64_t> start(nb + 1);
  std::vector<uint32_t> mode(nb);
  int32_t st = ZH_OK;
  hipStream_t s = ctx->stream;
  ZH_HIP(ctx, hipMemcpyAsync(start.data(), p->ca.b_start + b.first_block, nb * 8, hipMemcpyDeviceToHost, s));
  ZH_HIP(ctx, hipMemcpyAsync(&start[nb], p->ca.b_start + nb_all + buf, 8, hipMemcpyDeviceToHost, s));
  ZH_HIP(ctx, hipMemcpyAsync(mode.data(), p->ca.b_mode + b.first_block, nb * 4, hipMemcpyDeviceToHost, s));
  ZH_HIP(ctx, hipMemcpyAsync(&st, p->status + buf, 4, hipMemcpyDeviceToHost, s));
  ZH_HIP(ctx, hipStreamSynchronize(s));
  if (st != ZH_OK) return st;
  const uint64_t hdr_bits =
      8ull * (p->fmt == ZH_DF_GZIP ? 10 + b.fname_len + 1 : p->fmt == ZH_DF_ZLIB ? 2 : 0);
  std::vector<zh_block_entry> e;
  for (size_t k = 0; k < nb; k++) {
    const ZhBlockDesc& blk = p->h_blocks[b.first_block + k];
    const uint64_t out_off = blk.src_off - b.src_off;
    e.push_back(zh_block_entry{hdr_bits + start[k], out_off});
    if (mode[k] == ZH_MODE_STORED) {  // further stored chunks start on byte boundaries (deflate.nim:179-205)
      const uint64_t chunks = (blk.len + ZH_STORED_MAX - 1) / ZH_STORED_MAX;
      const uint64_t first_len_byte = (hdr_bits + start[k] + 3 + 7) >> 3;
      for (uint64_t c = 1; c < chunks; c++)
        e.push_back(zh_block_entry{(first_len_byte + c * (ZH_STORED_MAX + 5ull) - 1) * 8, out_off + c * ZH_STORED_MAX});
    }
  }
  e.push_back(zh_block_entry{hdr_bits + start[nb], b.src_len});
  *index = (zh_block_entry*)malloc(e.size() * sizeof(zh_block_entry));
  if (!*index) return ZH_ERR_NOMEM;
  memcpy(*index, e.data(), e.size() * sizeof(zh_block_entry));
  *n_entries = e.size();
  return ZH_OK;
}

// Large streams are decoded segment-wise (zh_inflate_seg.hip) in batches of up to 256 streams:
// ZH_SEG=0 turns that off, ZH_SEG_MIN is the smallest stream (compressed bytes, default 128 KiB),
// ZH_SEG_BYTES the segment length (default 32 KiB).
struct SegConfig {
  bool on = true;
  uint64_t min_stream = 131072, seg_bytes = 32768, tail_bytes = 4718592, setup_bytes = 8u << 20;
  size_t max_streams = 256;
};
static SegConfig seg_config() {  // (read per plan: the tests switch it)
  SegConfig v;
  if (const char* e = getenv("ZH_SEG")) v.on = strcmp(e, "0") != 0;
  if (const char* e = getenv("ZH_SEG_MIN")) v.min_stream = strtoull(e, nullptr, 10);
  if (const char* e = getenv("ZH_SEG_BYTES")) v.seg_bytes = std::max<uint64_t>(64, strtoull(e, nullptr, 10));
  if (const char* e = getenv("ZH_SEG_TAIL")) v.tail_bytes = strtoull(e, nullptr, 10);
  if (const char* e = getenv("ZH_SEG_SETUP")) v.setup_bytes = strtoull(e, nullptr, 10);  // (the tests' small streams: 0)
  return v;
}

// Cuts the plan's streams into segments and uploads the geometry; token regions of the segments
// are appended to the plan's token pool (`twords` is the pool's size so far).  Failure leaves the
// plan unsegmented.
static void plan_segments(zh_plan* p, const std::vector<ZhBufDesc>& bufs, uint64_t* twords) {
  const SegConfig c = seg_config();
  zh_ctx* ctx = p->ctx;
  const size_t n = bufs.size();
  if (!c.on || !n || n > c.max_streams) return;
  // the large streams of the batch are cut into segments, the others have none (and take the
  // ordinary kernels, like every stream whose chain of segments does not hold)
  auto large = [&](const ZhBufDesc& b) {
    return b.src_len >= c.min_stream && b.src_len >= 2 * c.seg_bytes && b.src_len <= (~0ull >> 4);
  };
  if (std::none_of(bufs.begin(), bufs.end(), large)) return;
  // Worth it?  The ordinary kernels give every stream one workgroup: the batch takes as long as its
  // longest stream (measured: 8.7 us per KB of compressed data); segment-wise the machine is full but a
  // byte costs four times more work (0.19 us per KB of the whole batch, 1.6 ms to set up).
  // 1 MiB streams (400 KB compressed): up to 20 of them; large ones: up to 39.
  {
    uint64_t longest = 0, total = 0;
    for (const ZhBufDesc& b : bufs) {
      longest = std::max<uint64_t>(longest, b.src_len);
      if (large(b)) total += b.src_len;
    }
    if (longest * 40 <= total + c.setup_bytes) return;
  }
  std::vector<uint32_t> parent, first_seg(n + 1), find_seg, find_batch;
  std::vector<uint64_t> nominal, search, toff, tcap, sym_base(n);
  uint64_t nsym = 0;
  for (size_t i = 0; i < n; i++) {
    const ZhBufDesc& b = bufs[i];
    const uint64_t ns = large(b) ? std::min<uint64_t>(std::max<uint64_t>(b.src_len / c.seg_bytes, 2), 4096) : 0;
    const uint64_t seg_bits = ns ? (b.src_len * 8 + ns - 1) / ns : 0, seg_len = (seg_bits + 7) / 8;
    first_seg[i] = (uint32_t)parent.size();
    for (uint64_t k = 0; k < ns; k++) {
      for (uint64_t bt = 0; k && bt * 65536 < seg_bits; bt++) {  // (the first segment's start is known)
        find_seg.push_back((uint32_t)parent.size());
        // the stream's last block (BFINAL = 1) is searched for in its last 4.5 MiB only -- this library's
        // own last block is 4 MiB of input at most --: everywhere would double the candidates
        const bool tail = (k * seg_bits + (bt + 1) * 65536) / 8 + c.tail_bytes >= b.src_len;
        find_batch.push_back((uint32_t)bt | (tail ? 0x80000000u : 0u));
      }
      if (!k) {
        find_seg.push_back((uint32_t)parent.size());
        find_batch.push_back(0);
      }
      parent.push_back((uint32_t)i);
      nominal.push_back(k * seg_bits);
      search.push_back(seg_bits);
      // room for six tokens per compressed byte of a nominal segment: a dozen segments' worth of
      // ordinary data, should the decoder have to carry on through segments without a block start
      // (a slot of no bytes is a sizing pass: the tokens are counted, never written out)
      const uint64_t cap = (b.dst_cap ? std::min<uint64_t>(b.dst_cap, 6 * seg_len) : 6 * seg_len) + 2 * (seg_len / 5 + 1) + 16;
      tcap.push_back(cap);
      toff.push_back(*twords);
      *twords += cap + 1024;
    }
    sym_base[i] = nsym;
    nsym += b.dst_cap;
  }
  first_seg[n] = (uint32_t)parent.size();
  const size_t ns = parent.size();
  if (ns > 0x7fffffffu) return;
  Arena ar;
  const size_t o_parent = ar.reserve(ns * 4), o_first = ar.reserve((n + 1) * 4), o_nom = ar.reserve(ns * 8),
               o_search = ar.reserve(ns * 8), o_toff = ar.reserve(ns * 8), o_tcap = ar.reserve(ns * 8),
               o_symb = ar.reserve(n * 8), o_start = ar.reserve(ns * 8), o_end = ar.reserve(ns * 8),
               o_final = ar.reserve(ns * 4), o_sst = ar.reserve(ns * 4), o_sout = ar.reserve(ns * 8),
               o_wlen = ar.reserve(ns * 8), o_valid = ar.reserve(ns * 4), o_prev = ar.reserve(ns * 4),
               o_ostart = ar.reserve(ns * 8), o_sok = ar.reserve(n * 4), o_order = ar.reserve(ns * 4),
               o_nchain = ar.reserve(n * 4), o_ordinal = ar.reserve(ns * 4), o_go = ar.reserve(n * 4), o_etoff = ar.reserve(ns * 8), o_etcap = ar.reserve(ns * 8), o_substart = ar.reserve(ns * 8), o_subhdr = ar.reserve(ns * 8),
               o_issub = ar.reserve(ns * 4);
  const size_t nfind = find_seg.size();
  const size_t o_fseg = ar.reserve(nfind * 4), o_fbatch = ar.reserve(nfind * 4), o_cn = ar.reserve(nfind * 4),
               o_coff = ar.reserve(nfind * 64 * 4);
  ar.reserve(256);
  if (ctx_malloc(p->ctx, (void**)&p->sg_arena, ar.size) != hipSuccess) {
    (void)hipGetLastError();
    p->sg_arena = nullptr;
    return;
  }
  uint8_t* base = p->sg_arena;
  hipStream_t s = ctx->stream;
  hipError_t up = hipMemsetAsync(base, 0, ar.size, s);
  auto put = [&](size_t off, const void* src, size_t bytes) {
    if (up == hipSuccess) up = hipMemcpyAsync(base + off, src, bytes, hipMemcpyHostToDevice, s);
  };
  put(o_parent, parent.data(), ns * 4);
  put(o_first, first_seg.data(), (n + 1) * 4);
  put(o_nom, nominal.data(), ns * 8);
  put(o_search, search.data(), ns * 8);
  put(o_toff, toff.data(), ns * 8);
  put(o_tcap, tcap.data(), ns * 8);
  put(o_symb, sym_base.data(), n * 8);
  put(o_fseg, find_seg.data(), nfind * 4);
  put(o_fbatch, find_batch.data(), nfind * 4);
  if (up == hipSuccess) up = hipStreamSynchronize(s);
  if (up != hipSuccess) {
    (void)hipGetLastError();
    ctx_free(p->ctx, p->sg_arena);
    p->sg_arena = nullptr;
    return;
  }
  ZhSegArgs& g = p->sg;
  g.nsegs = (uint32_t)ns;
  g.nstreams = (uint32_t)n;
  g.parent = carve<uint32_t>(base, o_parent);
  g.first_seg = carve<uint32_t>(base, o_first);
  g.nominal_bit = carve<uint64_t>(base, o_nom);
  g.search_bits = carve<uint64_t>(base, o_search);
  g.tok_off = carve<uint64_t>(base, o_toff);
  g.tok_cap = carve<uint64_t>(base, o_tcap);
  g.sym_base = carve<uint64_t>(base, o_symb);
  g.start_bit = carve<uint64_t>(base, o_start);
  g.end_bit = carve<uint64_t>(base, o_end);
  g.final_block = carve<uint32_t>(base, o_final);
  g.seg_status = carve<int32_t>(base, o_sst);
  g.seg_out = carve<uint64_t>(base, o_sout);
  g.wr_len = carve<uint64_t>(base, o_wlen);
  g.valid = carve<uint32_t>(base, o_valid);
  g.prev = carve<uint32_t>(base, o_prev);
  g.out_start = carve<uint64_t>(base, o_ostart);
  g.stream_ok = carve<uint32_t>(base, o_sok);
  g.order = carve<uint32_t>(base, o_order);
  g.nchain = carve<uint32_t>(base, o_nchain);
  g.ordinal = carve<uint32_t>(base, o_ordinal);
  g.go = carve<uint32_t>(base, o_go);
  g.eff_tok_off = carve<uint64_t>(base, o_etoff);
  g.eff_tok_cap = carve<uint64_t>(base, o_etcap);
  g.sub_start = carve<uint64_t>(base, o_substart);
  g.sub_hdr = carve<uint64_t>(base, o_subhdr);
  g.is_sub = carve<uint32_t>(base, o_issub);
  g.nfind = (uint32_t)nfind;
  g.find_seg = carve<uint32_t>(base, o_fseg);
  g.find_batch = carve<uint32_t>(base, o_fbatch);
  g.cand_n = carve<uint32_t>(base, o_cn);
  g.cand_off = carve<uint32_t>(base, o_coff);
  p->sg_sym_count = nsym + 64;
  p->segmented = true;
}

extern "C" int zh_plan_uncompress(zh_ctx* ctx, size_t n, const uint64_t* src_off,
                                  const uint64_t* src_len, const uint64_t* dst_off,
                                  const uint64_t* dst_cap, int data_format, zh_plan** out) {
  if (!ctx || !out || (n && (!src_off || !src_len || !dst_off || !dst_cap))) return ZH_ERR_ARGUMENT;
  *out = nullptr;
  if (data_format < ZH_DF_DETECT || data_format > ZH_DF_DEFLATE) return ZH_ERR_INVALID_FORMAT;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<ZhBufDesc> bufs(n);
  std::vector<ZhPieceDesc> pieces;
  for (size_t i = 0; i < n; i++) {
    ZhBufDesc& b = bufs[i];
    memset(&b, 0, sizeof(b));
    b.src_off = src_off[i];
    b.src_len = src_len[i];
    b.dst_off = dst_off[i];
    b.dst_cap = dst_cap[i];
    b.first_piece = (uint32_t)pieces.size();
    for (uint64_t o = 0; o < b.dst_cap; o += ZH_FRAG_SIZE)
      pieces.push_back(ZhPieceDesc{b.dst_off + o, (uint32_t)std::min<uint64_t>(b.dst_cap - o, ZH_FRAG_SIZE), (uint32_t)i, o});
    b.npieces = (uint32_t)pieces.size() - b.first_piece;
  }
  zh_plan* p = new zh_plan;
  p->ctx = ctx;
  p->is_compress = false;
  p->n = n;
  p->fmt = data_format;
  const size_t np = pieces.size();
  Arena ar;
  const size_t o_bufs = ar.reserve(n * sizeof(ZhBufDesc)), o_pieces = ar.reserve(np * sizeof(ZhPieceDesc));
  const size_t o_pcrc = ar.reserve(np * 4), o_pad = ar.reserve(np * 4), o_plen = ar.reserve(np * 4);
  const size_t o_bp = ar.reserve(n * 4), o_fmt = ar.reserve(n * 4), o_es = ar.reserve(n * 4),
               o_ei = ar.reserve(n * 4), o_bcrc = ar.reserve(n * 4), o_bad = ar.reserve(n * 4),
               o_olen = ar.reserve(n * 8), o_st = ar.reserve(n * 4);
  const size_t o_toff = ar.reserve(n * 8), o_tcap = ar.reserve(n * 8);
  ar.reserve(256);
  if (ctx_malloc(p->ctx, (void**)&p->arena, ar.size) != hipSuccess) {
    ctx->last_error = "hipMalloc(plan arena)";
    delete p;
    return ZH_ERR_NOMEM;
  }
  uint8_t* base = p->arena;
  // Token buffers of the split decode: a token makes at least one output byte and takes at least
  // one input bit; a stored block takes five input bytes and two more records than a token.
  std::vector<uint64_t> toff(n), tcap(n);
  uint64_t twords = 0;
  for (size_t i = 0; i < n; i++) {
    const uint64_t bits = bufs[i].src_len > (~0ull >> 3) ? ~0ull : bufs[i].src_len * 8;
    tcap[i] = std::min<uint64_t>(bufs[i].dst_cap, bits) + 2 * (bufs[i].src_len / 5 + 1) + 2;
    toff[i] = twords;
    twords += tcap[i] + 1024;  // (the writer reads whole batches of records, up to 640 behind the last)
  }
  plan_segments(p, bufs, &twords);
  if (!p->segmented && n) {
    // groups of streams whose token regions fit the scratch budget share the pool in turn (zh_plan_run); a
    // stream's region is then counted from its group's first
    const uint64_t budget_words = scratch_budget() / 4;
    uint64_t gwords = 0, gmax = 0;
    uint32_t g0 = 0;
    for (size_t i = 0; i < n; i++) {
      const uint64_t need = tcap[i] + 1024;
      if (i > g0 && gwords + need > budget_words) {
        p->tok_groups.push_back({g0, (uint32_t)(i - g0)});
        gmax = std::max(gmax, gwords);
        g0 = (uint32_t)i;
        gwords = 0;
      }
      toff[i] = gwords;
      gwords += need;
    }
    p->tok_groups.push_back({g0, (uint32_t)(n - g0)});
    gmax = std::max(gmax, gwords);
    if (p->tok_groups.size() > 1) twords = gmax;
    if (p->tok_groups.size() > 1 && getenv("ZH_TRACE"))
      fprintf(stderr, "zippy_hip: token pool for %zu groups of streams (%zu streams)\n", p->tok_groups.size(), n);
  }
  p->tok_words = twords + 32768;  // (... and stages them up to 8192 at a time, two stagings ahead)
  hipError_t up = hipMemcpyAsync(base + o_toff, toff.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
  if (up == hipSuccess) up = hipMemcpyAsync(base + o_tcap, tcap.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
  if (up == hipSuccess)
    up = hipMemcpyAsync(base + o_bufs, bufs.data(), n * sizeof(ZhBufDesc), hipMemcpyHostToDevice, ctx->stream);
  if (up == hipSuccess)
    up = hipMemcpyAsync(base + o_pieces, pieces.data(), np * sizeof(ZhPieceDesc), hipMemcpyHostToDevice, ctx->stream);
  if (up == hipSuccess) up = hipStreamSynchronize(ctx->stream);
  if (up != hipSuccess) {
    ctx->last_error = std::string("plan upload: ") + hipGetErrorString(up);
    zh_plan_destroy(p);
    return ZH_ERR_DEVICE;
  }
  p->d_bufs = carve<ZhBufDesc>(base, o_bufs);
  p->d_pieces = carve<ZhPieceDesc>(base, o_pieces);
  p->tok_off = carve<uint64_t>(base, o_toff);
  p->tok_cap = carve<uint64_t>(base, o_tcap);
  p->npieces = (uint32_t)np;
  p->piece_crc = carve<uint32_t>(base, o_pcrc);
  p->piece_adler = carve<uint32_t>(base, o_pad);
  p->piece_len = carve<uint32_t>(base, o_plen);
  p->buf_crc = carve<uint32_t>(base, o_bcrc);
  p->buf_adler = carve<uint32_t>(base, o_bad);
  ZhInflateArgs& a = p->ia;
  a.bufs = p->d_bufs;
  a.src_len_dev = nullptr;
  a.nbufs = (uint32_t)n;
  a.data_format = data_format;
  a.count_only = 0;
  a.body_pos = carve<uint32_t>(base, o_bp);
  a.fmt = carve<uint32_t>(base, o_fmt);
  a.expect_sum = carve<uint32_t>(base, o_es);
  a.expect_isize = carve<uint32_t>(base, o_ei);
  a.out_len = p->out_len = carve<uint64_t>(base, o_olen);
  a.status = p->status = carve<int32_t>(base, o_st);
  a.start_bit = nullptr;
  a.single_block = 0;
  a.skip = nullptr;
  *out = p;
  return ZH_OK;
}

// One stream, one decoder per deflate block (BASELINE config 5).  `ia` keeps describing the
// stream (container checks, checksum, result); `seg` describes its blocks as if they were streams.
extern "C" int zh_plan_uncompress_indexed(zh_ctx* ctx, uint64_t src_off, uint64_t src_len,
                                          uint64_t dst_off, uint64_t dst_cap, int data_format,
                                          const zh_block_entry* index, size_t n_entries, zh_plan** out) {
  if (!ctx || !out || !index || n_entries < 2 || n_entries > 0xfffffffeull) return ZH_ERR_ARGUMENT;
  *out = nullptr;
  const size_t nseg = n_entries - 1;
  for (size_t k = 0; k < nseg; k++)
    if (index[k + 1].out_off < index[k].out_off || index[k + 1].bit_off < index[k].bit_off ||
        index[k].bit_off >= src_len * 8)
      return ZH_ERR_ARGUMENT;
  if (index[0].out_off != 0) return ZH_ERR_ARGUMENT;
  if (index[nseg].out_off > dst_cap) return ZH_ERR_DST_TOO_SMALL;
  zh_plan* p = nullptr;
  const uint64_t total = index[nseg].out_off;
  int rc = zh_plan_uncompress(ctx, 1, &src_off, &src_len, &dst_off, &total, data_format, &p);
  if (rc) return rc;
  std::vector<ZhBufDesc> segs(nseg);
  std::vector<uint64_t> start(nseg);
  for (size_t k = 0; k < nseg; k++) {
    ZhBufDesc& b = segs[k];
    memset(&b, 0, sizeof(b));
    b.src_off = src_off;
    b.src_len = src_len;
    b.dst_off = dst_off + index[k].out_off;
    b.dst_cap = index[k + 1].out_off - index[k].out_off;
    start[k] = index[k].bit_off;
  }
  Arena ar;
  const size_t o_bufs = ar.reserve(nseg * sizeof(ZhBufDesc)), o_start = ar.reserve(nseg * 8),
               o_olen = ar.reserve(nseg * 8), o_st = ar.reserve(nseg * 4);
  ar.reserve(256);
  if (ctx_malloc(p->ctx, (void**)&p->seg_arena, ar.size) != hipSuccess) {
    zh_plan_destroy(p);
    return ZH_ERR_NOMEM;
  }
  uint8_t* base = p->seg_arena;
  hipError_t up = hipMemcpyAsync(base + o_bufs, segs.data(), nseg * sizeof(ZhBufDesc), hipMemcpyHostToDevice, ctx->stream);
  if (up == hipSuccess) up = hipMemcpyAsync(base + o_start, start.data(), nseg * 8, hipMemcpyHostToDevice, ctx->stream);
  if (up == hipSuccess) up = hipStreamSynchronize(ctx->stream);
  if (up != hipSuccess) {
    ctx->last_error = std::string("plan upload: ") + hipGetErrorString(up);
    zh_plan_destroy(p);
    return ZH_ERR_DEVICE;
  }
  ZhInflateArgs& g = p->seg;
  g = p->ia;
  g.bufs = carve<ZhBufDesc>(base, o_bufs);
  g.src_len_dev = nullptr;
  g.nbufs = (uint32_t)nseg;
  g.start_bit = carve<uint64_t>(base, o_start);
  g.single_block = 1;
  g.out_len = carve<uint64_t>(base, o_olen);
  g.status = carve<int32_t>(base, o_st);
  p->indexed = true;
  *out = p;
  return ZH_OK;
}

extern "C" int zh_plan_set_src_lens_device(zh_plan* plan, const uint64_t* d_lens) {
  if (!plan || plan->is_compress) return ZH_ERR_ARGUMENT;
  plan->ia.src_len_dev = d_lens;
  // The segment geometry of a plan (zh_inflate_seg.hip: where block starts are searched for, where
  // the last block may begin, the token regions) was laid over the HOST lengths -- here the slots'
  // capacities, not the streams: segments over padding, the tail window in the wrong place.  Such
  // a plan decodes with one workgroup a stream.
  if (d_lens) plan->segmented = false;
  return ZH_OK;
}

static void plan_set_count_only(zh_plan* plan, int on) { plan->ia.count_only = on; }

// ZH_INFLATE=serial keeps every stream on the two-wave serial decoder (zh_inflate.hip); the
// default decodes a stream's Huffman codes in parallel (zh_inflate_split.hip).  Sizing passes and
// the block-parallel form of one stream always use the serial kernel.
static bool inflate_split_enabled(const zh_ctx* ctx) {
  static const bool on = [] {
    const char* e = getenv("ZH_INFLATE");
    return !(e && strcmp(e, "serial") == 0);
  }();
  return ctx->inflate_mode < 0 ? on : ctx->inflate_mode == 0;
}
// BestSpeed parse: 0 the reference's (snappy.nim:12-136, byte-identical streams), 1 the parallel
// parse of zh_l1p_match.hip (valid streams of about the same size, not the reference's bytes)
static bool l1_parallel(const zh_ctx* ctx) {
  static const bool on = [] {
    const char* e = getenv("ZH_L1_PARSE");
    return e && strcmp(e, "parallel") == 0;
  }();
  return ctx->l1_parse < 0 ? on : ctx->l1_parse == 1;
}
extern "C" void zh_set_l1_parse(zh_ctx* ctx, int mode) {
  if (ctx) ctx->l1_parse = mode < 0 ? -1 : mode ? 1 : 0;
}
extern "C" void zh_set_inflate_mode(zh_ctx* ctx, int mode) {
  if (ctx) ctx->inflate_mode = mode < 0 ? -1 : mode ? 1 : 0;
}
// the token pool of a plan, allocated when it first runs in split mode; a failed allocation
// (it is several times the output) sends the plan to the serial kernel for good
static bool plan_token_pool(zh_plan* p) {
  if (!p->tok_pool) {
    if (p->tok_failed || !p->tok_words) return false;
    if (ctx_malloc(p->ctx, (void**)&p->tok_pool, p->tok_words * 4) != hipSuccess) {
      (void)hipGetLastError();
      p->tok_pool = nullptr;
      p->tok_failed = true;
      // not an error (the serial decoder gives the same bytes), but several times slower: leave a note
      p->ctx->last_error = "note: no memory for a token pool of " + std::to_string(p->tok_words * 4) +
                           " bytes; this plan decodes with the serial kernel (zh_inflate_kernel)";
      if (getenv("ZH_TRACE")) fprintf(stderr, "zippy_hip: %s\n", p->ctx->last_error.c_str());
      return false;
    }
  }
  if (p->segmented && !p->sg_sym) {  // without its buffers the plan simply is not segmented
    if (ctx_malloc(p->ctx, (void**)&p->sg_sym, p->sg_sym_count * 2) != hipSuccess ||
        ctx_malloc(p->ctx, (void**)&p->sg_windows, (size_t)p->sg.nsegs * 32768u) != hipSuccess ||
        ctx_malloc(p->ctx, (void**)&p->sg_winsym, (size_t)p->sg.nsegs * 65536u) != hipSuccess) {
      (void)hipGetLastError();
      if (p->sg_sym) ctx_free(p->ctx, p->sg_sym);
      if (p->sg_windows) ctx_free(p->ctx, p->sg_windows);
      p->sg_sym = nullptr;
      p->sg_windows = nullptr;
      p->sg_winsym = nullptr;
      p->segmented = false;
      p->ctx->last_error = "note: no memory for the segment-wise decode's buffers; large streams of this plan "
                           "are decoded by one workgroup each";
      if (getenv("ZH_TRACE")) fprintf(stderr, "zippy_hip: %s\n", p->ctx->last_error.c_str());
    } else {
      p->sg.sym = p->sg_sym;
      p->sg.windows = p->sg_windows;
      p->sg.winsym = p->sg_winsym;
    }
  }
  return true;
}
// A token pool that outlives the plan and is shared with other plans whose kernels run on the same
// stream one after the other (the pool is scratch of a run: tokens kernel -> writer).
static void plan_lend_token_pool(zh_plan* p, uint32_t* pool, uint64_t words) {
  if (p->tok_pool || p->tok_failed || !p->tok_words || p->tok_words > words) return;
  p->tok_pool = pool;
  p->tok_borrowed = true;
}

// Output slots start out zeroed (every shared output word is OR-ed into place).  Slots that tile
// one range are cleared with a single memset; slots with gaps between them are cleared one by
// one, byte-exact, so that caller data lying between two slots is never touched.
__global__ __launch_bounds__(256) void zh_zero_slots_kernel(uint8_t* __restrict__ d_dst,
                                                            const ZhBufDesc* __restrict__ bufs,
                                                            uint32_t parts) {
  // `parts` workgroups share a slot
  const uint32_t part = blockIdx.x % parts;
  const ZhBufDesc b = bufs[blockIdx.x / parts];
  uint8_t* const base = d_dst + b.dst_off;
  const uint64_t cap = b.dst_cap;
  uint64_t head = (16u - ((uintptr_t)base & 15u)) & 15u;
  if (head > cap) head = cap;
  const uint64_t nvec = (cap - head) >> 4;
  uint4* const body = reinterpret_cast<uint4*>(base + head);
  for (uint64_t i = (uint64_t)part * 256u + threadIdx.x; i < nvec; i += (uint64_t)parts * 256u)
    body[i] = make_uint4(0, 0, 0, 0);
  if (part == 0) {
    if (threadIdx.x < head) base[threadIdx.x] = 0;
    const uint64_t t0 = head + (nvec << 4);
    if (t0 + threadIdx.x < cap) base[t0 + threadIdx.x] = 0;
  }
}

extern "C" int zh_plan_run(zh_plan* p, const void* d_src_v, void* d_dst_v) {
  if (!p) return ZH_ERR_ARGUMENT;
  zh_ctx* ctx = p->ctx;
  hipStream_t s = ctx->stream;
  const uint8_t* d_src = (const uint8_t*)d_src_v;
  uint8_t* d_dst = (uint8_t*)d_dst_v;
  p->k_names.clear();
  if (!p->n) return ZH_OK;
  if (p->is_compress) {
    const ZhCompressArgs& a = p->ca;
    const int want_crc = p->fmt == ZH_DF_GZIP || p->force_crc, want_adler = p->fmt == ZH_DF_ZLIB;
    // every shared output word is OR-ed into place, so the slots start out zeroed
    // zh_emit_kernel and zh_layout_kernel address the output as aligned 32-bit words
    if ((uintptr_t)d_dst & 3u) return ZH_ERR_ARGUMENT;
    prof_mark(p, "memset_dst");
    if (p->dst_dense) {
      ZH_HIP(ctx, hipMemsetAsync(d_dst + p->dst_lo, 0, p->dst_hi - p->dst_lo, s));
    } else {
      uint32_t gy = (uint32_t)std::min<uint64_t>(64, (p->dst_max_cap >> 16) + 1);
      while (gy > 1 && (uint64_t)p->n * gy > 0x7fffffffull) gy >>= 1;  // (a grid has fewer than 2^31 workgroups)
      if ((uint64_t)p->n * gy > 0x7fffffffull) return ZH_ERR_ARGUMENT;
      hipLaunchKernelGGL(zh_zero_slots_kernel, dim3((uint32_t)p->n * gy), dim3(256), 0, s, d_dst, p->d_bufs, gy);
      ZH_HIP(ctx, hipGetLastError());
    }
    if (p->level == 1 && l1_parallel(ctx)) {
      prof_mark(p, "zh_l1p_match_kernel");
      zh_launch_l1p_match(s, d_src, a, p->l1_tables, p->l1_counter);
    } else if (p->level == 1 || p->level == -2) {
      prof_mark(p, "zh_l1_match_kernel");
      zh_launch_l1_match(s, d_src, a, p->level == -2, p->l1_tables, p->l1_counter);
    } else if (p->level != 0) {
      const int* cfg = kChainConfig[p->level == -1 ? 6 : p->level];
      if (p->chain_best_dirty)
        ZH_HIP(ctx, hipMemsetAsync(p->chain_best, 0, p->chain_scratch_frags * (size_t)ZH_FRAG_SIZE * 4u, s));
      p->chain_best_dirty = true;
      for (const auto& r : p->chain_ranges) {  // (one range unless the scratch budget says otherwise)
        ZhCompressArgs ar = a;
        ar.first_block = r.b0;
        ar.nblocks = r.nb;
        ar.first_frag = r.f0;
        ar.nfrags = r.nf;
        prof_mark(p, "zh_chain_prev_kernel");
        zh_launch_chain_prev(s, d_src, ar, p->head_scratch, p->chain_prev, p->chain_best);
        ZH_HIP(ctx, hipGetLastError());
        prof_mark(p, "zh_chain_walk_kernel");
        zh_launch_chain_search(s, d_src, ar, cfg[0], cfg[1], cfg[2], p->chain_prev, p->chain_best);
        prof_mark(p, "zh_chain_select_kernel");
        zh_launch_chain_select(s, d_src, ar, cfg[0], cfg[1], cfg[2], p->chain_prev, p->chain_best);
        ZH_HIP(ctx, hipGetLastError());
      }
      p->chain_best_dirty = false;  // (every launch was accepted: the links kernel hands best[] back cleared)
      prof_mark(p, "zh_frag_stats_kernel");
      zh_launch_frag_stats(s, d_src, a);
    }
    if (want_crc || want_adler) {
      prof_mark(p, "zh_checksum_pieces_kernel");
      zh_launch_checksum_pieces(s, ctx->cktabs, d_src, p->d_pieces, p->npieces, nullptr, want_crc,
                                want_adler, p->piece_crc, p->piece_adler, p->piece_len);
      prof_mark(p, "zh_checksum_combine_kernel");
      zh_launch_checksum_combine(s, p->d_bufs, (uint32_t)p->n, p->piece_crc, p->piece_adler,
                                 p->piece_len, want_crc, want_adler, p->buf_crc, p->buf_adler);
    }
    prof_mark(p, "zh_huffman_kernel");
    // (contract mode -- zh_set_l1_parse(ctx, 1), BestSpeed only -- also builds the block's codes without the
    // replay of the reference's heap: optimal codes, other tie-breaks)
    zh_launch_huffman(s, a, p->level == 1 && l1_parallel(ctx) ? 1 : 0);
    prof_mark(p, "zh_layout_kernel");
    zh_launch_layout(s, d_dst, a, p->buf_crc, p->buf_adler);
    prof_mark(p, "zh_emit_kernel");
    zh_launch_emit(s, d_src, d_dst, a);
    prof_mark(p, "end");
  } else {
    const ZhInflateArgs& a = p->ia;
    prof_mark(p, "zh_unwrap_kernel");
    zh_launch_unwrap(s, d_src, a);
    const bool split_ok = !p->indexed && inflate_split_enabled(ctx) && plan_token_pool(p);
    const bool split = split_ok && !a.count_only;
    ZhInflateArgs a1 = a;
    if (split_ok && p->segmented) {
      // a handful of large streams: many workgroups per stream (zh_inflate_seg.hip); streams whose
      // chain of segments does not hold are left to the ordinary kernels below.  A sizing pass
      // stops behind the chain kernel, which knows the output size by then.
      prof_mark(p, "zh_seg_find_kernel");
      zh_launch_seg_find(s, d_src, a, p->sg);
      prof_mark(p, "zh_seg_check_kernel");
      zh_launch_seg_check(s, d_src, a, p->sg);
      prof_mark(p, "zh_seg_substart_kernel");
      zh_launch_seg_tokens(s, d_src, a, p->tok_pool, p->sg, 0);
      zh_launch_seg_decide(s, a, p->sg);
      prof_mark(p, "zh_seg_tokens_kernel");
      zh_launch_seg_tokens(s, d_src, a, p->tok_pool, p->sg, 1);
      prof_mark(p, "zh_seg_chain_kernel");
      zh_launch_seg_chain(s, a, p->sg);
      if (!a.count_only) {
        prof_mark(p, "zh_seg_write_kernel");
        zh_launch_seg_write(s, d_src, a, p->tok_pool, p->sg);
        prof_mark(p, "zh_seg_windows_kernel");
        zh_launch_seg_windows(s, a, p->sg);
        prof_mark(p, "zh_seg_finish_kernel");
        zh_launch_seg_finish(s, d_dst, a, p->sg);
      }
      a1.skip = p->sg.stream_ok;
    }
    if (p->indexed) {
      prof_mark(p, "zh_inflate_kernel");
      ZH_HIP(ctx, hipMemsetAsync(p->seg.status, 0, (size_t)p->seg.nbufs * 4, s));
      zh_launch_inflate(s, d_src, d_dst, p->seg);
      prof_mark(p, "zh_segments_reduce_kernel");
      zh_launch_segments_reduce(s, p->seg, a);
    } else if (split) {
      // two kernels: tokens (parallel over each stream), then bytes (zh_inflate_split.hip)
      if (p->tok_groups.size() <= 1) {
        prof_mark(p, "zh_inflate_tokens_kernel");
        zh_launch_inflate_tokens(s, d_src, a1, p->tok_pool, p->tok_off, p->tok_cap);
        prof_mark(p, "zh_inflate_write_kernel");
        zh_launch_inflate_write(s, d_src, d_dst, a1, p->tok_pool, p->tok_off);
      } else {
        for (const auto& g : p->tok_groups) {  // the pool holds a group's records at a time
          ZhInflateArgs ag = a1;
          ag.first_buf = g.first;
          ag.nbufs = g.second;
          prof_mark(p, "zh_inflate_tokens_kernel");
          zh_launch_inflate_tokens(s, d_src, ag, p->tok_pool, p->tok_off, p->tok_cap);
          prof_mark(p, "zh_inflate_write_kernel");
          zh_launch_inflate_write(s, d_src, d_dst, ag, p->tok_pool, p->tok_off);
        }
      }
    } else {
      prof_mark(p, "zh_inflate_kernel");
      zh_launch_inflate(s, d_src, d_dst, a1);
    }
    if (!a.count_only) {
      // both checksums: with dfDetect the format is only known per stream on the device
      const int want_crc = p->fmt == ZH_DF_GZIP || p->fmt == ZH_DF_DETECT || p->force_crc;
      const int want_adler = p->fmt == ZH_DF_ZLIB || p->fmt == ZH_DF_DETECT;
      if (want_crc || want_adler) {
        prof_mark(p, "zh_checksum_pieces_kernel");
        zh_launch_checksum_pieces(s, ctx->cktabs, d_dst, p->d_pieces, p->npieces, p->out_len,
                                  want_crc, want_adler, p->piece_crc, p->piece_adler, p->piece_len);
        prof_mark(p, "zh_checksum_combine_kernel");
        zh_launch_checksum_combine(s, p->d_bufs, (uint32_t)p->n, p->piece_crc, p->piece_adler,
                                   p->piece_len, want_crc, want_adler, p->buf_crc, p->buf_adler);
        prof_mark(p, "zh_verify_kernel");
        zh_launch_verify(s, a, p->buf_crc, p->buf_adler);
      }
    }
    prof_mark(p, "end");
  }
  ZH_HIP(ctx, hipGetLastError());
  return ZH_OK;
}

extern "C" int zh_plan_results(zh_plan* p, uint64_t* out_lens, int32_t* statuses) {
  if (!p) return ZH_ERR_ARGUMENT;
  zh_ctx* ctx = p->ctx;
  if (out_lens) ZH_HIP(ctx, hipMemcpyAsync(out_lens, p->out_len, p->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (statuses) ZH_HIP(ctx, hipMemcpyAsync(statuses, p->status, p->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  ZH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZH_OK;
}
extern "C" int zh_plan_request_crc32(zh_plan* p, int on) {
  if (!p) return ZH_ERR_ARGUMENT;
  p->force_crc = on != 0;
  return ZH_OK;
}
extern "C" int zh_plan_crc32(zh_plan* p, uint32_t* crcs) {
  if (!p || !crcs || !p->buf_crc) return ZH_ERR_ARGUMENT;
  zh_ctx* ctx = p->ctx;
  ZH_HIP(ctx, hipMemcpyAsync(crcs, p->buf_crc, p->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  ZH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZH_OK;
}
extern "C" const uint64_t* zh_plan_device_lens(zh_plan* p) { return p ? p->out_len : nullptr; }
extern "C" const int32_t* zh_plan_device_statuses(zh_plan* p) { return p ? p->status : nullptr; }

// ---------------------------------------------------------------------------
// host-buffer API
// ---------------------------------------------------------------------------
namespace {
struct DevBuf {
  uint8_t* p = nullptr;
  zh_ctx* ctx = nullptr;
  ~DevBuf() {
    if (p) ctx_free(ctx, p);
  }
};
hipError_t dev_alloc(zh_ctx* ctx, DevBuf& b, size_t bytes) {
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

// ---- staging between pageable host memory and HBM ----
// The caller's buffers are pageable and the results are fresh `malloc`s: a plain hipMemcpy of
// either runs at a fraction of the link (a bounce copy inside the runtime, one page fault per
// 4 KiB of a fresh result).  Instead the batch moves in chunks through two pinned buffers: host
// threads gather/scatter one chunk while the DMA engine moves the other.
// bytes per staging chunk (ZH_PIN_CHUNK: test override, so that small cases cross chunk borders)
size_t pin_chunk() {
  static const size_t c = [] {
    const char* e = getenv("ZH_PIN_CHUNK");
    const long long v = e ? atoll(e) : 0;
    return v >= 65536 && v <= ((long long)1 << 30) ? (size_t)v & ~(size_t)4095 : (size_t)32 << 20;
  }();
  return c;
}

unsigned host_threads() {
  static const unsigned t = [] {
    const char* e = getenv("ZH_HOST_THREADS");
    const long v = e ? atol(e) : 0;
    if (v >= 1 && v <= 64) return (unsigned)v;
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(8u, hc / 2u));
  }();
  return t;
}

// The host threads that gather and scatter staging chunks: one pool per process, grown on demand,
// so that a chunk costs a wake-up, not a round of thread creation.
class HostPool {
 public:
  static HostPool& get() {
    static HostPool pool;
    return pool;
  }
  // f(t) for every share t in [0, nt), on the caller and up to nt - 1 pool threads; returns when
  // all shares are done.  (Threads that cannot be had only mean fewer helpers.)
  void run(unsigned nt, const std::function<void(unsigned)>& f) {
    if (nt <= 1) {
      f(0u);
      return;
    }
    std::lock_guard<std::mutex> one_job(call_m_);
    grow(nt - 1);
    {
      std::lock_guard<std::mutex> l(m_);
      job_ = &f;
      shares_ = nt;
      next_ = 0;
      left_ = nt;
      gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> l(m_);
    done_cv_.wait(l, [&] { return left_ == 0; });
    job_ = nullptr;
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }

 private:
  void grow(unsigned want) {
    while (th_.size() < want && th_.size() < 63) {
      try {
        th_.emplace_back([this] {
          uint64_t seen = 0;
          for (;;) {
            {
              std::unique_lock<std::mutex> l(m_);
              cv_.wait(l, [&] { return stop_ || gen_ != seen; });
              if (stop_) return;
              seen = gen_;
            }
            work();
          }
        });
      } catch (...) {
        return;
      }
    }
  }
  void work() {  // take shares until none is left
    for (;;) {
      unsigned t;
      const std::function<void(unsigned)>* job;
      {
        std::lock_guard<std::mutex> l(m_);
        if (!job_ || next_ >= shares_) return;
        t = next_++;
        job = job_;
      }
      (*job)(t);
      {
        std::lock_guard<std::mutex> l(m_);
        if (--left_ == 0) done_cv_.notify_all();
      }
    }
  }
  std::mutex call_m_, m_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> th_;
  const std::function<void(unsigned)>* job_ = nullptr;
  unsigned shares_ = 0, next_ = 0, left_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

int pin_init(zh_ctx* ctx) {
  for (int k = 0; k < 2; k++) {
    if (!ctx->pin[k]) ZH_HIP(ctx, hipHostMalloc(&ctx->pin[k], pin_chunk(), 0));
    if (!ctx->pin_ev[k]) ZH_HIP(ctx, hipEventCreate(&ctx->pin_ev[k]));
  }
  return ZH_OK;
}
// the DMA that last used staging chunk k has finished
int pin_wait(zh_ctx* ctx, int k) {
  if (ctx->pin_busy[k]) {
    ctx->pin_busy[k] = false;
    ZH_HIP(ctx, hipEventSynchronize(ctx->pin_ev[k]));
  }
  return ZH_OK;
}

// A batch laid out in one linear range: buffer i occupies [off[i], off[i] + len[i]) of it.
// Copies range [lo, hi) between that layout and a staging chunk that holds it from `lo`:
// to_stage: host buffers -> staging, else staging -> host buffers.
void stage_range(uint8_t* stage, uint64_t lo, uint64_t hi, const std::vector<uint64_t>& off,
                 const std::vector<uint64_t>& len, void* const* host, bool to_stage) {
  size_t i = (size_t)(std::upper_bound(off.begin(), off.end(), lo) - off.begin());
  if (i) i--;
  for (; i < off.size() && off[i] < hi; i++) {
    const uint64_t b = std::max(off[i], lo), e = std::min(off[i] + len[i], hi);
    if (b >= e || !host[i]) continue;
    uint8_t* h = (uint8_t*)host[i] + (b - off[i]);
    if (to_stage)
      memcpy(stage + (b - lo), h, e - b);
    else
      memcpy(h, stage + (b - lo), e - b);
  }
}
void stage_chunk(uint8_t* stage, uint64_t lo, uint64_t hi, const std::vector<uint64_t>& off,
                 const std::vector<uint64_t>& len, void* const* host, bool to_stage) {
  const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(host_threads(), (hi - lo) >> 16));
  const uint64_t per = ((hi - lo + nt - 1) / nt + 4095) & ~(uint64_t)4095;
  HostPool::get().run(nt, [&](unsigned t) {
    const uint64_t a = lo + per * t, b = std::min(hi, a + per);
    if (a < b) stage_range(stage + (a - lo), a, b, off, len, host, to_stage);
  });
}

// 256-byte aligned slices of one linear range; returns its size
uint64_t layout_slices(const size_t* lens, size_t n, std::vector<uint64_t>& off,
                       std::vector<uint64_t>& len64) {
  off.resize(n);
  len64.resize(n);
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) {
    off[i] = total;
    len64[i] = lens[i];
    total += (lens[i] + 255) & ~(uint64_t)255;
  }
  return total;
}
// host buffers -> dev[0, total) in that layout, chunk by chunk on `stream`
int upload_slices(zh_ctx* ctx, hipStream_t stream, const void* const* srcs,
                  const std::vector<uint64_t>& off, const std::vector<uint64_t>& len64,
                  uint64_t total, uint8_t* dev) {
  int st = pin_init(ctx);
  if (st) return st;
  int k = 0;
  for (uint64_t lo = 0; lo < total; lo += pin_chunk(), k ^= 1) {
    const uint64_t hi = std::min<uint64_t>(total, lo + pin_chunk());
    if ((st = pin_wait(ctx, k))) return st;
    stage_chunk(ctx->pin[k], lo, hi, off, len64, (void* const*)srcs, true);
    ZH_HIP(ctx, hipMemcpyAsync(dev + lo, ctx->pin[k], hi - lo, hipMemcpyHostToDevice, stream));
    ZH_HIP(ctx, hipEventRecord(ctx->pin_ev[k], stream));
    ctx->pin_busy[k] = true;
  }
  return ZH_OK;
}
// Pack host buffers into one device allocation.
int upload(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n, DevBuf& dev,
           std::vector<uint64_t>& off, std::vector<uint64_t>& len64) {
  const uint64_t total = layout_slices(lens, n, off, len64);
  if (dev_alloc(ctx, dev, total + 256) != hipSuccess) return ZH_ERR_NOMEM;
  return upload_slices(ctx, ctx->stream, srcs, off, len64, total, dev.p);
}

struct PackPiece {
  uint64_t src, dst;
  uint32_t len, pad;
};
struct alignas(16) Vec16 {
  uint32_t a, b, c, d;
};
}  // namespace

// results gathered from their (sparse) output slots into one dense range, 16 bytes at a time
__global__ __launch_bounds__(256) void zh_pack_kernel(const uint8_t* __restrict__ src,
                                                      uint8_t* __restrict__ dst,
                                                      const PackPiece* __restrict__ pieces) {
  const PackPiece p = pieces[blockIdx.x];
  const Vec16* s = reinterpret_cast<const Vec16*>(src + p.src);
  Vec16* d = reinterpret_cast<Vec16*>(dst + p.dst);
  const uint32_t nv = p.len >> 4;
  for (uint32_t i = threadIdx.x; i < nv; i += 256) d[i] = s[i];
  for (uint32_t i = (nv << 4) + threadIdx.x; i < p.len; i += 256) dst[p.dst + i] = src[p.src + i];
}

namespace {
// Results of the buffers with status ZH_OK: `malloc`ed and filled from their device slots
// d_dst + doff[i] (olen[i] bytes each), in two steps.
struct Download {
  std::vector<uint64_t> poff, plen;  // the dense layout the results are packed into
  uint64_t total = 0;
  DevBuf own_pack, d_pieces;
  uint8_t* pack = nullptr;
};
// step 1, on `stream`: allocate the results and pack them densely on the device (into `pack`,
// at least as large as the output slots together, or into a buffer of the Download's own)
int download_pack(zh_ctx* ctx, hipStream_t stream, Download& dl, const uint8_t* d_dst, size_t n,
                  const std::vector<uint64_t>& doff, const std::vector<uint64_t>& olen,
                  const std::vector<char>& take, uint8_t* pack, void** dsts, size_t* dst_lens,
                  int32_t* statuses) {
  constexpr uint32_t kPiece = 1u << 18;
  dl.poff.assign(n, 0);
  dl.plen.assign(n, 0);
  std::vector<PackPiece> pieces;
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) {
    dl.poff[i] = total;
    dl.plen[i] = take[i] ? olen[i] : 0;
    for (uint64_t o = 0; o < dl.plen[i]; o += kPiece)
      pieces.push_back({doff[i] + o, total + o, (uint32_t)std::min<uint64_t>(kPiece, dl.plen[i] - o), 0});
    total += (dl.plen[i] + 15) & ~(uint64_t)15;
  }
  dl.total = total;
  for (size_t i = 0; i < n; i++) {
    if (!take[i]) continue;
    if (ctx->into_ptrs) {  // the caller's buffer, if the result fits (its size is reported either way)
      const size_t gi = (size_t)((dsts + i) - ctx->into_base);
      dst_lens[i] = olen[i];
      if (olen[i] > ctx->into_caps[gi] || (!ctx->into_ptrs[gi] && olen[i])) {
        statuses[i] = ZH_ERR_DST_TOO_SMALL;
        continue;
      }
      dsts[i] = ctx->into_ptrs[gi];
      continue;
    }
    dsts[i] = malloc(olen[i] ? olen[i] : 1);
    if (!dsts[i]) {
      statuses[i] = ZH_ERR_NOMEM;
      continue;
    }
    dst_lens[i] = olen[i];
  }
  if (!total) return ZH_OK;
  if (!pack) {
    if (dev_alloc(ctx, dl.own_pack, total) != hipSuccess) return ZH_ERR_NOMEM;
    pack = dl.own_pack.p;
  }
  dl.pack = pack;
  if (dev_alloc(ctx, dl.d_pieces, pieces.size() * sizeof(PackPiece)) != hipSuccess) return ZH_ERR_NOMEM;
  ZH_HIP(ctx, hipMemcpyAsync(dl.d_pieces.p, pieces.data(), pieces.size() * sizeof(PackPiece),
                             hipMemcpyHostToDevice, stream));
  const PackPiece* const dev_pieces = reinterpret_cast<const PackPiece*>(dl.d_pieces.p);
  hipLaunchKernelGGL(zh_pack_kernel, dim3((uint32_t)pieces.size()), dim3(256), 0, stream, d_dst, pack,
                     dev_pieces);
  return ZH_OK;
}
// step 2, on `stream` (ordered behind step 1 by the caller): chunk c+1 is on the wire while the
// host threads scatter chunk c
int download_fetch(zh_ctx* ctx, hipStream_t stream, const Download& dl, void** dsts) {
  if (!dl.total) return ZH_OK;
  int st = pin_init(ctx);
  if (st) return st;
  const uint64_t total = dl.total, nchunks = (total + pin_chunk() - 1) / pin_chunk();
  auto fetch = [&](uint64_t c) -> int {
    const int k = (int)(c & 1);
    const uint64_t lo = c * pin_chunk(), hi = std::min<uint64_t>(total, lo + pin_chunk());
    ZH_HIP(ctx, hipMemcpyAsync(ctx->pin[k], dl.pack + lo, hi - lo, hipMemcpyDeviceToHost, stream));
    ZH_HIP(ctx, hipEventRecord(ctx->pin_ev[k], stream));
    ctx->pin_busy[k] = true;
    return ZH_OK;
  };
  if ((st = pin_wait(ctx, 0)) || (st = pin_wait(ctx, 1)) || (st = fetch(0))) return st;
  for (uint64_t c = 0; c < nchunks; c++) {
    if (c + 1 < nchunks && (st = fetch(c + 1))) return st;
    const int k = (int)(c & 1);
    if ((st = pin_wait(ctx, k))) return st;
    const uint64_t lo = c * pin_chunk(), hi = std::min<uint64_t>(total, lo + pin_chunk());
    stage_chunk(ctx->pin[k], lo, hi, dl.poff, dl.plen, dsts, false);
  }
  return ZH_OK;
}
int download(zh_ctx* ctx, const uint8_t* d_dst, size_t n, const std::vector<uint64_t>& doff,
             const std::vector<uint64_t>& olen, const std::vector<char>& take, void** dsts,
             size_t* dst_lens, int32_t* statuses) {
  Download dl;
  int st = download_pack(ctx, ctx->stream, dl, d_dst, n, doff, olen, take, nullptr, dsts, dst_lens, statuses);
  if (st) return st;
  return download_fetch(ctx, ctx->stream, dl, dsts);
}

// Batches of a GiB and more: groups of buffers (ZH_PIPE_GROUP bytes of input each) take turns, so
// that the kernels of one group run while the host threads and the DMA engine move the
// previous group's results out and the next group's buffers in.  A group must fill the machine
// by itself, or splitting costs more than the overlap hides (ZH_PIPE_MIN: smallest batch that is
// split).
uint64_t env_bytes(const char* name, uint64_t dflt) {
  const char* e = getenv(name);
  const long long v = e ? atoll(e) : 0;
  return v > 0 ? (uint64_t)v : dflt;
}
uint64_t pipe_group_bytes(const zh_ctx* ctx) {
  static const uint64_t v = env_bytes("ZH_PIPE_GROUP", (uint64_t)512 << 20);
  return ctx->pipe_group ? ctx->pipe_group : v;
}
uint64_t pipe_min_bytes(const zh_ctx* ctx) {
  static const uint64_t v = env_bytes("ZH_PIPE_MIN", (uint64_t)1 << 30);
  return ctx->pipe_min ? ctx->pipe_min : v;
}
constexpr int kPipeFallback = -1;  // not a status: "run this batch the plain way"

struct PipeGroup {
  size_t i0 = 0, n = 0;
  std::vector<uint64_t> soff, slen, doff, dcap;
  uint64_t src_total = 0, dst_total = 0;
  PlanGuard pg;
  hipEvent_t uploaded = nullptr, packed = nullptr;
  Download dl;
  ~PipeGroup() {
    if (uploaded) (void)hipEventDestroy(uploaded);
    if (packed) (void)hipEventDestroy(packed);
  }
};

// ZH_OK: done.  kPipeFallback: the batch does not split, memory for the groups' second set of
// buffers is not to be had, or some buffer outgrew its typical slot; nothing was returned, the
// caller runs the batch the plain way.
int compress_batch_pipelined(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                             int level, int data_format, void** dsts, size_t* dst_lens,
                             int32_t* statuses, uint32_t* crcs) {
  // at most 16 groups (each has its own plan scratch): very large batches get larger groups
  uint64_t in_total = 0;
  for (size_t i = 0; i < n; i++) in_total += lens[i];
  const uint64_t group_bytes = std::max<uint64_t>(pipe_group_bytes(ctx), in_total / 16);
  std::vector<size_t> cut{0};
  uint64_t acc = 0;
  for (size_t i = 0; i < n; i++) {
    acc += lens[i];
    if (acc >= group_bytes) {
      cut.push_back(i + 1);
      acc = 0;
    }
  }
  if (cut.back() != n) cut.push_back(n);
  const size_t G = cut.size() - 1;
  if (G < 2) return kPipeFallback;
  if (!ctx->copy_stream) ZH_HIP(ctx, hipStreamCreate(&ctx->copy_stream));
  hipStream_t cs = ctx->copy_stream, ks = ctx->stream;
  Trace tr;
  std::vector<PipeGroup> gs(G);
  int st;
  // The groups take turns in TWO sets of source / output / pack buffers (group g uses set g % 2: by
  // the time group g + 2 touches a buffer of the set, group g's last use of it lies before it on the
  // same stream or has been waited for on the host -- see the loop below); only the plans' own
  // scratch (match lists, histograms) is per group.  Everything is allocated before the pipeline
  // starts: hipMalloc / hipFree in the middle would serialise it.
  uint64_t set_src[2] = {0, 0}, set_dst[2] = {0, 0};
  for (size_t g = 0; g < G; g++) {
    PipeGroup& q = gs[g];
    q.i0 = cut[g];
    q.n = cut[g + 1] - cut[g];
    q.src_total = layout_slices(lens + q.i0, q.n, q.soff, q.slen);
    q.doff.resize(q.n);
    q.dcap.resize(q.n);
    for (size_t i = 0; i < q.n; i++) {
      q.doff[i] = q.dst_total;
      q.dcap[i] = typical_cap(lens[q.i0 + i], data_format);
      q.dst_total += (q.dcap[i] + 255) & ~(uint64_t)255;
    }
    set_src[g & 1] = std::max(set_src[g & 1], q.src_total);
    set_dst[g & 1] = std::max(set_dst[g & 1], q.dst_total);
  }
  DevBuf b_src[2], b_dst[2], b_pack[2];
  for (int k = 0; k < 2; k++)
    if (dev_alloc(ctx, b_src[k], set_src[k] + 256) != hipSuccess ||
        dev_alloc(ctx, b_dst[k], set_dst[k] + 256) != hipSuccess ||
        dev_alloc(ctx, b_pack[k], set_dst[k] + 256) != hipSuccess) {
      (void)hipGetLastError();
      return kPipeFallback;
    }
  for (size_t g = 0; g < G; g++) {
    PipeGroup& q = gs[g];
    ZH_HIP(ctx, hipEventCreate(&q.uploaded));
    ZH_HIP(ctx, hipEventCreate(&q.packed));
    st = zh_plan_compress(ctx, q.n, q.soff.data(), q.slen.data(), q.doff.data(), q.dcap.data(), level,
                          data_format, &q.pg.p);
    if (st == ZH_ERR_NOMEM) return kPipeFallback;
    if (st) return st;
    if (crcs) zh_plan_request_crc32(q.pg.p, 1);
  }
  ZH_HIP(ctx, hipStreamSynchronize(ks));  // the plans' descriptors are in place
  auto up = [&](size_t g) -> int {
    PipeGroup& q = gs[g];
    int e = upload_slices(ctx, cs, srcs + q.i0, q.soff, q.slen, q.src_total, b_src[g & 1].p);
    if (e) return e;
    ZH_HIP(ctx, hipEventRecord(q.uploaded, cs));
    return ZH_OK;
  };
  auto run = [&](size_t g) -> int {
    PipeGroup& q = gs[g];
    ZH_HIP(ctx, hipStreamWaitEvent(ks, q.uploaded, 0));
    return zh_plan_run(q.pg.p, b_src[g & 1].p, b_dst[g & 1].p);
  };
  auto give_up = [&](int code) -> int {  // nothing is handed out from a failed call
    (void)hipStreamSynchronize(cs);
    (void)hipStreamSynchronize(ks);
    for (int k = 0; k < 2; k++) ctx->pin_busy[k] = false;
    for (size_t i = 0; i < n; i++) {
      if (!ctx->into_ptrs) free(dsts[i]);  // (zh_*_batch_into: the buffers are the caller's)
      dsts[i] = nullptr;
      dst_lens[i] = 0;
      statuses[i] = ZH_OK;
    }
    return code;
  };
  if ((st = up(0)) || (st = run(0))) return give_up(st);
  for (size_t g = 0; g < G; g++) {
    PipeGroup& q = gs[g];
    if (g + 1 < G && (st = up(g + 1))) return give_up(st);  // while group g's kernels run
    std::vector<uint64_t> olen(q.n);
    std::vector<int32_t> ost(q.n);
    if ((st = zh_plan_results(q.pg.p, olen.data(), ost.data()))) return give_up(st);
    if (crcs && (st = zh_plan_crc32(q.pg.p, crcs + q.i0))) return give_up(st);
    std::vector<char> take(q.n);
    for (size_t i = 0; i < q.n; i++) {
      if (ost[i] == ZH_ERR_DST_TOO_SMALL) return give_up(kPipeFallback);
      statuses[q.i0 + i] = ost[i];
      take[i] = ost[i] == ZH_OK;
    }
    st = download_pack(ctx, ks, q.dl, b_dst[g & 1].p, q.n, q.doff, olen, take, b_pack[g & 1].p, dsts + q.i0,
                       dst_lens + q.i0, statuses + q.i0);
    if (st) return give_up(st);
    if (hipEventRecord(q.packed, ks) != hipSuccess) return give_up(ZH_ERR_DEVICE);
    if (g + 1 < G && (st = run(g + 1))) return give_up(st);  // next kernels behind the pack
    if (hipStreamWaitEvent(cs, q.packed, 0) != hipSuccess) return give_up(ZH_ERR_DEVICE);
    if ((st = download_fetch(ctx, cs, q.dl, dsts + q.i0))) return give_up(st);
  }
  ZH_HIP(ctx, hipStreamSynchronize(cs));
  tr.mark(ctx, "compress: pipelined groups");
  return ZH_OK;
}
}  // namespace

static int compress_batch_impl(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                               int level, int data_format, void** dsts, size_t* dst_lens,
                               int32_t* statuses, uint32_t* crcs) {
  if (!ctx || (n && (!srcs || !lens || !dsts || !dst_lens || !statuses))) return ZH_ERR_ARGUMENT;
  for (size_t i = 0; i < n; i++) {
    dsts[i] = nullptr;
    dst_lens[i] = 0;
    statuses[i] = ZH_OK;
  }
  if (level < -2 || level > 9) {
    for (size_t i = 0; i < n; i++) statuses[i] = ZH_ERR_INVALID_LEVEL;
    return ZH_ERR_INVALID_LEVEL;
  }
  if (data_format != ZH_DF_GZIP && data_format != ZH_DF_ZLIB && data_format != ZH_DF_DEFLATE) {
    for (size_t i = 0; i < n; i++) statuses[i] = ZH_ERR_INVALID_FORMAT;
    return ZH_ERR_INVALID_FORMAT;
  }
  if (!n) return ZH_OK;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  uint64_t in_total = 0;
  for (size_t i = 0; i < n; i++) in_total += lens[i];
  if (in_total >= pipe_min_bytes(ctx)) {
    const int ps = compress_batch_pipelined(ctx, srcs, lens, n, level, data_format, dsts, dst_lens,
                                            statuses, crcs);
    if (ps != kPipeFallback) return ps;
  }
  DevBuf d_src;
  std::vector<uint64_t> soff, slen;
  Trace tr;
  int st = upload(ctx, srcs, lens, n, d_src, soff, slen);
  if (st) return st;
  tr.mark(ctx, "compress: upload");

  for (int attempt = 0; attempt < 2; attempt++) {
    std::vector<uint64_t> doff(n), dcap(n);
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) {
      doff[i] = total;
      dcap[i] = attempt == 0 ? typical_cap(lens[i], data_format) : zh_compress_bound(lens[i], data_format);
      total += (dcap[i] + 255) & ~(uint64_t)255;
    }
    DevBuf d_dst;
    if (dev_alloc(ctx, d_dst, total + 256) != hipSuccess) return ZH_ERR_NOMEM;
    PlanGuard pg;
    st = zh_plan_compress(ctx, n, soff.data(), slen.data(), doff.data(), dcap.data(), level,
                          data_format, &pg.p);
    if (st) return st;
    tr.mark(ctx, "compress: alloc + plan");
    if (crcs) zh_plan_request_crc32(pg.p, 1);
    st = zh_plan_run(pg.p, d_src.p, d_dst.p);
    if (st) return st;
    std::vector<uint64_t> olen(n);
    std::vector<int32_t> ost(n);
    st = zh_plan_results(pg.p, olen.data(), ost.data());
    if (st) return st;
    tr.mark(ctx, "compress: kernels");
    if (crcs && (st = zh_plan_crc32(pg.p, crcs))) return st;
    bool retry = false;
    for (size_t i = 0; i < n; i++)
      if (ost[i] == ZH_ERR_DST_TOO_SMALL) retry = true;
    if (retry && attempt == 0) continue;
    std::vector<char> take(n);
    for (size_t i = 0; i < n; i++) {
      statuses[i] = ost[i];
      take[i] = ost[i] == ZH_OK;
    }
    if ((st = download(ctx, d_dst.p, n, doff, olen, take, dsts, dst_lens, statuses))) return st;
    tr.mark(ctx, "compress: download");
    break;
  }
  return ZH_OK;
}

extern "C" int zh_compress_batch(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                                 int level, int data_format, void** dsts, size_t* dst_lens,
                                 int32_t* statuses) {
  return compress_batch_impl(ctx, srcs, lens, n, level, data_format, dsts, dst_lens, statuses, nullptr);
}
// Results into buffers of the caller's: dsts[i] / caps[i] on entry.  A result that does not fit
// gets ZH_ERR_DST_TOO_SMALL and its size in dst_lens[i]; on return dsts[i] is the caller's pointer
// for every buffer that was filled and NULL otherwise.  Nothing here is to be given to zh_free.
struct IntoScope {
  zh_ctx* ctx;
  std::vector<void*> ptrs;
  IntoScope(zh_ctx* c, void** dsts, const size_t* caps, size_t n) : ctx(c), ptrs(dsts, dsts + n) {
    ctx->into_ptrs = ptrs.data();
    ctx->into_caps = caps;
    ctx->into_base = dsts;
  }
  ~IntoScope() { ctx->into_ptrs = nullptr, ctx->into_caps = nullptr, ctx->into_base = nullptr; }
};
extern "C" int zh_compress_batch_into(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                                      int level, int data_format, void** dsts, const size_t* caps,
                                      size_t* dst_lens, int32_t* statuses) {
  if (!ctx || (n && (!dsts || !caps))) return ZH_ERR_ARGUMENT;
  IntoScope scope(ctx, dsts, caps, n);
  return compress_batch_impl(ctx, srcs, lens, n, level, data_format, dsts, dst_lens, statuses, nullptr);
}
extern "C" int zh_compress_batch_crc32(zh_ctx* ctx, const void* const* srcs, const size_t* lens,
                                       size_t n, int level, int data_format, void** dsts,
                                       size_t* dst_lens, int32_t* statuses, uint32_t* crcs) {
  if (!crcs && n) return ZH_ERR_ARGUMENT;
  return compress_batch_impl(ctx, srcs, lens, n, level, data_format, dsts, dst_lens, statuses, crcs);
}

// Which container will the device see?  (zippy.nim:108-125, sizing only)
static int host_detect(const uint8_t* s, size_t len, int fmt) {
  if (fmt != ZH_DF_DETECT) return fmt;
  if (len > 18 && s[0] == 31 && s[1] == 139 && s[2] == 8 && (s[3] & 0xe0) == 0) return ZH_DF_GZIP;
  if (len > 6 && (s[0] & 0x0f) == 8 && (s[0] >> 4) <= 7 && (((unsigned)s[0] * 256u) + s[1]) % 31u == 0)
    return ZH_DF_ZLIB;
  return ZH_DF_DETECT;
}

// size_hints: expected output sizes (ZIP central directory, gzip.nim:72-76 trustSize): they
// replace the sizing pass of streams that carry no size; a stream that outgrows its hint falls
// back to the deflate expansion bound.  crcs: CRC-32 of every output (whatever the container).
// The uncompress counterpart of compress_batch_pipelined, for batches whose output sizes are all
// known up front (gzip members: ISIZE; ZIP entries: the central directory): groups of about
// ZH_PIPE_GROUP bytes of OUTPUT take turns, so that one group's kernels run while the group
// before it goes home and the next one comes in.  kPipeFallback: run the batch the plain way
// (does not split, no memory for the second set of buffers, or a stream outgrew its promise --
// a member of 4 GiB and more, or a damaged one -- which the plain path knows how to retry).
static int uncompress_batch_pipelined(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                                      int data_format, const std::vector<uint64_t>& cap, void** dsts,
                                      size_t* dst_lens, int32_t* statuses, uint32_t* crcs) {
  uint64_t out_total = 0;
  for (size_t i = 0; i < n; i++) out_total += cap[i];
  const uint64_t group_bytes = std::max<uint64_t>(pipe_group_bytes(ctx), out_total / 16);
  std::vector<size_t> cut{0};
  uint64_t acc = 0;
  for (size_t i = 0; i < n; i++) {
    acc += cap[i];
    if (acc >= group_bytes) {
      cut.push_back(i + 1);
      acc = 0;
    }
  }
  if (cut.back() != n) cut.push_back(n);
  const size_t G = cut.size() - 1;
  if (G < 2) return kPipeFallback;
  if (!ctx->copy_stream) ZH_HIP(ctx, hipStreamCreate(&ctx->copy_stream));
  hipStream_t cs = ctx->copy_stream, ks = ctx->stream;
  Trace tr;
  std::vector<PipeGroup> gs(G);
  int st;
  // Device memory is bounded by two groups, not by the batch: the groups take turns in TWO sets of
  // source / output / pack buffers (group g uses set g % 2: by the time group g + 2 touches a buffer
  // of the set, group g's last use of it lies before it on the same stream or has been waited for on
  // the host -- see the loop below), and ONE token pool serves every group (scratch of a run, and the
  // runs follow each other on `ks`).  Everything is allocated before the pipeline starts:
  // hipMalloc / hipFree in the middle would serialise it.
  uint64_t set_src[2] = {0, 0}, set_dst[2] = {0, 0};
  for (size_t g = 0; g < G; g++) {
    PipeGroup& q = gs[g];
    q.i0 = cut[g];
    q.n = cut[g + 1] - cut[g];
    q.src_total = layout_slices(lens + q.i0, q.n, q.soff, q.slen);
    q.doff.resize(q.n);
    q.dcap.resize(q.n);
    for (size_t i = 0; i < q.n; i++) {
      q.doff[i] = q.dst_total;
      q.dcap[i] = cap[q.i0 + i];
      q.dst_total += (q.dcap[i] + 255) & ~(uint64_t)255;
    }
    set_src[g & 1] = std::max(set_src[g & 1], q.src_total);
    set_dst[g & 1] = std::max(set_dst[g & 1], q.dst_total);
  }
  DevBuf b_src[2], b_dst[2], b_pack[2];
  for (int k = 0; k < 2; k++)
    if (dev_alloc(ctx, b_src[k], set_src[k] + 256) != hipSuccess ||
        dev_alloc(ctx, b_dst[k], set_dst[k] + 256) != hipSuccess ||
        dev_alloc(ctx, b_pack[k], set_dst[k] + 256) != hipSuccess) {
      (void)hipGetLastError();
      return kPipeFallback;
    }
  uint64_t tok_words = 0;
  for (size_t g = 0; g < G; g++) {
    PipeGroup& q = gs[g];
    ZH_HIP(ctx, hipEventCreate(&q.uploaded));
    ZH_HIP(ctx, hipEventCreate(&q.packed));
    st = zh_plan_uncompress(ctx, q.n, q.soff.data(), q.slen.data(), q.doff.data(), q.dcap.data(), data_format,
                            &q.pg.p);
    if (st == ZH_ERR_NOMEM) return kPipeFallback;
    if (st) return st;
    tok_words = std::max(tok_words, q.pg.p->tok_words);
    if (crcs) zh_plan_request_crc32(q.pg.p, 1);
  }
  DevBuf b_tok;
  if (inflate_split_enabled(ctx) && tok_words) {
    // (a pool that cannot be had leaves the plans to their own devices: plan_token_pool notes the fallback)
    if (dev_alloc(ctx, b_tok, tok_words * 4) != hipSuccess) (void)hipGetLastError();
    for (size_t g = 0; g < G; g++) {
      if (b_tok.p) plan_lend_token_pool(gs[g].pg.p, (uint32_t*)b_tok.p, tok_words);
      (void)plan_token_pool(gs[g].pg.p);  // (now, not in the middle of the pipeline)
    }
  }
  ZH_HIP(ctx, hipStreamSynchronize(ks));  // the plans' descriptors are in place
  auto up = [&](size_t g) -> int {
    PipeGroup& q = gs[g];
    int e = upload_slices(ctx, cs, srcs + q.i0, q.soff, q.slen, q.src_total, b_src[g & 1].p);
    if (e) return e;
    ZH_HIP(ctx, hipEventRecord(q.uploaded, cs));
    return ZH_OK;
  };
  auto run = [&](size_t g) -> int {
    PipeGroup& q = gs[g];
    ZH_HIP(ctx, hipStreamWaitEvent(ks, q.uploaded, 0));
    return zh_plan_run(q.pg.p, b_src[g & 1].p, b_dst[g & 1].p);
  };
  auto give_up = [&](int code) -> int {  // nothing is handed out from a failed call
    (void)hipStreamSynchronize(cs);
    (void)hipStreamSynchronize(ks);
    for (int k = 0; k < 2; k++) ctx->pin_busy[k] = false;
    for (size_t i = 0; i < n; i++) {
      if (!ctx->into_ptrs) free(dsts[i]);  // (zh_*_batch_into: the buffers are the caller's)
      dsts[i] = nullptr;
      dst_lens[i] = 0;
      statuses[i] = ZH_OK;
    }
    return code;
  };
  if ((st = up(0)) || (st = run(0))) return give_up(st);
  for (size_t g = 0; g < G; g++) {
    PipeGroup& q = gs[g];
    if (g + 1 < G && (st = up(g + 1))) return give_up(st);  // while group g's kernels run
    std::vector<uint64_t> olen(q.n);
    std::vector<int32_t> ost(q.n);
    if ((st = zh_plan_results(q.pg.p, olen.data(), ost.data()))) return give_up(st);
    if (crcs && (st = zh_plan_crc32(q.pg.p, crcs + q.i0))) return give_up(st);
    std::vector<char> take(q.n);
    for (size_t i = 0; i < q.n; i++) {
      if (ost[i] == ZH_ERR_DST_TOO_SMALL) return give_up(kPipeFallback);
      statuses[q.i0 + i] = ost[i];
      take[i] = ost[i] == ZH_OK;
    }
    st = download_pack(ctx, ks, q.dl, b_dst[g & 1].p, q.n, q.doff, olen, take, b_pack[g & 1].p, dsts + q.i0,
                       dst_lens + q.i0, statuses + q.i0);
    if (st) return give_up(st);
    if (hipEventRecord(q.packed, ks) != hipSuccess) return give_up(ZH_ERR_DEVICE);
    if (g + 1 < G && (st = run(g + 1))) return give_up(st);  // next kernels behind the pack
    if (hipStreamWaitEvent(cs, q.packed, 0) != hipSuccess) return give_up(ZH_ERR_DEVICE);
    if ((st = download_fetch(ctx, cs, q.dl, dsts + q.i0))) return give_up(st);
  }
  ZH_HIP(ctx, hipStreamSynchronize(cs));
  tr.mark(ctx, "uncompress: pipelined groups");
  return ZH_OK;
}

// hints_are_caps: the hints are capacities of buffers of the caller's (zh_uncompress_batch_into), not promised
// sizes: a stream that outgrows its hint takes the sizing pass (its size is all that is reported then) instead of
// a second decode at the 1032 x expansion bound.
static int uncompress_batch_impl(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                                 int data_format, const uint64_t* size_hints, void** dsts,
                                 size_t* dst_lens, int32_t* statuses, uint32_t* crcs, bool hints_are_caps = false) {
  if (!ctx || (n && (!srcs || !lens || !dsts || !dst_lens || !statuses))) return ZH_ERR_ARGUMENT;
  for (size_t i = 0; i < n; i++) {
    dsts[i] = nullptr;
    dst_lens[i] = 0;
    statuses[i] = ZH_OK;
  }
  if (data_format < ZH_DF_DETECT || data_format > ZH_DF_DEFLATE) {
    for (size_t i = 0; i < n; i++) statuses[i] = ZH_ERR_INVALID_FORMAT;
    return ZH_ERR_INVALID_FORMAT;
  }
  if (!n) return ZH_OK;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf d_src;
  std::vector<uint64_t> soff, slen;
  Trace tr;
  int st;

  // Output sizes: gzip members carry ISIZE (gzip.nim:64-66, trusted only as a capacity hint and
  // verified afterwards); zlib / raw streams carry nothing: they get a guess (4x their size,
  // enough for most data) and, if they outgrow it, a sizing pass (count only) and a second decode.
  std::vector<uint64_t> cap(n, 0);
  std::vector<char> guessed(n, 0), active(n, 1), hinted(n, 0);
  for (size_t i = 0; i < n; i++) {
    const uint8_t* s8 = (const uint8_t*)srcs[i];
    const int f = host_detect(s8, lens[i], data_format);
    const uint64_t max_out = (uint64_t)lens[i] * 1032 + 64;  // deflate cannot expand further
    if (f == ZH_DF_GZIP && lens[i] >= 18) {
      const uint8_t* t = s8 + lens[i] - 4;
      const uint64_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint64_t)t[3] << 24);
      cap[i] = std::min(isize, max_out);
    } else if (f == ZH_DF_ZLIB || f == ZH_DF_DEFLATE) {
      if (size_hints) {
        cap[i] = std::min<uint64_t>(size_hints[i], max_out);
        hinted[i] = 1;
      } else {
        guessed[i] = 1;
        cap[i] = std::min<uint64_t>((uint64_t)lens[i] * 4 + 65536, max_out);
      }
    }
  }
  {
    // every size known and a batch worth splitting: pipelined groups
    uint64_t out_total = 0;
    bool known = true;
    for (size_t i = 0; i < n; i++) {
      known = known && !guessed[i];
      out_total += cap[i];
    }
    if (known && out_total >= pipe_min_bytes(ctx)) {
      const int ps = uncompress_batch_pipelined(ctx, srcs, lens, n, data_format, cap, dsts, dst_lens, statuses, crcs);
      if (ps != kPipeFallback) return ps;
    }
  }
  if ((st = upload(ctx, srcs, lens, n, d_src, soff, slen))) return st;
  tr.mark(ctx, "uncompress: upload");
  // pass 1: decode.  Streams that need more room than they were given run again -- after pass 0
  // (sizing of the guessed ones) -- in pass 2; gzip members get the expansion bound there (more
  // data than ISIZE promised: a >= 4 GiB member, ISIZE being mod 2^32, or a corrupt stream).
  // A stream whose outcome is final is handed to later passes with length 0: it costs nothing.
  int pass = 1;
  for (int turn = 0; turn < 3; turn++) {
    std::vector<uint64_t> doff(n), dcap(n), slen_now(n);
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) {
      const bool runs = active[i] && (pass != 0 || guessed[i]);
      slen_now[i] = runs ? slen[i] : 0;
      doff[i] = total;
      dcap[i] = pass == 0 || !runs ? 0 : cap[i];
      total += (dcap[i] + 255) & ~(uint64_t)255;
    }
    DevBuf d_dst;
    if (dev_alloc(ctx, d_dst, total + 256) != hipSuccess) return ZH_ERR_NOMEM;
    PlanGuard pg;
    st = zh_plan_uncompress(ctx, n, soff.data(), slen_now.data(), doff.data(), dcap.data(), data_format, &pg.p);
    if (st) return st;
    tr.mark(ctx, "uncompress: alloc + plan");
    plan_set_count_only(pg.p, pass == 0);
    if (crcs && pass != 0) zh_plan_request_crc32(pg.p, 1);
    st = zh_plan_run(pg.p, d_src.p, d_dst.p);
    if (st) return st;
    std::vector<uint64_t> olen(n);
    std::vector<int32_t> ost(n);
    std::vector<uint32_t> ocrc(crcs ? n : 0);
    st = zh_plan_results(pg.p, olen.data(), ost.data());
    if (st) return st;
    if (crcs && pass != 0 && (st = zh_plan_crc32(pg.p, ocrc.data()))) return st;
    tr.mark(ctx, "uncompress: kernels");
    if (pass == 0) {
      for (size_t i = 0; i < n; i++)
        if (active[i] && guessed[i]) cap[i] = olen[i];
      pass = 2;
      continue;
    }
    bool again = false, size_first = false;
    std::vector<char> take(n, 0);
    for (size_t i = 0; i < n; i++) {
      if (!active[i]) continue;
      statuses[i] = ost[i];
      if (ost[i] == ZH_ERR_DST_TOO_SMALL && pass == 1) {
        if (hints_are_caps && hinted[i]) guessed[i] = 1;  // (only its size is wanted: count, then decode into as much)
        if (guessed[i])
          size_first = true;
        else
          cap[i] = (uint64_t)lens[i] * 1032 + 64;
        again = true;
        continue;
      }
      active[i] = 0;
      // the decoder ran out of room at the expansion bound (or at the size its own sizing pass counted): the
      // stream is not what it claims to be.  (A result that does not fit a buffer of the CALLER's is
      // download_pack's DST_TOO_SMALL below and stays that.)
      if (ost[i] == ZH_ERR_DST_TOO_SMALL) statuses[i] = ZH_ERR_CHECKSUM;
      if (ost[i] != ZH_OK) continue;
      take[i] = 1;
      if (crcs) crcs[i] = ocrc[i];
    }
    if ((st = download(ctx, d_dst.p, n, doff, olen, take, dsts, dst_lens, statuses))) return st;
    tr.mark(ctx, "uncompress: download");
    if (!again || pass == 2) break;
    pass = size_first ? 0 : 2;
  }
  return ZH_OK;
}

extern "C" int zh_uncompress_batch(zh_ctx* ctx, const void* const* srcs, const size_t* lens,
                                   size_t n, int data_format, void** dsts, size_t* dst_lens,
                                   int32_t* statuses) {
  return uncompress_batch_impl(ctx, srcs, lens, n, data_format, nullptr, dsts, dst_lens, statuses, nullptr);
}
extern "C" int zh_uncompress_batch_into(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                                        int data_format, void** dsts, const size_t* caps, size_t* dst_lens,
                                        int32_t* statuses) {
  if (!ctx || (n && (!dsts || !caps))) return ZH_ERR_ARGUMENT;
  IntoScope scope(ctx, dsts, caps, n);
  // (the capacities double as size hints: a stream without a size field is decoded into as much)
  std::vector<uint64_t> hints(caps, caps + n);
  return uncompress_batch_impl(ctx, srcs, lens, n, data_format, hints.data(), dsts, dst_lens, statuses, nullptr, true);
}
extern "C" int zh_uncompress_batch_sized(zh_ctx* ctx, const void* const* srcs, const size_t* lens,
                                         size_t n, int data_format, const uint64_t* size_hints,
                                         void** dsts, size_t* dst_lens, int32_t* statuses,
                                         uint32_t* crcs) {
  return uncompress_batch_impl(ctx, srcs, lens, n, data_format, size_hints, dsts, dst_lens, statuses, crcs);
}

// ---- one batch over several contexts (= several GPUs): contiguous index ranges, one host
// thread per context, no exchange between them (a buffer is a pure function of itself,
// zippy.nim:11-16).  Range r of n over k: the first n % k ranges get one buffer more -- the
// same split as zippy_amd/sharding.py shard_range.
extern "C" int zh_device_count(void) {
  int count = 0;
  return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

template <class Fn>
static int run_sharded(zh_ctx* const* ctxs, size_t n_ctx, size_t n, Fn&& fn) {
  if (!ctxs || !n_ctx) return ZH_ERR_ARGUMENT;
  for (size_t r = 0; r < n_ctx; r++) {
    if (!ctxs[r]) return ZH_ERR_ARGUMENT;
    for (size_t q = 0; q < r; q++)
      if (ctxs[q] == ctxs[r]) return ZH_ERR_ARGUMENT;  // a context serves one thread at a time
  }
  std::vector<int> rc(n_ctx, ZH_OK);
  auto shard = [&](size_t r) {
    const size_t base = n / n_ctx, extra = n % n_ctx;
    const size_t lo = r * base + std::min(r, extra), cnt = base + (r < extra ? 1 : 0);
    if (cnt) rc[r] = fn(ctxs[r], lo, cnt);
  };
#ifdef ZH_EMU
  for (size_t r = 0; r < n_ctx; r++) shard(r);  // (the emulator's fibers live on one thread)
#else
  std::vector<std::thread> th;
  for (size_t r = 1; r < n_ctx; r++) th.emplace_back(shard, r);
  shard(0);
  for (auto& t : th) t.join();
#endif
  for (int v : rc)
    if (v != ZH_OK) return v;
  return ZH_OK;
}

extern "C" int zh_compress_batch_multi(zh_ctx* const* ctxs, size_t n_ctx, const void* const* srcs,
                                       const size_t* lens, size_t n, int level, int data_format,
                                       void** dsts, size_t* dst_lens, int32_t* statuses) {
  if (n && (!srcs || !lens || !dsts || !dst_lens || !statuses)) return ZH_ERR_ARGUMENT;
  for (size_t i = 0; i < n; i++) dsts[i] = nullptr;
  return run_sharded(ctxs, n_ctx, n, [&](zh_ctx* c, size_t lo, size_t cnt) {
    return zh_compress_batch(c, srcs + lo, lens + lo, cnt, level, data_format, dsts + lo, dst_lens + lo,
                             statuses + lo);
  });
}
extern "C" int zh_uncompress_batch_multi(zh_ctx* const* ctxs, size_t n_ctx, const void* const* srcs,
                                         const size_t* lens, size_t n, int data_format, void** dsts,
                                         size_t* dst_lens, int32_t* statuses) {
  if (n && (!srcs || !lens || !dsts || !dst_lens || !statuses)) return ZH_ERR_ARGUMENT;
  for (size_t i = 0; i < n; i++) dsts[i] = nullptr;
  return run_sharded(ctxs, n_ctx, n, [&](zh_ctx* c, size_t lo, size_t cnt) {
    return zh_uncompress_batch(c, srcs + lo, lens + lo, cnt, data_format, dsts + lo, dst_lens + lo,
                               statuses + lo);
  });
}

extern "C" int zh_compress(zh_ctx* ctx, const void* src, size_t len, int level, int data_format,
                           void** dst, size_t* dst_len) {
  int32_t st = ZH_OK;
  const void* srcs[1] = {src};
  size_t lens[1] = {len};
  int rc = zh_compress_batch(ctx, srcs, lens, 1, level, data_format, dst, dst_len, &st);
  return rc ? rc : st;
}
extern "C" int zh_uncompress(zh_ctx* ctx, const void* src, size_t len, int data_format, void** dst,
                             size_t* dst_len) {
  int32_t st = ZH_OK;
  const void* srcs[1] = {src};
  size_t lens[1] = {len};
  int rc = zh_uncompress_batch(ctx, srcs, lens, 1, data_format, dst, dst_len, &st);
  return rc ? rc : st;
}

extern "C" int zh_compress_blocks(zh_ctx* ctx, const void* src, size_t len, int level, int data_format,
                                  size_t block_bytes, void** dst, size_t* dst_len,
                                  zh_block_entry** index, size_t* n_entries) {
  if (!ctx || !dst || !dst_len || !index || !n_entries || (len && !src)) return ZH_ERR_ARGUMENT;
  *dst = nullptr;
  *dst_len = 0;
  *index = nullptr;
  *n_entries = 0;
  if (level < -2 || level > 9) return ZH_ERR_INVALID_LEVEL;
  if (data_format != ZH_DF_GZIP && data_format != ZH_DF_ZLIB && data_format != ZH_DF_DEFLATE)
    return ZH_ERR_INVALID_FORMAT;
  if (!valid_block_bytes(block_bytes)) return ZH_ERR_ARGUMENT;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  const void* srcs[1] = {src};
  size_t lens[1] = {len};
  DevBuf d_src;
  std::vector<uint64_t> soff, slen;
  int st = upload(ctx, srcs, lens, 1, d_src, soff, slen);
  if (st) return st;
  const size_t nblocks = len / block_bytes + 1;
  for (int attempt = 0; attempt < 2; attempt++) {
    uint64_t doff = 0;
    uint64_t dcap = (attempt == 0 ? typical_cap(len, data_format) : zh_compress_bound(len, data_format)) +
                    1024 * nblocks;
    DevBuf d_dst;
    if (dev_alloc(ctx, d_dst, dcap + 256) != hipSuccess) return ZH_ERR_NOMEM;
    PlanGuard pg;
    st = zh_plan_compress_blocks(ctx, 1, soff.data(), slen.data(), &doff, &dcap, level, data_format,
                                 block_bytes, &pg.p);
    if (st) return st;
    st = zh_plan_run(pg.p, d_src.p, d_dst.p);
    if (st) return st;
    uint64_t olen = 0;
    int32_t ost = ZH_OK;
    st = zh_plan_results(pg.p, &olen, &ost);
    if (st) return st;
    if (ost == ZH_ERR_DST_TOO_SMALL && attempt == 0) continue;
    if (ost != ZH_OK) return ost;
    st = zh_plan_block_index(pg.p, 0, index, n_entries);
    if (st) return st;
    int32_t dst_st = ZH_OK;
    st = download(ctx, d_dst.p, 1, {doff}, {olen}, {1}, dst, dst_len, &dst_st);
    if (st || dst_st) {
      free(*index);
      free(*dst);
      *index = nullptr;
      *dst = nullptr;
      *n_entries = 0;
      *dst_len = 0;
      return st ? st : dst_st;
    }
    return ZH_OK;
  }
  return ZH_ERR_DST_TOO_SMALL;
}

extern "C" int zh_uncompress_indexed(zh_ctx* ctx, const void* src, size_t len, int data_format,
                                     const zh_block_entry* index, size_t n_entries, void** dst,
                                     size_t* dst_len) {
  if (!ctx || !dst || !dst_len || !index || n_entries < 2 || (len && !src)) return ZH_ERR_ARGUMENT;
  *dst = nullptr;
  *dst_len = 0;
  if (data_format < ZH_DF_DETECT || data_format > ZH_DF_DEFLATE) return ZH_ERR_INVALID_FORMAT;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  const void* srcs[1] = {src};
  size_t lens[1] = {len};
  DevBuf d_src;
  std::vector<uint64_t> soff, slen;
  int st = upload(ctx, srcs, lens, 1, d_src, soff, slen);
  if (st) return st;
  const uint64_t total = index[n_entries - 1].out_off;
  if (total > (uint64_t)len * 1032 + 64) return ZH_ERR_INVALID_BUFFER;  // deflate cannot expand further
  DevBuf d_dst;
  if (dev_alloc(ctx, d_dst, total + 256) != hipSuccess) return ZH_ERR_NOMEM;
  PlanGuard pg;
  st = zh_plan_uncompress_indexed(ctx, soff[0], slen[0], 0, total, data_format, index, n_entries, &pg.p);
  if (st) return st == ZH_ERR_ARGUMENT ? ZH_ERR_INVALID_BUFFER : st;
  st = zh_plan_run(pg.p, d_src.p, d_dst.p);
  if (st) return st;
  uint64_t olen = 0;
  int32_t ost = ZH_OK;
  st = zh_plan_results(pg.p, &olen, &ost);
  if (st) return st;
  if (ost != ZH_OK) return ost;
  int32_t dst_st = ZH_OK;
  st = download(ctx, d_dst.p, 1, {0}, {olen}, {1}, dst, dst_len, &dst_st);
  if (st || dst_st) {
    free(*dst);
    *dst = nullptr;
    *dst_len = 0;
    return st ? st : dst_st;
  }
  return ZH_OK;
}

// CRC-32 / Adler-32 of n host buffers in one launch pair (pieces of <= 32 KiB, then a fold per buffer).
static int checksum_host(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                         int want_crc, uint32_t* out) {
  if (!ctx || (n && (!srcs || !lens || !out))) return ZH_ERR_ARGUMENT;
  for (size_t i = 0; i < n; i++)
    if (lens[i] && !srcs[i]) return ZH_ERR_ARGUMENT;
  if (!n) return ZH_OK;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf d;
  std::vector<uint64_t> off, l64;
  int st = upload(ctx, srcs, lens, n, d, off, l64);
  if (st) return st;
  std::vector<ZhPieceDesc> pieces;
  std::vector<ZhBufDesc> bufs(n);
  for (size_t i = 0; i < n; i++) {
    ZhBufDesc& b = bufs[i];
    memset(&b, 0, sizeof(b));
    b.src_off = off[i];
    b.src_len = lens[i];
    b.first_piece = (uint32_t)pieces.size();
    for (uint64_t o = 0; o < lens[i]; o += ZH_FRAG_SIZE)
      pieces.push_back(ZhPieceDesc{off[i] + o, (uint32_t)std::min<uint64_t>(lens[i] - o, ZH_FRAG_SIZE), (uint32_t)i, o});
    b.npieces = (uint32_t)pieces.size() - b.first_piece;
  }
  const size_t np = pieces.size();
  if (np >= 0xffffffffull) return ZH_ERR_ARGUMENT;
  Arena ar;
  const size_t o_b = ar.reserve(n * sizeof(ZhBufDesc)), o_p = ar.reserve(np * sizeof(ZhPieceDesc)),
               o_pc = ar.reserve(np * 4), o_pl = ar.reserve(np * 4), o_pa = ar.reserve(np * 4),
               o_oc = ar.reserve(n * 4), o_oa = ar.reserve(n * 4);
  ar.reserve(256);
  DevBuf scratch;
  if (dev_alloc(ctx, scratch, ar.size) != hipSuccess) return ZH_ERR_NOMEM;
  uint8_t* base = scratch.p;
  hipStream_t s = ctx->stream;
  ZH_HIP(ctx, hipMemcpyAsync(base + o_b, bufs.data(), n * sizeof(ZhBufDesc), hipMemcpyHostToDevice, s));
  if (np) ZH_HIP(ctx, hipMemcpyAsync(base + o_p, pieces.data(), np * sizeof(ZhPieceDesc), hipMemcpyHostToDevice, s));
  zh_launch_checksum_pieces(s, ctx->cktabs, d.p, carve<ZhPieceDesc>(base, o_p), (uint32_t)np, nullptr, want_crc,
                            !want_crc, carve<uint32_t>(base, o_pc), carve<uint32_t>(base, o_pa),
                            carve<uint32_t>(base, o_pl));
  zh_launch_checksum_combine(s, carve<ZhBufDesc>(base, o_b), (uint32_t)n, carve<uint32_t>(base, o_pc),
                             carve<uint32_t>(base, o_pa), carve<uint32_t>(base, o_pl), want_crc, !want_crc,
                             carve<uint32_t>(base, o_oc), carve<uint32_t>(base, o_oa));
  ZH_HIP(ctx, hipMemcpyAsync(out, base + (want_crc ? o_oc : o_oa), n * 4, hipMemcpyDeviceToHost, s));
  ZH_HIP(ctx, hipStreamSynchronize(s));
  return ZH_OK;
}
extern "C" int zh_crc32_batch(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n,
                              uint32_t* out) {
  return checksum_host(ctx, srcs, lens, n, 1, out);
}
extern "C" int zh_crc32(zh_ctx* ctx, const void* src, size_t len, uint32_t* out) {
  const void* srcs[1] = {src};
  size_t lens[1] = {len};
  return checksum_host(ctx, srcs, lens, 1, 1, out);
}
extern "C" int zh_adler32(zh_ctx* ctx, const void* src, size_t len, uint32_t* out) {
  const void* srcs[1] = {src};
  size_t lens[1] = {len};
  return checksum_host(ctx, srcs, lens, 1, 0, out);
}

// ---------------------------------------------------------------------------
// parity introspection: device parse -> reference token stream
// ---------------------------------------------------------------------------
// Debug hook: one prefix code from a histogram -- contract 0: the replay of deflate.nim:13-151 huffmanCodes
// (byte-identical mode), 1: the wave-parallel optimal builder of contract mode.  codes / lens: num_freq + 2 entries.
extern "C" int zh_debug_huffman(zh_ctx* ctx, const uint32_t* freq, int num_freq, int min_codes, int limit, int contract,
                                uint16_t* codes, uint8_t* lens, int* num_codes) {
  if (!ctx || !freq || !codes || !lens || !num_codes || num_freq < 1 || num_freq > 288 || min_codes < 1 || min_codes > 287 ||
      limit < 1 || limit > 15)
    return ZH_ERR_ARGUMENT;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf d;
  const size_t o_codes = 2048, o_lens = 4096, o_n = 5120;
  if (dev_alloc(ctx, d, 8192) != hipSuccess) return ZH_ERR_NOMEM;
  hipStream_t s = ctx->stream;
  ZH_HIP(ctx, hipMemcpyAsync(d.p, freq, (size_t)num_freq * 4, hipMemcpyHostToDevice, s));
  zh_launch_huffman_probe(s, reinterpret_cast<const uint32_t*>(d.p), num_freq, min_codes, limit, contract,
                          reinterpret_cast<uint16_t*>(d.p + o_codes), d.p + o_lens, reinterpret_cast<int*>(d.p + o_n));
  ZH_HIP(ctx, hipGetLastError());
  int n = 0;
  ZH_HIP(ctx, hipMemcpyAsync(&n, d.p + o_n, 4, hipMemcpyDeviceToHost, s));
  ZH_HIP(ctx, hipStreamSynchronize(s));
  if (n < 0 || n > 288) return ZH_ERR_COMPRESS_INTERNAL;
  ZH_HIP(ctx, hipMemcpyAsync(codes, d.p + o_codes, (size_t)n * 2, hipMemcpyDeviceToHost, s));
  ZH_HIP(ctx, hipMemcpyAsync(lens, d.p + o_lens, (size_t)n, hipMemcpyDeviceToHost, s));
  ZH_HIP(ctx, hipStreamSynchronize(s));
  *num_codes = n;
  return ZH_OK;
}

extern "C" int zh_debug_tokens(zh_ctx* ctx, const void* src, size_t len, int level,
                               uint16_t** tokens, size_t* num_tokens) {
  if (!ctx || !tokens || !num_tokens || (len && !src)) return ZH_ERR_ARGUMENT;
  if (level < -2 || level > 9 || level == 0) return ZH_ERR_INVALID_LEVEL;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  const void* srcs[1] = {src};
  size_t lens[1] = {len};
  DevBuf d_src;
  std::vector<uint64_t> soff, slen;
  int st = upload(ctx, srcs, lens, 1, d_src, soff, slen);
  if (st) return st;
  uint64_t doff = 0, dcap = 0;
  PlanGuard pg;
  st = zh_plan_compress(ctx, 1, soff.data(), slen.data(), &doff, &dcap, level, ZH_DF_DEFLATE, &pg.p);
  if (st) return st;
  zh_plan* p = pg.p;
  hipStream_t s = ctx->stream;
  const ZhCompressArgs& a = p->ca;
  if (level == 1 || level == -2) {
    zh_launch_l1_match(s, d_src.p, a, level == -2, p->l1_tables, p->l1_counter);
  } else {
    const int* cfg = kChainConfig[level == -1 ? 6 : level];
    for (const auto& r : p->chain_ranges) {
      ZhCompressArgs ar = a;
      ar.first_block = r.b0;
      ar.nblocks = r.nb;
      ar.first_frag = r.f0;
      ar.nfrags = r.nf;
      zh_launch_chain_prev(s, d_src.p, ar, p->head_scratch, p->chain_prev, p->chain_best);
      zh_launch_chain_search(s, d_src.p, ar, cfg[0], cfg[1], cfg[2], p->chain_prev, p->chain_best);
      zh_launch_chain_select(s, d_src.p, ar, cfg[0], cfg[1], cfg[2], p->chain_prev, p->chain_best);
    }
  }
  const size_t nf = a.nfrags;
  std::vector<uint32_t> nmatch(nf);
  std::vector<uint16_t> mpos(nf * ZH_MAX_MATCHES_PER_FRAG), mlen(mpos.size()), moff(mpos.size());
  if (nf) {
    ZH_HIP(ctx, hipMemcpyAsync(nmatch.data(), a.f_nmatch, nf * 4, hipMemcpyDeviceToHost, s));
    ZH_HIP(ctx, hipMemcpyAsync(mpos.data(), a.m_pos, mpos.size() * 2, hipMemcpyDeviceToHost, s));
    ZH_HIP(ctx, hipMemcpyAsync(mlen.data(), a.m_len, mpos.size() * 2, hipMemcpyDeviceToHost, s));
    ZH_HIP(ctx, hipMemcpyAsync(moff.data(), a.m_off, mpos.size() * 2, hipMemcpyDeviceToHost, s));
  }
  ZH_HIP(ctx, hipStreamSynchronize(s));

  std::vector<uint16_t> out;
  auto add_literals = [&](uint64_t count) {  // snappy.nim:39-47
    while (count > 0) {
      const uint64_t added = std::min<uint64_t>(count, 32767);
      out.push_back((uint16_t)added);
      count -= added;
    }
  };
  // Level 1 closes its literal run at every fragment end (emitRemainder,
  // snappy.nim:66-68); the chain levels and -2 run literals across the block.
  const bool per_fragment = level == 1;
  size_t f = 0;
  for (uint64_t bstart = 0; bstart < len || (len == 0 && bstart == 0); bstart += ZH_BLOCK_SIZE) {
    const uint64_t blen = std::min<uint64_t>(len - bstart, ZH_BLOCK_SIZE);
    uint64_t run = 0;  // pending literals
    uint64_t covered_until = 0;  // block-relative end of the last match
    for (uint64_t o = 0; o < blen; o += ZH_FRAG_SIZE, f++) {
      const uint64_t flen = std::min<uint64_t>(blen - o, ZH_FRAG_SIZE);
      uint64_t cursor = std::max<uint64_t>(o, covered_until);
      for (uint32_t m = 0; m < nmatch[f]; m++) {
        const size_t k = f * ZH_MAX_MATCHES_PER_FRAG + m;
        const uint64_t mp = o + mpos[k];
        run += mp - cursor;
        add_literals(run);
        run = 0;
        const uint32_t l = mlen[k], off = moff[k];
        uint32_t li = 0;
        {
          static const uint16_t base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                            31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
          for (int i = 0; i < 29; i++)
            if (base[i] <= l) li = i;
          if (l == 258) li = 28;
        }
        const uint32_t di = zh_dist_code(off);
        out.push_back((uint16_t)(0x8000u | (li << 8) | di));
        out.push_back((uint16_t)off);
        out.push_back((uint16_t)l);
        cursor = mp + l;
        covered_until = cursor;
      }
      const uint64_t fend = o + flen;
      if (cursor < fend) run += fend - cursor;
      if (per_fragment) {
        add_literals(run);
        run = 0;
      }
    }
    add_literals(run);
    if (len == 0) break;
  }
  *num_tokens = out.size();
  *tokens = (uint16_t*)malloc(out.size() * 2 + 2);
  if (!*tokens) return ZH_ERR_NOMEM;
  memcpy(*tokens, out.data(), out.size() * 2);
  return ZH_OK;
}

#ifdef ZH_KPROF
// tuning builds only (zh_kprof.h): phase timers summed by the kernels
__device__ unsigned long long zh_kprof_slots[ZH_KPROF_SLOTS];
extern "C" int zh_kprof_read(unsigned long long* out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return ZH_ERR_DEVICE;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(zh_kprof_slots), sizeof(zh_kprof_slots)) != hipSuccess)
    return ZH_ERR_DEVICE;
  if (reset) {
    static const unsigned long long zeros[ZH_KPROF_SLOTS] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(zh_kprof_slots), zeros, sizeof(zeros)) != hipSuccess) return ZH_ERR_DEVICE;
  }
  return ZH_OK;
}
#endif
