// Host side of the C ABI: host-buffer batches -- staging through pinned chunks, pipelined groups, results into
// fresh or caller-owned buffers, one batch over several contexts (GPUs).
#include "zh_host.h"

// ---------------------------------------------------------------------------
// host-buffer API
// ---------------------------------------------------------------------------
namespace {
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

int zhh_upload(zh_ctx* ctx, const void* const* srcs, const size_t* lens, size_t n, DevBuf& dev, std::vector<uint64_t>& off,
               std::vector<uint64_t>& len64) {
  return upload(ctx, srcs, lens, n, dev, off, len64);
}
int zhh_download(zh_ctx* ctx, const uint8_t* d_dst, size_t n, const std::vector<uint64_t>& doff,
                 const std::vector<uint64_t>& olen, const std::vector<char>& take, void** dsts, size_t* dst_lens,
                 int32_t* statuses) {
  return download(ctx, d_dst, n, doff, olen, take, dsts, dst_lens, statuses);
}
