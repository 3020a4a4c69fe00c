// Host side of the C ABI (include/zippy_hip.h): contexts, the device block cache, status strings, size bounds.
#include "zh_host.h"

// Device allocations of a context.  hipMalloc / hipFree cost a call of one small buffer more than
// its kernels (hipFree also waits for the device), so freed blocks are kept -- up to ZH_DEV_CACHE_MB
// (default 2048; 0: none), blocks of up to half of that -- and handed out again to requests they fit
// without wasting more than half.  Everything a context does is ordered on its stream, so a block
// may be reused as soon as it has been given back.  ZH_DEV_CACHE_POISON=1 fills every block handed
// out (test aid: nothing may rely on fresh memory being zero).
hipError_t ctx_malloc(zh_ctx* ctx, void** out, size_t bytes) {
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
void ctx_free(zh_ctx* ctx, void* p) {
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
  if (!zh_chain_lds_order_ok(device, c->stream)) {
    c->chain_links_serial = true;
    c->last_error = "this device does not serve the lanes of an LDS atomic in ascending order (or could not be asked): "
                    "levels -1, 2..9 build their chain links with the in-order kernels";
  }
#ifdef ZH_XCHECK  // (the test build: the in-order kernels as a cross-check of the class-sorted ones)
  if (const char* e = getenv("ZH_CHAIN_PREV"))
    if (strcmp(e, "serial") == 0) c->chain_links_serial = true;
#endif
  {  // zh_debug_segment_stats' two counters: here, zeroed on the context's stream, not lazily inside some plan's run
    void* q = nullptr;
    if (ctx_malloc(c, &q, 16) == hipSuccess && hipMemsetAsync(q, 0, 16, c->stream) == hipSuccess) {
      c->d_seg_stats = static_cast<uint64_t*>(q);
    } else {
      (void)hipGetLastError();  // (the counters are a test aid: without them they read 0)
    }
  }
  *out = c;
  return ZH_OK;
}
extern "C" int zh_chain_links_parallel(zh_ctx* ctx) {
  return ctx && !ctx->chain_links_serial ? 1 : 0;
}

extern "C" void zh_destroy(zh_ctx* ctx) {
  if (!ctx) return;
  for (int k = 0; k < 2; k++) {
    if (ctx->pin_ev[k]) (void)hipEventDestroy(ctx->pin_ev[k]);
    if (ctx->pin[k]) (void)hipHostFree(ctx->pin[k]);
  }
  for (auto& b : ctx->dev_blocks) (void)hipFree(b.p);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
  if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
  if (ctx->aux_join) (void)hipEventDestroy(ctx->aux_join);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}
extern "C" const char* zh_last_error(zh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
extern "C" void* zh_stream(zh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" void zh_set_gzip_fname_len(zh_ctx* ctx, int k) {
  if (ctx) ctx->fname_len = k > 25 ? 25 : k;
}
extern "C" void zh_free(void* p) { free(p); }

// Device memory for callers that have no HIP binding of their own (a Nim / cgo / C shim that uses the plans):
// blocks from the context's cache, copies on the context's stream and waited for.
extern "C" int zh_device_malloc(zh_ctx* ctx, size_t bytes, void** d_out) {
  if (!ctx || !d_out) return ZH_ERR_ARGUMENT;
  *d_out = nullptr;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  ZH_HIP(ctx, ctx_malloc(ctx, d_out, bytes ? bytes : 1));
  return ZH_OK;
}
extern "C" void zh_device_free(zh_ctx* ctx, void* d) {
  if (!ctx || !d) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);  // (kernels of the context may still be reading it)
  if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);  // (a run joins it; a run that failed half-way too)
  ctx_free(ctx, d);
}
extern "C" int zh_device_upload(zh_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
  if (!ctx || (bytes && (!d_dst || !src))) return ZH_ERR_ARGUMENT;
  if (!bytes) return ZH_OK;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  ZH_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  ZH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZH_OK;
}
extern "C" int zh_device_download(zh_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !d_src))) return ZH_ERR_ARGUMENT;
  if (!bytes) return ZH_OK;
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  ZH_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZH_OK;
}
extern "C" void zh_set_host_pipeline(zh_ctx* ctx, size_t min_batch_bytes, size_t group_bytes) {
  if (!ctx) return;
  ctx->pipe_min = min_batch_bytes;
  ctx->pipe_group = group_bytes;
}

// Worst case of the reference's encoder: it has no "stored if larger" fallback, so
// a block that escapes the 98 % literal test can still use up to 15 bits per
// literal; every block adds a <= 1 KiB header.
extern "C" size_t zh_compress_bound(size_t len, int data_format) {
  size_t nblocks = (len + ZH_BLOCK_SIZE - 1) / ZH_BLOCK_SIZE;
  if (!nblocks) nblocks = 1;
  return len * 2 + 1024 * nblocks + 5 * (len / ZH_STORED_MAX + 1) + container_overhead(data_format) + 64;
}
