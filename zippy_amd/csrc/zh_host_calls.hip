// Host side of the C ABI: single-buffer calls, the block-parallel form of one large buffer, checksums, the
// debug hooks.
#include "zh_host.h"

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
  int st = zhh_upload(ctx, srcs, lens, 1, d_src, soff, slen);
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
    st = zhh_download(ctx, d_dst.p, 1, {doff}, {olen}, {1}, dst, dst_len, &dst_st);
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
  int st = zhh_upload(ctx, srcs, lens, 1, d_src, soff, slen);
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
  st = zhh_download(ctx, d_dst.p, 1, {0}, {olen}, {1}, dst, dst_len, &dst_st);
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
  int st = zhh_upload(ctx, srcs, lens, n, d, off, l64);
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
  zh_launch_checksum_combine(s, ctx->cktabs, carve<ZhBufDesc>(base, o_b), (uint32_t)n, carve<uint32_t>(base, o_pc),
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
  // (min_codes <= num_freq: numCodes = max(highest, minCodes) + 1 then fits the num_freq + 2 entries the caller holds;
  // the reference's own calls are 286 / 257, 30 / 2, 19 / 19, deflate.nim:285-290,355)
  if (!ctx || !freq || !codes || !lens || !num_codes || num_freq < 1 || num_freq > 288 || min_codes < 1 ||
      min_codes > num_freq || limit < 1 || limit > 15)
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
  if (n < 0 || n > num_freq + 2) return ZH_ERR_COMPRESS_INTERNAL;
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
  int st = zhh_upload(ctx, srcs, lens, 1, d_src, soff, slen);
  if (st) return st;
  uint64_t doff = 0, dcap = 0;
  PlanGuard pg;
  st = zh_plan_compress(ctx, 1, soff.data(), slen.data(), &doff, &dcap, level, ZH_DF_DEFLATE, &pg.p);
  if (st) return st;
  zh_plan* p = pg.p;
  hipStream_t s = ctx->stream;
  const ZhCompressArgs& a = p->ca;
  if (level == 1 || level == -2) {
    zh_launch_l1_match(s, d_src.p, a, level == -2, p->l1_tables, p->l1_counter, nullptr, nullptr, nullptr, 0);
  } else {
    const int* cfg = kChainConfig[level == -1 ? 6 : level];
    for (const auto& r : p->chain_ranges) {
      ZhCompressArgs ar = a;
      ar.first_block = r.b0;
      ar.nblocks = r.nb;
      ar.first_frag = r.f0;
      ar.nfrags = r.nf;
      zh_launch_chain_prev(s, d_src.p, ar, p->head_scratch, p->chain_prev, p->chain_best, p->ctx->chain_links_serial ? 1 : 0, cfg[0]);
      zh_launch_chain_search(s, d_src.p, ar, cfg[0], cfg[1], cfg[2], p->chain_prev, p->chain_best, p->ctx->chain_links_serial ? 1 : 0);
      zh_launch_chain_select(s, d_src.p, ar, cfg[0], cfg[1], cfg[2], p->chain_prev, p->chain_best, p->ctx->chain_links_serial ? 1 : 0);
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
