// Host side of the C ABI: compress plans -- the descriptors of a batch (buffers, deflate blocks, fragments,
// checksum pieces) and its scratch in one arena --, and the block index of the block-parallel form.
#include "zh_host.h"

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
  uint64_t cap_max = 0;
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
    cap_max = std::max(cap_max, b.dst_cap);
  }
  if (blocks.size() >= 0xffffffffull || frags.size() >= 0xffffffffull) return ZH_ERR_ARGUMENT;

  zh_plan* p = new zh_plan;
  p->ctx = ctx;
  p->is_compress = true;
  p->n = n;
  p->level = level;
  p->fmt = data_format;
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
  // (the exact BestSpeed parse and level -2 hand the emission their coverage bitmap: 4 KiB a fragment)
  const bool cover_out = level == 1 || level == -2;
  const size_t o_fcov = ar.reserve(cover_out ? nf * 4096 : 0);
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
  const size_t l1tab_bytes = level == 1 ? std::max(std::min<size_t>(nf, zh_l1_table_slots()) * 32768,
                                                   std::min<size_t>(nf, zh_l1p_slots()) * 131072)
                                        : 0;
  // ZH_L1_POOL=uncached / fine (measurement): the pool as an allocation of its own in memory the L2 does
  // not allocate lines for (hipDeviceMallocUncached) / fine-grained memory, instead of a range of the arena
  const int l1_pool_mode = [] {
    const char* e = getenv("ZH_L1_POOL");
    return !e ? 0 : strcmp(e, "uncached") == 0 ? 1 : strcmp(e, "fine") == 0 ? 2 : 0;
  }();
  const bool l1_pool_own = l1_pool_mode != 0 && l1tab_bytes != 0;
  const size_t o_l1tab = ar.reserve(l1_pool_own ? 0 : l1tab_bytes);
  const size_t o_l1ctr = ar.reserve(256);
  const bool l1 = level == 1 || level == -2;
  const size_t o_l1cost = ar.reserve(l1 ? nf * 4 : 4), o_l1order = ar.reserve(l1 ? nf * 4 : 4), o_l1hist = ar.reserve(512 + (l1 ? (nf / 256 + 2) * 4 : 0));
  const size_t o_cprev = ar.reserve(chain ? range_frags * (size_t)ZH_FRAG_SIZE * 8 : 0);
  const size_t o_cbest = ar.reserve(chain ? range_frags * (size_t)ZH_FRAG_SIZE * 4 : 0);
  ar.reserve(256);

  if (ctx_malloc(p->ctx, (void**)&p->arena, ar.size) != hipSuccess) {
    ctx->last_error = "hipMalloc(plan arena, " + std::to_string(ar.size) + " bytes)";
    delete p;
    return ZH_ERR_NOMEM;
  }
#ifndef ZH_EMU
  if (l1_pool_own &&
      hipExtMallocWithFlags(&p->l1_pool_own, l1tab_bytes, l1_pool_mode == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    ctx->last_error = "hipExtMallocWithFlags(BestSpeed table pool, " + std::to_string(l1tab_bytes) + " bytes)";
    zh_plan_destroy(p);
    return ZH_ERR_NOMEM;
  }
#else
  if (l1_pool_own && hipMalloc(&p->l1_pool_own, l1tab_bytes) != hipSuccess) {
    zh_plan_destroy(p);
    return ZH_ERR_NOMEM;
  }
#endif
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
  // the checksum beside the emission as well, the trailer by a kernel of its own behind it: worth its launch from
  // 256 MiB of input on (4096 x 1 MiB: 0.25 ms of 73; 1024 x 64 KiB: 0.5 % slower).  ZH_TRAILER_LATE=0 / 1: never / always.
  p->trailer_late = nf >= 8192;
  if (const char* e = getenv("ZH_TRAILER_LATE")) p->trailer_late = strcmp(e, "0") != 0;
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
  a.f_cover = cover_out ? carve<uint32_t>(base, o_fcov) : nullptr;
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
  p->l1_tables = p->l1_pool_own ? static_cast<uint16_t*>(p->l1_pool_own) : carve<uint16_t>(base, o_l1tab);
  p->l1_counter = carve<uint32_t>(base, o_l1ctr);
  if (l1) {
    p->l1_cost = carve<uint32_t>(base, o_l1cost);
    p->l1_order = carve<uint32_t>(base, o_l1order);
    p->l1_hist = carve<uint32_t>(base, o_l1hist);
  }
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
