// Host side of the C ABI: uncompress plans -- stream descriptors, token regions, the segmentation of large
// streams (zh_inflate_seg.hip) --, the indexed (block-parallel) form.
#include "zh_host.h"

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
  // test aids, read here and not on the run path: a found start that is none planted at a stream bit, the chain's story
  if (const char* e = getenv("ZH_SEG_FAKE_START")) p->sg_fake_start = strtoull(e, nullptr, 10);
  p->sg_trace = getenv("ZH_TRACE_SEG") != nullptr;
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
               o_symb = ar.reserve(n * 8), o_start = ar.reserve(ns * 8), o_start2 = ar.reserve(ns * 8), o_end = ar.reserve(ns * 8),
               o_final = ar.reserve(ns * 4), o_sst = ar.reserve(ns * 4), o_sout = ar.reserve(ns * 8),
               o_wlen = ar.reserve(ns * 8), o_valid = ar.reserve(ns * 4), o_prev = ar.reserve(ns * 4),
               o_ostart = ar.reserve(ns * 8), o_sok = ar.reserve(n * 4), o_order = ar.reserve(ns * 4),
               o_nchain = ar.reserve(n * 4), o_repair = ar.reserve(n * 4), o_ordinal = ar.reserve(ns * 4), o_go = ar.reserve(n * 4), o_etoff = ar.reserve(ns * 8), o_etcap = ar.reserve(ns * 8), o_substart = ar.reserve(ns * 8), o_subhdr = ar.reserve(ns * 8),
               o_issub = ar.reserve(ns * 4), o_held = ar.reserve(ns * 8), o_stored = ar.reserve(ns * 8);
  const size_t nfind = find_seg.size();
  const size_t o_fseg = ar.reserve(nfind * 4), o_fbatch = ar.reserve(nfind * 4), o_cn = ar.reserve(nfind * 4),
               o_coff = ar.reserve(nfind * (size_t)kSegFindSlots * 4);
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
  g.start2_bit = carve<uint64_t>(base, o_start2);
  g.stored_bit = carve<uint64_t>(base, o_stored);
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
  g.repair = carve<uint32_t>(base, o_repair);
  g.ordinal = carve<uint32_t>(base, o_ordinal);
  g.go = carve<uint32_t>(base, o_go);
  g.eff_tok_off = carve<uint64_t>(base, o_etoff);
  g.eff_tok_cap = carve<uint64_t>(base, o_etcap);
  g.sub_start = carve<uint64_t>(base, o_substart);
  g.sub_hdr = carve<uint64_t>(base, o_subhdr);
  g.is_sub = carve<uint32_t>(base, o_issub);
  g.held_start = carve<uint64_t>(base, o_held);
  g.nfind = (uint32_t)nfind;
  g.find_seg = carve<uint32_t>(base, o_fseg);
  g.find_batch = carve<uint32_t>(base, o_fbatch);
  g.cand_n = carve<uint32_t>(base, o_cn);
  g.cand_off = carve<uint32_t>(base, o_coff);
  p->sg_sym_count = nsym + 64;
  {  // (a false start a GiB or so: a second round of repairs only pays where the first is likely to leave something)
    uint64_t cut_bytes = 0;
    for (const ZhBufDesc& b : bufs)
      if (large(b)) cut_bytes += b.src_len;
    p->sg_repair_rounds = cut_bytes >= (256ull << 20) ? 2 : 1;
  }
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
  p->half_piece = n >= 2 ? bufs[n / 2].first_piece : 0;
  p->n = n;
  p->fmt = data_format;
  for (const ZhBufDesc& b : bufs) {
    p->src_max_len = std::max<uint64_t>(p->src_max_len, b.src_len);
    p->dst_max_cap = std::max<uint64_t>(p->dst_max_cap, b.dst_cap);
  }
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
  // large batches decode as two halves on two streams (zh_plan_run.hip): from 2048 streams on, so that a half still is
  // a batch for the narrow kernels; ZH_INFLATE_HALVES=<smallest such batch> (0: never; the tests: 4)
  p->halves_min = 2048;
  if (const char* e = getenv("ZH_INFLATE_HALVES")) p->halves_min = (uint32_t)strtoul(e, nullptr, 10);
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
