// Host side of the C ABI: running a plan -- kernel sequencing on the context's stream, the per-context
// switches, the token pool, per-kernel timing, results.
#include "zh_host.h"

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
  if (p->k_aux_used) {  // the checksum kernels of a compress run, timed on the second stream (beside "zh_huffman_kernel")
    static const char* const aux_names[2] = {"zh_checksum_pieces_kernel", "zh_checksum_combine_kernel"};
    for (int k = 0; k < 2 && cnt < max_entries; k++) {
      float t = 0;
      if (hipEventElapsedTime(&t, p->k_aux[k], p->k_aux[k + 1]) != hipSuccess) break;
      names[cnt] = aux_names[k];
      ms[cnt] = t;
      cnt++;
    }
  }
  if (p->k_half_used) {  // the second half of an uncompress batch, on the second stream beside the first half's writer
    static const char* const half_names[2] = {"zh_inflate_tokens_kernel", "zh_inflate_write_kernel"};
    for (int k = 0; k < 2 && cnt < max_entries; k++) {
      float t = 0;
      if (hipEventElapsedTime(&t, p->k_half[k], p->k_half[k + 1]) != hipSuccess) break;
      names[cnt] = half_names[k];
      ms[cnt] = t;
      cnt++;
    }
  }
  return cnt;
}

extern "C" void zh_plan_destroy(zh_plan* p) {
  if (!p) return;
  // (nothing useful can be done about a failure while tearing down)
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  if (p->ctx->aux_stream) (void)hipStreamSynchronize(p->ctx->aux_stream);  // (joined by every run; cheap when idle)
  if (p->arena) ctx_free(p->ctx, p->arena);
  if (p->l1_pool_own) (void)hipFree(p->l1_pool_own);
  if (p->seg_arena) ctx_free(p->ctx, p->seg_arena);
  if (p->tok_pool && !p->tok_borrowed) ctx_free(p->ctx, p->tok_pool);
  if (p->sg_arena) ctx_free(p->ctx, p->sg_arena);
  if (p->sg_sym) ctx_free(p->ctx, p->sg_sym);
  if (p->sg_windows) ctx_free(p->ctx, p->sg_windows);
  if (p->sg_winsym) ctx_free(p->ctx, p->sg_winsym);
  if (p->unpack_lens) ctx_free(p->ctx, p->unpack_lens);
  for (auto e : p->k_events) (void)hipEventDestroy(e);
  for (auto e : p->k_aux)
    if (e) (void)hipEventDestroy(e);
  for (auto e : p->k_half)
    if (e) (void)hipEventDestroy(e);
  delete p;
}


// ZH_INFLATE=serial keeps every stream on the two-wave serial decoder (zh_inflate.hip); the
// default decodes a stream's Huffman codes in parallel (zh_inflate_split.hip).  Sizing passes and
// the block-parallel form of one stream always use the serial kernel.
bool inflate_split_enabled(const zh_ctx* ctx) {
  static const bool on = [] {
    const char* e = getenv("ZH_INFLATE");
    return !(e && strcmp(e, "serial") == 0);
  }();
  return ctx->inflate_mode < 0 ? on : ctx->inflate_mode == 0;
}
// The context's second stream and its two events, made when first asked for.  Anything that fails leaves the work in
// line on the context's stream.
static bool second_stream(zh_ctx* ctx) {
  if (!ctx->aux_stream) {
    if (hipStreamCreate(&ctx->aux_stream) != hipSuccess) {
      ctx->aux_stream = nullptr;
      (void)hipGetLastError();
      return false;
    }
  }
  if (!ctx->aux_fork && hipEventCreate(&ctx->aux_fork) != hipSuccess) ctx->aux_fork = nullptr;
  if (!ctx->aux_join && hipEventCreate(&ctx->aux_join) != hipSuccess) ctx->aux_join = nullptr;
  return ctx->aux_fork && ctx->aux_join;
}
// ZH_CHECKSUM_ASIDE=0 keeps a compress run's checksum kernels in line on the context's stream (measurement); it says
// nothing about the uncompress batch's two halves, which have their own switch (ZH_INFLATE_HALVES)
static bool checksum_aside(zh_ctx* ctx) {
  static const bool on = [] {
    const char* e = getenv("ZH_CHECKSUM_ASIDE");
    return !(e && strcmp(e, "0") == 0);
  }();
  return on && second_stream(ctx);
}
// Work that was forked onto the second stream is joined on every way out of a run: an early return (a failed HIP
// call between fork and join) must not leave kernels of the second stream reading the caller's buffers unordered
// with the caller's stream -- the caller may free or reuse them as soon as its own stream has drained.
namespace {
struct AuxJoinGuard {
  zh_ctx* ctx;
  hipStream_t s;
  bool forked = false;
  ~AuxJoinGuard() {
    if (!forked) return;
    if (hipEventRecord(ctx->aux_join, ctx->aux_stream) != hipSuccess ||
        hipStreamWaitEvent(s, ctx->aux_join, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(ctx->aux_stream);
    }
  }
};
}  // namespace
// ZH_L1_ORDER=0: fragments as numbered in every run (measurement: what a plan's first run costs)
static bool l1_longest_first() {
  static const bool on = [] {
    const char* e = getenv("ZH_L1_ORDER");
    return !(e && strcmp(e, "0") == 0);
  }();
  return on;
}
// BestSpeed parse: 0 the reference's (snappy.nim:12-136, byte-identical streams), 1 the parallel
// parse of zh_l1p_match.hip (valid streams of about the same size, not the reference's bytes)
bool l1_parallel(const zh_ctx* ctx) {
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
bool plan_token_pool(zh_plan* p) {
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
void plan_lend_token_pool(zh_plan* p, uint32_t* pool, uint64_t words) {
  if (p->tok_pool || p->tok_failed || !p->tok_words || p->tok_words > words) return;
  p->tok_pool = pool;
  p->tok_borrowed = true;
}

// ZH_TRACE_SEG: why a large stream was not decoded segment-wise (the run waits for the chain kernel and reads its
// verdicts back; a stream whose chain does not hold is decoded by one workgroup, correctly and slowly)
static void seg_trace(zh_plan* p, hipStream_t s) {
  const ZhSegArgs& g = p->sg;
  if (hipStreamSynchronize(s) != hipSuccess) return;
  std::vector<uint32_t> first(g.nstreams + 1), go(g.nstreams), ok(g.nstreams), nchain(g.nstreams);
  (void)hipMemcpy(first.data(), g.first_seg, first.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(go.data(), g.go, go.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(ok.data(), g.stream_ok, ok.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(nchain.data(), g.nchain, nchain.size() * 4, hipMemcpyDeviceToHost);
  {
    std::vector<uint32_t> cn(g.nfind);
    (void)hipMemcpy(cn.data(), g.cand_n, cn.size() * 4, hipMemcpyDeviceToHost);
    uint64_t total = 0, over = 0, most = 0;
    for (uint32_t c : cn) {
      total += c;
      over += c > 256u;
      most = std::max<uint64_t>(most, c);
    }
    fprintf(stderr, "zippy_hip: %u search batches of 65536 bits: %llu candidates for a block start, %llu at most, %llu batches with more than the queue's 256\n",
            g.nfind, (unsigned long long)total, (unsigned long long)most, (unsigned long long)over);
  }
  for (uint32_t i = 0; i < g.nstreams; i++) {
    const uint32_t n = first[i + 1] - first[i];
    if (!n) continue;
    fprintf(stderr, "zippy_hip: stream %u: %u segments, go %u, chain holds %u, %u on it\n", i, n, go[i], ok[i], nchain[i]);
    if (ok[i]) continue;
    std::vector<uint64_t> nominal(n), start(n), end(n), out(n), cap(n), shdr(n);
    std::vector<int32_t> st(n);
    std::vector<uint32_t> fin(n), sub(n);
    (void)hipMemcpy(nominal.data(), g.nominal_bit + first[i], n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(start.data(), g.start_bit + first[i], n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(end.data(), g.end_bit + first[i], n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(out.data(), g.seg_out + first[i], n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(cap.data(), g.eff_tok_cap + first[i], n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(shdr.data(), g.sub_hdr + first[i], n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(st.data(), g.seg_status + first[i], n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(fin.data(), g.final_block + first[i], n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(sub.data(), g.is_sub + first[i], n * 4, hipMemcpyDeviceToHost);
    // follow the chain the way zh_seg_chain_kernel does, and say where it ends
    uint64_t want = start[0];
    uint32_t k = 0, links = 0, none = 0, bad = 0;
    for (uint32_t j = 0; j < n; j++) {
      none += start[j] == ~0ull;
      bad += start[j] != ~0ull && st[j] != 0;
    }
    fprintf(stderr, "  %u segments without a start, %u with a failed decoder\n", none, bad);
    while (k < n) {
      while (k < n && start[k] != want) k++;
      if (k == n) break;
      links++;
      if (st[k] != 0 || fin[k]) break;
      want = end[k];
      k++;
    }
    const uint32_t lo = k < n ? k : 0;
    fprintf(stderr, "  chain: %u links, %s; wanted bit %llu\n", links,
            k == n ? "no segment starts where the last one stopped" : st[k] ? "a decoder failed" : "reached the last block",
            (unsigned long long)want);
    uint32_t near = 0;
    while (near + 1 < n && nominal[near + 1] <= want) near++;
    for (uint32_t j = (k == n ? near : lo) > 2 ? (k == n ? near : lo) - 2 : 0; j < n && j < (k == n ? near : lo) + 4; j++)
      fprintf(stderr, "  seg %u: nominal %llu start %lld (sub %u of the block at %lld) end %llu status %d final %u out %llu tokens' room %llu\n", j,
              (unsigned long long)nominal[j], (long long)start[j], sub[j], (long long)shdr[j], (unsigned long long)end[j], st[j], fin[j],
              (unsigned long long)out[j], (unsigned long long)cap[j]);
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
  // (a caller that drives two contexts' plans from one thread: the launches below go to the current device)
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  AuxJoinGuard aux{ctx, s};
  if (p->is_compress) {
    const ZhCompressArgs& a = p->ca;
    const int want_crc = p->fmt == ZH_DF_GZIP || p->force_crc, want_adler = p->fmt == ZH_DF_ZLIB;
    // zh_emit_kernel and the layout kernels address the output as aligned 32-bit words.  Nothing is cleared beforehand:
    // every word that is OR-ed into is zeroed by the layout kernels first, everything else is stored whole (zh_huffman.hip)
    if ((uintptr_t)d_dst & 3u) return ZH_ERR_ARGUMENT;
    if (p->level == 1 && l1_parallel(ctx)) {
      prof_mark(p, "zh_l1p_match_kernel");
      zh_launch_l1p_match(s, d_src, a, p->l1_tables, p->l1_counter);
    } else if (p->level == 1 || p->level == -2) {
      prof_mark(p, "zh_l1_match_kernel");
      // (from the second run on the cheapest fragments -- by what they cost the run before -- are handed out last)
      zh_launch_l1_match(s, d_src, a, p->level == -2, p->l1_tables, p->l1_counter, p->l1_cost, p->l1_order, p->l1_hist,
                         p->l1_runs > 0 && l1_longest_first());
      p->l1_runs++;
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
        zh_launch_chain_prev(s, d_src, ar, p->head_scratch, p->chain_prev, p->chain_best, p->ctx->chain_links_serial ? 1 : 0, cfg[0]);
        ZH_HIP(ctx, hipGetLastError());
        prof_mark(p, "zh_chain_walk_kernel");
        zh_launch_chain_search(s, d_src, ar, cfg[0], cfg[1], cfg[2], p->chain_prev, p->chain_best, p->ctx->chain_links_serial ? 1 : 0);
        prof_mark(p, "zh_chain_select_kernel");
        zh_launch_chain_select(s, d_src, ar, cfg[0], cfg[1], cfg[2], p->chain_prev, p->chain_best, p->ctx->chain_links_serial ? 1 : 0);
        ZH_HIP(ctx, hipGetLastError());
      }
      p->chain_best_dirty = false;  // (every launch was accepted: the links kernel hands best[] back cleared)
      prof_mark(p, "zh_frag_stats_kernel");
      zh_launch_frag_stats(s, d_src, a);
    }
    // The checksum of the source and the blocks' code builder need nothing from each other and wait for different
    // things -- the one for the LDS (its table look-ups) on every CU, the other, a lane a block, for its own chain of
    // dependent instructions --: side by side, the checksum on the context's second stream, forked behind the match
    // finder (beside THAT it only costs) and joined in front of the layout, which writes the trailer.
    p->k_aux_used = false;
    bool trailer_late = false;
    if (want_crc || want_adler) {
      hipStream_t cs = s;
      const bool aside = checksum_aside(ctx);
      if (aside) {
        cs = ctx->aux_stream;
        ZH_HIP(ctx, hipEventRecord(ctx->aux_fork, s));
        ZH_HIP(ctx, hipStreamWaitEvent(cs, ctx->aux_fork, 0));
        aux.forked = true;
        if (p->profiling) {
          for (hipEvent_t& e : p->k_aux)
            if (!e && hipEventCreate(&e) != hipSuccess) p->profiling = false;
          if (p->profiling) {
            ZH_HIP(ctx, hipEventRecord(p->k_aux[0], cs));
            p->k_aux_used = true;
          }
        }
      } else {
        prof_mark(p, "zh_checksum_pieces_kernel");
      }
      zh_launch_checksum_pieces(cs, ctx->cktabs, d_src, p->d_pieces, p->npieces, nullptr, want_crc,
                                want_adler, p->piece_crc, p->piece_adler, p->piece_len);
      if (!aside) prof_mark(p, "zh_checksum_combine_kernel");
      else if (p->k_aux_used) ZH_HIP(ctx, hipEventRecord(p->k_aux[1], cs));
      zh_launch_checksum_combine(cs, ctx->cktabs, p->d_bufs, (uint32_t)p->n, p->piece_crc, p->piece_adler,
                                 p->piece_len, want_crc, want_adler, p->buf_crc, p->buf_adler);
      if (aside) {
        if (p->k_aux_used) ZH_HIP(ctx, hipEventRecord(p->k_aux[2], cs));
        ZH_HIP(ctx, hipEventRecord(ctx->aux_join, cs));
      }
      prof_mark(p, "zh_huffman_kernel");
      zh_launch_huffman(s, a, p->level == 1 && l1_parallel(ctx) ? 1 : 0);
      // (the trailer is all that needs the checksum: in large batches it is written behind the emission -- a kernel of
      // its own --, and the checksum has the emission to run beside as well: it takes three times as long there, the
      // emission 5 % longer, and the pair ends 0.25 ms sooner)
      trailer_late = aside && p->trailer_late;
      if (aside && !trailer_late) {
        ZH_HIP(ctx, hipStreamWaitEvent(s, ctx->aux_join, 0));
        aux.forked = false;
      }
    } else {
      prof_mark(p, "zh_huffman_kernel");
      // (contract mode -- zh_set_l1_parse(ctx, 1), BestSpeed only -- also builds the block's codes without the
      // replay of the reference's heap: optimal codes, other tie-breaks)
      zh_launch_huffman(s, a, p->level == 1 && l1_parallel(ctx) ? 1 : 0);
    }
    prof_mark(p, "zh_layout_kernel");
    zh_launch_layout(s, d_dst, a, p->buf_crc, p->buf_adler, trailer_late ? 0 : 1);
    prof_mark(p, "zh_emit_kernel");
    // (the exact BestSpeed parse and level -2 have left their coverage bitmaps for it; ZH_EMIT_COVER=0: measurement)
    static const bool cover_ok = [] { const char* e = getenv("ZH_EMIT_COVER"); return !(e && strcmp(e, "0") == 0); }();
    zh_launch_emit(s, d_src, d_dst, a, cover_ok && ((p->level == 1 && !l1_parallel(ctx)) || p->level == -2) ? 1 : 0);
    if (trailer_late) {
      prof_mark(p, "zh_trailer_kernel");
      ZH_HIP(ctx, hipStreamWaitEvent(s, ctx->aux_join, 0));
      aux.forked = false;
      zh_launch_trailer(s, d_dst, a, p->buf_crc, p->buf_adler);
    }
    prof_mark(p, "end");
  } else {
    const ZhInflateArgs& a = p->ia;
    prof_mark(p, "zh_unwrap_kernel");
    zh_launch_unwrap(s, d_src, a);
    const bool split_ok = !p->indexed && inflate_split_enabled(ctx) && plan_token_pool(p);
    const bool split = split_ok && !a.count_only;
    ZhInflateArgs a1 = a;
    // both checksums: with dfDetect the format is only known per stream on the device
    const int want_crc_u = p->fmt == ZH_DF_GZIP || p->fmt == ZH_DF_DETECT || p->force_crc;
    const int want_adler_u = p->fmt == ZH_DF_ZLIB || p->fmt == ZH_DF_DETECT;
    uint32_t ck_done = 0;  // checksum pieces already taken care of (the first half's, below)
    if (split_ok && p->segmented) {
      // a handful of large streams: many workgroups per stream (zh_inflate_seg.hip); streams whose
      // chain of segments does not hold are left to the ordinary kernels below.  A sizing pass
      // stops behind the chain kernel, which knows the output size by then.
      prof_mark(p, "zh_seg_find_kernel");
      zh_launch_seg_find(s, d_src, a, p->sg);
      prof_mark(p, "zh_seg_check_kernel");
      zh_launch_seg_check(s, d_src, a, p->sg);
      if (p->sg_fake_start != ~0ull) zh_launch_seg_fake_start(s, p->sg, p->sg_fake_start);
      prof_mark(p, "zh_seg_substart_kernel");
      zh_launch_seg_tokens(s, d_src, a, p->tok_pool, p->sg, 0);
      zh_launch_seg_decide(s, a, p->sg, 0);
      prof_mark(p, "zh_seg_tokens_kernel");
      zh_launch_seg_tokens(s, d_src, a, p->tok_pool, p->sg, 1);
      prof_mark(p, "zh_seg_chain_kernel");
      zh_launch_seg_chain(s, a, p->sg, 0);
      // a chain broken by found starts that were none: once more without them (streams whose chain holds sit it out)
      for (int round = 0; round < p->sg_repair_rounds; round++) {
        prof_mark(p, "zh_seg_repair_kernel");
        zh_launch_seg_repair(s, a, p->sg);
        zh_launch_seg_tokens(s, d_src, a, p->tok_pool, p->sg, 4);
        zh_launch_seg_decide(s, a, p->sg, 1);
        zh_launch_seg_tokens(s, d_src, a, p->tok_pool, p->sg, 5);
        zh_launch_seg_chain(s, a, p->sg, 1);
      }
      zh_launch_seg_stats(s, p->sg, ctx->d_seg_stats);  // (the context's two counters: made and zeroed by zh_create)
      if (p->sg_trace) seg_trace(p, s);
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
      p->k_half_used = false;
      if (p->tok_groups.size() <= 1 && p->halves_min >= 2u && a1.nbufs >= p->halves_min && second_stream(ctx)) {
        // Two halves, the second on the context's second stream: 4096 workgroups are 3.2 rounds of the machine for
        // either kernel, and both wait more than they work -- side by side a half's writer fills what
        // the other half's tokens kernel leaves idle, and only the last tail is nobody's to fill (own streams 20.8 ->
        // 19.7 ms, zlib level-6 members 24.3 -> 23.4; three and four parts: slower).
        // Kernel times: both halves' launches are reported, each as long as it took BESIDE the other.
        ZhInflateArgs ah[2] = {a1, a1};
        ah[0].nbufs = a1.nbufs / 2u;
        ah[1].first_buf = a1.first_buf + ah[0].nbufs;
        ah[1].nbufs = a1.nbufs - ah[0].nbufs;
        hipStream_t s2 = ctx->aux_stream;
        ZH_HIP(ctx, hipEventRecord(ctx->aux_fork, s));
        ZH_HIP(ctx, hipStreamWaitEvent(s2, ctx->aux_fork, 0));
        aux.forked = true;
        if (p->profiling) {
          for (hipEvent_t& e : p->k_half)
            if (!e && hipEventCreate(&e) != hipSuccess) p->profiling = false;
          p->k_half_used = p->profiling;
        }
        if (p->k_half_used) ZH_HIP(ctx, hipEventRecord(p->k_half[0], s2));
        zh_launch_inflate_tokens(s2, d_src, ah[1], p->tok_pool, p->tok_off, p->tok_cap);
        if (p->k_half_used) ZH_HIP(ctx, hipEventRecord(p->k_half[1], s2));
        zh_launch_inflate_write(s2, d_src, d_dst, ah[1], p->tok_pool, p->tok_off);
        if (p->k_half_used) ZH_HIP(ctx, hipEventRecord(p->k_half[2], s2));
        ZH_HIP(ctx, hipEventRecord(ctx->aux_join, s2));
        prof_mark(p, "zh_inflate_tokens_kernel");
        zh_launch_inflate_tokens(s, d_src, ah[0], p->tok_pool, p->tok_off, p->tok_cap);
        prof_mark(p, "zh_inflate_write_kernel");
        zh_launch_inflate_write(s, d_src, d_dst, ah[0], p->tok_pool, p->tok_off);
        // (the first half's output is complete: its checksum pieces fill what the second half's tail leaves idle
        // instead of waiting for it)
        if ((want_crc_u || want_adler_u) && a1.first_buf == 0 && a1.nbufs == (uint32_t)p->n && p->half_piece) {
          prof_mark(p, "zh_checksum_pieces_kernel");
          zh_launch_checksum_pieces(s, ctx->cktabs, d_dst, p->d_pieces, p->half_piece, p->out_len, want_crc_u, want_adler_u,
                                    p->piece_crc, p->piece_adler, p->piece_len);
          ck_done = p->half_piece;
        }
        prof_mark(p, "(waiting for the other half)");
        ZH_HIP(ctx, hipStreamWaitEvent(s, ctx->aux_join, 0));
        aux.forked = false;
      } else if (p->tok_groups.size() <= 1) {
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
    } else if (a.count_only && !p->indexed && inflate_split_enabled(ctx)) {
      // a sizing pass: the tokens kernel without its records (the serial decoder took several times the decode itself)
      prof_mark(p, "zh_inflate_tokens_kernel");
      zh_launch_inflate_count(s, d_src, a1);
    } else {
      prof_mark(p, "zh_inflate_kernel");
      zh_launch_inflate(s, d_src, d_dst, a1);
    }
    if (!a.count_only) {
      const int want_crc = want_crc_u, want_adler = want_adler_u;
      if (want_crc || want_adler) {
        prof_mark(p, "zh_checksum_pieces_kernel");
        zh_launch_checksum_pieces(s, ctx->cktabs, d_dst, p->d_pieces + ck_done, p->npieces - ck_done, p->out_len,
                                  want_crc, want_adler, p->piece_crc + ck_done, p->piece_adler + ck_done,
                                  p->piece_len + ck_done);
        prof_mark(p, "zh_checksum_combine_kernel");
        zh_launch_checksum_combine(s, ctx->cktabs, p->d_bufs, (uint32_t)p->n, p->piece_crc, p->piece_adler,
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
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  if (out_lens) ZH_HIP(ctx, hipMemcpyAsync(out_lens, p->out_len, p->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (statuses) ZH_HIP(ctx, hipMemcpyAsync(statuses, p->status, p->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  ZH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZH_OK;
}
extern "C" int zh_debug_segment_stats(zh_ctx* ctx, uint64_t* cut, uint64_t* held) {
  if (!ctx) return ZH_ERR_ARGUMENT;
  uint64_t h[2] = {0, 0};
  if (ctx->d_seg_stats) {
    ZH_HIP(ctx, hipSetDevice(ctx->device));
    ZH_HIP(ctx, hipMemcpyAsync(h, ctx->d_seg_stats, 16, hipMemcpyDeviceToHost, ctx->stream));
    ZH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (cut) *cut = h[0];
  if (held) *held = h[1];
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
  ZH_HIP(ctx, hipSetDevice(ctx->device));
  ZH_HIP(ctx, hipMemcpyAsync(crcs, p->buf_crc, p->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  ZH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZH_OK;
}
extern "C" const uint64_t* zh_plan_device_lens(zh_plan* p) { return p ? p->out_len : nullptr; }
extern "C" const int32_t* zh_plan_device_statuses(zh_plan* p) { return p ? p->status : nullptr; }
