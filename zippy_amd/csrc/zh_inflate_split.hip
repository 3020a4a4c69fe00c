// Batched inflate in two kernels ("split" mode): a stream's Huffman decoding is parallel over the
// stream, only its LZ copies are serial.
//
// A deflate block is one long run of variable-length codes, so a serial decoder -- the reference's
// inflate.nim:173-250 and zh_inflate.hip's two-wave form of it -- spends its time finding where the
// next code starts.  Huffman decoding re-synchronises by itself, though: a decoder started at an
// arbitrary bit falls in step with the real code sequence after a few dozen symbols.  So:
//
//   zh_inflate_tokens_kernel (256 or 1024 threads per stream)
//     blocks in order (inflate.nim:273-289).  A block's header: wave 0 reads the fixed fields and -- for a clean
//     dynamic header -- the code lengths with all its lanes (code_lengths_wave: the symbols that start at 64
//     consecutive bits at once, a readlane walk picks the real ones); anything else goes to the serial reader,
//     which raises the reference's errors.  The two decode tables are built by the whole workgroup
//     (build_tables_wg, zh_inflate_tables.h).  Then the block's bits are taken 131 072 at a time ("superchunk",
//     staged in LDS), one 512-bit SUBCHUNK per thread.  Every thread decodes whole tokens (literal, or
//     length+extra+distance+extra, inflate.nim:93-100 / 199-222) from its guessed start -- found by a run-up
//     through the subchunk before -- to the first token start at or behind its subchunk's end; that end is the
//     next thread's true start, so starts are handed on and threads whose start changed decode again until no
//     start changes.  Thread 0's start is exact, so by induction every start then is: the result is the serial
//     decoder's token sequence -- self-synchronisation only decides how many turns that takes (one or two),
//     never what comes out.  Token counts are prefix-summed and a last pass writes the tokens (one 32-bit record
//     each -- a pair of literals shares one --, the serial kernel's round-record format) to the stream's token
//     buffer in HBM.  End of block, invalid symbols and the end of the input are found by the thread that owns
//     the bit position, in stream order; stored blocks (inflate.nim:252-266) become one record.
//   zh_inflate_write_kernel (256, 512 or 1024 threads per stream)
//     records into bytes, rounds of eight output bytes a thread: a prefix sum places a round's records, a byte ->
//     record map tells every byte its record; literals carry their value, bytes copied from before the round are
//     read back through L2 (the LZ window is the output itself), bytes copied from inside it chase their source
//     by pointer doubling in LDS.  Keeps inflate.nim:224-225's distance check and the capacity check; produces
//     out_len and the status.
//
// Same checks, same accept/reject decision and the same bytes as zh_inflate_kernel (the tests
// run both against the oracle).  Algorithmic traffic: C read + N written, plus the token
// records (4 bytes per token, written once and read once).
#include <cstdlib>
#include <type_traits>

#ifndef ZH_WR_OCC
#define ZH_WR_OCC 5
#endif
#ifndef ZH_WR_BYTES
#define ZH_WR_BYTES 8
#endif
#ifndef ZH_RUNUP
#define ZH_RUNUP 512
#endif
#ifndef ZH_SUBBITS
#define ZH_SUBBITS 512
#endif
#ifndef ZH_TOK_OCC
#define ZH_TOK_OCC 5  // workgroups a CU of the tokens kernel's 256-thread form (4: 8.03 ms, 5: 7.26 ms for 4096 x 1 MiB of own streams)
#endif
#ifndef ZH_SERIAL_HEADER
#define ZH_SERIAL_HEADER 0  // 1: the code lengths of every dynamic header by the serial reader (cross-check, measurement)
#endif

#include "zh_common.h"
#include "zh_kprof.h"
#ifdef ZH_KPROF_HDR
#define KPROF_HDR_MARK(i) KPROF_MARK(i)
#else
#define KPROF_HDR_MARK(i) ((void)0)
#endif
#include "zh_tables.h"
#include "zh_inflate_tables.h"

namespace {

constexpr bool kSerialHeader = ZH_SERIAL_HEADER != 0;
constexpr uint32_t kSubBits = ZH_SUBBITS;                            // one thread's share of a superchunk
constexpr uint32_t kSubWords = kSubBits / 32u;                 // 16
// The staged superchunk gives every subchunk 19 dwords: its own 16 and a copy of the next three
// (a token that starts in the subchunk reads at most that far), so dword w of the superchunk sits
// at w + 3 * (w / 16) and a token's three dwords are consecutive.  19 is odd: the lanes of a wave,
// one subchunk apart, read 64 different banks (16 apart they would share four).
constexpr uint32_t kSubStride = kSubWords + 3u;
constexpr uint32_t kHeaderWords = 288;                        // a dynamic header is < 900 bytes
constexpr uint32_t kDistSub = 256;                            // second-level distance tables
constexpr uint32_t kNoStart = 0xffffffffu;                    // "the thread before me ended the block"
// bits decoded ahead of a subchunk to find its first token boundary (4096 x 1 MiB, tokens kernel: none
// 20.3 ms, 128 bits 17.5, 256 16.4, 512 15.5; subchunks of 1024 bits with a run-up of 512: 17.8)
constexpr uint32_t kRunUp = ZH_RUNUP;
static_assert(kRunUp <= ZH_SUBBITS, "a run-up stays inside the subchunk before");
constexpr uint32_t kMinTurns = 3;     // speculative turns before the all-starts pass may take over
constexpr uint32_t kSlowGain = 12;    // ... when a turn added fewer final threads than this
constexpr uint32_t kStartSpan = 48;   // a token is at most 48 bits: a subchunk's true start is one of 48
constexpr uint32_t kMapTerm = 0xffu;  // all-starts map: "ends the block (or fails) in this subchunk"

// token records (uint32): the serial kernel's round record
//   bits 0-8 output length | bit 9 literal | (bit 10: in chain, set by the writer) | bits 16-31 value
// special records have bit 15 set and length 0:
//   bits 11-12 = 1: stored run of `value` bytes, followed by two words: byte offset in the stream
//   bits 11-12 = 2: end of the stream, value = status
constexpr uint32_t kRecSpecial = 0x8000u, kRecStored = 1u << 11, kRecEnd = 2u << 11;

struct RunResult {
  uint32_t end;    // bit position (superchunk-relative) behind the last token taken
  uint32_t n;      // tokens
  uint32_t term;   // 0 none, 1 end of block, else a ZH_ERR_* status
  uint32_t bytes;  // output bytes of the tokens written (segment mode)
};

}  // namespace

// kSplitThreads: 256 (a superchunk of 16 KiB of the stream, five workgroups per CU: batches) or
// 1024 (64 KiB, one workgroup per CU: up to a few streams per CU, where the chain of a stream's
// superchunks is what takes the time).
// kSeg: a workgroup takes a SEGMENT of a stream (ZhSegArgs, zh_inflate_seg.hip): it starts at the
// block the segment's search found, stops at the first block boundary at or behind the next found
// start, and reports where that was, how many bytes its tokens make and whether the stream ended.
// kCount: a SIZING pass (ZhInflateArgs::count_only: streams that carry no size and outgrew the guess made for them,
// zippy.nim:130-165, zh_host_batch.hip): the same decode, but no record is written and no room is asked for -- the
// stream's output bytes are summed and left in out_len.  An instantiation of its own: the batches' kernel keeps its
// registers and its code.
template <uint32_t kSplitThreads, bool kSeg, bool kCount = false>
__global__ __launch_bounds__(kSplitThreads, kSplitThreads == 256 ? ZH_TOK_OCC : kSplitThreads < 256 ? 4 : 1) void zh_inflate_tokens_kernel(const uint8_t* __restrict__ d_src,
                                                                ZhInflateArgs a,
                                                                uint32_t* __restrict__ tok_pool,
                                                                const uint64_t* __restrict__ tok_off,
                                                                const uint64_t* __restrict__ tok_cap,
                                                                ZhSegArgs g, int phase) {
  constexpr uint32_t kSuperBits = kSplitThreads * kSubBits;
  constexpr uint32_t kStageWords = (kSplitThreads + 1u) * kSubStride;
  constexpr uint32_t kWaves = kSplitThreads / 64u;
  // subchunks mapped per all-starts pass (the map is what decides how many workgroups a CU's LDS holds)
  constexpr uint32_t kMapGroup = kSplitThreads == 256u && ZH_TOK_OCC > 4 ? 48u : kSplitThreads >= 256u ? 64u : 16u;
  __shared__ uint32_t s_lit[(1u << kLitBits) + kLitSub];
  __shared__ uint32_t s_dst[(1u << kDistBits) + kDistSub];  // also hosts the 7-bit code-length table
  __shared__ uint32_t s_in[kStageWords];
  __shared__ HuffTab s_tab_lit, s_tab_dist, s_tab_cl;
  __shared__ uint16_t s_val_lit[288], s_val_dist[32], s_val_cl[20];
  __shared__ uint8_t s_lens[320 + 16];
  __shared__ uint32_t s_cnt[16];
  __shared__ uint32_t s_end[kSplitThreads];   // where every thread's run ended
  __shared__ uint32_t s_wsum[kWaves];
  __shared__ uint32_t s_first_dirty[2], s_first_term[2];  // per turn parity
  // all-starts map of a subchunk: entry i = where the decode that starts i bits into the subchunk
  // comes out, in bits behind the subchunk's end (or kMapTerm)
  __shared__ uint8_t s_map[kMapGroup][kStartSpan];
  // block / superchunk control words written by one thread, read by all
  __shared__ uint32_t s_c_btype, s_c_final, s_c_st, s_c_stored_len, s_c_term, s_c_endrel, s_c_hlit, s_c_hdist;
  __shared__ uint64_t s_c_pos;

  const uint32_t tid = threadIdx.x;
  const unsigned lane = zh_lane();
  const uint32_t sid = kSeg ? blockIdx.x : a.first_buf + blockIdx.x;  // token region (a stream, or a segment of one)
  const uint32_t bid = kSeg ? g.parent[sid] : sid;   // the stream
  if (a.status[bid] != ZH_OK) return;  // unwrap already failed this stream
  if (!kSeg && a.skip && a.skip[sid]) return;  // decoded segment-wise
  // segment mode: the next found start at or behind the decoder's position.  It stops only where it
  // lands on one EXACTLY; a start it runs past was a wrong guess (bits inside a stored block that read
  // like a header -- a compressed file inside an archive is full of them) and is ignored.
  // A segment inside a long block can have a SUB-START: a token boundary of that block, guessed in
  // phase 0 of this kernel (below) from the block's header (g.sub_hdr) by decoding towards the
  // segment from 4096 bits before it -- Huffman decoding falls in step by itself.  Its decoder reads
  // the tables from that header and starts at the boundary; the decoder before lands on it only if
  // it is in the same block (and stops there, in the middle of the block).
  uint64_t seg_start = 0, seg_target = 0, cur_hdr = kSegNone;
  uint32_t seg_tj = sid, seg_last = 0;
  bool sub_first = false, seg_landed = false;
  // A rerun (phases 4 / 5 = 0 / 1 once more, zh_seg_repair_kernel): only the streams under repair.
  const bool rerun = kSeg && phase >= 4;
  const int route = kSeg ? 0 : phase;  // (batches: 9 / 8, see below; 1: every stream)
  phase &= 3;
  if (rerun && !g.repair[bid]) return;
  if (kSeg && phase != 0) {
    if (!g.go[bid]) return;  // too few starts were found: the stream is left to the ordinary kernels
    seg_start = g.start_bit[sid];
    if (seg_start == kSegNone) return;  // no start in this segment: the decoder before carries on through it
    seg_last = g.first_seg[bid + 1u];
    sub_first = g.is_sub[sid] != 0u;
  }
  if (kSeg && phase == 0) {
    if (sid == g.first_seg[bid] || g.start_bit[sid] != kSegNone) return;  // (it has a real start)
  }

  const ZhBufDesc bd = a.bufs[bid];
  const uint8_t* src = d_src + bd.src_off;
  const uint64_t src_len = a.src_len_dev ? a.src_len_dev[bid] : bd.src_len;
  const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
  const uint32_t* asrc = reinterpret_cast<const uint32_t*>(src - mis);
  const uint64_t end = mis + src_len;  // first byte offset (from asrc) past the stream
  auto load_dword = [&](uint64_t off) -> uint32_t {  // bytes past the end read as zero
    if (off >= end) return 0u;
    uint32_t v = asrc[off >> 2];
    if (off + 4 > end) v &= (1u << (8 * (uint32_t)(end - off))) - 1u;
    return v;
  };
  // Batches take two forms of this kernel (zh_launch_inflate_tokens): a superchunk's worth of lanes is busy only while
  // the block lasts, and a stream of short blocks -- system zlib's are ~ 15 KiB of codes each -- ends most
  // superchunks of 256 subchunks early; 128 fill twice as well (zlib-6 members 12.7 -> 10.7 ms), at ~ 7 % more
  // on long blocks (barriers and scans over half the lanes).  What a stream is made of is not known before it is
  // decoded; whether its first block is its last is: bit 0 of the body.  route 1: streams of ONE block (this
  // library's own up to 4 MiB of input), 2: the others.
  if (!kSeg && route >= 8) {
    const uint64_t b0 = (uint64_t)mis + a.body_pos[sid];
    const bool one_block = ((load_dword(b0 & ~(uint64_t)3) >> (8u * (uint32_t)(b0 & 3u))) & 1u) != 0u;
    if (one_block != (route == 9)) return;
  }
  uint32_t* const tok = kCount ? nullptr : tok_pool + (kSeg ? g.eff_tok_off[sid] : tok_off[sid]);
  const uint64_t cap = kCount ? ~0ull : kSeg ? g.eff_tok_cap[sid] : tok_cap[sid];  // records this stream may write (the end record included)
  __shared__ uint32_t s_wbytes[kWaves];

  // stream position in bits (from asrc)
  uint64_t pos = kSeg ? (uint64_t)mis * 8 + (sub_first ? g.sub_hdr[sid] : seg_start) : ((uint64_t)mis + a.body_pos[sid]) * 8;
  uint64_t ntok = 0, out_bytes = 0;
  KPROF_DECL(8);  // cycles: 0 header + tables, 1 staging, 2 sync turns, 3 scan, 4 token pass; counts: 5 superchunks, 6 turns, 7 streams
  int st = ZH_OK;
  bool final_block = false;

  // ---- one token at superchunk-relative bit p, decoded by the calling lane alone ----
  // returns the token's bits (0: not a token, see *kind) and its record (the all-starts pass; the passes proper
  // live in run() below)
  auto decode_at = [&](uint32_t p, uint32_t* rec, uint32_t* kind) -> uint32_t {
    const uint32_t wi = p >> 5, sh = p & 31u;
    const uint32_t si = wi + 3u * (p / kSubBits);
    const uint32_t d0 = s_in[si], d1 = s_in[si + 1u], d2 = s_in[si + 2u];
    const uint32_t v_lo = zh_alignbit(d1, d0, sh), v_hi = zh_alignbit(d2, d1, sh);
    uint32_t e = s_lit[v_lo & ((1u << kLitBits) - 1u)];
    if (e & 0x400u) e = s_lit[(e >> 16) + ((v_lo >> kLitBits) & ((1u << (e & 15u)) - 1u))];
    if (e == 0) {  // inflate.nim:67-91 decodeSymbolSlow: longer than the tables reach, or unassigned
      const uint32_t k = __brev(v_lo) >> 16;
      uint32_t cl = kLitBits + 1;
      while (cl < 16 && k >= s_tab_lit.max_codes[cl]) cl++;
      uint32_t sym = 0xffffu;
      if (cl < 16)
        sym = s_val_lit[((k >> (16 - cl)) - s_tab_lit.first_code[cl] + s_tab_lit.first_symbol[cl]) & 0xffffu];
      e = litlen_entry(sym, cl < 16 ? cl : 0);
    }
    const uint32_t L = e & 15u;
    if (e & 0x8000u) {
      *kind = 0;
      *rec = 1u | (1u << 9) | (((e >> 16) & 0xffu) << 16);
      return L;
    }
    const uint32_t k1 = (e >> 8) & 3u;
    if (k1 != kKindBase) {
      *kind = k1 == kKindEob ? 1u : (uint32_t)ZH_ERR_INVALID_BUFFER;  // inflate.nim:202-204
      *rec = 0;
      return L;
    }
    const uint64_t v = (uint64_t)v_lo | ((uint64_t)v_hi << 32);
    const uint32_t eb = (e >> 4) & 15u;
    const uint32_t length = (e >> 16) + ((v_lo >> L) & ((1u << eb) - 1u));
    const uint32_t o2 = L + eb;  // <= 20
    const uint32_t dv = zh_alignbit(v_hi, v_lo, o2);
    uint32_t de = s_dst[dv & ((1u << kDistBits) - 1u)];
    if (de & 0x400u) de = s_dst[(de >> 16) + ((dv >> kDistBits) & ((1u << (de & 15u)) - 1u))];
    if (de == 0) {
      const uint32_t k = __brev(dv) >> 16;
      uint32_t cl = kDistBits + 1;
      while (cl < 16 && k >= s_tab_dist.max_codes[cl]) cl++;
      uint32_t sym = 0xffffu;
      if (cl < 16)
        sym = s_val_dist[((k >> (16 - cl)) - s_tab_dist.first_code[cl] + s_tab_dist.first_symbol[cl]) & 0xffffu];
      de = dist_entry(sym, cl < 16 ? cl : 0);
    }
    if (((de >> 8) & 3u) != kKindBase) {  // inflate.nim:211-213
      *kind = (uint32_t)ZH_ERR_INVALID_BUFFER;
      *rec = 0;
      return 0;
    }
    const uint32_t o3 = o2 + (de & 15u), deb = (de >> 4) & 15u;  // o3 <= 35, the token <= 48 bits
    const uint32_t dist = (de >> 16) + ((uint32_t)(v >> o3) & ((1u << deb) - 1u));
    *kind = 0;
    *rec = length | (dist << 16);
    return o3 + deb;
  };
  // the rare litlen entries (not "plain", zh_inflate_tables.h): a link to a second-level table, or an empty entry --
  // inflate.nim:67-91 decodeSymbolSlow: longer than the tables reach, or unassigned.  What comes back is a plain
  // entry, end of block, or an invalid symbol (kKindBad).
  auto lit_rare = [&](uint32_t e, uint32_t v_lo) -> uint32_t {
    if (e & 0x400u) e = s_lit[(e >> 16) + ((v_lo >> kLitBits) & ((1u << (e & 15u)) - 1u))];
    if (e == 0) {
      const uint32_t k = __brev(v_lo) >> 16;
      uint32_t cl = kLitBits + 1;
      while (cl < 16 && k >= s_tab_lit.max_codes[cl]) cl++;
      uint32_t sym = 0xffffu;
      if (cl < 16)
        sym = s_val_lit[((k >> (16 - cl)) - s_tab_lit.first_code[cl] + s_tab_lit.first_symbol[cl]) & 0xffffu];
      e = litlen_entry(sym, cl < 16 ? cl : 0);
    }
    return e;
  };
  auto dist_rare = [&](uint32_t de, uint32_t dv) -> uint32_t {
    if (de & 0x400u) de = s_dst[(de >> 16) + ((dv >> kDistBits) & ((1u << (de & 15u)) - 1u))];
    if (de == 0) {
      const uint32_t k = __brev(dv) >> 16;
      uint32_t cl = kDistBits + 1;
      while (cl < 16 && k >= s_tab_dist.max_codes[cl]) cl++;
      uint32_t sym = 0xffffu;
      if (cl < 16)
        sym = s_val_dist[((k >> (16 - cl)) - s_tab_dist.first_code[cl] + s_tab_dist.first_symbol[cl]) & 0xffffu];
      de = dist_entry(sym, cl < 16 ? cl : 0);
    }
    return de;
  };
  // tokens from p up to the first token start at or behind `limit`; `end_rel`: the input's end.
  // The loop every pass of the kernel lives in (run-up, turns, writing pass), so it is written for the instructions it
  // issues (round 5: 123 -> ~ 90 an iteration of a wave with literals and copies in it): ONE test sends everything rare
  // out of the way (table entries carry a "plain" bit), the endings -- an invalid symbol (inflate.nim:202-204, 211-213),
  // the input's end (the role of `bitsBuffered < 0`), end of block -- are selected into one `term` and leave through
  // one exit, in the order the checks had as branches: an invalid symbol before the token's bits are taken, the
  // input's end before the end of block.
  // `one_row` (std::true_type): the run stays in the subchunk it starts in -- every pass of a superchunk does: a
  // run-up ends where it enters the next subchunk, a thread's own run where it leaves its own, and what a token
  // needs beyond a subchunk's end is in the row's three spare dwords.  Positions then count in the STAGED layout
  // (19 dwords a subchunk: + 96 bits a row, which leaves a position's low five bits alone), and a token's dwords
  // are at (position >> 5) without the row arithmetic.
  auto run = [&](auto one_row, uint32_t p, uint32_t limit, uint32_t end_rel, uint32_t* out) -> RunResult {
    constexpr bool kRow = decltype(one_row)::value;
    RunResult r;
    r.n = 0;
    r.term = 0;
    r.bytes = 0;
    uint32_t t = 0;
    bool go = p < limit;
    const uint32_t row_bits = kRow ? (p / kSubBits) * (32u * (kSubStride - kSubWords)) : 0u;
    if (kRow) {
      p += row_bits;
      limit += row_bits;
      end_rel = end_rel < 0xf0000000u ? end_rel + row_bits : end_rel;  // (a superchunk is 2^19 bits at most)
    }
    while (go) {
      const uint32_t si = kRow ? p >> 5 : (p >> 5) + 3u * (p / kSubBits);
      const uint32_t d0 = s_in[si], d1 = s_in[si + 1u], d2 = s_in[si + 2u];
      const uint32_t v_lo = zh_alignbit(d1, d0, p);
      uint32_t e = s_lit[v_lo & ((1u << kLitBits) - 1u)];
      if (__builtin_expect(!(e & kEntryPlain), 0)) e = lit_rare(e, v_lo);
      const uint32_t L = e & 15u;
      uint32_t tb, rec, tb2 = 0, rec2 = 0;
      t = 0;
      if (e & 0x8000u) {
        tb = L;
        rec = 1u | (1u << 9) | (e & 0xff0000u);
        // a literal right behind a literal comes out of the same 32 bits (L <= 15: at least 17 are left, a root
        // entry needs kLitBits): its bits and what it adds to the record, 0 if what follows is anything else
        const uint32_t e2 = s_lit[(v_lo >> L) & ((1u << kLitBits) - 1u)];
        if (e2 & 0x8000u) {
          tb2 = e2 & 15u;
          rec2 = ((e2 & 0xff0000u) << 8) + 1u;
        }
      } else {
        // a length and its distance (for an end of block or an invalid symbol the same arithmetic on whatever
        // the bits hold: in range, and thrown away below)
        const uint32_t v_hi = zh_alignbit(d2, d1, p);
        const uint32_t eb = (e >> 4) & 15u;
        const uint32_t length = (e >> 16) + ((v_lo >> L) & ((1u << eb) - 1u));
        const uint32_t o2 = L + eb;  // <= 20
        const uint32_t dv = zh_alignbit(v_hi, v_lo, o2);
        uint32_t de = s_dst[dv & ((1u << kDistBits) - 1u)];
        if (__builtin_expect(!(de & kEntryPlain), 0)) de = dist_rare(de, dv);
        const uint32_t dl = de & 15u, deb = (de >> 4) & 15u;  // dl + deb <= 28: the extra bits are in dv
        const uint32_t dist = (de >> 16) + ((dv >> dl) & ((1u << deb) - 1u));
        tb = o2 + dl + deb;  // the token <= 48 bits
        rec = length | (dist << 16);
        if (__builtin_expect(!(e & de & kEntryPlain), 0)) {
          const bool eob = !(e & kEntryPlain) && ((e >> 8) & 3u) == kKindEob;
          t = eob ? 1u : (uint32_t)ZH_ERR_INVALID_BUFFER;
          tb = eob ? L : 0u;
        }
      }
      p += tb;
      // the last token of a run: an ending (t), or one that reaches past the input (the role of `bitsBuffered < 0`)
      const bool last = t != 0u || p > end_rel;
      // A second literal out of the same bits, if it starts before `limit` and ends inside the input: the
      // decisions the loop would take on its next turn, so every pass over these bits finds the same tokens.
      // (Two tokens in three of the bench data are literals: 15.4 -> 11.6 ms.  Up to one / two / three more out
      // of a 64-bit window: 12.6 / 12.8 / 14.0 ms; a literal behind a COPY out of the copy's 64 bits as well: 12.3
      // against 11.4.)  The two share ONE record: length 2, the second byte on top.  (Nothing follows: tb2 = rec2 = 0.)
      const bool pair = !last && p < limit && p + tb2 <= end_rel;
      if (pair) {
        p += tb2;
        rec += rec2;
      }
      if (kCount) {
        if (!last) r.bytes += rec & 0x1ffu;
      } else if (out && !last) {
        out[r.n] = rec;
        if (kSeg) r.bytes += rec & 0x1ffu;
      }
      r.n += last ? 0u : 1u;
      go = !last && p < limit;
    }
    // the ending: an invalid symbol counts before the input's end (its bits are never taken), the input's end before
    // an end of block
    r.term = t > 1u ? t : p > end_rel ? (uint32_t)ZH_ERR_END_OF_BUFFER : t;
    r.end = p - row_bits;
    return r;
  };

  // the block header at `pos`: staged, then read by wave 0 like the serial kernel does; leaves the
  // tables in LDS and the s_c_* words
  uint32_t hdr_st = 0;  // the header's status (the same in every thread)
  auto parse_header = [&]() {
    // ---- block header: staged, then read by wave 0 like the serial kernel does ----
    const uint64_t hbase = pos >> 5;  // dword of the header's first bit
    KPROF_MARK(4);
    __syncthreads();
    for (uint32_t i = tid; i < kHeaderWords; i += kSplitThreads) s_in[i] = load_dword((hbase + i) * 4);
    __syncthreads();
    KPROF_HDR_MARK(1);  // (-DZH_KPROF_HDR: the header's phases in slots 1 / 3 / 0 -- staged, lengths read, tables built)
    if (tid < 64) {
      uint64_t bp = pos & 31u;  // relative to hbase * 32
      uint64_t hb = 0;
      uint32_t hc = 0;
      int hst = ZH_OK;
      auto fetch = [&]() -> uint64_t {
        const uint32_t wi = (uint32_t)(bp >> 5), sh = (uint32_t)bp & 31u;
        const uint32_t d0 = zh_bcast(s_in[wi]), d1 = zh_bcast(s_in[wi + 1u]), d2 = zh_bcast(s_in[wi + 2u]);
        return (uint64_t)zh_alignbit(d1, d0, sh) | ((uint64_t)zh_alignbit(d2, d1, sh) << 32);
      };
      auto need = [&]() {
        if (hc < 32u) {
          hb = fetch();
          hc = 64;
        }
      };
      auto take = [&](uint32_t nbits) -> uint32_t {
        const uint32_t v = (uint32_t)hb & ((1u << nbits) - 1u);
        hb >>= nbits;
        hc -= nbits;
        bp += nbits;
        return v;
      };
      auto past_end = [&]() -> bool { return hbase * 32 + bp > end * 8; };
      auto decode_slow = [&](uint32_t bits, uint32_t lut_bits, const HuffTab* tab, const uint16_t* values,
                             uint32_t* nb) -> uint32_t {
        const uint32_t k = __brev(bits) >> 16;
        uint32_t cl = lut_bits + 1;
        while (cl < 16 && k >= zh_bcast(tab->max_codes[cl])) cl++;
        *nb = 0;
        if (cl >= 16) return 0xffffu;
        const uint32_t id = ((k >> (16 - cl)) - zh_bcast(tab->first_code[cl]) +
                             zh_bcast(tab->first_symbol[cl])) & 0xffffu;
        *nb = cl;
        return zh_bcast(values[id]);
      };
      need();
      const uint32_t bfinal = take(1), btype = take(2);
      uint32_t stored_len = 0;
      if (btype == 0) {  // inflate.nim:252-266 inflateNoCompression
        bp = ((hbase * 32 + bp + 7u) & ~(uint64_t)7) - hbase * 32;
        hc = 0;
        need();
        const uint32_t len = take(16), nlen = take(16);
        if (len + nlen != 65535u) hst = ZH_ERR_INVALID_BUFFER;
        else if (((hbase * 32 + bp) >> 3) + len > end) hst = ZH_ERR_END_OF_BUFFER;
        stored_len = len;
      } else if (btype == 3) {
        hst = ZH_ERR_BLOCK_HEADER;
      } else {
        uint32_t hlit = 288, hdist = 30;
        if (btype == 1) {  // fixed codes, inflate.nim:111-113
          zh_wave_sync();
          for (uint32_t s = lane; s < 288; s += 64) s_lens[s] = (uint8_t)(s <= 143 ? 8 : s <= 255 ? 9 : s <= 279 ? 7 : 8);
          if (lane < 30) s_lens[288 + lane] = 5;
        } else {  // dynamic header, inflate.nim:115-171
          hlit = take(5) + 257;
          hdist = take(5) + 1;
          const uint32_t hclen = take(4) + 4;
          if (hlit > 286 || hdist > 30) hst = ZH_ERR_INVALID_BUFFER;
          // the code lengths by the whole wave (zh_inflate_tables.h); anything but a clean header comes back as 0
          // and is read again by the serial reader below, which raises the reference's error for it
          uint32_t fast_q = 0;
          if (hst == ZH_OK && !kSerialHeader)
            fast_q = code_lengths_wave(s_in, (uint32_t)bp, hlit, hdist, hclen, end * 8 > hbase * 32 ? end * 8 - hbase * 32 : 0,
                                       s_lens, reinterpret_cast<uint8_t*>(s_dst));
#ifdef ZH_EMU
          if (lane == 0 && hst == ZH_OK && getenv("ZH_DBG_HDR")) fprintf(stderr, "dynamic header: %s\n", fast_q ? "wave" : "serial reader");
#endif
          if (fast_q) {
            bp = fast_q;
            hc = 0;
          }
          if (hst == ZH_OK && !fast_q) {
            zh_wave_sync();
            if (lane < 20) s_lens[lane] = 0;
            zh_wave_sync();
            for (uint32_t i = 0; i < hclen; i++) {
              need();
              const uint32_t v = take(3);
              if (lane == 0) s_lens[c_clcl_order[i]] = (uint8_t)v;
            }
            hst = build_table(s_lens, 19, s_dst, 7, 2, &s_tab_cl, s_val_cl, s_cnt);
          }
          if (hst == ZH_OK && !fast_q) {
            uint32_t i = 0;
            const uint32_t total = hlit + hdist;
            uint32_t prev = 0;
            while (i != total) {
              need();
              uint32_t sym;
              const uint32_t e = zh_bcast(s_dst[(uint32_t)hb & 127u]);
              if (e) {
                take(e & 15u);
                sym = e >> 16;
              } else {
                uint32_t nb;
                sym = decode_slow((uint32_t)hb, 7, &s_tab_cl, s_val_cl, &nb);
                take(nb);
              }
              if (past_end()) { hst = ZH_ERR_END_OF_BUFFER; break; }
              if (sym <= 15) {
                if (lane == 0) s_lens[i] = (uint8_t)sym;
                prev = sym;
                i++;
              } else if (sym == 16) {
                if (i == 0) { hst = ZH_ERR_INVALID_BUFFER; break; }
                const uint32_t rep = take(2) + 3;
                if (i + rep > 320) { hst = ZH_ERR_INVALID_BUFFER; break; }
                if (lane < rep) s_lens[i + lane] = (uint8_t)prev;
                i += rep;
              } else if (sym == 17 || sym == 18) {
                const uint32_t rep = sym == 17 ? take(3) + 3 : take(7) + 11;
                for (uint32_t j = lane; j < rep && i + j < 320 + 16; j += 64) s_lens[i + j] = 0;
                prev = 0;
                i += rep;
              } else {
                hst = ZH_ERR_INVALID_SYMBOL;
                break;
              }
              if (i > total) { hst = ZH_ERR_INVALID_BUFFER; break; }
            }
          }
        }
        if (lane == 0) {
          s_c_hlit = hlit;
          s_c_hdist = hdist;
        }
      }
      if (lane == 0) {
        s_c_btype = btype;
        s_c_final = bfinal;
        s_c_st = (uint32_t)hst;
        s_c_stored_len = stored_len;
        s_c_pos = hbase * 32 + bp;  // behind the header (stored: the first raw byte)
      }
    }
    __syncthreads();
    KPROF_HDR_MARK(3);
    hdr_st = s_c_st;
    if (hdr_st == (uint32_t)ZH_OK && s_c_btype != 0u) {
      // the decode tables, by everybody (the staged header is done with: its bytes behind the first 320 words are scratch)
      const uint32_t hlit = s_c_hlit, hdist = s_c_hdist, dist_at = s_c_btype == 1u ? 288u : hlit;
      static_assert(kDistSub == 256u, "");
      const int t = build_tables_wg<kSplitThreads>(s_lens, hlit, s_lit, &s_tab_lit, s_val_lit, s_lens + dist_at, hdist, s_dst,
                                                   &s_tab_dist, s_val_dist, s_in + 320);
      hdr_st = (uint32_t)t;
    }
    KPROF_MARK(0);
  };
  // the superchunk that starts at the dword of `base_bit`, into s_in
  auto stage_super = [&](uint64_t base_bit) {
      {
        constexpr uint32_t kWords = kSuperBits / 32u + kSubWords;  // the superchunk + the row behind it
        constexpr uint32_t kLoads = (kWords + kSplitThreads - 1u) / kSplitThreads;  // 17 dwords a thread
        const uint64_t w0 = base_bit >> 5;
        uint32_t dw[kLoads];
        if ((w0 + kLoads * kSplitThreads) * 4 <= end) {  // all of it inside the input: plain loads
#pragma unroll
          for (uint32_t j = 0; j < kLoads; j++) dw[j] = asrc[w0 + tid + j * kSplitThreads];
        } else {
#pragma unroll
          for (uint32_t j = 0; j < kLoads; j++) dw[j] = load_dword((w0 + tid + j * kSplitThreads) * 4);
        }
#pragma unroll
        for (uint32_t j = 0; j < kLoads; j++) {
          const uint32_t w = tid + j * kSplitThreads;
          if (w < kWords) {
            const uint32_t at = w + 3u * (w / kSubWords);
            s_in[at] = dw[j];
            if ((w & (kSubWords - 1u)) < 3u && w >= kSubWords) s_in[at - 3u] = dw[j];  // the copy behind the subchunk before
          }
        }
      }
  };

  // segment mode: the next found start at or behind stream bit `rel` (segments without one are stepped
  // over; kSegNone is behind every position)
  auto next_start = [&](uint64_t rel) {
    while (seg_tj == sid || (seg_tj < seg_last && (seg_target == kSegNone || seg_target < rel))) {
      seg_tj++;
      seg_target = seg_tj < seg_last ? g.start_bit[seg_tj] : kSegNone;
    }
  };

  if (kSeg && phase == 0) {
    // ---- phase 0: a sub-start for this segment ----
    // The block it lies in, if a found start tells: the nearest one before it -- if that IS a block's start.  Bits of a
    // payload that read like a header (about one a GiB; dozens in a stream of literals only) mislead every segment
    // from there to their block's end.  Tables that are not the payload's all but never bring the 64 decoders below
    // in step (and what does not even read as a dynamic block, the only kind the search finds, is no header at all),
    // while a real block's tables all but always do: when a header yields nothing it is doubted, not the method, and
    // the found start before it is asked as well.  (What this lets through, its own decoder gives away:
    // zh_seg_repair_kernel.)
    uint64_t hdr = kSegNone, found = kSegNone, payload = 0;
    const uint64_t target = (uint64_t)mis * 8 + g.nominal_bit[sid];
    const uint64_t end_bit = end * 8;
    bool readable = false;
    for (uint32_t k = sid, lo = g.first_seg[bid], n = 0, tried = 0; k > lo && n < 256u && tried < 2u && target < end_bit; n++) {
      k--;
      const uint64_t sk = g.start_bit[k];
      if (sk == kSegNone) continue;
      hdr = sk;
      // (a header less than a segment and a half back: the decoder that starts there is about to arrive
      // anyway -- ordinary blocks of a few tens of KiB -- and parsing it once more costs more than it saves)
      if (g.nominal_bit[sid] - hdr < g.search_bits[sid] + g.search_bits[sid] / 2) break;
      tried++;
      pos = (uint64_t)mis * 8 + hdr;
      parse_header();
      payload = s_c_pos;
      // (the search only ever finds dynamic blocks; other kinds begin a stream -- or sit where ZH_SEG_FAKE_START put them)
      readable = hdr_st == (uint32_t)ZH_OK && (s_c_btype == 2u || (k == lo && s_c_btype != 0u)) && payload < target;
      if (readable) {
        constexpr uint64_t kBefore = 4096;  // bits of run-up
        const bool exact = payload + kBefore >= target;  // (the block starts that close: no guessing)
        const uint64_t s0 = exact ? payload : target - kBefore;
        const uint64_t base_bit = s0 & ~(uint64_t)31;
        __syncthreads();
        stage_super(base_bit);
        __syncthreads();
        const uint32_t end_rel = (uint32_t)(end_bit - base_bit < 0xfffffff0ull ? end_bit - base_bit : 0xfffffff0ull);
        if (tid < 64u) {
          // 64 decoders a bit apart: after 4096 bits they have all fallen in step with the block's
          // real token sequence, or this is no place to start
          const uint32_t from = (uint32_t)(s0 - base_bit) + (exact ? 0u : tid);
          const RunResult r = run(std::false_type{}, from, (uint32_t)(target - base_bit), end_rel, nullptr);
          const uint32_t e0 = zh_bcast(r.end);
          const uint64_t ok = __ballot(r.term == 0u && r.end == e0);
          if ((ok & 1ull) && __popcll(ok) >= 56) found = base_bit + e0 - (uint64_t)mis * 8;
        }
      }
      // (every thread has to know: the next attempt is the workgroup's)
      if (tid == 0) s_wbytes[0] = found != kSegNone ? 1u : 0u;
      __syncthreads();
      const bool got = s_wbytes[0] != 0u;
      __syncthreads();
      if (got) break;
      if (k == lo) break;  // (nothing lies before the stream's first block)
    }
    if (tid == 0) {
#ifdef ZH_EMU
      if (getenv("ZH_DBG_SUB"))
        fprintf(stderr, "  segment %u (nominal %llu): header %lld readable %d payload %llu -> sub-start %lld\n", sid,
                (unsigned long long)g.nominal_bit[sid], (long long)hdr, (int)readable, (unsigned long long)payload, (long long)found);
#endif
      g.sub_start[sid] = found;
      g.sub_hdr[sid] = hdr;
    }
    return;
  }

  while (!final_block && st == ZH_OK) {  // inflate.nim:273-289
    if (kSeg && !sub_first) {
      const uint64_t rel = pos - (uint64_t)mis * 8;
      next_start(rel);
#ifdef ZH_EMU
      if (tid == 0 && getenv("ZH_DBG_BLOCKS") && atoi(getenv("ZH_DBG_BLOCKS")) > 1)
        fprintf(stderr, "  region %u at %llu: next start %llu (segment %u), %llu tokens of %llu\n", sid, (unsigned long long)rel,
                (unsigned long long)seg_target, seg_tj, (unsigned long long)ntok, (unsigned long long)cap);
#endif
      if (seg_target == rel && !g.is_sub[seg_tj]) break;  // that segment's decoder takes over
    }
    const uint64_t hdr_at = pos;
    parse_header();
    if (kSeg) cur_hdr = hdr_at - (uint64_t)mis * 8;
    const uint32_t btype = s_c_btype;
#ifdef ZH_EMU
    if (tid == 0 && getenv("ZH_DBG_BLOCKS"))
      fprintf(stderr, "block at bit %llu type %u (region %u)\n", (unsigned long long)(hdr_at - mis * 8), btype, sid);
#endif
    st = (int)hdr_st;
    if (s_c_final) final_block = true;
    pos = s_c_pos;
    if (st != ZH_OK) break;
    if (kSeg && sub_first) {  // this decoder starts inside the block
      sub_first = false;
      if (btype == 0) {
        st = ZH_ERR_INVALID_BUFFER;
        break;
      }
      pos = (uint64_t)mis * 8 + seg_start;
    }

    if (btype == 0) {
      const uint32_t len = s_c_stored_len;
      const uint64_t byte_pos = pos >> 3;  // from asrc
      if (len) {  // (an empty stored block leaves no record: every record group makes output)
        if (!kCount && ntok + 3 + 1 > cap) {
          st = ZH_ERR_DST_TOO_SMALL;
          break;
        }
        if (!kCount && tid == 0) {
          const uint64_t off = byte_pos - mis;  // from the stream's first byte
          tok[ntok] = kRecSpecial | kRecStored | (len << 16);
          tok[ntok + 1] = (uint32_t)off;
          tok[ntok + 2] = (uint32_t)(off >> 32);
        }
        ntok += 3;
        out_bytes += len;
      }
      pos = (byte_pos + len) * 8;
      // Stored blocks come in chains -- incompressible data is one of 65 535 bytes after the other, 16 K of them a
      // GiB (deflate.nim:186-199), each a header read with the whole workgroup waiting at three barriers --, and
      // behind a stored block the next header is byte-aligned: wave 0 reads 64 of them at once, lane k where the
      // k-th would be if all before it were full, and keeps the lanes up to the first that is not a full block of a
      // chain that goes on (anything that is not a clean stored block is left to the header reader above, which
      // knows the reference's error for it).  One record a block, as before.  In segment mode a decoder stops at the
      // found start it lands on: a step goes no further than the next one (a found start inside a block's bytes is a
      // wrong guess and is run past with the block).
      if (len == ZH_STORED_MAX && !final_block) {
        for (;;) {
          uint64_t seg_stop = ~0ull;  // (stream bit: the chain takes no block that starts at or behind it)
          if (kSeg) {
            const uint64_t rel = pos - (uint64_t)mis * 8;
            next_start(rel);
            if (seg_target == rel && !g.is_sub[seg_tj]) break;  // (the loop above hands over to that segment's decoder)
            if (seg_tj < seg_last && seg_target != kSegNone && seg_target > rel) seg_stop = seg_target;
          }
          // every wave has read the control words the header reader left (s_c_stored_len above all: a wave that
          // read wave 0's count instead would skip the chain and leave the others at its barrier) before wave 0
          // overwrites them for the first round
          __syncthreads();
          if (tid < 64u) {
            const uint64_t o = (pos >> 3) + (uint64_t)lane * (ZH_STORED_MAX + 5u);  // (from asrc)
            const uint32_t w0 = load_dword(o & ~(uint64_t)3), w1 = load_dword((o & ~(uint64_t)3) + 4u),
                           w2 = load_dword((o & ~(uint64_t)3) + 8u);
            const uint32_t sh = (uint32_t)(o & 3u) * 8u;
            const uint64_t five = (((uint64_t)zh_alignbit(w2, w1, sh) << 32) | zh_alignbit(w1, w0, sh)) & 0xffffffffffull;
            const uint32_t h = (uint32_t)five & 0xffu, blen = (uint32_t)(five >> 8) & 0xffffu, nlen = (uint32_t)(five >> 24) & 0xffffu;
            const bool ok = o + 5u <= end && ((h >> 1) & 3u) == 0u && blen + nlen == 65535u && o + 5u + blen <= end &&
                            (!kSeg || (o - mis) * 8u < seg_stop);
            const bool full = ok && blen == ZH_STORED_MAX && !(h & 1u);
            const uint64_t fulls = __ballot(full);
            const uint32_t m = ~fulls ? (uint32_t)__builtin_ctzll(~fulls) : 64u;  // lanes 0 .. m - 1: full blocks of the chain
            const bool tail_ok = m < 64u && ((__ballot(ok) >> m) & 1ull);         // lane m: a last, shorter or final one
            uint32_t cnt = m + (tail_ok ? 1u : 0u);
            // records: the blocks that still fit the region (the check a block had: ntok + 3 + 1 <= cap), none for an
            // empty block
            const uint32_t tail_len = (uint32_t)__shfl((int)blen, (int)(m < 64u ? m : 0u), 64);
            const uint32_t nrec = cnt - (tail_ok && tail_len == 0u ? 1u : 0u);
            const uint64_t room = kCount ? 64u : cap > ntok + 4u ? (cap - ntok - 4u) / 3u + 1u : (cap == ntok + 4u ? 1u : 0u);
            const bool over = nrec > room;
            const uint32_t nfit = over ? (uint32_t)room : nrec;
            if (!kCount && lane < nfit) {
              const uint64_t off = o + 5u - mis;
              tok[ntok + 3u * lane] = kRecSpecial | kRecStored | (blen << 16);
              tok[ntok + 3u * lane + 1u] = (uint32_t)off;
              tok[ntok + 3u * lane + 2u] = (uint32_t)(off >> 32);
            }
            if (over) cnt = nfit;  // (the stream ends here: ZH_ERR_DST_TOO_SMALL)
            if (lane == 0) {
              const uint32_t lastl = cnt ? cnt - 1u : 0u;
              s_c_stored_len = cnt;                                       // blocks taken
              s_c_term = over ? 1u : 0u;
              s_c_endrel = nfit;                                          // records written
            }
            if (cnt && lane == cnt - 1u) {
              s_c_pos = (o + 5u + blen) * 8u;                             // behind the last block taken
              s_c_final = h & 1u;
            }
            if (kSeg || kCount) {  // the bytes the records make (blocks 0 .. nfit - 1; an empty last one adds nothing)
              const uint32_t made = zh_wave_sum(lane < nfit ? blen : 0u);
              if (lane == 0) s_wbytes[0] = made;
            }
          }
          __syncthreads();
          if (kSeg || kCount) out_bytes += s_wbytes[0];
          const uint32_t took = s_c_stored_len, nrec = s_c_endrel;
          const bool over = s_c_term != 0u;
          if (took) {
            pos = s_c_pos;
            if (s_c_final) final_block = true;
          }
          ntok += 3u * nrec;
          __syncthreads();  // (the control words are read: the next round, or the header reader, may write them)
          if (over) {
            st = ZH_ERR_DST_TOO_SMALL;
            break;
          }
          if (took < 64u || final_block) break;  // (a block that is not a full one of the chain: the ordinary way)
        }
        if (st != ZH_OK) break;
      }
      continue;
    }

    // ---- the block's codes, a superchunk at a time ----
    for (;;) {
      const uint64_t base_bit = pos & ~(uint64_t)31;
      const uint32_t rel0 = (uint32_t)(pos - base_bit);
      // segment mode: a sub-start of this block ahead?  The superchunk is then cut there: the
      // thread that covers the position decodes up to it, the threads behind sit out.
      uint32_t cut_rel = 0xffffffffu, cut_t = kSplitThreads;
      if (kSeg) {
        const uint64_t rel = pos - (uint64_t)mis * 8;
        next_start(rel);
        if (seg_tj < seg_last && seg_target == rel && g.is_sub[seg_tj] && g.sub_hdr[seg_tj] == cur_hdr) {
          seg_landed = true;  // landed on it: that segment's decoder takes over
          break;
        }
        // the first sub-start of THIS block inside the superchunk.  Found starts before it that are no such thing --
        // bits of the payload that read like a header: the decoder runs past them -- must not hide it: without the
        // cut this decoder takes the rest of the superchunk, the tokens of a dozen segments that have decoders
        // and regions of their own, and runs out of room.
        const uint64_t super_end = base_bit + kSuperBits - (uint64_t)mis * 8;  // (stream bits)
        for (uint32_t cj = seg_tj; cj < seg_last && g.nominal_bit[cj] < super_end; cj++) {
          const uint64_t ct = cj == seg_tj ? seg_target : g.start_bit[cj];
          if (ct == kSegNone || ct <= rel || !g.is_sub[cj] || g.sub_hdr[cj] != cur_hdr) continue;
          if (ct < super_end) {
            cut_rel = (uint32_t)(ct + (uint64_t)mis * 8 - base_bit);
            cut_t = (cut_rel - 1u) / kSubBits;
          }
          break;
        }
      }
      __syncthreads();  // (everybody is done with the previous contents of s_in)
      stage_super(base_bit);
      if (tid == 0) {
        s_first_dirty[0] = s_first_dirty[1] = kSplitThreads;
        s_first_term[0] = s_first_term[1] = kSplitThreads;
      }
      const uint64_t end_bit = end * 8;
      const uint32_t end_rel = end_bit > base_bit
                                   ? (uint32_t)(end_bit - base_bit < 0xfffffff0ull ? end_bit - base_bit : 0xfffffff0ull)
                                   : 0u;
      __syncthreads();
#ifdef ZH_KPROF_HDR
      KPROF_MARK(2);
#else
      KPROF_MARK(1);
#endif
      KPROF_COUNT(5, 1);
      const uint32_t limit = kSeg && tid == cut_t ? cut_rel : (tid + 1u) * kSubBits;
      uint32_t my_start = tid == 0 ? rel0 : (kSeg && tid > cut_t ? kNoStart : tid * kSubBits);
      // A run-up: a thread first decodes the last kRunUp bits of the subchunk before its own, from a
      // guessed bit; by the time it crosses into its subchunk it has usually fallen in step, and the
      // boundary it crosses at is the start the thread before will hand it -- no second turn for it.
      if (kRunUp && tid != 0 && my_start != kNoStart) {
        const uint32_t from = tid * kSubBits - kRunUp;  // (kRunUp <= kSubBits; thread 1 may as well start where thread 0 does)
        const RunResult pre = run(std::true_type{}, from > rel0 ? from : rel0, tid * kSubBits, end_rel, nullptr);
        if (pre.term == 0u) my_start = pre.end;
      }
      bool dirty = true;
      RunResult r = {0, 0, 0, 0};
      // A turn: threads whose start changed decode again; then every thread takes the end of the
      // thread before it as its start.  Threads below the first one whose start changed ("dirty")
      // have starts that follow from thread 0's exact one: they are final.  The superchunk is done
      // when that final prefix reaches its end, or a thread that met the end of the block (or an
      // error) -- what lies behind that is not this block's code and never settles.
      uint32_t tterm = kSplitThreads, prev_fd = 0;
      for (uint32_t turn = 1;; turn++) {
        const uint32_t par = turn & 1u;
        if (dirty) {
          if (my_start == kNoStart) {
            r.end = kNoStart;
            r.n = 0;
            r.term = 0;
          } else {
            r = run(std::true_type{}, my_start, limit, end_rel, nullptr);
          }
          s_end[tid] = r.term ? kNoStart : r.end;
        }
        __syncthreads();
        const uint32_t ns = tid == 0 ? rel0 : (kSeg && tid > cut_t ? kNoStart : s_end[tid - 1u]);
        dirty = ns != my_start;
        my_start = ns;
        if (dirty) atomicMin(&s_first_dirty[par], tid);
        else if (my_start != kNoStart && r.term) atomicMin(&s_first_term[par], tid);
        if (tid == 0) {  // the other parity's words are free: nobody reads them before the next barrier
          s_first_dirty[par ^ 1u] = kSplitThreads;
          s_first_term[par ^ 1u] = kSplitThreads;
        }
        __syncthreads();
        KPROF_COUNT(6, 1);
        const uint32_t fd = s_first_dirty[par], ft = s_first_term[par];
        if (ft < fd || fd == kSplitThreads) {
          tterm = ft < fd ? ft : kSplitThreads;
#ifdef ZH_EMU
          if (tid == 0 && getenv("ZH_DBG_TURNS")) fprintf(stderr, "superchunk turns %u\n", turn);
#endif
          break;
        }
        const uint32_t gain = fd - prev_fd;
        prev_fd = fd;
        if (turn < kMinTurns || gain >= kSlowGain) continue;
        // ---- the speculation has not settled (periodic data keeps a wrong start out of step for
        // ever: the final prefix would grow by one thread a turn).  All-starts pass over the threads
        // from `fd` on: a wave takes a subchunk, its lanes 64 consecutive bit positions, blocks of
        // 64 positions from the subchunk's last to its first; every lane decodes the token at its
        // position, and where that token leads is known already (a later lane of the block, by
        // pointer doubling, or the block done before).  The first 48 positions' results are the
        // subchunk's map start -> end.
        KPROF_COUNT(5, 1u << 16);
        uint32_t map_end = fd + kMapGroup < kSplitThreads ? fd + kMapGroup : kSplitThreads;
        if (kSeg && map_end > cut_t) map_end = cut_t;  // (the cut thread's subchunk ends early: it settles turn by turn)
        if (map_end <= fd) continue;
        for (uint32_t t = fd + (tid >> 6); t < map_end; t += kSplitThreads / 64u) {
          const uint32_t t_lim = (t + 1u) * kSubBits;
          uint32_t next_e = 0;  // block behind this one: lane l = result of position blockEnd + l
          for (int b = (int)(kSubBits / 64u) - 1; b >= 0; b--) {
            const uint32_t p = t * kSubBits + (uint32_t)b * 64u + lane, block_end = t * kSubBits + (uint32_t)b * 64u + 64u;
            uint32_t rec, kind;
            const uint32_t tb = decode_at(p, &rec, &kind);
            const uint32_t q = p + tb;
            // state: 0x100 | result once known, else the lane (of this block) the token leads to
            uint32_t v;
            if (kind != 0u || q > end_rel) v = 0x100u | kMapTerm;
            else if (q >= t_lim) v = 0x100u | (q - t_lim);
            else if (q >= block_end) v = 0x200u | (q - block_end);  // a lane of the block behind
            else v = q - (block_end - 64u);
            const uint32_t from_next = (uint32_t)__shfl((int)next_e, (int)(v & 63u), 64);
            if (v & 0x200u) v = 0x100u | from_next;
            while (__ballot(v < 0x100u)) {
              const uint32_t w = (uint32_t)__shfl((int)v, (int)(v & 63u), 64);
              if (v < 0x100u) v = w;
            }
            next_e = v & 0xffu;
          }
          if (lane < kStartSpan) s_map[t - fd][lane] = (uint8_t)next_e;
        }
        __syncthreads();
        if (tid == 0) {  // follow the maps from the last final end
          uint32_t sp = s_end[fd - 1u];  // (thread 0 is never dirty: fd >= 1)
          for (uint32_t t = fd; t < map_end; t++) {
            uint32_t e = kNoStart;
            if (sp != kNoStart) {
              const uint32_t m = s_map[t - fd][sp - t * kSubBits];
              if (m != kMapTerm) e = (t + 1u) * kSubBits + m;
            }
            s_end[t] = e;
            sp = e;
          }
        }
        __syncthreads();
        {  // the mapped threads (and the one behind them) take their exact starts; whoever's start
           // changed decodes again in the next turn, which also settles what lies behind the group
          const uint32_t ns2 = tid == 0 ? rel0 : (kSeg && tid > cut_t ? kNoStart : s_end[tid - 1u]);
          if (ns2 != my_start) {
            my_start = ns2;
            dirty = true;
          }
        }
        prev_fd = map_end - 1u;  // (the group is final after the next turn: judge its gain from there)
      }
      KPROF_MARK(2);
      const bool active = my_start != kNoStart && tid <= tterm;
      const uint32_t n_eff = active ? r.n : 0u;
      const uint32_t incl = zh_wave_scan(n_eff);
      if (lane == 63) s_wsum[tid >> 6] = incl;
      if (tid == tterm) {
        s_c_term = r.term;
        s_c_endrel = r.end;
      }
      __syncthreads();
      uint32_t before = incl - n_eff, total = 0;
      for (uint32_t w = 0; w < kWaves; w++) {
        const uint32_t ws = s_wsum[w];
        if (w < (tid >> 6)) before += ws;
        total += ws;
      }
      if (!kCount && ntok + total + 1 > cap) {  // more tokens than output bytes fit the slot
        st = ZH_ERR_DST_TOO_SMALL;
        break;
      }
#ifdef ZH_KPROF_HDR
      KPROF_MARK(2);
#else
      KPROF_MARK(3);
#endif
      // (parking every run's records in HBM and copying them into place here was tried: the
      // scattered 4-byte stores of the speculative turns cost more than this second decode)
      uint32_t made = 0;
      if (active && r.n) made = run(std::true_type{}, my_start, limit, end_rel, kCount ? nullptr : tok + ntok + before).bytes;
      if (kSeg || kCount) {
        made = zh_wave_sum(made);
        if (lane == 0) s_wbytes[tid >> 6] = made;
        __syncthreads();
        for (uint32_t w = 0; w < kWaves; w++) out_bytes += s_wbytes[w];
      }
      KPROF_MARK(4);
      ntok += total;
      if (tterm < kSplitThreads) {
        const uint32_t term = s_c_term;
        if (term != 1u) st = (int)term;
        pos = base_bit + s_c_endrel;  // behind the end-of-block code
        break;
      }
      pos = base_bit + s_end[kSeg && cut_t < kSplitThreads ? cut_t : kSplitThreads - 1u];
    }
    if (kSeg && seg_landed) break;
  }
  if (tid == 0) {
    if (!kCount) tok[ntok] = kRecSpecial | kRecEnd | ((uint32_t)st << 16);
    if (kCount) {  // the sizing pass's answer (what zh_inflate_kernel leaves when it only counts)
      a.out_len[sid] = out_bytes;
      a.status[sid] = st;
    }
    if (kSeg) {
      g.end_bit[sid] = pos - (uint64_t)mis * 8;
      g.final_block[sid] = final_block && !seg_landed ? 1u : 0u;  // (a decoder that hands over inside the last block has not ended the stream)
      g.seg_status[sid] = st;
      g.seg_out[sid] = out_bytes;
    }
  }
  KPROF_COUNT(7, 1);
  if (tid == 0) KPROF_FLUSH(48, 8);
}

// ---------------------------------------------------------------------------
// Writer: four waves per stream turn the token records into bytes, up to 1024 output bytes a
// round; a thread owns FOUR consecutive output bytes (one unaligned dword store).  A round takes up
// to 512 records (two per thread) whose output fits: a prefix sum places them, their starts are
// scattered into a byte -> record map in LDS and a running maximum tells every byte its record.  A
// byte is a literal, a copy of a byte before the round (read back through L2: the LZ window is the
// output itself), or a copy of a byte of this round -- those chase their source by pointer
// doubling in LDS, which is inflate.nim:227-250's byte-sequential copy semantics (overlapping
// copies included) without a loop over the tokens.  A stream is a chain of rounds, each a chain of
// LDS and L2 round trips: wide rounds are what shortens it (64-byte rounds by one wave: 13 ms for
// a 1 MiB stream whatever the batch).  A single copy that does not fit a round goes alone.
// ---------------------------------------------------------------------------
// (Round 6 measured a round with FEWER BARRIERS -- the cut (records that fit, their bytes) and the waves' carries written by
// the threads that own them instead of through per-wave sums and a barrier each, the ring's commit merged into the round's
// last barrier: 4 + k barriers a round instead of 7 + k, byte-identical -- and it was SLOWER, 12.00 -> 12.42 ms for 4096 x
// 1 MiB in one launch (the merged barrier alone: 12.11; the owner-written cut alone: 12.44): with five workgroups a CU a
// barrier is time another workgroup fills, the two dozen instructions that replaced two of them are not.  profiles/r06_d_*.)
// kWrThreads: 256 (batches), 512 (a few hundred streams) or 1024 (up to a stream per CU); a round is kB bytes a thread.
// (Round 4 measured the same kernel on ONE wave a stream -- 512-byte rounds, no barrier that costs anything, all 4096
// streams of the bench batch resident at once, 16 a CU --: 15.95 ms against 13.57 on the same box, and 7.8 ms against
// 2.3 for 512 streams: a round is a chain of ~ 9 000 cycles for a lone wave, and four waves a SIMD do not hide it.)
// kSeg: a workgroup writes a chain SEGMENT of a stream (zh_inflate_seg.hip) as 16-bit symbols into
// g.sym: a byte, or -- for a byte copied from the 32 KiB before the segment, which some other
// workgroup is writing at the same time -- 0x8000 | its index in that window.  Copies of symbols are
// copies whatever the symbol is; zh_seg_windows_kernel / zh_seg_finish_kernel turn them into bytes.
template <uint32_t kWrThreads, bool kSeg, uint32_t kB>
__global__ __launch_bounds__(kWrThreads, kWrThreads == 256 ? ZH_WR_OCC : kWrThreads == 512 ? 4 : 1) void zh_inflate_write_kernel(const uint8_t* __restrict__ d_src,
                                                               uint8_t* __restrict__ d_dst, ZhInflateArgs a,
                                                               const uint32_t* __restrict__ tok_pool,
                                                               const uint64_t* __restrict__ tok_off,
                                                               ZhSegArgs g) {
  typedef typename std::conditional<kSeg, uint16_t, uint8_t>::type Sym;
  constexpr uint32_t kWrWaves = kWrThreads / 64u;
  constexpr uint32_t kR = kB / 2u;                    // records a thread looks at per round (kB: output bytes it owns)
  constexpr uint32_t kWrRound = kWrThreads * kB;      // output bytes per round
  constexpr uint32_t kWrRecs = kWrThreads * kR;       // records looked at per round
  constexpr uint32_t kWrRing = kWrThreads * kR * 4u;  // token records staged in LDS (a power of two)
  __shared__ uint32_t s_tok[kWrRing];
  __shared__ uint32_t s_map32[kWrRound / 2];  // u16 per byte: (index in the round of the record that starts there) + 1
  __shared__ uint32_t s_par32[2][kWrRound / 2];  // u16 per byte: the byte of this round it copies (itself: a root); two copies take turns
  __shared__ uint32_t s_val32[kWrRound * sizeof(Sym) / 4];  // a symbol per byte: its value (valid for roots)
  __shared__ uint32_t s_w[6][kWrWaves];       // per-wave partial results
  __shared__ uint32_t s_flag[4];              // round-wide flags (see below)
  uint16_t* const s_map = reinterpret_cast<uint16_t*>(s_map32);
  Sym* const s_val = reinterpret_cast<Sym*>(s_val32);
  const uint32_t tid = threadIdx.x, wv = tid >> 6;
  const unsigned lane = zh_lane();
  const uint32_t sid = kSeg ? blockIdx.x : a.first_buf + blockIdx.x;
  const uint32_t bid = kSeg ? g.parent[sid] : sid;
  if (a.status[bid] != ZH_OK) return;
  if (kSeg ? !g.valid[sid] : (a.skip && a.skip[sid])) return;

  const ZhBufDesc bd = a.bufs[bid];
  const uint8_t* src = d_src + bd.src_off;
  const uint64_t gbase = kSeg ? g.out_start[sid] : 0;  // output bytes of the stream before this workgroup's
  Sym* dst = kSeg ? reinterpret_cast<Sym*>(g.sym + g.sym_base[bid] + gbase) : reinterpret_cast<Sym*>(d_dst + bd.dst_off);
  const uint64_t cap = bd.dst_cap - gbase;
  const uint32_t* tok = tok_pool + (kSeg ? g.eff_tok_off[sid] : tok_off[sid]);

  uint64_t op = 0;  // bytes produced (the same in every thread)
  int st = ZH_OK;
  auto ld_out = [&](int64_t at) -> uint32_t {
    if (kSeg && at < 0) return 0x8000u | (uint32_t)(32768 + at);  // (a distance is at most 32768)
    return __hip_atomic_load(dst + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // every thread's stores have reached L2 and every thread knows it: match sources are read back
  // past this CU's L1 (which may hold a stale copy of a line the workgroup has since extended)
  auto output_visible = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
  };
  // inflate.nim:224-250: one LZ copy of `length` bytes from `dist` back, at op (uniform arguments;
  // the output written so far is visible)
  auto lz_copy = [&](uint32_t length, uint32_t dist) {
    if (dist > gbase + op) {  // inflate.nim:224-225
      st = ZH_ERR_INVALID_BUFFER;
      return;
    }
    if (op + length > cap) {
      st = ZH_ERR_DST_TOO_SMALL;
      return;
    }
    const int64_t sb = (int64_t)op - dist;
    if (dist >= length) {
      for (uint32_t i = tid; i < length; i += kWrThreads) dst[op + i] = (Sym)ld_out(sb + i);
    } else if (dist == 1) {
      const Sym v = (Sym)ld_out(sb);
      for (uint32_t i = tid; i < length; i += kWrThreads) dst[op + i] = v;
    } else {
      for (uint32_t i = tid; i < length; i += kWrThreads) dst[op + i] = (Sym)ld_out(sb + i % dist);
    }
    op += length;
  };

  // records [ti, hi) are in the ring; `pre` holds records [hi, hi + 2 kR threads) on their way from HBM
  uint64_t ti = 0, hi = 0;
  uint32_t pre[2 * kR];
  auto fetch_ahead = [&]() {
#pragma unroll
    for (uint32_t k = 0; k < 2 * kR; k++) pre[k] = tok[hi + tid + kWrThreads * k];
  };
  auto commit_ahead = [&]() {  // (callers keep a barrier between this and the ring's readers)
#pragma unroll
    for (uint32_t k = 0; k < 2 * kR; k++) s_tok[(uint32_t)(hi + tid + kWrThreads * k) & (kWrRing - 1u)] = pre[k];
    hi += 2u * kR * kWrThreads;
  };
  fetch_ahead();
  if (tid < 4) s_flag[tid] = 0;
#pragma unroll
  for (uint32_t j = 0; j < kB / 2u; j++) s_map32[(kB / 2u) * tid + j] = 0;

  for (uint32_t round = 0;; round++) {
    if (tid < 4) s_w[5][tid] = 0;  // (the pointer-doubling loop's flags)
    if (hi < ti + kWrRecs + 128u) {  // a round looks at kWrRecs records (+ 2 behind a stored-run record)
      commit_ahead();
      fetch_ahead();
    }
    __syncthreads();
    const uint32_t i0 = kR * tid;
    uint32_t r[kR];
#pragma unroll
    for (uint32_t k = 0; k < kR; k++) r[k] = s_tok[(uint32_t)(ti + i0 + k) & (kWrRing - 1u)];
    // ---- the records that fit the round: a prefix, which ends at the first special record at the
    // latest (it counts as longer than a round, so neither it nor anything behind it fits) ----
    uint32_t len[kR], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kR; k++) {
      len[k] = r[k] & kRecSpecial ? kWrRound + 1u : r[k] & 0x1ffu;
      sum += len[k];
    }
    const uint32_t incl = zh_wave_scan(sum);
    if (lane == 63) s_w[1][wv] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWrWaves; w++)
      if (w < wv) before += s_w[1][w];
    // where the records' output starts in the round; the records that fit are a prefix (the sums grow)
    uint32_t o[kR];
    bool fit[kR];
    {
      uint32_t at = before + incl - sum, endl = 0, nfit = 0;
      bool bad = false;
#pragma unroll
      for (uint32_t k = 0; k < kR; k++) {
        o[k] = at;
        fit[k] = at + len[k] <= kWrRound;
        if (fit[k]) endl = at + len[k];
        // inflate.nim:224-225 `distance > op` (a distance is at most 32768)
        bad = bad || (fit[k] && !((r[k] >> 9) & 1u) && (uint64_t)(r[k] >> 16) > gbase + op + at);
        nfit += (uint32_t)__popcll(__ballot(fit[k]));
        at += len[k];
      }
      const uint64_t b0 = __ballot(fit[0]);
      const uint32_t wend = b0 ? (uint32_t)__builtin_amdgcn_readlane(endl, (uint32_t)__popcll(b0) - 1u) : 0u;
      if (lane == 0) {
        s_w[2][wv] = nfit;
        s_w[3][wv] = wend;
      }
      if (gbase + op < 32768u && bad) s_flag[round & 1u] = 1;  // (flag words alternate between rounds; the idle one is cleared below)
    }
    __syncthreads();
    uint32_t n = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWrWaves; w++) {
      n += s_w[2][w];
      total = max(total, s_w[3][w]);
    }
    const bool bad_dist = s_flag[round & 1u] != 0;
    if (tid == 0) {
      s_flag[(round & 1u) ^ 1u] = 0;
      s_flag[2u + ((round & 1u) ^ 1u)] = 0;
    }
    if (n == 0) {
      const uint32_t q0 = s_tok[(uint32_t)ti & (kWrRing - 1u)];
      if (q0 & kRecSpecial) {
        if ((q0 & (3u << 11)) == kRecStored) {  // inflate.nim:252-266: raw bytes
          if (kSeg) {
            const uint32_t length = q0 >> 16;
            const uint64_t off = (uint64_t)s_tok[(uint32_t)(ti + 1u) & (kWrRing - 1u)] |
                                 ((uint64_t)s_tok[(uint32_t)(ti + 2u) & (kWrRing - 1u)] << 32);
            if (op + length > cap) {
              st = ZH_ERR_DST_TOO_SMALL;
              break;
            }
            for (uint32_t i = tid; i < length; i += kWrThreads) dst[op + i] = (Sym)src[off + i];
            op += length;
            ti += 3;
          } else {
            // Stored blocks come in chains (incompressible data: 16 K of them a GiB): up to four records at a time,
            // their bytes sixteen at a time with four (one stream a workgroup: sixteen) loads of a thread in flight before the first store -- not a
            // round of this loop, with its scans and barriers, a block, and not a byte a thread and trip.
            struct __attribute__((packed)) V16 { uint32_t w[4]; };
            // (kept small: what this path holds in registers must not cost the rounds above theirs -- with eight records
            // and eight loads the 256-thread form spilt inside its round, 12.2 -> 13.7 ms on the bench batch, and the
            // 512-thread form lost a workgroup a CU)
            constexpr uint32_t kGroup = 4, kFly = kWrThreads >= 1024u ? 16u : 4u;
            bool bad = false;
            for (;;) {
              // (every array below is indexed by unrolled constants only: registers, not scratch)
              uint64_t goff[kGroup], gpre[kGroup];   // a record's source offset; the group's bytes before it
              uint32_t glen[kGroup], gch[kGroup + 1], ng = 0;  // its length; the group's whole pieces before it
              uint64_t gtotal = 0;
              bool open = true;
              gch[0] = 0;
#pragma unroll
              for (uint32_t k = 0; k < kGroup; k++) {
                goff[k] = 0;
                gpre[k] = gtotal;
                glen[k] = 0;
                gch[k + 1] = gch[k];
                if (open && ti + 3u * (k + 1u) <= hi) {
                  const uint32_t q = s_tok[(uint32_t)(ti + 3u * k) & (kWrRing - 1u)];
                  const uint32_t length = q >> 16;
                  if ((q & (kRecSpecial | (3u << 11))) != (kRecSpecial | kRecStored)) {
                    open = false;
                  } else if (op + gtotal + length > cap) {
                    bad = k == 0;  // (the first of a group: this is where the stream ends)
                    open = false;
                  } else {
                    glen[k] = length;
                    goff[k] = (uint64_t)s_tok[(uint32_t)(ti + 3u * k + 1u) & (kWrRing - 1u)] |
                              ((uint64_t)s_tok[(uint32_t)(ti + 3u * k + 2u) & (kWrRing - 1u)] << 32);
                    gch[k + 1] = gch[k] + (length >> 4);
                    gtotal += length;
                    ng = k + 1u;
                  }
                } else {
                  open = false;
                }
              }
              if (!ng) break;
              // whole 16-byte pieces, numbered through the group; the bytes behind a record's last whole piece one by one
              // piece e of the group -> where it comes from (`to` false) or goes (true); ~0: none
              auto piece = [&](uint32_t e, bool to) -> uint64_t {
                uint32_t rch = 0;
                uint64_t roff = goff[0], rpre = 0;
#pragma unroll
                for (uint32_t k = 1; k < kGroup; k++)
                  if (k < ng && e >= gch[k]) {
                    rch = gch[k];
                    roff = goff[k];
                    rpre = gpre[k];
                  }
                if (e >= gch[kGroup]) return to ? ~0ull : goff[0];
                return (to ? op + rpre : roff) + (uint64_t)(e - rch) * 16u;
              };
              for (uint32_t base = 0; base < gch[kGroup]; base += kWrThreads * kFly) {
                V16 v[kFly];
#pragma unroll
                for (uint32_t u = 0; u < kFly; u++) v[u] = *reinterpret_cast<const V16*>(src + piece(base + u * kWrThreads + tid, false));
#pragma unroll
                for (uint32_t u = 0; u < kFly; u++) {
                  const uint64_t at = piece(base + u * kWrThreads + tid, true);
                  if (at != ~0ull) *reinterpret_cast<V16*>(dst + at) = v[u];
                }
              }
#pragma unroll
              for (uint32_t k = 0; k < kGroup; k++) {
                const uint32_t done = glen[k] & ~15u;
                if (k < ng && tid < (glen[k] & 15u)) dst[op + gpre[k] + done + tid] = src[goff[k] + done + tid];
              }
              op += gtotal;
              ti += 3u * ng;
              if (hi < ti + kWrRecs + 128u) {
                commit_ahead();
                fetch_ahead();
              }
              __syncthreads();
            }
            if (bad) {
              st = ZH_ERR_DST_TOO_SMALL;
              break;
            }
          }
          output_visible();
          continue;
        }
        st = (int)(q0 >> 16);  // end of the stream
        break;
      }
      lz_copy(q0 & 0x1ffu, q0 >> 16);  // one copy that does not fit a round
      if (st != ZH_OK) break;
      ti += 1;
      output_visible();
      continue;
    }
    if (bad_dist) {
      st = ZH_ERR_INVALID_BUFFER;
      break;
    }
    if (op + total > cap) {
      st = ZH_ERR_DST_TOO_SMALL;
      break;
    }
    // ---- byte -> record (the map is all zeros here: whoever reads a word clears it) ----
#pragma unroll
    for (uint32_t k = 0; k < kR; k++)
      if (fit[k]) s_map[o[k]] = (uint16_t)(i0 + k + 1u);
    __syncthreads();
    uint32_t t[kB];
    uint32_t starts = 0;  // bit j: a record starts at this thread's byte j
    {
      uint32_t m = 0;
#pragma unroll
      for (uint32_t j = 0; j < kB / 2u; j++) {
        const uint32_t w = s_map32[(kB / 2u) * tid + j];
        s_map32[(kB / 2u) * tid + j] = 0;
        starts |= ((w & 0xffffu) ? 1u : 0u) << (2u * j) | ((w >> 16) ? 2u : 0u) << (2u * j);
        m = max(m, w & 0xffffu);
        t[2u * j] = m;
        m = max(m, w >> 16);
        t[2u * j + 1u] = m;
      }
    }
    const uint32_t run = zh_wave_scan_max(t[kB - 1u]);
    if (lane == 63) s_w[4][wv] = run;
    uint32_t carry = (uint32_t)__shfl_up((int)run, 1, 64);
    if (lane == 0) carry = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t w = 0; w < kWrWaves; w++)
      if (w < wv) carry = max(carry, s_w[4][w]);
    // ---- every byte: literal, copy from before the round (far), or from inside it (near) ----
    // Without a branch a byte (under each byte's own condition the compiler kept an exec mask a byte: a hundred
    // scalar instructions a round), and with the stream's output position in scalar registers -- it is the same in
    // every thread, but only the hardware knows --, so that a far byte's address is one base for the wave plus a
    // 32-bit offset a lane: a byte from before the round lies at most 32768 before the byte it makes.
    uint32_t par[kB], val[kB];
    uint32_t far_off[kB];  // where a byte copied from before the round comes from: bytes behind (op - 32768)
    uint32_t far_mask = 0, near_mask = 0;
    const uint64_t op_s = (uint64_t)zh_bcast((uint32_t)op) | ((uint64_t)zh_bcast((uint32_t)(op >> 32)) << 32);
    const Sym* const far_base = dst + ((int64_t)op_s - 32768);  // (pointer arithmetic: the loads stay global_load, a base for the wave + an offset a lane)
#pragma unroll
    for (uint32_t j = 0; j < kB; j++) {
      const uint32_t pb = kB * tid + j;       // byte of the round
      const uint32_t tk = max(carry, t[j]);  // its record + 1 (>= 1 for every live byte)
      const uint32_t rec = s_tok[(uint32_t)(ti + tk - 1u) & (kWrRing - 1u)];
      // a literal record holds one byte or two: its first at bit 16, the one behind it at bit 24
      val[j] = (rec >> ((starts >> j) & 1u ? 16u : 24u)) & 0xffu;
      const uint32_t dist = rec >> 16;
      const bool copy = pb < total && !((rec >> 9) & 1u);
      const bool near = copy && dist <= pb;
      const bool far = copy && dist > pb;
      par[j] = near ? pb - dist : pb;
      far_off[j] = far ? 32768u + pb - dist : 32768u;  // (no far byte: the round's first byte, read in vain)
      near_mask |= near ? 1u << j : 0u;
      far_mask |= far ? 1u << j : 0u;
    }
    const bool any_near = near_mask != 0u;
    // the far bytes of a thread are read TOGETHER: unconditional loads that are all in flight at once -- under
    // their bytes' conditions the compiler waited for each of the eight before it issued the next, eight trips
    // to L2 a round.  (Written before this round: visible since its start.)
    {
      uint32_t got[kB];
      // segment mode: a byte from before the segment is not there yet: it stays a reference into the window (ld_out's rule)
      uint32_t before_seg = 0;
      if (kSeg && op_s < 32768u) {
#pragma unroll
        for (uint32_t j = 0; j < kB; j++)
          before_seg |= ((far_mask >> j) & 1u) && (uint32_t)op_s + far_off[j] < 32768u ? 1u << j : 0u;
      }
#pragma unroll
      for (uint32_t j = 0; j < kB; j++)
        got[j] = __hip_atomic_load(far_base + ((before_seg >> j) & 1u ? 32768u : far_off[j]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (uint32_t j = 0; j < kB; j++)
        if ((far_mask >> j) & 1u) val[j] = (before_seg >> j) & 1u ? 0x8000u | ((uint32_t)op_s + far_off[j]) : got[j];
    }
    if (any_near) s_flag[2u + (round & 1u)] = 1;
#pragma unroll
    for (uint32_t j = 0; j < kB / 2u; j++) s_par32[0][(kB / 2u) * tid + j] = par[2u * j] | (par[2u * j + 1u] << 16);
    if (kSeg) {
#pragma unroll
      for (uint32_t j = 0; j < kB / 2u; j++)
        s_val32[(kB / 2u) * tid + j] = (val[2u * j] & 0xffffu) | (val[2u * j + 1u] << 16);
    } else {
#pragma unroll
      for (uint32_t j = 0; j < kB / 4u; j++)
        s_val32[(kB / 4u) * tid + j] = (val[4u * j] & 0xffu) | ((val[4u * j + 1u] & 0xffu) << 8) |
                                      ((val[4u * j + 2u] & 0xffu) << 16) | (val[4u * j + 3u] << 24);
    }
    __syncthreads();
    if (s_flag[2u + (round & 1u)]) {
      // par <- par[par] until every byte points at a root: a step reads one copy of the pointers and
      // writes the other, one barrier a step.  s_w[5][k & 3] says that somebody moved in step k (all
      // four are cleared when a round starts, and two steps ahead by thread 0 in long loops).
      for (uint32_t k = 0;; k++) {
        const uint16_t* const rd = reinterpret_cast<const uint16_t*>(s_par32[k & 1u]);
        bool changed = false;
#pragma unroll
        for (uint32_t j = 0; j < kB; j++) {
          const uint32_t q = rd[par[j]];
          changed |= q != par[j];
          par[j] = q;
        }
        if (changed) s_w[5][k & 3u] = 1;
#pragma unroll
        for (uint32_t j = 0; j < kB / 2u; j++)
          s_par32[(k & 1u) ^ 1u][(kB / 2u) * tid + j] = par[2u * j] | (par[2u * j + 1u] << 16);
        if (tid == 0) s_w[5][(k + 2u) & 3u] = 0;
        __syncthreads();
        if (!s_w[5][k & 3u]) break;
      }
#pragma unroll
      for (uint32_t j = 0; j < kB; j++) val[j] = s_val[par[j]];
    }
    const uint32_t pb0 = kB * tid;
    if (pb0 + kB <= total) {  // (gfx950 global stores need no alignment)
      struct __attribute__((packed)) U64 { uint64_t v; };
      struct __attribute__((packed)) U32 { uint32_t v; };
      if (kSeg) {
#pragma unroll
        for (uint32_t j = 0; j < kB / 4u; j++)
          reinterpret_cast<U64*>(dst + op + pb0 + 4u * j)->v =
              (uint64_t)((val[4u * j] & 0xffffu) | (val[4u * j + 1u] << 16)) |
              ((uint64_t)((val[4u * j + 2u] & 0xffffu) | (val[4u * j + 3u] << 16)) << 32);
      } else if (kB == 8u) {
        reinterpret_cast<U64*>(dst + op + pb0)->v =
            (uint64_t)((val[0] & 0xffu) | ((val[1] & 0xffu) << 8) | ((val[2] & 0xffu) << 16) | (val[3] << 24)) |
            ((uint64_t)((val[kB - 4u] & 0xffu) | ((val[kB - 3u] & 0xffu) << 8) | ((val[kB - 2u] & 0xffu) << 16) |
                        (val[kB - 1u] << 24)) << 32);
      } else {
        reinterpret_cast<U32*>(dst + op + pb0)->v =
            (val[0] & 0xffu) | ((val[1] & 0xffu) << 8) | ((val[2] & 0xffu) << 16) | (val[3] << 24);
      }
    } else {
#pragma unroll
      for (uint32_t j = 0; j < kB; j++)
        if (pb0 + j < total) dst[op + pb0 + j] = (Sym)val[j];
    }
    op += total;
    ti += n;
    output_visible();
  }
  if (kSeg && st == ZH_OK) {
    // the 32 KiB that end the segment, for zh_seg_windows_kernel: symbols of this segment, or --
    // the segment being shorter -- "byte k of the window before it"
    uint16_t* ws = g.winsym + (size_t)sid * 32768u;
    for (uint32_t j = tid; j < 32768u; j += kWrThreads) {
      const int64_t at = (int64_t)op + j - 32768;
      ws[j] = (uint16_t)(at >= 0 ? ld_out(at) : 0x8000u | (uint32_t)(j + op));
    }
  }
  if (tid == 0) {
    if (kSeg) {
      g.wr_len[sid] = op;
      if (st != ZH_OK) g.seg_status[sid] = st;
    } else {
      a.out_len[sid] = op;
      a.status[sid] = st;
    }
  }
}

// Few streams: one wide workgroup each (a stream is a chain of superchunks and of rounds, and wide
// ones shorten it); many: narrow workgroups, several per CU (ZH_INFLATE_WIDE = largest batch that
// still gets the wide form).
// -> threads per workgroup.  Few streams: one wide workgroup each (a stream is a chain of superchunks and of
// rounds, and wide ones shorten it); many: narrow workgroups, several per CU.  Measured on 1 MiB streams
// (tokens / writer, ms): 256 streams 1.5 / 1.5 wide, 2.1 / 2.0 with 512 threads, 3.8 / 3.2 narrow; 512 streams
// 2.8 / 2.9, 4.1 / 2.6, 4.0 / 3.6; 768 streams 4.0 / 4.4, 6.0 / 4.3, 4.2 / 4.1; 1024 streams 5.4 / 5.8, 8.1 / 5.0,
// 4.3 / 4.4.  ZH_INFLATE_WIDE / ZH_INFLATE_MID: the tokens kernel's / the writer's largest wide batch.
static uint32_t zh_tokens_width(uint32_t nbufs) {
  static const uint32_t wide = [] {
    const char* e = getenv("ZH_INFLATE_WIDE");
    return e ? (uint32_t)atoi(e) : 768u;
  }();
  return nbufs <= wide ? 1024u : 256u;
}
static uint32_t zh_writer_width(uint32_t nbufs) {
  static const uint32_t wide = [] {
    const char* e = getenv("ZH_INFLATE_WIDE");
    return e ? (uint32_t)atoi(e) : 256u;
  }();
  static const uint32_t mid = [] {
    const char* e = getenv("ZH_INFLATE_MID");
    return e ? (uint32_t)atoi(e) : 640u;
  }();
  return nbufs <= wide ? 1024u : nbufs <= mid ? 512u : 256u;
}

extern "C" void zh_launch_inflate_tokens(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a,
                                         uint32_t* tok_pool, const uint64_t* tok_off, const uint64_t* tok_cap) {
  if (!a.nbufs) return;
  if (zh_tokens_width(a.nbufs) == 1024u)
    hipLaunchKernelGGL((zh_inflate_tokens_kernel<1024, false>), dim3(a.nbufs), dim3(1024), 0, stream, d_src, a, tok_pool, tok_off, tok_cap, ZhSegArgs{}, 1);
  else {  // (phase 9: the streams of one block, 8: the others -- see the kernel)
    hipLaunchKernelGGL((zh_inflate_tokens_kernel<256, false>), dim3(a.nbufs), dim3(256), 0, stream, d_src, a, tok_pool, tok_off, tok_cap, ZhSegArgs{}, 9);
    hipLaunchKernelGGL((zh_inflate_tokens_kernel<128, false>), dim3(a.nbufs), dim3(128), 0, stream, d_src, a, tok_pool, tok_off, tok_cap, ZhSegArgs{}, 8);
  }
}
// A sizing pass over a batch (count_only): the same routing, no token pool.
extern "C" void zh_launch_inflate_count(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a) {
  if (!a.nbufs) return;
  if (zh_tokens_width(a.nbufs) == 1024u) {
    hipLaunchKernelGGL((zh_inflate_tokens_kernel<1024, false, true>), dim3(a.nbufs), dim3(1024), 0, stream, d_src, a, nullptr, nullptr, nullptr, ZhSegArgs{}, 1);
  } else {
    hipLaunchKernelGGL((zh_inflate_tokens_kernel<256, false, true>), dim3(a.nbufs), dim3(256), 0, stream, d_src, a, nullptr, nullptr, nullptr, ZhSegArgs{}, 9);
    hipLaunchKernelGGL((zh_inflate_tokens_kernel<128, false, true>), dim3(a.nbufs), dim3(128), 0, stream, d_src, a, nullptr, nullptr, nullptr, ZhSegArgs{}, 8);
  }
}
// phase 0: sub-starts for the segments inside long blocks
// (before zh_seg_decide_kernel); phase 1: the tokens; 4 / 5: the same once more for the streams under repair
extern "C" void zh_launch_seg_tokens(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a, uint32_t* tok_pool,
                                     ZhSegArgs g, int phase) {
  if (!g.nsegs) return;
  if (g.nsegs <= 512u)  // (segments are short: a wide workgroup only pays while there is a CU for each or so)
    hipLaunchKernelGGL((zh_inflate_tokens_kernel<1024, true>), dim3(g.nsegs), dim3(1024), 0, stream, d_src, a, tok_pool, nullptr, nullptr, g, phase);
  else
    hipLaunchKernelGGL((zh_inflate_tokens_kernel<256, true>), dim3(g.nsegs), dim3(256), 0, stream, d_src, a, tok_pool, nullptr, nullptr, g, phase);
}
extern "C" void zh_launch_seg_write(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a, const uint32_t* tok_pool,
                                    ZhSegArgs g) {
  if (!g.nsegs) return;
  if (g.nsegs <= 512u)
    hipLaunchKernelGGL((zh_inflate_write_kernel<1024, true, ZH_WR_BYTES>), dim3(g.nsegs), dim3(1024), 0, stream, d_src, nullptr, a, tok_pool, nullptr, g);
  else
    hipLaunchKernelGGL((zh_inflate_write_kernel<256, true, ZH_WR_BYTES>), dim3(g.nsegs), dim3(256), 0, stream, d_src, nullptr, a, tok_pool, nullptr, g);
}
extern "C" void zh_launch_inflate_write(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, ZhInflateArgs a,
                                        const uint32_t* tok_pool, const uint64_t* tok_off) {
  if (!a.nbufs) return;
  const uint32_t th = zh_writer_width(a.nbufs);
  if (th == 1024u)
    hipLaunchKernelGGL((zh_inflate_write_kernel<1024, false, ZH_WR_BYTES>), dim3(a.nbufs), dim3(1024), 0, stream, d_src, d_dst, a, tok_pool, tok_off, ZhSegArgs{});
  else if (th == 512u)
    hipLaunchKernelGGL((zh_inflate_write_kernel<512, false, ZH_WR_BYTES>), dim3(a.nbufs), dim3(512), 0, stream, d_src, d_dst, a, tok_pool, tok_off, ZhSegArgs{});
  else
    hipLaunchKernelGGL((zh_inflate_write_kernel<256, false, ZH_WR_BYTES>), dim3(a.nbufs), dim3(256), 0, stream, d_src, d_dst, a, tok_pool, tok_off, ZhSegArgs{});
}
