// Batched inflate: container unwrap + RFC 1951 decode, one workgroup of two 64-lane waves
// per stream (a foreign deflate stream has no block index, so the only parallelism across a
// stream is lane-level; throughput comes from thousands of streams -- or, for streams this
// library wrote with a block index, from zh_plan_uncompress_indexed, which makes every block
// a "stream" of this kernel).
//
// Replaces src/zippy.nim:100-165 (format detect / zlib header), gzip.nim:3-88
// (gzip header + trailer), inflate.nim:24-291 (Huffman tables, decode loop,
// stored blocks) and the BitStreamReader of bitstreams.nim:22-82.
//
// Per workgroup, in LDS (9.3 KiB, 16 workgroups = 32 waves per CU -- a wave alone on its SIMD
// issues one instruction per ~4.4 cycles, so resident waves are what buys throughput): a 10-bit
// literal/length root LUT with second-level tables behind it for longer codes and an 8-bit
// distance LUT, all of self-describing 32-bit entries (inflate.nim's 9-bit `fast` table,
// re-shaped), the canonical slow-path arrays (firstCode / firstSymbol / maxCodes / values,
// inflate.nim:14-19), a 512-byte staging ring of the compressed stream and two round
// descriptors.  There is no window copy: the LZ window is the output itself, read back through
// L2.  The decode wave and the output wave are described above zh_inflate_kernel.
// Algorithmic traffic: compressed bytes read once, output written once.
#include "zh_common.h"
#include "zh_kprof.h"
#include "zh_tables.h"
#include "zh_inflate_tables.h"


// ---------------------------------------------------------------------------
// Container unwrap: one thread per stream.
// ---------------------------------------------------------------------------
__global__ void zh_unwrap_kernel(const uint8_t* __restrict__ d_src, ZhInflateArgs a) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nbufs) return;
  const ZhBufDesc b = a.bufs[i];
  const uint8_t* src = d_src + b.src_off;
  const uint64_t len = a.src_len_dev ? a.src_len_dev[i] : b.src_len;
  int fmt = a.data_format;
  int st = ZH_OK;
  uint32_t pos = 0, sum = 0, isize = 0;

  if (fmt == ZH_DF_DETECT) {  // zippy.nim:108-125
    if (len > 18 && src[0] == 31 && src[1] == 139 && src[2] == 8 && (src[3] & 0xe0) == 0)
      fmt = ZH_DF_GZIP;
    else if (len > 6 && (src[0] & 0x0f) == 8 && (src[0] >> 4) <= 7 &&
             (((uint32_t)src[0] * 256u) + src[1]) % 31u == 0)
      fmt = ZH_DF_ZLIB;
    else
      st = ZH_ERR_DETECT;
  }
  if (st == ZH_OK && fmt == ZH_DF_GZIP) {  // gzip.nim:9-66
    if (len < 18) {
      st = ZH_ERR_INVALID_BUFFER;
    } else {
      const uint8_t flg = src[3];
      if (src[0] != 31 || src[1] != 139) st = ZH_ERR_GZIP_ID;
      else if (src[2] != 8) st = ZH_ERR_UNSUPPORTED_METHOD;
      else if (flg & 0xe0) st = ZH_ERR_RESERVED_FLAGS;
      else if (flg & 4) st = ZH_ERR_UNSUPPORTED_FLAGS;  // FEXTRA
      uint64_t p = 10;
      for (int field = 0; field < 2 && st == ZH_OK; field++) {  // FNAME, FCOMMENT
        if (flg & (field == 0 ? 8 : 16)) {
          while (p < len && src[p] != 0) p++;
          if (p >= len) st = ZH_ERR_INVALID_BUFFER;
          p++;
        }
      }
      if (st == ZH_OK && (flg & 2)) {  // FHCRC: skipped, not verified (gzip.nim:55-59)
        if (p + 2 >= len) st = ZH_ERR_INVALID_BUFFER;
        p += 2;
      }
      if (st == ZH_OK && p + 8 >= len) st = ZH_ERR_INVALID_BUFFER;
      if (st == ZH_OK) {
        pos = (uint32_t)p;
        const uint8_t* t = src + len - 8;
        sum = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
      }
    }
  } else if (st == ZH_OK && fmt == ZH_DF_ZLIB) {  // zippy.nim:130-150
    if (len < 6) {
      st = ZH_ERR_INVALID_BUFFER;
    } else {
      const uint8_t cmf = src[0], flg = src[1];
      if ((cmf & 0x0f) != 8) st = ZH_ERR_UNSUPPORTED_METHOD;
      else if ((cmf >> 4) > 7) st = ZH_ERR_COMPRESSION_INFO;
      else if ((((uint32_t)cmf * 256u) + flg) % 31u != 0) st = ZH_ERR_INVALID_HEADER;
      else if (flg & 0x20) st = ZH_ERR_PRESET_DICT;
      pos = 2;
      const uint8_t* t = src + len - 4;
      sum = ((uint32_t)t[0] << 24) | (t[1] << 16) | (t[2] << 8) | t[3];
    }
  } else if (st == ZH_OK && fmt != ZH_DF_DEFLATE) {
    st = ZH_ERR_INVALID_FORMAT;
  }
  a.body_pos[i] = pos;
  a.fmt[i] = (uint32_t)fmt;
  a.expect_sum[i] = sum;
  a.expect_isize[i] = isize;
  a.status[i] = st;
  a.out_len[i] = 0;
}


// Two waves per stream, one producer and one consumer, a barrier per round:
//
// DECODE wave: owns the bit stream.  A round covers up to 128 bit positions: lane l decodes
//   the whole tokens -- literal, or length + extra + distance + extra -- that WOULD start l
//   and 64 + l bits behind the current position, each from its own 64-bit view of the stream
//   (two LDS lookups, inflate.nim:93-100 / 199-222 for all offsets at once).  A wave-uniform
//   walk then follows the real chain of token starts through those lanes (seven instructions
//   per symbol), DPP prefix sums of the chain's output lengths place every token, and the
//   round goes into one of two descriptor buffers in LDS.  Anything the LUTs cannot finish in
//   the lane (codes past the tables' reach, end of block, invalid symbols) stops the chain and is decoded alone
//   with the canonical slow path (inflate.nim:67-91) as the round's "tail".  Block headers,
//   table construction and stored blocks happen here too.
// OUTPUT wave: owns the output.  Rounds of <= 64 bytes (the usual case) get one lane per
//   OUTPUT byte: token starts are scattered by output offset and a running maximum tells
//   every byte its token; literals carry their value, match bytes read earlier output back
//   through L2 (the LZ window is the output itself), bytes copied from the round's own output
//   chase their source down by pointer doubling -- inflate.nim:227-250's byte-sequential copy
//   semantics without a loop over the tokens.  Longer rounds run their copies in order.
// While the output wave works on round k the decode wave is already on round k + 1.
namespace {

constexpr uint32_t kTailNone = 0, kTailLiteral = 1, kTailMatch = 2, kTailStored = 3, kTailEnd = 4;

struct RoundDesc {
  // per lane and window: output length (9) | literal << 9 | in chain << 10 | value << 16
  // (value: the literal byte, or the match distance)
  uint32_t rec[2][64];
  uint32_t use_b;    // window B holds tokens too
  uint32_t tail;     // what follows the chain (kTail*)
  uint32_t tail_a;   // literal byte | match length | stored length | final status
  uint32_t tail_b;   // match distance
  uint64_t tail_off; // stored run: offset of its first byte in the compressed stream
};

}  // namespace

__global__ __launch_bounds__(128, 8) void zh_inflate_kernel(const uint8_t* __restrict__ d_src,
                                                         uint8_t* __restrict__ d_dst,
                                                         ZhInflateArgs a) {
  __shared__ uint32_t s_lit[(1u << kLitBits) + kLitSub];
  __shared__ uint32_t s_dst[1u << kDistBits];  // also hosts the 7-bit code-length table
  __shared__ uint32_t s_in[kInWords];          // staging ring of the compressed stream
  __shared__ uint8_t s_map[64];                // output byte -> token lane of the current round
  __shared__ HuffTab s_tab_lit, s_tab_dist, s_tab_cl;
  __shared__ uint16_t s_val_lit[288], s_val_dist[32], s_val_cl[20];
  __shared__ uint8_t s_lens[320 + 16];
  __shared__ uint32_t s_cnt[16];
  __shared__ RoundDesc s_desc[2];
  __shared__ int32_t s_ostatus;  // raised by the output wave, polled by the decode wave

  const unsigned lane = zh_lane();
  const bool decoder = threadIdx.x < 64;
  const uint32_t sid = blockIdx.x;
  if (a.status[sid] != ZH_OK) return;  // unwrap already failed this stream (both waves leave)
  if (a.skip && a.skip[sid]) return;  // decoded (or sized) segment-wise, zh_inflate_seg.hip

  const ZhBufDesc bd = a.bufs[sid];
  const uint8_t* src = d_src + bd.src_off;
  const uint64_t src_len = a.src_len_dev ? a.src_len_dev[sid] : bd.src_len;
  uint8_t* dst = d_dst + bd.dst_off;
  const uint64_t cap = bd.dst_cap;
  const int count_only = a.count_only;
  const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
  const uint32_t* asrc = reinterpret_cast<const uint32_t*>(src - mis);
  if (threadIdx.x == 0) s_ostatus = ZH_OK;
  __syncthreads();

  if (!decoder) {
    // =========================== OUTPUT wave ===========================
    KPROF_DECL(8);  // output wave: 0 waiting for a round, 1 working; 2 rounds; 3 waves; 4 long rounds; 5 far rounds; 6 doubling turns; 7 tail matches
    uint64_t op = 0;  // bytes produced
    int st = ZH_OK;
    // match sources are read back past this CU's L1 (which may hold a stale copy of a line the
    // wave has since extended), after the wave's earlier stores have completed
    auto own_output_visible = [&]() {
      zh_wave_sync();  // (lanes read what other lanes stored: a rendezvous for the emulator)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    };
    auto ld_out = [&](uint64_t at) -> uint32_t {
      return __hip_atomic_load(dst + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // inflate.nim:224-250: one LZ copy of `length` bytes from `dist` back, at op (wave-uniform)
    auto lz_copy = [&](uint32_t length, uint32_t dist) {
      if (dist > op) {  // inflate.nim:224-225
        st = ZH_ERR_INVALID_BUFFER;
        return;
      }
      if (!count_only) {
        if (op + length > cap) {
          st = ZH_ERR_DST_TOO_SMALL;
          return;
        }
        // an overlapping copy (dist < length) repeats the dist-byte pattern, so every lane
        // reads its source from the region already written
        own_output_visible();
        const uint64_t sb = op - dist;
        if (dist >= length) {
          for (uint32_t i = lane; i < length; i += 64) dst[op + i] = (uint8_t)ld_out(sb + i);
        } else if (dist == 1) {
          const uint8_t v = (uint8_t)ld_out(sb);
          for (uint32_t i = lane; i < length; i += 64) dst[op + i] = v;
        } else {
          for (uint32_t i = lane; i < length; i += 64) dst[op + i] = (uint8_t)ld_out(sb + i % dist);
        }
      }
      op += length;
    };
    for (uint32_t k = 0;; k++) {
      KPROF_MARK(1);
      __syncthreads();  // round k is in s_desc[k & 1]
      KPROF_MARK(0);
      KPROF_COUNT(2, 1);
      // wave-uniform by construction; saying so keeps them in scalar registers and the control
      // flow below on scalar branches
      op = zh_bcast64(op);
      st = (int)zh_bcast((uint32_t)st);
      const RoundDesc& d = s_desc[k & 1u];
      const uint32_t use_b = d.use_b, tail = d.tail, tail_a = d.tail_a, tail_b = d.tail_b;
      const uint64_t tail_off = d.tail_off;
      const uint32_t recA = d.rec[0][lane];
      const uint32_t recB = use_b ? d.rec[1][lane] : 0u;
      if (tail == kTailEnd) {
        if (st == ZH_OK) st = (int)tail_a;
        break;
      }
      if (st != ZH_OK) continue;  // failed: keep taking rounds until the decode wave stops
      // where every token's output goes: prefix sums of the chain's output lengths
      const bool in_chain = (recA >> 10) & 1u, is_lit = (recA >> 9) & 1u;
      const bool in_chainB = (recB >> 10) & 1u, is_litB = (recB >> 9) & 1u;
      const uint32_t valA = recA >> 16, valB = recB >> 16;
      const uint32_t lenA = in_chain ? recA & 0x1ffu : 0u, lenB = in_chainB ? recB & 0x1ffu : 0u;
      // one scan for both windows: A's lengths in the low half, B's in the high half (a window's
      // sum is at most 64 * 258)
      const uint32_t incl = zh_wave_scan(lenA | (lenB << 16));
      const uint32_t sums = (uint32_t)__builtin_amdgcn_readlane(incl, 63);
      const uint32_t totalA = sums & 0xffffu, totalB = sums >> 16;
      const uint32_t opre = (incl & 0xffffu) - lenA;
      const uint32_t opreB = totalA + (incl >> 16) - lenB;
      const uint32_t total = totalA + totalB;
      if (total - 1u < 64u) {
        const bool is_match = in_chain && !is_lit, is_matchB = in_chainB && !is_litB;
        // inflate.nim:224-225 `distance > op`: a distance is at most 32768 (30 codes), so the lanes
        // only need asking while the stream is that young
        if (op < 32768u && __ballot((is_match && (uint64_t)valA > op + opre) ||
                                    (is_matchB && (uint64_t)valB > op + opreB))) {
          st = ZH_ERR_INVALID_BUFFER;
        } else if (!count_only && op + total > cap) {
          st = ZH_ERR_DST_TOO_SMALL;
        } else if (!count_only) {
          zh_wave_sync();
          s_map[lane] = 0;
          zh_wave_sync();
          if (in_chain) s_map[opre] = (uint8_t)(lane + 1u);
          if (in_chainB) s_map[opreB] = (uint8_t)(lane + 65u);
          zh_wave_sync();
          const uint32_t tk = zh_wave_scan_max(s_map[lane]);  // token (window, lane) + 1 of output byte `lane`
          const uint32_t j = (tk - 1u) & 63u;
          // both windows' tokens of a lane travel in one cross-lane fetch: 16 bits each, a distance
          // (<= 0x8000) as it is, a literal as 0x9000 | byte
          const uint32_t f1 = (is_lit ? 0x9000u | valA : valA) | ((is_litB ? 0x9000u | valB : valB) << 16);
          const uint32_t g2 = (uint32_t)__shfl((int)f1, (int)j, 64);
          const uint32_t gd = tk > 64u ? g2 >> 16 : g2 & 0xffffu;
          const bool live = lane < total;
          uint32_t val = gd & 0xffu;
          uint32_t par = lane;  // source byte inside this round (itself: a root)
          bool far = false;
          uint32_t back = 0;
          if (live && gd < 0x9000u) {
            if (gd <= lane) {
              par = lane - gd;
            } else {
              back = gd - lane;  // bytes before this round's first
              far = true;
            }
          }
          if (__ballot(far)) {
            KPROF_COUNT(5, 1);
            own_output_visible();
            if (far) val = ld_out(op - back);
          }
          if (__ballot(par != lane)) {
            for (;;) {
              KPROF_COUNT(6, 1);
              const uint32_t pp = (uint32_t)__shfl((int)par, (int)par, 64);
              const bool changed = pp != par;
              par = pp;
              if (!__ballot(changed)) break;
            }
            val = (uint32_t)__shfl((int)val, (int)par, 64);
          }
          if (live) dst[op + lane] = (uint8_t)val;
        }
        // a failed stream reports the bytes it really wrote (out_len <= cap): `op` stops at the
        // last round that fitted
        if (st == ZH_OK) op += total;
      } else if (total) {
        // long rounds: per window, literal runs by their lanes and copies in order
        KPROF_COUNT(4, 1);
        const uint64_t op0 = op;
        auto long_window = [&](bool chain_w, bool lit_w, uint32_t rec_w, uint32_t opre_w) {
          const uint64_t litmask = __ballot(chain_w && lit_w);
          uint64_t mm = __ballot(chain_w && !lit_w);
          uint32_t done_lanes = 0;  // chain lanes below this bit offset are finished
          while (mm && st == ZH_OK) {
            const uint32_t g = (uint32_t)__ffsll((long long)mm) - 1u;
            mm &= mm - 1;
            const uint64_t grp = litmask & ((1ull << g) - 1ull) & (~0ull << done_lanes);
            if (grp) {
              const uint32_t nl = (uint32_t)__popcll(grp);
              if (!count_only) {
                if (op + nl > cap) { st = ZH_ERR_DST_TOO_SMALL; break; }
                if ((grp >> lane) & 1ull) dst[op0 + opre_w] = (uint8_t)(rec_w >> 16);
              }
              op += nl;
            }
            const uint32_t r = __builtin_amdgcn_readlane(rec_w, g);
            lz_copy(r & 0x1ffu, r >> 16);
            done_lanes = g + 1u;
          }
          if (st == ZH_OK) {
            const uint64_t grp = done_lanes < 64u ? litmask & (~0ull << done_lanes) : 0ull;
            if (grp) {
              const uint32_t nl = (uint32_t)__popcll(grp);
              if (!count_only) {
                if (op + nl > cap) st = ZH_ERR_DST_TOO_SMALL;
                else if ((grp >> lane) & 1ull) dst[op0 + opre_w] = (uint8_t)(rec_w >> 16);
              }
              if (st == ZH_OK) op += nl;
            }
          }
        };
        long_window(in_chain, is_lit, recA, opre);
        if (use_b && st == ZH_OK) long_window(in_chainB, is_litB, recB, opreB);
      }
      if (st == ZH_OK) {
        if (tail == kTailLiteral) {
          if (!count_only) {
            if (op + 1 > cap) st = ZH_ERR_DST_TOO_SMALL;
            else if (lane == 0) dst[op] = (uint8_t)tail_a;
          }
          if (st == ZH_OK) op += 1;
        } else if (tail == kTailMatch) {
          KPROF_COUNT(7, 1);
          lz_copy(tail_a, tail_b);
        } else if (tail == kTailStored) {  // inflate.nim:252-266: tail_a raw bytes
          if (!count_only) {
            if (op + tail_a > cap) {
              st = ZH_ERR_DST_TOO_SMALL;
            } else {
              const uint8_t* raw = reinterpret_cast<const uint8_t*>(asrc) + tail_off;
              for (uint32_t i = lane; i < tail_a; i += 64) dst[op + i] = raw[i];
            }
          }
          if (st == ZH_OK) op += tail_a;
        }
      }
      if (st != ZH_OK && lane == 0) s_ostatus = st;
    }
    if (st == ZH_OK && op > cap && !count_only) st = ZH_ERR_DST_TOO_SMALL;
    if (lane == 0) {
      a.out_len[sid] = op;
      a.status[sid] = st;
    }
    KPROF_COUNT(3, 1);
    KPROF_FLUSH(16, 8);
    return;
  }

  // =========================== DECODE wave ===========================
  // (it is the critical one of the pair at equal priority -- the output wave waits a fifth of
  // the time -- so its decode + chain walk take first pick of the SIMD's issue slots, below)
  // input: the stream is addressed in bits from the 4-byte aligned base below `src`; 64 dwords
  // at a time go through a 512-byte LDS ring (bitstreams.nim:22-49's refill)
  const uint64_t end = mis + src_len;  // first byte offset (from asrc) past the stream
  auto load_dword = [&](uint64_t off) -> uint32_t {  // bytes past the end read as zero
    if (off >= end) return 0u;
    uint32_t v = asrc[off >> 2];
    if (off + 4 > end) v &= (1u << (8 * (uint32_t)(end - off))) - 1u;
    return v;
  };
  uint64_t bp = 0;      // stream position in bits (from asrc)
  uint64_t in_hi = 0;   // dwords below in_hi are staged (the ring holds the last kInWords of them)
  uint32_t wnxt = 0;    // lane l: dword in_hi + l, loaded ahead of its use
  auto stage = [&]() {
    zh_wave_sync();
    s_in[(uint32_t)(in_hi + lane) & (kInWords - 1u)] = wnxt;
    in_hi += 64;
    wnxt = load_dword((in_hi + lane) * 4);
    zh_wave_sync();
  };
  auto seek = [&](uint64_t bitpos) {  // restart the staging at bitpos
    bp = bitpos;
    in_hi = bitpos >> 5;
    wnxt = load_dword((in_hi + lane) * 4);
    stage();
  };
  auto ensure = [&]() {  // the round below reads up to five dwords from bp >> 5
    if (in_hi < (bp >> 5) + 8u) stage();
  };
  auto fetch = [&]() -> uint64_t {  // the 64 stream bits at bp (wave-uniform)
    ensure();
    const uint32_t wi = (uint32_t)(bp >> 5), sh = (uint32_t)bp & 31u;
    const uint32_t d0 = zh_bcast(s_in[wi & (kInWords - 1u)]), d1 = zh_bcast(s_in[(wi + 1u) & (kInWords - 1u)]),
                   d2 = zh_bcast(s_in[(wi + 2u) & (kInWords - 1u)]);
    return (uint64_t)zh_alignbit(d1, d0, sh) | ((uint64_t)zh_alignbit(d2, d1, sh) << 32);
  };
  // header bit reader: hb caches the bits at bp
  uint64_t hb = 0;
  uint32_t hc = 0;
  auto need = [&]() {
    if (hc < 32u) {
      hb = fetch();
      hc = 64;
    }
  };
  auto take = [&](uint32_t nbits) -> uint32_t {  // nbits <= 16, after need()
    const uint32_t v = (uint32_t)hb & ((1u << nbits) - 1u);
    hb >>= nbits;
    hc -= nbits;
    bp += nbits;
    return v;
  };
  auto past_end = [&]() -> bool { return bp > end * 8; };  // the role of `bitsBuffered < 0`
  // inflate.nim:67-91 decodeSymbolSlow for codes longer than the LUT on the bits in `bits`;
  // returns the symbol (0xffff = unassigned code) and its length in *nb
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

  KPROF_DECL(8);  // decode wave: 0 waiting for the output wave, 1 other work; 2 rounds; 3 waves; 4 vector decode; 5 chain walks; 6 staging
  uint32_t rk = 0;  // rounds handed over so far
  int st = ZH_OK;
  // hand a round without chain tokens to the output wave
  auto send_tail = [&](uint32_t tail, uint32_t ta, uint32_t tb, uint64_t toff) {
    RoundDesc& d = s_desc[rk & 1u];
    d.rec[0][lane] = 0;
    if (lane == 0) {
      d.use_b = 0;
      d.tail = tail;
      d.tail_a = ta;
      d.tail_b = tb;
      d.tail_off = toff;
    }
    KPROF_MARK(1);
    __syncthreads();
    KPROF_MARK(0);
    rk++;
    if (st == ZH_OK && s_ostatus != ZH_OK) st = s_ostatus;
  };

  // a whole stream starts at its deflate body; a block of an indexed stream at its own first bit
  seek(a.start_bit ? (uint64_t)mis * 8 + a.start_bit[sid] : ((uint64_t)mis + a.body_pos[sid]) * 8);
  const bool single_block = a.single_block != 0;
  bool final_block = false;
  while (!final_block && st == ZH_OK) {  // inflate.nim:273-289
    hc = 0;
    need();
    const uint32_t bfinal = take(1), btype = take(2);
    if (bfinal || single_block) final_block = true;

    if (btype == 0) {  // inflate.nim:252-266 inflateNoCompression
      bp = (bp + 7u) & ~(uint64_t)7;
      hc = 0;
      need();
      const uint32_t len = take(16), nlen = take(16);
      if (len + nlen != 65535u) { st = ZH_ERR_INVALID_BUFFER; break; }
      const uint64_t byte_pos = bp >> 3;  // from asrc
      if (byte_pos + len > end) { st = ZH_ERR_END_OF_BUFFER; break; }
      send_tail(kTailStored, len, 0, byte_pos);
      seek((byte_pos + len) * 8);
      continue;
    }
    if (btype == 3) { st = ZH_ERR_BLOCK_HEADER; break; }

    uint32_t hlit = 288, hdist = 30;
    if (btype == 1) {  // fixed codes, inflate.nim:111-113 (rebuilt per block like the reference)
      zh_wave_sync();
      for (uint32_t s = lane; s < 288; s += 64) s_lens[s] = (uint8_t)(s <= 143 ? 8 : s <= 255 ? 9 : s <= 279 ? 7 : 8);
      if (lane < 30) s_lens[288 + lane] = 5;
    } else {  // dynamic header, inflate.nim:115-171
      hlit = take(5) + 257;
      hdist = take(5) + 1;
      const uint32_t hclen = take(4) + 4;
      if (hlit > 286 || hdist > 30) { st = ZH_ERR_INVALID_BUFFER; break; }
      zh_wave_sync();
      if (lane < 20) s_lens[lane] = 0;
      zh_wave_sync();
      for (uint32_t i = 0; i < hclen; i++) {
        need();
        const uint32_t v = take(3);
        if (lane == 0) s_lens[c_clcl_order[i]] = (uint8_t)v;
      }
      st = build_table(s_lens, 19, s_dst, 7, 2, &s_tab_cl, s_val_cl, s_cnt);
      if (st != ZH_OK) break;
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
        if (past_end()) { st = ZH_ERR_END_OF_BUFFER; break; }
        if (sym <= 15) {
          if (lane == 0) s_lens[i] = (uint8_t)sym;
          prev = sym;
          i++;
        } else if (sym == 16) {
          if (i == 0) { st = ZH_ERR_INVALID_BUFFER; break; }
          const uint32_t rep = take(2) + 3;
          if (i + rep > 320) { st = ZH_ERR_INVALID_BUFFER; break; }
          if (lane < rep) s_lens[i + lane] = (uint8_t)prev;
          i += rep;
        } else if (sym == 17 || sym == 18) {
          const uint32_t rep = sym == 17 ? take(3) + 3 : take(7) + 11;
          for (uint32_t j = lane; j < rep && i + j < 320 + 16; j += 64) s_lens[i + j] = 0;
          prev = 0;
          i += rep;
        } else {
          st = ZH_ERR_INVALID_SYMBOL;
          break;
        }
        if (i > total) { st = ZH_ERR_INVALID_BUFFER; break; }
      }
      if (st != ZH_OK) break;
    }
    {
      const uint32_t dist_at = btype == 1 ? 288u : hlit;
      st = build_table(s_lens, hlit, s_lit, kLitBits, 0, &s_tab_lit, s_val_lit, s_cnt, kLitSub);
      if (st != ZH_OK) break;
      st = build_table(s_lens + dist_at, hdist, s_dst, kDistBits, 1, &s_tab_dist, s_val_dist, s_cnt);
      if (st != ZH_OK) break;
    }

    for (;;) {  // inflate.nim:173-250, one round = up to 128 bit positions
      KPROF_MARK(1);
#ifndef ZH_EMU
      __builtin_amdgcn_s_setprio(1);
#endif
      ensure();
      KPROF_MARK(6);
      // ---- every lane decodes the tokens that would start at bits bp + lane (window A)
      // and bp + 64 + lane (window B): two independent dependency chains per lane; B is
      // used when A's chain runs into it cleanly and both together make at most 64 bytes ----
      const uint32_t wi = (uint32_t)((bp + lane) >> 5), sh = ((uint32_t)bp + lane) & 31u;
      // (the output wave's status travels with these reads: a round's delay in noticing a failure
      // costs nothing, the exposed LDS round trip of a read behind the barrier did)
      const int32_t ostatus = s_ostatus;
      const uint32_t d0 = s_in[wi & (kInWords - 1u)], d1 = s_in[(wi + 1u) & (kInWords - 1u)],
                     d2 = s_in[(wi + 2u) & (kInWords - 1u)], d3 = s_in[(wi + 3u) & (kInWords - 1u)],
                     d4 = s_in[(wi + 4u) & (kInWords - 1u)];
      struct Tok {
        uint32_t v_lo, v_hi, e, tbits, outlen, val;  // val: literal byte or match distance
        bool is_lit;
      };
      // `flag` in the lanes whose bit is set in the wave-uniform mask (one select on the mask
      // instead of a 64-bit shift per lane)
      auto lane_flag = [&](uint64_t mask, uint32_t flag) -> uint32_t {
#ifdef ZH_EMU
        return (mask >> lane) & 1ull ? flag : 0u;
#else
        uint32_t r;
        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(flag), "s"(mask));
        return r;
#endif
      };
      auto decode_tok = [&](uint32_t v_lo, uint32_t v_hi) -> Tok {
        Tok t;
        t.v_lo = v_lo;
        t.v_hi = v_hi;
        uint32_t e = s_lit[v_lo & ((1u << kLitBits) - 1u)];
        if (e & 0x400u) e = s_lit[(e >> 16) + ((v_lo >> kLitBits) & ((1u << (e & 15u)) - 1u))];  // a longer code
        const uint32_t L = e & 15u, eb = (e >> 4) & 15u;
        t.e = e;
        t.is_lit = (e & 0x8000u) != 0;
        const uint32_t lenval = (e >> 16) + ((v_lo >> L) & ((1u << eb) - 1u));
        // (32-bit funnel shifts: L + eb <= 20 and, with a distance code from the 8-bit table, o3 <= 28)
        const uint32_t o2 = L + eb;
        const uint32_t de = s_dst[zh_alignbit(v_hi, v_lo, o2) & ((1u << kDistBits) - 1u)];
        const uint32_t o3 = o2 + (de & 15u), deb = (de >> 4) & 15u;
        t.val = t.is_lit ? (e >> 16) & 0xffu : (de >> 16) + (zh_alignbit(v_hi, v_lo, o3) & ((1u << deb) - 1u));
        if (t.is_lit) t.tbits = L;  // bits of the whole token; 0x8000: not decodable here
        else if (e != 0 && ((e >> 8) & 3u) == kKindBase && de != 0 && ((de >> 8) & 3u) == kKindBase) t.tbits = o3 + deb;
        else t.tbits = 0x8000u;
        t.outlen = t.is_lit ? 1u : lenval;
        return t;
      };
      const Tok A = decode_tok(zh_alignbit(d1, d0, sh), zh_alignbit(d2, d1, sh));
      const Tok B = decode_tok(zh_alignbit(d3, d2, sh), zh_alignbit(d4, d3, sh));

      // ---- the chain of real token starts ----
      uint64_t chain = 0, chainB = 0;
      uint32_t pos = 0;
#ifdef ZH_KPROF
      asm volatile("" ::"v"(A.tbits), "v"(B.tbits));  // (timers: the decode is done here)
#endif
      KPROF_MARK(4);
#ifdef ZH_EMU
      while (pos < 64u) {
        const uint32_t tv = __builtin_amdgcn_readlane(A.tbits, pos);
        if (tv & 0x8000u) break;
        chain |= 1ull << pos;
        pos += tv;
      }
      const bool useB = pos >= 64u;
      if (useB) {
        while (pos < 128u) {
          const uint32_t tv = __builtin_amdgcn_readlane(B.tbits, pos - 64u);
          if (tv & 0x8000u) break;
          chainB |= 1ull << (pos - 64u);
          pos += tv;
        }
      }
#else
      // the same two loops, five instructions and one taken branch per symbol: a token that
      // cannot be decoded here has length 0x8000, which ends the loop like the end of the window
      // does; its lane's bit is taken back afterwards (lane select and s_bitset use pos[5:0])
      {
        uint32_t tv;
        asm volatile(
            "1:\n\t"
            "v_readlane_b32 %[tv], %[tb], %[pos]\n\t"
            "s_bitset1_b64 %[ch], %[pos]\n\t"
            "s_add_u32 %[pos], %[pos], %[tv]\n\t"
            "s_cmp_lt_u32 %[pos], 64\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_bitcmp0_b32 %[pos], 15\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_and_b32 %[pos], %[pos], 0x7fff\n\t"
            "s_bitset0_b64 %[ch], %[pos]\n\t"
            "2:"
            : [tv] "=&s"(tv), [pos] "+s"(pos), [ch] "+s"(chain)
            : [tb] "v"(A.tbits)
            : "scc");
      }
      const bool useB = pos >= 64u;  // (a chain that stopped in window A left pos below 64)
      if (useB) {
        uint32_t tv;
        asm volatile(
            "1:\n\t"
            "v_readlane_b32 %[tv], %[tb], %[pos]\n\t"
            "s_bitset1_b64 %[ch], %[pos]\n\t"
            "s_add_u32 %[pos], %[pos], %[tv]\n\t"
            "s_cmp_lt_u32 %[pos], 0x80\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_bitcmp0_b32 %[pos], 15\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_and_b32 %[pos], %[pos], 0x7fff\n\t"
            "s_bitset0_b64 %[ch], %[pos]\n\t"
            "2:"
            : [tv] "=&s"(tv), [pos] "+s"(pos), [ch] "+s"(chainB)
            : [tb] "v"(B.tbits)
            : "scc");
      }
#endif
      KPROF_MARK(5);
#ifndef ZH_EMU
      __builtin_amdgcn_s_setprio(0);
#endif
      // ---- hand the round over (the output wave works out the offsets) ----
      RoundDesc& d = s_desc[rk & 1u];
      d.rec[0][lane] = A.outlen | (A.is_lit ? 1u << 9 : 0u) | lane_flag(chain, 1u << 10) | (A.val << 16);
      if (useB)
        d.rec[1][lane] = B.outlen | (B.is_lit ? 1u << 9 : 0u) | lane_flag(chainB, 1u << 10) | (B.val << 16);
      bp += pos;
      bool block_done = false;
      uint32_t tail = kTailNone, tail_a = 0, tail_b = 0;
      if (pos < 64u || (useB && pos < 128u)) {
        // ---- the token at bp stopped the chain: decode it alone (inflate.nim:67-102) ----
        // (the builtin returns int: widen through uint32_t, or the low word sign-extends)
        uint32_t sv_lo, sv_hi, se;
        if (pos < 64u) {
          sv_lo = __builtin_amdgcn_readlane(A.v_lo, pos);
          sv_hi = __builtin_amdgcn_readlane(A.v_hi, pos);
          se = __builtin_amdgcn_readlane(A.e, pos);
        } else {
          sv_lo = __builtin_amdgcn_readlane(B.v_lo, pos - 64u);
          sv_hi = __builtin_amdgcn_readlane(B.v_hi, pos - 64u);
          se = __builtin_amdgcn_readlane(B.e, pos - 64u);
        }
        uint64_t sv = (uint64_t)sv_lo | ((uint64_t)sv_hi << 32);
        uint32_t used;
        if (se == 0) {  // longer than the LUT, or unassigned
          uint32_t nb;
          const uint32_t sym = decode_slow((uint32_t)sv, kLitBits, &s_tab_lit, s_val_lit, &nb);
          se = litlen_entry(sym, 0);
          used = nb;
        } else {
          used = se & 15u;
        }
        sv >>= used;
        const uint32_t kind = (se >> 8) & 3u;
        if (se & 0x8000u) {  // a literal with a long code
          tail = kTailLiteral;
          tail_a = (se >> 16) & 0xffu;
        } else if (kind == kKindEob) {
          block_done = true;
        } else if (kind == kKindBad) {  // inflate.nim:202-204
          st = ZH_ERR_INVALID_BUFFER;
        } else {
          const uint32_t seb = (se >> 4) & 15u;
          const uint32_t length = (se >> 16) + ((uint32_t)sv & ((1u << seb) - 1u));
          sv >>= seb;
          used += seb;
          uint32_t sde = zh_bcast(s_dst[(uint32_t)sv & ((1u << kDistBits) - 1u)]);
          uint32_t dnb;
          if (sde == 0) {
            const uint32_t dsym = decode_slow((uint32_t)sv, kDistBits, &s_tab_dist, s_val_dist, &dnb);
            sde = dist_entry(dsym, 0);
          } else {
            dnb = sde & 15u;
          }
          sv >>= dnb;
          used += dnb;
          if (((sde >> 8) & 3u) == kKindBad) {  // inflate.nim:211-213
            st = ZH_ERR_INVALID_BUFFER;
          } else {
            const uint32_t sdeb = (sde >> 4) & 15u;
            tail = kTailMatch;
            tail_a = length;
            tail_b = (sde >> 16) + ((uint32_t)sv & ((1u << sdeb) - 1u));
            used += sdeb;
          }
        }
        bp += used;
      }
      // tokens decoded from beyond the end of the input are caught here at the latest
      if (st == ZH_OK && past_end()) st = ZH_ERR_END_OF_BUFFER;
      if (lane == 0) {
        d.use_b = useB ? 1u : 0u;
        d.tail = st == ZH_OK ? tail : kTailNone;
        d.tail_a = tail_a;
        d.tail_b = tail_b;
      }
      KPROF_MARK(1);
      __syncthreads();
      KPROF_MARK(0);
      KPROF_COUNT(2, 1);
      rk++;
      if (st == ZH_OK && ostatus != ZH_OK) st = ostatus;
      if (st != ZH_OK || block_done) break;
    }
  }
  send_tail(kTailEnd, (uint32_t)st, 0, 0);
  KPROF_COUNT(3, 1);
  KPROF_FLUSH(24, 8);
}

// Final check of each stream against its trailer (gzip.nim:80-88, zippy.nim:152-162).
__global__ void zh_verify_kernel(ZhInflateArgs a, const uint32_t* __restrict__ buf_crc,
                                 const uint32_t* __restrict__ buf_adler) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nbufs) return;
  if (a.status[i] != ZH_OK) return;
  const uint32_t fmt = a.fmt[i];
  if (fmt == ZH_DF_GZIP) {
    if (a.expect_sum[i] != buf_crc[i]) a.status[i] = ZH_ERR_CHECKSUM;
    else if (a.expect_isize[i] != (uint32_t)(a.out_len[i] & 0xffffffffu)) a.status[i] = ZH_ERR_SIZE;
  } else if (fmt == ZH_DF_ZLIB) {
    if (a.expect_sum[i] != buf_adler[i]) a.status[i] = ZH_ERR_CHECKSUM;
  }
}

// Block-parallel decode: fold the per-block results into the stream's.  Every block must have
// produced exactly the bytes its index entry promised.
__global__ __launch_bounds__(64) void zh_segments_reduce_kernel(ZhInflateArgs seg, ZhInflateArgs stream) {
  __shared__ uint64_t s_total[64];
  __shared__ uint32_t s_bad[64];
  const unsigned lane = zh_lane();
  uint64_t total = 0;
  uint32_t bad = 0xffffffffu;
  for (uint32_t i = lane; i < seg.nbufs; i += 64) {
    const uint64_t len = seg.out_len[i];
    total += len;
    if ((seg.status[i] != ZH_OK || len != seg.bufs[i].dst_cap) && i < bad) bad = i;
  }
  s_total[lane] = total;
  s_bad[lane] = bad;
  zh_wave_sync();
  if (lane == 0 && stream.status[0] == ZH_OK) {
    for (uint32_t l = 1; l < 64; l++) {
      total += s_total[l];
      if (s_bad[l] < bad) bad = s_bad[l];
    }
    stream.out_len[0] = total;
    if (bad != 0xffffffffu) {
      const int32_t st = seg.status[bad];
      stream.status[0] = (st == ZH_OK || st == ZH_ERR_DST_TOO_SMALL) ? (int32_t)ZH_ERR_INVALID_BUFFER : st;
    }
  }
}

extern "C" void zh_launch_segments_reduce(hipStream_t stream, ZhInflateArgs seg, ZhInflateArgs whole) {
  hipLaunchKernelGGL(zh_segments_reduce_kernel, dim3(1), dim3(64), 0, stream, seg, whole);
}

extern "C" void zh_launch_unwrap(hipStream_t stream, const uint8_t* d_src, ZhInflateArgs a) {
  if (!a.nbufs) return;
  hipLaunchKernelGGL(zh_unwrap_kernel, dim3((a.nbufs + 63) / 64), dim3(64), 0, stream, d_src, a);
}
extern "C" void zh_launch_inflate(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst,
                                  ZhInflateArgs a) {
  if (!a.nbufs) return;
  hipLaunchKernelGGL(zh_inflate_kernel, dim3(a.nbufs), dim3(128), 0, stream, d_src, d_dst, a);
}
extern "C" void zh_launch_verify(hipStream_t stream, ZhInflateArgs a, const uint32_t* buf_crc,
                                 const uint32_t* buf_adler) {
  if (!a.nbufs) return;
  hipLaunchKernelGGL(zh_verify_kernel, dim3((a.nbufs + 63) / 64), dim3(64), 0, stream, a, buf_crc,
                     buf_adler);
}
