// Shared definitions of the MI355X DEFLATE engine (device + host side).
//
// Vocabulary (follows the reference):
//   buffer    one independent input of the batch (one compress()/uncompress() call)
//   block     <= 4 MiB of a buffer = one deflate block with one Huffman table
//             (deflate.nim:228-237, internal.nim:16)
//   fragment  <= 32 KiB of a block.  At BestSpeed fragments are LZ-independent
//             (snappy.nim:150-163), so a fragment is the unit of device
//             parallelism: one 64-lane wave parses / emits one fragment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/zippy_hip.h"

#define ZH_WAVE 64
#define ZH_FRAG_SIZE 32768u          // internal.nim:14 maxWindowSize
#define ZH_BLOCK_SIZE 4194304u       // internal.nim:16 maxBlockSize
#define ZH_STORED_MAX 65535u         // internal.nim:15 maxUncompressedBlockSize
#define ZH_MAX_MATCHES_PER_FRAG 8192 // a >=4-byte match every 4 bytes
#define ZH_HIST_STRIDE 320           // 286 litlen + 30 distance counters, padded
#define ZH_HDR_WORDS 256             // dynamic block header bit string, <= 1 KiB
#define ZH_NUM_LITLEN 286
#define ZH_NUM_DIST 30
#define ZH_CHAIN_HEAD_WORDS ((1u << 17) + 64u)  // chain levels: a block's `head` (lz77.nim:5-6) + a scratch slot a lane

enum { ZH_MODE_STORED = 0, ZH_MODE_FIXED = 1, ZH_MODE_DYNAMIC = 2 };

struct ZhFragDesc {
  uint64_t src_off;  // absolute byte offset of the fragment in d_src
  uint32_t len;      // 1..32768
  uint32_t block;    // owning block
};

struct ZhBlockDesc {
  uint64_t src_off;  // absolute byte offset of the block in d_src
  uint64_t len;      // <= 4 MiB (whole buffer at level 0)
  uint32_t buf;
  uint32_t first_frag;
  uint32_t nfrag;
  uint32_t is_final;
};

struct ZhBufDesc {
  uint64_t src_off, src_len;
  uint64_t dst_off, dst_cap;
  uint32_t first_block, nblocks;
  uint32_t first_piece, npieces;  // checksum pieces (== fragments on the compress side)
  uint32_t fname_len;             // gzip FNAME letters (zippy.nim:26-42)
  uint32_t pad;
};

// Device pointers of one compress plan (SoA scratch in HBM).
struct ZhCompressArgs {
  const ZhFragDesc* frags;
  const ZhBlockDesc* blocks;
  const ZhBufDesc* bufs;
  uint32_t nfrags, nblocks, nbufs;
  int32_t level, data_format;
  // The chain levels' kernels run on a RANGE of the plan's blocks at a time when their scratch (12 bytes a
  // position) is smaller than the batch: blocks first_block .. first_block + nblocks - 1 = fragments first_frag ..
  // first_frag + nfrags - 1; the scratch is indexed from the range's first block / fragment, everything else
  // by the plan's own numbers.  (0, 0 and the plan's counts everywhere else.)
  uint32_t first_block, first_frag;
  // per fragment, written by the matcher
  uint16_t* m_pos;   // [nfrags][8192] match start, relative to the fragment
  uint16_t* m_len;   // [nfrags][8192] 4..258 (5..258 for the chain levels)
  uint16_t* m_off;   // [nfrags][8192] 1..32767
  uint32_t* f_nmatch;
  uint32_t* f_spill;  // leading bytes covered by a match begun in the previous fragment
  uint32_t* f_nlit;
  uint32_t* f_extra_bits;  // sum of length+distance extra bits of the fragment's matches
  uint16_t* f_hist;        // [nfrags][320]
  // [nfrags][1024] levels 1 (exact parse) and -2, else null: bit p of a fragment's 4 KiB = byte p lies inside a match,
  // behind its first byte -- written by zh_l1_match_kernel in whole groups of 64 words, read by zh_emit_kernel
  uint32_t* f_cover;
  uint32_t* f_crc;         // CRC-32 (gzip) or adler s1 | s2<<16 pieces, see zh_checksum
  uint32_t* f_adler;
  // per fragment, written by the Huffman / layout kernels
  uint32_t* f_bits;        // encoded bit length under the block's codes
  uint64_t* f_bit_start;   // absolute bit position in d_dst
  // per block
  uint32_t* b_mode;
  uint32_t* b_litcode;     // [nblocks][288]  code | len << 16
  uint32_t* b_distcode;    // [nblocks][32]
  uint32_t* b_hdr;         // [nblocks][ZH_HDR_WORDS] header bit string (LSB first)
  uint32_t* b_hdr_bits;
  uint64_t* b_bits;        // total bits of the block (header + payload + EOB), compressed modes
  uint64_t* b_stored_d0;   // stored mode: absolute byte of block byte 0 in d_dst
  uint64_t* b_start;       // bit position of the block's first bit, from the start of the deflate body
  // per buffer
  uint64_t* out_len;
  int32_t* status;
};

// Checksum piece: d_data[off .. off+len), len <= 32768.  With a device-side length
// array (inflate output) len = clamp(dyn_len[buf] - rel_off, 0, len): `len` is then the
// piece's share of the slot capacity.
struct ZhPieceDesc {
  uint64_t off;      // absolute byte offset in d_data
  uint32_t len;      // static length (an upper bound when dyn_len is used)
  uint32_t buf;
  uint64_t rel_off;  // offset of the piece inside its buffer
};

struct ZhInflateArgs {
  const ZhBufDesc* bufs;         // src_off/src_len = compressed stream, dst_off/dst_cap = slot
  const uint64_t* src_len_dev;   // optional device override of src_len (compress -> uncompress)
  uint32_t nbufs;
  int32_t data_format;
  int32_t count_only;            // sizing pass for zlib/raw streams: decode without writing
  // unwrap results (per stream)
  uint32_t* body_pos;
  uint32_t* fmt;                 // resolved format
  uint32_t* expect_sum;          // CRC-32 (gzip) / Adler-32 (zlib) from the trailer
  uint32_t* expect_isize;
  uint64_t* out_len;
  int32_t* status;
  // block-parallel decode (zh_plan_uncompress_indexed): every "stream" is one deflate block
  const uint64_t* start_bit;     // bit position of the block in its compressed buffer, or null
  int32_t single_block;          // stop after one block whatever BFINAL says
  // split decode: streams whose flag is set were decoded segment-wise (ZhSegArgs) and are left alone
  const uint32_t* skip;
  // split decode: the launch covers streams first_buf .. first_buf + nbufs - 1 (the token pool holds a group of
  // streams at a time when it is smaller than the batch; 0 and the plan's count otherwise)
  uint32_t first_buf;
};

// One large stream decoded by many workgroups (zh_inflate_seg.hip).  The compressed bytes are cut
// into segments; a segment's decoder starts at the first deflate block found at or behind the
// segment's nominal first bit and stops at the block boundary where the next segment's begins.
constexpr uint64_t kSegNone = ~0ull;
constexpr uint32_t kSegFindSlots = 256;  // candidates a search batch hands on (ZhSegArgs.cand_off: zh_inflate_seg.hip, the plan's arena)
struct ZhSegArgs {
  uint32_t nsegs, nstreams;
  const uint32_t* parent;       // [nsegs] the stream a segment belongs to
  const uint32_t* first_seg;    // [nstreams + 1] a stream's segments are first_seg[i] .. first_seg[i + 1] - 1
  const uint64_t* nominal_bit;  // [nsegs] where the search for the segment's first block starts (bits from the stream's first byte)
  const uint64_t* search_bits;  // [nsegs] ... and how far it goes
  const uint64_t* tok_off;      // [nsegs] token region of the segment in the plan's token pool
  const uint64_t* tok_cap;
  const uint64_t* sym_base;     // [nstreams] first symbol of the stream's output in `sym`
  // the search for block starts: one workgroup per batch of 65536 bit positions of a segment
  uint32_t nfind;
  const uint32_t* find_seg;     // [nfind]
  const uint32_t* find_batch;   // [nfind] (bit 31: the stream's last block may start in this batch: BFINAL = 1 counts too)
  uint32_t* cand_n;             // [nfind] positions of the batch that passed the cheap tests ...
  uint32_t* cand_off;           // [nfind][kSegFindSlots] ... as offsets into the batch
  uint32_t* go;                 // [nstreams] enough segments of the stream have a start: decode it segment-wise
  uint64_t* eff_tok_off;        // [nsegs] token region of a segment that keeps its decoder (zh_seg_decide_kernel:
  uint64_t* eff_tok_cap;        //   its own, or those of its whole group of segments)
  // sub-starts (zh_inflate_tokens_kernel phase 0, merged into start_bit by zh_seg_decide_kernel)
  uint64_t* sub_start;          // [nsegs] a token boundary at or behind the segment's nominal first bit, or kSegNone
  uint64_t* sub_hdr;            // [nsegs] first bit of the header of the block it lies in
  uint32_t* is_sub;             // [nsegs] start_bit is such a boundary, not a block start
  // find / tokens results
  uint64_t* start_bit;          // [nsegs] first bit of the segment's first block, or kSegNone
  uint64_t* start2_bit;         // [nsegs] the next position of the segment that reads like a block's start (the probe of
                                //   zh_inflate_tokens_kernel falls back on it), or kSegNone
  uint64_t* stored_bit;         // [nsegs] first bit of the segment's first STORED block that is followed by another (a start
                                //   only where the segment has no compressed block's: zh_seg_check's merge), or kSegNone
  uint64_t* held_start;         // [nsegs] a found start that zh_seg_decide_kernel's grouping set aside (a group keeps its first
                                //   start's decoder): a repair round gets it back (kSegNone: none)
  uint64_t* end_bit;            // [nsegs] the block boundary the tokens kernel stopped at
  uint32_t* final_block;        // [nsegs] ... which was the end of the stream
  int32_t* seg_status;          // [nsegs] tokens kernel, then writer
  uint64_t* seg_out;            // [nsegs] output bytes of the segment's tokens
  uint64_t* wr_len;             // [nsegs] bytes the writer made (equal to seg_out unless it failed)
  // chain results
  uint32_t* valid;              // [nsegs] the segment lies on the stream's chain of blocks
  uint32_t* prev;               // [nsegs] the chain segment before it (0xffffffff: none)
  uint64_t* out_start;          // [nsegs] first output byte of the segment in its stream
  uint32_t* stream_ok;          // [nstreams] the chain holds: the stream is decoded segment-wise
  uint32_t* order;              // [nsegs] a stream's chain segments in order, from first_seg[i] on
  uint32_t* nchain;             // [nstreams] ... and how many they are
  uint32_t* repair;             // [nstreams] the chain did not hold, found starts that were none are out: once more (zh_seg_repair_kernel)
  uint32_t* ordinal;            // [nsegs] a chain segment's place in that order
  // 16-bit output symbols (a byte, or 0x8000 | index into the 32 KiB window before the segment)
  uint16_t* sym;
  uint16_t* winsym;             // [nsegs][32768] the last 32 KiB of output at the end of a chain segment as symbols
                                //   (0x8000 | k here also stands for byte k of the window before a short segment)
  uint8_t* windows;             // [nsegs][32768] ... and as bytes
};

// ---- wave helpers (single-wave workgroups; lockstep execution on gfx950) ----
__device__ __forceinline__ unsigned zh_lane() { return threadIdx.x & 63u; }

// Orders LDS/global traffic between lanes of one wave: nothing is emitted on the
// GPU (a wave executes in lockstep and LDS is in-order per wave); it only pins
// the compiler.  The CPU emulator used by the tests turns it into a rendezvous.
__device__ __forceinline__ void zh_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A store the caches need not keep (streamed results next to hot tables).
template <class T>
__device__ __forceinline__ void zh_store_nt(T* p, T v) {
#ifdef ZH_EMU
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}

__device__ __forceinline__ uint32_t zh_bcast(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t zh_bcast64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t zh_wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint64_t zh_wave_sum64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t zh_wave_xor(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t zh_wave_scan(uint32_t v) {
#ifdef ZH_EMU
  const unsigned lane = zh_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o, 64);
    if (lane >= (unsigned)o) v += t;
  }
  return v;
#else
  // DPP: shifts inside the four 16-lane rows, then the row totals are handed on
  // (row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143); no LDS round trip
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
#endif
}
// inclusive running maximum over the 64 lanes
__device__ __forceinline__ uint32_t zh_wave_scan_max(uint32_t v) {
#ifdef ZH_EMU
  const unsigned lane = zh_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o, 64);
    if (lane >= (unsigned)o && t > v) v = t;
  }
  return v;
#else
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return v;
#endif
}
// the shader clock (scheduling hints only: nothing that comes out depends on it; the emulator's is a sequence of
// uneven steps, so that its "costs" make a real permutation)
__device__ __forceinline__ uint64_t zh_clock() {
#ifdef ZH_EMU
  static uint64_t t = 0;  // (a sequence of uneven steps)
  return t += 1000u + (uint32_t)((t * 2654435761ull) >> 9) % 3000000u;
#else
  return __builtin_readcyclecounter();
#endif
}
__device__ __forceinline__ uint64_t zh_lanemask_lt() { return (1ull << zh_lane()) - 1ull; }

// unaligned little-endian loads from a dword-typed LDS/global array
__device__ __forceinline__ uint32_t zh_ld32(const uint32_t* w, uint32_t byte_pos) {
  uint32_t i = byte_pos >> 2;
  return __builtin_amdgcn_alignbyte(w[i + 1], w[i], byte_pos & 3u);
}
__device__ __forceinline__ uint64_t zh_ld64(const uint32_t* w, uint32_t byte_pos) {
  uint32_t i = byte_pos >> 2, s = byte_pos & 3u;
  uint32_t a = w[i], b = w[i + 1], c = w[i + 2];
  return (uint64_t)__builtin_amdgcn_alignbyte(b, a, s) |
         ((uint64_t)__builtin_amdgcn_alignbyte(c, b, s) << 32);
}
// (hi:lo) >> (sh & 31), low 32 bits
__device__ __forceinline__ uint32_t zh_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
#ifdef ZH_EMU
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u));
#else
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#endif
}
__device__ __forceinline__ uint32_t zh_ld8(const uint32_t* w, uint32_t byte_pos) {
  return (w[byte_pos >> 2] >> ((byte_pos & 3u) * 8u)) & 0xffu;
}
