// CRC-32 / Adler-32 of byte ranges in HBM ("pieces" of <= 32 KiB) plus the
// GF(2) / modular combination into whole-buffer checksums.
//
// Replaces crc.nim:29-72 (slice-by-8 / PCLMUL folding; gfx950 has no carry-less
// multiply) and adler32.nim:19-63.  One 64-lane wave per piece (four waves a workgroup share the tables):
//   * lane k owns the 32-byte column k of every 2 KiB row: two 16-byte loads a lane whose 64 pieces each lie 32
//     bytes apart (16 bytes a lane and 1 KiB rows, fully coalesced, until round 5: the skip below is then paid twice
//     as often -- 1.30 against 1.23 ms for 4 GiB; at 64 bytes a lane the loads' stride costs more than the skips
//     save, 1.80 ms);
//   * per row a lane takes its 16 bytes a dword at a time through THREE tables of 11 + 11 + 10 index bits
//     (what advancing the state by four bytes does is linear in the state's bits: any split of the 32 will
//     do; the kernel is bound by its LDS lookups -- a 64-lane lookup at random addresses is ~ 16 cycles of
//     the CU's LDS --, so a dword is 3 of them instead of slice-by-4's 4) and then "skips" the other lanes'
//     bytes of the row with one 4-lookup multiplication by x^(8*(row - its own bytes)) mod P (tables Z0..Z3);
//   * lanes are aligned to the end of the piece by one multiplication with
//     x^(8*d) (d from a small table) and XOR-reduced across the wave.
// Algorithmic traffic: each input byte is read once.
#include <mutex>

#include "zh_common.h"
#include "zh_tables.h"

namespace {

constexpr uint32_t kPoly = 0xedb88320u;

// a(x) * b(x) mod P in the reflected representation (x^0 == bit 31), the role of
// zlib's multmodp; 32 shift/xor steps.
__host__ __device__ inline uint32_t gf2_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; i++) {
    p ^= (a & 0x80000000u) ? b : 0u;
    a <<= 1;
    b = (b & 1u) ? (b >> 1) ^ kPoly : (b >> 1);
  }
  return p;
}

// x^(8*n) mod P
__host__ __device__ inline uint32_t gf2_xpow8(uint64_t n) {
  uint32_t result = 0x80000000u;   // x^0
  uint32_t sq = 0x00800000u;       // x^8
  while (n) {
    if (n & 1) result = gf2_mul(sq, result);
    sq = gf2_mul(sq, sq);
    n >>= 1;
  }
  return result;
}

#ifndef ZH_CK_LANE
#define ZH_CK_LANE 32
#endif
constexpr uint32_t kLaneBytes = ZH_CK_LANE;       // contiguous bytes a lane owns in a row (16-byte loads)
constexpr uint32_t kRowBytes = 64u * kLaneBytes;  // a wave's row
static_assert(kLaneBytes % 16u == 0 && kLaneBytes <= 64u, "");
struct ChecksumTables {
  uint32_t t0[256];     // one byte (crc.nim's table 0): heads and tails
  uint32_t a[3][2048];  // four bytes: state ^ dword -> a[0][bits 0..10] ^ a[1][bits 11..21] ^ a[2][bits 22..31]
  uint32_t z[4][256];   // multiply a state by x^(8*(kRowBytes - kLaneBytes)): Zj[b] = (b << 8j) * x^(...), the other lanes' bytes of a row
  uint32_t xz[2 * kRowBytes];  // x^(8*j) mod P
  uint32_t zp[4][256];  // multiply a state by x^(8*32768), a whole piece: the combine's step
  uint32_t init_whole;  // what the initial 0xffffffff has become behind a whole piece: 0xffffffff * x^(8*32768)
};
constexpr uint32_t kWavesPerGroup = 4;

}  // namespace


__global__ __launch_bounds__(64 * kWavesPerGroup) void zh_checksum_pieces_kernel(
    const uint8_t* __restrict__ d_data, const ZhPieceDesc* __restrict__ pieces, uint32_t npieces,
    const uint64_t* __restrict__ dyn_len, const ChecksumTables* __restrict__ tabs, int want_crc,
    int want_adler, uint32_t* __restrict__ out_crc, uint32_t* __restrict__ out_adler,
    uint32_t* __restrict__ out_len) {
  __shared__ uint32_t s_t0[256];
  __shared__ uint32_t s_a0[2048], s_a1[2048], s_a2[1024];
  __shared__ uint32_t s_z[4][256];
  const unsigned lane = zh_lane();
  if (want_crc) {
    for (unsigned i = threadIdx.x; i < 2048; i += 64 * kWavesPerGroup) {
      s_a0[i] = tabs->a[0][i];
      s_a1[i] = tabs->a[1][i];
      if (i < 1024) {
        s_a2[i] = tabs->a[2][i];
        (&s_z[0][0])[i] = (&tabs->z[0][0])[i];
      }
      if (i < 256) s_t0[i] = tabs->t0[i];
    }
  }
  __syncthreads();

  // (a wave a piece; no barrier below)
  for (uint32_t p = blockIdx.x * kWavesPerGroup + (threadIdx.x >> 6); p < npieces; p += gridDim.x * kWavesPerGroup) {
    const ZhPieceDesc pd = pieces[p];
    uint32_t len = pd.len;
    if (dyn_len) {
      uint64_t total = dyn_len[pd.buf];
      len = total > pd.rel_off ? (uint32_t)(total - pd.rel_off < 32768u ? total - pd.rel_off : 32768u) : 0u;
      // never past the slot: pd.len is the piece's share of the slot's capacity (a stream that
      // failed with DST_TOO_SMALL must not make this kernel read behind its slot)
      if (len > pd.len) len = pd.len;
    }
    const uint8_t* base = d_data + pd.off;
    // head: bytes before the first 16-byte boundary (handled by lane 0)
    uint32_t head = (uint32_t)((16u - ((uintptr_t)base & 15u)) & 15u);
    if (head > len) head = len;
    const uint32_t body = len - head;
    const uint32_t rows = body / kRowBytes;
    const uint32_t tail = body % kRowBytes;
    const uint8_t* bp = base + head;

    uint32_t crc_rows = 0, crc_tail = 0, crc_head = 0;
    uint64_t sum_b = 0, sum_ib = 0;  // adler: sum of bytes, sum of index*byte

    if (lane == 0) {
      for (uint32_t i = 0; i < head; i++) {
        uint32_t b = base[i];
        if (want_crc) crc_head = s_t0[(crc_head ^ b) & 255u] ^ (crc_head >> 8);
        sum_b += b;
        sum_ib += (uint64_t)i * b;
      }
    }
    auto four_bytes = [&](uint32_t state, uint32_t w) -> uint32_t {
      const uint32_t c = state ^ w;
      return s_a0[c & 2047u] ^ s_a1[(c >> 11) & 2047u] ^ s_a2[c >> 22];
    };
    auto adler_dword = [&](uint32_t w, uint32_t at) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t b = (w >> (8 * j)) & 255u;
        sum_b += b;
        sum_ib += (uint64_t)(at + j) * b;
      }
    };
    // (a row ahead: the next row's bytes are asked for before this row's go through the tables)
    constexpr uint32_t kVec = kLaneBytes / 16u;
    uint4 vn[kVec];
#pragma unroll
    for (uint32_t q = 0; q < kVec; q++)
      vn[q] = rows ? *reinterpret_cast<const uint4*>(bp + lane * kLaneBytes + 16u * q) : make_uint4(0, 0, 0, 0);
    for (uint32_t r = 0; r < rows; r++) {
      uint4 v[kVec];
#pragma unroll
      for (uint32_t q = 0; q < kVec; q++) {
        v[q] = vn[q];
        vn[q] = *reinterpret_cast<const uint4*>(bp + (size_t)(r + 1u < rows ? r + 1u : r) * kRowBytes + lane * kLaneBytes + 16u * q);
      }
      if (want_crc && r) {  // skip the other 63 lanes' bytes between this lane's bytes of two rows
        crc_rows = s_z[0][crc_rows & 255u] ^ s_z[1][(crc_rows >> 8) & 255u] ^
                   s_z[2][(crc_rows >> 16) & 255u] ^ s_z[3][crc_rows >> 24];
      }
#pragma unroll
      for (uint32_t q = 0; q < kVec; q++) {
        const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (want_crc) crc_rows = four_bytes(crc_rows, w[k]);
          if (want_adler) adler_dword(w[k], head + r * kRowBytes + lane * kLaneBytes + 16u * q + 4u * k);
        }
      }
    }
    // tail: lane k takes bytes [kLaneBytes k, kLaneBytes (k + 1)) of the last partial row, whole dwords first
    uint32_t t_begin = lane * kLaneBytes, t_end = t_begin + kLaneBytes;
    if (t_begin > tail) t_begin = tail;
    if (t_end > tail) t_end = tail;
    {
      const uint8_t* tp = bp + (size_t)rows * kRowBytes;
      uint32_t i = t_begin;
      for (; i + 4u <= t_end; i += 4u) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(tp + i);  // (bp is 16-byte aligned, i a multiple of four)
        if (want_crc) crc_tail = four_bytes(crc_tail, w);
        if (want_adler) adler_dword(w, head + rows * kRowBytes + i);
      }
      for (; i < t_end; i++) {
        const uint32_t b = tp[i];
        if (want_crc) crc_tail = s_t0[(crc_tail ^ b) & 255u] ^ (crc_tail >> 8);
        sum_b += b;
        sum_ib += (uint64_t)(head + rows * kRowBytes + i) * b;
      }
    }

    if (want_crc) {
      // align every partial state to the end of the piece and fold
      uint32_t acc = 0;
      if (rows) acc ^= gf2_mul(tabs->xz[kLaneBytes * (63u - lane) + tail], crc_rows);
      if (t_end > t_begin) acc ^= gf2_mul(tabs->xz[tail - t_end], crc_tail);
      acc = zh_wave_xor(acc);
      if (lane == 0) {
        if (head) acc ^= gf2_mul(gf2_xpow8(len - head), crc_head);
        // standard conditioning: init 0xffffffff travels through len bytes, final NOT.  (For a whole piece a constant:
        // gf2_xpow8() is thirty 32-step multiplications by ONE lane -- the counters had three quarters of this kernel's
        // vector instructions on a single lane.)
        uint32_t crc = ~(acc ^ (len == 32768u ? tabs->init_whole : gf2_mul(gf2_xpow8(len), 0xffffffffu)));
        out_crc[p] = crc;
      }
    }
    if (want_adler) {
      sum_b = zh_wave_sum64(sum_b);
      sum_ib = zh_wave_sum64(sum_ib);
      if (lane == 0) {
        // s1 = 1 + sum b ; s2 = len + sum (len - i) * b   (adler32.nim:28-31 unrolled)
        uint64_t s1 = (1 + sum_b) % 65521u;
        uint64_t s2 = ((uint64_t)len + (uint64_t)len * sum_b - sum_ib) % 65521u;
        out_adler[p] = (uint32_t)((s2 << 16) | s1);
      }
    }
    if (lane == 0) out_len[p] = len;
  }
}

// Combine per-piece checksums of each buffer in order:
//   crc(A||B)   = crc(A) * x^(8 len B) xor crc(B)                (zlib crc32_combine)
//   adler(A||B) : s1 = s1A + s1B - 1 ; s2 = s2A + s2B + lenB * (s1A - 1)   (mod 65521)
// One wave per buffer: lane l folds its contiguous share of the pieces, then the 64 partial
// results are folded pairwise (both combinations are associative).
__global__ __launch_bounds__(64) void zh_checksum_combine_kernel(
    const ZhBufDesc* __restrict__ bufs, uint32_t nbufs, const uint32_t* __restrict__ piece_crc,
    const uint32_t* __restrict__ piece_adler, const uint32_t* __restrict__ piece_len, int want_crc,
    int want_adler, uint32_t* __restrict__ buf_crc, uint32_t* __restrict__ buf_adler,
    const ChecksumTables* __restrict__ tabs) {
  const uint32_t i = blockIdx.x;
  const unsigned lane = threadIdx.x & 63u;
  const ZhBufDesc b = bufs[i];
  const uint32_t share = (b.npieces + 63u) / 64u;
  uint32_t k = lane * share, k_end = k + share;
  if (k > b.npieces) k = b.npieces;
  if (k_end > b.npieces) k_end = b.npieces;
  uint32_t crc = 0;
  uint64_t s1 = 1, s2 = 0, total = 0;
  const uint32_t x_full = gf2_xpow8(32768);
  // (a lane with many pieces -- one buffer of GiBs -- folds them as a chain of table look-ups: the table in LDS, a
  // hundred cycles a link instead of a trip to L2)
  __shared__ uint32_t s_zp[4][256];
  const bool in_lds = want_crc && share > 4u;
  if (in_lds)
    for (uint32_t t = lane; t < 1024u; t += 64u) (&s_zp[0][0])[t] = (&tabs->zp[0][0])[t];
  zh_wave_sync();
  // (eight pieces' lengths and checksums asked for at once: a piece a trip to L2 was what one buffer of GiBs waited for)
  constexpr uint32_t kAhead = 8;
  for (; k < k_end; k += kAhead) {
    uint32_t lens[kAhead], crcs[kAhead], adlers[kAhead];
#pragma unroll
    for (uint32_t u = 0; u < kAhead; u++) {
      const uint32_t p = b.first_piece + (k + u < k_end ? k + u : k_end - 1u);
      lens[u] = piece_len[p];
      crcs[u] = want_crc ? piece_crc[p] : 0u;
      adlers[u] = want_adler ? piece_adler[p] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < kAhead; u++) {
      const uint32_t len = k + u < k_end ? lens[u] : 0u;
      if (len == 0) continue;
      total += len;
      if (want_crc) {
        // (a whole piece behind: four table entries instead of a 32-step multiplication -- one buffer of 4 GiB is
        // 2048 pieces a lane)
        if (len == 32768u && in_lds)
          crc = s_zp[0][crc & 255u] ^ s_zp[1][(crc >> 8) & 255u] ^ s_zp[2][(crc >> 16) & 255u] ^ s_zp[3][crc >> 24];
        else if (len == 32768u)
          crc = tabs->zp[0][crc & 255u] ^ tabs->zp[1][(crc >> 8) & 255u] ^ tabs->zp[2][(crc >> 16) & 255u] ^ tabs->zp[3][crc >> 24];
        else
          crc = gf2_mul(gf2_xpow8(len), crc);
        crc ^= crcs[u];
      }
      if (want_adler) {
        const uint32_t a = adlers[u];
        const uint64_t s1b = a & 0xffffu, s2b = a >> 16;
        s2 = (s2 + s2b + (uint64_t)(len % 65521u) * ((s1 + 65520u) % 65521u)) % 65521u;
        s1 = (s1 + s1b + 65520u) % 65521u;
      }
    }
  }
  // x^(8 * bytes) for a right-hand side of `share << k` whole pieces, squared from level to level: a tree of
  // gf2_xpow8() calls (twenty squarings and multiplications each) was most of this kernel; only the side that holds a
  // buffer's short last piece still needs one
  uint32_t x_whole = share == 1u ? x_full : gf2_xpow8((uint64_t)share << 15);
  uint64_t whole = (uint64_t)share << 15;
  for (unsigned d = 1; d < 64; d <<= 1, x_whole = gf2_mul(x_whole, x_whole), whole <<= 1) {  // (this) || (lane + d)
    const uint32_t crc_r = (uint32_t)__shfl((int)crc, (int)((lane + d) & 63u), 64);
    const uint64_t tot_r = (uint64_t)(uint32_t)__shfl((int)(uint32_t)total, (int)((lane + d) & 63u), 64) |
                           ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(total >> 32), (int)((lane + d) & 63u), 64) << 32);
    const uint32_t s1_r = (uint32_t)__shfl((int)(uint32_t)s1, (int)((lane + d) & 63u), 64);
    const uint32_t s2_r = (uint32_t)__shfl((int)(uint32_t)s2, (int)((lane + d) & 63u), 64);
    if ((lane & (2 * d - 1)) == 0 && lane + d < 64) {
      if (want_crc) crc = gf2_mul(tot_r == whole ? x_whole : gf2_xpow8(tot_r), crc) ^ crc_r;
      if (want_adler) {
        s2 = (s2 + s2_r + (tot_r % 65521u) * ((s1 + 65520u) % 65521u)) % 65521u;
        s1 = (s1 + s1_r + 65520u) % 65521u;
      }
      total += tot_r;
    }
  }
  if (lane == 0) {
    if (want_crc) buf_crc[i] = crc;
    if (want_adler) buf_adler[i] = (uint32_t)((s2 << 16) | s1);
  }
}

// ---- host side ------------------------------------------------------------
// one table set per device, shared by the contexts on it (contexts may be created concurrently)
static ChecksumTables* g_tabs_dev[64] = {nullptr};
static std::mutex g_tabs_mutex;

extern "C" const void* zh_checksum_tables(int device) {
  if (device < 0 || device >= 64) device = 0;
  std::lock_guard<std::mutex> lock(g_tabs_mutex);
  if (g_tabs_dev[device]) return g_tabs_dev[device];
  ChecksumTables* h = new ChecksumTables;
  constexpr zh::CrcTables ct = zh::make_crc_tables();
  for (int i = 0; i < 256; i++) h->t0[i] = ct.t[0][i];
  // what slice-by-4 does to a state: linear, so the contribution of any group of its bits is a table
  auto four_bytes = [&](uint32_t c) {
    return ct.t[3][c & 255u] ^ ct.t[2][(c >> 8) & 255u] ^ ct.t[1][(c >> 16) & 255u] ^ ct.t[0][c >> 24];
  };
  for (uint32_t v = 0; v < 2048; v++) {
    h->a[0][v] = four_bytes(v);
    h->a[1][v] = four_bytes(v << 11);
    h->a[2][v] = v < 1024 ? four_bytes(v << 22) : 0u;
  }
  const uint32_t x_skip = gf2_xpow8(kRowBytes - kLaneBytes);
  for (int j = 0; j < 4; j++)
    for (uint32_t b = 0; b < 256; b++) h->z[j][b] = gf2_mul(x_skip, b << (8 * j));
  const uint32_t x32768 = gf2_xpow8(32768);
  h->init_whole = gf2_mul(x32768, 0xffffffffu);
  for (int j = 0; j < 4; j++)
    for (uint32_t b = 0; b < 256; b++) h->zp[j][b] = gf2_mul(x32768, b << (8 * j));
  uint32_t x = 0x80000000u;
  const uint32_t x8 = 0x00800000u;
  for (uint32_t j = 0; j < 2 * kRowBytes; j++) {
    h->xz[j] = x;
    x = gf2_mul(x8, x);
  }
  ChecksumTables* d = nullptr;
  if (hipMalloc(&d, sizeof(ChecksumTables)) != hipSuccess) {
    delete h;
    return nullptr;
  }
  const hipError_t e = hipMemcpy(d, h, sizeof(ChecksumTables), hipMemcpyHostToDevice);
  delete h;
  if (e != hipSuccess) {
    (void)hipFree(d);
    return nullptr;
  }
  g_tabs_dev[device] = d;
  return d;
}

extern "C" void zh_launch_checksum_pieces(hipStream_t stream, const void* tabs, const uint8_t* d_data,
                                          const ZhPieceDesc* pieces, uint32_t npieces,
                                          const uint64_t* dyn_len, int want_crc, int want_adler,
                                          uint32_t* out_crc, uint32_t* out_adler, uint32_t* out_len) {
  if (!npieces) return;
  // (25 KiB of tables a workgroup: six of them a CU, and each takes them from L2 once)
  uint32_t grid = (npieces + kWavesPerGroup - 1u) / kWavesPerGroup;
  if (grid > 3072u) grid = 3072u;
  hipLaunchKernelGGL(zh_checksum_pieces_kernel, dim3(grid), dim3(64 * kWavesPerGroup), 0, stream, d_data, pieces,
                     npieces, dyn_len, (const ChecksumTables*)tabs, want_crc, want_adler, out_crc,
                     out_adler, out_len);
}

extern "C" void zh_launch_checksum_combine(hipStream_t stream, const void* tabs, const ZhBufDesc* bufs, uint32_t nbufs,
                                           const uint32_t* piece_crc, const uint32_t* piece_adler,
                                           const uint32_t* piece_len, int want_crc, int want_adler,
                                           uint32_t* buf_crc, uint32_t* buf_adler) {
  if (!nbufs) return;
  hipLaunchKernelGGL(zh_checksum_combine_kernel, dim3(nbufs), dim3(64), 0, stream, bufs,
                     nbufs, piece_crc, piece_adler, piece_len, want_crc, want_adler, buf_crc,
                     buf_adler, (const ChecksumTables*)tabs);
}
