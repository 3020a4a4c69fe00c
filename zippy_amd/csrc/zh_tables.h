// RFC 1951 constant tables (the roles of internal.nim:26-111,133-175,224-249),
// generated at compile time rather than transcribed.
#pragma once
#include <stdint.h>

namespace zh {

struct LenTables {
  uint16_t base[29];
  uint8_t extra[29];
  uint8_t index_of[256];  // (length - 3) -> length code index 0..28
};
struct DistTables {
  uint16_t base[30];
  uint8_t extra[30];
};

constexpr LenTables make_len_tables() {
  LenTables t{};
  int len = 3;
  for (int i = 0; i < 28; i++) {
    int extra = i < 8 ? 0 : (i - 4) / 4;
    t.base[i] = (uint16_t)len;
    t.extra[i] = (uint8_t)extra;
    len += 1 << extra;
  }
  t.base[28] = 258;
  t.extra[28] = 0;
  for (int l = 3; l <= 258; l++) {
    int idx = 0;
    for (int i = 0; i < 28; i++)
      if (t.base[i] <= l) idx = i;
    if (l == 258) idx = 28;
    t.index_of[l - 3] = (uint8_t)idx;
  }
  return t;
}
constexpr DistTables make_dist_tables() {
  DistTables t{};
  int d = 1;
  for (int i = 0; i < 30; i++) {
    int extra = i < 4 ? 0 : (i - 2) / 2;
    t.base[i] = (uint16_t)d;
    t.extra[i] = (uint8_t)extra;
    d += 1 << extra;
  }
  return t;
}

constexpr uint8_t kClclOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// CRC-32 (reflected 0xedb88320) byte tables T0..T3 for slice-by-4 (crc.nim:6-23)
struct CrcTables {
  uint32_t t[4][256];
};
constexpr CrcTables make_crc_tables() {
  CrcTables c{};
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t v = i;
    for (int j = 0; j < 8; j++) v = (v >> 1) ^ ((v & 1) * 0xedb88320u);
    c.t[0][i] = v;
  }
  for (int k = 1; k < 4; k++)
    for (int i = 0; i < 256; i++) c.t[k][i] = (c.t[k - 1][i] >> 8) ^ c.t[0][c.t[k - 1][i] & 255];
  return c;
}

}  // namespace zh

// distance (offset - 1 is NOT used here: argument is the distance 1..32768) -> code 0..29
__host__ __device__ inline uint32_t zh_dist_code(uint32_t dist) {
  uint32_t v = dist - 1;
  if (v < 4) return v;
  uint32_t hb = 31u - (uint32_t)__builtin_clz(v);  // v >= 4
  return 2u * hb + ((v >> (hb - 1)) & 1u);
}

// The same tables as arithmetic, for the device hot paths (a per-lane lookup in a __constant__
// array is a dependent trip to memory; these are a dozen ALU operations).  zh_selfcheck_tables()
// in the tests compares them with the tables above for every length and distance.
// length 3..258 -> code index 0..28 (internal.nim:46-73 baseLengthIndices)
__host__ __device__ inline uint32_t zh_len_code(uint32_t length) {
  const uint32_t l = length - 3u;
  if (l < 8u) return l;
  if (l == 255u) return 28u;
  const uint32_t hb = 31u - (uint32_t)__builtin_clz(l);  // 3..7
  return 4u * (hb - 1u) + ((l >> (hb - 2u)) & 3u);
}
__host__ __device__ inline uint32_t zh_len_extra_bits(uint32_t li) { return li < 8u || li == 28u ? 0u : (li - 4u) >> 2; }
__host__ __device__ inline uint32_t zh_len_base(uint32_t li) {
  return li < 8u ? li + 3u : li == 28u ? 258u : 3u + ((4u + (li & 3u)) << ((li - 4u) >> 2));
}
__host__ __device__ inline uint32_t zh_dist_extra_bits(uint32_t di) { return di < 4u ? 0u : (di - 2u) >> 1; }
__host__ __device__ inline uint32_t zh_dist_base(uint32_t di) {
  return di < 4u ? di + 1u : 1u + ((2u + (di & 1u)) << ((di - 2u) >> 1));
}

