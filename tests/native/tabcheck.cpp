#include <cstdio>
#include <cstdint>
#define __host__
#define __device__
#include "zh_tables.h"
int main() {
  constexpr zh::LenTables L = zh::make_len_tables();
  constexpr zh::DistTables D = zh::make_dist_tables();
  int bad = 0;
  for (uint32_t len = 3; len <= 258; len++) {
    uint32_t li = zh_len_code(len);
    if (li != L.index_of[len - 3] || zh_len_base(li) != L.base[li] || zh_len_extra_bits(li) != L.extra[li]) { bad++; printf("len %u\n", len); }
  }
  for (uint32_t d = 1; d <= 32768; d++) {
    uint32_t di = zh_dist_code(d);
    if (di >= 30 || zh_dist_base(di) != D.base[di] || zh_dist_extra_bits(di) != D.extra[di] || d < D.base[di] || d - D.base[di] >= (1u << D.extra[di])) { bad++; if (bad < 10) printf("dist %u\n", d); }
  }
  printf("bad %d\n", bad);
  return bad != 0;
}
