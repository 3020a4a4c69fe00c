// How many returning 64-lane LDS atomics a CU serves: one wave issuing them back to back (independent,
// results consumed late) against 4 / 16 waves of one workgroup doing the same on disjoint slices.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int kWaves, int kSlots>
__global__ __launch_bounds__(64 * kWaves) void k(uint64_t* out, uint32_t* sink, uint32_t seed) {
  __shared__ uint32_t tab[kWaves * kSlots];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kWaves * kSlots; i += 64 * kWaves) tab[i] = 0;
  __syncthreads();
  uint32_t x = (lane + 1u) * 2654435761u + seed + wv * 977u, acc = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 1024; i += 8) {
    uint32_t r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      x = x * 1664525u + 1013904223u;
      r[j] = atomicMax(&tab[wv * kSlots + ((x >> 10) & (kSlots - 1))], (uint32_t)(i + j) << 8 | lane);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) acc += r[j];
  }
  __syncthreads();
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int kWaves, int kSlots>
void run(const char* what, int blocks) {
  uint64_t* d; uint32_t* s;
  hipMalloc(&d, blocks * 8); hipMalloc(&s, 4);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<kWaves, kSlots>), dim3(blocks), dim3(64 * kWaves), 0, 0, d, s, 7u + rep);
  hipDeviceSynchronize();
  uint64_t h[4096]; hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; i++) avg += h[i]; avg /= blocks;
  printf("%-46s %8.0f cycles for 1024 atomics a wave: %6.1f a wave-atomic, %6.1f a CU-atomic\n", what, avg, avg / 1024.0,
         avg / 1024.0 / kWaves);
  hipFree(d); hipFree(s);
}

int main() {
  run<1, 16384>("1 wave, 16384 slots (one workgroup a CU)", 256);
  run<1, 4096>("1 wave, 4096 slots", 256);
  run<4, 4096>("4 waves, 4096 slots each", 256);
  run<16, 1024>("16 waves, 1024 slots each", 256);
  run<16, 1024>("16 waves, 1024 slots each, 2 workgroups a CU", 512);
  return 0;
}
