// The parallel parse's link phase, a step = 64 positions against a 16384-slot table in LDS, three ways:
//   0  one returning atomicMax a lane (what zh_l1p_match_kernel does)
//   1  a plain read and a plain write a lane (misses a duplicate of the slot inside the step)
//   2  read, write, read back, and a mailbox write + read that hands a duplicate inside the step to the later lane
// by one wave alone, and by one wave while 15 others of the workgroup read LDS as fast as they can (the other
// workgroup's walks on the CU).   hipcc --offload-arch=gfx950 -O3 -o lds_link_step lds_link_step.hip && ./lds_link_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int kMode, int kNoise>
__global__ __launch_bounds__(1024) void k(uint64_t* out, uint32_t* sink, uint32_t seed) {
  __shared__ uint32_t tab[16384];
  __shared__ uint32_t mbox[64];
  __shared__ uint32_t other[8192];
  __shared__ volatile uint32_t stop;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 1024) tab[i] = 0;
  for (int i = threadIdx.x; i < 8192; i += 1024) other[i] = i * 2654435761u;
  if (threadIdx.x < 64) mbox[threadIdx.x] = 0;
  if (threadIdx.x == 0) stop = 0;
  __syncthreads();
  uint32_t acc = 0;
  if (wv == 0) {
    uint32_t x = (lane + 1u) * 2654435761u + seed;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 512; i += 8) {
      uint32_t r[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t slot = (x >> 10) & 16383u, v = ((uint32_t)((i + j) * 64 + lane) << 16) | (x & 0xffffu);
        if (kMode == 0) {
          r[j] = atomicMax(&tab[slot], v);
        } else if (kMode == 1) {
          r[j] = tab[slot];
          tab[slot] = v;
        } else {
          uint32_t old = tab[slot];
          tab[slot] = v;
          const uint32_t won = tab[slot];
          if (won != v) mbox[(won >> 16) & 63u] = v;
          const uint32_t m = mbox[lane];
          mbox[lane] = 0;
          r[j] = m ? m : old;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) acc += r[j];
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) {
      out[blockIdx.x] = t1 - t0;
      stop = 1;
    }
  } else if (kNoise) {
    uint32_t a = threadIdx.x * 4u;
    while (!stop) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint2 v = *reinterpret_cast<const uint2*>(&other[(a >> 2 << 1) & 8190u]);
        a = a * 1664525u + v.x + v.y;
      }
    }
    acc = a;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int kMode, int kNoise>
void run(const char* what) {
  const int blocks = 256;
  uint64_t* d; uint32_t* s;
  hipMalloc(&d, blocks * 8); hipMalloc(&s, 4);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<kMode, kNoise>), dim3(blocks), dim3(1024), 0, 0, d, s, 7u + rep);
  hipDeviceSynchronize();
  uint64_t h[256]; hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; i++) avg += h[i]; avg /= blocks;
  printf("%-64s %8.0f cycles for 512 steps: %6.1f a step\n", what, avg, avg / 512.0);
  hipFree(d); hipFree(s);
}

int main() {
  run<0, 0>("atomicMax, the wave alone");
  run<1, 0>("read + write, the wave alone");
  run<2, 0>("read + write + read back + mailbox, the wave alone");
  run<0, 1>("atomicMax, 15 waves reading LDS beside it");
  run<1, 1>("read + write, 15 waves reading LDS beside it");
  run<2, 1>("read + write + read back + mailbox, 15 waves reading LDS beside it");
  return 0;
}
