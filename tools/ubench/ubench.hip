// Single-wave instruction-cost microbenchmarks for gfx950 (tuning aid, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N 4096
__global__ __launch_bounds__(64) void k(uint64_t* out, uint32_t* sink, int which, uint32_t seed) {
  __shared__ uint32_t lds[4096];
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (i * 7 + seed) & 4095;
  __syncthreads();
  uint32_t s = __builtin_amdgcn_readfirstlane(seed), v = lane + seed, acc = 0;
  uint64_t m = 0x123456789abcdefull ^ seed;
  uint64_t t0 = __builtin_readcyclecounter();
  switch (which) {
    case 0:  // dependent SALU adds
#pragma unroll 64
      for (int i = 0; i < N; i++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
      break;
    case 1:  // dependent VALU adds
#pragma unroll 64
      for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(v));
      break;
    case 2:  // loop with a taken branch per iteration (1 SALU + cmp + branch)
      for (int i = 0; i < N; i++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
      break;
    case 3:  // v_readlane -> SALU -> v_readlane dependent chain (lane select from result)
#pragma unroll 16
      for (int i = 0; i < N; i++) {
        uint32_t r = __builtin_amdgcn_readlane(v, s & 63u);
        s = r + 1;
      }
      break;
    case 4:  // dependent LDS read chain (pointer chase, uniform address)
#pragma unroll 16
      for (int i = 0; i < N; i++) v = lds[v & 4095u];
      break;
    case 5:  // ds_bpermute dependent chain
#pragma unroll 16
      for (int i = 0; i < N; i++) v = __builtin_amdgcn_ds_bpermute((v & 63u) << 2, v + 1);
      break;
    case 6:  // 64-bit scalar shift/and/or dependent chain (3 ops per iteration)
#pragma unroll 16
      for (int i = 0; i < N; i++) {
        m = (m << (s & 7u)) | (m >> 61);
        s = (uint32_t)m;
      }
      break;
    case 7:  // ballot -> ffs -> readlane chain (typical walk step)
#pragma unroll 8
      for (int i = 0; i < N; i++) {
        uint64_t b = __ballot(v > s);
        uint32_t g = b ? __builtin_ctzll(b) : 0;
        s = __builtin_amdgcn_readlane(v, g) & 0xffff;
        v += 1;
      }
      break;
    case 8:  // data-dependent (unpredictable) scalar branch per iteration
      for (int i = 0; i < N; i++) {
        if (s & 1) s = s * 3 + 1; else s = (s >> 1) + 7;
        asm volatile("" : "+s"(s));
      }
      break;
    case 9:  // independent SALU pairs (2 chains)
      {
        uint32_t s2 = s ^ 5;
#pragma unroll 64
        for (int i = 0; i < N; i++) { asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc"); asm volatile("s_add_u32 %0, %0, 5" : "+s"(s2) : : "scc"); }
        s += s2;
      }
      break;
    case 10:  // DPP row_shr add chain
#pragma unroll 16
      for (int i = 0; i < N; i++) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
      break;
    case 11:  // LDS atomic add then read (round trip with return)
#pragma unroll 8
      for (int i = 0; i < N; i++) v = atomicAdd(&lds[v & 4095u], 1u);
      break;
  }
  uint64_t t1 = __builtin_readcyclecounter();
  acc = s + v + (uint32_t)m;
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + lane] = acc;
}

int main() {
  uint64_t* out; uint32_t* sink;
  (void)hipMalloc(&out, 8 * 8192); (void)hipMalloc(&sink, 4 * 64 * 8192);
  const char* names[] = {"dependent s_add", "dependent v_add", "loop: s_add+cmp+taken branch", "readlane->salu chain",
                         "LDS read chain", "ds_bpermute chain", "64-bit salu x3 chain", "ballot+ffs+readlane chain",
                         "data-dependent scalar branch", "2 independent s_add chains (per pair)", "DPP add chain", "LDS atomic rtn chain"};
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (int blocks : {1, 1024, 4096}) {
    printf("== %d single-wave blocks\n", blocks);
    for (int w = 0; w < 12; w++) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, sink, w, 12345u);
      (void)hipDeviceSynchronize();
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, sink, w, 777u);
      (void)hipDeviceSynchronize();
      uint64_t h[4096];
      (void)hipMemcpy(h, out, 8 * blocks, hipMemcpyDeviceToHost);
      double sum = 0; for (int i = 0; i < blocks; i++) sum += (double)h[i];
      printf("  %-40s %7.2f cycles/iter\n", names[w], sum / blocks / N);
    }
  }
  return 0;
}
