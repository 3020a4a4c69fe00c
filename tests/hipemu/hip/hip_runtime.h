// TEST INFRASTRUCTURE ONLY -- a tiny single-process emulator of the HIP subset
// used by zippy_amd/csrc, so that the *same* kernel sources can be logic-tested
// with g++ in the GPU-less build container (tests/test_emu_*.py).  It is never
// linked into the product library (zippy_amd/libzippy_hip.so is built by hipcc
// against the real ROCm headers) and nothing in zippy_amd/ loads it.
//
// Model: every thread of a workgroup is a fiber (own stack, cooperative
// switching); one workgroup runs at a time; a wave is 64 consecutive threads.
// Cross-lane operations (__ballot, __shfl, readlane ...) and __syncthreads are
// rendezvous points: all live lanes of the wave (block) must reach the same call
// site sequence, otherwise the emulator aborts -- i.e. kernels must keep
// cross-lane ops in wave-uniform control flow, which is also what gfx950 wants.
#pragma once
#define ZH_EMU 1
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __constant__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct emu_stream_t* hipStream_t;
typedef struct emu_event_t* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
extern emu_uint3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
// rendezvous of the calling lane's wave; returns pointer to the 64-slot exchange
// array (uint64 per lane; dead lanes hold 0) valid until the lane's next collective.
const uint64_t* wave_exchange(uint64_t my_value, uint64_t* live_mask);
void block_barrier();
unsigned lane_id();
extern unsigned char* dyn_shared;
}  // namespace emu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })

#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::dyn_shared;

// ---- runtime API (host memory stands in for device memory) ----
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc((void**)p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// ---- device intrinsics ----
inline void __syncthreads() { emu::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

inline unsigned long long __ballot(int pred) {
  uint64_t live;
  const uint64_t* ex = emu::wave_exchange(pred ? 1 : 0, &live);
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) if (ex[i]) m |= 1ull << i;
  return m & live;
}
template <class T> inline T emu_shfl_from(T v, unsigned src) {
  static_assert(sizeof(T) <= 8, "shfl width");
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  uint64_t live;
  const uint64_t* ex = emu::wave_exchange(raw, &live);
  uint64_t r = ex[src & 63];
  T out;
  memcpy(&out, &r, sizeof(T));
  return out;
}
template <class T> inline T __shfl(T v, int src_lane, int width = 64) {
  unsigned l = emu::lane_id();
  unsigned base = l & ~(unsigned)(width - 1);
  return emu_shfl_from(v, base + ((unsigned)src_lane & (unsigned)(width - 1)));
}
template <class T> inline T __shfl_up(T v, unsigned delta, int width = 64) {
  unsigned l = emu::lane_id();
  unsigned pos = l & (unsigned)(width - 1);
  return emu_shfl_from(v, pos >= delta ? l - delta : l);
}
template <class T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
  unsigned l = emu::lane_id();
  unsigned pos = l & (unsigned)(width - 1);
  return emu_shfl_from(v, pos + delta < (unsigned)width ? l + delta : l);
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
  unsigned l = emu::lane_id();
  return emu_shfl_from(v, l ^ (unsigned)mask);
}
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline unsigned __brev(unsigned v) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
  return __builtin_bswap32(v);
}

// names of the amdgcn builtins the kernels use (wrapped in zh_wave.h)
// (both return int, like the real builtins: a careless widening sign-extends here too)
inline int emu_amdgcn_readfirstlane(unsigned v) {
  uint64_t live;
  const uint64_t* ex = emu::wave_exchange(v, &live);
  return (int)(live ? (unsigned)ex[__builtin_ctzll(live)] : v);
}
inline int emu_amdgcn_readlane(unsigned v, unsigned lane) {
  uint64_t live;
  const uint64_t* ex = emu::wave_exchange(v, &live);
  return (int)(unsigned)ex[lane & 63];
}
inline unsigned emu_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) {
  return (unsigned)(((((uint64_t)hi) << 32) | lo) >> ((sh & 3) * 8));
}
#define __builtin_amdgcn_readfirstlane emu_amdgcn_readfirstlane
#define __builtin_amdgcn_readlane emu_amdgcn_readlane
#define __builtin_amdgcn_alignbyte emu_amdgcn_alignbyte
inline void emu_amdgcn_wave_barrier() { uint64_t live; (void)emu::wave_exchange(0, &live); }
#define __builtin_amdgcn_wave_barrier emu_amdgcn_wave_barrier
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)

#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(ptr, order, scope) (*(ptr))
#define __hip_atomic_store(ptr, val, order, scope) (*(ptr) = (val))
#define __hip_atomic_exchange(ptr, val, order, scope) atomicExch(ptr, val)
// atomics: fibers are cooperative, so plain read-modify-write is atomic
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> inline T atomicSub(T* p, T v) { T o = *p; *p = (T)(o - v); return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> inline T atomicAnd(T* p, T v) { T o = *p; *p = (T)(o & v); return o; }
template <class T> inline T atomicXor(T* p, T v) { T o = *p; *p = (T)(o ^ v); return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
