// In-kernel phase timers for tuning builds (-DZH_KPROF, `python -m zippy_amd.build --kprof`
// -> libzippy_hip_kprof.so, read by tools/kprof.py).  The product build compiles all of
// this away.  Slots are summed over every wave of every launch since the last reset.
#pragma once
#include <stdint.h>

#ifdef ZH_KPROF
#define ZH_KPROF_SLOTS 64
extern __device__ unsigned long long zh_kprof_slots[ZH_KPROF_SLOTS];
#define KPROF_DECL(n) unsigned long long kp_acc[n] = {}; unsigned long long kp_t = __builtin_readcyclecounter()
// charge the cycles since the previous mark to accumulator i
#define KPROF_MARK(i)                                                \
  do {                                                               \
    const unsigned long long kp_now = __builtin_readcyclecounter();  \
    kp_acc[i] += kp_now - kp_t;                                      \
    kp_t = kp_now;                                                   \
  } while (0)
#define KPROF_COUNT(i, v) (kp_acc[i] += (v))
#define KPROF_FLUSH(base, n)                                                          \
  do {                                                                                \
    if ((threadIdx.x & 63u) == 0)                                                     \
      for (int kp_i = 0; kp_i < (n); kp_i++) atomicAdd(&zh_kprof_slots[(base) + kp_i], kp_acc[kp_i]); \
  } while (0)
#else
#define KPROF_DECL(n) ((void)0)
#define KPROF_MARK(i) ((void)0)
#define KPROF_COUNT(i, v) ((void)0)
#define KPROF_FLUSH(base, n) ((void)0)
#endif
