// Common helpers for the gfx950 kernels of libmmtpsm.so (see include/mmtpsm.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmtpsm.h"

#define MMT_LAUNCH_CHECK()                    \
  do {                                        \
    hipError_t e__ = hipGetLastError();       \
    if (e__ != hipSuccess) return (int)e__;   \
  } while (0)

static inline int mmt_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
