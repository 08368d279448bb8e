// Flat-buffer optimiser and EMA kernels (HBM-bound streaming, float4 per lane, grid-stride).
//
// The host keeps every model's parameters in ONE contiguous fp32 buffer (gradients and momentum
// likewise), so the 132-tensor loops of the reference -- engine/MTtrainer.py:277-281 (EMA: 264 tiny
// launches) and torch.optim.SGD over 121 parameter groups (solver/build.py:5-23) -- become one launch
// each.  Algorithmic HBM traffic: EMA 3 x 4 B/param (read t, read s, write t); SGD 5 x 4 B/param.
#include "common.h"

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ t, const float* __restrict__ s, long n4,
                                                  long n, float alpha, float beta) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 a = ((const f32x4*)t)[i];
    const f32x4 b = ((const f32x4*)s)[i];
#pragma unroll
    for (int e = 0; e < 4; e++) a[e] = fmaf(b[e], beta, a[e] * alpha);  // mul_(alpha).add_(s, alpha=1-alpha)
    ((f32x4*)t)[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    t[i] = fmaf(s[i], beta, t[i] * alpha);
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, long n4, long n, float lr, float wd,
                                                  float mom, int first) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 w = ((f32x4*)p)[i];
    const f32x4 gr = ((const f32x4*)g)[i];
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (!first) b = ((f32x4*)buf)[i];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float d = gr[e] + wd * w[e];
      b[e] = first ? d : mom * b[e] + d;
      w[e] = w[e] - lr * b[e];
    }
    ((f32x4*)buf)[i] = b;
    ((f32x4*)p)[i] = w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float d = g[i] + wd * p[i];
    const float b = first ? d : mom * buf[i] + d;
    buf[i] = b;
    p[i] = p[i] - lr * b;
  }
}

static int stream_blocks(long n4) {
  long b = (n4 + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int mmt_ema_update(float* teacher, const float* student, int64_t n, double alpha, void* stream) {
  if (n <= 0) return 0;
  if (((uintptr_t)teacher | (uintptr_t)student) & 15) return MMT_EINVAL;
  hipLaunchKernelGGL(ema_kernel, dim3(stream_blocks(n >> 2)), dim3(256), 0, (hipStream_t)stream, teacher, student,
                     (long)(n >> 2), (long)n, (float)alpha, (float)(1.0 - alpha));
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float wd, float momentum,
                                int first, void* stream) {
  if (n <= 0) return 0;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) return MMT_EINVAL;
  hipLaunchKernelGGL(sgd_kernel, dim3(stream_blocks(n >> 2)), dim3(256), 0, (hipStream_t)stream, p, g, buf,
                     (long)(n >> 2), (long)n, lr, wd, momentum, first);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_version(void) { return 1; }
