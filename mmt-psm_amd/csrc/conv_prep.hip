// Operand preparation and the small passes around the convolutions: weight matrices -> K-panel planes
// (bf16 x 3 and two-term fp16, forward and flipped for the data gradient, one launch per model), activation tensors -> planes for the
// callers that still ask for a pass of their own, the magnitude statistics the fp16 split's scales and range guard read, the
// residual / multi-input sums with statistics, the dense flipped weight of the fp32 data gradient, and the stem's 3x3 / stride 2
// max-pool (backbone/resnet.py:324-330 in the reference).  All HBM-bound element-wise work: 16-byte accesses, grid-stride.
#include <stdlib.h>
#include <type_traits>
#include "conv_shared.h"

namespace {

// x (n fp32 values, n % 8 == 0) -> NS bf16 planes of the same indexing: x = p0 + p1 + p2, round-to-nearest at each level
// (the split the kernels above do in registers, done ONCE per tensor instead of once per use)
template <int NS>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ pl,
                                                           const long plane_stride, const long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const f32x4 v0 = ((const f32x4*)x)[2 * i], v1 = ((const f32x4*)x)[2 * i + 1];
    uint2 o0[NS], o1[NS];
    split4<NS>(v0, o0);
    split4<NS>(v1, o1);
#pragma unroll
    for (int q = 0; q < NS; q++)
      ((uint4*)(pl + q * plane_stride))[i] = uint4{o0[q].x, o0[q].y, o1[q].x, o1[q].y};
  }
}

// one atomic per BLOCK, and only when the block's maximum beats what is already there (atomics on one address serialise
// at the L2: 32 k of them cost milliseconds): wave reduce, LDS reduce over the 4 waves, test, atomicMax
__device__ __forceinline__ void block_amax_commit(float m, unsigned* __restrict__ out) {
  __shared__ float wmax[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const unsigned bits = __builtin_bit_cast(unsigned, m);
    if (m > 0.f && bits > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, bits);
  }
}

// out[0] = max(out[0], max |x[i] * rowscale[(i / inner) % rows]|) as a float (non-negative floats order like their bits)
template <bool STATS>  // STATS: out is a 33-float slot, sum |x| and count of every 16th block's share added to it (see ConvP.amax_stats)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, const long n4, const float* __restrict__ rowscale,
                                                   const long inner4, const int rows, unsigned* __restrict__ out) {
  float m = 0.f, sum = 0.f, cnt = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)x)[i];
    const float rs = rowscale ? fabsf(rowscale[(i / inner4) % rows]) : 1.f;
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) * rs);
    if (STATS) { sum += ((fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3]))) * rs; cnt += 4.f; }
  }
  block_amax_commit(m, out);
  if (STATS && (blockIdx.x & 15) == 0) {   // the grid-stride loop gives every block a share spread over the whole tensor
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if ((threadIdx.x & 63) == 0 && cnt > 0.f) {
      const int k = (blockIdx.x >> 4) & 15;
      atomicAdd((float*)out + 1 + k, sum);
      atomicAdd((float*)out + 17 + k, cnt);
    }
  }
}

// out = a + b (+ c (+ d)) in that order, and the statistics of the sum into a 33-float slot as amax_kernel<true> records them: the
// gradient of a tensor with several consumers in ONE pass (instead of n - 1 library additions and a reduction pass)
__global__ __launch_bounds__(256) void sum_stats_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ c, const float* __restrict__ d,
                                                        float* __restrict__ y, const long n4, unsigned* __restrict__ out) {
  float m = 0.f, sum = 0.f, cnt = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = ((const f32x4*)a)[i] + ((const f32x4*)b)[i];
    if (c) v += ((const f32x4*)c)[i];
    if (d) v += ((const f32x4*)d)[i];
    ((f32x4*)y)[i] = v;
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    sum += (fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3]));
    cnt += 4.f;
  }
  block_amax_commit(m, out);
  if ((blockIdx.x & 15) == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if ((threadIdx.x & 63) == 0 && cnt > 0.f) {
      const int k = (blockIdx.x >> 4) & 15;
      atomicAdd((float*)out + 1 + k, sum);
      atomicAdd((float*)out + 17 + k, cnt);
    }
  }
}

__global__ __launch_bounds__(256) void split_planes_f16_kernel(const float* __restrict__ x, unsigned short* __restrict__ pl,
                                                               const long plane_stride, const long n8, const float s_host,
                                                               const float* __restrict__ amax, float* __restrict__ s_out,
                                                               unsigned* __restrict__ amax_next, float* __restrict__ zero_slot) {
  const float s = amax ? f16_scale_of(*amax) : s_host;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (s_out) *s_out = s;
    if (zero_slot) *zero_slot = 0.f;   // the accumulator of the NEXT call in this role (nobody touches it during this one)
  }
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const f32x4 v0 = ((const f32x4*)x)[2 * i], v1 = ((const f32x4*)x)[2 * i + 1];
    if (amax_next) {
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))));
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
    }
    uint2 o0[2], o1[2];
    split4h(v0, s, o0);
    split4h(v1, s, o1);
#pragma unroll
    for (int q = 0; q < 2; q++)
      ((uint4*)(pl + q * plane_stride))[i] = uint4{o0[q].x, o0[q].y, o1[q].x, o1[q].y};
  }
  if (amax_next) block_amax_commit(m, amax_next);   // the maximum of THIS tensor, for the scale of the next one in this role
}

struct PackDesc { long src_off, dst_off; int Cout, K, unit0, pad; };

__device__ __forceinline__ void pack_unit(const float* __restrict__ w, unsigned short* __restrict__ dst, long plane_stride,
                                          int Cout, int K, int unit, int lane) {
  const int nb32 = (Cout + 31) >> 5;
  const int step = unit / nb32, blk = unit - step * nb32;
  const int r = lane >> 1, h = lane & 1;
  const int n = blk * 32 + r;
  const int lh = h ^ ((r >> 3) & 1);
  f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
  if (n < Cout) {
    const float* src = w + (long)n * K + step * 16 + lh * 8;
    v0 = ldg4(src);
    v1 = ldg4(src + 4);
  }
  uint2 o0[3], o1[3];
  split4<3>(v0, o0);
  split4<3>(v1, o1);
#pragma unroll
  for (int q = 0; q < 3; q++)
    *(uint4*)(dst + q * plane_stride + (long)unit * 512 + lane * 8) = uint4{o0[q].x, o0[q].y, o1[q].x, o1[q].y};
}

__global__ __launch_bounds__(256) void pack_one_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst,
                                                       long plane_stride, int Cout, int K, int n_units) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit < n_units) pack_unit(w, dst, plane_stride, Cout, K, unit, threadIdx.x & 63);
}

// the same tiling with the two fp16 terms of w * s (experiment, see split4h)
__global__ __launch_bounds__(256) void pack_one_f16_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst,
                                                           long plane_stride, int Cout, int K, int n_units, float s_host,
                                                           const float* __restrict__ amax, float* __restrict__ s_out) {
  const float s = amax ? f16_scale_of(*amax) : s_host;
  if (s_out && blockIdx.x == 0 && threadIdx.x == 0) *s_out = s;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;
  const int lane = threadIdx.x & 63;
  const int nb32 = (Cout + 31) >> 5;
  const int step = unit / nb32, blk = unit - step * nb32;
  const int r = lane >> 1, h = lane & 1;
  const int n = blk * 32 + r;
  const int lh = h ^ ((r >> 3) & 1);
  f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
  if (n < Cout) {
    const float* src = w + (long)n * K + step * 16 + lh * 8;
    v0 = ldg4(src);
    v1 = ldg4(src + 4);
  }
  uint2 o0[2], o1[2];
  split4h(v0, s, o0);
  split4h(v1, s, o1);
#pragma unroll
  for (int q = 0; q < 2; q++)
    *(uint4*)(dst + q * plane_stride + (long)unit * 512 + lane * 8) = uint4{o0[q].x, o0[q].y, o1[q].x, o1[q].y};
}

// packed planes of the DATA-GRADIENT weights straight from w: the matrix wd[ci][(KH-1-kh, KW-1-kw, co)] = w[co][kh][kw][ci]
// * scale[co] (what weight_flip_kernel materialises in fp32) is never written; needs Cout % 16 == 0
__device__ __forceinline__ void pack_flip_unit(const float* __restrict__ w, const float* __restrict__ scale,
                                               unsigned short* __restrict__ dst, long plane_stride, int Cout, int KH, int KW,
                                               int Cin, int unit, int lane) {
  const int nb32 = (Cin + 31) >> 5;
  const int step = unit / nb32, blk = unit - step * nb32;
  const int r = lane >> 1, h = lane & 1;
  const int ci = blk * 32 + r;
  const int lh = h ^ ((r >> 3) & 1);
  const int k0 = step * 16 + lh * 8;          // first of this lane's 8 k' = (flipped tap, co)
  const int ft = k0 / Cout, co0 = k0 - ft * Cout;
  const int tap = KH * KW - 1 - ft;           // un-flipped tap index kh*KW + kw
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int co = co0 + j;
    v[j] = ci < Cin ? w[((long)co * KH * KW + tap) * Cin + ci] * (scale ? scale[co] : 1.f) : 0.f;
  }
  uint2 o0[3], o1[3];
  split4<3>(f32x4{v[0], v[1], v[2], v[3]}, o0);
  split4<3>(f32x4{v[4], v[5], v[6], v[7]}, o1);
#pragma unroll
  for (int q = 0; q < 3; q++)
    *(uint4*)(dst + q * plane_stride + (long)unit * 512 + lane * 8) = uint4{o0[q].x, o0[q].y, o1[q].x, o1[q].y};
}

// data-gradient weights (see pack_flip_unit) as the two fp16 terms of wd * s, s from the device-side maximum (experiment)
__global__ __launch_bounds__(256) void pack_flip_f16_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                            unsigned short* __restrict__ dst, long plane_stride, int Cout,
                                                            int KH, int KW, int Cin, int n_units,
                                                            const float* __restrict__ amax, float* __restrict__ s_out) {
  const float s = f16_scale_of(*amax);
  if (s_out && blockIdx.x == 0 && threadIdx.x == 0) *s_out = s;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;
  const int lane = threadIdx.x & 63;
  const int nb32 = (Cin + 31) >> 5;
  const int step = unit / nb32, blk = unit - step * nb32;
  const int r = lane >> 1, h = lane & 1;
  const int ci = blk * 32 + r;
  const int lh = h ^ ((r >> 3) & 1);
  const int k0 = step * 16 + lh * 8;
  const int ft = k0 / Cout, co0 = k0 - ft * Cout;
  const int tap = KH * KW - 1 - ft;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int co = co0 + j;
    v[j] = ci < Cin ? w[((long)co * KH * KW + tap) * Cin + ci] * (scale ? scale[co] : 1.f) : 0.f;
  }
  uint2 o0[2], o1[2];
  split4h(f32x4{v[0], v[1], v[2], v[3]}, s, o0);
  split4h(f32x4{v[4], v[5], v[6], v[7]}, s, o1);
#pragma unroll
  for (int q = 0; q < 2; q++)
    *(uint4*)(dst + q * plane_stride + (long)unit * 512 + lane * 8) = uint4{o0[q].x, o0[q].y, o1[q].x, o1[q].y};
}

__global__ __launch_bounds__(256) void pack_flip_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                        unsigned short* __restrict__ dst, long plane_stride, int Cout,
                                                        int KH, int KW, int Cin, int n_units) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (unit >= n_units) return;
  pack_flip_unit(w, scale, dst, plane_stride, Cout, KH, KW, Cin, unit, lane);
}

// all data-gradient weight planes of a model in ONE launch (once per optimiser step): descriptor table on the device
struct FlipDesc { const float* w; const float* scale; unsigned short* dst; long plane_stride; int Cout, KH, KW, Cin, unit0, pad; };
__global__ __launch_bounds__(256) void pack_flip_many_kernel(const FlipDesc* __restrict__ descs, const int* __restrict__ unit_desc,
                                                             int n_units) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (unit >= n_units) return;
  const FlipDesc d = descs[unit_desc[unit]];
  pack_flip_unit(d.w, d.scale, d.dst, d.plane_stride, d.Cout, d.KH, d.KW, d.Cin, unit - d.unit0, lane);
}

__global__ __launch_bounds__(256) void pack_many_kernel(const float* __restrict__ base, unsigned short* __restrict__ dst,
                                                        long plane_stride, const PackDesc* __restrict__ descs,
                                                        const int* __restrict__ unit_desc, int n_units) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;
  const PackDesc d = descs[unit_desc[unit]];
  pack_unit(base + d.src_off, dst + d.dst_off, plane_stride, d.Cout, d.K, unit - d.unit0, threadIdx.x & 63);
}

// ---- fp16 two-term planes (the default arithmetic of mode 3) of EVERY weight matrix of a model, once per optimiser / EMA
// step like pack_many_kernel: launch 1 reduces max |w| per matrix into stat[2 d] (one conditional atomic per 1 KiB unit),
// launch 2 derives the matrix's power-of-two scale from it, leaves it in stat[2 d + 1] and writes the planes of w * scale.
__device__ __forceinline__ void load_pack_unit(const float* __restrict__ w, int Cout, int K, int unit, int lane, f32x4& v0, f32x4& v1) {
  const int nb32 = (Cout + 31) >> 5;
  const int step = unit / nb32, blk = unit - step * nb32;
  const int r = lane >> 1, h = lane & 1;
  const int n = blk * 32 + r;
  const int lh = h ^ ((r >> 3) & 1);
  v0 = f32x4{0.f, 0.f, 0.f, 0.f};
  v1 = v0;
  if (n < Cout) {
    const float* src = w + (long)n * K + step * 16 + lh * 8;
    v0 = ldg4(src);
    v1 = ldg4(src + 4);
  }
}

__device__ __forceinline__ void load_flip_unit(const float* __restrict__ w, const float* __restrict__ scale, int Cout, int KH, int KW,
                                               int Cin, int unit, int lane, f32x4& v0, f32x4& v1) {
  const int nb32 = (Cin + 31) >> 5;
  const int step = unit / nb32, blk = unit - step * nb32;
  const int r = lane >> 1, h = lane & 1;
  const int ci = blk * 32 + r;
  const int lh = h ^ ((r >> 3) & 1);
  const int k0 = step * 16 + lh * 8;
  const int ft = k0 / Cout, co0 = k0 - ft * Cout;
  const int tap = KH * KW - 1 - ft;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int co = co0 + j;
    v[j] = ci < Cin ? w[((long)co * KH * KW + tap) * Cin + ci] * (scale ? scale[co] : 1.f) : 0.f;
  }
  v0 = f32x4{v[0], v[1], v[2], v[3]};
  v1 = f32x4{v[4], v[5], v[6], v[7]};
}

__device__ __forceinline__ void store_unit_f16(const f32x4 v0, const f32x4 v1, const float s, unsigned short* __restrict__ dst,
                                               long plane_stride, int unit, int lane) {
  uint2 o0[2], o1[2];
  split4h(v0, s, o0);
  split4h(v1, s, o1);
#pragma unroll
  for (int q = 0; q < 2; q++)
    *(uint4*)(dst + q * plane_stride + (long)unit * 512 + lane * 8) = uint4{o0[q].x, o0[q].y, o1[q].x, o1[q].y};
}

// wave-level running maximum over a contiguous run of units: committed (one conditional atomic) when the run moves on to the
// next matrix and at its end -- one atomic per 1 KiB unit made this launch 5x slower than the packing launch itself
__device__ __forceinline__ void wave_amax_commit(float m, unsigned* __restrict__ out) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const unsigned bits = __builtin_bit_cast(unsigned, m);
  if ((threadIdx.x & 63) == 0 && m > 0.f && bits > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, bits);
}
__device__ __forceinline__ float amax8(const f32x4 v0, const f32x4 v1) {
  return fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
               fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
}
constexpr int PACK_RUN = 32;   // units per wave in the reduction launches

__global__ __launch_bounds__(256) void pack_many_amax_kernel(const float* __restrict__ base, const PackDesc* __restrict__ descs,
                                                             const int* __restrict__ unit_desc, int n_units, float* __restrict__ stat) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int u0 = wave * PACK_RUN, u1 = min(u0 + PACK_RUN, n_units);
  int cur = -1;
  float m = 0.f;
  PackDesc d{};
  for (int unit = u0; unit < u1; unit++) {
    const int di = unit_desc[unit];
    if (di != cur) {
      if (cur >= 0) wave_amax_commit(m, (unsigned*)stat + 2 * cur);
      cur = di; m = 0.f; d = descs[di];
    }
    // the maximum does not care about the order: unit u of a matrix reads the u-th 2 KiB of the matrix as it lies in memory (one
    // coalesced run per wave) instead of the 32-byte pieces of 32 rows the packing launch needs (64 us -> 40 for 44 M parameters);
    // a matrix has at least as many units as it has 512-element runs (its packed image is padded to 32 rows)
    const long e = (long)(unit - d.unit0) * 512 + lane * 8, numel = (long)d.Cout * d.K;
    if (e < numel) {   // (K % 16 == 0: the 8 elements of a lane lie inside the matrix)
      const float* src = base + d.src_off + e;
      m = fmaxf(m, amax8(ldg4(src), ldg4(src + 4)));
    }
  }
  if (cur >= 0) wave_amax_commit(m, (unsigned*)stat + 2 * cur);
}

__global__ __launch_bounds__(256) void pack_many_f16_kernel(const float* __restrict__ base, unsigned short* __restrict__ dst,
                                                            long plane_stride, const PackDesc* __restrict__ descs,
                                                            const int* __restrict__ unit_desc, int n_units, float* __restrict__ stat) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (unit >= n_units) return;
  const int di = unit_desc[unit];
  const PackDesc d = descs[di];
  f32x4 v0, v1;
  load_pack_unit(base + d.src_off, d.Cout, d.K, unit - d.unit0, lane, v0, v1);
  const float s = f16_scale_of(stat[2 * di]);
  if (unit == d.unit0 && lane == 0) stat[2 * di + 1] = s;
  store_unit_f16(v0, v1, s, dst + d.dst_off, plane_stride, unit - d.unit0, lane);
}

__global__ __launch_bounds__(256) void pack_flip_many_amax_kernel(const FlipDesc* __restrict__ descs, const int* __restrict__ unit_desc,
                                                                  int n_units, float* __restrict__ stat) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int u0 = wave * PACK_RUN, u1 = min(u0 + PACK_RUN, n_units);
  int cur = -1;
  float m = 0.f;
  FlipDesc d{};
  for (int unit = u0; unit < u1; unit++) {
    const int di = unit_desc[unit];
    if (di != cur) {
      if (cur >= 0) wave_amax_commit(m, (unsigned*)stat + 2 * cur);
      cur = di; m = 0.f; d = descs[di];
    }
    // (as above: max |w[co][tap][ci] * scale[co]| over the matrix in memory order -- the flipped / transposed gather of the packing
    // launch costs 8 strided scalar loads per lane, 148 us per step for a number that does not depend on the order)
    const int rowlen = d.KH * d.KW * d.Cin;
    const long e = (long)(unit - d.unit0) * 512 + lane * 8, numel = (long)d.Cout * rowlen;
    if (e < numel) {
      if ((rowlen & 7) == 0) {   // the 8 elements of a lane share their output channel
        const float sc = d.scale ? d.scale[e / rowlen] : 1.f;
        const f32x4 a = ldg4(d.w + e), b = ldg4(d.w + e + 4);
        m = fmaxf(m, amax8(f32x4{a[0] * sc, a[1] * sc, a[2] * sc, a[3] * sc}, f32x4{b[0] * sc, b[1] * sc, b[2] * sc, b[3] * sc}));
      } else {
        for (int j = 0; j < 8 && e + j < numel; j++)
          m = fmaxf(m, fabsf(d.w[e + j] * (d.scale ? d.scale[(e + j) / rowlen] : 1.f)));
      }
    }
  }
  if (cur >= 0) wave_amax_commit(m, (unsigned*)stat + 2 * cur);
}

__global__ __launch_bounds__(256) void pack_flip_many_f16_kernel(const FlipDesc* __restrict__ descs, const int* __restrict__ unit_desc,
                                                                 int n_units, float* __restrict__ stat) {
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (unit >= n_units) return;
  const int di = unit_desc[unit];
  const FlipDesc d = descs[di];
  f32x4 v0, v1;
  load_flip_unit(d.w, d.scale, d.Cout, d.KH, d.KW, d.Cin, unit - d.unit0, lane, v0, v1);
  const float s = f16_scale_of(stat[2 * di]);
  if (unit == d.unit0 && lane == 0) stat[2 * di + 1] = s;
  store_unit_f16(v0, v1, s, (unsigned short*)d.dst, d.plane_stride, unit - d.unit0, lane);
}

__global__ void weight_flip_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                   float* __restrict__ wd, int Cout, int KH, int KW, int Cin) {
  // wd[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci] * scale[co]; 32x32 LDS transpose over (co, ci)
  __shared__ float tile[32][33];
  const int tap = blockIdx.z, kh = tap / KW, kw = tap % KW;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int co = co0 + i, ci = ci0 + tx;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[(((long)co * KH + kh) * KW + kw) * Cin + ci] * (scale ? scale[co] : 1.f);
    tile[i][tx] = v;
  }
  __syncthreads();
  const int ftap = (KH - 1 - kh) * KW + (KW - 1 - kw);
  for (int i = ty; i < 32; i += 8) {
    const int ci = ci0 + i, co = co0 + tx;
    if (co < Cout && ci < Cin) wd[((long)ci * KH * KW + ftap) * Cout + co] = tile[tx][i];
  }
}

template <bool HALF>  // HALF: x and y are bf16 tensors (max of bf16 values is a bf16 value: exact)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                      int H, int W, int C, int Ho, int Wo) {
  // 3 x 3 / stride 2 / pad 1.  Taps outside the image are CLAMPED to the nearest inside one instead of skipped: the maximum does not
  // change (a clamped tap repeats a value of the window) and the nine loads of an output are unconditional, all in flight at once
  // (round 5: the skipping form -- a branch around each load -- waited for every load in turn: 230 us for the teacher's 8 x 64 x 512^2
  // stem output, 1.7 x its HBM time)
  const long total = (long)N * Ho * Wo * (C / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % (C / 4));
    long r = i / (C / 4);
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    f32x4 v[9];
#pragma unroll
    for (int dh = 0; dh < 3; dh++) {
      const int ih = min(max(ho * 2 - 1 + dh, 0), H - 1);
#pragma unroll
      for (int dwi = 0; dwi < 3; dwi++) {
        const int iw = min(max(wo * 2 - 1 + dwi, 0), W - 1);
        const long xi = (((long)n * H + ih) * W + iw) * C + c4 * 4;
        if (HALF) {
          const uint2 t = *(const uint2*)((const unsigned short*)x + xi);
          v[dh * 3 + dwi] = f32x4{__builtin_bit_cast(float, t.x << 16), __builtin_bit_cast(float, t.x & 0xffff0000u),
                                  __builtin_bit_cast(float, t.y << 16), __builtin_bit_cast(float, t.y & 0xffff0000u)};
        } else {
          v[dh * 3 + dwi] = ldg4(x + xi);
        }
      }
    }
    f32x4 m = v[0];
#pragma unroll
    for (int t = 1; t < 9; t++)
#pragma unroll
      for (int e = 0; e < 4; e++) m[e] = fmaxf(m[e], v[t][e]);
    const long yi = (((long)n * Ho + ho) * Wo + wo) * C + c4 * 4;
    if (HALF) *(uint2*)((unsigned short*)y + yi) = uint2{pk_bf16(m[0], m[1]), pk_bf16(m[2], m[3])};
    else *(f32x4*)(y + yi) = m;
  }
}

}  // namespace

extern "C" long mmt_packed_weight_elems(int Cout, int K) {
  if (Cout <= 0 || K <= 0 || (K & 15)) return -1;
  return (long)(K / 16) * ((Cout + 31) / 32) * 512;
}

extern "C" int mmt_pack_weight(const float* w, void* planes, long plane_stride, int Cout, int K, void* stream) {
  const long n = mmt_packed_weight_elems(Cout, K);
  if (!w || !planes || n < 0 || plane_stride < n || (plane_stride & 7) || ((size_t)planes & 15)) return MMT_EINVAL;
  const int units = (int)(n / 512);
  hipLaunchKernelGGL(pack_one_kernel, dim3((units + 3) / 4), dim3(256), 0, (hipStream_t)stream, w,
                     (unsigned short*)planes, plane_stride, Cout, K, units);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_pack_weight_flipped(const float* w, const float* scale, void* planes, long plane_stride, int Cout,
                                       int KH, int KW, int Cin, void* stream) {
  const long n = mmt_packed_weight_elems(Cin, KH * KW * Cout);  // rows = Cin, K' = KH*KW*Cout
  if (!w || !planes || n < 0 || (Cout & 15) || plane_stride < n || (plane_stride & 7) || ((size_t)planes & 15)) return MMT_EINVAL;
  const int units = (int)(n / 512);
  hipLaunchKernelGGL(pack_flip_kernel, dim3((units + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, scale,
                     (unsigned short*)planes, plane_stride, Cout, KH, KW, Cin, units);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_pack_weights_flipped(const mmt_flip_desc* descs, const int* unit_desc, int n_units, void* stream) {
  if (!descs || !unit_desc) return MMT_EINVAL;
  if (n_units <= 0) return 0;
  static_assert(sizeof(mmt_flip_desc) == sizeof(FlipDesc), "descriptor layout");
  hipLaunchKernelGGL(pack_flip_many_kernel, dim3((n_units + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const FlipDesc*)descs,
                     unit_desc, n_units);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_pack_weights(const float* base, void* planes, long plane_stride, const mmt_pack_desc* descs,
                                const int* unit_desc, int n_units, void* stream) {
  if (!base || !planes || !descs || !unit_desc || (plane_stride & 7) || ((size_t)planes & 15)) return MMT_EINVAL;
  if (n_units <= 0) return 0;
  static_assert(sizeof(mmt_pack_desc) == sizeof(PackDesc), "descriptor layout");
  hipLaunchKernelGGL(pack_many_kernel, dim3((n_units + 3) / 4), dim3(256), 0, (hipStream_t)stream, base,
                     (unsigned short*)planes, plane_stride, (const PackDesc*)descs, unit_desc, n_units);
  MMT_LAUNCH_CHECK();
  return 0;
}

// the two fp16 planes of w * s_d for every matrix d of the table (see pack_many_f16_kernel); stat[2 d] <- max |w_d|,
// stat[2 d + 1] <- s_d (what mmt_conv_forward_f16x2 / mmt_conv3x3_strip_f16x2 take as s_w)
extern "C" int mmt_pack_weights_f16(const float* base, void* planes, long plane_stride, const mmt_pack_desc* descs,
                                    const int* unit_desc, int n_units, int n_descs, float* stat, void* stream) {
  if (!base || !planes || !descs || !unit_desc || !stat || (plane_stride & 7) || ((size_t)planes & 15)) return MMT_EINVAL;
  if (n_units <= 0 || n_descs <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(stat, 0, (size_t)n_descs * 2 * sizeof(float), s) != hipSuccess) return MMT_EINVAL;
  hipLaunchKernelGGL(pack_many_amax_kernel, dim3(mmt_cdiv(n_units, 4 * PACK_RUN)), dim3(256), 0, s, base, (const PackDesc*)descs, unit_desc,
                     n_units, stat);
  hipLaunchKernelGGL(pack_many_f16_kernel, dim3((n_units + 3) / 4), dim3(256), 0, s, base, (unsigned short*)planes, plane_stride,
                     (const PackDesc*)descs, unit_desc, n_units, stat);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_pack_weights_flipped_f16(const mmt_flip_desc* descs, const int* unit_desc, int n_units, int n_descs, float* stat,
                                            void* stream) {
  if (!descs || !unit_desc || !stat) return MMT_EINVAL;
  if (n_units <= 0 || n_descs <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(stat, 0, (size_t)n_descs * 2 * sizeof(float), s) != hipSuccess) return MMT_EINVAL;
  hipLaunchKernelGGL(pack_flip_many_amax_kernel, dim3(mmt_cdiv(n_units, 4 * PACK_RUN)), dim3(256), 0, s, (const FlipDesc*)descs, unit_desc,
                     n_units, stat);
  hipLaunchKernelGGL(pack_flip_many_f16_kernel, dim3((n_units + 3) / 4), dim3(256), 0, s, (const FlipDesc*)descs, unit_desc,
                     n_units, stat);
  MMT_LAUNCH_CHECK();
  return 0;
}

// amax[0] = max(amax[0], max |x * rowscale|) (amax zeroed by the caller); rowscale indexes rows of `inner` elements
extern "C" int mmt_amax(const float* x, long n, const float* rowscale, long inner, int rows, float* amax, void* stream) {
  if (!x || !amax || n < 0 || (n & 3) || ((size_t)x & 15) || (rowscale && (inner <= 0 || (inner & 3) || rows <= 0))) return MMT_EINVAL;
  if (n == 0) return 0;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(amax_kernel<false>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, n / 4, rowscale, rowscale ? inner / 4 : 1,
                     rowscale ? rows : 1, (unsigned*)amax);
  MMT_LAUNCH_CHECK();
  return 0;
}

// the same reduction into a 33-float statistics slot (zeroed by the caller): slot[0] = max |x|, slot[1..32] = partial sums of |x|
extern "C" int mmt_amax_stats(const float* x, long n, float* slot, void* stream) {
  if (!x || !slot || n < 0 || (n & 3) || ((size_t)x & 15)) return MMT_EINVAL;
  if (n == 0) return 0;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(amax_kernel<true>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, n / 4, (const float*)nullptr, 1L, 1,
                     (unsigned*)slot);
  MMT_LAUNCH_CHECK();
  return 0;
}

// out <- the statistics of a tensor whose every element is a CONVEX combination of elements of the tensors behind `slots`
// (ROIAlign: bilinear taps of the pyramid levels, averaged): max = the largest of the maxima -- an upper bound, which is all the
// consumer's power-of-two scale needs -- sums and counts added (the mean of such a tensor is about that of its sources)
struct StatSlots { const float* s[8]; };
__global__ void stats_combine_kernel(const StatSlots slots, const int n, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i >= 33) return;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (k < n) v = i == 0 ? fmaxf(v, slots.s[k][0]) : v + slots.s[k][i];
  out[i] = v;
}

// (round 5, ADVICE r4: the slot addresses travel in the kernel arguments -- the per-call device table of round 4 was a pageable
// host-to-device copy on the ROIAlign path of both launch threads)
extern "C" int mmt_stats_combine(const float* const* slots, int n, float* out, void* stream) {
  if (!slots || !out || n <= 0 || n > 8) return MMT_EINVAL;
  StatSlots t{};
  for (int k = 0; k < n; k++) {
    if (!slots[k]) return MMT_EINVAL;
    t.s[k] = slots[k];
  }
  hipLaunchKernelGGL(stats_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, n, out);
  MMT_LAUNCH_CHECK();
  return 0;
}

// y = a + b (+ c) (+ d) elementwise (n % 4 == 0, 16-byte aligned; y may be one of the inputs) and the statistics of y into `slot`
// (33 floats, zeroed by the caller) as mmt_amax_stats records them
extern "C" int mmt_sum_stats(const float* a, const float* b, const float* c, const float* d, float* y, long n, float* slot,
                             void* stream) {
  if (!a || !b || !y || !slot || n < 0 || (n & 3) || (((size_t)a | (size_t)b | (size_t)c | (size_t)d | (size_t)y) & 15)) return MMT_EINVAL;
  if (!c && d) return MMT_EINVAL;
  if (n == 0) return 0;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sum_stats_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, d, y, n / 4, (unsigned*)slot);
  MMT_LAUNCH_CHECK();
  return 0;
}

// scale: the power of two to use, or -- with `amax` (device) -- derived from it on the device and written to scale_out (device)
extern "C" int mmt_split_planes_f16(const float* x, void* planes, long plane_stride, long n, float scale, const float* amax,
                                    float* scale_out, float* amax_next, float* zero_slot, void* stream) {
  if (!x || !planes || n < 0 || (n & 7) || plane_stride < n || (plane_stride & 7) || ((size_t)planes & 15) || ((size_t)x & 15))
    return MMT_EINVAL;
  if (n == 0) return 0;
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(split_planes_f16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)planes,
                     plane_stride, n / 8, scale, amax, scale_out, (unsigned*)amax_next, zero_slot);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_pack_weight_f16(const float* w, void* planes, long plane_stride, int Cout, int K, float scale, const float* amax,
                                   float* scale_out, void* stream) {
  const long n = mmt_packed_weight_elems(Cout, K);
  if (!w || !planes || n < 0 || plane_stride < n || (plane_stride & 7) || ((size_t)planes & 15)) return MMT_EINVAL;
  const int units = (int)(n / 512);
  hipLaunchKernelGGL(pack_one_f16_kernel, dim3((units + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)planes,
                     plane_stride, Cout, K, units, scale, amax, scale_out);
  MMT_LAUNCH_CHECK();
  return 0;
}

// data-gradient weights of conv(x, w) (* scale[co]), as mmt_pack_weight_flipped; amax = device max of |w scale| (mmt_amax)
extern "C" int mmt_pack_weight_flipped_f16(const float* w, const float* scale, void* planes, long plane_stride, int Cout, int KH,
                                           int KW, int Cin, const float* amax, float* scale_out, void* stream) {
  const long n = mmt_packed_weight_elems(Cin, KH * KW * Cout);
  if (!w || !planes || !amax || n < 0 || (Cout & 15) || plane_stride < n || (plane_stride & 7) || ((size_t)planes & 15)) return MMT_EINVAL;
  const int units = (int)(n / 512);
  hipLaunchKernelGGL(pack_flip_f16_kernel, dim3((units + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, scale,
                     (unsigned short*)planes, plane_stride, Cout, KH, KW, Cin, units, amax, scale_out);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_split_planes(const float* x, void* planes, long plane_stride, long n, void* stream) {
  if (!x || !planes || n < 0 || (n & 7) || plane_stride < n || (plane_stride & 7) || ((size_t)planes & 15) || ((size_t)x & 15))
    return MMT_EINVAL;
  if (n == 0) return 0;
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_planes_kernel<3>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)planes,
                     plane_stride, n / 8);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_weight_flip_transpose(const float* w, const float* scale, float* wd, int Cout, int KH, int KW,
                                         int Cin, void* stream) {
  if (!w || !wd) return MMT_EINVAL;
  hipLaunchKernelGGL(weight_flip_kernel, dim3(mmt_cdiv(Cin, 32), mmt_cdiv(Cout, 32), KH * KW), dim3(256), 0,
                     (hipStream_t)stream, w, scale, wd, Cout, KH, KW, Cin);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  if (C & 3) return MMT_EINVAL;
  const long total = (long)N * Ho * Wo * (C / 4);
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(maxpool_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_maxpool3x3s2_bf16(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  if (C & 3) return MMT_EINVAL;
  const long total = (long)N * Ho * Wo * (C / 4);
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(maxpool_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, N, H, W,
                     C, Ho, Wo);
  MMT_LAUNCH_CHECK();
  return 0;
}
