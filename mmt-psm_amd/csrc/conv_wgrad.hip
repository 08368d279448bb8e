// The weight gradients of the convolutions (the dW / dbias half of the backward pass the
// reference gets from ATen's conv2d / linear backward): the fp32 MFMA kernel, the register-splitting pipelined kernel of the
// default arithmetic, their split-K reduction, and the batch entry point that packs a backward pass's jobs into grouped launches.
// (The plane-fed kernel and the grouped launches' plumbing are conv_wgpl.hip.)
#include <stdlib.h>
#include <type_traits>
#include "conv_shared.h"

namespace {

// ------------------------------------------------------------------------------------ weight gradient
// dW[co][tap][ci] += rowscale[co] * sum_m dy[m][co] * xg[m][tap][ci]
// GEMM: M' = Cout, N' = KH*KW*Cin, K' = m.  Both operands are MN-contiguous in memory, so their LDS
// images are [k][128] and fragments are single ds_read_b32 (lanes along the 128 => conflict-free).
// K' is split across blockIdx.z; partial tiles are combined with fp32 atomics straight into the
// caller's gradient buffer (which also sums the contributions of every use of a shared weight).
template <bool FAST>  // FAST: Cout % 4 == 0 and Ho, Wo >= 8 -> straight-line vector loads, carry-select row decode
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const ConvP p, const float* __restrict__ dy,
                                                         const float* __restrict__ rowscale,
                                                         float* __restrict__ dw, int m_per_split,
                                                         float* __restrict__ ws) {
  constexpr int BM = 128, BN = 128;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsA = lds;               // [2][32][128]  dy
  float* ldsB = lds + 2 * 32 * 128;  // [2][32][128]  gathered x
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int co0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int NP = p.KH * p.KW * p.Cin;
  const int ms = blockIdx.z * m_per_split;
  const int me = min(p.M, ms + m_per_split);
  if (ms >= me) return;

  const int c4 = tid & 31;   // float4 column within the 128-wide row
  const int kr = tid >> 5;   // row 0..7 (+8*j)
  // A (dy) column validity
  const int aco = co0 + c4 * 4;
  const bool avec = FAST || (p.Cout & 3) == 0;
  // B column -> (tap, ci) fixed for the whole K' loop
  const int ncol = n0 + c4 * 4;
  const bool bcol_ok = ncol < NP;
  int bkh = 0, bkw = 0, bci = 0;
  if (bcol_ok) { const int tap = ncol / p.Cin; bci = ncol - tap * p.Cin; bkh = tap / p.KW; bkw = tap - bkh * p.KW; }
  const int HoWo = p.Ho * p.Wo;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  f32x4 ra[4], rb[4];
  bool pa[4], pb[4];
  // (img, ho, wo) of this thread's 4 rows, advanced by 32 rows per tile with carries instead of divisions
  int r_img[4], r_ho[4], r_wo[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int m = ms + kr + 8 * j;
    r_img[j] = m / HoWo;
    const int rem = m - r_img[j] * HoWo;
    r_ho[j] = rem / p.Wo;
    r_wo[j] = rem - r_ho[j] * p.Wo;
  }
  const bool inc_ok = FAST;
  const int step_q = 32 / p.Wo, step_r = 32 - step_q * p.Wo;  // +32 rows = +step_q image rows, +step_r columns
  auto load_tile = [&](int mt) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int m = mt + kr + 8 * j;
      const bool mok = m < me;
      if (FAST || avec) {  // uniform; the loads are branch-free (invalid lanes read element 0, select 0)
        const bool ok = mok && aco < p.Cout;
        ra[j] = ldg4(dy + (ok ? (unsigned)m * (unsigned)p.Cout + (unsigned)aco : 0u));
        pa[j] = ok;
      } else {
        f32x4 v = zero4;
        if (mok) {
          const float* src = dy + (long)m * p.Cout + aco;
#pragma unroll
          for (int e = 0; e < 4; e++) if (aco + e < p.Cout) v[e] = src[e];
        }
        ra[j] = v;
        pa[j] = true;
      }
      int img, ho, wo;
      if (inc_ok) { img = r_img[j]; ho = r_ho[j]; wo = r_wo[j]; }
      else { const int mm = mok ? m : 0; img = mm / HoWo; const int rem = mm - img * HoWo; ho = rem / p.Wo; wo = rem - ho * p.Wo; }
      const int ih = ho * p.stride - p.pad + bkh, iw = wo * p.stride - p.pad + bkw;
      const bool okb = mok && bcol_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      rb[j] = ldg4(p.x + (okb ? (unsigned)(((img * p.H + ih) * p.W + iw) * p.Cin + bci) : 0u));
      pb[j] = okb;
      if (inc_ok) {  // carries as selects (step_q + 1 <= 5 < Ho): no divergent loops, no divisions
        int wo2 = r_wo[j] + step_r, ho2 = r_ho[j] + step_q;
        const bool cw = wo2 >= p.Wo;
        wo2 = cw ? wo2 - p.Wo : wo2;
        ho2 = cw ? ho2 + 1 : ho2;
        const bool ch = ho2 >= p.Ho;
        r_wo[j] = wo2;
        r_ho[j] = ch ? ho2 - p.Ho : ho2;
        r_img[j] = ch ? r_img[j] + 1 : r_img[j];
      }
    }
  };
  auto store_tile = [&](int buf) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      *(f32x4*)(ldsA + buf * 4096 + (kr + 8 * j) * 128 + c4 * 4) = pa[j] ? ra[j] : zero4;
      *(f32x4*)(ldsB + buf * 4096 + (kr + 8 * j) * 128 + c4 * 4) = pb[j] ? rb[j] : zero4;
    }
  };
  const int lr = lane & 31, kh2 = lane >> 5;
  const int ntile = (me - ms + 31) / 32;
  load_tile(ms);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntile; t++) {
    const int buf = t & 1;
    if (t + 1 < ntile) load_tile(ms + (t + 1) * 32);
    const float* A = ldsA + buf * 4096 + wm * 64 + lr;
    const float* B = ldsB + buf * 4096 + wn * 64 + lr;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int k = 8 * jj + 4 * kh2 + q;
        const float a0 = A[k * 128], a1 = A[k * 128 + 32];
        const float b0 = B[k * 128], b1 = B[k * 128 + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    if (t + 1 < ntile) store_tile(buf ^ 1);
    __syncthreads();
  }
  // ---- epilogue.  fp32 atomics from every split-K block cost ~20 us per block (measured: mid-size layers ran at
  // 62 TFLOP/s with them, 100 without), so the partial tile goes through LDS and is written with plain, fully
  // coalesced float4 stores: to workspace slab `blockIdx.z` when K' is split (wgrad_reduce_kernel then sums the slabs
  // into dw), or read-modify-written into dw directly when there is a single split.
  {
    float* ct = lds;  // [128][128]
    const int rq = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
          ct[row * 128 + (wn * 2 + b) * 32 + lr] = acc[a][b][r];
        }
    __syncthreads();
    const int cc = tid & 31, r0 = tid >> 5;
    const int n = n0 + cc * 4;
    if (n < NP) {
      const bool direct = ws == nullptr;
      float* dst = direct ? dw : ws + (long)blockIdx.z * p.Cout * NP;
      for (int row = r0; row < 128; row += 8) {
        const int co = co0 + row;
        if (co >= p.Cout) break;
        f32x4 v = *(const f32x4*)(ct + row * 128 + cc * 4);
        float* q = dst + (long)co * NP + n;
        if (direct) {
          const float sc = rowscale ? rowscale[co] : 1.f;
          const f32x4 o = *(const f32x4*)q;
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = o[e] + v[e] * sc;
        }
        *(f32x4*)q = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ split-bf16 weight gradient
// Same GEMM as conv_wgrad_kernel (M' = Cout, N' = KH*KW*Cin, K' = pixels) on the bf16 matrix pipe with the fp32
// operands split into NS bf16 terms while they are staged.  The reduction index (pixels) is the STRIDED index of both
// operands in memory (dy[m][co], x[m][ci]) while the MFMA wants 8 consecutive k per lane, so the staging transposes:
// a thread owns a 4-pixel x 4-channel block (4 float4 loads, lanes along channels -> 256 B contiguous per pixel), splits
// it, and writes per channel one 8-byte group of 4 consecutive pixels.  LDS image per operand and plane:
//   [k/8][row][8 k] bf16 with row = (c % 4) * 32 + c / 4  for channel c of the 128-wide tile
// (a thread's four channels land in four different 32-row MFMA tiles; consecutive lanes write consecutive rows: 2-way
// bank conflict on the ds_write_b64 instead of 8-way for row = c).  MFMA tile T, row i therefore is channel 4 i + T;
// the epilogue undoes the permutation when it lays the accumulators out in LDS.  Fragment reads are ds_read_b128 at
// (k/8, row), conflict-free without a swizzle.  Tile 128 x 128 x 16 pixels, 2-deep ring, waves 2 x 2 (each 64 x 64):
// waves 0,1 stage dy, waves 2,3 stage the gathered input.
template <int NS, bool INC>  // INC: Ho, Wo >= 8 -> carry-select pixel decode (else divisions)
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_kernel(const ConvP p, const float* __restrict__ dy,
                                                                  const float* __restrict__ rowscale,
                                                                  float* __restrict__ dw, int m_per_split,
                                                                  float* __restrict__ ws, float* __restrict__ dbias) {
  constexpr int PL = 2 * 128 * 16;       // bytes per plane per operand: [2][128][16 B]
  constexpr int STAGE = 2 * NS * PL;     // A planes | B planes
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ring = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int co0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int NP = p.KH * p.KW * p.Cin;
  const int ms = blockIdx.z * m_per_split;
  const int me = min(p.M, ms + m_per_split);
  if (ms >= me) return;
  const int HoWo = p.Ho * p.Wo;

  // ---- staging role of this thread
  const bool roleB = wave >= 2;                       // wave-uniform
  const int cq = (wave & 1) * 16 + (lane & 15);       // channel quad 0..31 of the 128-wide tile
  const int pg = lane >> 4;                           // pixel group 0..3 (4 pixels each) of the 16-pixel step
  const int ch = (roleB ? n0 : co0) + cq * 4;         // first of this thread's 4 channels / columns
  bool col_ok;
  int bkh = 0, bkw = 0, bci = 0;
  if (roleB) {
    col_ok = ch < NP;
    if (col_ok) { const int tap = ch / p.Cin; bci = ch - tap * p.Cin; bkh = tap / p.KW; bkw = tap - bkh * p.KW; }
  } else {
    col_ok = ch < p.Cout;
  }
  // LDS byte offset (inside a plane) of channel e = 0: k-group pg>>1, row e*32 + cq, 8-byte half pg&1
  const int woff = (pg >> 1) * 2048 + cq * 16 + (pg & 1) * 8 + (roleB ? NS * PL : 0);
  int r_img[4], r_ho[4], r_wo[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int m = ms + pg * 4 + j;
    r_img[j] = m / HoWo;
    const int rem = m - r_img[j] * HoWo;
    r_ho[j] = rem / p.Wo;
    r_wo[j] = rem - r_ho[j] * p.Wo;
  }
  const int step_q = 16 / p.Wo, step_r = 16 - step_q * p.Wo;
  f32x4 rg[4];
  bool pr[4];
  // bias gradient rides along: the dy values pass through the registers of waves 0,1 anyway (first column tile only)
  const bool do_bias = dbias != nullptr && blockIdx.x == 0 && !roleB;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int m = mt + pg * 4 + j;
      const bool mok = m < me && col_ok;
      if (!roleB) {
        rg[j] = ldg4(dy + (mok ? (unsigned)m * (unsigned)p.Cout + (unsigned)ch : 0u));
        pr[j] = mok;
      } else {
        int img, ho, wo;
        if (INC) { img = r_img[j]; ho = r_ho[j]; wo = r_wo[j]; }
        else { const int mm = m < me ? m : 0; img = mm / HoWo; const int rem = mm - img * HoWo; ho = rem / p.Wo; wo = rem - ho * p.Wo; }
        const int ih = ho * p.stride - p.pad + bkh, iw = wo * p.stride - p.pad + bkw;
        const bool ok = mok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        rg[j] = ldg4(p.x + (ok ? (unsigned)(((img * p.H + ih) * p.W + iw) * p.Cin + bci) : 0u));
        pr[j] = ok;
        if (INC) {
          int wo2 = r_wo[j] + step_r, ho2 = r_ho[j] + step_q;
          const bool cw = wo2 >= p.Wo;
          wo2 = cw ? wo2 - p.Wo : wo2;
          ho2 = cw ? ho2 + 1 : ho2;
          const bool chh = ho2 >= p.Ho;
          r_wo[j] = wo2;
          r_ho[j] = chh ? ho2 - p.Ho : ho2;
          r_img[j] = chh ? r_img[j] + 1 : r_img[j];
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* base = ring + buf * STAGE + woff;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = pr[j] ? rg[j] : zero4;
    if (do_bias) bsum += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int e = 0; e < 4; e++) {  // channel e of the block: its 4 pixels are 4 consecutive k
      uint2 o[NS];
      split4<NS>(f32x4{v[0][e], v[1][e], v[2][e], v[3][e]}, o);
#pragma unroll
      for (int q = 0; q < NS; q++) *(uint2*)(base + q * PL + e * 512) = o[q];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const int lr = lane & 31, kh2 = lane >> 5;
  const int froff = kh2 * 2048 + lr * 16;
  const int ntile = (me - ms + 15) / 16;
  load_tile(ms);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntile; t++) {
    const int buf = t & 1;
    const bool more = t + 1 < ntile;
    const char* A = ring + buf * STAGE + froff + (wm * 2) * 512;
    const char* B = ring + buf * STAGE + NS * PL + froff + (wn * 2) * 512;
    bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
      for (int a = 0; a < 2; a++) {
        fa[q][a] = *(const bf16x8*)(A + q * PL + a * 512);
        fb[q][a] = *(const bf16x8*)(B + q * PL + a * 512);
      }
    int g = 0;
#pragma unroll
    for (int sum = NS - 1; sum >= 0; sum--)
#pragma unroll
      for (int qa = 0; qa <= sum; qa++) {
        const int qb = sum - qa;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
        if (g == 0) {  // the global loads of the next step go out behind the first MFMA group
          __builtin_amdgcn_sched_barrier(0);
          if (more) load_tile(ms + (t + 1) * 16);
          __builtin_amdgcn_sched_barrier(0);
        }
        g++;
      }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (do_bias) {  // lanes cq + 16 * pg hold partial sums of the same 4 channels: fold the 4 pixel groups, one atomic each
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float t = bsum[e];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (pg == 0 && ch + e < p.Cout) atomicAdd(dbias + ch + e, t);
    }
  }
  // ---- epilogue: as conv_wgrad_kernel, with the (tile, row) -> channel permutation undone while writing to LDS
  {
    float* ct = lds;  // [128][128]
    const int rq = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * rq;
          ct[(4 * i + wm * 2 + a) * 128 + 4 * lr + wn * 2 + b] = acc[a][b][r];
        }
    __syncthreads();
    const int cc = tid & 31, r0 = tid >> 5;
    const int n = n0 + cc * 4;
    if (n < NP) {
      const bool direct = ws == nullptr;
      float* dst = direct ? dw : ws + (long)blockIdx.z * p.Cout * NP;
      for (int row = r0; row < 128; row += 8) {
        const int co = co0 + row;
        if (co >= p.Cout) break;
        f32x4 v = *(const f32x4*)(ct + row * 128 + cc * 4);
        float* q = dst + (long)co * NP + n;
        if (direct) {
          const float sc = rowscale ? rowscale[co] : 1.f;
          const f32x4 o = *(const f32x4*)q;
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = o[e] + v[e] * sc;
        }
        *(f32x4*)q = v;
      }
    }
  }
}

// the weight gradient's slow, exact path (see f16_guard_bad): the block's 128 x 128 tile of dW over its pixel range [ms, me)
// with fp32 FMAs straight from global memory, times s (= s_x s_dy: the stores divide it out again), into ct; bias sums by atomics
__device__ __forceinline__ void wgrad_slow_fill(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ ct, const int co0,
                                             const int n0, const int ms, const int me, const float s, float* __restrict__ dbias,
                                             const int tid) {
  const ConvPK pk = kernarg_convp();   // fields are fetched where they are used (scalar loads from the argument segment)
  const int NP = pk->KH * pk->KW * pk->Cin, HoWo = pk->Ho * pk->Wo;
  for (int o = tid; o < 128 * 128; o += 256) {
    const int row = o >> 7, col = o & 127;
    const int co = co0 + row, n = n0 + col;
    float acc = 0.f;
    if (co < pk->Cout && n < NP) {
      const int tap = n / pk->Cin, ci = n - tap * pk->Cin, kh = tap / pk->KW, kw = tap - kh * pk->KW;
      for (int m = ms; m < me; m++) {
        const int img = m / HoWo, rem = m - img * HoWo;
        const int ho = rem / pk->Wo, wo = rem - ho * pk->Wo;
        const int ih = ho * pk->stride - pk->pad + kh, iw = wo * pk->stride - pk->pad + kw;
        if ((unsigned)ih < (unsigned)pk->H && (unsigned)iw < (unsigned)pk->W)
          acc = fmaf(dy[(long)m * pk->Cout + co], x[((long)(img * pk->H + ih) * pk->W + iw) * pk->Cin + ci], acc);
      }
    }
    ct[o] = acc * s;
  }
  if (dbias && tid < 128 && co0 + tid < pk->Cout) {
    float b = 0.f;
    for (int m = ms; m < me; m++) b += dy[(long)m * pk->Cout + co0 + tid];
    atomicAdd(dbias + co0 + tid, b);
  }
}

// Same tile, same LDS image and same arithmetic as conv_wgrad_split_kernel, software-pipelined one step deeper: the
// global loads of pixel step t+2 are issued behind the first MFMAs of step t, and the split + LDS stores of step t+1
// (whose loads went out a whole step earlier) are cut into micro-ops that sit behind the individual MFMAs of step t.
// In the kernel above the three parts of a step -- MFMAs, split arithmetic, load latency -- simply add up (0.47 + 0.31
// + 0.33 ms on the 3x3 256-channel FPN shape); here the matrix pipe covers the other two.  Branch-free: steps past
// the end load nothing (predicated to offset 0) and store zeros into a buffer nobody reads.
// MODE = pixel decode: 0 divisions, 1 carry-select per pixel (Ho, Wo >= 8), 2 per thread (+ Wo % 4 == 0);
// VEC4 = Cout % 4 == 0 (16-byte dy loads)
// BF (mode 1, bf16 storage): bit 0 = x is a bf16 tensor, bit 1 = dy is; the loads fetch 8 bytes per 4 channels and widen
// them (exact), everything after the load is unchanged -- the one-term "split" of a bf16 value is the value itself
// F16 (opt-in fp16 two-term split, NS == 2): both operands are scaled by the power of two of their recorded maximum
// (p.f16_sx -> max |x|, p.f16_sw -> max |dy|, device scalars), split into two fp16 terms, multiplied with 3 f16 MFMAs, and the
// tile is divided by s_x s_dy where it is stored (directly, or in wgrad_reduce_kernel for the split form)
// (tx, ty, tz, lin: the grid of this launch and the block's linear index in it -- or, in a grouped launch, of its ITEM)
template <int NS, int MODE, bool VEC4, int BF = 0, bool F16 = false>
__device__ __forceinline__ void conv_wgrad_pipe_body(const ConvP& p_in, const float* __restrict__ dy_in,
                                                     const float* __restrict__ rowscale,
                                                     float* __restrict__ dw, int m_per_split,
                                                     float* __restrict__ ws, float* __restrict__ dbias,
                                                     const int tx, const int ty, const int tz, const int lin) {
  constexpr int PL = 2 * 128 * 16;
  constexpr int STAGE = 2 * NS * PL;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ring = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware block order.  The hardware deals workgroups to the 8 XCDs round-robin in launch order (x fastest); the
  // tx * ty blocks of one pixel range (same z) read the same slices of x and dy, so they are made neighbours on ONE XCD
  // (consecutive remapped ids) and the slices cross the fabric once instead of once per XCD
  int bx, by, bz;
  {
    const int nwg = tx * ty * tz;
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7, idx = lin >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = id % tx;
    const int t = id / tx;
    by = t % ty;
    bz = t / ty;
  }
  // two-segment form (p_in.seg_z > 0): slices z >= seg_z belong to the second (x, dy) pair -- same shapes, its own scales
  ConvP p = p_in;
  const float* __restrict__ dy = dy_in;
  int bzl = bz;
  if constexpr (F16) {
    if (p_in.seg_z > 0 && bz >= p_in.seg_z) {
      p.x = p_in.x2; dy = p_in.dy2; p.f16_sx = p_in.f16_sx2; p.f16_sw = p_in.f16_sw2;
      p.guard_x = p_in.guard_x2; p.guard_dy = p_in.guard_dy2;
      bzl = bz - p_in.seg_z;
    }
  }
  const int co0 = by * 128, n0 = bx * 128;
  const int NP = p.KH * p.KW * p.Cin;
  const int ms = bzl * m_per_split;
  const int me = min(p.M, ms + m_per_split);
  if (ms >= me) return;
  const int HoWo = p.Ho * p.Wo;
  F16Guard guard_x = {}, guard_dy = {};
  if constexpr (F16) { guard_x = f16_guard_load(p.guard_x); guard_dy = f16_guard_load(p.guard_dy); }   // tested before the pipeline starts

  const bool roleB = wave >= 2;
  const int cq = (wave & 1) * 16 + (lane & 15);
  const int pg = lane >> 4;
  const int ch = (roleB ? n0 : co0) + cq * 4;
  bool col_ok;
  int bkh = 0, bkw = 0, bci = 0;
  if (roleB) {
    col_ok = ch < NP;
    if (col_ok) { const int tap = ch / p.Cin; bci = ch - tap * p.Cin; bkh = tap / p.KW; bkw = tap - bkh * p.KW; }
  } else {
    col_ok = ch < p.Cout;
  }
  const int woff = (pg >> 1) * 2048 + cq * 16 + (pg & 1) * 8 + (roleB ? NS * PL : 0);
  int r_img[4], r_ho[4], r_wo[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int m = ms + pg * 4 + j;
    r_img[j] = m / HoWo;
    const int rem = m - r_img[j] * HoWo;
    r_ho[j] = rem / p.Wo;
    r_wo[j] = rem - r_ho[j] * p.Wo;
  }
  const int step_q = 16 / p.Wo, step_r = 16 - step_q * p.Wo;
  const bool do_bias = dbias != nullptr && bx == 0 && !roleB;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  const int lr = lane & 31, kh2 = lane >> 5;
  const int froff = kh2 * 2048 + lr * 16;
  const int ntile = (me - ms + 15) / 16;
  float f16_s[2] = {1.f, 1.f};   // F16: scale of dy (role A) / of x (role B)
  if constexpr (F16) { f16_s[0] = f16_scale_of(*p.f16_sw); f16_s[1] = f16_scale_of(*p.f16_sx); }

  // the whole pipeline once per staging role (wave-uniform), so that each copy is straight-line code
  auto run = [&](auto role_tag) {
    constexpr bool RB = decltype(role_tag)::value;
    const float f16_role = f16_s[RB ? 1 : 0];
    int m_load = ms + pg * 4;  // first pixel of this thread's next load (advances 16 per step)
    // Loads are raw buffer loads: 32-bit byte offset against a scalar descriptor (no 64-bit address arithmetic), and
    // lanes outside the tensor / in the halo get offset 2^31 >= num_records, for which the hardware returns zeros.
    constexpr unsigned OOB = 0x80000000u;
    constexpr bool HALF = RB ? (BF & 1) != 0 : (BF & 2) != 0;  // this role's tensor is stored as bf16
    static_assert(!HALF || VEC4, "bf16 storage: 4-channel loads");
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RB ? p.x : dy), 0, (int)((RB ? (long)p.N * p.H * p.W * p.Cin : (long)p.M * p.Cout) * (HALF ? 2 : 4)), 0x00020000);
    auto bload = [&](unsigned voff) {   // voff: byte offset in the fp32 tensor (OOB = 2^31: beyond either size)
      if constexpr (HALF) {
        const uint2 t = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff >> 1), 0, 0));
        return f32x4{__builtin_bit_cast(float, t.x << 16), __builtin_bit_cast(float, t.x & 0xffff0000u),
                     __builtin_bit_cast(float, t.y << 16), __builtin_bit_cast(float, t.y & 0xffff0000u)};
      } else {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0));
      }
    };
    // MODE 2 (Wo % 4 == 0): the thread's four pixels share an output row, so one (ih, iw, offset) triple is carried
    // per thread, with the tap folded into the wrap limits; stepping 16 pixels wraps at most once in each direction
    int t_iw = 0, t_ih = 0, t_off = 0;
    const int lim_w = p.Wo * p.stride - p.pad + bkw, lim_h = p.Ho * p.stride - p.pad + bkh;
    const int a_w = step_r * p.stride, a_h = step_q * p.stride, WoS = p.Wo * p.stride, HoS = p.Ho * p.stride;
    const int d_step = (a_h * p.W + a_w) * p.Cin * 4, d_cw = (p.stride * p.W - WoS) * p.Cin * 4;
    const int d_ch = (p.H - HoS) * p.W * p.Cin * 4, d_px = p.stride * p.Cin * 4;
    if (RB && MODE == 2) {
      t_iw = r_wo[0] * p.stride - p.pad + bkw;
      t_ih = r_ho[0] * p.stride - p.pad + bkh;
      t_off = (((r_img[0] * p.H + t_ih) * p.W + t_iw) * p.Cin + bci) * 4;
    }
    unsigned a_off = ((unsigned)m_load * (unsigned)p.Cout + (unsigned)ch) * 4u;  // dy role: byte offset of pixel 0
    auto load_px = [&](int j, f32x4 (&rg)[4]) {
      if (!RB) {
        const bool mok = m_load + j < me && col_ok;
        const unsigned off = a_off + (unsigned)(j * p.Cout * 4);
        if (VEC4) {
          rg[j] = bload(mok ? off : OOB);
        } else {  // Cout % 4 != 0 (the 15-channel predictors): rows are not 16-byte aligned and may end inside the quad
#pragma unroll
          for (int e = 0; e < 4; e++)
            rg[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rsrc, (int)(mok && ch + e < p.Cout ? off + 4u * e : OOB), 0, 0));
        }
        if (j == 3) { a_off += 16u * (unsigned)p.Cout * 4u; m_load += 16; }
        return;
      }
      if (MODE == 2) {
        const int iw = t_iw + j * p.stride;
        const bool ok = m_load < me && col_ok && (unsigned)t_ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        rg[j] = bload(ok ? (unsigned)(t_off + j * d_px) : OOB);
        if (j == 3) {
          t_iw += a_w;
          const bool cw = t_iw >= lim_w;
          t_iw = cw ? t_iw - WoS : t_iw;
          t_ih += a_h;
          t_ih = cw ? t_ih + p.stride : t_ih;
          const bool chh = t_ih >= lim_h;
          t_ih = chh ? t_ih - HoS : t_ih;
          t_off += d_step;
          t_off = cw ? t_off + d_cw : t_off;
          t_off = chh ? t_off + d_ch : t_off;
          m_load += 16;
        }
        return;
      }
      const int m = m_load + j;
      const bool mok = m < me && col_ok;
      int img, ho, wo;
      if (MODE == 1) { img = r_img[j]; ho = r_ho[j]; wo = r_wo[j]; }
      else { const int mm = m < me ? m : 0; img = mm / HoWo; const int rem = mm - img * HoWo; ho = rem / p.Wo; wo = rem - ho * p.Wo; }
      const int ih = ho * p.stride - p.pad + bkh, iw = wo * p.stride - p.pad + bkw;
      const bool ok = mok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      rg[j] = bload(ok ? (unsigned)(((img * p.H + ih) * p.W + iw) * p.Cin + bci) * 4u : OOB);
      if (MODE == 1) {
        int wo2 = r_wo[j] + step_r, ho2 = r_ho[j] + step_q;
        const bool cw = wo2 >= p.Wo;
        wo2 = cw ? wo2 - p.Wo : wo2;
        ho2 = cw ? ho2 + 1 : ho2;
        const bool chh = ho2 >= p.Ho;
        r_wo[j] = wo2;
        r_ho[j] = chh ? ho2 - p.Ho : ho2;
        r_img[j] = chh ? r_img[j] + 1 : r_img[j];
      }
      if (j == 3) m_load += 16;
    };
    // split state of the tile being stored: residuals per channel e (4 pixels each), packed pairs of the level
    float sv[4][4];
    unsigned su[4][2];
    auto split_begin = [&](const f32x4 (&rg)[4]) {
      if (!RB && do_bias) bsum += (rg[0] + rg[1]) + (rg[2] + rg[3]);
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int j = 0; j < 4; j++) sv[e][j] = rg[j][e];   // (F16: the scale rides in the conversions below)
    };
    auto split_cvt = [&](int e, int q, char* base) {
      if constexpr (F16) {
        // h = rn16(s x), l = rn16(s x - h) with the mixed-precision FMA: the scale (a power of two: s x is exact) and the
        // conversion in ONE instruction per value and term, results packed in place -- 32 vector instructions per 16
        // values where multiply / convert / pack / widen / subtract / convert / pack was 56 (the kernel's time is the SUM
        // of its vector-ALU and matrix time: the two hardly co-issue on this part)
        if (q == 0) {
          asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(su[e][0]) : "v"(f16_role), "v"(sv[e][0]));
          asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(su[e][0]) : "v"(f16_role), "v"(sv[e][1]));
          asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(su[e][1]) : "v"(f16_role), "v"(sv[e][2]));
          asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(su[e][1]) : "v"(f16_role), "v"(sv[e][3]));
        } else {
          unsigned l0, l1;
          asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(f16_role), "v"(sv[e][0]), "v"(su[e][0]));
          asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(f16_role), "v"(sv[e][1]), "v"(su[e][0]));
          asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(f16_role), "v"(sv[e][2]), "v"(su[e][1]));
          asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(f16_role), "v"(sv[e][3]), "v"(su[e][1]));
          su[e][0] = l0; su[e][1] = l1;
        }
      } else {
        su[e][0] = pk_bf16(sv[e][0], sv[e][1]);
        su[e][1] = pk_bf16(sv[e][2], sv[e][3]);
      }
      *(uint2*)(base + q * PL + e * 512) = uint2{su[e][0], su[e][1]};
    };
    // plain v_sub_f32: the compiler would pair these into v_pk_add_f32, which is the slower choice beside MFMAs
    auto fsub = [](float a, unsigned b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto split_sub = [&](int e) {
      if constexpr (F16) {
        (void)e;   // the subtraction is inside the second conversion
      } else {
        sv[e][0] = fsub(sv[e][0], su[e][0] << 16);
        sv[e][1] = fsub(sv[e][1], su[e][0] & 0xffff0000u);
        sv[e][2] = fsub(sv[e][2], su[e][1] << 16);
        sv[e][3] = fsub(sv[e][3], su[e][1] & 0xffff0000u);
      }
    };
    constexpr int SPL = 2 * NS - 1;          // cvt, (sub, cvt) x (NS-1) per channel
    constexpr int NMICRO = 4 + 1 + 4 * SPL;  // split_begin, 4 pixel loads, 4 channels x SPL pieces
    auto micro = [&](int idx, char* base, f32x4 (&rg_g)[4], const f32x4 (&rg_w)[4]) {
      if (idx == 0) { split_begin(rg_w); return; }
      idx -= 1;
      if (idx < 4) { load_px(idx, rg_g); return; }
      idx -= 4;
      // level-major over the four channels: consecutive pieces are independent
      const int lvl = idx >> 2, e = idx & 3;
      if (lvl & 1) split_sub(e); else split_cvt(e, lvl >> 1, base);
    };
    // one step: MFMAs on the fragments already in registers (fa, fb: tile t); behind them the fragment reads of tile t+1
    // (complete in LDS stage s_next since the barrier that ended the previous step) into (fan, fbn), the split + LDS stores of
    // tile t+2 (from rg_w, into stage s_store, whose last reader finished before that same barrier) and the global loads of
    // tile t+4 (into rg_g).  Three LDS stages: no wave waits for a fragment at the top of a step.
    constexpr int NFR = 4 * NS;
    auto step = [&](int s_next, int s_store, const bf16x8 (&fa)[NS][2], const bf16x8 (&fb)[NS][2], bf16x8 (&fan)[NS][2],
                    bf16x8 (&fbn)[NS][2], f32x4 (&rg_g)[4], const f32x4 (&rg_w)[4]) {
      const char* A = ring + s_next * STAGE + froff + (wm * 2) * 512;
      const char* B = ring + s_next * STAGE + NS * PL + froff + (wn * 2) * 512;
      char* base = ring + s_store * STAGE + woff;
      auto fread = [&](int r) {   // in the order the MFMAs want them (smallest terms first: fa[0], fb[NS-1] lead)
        const int q = r >> 2, a = (r >> 1) & 1;
        if (r & 1) fbn[NS - 1 - q][a] = *(const bf16x8*)(B + (NS - 1 - q) * PL + a * 512);
        else fan[q][a] = *(const bf16x8*)(A + q * PL + a * 512);
      };
      constexpr int NM = 4 * (NS * (NS + 1) / 2);
      constexpr int NMI = NMICRO + NFR;
      int j = 0, mi = 0;
#pragma unroll
      for (int sum = NS - 1; sum >= 0; sum--)
#pragma unroll
        for (int qa = 0; qa <= sum; qa++) {
          const int qb = sum - qa;
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
              if constexpr (F16)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[qa][a]), __builtin_bit_cast(f16x8, fb[qb][b]), acc[a][b], 0, 0, 0);
              else
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
              j++;
#pragma unroll
              for (int r = 0; r < (NMI + NM - 1) / NM; r++)
                if (mi < (j * NMI + NM - 1) / NM) {
                  // the staging pieces and the fragment reads alternate (a read is one instruction)
                  const int k = mi;
                  const int nf = k < 2 * NFR ? (k + 1) / 2 : NFR;          // fragment reads among the first k pieces
                  if (k < 2 * NFR && (k & 1) == 0) fread(k >> 1); else micro(k - nf, base, rg_g, rg_w);
                  mi++;
                }
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      __syncthreads();
    };

    // three register sets: the loads of pixel step t + 4 go out during step t (two steps ahead of the split that consumes
    // them: one step -- about a microsecond -- did not cover the L2 / fabric latency under load)
    f32x4 rgP[4], rgQ[4], rgR[4];
    bf16x8 fa0[NS][2], fb0[NS][2], fa1[NS][2], fb1[NS][2];
    {  // prologue: tiles 0, 1 -> LDS stages 0, 1; tiles 2, 3 -> registers; fragments of tile 0
#pragma unroll
      for (int j = 0; j < 4; j++) load_px(j, rgP);
#pragma unroll
      for (int j = 0; j < 4; j++) load_px(j, rgQ);
#pragma unroll
      for (int t = 0; t < 2; t++) {
        char* base = ring + t * STAGE + woff;
        split_begin(t ? rgQ : rgP);
#pragma unroll
        for (int lvl = 0; lvl < SPL; lvl++)
#pragma unroll
          for (int e = 0; e < 4; e++) { if (lvl & 1) split_sub(e); else split_cvt(e, lvl >> 1, base); }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) load_px(j, rgP);
#pragma unroll
      for (int j = 0; j < 4; j++) load_px(j, rgQ);
      __syncthreads();
      const char* A = ring + froff + (wm * 2) * 512;
      const char* B = ring + NS * PL + froff + (wn * 2) * 512;
#pragma unroll
      for (int q = 0; q < NS; q++)
#pragma unroll
        for (int a = 0; a < 2; a++) {
          fa0[q][a] = *(const bf16x8*)(A + q * PL + a * 512);
          fb0[q][a] = *(const bf16x8*)(B + q * PL + a * 512);
        }
    }
    for (int t = 0; t < ntile; t += 6) {
      step(1, 2, fa0, fb0, fa1, fb1, rgR, rgP);
      if (t + 1 < ntile) step(2, 0, fa1, fb1, fa0, fb0, rgP, rgQ);
      if (t + 2 < ntile) step(0, 1, fa0, fb0, fa1, fb1, rgQ, rgR);
      if (t + 3 < ntile) step(1, 2, fa1, fb1, fa0, fb0, rgR, rgP);
      if (t + 4 < ntile) step(2, 0, fa0, fb0, fa1, fb1, rgP, rgQ);
      if (t + 5 < ntile) step(0, 1, fa1, fb1, fa0, fb0, rgQ, rgR);
    }
  };
  bool slow = false;   // fp16 split: an operand whose dynamic range defeats fp16 (f16_guard_bad) -> exact fp32 products for this tile
  if constexpr (F16) slow = f16_guard_bad(guard_x) || f16_guard_bad(guard_dy);
  if (slow) {
    wgrad_slow_fill(p.x, dy, lds, co0, n0, ms, me, f16_s[0] * f16_s[1], dbias != nullptr && bx == 0 ? dbias : nullptr, tid);
    __syncthreads();
  } else {
  if (roleB) run(std::true_type{}); else run(std::false_type{});
  if (do_bias) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float t = bsum[e];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (pg == 0 && ch + e < p.Cout) atomicAdd(dbias + ch + e, t);
    }
  }
  }
  {
    float* ct = lds;  // [128][128]
    const int rq = lane >> 5;
    if (!slow) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * rq;
          ct[(4 * i + wm * 2 + a) * 128 + 4 * lr + wn * 2 + b] = acc[a][b][r];
        }
    __syncthreads();
    }
    const int cc = tid & 31, r0 = tid >> 5;
    const int n = n0 + cc * 4;
    if (n < NP) {
      const bool direct = ws == nullptr;
      float* dst = direct ? dw : ws + (long)bz * p.Cout * NP;
      for (int row = r0; row < 128; row += 8) {
        const int co = co0 + row;
        if (co >= p.Cout) break;
        f32x4 v = *(const f32x4*)(ct + row * 128 + cc * 4);
        float* q = dst + (long)co * NP + n;
        if (direct) {
          const float sc = (rowscale ? rowscale[co] : 1.f) * (F16 ? 1.f / (f16_s[0] * f16_s[1]) : 1.f);
          const f32x4 o = *(const f32x4*)q;
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = o[e] + v[e] * sc;
        } else if constexpr (F16) {   // the slab carries the true partial sum: each segment has its own power-of-two scales
          const float inv = 1.f / (f16_s[0] * f16_s[1]);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] *= inv;
        }
        *(f32x4*)q = v;
      }
    }
  }
}

template <int NS, int MODE, bool VEC4, int BF = 0, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_pipe_kernel(const ConvP p_in, const float* __restrict__ dy_in,
                                                                 const float* __restrict__ rowscale,
                                                                 float* __restrict__ dw, int m_per_split,
                                                                 float* __restrict__ ws, float* __restrict__ dbias) {
  conv_wgrad_pipe_body<NS, MODE, VEC4, BF, F16>(p_in, dy_in, rowscale, dw, m_per_split, ws, dbias, (int)gridDim.x, (int)gridDim.y,
                                                (int)gridDim.z, (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)));
}

// round 6: the register-splitting weight gradient of SEVERAL layers in one launch (the 1x1 layers, whose operands
// have no planes: 16 tiles each at N = 2, cut into 16 pixel ranges of 16 steps when launched alone).  See wgrad_pl_group_kernel
// (conv_wgpl.hip) for the idea; items' block ranges start at multiples of 8.
constexpr int WGP_MAXG = 6;
struct WgPipeItem { ConvP p; const float* dy; const float* rowscale; float* dw; float* ws; float* dbias; int mps, tx, ty, tz; };
struct WgPipeGroup { WgPipeItem it[WGP_MAXG]; int first[WGP_MAXG + 1]; int n; };
static_assert(sizeof(WgPipeGroup) <= 3840, "kernel-argument segment");
template <int MODE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_pipe_group_kernel(const WgPipeGroup g) {
  int i = 0;
  for (int k = 1; k < g.n; k++) i = (int)blockIdx.x >= g.first[k] ? k : i;
  const int local = (int)blockIdx.x - g.first[i];
  const int tx = g.it[i].tx, ty = g.it[i].ty, tz = g.it[i].tz;
  if (local >= tx * ty * tz) return;   // (padding up to the next multiple of 8)
  conv_wgrad_pipe_body<2, MODE, true, 0, true>(g.it[i].p, g.it[i].dy, g.it[i].rowscale, g.it[i].dw, g.it[i].mps, g.it[i].ws,
                                                g.it[i].dbias, tx, ty, tz, local);
}

// dw[co][n] += rowscale[co] * sum_s ws[s][co][n]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int Cout, int NP,
                                                           const float* __restrict__ rowscale,
                                                           float* __restrict__ dw, const float* __restrict__ f16_ax = nullptr,
                                                           const float* __restrict__ f16_ady = nullptr) {
  const float f16_inv = f16_ax ? 1.f / (f16_scale_of(*f16_ax) * f16_scale_of(*f16_ady)) : 1.f;   // fp16 split: the slabs hold scaled sums
  const long n4 = (long)Cout * NP / 4;
  const long slab = (long)Cout * NP;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    // slabs added in order; eight loads in flight (the plain loop is a chain of load latencies: 13 us per launch)
    f32x4 o = ((f32x4*)dw)[i];
    f32x4 a = ((const f32x4*)ws)[i];
    int s = 1;
    for (; s + 8 <= splits; s += 8) {
      f32x4 b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) b[u] = *(const f32x4*)(ws + (s + u) * slab + i * 4);
#pragma unroll
      for (int u = 0; u < 8; u++) a += b[u];
    }
    for (; s < splits; s++) a += *(const f32x4*)(ws + s * slab + i * 4);
    const float sc = (rowscale ? rowscale[(int)((i * 4) / NP)] : 1.f) * f16_inv;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] += a[e] * sc;
    ((f32x4*)dw)[i] = o;
  }
}

// dbias[c] += scale * sum_m dy[m][c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, int M, int C,
                                                     float* __restrict__ out, int rows_per_block) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < C)
    for (int r = r0 + sub; r < r1; r += 4) s += dy[(long)r * C + c];
  __shared__ float red[4][64];
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && c < C) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" int mmt_conv_wgrad_splits(const mmt_conv_args* a) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  if (p.M == 0 || p.Cout == 0) return 1;
  const int NP = p.KH * p.KW * p.Cin;
  const int tx = mmt_cdiv(NP, 128), ty = mmt_cdiv(p.Cout, 128);
  // all blocks of a launch run equally long: fill the 512 resident slots (256 CUs x 2 blocks) ONCE.  (640 = 1.25
  // rounds cost a second, 20 %-full round: 92 -> 105 TFLOP/s fp32, 103 -> 136 split-bf16 on the FPN 3x3 shapes)
  const long tiles = (long)tx * ty;
  static const int slots = getenv("MMT_WG_SLOTS") ? atoi(getenv("MMT_WG_SLOTS")) : 512;   // (tuned: profiles/r04_dispatch_sweep.txt; in the step: r04 / r06 history; the switch is for that sweep)
  constexpr int min_px = 512;   // (tuned: profiles/r04_dispatch_sweep.txt; in the step: profiles/r04_history.md)
  int split = (int)(tiles >= slots ? 1 : slots / tiles);
  if (a->x2) {   // two segments of p.M pixels each: the same number of blocks reduces twice the pixels; an even number of slices
    const int max2 = mmt_cdiv(2 * p.M, min_px);
    if (split > max2) split = max2;
    int half = split / 2;
    if (half < 1) half = 1;
    int mps2 = mmt_cdiv(p.M, half);
    mps2 = (mps2 + 31) / 32 * 32;
    return 2 * mmt_cdiv(p.M, mps2);
  }
  const int max_split = mmt_cdiv(p.M, min_px);  // at least min_px / 16 k-tiles per block
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  int mps = mmt_cdiv(p.M, split);
  mps = (mps + 31) / 32 * 32;
  return mmt_cdiv(p.M, mps);
}

extern "C" int mmt_conv_wgrad(const mmt_conv_args* a, const float* dy, const float* rowscale, float* dw,
                              float* dbias, float* workspace, void* stream) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  if (!dy || !dw || !p.cin4) return MMT_EINVAL;
  if (p.M == 0 || p.Cout == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int NP = p.KH * p.KW * p.Cin;
  const int tx = mmt_cdiv(NP, 128), ty = mmt_cdiv(p.Cout, 128);
  const int split = mmt_conv_wgrad_splits(a);
  const bool two = a->x2 != nullptr;
  int mps = mmt_cdiv(p.M, two ? split / 2 : split);
  mps = (mps + 31) / 32 * 32;
  if (two && (!a->dy2 || !a->f16_x_amax || !a->f16_dy_amax || !a->f16_x_amax2 || !a->f16_dy_amax2 || split != 2 * mmt_cdiv(p.M, mps)))
    return MMT_EINVAL;   // the two-segment form exists on the fp16 split only
  if (split > 1 && !workspace) return MMT_EINVAL;
  float* ws = split > 1 ? workspace : nullptr;
  const bool fast = (p.Cout & 3) == 0 && p.Wo >= 8 && p.Ho >= 8;
  const int prec = precision();
  // bf16 storage of x (IO_X) / dy (IO_DY): mode 1, the pipelined kernel with 4-channel loads only
  const int bf = ((p.io & IO_X) ? 1 : 0) | ((p.io & IO_DY) ? 2 : 0);
  // opt-in fp16 two-term split (mode 3 only): x_amax / dy_amax = device maxima of the two operands
  const bool f16 = a->f16_x_amax && a->f16_dy_amax;
  if (f16) {
    if (prec != 3 || bf || (p.Cout & 3) || (mps & 15) || !((long)p.N * p.H * p.W * p.Cin * 4 < (1L << 31) && (long)p.M * p.Cout * 4 < (1L << 31)))
      return MMT_EINVAL;
    p.f16_sx = (const float*)a->f16_x_amax;
    p.f16_sw = (const float*)a->f16_dy_amax;
    if (two) {
      p.x2 = (const float*)a->x2; p.dy2 = (const float*)a->dy2;
      p.f16_sx2 = (const float*)a->f16_x_amax2; p.f16_sw2 = (const float*)a->f16_dy_amax2;
      p.guard_x2 = (const float*)a->f16_guard_x2; p.guard_dy2 = (const float*)a->f16_guard_dy2;
      p.seg_z = split / 2;
    }
    const dim3 grid(tx, ty, split);
    const int mode = !(p.Wo >= 8 && p.Ho >= 8) ? 0 : ((p.Wo & 3) == 0 ? 2 : 1);
#define WGF(MODE) hipLaunchKernelGGL((conv_wgrad_pipe_kernel<2, MODE, true, 0, true>), grid, dim3(256), (size_t)65536, s, p, dy, rowscale, dw, mps, ws, dbias)   /* 3 stages of 16 KB <= the 64 KB epilogue tile */
    if (mode == 2) WGF(2); else if (mode == 1) WGF(1); else WGF(0);
#undef WGF
    MMT_LAUNCH_CHECK();
    if (split > 1) {
      const long n4 = (long)p.Cout * NP / 4;
      int blocks = (int)((n4 + 255) / 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, split, p.Cout, NP, rowscale, dw);   // (the slabs are un-scaled)
      MMT_LAUNCH_CHECK();
    }
    return 0;
  }
  if (bf && !(prec == 1 && (p.Cout & 3) == 0 && (mps & 15) == 0 && (long)p.N * p.H * p.W * p.Cin * 4 < (1L << 31) &&
              (long)p.M * p.Cout * 4 < (1L << 31)))
    return MMT_EINVAL;
  constexpr int pipe_any = 1;   // (tuned: profiles/r04_dispatch_sweep.txt)
  const bool small_t = (long)p.N * p.H * p.W * p.Cin * 4 < (1L << 31) && (long)p.M * p.Cout * 4 < (1L << 31);
  if (prec > 0 && ((p.Cout & 3) == 0 || (pipe_any && small_t)) && (mps & 15) == 0) {
    const dim3 grid(tx, ty, split);
    constexpr int pipe = 1;   // (tuned: profiles/r04_dispatch_sweep.txt)
    // the pipelined kernel addresses both operands with 32-bit byte offsets (buffer loads)
    const bool small = (long)p.N * p.H * p.W * p.Cin * 4 < (1L << 31) && (long)p.M * p.Cout * 4 < (1L << 31);
    const int mode = !(p.Wo >= 8 && p.Ho >= 8) ? 0 : ((p.Wo & 3) == 0 ? 2 : 1);
    const bool vec4 = (p.Cout & 3) == 0;
    // LDS: the 64 KB epilogue tile, or the three operand stages of the 3-term split (3 x 24 KB)
    const size_t wg_lds = prec >= 3 ? 73728 : 65536;
#define WGPL(K) do { if (wg_lds > 65536) { const hipError_t er = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wg_lds); if (er != hipSuccess) return (int)er; } hipLaunchKernelGGL(K, grid, dim3(256), wg_lds, s, p, dy, rowscale, dw, mps, ws, dbias); } while (0)
#define WGP(NS, MODE) do { if (bf && NS == 1) { if (bf == 1) WGPL((conv_wgrad_pipe_kernel<1, MODE, true, 1>)); else if (bf == 2) WGPL((conv_wgrad_pipe_kernel<1, MODE, true, 2>)); else WGPL((conv_wgrad_pipe_kernel<1, MODE, true, 3>)); } else if (vec4) WGPL((conv_wgrad_pipe_kernel<NS, MODE, true>)); else WGPL((conv_wgrad_pipe_kernel<NS, MODE, false>)); } while (0)
#define WGP3(NS) do { if (mode == 2) WGP(NS, 2); else if (mode == 1) WGP(NS, 1); else WGP(NS, 0); } while (0)
#define WGS(NS, INC) do { if ((pipe && small) || bf) WGP3(NS); else hipLaunchKernelGGL((conv_wgrad_split_kernel<NS, INC>), grid, dim3(256), (size_t)65536, s, p, dy, rowscale, dw, mps, ws, dbias); } while (0)
    if (fast) { if (prec == 1) WGS(1, true); else if (prec == 2) WGS(2, true); else WGS(3, true); }
    else { if (prec == 1) WGS(1, false); else if (prec == 2) WGS(2, false); else WGS(3, false); }
#undef WGP3
#undef WGP
#undef WGPL
#undef WGS
    dbias = nullptr;  // summed inside the kernel
  } else if (fast)
    hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3(tx, ty, split), dim3(256), (size_t)4 * 4096 * sizeof(float), s, p,
                       dy, rowscale, dw, mps, ws);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3(tx, ty, split), dim3(256), (size_t)4 * 4096 * sizeof(float), s, p,
                       dy, rowscale, dw, mps, ws);
  MMT_LAUNCH_CHECK();
  if (split > 1) {
    const long n4 = (long)p.Cout * NP / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, split, p.Cout, NP, rowscale, dw);
    MMT_LAUNCH_CHECK();
  }
  if (dbias) {
    int rpb = 1024;
    hipLaunchKernelGGL(colsum_kernel, dim3(mmt_cdiv(p.Cout, 64), mmt_cdiv(p.M, rpb)), dim3(256), 0, s, dy, p.M,
                       p.Cout, dbias, rpb);
    MMT_LAUNCH_CHECK();
  }
  return 0;
}

// ---- round 6: the weight gradients of a BATCH of layers -- what a backward pass hands to the side stream at a
// time (layers/fused.py::flush_wgrads) -- as grouped launches: the plane-fed jobs (both operands' row-blocked planes given) in groups of
// <= 12 on wgrad_pl_group_kernel, the fp16-split jobs without planes in groups of <= 6 per pixel-decode mode on
// conv_wgrad_pipe_group_kernel, ONE reduce launch for every slab of the batch; whatever fits neither (and any group of one) goes out
// as the single launch it always was.  Inside a group the tiles of all layers fill the chip together, so a layer needs fewer pixel
// ranges than alone -- a quarter of them by default (see wg_plan): longer reductions per block, a quarter of the slab traffic.
// Summation order: fixed by (the batch's composition, shapes) -- repeatable, but not the single launches' order when ranges differ.
namespace {
// jobs per grouped launch (MMT_WGRAD_GROUP_CAP_PL / _PIPE: the sweep of profiles/r06_history.md; at most what the kernel-argument segment holds)
static int wg_cap(int k) {
  const char* e = getenv(k == 1 ? "MMT_WGRAD_GROUP_CAP_PL" : "MMT_WGRAD_GROUP_CAP_PIPE");
  const int mx = k == 1 ? 12 : WGP_MAXG;
  const int v = e ? atoi(e) : mx;
  return v < 1 ? 1 : (v > mx ? mx : v);
}
struct WgJobPlan { int kind; int split; int mps; long ws_off; };   // kind 0: single launch (mmt_conv_wgrad / _planes); 1: plane-fed group; 2 + mode: pipe group
constexpr int WGJ_MAX = 96;

static bool wg_job_pipe_ok(const mmt_wgrad_job& j, ConvP& p) {
  if (fill(p, &j.a) || !j.dy || !j.dw || !p.cin4 || p.M == 0 || p.Cout == 0) return false;
  if (!j.a.f16_x_amax || !j.a.f16_dy_amax || j.a.x2 || precision() != 3 || p.io || (p.Cout & 3)) return false;
  return (long)p.N * p.H * p.W * p.Cin * 4 < (1L << 31) && (long)p.M * p.Cout * 4 < (1L << 31);
}

// -> workspace floats; plan[i] filled; chunk boundaries are recomputed by the launcher the same way
static long wg_plan(const mmt_wgrad_job* jobs, int n, WgJobPlan* plan) {
  const char* e = getenv("MMT_WGRAD_GROUP");   // read per call (A/B timing, the bit-equality tests of the schedules)
  const bool on = !(e && atoi(e) == 0);
  // Pixel ranges per job inside a group: 1 / WG_DIV of what the job would use ALONE (MMT_WGRAD_GROUP_DIV; 0 = as few as fill the chip
  // as a group).  Measured in the step (profiles/r06_history.md section 7): filling the chip as a group makes blocks that own a CU for
  // ~150 us and hold up the step stream's latency-bound chain (+2 ms); the jobs' own ranges only save launches and reduces (-0.3 ms);
  // a quarter of them is the optimum (-0.5 ... -0.75 ms): blocks four times as long, a quarter of the slab traffic, still short
  const char* dv = getenv("MMT_WGRAD_GROUP_DIV");
  const int sdiv = dv ? atoi(dv) : 4;
  const bool solo = sdiv > 0;
  const int sdiv_pl = getenv("MMT_WGRAD_GROUP_PL") ? atoi(getenv("MMT_WGRAD_GROUP_PL")) : sdiv;       // (sweep: the two kinds apart)
  const int sdiv_pipe = getenv("MMT_WGRAD_GROUP_PIPE") ? atoi(getenv("MMT_WGRAD_GROUP_PIPE")) : sdiv;
  int kind[WGJ_MAX];
  for (int i = 0; i < n; i++) {
    ConvP p;
    kind[i] = 0;
    if (on && jobs[i].x_planes && jobs[i].dy_planes && !jobs[i].a.x2 && wgpl_eligible_splits(&jobs[i].a) > 0) kind[i] = 1;
    else if (on && wg_job_pipe_ok(jobs[i], p)) kind[i] = 2 + (!(p.Wo >= 8 && p.Ho >= 8) ? 0 : ((p.Wo & 3) == 0 ? 2 : 1));
  }
  // a weight shared by several jobs of the batch (the RPN head over the pyramid levels): only its FIRST job may ride in a group -- the
  // items of a group run concurrently and accumulate into dw without atomics; the others follow as single launches, in order
  for (int i = 1; i < n; i++)
    for (int j = 0; j < i; j++)
      if (jobs[j].dw == jobs[i].dw || (jobs[i].dbias && jobs[j].dbias == jobs[i].dbias)) { kind[i] = 0; break; }
  // a kind with a single member is a single launch
  for (int k = 1; k <= 4; k++) {
    int cnt = 0, last = -1;
    for (int i = 0; i < n; i++) if (kind[i] == k) { cnt++; last = i; }
    if (cnt == 1) kind[last] = 0;
  }
  long ws = 0;
  for (int k = 1; k <= 4; k++) {
    const int cap = wg_cap(k), target = k == 1 ? 256 : 512;
    int idx[WGJ_MAX], m = 0;
    for (int i = 0; i < n; i++) if (kind[i] == k) idx[m++] = i;
    for (int c0 = 0; c0 < m; c0 += cap) {
      const int c1 = c0 + cap < m ? c0 + cap : m;   // (the launcher cuts its groups at the same counts)
      long tiles = 0;
      for (int c = c0; c < c1; c++) {
        const mmt_conv_args& a = jobs[idx[c]].a;
        const int NP = a.KH * a.KW * a.Cin;
        tiles += k == 1 ? (long)(a.Cout >> 7) * (NP >> 7) : (long)mmt_cdiv(NP, 128) * mmt_cdiv(a.Cout, 128);
      }
      long f = target / (tiles > 0 ? tiles : 1);
      if (f < 1) f = 1;
      if (solo) f = 1L << 20;
      for (int c = c0; c < c1; c++) {
        const int i = idx[c];
        const mmt_conv_args& a = jobs[i].a;
        const int NP = a.KH * a.KW * a.Cin;
        WgJobPlan& pl = plan[i];
        pl.kind = k; pl.mps = 0;
        if (k == 1) {
          long ks = f, T = wgpl_super_steps(&a);
          if (solo) ks = (wgpl_eligible_splits(&a) + sdiv_pl - 1) / sdiv_pl;
          if (ks > T / 8) ks = T / 8;
          if (ks < 1) ks = 1;
          pl.split = (int)ks;
        } else {
          const int M = a.N * a.Ho * a.Wo;
          long sp = solo ? (mmt_conv_wgrad_splits(&a) + sdiv_pipe - 1) / sdiv_pipe : f;
          const long mx = mmt_cdiv(M, 512);
          if (sp > mx) sp = mx;
          if (sp < 1) sp = 1;
          int mps = mmt_cdiv(M, (int)sp);
          mps = (mps + 31) / 32 * 32;
          pl.mps = mps;
          pl.split = mmt_cdiv(M, mps);
        }
        pl.ws_off = ws;
        if (pl.split > 1) ws += (long)pl.split * a.Cout * NP;
      }
    }
  }
  for (int i = 0; i < n; i++) {
    if (kind[i] != 0) continue;
    WgJobPlan& pl = plan[i];
    const mmt_conv_args& a = jobs[i].a;
    pl.kind = 0; pl.mps = 0;
    int sp = 0;
    if (jobs[i].x_planes && jobs[i].dy_planes && !a.x2) sp = wgpl_eligible_splits(&a);
    if (sp <= 0) sp = mmt_conv_wgrad_splits(&a);
    pl.split = sp < 1 ? 1 : sp;
    pl.ws_off = ws;
    if (pl.split > 1) ws += (long)pl.split * a.Cout * a.KH * a.KW * a.Cin;
  }
  return ws;
}
}  // namespace

extern "C" int mmt_conv_wgrad_group_workspace(const mmt_wgrad_job* jobs, int n, long* floats_out) {
  if (!jobs || !floats_out || n < 0 || n > WGJ_MAX) return MMT_EINVAL;
  WgJobPlan plan[WGJ_MAX];
  *floats_out = wg_plan(jobs, n, plan);
  return 0;
}

extern "C" int mmt_conv_wgrad_group(const mmt_wgrad_job* jobs, int n, float* workspace, long workspace_floats, void* stream) {
  if (!jobs || n < 0 || n > WGJ_MAX) return MMT_EINVAL;
  if (n == 0) return 0;
  WgJobPlan plan[WGJ_MAX];
  const long need = wg_plan(jobs, n, plan);
  if (need > 0 && (!workspace || workspace_floats < need)) return MMT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  WgReduceItem red[WGJ_MAX];
  int nred = 0;
  // plane-fed groups
  {
    WgPlJob g[12];
    int m = 0;
    auto flush = [&]() -> int {
      if (m == 0) return 0;
      const int e = launch_wgpl_group(g, m, s);
      m = 0;
      return e;
    };
    for (int i = 0; i < n; i++) {
      if (plan[i].kind != 1) continue;
      const mmt_wgrad_job& j = jobs[i];
      float* ws = plan[i].split > 1 ? workspace + plan[i].ws_off : nullptr;
      g[m++] = WgPlJob{&j.a, j.dy, j.x_planes, j.x_plane_stride, j.dy_planes, j.dy_plane_stride, j.s_x, j.s_dy, j.rowscale, j.dw, j.dbias, ws,
                       plan[i].split};
      if (plan[i].split > 1)
        red[nred++] = WgReduceItem{ws, j.rowscale, j.dw, plan[i].split, j.a.Cout, j.a.KH * j.a.KW * j.a.Cin, 0};
      if (m == wg_cap(1)) { const int e = flush(); if (e) return e; }
    }
    const int e = flush();
    if (e) return e;
  }
  // register-splitting groups, one pixel-decode mode at a time
  for (int mode = 0; mode < 3; mode++) {
    WgPipeGroup g;
    g.n = 0;
    int nb = 0;
    auto flush = [&]() -> int {
      if (g.n == 0) return 0;
      g.first[g.n] = nb;
      if (mode == 2) hipLaunchKernelGGL(conv_wgrad_pipe_group_kernel<2>, dim3(nb), dim3(256), (size_t)65536, s, g);
      else if (mode == 1) hipLaunchKernelGGL(conv_wgrad_pipe_group_kernel<1>, dim3(nb), dim3(256), (size_t)65536, s, g);
      else hipLaunchKernelGGL(conv_wgrad_pipe_group_kernel<0>, dim3(nb), dim3(256), (size_t)65536, s, g);
      g.n = 0;
      nb = 0;
      MMT_LAUNCH_CHECK();
      return 0;
    };
    for (int i = 0; i < n; i++) {
      if (plan[i].kind != 2 + mode) continue;
      const mmt_wgrad_job& j = jobs[i];
      WgPipeItem& it = g.it[g.n];
      if (!wg_job_pipe_ok(j, it.p) || (plan[i].mps & 15)) return MMT_EINVAL;
      it.p.f16_sx = (const float*)j.a.f16_x_amax;
      it.p.f16_sw = (const float*)j.a.f16_dy_amax;
      const int NP = it.p.KH * it.p.KW * it.p.Cin;
      it.dy = j.dy; it.rowscale = j.rowscale; it.dw = j.dw; it.dbias = j.dbias;
      it.ws = plan[i].split > 1 ? workspace + plan[i].ws_off : nullptr;
      it.mps = plan[i].mps; it.tx = mmt_cdiv(NP, 128); it.ty = mmt_cdiv(it.p.Cout, 128); it.tz = plan[i].split;
      if (plan[i].split > 1) red[nred++] = WgReduceItem{it.ws, j.rowscale, j.dw, plan[i].split, it.p.Cout, NP, 0};
      g.first[g.n] = nb;
      nb += (it.tx * it.ty * it.tz + 7) & ~7;
      g.n++;
      if (g.n == wg_cap(2)) { const int e = flush(); if (e) return e; }
    }
    const int e = flush();
    if (e) return e;
  }
  if (nred) { const int e = launch_wgrad_reduce_group(red, nred, s); if (e) return e; }
  // everything else: the single launches
  for (int i = 0; i < n; i++) {
    if (plan[i].kind != 0) continue;
    const mmt_wgrad_job& j = jobs[i];
    float* ws = plan[i].split > 1 ? workspace + plan[i].ws_off : nullptr;
    int e = 1;
    if (j.x_planes && j.dy_planes && !j.a.x2)
      e = mmt_conv_wgrad_planes(&j.a, j.dy, j.x_planes, j.x_plane_stride, j.dy_planes, j.dy_plane_stride, j.s_x, j.s_dy, j.rowscale, j.dw,
                                j.dbias, ws, stream);
    if (e == 1) e = mmt_conv_wgrad(&j.a, j.dy, j.rowscale, j.dw, j.dbias, ws, stream);
    if (e) return e;
  }
  return 0;
}

extern "C" int mmt_colsum(const float* dy, int M, int C, float* out, void* stream) {
  if (M <= 0 || C <= 0) return 0;
  const int rpb = 1024;
  hipLaunchKernelGGL(colsum_kernel, dim3(mmt_cdiv(C, 64), mmt_cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, dy, M,
                     C, out, rpb);
  MMT_LAUNCH_CHECK();
  return 0;
}
