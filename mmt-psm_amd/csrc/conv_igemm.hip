// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the hot path, the ATen/cuDNN calls behind maskrcnn_benchmark.layers.Conv2d /
// nn.Conv2d / nn.Linear (layers/misc.py:30-43, backbone/fpn.py:30-31, rpn/rpn.py:27-33,
// box_head/roi_box_feature_extractors.py:97-98) together with the elementwise passes the reference
// runs as separate kernels around them: FrozenBatchNorm2d scale/shift (layers/batch_norm.py:19-24),
// ReLU, the bottleneck residual add (backbone/resnet.py:254-274) and the FPN "lateral +
// nearest-upsampled top-down" add (backbone/fpn.py:57-62).
//
// GEMM view (NHWC activations, [Cout][KH][KW][Cin] weights => both operands K-contiguous):
//     C[m][n] = sum_k A[m][k] * B[n][k],  m = (img,ho,wo), n = cout, k = (kh,kw,ci)
// A is gathered on the fly (implicit im2col, zero-filled halo), never materialised in HBM.
//
// Block = 256 threads = 4 waves; block tile BM x BN x 32; wave tile (TM*32) x (TN*32) built from
// 32x32x2 fp32 MFMAs (64 cycles each, exact fp32 FMA chains).  Operand tiles are staged
// global -> VGPR -> LDS with a 2-deep LDS ring: the global loads of tile t+1 are issued before the
// MFMA block of tile t and written to the other LDS buffer after it, one __syncthreads per tile.
// LDS image of a tile is [row][32 floats] with the 16-byte slot index XOR-ed by (row>>1)&7, which makes
// the ds_read_b128 fragment reads (lane = row, 4 consecutive k) conflict-free for all four 16-lane
// service groups of that instruction.  A lane's b128 gives it k = 8*jj + 4*(lane>>5) + {0..3}; the four
// values feed four consecutive MFMAs, and since A and B use the same k permutation the sum is unchanged.
//
// Roofline: MFMA-bound (157.3 TFLOP/s fp32 dense).  Per block tile of 128x128x32: 1.05 MFLOP for
// 32 KiB of operand traffic (mostly L2 hits: neighbouring pixels/taps) => ~32 FLOP/B >> the 26 FLOP/B
// ridge of HBM, so the algorithmic HBM traffic is input + weights + output once.
#include <stdlib.h>
#include "common.h"

namespace {

struct ConvP {
  const float* x; const float* w; const float* scale; const float* shift; const float* res;
  const float* mask; const float* mul; float* y;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
  int relu, res_mode, out_stride, out_H, out_W;
  float mask_scale;
  int M, K, cin32, cin4;  // derived
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const f32x4*)p; }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(const ConvP p) {
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int NA = BM / 32, NB = BN / 32;  // float4 loads per thread per tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsA = lds;                 // [2][BM*32]
  float* ldsB = lds + 2 * BM * 32;   // [2][BN*32]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: each XCD owns a contiguous range of M (its own slice of the activations,
  // read once); consecutive blocks sweep the Cout panels of the same rows, weights stay L2-resident
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = tiles_m * tiles_n, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging geometry
  const int c4 = tid & 7;       // which 16-byte slot of the 32-float k-chunk
  const int r8 = tid >> 3;      // row within a 32-row group
  unsigned abase[NA];           // image base offset of each A row (element offsets fit 32 bits, checked on the host)
  int aih0[NA], aiw0[NA];
  bool aok[NA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < NA; j++) {
    const int m = m0 + r8 + 32 * j;
    aok[j] = m < p.M;
    const int mm = aok[j] ? m : 0;
    const int img = mm / HoWo, rem = mm - img * HoWo;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    aih0[j] = ho * p.stride - p.pad;
    aiw0[j] = wo * p.stride - p.pad;
    abase[j] = (unsigned)img * (unsigned)(p.H * p.W * p.Cin);
  }
  unsigned bbase[NB];
  bool bok[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) {
    const int n = n0 + r8 + 32 * j;
    bok[j] = n < p.Cout;
    bbase[j] = (unsigned)(bok[j] ? n : 0) * (unsigned)p.K;
  }
  // LDS write offsets (floats) for this thread's slots
  int awoff[NA], bwoff[NB];
#pragma unroll
  for (int j = 0; j < NA; j++) { const int row = r8 + 32 * j; awoff[j] = row * 32 + ((c4 ^ ((row >> 1) & 7)) << 2); }
#pragma unroll
  for (int j = 0; j < NB; j++) { const int row = r8 + 32 * j; bwoff[j] = row * 32 + ((c4 ^ ((row >> 1) & 7)) << 2); }
  // LDS fragment read offsets: row = lane&31 (+ tile row base), slot = 2*jj + (lane>>5)
  const int lr = lane & 31, kh2 = lane >> 5, swz = (lr >> 1) & 7;
  int froff[4];
#pragma unroll
  for (int jj = 0; jj < 4; jj++) froff[jj] = lr * 32 + (((2 * jj + kh2) ^ swz) << 2);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const int nkt = (p.K + 31) / 32;
  f32x4 ra[NA], rb[NB];
  bool pa[NA], pb[NB];  // validity of the staged values; the zero-select happens at LDS-store time, so that the
                        // loads stay in flight across the MFMA block (a select right after the load would wait)
  // scalar tap state for the Cin%32==0 fast path
  int s_kh = 0, s_kw = 0, s_ci = 0;

  auto load_tile = [&](int kt) {
    const int k0 = kt * 32;
    if (!p.cin4) {
      // slow generic path (Cin % 4 != 0: data-gradients of the 3/12/15-channel predictor convs): every k
      // element of this thread's 16-byte slot has its own (tap, ci) and is fetched with a scalar load
#pragma unroll
      for (int j = 0; j < NA; j++) { ra[j] = f32x4{0.f, 0.f, 0.f, 0.f}; pa[j] = true; }
#pragma unroll
      for (int j = 0; j < NB; j++) { rb[j] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[j] = true; }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int kidx = k0 + c4 * 4 + e;
        if (kidx >= p.K) continue;
        const int tap = kidx / p.Cin, ci = kidx - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int j = 0; j < NA; j++) {
          const int ih = aih0[j] + kh, iw = aiw0[j] + kw;
          if (aok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
            ra[j][e] = p.x[abase[j] + (unsigned)((ih * p.W + iw) * p.Cin + ci)];
        }
#pragma unroll
        for (int j = 0; j < NB; j++)
          if (bok[j]) rb[j][e] = p.w[bbase[j] + kidx];
      }
      return;
    }
    int kh, kw, ci;
    bool kok;
    if (p.cin32) {
      kh = s_kh; kw = s_kw; ci = s_ci + c4 * 4; kok = true;
      s_ci += 32;
      if (s_ci >= p.Cin) { s_ci = 0; if (++s_kw == p.KW) { s_kw = 0; ++s_kh; } }
    } else {
      const int kidx = k0 + c4 * 4;
      kok = kidx < p.K;
      const int tap = kidx / p.Cin;
      ci = kidx - tap * p.Cin;
      kh = tap / p.KW; kw = tap - kh * p.KW;
    }
    // branch-free: an invalid lane reads element 0 of the tensor and the value is discarded by a select, so the
    // compiler emits straight-line loads (no exec-mask branch + vmcnt(0) per row)
#pragma unroll
    for (int j = 0; j < NA; j++) {
      const int ih = aih0[j] + kh, iw = aiw0[j] + kw;
      const bool ok = kok && aok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned off = ok ? abase[j] + (unsigned)((ih * p.W + iw) * p.Cin + ci) : 0u;
      ra[j] = ldg4(p.x + off);
      pa[j] = ok;
    }
    const int kidx = k0 + c4 * 4;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const bool ok = bok[j] && kidx < p.K;
      rb[j] = ldg4(p.w + (ok ? bbase[j] + (unsigned)kidx : 0u));
      pb[j] = ok;
    }
  };
  auto store_tile = [&](int buf) {
    float* A = ldsA + buf * BM * 32;
    float* B = ldsB + buf * BN * 32;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NA; j++) *(f32x4*)(A + awoff[j]) = pa[j] ? ra[j] : zero4;
#pragma unroll
    for (int j = 0; j < NB; j++) *(f32x4*)(B + bwoff[j]) = pb[j] ? rb[j] : zero4;
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const float* A = ldsA + buf * BM * 32 + (wm * TM * 32) * 32;
    const float* B = ldsB + buf * BN * 32 + (wn * TN * 32) * 32;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; a++) fa[a] = *(const f32x4*)(A + a * 32 * 32 + froff[jj]);
#pragma unroll
      for (int b = 0; b < TN; b++) fb[b] = *(const f32x4*)(B + b * 32 * 32 + froff[jj]);
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
          for (int b = 0; b < TN; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][t], fb[b][t], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (the operand ring is free now) -> row-major float4 rows, so that stores,
  // residual / mask / mul loads are all 16 B per lane and fully coalesced (a 128-wide tile row = 512 B), and the
  // (img,ho,wo) decode is per row instead of per element.  Matters for the low-K 1x1 layers, which are
  // store-bound: their whole runtime is this epilogue.
  {
    float* ct = lds;  // [BM][BN]
    const int col_l = lane & 31, rq = lane >> 5;
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int b = 0; b < TN; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
          ct[row * BN + (wn * TN + b) * 32 + col_l] = acc[a][b][r];
        }
    __syncthreads();
    constexpr int C4 = BN / 4, RPP = 256 / C4;
    const int cc = tid % C4, r0 = tid / C4;
    const int c = n0 + cc * 4;
    if (c < p.Cout) {
      const bool vec = (p.Cout & 3) == 0;
      const int nv = vec ? 4 : min(4, p.Cout - c);
      float sc[4], sh[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        sc[e] = (p.scale && e < nv) ? p.scale[c + e] : 1.f;
        sh[e] = (p.shift && e < nv) ? p.shift[c + e] : 0.f;
      }
      auto ld = [&](const float* q, float* o) {
        if (vec) { const f32x4 t = ldg4(q); o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; }
        else { for (int e = 0; e < 4; e++) o[e] = e < nv ? q[e] : 0.f; }
      };
      for (int row = r0; row < BM; row += RPP) {
        const int m = m0 + row;
        if (m >= p.M) break;
        const f32x4 t = *(const f32x4*)(ct + row * BN + cc * 4);
        float v[4] = {t[0] * sc[0] + sh[0], t[1] * sc[1] + sh[1], t[2] * sc[2] + sh[2], t[3] * sc[3] + sh[3]};
        long oidx = (long)m * p.Cout + c;
        float u[4];
        if (p.res_mode >= 2 || p.out_stride > 1) {
          const int img = m / HoWo, rem = m - img * HoWo;
          const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
          if (p.res_mode == 2) {
            const int h2 = p.Ho >> 1, w2 = p.Wo >> 1;
            ld(p.res + (((long)img * h2 + (ho >> 1)) * w2 + (wo >> 1)) * p.Cout + c, u);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += u[e];
          } else if (p.res_mode == 3) {
            const int h2 = p.Ho * 2, w2 = p.Wo * 2;
            const float* rp = p.res + (((long)img * h2 + 2 * ho) * w2 + 2 * wo) * p.Cout + c;
            float u1[4], u2[4], u3[4];
            ld(rp, u); ld(rp + p.Cout, u1); ld(rp + (long)w2 * p.Cout, u2); ld(rp + (long)w2 * p.Cout + p.Cout, u3);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += (u[e] + u1[e]) + (u2[e] + u3[e]);
          }
          if (p.out_stride > 1)
            oidx = (((long)img * p.out_H + ho * p.out_stride) * p.out_W + wo * p.out_stride) * p.Cout + c;
        }
        if (p.res_mode == 1) {
          ld(p.res + (long)m * p.Cout + c, u);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += u[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.mask) {
          ld(p.mask + oidx, u);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = u[e] > 0.f ? v[e] * p.mask_scale : 0.f;
        }
        if (p.mul) {
          ld(p.mul + (long)m * p.Cout + c, u);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] *= u[e];
        }
        if (vec) *(f32x4*)(p.y + oidx) = f32x4{v[0], v[1], v[2], v[3]};
        else for (int e = 0; e < nv; e++) p.y[oidx + e] = v[e];
      }
    }
  }
}

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

// dw[co][n] += rowscale[co] * sum_s ws[s][co][n]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int Cout, int NP,
                                                           const float* __restrict__ rowscale,
                                                           float* __restrict__ dw) {
  const long n4 = (long)Cout * NP / 4;
  const long slab = (long)Cout * NP;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 a = ((const f32x4*)ws)[i];
    for (int s = 1; s < splits; s++) {
      const f32x4 b = *(const f32x4*)(ws + s * slab + i * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) a[e] += b[e];
    }
    const float sc = rowscale ? rowscale[(int)((i * 4) / NP)] : 1.f;
    f32x4 o = ((f32x4*)dw)[i];
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

__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                      int H, int W, int C, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * (C / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % (C / 4));
    long r = i / (C / 4);
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dh = 0; dh < 3; dh++) {
      const int ih = ho * 2 - 1 + dh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int dwi = 0; dwi < 3; dwi++) {
        const int iw = wo * 2 - 1 + dwi;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = ldg4(x + (((long)n * H + ih) * W + iw) * C + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; e++) m[e] = fmaxf(m[e], v[e]);
      }
    }
    *(f32x4*)(y + (((long)n * Ho + ho) * Wo + wo) * C + c4 * 4) = m;
  }
}

int fill(ConvP& p, const mmt_conv_args* a) {
  if (!a || !a->x) return MMT_EINVAL;
  p.x = a->x; p.w = a->w; p.scale = a->scale; p.shift = a->shift; p.res = a->res; p.mask = a->mask;
  p.mul = a->mul; p.y = a->y;
  p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW;
  p.stride = a->stride; p.pad = a->pad; p.Ho = a->Ho; p.Wo = a->Wo;
  p.relu = a->relu; p.res_mode = a->res_mode; p.out_stride = a->out_stride < 1 ? 1 : a->out_stride;
  p.out_H = a->out_H; p.out_W = a->out_W; p.mask_scale = a->mask_scale;
  if ((long)p.N * p.Ho * p.Wo > 0x7fffffffL) return MMT_EINVAL;
  if ((long)p.N * p.H * p.W * p.Cin >= 0x7fffffffL || (long)p.N * p.Ho * p.Wo * p.Cout >= 0x7fffffffL ||
      (long)p.Cout * p.KH * p.KW * p.Cin >= 0x7fffffffL) return MMT_EINVAL;  // kernels use 32-bit element offsets
  p.M = p.N * p.Ho * p.Wo;
  p.K = p.KH * p.KW * p.Cin;
  p.cin32 = (p.Cin % 32) == 0;
  p.cin4 = (p.Cin % 4) == 0;
  return 0;
}

template <int BM, int BN, int WM, int WN>
int launch_fwd(const ConvP& p, hipStream_t s) {
  const int tiles = mmt_cdiv(p.M, BM) * mmt_cdiv(p.Cout, BN);
  const size_t lds = (size_t)(2 * BM * 32 + 2 * BN * 32) * sizeof(float);
  hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN>), dim3(tiles), dim3(256), lds, s, p);
  MMT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

static int pick_variant(const ConvP& p) {
  if (p.Cout <= 32) return 0;
  // K <= 256 (the 1x1 layers of layer1/layer2/FPN laterals): 2-8 k-tiles per output tile, so prologue and epilogue
  // dominate; the 64x64 configuration keeps ~4x more blocks resident to overlap them (measured +10..25 %)
  static const int lowk = getenv("MMT_LOWK") ? atoi(getenv("MMT_LOWK")) : 256;
  if (p.K <= lowk) return 2;
  // enough 128x128 tiles to fill 256 CUs (2 resident blocks each) -> 128x128; else 128x64 (twice the blocks, A tile
  // still reused across 64 output channels); else 64x64 (4x the blocks)
  const long t128 = (long)mmt_cdiv(p.M, 128) * mmt_cdiv(p.Cout, 128);
  if (t128 >= 384 && p.Cout > 64) return 1;
  static const int mid = getenv("MMT_MID") ? atoi(getenv("MMT_MID")) : 1;
  const long t12864 = (long)mmt_cdiv(p.M, 128) * mmt_cdiv(p.Cout, 64);
  if (mid && t12864 >= 256 && p.Cout >= 64) return 3;
  return 2;
}

extern "C" int mmt_conv_variant(const mmt_conv_args* a) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  return pick_variant(p);
}

extern "C" int mmt_conv_forward(const mmt_conv_args* a, void* stream) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  if (!p.w || !p.y) return MMT_EINVAL;
  if (p.M == 0 || p.Cout == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  switch (pick_variant(p)) {
    case 0: return launch_fwd<128, 32, 4, 1>(p, s);
    case 1: return launch_fwd<128, 128, 2, 2>(p, s);
    case 3: return launch_fwd<128, 64, 2, 2>(p, s);
    default: return launch_fwd<64, 64, 2, 2>(p, s);
  }
}

extern "C" int mmt_conv_wgrad_splits(const mmt_conv_args* a) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  if (p.M == 0 || p.Cout == 0) return 1;
  const int NP = p.KH * p.KW * p.Cin;
  const int tx = mmt_cdiv(NP, 128), ty = mmt_cdiv(p.Cout, 128);
  int split = mmt_cdiv(640, (long)tx * ty);  // ~1.25 resident rounds of 2 blocks x 256 CUs
  const int max_split = mmt_cdiv(p.M, 512);  // at least 16 k-tiles per block
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
  int mps = mmt_cdiv(p.M, split);
  mps = (mps + 31) / 32 * 32;
  if (split > 1 && !workspace) return MMT_EINVAL;
  float* ws = split > 1 ? workspace : nullptr;
  const bool fast = (p.Cout & 3) == 0 && p.Wo >= 8 && p.Ho >= 8;
  if (fast)
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

extern "C" int mmt_colsum(const float* dy, int M, int C, float* out, void* stream) {
  if (M <= 0 || C <= 0) return 0;
  const int rpb = 1024;
  hipLaunchKernelGGL(colsum_kernel, dim3(mmt_cdiv(C, 64), mmt_cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, dy, M,
                     C, out, rpb);
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
  hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo);
  MMT_LAUNCH_CHECK();
  return 0;
}
