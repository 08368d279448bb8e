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
//
// Round 6: this translation unit holds the FORWARD / data-gradient kernels that read fp32 tensors (fp32 MFMA, bf16 x 3 and two-term
// fp16 tiled kernels, the tap-strip 3x3 kernel, the row-resident 1x1 kernel, split-K finish) and the dispatch behind mmt_conv_forward /
// mmt_conv_forward_f16x2.  The weight gradients are conv_wgrad.hip (+ conv_wgpl.hip), operand preparation (weight packing, plane
// splits, statistics, sums, max-pool) is conv_prep.hip, the plane-fed GEMM conv_pgemm.hip, the stem and frozen-stage kernels
// conv_stem.hip; what they share is conv_shared.h.  One experiment rebuilds one object.
#include <stdlib.h>
#include <map>
#include <mutex>
#include <type_traits>
#include "conv_shared.h"

namespace {

// ---- shared epilogue of the forward kernels: accumulators -> LDS (the operand ring is free by then) -> row-major
// float4 rows, so that stores and residual / mask / mul loads are all 16 B per lane and fully coalesced (a 128-wide
// tile row = 512 B), and the (img,ho,wo) decode is per row instead of per element.  Matters for the low-K 1x1
// layers, which are store-bound: their whole runtime is this epilogue.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_stage(f32x16 (&acc)[BM / (32 * WM)][BN / (32 * WN)], float* lds,
                                                    const int lane, const int wm, const int wn) {
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
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
}

// second half of the epilogue: the [BM][BN] tile in LDS (complete, synchronised) -> scale/shift, residual, ReLU, ... -> y

// the block's contribution to the statistics slot of its output: ONE conditional atomic per block (wave reduce, LDS reduce
// over the waves, then the test against what is already there).  One per wave was measured at 3 - 8 us of a 36 us launch
// when a single round of tiles finishes together (1024 same-address atomics at 11 - 13 ns each).  Every thread of the
// block calls this (it holds a barrier).
__device__ __forceinline__ void conv_amax_commit(const ConvP& p, AmaxAcc a, const int lin) {
  __shared__ float red[3][8];
  const int t = threadIdx.x, w = t >> 6, nw = (blockDim.x + 63) >> 6;
  const bool stats = p.amax_stats && (lin & 63) == 0;   // mean |y| from a SAMPLE of the blocks: atomics on the two cache
                                                         // lines of a slot serialise at the L2
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a.amx = fmaxf(a.amx, __shfl_xor(a.amx, o, 64));
  if (stats) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a.asum += __shfl_xor(a.asum, o, 64); a.acnt += __shfl_xor(a.acnt, o, 64); }
  }
  if ((t & 63) == 0) { red[0][w] = a.amx; red[1][w] = a.asum; red[2][w] = a.acnt; }
  __syncthreads();
  if (t == 0) {
    float m = red[0][0], sm = red[1][0], cn = red[2][0];
    for (int i = 1; i < nw; i++) { m = fmaxf(m, red[0][i]); sm += red[1][i]; cn += red[2][i]; }
    const unsigned bits = __builtin_bit_cast(unsigned, m);
    if (m > 0.f && bits > __hip_atomic_load(p.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p.amax_out, bits);
    if (p.amax_next && m > 0.f && bits > __hip_atomic_load(p.amax_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p.amax_next, bits);
    if (stats && cn > 0.f) {
      const int k = (lin >> 6) & 15;
      atomicAdd((float*)p.amax_out + 1 + k, sm);
      atomicAdd((float*)p.amax_out + 17 + k, cn);
    }
  }
}

// acc: the caller runs this more than once per block and commits the statistics itself (conv_amax_commit); null: committed here
template <int BM, int BN>
__device__ __forceinline__ void conv_epilogue_finish(const ConvP& p, float* lds, const int m0, const int n0, const int tid,
                                                     const int HoWo, AmaxAcc* acc = nullptr) {
  float amx = 0.f, asum = 0.f, acnt = 0.f;
  {
    float* ct = lds;  // [BM][BN]
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
      // the same for a tensor stored as bf16 (element index = the fp32 tensor's)
      auto ldh = [&](const float* base, long idx, float* o) {
        const unsigned short* q = (const unsigned short*)base + idx;
        if (vec) {
          const uint2 t = *(const uint2*)q;
          o[0] = __builtin_bit_cast(float, t.x << 16); o[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
          o[2] = __builtin_bit_cast(float, t.y << 16); o[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
        } else { for (int e = 0; e < 4; e++) o[e] = e < nv ? __builtin_bit_cast(float, (unsigned)q[e] << 16) : 0.f; }
      };
      if (p.f16_sx) {  // operands were scaled by powers of two: exact rescale of the accumulated sum
        const float inv = 1.f / ((p.f16_ax ? f16_scale_of(*p.f16_sx) : *p.f16_sx) * *p.f16_sw);
#pragma unroll
        for (int e = 0; e < 4; e++) sc[e] *= inv;
      }
      const bool res_h = p.io & IO_RES, mask_h = p.io & IO_MASK;
      const RbGeom rbg = rb_geom(p);                       // round 6: y also as row-blocked fp16 planes (host: Cout % 16 == 0)
      const float rbs = p.yrb ? *p.yrb_s : 1.f;
      // rows in groups of G: every residual / mask / mul load of a group is issued before the first use, so a thread
      // pays one global-load latency per group instead of one per row (the row loop is not unrollable past its stores)
      constexpr int ROWS = BM / RPP, G = ROWS < 4 ? ROWS : 4;
      static_assert(ROWS % G == 0, "row groups");
      for (int g0 = 0; g0 < ROWS; g0 += G) {
        if (m0 + r0 + g0 * RPP >= p.M) break;
        float ur[G][4], um[G][4], ul[G][4], u1[G][4], u2[G][4], u3[G][4];
        long oidx[G];
        bool ok[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
          const int m = m0 + r0 + (g0 + g) * RPP;
          ok[g] = m < p.M;
          oidx[g] = (long)m * p.Cout + c;
          if (!ok[g]) continue;
          if (p.res_mode >= 2 || p.out_stride > 1) {
            const int img = m / HoWo, rem = m - img * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            if (p.res_mode == 2) {
              const int h2 = p.Ho >> 1, w2 = p.Wo >> 1;
              const long ri = (((long)img * h2 + (ho >> 1)) * w2 + (wo >> 1)) * p.Cout + c;
              if (res_h) ldh(p.res, ri, ur[g]); else ld(p.res + ri, ur[g]);
            } else if (p.res_mode == 3) {
              const int h2 = p.Ho * 2, w2 = p.Wo * 2;
              const float* rp = p.res + (((long)img * h2 + 2 * ho) * w2 + 2 * wo) * p.Cout + c;
              ld(rp, ur[g]); ld(rp + p.Cout, u1[g]); ld(rp + (long)w2 * p.Cout, u2[g]);
              ld(rp + (long)w2 * p.Cout + p.Cout, u3[g]);
            }
            if (p.out_stride > 1)
              oidx[g] = (((long)img * p.out_H + ho * p.out_stride) * p.out_W + wo * p.out_stride) * p.Cout + c;
          }
          if (p.res_mode == 1) { if (res_h) ldh(p.res, (long)m * p.Cout + c, ur[g]); else ld(p.res + (long)m * p.Cout + c, ur[g]); }
          if (p.mask) { if (mask_h) ldh(p.mask, oidx[g], um[g]); else ld(p.mask + oidx[g], um[g]); }
          if (p.mul) ld(p.mul + (long)m * p.Cout + c, ul[g]);
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
          if (!ok[g]) continue;
          const int row = r0 + (g0 + g) * RPP;
          const f32x4 t = *(const f32x4*)(ct + row * BN + cc * 4);
          float v[4] = {t[0] * sc[0] + sh[0], t[1] * sc[1] + sh[1], t[2] * sc[2] + sh[2], t[3] * sc[3] + sh[3]};
          if (p.res_mode == 1 || p.res_mode == 2) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += ur[g][e];
          } else if (p.res_mode == 3) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += (ur[g][e] + u1[g][e]) + (u2[g][e] + u3[g][e]);
          }
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
          }
          if (p.mask) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = um[g][e] > 0.f ? v[e] * p.mask_scale : 0.f;
          }
          if (p.mul) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] *= ul[g][e];
          }
          if (p.amax_out) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const float av = e < nv ? fabsf(v[e]) : 0.f; amx = fmaxf(amx, av); asum += av; }
            acnt += (float)nv;
          }
          if (p.io & IO_Y) {  // y is a bf16 tensor: round to nearest even, nothing else is written
            unsigned short* yh = (unsigned short*)p.y + oidx[g];
            const unsigned lo = pk_bf16(v[0], v[1]), hi = pk_bf16(v[2], v[3]);
            if (vec) *(uint2*)yh = uint2{lo, hi};
            else { const unsigned short h4[4] = {(unsigned short)lo, (unsigned short)(lo >> 16), (unsigned short)hi, (unsigned short)(hi >> 16)};
                   for (int e = 0; e < nv; e++) yh[e] = h4[e]; }
            continue;
          }
          if (vec) *(f32x4*)(p.y + oidx[g]) = f32x4{v[0], v[1], v[2], v[3]};
          else for (int e = 0; e < nv; e++) p.y[oidx[g] + e] = v[e];
          if (p.ypl) {  // (host: only with Cout % 4 == 0) the consumer's plane split, done here while the values are in registers
            uint2 o[3];
            split4<3>(f32x4{v[0], v[1], v[2], v[3]}, o);
#pragma unroll
            for (int q = 0; q < 3; q++) *(uint2*)(p.ypl + q * p.ypl_stride + oidx[g]) = o[q];
          }
          if (p.yrb && m0 + r0 + (g0 + g) * RPP < p.yrb_M) rb_store4(p, rbg, m0 + r0 + (g0 + g) * RPP, c, f32x4{v[0], v[1], v[2], v[3]}, rbs);
        }
      }
    }
  }
  if (p.amax_out) {
    if (acc) { acc->amx = fmaxf(acc->amx, amx); acc->asum += asum; acc->acnt += acnt; }
    else conv_amax_commit(p, AmaxAcc{amx, asum, acnt}, blockIdx.x);
  }
}

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[BM / (32 * WM)][BN / (32 * WN)], float* lds,
                                              const int m0, const int n0, const int tid, const int lane,
                                              const int wm, const int wn, const int HoWo) {
  conv_epilogue_stage<BM, BN, WM, WN>(acc, lds, lane, wm, wn);
  __syncthreads();
  conv_epilogue_finish<BM, BN>(p, lds, m0, n0, tid, HoWo);
}

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

  conv_epilogue<BM, BN, WM, WN>(p, acc, lds, m0, n0, tid, lane, wm, wn, HoWo);
}

// ------------------------------------------------------------------------------------ split-bf16 forward
// The same implicit GEMM on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate) with fp32
// operands in HBM: while a tile is staged into LDS every fp32 value x is split into NS bf16 terms (round-to-nearest at
// each level, residuals exact in fp32):   x = x0 + x1 (+ x2),  |x1| <= 2^-9 |x|,  |x2| <= 2^-18 |x|.
//   NS = 3: a*b ~= a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0); dropped terms <= 3 * 2^-27 |ab|, i.e. below the fp32
//           rounding of the product itself -> fp32-grade results (accumulation is fp32, small terms first) at 6 MFMAs
//           per 16-k step = 2500/6 = 417 TFLOP/s effective peak vs 157 for the fp32-input MFMA
//   NS = 2: 3 products, ~2^-17 relative per product;   NS = 1: plain bf16 inputs (BASELINE config 5's "bf16 MFMA path")
// LDS image per plane: [row][16 bf16] (32 B rows); the 16-byte half holding k 0..7 / 8..15 is XOR-ed with (row>>3)&1,
// which makes the ds_read_b128 fragment reads conflict-free for the four 16-lane service groups; A and B use the same
// lane->k assignment, so the k order inside the instruction does not matter.  Tile = BM x BN x 16, 2-deep LDS ring.
// Requires Cin % 16 == 0 (every heavy layer of the path); other shapes run on the fp32 kernel above.

template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256, 2) void conv_fwd_split_kernel(const ConvP p) {
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int NA = BM / 64, NB = BN / 64;  // float4 loads per thread per 16-k tile
  constexpr int PA = BM * 32, PB = BN * 32;  // bytes per bf16 plane
  constexpr int STAGE = NS * (PA + PB);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ring = (char*)lds;  // [2][NS planes of A | NS planes of B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = tiles_m * tiles_n, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int c4 = tid & 3;    // which float4 of the 16-float k-chunk
  const int r64 = tid >> 2;  // row within a 64-row group
  unsigned abase[NA];
  int aih0[NA], aiw0[NA];
  bool aok[NA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < NA; j++) {
    const int m = m0 + r64 + 64 * j;
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
    const int n = n0 + r64 + 64 * j;
    bok[j] = n < p.Cout;
    bbase[j] = (unsigned)(bok[j] ? n : 0) * (unsigned)p.K;
  }
  // byte offset of this thread's 8-byte slot inside a plane (rows r64 + 64 j: (row>>3)&1 does not depend on j)
  const int woff = r64 * 32 + ((((c4 >> 1) ^ (r64 >> 3)) & 1) << 4) + ((c4 & 1) << 3);
  const int lr = lane & 31, kh2 = lane >> 5;
  const int froff = lr * 32 + (((kh2 ^ (lr >> 3)) & 1) << 4);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const int nkt = p.K >> 4;
  f32x4 ra[NA], rb[NB];
  bool pa[NA], pb[NB];
  int s_kh = 0, s_kw = 0, s_ci = 0;

  auto load_tile = [&](int kt) {
    const int kh = s_kh, kw = s_kw, ci = s_ci + c4 * 4;
    s_ci += 16;
    if (s_ci >= p.Cin) { s_ci = 0; if (++s_kw == p.KW) { s_kw = 0; ++s_kh; } }
#pragma unroll
    for (int j = 0; j < NA; j++) {
      const int ih = aih0[j] + kh, iw = aiw0[j] + kw;
      const bool ok = aok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned off = ok ? abase[j] + (unsigned)((ih * p.W + iw) * p.Cin + ci) : 0u;
      ra[j] = ldg4(p.x + off);
      pa[j] = ok;
    }
    const int kidx = kt * 16 + c4 * 4;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      rb[j] = ldg4(p.w + (bok[j] ? bbase[j] + (unsigned)kidx : 0u));
      pb[j] = bok[j];
    }
  };
  auto store_tile = [&](int buf) {
    char* A = ring + buf * STAGE + woff;
    char* B = A + NS * PA;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NA; j++) {
      uint2 o[NS];
      split4<NS>(pa[j] ? ra[j] : zero4, o);
#pragma unroll
      for (int q = 0; q < NS; q++) *(uint2*)(A + q * PA + j * 64 * 32) = o[q];
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
      uint2 o[NS];
      split4<NS>(pb[j] ? rb[j] : zero4, o);
#pragma unroll
      for (int q = 0; q < NS; q++) *(uint2*)(B + q * PB + j * 64 * 32) = o[q];
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* A = ring + buf * STAGE + (wm * TM * 32) * 32 + froff;
    const char* B = ring + buf * STAGE + NS * PA + (wn * TN * 32) * 32 + froff;
    bf16x8 fa[NS][TM], fb[NS][TN];
#pragma unroll
    for (int q = 0; q < NS; q++) {
#pragma unroll
      for (int a = 0; a < TM; a++) fa[q][a] = *(const bf16x8*)(A + q * PA + a * 32 * 32);
#pragma unroll
      for (int b = 0; b < TN; b++) fb[q][b] = *(const bf16x8*)(B + q * PB + b * 32 * 32);
    }
    // smallest terms first: order s = qa + qb descending
#pragma unroll
    for (int sum = 2 * (NS - 1) > NS - 1 ? NS - 1 : 0; sum >= 0; sum--)
#pragma unroll
      for (int qa = 0; qa <= sum; qa++) {
        const int qb = sum - qa;
        if (qa < NS && qb < NS) {
#pragma unroll
          for (int a = 0; a < TM; a++)
#pragma unroll
            for (int b = 0; b < TN; b++)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
        }
      }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }
  conv_epilogue<BM, BN, WM, WN>(p, acc, lds, m0, n0, tid, lane, wm, wn, HoWo);
}

// ------------------------------------------------------------------------------------ split-bf16 forward, DMA-fed
// Same arithmetic as conv_fwd_split_kernel, restructured around what that kernel measured as its limit (ds_write
// bandwidth of ~80 B/clk/CU and one store-phase + barrier per 16-k step; the split VALU work itself was free):
//   * B (weights) arrives PRE-SPLIT: NS bf16 planes of the weight tensor in HBM (mmt_pack_weights, refreshed once per
//     optimiser / EMA step), copied global -> LDS by the DMA path (global_load_lds_dwordx4: no VGPR staging, no ds_write);
//   * A (activations) is DMA-copied as raw fp32 and split in registers AFTER the fragment read; with the 4x1 wave grid
//     the rows of a wave are private to it, so every activation value is split once per block (8 values per lane per
//     16-k step);
//   * 3-deep LDS ring, one raw s_barrier per step, counted vmcnt: the DMA of step t+2 is issued right after the barrier
//     of step t and stays in flight across the next barrier.
// LDS images are lane-linear per DMA instruction (hardware: dest = base + 16 * lane); the bank swizzles are applied on
// the SOURCE address and again on the fragment read (A: 16-B chunk ^= (row>>2)&3 of a 64-B row; B: half ^= (row>>3)&1).
__device__ __attribute__((aligned(16))) const float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};  // source of halo / OOB lanes

__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_ptr_t)l, 16, 0, 0);
}

// The same copy as an instruction the compiler does not see.  It orders every LDS read that might alias an LDS-DMA target
// behind `s_waitcnt vmcnt(0)`; in a loop that keeps register loads and stores in flight across the copy (the row-resident
// 1x1 kernel) that wait would drain them all before each matrix phase.  The kernel waits for its copies with counted
// `s_waitcnt vmcnt(n)` itself.  `l` is wave-uniform (M0 = LDS byte address, lane i lands at + 16 i).
__device__ __forceinline__ void dma16_unseen(const void* g, void* l) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)l);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "m0", "memory");
#endif
}

// F16 (opt-in, NS == 2): the weight planes hold the two fp16 terms of w * s_w; the activations are scaled by the power of
// two of their recorded maximum (p.f16_sx -> max |x|, a device scalar) and split into two fp16 terms in registers; 3 products
template <int BM, int BN, int WM, int WN, int NS, int S, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv_fwd_glds_kernel(const ConvP p, const unsigned short* __restrict__ wpl,
                                                               const long wpl_stride, const int ksplit,
                                                               float* __restrict__ ws) {
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int A_BYTES = BM * 64;           // [BM][16] fp32
  constexpr int PB = BN * 32;                // one bf16 plane [BN][16]
  constexpr int STAGE = A_BYTES + NS * PB;
  constexpr int NIA = BM / 64;               // A DMA instructions per wave per step (16 rows each)
  constexpr int NIB_TOT = NS * BN / 32;      // B DMA instructions per step (32 rows of one plane each)
  constexpr int NIB = (NIB_TOT + 3) / 4;     // per wave (the tail wraps around: same data written twice, harmless)
  constexpr int NI = NIA + NIB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ring = (char*)lds;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = tiles_m * tiles_n * ksplit, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // split-K (few tiles, long K: the deep 3x3 layers at small batch, fc6): the ksplit blocks of a tile are neighbours
  const int ks = bid % ksplit;
  bid /= ksplit;
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int HoWo = p.Ho * p.Wo;
  F16Guard guard = {};
  if constexpr (F16) guard = f16_guard_load(p.guard_x);   // issued here, tested behind the prologue's copies
  // ---- DMA source geometry.  A instruction i of this wave covers rows (wave + 4 i) * 16 + (lane >> 2), 16-B chunk
  // (lane & 3) of the LDS row, which holds logical chunk (lane & 3) ^ ((row >> 2) & 3).
  unsigned abase[NIA];
  int aih0[NIA], aiw0[NIA], achunk[NIA];
  bool aok[NIA];
#pragma unroll
  for (int i = 0; i < NIA; i++) {
    const int row = (wave + 4 * i) * 16 + (lane >> 2);
    const int m = m0 + row;
    aok[i] = m < p.M;
    const int mm = aok[i] ? m : 0;
    const int img = mm / HoWo, rem = mm - img * HoWo;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    aih0[i] = ho * p.stride - p.pad;
    aiw0[i] = wo * p.stride - p.pad;
    abase[i] = (unsigned)img * (unsigned)(p.H * p.W * p.Cin);
    achunk[i] = ((lane & 3) ^ ((row >> 2) & 3)) * 4;
  }
  // B instruction t = wave + 4 i (mod NIB_TOT): plane t / (BN/32), 32-row block t % (BN/32).  The packed weight planes
  // (mmt_pack_weights) hold, for every 16-k step and 32-channel block, the 1 KiB LDS image itself (swizzle included),
  // so a DMA instruction reads 1 KiB of consecutive memory: 8 full cache lines instead of 32 quarter-used ones
  const int nb32 = (p.Cout + 31) >> 5;
  const long kt_stride = (long)nb32 * 512;  // bf16 elements per 16-k step
  const unsigned short* bsrc[NIB];
  int bdst[NIB];
#pragma unroll
  for (int i = 0; i < NIB; i++) {
    const int t = (wave + 4 * i) % NIB_TOT;
    const int q = t / (BN / 32), rb = t % (BN / 32);
    int nb = n0 / 32 + rb;
    if (nb >= nb32) nb = nb32 - 1;  // tile hanging over Cout: any valid block (those columns are never stored)
    bsrc[i] = wpl + q * wpl_stride + (long)nb * 512 + lane * 8;
    bdst[i] = A_BYTES + q * PB + rb * 1024;
  }
  // ---- staging micro-operations.  Everything a wave does for FUTURE steps is cut into small pieces that are placed,
  // one or two at a time, behind the individual MFMAs of the current step (hard scheduling fences in between: left to
  // itself the compiler issues the staging code in one burst and chains MFMAs on the same accumulator).  Measured on
  // this part: vector-ALU work hardly ever co-executes with an MFMA of the OTHER wave of the SIMD (SQ_VALU_MFMA_COEXEC
  // 7 % of MFMA-busy), so hiding it in the MFMA shadow of the same wave is what counts.
  const int nkt_all = p.K >> 4;
  const int kt0 = (int)((long)ks * nkt_all / ksplit), nkt = (int)((long)(ks + 1) * nkt_all / ksplit) - kt0;
  int s_kh, s_kw, s_ci;   // tap / channel position of the next A chunk to fetch (this block's K range starts at step kt0)
  {
    const int k0 = kt0 * 16, tap0 = k0 / p.Cin;
    s_ci = k0 - tap0 * p.Cin;
    s_kh = tap0 / p.KW;
    s_kw = tap0 - s_kh * p.KW;
  }
  bool fresh = true;  // a K range may start in the middle of a tap
  int c_kh = 0, c_kw = 0, c_ci = 0;   // ... latched for the DMAs of this step
  unsigned roff[NIA];  // element offset of (row, current tap, channel 0 + this lane's chunk); valid flag
  bool rok[NIA];
  const float* const zsrc = g_zero16;
  auto tap_next = [&]() {
    c_kh = s_kh; c_kw = s_kw; c_ci = s_ci;
    if (c_ci == 0 || fresh) {  // first chunk of a tap (once per Cin/16 steps): bounds and row offsets for the whole tap
      fresh = false;
#pragma unroll
      for (int i = 0; i < NIA; i++) {
        const int ih = aih0[i] + c_kh, iw = aiw0[i] + c_kw;
        rok[i] = aok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        roff[i] = abase[i] + (unsigned)((ih * p.W + iw) * p.Cin + achunk[i]);
      }
    }
    s_ci += 16;
    if (s_ci >= p.Cin) { s_ci = 0; if (++s_kw == p.KW) { s_kw = 0; ++s_kh; } }
  };
  bool filling = true;  // false in the tail steps (nothing left to fetch): their DMAs read the 16 zero bytes instead
  auto dma_a = [&](int i, int stage) {
    const float* src = rok[i] && filling ? p.x + (roff[i] + (unsigned)c_ci) : zsrc;
    dma16(src, ring + stage * STAGE + (wave + 4 * i) * 1024);
  };
  auto dma_b = [&](int i, int kt, int stage) {
    const void* src = filling ? (const void*)(bsrc[i] + (long)(kt0 + kt) * kt_stride) : (const void*)zsrc;
    dma16(src, ring + stage * STAGE + bdst[i]);
  };
  auto issue_all = [&](int kt, int stage) {  // prologue form
    tap_next();
#pragma unroll
    for (int i = 0; i < NIA; i++) dma_a(i, stage);
#pragma unroll
    for (int i = 0; i < NIB; i++) dma_b(i, kt, stage);
  };

  // ---- fragment read offsets
  const int lr = lane & 31, kh2 = lane >> 5;
  int aoff[TM][2];
#pragma unroll
  for (int a = 0; a < TM; a++) {
    const int row = (wm * TM + a) * 32 + lr;
#pragma unroll
    for (int h = 0; h < 2; h++) aoff[a][h] = row * 64 + (((2 * kh2 + h) ^ ((row >> 2) & 3)) << 4);
  }
  const int boff = A_BYTES + (wn * TN * 32 + lr) * 32 + (((kh2 ^ (lr >> 3)) & 1) << 4);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  f32x4 va[TM][2];        // raw fp32 A fragment halves of the next step; turned into residuals by the split levels
  unsigned ua[TM][2][2];  // packed bf16 pairs of the level being produced
  uint2 oa[TM][2][NS];    // bf16 terms of the A fragment halves
  float f16_sx = 1.f;
  if constexpr (F16) f16_sx = f16_scale_of(*p.f16_sx);
  auto split_cvt = [&](int a, int h, int q) {
    if constexpr (F16) {
      if (q == 0) va[a][h] *= f16_sx;
      ua[a][h][0] = __builtin_bit_cast(unsigned, f16x2{(_Float16)va[a][h][0], (_Float16)va[a][h][1]});
      ua[a][h][1] = __builtin_bit_cast(unsigned, f16x2{(_Float16)va[a][h][2], (_Float16)va[a][h][3]});
    } else {
      ua[a][h][0] = pk_bf16(va[a][h][0], va[a][h][1]);
      ua[a][h][1] = pk_bf16(va[a][h][2], va[a][h][3]);
    }
    oa[a][h][q] = uint2{ua[a][h][0], ua[a][h][1]};
  };
  auto split_sub = [&](int a, int h) {
    if constexpr (F16) {
      const f16x2 h0 = __builtin_bit_cast(f16x2, ua[a][h][0]), h1 = __builtin_bit_cast(f16x2, ua[a][h][1]);
      va[a][h][0] -= (float)h0[0]; va[a][h][1] -= (float)h0[1]; va[a][h][2] -= (float)h1[0]; va[a][h][3] -= (float)h1[1];
    } else {
      va[a][h][0] -= __builtin_bit_cast(float, ua[a][h][0] << 16);
      va[a][h][1] -= __builtin_bit_cast(float, ua[a][h][0] & 0xffff0000u);
      va[a][h][2] -= __builtin_bit_cast(float, ua[a][h][1] << 16);
      va[a][h][3] -= __builtin_bit_cast(float, ua[a][h][1] & 0xffff0000u);
    }
  };
  auto pack_a = [&](bf16x8 (&fa)[NS][TM]) {
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int q = 0; q < NS; q++) {
        const uint4 u = {oa[a][0][q].x, oa[a][0][q].y, oa[a][1][q].x, oa[a][1][q].y};
        fa[q][a] = __builtin_bit_cast(bf16x8, u);
      }
  };
  constexpr int SPL = 2 * NS - 1;  // split pieces per fragment half: cvt, (sub, cvt) x (NS-1)
  constexpr int M_DMA = NIA + NIB, M_RA = 2 * TM, M_RB = NS * TN, M_SPLIT = 2 * TM * SPL;
  constexpr int NMICRO = M_DMA + M_RA + M_RB + M_SPLIT + 1;
  // idx is a compile-time constant wherever this is called (fully unrolled callers)
  auto micro = [&](int idx, int kt_fill, int stage_free, int stage_next, bf16x8 (&fan)[NS][TM], bf16x8 (&fbn)[NS][TN]) {
    if (idx < NIA) { dma_a(idx, stage_free); return; }
    idx -= NIA;
    if (idx < NIB) { dma_b(idx, kt_fill, stage_free); return; }
    idx -= NIB;
    const char* st = ring + stage_next * STAGE;
    if (idx < M_RA) { va[idx >> 1][idx & 1] = *(const f32x4*)(st + aoff[idx >> 1][idx & 1]); return; }
    idx -= M_RA;
    if (idx < M_RB) { const int q = idx / TN, b = idx % TN; fbn[q][b] = *(const bf16x8*)(st + boff + q * PB + b * 1024); return; }
    idx -= M_RB;
    if (idx < M_SPLIT) {
      // level-major over the 2*TM fragment halves, so that consecutive pieces are independent of each other
      const int lvl = idx / (2 * TM), f = idx % (2 * TM), a = f >> 1, h = f & 1;
      if (lvl & 1) split_sub(a, h); else split_cvt(a, h, lvl >> 1);
      return;
    }
    pack_a(fan);
  };
  auto wait_dma = [&](int steps_in_flight) {  // counted wait: the newest `steps_in_flight` steps of DMAs may stay pending
    static_assert(S >= 3 && S <= 5 && 3 * NI < 64, "vmcnt range");
    if (steps_in_flight <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (steps_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NI) : "memory");
    else if (steps_in_flight == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * NI) : "memory");
  };
  // One pipeline step, branch-free: the tail steps still issue their DMAs (the counted waits stay uniform) but point
  // them at 16 zero bytes, and pre-read a stale buffer, instead of testing for the end.
  auto step = [&](int kt, int stage_next, int stage_free, const bf16x8 (&fa)[NS][TM], const bf16x8 (&fb)[NS][TN],
                  bf16x8 (&fan)[NS][TM], bf16x8 (&fbn)[NS][TN]) {
    // this wave's DMAs of step kt+1 have landed (those of steps kt+2 .. kt+S-1 may stay in flight) and its own fragment
    // reads of step kt are complete, and so are everybody else's: step kt+1 is readable, the buffer of step kt is free
    wait_dma(S - 2);
    __builtin_amdgcn_s_barrier();
    tap_next();
    filling = kt + S < nkt;
    const int kt_fill = min(kt + S, nkt - 1);
    constexpr int NM = TM * TN * (NS * (NS + 1) / 2);
    int j = 0, mi = 0;
#pragma unroll
    for (int sum = NS - 1; sum >= 0; sum--)
#pragma unroll
      for (int qa = 0; qa <= sum; qa++) {
        const int qb = sum - qa;
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
          for (int b = 0; b < TN; b++) {
            if constexpr (F16)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[qa][a]), __builtin_bit_cast(f16x8, fb[qb][b]), acc[a][b], 0, 0, 0);
            else
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
            j++;
#pragma unroll
            for (int r = 0; r < (NMICRO + NM - 1) / NM; r++)
              if (mi < (j * NMICRO + NM - 1) / NM) { micro(mi, kt_fill, stage_free, stage_next, fan, fbn); mi++; }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
  };
#pragma unroll
  for (int t = 0; t < S - 1; t++) { filling = t < nkt; issue_all(min(t, nkt - 1), t); }
  if constexpr (F16) {
    // the range guard sits HERE, behind the prologue's copies: the slot's scalar loads ride in the shadow of the HBM latency the
    // block waits for anyway (at the top of the kernel they cost ~1 us per launch: +0.3 ms of conv time per step, measured)
    if (f16_guard_bad(guard)) {   // this tensor's dynamic range defeats fp16: exact fp32 products for the tile (uniform branch)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the copies in flight land in LDS nobody reads
      float* slab = ksplit > 1 ? ws + ((long)bid * ksplit + ks) * (BM * BN) : nullptr;
      conv_slow_tile(m0, BM, 0, BM, n0, BN, slab, ks != 0, tid, 256, blockIdx.x);
      return;
    }
  }
  wait_dma(S - 2);
  __builtin_amdgcn_s_barrier();
  filling = S - 1 < nkt;
  issue_all(min(S - 1, nkt - 1), S - 1);
  bf16x8 fa0[NS][TM], fb0[NS][TN], fa1[NS][TM], fb1[NS][TN];
  {
    const char* st = ring;
#pragma unroll
    for (int a = 0; a < TM; a++) { va[a][0] = *(const f32x4*)(st + aoff[a][0]); va[a][1] = *(const f32x4*)(st + aoff[a][1]); }
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
      for (int b = 0; b < TN; b++) fb0[q][b] = *(const bf16x8*)(st + boff + q * PB + b * 1024);
#pragma unroll
    for (int lvl = 0; lvl < SPL; lvl++)
#pragma unroll
      for (int f = 0; f < 2 * TM; f++) { if (lvl & 1) split_sub(f >> 1, f & 1); else split_cvt(f >> 1, f & 1, lvl >> 1); }
    pack_a(fa0);
  }
  int stage = 0;  // buffer of step kt
  for (int kt = 0; kt < nkt; kt += 2) {
    const int s1 = stage == S - 1 ? 0 : stage + 1, s2 = s1 == S - 1 ? 0 : s1 + 1;
    step(kt, s1, stage, fa0, fb0, fa1, fb1);
    if (kt + 1 < nkt) step(kt + 1, s2, s1, fa1, fb1, fa0, fb0);
    stage = s2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus DMAs of the tail steps must land before LDS is reused
  __syncthreads();
  if (ksplit == 1) {
    // round 5: straight from the accumulator registers (conv_shared.h: conv_epilogue_direct; bit-identical) wherever an output row's
    // address is linear in the GEMM row -- fp32 tensors, no residual of another resolution, no scatter; else the staged epilogue
    if (!p.staged_epilogue && !p.io && !p.ypl && !p.mul && p.res_mode <= 1 && p.out_stride == 1 && (long)p.M * p.Cout * 4 < (1L << 31)) {
      const int mrow0 = m0 + wm * TM * 32 + 4 * (lane >> 5);
      conv_epilogue_direct<TM, TN>(p, acc, (unsigned)mrow0 * ((unsigned)p.Cout * 4u), p.M - mrow0, n0 + wn * TN * 32 + (lane & 31),
                                   (1 << (TM * TN)) - 1, lds + 64, wave, lane, tid, blockIdx.x);
      return;
    }
    conv_epilogue<BM, BN, WM, WN>(p, acc, lds, m0, n0, tid, lane, wm, wn, HoWo);
    return;
  }
  // split-K: every block parks its partial tile in the workspace; conv_splitk_finish_kernel (next launch on the stream) adds
  // the partials in the fixed order 0..ksplit-1 and runs the epilogue.  (A single-launch form -- last block to arrive at a
  // per-tile counter finishes -- needs device-scope fences, which on this part write back the XCD's L2: 0.14 vs 0.08 ms.)
  conv_epilogue_stage<BM, BN, WM, WN>(acc, lds, lane, wm, wn);
  __syncthreads();
  constexpr int TILE4 = BM * BN / 4;
  f32x4* slab = (f32x4*)ws + ((long)bid * ksplit + ks) * TILE4;
  const f32x4* ct4 = (const f32x4*)lds;
  for (int i = tid; i < TILE4; i += 256) slab[i] = ct4[i];
}

// ------------------------------------------------------------------------------------ split-bf16 forward, both operands as planes
// Same GEMM, same products in the same order as conv_fwd_glds_kernel (bit-identical results), with the ACTIVATIONS
// pre-split as well: x arrives as NS bf16 planes with x's own NHWC indexing (mmt_split_planes, or written by the
// producing convolution's epilogue).  What that buys, from the counters of the kernel above:
//   * a 3x3 convolution re-reads every input value 9 x Cout/128 times; there each read is split again in registers
//     (~45 vector instructions per wave per 16-k step beside 24 MFMAs: with two waves per SIMD that is the issue budget
//     of the MFMA gaps).  Here the main loop has NO vector arithmetic: per step and wave 6 DMA issues, 12 ds_read_b128,
//     24 MFMAs;
//   * without a per-wave split the wave grid is free: 2 x 2 waves of 64 x 64 (12 fragment reads per 24 MFMAs, 14 before).
// A-plane LDS image = the B-plane image: per 32-row block [32 rows][2 halves][8 bf16] with half ^= (row >> 3) & 1; a DMA
// instruction fills one block of one plane (lane = (row, half): 16 B from the row's pixel, channel c0 + 8 * logical half).
template <int BM, int BN, int WM, int WN, int NS, int S>
__global__ __launch_bounds__(256, 2) void conv_fwd_pp_kernel(const ConvP p, const int ksplit, float* __restrict__ ws) {
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int PA = BM * 32, PB = BN * 32;  // bytes of one plane of the A / B tile
  constexpr int A_BYTES = NS * PA;
  constexpr int STAGE = A_BYTES + NS * PB;
  constexpr int NIA_TOT = NS * BM / 32, NIB_TOT = NS * BN / 32;
  constexpr int NIA = (NIA_TOT + 3) / 4, NIB = (NIB_TOT + 3) / 4, NI = NIA + NIB;
  static_assert(BM / 32 == 4 || BM / 32 == 2, "row blocks per wave");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ring = (char*)lds;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = tiles_m * tiles_n * ksplit, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ks = bid % ksplit;
  bid /= ksplit;
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int HoWo = p.Ho * p.Wo;

  // ---- A DMA geometry: instruction t = wave + 4 i -> plane t / (BM/32), row block t % (BM/32); with BM/32 == 4 a wave
  // always fills ITS OWN row block (one pixel per lane pair) in every plane
  const unsigned short* asrc[NIA];
  int adst[NIA];
  int arow_blk[NIA];
#pragma unroll
  for (int i = 0; i < NIA; i++) {
    const int t = (wave + 4 * i) % NIA_TOT;
    const int q = t / (BM / 32), rb = t % (BM / 32);
    asrc[i] = p.xpl + q * p.xpl_stride;
    adst[i] = q * PA + rb * 1024;
    arow_blk[i] = rb;
  }
  // rows: with BM/32 == 4 all of this wave's instructions share one row block; with BM/32 == 2 they alternate (0,1,0,..)
  constexpr int NRB = (BM / 32 == 4) ? 1 : 2;
  unsigned abase[NRB];
  int aih0[NRB], aiw0[NRB];
  bool aok[NRB];
  const int lrow = lane >> 1;
  const int achunk = (((lane & 1) ^ ((lrow >> 3) & 1)) << 3);  // bf16 elements: logical half held by this lane
#pragma unroll
  for (int j = 0; j < NRB; j++) {
    const int rb = (BM / 32 == 4) ? (wave % 4) : ((wave + 4 * j) % NIA_TOT) % (BM / 32);
    const int m = m0 + rb * 32 + lrow;
    aok[j] = m < p.M;
    const int mm = aok[j] ? m : 0;
    const int img = mm / HoWo, rem = mm - img * HoWo;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    aih0[j] = ho * p.stride - p.pad;
    aiw0[j] = wo * p.stride - p.pad;
    abase[j] = (unsigned)img * (unsigned)(p.H * p.W * p.Cin);
  }
  const int nb32 = (p.Cout + 31) >> 5;
  const long kt_stride = (long)nb32 * 512;  // bf16 elements per 16-k step
  const unsigned short* bsrc[NIB];
  int bdst[NIB];
#pragma unroll
  for (int i = 0; i < NIB; i++) {
    const int t = (wave + 4 * i) % NIB_TOT;
    const int q = t / (BN / 32), rb = t % (BN / 32);
    int nb = n0 / 32 + rb;
    if (nb >= nb32) nb = nb32 - 1;
    bsrc[i] = p.wpl + q * p.wpl_stride + (long)nb * 512 + lane * 8;
    bdst[i] = A_BYTES + q * PB + rb * 1024;
  }
  const int nkt_all = p.K >> 4;
  const int kt0 = (int)((long)ks * nkt_all / ksplit), nkt = (int)((long)(ks + 1) * nkt_all / ksplit) - kt0;
  int s_kh, s_kw, s_ci;
  {
    const int k0 = kt0 * 16, tap0 = k0 / p.Cin;
    s_ci = k0 - tap0 * p.Cin;
    s_kh = tap0 / p.KW;
    s_kw = tap0 - s_kh * p.KW;
  }
  bool fresh = true;
  int c_ci = 0;
  unsigned roff[NRB];
  bool rok[NRB];
  const unsigned short* const zsrc = (const unsigned short*)g_zero16;
  auto tap_next = [&]() {
    c_ci = s_ci;
    if (s_ci == 0 || fresh) {
      fresh = false;
#pragma unroll
      for (int j = 0; j < NRB; j++) {
        const int ih = aih0[j] + s_kh, iw = aiw0[j] + s_kw;
        rok[j] = aok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        roff[j] = abase[j] + (unsigned)((ih * p.W + iw) * p.Cin + achunk);
      }
    }
    s_ci += 16;
    if (s_ci >= p.Cin) { s_ci = 0; if (++s_kw == p.KW) { s_kw = 0; ++s_kh; } }
  };
  bool filling = true;
  auto dma_a = [&](int i, int stage) {
    const int j = NRB == 1 ? 0 : (i & 1);
    const unsigned short* src = rok[j] && filling ? asrc[i] + (roff[j] + (unsigned)c_ci) : zsrc;
    dma16(src, ring + stage * STAGE + adst[i]);
  };
  auto dma_b = [&](int i, int kt, int stage) {
    const void* src = filling ? (const void*)(bsrc[i] + (long)(kt0 + kt) * kt_stride) : (const void*)zsrc;
    dma16(src, ring + stage * STAGE + bdst[i]);
  };
  auto issue_all = [&](int kt, int stage) {
    tap_next();
#pragma unroll
    for (int i = 0; i < NIA; i++) dma_a(i, stage);
#pragma unroll
    for (int i = 0; i < NIB; i++) dma_b(i, kt, stage);
  };

  const int lr = lane & 31, kh2 = lane >> 5;
  const int foff = lr * 32 + (((kh2 ^ (lr >> 3)) & 1) << 4);
  const int aoff = (wm * TM) * 1024 + foff;
  const int boff = A_BYTES + (wn * TN) * 1024 + foff;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  constexpr int M_DMA = NIA + NIB, M_RA = NS * TM, M_RB = NS * TN;
  constexpr int NMICRO = M_DMA + M_RA + M_RB;
  auto micro = [&](int idx, int kt_fill, int stage_free, int stage_next, bf16x8 (&fan)[NS][TM], bf16x8 (&fbn)[NS][TN]) {
    if (idx < NIA) { dma_a(idx, stage_free); return; }
    idx -= NIA;
    if (idx < NIB) { dma_b(idx, kt_fill, stage_free); return; }
    idx -= NIB;
    const char* st = ring + stage_next * STAGE;
    if (idx < M_RA) { const int q = idx / TM, a = idx % TM; fan[q][a] = *(const bf16x8*)(st + aoff + q * PA + a * 1024); return; }
    idx -= M_RA;
    { const int q = idx / TN, b = idx % TN; fbn[q][b] = *(const bf16x8*)(st + boff + q * PB + b * 1024); }
  };
  auto wait_dma = [&](int steps_in_flight) {
    static_assert(S >= 3 && S <= 5 && 3 * NI < 64, "vmcnt range");
    if (steps_in_flight <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (steps_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NI) : "memory");
    else if (steps_in_flight == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * NI) : "memory");
  };
  auto step = [&](int kt, int stage_next, int stage_free, const bf16x8 (&fa)[NS][TM], const bf16x8 (&fb)[NS][TN],
                  bf16x8 (&fan)[NS][TM], bf16x8 (&fbn)[NS][TN]) {
    wait_dma(S - 2);
    __builtin_amdgcn_s_barrier();
    tap_next();
    filling = kt + S < nkt;
    const int kt_fill = min(kt + S, nkt - 1);
    constexpr int NM = TM * TN * (NS * (NS + 1) / 2);
    int j = 0, mi = 0;
#pragma unroll
    for (int sum = NS - 1; sum >= 0; sum--)
#pragma unroll
      for (int qa = 0; qa <= sum; qa++) {
        const int qb = sum - qa;
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
          for (int b = 0; b < TN; b++) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
            j++;
#pragma unroll
            for (int r = 0; r < (NMICRO + NM - 1) / NM; r++)
              if (mi < (j * NMICRO + NM - 1) / NM) { micro(mi, kt_fill, stage_free, stage_next, fan, fbn); mi++; }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
  };
#pragma unroll
  for (int t = 0; t < S - 1; t++) { filling = t < nkt; issue_all(min(t, nkt - 1), t); }
  wait_dma(S - 2);
  __builtin_amdgcn_s_barrier();
  filling = S - 1 < nkt;
  issue_all(min(S - 1, nkt - 1), S - 1);
  bf16x8 fa0[NS][TM], fb0[NS][TN], fa1[NS][TM], fb1[NS][TN];
  {
    const char* st = ring;
#pragma unroll
    for (int q = 0; q < NS; q++) {
#pragma unroll
      for (int a = 0; a < TM; a++) fa0[q][a] = *(const bf16x8*)(st + aoff + q * PA + a * 1024);
#pragma unroll
      for (int b = 0; b < TN; b++) fb0[q][b] = *(const bf16x8*)(st + boff + q * PB + b * 1024);
    }
  }
  int stage = 0;
  for (int kt = 0; kt < nkt; kt += 2) {
    const int s1 = stage == S - 1 ? 0 : stage + 1, s2 = s1 == S - 1 ? 0 : s1 + 1;
    step(kt, s1, stage, fa0, fb0, fa1, fb1);
    if (kt + 1 < nkt) step(kt + 1, s2, s1, fa1, fb1, fa0, fb0);
    stage = s2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ksplit == 1) {
    conv_epilogue<BM, BN, WM, WN>(p, acc, lds, m0, n0, tid, lane, wm, wn, HoWo);
    return;
  }
  conv_epilogue_stage<BM, BN, WM, WN>(acc, lds, lane, wm, wn);
  __syncthreads();
  constexpr int TILE4 = BM * BN / 4;
  f32x4* slab = (f32x4*)ws + ((long)bid * ksplit + ks) * TILE4;
  const f32x4* ct4 = (const f32x4*)lds;
  for (int i = tid; i < TILE4; i += 256) slab[i] = ct4[i];
}

// ------------------------------------------------------------------------------------ 3x3 / stride 1 / pad 1, all planes, tap strips
// What the experiments on conv_fwd_pp_kernel showed (tools/bench_pp_dbg.py, FPN 3x3 N=8): without its DMAs the main loop
// takes 2.3 ms, with them 3.3 ms, and the cost is linear in the BYTES moved L2 -> LDS (24 KB per 128x128x16 step), not in
// their issue pattern; fragment reads and barriers are free.  So this kernel moves fewer bytes per multiply:
//   * the three horizontal taps (kw = 0, 1, 2) of one (kh, 16-channel slab) read the SAME input pixels shifted by one: a
//     strip of TW + 2 pixels per image row is copied once and the three taps read it at row offsets 0, 1, 2
//     (the half-swizzle of the plane image, half ^= (row >> 3) & 1 on absolute strip rows, stays conflict-free for every
//     shift) -- A traffic / 2.4;
//   * tile = 256 output pixels (R = 256 / TW image rows x TW pixels) x 128 channels on EIGHT waves (2 per SIMD, 64 x 64
//     each): the weight planes of a step feed twice the pixels -- B traffic per multiply / 2.
//   -> 11 KB per 128x128x16 step-equivalent instead of 24.
// K order: (group of G channel slabs, kh, slab of the group, kw) since round 6 ((kh, channel slab, kw) before = one group of all
// slabs; G in bits 24-30 of the ksplit argument) -- the packed weight
// planes are indexed, not re-packed.  One barrier per super-step
// (three taps = 72 MFMAs per wave), placed before the last tap: the fragments of that tap are in registers by then, so the
// stage is free for the copies of super-step s + 2 while tap 0 of s + 1 is pre-read from the other stage.
template <int TW, int NS, bool F16 = false>  // F16: the planes hold the two fp16 terms of x * s_x / w * s_w (default arithmetic of mode 3)
__global__ __launch_bounds__(512, 2) void conv3x3_strip_kernel(const ConvP p, const int ksplit_arg, float* __restrict__ ws) {
  // bits 24-30 of the argument = G: super-steps in (group of G channel slabs, kh, slab of the group) order; 0 = (kh, slab), the order of
  // rounds 2-5.  The three kh of a slab read the same input rows shifted by one: with kh outermost the whole input streams through the
  // L2 three times (FETCH_SIZE of this kernel was 3 x its input), with a small group outermost a row is fetched once and found in
  // the L2 twice, 3 G super-steps apart at most (profiles/r06_history.md section 9)
  const int ksplit = ksplit_arg & 0xffffff;
  const int korder_g = (ksplit_arg >> 24) & 127;
  constexpr int BM = 256, BN = 128, R = BM / TW, SW = TW + 32, TM = 2, TN = 2;
  constexpr int PA = R * SW * 32;              // bytes of one A plane of a stage (strip rows x 32 B)
  constexpr int PB = BN * 32;                  // one B plane of one tap
  constexpr int A_BYTES = NS * PA, B_TAP = NS * PB, STAGE = A_BYTES + 3 * B_TAP;
  constexpr int NA = R * (SW / 32) * NS, NB = 3 * (BN / 32) * NS, NSLOT = (NA + 7) / 8 + (NB + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ring = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;     // 4 x 2 waves of 64 x 64

  const int tiles_w = p.Wo / TW, tiles_h = p.Ho / R, tiles_n = (p.Cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // split-K (few tiles: the N = 2 passes on the 64^2 / 128^2 maps): the ksplit blocks of a tile are neighbours and take
  // consecutive, even-sized ranges of super-steps; partial tiles go to the workspace (host: only when Wo == TW, where a
  // tile's 256 pixels are consecutive in memory and conv_splitk_finish_kernel<256, 128> finishes them)
  const int ks = bid % ksplit;
  bid /= ksplit;
  const int tile_lin = bid;
  const int tile_n = bid % tiles_n;
  int t = bid / tiles_n;
  const int tw = t % tiles_w; t /= tiles_w;
  const int th = t % tiles_h, img = t / tiles_h;
  const int ho0 = th * R, wo0 = tw * TW, n0 = tile_n * BN;
  const int slabs = p.Cin >> 4;                      // super-steps: (kh, slab), 3 * slabs of them
  const int pairs = (3 * slabs) >> 1;                // ... handed out in pairs (the loop body is two super-steps)
  const int ss0 = 2 * (int)((long)ks * pairs / ksplit), nss = 2 * (int)((long)(ks + 1) * pairs / ksplit) - ss0;
  F16Guard guard = {};
  if constexpr (F16) guard = f16_guard_load(p.guard_x);   // issued here, tested behind the first copies (see conv_fwd_glds_kernel)
  const float s_lag = (F16 && p.xpl_lag) ? *p.f16_sx : 0.f;   // (round 6) planes written by the producer with a scale fixed beforehand

  // ---- copy slots of this wave.  Slots 0 .. SA-1 carry A items (wave + 8 i < NA: plane, strip row r, 32-pixel block),
  // slots SA .. SA+SB-1 B items (tap, plane, 32-channel block).  Everything that does not change from super-step to
  // super-step is computed here: a copy in the loop is one buffer_load ... lds with a per-lane VGPR offset fixed for the
  // whole kernel and a scalar offset per super-step; lanes outside the image (and strip pixels past TW + 2) carry an
  // offset beyond the descriptor's range and read zeros -- no address arithmetic, no predication in the loop.
  constexpr int SA = (NA + 7) / 8, SB = (NB + 7) / 8;
  constexpr unsigned OOB = 0x80000000u;
  const int nb32 = (p.Cout + 31) >> 5;
  const long kt_stride = (long)nb32 * 512;
  const int lpx = lane >> 1;
  const int achunk = (((lane & 1) ^ ((lpx >> 3) & 1)) << 3);
  const long n_x = (long)p.N * p.H * p.W * p.Cin;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.xpl, 0, (int)(unsigned)(((NS - 1) * p.xpl_stride + n_x) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpl, 0, 0x7ffffff0, 0x00020000);
  unsigned voff[SA + SB];   // per-lane byte offset of the slot (OOB = zeros)
  int sdst[SA + SB];        // LDS byte offset inside a stage
  int srow[SA];             // A: strip row r (decides with kh whether the input row exists); -1 = no item
#pragma unroll
  for (int i = 0; i < SA; i++) {
    const int item = wave + 8 * i;
    srow[i] = -1; voff[i] = OOB; sdst[i] = 0;
    if (item < NA) {
      const int q = item / (R * (SW / 32)), rem = item % (R * (SW / 32)), r = rem / (SW / 32), blk = rem % (SW / 32);
      const int px = blk * 32 + lpx, iw = wo0 - 1 + px;
      const bool ok = px < TW + 2 && (unsigned)iw < (unsigned)p.W;
      // input row ho0 + r + kh - 1: the (kh - 1) rows and the channel slab go into the scalar offset
      // (row-blocked planes [N H][Cin / 16][W][16], p.xpl_rb: the 32 pixels of an item are one run of 1 KiB instead of 32 pieces of
      // 32 bytes in 32 cache lines; the channel slab moves by W x 32 bytes in the scalar offset)
      const long e = p.xpl_rb ? q * p.xpl_stride + ((long)(img * p.H + ho0 + r) * slabs * p.W + iw) * 16 + achunk
                              : q * p.xpl_stride + ((long)(img * p.H + ho0 + r) * p.W + iw) * p.Cin + achunk;
      voff[i] = ok ? (unsigned)(e * 2) : OOB;
      sdst[i] = q * PA + (r * SW + blk * 32) * 32;
      srow[i] = r;
    }
  }
#pragma unroll
  for (int i = 0; i < SB; i++) {
    const int item = wave + 8 * i;
    voff[SA + i] = OOB; sdst[SA + i] = -1;
    if (item < NB) {
      const int kw = item / ((BN / 32) * NS), rem = item % ((BN / 32) * NS), q = rem / (BN / 32), rb = rem % (BN / 32);
      int nb = n0 / 32 + rb;
      if (nb >= nb32) nb = nb32 - 1;
      voff[SA + i] = (unsigned)((q * p.wpl_stride + (long)kw * slabs * kt_stride + (long)nb * 512 + lane * 8) * 2);
      sdst[SA + i] = A_BYTES + kw * B_TAP + q * PB + rb * 1024;
    }
  }
  // per input-row offset kh the A slots' VGPR offsets are fixed (a slot whose input row does not exist reads zeros);
  // recomputed when the super-step being FILLED enters a new kh (three times per kernel), never inside the slab loop
  const int row_bytes = p.W * p.Cin * 2;
  const int a_step = p.xpl_rb ? p.W * 32 : 32;   // bytes from one 16-channel slab to the next
  unsigned vo_a[SA];
  const int G = (korder_g == 0 || korder_g > slabs) ? slabs : korder_g;   // (slabs % G == 0: the host picks a power of two <= 8, slabs is a multiple of 8)
  int f_grp = ss0 / (3 * G), f_kh = (ss0 % (3 * G)) / G, f_sub = ss0 % G;   // super-step ss0 = (group, kh, slab of the group)
  int f_cs = f_grp * G + f_sub;                 // (kh, slab) of the super-step whose copies are being issued
  const int b_step = (int)(kt_stride * 2);
  int soff_a = 0;                               // its scalar byte offsets: A = max(kh - 1, 0) rows + slab,
  int soff_b = (f_kh * 3 * slabs + f_cs) * b_step;   // B = (kh * 3 * slabs + slab) steps
  auto fill_enter_kh = [&]() {
#pragma unroll
    for (int i = 0; i < SA; i++) {
      const bool rowok = srow[i] >= 0 && (unsigned)(ho0 + srow[i] + f_kh - 1) < (unsigned)p.H;
      // voff addresses input row ho0 + r; kh = 0 wants the row above (offsets are unsigned: subtract here, add in soff else)
      vo_a[i] = (rowok && voff[i] != OOB) ? (f_kh == 0 ? voff[i] - (unsigned)row_bytes : voff[i]) : OOB;
    }
    soff_a = (f_kh >= 1 ? (f_kh - 1) * row_bytes : 0) + f_cs * a_step;
  };
  fill_enter_kh();
  auto fill_advance = [&]() {      // next super-step to fill
    if (++f_sub < G) {             // the next slab of the group, same kh
      ++f_cs;
      soff_a += a_step;
      soff_b += b_step;
      return;
    }
    f_sub = 0;
    if (++f_kh == 3) { f_kh = 0; ++f_grp; }
    f_cs = f_grp * G;
    soff_b = (f_kh * 3 * slabs + f_cs) * b_step;   // (the weight steps of kw = 1, 2 of a kh lie between those of kw = 0 of two kh)
    fill_enter_kh();
  };
  auto issue_slot = [&](int i, int stage) {
    char* const st = ring + stage * STAGE;
    if (i < SA) {
      if (srow[i] < 0) return;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(st + sdst[i]), 16, (int)vo_a[i], soff_a, 0, 0);
    } else if (i < SA + SB) {
      if (sdst[i] < 0) return;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(st + sdst[i]), 16, (int)voff[i], soff_b, 0, 0);
    }
  };

  // ---- fragment read addresses, fixed for the kernel (per stage: the stage is a literal at every call site): A row =
  // strip row of (image row, pixel) + kw, the half-swizzle follows the ABSOLUTE strip row; B = plane image of the tap
  const int lr = lane & 31, kh2 = lane >> 5;
  const char* aaddr[2][3][TM];
  const char* baddr[2][3];
#pragma unroll
  for (int sg = 0; sg < 2; sg++)
#pragma unroll
    for (int kw = 0; kw < 3; kw++) {
#pragma unroll
      for (int a = 0; a < TM; a++) {
        const int mrow = wm * 64 + a * 32 + lr;
        const int row = (mrow / TW) * SW + (mrow % TW) + kw;
        aaddr[sg][kw][a] = ring + sg * STAGE + row * 32 + (((kh2 ^ (row >> 3)) & 1) << 4);
      }
      baddr[sg][kw] = ring + sg * STAGE + A_BYTES + kw * B_TAP + (wn * TN) * 1024 + lr * 32 + (((kh2 ^ (lr >> 3)) & 1) << 4);
    }
  auto read_tap = [&](int sg, int kw, bf16x8 (&fa)[NS][TM], bf16x8 (&fb)[NS][TN], int idx) {
    // idx-th fragment read of the tap (0 .. NS*(TM+TN) - 1), dealt out behind individual MFMAs in the order the next tap
    // consumes them: its first products are a0*b2, then a1*b1, a2*b0, so (a plane g, b plane NS-1-g) for g = 0, 1, ..
    const int g = idx / (TM + TN), k = idx % (TM + TN);
    if (k < TM) {
      fa[g][k] = *(const bf16x8*)(aaddr[sg][kw][k] + g * PA);
    } else {
      const int q = NS - 1 - g, b = k - TM;
      fb[q][b] = *(const bf16x8*)(baddr[sg][kw] + q * PB + b * 1024);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  constexpr int NM = TM * TN * (NS * (NS + 1) / 2), NF = NS * (TM + TN);
  constexpr int NRD = NM * 2 / 3;  // the reads of the next tap ride behind the first two thirds of the MFMAs: the last one
                                   // has a third of the tap (~250 cycles) to land before the next tap's first MFMA wants it
  // one tap: NM MFMAs on (fa, fb); behind them the NF fragment reads of the NEXT tap (into fan / fbn, from stage sg_next) and
  // the copy slots [dma0, dma0 + ndma) of the super-step being filled into stage `stage_fill`
  auto tap = [&](const bf16x8 (&fa)[NS][TM], const bf16x8 (&fb)[NS][TN], bf16x8 (&fan)[NS][TM], bf16x8 (&fbn)[NS][TN],
                 int sg_next, int kw_next, bool do_read, int dma0, int ndma, int stage_fill, bool do_dma) {
    int j = 0, ri = 0, di = 0;
#pragma unroll
    for (int sum = NS - 1; sum >= 0; sum--)
#pragma unroll
      for (int qa = 0; qa <= sum; qa++) {
        const int qb = sum - qa;
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
          for (int b = 0; b < TN; b++) {
            if constexpr (F16)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[qa][a]), __builtin_bit_cast(f16x8, fb[qb][b]), acc[a][b], 0, 0, 0);
            else
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
            j++;
            if (ri < (j * NF + NRD - 1) / NRD && ri < NF) { if (do_read) read_tap(sg_next, kw_next, fan, fbn, ri); ri++; }
            if (di < (j * ndma + NM - 1) / NM && di < ndma) { if (do_dma) issue_slot(dma0 + di, stage_fill); di++; }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
  };

  // ---- prologue: stage 0 <- super-step 0, stage 1 <- super-step 1, fragments of tap 0
#pragma unroll
  for (int i = 0; i < NSLOT; i++) issue_slot(i, 0);
  fill_advance();
  if constexpr (F16) {   // the range guard, in the shadow of the first copies' latency (see conv_fwd_glds_kernel)
    if (p.xpl_lag ? f16_guard_bad_lag(guard, s_lag) : f16_guard_bad(guard)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float* slab = ksplit > 1 ? ws + ((long)tile_lin * ksplit + ks) * (BM * BN) : nullptr;
      conv_slow_tile((img * p.Ho + ho0) * p.Wo + wo0, TW, p.Wo, BM, n0, BN, slab, ks != 0, tid, 512, blockIdx.x);
      return;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < NSLOT; i++) issue_slot(i, 1);
  fill_advance();                  // the fill state now describes super-step 2
  bf16x8 fa0[NS][TM], fb0[NS][TN], fa1[NS][TM], fb1[NS][TN];
#pragma unroll
  for (int i = 0; i < NF; i++) read_tap(0, 0, fa0, fb0, i);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  constexpr int D0 = (NSLOT + 1) / 2, D1 = NSLOT - D0;   // copy slots issued behind tap 2 / behind tap 0 of the next super-step
  // one super-step in stage SG (a literal at the call sites); P / Q = register sets, P holds its tap 0.  The copies of
  // super-step ss + 2 go out behind tap 2 of ss (first half) and tap 0 of ss + 1 (second half), into this same stage
  auto superstep = [&](int ss, int SG, bf16x8 (&fa_p)[NS][TM], bf16x8 (&fb_p)[NS][TN], bf16x8 (&fa_q)[NS][TM], bf16x8 (&fb_q)[NS][TN]) {
    // tap 0 (regs p) | pre-read tap 1 -> q | second half of the copies of super-step ss + 1 (into the other stage)
    const bool second = ss >= 1 && ss + 1 < nss;
    tap(fa_p, fb_p, fa_q, fb_q, SG, 1, true, D0, D1, SG ^ 1, second);
    if (second) fill_advance();
    // tap 1 (regs q) | pre-read tap 2 -> p
    tap(fa_q, fb_q, fa_p, fb_p, SG, 2, true, 0, 0, 0, false);
    // every copy into the other stage has landed, everybody's reads of this stage are complete
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // tap 2 (regs p) | pre-read tap 0 of ss + 1 -> q from the other stage | first half of the copies of ss + 2 into this stage
    // (after the last super-step this pre-read fetches stale bytes nobody uses: unconditional, because a branch around
    // register-producing reads makes the compiler re-pack every bf16 fragment register at the join -- 600 v_perm / v_lshr
    // per loop body, measured)
    tap(fa_p, fb_p, fa_q, fb_q, SG ^ 1, 0, true, 0, D0, SG, ss + 2 < nss);
  };
  for (int ss = 0; ss < nss; ss += 2) {
    superstep(ss, 0, fa0, fb0, fa1, fb1);
    superstep(ss + 1, 1, fa1, fb1, fa0, fb0);   // nss is even (Cin % 32 == 0)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- epilogue, round 5: straight from the accumulator registers (conv_shared.h: conv_epilogue_direct) for the un-split form on
  // fp32 tensors -- a wave's 64 tile rows are 64 consecutive pixels of one image row (TW >= 64).  The staged form below (tile ->
  // LDS -> float4 rows, whose residual / mask loads are waited for one by one) took 18 - 30 us per tile with the matrix pipe idle
  // (profiles/r04_history.md); same expressions, bit-identical outputs.
  static_assert(TW >= 64, "a wave's rows lie in one image row");
  if (ksplit == 1 && !p.staged_epilogue && !p.io && !p.ypl && !p.mul && p.res_mode <= 1 && (long)p.M * p.Cout * 4 < (1L << 31)) {
    const int mrow = wm * 64;
    const unsigned pix = (unsigned)((img * p.Ho + ho0 + mrow / TW) * p.Wo + wo0 + mrow % TW + 4 * (lane >> 5));
    conv_epilogue_direct<2, 2>(p, acc, pix * ((unsigned)p.Cout * 4u), 1 << 30, n0 + wn * 64 + (lane & 31), 15, (float*)lds + 64, wave,
                               lane, tid, blockIdx.x);
    return;
  }
  // ---- epilogue: the 256 x 128 tile as two 128-row halves, each finished by 256 threads with the shared row epilogue.
  // Tile row m (0..255) = output pixel (ho0 + m / TW, wo0 + m % TW); the row epilogue wants the linear pixel index, which
  // is contiguous only inside one image row, so each 128-row half is handed over in runs of min(TW, 128) rows
  conv_epilogue_stage<BM, BN, 4, 2>(acc, lds, lane, wm, wn);
  __syncthreads();
  if (ksplit > 1) {
    constexpr int TILE4 = BM * BN / 4;
    f32x4* slab = (f32x4*)ws + ((long)tile_lin * ksplit + ks) * TILE4;
    const f32x4* ct4 = (const f32x4*)lds;
    for (int i = tid; i < TILE4; i += 512) slab[i] = ct4[i];
    return;
  }
  {
    const int half = tid >> 8, t2 = tid & 255;
    constexpr int RUN = TW < 128 ? TW : 128;
    AmaxAcc st;
#pragma unroll
    for (int run = 0; run < 128 / RUN; run++) {
      const int mrow = half * 128 + run * RUN;
      const int m_lin = (img * p.Ho + ho0 + mrow / TW) * p.Wo + wo0 + mrow % TW;
      conv_epilogue_finish<RUN, BN>(p, lds + mrow * BN, m_lin, n0, t2, p.Ho * p.Wo, &st);
    }
    if (p.amax_out) conv_amax_commit(p, st, blockIdx.x);
  }
}

// one block per 1/PARTS of a tile (BM / PARTS rows), PARTS = 4 (quarter tiles: four times the blocks of the main launch's tile
// count).  The kernel is a pure memory pass of ksplit x tile reads: 21 us for 40 MB on the student's 64-tile 3x3 layers at
// N = 2.  More, smaller blocks do NOT help (MMT_FINISH_PARTS=16: 33.6 us -- 8 KB instead of 32 KB runs per slab)
template <int BM, int BN, int PARTS>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvP p, const int ksplit, const float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int QM = BM / PARTS;
  const int tid = threadIdx.x;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int bid = blockIdx.x / PARTS, part = blockIdx.x % PARTS;
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  constexpr int TILE4 = BM * BN / 4, PART4 = QM * BN / 4, U = PART4 / 256 < 4 ? PART4 / 256 : 4;
  static_assert(PART4 % (256 * U) == 0 && U >= 1, "tile size");
  const f32x4* base = (const f32x4*)ws + (long)bid * ksplit * TILE4 + part * PART4;
  for (int i0 = tid; i0 < PART4; i0 += 256 * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = base[i0 + u * 256];
    // slabs four at a time: all loads of a group are in flight before the first add; the order of the additions is unchanged
    for (int j = 1; j < ksplit; j += 4) {
      f32x4 t[4][U];
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int u = 0; u < U; u++)
          t[jj][u] = j + jj < ksplit ? base[(long)(j + jj) * TILE4 + i0 + u * 256] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
        if (j + jj < ksplit) {
#pragma unroll
          for (int u = 0; u < U; u++) v[u] += t[jj][u];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) ((f32x4*)lds)[i0 + u * 256] = v[u];
  }
  __syncthreads();
  conv_epilogue_finish<QM, BN>(p, lds, tile_m * BM + part * QM, tile_n * BN, tid, p.Ho * p.Wo);
}

template <int BM, int BN>
static void launch_finish(const ConvP& p, int tiles, int ksplit, const float* ws, hipStream_t s) {
  constexpr int parts = 4;   // (tuned: profiles/r04_dispatch_sweep.txt)
  if (parts == 16)
    hipLaunchKernelGGL((conv_splitk_finish_kernel<BM, BN, 16>), dim3(tiles * 16), dim3(256), (size_t)(BM / 16) * BN * 4, s, p, ksplit, ws);
  else
    hipLaunchKernelGGL((conv_splitk_finish_kernel<BM, BN, 4>), dim3(tiles * 4), dim3(256), (size_t)(BM / 4) * BN * 4, s, p, ksplit, ws);
}

// Weight packing for the DMA-fed kernels.  For a weight matrix [Cout][K] (K % 16 == 0) plane q of the packed form is
//   [K/16 steps][ceil(Cout/32) blocks][32 rows][2 halves][8 bf16]
// where (row r, physical half h) holds the q-th bf16 term of w[32 blk + r][16 step + 8 (h ^ ((r>>3)&1)) + 0..7]
// (zeros for rows >= Cout).  One wave produces one 1 KiB unit of each plane: lane = (row, half).

// ------------------------------------------------------------------------- 1x1 convolutions with K = 64 / 128 / 256
// The expanding 1x1 layers of layer1 / layer2 (64 -> 256, 128 -> 512: output + residual dominate, 2-6 FLOP/B) are not
// matrix-bound and not HBM-bound in the tiled kernel above but INSTRUCTION-bound: a block owns one 128 x 64 tile with
// only 4-8 k-steps, so its ~600-instruction prologue (row decode, ring fill, first split), the per-step bookkeeping
// and the epilogue are paid per 48 MFMAs (measured: 1230 vector + 630 scalar instructions per wave per tile).
// Here a block owns 128 ROWS and walks across Cout in panels of 32 columns:
//   * each wave loads its 32 rows of A straight into registers (no LDS) and splits them into bf16 / fp16 fragments ONCE
//     for the whole K and for every panel (instead of once per panel);
//   * the pre-split weight planes of panel nt+1 are DMA-copied into the other LDS buffer while panel nt is multiplied;
//   * per panel: KT x 6 (x 3 on the fp16 split) MFMAs from registers / LDS fragments, then the epilogue straight from the
//     accumulator registers.
// Arithmetic and summation order are those of conv_fwd_glds_kernel (bit-identical results).
// F16 (round 3): the two-term fp16 split of the default arithmetic -- A scaled by the power of two of its recorded maximum
// (p.f16_sx -> max |x|) and split ONCE into (h, l) fragments that stay in registers for every panel; weight planes are the
// fp16 pair of the packed weight (scale *p.f16_sw); 3 products per multiply; the epilogue divides by both scales.  With that,
// K = 256 fits too (KT = 16: 128 fragment registers per wave), i.e. the expanding 1x1 layers of layer3 (256 -> 1024), the
// FPN lateral of C2, the hint adaptors and the data gradients of every 1x1 layer with 256 output channels.  In the tiled
// kernel those re-read the raw fp32 rows once per 64-column tile through the L2 -> LDS path (256 -> 1024 at 8 x 64 x 64:
// 786 MB for 17 GFLOP, 170 us); here a row is read once per column group.  blockIdx.y = column group (`ppb` panels each):
// the few-tile student shapes (N = 2: 64 row blocks) still fill the chip.
// Epilogue (round 4; plain / residual-add / masked output): an accumulator register of the 32 x 32 tile is one output row x
// 32 consecutive channels per half wave, so every store / residual / mask instruction of a wave touches two complete
// 128-byte lines -- nothing is staged through LDS and the weight ring's barrier is the only one.  What bounds these layers
// is the number of residual bytes a CU keeps in flight (8 waves; 2 - 3 us of loaded HBM latency), so the residual / mask values
// of LATER panels are requested inside the epilogue of panel nt, each into the register its predecessor was just consumed
// from: LA = 2 panels ahead (two register sets, the loop unrolled by two), one panel ahead where the mask values need the
// registers (MASK with K = 256).  They fly through the stores, the barrier and the matrix phases in between.  The staged
// epilogue of round 3 (tile -> LDS -> float4 rows, two more barriers per panel, requests in flight during the matrix phase
// only) ran 256 -> 1024 at 8 x 64 x 64 in 128 us; this one in 87 us (profiles/r04_rows_epilogue.txt).
// Every request is issued unconditionally (an absent operand is a zero-sized buffer: the load returns 0 without touching
// memory): no branches in the loop and a fixed number of vector memory operations per panel, which the counted wait for
// the weight planes at the end of an iteration relies on.
template <int KT, int NS, bool F16 = false, bool MASK = false>
__global__ __launch_bounds__(256, 2) void conv1x1_rows_kernel(const ConvP p, const unsigned short* __restrict__ wpl,
                                                              const long wpl_stride, const int ppb) {
  constexpr int BN = 32;
  constexpr int PIECES = KT * NS;            // 1 KiB DMA pieces per panel: [kt][plane]
  constexpr int BBUF = PIECES * 1024;
  constexpr int LA = (MASK && KT == 16) ? 1 : 2;   // panels of look-ahead of the residual / mask requests
  static_assert(PIECES % 4 == 0, "pieces per wave");
  static_assert(!F16 || NS == 2, "fp16 split: two terms");
  static_assert(!MASK || F16, "masked output: fp16 split only");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const bring = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, kh2 = lane >> 5;
  const int tiles_m = (p.M + 127) >> 7;
  int bid = blockIdx.x;
  {  // each XCD walks its own contiguous range of rows
    const int q = tiles_m >> 3, r = tiles_m & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = bid * 128;
  const int K = 16 * KT;
  const int nb32 = (p.Cout + 31) >> 5;
  const int nt_lo = blockIdx.y * ppb, nt_hi = min(nb32, nt_lo + ppb);
  F16Guard guard = {};
  if constexpr (F16) guard = f16_guard_load(p.guard_x);   // issued here, tested behind the first weight copies
  auto issue_b = [&](int nt, int buf) {
#pragma unroll
    for (int i = 0; i < PIECES / 4; i++) {
      const int piece = wave + 4 * i;  // wave-uniform
      const int kt = piece / NS, q = piece % NS;
      dma16_unseen(wpl + q * wpl_stride + ((long)kt * nb32 + nt) * 512 + lane * 8, bring + buf * BBUF + piece * 1024);
    }
  };
  // Panel order: block `bid` starts `bid % cnt` panels into its column group and wraps around.  All blocks start together and
  // run in step; walking the panels in the same order they would all touch the same 128 bytes of their 4 KiB output rows
  // (Cout = 1024) at the same time, i.e. the same few memory channels.
  const int cnt = nt_hi - nt_lo;
  if (cnt <= 0) return;
  const int rot = bid % cnt;
  auto pn = [&](int i) {   // i-th panel of this block; past the end: none
    int t = i + rot;
    if (t >= cnt) t -= cnt;
    return i < cnt ? nt_lo + t : -1;
  };
  issue_b(pn(0), 0);
  if constexpr (F16) {   // the range guard, in the shadow of the first weight copies (see conv_fwd_glds_kernel)
    if (f16_guard_bad(guard)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (nt_lo < nt_hi) conv_slow_tile(m0, 128, 0, 128, nt_lo * BN, (nt_hi - nt_lo) * BN, nullptr, false, tid, 256,
                                        blockIdx.x + gridDim.x * blockIdx.y);
      return;
    }
  }
  // ---- A: rows m0 + 32 wave + lr, k = 16 kt + 8 kh2 .. + 7 per step; rows past M read zeros (buffer bounds)
  uint4 fa[KT][NS];   // bf16x8 / f16x8 fragments as raw words
  const float sx = F16 ? f16_scale_of(*p.f16_sx) : 1.f;
  const int boff = lr * 32 + (((kh2 ^ (lr >> 3)) & 1) << 4);
  const int col_l = lane & 31, rq = lane >> 5;
  const unsigned cout4 = (unsigned)p.Cout * 4u;
  const int mlane = m0 + wave * 32 + 4 * rq;            // row of accumulator register 0; register r: + 8 (r / 4) + r % 4
  const unsigned ylane = (unsigned)mlane * cout4;       // rows past M: beyond the buffers' bounds (loads 0, stores dropped)
  // residual rows of the four 8-row groups: same pixel (mode 1) or the nearest-x2 upsampled coarser map (mode 2: the FPN
  // top-down add, backbone/fpn.py:57-62; Wo % 8 == 0 -- the launcher checks -- so a group lies in one image row)
  unsigned rbase[4];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    rbase[g] = ylane + (unsigned)(8 * g) * cout4;
    if (p.res_mode == 2) {
      const int m = m0 + wave * 32 + 8 * g, HoWo = p.Ho * p.Wo;
      const int img = m / HoWo, rem = m - img * HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      rbase[g] = (unsigned)((img * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1) + 2 * rq) * cout4;
    }
  }
  const int res_rows = p.res_mode == 2 ? p.N * (p.Ho >> 1) * (p.Wo >> 1) : p.M;
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res_mode ? p.res : p.x), 0, p.res_mode ? (int)((long)res_rows * p.Cout * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.mask ? p.mask : p.x), 0, p.mask ? (int)((long)p.M * p.Cout * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)((long)p.M * p.Cout * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.scale ? p.scale : p.x), 0, p.scale ? p.Cout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.shift ? p.shift : p.x), 0, p.shift ? p.Cout * 4 : 0, 0x00020000);
  const bool up = p.res_mode == 2, has_res = p.res_mode != 0, has_scale = p.scale != nullptr;
  const float finv = F16 ? 1.f / (sx * *p.f16_sw) : 1.f;
  // round 6: y also as row-blocked fp16 planes (p.yrb; conv_shared.h: quad_transpose).  After the 4 x 4 transpose inside a quad of
  // lanes, lane k holds the quad's four channels of row 8 j + k of register group j: this lane's four rows, fixed for the block
  const bool rb = F16 && p.yrb != nullptr;
  const float rbs = rb ? *p.yrb_s : 1.f;
  unsigned rboff[4];
  {
    const RbGeom rbg = rb_geom(p);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int m = mlane + 8 * j + (lane & 3);
      rboff[j] = (rb && m < p.yrb_M) ? rb_row_off(rbg, m) : 0x80000000u;
    }
  }
  const int rb_bytes = rb ? (int)((long)p.yrb_M * p.Cout * 2) : 0;
  const __amdgpu_buffer_rsrc_t rrb0 = __builtin_amdgcn_make_buffer_rsrc((void*)(rb ? p.yrb : (unsigned short*)p.y), 0, rb_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrb1 = __builtin_amdgcn_make_buffer_rsrc((void*)(rb ? p.yrb + p.yrb_stride : (unsigned short*)p.y), 0, rb_bytes, 0x00020000);
  const unsigned rb_cstep = (unsigned)p.Wo * 32u;   // bytes from one 16-channel block of an image row to the next
  float amx = 0.f, asum = 0.f, acnt = 0.f;   // max |y| / sum |y| / count over what this thread stores (p.amax_out)
  float ur[LA][16], um[MASK ? LA : 1][16], scn[LA], shn[LA];
  auto row_off = [&](int r) { return (unsigned)(8 * (r >> 2) + (r & 3)) * cout4; };
  auto res_off = [&](int r) { return rbase[r >> 2] + (unsigned)(up ? (r & 3) >> 1 : (r & 3)) * cout4; };
  auto col_bytes = [&](int nt) {   // byte offset of this lane's channel in panel nt; out of range -> beyond every bound
    const int c = nt * BN + col_l;
    return (nt >= 0 && c < p.Cout) ? (unsigned)c * 4u : 0x80000000u;
  };
  auto request = [&](auto slot, int r, unsigned cb) {
    constexpr int S = decltype(slot)::value;
    ur[S][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)(res_off(r) + cb), 0, 0));
    if constexpr (MASK)
      um[S][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, (int)(ylane + row_off(r) + cb), 0, 0));
  };
  auto request_affine = [&](auto slot, unsigned cb) {
    constexpr int S = decltype(slot)::value;
    scn[S] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsc, (int)cb, 0, 0));
    shn[S] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsh, (int)cb, 0, 0));
  };
  auto first = [&](auto slot) {
    const unsigned cb = col_bytes(pn(decltype(slot)::value));
    request_affine(slot, cb);
#pragma unroll
    for (int r = 0; r < 16; r++) request(slot, r, cb);
  };
  first(std::integral_constant<int, 0>{});   // before the rows of A: these requests fly through the whole A phase
  if constexpr (LA == 2) first(std::integral_constant<int, 1>{});
  asm volatile("" ::: "memory");
  {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long)p.M * K * 4), 0x00020000);
    const int m = m0 + wave * 32 + lr;
    const unsigned base = m < p.M ? ((unsigned)m * (unsigned)K + 8u * kh2) * 4u : 0x80000000u;
    constexpr int CH = KT < 8 ? KT : 8;   // raw rows in flight: 8 k-steps (64 registers) at a time, then split
#pragma unroll
    for (int k0 = 0; k0 < KT; k0 += CH) {
      f32x4 ra[CH][2];
#pragma unroll
      for (int kt = 0; kt < CH; kt++)
#pragma unroll
        for (int h = 0; h < 2; h++)
          ra[kt][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + (k0 + kt) * 64 + h * 16), 0, 0));
#pragma unroll
      for (int kt = 0; kt < CH; kt++) {
        uint2 o0[NS], o1[NS];
        if constexpr (F16) {
          split4h(ra[kt][0], sx, o0);
          split4h(ra[kt][1], sx, o1);
        } else {
          split4<NS>(ra[kt][0], o0);
          split4<NS>(ra[kt][1], o1);
        }
#pragma unroll
        for (int q = 0; q < NS; q++) fa[k0 + kt][q] = uint4{o0[q].x, o0[q].y, o1[q].x, o1[q].y};
      }
      if (KT > CH) __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first panel's planes (and everything above)
  auto panel = [&](auto slot, const int i) {   // i-th panel of this block
    constexpr int S = decltype(slot)::value;
    const int buf = i & 1;
    __syncthreads();  // this panel readable by everybody; everybody is done with the other buffer
    asm volatile("" ::: "memory");
    if (i + 1 < cnt) issue_b(pn(i + 1), buf ^ 1);
    asm volatile("" ::: "memory");
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const char* bb = bring + buf * BBUF + boff;
    // B fragments one k-step ahead, fenced: left alone the compiler hoists the fragment reads of ALL k-steps (and spills)
    uint4 fb[2][NS];
    auto read_b = [&](int kt, uint4 (&f)[NS]) {
#pragma unroll
      for (int q = 0; q < NS; q++) f[q] = *(const uint4*)(bb + (kt * NS + q) * 1024);
    };
    read_b(0, fb[0]);
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
      if (kt + 1 < KT) read_b(kt + 1, fb[(kt + 1) & 1]);
#pragma unroll
      for (int sum = NS - 1; sum >= 0; sum--)
#pragma unroll
        for (int qa = 0; qa <= sum; qa++) {
          const int qb = sum - qa;
          if (F16 && qa + qb > 1) continue;   // (never: NS = 2)
          if constexpr (F16)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[kt][qa]),
                                                         __builtin_bit_cast(f16x8, fb[kt & 1][qb]), acc, 0, 0, 0);
          else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[kt][qa]),
                                                          __builtin_bit_cast(bf16x8, fb[kt & 1][qb]), acc, 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned cb = col_bytes(pn(i)), cbn = col_bytes(pn(i + LA));
    float sc = has_scale ? scn[S] : 1.f;
    const float sh = shn[S];
    if (F16) sc *= finv;   // operands were scaled by powers of two: exact rescale of the accumulated sum
    request_affine(slot, cbn);
    float vv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {   // same expressions as conv_epilogue
      float v = acc[r] * sc + sh;
      v += has_res ? ur[S][r] : 0.f;
      v = p.relu ? fmaxf(v, 0.f) : v;
      if constexpr (MASK) v = um[S][r] > 0.f ? v * p.mask_scale : 0.f;
      const bool ok = cb != 0x80000000u && mlane + 8 * (r >> 2) + (r & 3) < p.M;
      const float av = ok ? fabsf(v) : 0.f;
      amx = fmaxf(amx, av);
      asum += av;
      acnt += ok ? 1.f : 0.f;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)(ylane + row_off(r) + cb), 0, 0);
      request(slot, r, cbn);
      vv[r] = v;
    }
    // planes of the next panel (issued at the top of this iteration): everything issued after them may stay in flight --
    // 2 + 16 x (store, residual (, mask)) operations (+ 8 plane stores); vector memory operations retire in order
    // (`s_setprio 0` changes nothing: it marks this wait for tests/test_rows_kernel_isa.py, which counts the operations
    // between the copies and the wait in the shipped code object)
    if (rb) {   // (uniform) round 6: the panel's 32 x 32 values as row-blocked fp16 planes, eight 8-byte stores per lane
      const int cq = (pn(i) * BN + col_l) & ~3;
      const unsigned coff = cb != 0x80000000u ? (unsigned)(cq >> 4) * rb_cstep + (unsigned)(cq & 15) * 2u : 0x80000000u;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        unsigned w4[4] = {rb_split1(vv[4 * j], rbs), rb_split1(vv[4 * j + 1], rbs), rb_split1(vv[4 * j + 2], rbs), rb_split1(vv[4 * j + 3], rbs)};
        quad_transpose(w4, lane);
        const unsigned o = (rboff[j] | coff) & 0x80000000u ? 0x80000000u : rboff[j] + coff;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 hh = {__builtin_amdgcn_perm(w4[1], w4[0], 0x05040100u), __builtin_amdgcn_perm(w4[3], w4[2], 0x05040100u)};
        const u32x2 ll = {__builtin_amdgcn_perm(w4[1], w4[0], 0x07060302u), __builtin_amdgcn_perm(w4[3], w4[2], 0x07060302u)};
        __builtin_amdgcn_raw_buffer_store_b64(hh, rrb0, (int)o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(ll, rrb1, (int)o, 0, 0);
      }
      if constexpr (MASK) asm volatile("s_waitcnt vmcnt(58)\n\ts_setprio 0" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(42)\n\ts_setprio 0" ::: "memory");
    } else {
      if constexpr (MASK) asm volatile("s_waitcnt vmcnt(50)\n\ts_setprio 0" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(34)\n\ts_setprio 0" ::: "memory");
    }
  };
  for (int i = 0; i < cnt; i += 2) {   // unrolled by two: register set and ring buffer of a panel are compile-time
    panel(std::integral_constant<int, 0>{}, i);
    if (i + 1 < cnt) panel(std::integral_constant<int, LA == 2 ? 1 : 0>{}, i + 1);
  }
  if (p.amax_out) conv_amax_commit(p, AmaxAcc{amx, asum, acnt}, blockIdx.y * gridDim.x + blockIdx.x);
}

}  // namespace

namespace mmtconv {
int fill(ConvP& p, const mmt_conv_args* a) {
  if (!a || !a->x) return MMT_EINVAL;
  p.x = a->x; p.w = a->w; p.scale = a->scale; p.shift = a->shift; p.res = a->res; p.mask = a->mask;
  p.mul = a->mul; p.y = a->y;
  p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW;
  p.stride = a->stride; p.pad = a->pad; p.Ho = a->Ho; p.Wo = a->Wo;
  p.relu = a->relu; p.res_mode = a->res_mode; p.out_stride = a->out_stride < 1 ? 1 : a->out_stride;
  p.out_H = a->out_H; p.out_W = a->out_W; p.mask_scale = a->mask_scale;
  p.wpl = (const unsigned short*)a->w_planes; p.wpl_stride = a->w_plane_stride;
  p.xpl = (const unsigned short*)a->x_planes; p.xpl_stride = a->x_plane_stride;
  p.ypl = (unsigned short*)a->y_planes; p.ypl_stride = a->y_plane_stride;
  p.xpl_rb = a->x_planes_layout;
  p.yrb = (unsigned short*)a->y_rb; p.yrb_stride = a->y_rb_stride; p.yrb_s = a->y_rb_scale;
  p.amax_next = (unsigned*)a->y_amax_next; p.xpl_lag = a->x_planes_lag;
  p.f16_sx = p.f16_sw = nullptr;
  p.f16_ax = 0;
  p.guard_x = (const float*)a->f16_guard_x; p.guard_dy = (const float*)a->f16_guard_dy;
  p.w_src = (const float*)a->w_src; p.w_src_scale = (const float*)a->w_src_scale;
  p.x2 = p.dy2 = p.f16_sx2 = p.f16_sw2 = p.guard_x2 = p.guard_dy2 = nullptr;
  p.seg_z = 0;
  { const char* e = getenv("MMT_DIRECT_EPI"); p.staged_epilogue = e && atoi(e) == 0; }   // read per call (A/B timing)
  p.amax_out = (unsigned*)a->y_amax;
  p.amax_stats = a->y_amax_stats;
  p.io = a->io_bf16;
  if (p.io & ~(IO_X | IO_Y | IO_RES | IO_MASK | IO_DY)) return MMT_EINVAL;
  if (p.io & IO_X) {  // x itself is the (only) bf16 plane: the all-planes kernels with one term
    if (p.xpl || ((size_t)a->x & 15)) return MMT_EINVAL;
    p.xpl = (const unsigned short*)a->x; p.xpl_stride = 0;
  }
  if ((p.io & IO_Y) && (p.ypl || a->mul)) return MMT_EINVAL;
  if (p.ypl && ((a->Cout & 3) || ((size_t)p.ypl & 7) || (p.ypl_stride & 3) || a->out_stride > 1)) return MMT_EINVAL;
  p.yrb_M = a->y_rb_rows > 0 && (long)a->y_rb_rows < (long)a->N * a->Ho * a->Wo ? a->y_rb_rows : a->N * a->Ho * a->Wo;
  if (p.yrb && (!p.yrb_s || (a->Cout & 15) || a->out_stride > 1 || (p.io & IO_Y) || ((size_t)p.yrb & 15) || (p.yrb_stride & 7) ||
                a->y_rb_rows < 0 || (long)a->N * a->Ho * a->Wo * a->Cout >= (1L << 30) || p.yrb_stride < (long)p.yrb_M * a->Cout))
    return MMT_EINVAL;
  if ((long)p.N * p.Ho * p.Wo > 0x7fffffffL) return MMT_EINVAL;
  if ((long)p.N * p.H * p.W * p.Cin >= 0x7fffffffL || (long)p.N * p.Ho * p.Wo * p.Cout >= 0x7fffffffL ||
      (long)p.Cout * p.KH * p.KW * p.Cin >= 0x7fffffffL) return MMT_EINVAL;  // kernels use 32-bit element offsets
  p.M = p.N * p.Ho * p.Wo;
  p.K = p.KH * p.KW * p.Cin;
  p.cin32 = (p.Cin % 32) == 0;
  p.cin4 = (p.Cin % 4) == 0;
  return 0;
}
}  // namespace mmtconv

namespace {

template <int BM, int BN, int WM, int WN>
int launch_fwd(const ConvP& p, hipStream_t s) {
  const int tiles = mmt_cdiv(p.M, BM) * mmt_cdiv(p.Cout, BN);
  const size_t lds = (size_t)(2 * BM * 32 + 2 * BN * 32) * sizeof(float);
  hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN>), dim3(tiles), dim3(256), lds, s, p);
  MMT_LAUNCH_CHECK();
  return 0;
}

template <int BM, int BN, int WM, int WN, int NS>
int launch_split(const ConvP& p, hipStream_t s) {
  const int tiles = mmt_cdiv(p.M, BM) * mmt_cdiv(p.Cout, BN);
  size_t ring = (size_t)2 * NS * (BM + BN) * 32, epi = (size_t)BM * BN * sizeof(float);
  hipLaunchKernelGGL((conv_fwd_split_kernel<BM, BN, WM, WN, NS>), dim3(tiles), dim3(256), ring > epi ? ring : epi, s, p);
  MMT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

namespace mmtconv {
SplitWs split_workspace(hipStream_t s) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SplitWs> table;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SplitWs{nullptr, nullptr};
  std::lock_guard<std::mutex> g(mu);
  auto it = table.find({dev, s});
  if (it != table.end()) return it->second;
  SplitWs w{nullptr, nullptr};
  if (hipMalloc((void**)&w.ws, SPLITK_WS_BYTES) != hipSuccess) return SplitWs{nullptr, nullptr};
  if (hipMalloc((void**)&w.tickets, SPLITK_TICKETS * sizeof(unsigned)) != hipSuccess ||
      hipMemset(w.tickets, 0, SPLITK_TICKETS * sizeof(unsigned)) != hipSuccess)
    return SplitWs{nullptr, nullptr};
  table[{dev, s}] = w;
  return w;
}
}  // namespace mmtconv

namespace {

// number of K ranges for the 128 x 128 kernel: fill the 512 resident block slots when the tiles alone do not, keeping
// at least 32 steps per range
static int pick_ksplit(const ConvP& p) {
  const char* on_env = getenv("MMT_SPLITK");  // read per call: the schedule-equivalence test switches it
  if ((on_env && atoi(on_env) == 0) || p.Cout < 128) return 1;
  const long t128 = (long)mmt_cdiv(p.M, 128) * mmt_cdiv(p.Cout, 128);
  const int nkt = p.K >> 4;
  constexpr int tmax = 512;   // (tuned: profiles/r04_dispatch_sweep.txt)
  constexpr int kmin = 128;   // (tuned: profiles/r04_dispatch_sweep.txt)
  if (t128 >= tmax || nkt < kmin) return 1;  // K >= 2048: shorter sums lose more in the second launch than they gain
  int ks = (int)(512 / t128);
  if (ks > nkt / 32) ks = nkt / 32;
  if (ks > 16) ks = 16;
  return ks < 2 ? 1 : ks;
}

template <int BM, int BN, int WM, int WN, int NS, int S>
int launch_glds(const ConvP& p, hipStream_t s, int ksplit = 1) {
  const int tiles = mmt_cdiv(p.M, BM) * mmt_cdiv(p.Cout, BN) * ksplit;
  SplitWs w{nullptr, nullptr};
  if (ksplit > 1) {
    w = split_workspace(s);
    if (!w.ws || tiles > 1024) return MMT_EINVAL;
  }
  const size_t ring = (size_t)S * (BM * 64 + NS * BN * 32), epi = (size_t)BM * BN * sizeof(float);
  const size_t lds = ring > epi ? ring : epi;
  auto kern = conv_fwd_glds_kernel<BM, BN, WM, WN, NS, S>;
  if (lds > 65536) {
    static bool done = false;  // per instantiation
    if (!done) {
      const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, s, p, p.wpl, p.wpl_stride, ksplit, w.ws);
  MMT_LAUNCH_CHECK();
  if (ksplit > 1) {
    if constexpr (BM == 128 && BN == 128)
      launch_finish<BM, BN>(p, tiles / ksplit, ksplit, w.ws, s);
    else
      return MMT_EINVAL;
    MMT_LAUNCH_CHECK();
  }
  return 0;
}

template <int BM, int BN, int WM, int WN, int NS, int S>
int launch_pp(const ConvP& p, hipStream_t s, int ksplit = 1) {
  const int tiles = mmt_cdiv(p.M, BM) * mmt_cdiv(p.Cout, BN) * ksplit;
  SplitWs w{nullptr, nullptr};
  if (ksplit > 1) {
    w = split_workspace(s);
    if (!w.ws || tiles > 1024) return MMT_EINVAL;
  }
  const size_t ring = (size_t)S * NS * (BM + BN) * 32, epi = (size_t)BM * BN * sizeof(float);
  const size_t lds = ring > epi ? ring : epi;
  void (*kern)(const ConvP, const int, float*) = conv_fwd_pp_kernel<BM, BN, WM, WN, NS, S>;
  if (lds > 65536) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, s, p, ksplit, w.ws);
  MMT_LAUNCH_CHECK();
  if (ksplit > 1) {
    if constexpr (BM == 128 && BN == 128)
      launch_finish<BM, BN>(p, tiles / ksplit, ksplit, w.ws, s);
    else
      return MMT_EINVAL;
    MMT_LAUNCH_CHECK();
  }
  return 0;
}

// K ranges for the tap-strip kernel: 1 when its 256 x 128 tiles fill the chip (one 512-thread block per CU), else enough
// ranges to do so -- possible when a tile's pixels are consecutive in memory (Wo == TW: the finish kernel's row mapping),
// Cout is a multiple of 128 and every range keeps >= 6 super-steps; 0 = not a shape for this kernel
static int strip_ksplit(const ConvP& p, int tw) {
  const long tiles = (long)p.N * (p.Ho * p.Wo / 256) * mmt_cdiv(p.Cout, 128);
  if (tiles >= 256) return 1;
  // fp32 tensors (the split arithmetics): a few-tile shape is NOT taken -- this kernel needs its input as pre-split planes, a pass
  // of its own over the tensor (~9 us on the student's N = 2 maps) that the tiled kernel, which splits in registers, does not; with
  // equal kernel times (profiles/r04_dispatch_sweep.txt: 35.07 vs 35.12-35.23 ms) the pass is what counts: 32 fewer launches per
  // step, 35.55 -> 35.34 ms in three same-box alternations (round 4).  bf16-stored tensors ARE their plane: split form kept.
  if (!(p.io & IO_X)) return 0;
  const char* e = getenv("MMT_SPLITK");
  if ((e && atoi(e) == 0) || p.Wo != tw || (p.Cout & 127) || tiles < 32) return 0;
  int ks = (int)((256 + tiles - 1) / tiles);
  const int pairs = 3 * (p.Cin >> 4) / 2;
  if (ks > pairs / 3) ks = pairs / 3;
  if (ks > 8) ks = 8;
  if (tiles * ks > 512) ks = (int)(512 / tiles);
  return ks >= 2 ? ks : 0;
}

// 3x3 / stride 1 / pad 1 with both operands as planes: which strip width (0 = not taken)
// K order of the tap-strip kernel (bits 24-30 of its ksplit argument = slabs per group, 0 = kh outermost as in rounds 2-5):
// MMT_STRIP_KORDER = 0 | 1 | 2 | 4 | 8 (read per call: A/B timing).  Default 4 (tools/strip_korder.py, profiles/r06_strip_korder.txt):
// the fabric reads of 1 (141 MB per launch instead of 278) at the cycle count of 0 -- every change of kh recomputes the copy slots'
// row offsets, which costs 7 % of the kernel's cycles when it happens every super-step and 1 % every fourth
static int strip_korder(const ConvP& p) {
  const char* e = getenv("MMT_STRIP_KORDER");
  int g = e ? atoi(e) : 4;
  if (g != 0 && g != 1 && g != 2 && g != 4 && g != 8) g = 4;
  while (g > 1 && ((p.Cin >> 4) % g) != 0) g >>= 1;
  return g << 24;
}

static int strip_tw(const ConvP& p, bool need_planes = true) {
  if ((need_planes && !p.xpl) || !p.wpl || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.out_stride != 1 || p.Ho != p.H ||
      p.Wo != p.W || (p.Cin & 31) || p.Cout <= 32 || ((size_t)p.xpl & 15) || (p.xpl_stride & 7))
    return 0;
  const char* e = getenv("MMT_STRIP");  // read per call (A/B timing)
  if (e && atoi(e) == 0) return 0;
  int tw = 0;
  const char* w64 = getenv("MMT_STRIP_TW64");   // A/B: 4 image rows x 64 pixels per tile (6 input rows for 4: halo x 1.5) instead of 2 x 128 (4 for 2: x 2)
  if (p.Wo % 128 == 0 && p.Ho % 2 == 0 && !(w64 && atoi(w64) && p.Ho % 4 == 0)) tw = 128;
  else if (p.Wo % 64 == 0 && p.Ho % 4 == 0) tw = 64;
  constexpr int minc = 128;   // (tuned: profiles/r04_dispatch_sweep.txt)
  if (!tw || p.Cin < minc) return 0;  // K = 576 (the 64-channel layer1 convs): 12 super-steps do not amortise the fill
  return strip_ksplit(p, tw) ? tw : 0;
}

template <int TW, int NS = 3>
int launch_strip(const ConvP& p, hipStream_t s) {
  constexpr int R = 256 / TW, SW = TW + 32;
  const size_t ring = (size_t)2 * (NS * R * SW * 32 + 3 * NS * 128 * 32), epi = (size_t)256 * 128 * sizeof(float);
  const size_t lds = ring > epi ? ring : epi;
  const int ksplit = strip_ksplit(p, TW);
  SplitWs w{nullptr, nullptr};
  if (ksplit > 1) {
    w = split_workspace(s);
    if (!w.ws) return MMT_EINVAL;
  }
  void (*kern)(const ConvP, const int, float*) = conv3x3_strip_kernel<TW, NS>;
  {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int tiles = p.N * (p.Ho * p.Wo / 256) * mmt_cdiv(p.Cout, 128);
  hipLaunchKernelGGL(kern, dim3(tiles * ksplit), dim3(512), lds, s, p, ksplit | strip_korder(p), w.ws);
  MMT_LAUNCH_CHECK();
  if (ksplit > 1) {  // Wo == TW: tile t covers the 256 consecutive pixels [256 t', 256 t' + 256) of its channel block
    launch_finish<256, 128>(p, tiles, ksplit, w.ws, s);
    MMT_LAUNCH_CHECK();
  }
  return 0;
}

// panels per block of the row-resident 1x1 kernel: all of them when the row blocks alone fill the chip (>= 512 resident
// slots), else column groups so that about 512 blocks exist (256 / 1024 / 2048 measured slower: profiles/r04_rows_epilogue.txt)
static int rows_ppb(const ConvP& p) {
  const int tiles_m = mmt_cdiv(p.M, 128), npanel = mmt_cdiv(p.Cout, 32);
  if (tiles_m >= 512) return npanel;
  int groups = mmt_cdiv(512, tiles_m);
  if (groups > npanel) groups = npanel;
  return mmt_cdiv(npanel, groups);
}

template <int KT, int NS, bool F16 = false, bool MASK = false>
int launch_rows(const ConvP& p, hipStream_t s) {
  const size_t lds = 2 * (size_t)KT * NS * 1024;
  auto kern = conv1x1_rows_kernel<KT, NS, F16, MASK>;
  if (lds > 65536) {
    static bool done = false;  // per instantiation
    if (!done) {
      const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      done = true;
    }
  }
  const int ppb = rows_ppb(p);
  hipLaunchKernelGGL(kern, dim3(mmt_cdiv(p.M, 128), mmt_cdiv(mmt_cdiv(p.Cout, 32), ppb)), dim3(256), lds, s, p, p.wpl, p.wpl_stride, ppb);
  MMT_LAUNCH_CHECK();
  return 0;
}

// is this call one for the row-resident 1x1 kernel?  (K = 64 / 128 in every split mode; K = 256 on the fp16 split only)
static bool rows_shape(const ConvP& p, bool f16) {
  const char* rows_env = getenv("MMT_ROWS");  // read per call: the parity tests switch it
  const int rows = rows_env ? atoi(rows_env) : 1;
  constexpr int rows_min = 256;   // (tuned: profiles/r04_dispatch_sweep.txt)  // blocks of 128 rows (bf16 split)
  constexpr int rows_min16 = 16;   // (tuned: profiles/r04_dispatch_sweep.txt)
  if (!rows || p.io || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.Cout < 64 || (p.Cout & 3) || p.res_mode > 2 || p.mul ||
      p.out_stride != 1 || (long)p.M * p.Cin * 4 >= (1L << 31) || (long)p.M * p.Cout * 4 >= (1L << 31) ||
      (p.res_mode == 2 && (p.Wo & 7)))
    return false;
  if (f16) return (p.Cin == 64 || p.Cin == 128 || p.Cin == 256) && p.M >= 128 * rows_min16;
  return !p.mask && p.res_mode <= 1 && (p.Cin == 64 || p.Cin == 128) && p.M >= 128 * rows_min;
}

template <int NS>
int launch_glds_variant(int variant, const ConvP& p, hipStream_t s) {
  // 1x1 / stride 1 layers with K = 64 or 128 and many rows: one block per 128 rows, all Cout panels (see the kernel)
  if (rows_shape(p, false)) {
    if (p.Cin == 64) return launch_rows<4, NS>(p, s);
    if (p.Cin == 128) return launch_rows<8, NS>(p, s);
  }
  if (NS == 3) {
    const int tw = strip_tw(p);
    if (tw == 128) return launch_strip<128>(p, s);
    if (tw == 64) return launch_strip<64>(p, s);
  }
  const int ksplit = pick_ksplit(p);
  if (p.io & IO_X) {  // x stored as bf16 (mode 1 only, checked by the caller): both operands are DMA-copied as they are
    if constexpr (NS == 1) {
      const int tw = strip_tw(p);
      if (tw == 128) return launch_strip<128, 1>(p, s);
      if (tw == 64) return launch_strip<64, 1>(p, s);
      if (ksplit > 1) return launch_pp<128, 128, 2, 2, 1, 3>(p, s, ksplit);
      switch (variant) {
        case 1: return launch_pp<128, 128, 2, 2, 1, 3>(p, s);
        case 3: return launch_pp<128, 64, 2, 2, 1, 3>(p, s);
        default: return launch_pp<64, 64, 2, 2, 1, 3>(p, s);
      }
    }
    return MMT_EINVAL;
  }
  // activations pre-split into planes by the caller: the all-planes kernel (128 x 128 tiles, 2 x 2 waves)
  if (ksplit > 1) return launch_glds<128, 128, 4, 1, NS, 3>(p, s, ksplit);
  switch (variant) {
    case 1: return launch_glds<128, 128, 4, 1, NS, 3>(p, s);
    case 3: return launch_glds<128, 64, 4, 1, NS, 3>(p, s);
    default: return launch_glds<64, 64, 2, 2, NS, 3>(p, s);
  }
}

template <int NS>
int launch_split_variant(int variant, const ConvP& p, hipStream_t s) {
  switch (variant) {
    case 1: return launch_split<128, 128, 2, 2, NS>(p, s);
    case 3: return launch_split<128, 64, 2, 2, NS>(p, s);
    default: return launch_split<64, 64, 2, 2, NS>(p, s);
  }
}

}  // namespace

namespace mmtconv {
static int g_precision = -1;
int precision() {
  if (g_precision < 0) {
    g_precision = getenv("MMT_CONV_PRECISION") ? atoi(getenv("MMT_CONV_PRECISION")) : 3;
    if (g_precision < 0 || g_precision > 3) g_precision = 3;
  }
  return g_precision;
}
}  // namespace mmtconv

extern "C" int mmt_set_conv_precision(int mode) {
  if (mode < 0 || mode > 3) return MMT_EINVAL;
  mmtconv::g_precision = mode;
  return 0;
}

extern "C" int mmt_get_conv_precision(void) { return precision(); }

// ---- experiment: 3x3 convolution on the tap-strip kernel with a two-term fp16 split (3 products instead of 6)
// x_planes / w_planes: the two fp16 planes of x * s_x and of the packed weight * s_w; s_x, s_w: device scalars
extern "C" int mmt_conv3x3_strip_f16x2(const mmt_conv_args* a, const float* s_x, const float* s_w, void* stream) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  if (!p.y || !p.xpl || !p.wpl || !s_x || !s_w || p.io || p.ypl) return MMT_EINVAL;
  p.f16_sx = s_x; p.f16_sw = s_w;
  const int tw = strip_tw(p);
  if (!tw) return MMT_EINVAL;
  const int ksplit = strip_ksplit(p, tw);
  constexpr int NS = 2;
  hipStream_t s = (hipStream_t)stream;
  SplitWs w{nullptr, nullptr};
  if (ksplit > 1) {
    w = split_workspace(s);
    if (!w.ws) return MMT_EINVAL;
  }
  const int tiles = p.N * (p.Ho * p.Wo / 256) * mmt_cdiv(p.Cout, 128);
  auto go = [&](auto kern, int TW) {
    const int R = 256 / TW, SW = TW + 32;
    const size_t ring = (size_t)2 * (NS * R * SW * 32 + 3 * NS * 128 * 32), epi = (size_t)256 * 128 * sizeof(float);
    const size_t lds = ring > epi ? ring : epi;
    const hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (er != hipSuccess) return (int)er;
    hipLaunchKernelGGL(kern, dim3(tiles * ksplit), dim3(512), lds, s, p, ksplit | strip_korder(p), w.ws);
    return 0;
  };
  if (tw == 128) e = go(conv3x3_strip_kernel<128, NS, true>, 128);
  else e = go(conv3x3_strip_kernel<64, NS, true>, 64);
  if (e) return e;
  MMT_LAUNCH_CHECK();
  if (ksplit > 1) {
    launch_finish<256, 128>(p, tiles, ksplit, w.ws, s);
    MMT_LAUNCH_CHECK();
  }
  return 0;
}

static int pick_variant(const ConvP& p);
// round 6: would the launch this call takes on the fp16 split write row-blocked planes of y from its epilogue (mmt_conv_args.y_rb)?
// The register-direct, LDS-staged and split-K-finish epilogues do; the patch kernels (layer1's 3x3, the stem) do not
static bool rb_epilogue_ok(const ConvP& p) {
  if ((p.Cout & 15) || p.out_stride > 1 || (p.io & IO_Y) || p.mul || (long)p.M * p.Cout >= (1L << 30)) return false;
  if (c64_shape(p)) return false;
  return true;   // (the row-resident 1x1 kernel writes them from its register epilogue)
}
extern "C" int mmt_conv_writes_rb(const mmt_conv_args* a) {
  ConvP p;
  if (fill(p, a) || precision() != 3 || p.io) return 0;
  p.f16_ax = 1;
  p.f16_sx = p.f16_sw = (const float*)16;   // (shape question: placeholders for the shape tests that look at them)
  return rb_epilogue_ok(p) ? 1 : 0;
}
// any convolution the DMA-fed kernel takes (Cin % 16 == 0, Cout > 32), raw fp32 x: x_amax = device max |x|, w_planes = the two
// fp16 planes of the packed weight, s_w their device scale
extern "C" int mmt_conv_forward_f16x2(const mmt_conv_args* a, const float* x_amax, const float* s_w, void* stream) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  if (!p.y || !p.wpl || !x_amax || !s_w || p.io || p.ypl || p.xpl_rb || (p.Cin & 15) || ((size_t)p.wpl & 15) || (p.wpl_stride & 7)) return MMT_EINVAL;
  if (p.M == 0 || p.Cout == 0) return 0;
  const int variant = pick_variant(p);
  if (variant == 0) return MMT_EINVAL;
  p.f16_sx = x_amax; p.f16_sw = s_w; p.f16_ax = 1;
  hipStream_t s = (hipStream_t)stream;
  if (p.yrb && !rb_epilogue_ok(p)) return MMT_EINVAL;   // (the caller asks mmt_conv_writes_rb first)
  if (c64_shape(p)) return launch_c64(p, s);   // layer1's 3x3, 64 -> 64 channels: the patch kernel (conv_stem.hip, round 5)
  if (rows_shape(p, true)) {   // 1x1 layers with K = 64 / 128 / 256: rows resident in registers as fp16 fragments
    if (p.mask) {
      if (p.Cin == 64) return launch_rows<4, 2, true, true>(p, s);
      if (p.Cin == 128) return launch_rows<8, 2, true, true>(p, s);
      return launch_rows<16, 2, true, true>(p, s);
    }
    if (p.Cin == 64) return launch_rows<4, 2, true>(p, s);
    if (p.Cin == 128) return launch_rows<8, 2, true>(p, s);
    return launch_rows<16, 2, true>(p, s);
  }
  const int ksplit = pick_ksplit(p);
  constexpr int S = 3;   // LDS stages of the operand ring (4 / 5 measured no faster: 36.4 / 34.8 / 35.0 us, profiles/r03_history.md)
  auto go = [&](auto kern, int BM, int BN, int ks) -> int {
    const int tiles = mmt_cdiv(p.M, BM) * mmt_cdiv(p.Cout, BN) * ks;
    SplitWs w{nullptr, nullptr};
    if (ks > 1) {
      w = split_workspace(s);
      if (!w.ws || tiles > 1024) return MMT_EINVAL;
    }
    const size_t ring = (size_t)S * (BM * 64 + 2 * BN * 32), epi = (size_t)BM * BN * sizeof(float);
    const size_t lds = ring > epi ? ring : epi;
    if (lds > 65536) {
      const hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (er != hipSuccess) return (int)er;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, s, p, p.wpl, p.wpl_stride, ks, w.ws);
    if (ks > 1)
      launch_finish<128, 128>(p, tiles / ks, ks, w.ws, s);
    return 0;
  };
#define MMT_GLDS_GO(SS)                                                                                      \
  do {                                                                                                       \
    if (ksplit > 1) e = go(conv_fwd_glds_kernel<128, 128, 4, 1, 2, SS, true>, 128, 128, ksplit);            \
    else if (variant == 1) e = go(conv_fwd_glds_kernel<128, 128, 4, 1, 2, SS, true>, 128, 128, 1);          \
    else if (variant == 3) e = go(conv_fwd_glds_kernel<128, 64, 4, 1, 2, SS, true>, 128, 64, 1);            \
    else e = go(conv_fwd_glds_kernel<64, 64, 2, 2, 2, SS, true>, 64, 64, 1);                                \
  } while (0)
  MMT_GLDS_GO(3);
#undef MMT_GLDS_GO
  if (e) return e;
  MMT_LAUNCH_CHECK();
  return 0;
}

static int pick_variant(const ConvP& p) {
  if (p.Cout <= 32) return 0;
  // K <= 256 (the 1x1 layers of layer1/layer2/FPN laterals): 2-8 k-tiles per output tile, so prologue and epilogue
  // dominate; the 64x64 configuration keeps ~4x more blocks resident to overlap them (measured +10..25 %)
  constexpr int lowk = 256;   // (tuned: profiles/r04_dispatch_sweep.txt)
  constexpr int lowv = 3;   // (tuned: profiles/r04_dispatch_sweep.txt)
  if (p.K <= lowk) return (p.Cout >= 64 && (long)mmt_cdiv(p.M, 128) * mmt_cdiv(p.Cout, 64) >= 768) ? lowv : 2;
  // enough 128x128 tiles to fill 256 CUs (2 resident blocks each) -> 128x128; else 128x64 (twice the blocks, A tile
  // still reused across 64 output channels); else 64x64 (4x the blocks)
  const long t128 = (long)mmt_cdiv(p.M, 128) * mmt_cdiv(p.Cout, 128);
  constexpr int t128min = 256;   // (tuned: profiles/r04_dispatch_sweep.txt)
  if (t128 >= t128min && p.Cout > 64) return 1;
  constexpr int mid = 1;   // (tuned: profiles/r04_dispatch_sweep.txt)
  const long t12864 = (long)mmt_cdiv(p.M, 128) * mmt_cdiv(p.Cout, 64);
  if (mid && t12864 >= 256 && p.Cout >= 64) return 3;
  return 2;
}

extern "C" int mmt_conv_variant(const mmt_conv_args* a) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  const int v = pick_variant(p);
  // conv3x3_strip_kernel: mode 3 with x_planes given, mode 1 with x stored as bf16
  if (v != 0 && (precision() == 3 || (precision() == 1 && (p.io & IO_X))) && strip_tw(p)) return 4;
  // the split-K form runs on the 128 x 128 kernel whatever the tile variant would have been
  if (v != 0 && precision() > 0 && (p.Cin & 15) == 0 && p.wpl && pick_ksplit(p) > 1) return 1;
  return v;
}

extern "C" int mmt_conv_wants_planes(const mmt_conv_args* a) {
  ConvP p;
  int e = fill(p, a);
  if (e) return 0;
  if (precision() != 3 || pick_variant(p) == 0) return 0;
  return strip_tw(p, false) ? 1 : 0;
}

extern "C" int mmt_conv_ksplit(const mmt_conv_args* a) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  if (pick_variant(p) != 0 && (precision() == 3 || (precision() == 1 && (p.io & IO_X))) && strip_tw(p)) return 1;
  if (pick_variant(p) != 0 && precision() > 0 && (p.Cin & 15) == 0 && p.wpl) return pick_ksplit(p);
  return 1;
}

extern "C" int mmt_conv_forward(const mmt_conv_args* a, void* stream) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  if (!p.y || p.xpl_rb || p.yrb) return MMT_EINVAL;   // (row-blocked planes: the fp16-split entry points only)
  if (p.M == 0 || p.Cout == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int variant = pick_variant(p);
  const int prec = precision();
  if ((p.io & IO_X) && !(prec == 1 && variant != 0 && (p.Cin & 15) == 0 && p.wpl)) return MMT_EINVAL;  // bf16 x: DMA kernels only
  if (!p.w && !(prec > 0 && variant != 0 && (p.Cin & 15) == 0 && p.wpl)) return MMT_EINVAL;  // planes-only call
  if (prec > 0 && variant != 0 && (p.Cin & 15) == 0 && p.wpl) {
    if (((size_t)p.wpl & 15) || (p.wpl_stride & 7)) return MMT_EINVAL;
    if (prec == 1) return launch_glds_variant<1>(variant, p, s);
    if (prec == 2) return launch_glds_variant<2>(variant, p, s);
    return launch_glds_variant<3>(variant, p, s);
  }
  if (prec > 0 && variant != 0 && (p.Cin & 15) == 0) {
    if (prec == 1) return launch_split_variant<1>(variant, p, s);
    if (prec == 2) return launch_split_variant<2>(variant, p, s);
    return launch_split_variant<3>(variant, p, s);
  }
  switch (variant) {
    case 0: return launch_fwd<128, 32, 4, 1>(p, s);
    case 1: return launch_fwd<128, 128, 2, 2>(p, s);
    case 3: return launch_fwd<128, 64, 2, 2>(p, s);
    default: return launch_fwd<64, 64, 2, 2>(p, s);
  }
}
