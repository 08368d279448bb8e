// ROIAlign forward/backward for gfx950, NHWC, all FPN levels in one launch.
//
// Replaces cuda/ROIAlign_cuda.cu:64-122 (forward, 1 thread per output element over NCHW) and
// :177-254 (backward) of the reference, plus the per-level gather/scatter loop of
// modeling/poolers.py:116-119.  Mapping here: one 64-lane wave owns one (roi, bin); lanes run over
// channels (float4 per lane => 256 channels per pass), so every tap is a coalesced 1 KiB read and
// every gradient scatter is a coalesced run of fp32 atomics.  The bilinear geometry is wave-uniform
// (derived from blockIdx / readfirstlane) and therefore lives on the scalar unit.
//
// Numerics: same expression order as cpu/ROIAlign_cpu.cpp:113-219 (w1*v1 + w2*v2 + w3*v3 + w4*v4
// accumulated sample by sample, then / count); this file is built with -ffp-contract=off so no FMA
// is formed and results are bit-identical to the reference CPU path.
#include "common.h"

struct Pyr {
  const float* feat[4];
  float* grad[4];
  int H[4], W[4];
  float scale[4];
  int N, C;
};

struct Bilin {
  int yl, xl, yh, xh;
  float w1, w2, w3, w4;
  bool empty;
};

__device__ __forceinline__ Bilin bilin(int H, int W, float y, float x) {
  Bilin b;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    b.empty = true; b.yl = b.xl = b.yh = b.xh = 0; b.w1 = b.w2 = b.w3 = b.w4 = 0.f; return b;
  }
  b.empty = false;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
  float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
  b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
  b.yl = y_low; b.xl = x_low; b.yh = y_high; b.xh = x_high;
  return b;
}

// HALF (forward only, VEC == 4): the pyramid levels are bf16 tensors (bf16 activation storage, mode 1); taps are widened
// exactly, arithmetic and output stay fp32
template <bool BWD, int VEC, bool HALF = false>
__global__ __launch_bounds__(256) void roi_align_kernel(Pyr p, const float* __restrict__ rois,
                                                        const int* __restrict__ levels, int K, int PH, int PW,
                                                        int sr, float* __restrict__ out_or_gout) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long bin = (long)blockIdx.x * 4 + wave;
  const long nbins = (long)K * PH * PW;
  if (bin >= nbins) return;
  const int pw = (int)(bin % PW);
  const int ph = (int)((bin / PW) % PH);
  const int k = (int)(bin / ((long)PW * PH));
  const int lv = levels[k];
  const float* r = rois + (long)k * 5;
  const int b = (int)r[0];
  const float scale = p.scale[lv];
  const int H = p.H[lv], W = p.W[lv], C = p.C;
  const float rsw = r[1] * scale, rsh = r[2] * scale, rew = r[3] * scale, reh = r[4] * scale;
  const float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
  const float bh = rh / (float)PH, bw = rw / (float)PW;
  const int gh = sr > 0 ? sr : (int)ceilf(rh / PH);
  const int gw = sr > 0 ? sr : (int)ceilf(rw / PW);
  const float count = (float)(gh * gw);
  const float* feat = HALF ? (const float*)((const unsigned short*)p.feat[lv] + (long)b * H * W * C) : p.feat[lv] + (long)b * H * W * C;
  float* gfeat = BWD ? p.grad[lv] + (long)b * H * W * C : nullptr;
  float* o = out_or_gout + bin * C;

  if (BWD && gh == 2 && gw == 2) {
    // Backward, 2x2 sampling grid (the only one the model uses).  The sample grid of a bin is a product grid, so
    // the bilinear weights factor:  d feat[y][x] += g/count * WY[y] * WX[x],  WY[y] = sum over the two sample rows
    // of their weight on row y.  Coinciding rows/columns of neighbouring samples are folded together
    // (wave-uniform scalar work), cutting the fp32 atomics per bin from 16 to typically 9 (box) / 4-9 (mask).
    int ry[4], rx[4];
    float wy[4], wx[4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      float y = rsh + ph * bh + (float)(i + .5f) * bh / 2.f;
      float x = rsw + pw * bw + (float)(i + .5f) * bw / 2.f;
      const bool yok = !(y < -1.0f || y > (float)H), xok = !(x < -1.0f || x > (float)W);
      if (y <= 0) y = 0;
      if (x <= 0) x = 0;
      int yl = (int)y, yh, xl = (int)x, xh;
      if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
      if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
      const float ly = y - yl, lx = x - xl;
      ry[2 * i] = yok ? yl : 0; ry[2 * i + 1] = yok ? yh : 0;
      wy[2 * i] = yok ? 1.f - ly : 0.f; wy[2 * i + 1] = yok ? ly : 0.f;
      rx[2 * i] = xok ? xl : 0; rx[2 * i + 1] = xok ? xh : 0;
      wx[2 * i] = xok ? 1.f - lx : 0.f; wx[2 * i + 1] = xok ? lx : 0.f;
    }
#pragma unroll
    for (int a = 1; a < 4; a++)
#pragma unroll
      for (int b = 0; b < a; b++) {
        if (ry[a] == ry[b] && wy[a] != 0.f) { wy[b] += wy[a]; wy[a] = 0.f; }
        if (rx[a] == rx[b] && wx[a] != 0.f) { wx[b] += wx[a]; wx[a] = 0.f; }
      }
    // lanes run over CONSECUTIVE channels (lane + 64 j): every atomic instruction then covers 256 contiguous bytes (two
    // full cache lines at the L2 atomic units) instead of 64 four-byte pieces spread over 1 KiB
    for (int cb = 0; cb < C; cb += 256) {
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int c = cb + lane + 64 * j;
        g[j] = c < C ? o[c] / count : 0.f;
      }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const float wgt = wy[a] * wx[b];
          if (wgt != 0.f) {
            float* dst = gfeat + ((long)ry[a] * W + rx[b]) * C + cb + lane;
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (cb + lane + 64 * j < C) atomicAdd(dst + 64 * j, g[j] * wgt);
          }
        }
    }
    return;
  }
  for (int c0 = lane * VEC; c0 < C; c0 += 64 * VEC) {
    float acc[VEC], g[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) { acc[j] = 0.f; g[j] = 0.f; }
    if (BWD) {
      if (VEC == 4) { const f32x4 t = *(const f32x4*)(o + c0); g[0] = t[0]; g[1 % VEC] = t[1]; g[2 % VEC] = t[2]; g[3 % VEC] = t[3]; }
      else g[0] = o[c0];
    }
    for (int iy = 0; iy < gh; iy++) {
      const float yy = rsh + ph * bh + (float)(iy + .5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ix++) {
        const float xx = rsw + pw * bw + (float)(ix + .5f) * bw / (float)gw;
        const Bilin q = bilin(H, W, yy, xx);
        const long o1 = ((long)q.yl * W + q.xl) * C + c0, o2 = ((long)q.yl * W + q.xh) * C + c0;
        const long o3 = ((long)q.yh * W + q.xl) * C + c0, o4 = ((long)q.yh * W + q.xh) * C + c0;
        if (!BWD) {
          float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
          if (VEC == 4 && HALF) {
            const unsigned short* fh = (const unsigned short*)feat;
            const uint2 h1 = *(const uint2*)(fh + o1), h2 = *(const uint2*)(fh + o2), h3 = *(const uint2*)(fh + o3), h4 = *(const uint2*)(fh + o4);
            auto wide = [](const uint2 t, float (&v)[VEC]) {
              v[0] = __builtin_bit_cast(float, t.x << 16); v[1 % VEC] = __builtin_bit_cast(float, t.x & 0xffff0000u);
              v[2 % VEC] = __builtin_bit_cast(float, t.y << 16); v[3 % VEC] = __builtin_bit_cast(float, t.y & 0xffff0000u);
            };
            wide(h1, v1); wide(h2, v2); wide(h3, v3); wide(h4, v4);
          } else if (VEC == 4) {
            const f32x4 a1 = *(const f32x4*)(feat + o1), a2 = *(const f32x4*)(feat + o2);
            const f32x4 a3 = *(const f32x4*)(feat + o3), a4 = *(const f32x4*)(feat + o4);
#pragma unroll
            for (int j = 0; j < VEC; j++) { v1[j] = a1[j]; v2[j] = a2[j]; v3[j] = a3[j]; v4[j] = a4[j]; }
          } else { v1[0] = feat[o1]; v2[0] = feat[o2]; v3[0] = feat[o3]; v4[0] = feat[o4]; }
#pragma unroll
          for (int j = 0; j < VEC; j++) acc[j] += q.w1 * v1[j] + q.w2 * v2[j] + q.w3 * v3[j] + q.w4 * v4[j];
        } else if (!q.empty) {
#pragma unroll
          for (int j = 0; j < VEC; j++) {
            atomicAdd(gfeat + o1 + j, g[j] * q.w1 / count);
            atomicAdd(gfeat + o2 + j, g[j] * q.w2 / count);
            atomicAdd(gfeat + o3 + j, g[j] * q.w3 / count);
            atomicAdd(gfeat + o4 + j, g[j] * q.w4 / count);
          }
        }
      }
    }
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < VEC; j++) acc[j] = acc[j] / count;
      if (VEC == 4) { f32x4 t; t[0] = acc[0]; t[1] = acc[1 % VEC]; t[2] = acc[2 % VEC]; t[3] = acc[3 % VEC]; *(f32x4*)(o + c0) = t; }
      else o[c0] = acc[0];
    }
  }
}

// ---- backward without atomics (round 5): one block per 8 x 8 pixel tile of one (level, image) map.  The block
// lists the ROIs of its image and level whose footprint can touch the tile (ordered compaction over k), then every wave walks the
// listed ROIs' bins in (k, ph, pw) order and adds  g / count * WY[y] * WX[x]  (the factored weights of the scatter kernel above) into
// ITS 64-channel slice of a tile accumulator in LDS -- no two waves share a cell, so plain read-modify-write, a fixed summation
// order, and every element of every level is written exactly once (zeros where no ROI reaches: the caller does not clear).
// Against the scatter form: 9 fp32 atomics per (bin, channel) + a 178 MB clear per call become one coalesced store per element --
// and the additions of a crowded tile become the work of ONE block: slower on the bench's proposals (see the entry point), opt-in.
struct TileMap { int first[5]; int tx[4], ty[4]; };   // first[l] = index of level l's first tile, first[4] = all tiles

__global__ __launch_bounds__(256) void roi_align_bwd_tiles_kernel(Pyr p, TileMap tm, const float* __restrict__ rois,
                                                                  const int* __restrict__ levels, int K, int PH, int PW,
                                                                  const float* __restrict__ gout) {
  extern __shared__ __attribute__((aligned(16))) float acc[];   // [64 pixels][C] then the ROI list (K ints) and 8 counters
  const int C = p.C;
  int* const list = (int*)(acc + 64 * C);
  int* const wcnt = list + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int lv = 0;
#pragma unroll
  for (int l = 1; l < 4; l++) if ((int)blockIdx.x >= tm.first[l]) lv = l;
  const int t = blockIdx.x - tm.first[lv];
  const int per = tm.tx[lv] * tm.ty[lv];
  const int b = t / per, ty0 = ((t % per) / tm.tx[lv]) * 8, tx0 = ((t % per) % tm.tx[lv]) * 8;
  const int H = p.H[lv], W = p.W[lv];
  const float scale = p.scale[lv];
  // ---- the tile's ROI list, ascending k
  int n = 0;
  for (int k0 = 0; k0 < K; k0 += 256) {
    const int k = k0 + tid;
    bool hit = false;
    if (k < K && levels[k] == lv) {
      const float* r = rois + (long)k * 5;
      if ((int)r[0] == b) {
        const float rsw = r[1] * scale, rsh = r[2] * scale, rew = r[3] * scale, reh = r[4] * scale;
        const float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
        // samples lie in (rsh, rsh + rh) x (rsw, rsw + rw); one at y touches rows floor(y), floor(y) + 1 (clamped into the map)
        const float ylo = floorf(rsh), yhi = floorf(rsh + rh) + 1.f, xlo = floorf(rsw), xhi = floorf(rsw + rw) + 1.f;
        hit = yhi >= (float)ty0 && ylo <= (float)(ty0 + 7) && xhi >= (float)tx0 && xlo <= (float)(tx0 + 7);
        hit = hit || (ty0 == 0 && yhi < 0.f && yhi >= -1.f) || (tx0 == 0 && xhi < 0.f && xhi >= -1.f);   // (clamped up to row / column 0)
        hit = hit || (ty0 + 8 >= H && ylo > (float)(H - 1)) || (tx0 + 8 >= W && xlo > (float)(W - 1));   // (clamped down to the last one)
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int base = n;
    for (int w = 0; w < wave; w++) base += wcnt[w];
    if (hit) list[base + __popcll(m & ((1ull << lane) - 1ull))] = k;
    n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  const int npx = 64 * C / 4;
  if (n == 0) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < npx; i += 256) {
      const int pix = i / (C / 4), y = ty0 + (pix >> 3), x = tx0 + (pix & 7);
      if (y < H && x < W) *(f32x4*)(p.grad[lv] + (((long)b * H + y) * W + x) * C + (i % (C / 4)) * 4) = z;
    }
    return;
  }
  for (int i = tid; i < npx; i += 256) ((f32x4*)acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const float count = 4.f;
  for (int q = 0; q < n; q++) {
    const int k = list[q];
    const float* r = rois + (long)k * 5;
    const float rsw = r[1] * scale, rsh = r[2] * scale, rew = r[3] * scale, reh = r[4] * scale;
    const float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
    const float bh = rh / (float)PH, bw = rw / (float)PW;
    // bins whose samples can touch the tile (one bin of slack on either side; the exact test is per sample below)
    int ph0 = (int)floorf(((float)(ty0 - 1) - rsh) / bh) - 1, ph1 = (int)floorf(((float)(ty0 + 8) - rsh) / bh) + 1;
    int pw0 = (int)floorf(((float)(tx0 - 1) - rsw) / bw) - 1, pw1 = (int)floorf(((float)(tx0 + 8) - rsw) / bw) + 1;
    if (ty0 == 0) ph0 = 0;
    if (tx0 == 0) pw0 = 0;
    if (ty0 + 8 >= H) ph1 = PH - 1;
    if (tx0 + 8 >= W) pw1 = PW - 1;
    ph0 = max(ph0, 0); pw0 = max(pw0, 0); ph1 = min(ph1, PH - 1); pw1 = min(pw1, PW - 1);
    for (int ph = ph0; ph <= ph1; ph++) {
      int ry[4]; float wy[4];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        float y = rsh + ph * bh + (float)(i + .5f) * bh / 2.f;
        const bool yok = !(y < -1.0f || y > (float)H);
        if (y <= 0) y = 0;
        int yl = (int)y, yh;
        if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
        const float ly = y - yl;
        ry[2 * i] = yok ? yl : 0; ry[2 * i + 1] = yok ? yh : 0;
        wy[2 * i] = yok ? 1.f - ly : 0.f; wy[2 * i + 1] = yok ? ly : 0.f;
      }
#pragma unroll
      for (int a = 1; a < 4; a++)
#pragma unroll
        for (int c = 0; c < a; c++)
          if (ry[a] == ry[c] && wy[a] != 0.f) { wy[c] += wy[a]; wy[a] = 0.f; }
      bool any = false;
#pragma unroll
      for (int a = 0; a < 4; a++) {
        ry[a] -= ty0;
        if (ry[a] < 0 || ry[a] > 7) wy[a] = 0.f;
        any = any || wy[a] != 0.f;
      }
      if (!any) continue;
      for (int pw = pw0; pw <= pw1; pw++) {
        int rx[4]; float wx[4];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          float x = rsw + pw * bw + (float)(i + .5f) * bw / 2.f;
          const bool xok = !(x < -1.0f || x > (float)W);
          if (x <= 0) x = 0;
          int xl = (int)x, xh;
          if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
          const float lx = x - xl;
          rx[2 * i] = xok ? xl : 0; rx[2 * i + 1] = xok ? xh : 0;
          wx[2 * i] = xok ? 1.f - lx : 0.f; wx[2 * i + 1] = xok ? lx : 0.f;
        }
#pragma unroll
        for (int a = 1; a < 4; a++)
#pragma unroll
          for (int c = 0; c < a; c++)
            if (rx[a] == rx[c] && wx[a] != 0.f) { wx[c] += wx[a]; wx[a] = 0.f; }
        bool anyx = false;
#pragma unroll
        for (int a = 0; a < 4; a++) {
          rx[a] -= tx0;
          if (rx[a] < 0 || rx[a] > 7) wx[a] = 0.f;
          anyx = anyx || wx[a] != 0.f;
        }
        if (!anyx) continue;
        const float* o = gout + (((long)k * PH + ph) * PW + pw) * C;
        for (int c = wave * 64 + lane; c < C; c += 256) {
          const float g = o[c] / count;
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const float wgt = wy[a] * wx[e];
              if (wgt != 0.f) acc[(ry[a] * 8 + rx[e]) * C + c] += g * wgt;
            }
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < npx; i += 256) {
    const int pix = i / (C / 4), y = ty0 + (pix >> 3), x = tx0 + (pix & 7);
    if (y < H && x < W) *(f32x4*)(p.grad[lv] + (((long)b * H + y) * W + x) * C + (i % (C / 4)) * 4) = ((const f32x4*)acc)[i];
  }
}

static int fill(Pyr& q, const mmt_pyramid* p) {
  if (!p || p->num_levels < 1 || p->num_levels > 4 || p->C < 1) return MMT_EINVAL;
  for (int l = 0; l < 4; l++) {
    int s = l < p->num_levels ? l : 0;
    q.feat[l] = p->feat[s]; q.grad[l] = p->grad_feat[s];
    q.H[l] = p->H[s]; q.W[l] = p->W[s]; q.scale[l] = p->scale[s];
  }
  q.N = p->N; q.C = p->C;
  return 0;
}

extern "C" int mmt_roi_align_forward(const mmt_pyramid* pyr, const float* rois, const int32_t* levels, int K,
                                     int PH, int PW, int sampling_ratio, float* out, void* stream) {
  Pyr q;
  int e = fill(q, pyr);
  if (e) return e;
  long nbins = (long)K * PH * PW;
  if (nbins == 0) return 0;
  if (q.C & 3)
    hipLaunchKernelGGL((roi_align_kernel<false, 1>), dim3(mmt_cdiv(nbins, 4)), dim3(256), 0, (hipStream_t)stream, q,
                       rois, levels, K, PH, PW, sampling_ratio, out);
  else
    hipLaunchKernelGGL((roi_align_kernel<false, 4>), dim3(mmt_cdiv(nbins, 4)), dim3(256), 0, (hipStream_t)stream, q,
                       rois, levels, K, PH, PW, sampling_ratio, out);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_roi_align_forward_bf16(const mmt_pyramid* pyr, const float* rois, const int32_t* levels, int K,
                                          int PH, int PW, int sampling_ratio, float* out, void* stream) {
  Pyr q;
  int e = fill(q, pyr);
  if (e) return e;
  if (q.C & 3) return MMT_EINVAL;
  long nbins = (long)K * PH * PW;
  if (nbins == 0) return 0;
  hipLaunchKernelGGL((roi_align_kernel<false, 4, true>), dim3(mmt_cdiv(nbins, 4)), dim3(256), 0, (hipStream_t)stream, q,
                     rois, levels, K, PH, PW, sampling_ratio, out);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_roi_align_backward(const mmt_pyramid* pyr, const float* rois, const int32_t* levels, int K,
                                      int PH, int PW, int sampling_ratio, const float* grad_out, void* stream) {
  Pyr q;
  int e = fill(q, pyr);
  if (e) return e;
  long nbins = (long)K * PH * PW;
  if (nbins == 0) return 0;
  if (q.C & 3)
    hipLaunchKernelGGL((roi_align_kernel<true, 1>), dim3(mmt_cdiv(nbins, 4)), dim3(256), 0, (hipStream_t)stream, q,
                       rois, levels, K, PH, PW, sampling_ratio, const_cast<float*>(grad_out));
  else
    hipLaunchKernelGGL((roi_align_kernel<true, 4>), dim3(mmt_cdiv(nbins, 4)), dim3(256), 0, (hipStream_t)stream, q,
                       rois, levels, K, PH, PW, sampling_ratio, const_cast<float*>(grad_out));
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_roi_align_backward_dense(const mmt_pyramid* pyr, const float* rois, const int32_t* levels, int K,
                                            int PH, int PW, int sampling_ratio, const float* grad_out, void* stream) {
  Pyr q;
  int e = fill(q, pyr);
  if (e) return e;
  // opt-in (MMT_ROI_BWD_DENSE=1): repeatable bit for bit, but proposals cluster on the objects -- a tile under 100 ROIs keeps one CU
  // busy for 0.4-6 ms while the scatter kernel spreads the same additions over the L2 atomic units of the whole chip in 0.2 ms
  // (profiles/r05_history.md): the step is 0.3 ms slower with it
  const char* env = getenv("MMT_ROI_BWD_DENSE");
  if (sampling_ratio != 2 || (q.C & 63) || q.C > 256 || K > 8192 || PH < 1 || PW < 1 || !env || !atoi(env)) return 1;   // not taken
  TileMap tm;
  int tiles = 0;
  for (int l = 0; l < 4; l++) {
    tm.first[l] = tiles;
    const bool live = l < pyr->num_levels;
    tm.tx[l] = live ? mmt_cdiv(q.W[l], 8) : 1;
    tm.ty[l] = live ? mmt_cdiv(q.H[l], 8) : 1;
    if (live) tiles += tm.tx[l] * tm.ty[l] * q.N;
  }
  tm.first[4] = tiles;
  for (int l = pyr->num_levels; l < 4; l++) tm.first[l] = tiles;   // (absent levels own no tile)
  if (tiles == 0) return 0;
  const size_t lds = (size_t)64 * q.C * 4 + (size_t)K * 4 + 32;
  if (lds > 160 * 1024) return 1;
  static bool done = false;
  if (!done) {
    const hipError_t er = hipFuncSetAttribute((const void*)roi_align_bwd_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (er != hipSuccess) return (int)er;
    done = true;
  }
  hipLaunchKernelGGL(roi_align_bwd_tiles_kernel, dim3(tiles), dim3(256), lds, (hipStream_t)stream, q, tm, rois, levels, K, PH, PW, grad_out);
  MMT_LAUNCH_CHECK();
  return 0;
}
