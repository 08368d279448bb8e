// Fused loss kernels of the mean-teacher hot path (all HBM-bound; value and gradient come from one
// or two streaming passes instead of the ~100 small ATen launches the reference issues).
//
//   mask_bce     <- mask_head/loss.py:177-179   F.binary_cross_entropy_with_logits(logits[pos,label], tgt)
//   mgd_level_*  <- detector/generalized_rcnn.py:243-282  fg_hint_loss (MGD), one level x all teachers
//   mask_pool    <- generalized_rcnn.py:259-264  adaptive_avg_pool2d(mask) binarised at 0.5
//   psm_rows     <- box_head/loss.py:185-237,267-287,311-315  evaluatePSM / cls_loss / sharpen
//   psm_variance <- box_head/loss.py:164-173,191-194  std over the K teacher views of softmax probs
#include "common.h"

// ----------------------------------------------------------------------------- mask BCE
__global__ __launch_bounds__(256) void mask_bce_kernel(const float* __restrict__ logits,
                                                       const int* __restrict__ labels,
                                                       const float* __restrict__ tgt, long total, int HW, int NC,
                                                       float inv_n, float gscale, float* __restrict__ loss,
                                                       float* __restrict__ grad) {
  float part = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int p = (int)(i / HW);
    const int lab = labels[p];
    const float x = logits[i * NC + lab];
    const float t = tgt[i];
    // (1-t)*x + max(-x,0) + log(1 + exp(-|x|))
    part += (1.f - t) * x + fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));
    const float sg = 1.f / (1.f + expf(-x));
    for (int c = 0; c < NC; c++) grad[i * NC + c] = (c == lab) ? (sg - t) * inv_n * gscale : 0.f;
  }
  part = wave_sum(part);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}

extern "C" int mmt_mask_bce(const float* logits, const int32_t* labels, const float* targets, int P, int HW,
                            int NC, float grad_scale, float* loss, float* grad, void* stream) {
  const long total = (long)P * HW;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mask_bce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, labels, targets,
                     total, HW, NC, 1.f / (float)total, grad_scale, loss, grad);
  MMT_LAUNCH_CHECK();
  return 0;
}

// ----------------------------------------------------------------------------- MGD
struct MgdT {
  const float* t[8];
  int flip[8];
  int nt;
};

template <bool BWD>
__global__ __launch_bounds__(256) void mgd_kernel(const float* __restrict__ s, MgdT T, const float* __restrict__ m,
                                                  long npix, int W, int C4, float* __restrict__ acc,
                                                  const float* __restrict__ coef, float* __restrict__ grad) {
  // one thread per (pixel, float4 of channels)
  float num[8];
#pragma unroll
  for (int i = 0; i < 8; i++) num[i] = 0.f;
  float msum = 0.f;
  float cf[8];
  if (BWD) {
#pragma unroll
    for (int i = 0; i < 8; i++) cf[i] = i < T.nt ? coef[i] : 0.f;
  }
  const long total = npix * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / C4;
    const int c4 = (int)(i - pix * C4);
    const float mk = m[pix];
    const int w = (int)(pix % W);
    const long fpix = pix - w + (W - 1 - w);
    const f32x4 sv = ((const f32x4*)s)[i];
    if (!BWD && c4 == 0) msum += mk;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k >= T.nt) break;
      const f32x4 tv = ((const f32x4*)T.t[k])[(T.flip[k] ? fpix : pix) * C4 + c4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float d = sv[e] - tv[e];
        if (BWD) g[e] += cf[k] * d; else num[k] += d * d * mk;
      }
    }
    if (BWD) {
#pragma unroll
      for (int e = 0; e < 4; e++) g[e] *= 2.f * mk;
      ((f32x4*)grad)[i] = g;
    }
  }
  if (!BWD) {
    __shared__ float red[4][9];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float v = wave_sum(num[k]);
      if ((threadIdx.x & 63) == 0) red[wv][k] = v;
    }
    const float ms = wave_sum(msum);
    if ((threadIdx.x & 63) == 0) red[wv][8] = ms;
    __syncthreads();
    if (threadIdx.x < 9) {
      const int k = threadIdx.x;
      const float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
      if (k == 8) atomicAdd(acc + T.nt, v);
      else if (k < T.nt) atomicAdd(acc + k, v);
    }
  }
}

static int mgd_launch(bool bwd, const float* s, const mmt_mgd_teachers* T, const float* m, int N, int H, int W,
                      int C, float* acc, const float* coef, float* grad, void* stream) {
  if (!T || T->nt < 1 || T->nt > 8 || (C & 3)) return MMT_EINVAL;
  MgdT q;
  for (int i = 0; i < 8; i++) { q.t[i] = T->t[i < T->nt ? i : 0]; q.flip[i] = T->flip[i < T->nt ? i : 0]; }
  q.nt = T->nt;
  const long npix = (long)N * H * W;
  const long total = npix * (C / 4);
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (bwd)
    hipLaunchKernelGGL(mgd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, q, m, npix, W, C / 4,
                       acc, coef, grad);
  else
    hipLaunchKernelGGL(mgd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, q, m, npix, W, C / 4,
                       acc, coef, grad);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_mgd_level_forward(const float* s, const mmt_mgd_teachers* T, const float* m, int N, int H, int W,
                                     int C, float* acc, void* stream) {
  return mgd_launch(false, s, T, m, N, H, W, C, acc, nullptr, nullptr, stream);
}
extern "C" int mmt_mgd_level_backward(const float* s, const mmt_mgd_teachers* T, const float* m, int N, int H,
                                      int W, int C, const float* coef, float* grad_s, void* stream) {
  return mgd_launch(true, s, T, m, N, H, W, C, nullptr, coef, grad_s, stream);
}

__global__ __launch_bounds__(256) void mask_pool_kernel(const int* __restrict__ seg, int N, int IH, int IW, int H,
                                                        int W, float* __restrict__ m) {
  // one wave per output pixel, lanes sweep the pooling window; the sum of small integers is exact in fp32 in any
  // order, so this equals adaptive_avg_pool2d's sequential sum bit for bit
  const long total = (long)N * H * W;
  const int lane = threadIdx.x & 63;
  for (long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6); i < total; i += (long)gridDim.x * 4) {
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const int n = (int)(i / ((long)W * H));
    // adaptive pooling window: [floor(h*IH/H), ceil((h+1)*IH/H))
    const int h0 = (int)(((long)h * IH) / H), h1 = (int)((((long)h + 1) * IH + H - 1) / H);
    const int w0 = (int)(((long)w * IW) / W), w1 = (int)((((long)w + 1) * IW + W - 1) / W);
    const int ww = w1 - w0, cnt = (h1 - h0) * ww;
    float sum = 0.f;
    for (int k = lane; k < cnt; k += 64) {
      const int y = h0 + k / ww, x = w0 + k % ww;
      sum += (float)seg[((long)n * IH + y) * IW + x];
    }
    sum = wave_sum(sum);
    if (lane == 0) m[i] = (sum / (float)cnt) > 0.5f ? 1.f : 0.f;
  }
}

extern "C" int mmt_mask_pool(const int32_t* seg, int N, int IH, int IW, int H, int W, float* m, void* stream) {
  const long total = (long)N * H * W;
  if (total == 0) return 0;
  int blocks = (int)((total + 3) / 4);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(mask_pool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seg, N, IH, IW, H, W, m);
  MMT_LAUNCH_CHECK();
  return 0;
}

// ----------------------------------------------------------------------------- PSM
#define PSM_MAXC 16

__global__ __launch_bounds__(256) void psm_rows_kernel(const float* __restrict__ teacher, int K,
                                                       const float* __restrict__ student, int R, int NC,
                                                       const float* __restrict__ roww, float temp, int do_sharpen,
                                                       int kind, float* __restrict__ rowloss,
                                                       float* __restrict__ rowgrad) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float w = roww[r];
  float t[PSM_MAXC], sl[PSM_MAXC];
  float tmax = -INFINITY, smax = -INFINITY;
  for (int c = 0; c < NC; c++) {
    float a = 0.f;
    for (int k = 0; k < K; k++) a += teacher[((long)k * R + r) * NC + c];
    t[c] = a / (float)K;
    tmax = fmaxf(tmax, t[c]);
    sl[c] = student[(long)r * NC + c];
    smax = fmaxf(smax, sl[c]);
  }
  if (kind == 2) {  // 'mse' (box_head/loss.py:273-274): squared error against the mean teacher LOGITS
    float l2 = 0.f;
    for (int c = 0; c < NC; c++) {
      const float d = sl[c] - t[c];
      l2 += d * d;
      rowgrad[(long)r * NC + c] = w * 2.f * d;
    }
    rowloss[r] = w * l2;
    return;
  }
  float tsum = 0.f, ssum = 0.f;
  for (int c = 0; c < NC; c++) { t[c] = expf(t[c] - tmax); tsum += t[c]; ssum += expf(sl[c] - smax); }
  const float lse = smax + logf(ssum);
  for (int c = 0; c < NC; c++) t[c] /= tsum;
  if (kind == 0 && do_sharpen) {
    float ps = 0.f;
    for (int c = 0; c < NC; c++) { t[c] = powf(t[c], 1.f / temp); ps += t[c]; }
    for (int c = 0; c < NC; c++) t[c] /= ps;
  }
  float l = 0.f;
  for (int c = 0; c < NC; c++) {
    const float logp = sl[c] - lse;
    if (kind == 0) l += -t[c] * logp;
    else l += t[c] > 0.f ? t[c] * (logf(t[c]) - logp) : 0.f;
    rowgrad[(long)r * NC + c] = w * (expf(logp) - t[c]);
  }
  rowloss[r] = w * l;
}

extern "C" int mmt_psm_rows(const float* teacher, int Kaug, const float* student, int R, int NC, const float* roww,
                            float temp, int sharpen, int kind, float* rowloss, float* rowgrad, void* stream) {
  if (NC > PSM_MAXC || Kaug < 1) return MMT_EINVAL;
  if (R == 0) return 0;
  hipLaunchKernelGGL(psm_rows_kernel, dim3(mmt_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, teacher, Kaug,
                     student, R, NC, roww, temp, sharpen, kind, rowloss, rowgrad);
  MMT_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void psm_var_kernel(const float* __restrict__ teacher, int K, int R, int NC,
                                                      int use_softmax, float* __restrict__ v) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float mean[PSM_MAXC], m2[PSM_MAXC];
  for (int c = 0; c < NC; c++) { mean[c] = 0.f; m2[c] = 0.f; }
  // two-pass: probabilities are recomputed (K*NC tiny)
  for (int pass = 0; pass < 2; pass++) {
    for (int k = 0; k < K; k++) {
      float p[PSM_MAXC];
      float mx = -INFINITY, sm = 0.f;
      for (int c = 0; c < NC; c++) { p[c] = teacher[((long)k * R + r) * NC + c]; mx = fmaxf(mx, p[c]); }
      if (use_softmax) for (int c = 0; c < NC; c++) { p[c] = expf(p[c] - mx); sm += p[c]; }
      else sm = 1.f;
      for (int c = 0; c < NC; c++) {
        const float q = p[c] / sm;
        if (pass == 0) mean[c] += q; else { const float d = q - mean[c]; m2[c] += d * d; }
      }
    }
    if (pass == 0) for (int c = 0; c < NC; c++) mean[c] /= (float)K;
  }
  float out = 0.f;
  for (int c = 0; c < NC; c++) out += sqrtf(m2[c] / (float)(K - 1));
  v[r] = out;
}

extern "C" int mmt_psm_variance(const float* teacher, int Kaug, int R, int NC, int use_softmax, float* v,
                                void* stream) {
  if (NC > PSM_MAXC || Kaug < 2) return MMT_EINVAL;
  if (R == 0) return 0;
  hipLaunchKernelGGL(psm_var_kernel, dim3(mmt_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, teacher, Kaug, R, NC,
                     use_softmax, v);
  MMT_LAUNCH_CHECK();
  return 0;
}

// ----------------------------------------------------------------------------- RPN loss
// rpn/loss.py:183-194 over all anchors of the batch with the sampler's masks instead of gathers:
//   n = max(#(pos | neg), 1);  objectness = sum_{pos | neg} BCEWithLogits(obj, max(label, 0)) / n;
//   box = sum_{pos} smooth_l1(reg - regt, beta, sum) / n
// two streaming passes (the gradients need n): sums, then losses + unit gradients -- instead of ~22 elementwise / reduction
// launches forward and ~25 backward.
__global__ __launch_bounds__(256) void rpn_loss_sums_kernel(const float* __restrict__ obj, const float4* __restrict__ reg,
                                                            const float* __restrict__ labels, const float4* __restrict__ regt,
                                                            const uint8_t* __restrict__ pos, const uint8_t* __restrict__ neg,
                                                            long R, float beta, float* __restrict__ sums) {
  float cnt = 0.f, bce = 0.f, box = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < R; i += (long)gridDim.x * 256) {
    const bool p = pos[i] != 0, s = p || neg[i] != 0;
    if (s) {
      const float x = obj[i], t = fmaxf(labels[i], 0.f);
      cnt += 1.f;
      bce += (1.f - t) * x + fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));
    }
    if (p) {
      const float4 a = reg[i], b = regt[i];
      const float d[4] = {fabsf(a.x - b.x), fabsf(a.y - b.y), fabsf(a.z - b.z), fabsf(a.w - b.w)};
#pragma unroll
      for (int j = 0; j < 4; ++j) box += d[j] < beta ? 0.5f * d[j] * d[j] / beta : d[j] - 0.5f * beta;
    }
  }
  cnt = wave_sum(cnt);
  bce = wave_sum(bce);
  box = wave_sum(box);
  __shared__ float red[4][3];
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][0] = cnt;
    red[threadIdx.x >> 6][1] = bce;
    red[threadIdx.x >> 6][2] = box;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (v != 0.f) atomicAdd(sums + threadIdx.x, v);
  }
}

__global__ __launch_bounds__(256) void rpn_loss_grad_kernel(const float* __restrict__ obj, const float4* __restrict__ reg,
                                                            const float* __restrict__ labels, const float4* __restrict__ regt,
                                                            const uint8_t* __restrict__ pos, const uint8_t* __restrict__ neg,
                                                            long R, float beta, const float* __restrict__ sums,
                                                            float* __restrict__ out, float* __restrict__ dobj,
                                                            float4* __restrict__ dreg) {
  const float inv = 1.f / fmaxf(sums[0], 1.f);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = sums[1] * inv;
    out[1] = sums[2] * inv;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < R; i += (long)gridDim.x * 256) {
    const bool p = pos[i] != 0, s = p || neg[i] != 0;
    float g = 0.f;
    if (s) {
      const float x = obj[i], t = fmaxf(labels[i], 0.f);
      g = (1.f / (1.f + expf(-x)) - t) * inv;
    }
    dobj[i] = g;
    float4 o = {0.f, 0.f, 0.f, 0.f};
    if (p) {
      const float4 a = reg[i], b = regt[i];
      const float e[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
      float r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = fabsf(e[j]);
        r[j] = (d < beta ? e[j] / beta : (e[j] > 0.f ? 1.f : (e[j] < 0.f ? -1.f : 0.f))) * inv;
      }
      o = {r[0], r[1], r[2], r[3]};
    }
    dreg[i] = o;
  }
}

extern "C" int mmt_rpn_loss(const float* obj, const float* reg, const float* labels, const float* regt, const uint8_t* pos,
                            const uint8_t* neg, long R, float beta, float* sums, float* out, float* dobj, float* dreg,
                            void* stream) {
  if (!obj || !reg || !labels || !regt || !pos || !neg || !sums || !out || !dobj || !dreg || R < 1 || !(beta > 0.f) ||
      (((size_t)reg | (size_t)regt | (size_t)dreg) & 15))
    return MMT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, 3 * sizeof(float), s) != hipSuccess) return MMT_EINVAL;
  int blocks = (int)((R + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rpn_loss_sums_kernel, dim3(blocks), dim3(256), 0, s, obj, (const float4*)reg, labels, (const float4*)regt, pos,
                     neg, R, beta, sums);
  MMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(rpn_loss_grad_kernel, dim3(blocks), dim3(256), 0, s, obj, (const float4*)reg, labels, (const float4*)regt, pos,
                     neg, R, beta, sums, out, dobj, (float4*)dreg);
  MMT_LAUNCH_CHECK();
  return 0;
}

// ----------------------------------------------------------------------------- box-head loss
// box_head/loss.py:118-162: classification = mean_i CE(logits_i, label_i); box = sum_{label > 0} smooth_l1(breg[i, 4 label ..] -
// regt_i, beta = 1, sum) / R.  One launch: both values (out[2], zeroed here) and the unit gradients of both inputs.
__global__ __launch_bounds__(256) void box_loss_kernel(const float* __restrict__ logits, const float* __restrict__ breg,
                                                       const int64_t* __restrict__ labels, const float* __restrict__ regt, int R,
                                                       int NC, float* __restrict__ out, float* __restrict__ dlogits,
                                                       float* __restrict__ dbreg, const int64_t* __restrict__ n_rows) {
  // n_rows (device, or null = R): the rows that count -- fixed-capacity lists carry rows labelled -1 behind an image's sampled set
  // (box_head.py::subsample_fixed); those contribute nothing and are not rows of the mean
  const float inv = 1.f / (n_rows ? fmaxf((float)*n_rows, 1.f) : (float)R);
  float ce = 0.f, box = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < R; i += gridDim.x * 256) {
    const float* l = logits + (long)i * NC;
    const int lab = (int)labels[i];
    if (lab < 0) {
      for (int c = 0; c < NC; ++c) dlogits[(long)i * NC + c] = 0.f;
      for (int c = 0; c < 4 * NC; ++c) dbreg[(long)i * 4 * NC + c] = 0.f;
      continue;
    }
    float mx = l[0];
    for (int c = 1; c < NC; ++c) mx = fmaxf(mx, l[c]);
    float se = 0.f;
    for (int c = 0; c < NC; ++c) se += expf(l[c] - mx);
    const float lse = mx + logf(se);
    ce += lse - l[lab];
    for (int c = 0; c < NC; ++c) dlogits[(long)i * NC + c] = (expf(l[c] - lse) - (c == lab ? 1.f : 0.f)) * inv;
    float* db = dbreg + (long)i * 4 * NC;
    for (int c = 0; c < 4 * NC; ++c) db[c] = 0.f;
    if (lab > 0) {
      for (int j = 0; j < 4; ++j) {
        const float e = breg[(long)i * 4 * NC + 4 * lab + j] - regt[(long)i * 4 + j], d = fabsf(e);
        box += d < 1.f ? 0.5f * d * d : d - 0.5f;
        db[4 * lab + j] = (d < 1.f ? e : (e > 0.f ? 1.f : -1.f)) * inv;
      }
    }
  }
  ce = wave_sum(ce);
  box = wave_sum(box);
  __shared__ float red[4][2];
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][0] = ce;
    red[threadIdx.x >> 6][1] = box;
  }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(out + threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * inv);
}

extern "C" int mmt_box_loss_rows(const float* logits, const float* breg, const int64_t* labels, const float* regt, int R, int NC,
                                 const int64_t* n_rows, float* out, float* dlogits, float* dbreg, void* stream) {
  if (!logits || !breg || !labels || !regt || !out || !dlogits || !dbreg || R < 1 || NC < 2) return MMT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, 2 * sizeof(float), s) != hipSuccess) return MMT_EINVAL;
  int blocks = (R + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(box_loss_kernel, dim3(blocks), dim3(256), 0, s, logits, breg, labels, regt, R, NC, out, dlogits, dbreg, n_rows);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_box_loss(const float* logits, const float* breg, const int64_t* labels, const float* regt, int R, int NC,
                            float* out, float* dlogits, float* dbreg, void* stream) {
  return mmt_box_loss_rows(logits, breg, labels, regt, R, NC, nullptr, out, dlogits, dbreg, stream);
}
