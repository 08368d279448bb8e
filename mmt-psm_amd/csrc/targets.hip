// Ground-truth assignment for anchors / proposals in one launch (two with low-quality matches).
//
// Replaces, per image, the chain boxlist_iou (structures/boxlist_ops.py:53-87) -> Matcher (modeling/matcher.py:37-139)
// -> label rules (rpn/loss.py:56-83, box_head/loss.py:38-80) -> BoxCoder.encode (modeling/box_coder.py:23-53) that the
// reference (and, until now, this port) runs as ~45 separate elementwise / reduction launches per image over a
// [G x A] IoU matrix.  Here one thread owns one candidate box and walks the (few) ground-truth boxes of its image.
// Arithmetic is the same expression order as the tensor code (built with -ffp-contract=off), so IoUs, the argmax and the
// regression targets are bit-identical to it.
#include "common.h"

namespace {
__device__ __forceinline__ float iou_pm1(const float* g, float garea, const float* c, float carea) {
  const float ltx = fmaxf(g[0], c[0]), lty = fmaxf(g[1], c[1]);
  const float rbx = fminf(g[2], c[2]), rby = fminf(g[3], c[3]);
  const float w = fmaxf(rbx - ltx + 1.f, 0.f), h = fmaxf(rby - lty + 1.f, 0.f);
  const float inter = w * h;
  return inter / (garea + carea - inter);
}

__device__ __forceinline__ int image_of(const int* __restrict__ off, int N, int a) {
  int n = 0;
  while (n + 1 < N && a >= off[n + 1]) n++;
  return n;
}

// top[g] = max over the candidates of g's image of IoU(g, candidate)   (IoU >= 0: uint order == float order).
// Block-level reduction first: one atomic per (block, gt) -- per-thread atomics on the dozen gt addresses serialised
// (0.84 ms for the 2 x 262k anchors of a batch; now ~20 us).
__global__ __launch_bounds__(256) void match_top_kernel(const float* __restrict__ cand, const int* __restrict__ cand_off,
                                                        const float* __restrict__ gt, const int* __restrict__ gt_off,
                                                        int N, int A_total, int shared_cand, unsigned* __restrict__ top) {
  __shared__ float red[4];
  const int a = blockIdx.x * 256 + threadIdx.x;
  const bool valid = a < A_total;
  const int n = image_of(cand_off, N, valid ? a : A_total - 1);
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    const float* cp = cand + (long)(shared_cand ? a - cand_off[n] : a) * 4;
    c[0] = cp[0]; c[1] = cp[1]; c[2] = cp[2]; c[3] = cp[3];
  }
  const float carea = (c[2] - c[0] + 1.f) * (c[3] - c[1] + 1.f);
  const int n_lo = image_of(cand_off, N, blockIdx.x * 256);
  const int n_hi = image_of(cand_off, N, min(blockIdx.x * 256 + 255, A_total - 1));
  for (int g = gt_off[n_lo]; g < gt_off[n_hi + 1]; g++) {   // block-uniform loop over the gts of the images it touches
    const float* gb = gt + (long)g * 4;
    float v = 0.f;
    if (valid && g >= gt_off[n] && g < gt_off[n + 1])
      v = iou_pm1(gb, (gb[2] - gb[0] + 1.f) * (gb[3] - gb[1] + 1.f), c, carea);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      if (m > 0.f) atomicMax(top + g, __float_as_uint(m));
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void match_kernel(const float* __restrict__ cand, const int* __restrict__ cand_off,
                                                    const float* __restrict__ gt, const int* __restrict__ gt_off,
                                                    const int64_t* __restrict__ gt_labels,
                                                    const uint8_t* __restrict__ visible, int N, int A_total,
                                                    int shared_cand, float high, float low, const unsigned* __restrict__ top,
                                                    float wx, float wy, float ww, float wh, int32_t* __restrict__ matches,
                                                    float* __restrict__ labels_f, int64_t* __restrict__ labels_i,
                                                    float* __restrict__ reg) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= A_total) return;
  const int n = image_of(cand_off, N, a);
  const int al = a - cand_off[n];
  const float* c = cand + (long)(shared_cand ? al : a) * 4;
  const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
  const float cc[4] = {c0, c1, c2, c3};
  const float carea = (c2 - c0 + 1.f) * (c3 - c1 + 1.f);
  const int g0 = gt_off[n], g1 = gt_off[n + 1];
  float best = -1.f;
  int arg = 0;
  bool is_best = false;
  for (int g = g0; g < g1; g++) {
    const float* gb = gt + (long)g * 4;
    const float v = iou_pm1(gb, (gb[2] - gb[0] + 1.f) * (gb[3] - gb[1] + 1.f), cc, carea);
    if (v > best) { best = v; arg = g - g0; }            // first maximum, as torch.max over dim 0
    if (top != nullptr && v == __uint_as_float(top[g])) is_best = true;
  }
  int m = arg;
  if (best < low) m = -1;                                  // Matcher.BELOW_LOW_THRESHOLD
  else if (best < high) m = -2;                            // Matcher.BETWEEN_THRESHOLDS
  if (top != nullptr && is_best) m = arg;                  // allow_low_quality_matches (matcher.py:118-139)
  matches[a] = m;
  const int mi = m < 0 ? 0 : m;
  if (labels_f != nullptr) {                               // RPN: 1 matched, 0 background, -1 ignored / not visible
    float l = m >= 0 ? 1.f : 0.f;
    if (visible != nullptr && !visible[shared_cand ? al : a]) l = -1.f;
    if (m == -2) l = -1.f;
    labels_f[a] = l;
  }
  if (labels_i != nullptr) {                               // box head: class of the matched gt, 0 background, -1 ignored
    int64_t l = gt_labels[g0 + mi];
    if (m == -1) l = 0;
    if (m == -2) l = -1;
    labels_i[a] = l;
  }
  if (reg != nullptr) {                                    // BoxCoder.encode(gt[mi], candidate)
    const float* gb = gt + (long)(g0 + mi) * 4;
    const float pw = c2 - c0 + 1.f, ph = c3 - c1 + 1.f;
    const float px = c0 + 0.5f * pw, py = c1 + 0.5f * ph;
    const float gw = gb[2] - gb[0] + 1.f, gh = gb[3] - gb[1] + 1.f;
    const float gx = gb[0] + 0.5f * gw, gy = gb[1] + 0.5f * gh;
    float* o = reg + (long)a * 4;
    o[0] = wx * (gx - px) / pw;
    o[1] = wy * (gy - py) / ph;
    o[2] = ww * logf(gw / pw);
    o[3] = wh * logf(gh / ph);
  }
}
// BoxCoder.decode (modeling/box_coder.py:52-95) for `ncls` boxes per row, + BoxList.clip_to_image(remove_empty=False)
// (structures/bounding_box.py:229-238) when `lim` is given: one thread per (row, class).  Same operations in the same order
// as the reference's CPU code (true divisions by the weights, every intermediate rounded to fp32, no FMA contraction).
__global__ __launch_bounds__(256) void box_decode_kernel(const float* __restrict__ codes, const float* __restrict__ boxes,
                                                         const int R, const int ncls, const float wx, const float wy,
                                                         const float ww, const float wh, const float clipv,
                                                         const int32_t* __restrict__ row_off, const float* __restrict__ lim,
                                                         const int n_img, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * ncls) return;
  const int r = (int)(i / ncls);
  const f32x4 b = *(const f32x4*)(boxes + (long)r * 4);
  const f32x4 c = *(const f32x4*)(codes + i * 4);
  const float w = b[2] - b[0] + 1.f, h = b[3] - b[1] + 1.f;
  const float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
  const float dx = c[0] / wx, dy = c[1] / wy;
  const float dw = fminf(c[2] / ww, clipv), dh = fminf(c[3] / wh, clipv);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  f32x4 o = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw - 1.f, pcy + 0.5f * ph - 1.f};
  if (lim) {
    int img = 0;
    while (img + 1 < n_img && r >= row_off[img + 1]) img++;
    const float lw = lim[2 * img], lh = lim[2 * img + 1];
    o[0] = fminf(fmaxf(o[0], 0.f), lw);
    o[1] = fminf(fmaxf(o[1], 0.f), lh);
    o[2] = fminf(fmaxf(o[2], 0.f), lw);
    o[3] = fminf(fmaxf(o[3], 0.f), lh);
  }
  *(f32x4*)(out + i * 4) = o;
}

// Pooler.convert_to_roi_format + LevelMapper (modeling/poolers.py:29-32, 91-104) for the boxes of all images of a call in one
// launch: rois[r] = (image, x1, y1, x2, y2), levels[r] = clamp(floor(lvl0 + log2(sqrt(area) / s0 + eps)), k_min, k_max) - k_min
// in the tensor code's fp32 expression order (area with the +1 convention; built with -ffp-contract=off).
struct RoiFmtArgs { const float* boxes[32]; int off[33]; int n_img; float s0, lvl0, eps, k_min, k_max; float* rois; int* levels; };
__global__ __launch_bounds__(256) void roi_format_kernel(const RoiFmtArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.off[a.n_img]) return;
  int n = 0;
  while (n + 1 < a.n_img && r >= a.off[n + 1]) n++;
  const f32x4 b = *(const f32x4*)(a.boxes[n] + (long)(r - a.off[n]) * 4);
  float* o = a.rois + (long)r * 5;
  o[0] = (float)n; o[1] = b[0]; o[2] = b[1]; o[3] = b[2]; o[4] = b[3];
  if (a.levels) {
    const float area = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
    const float s = sqrtf(area);
    float lv = floorf(a.lvl0 + log2f(s / a.s0 + a.eps));
    lv = fminf(fmaxf(lv, a.k_min), a.k_max);   // torch.clamp: NaN (area < 0 never occurs for xyxy boxes with x2 >= x1 - 1) aside
    a.levels[r] = (int)(long)lv - (int)a.k_min;
  }
}

// IR-Net relation NMS, regression labels of the ranked boxes of ONE image (reference relation_module.py:323-391: a D2H copy
// and numpy loops per class; the device tensor formulation of modeling/relation/relation_module.py::prepare_reg_label was
// ~100 launches per image).  One block per foreground class: IoU of the n ranked boxes of the class with every gt box,
// each box's best gt of the class (first maximal index), per threshold every gt's best-scoring box among the boxes that
// overlap it above the threshold and have it as their best gt (first maximal index), and for every box the IoU recorded by
// the FIRST gt that chose it -- numpy's tie rules, fp32 expressions in the tensor code's order.
constexpr int REL_MAX = 8192;   // n * G cells of the IoU matrix in LDS
__global__ __launch_bounds__(128) void relation_labels_kernel(const float* __restrict__ boxes /*[n][fg][4]*/,
                                                              const float* __restrict__ score /*[n][fg]*/,
                                                              const float* __restrict__ gt /*[G][4]*/, const long* __restrict__ gl /*[G]*/,
                                                              int n, int fg, int G, int T, float t0, float t1, float t2, float t3,
                                                              float* __restrict__ out /*[n][fg][T]*/) {
  __shared__ float iou[REL_MAX];
  __shared__ int best[128], msi[256];
  __shared__ float moi[256];
  // the ground truth, its class-membership flags and the class's score column once into LDS: the loops below walk them n x G
  // times (from global memory every one of those reads was a dependent cache access: 128 -> ~40 us per call)
  __shared__ float sgt[256 * 4], ssc[128];
  __shared__ int scm[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const float thr[4] = {t0, t1, t2, t3};
  for (int e = tid; e < G * 4; e += 128) sgt[e] = gt[e];
  for (int g = tid; g < G; g += 128) scm[g] = gl[g] == (long)(c + 1);
  for (int b = tid; b < n; b += 128) ssc[b] = score[(long)b * fg + c];
  __syncthreads();
  for (int b = tid; b < n; b += 128) {
    const float* bx = boxes + ((long)b * fg + c) * 4;
    const float a1 = (bx[2] - bx[0] + 1.f) * (bx[3] - bx[1] + 1.f);
    float bv = 0.f;
    int bi = -1;
    for (int g = 0; g < G; g++) {
      const float* t = sgt + g * 4;
      const float a2 = (t[2] - t[0] + 1.f) * (t[3] - t[1] + 1.f);
      const float w = fmaxf(fminf(bx[2], t[2]) - fmaxf(bx[0], t[0]) + 1.f, 0.f);
      const float h = fmaxf(fminf(bx[3], t[3]) - fmaxf(bx[1], t[1]) + 1.f, 0.f);
      const float inter = w * h;
      const float v = inter / (a1 + a2 - inter);
      iou[b * G + g] = v;
      const float vc = scm[g] ? v : -1.f;
      if (bi < 0 || vc > bv) { bv = vc; bi = g; }   // first maximal index
    }
    best[b] = bi;
  }
  __syncthreads();
  for (int k = 0; k < T; k++) {
    for (int g = tid; g < G; g += 128) {
      const bool cm = scm[g] != 0;
      float ms = 0.f;
      int mb = 0;
      bool any = false;
      for (int b = 0; b < n; b++) {
        const float v = iou[b * G + g];
        const float osc = (cm && v > thr[k] && best[b] == g) ? ssc[b] : 0.f;
        if (!any || osc > ms) { ms = osc; mb = b; any = true; }
      }
      msi[g] = mb;
      const float v = iou[mb * G + g];
      moi[g] = (cm && v > thr[k] && best[mb] == g) ? v : 0.f;
    }
    __syncthreads();
    for (int b = tid; b < n; b += 128) {
      bool valid = false;
      int first = G;
      for (int g = 0; g < G; g++) {
        if (!scm[g]) continue;
        if (iou[b * G + g] > thr[k]) valid = true;
        if (msi[g] == b && first == G) first = g;
      }
      out[((long)b * fg + c) * T + k] = (valid && first < G) ? moi[first] : 0.f;
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int mmt_box_decode(const float* codes, const float* boxes, int R, int ncls, float wx, float wy, float ww, float wh,
                              float clip, const int32_t* row_off, const float* lim, int n_img, float* out, void* stream) {
  if (!codes || !boxes || !out || R < 0 || ncls < 1 || (lim && (!row_off || n_img < 1))) return MMT_EINVAL;
  if (R == 0) return 0;
  hipLaunchKernelGGL(box_decode_kernel, dim3(mmt_cdiv((long)R * ncls, 256)), dim3(256), 0, (hipStream_t)stream, codes, boxes,
                     R, ncls, wx, wy, ww, wh, clip, row_off, lim, n_img, out);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_match_targets(const float* cand, const int32_t* cand_off, const float* gt, const int32_t* gt_off,
                                 const int64_t* gt_labels, const uint8_t* visible, int N, int A_total, int G_total,
                                 int shared_cand, float high, float low, int allow_low_quality, float wx, float wy, float ww,
                                 float wh, uint32_t* top_ws, int32_t* matches, float* labels_f, int64_t* labels_i, float* reg,
                                 void* stream) {
  if (!cand || !cand_off || !gt || !gt_off || !matches || N < 1 || (labels_i && !gt_labels)) return MMT_EINVAL;
  if (allow_low_quality && !top_ws) return MMT_EINVAL;
  if (A_total <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = mmt_cdiv(A_total, 256);
  if (allow_low_quality) {
    hipError_t e = hipMemsetAsync(top_ws, 0, (size_t)(G_total > 0 ? G_total : 1) * sizeof(uint32_t), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(match_top_kernel, dim3(blocks), dim3(256), 0, s, cand, cand_off, gt, gt_off, N, A_total, shared_cand,
                       top_ws);
    MMT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(match_kernel, dim3(blocks), dim3(256), 0, s, cand, cand_off, gt, gt_off, gt_labels, visible, N, A_total,
                     shared_cand, high, low, allow_low_quality ? top_ws : (const unsigned*)nullptr, wx, wy, ww, wh, matches,
                     labels_f, labels_i, reg);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_roi_format_levels(const float* const* boxes, const int32_t* counts, int n_img, float s0, float lvl0, float eps,
                                     int k_min, int k_max, float* rois, int32_t* levels, void* stream) {
  if (!boxes || !counts || n_img < 1 || n_img > 32 || !rois) return MMT_EINVAL;
  RoiFmtArgs a;
  a.off[0] = 0;
  for (int i = 0; i < n_img; i++) {
    if (counts[i] < 0 || (counts[i] > 0 && (!boxes[i] || ((size_t)boxes[i] & 15)))) return MMT_EINVAL;
    a.boxes[i] = boxes[i];
    a.off[i + 1] = a.off[i] + counts[i];
  }
  a.n_img = n_img; a.s0 = s0; a.lvl0 = lvl0; a.eps = eps; a.k_min = (float)k_min; a.k_max = (float)k_max;
  a.rois = rois; a.levels = levels;
  const int total = a.off[n_img];
  if (total == 0) return 0;
  hipLaunchKernelGGL(roi_format_kernel, dim3(mmt_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_relation_reg_labels(const float* boxes, const float* score, const float* gt, const int64_t* gt_labels, int n, int fg,
                                       int G, const float* thresholds, int T, float* out, void* stream) {
  if (!boxes || !score || !out || n < 1 || n > 128 || fg < 1 || T < 1 || T > 4 || G < 0 || G > 256 || (long)n * G > REL_MAX || (G && (!gt || !gt_labels)))
    return MMT_EINVAL;
  if (G == 0) return hipMemsetAsync(out, 0, (size_t)n * fg * T * sizeof(float), (hipStream_t)stream) == hipSuccess ? 0 : MMT_EINVAL;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < T; i++) t[i] = thresholds[i];
  hipLaunchKernelGGL(relation_labels_kernel, dim3(fg), dim3(128), 0, (hipStream_t)stream, boxes, score, gt, (const long*)gt_labels, n, fg, G, T,
                     t[0], t[1], t[2], t[3], out);
  MMT_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- IR-Net geometric position embedding
// extract_multi_position_matrix (relation/relation_module.py:93-135): for every class c and ordered box pair (i, j)
//   d = (log max(|cx_i - cx_j| / w_i, 1e-3), log max(|cy_i - cy_j| / h_i, 1e-3), log(w_i / w_j), log(h_i / h_j)),
//   out[c][i][j] = (sin(100 d_k f_m), k = 0..3, m = 0..F-1; then cos of the same) with f_m = wave_len^(-m / F), F = dim_g / 8
// -- thirty-odd elementwise launches of the tensor formulation in one; fp32 expressions in the tensor code's order (this file is
// built without FMA contraction).  One lane per output value: a wave covers the 8 F values of one pair.
__global__ __launch_bounds__(256) void position_embedding_kernel(const float* __restrict__ boxes /*[n][C][4]*/, int n, int C, int F,
                                                                 const float* __restrict__ freq /*[F]*/,
                                                                 float* __restrict__ out /*[C][n][n][8 F]*/) {
  const int D = 8 * F;
  const long total = (long)C * n * n * D;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int d = (int)(e % D);
    long t = e / D;
    const int j = (int)(t % n); t /= n;
    const int i = (int)(t % n);
    const int c = (int)(t / n);
    const float* bi = boxes + ((long)i * C + c) * 4;
    const float* bj = boxes + ((long)j * C + c) * 4;
    const int half = d / (4 * F), k = (d % (4 * F)) / F, m = d % F;
    float v;
    if (k == 0) {
      const float cxi = (bi[0] + bi[2]) * 0.5f, cxj = (bj[0] + bj[2]) * 0.5f, wi = (bi[2] - bi[0]) + 1.f;
      v = logf(fmaxf(fabsf((cxi - cxj) / wi), 1e-3f));
    } else if (k == 1) {
      const float cyi = (bi[1] + bi[3]) * 0.5f, cyj = (bj[1] + bj[3]) * 0.5f, hi = (bi[3] - bi[1]) + 1.f;
      v = logf(fmaxf(fabsf((cyi - cyj) / hi), 1e-3f));
    } else if (k == 2) {
      const float wi = (bi[2] - bi[0]) + 1.f, wj = (bj[2] - bj[0]) + 1.f;
      v = logf(wi / wj);
    } else {
      const float hi = (bi[3] - bi[1]) + 1.f, hj = (bj[3] - bj[1]) + 1.f;
      v = logf(hi / hj);
    }
    const float a = (100.f * v) * freq[m];
    out[e] = half ? cosf(a) : sinf(a);
  }
}

extern "C" int mmt_position_embedding(const float* boxes, int n, int C, int dim_g, const float* freq, float* out, void* stream) {
  if (!boxes || !freq || !out || n < 0 || C < 1 || dim_g < 8 || (dim_g & 7)) return MMT_EINVAL;
  if (n == 0) return 0;
  const long total = (long)C * n * n * dim_g;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(position_embedding_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, boxes, n, C, dim_g / 8, freq, out);
  MMT_LAUNCH_CHECK();
  return 0;
}
