// On-device replacements for the two per-ROI CPU Python loops of the reference step:
//
//   paste_masks      <- mask_head/inference.py:29-65 (sigmoid, pick the predicted class),
//                       :120-206 (expand_masks / expand_boxes / paste_mask_in_image), :209-246 (Masker)
//                       and detector/generalized_rcnn.py:129-132 (sum over detections) -- the teacher's
//                       integral pseudo-mask, built with int atomics instead of D x (H,W) canvases.
//   polygon_targets  <- mask_head/loss.py:37-75 (project_masks_on_boxes),
//                       structures/segmentation_mask.py:96-133 (crop / resize / convert) and the
//                       rasteriser pycoco/maskApi.c:166-206 (rleFrPoly) + :53-74 (union merge).
//
// Built with -ffp-contract=off: thresholds (> 0.5) and floor/ceil decisions must round like the CPU code.
#include "common.h"

// ------------------------------------------------------------------------------------ paste
// STACK: the evaluator's form (mask_head/inference.py:209-246 as pap_eval.py:107-109 calls it): the input holds the
// PROBABILITIES of the predicted class ((D, 1, M, M): MaskPostProcessor's `mask` field), and every detection gets its own
// binary canvas stack[d] (bytes) instead of a vote in the integral map of its image.
template <bool STACK>
__global__ __launch_bounds__(256) void paste_kernel(const float* __restrict__ logits, const int* __restrict__ labels,
                                                    const float* __restrict__ boxes, const int* __restrict__ img,
                                                    int M, int NC, int IH, int IW, float thresh,
                                                    int* __restrict__ seg, unsigned char* __restrict__ stack) {
  extern __shared__ float pm[];  // (M+2)^2 padded probabilities
  const int d = blockIdx.x;
  if (!STACK && img[d] < 0) return;   // a row behind its image's count in a fixed-capacity detection list: no vote
  const int P = M + 2;
  const int lab = STACK ? 0 : labels[d];
  for (int i = threadIdx.x; i < P * P; i += 256) {
    const int y = i / P, x = i - y * P;
    float v = 0.f;
    if (y >= 1 && y <= M && x >= 1 && x <= M) {
      const float z = logits[(((long)d * M + (y - 1)) * M + (x - 1)) * NC + lab];
      v = STACK ? z : 1.f / (1.f + expf(-z));
    }
    pm[i] = v;
  }
  __syncthreads();
  // expand_boxes (inference.py:120-135) with scale = (M+2)/M, then .to(int32) (truncation)
  const float scale = (float)(M + 2) / (float)M;
  const float bx0 = boxes[d * 4 + 0], by0 = boxes[d * 4 + 1], bx1 = boxes[d * 4 + 2], by1 = boxes[d * 4 + 3];
  float wh = (bx1 - bx0) * .5f, hh = (by1 - by0) * .5f;
  const float xc = (bx1 + bx0) * .5f, yc = (by1 + by0) * .5f;
  wh *= scale; hh *= scale;
  const int x0 = (int)(xc - wh), x1 = (int)(xc + wh), y0 = (int)(yc - hh), y1 = (int)(yc + hh);
  const int w = max(x1 - x0 + 1, 1), h = max(y1 - y0 + 1, 1);
  const int cx0 = max(x0, 0), cx1 = min(x1 + 1, IW), cy0 = max(y0, 0), cy1 = min(y1 + 1, IH);
  const int cw = cx1 - cx0, ch = cy1 - cy0;
  if (cw <= 0 || ch <= 0) return;
  // F.interpolate(bilinear, align_corners=False): src = (dst+0.5)*in/out - 0.5 clamped at 0
  const float sy = (float)P / (float)h, sx = (float)P / (float)w;
  int* out = STACK ? nullptr : seg + (long)img[d] * IH * IW;
  unsigned char* outb = STACK ? stack + (long)d * IH * IW : nullptr;
  for (int i = threadIdx.x; i < cw * ch; i += 256) {
    const int yy = cy0 + i / cw, xx = cx0 + i % cw;
    const int dy = yy - y0, dx = xx - x0;
    float fy = sy * ((float)dy + 0.5f) - 0.5f;
    if (fy < 0.f) fy = 0.f;
    float fx = sx * ((float)dx + 0.5f) - 0.5f;
    if (fx < 0.f) fx = 0.f;
    const int iy0 = (int)fy, ix0 = (int)fx;
    const int iy1 = iy0 + (iy0 < P - 1 ? 1 : 0), ix1 = ix0 + (ix0 < P - 1 ? 1 : 0);
    const float ly1 = fy - (float)iy0, ly0 = 1.f - ly1, lx1 = fx - (float)ix0, lx0 = 1.f - lx1;
    const float v = ly0 * (lx0 * pm[iy0 * P + ix0] + lx1 * pm[iy0 * P + ix1]) +
                    ly1 * (lx0 * pm[iy1 * P + ix0] + lx1 * pm[iy1 * P + ix1]);
    if (v > thresh) {
      if (STACK) outb[(long)yy * IW + xx] = 1;
      else atomicAdd(out + (long)yy * IW + xx, 1);
    }
  }
}

extern "C" int mmt_paste_masks(const float* logits, const int32_t* labels, const float* boxes, const int32_t* img,
                               int D, int M, int NC, int IH, int IW, float thresh, int32_t* seg, void* stream) {
  if (D <= 0) return 0;
  const size_t lds = (size_t)(M + 2) * (M + 2) * sizeof(float);
  hipLaunchKernelGGL(paste_kernel<false>, dim3(D), dim3(256), lds, (hipStream_t)stream, logits, labels, boxes, img, M, NC, IH,
                     IW, thresh, seg, (unsigned char*)nullptr);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_paste_mask_stack(const float* prob, const float* boxes, int D, int M, int IH, int IW, float thresh,
                                    uint8_t* stack, void* stream) {
  if (D <= 0) return 0;
  if (!prob || !boxes || !stack || M <= 0 || IH <= 0 || IW <= 0) return MMT_EINVAL;
  const size_t lds = (size_t)(M + 2) * (M + 2) * sizeof(float);
  hipLaunchKernelGGL(paste_kernel<true>, dim3(D), dim3(256), lds, (hipStream_t)stream, prob, (const int*)nullptr, boxes,
                     (const int*)nullptr, M, 1, IH, IW, thresh, (int*)nullptr, stack);
  MMT_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------ polygons
// rleFrPoly builds, from the x5-upsampled dense boundary walk, the sorted list of column crossings
// a_j = x*h + y and emits alternating run lengths; zero-length runs are folded.  That is exactly
//   mask[q] = (#{j : a_j <= q}) mod 2     (q = column-major pixel index)
// so the sort + run-length stage is replaced by a parity count over an (unsorted) crossing list kept
// in LDS.  One wave per ROI; lanes own polygon edges for the boundary walk and pixels for the fill.
#define POLY_CAP 1536

__global__ __launch_bounds__(64) void polygon_kernel(const float* __restrict__ xy, const int* __restrict__ poly_off,
                                                     const int* __restrict__ roi_poly, const float* __restrict__ boxes,
                                                     int M, float* __restrict__ out, int* __restrict__ overflow) {
  __shared__ unsigned cross[POLY_CAP];
  __shared__ int ncross;
  const int p = blockIdx.x, lane = threadIdx.x;
  const int h = M, w = M;
  // crop + resize of structures/segmentation_mask.py:96-120 in float32, ratio cast like `tensor * python_float`
  const float b0 = boxes[p * 4 + 0], b1 = boxes[p * 4 + 1], b2 = boxes[p * 4 + 2], b3 = boxes[p * 4 + 3];
  float bw = b2 - b0, bh = b3 - b1;
  if (!(bw >= 1.f)) bw = 1.f;  // max(w, 1)
  if (!(bh >= 1.f)) bh = 1.f;
  const float rw = (float)((double)M / (double)bw), rh = (float)((double)M / (double)bh);
  const int npix = h * w;
  // per-pixel accumulated mask (union over polygons): each lane owns pixels lane, lane+64, ...
  unsigned char acc[16];  // ceil(28*28/64)=13
  const int per = (npix + 63) / 64;
  for (int i = 0; i < per; i++) acc[i] = 0;

  for (int pi = roi_poly[2 * p]; pi < roi_poly[2 * p + 1]; pi++) {
    const int v0 = poly_off[pi], k = poly_off[pi + 1] - v0;
    if (lane == 0) ncross = 0;
    __syncthreads();
    const double scale = 5;
    for (int j = lane; j < k; j += 64) {
      const int j1 = (j + 1 == k) ? 0 : j + 1;
      const float fxs = (xy[(v0 + j) * 2 + 0] - b0) * rw, fys = (xy[(v0 + j) * 2 + 1] - b1) * rh;
      const float fxe = (xy[(v0 + j1) * 2 + 0] - b0) * rw, fye = (xy[(v0 + j1) * 2 + 1] - b1) * rh;
      int xs = (int)(scale * (double)fxs + .5), ys = (int)(scale * (double)fys + .5);
      int xe = (int)(scale * (double)fxe + .5), ye = (int)(scale * (double)fye + .5);
      const int dx = abs(xe - xs), dy = abs(ys - ye);
      const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
      if (flip) { int t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
      const double s = dx >= dy ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;
      const int n = (dx >= dy ? dx : dy);
      int pu = 0, pv = 0;
      for (int d = 0; d <= n; d++) {
        const int t = flip ? n - d : d;
        int u, v;
        if (dx >= dy) { u = t + xs; v = (int)(ys + s * t + .5); }
        else { v = t + ys; u = (int)(xs + s * t + .5); }
        if (d > 0 && u != pu) {
          double xd = (double)(u < pu ? u : u - 1);
          xd = (xd + .5) / scale - .5;
          if (!(floor(xd) != xd || xd < 0 || xd > w - 1)) {
            double yd = (double)(v < pv ? v : pv);
            yd = (yd + .5) / scale - .5;
            if (yd < 0) yd = 0; else if (yd > h) yd = h;
            yd = ceil(yd);
            const int slot = atomicAdd(&ncross, 1);
            if (slot < POLY_CAP) cross[slot] = (unsigned)((int)xd * h + (int)yd);
          }
        }
        pu = u; pv = v;
      }
    }
    __syncthreads();
    int nc = ncross;
    if (nc > POLY_CAP) { if (lane == 0) atomicExch(overflow, 1); nc = POLY_CAP; }
    for (int i = 0; i < per; i++) {
      const int q = lane + 64 * i;  // column-major index q = x*h + y
      if (q >= npix) break;
      unsigned cnt = 0;
      for (int j = 0; j < nc; j++) cnt += (cross[j] <= (unsigned)q) ? 1u : 0u;
      acc[i] |= (unsigned char)(cnt & 1u);
    }
    __syncthreads();
  }
  for (int i = 0; i < per; i++) {
    const int q = lane + 64 * i;
    if (q >= npix) break;
    const int x = q / h, y = q - x * h;
    out[((long)p * h + y) * w + x] = acc[i] ? 1.f : 0.f;
  }
}

extern "C" int mmt_polygon_targets(const float* poly_xy, const int32_t* poly_off, const int32_t* roi_poly,
                                   const float* boxes, int P, int M, float* out, int32_t* overflow, void* stream) {
  if (P <= 0) return 0;
  if (M * M > 16 * 64) return MMT_EINVAL;
  hipLaunchKernelGGL(polygon_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, poly_xy, poly_off, roi_poly, boxes, M,
                     out, overflow);
  MMT_LAUNCH_CHECK();
  return 0;
}
