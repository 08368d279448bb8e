/* ORACLE -- test infrastructure only.  Never linked, imported or executed by the product
 * (mmt-psm_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Plain-C restatement of the reference's native ops on the hot path.  Layout is the
 * REFERENCE's (NCHW, row-major), deliberately unlike the product's NHWC kernels.
 *
 *   orc_nms                 <- maskrcnn_benchmark/csrc/cpu/nms_cpu.cpp:5-68
 *   orc_roi_align_forward_* <- maskrcnn_benchmark/csrc/cpu/ROIAlign_cpu.cpp:17-219
 *   orc_roi_align_backward_*<- maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:126-254
 *                              (the reference has NO CPU backward, csrc/ROIAlign.h:44)
 *   orc_poly_mask           <- pycoco/maskApi.c:166-206 (rleFrPoly) + :47-51 (rleDecode)
 *                              + :53-74 (rleMerge, union) as used by
 *                              structures/segmentation_mask.py:122-133
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function here against golden
 * vectors captured from the reference's own compiled code (tests/golden/gen_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ NMS */
typedef struct { float s; int64_t i; } orc_si;
static int orc_si_cmp(const void* a, const void* b) {
  const orc_si* x = (const orc_si*)a; const orc_si* y = (const orc_si*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0); /* stable: ties keep index order */
}

/* dets [n,4] xyxy, scores [n]; keep_out [n] receives ascending original indices; returns count.
 * Greedy, "+1" areas, suppress on ovr >= thr (nms_cpu.cpp:22,56-60). */
int64_t orc_nms(const float* dets, const float* scores, int64_t n, float thr, int64_t* keep_out) {
  if (n <= 0) return 0;
  orc_si* ord = (orc_si*)malloc(sizeof(orc_si) * n);
  float* area = (float*)malloc(sizeof(float) * n);
  uint8_t* sup = (uint8_t*)calloc(n, 1);
  for (int64_t i = 0; i < n; i++) {
    ord[i].s = scores[i]; ord[i].i = i;
    area[i] = (dets[i * 4 + 2] - dets[i * 4 + 0] + 1) * (dets[i * 4 + 3] - dets[i * 4 + 1] + 1);
  }
  qsort(ord, n, sizeof(orc_si), orc_si_cmp);
  for (int64_t a = 0; a < n; a++) {
    int64_t i = ord[a].i;
    if (sup[i]) continue;
    float ix1 = dets[i * 4], iy1 = dets[i * 4 + 1], ix2 = dets[i * 4 + 2], iy2 = dets[i * 4 + 3];
    float ia = area[i];
    for (int64_t b = a + 1; b < n; b++) {
      int64_t j = ord[b].i;
      if (sup[j]) continue;
      float xx1 = ix1 > dets[j * 4] ? ix1 : dets[j * 4];
      float yy1 = iy1 > dets[j * 4 + 1] ? iy1 : dets[j * 4 + 1];
      float xx2 = ix2 < dets[j * 4 + 2] ? ix2 : dets[j * 4 + 2];
      float yy2 = iy2 < dets[j * 4 + 3] ? iy2 : dets[j * 4 + 3];
      float w = xx2 - xx1 + 1; if (w < 0) w = 0;
      float h = yy2 - yy1 + 1; if (h < 0) h = 0;
      float inter = w * h;
      float ovr = inter / (ia + area[j] - inter);
      if (ovr >= thr) sup[j] = 1;
    }
  }
  int64_t k = 0;
  for (int64_t i = 0; i < n; i++) if (!sup[i]) keep_out[k++] = i;
  free(ord); free(area); free(sup);
  return k;
}

/* ------------------------------------------------------------------ ROIAlign */
#define ORC_ROI_IMPL(T, SUF)                                                                      \
  static void orc_bilin_##SUF(int H, int W, T y, T x, int* yl, int* xl, int* yh, int* xh,          \
                              T* w1, T* w2, T* w3, T* w4, int* empty) {                            \
    if (y < (T)-1.0 || y > (T)H || x < (T)-1.0 || x > (T)W) {                                     \
      *empty = 1; *w1 = *w2 = *w3 = *w4 = 0; *yl = *xl = *yh = *xh = 0; return; }                  \
    *empty = 0;                                                                                    \
    if (y <= 0) y = 0;                                                                             \
    if (x <= 0) x = 0;                                                                             \
    int y_low = (int)y, x_low = (int)x, y_high, x_high;                                            \
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (T)y_low; } else y_high = y_low + 1;         \
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (T)x_low; } else x_high = x_low + 1;         \
    T ly = y - y_low, lx = x - x_low, hy = (T)1. - ly, hx = (T)1. - lx;                            \
    *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;                                    \
    *yl = y_low; *xl = x_low; *yh = y_high; *xh = x_high;                                          \
  }                                                                                                \
  /* input [N,C,H,W], rois [K,5]=(batch,x1,y1,x2,y2), out [K,C,PH,PW] */                            \
  void orc_roi_align_forward_##SUF(const T* in, const T* rois, int64_t K, int C, int H, int W,     \
                                   T scale, int PH, int PW, int sr, T* out) {                      \
    for (int64_t n = 0; n < K; n++) {                                                              \
      const T* r = rois + n * 5;                                                                   \
      int b = (int)r[0];                                                                           \
      T rsw = r[1] * scale, rsh = r[2] * scale, rew = r[3] * scale, reh = r[4] * scale;            \
      T rw = rew - rsw; if (rw < (T)1.) rw = (T)1.;                                                \
      T rh = reh - rsh; if (rh < (T)1.) rh = (T)1.;                                                \
      T bh = rh / (T)PH, bw = rw / (T)PW;                                                          \
      int gh = sr > 0 ? sr : (int)ceil(rh / PH);                                                   \
      int gw = sr > 0 ? sr : (int)ceil(rw / PW);                                                   \
      const T count = (T)(gh * gw);                                                                \
      for (int c = 0; c < C; c++) {                                                                \
        const T* plane = in + ((int64_t)b * C + c) * H * W;                                        \
        for (int ph = 0; ph < PH; ph++) for (int pw = 0; pw < PW; pw++) {                          \
          T acc = 0;                                                                               \
          for (int iy = 0; iy < gh; iy++) {                                                        \
            const T yy = rsh + ph * bh + (T)(iy + .5f) * bh / (T)gh;                               \
            for (int ix = 0; ix < gw; ix++) {                                                      \
              const T xx = rsw + pw * bw + (T)(ix + .5f) * bw / (T)gw;                             \
              int yl, xl, yh, xh, empty; T w1, w2, w3, w4;                                         \
              orc_bilin_##SUF(H, W, yy, xx, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4, &empty);       \
              /* same association as ROIAlign_cpu.cpp:203-206 */                                   \
              acc += w1 * plane[yl * W + xl] + w2 * plane[yl * W + xh] +                           \
                     w3 * plane[yh * W + xl] + w4 * plane[yh * W + xh];                            \
            }                                                                                      \
          }                                                                                        \
          out[(((int64_t)n * C + c) * PH + ph) * PW + pw] = acc / count;                           \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
  /* grad_out [K,C,PH,PW] -> grad_in [N,C,H,W] (must be zeroed by the caller) */                    \
  void orc_roi_align_backward_##SUF(const T* gout, const T* rois, int64_t K, int C, int H, int W,  \
                                    T scale, int PH, int PW, int sr, T* gin) {                     \
    for (int64_t n = 0; n < K; n++) {                                                              \
      const T* r = rois + n * 5;                                                                   \
      int b = (int)r[0];                                                                           \
      T rsw = r[1] * scale, rsh = r[2] * scale, rew = r[3] * scale, reh = r[4] * scale;            \
      T rw = rew - rsw; if (rw < (T)1.) rw = (T)1.;                                                \
      T rh = reh - rsh; if (rh < (T)1.) rh = (T)1.;                                                \
      T bh = rh / (T)PH, bw = rw / (T)PW;                                                          \
      int gh = sr > 0 ? sr : (int)ceil(rh / PH);                                                   \
      int gw = sr > 0 ? sr : (int)ceil(rw / PW);                                                   \
      const T count = (T)(gh * gw);                                                                \
      for (int c = 0; c < C; c++) {                                                                \
        T* plane = gin + ((int64_t)b * C + c) * H * W;                                             \
        for (int ph = 0; ph < PH; ph++) for (int pw = 0; pw < PW; pw++) {                          \
          const T g = gout[(((int64_t)n * C + c) * PH + ph) * PW + pw];                            \
          for (int iy = 0; iy < gh; iy++) {                                                        \
            const T yy = rsh + ph * bh + (T)(iy + .5f) * bh / (T)gh;                               \
            for (int ix = 0; ix < gw; ix++) {                                                      \
              const T xx = rsw + pw * bw + (T)(ix + .5f) * bw / (T)gw;                             \
              int yl, xl, yh, xh, empty; T w1, w2, w3, w4;                                         \
              orc_bilin_##SUF(H, W, yy, xx, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4, &empty);       \
              if (empty) continue;                                                                 \
              plane[yl * W + xl] += g * w1 / count;                                                \
              plane[yl * W + xh] += g * w2 / count;                                                \
              plane[yh * W + xl] += g * w3 / count;                                                \
              plane[yh * W + xh] += g * w4 / count;                                                \
            }                                                                                      \
          }                                                                                        \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }

ORC_ROI_IMPL(float, f32)
ORC_ROI_IMPL(double, f64)

/* ------------------------------------------------------------------ polygon -> mask */
static int orc_u32_cmp(const void* a, const void* b) {
  uint32_t c = *(const uint32_t*)a, d = *(const uint32_t*)b;
  return c > d ? 1 : (c < d ? -1 : 0);
}

/* One polygon (k vertices, xy interleaved, double) OR-ed into mask[h*w] stored COLUMN-major
 * like pycocotools (index = x*h + y).  Follows rleFrPoly's three stages: x5 upsampled dense
 * boundary walk; column crossings; sorted run lengths with zero-run folding; then decode. */
static void orc_poly_or(const double* xy, int k, int h, int w, uint8_t* mask) {
  const double scale = 5;
  int* x = (int*)malloc(sizeof(int) * (k + 1));
  int* y = (int*)malloc(sizeof(int) * (k + 1));
  long m = 0;
  for (int j = 0; j < k; j++) x[j] = (int)(scale * xy[j * 2 + 0] + .5);
  x[k] = x[0];
  for (int j = 0; j < k; j++) y[j] = (int)(scale * xy[j * 2 + 1] + .5);
  y[k] = y[0];
  for (int j = 0; j < k; j++) {
    int ax = abs(x[j] - x[j + 1]), ay = abs(y[j] - y[j + 1]);
    m += (ax > ay ? ax : ay) + 1;
  }
  int* u = (int*)malloc(sizeof(int) * m);
  int* v = (int*)malloc(sizeof(int) * m);
  m = 0;
  for (int j = 0; j < k; j++) {
    int xs = x[j], xe = x[j + 1], ys = y[j], ye = y[j + 1], t;
    int dx = abs(xe - xs), dy = abs(ys - ye);
    int flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) { t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
    double s = dx >= dy ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;
    if (dx >= dy) for (int d = 0; d <= dx; d++) {
      t = flip ? dx - d : d; u[m] = t + xs; v[m] = (int)(ys + s * t + .5); m++;
    } else for (int d = 0; d <= dy; d++) {
      t = flip ? dy - d : d; v[m] = t + ys; u[m] = (int)(xs + s * t + .5); m++;
    }
  }
  long kk = m; m = 0;
  free(x); free(y);
  x = (int*)malloc(sizeof(int) * (kk > 0 ? kk : 1));
  y = (int*)malloc(sizeof(int) * (kk > 0 ? kk : 1));
  for (long j = 1; j < kk; j++) if (u[j] != u[j - 1]) {
    double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1); xd = (xd + .5) / scale - .5;
    if (floor(xd) != xd || xd < 0 || xd > w - 1) continue;
    double yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]); yd = (yd + .5) / scale - .5;
    if (yd < 0) yd = 0; else if (yd > h) yd = h;
    yd = ceil(yd);
    x[m] = (int)xd; y[m] = (int)yd; m++;
  }
  kk = m;
  uint32_t* a = (uint32_t*)malloc(sizeof(uint32_t) * (kk + 1));
  for (long j = 0; j < kk; j++) a[j] = (uint32_t)(x[j] * (int)h + y[j]);
  a[kk++] = (uint32_t)(h * w);
  free(u); free(v); free(x); free(y);
  qsort(a, kk, sizeof(uint32_t), orc_u32_cmp);
  uint32_t p = 0;
  for (long j = 0; j < kk; j++) { uint32_t t = a[j]; a[j] -= p; p = t; }
  uint32_t* b = (uint32_t*)malloc(sizeof(uint32_t) * kk);
  long j = 0; m = 0; b[m++] = a[j++];
  while (j < kk) if (a[j] > 0) b[m++] = a[j++]; else { j++; if (j < kk) b[m - 1] += a[j++]; }
  /* decode (rleDecode) OR-ed into mask: union == rleMerge(intersect=0) then decode */
  long pos = 0; uint8_t val = 0;
  for (long r = 0; r < m; r++) {
    for (uint32_t c = 0; c < b[r] && pos < (long)h * w; c++, pos++) if (val) mask[pos] = 1;
    val = !val;
  }
  free(a); free(b);
}

/* npoly polygons of one instance: xy concatenated, lens[i] = number of VERTICES of polygon i.
 * out: row-major uint8 [h,w] (transposed from pycocotools' column-major, i.e. what
 * mask_utils.decode returns as a numpy (h,w) array). */
void orc_poly_mask(const double* xy, const int* lens, int npoly, int h, int w, uint8_t* out) {
  uint8_t* cm = (uint8_t*)calloc((size_t)h * w, 1);
  const double* p = xy;
  for (int i = 0; i < npoly; i++) { orc_poly_or(p, lens[i], h, w, cm); p += 2 * lens[i]; }
  for (int yy = 0; yy < h; yy++) for (int xx = 0; xx < w; xx++) out[yy * w + xx] = cm[xx * h + yy];
  free(cm);
}
