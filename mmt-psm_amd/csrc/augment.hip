// Input augmentation on the device: the Pillow / torchvision pixel arithmetic behind the reference's
// data/transforms/transforms.py:28-205 (Resize, RandomHorizontalFlip, AdjustBrightness, AdjustContrast, AdjustHue,
// RandomErasing, ToTensor, Normalize), bit-exact:
//   * resample_kernel   Pillow ImagingResample, 8-bit path: one separable pass with the 22-bit fixed-point triangle
//                       (BILINEAR, antialiased when shrinking) coefficients computed on the host exactly like
//                       precompute_coeffs / normalize_coeffs_8bpc; horizontal pass first, 8-bit intermediate;
//   * luma_sum_kernel   sum of convert('L') of the brightness-adjusted image (ImageEnhance.Contrast's grey level);
//   * view_kernel       flip -> ImagingBlend with black (brightness) -> ImagingBlend with the mean grey (contrast) ->
//                       rgb2hsv, wrap-around hue add, hsv2rgb (Convert.c; float / double mix as in the C code) ->
//                       to_tensor (/255), BGR * 255, - mean: one pass, uint8 in, fp32 NHWC out (zero-padded batch slot);
//   * erase_kernel      RandomErasing rectangles (host-drawn positions and fill bytes, as the reference draws them).
// HBM-bound: 3 B read + 12 B written per pixel and view.  Built with -ffp-contract=off (Pillow is not FMA-contracted).
#include "common.h"

namespace {
__device__ __forceinline__ int blend8(int deg, int px, float alpha, bool inside) {
  const float t = (float)deg + alpha * (float)(px - deg);   // ImagingBlend: float arithmetic
  if (inside) return (int)(unsigned char)t;                 // 0 <= alpha <= 1: plain (UINT8) cast
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);        // extrapolation: clipped
}
__device__ __forceinline__ int luma8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__global__ __launch_bounds__(256) void resample_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H,
                                                       int W, int out_size, int horizontal,
                                                       const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                                       int ksize) {
  const int oH = horizontal ? H : out_size, oW = horizontal ? out_size : W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)oH * oW) return;
  const int y = (int)(i / oW), x = (int)(i - (long)y * oW);
  const int o = horizontal ? x : y;
  const int lo = bounds[2 * o], n = bounds[2 * o + 1];
  const int* k = coeffs + (long)o * ksize;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;  // 1 << (PRECISION_BITS - 1)
  for (int t = 0; t < n; t++) {
    const uint8_t* p = horizontal ? src + ((long)y * W + lo + t) * 3 : src + ((long)(lo + t) * W + x) * 3;
    const int c = k[t];
    a0 += p[0] * c; a1 += p[1] * c; a2 += p[2] * c;
  }
  uint8_t* q = dst + i * 3;
  a0 >>= 22; a1 >>= 22; a2 >>= 22;
  q[0] = (uint8_t)(a0 < 0 ? 0 : (a0 > 255 ? 255 : a0));
  q[1] = (uint8_t)(a1 < 0 ? 0 : (a1 > 255 ? 255 : a1));
  q[2] = (uint8_t)(a2 < 0 ? 0 : (a2 > 255 ? 255 : a2));
}

__global__ __launch_bounds__(256) void luma_sum_kernel(const uint8_t* __restrict__ img, long npix,
                                                       const float* __restrict__ brightness, int V,
                                                       unsigned long long* __restrict__ sums) {
  const int v = blockIdx.y;
  const float a = brightness[v];
  const bool inside = a >= 0.f && a <= 1.f;
  unsigned s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const uint8_t* p = img + i * 3;
    s += luma8(blend8(0, p[0], a, inside), blend8(0, p[1], a, inside), blend8(0, p[2], a, inside));
  }
  __shared__ unsigned red[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sums + v, (unsigned long long)red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void view_kernel(const uint8_t* __restrict__ img, int H, int W, int flip,
                                                   const float* __restrict__ brightness, const float* __restrict__ contrast,
                                                   const int* __restrict__ hue_shift,
                                                   const unsigned long long* __restrict__ sums, float m0, float m1, float m2,
                                                   float* __restrict__ out, long view_stride, int out_W, int out_C) {
  const int v = blockIdx.y;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)H * W) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  const uint8_t* p = img + ((long)y * W + (flip ? W - 1 - x : x)) * 3;
  const float ab = brightness[v], ac = contrast[v];
  const bool ib = ab >= 0.f && ab <= 1.f, ic = ac >= 0.f && ac <= 1.f;
  int r = p[0], g = p[1], b = p[2];
  float* o = out + (long)v * view_stride + ((long)y * out_W + x) * out_C;
  if (hue_shift[v] < 0) {  // no colour chain (test-time pipeline): ToTensor + Normalize only
    o[0] = ((float)b / 255.f) * 255.f - m0;
    o[1] = ((float)g / 255.f) * 255.f - m1;
    o[2] = ((float)r / 255.f) * 255.f - m2;
    return;
  }
  r = blend8(0, r, ab, ib); g = blend8(0, g, ab, ib); b = blend8(0, b, ab, ib);
  const int mean = (int)((double)sums[v] / (double)((long)H * W) + 0.5);  // int(ImageStat.Stat(L).mean[0] + 0.5)
  r = blend8(mean, r, ac, ic); g = blend8(mean, g, ac, ic); b = blend8(mean, b, ac, ic);
  // ---- rgb2hsv (Convert.c): float ratios, double hue wrap and x255
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = (int)((double)h * 255.0); uh = uh < 0 ? 0 : (uh > 255 ? 255 : uh);
    us = (int)((double)s * 255.0); us = us < 0 ? 0 : (us > 255 ? 255 : us);
  }
  uh = (uh + hue_shift[v]) & 255;  // uint8 wrap-around add of F.adjust_hue
  // ---- hsv2rgb (Convert.c): double arithmetic, round half up
  if (us == 0) {
    r = g = b = uv;
  } else {
    const double fh = (double)uh * 6.0 / 255.0, fs = (double)us / 255.0, mv = (double)uv;
    const int ii = (int)floor(fh);
    const double f = fh - (double)ii;
    int pp = (int)floor(mv * (1.0 - fs) + 0.5), qq = (int)floor(mv * (1.0 - fs * f) + 0.5),
        tt = (int)floor(mv * (1.0 - fs * (1.0 - f)) + 0.5);
    pp = pp < 0 ? 0 : (pp > 255 ? 255 : pp); qq = qq < 0 ? 0 : (qq > 255 ? 255 : qq); tt = tt < 0 ? 0 : (tt > 255 ? 255 : tt);
    switch (ii % 6) {
      case 0: r = uv; g = tt; b = pp; break;
      case 1: r = qq; g = uv; b = pp; break;
      case 2: r = pp; g = uv; b = tt; break;
      case 3: r = pp; g = qq; b = uv; break;
      case 4: r = tt; g = pp; b = uv; break;
      default: r = uv; g = pp; b = qq; break;
    }
  }
  // ---- to_tensor (/255), image[[2,1,0]] * 255, - mean (transforms.py:84-99)
  o[0] = ((float)b / 255.f) * 255.f - m0;
  o[1] = ((float)g / 255.f) * 255.f - m1;
  o[2] = ((float)r / 255.f) * 255.f - m2;
}

__global__ __launch_bounds__(256) void erase_kernel(float* __restrict__ out, long view_stride, int out_W, int out_C,
                                                    const int* __restrict__ rects, const long* __restrict__ fill_off,
                                                    const uint8_t* __restrict__ fills, float m0, float m1, float m2) {
  const int* rc = rects + blockIdx.y * 5;  // view, top, left, h, w
  const int h = rc[3], w = rc[4];
  const uint8_t* f = fills + fill_off[blockIdx.y];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)h * w; i += (long)gridDim.x * 256) {
    const int y = (int)(i / w), x = (int)(i - (long)y * w);
    float* o = out + (long)rc[0] * view_stride + ((long)(rc[1] + y) * out_W + rc[2] + x) * out_C;
    const uint8_t* p = f + i * 3;  // RGB bytes
    o[0] = ((float)p[2] / 255.f) * 255.f - m0;
    o[1] = ((float)p[1] / 255.f) * 255.f - m1;
    o[2] = ((float)p[0] / 255.f) * 255.f - m2;
  }
}
}  // namespace

extern "C" int mmt_resample_u8(const uint8_t* src, uint8_t* dst, int H, int W, int out_size, int horizontal,
                               const int32_t* bounds, const int32_t* coeffs, int ksize, void* stream) {
  if (!src || !dst || !bounds || !coeffs || H < 1 || W < 1 || out_size < 1 || ksize < 1) return MMT_EINVAL;
  const long n = horizontal ? (long)H * out_size : (long)out_size * W;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W,
                     out_size, horizontal, bounds, coeffs, ksize);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_aug_views(const uint8_t* img, int H, int W, int flip, const float* brightness, const float* contrast,
                             const int32_t* hue_shift, unsigned long long* sums_ws, int V, const float* mean3, float* out,
                             long view_stride, int out_W, int out_C, void* stream) {
  if (!img || !brightness || !contrast || !hue_shift || !sums_ws || !mean3 || !out || V < 1 || out_C < 3 || out_W < W)
    return MMT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(sums_ws, 0, (size_t)V * sizeof(unsigned long long), s);
  if (e != hipSuccess) return (int)e;
  const long npix = (long)H * W;
  int blocks = (int)((npix + 255) / 256);
  hipLaunchKernelGGL(luma_sum_kernel, dim3(blocks > 1024 ? 1024 : blocks, V), dim3(256), 0, s, img, npix, brightness, V,
                     sums_ws);
  MMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(view_kernel, dim3(blocks, V), dim3(256), 0, s, img, H, W, flip, brightness, contrast, hue_shift, sums_ws,
                     mean3[0], mean3[1], mean3[2], out, view_stride, out_W, out_C);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_aug_erase(float* out, long view_stride, int out_W, int out_C, const int32_t* rects, const long* fill_off,
                             const uint8_t* fills, int R, const float* mean3, void* stream) {
  if (R <= 0) return 0;
  if (!out || !rects || !fill_off || !fills || !mean3) return MMT_EINVAL;
  hipLaunchKernelGGL(erase_kernel, dim3(16, R), dim3(256), 0, (hipStream_t)stream, out, view_stride, out_W, out_C, rects,
                     fill_off, fills, mean3[0], mean3[1], mean3[2]);
  MMT_LAUNCH_CHECK();
  return 0;
}
