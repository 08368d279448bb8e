// Shared by the convolution translation units (conv_igemm.hip, conv_wgrad.hip, conv_prep.hip, conv_pgemm.hip, conv_stem.hip, conv_wgpl.hip): the parameter block of every convolution kernel,
// the fp16 split's range guard and its slow exact path, small vector types.  Device helpers live in an anonymous namespace (one
// copy per translation unit); the host-side pieces with one definition (conv_igemm.hip) are declared in namespace mmtconv.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace mmtconv {

struct ConvP {
  const float* x; const float* w; const float* scale; const float* shift; const float* res;
  const float* mask; const float* mul; float* y;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
  int relu, res_mode, out_stride, out_H, out_W;
  float mask_scale;
  int M, K, cin32, cin4;  // derived
  const unsigned short* wpl; long wpl_stride;  // pre-split bf16 planes of w (or null)
  const unsigned short* xpl; long xpl_stride;  // pre-split bf16 planes of x, same NHWC indexing as x (or null)
  unsigned short* ypl; long ypl_stride;        // also write y as three bf16 planes (for a 3x3 consumer), or null
  unsigned* amax_out;  // device float (bits) accumulating max |y| of this launch's output, or null (fp16 split: the consumer's scale)
  int amax_stats;      // amax_out is a 33-float slot: every 64th block adds sum |y| of what it stores to [1 + k] and the element count
                       // to [17 + k], k = (block >> 6) & 15 (crest factor max / mean of a SAMPLE: the fp16 split's fall-back test)
  int f16_ax;  // f16_sx points to max |x| (the scale is derived from it) instead of to the scale itself
  const float* f16_sx; const float* f16_sw;  // fp16 two-term split (experiment): device scalars s_x, s_w; the epilogue divides by s_x s_w
  int io;  // bf16 STORAGE of operands (mode 1): IO_X x, IO_Y y, IO_RES res, IO_MASK mask are bf16 tensors of the same indexing
  // fp16 split, range guard (round 4): the 33-float statistics slots of x (and, weight gradient, of dy) -- max, sampled sum and
  // count -- from which every block derives the crest factor of ITS OWN operand before its first instruction of arithmetic;
  // w_src / w_src_scale: where the fp32 weights of a planes-only (data-gradient) call come from (see conv_slow_tile)
  const float* guard_x; const float* guard_dy;
  const float* w_src; const float* w_src_scale;
  // weight gradient over TWO segments of pixels (round 4: the two student passes of a step share every weight; their activations
  // and gradients are separate tensors of one shape): blocks with z >= seg_z work on (x2, dy2) with that segment's own scales /
  // statistics slots; seg_z == 0: one segment
  const float* x2; const float* dy2; const float* f16_sx2; const float* f16_sw2; const float* guard_x2; const float* guard_dy2;
  int seg_z;
  int staged_epilogue;   // (A/B timing: MMT_DIRECT_EPI=0) the tiled and tap-strip kernels leave through the LDS-staged epilogue
  int xpl_rb;            // xpl is row-blocked: [N * H][Cin / 16][W][16] per plane (conv_pg_kernel only; mmt_conv_args.x_planes_layout)
  // round 6: y also as two row-blocked fp16 planes of y * *yrb_s ([N Ho][Cout / 16][Wo][16]), written by the epilogue for a plane-fed
  // consumer; amax_next: the producing site's pending maximum (-> next step's scale); xpl_lag: the scale of xpl was chosen before the
  // tensor existed (the guard tests it against the recorded statistics, f16_guard_bad_lag)
  unsigned short* yrb; long yrb_stride; const float* yrb_s; unsigned* amax_next; int xpl_lag;
  int yrb_M;   // output pixels [0, yrb_M) get planes (mmt_conv_args.y_rb_rows; M: all)
};
constexpr float F16_CREST_HI = 131072.f;   // 2^17: max / mean |x| above which fp16's five exponent bits lose the bulk of the tensor
constexpr int IO_X = 1, IO_Y = 2, IO_RES = 4, IO_MASK = 8, IO_DY = 16;

// mmt_conv_args -> ConvP (conv_igemm.hip); 0 or an MMT_E* code
int fill(ConvP& p, const mmt_conv_args* a);
// 0: fp32-input MFMA (exact fp32 products)   1: bf16   2: 2-term split (3 products)   3: 3-term split (6 products)
int precision();
// split-K workspace (partial tiles) + arrival counters (zeroed; whoever uses one leaves it zero), one pair per stream: kernels of a
// stream run in order and may share it; the teacher's and the student's streams launch concurrently (two host threads) and may not
struct SplitWs { float* ws; unsigned* tickets; };
constexpr size_t SPLITK_WS_BYTES = (size_t)1024 * 128 * 128 * 4;  // 1024 partial tiles of 128 x 128 (64 MiB)
constexpr int SPLITK_TICKETS = 4096;
SplitWs split_workspace(hipStream_t s);
// round 6: weight gradients of several layers in one launch (conv_wgpl.hip; orchestrated by mmt_conv_wgrad_group in conv_wgrad.hip)
struct WgReduceItem { const float* ws; const float* rowscale; float* dw; int splits, Cout, NP, pad; };
int launch_wgrad_reduce_group(const WgReduceItem* items, int n, hipStream_t s);
struct WgPlJob { const mmt_conv_args* a; const float* dy; const void* xpl; long xpl_stride; const void* dpl; long dpl_stride;
                 const float* s_x; const float* s_dy; const float* rowscale; float* dw; float* dbias; float* ws; int ksplit; };
int wgpl_eligible_splits(const mmt_conv_args* a);   // > 0: the plane-fed weight gradient takes the layer
long wgpl_super_steps(const mmt_conv_args* a);      // 32-pixel super-steps of the layer's reduction
int launch_wgpl_group(const WgPlJob* jobs, int n, hipStream_t s);
// layer1's 3x3 (64 -> 64 channels) on the patch kernel of conv_stem.hip: is this call one, and its launch
bool c64_shape(const ConvP& p);
int launch_c64(const ConvP& p, hipStream_t s);
}  // namespace mmtconv

namespace {
using namespace mmtconv;

// The unlagged, per-tensor form of the range decision.  The two-term fp16 split represents x s as h + l with
// s set by max |x|; when ONE element is 10^8 x the rest, everything else lands where l is subnormal and carries 11 bits instead
// of 22.  The host cannot know that about the tensor at hand (it picks the kernel before the producer has run); the device can:
// the producer recorded max / sampled mean of x in its statistics slot.  Every block of an fp16-split kernel reads the slot
// first and, for such a tensor, computes its tile with plain fp32 FMAs from the fp32 operands instead (conv_slow_tile: exact
// products, ~100 x slower, a handful of launches) -- until the host has seen the same statistics and moved the site to the
// 3-term bf16 split (_hip._site_ok), which is fast and range-free.  NaN / inf maxima take the slow path too (they propagate).
// Two halves, so that the slot's scalar loads are ISSUED at the top of a kernel and WAITED FOR behind its prologue copies: read at
// the point of the branch they cost ~1 us of exposed latency per launch (+0.5 ms of conv time per step, measured; the slot was
// last written by L2 atomics of several XCDs and misses this XCD's L2).  Nine values: the maximum and the first four of the
// sixteen (sum, count) partials -- the share of the producer's sampled blocks 0, 64, 128, 192 (mod 1024); block 0 always samples.
struct F16Guard { float amax, s0, s1, s2, s3, c0, c1, c2, c3; };
__device__ __forceinline__ F16Guard f16_guard_load(const float* __restrict__ slot) {
  F16Guard g = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (slot) { g.amax = slot[0]; g.s0 = slot[1]; g.s1 = slot[2]; g.s2 = slot[3]; g.s3 = slot[4];
              g.c0 = slot[17]; g.c1 = slot[18]; g.c2 = slot[19]; g.c3 = slot[20]; }
  return g;
}
__device__ __forceinline__ bool f16_guard_bad(const F16Guard& g) {
  const float tot = (g.s0 + g.s1) + (g.s2 + g.s3), cnt = (g.c0 + g.c1) + (g.c2 + g.c3);
  if (!(tot > 0.f)) return false;                       // nothing sampled (or an all-zero sample): no statement
  if (!(g.amax == g.amax) || g.amax > 3.0e38f) return true;
  // (Tried in round 6 and withdrawn: the mean of the BULK, the maximum taken out of the sample's sum -- it also fires on SPARSE
  // tensors, whose few non-zero values fp16 holds exactly: the RPN's gradient is non-zero at 512 of 1.3 M anchors, and every tenth
  // step took the exact path somewhere: p90 of the step 32.5 -> 42 ms.  A lone outlier INSIDE a small sample therefore stays
  // undetected by this test -- max / mean degenerates to the sample's count -- and is left to the host's lagged test.)
  return g.amax * cnt > tot * F16_CREST_HI;
}

// the same decision when the scale `s` of the planes was fixed BEFORE the tensor existed (planes written by the producer's epilogue
// with last step's maximum x 2 head-room, max |x| s in [2^12, 2^13) for an unchanged tensor): the fp16 range is tested with the
// scale actually applied -- max |x| s must stay below the largest fp16 number (a tensor may grow 7 x from one step to the next before
// it does not), and the sampled mean |x| s above 2^-5: the crest factor 2^17 ... 2^18 of the test above at that placement (a stricter
// 2^-4 with 8 x head-room, the first form, sent tensors with a crest factor of 2^14 ... 2^17 -- sparse gradients -- to the exact path)
__device__ __forceinline__ bool f16_guard_bad_lag(const F16Guard& g, const float s) {
  const float tot = (g.s0 + g.s1) + (g.s2 + g.s3), cnt = (g.c0 + g.c1) + (g.c2 + g.c3);
  if (!(g.amax == g.amax) || g.amax > 3.0e38f) return true;
  if (g.amax * s > 60000.f) return true;
  if (!(tot > 0.f)) return false;
  return cnt > tot * s * 32.f;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32 (RNE)
}
// x = x0 + x1 (+ x2) in bf16 terms, round-to-nearest at each level, the residuals exact in fp32 (4 values -> NS packed pairs of pairs)
template <int NS>
__device__ __forceinline__ void split4(const f32x4 v, uint2 (&o)[NS]) {
  float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
#pragma unroll
  for (int q = 0; q < NS; q++) {
    const unsigned a = pk_bf16(r0, r1), b = pk_bf16(r2, r3);
    o[q] = uint2{a, b};
    if (q + 1 < NS) {
      r0 -= __builtin_bit_cast(float, a << 16);
      r1 -= __builtin_bit_cast(float, a & 0xffff0000u);
      r2 -= __builtin_bit_cast(float, b << 16);
      r3 -= __builtin_bit_cast(float, b & 0xffff0000u);
    }
  }
}
// power-of-two scale that puts the largest magnitude `amax` into [2^13, 2^14] (an all-zero tensor: 1)
__device__ __forceinline__ float f16_scale_of(const float amax) {
  if (!(amax > 0.f)) return 1.f;
  int e;
  frexpf(amax, &e);
  e = 14 - e;
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.f, e);
}

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const f32x4*)p; }
// ---- EXPERIMENT (mmt_conv3x3_strip_f16x2, tools/bench_f16x2.py): two-term fp16 split of a pre-scaled value, x * s = h + l
// with h = fp16(x s), l = fp16(x s - h) (the residual is exact in fp32): 22 significant bits, |x s - h - l| <= 2^-22 |x s|.
// The caller scales each tensor by a power of two so that its largest magnitude sits near 2^14: h is then a normal fp16
// number down to 2^-28 of the tensor's maximum and l down to 2^-17 of it (below that l turns subnormal: absolute error
// <= 2^-25 in scaled units, i.e. <= 2^-39 of the maximum).
__device__ __forceinline__ void split4h(const f32x4 v, const float s, uint2 (&o)[2]) {
  float r[4] = {v[0] * s, v[1] * s, v[2] * s, v[3] * s};
#pragma unroll
  for (int e = 0; e < 4; e++) r[e] = fminf(fmaxf(r[e], -65504.f), 65504.f);   // a scale from an older tensor may be too large: saturate
  _Float16 h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; e++) { h[e] = (_Float16)r[e]; l[e] = (_Float16)(r[e] - (float)h[e]); }
  o[0] = uint2{__builtin_bit_cast(unsigned, f16x2{h[0], h[1]}), __builtin_bit_cast(unsigned, f16x2{h[2], h[3]})};
  o[1] = uint2{__builtin_bit_cast(unsigned, f16x2{l[0], l[1]}), __builtin_bit_cast(unsigned, f16x2{l[2], l[3]})};
}

struct AmaxAcc { float amx = 0.f, asum = 0.f, acnt = 0.f; };   // max |y| / sum |y| / count over what a thread stores (p.amax_out)

// ---- round 6: row-blocked fp16 planes of y from the epilogue (ConvP::yrb).  (h, l) of one value packed into a word: h in the low half
__device__ __forceinline__ unsigned rb_split1(const float v, const float s) {
  const float r = __builtin_amdgcn_fmed3f(v * s, -65504.f, 65504.f);   // (same clamp as split4h: a stale scale saturates, never inf)
  const _Float16 h = (_Float16)r;
  const _Float16 l = (_Float16)(r - (float)h);
  return __builtin_bit_cast(unsigned, f16x2{h, l});
}
// byte offset of (output pixel m, channel c) inside one plane; pixel -> (image row, column) with one float reciprocal (exact for
// m < 2^22 after the correction step)
struct RbGeom { int Wo; float rWo; unsigned cb32; };   // cb32 = (Cout / 16) * Wo * 32: bytes of one image row of a plane
__device__ __forceinline__ RbGeom rb_geom(const ConvP& p) {
  return RbGeom{p.Wo, 1.f / (float)p.Wo, (unsigned)(p.Cout >> 4) * (unsigned)p.Wo * 32u};
}
__device__ __forceinline__ unsigned rb_row_off(const RbGeom& g, const int m) {   // offset of (pixel m, channel 0)
  int row = (int)(((float)m + 0.5f) * g.rWo);
  int w = m - row * g.Wo;
  if (w < 0) { row--; w += g.Wo; } else if (w >= g.Wo) { row++; w -= g.Wo; }
  return (unsigned)row * g.cb32 + (unsigned)w * 32u;
}
// four consecutive channels c .. c + 3 (c % 4 == 0) of pixel m, already split: 8 bytes into each plane
__device__ __forceinline__ void rb_store4(const ConvP& p, const RbGeom& g, const int m, const int c, const f32x4 v, const float s) {
  uint2 o[2];
  split4h(v, s, o);
  char* const d = (char*)p.yrb + rb_row_off(g, m) + (unsigned)(c >> 4) * ((unsigned)g.Wo * 32u) + (unsigned)(c & 15) * 2u;
  *(uint2*)d = o[0];
  *(uint2*)(d + p.yrb_stride * 2) = o[1];
}
// 4 x 4 transpose inside every quad of lanes: in[j] of lane k -> out[k] of lane j (two butterfly stages on DPP quad permutes)
__device__ __forceinline__ void quad_transpose(unsigned (&w)[4], const int lane) {
  const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
  for (int pr = 0; pr < 2; pr++) {   // pairs (0, 1), (2, 3) with the lane across bit 0
    const unsigned send = o1 ? w[2 * pr] : w[2 * pr + 1];
    const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]
    if (o1) w[2 * pr] = recv; else w[2 * pr + 1] = recv;
  }
#pragma unroll
  for (int pr = 0; pr < 2; pr++) {   // pairs (0, 2), (1, 3) with the lane across bit 1
    const unsigned send = o2 ? w[pr] : w[pr + 2];
    const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);   // quad_perm [2, 3, 0, 1]
    if (o2) w[pr] = recv; else w[pr + 2] = recv;
  }
}

// ---- the fp16 split's slow, exact path (see f16_guard_bad): outputs (row i, column c) of a block's tile, i < rows,
// c < ncols; row i is output pixel m = m_first + (i / run) * run_stride + (i % run) (one run of consecutive pixels for the tiled
// and row-resident kernels, 256 / TW image rows for the strip kernel), column c is channel n0 + c.  fp32 FMA over the im2col
// row straight from global memory; the weights are p.w ([Cout][KH][KW][Cin]) or, for a planes-only data-gradient call, read
// through the flip / transpose / BN scale of pack_flip_unit from the forward weight p.w_src ([Cin'][KH][KW][Cout'], scale per
// row).  raw != null (split-K forms): the sum times s_x s_w goes to raw[i * ncols + c] -- what the finish launch divides out
// again -- or zeros when `zero`; else the epilogue of conv_epilogue_finish, element by element, statistics included.
// The parameter block is read from the KERNEL-ARGUMENT SEGMENT (every kernel that calls this has its ConvP first), not from the
// caller's registers: inlined with `p` in registers the slow path kept ~60 more scalar values alive across the callers' prologues
// (SGPR spills 15 -> 80 in the tiled kernel, +0.6 ms of conv time per step); as a real call it costs the callers 288 B of stack.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) ConvP* ConvPK;
__device__ __forceinline__ ConvPK kernarg_convp() { return (ConvPK)__builtin_amdgcn_kernarg_segment_ptr(); }
#else   // (host pass of the same translation unit: never executed)
typedef const ConvP* ConvPK;
__device__ __forceinline__ ConvPK kernarg_convp() { return nullptr; }
#endif

__device__ __forceinline__ void conv_slow_tile(const int m_first, const int run, const int run_stride, const int rows,
                                            const int n0, const int ncols, float* __restrict__ raw, const bool zero,
                                            const int tid, const int nthreads, const int lin) {
  const ConvPK pk = kernarg_convp();   // fields are fetched where they are used (scalar loads from the argument segment)
  const int HoWo = pk->Ho * pk->Wo;
  const float S = (pk->f16_ax ? f16_scale_of(*pk->f16_sx) : *pk->f16_sx) * *pk->f16_sw;
  float amx = 0.f, asum = 0.f, acnt = 0.f;
  for (int o = tid; o < rows * ncols; o += nthreads) {
    const int i = o / ncols, c = o - i * ncols;
    const int m = m_first + (i / run) * run_stride + (i % run), n = n0 + c;
    if (raw && (zero || m >= pk->M || n >= pk->Cout)) { raw[o] = 0.f; continue; }
    if (m >= pk->M || n >= pk->Cout) continue;
    const int img = m / HoWo, rem = m - img * HoWo;
    const int ho = rem / pk->Wo, wo = rem - ho * pk->Wo;
    float acc = 0.f;
    for (int kh = 0; kh < pk->KH; kh++) {
      const int ih = ho * pk->stride - pk->pad + kh;
      if ((unsigned)ih >= (unsigned)pk->H) continue;
      for (int kw = 0; kw < pk->KW; kw++) {
        const int iw = wo * pk->stride - pk->pad + kw;
        if ((unsigned)iw >= (unsigned)pk->W) continue;
        const float* xr = pk->x + ((long)(img * pk->H + ih) * pk->W + iw) * pk->Cin;
        if (pk->w) {
          const float* wr = pk->w + ((long)(n * pk->KH + kh) * pk->KW + kw) * pk->Cin;
          for (int ci = 0; ci < pk->Cin; ci++) acc = fmaf(xr[ci], wr[ci], acc);
        } else {   // W'(n, (kh, kw), c) = w_src[c][KH - 1 - kh][KW - 1 - kw][n] * scale[c]
          const int tap = (pk->KH - 1 - kh) * pk->KW + (pk->KW - 1 - kw);
          const float* wr = pk->w_src + (long)tap * pk->Cout + n;
          const long cs = (long)pk->KH * pk->KW * pk->Cout;
          for (int ci = 0; ci < pk->Cin; ci++)
            acc = fmaf(xr[ci], wr[ci * cs] * (pk->w_src_scale ? pk->w_src_scale[ci] : 1.f), acc);
        }
      }
    }
    if (raw) { raw[o] = acc * S; continue; }
    float v = acc * (pk->scale ? pk->scale[n] : 1.f) + (pk->shift ? pk->shift[n] : 0.f);
    long oidx = (long)m * pk->Cout + n;
    if (pk->out_stride > 1) oidx = (((long)img * pk->out_H + ho * pk->out_stride) * pk->out_W + wo * pk->out_stride) * pk->Cout + n;
    if (pk->res_mode == 1) v += pk->res[(long)m * pk->Cout + n];
    else if (pk->res_mode == 2) v += pk->res[(((long)img * (pk->Ho >> 1) + (ho >> 1)) * (pk->Wo >> 1) + (wo >> 1)) * pk->Cout + n];
    else if (pk->res_mode == 3) {
      const int w2 = pk->Wo * 2;
      const float* rp = pk->res + (((long)img * (pk->Ho * 2) + 2 * ho) * w2 + 2 * wo) * pk->Cout + n;
      v += (rp[0] + rp[pk->Cout]) + (rp[(long)w2 * pk->Cout] + rp[(long)w2 * pk->Cout + pk->Cout]);
    }
    if (pk->relu) v = fmaxf(v, 0.f);
    if (pk->mask) v = pk->mask[oidx] > 0.f ? v * pk->mask_scale : 0.f;
    if (pk->mul) v *= pk->mul[(long)m * pk->Cout + n];
    pk->y[oidx] = v;
    if (pk->yrb && m < pk->yrb_M) {   // the planes a plane-fed consumer was promised (one element at a time: this path is ~100 x slower anyway)
      const unsigned hl = rb_split1(v, *pk->yrb_s);
      const long row = (long)img * pk->Ho + ho;
      const long e = ((row * (pk->Cout >> 4) + (n >> 4)) * pk->Wo + wo) * 16 + (n & 15);
      pk->yrb[e] = (unsigned short)(hl & 0xffffu);
      pk->yrb[pk->yrb_stride + e] = (unsigned short)(hl >> 16);
    }
    const float av = fabsf(v);
    amx = fmaxf(amx, av); asum += av; acnt += 1.f;
  }
  if (!raw && pk->amax_out) {
    if (amx > 0.f) atomicMax(pk->amax_out, __builtin_bit_cast(unsigned, amx));
    if (amx > 0.f && pk->amax_next) atomicMax(pk->amax_next, __builtin_bit_cast(unsigned, amx));
    if (pk->amax_stats && (lin & 63) == 0 && acnt > 0.f) {
      const int k = (lin >> 6) & 15;
      atomicAdd((float*)pk->amax_out + 1 + k, asum);
      atomicAdd((float*)pk->amax_out + 17 + k, acnt);
    }
  }
}

// ---- epilogue straight from the accumulator registers of a wave's 64 x 64 tile (2 x 2 MFMA tiles of 32 x 32; round 5: the plane-fed
// kernel and the tap-strip kernel).  acc[a][b][r] = tile row 32 a + 8 (r / 4) + r % 4 + 4 (lane / 32), channel 32 b + lane % 32: an
// accumulator register is one output row x 32 consecutive channels per half wave -- stores, residual and mask loads of a wave touch
// complete 128-byte lines with nothing staged through LDS.  (The staged epilogue of conv_igemm.hip waits for its residual / mask
// loads one by one -- the compiler's answer to a load under a lane predicate -- and cost 18 - 30 us per 256 x 128 tile outside the
// main loop, profiles/r04_history.md; here every request is an unconditional buffer operation: an absent operand is a zero-sized
// buffer, a row past `rows_left` or a channel past Cout is an offset beyond the buffer.)
//   ybase      byte offset of (the wave's tile row 0 + 4 (lane / 32), channel 0) in y; the wave's 64 rows are consecutive pixels
//   rows_left  rows from there to the end of the tensor (<= 0: nothing of this lane's is stored)
//   c0         channel of (b = 0, this lane)
//   own        bit i = (TN a + b) set: this wave finishes that sub-tile (groups of waves that share a tile split its sub-tiles)
//   red        24 floats of LDS free at this point, block-wide (the statistics' reduction); every wave of the block calls this
// Same expressions, in the same order, as conv_epilogue_finish: bit-identical outputs.  fp32 tensors, res_mode <= 1, out_stride 1.
template <int TM, int TN>   // the wave's tile: TM x TN MFMA tiles of 32 x 32 (2 x 2: the plane-fed and tap-strip kernels; 1 x 4 / 1 x 2 / 1 x 1: the tiled kernel)
__device__ __forceinline__ void conv_epilogue_direct(const ConvP& p, const f32x16 (&acc)[TM][TN], const unsigned ybase, const int rows_left,
                                                     const int c0, const int own, float* const red, const int wave, const int lane,
                                                     const int tid, const int lin) {
  constexpr unsigned OOB = 0x80000000u;
  const unsigned cout4 = (unsigned)p.Cout * 4u;
  unsigned cb[TN];
  bool cok[TN];
#pragma unroll
  for (int b = 0; b < TN; b++) {
    const int c = c0 + b * 32;
    cok[b] = c < p.Cout;
    cb[b] = cok[b] ? (unsigned)c * 4u : OOB;
  }
  const int ybytes = (int)((long)p.M * p.Cout * 4);
  const bool has_res = p.res_mode != 0, has_mask = p.mask != nullptr;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(has_res ? p.res : p.y), 0, has_res ? ybytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc((void*)(has_mask ? p.mask : p.y), 0, has_mask ? ybytes : 0, 0x00020000);
  float sc[TN], sh[TN];
  // operands scaled by powers of two (fp16 split): exact rescale of the accumulated sum
  const float inv = p.f16_sx ? 1.f / ((p.f16_ax ? f16_scale_of(*p.f16_sx) : *p.f16_sx) * *p.f16_sw) : 1.f;
#pragma unroll
  for (int b = 0; b < TN; b++) {
    sc[b] = p.scale && cok[b] ? p.scale[c0 + b * 32] : 1.f;
    if (p.f16_sx) sc[b] *= inv;
    sh[b] = p.shift && cok[b] ? p.shift[c0 + b * 32] : 0.f;
  }
  auto off = [&](int a, int b, int r) { return ybase + (unsigned)(a * 32 + 8 * (r >> 2) + (r & 3)) * cout4 + cb[b]; };
  float amx = 0.f, asum = 0.f, acnt = 0.f;
  // round 6: y also as row-blocked fp16 planes (p.yrb).  After a 4 x 4 transpose inside every quad of lanes, lane k of a quad holds
  // the quad's four channels of row (4 j + k) of a register group j: 8 bytes per plane and lane, and the 16 lanes of a 16-channel block
  // write 4 consecutive pixels x 32 bytes = one 128-byte run (the half wave above: the next four pixels)
  const bool rb = p.yrb != nullptr;
  const float rbs = rb ? *p.yrb_s : 1.f;
  const RbGeom rbg = rb_geom(p);
  const int rb_bytes = rb ? (int)((long)p.yrb_M * p.Cout * 2) : 0;
  const __amdgpu_buffer_rsrc_t rrb0 = __builtin_amdgcn_make_buffer_rsrc((void*)(rb ? p.yrb : (unsigned short*)p.y), 0, rb_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrb1 = __builtin_amdgcn_make_buffer_rsrc((void*)(rb ? p.yrb + p.yrb_stride : (unsigned short*)p.y), 0, rb_bytes, 0x00020000);
  const int mrow = (int)(ybase / cout4);   // output pixel of (tile row 0 + 4 (lane / 32))
#pragma unroll
  for (int i = 0; i < TM * TN; i++) {
    if (!((own >> i) & 1)) continue;
    const int a = i / TN, b = i % TN;
    float vv[16];
    float ur[16], um[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { ur[r] = 0.f; um[r] = 1.f; }
    if (has_res) {   // whole blocks of 16 requests behind one uniform branch each (an absent operand costs nothing)
#pragma unroll
      for (int r = 0; r < 16; r++)
        ur[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)off(a, b, r), 0, 0));
    }
    if (has_mask) {
#pragma unroll
      for (int r = 0; r < 16; r++)
        um[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmask, (int)off(a, b, r), 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float v = acc[a][b][r] * sc[b] + sh[b];
      if (has_res) v += ur[r];
      if (p.relu) v = fmaxf(v, 0.f);
      if (has_mask) v = um[r] > 0.f ? v * p.mask_scale : 0.f;
      const bool ok = cok[b] && a * 32 + 8 * (r >> 2) + (r & 3) < rows_left;
      const float av = ok ? fabsf(v) : 0.f;
      amx = fmaxf(amx, av);
      asum += av;
      acnt += ok ? 1.f : 0.f;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)(ok ? off(a, b, r) : OOB), 0, 0);
      vv[r] = v;
    }
    if (rb) {   // (one uniform branch per sub-tile: launches without planes pay nothing)
      const int cq = (c0 + b * 32) & ~3;   // first channel of this lane's quad
#pragma unroll
      for (int j = 0; j < 4; j++) {
        unsigned w4[4] = {rb_split1(vv[4 * j], rbs), rb_split1(vv[4 * j + 1], rbs), rb_split1(vv[4 * j + 2], rbs), rb_split1(vv[4 * j + 3], rbs)};
        quad_transpose(w4, lane);
        const int rr = a * 32 + 8 * j + (lane & 3);
        const bool ok = cok[b] && rr < rows_left && mrow + rr < p.yrb_M;
        const unsigned o = ok ? rb_row_off(rbg, mrow + rr) + (unsigned)(cq >> 4) * ((unsigned)rbg.Wo * 32u) + (unsigned)(cq & 15) * 2u : OOB;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 hh = {__builtin_amdgcn_perm(w4[1], w4[0], 0x05040100u), __builtin_amdgcn_perm(w4[3], w4[2], 0x05040100u)};
        const u32x2 ll = {__builtin_amdgcn_perm(w4[1], w4[0], 0x07060302u), __builtin_amdgcn_perm(w4[3], w4[2], 0x07060302u)};
        __builtin_amdgcn_raw_buffer_store_b64(hh, rrb0, (int)o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(ll, rrb1, (int)o, 0, 0);
      }
    }
  }
  // statistics of the output (max |y|; sum |y| and count from every 64th block), reduced through `red`: a second __shared__ object
  // would make the compiler drain the copy queue in front of every fragment read of the main loop
  if (p.amax_out) {
    const int nw = (int)(blockDim.x >> 6);
    const bool stats = p.amax_stats && (lin & 63) == 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
    if (stats) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { asum += __shfl_xor(asum, o, 64); acnt += __shfl_xor(acnt, o, 64); }
    }
    if (lane == 0) { red[wave] = amx; red[16 + wave] = asum; red[32 + wave] = acnt; }
    __syncthreads();
    if (tid == 0) {
      float m = red[0], sm = red[16], cn = red[32];
      for (int i = 1; i < nw; i++) { m = fmaxf(m, red[i]); sm += red[16 + i]; cn += red[32 + i]; }
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
}

}  // namespace
