// The ResNet stem as ONE kernel (round 5): conv 7x7 / stride 2 / pad 3 (3 -> 64) + FrozenBatchNorm + ReLU + max pool 3x3 / stride 2 /
// pad 1 -- reference modeling/backbone/resnet.py:288-293 `StemWithFixedBatchNorm.forward` with layers/batch_norm.py:19-24 folded
// in.  The stem is frozen (FREEZE_CONV_BODY_AT = 2): forward only.
//
// Before: a library copy built the 16-channel space-to-depth image, the tiled kernel ran the 4x4 / stride-1 convolution over it
// (429 us for the teacher's 8 x 1024^2 batch, 3.2 x its HBM time: K = 256 leaves 16 steps per tile to pay for a tile's prologue and
// epilogue), wrote 537 MB of ReLU output, and maxpool_kernel read them back (108 us).  The round trip of those 537 MB is the
// larger half of the cost, and nothing downstream wants the un-pooled tensor.  Here a block owns a tile of 7 x 15 POOLED pixels:
//   * the 16 x 32 convolution outputs under it (rows 2 py0 - 1 .., columns 2 px0 - 1 ..: every pooling window of the tile; one MFMA
//     row tile = one output row) need a 19 x 35 patch of space-to-depth pixels (16 channels = the 2 x 2 parity x RGB + 0).  The patch
//     is gathered ONCE from the NCHW image -- 8-byte loads, coalesced along W --, scaled by the power of two of the image's recorded
//     maximum, split into its two fp16 terms and written to LDS as the A-plane image of the convolution kernels (32 bytes per pixel
//     and plane, half ^= (pixel >> 3) & 1); the 16 taps (a, b) of the 4 x 4 filter read their fragments from it at pixel offset
//     35 a + b: no im2col traffic at all;
//   * the packed weight planes of all 16 steps (64 KB) are copied into LDS once; the main loop is 16 steps of 8 fragment reads + 12
//     MFMAs per wave (8 waves, 2 output rows x 64 channels each) with no barrier and no copy in it;
//   * epilogue: scale / shift / ReLU in registers; vertical maximum of a wave's two rows in registers, the third row (the next
//     wave's first) and the horizontal 3-maximum through LDS; 16-byte coalesced stores of the pooled tile; statistics of the output.
// Same products in the same order as the tiled kernel on the space-to-depth image and the same maximum: bit-identical to the
// three-launch path (tests/test_stem_gpu.py).  Arithmetic: two-term fp16 split, 3 products per multiply (mode 3 default).
//
// Roofline: HBM -- 12 B read per input pixel, 64 B written per output pixel (8 x 1024^2: 101 + 134 MB = 47 us at 5 TB/s); the matrix
// work (39.5 GFLOP x 3 products, 1.22 x redundant rows / columns of overlapping tiles) is 58 us at the dense fp16 peak.
#include "conv_shared.h"

namespace {

struct StemP {
  const float* x;                      // [N][3][H][W] fp32
  const unsigned short* wpl; long wpl_stride;   // packed fp16 planes of the space-to-depth filter [64][(a, b, 16 ch) = 256]
  const float* w;                      // the same filter in fp32 ([64][4][4][16]), for the exact path
  const float* scale; const float* shift;
  float* y;                            // [N][Hp][Wp][64] fp32
  const float* x_slot;                 // statistics slot of x: [0] = max |x| (the scale is derived from it), sums / counts for the guard
  const float* s_w;                    // scale of the weight planes
  unsigned* amax_out; int amax_stats;
  int N, H, W, Hc, Wc, Hp, Wp, tiles_y, tiles_x;
};

constexpr int SPH = 19, SPW = 35, SPIX = SPH * SPW;          // patch of space-to-depth pixels
constexpr int SPL = 672 * 32;                                 // bytes of one plane of the patch (665 pixels, padded)
constexpr int SB_OFF = 2 * SPL;                               // weight planes behind the two patch planes
constexpr int STEM_LDS = 128 * 1024;                          // loop: 2 x 21 KB + 64 KB; epilogue: 64 KB + 56 KB

__device__ __forceinline__ float stem_exact(const StemP& p, int n, int oy, int ox, int co) {
  // conv output (oy, ox, co) with fp32 FMAs straight from the image (the fp16 split's slow, exact path)
  float acc = 0.f;
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++) {
      const int si = oy - 2 + a, sj = ox - 2 + b;
      if ((unsigned)si >= (unsigned)p.Hc || (unsigned)sj >= (unsigned)p.Wc) continue;
      for (int ch = 0; ch < 16; ch++) {
        const int c = ch & 3;
        if (c == 3) continue;
        const float xv = p.x[(((long)n * 3 + c) * p.H + 2 * si + (ch >> 3)) * p.W + 2 * sj + ((ch >> 2) & 1)];
        acc = fmaf(xv, p.w[((co * 4 + a) * 4 + b) * 16 + ch], acc);
      }
    }
  return acc;
}

__global__ __launch_bounds__(512) void stem_fused_kernel(const StemP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const sm = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int py0 = ty * 7, px0 = tx * 15;
  const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;              // first convolution output of the tile
  const int sy0 = oy0 - 2, sx0 = ox0 - 2;                      // first space-to-depth pixel of the patch
  const F16Guard guard = f16_guard_load(p.x_slot);
  const float sx = f16_scale_of(*p.x_slot);

  if (f16_guard_bad(guard)) {   // an image whose dynamic range defeats fp16 (uniform over the grid): exact products, element by element
    float amx = 0.f;
    for (int o = tid; o < 7 * 15 * 64; o += 512) {
      const int co = o & 63, q = (o >> 6) % 15, j = (o >> 6) / 15;
      const int py = py0 + j, px = px0 + q;
      if (py >= p.Hp || px >= p.Wp) continue;
      float m = 0.f;
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          const int oy = 2 * py + dy, ox = 2 * px + dx;
          if ((unsigned)oy >= (unsigned)p.Hc || (unsigned)ox >= (unsigned)p.Wc) continue;
          m = fmaxf(m, fmaxf(stem_exact(p, n, oy, ox, co) * p.scale[co] + p.shift[co], 0.f));
        }
      p.y[(((long)n * p.Hp + py) * p.Wp + px) * 64 + co] = m;
      amx = fmaxf(amx, m);
    }
    if (p.amax_out && amx > 0.f) atomicMax(p.amax_out, __builtin_bit_cast(unsigned, amx));
    return;
  }

  // ---- the weight planes of all 16 steps: 64 copies of 1 KiB, eight per wave.  LDS image: [plane][step][32-channel block][1 KiB]
  {
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpl, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int item = wave + 8 * i, q = item >> 5, rest = item & 31;   // rest = step * 2 + block: the planes' own order
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(sm + SB_OFF + item * 1024), 16,
                                               (int)((q * p.wpl_stride + (long)rest * 512 + lane * 8) * 2), 0, 0, 0);
    }
  }
  // ---- the patch: item = (pixel, row parity bh): three 8-byte loads (c = 0, 1, 2 at columns 2 sj, 2 sj + 1), one 16-byte LDS store
  // per plane.  16 channels of a pixel = (bh, bw, c) with c = 3 zero: the order of StemWithFixedBatchNorm._s2d_weight
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int rnd = 0; rnd < 3; rnd++) {
    const int item = tid + 512 * rnd;
    if (item < 2 * SPIX) {
      const int pix = item >> 1, bh = item & 1;
      const int pr = pix / SPW, pc = pix - pr * SPW;
      const int si = sy0 + pr, sj = sx0 + pc;
      f32x2 v[3];
      const bool ok = (unsigned)si < (unsigned)p.Hc && (unsigned)sj < (unsigned)p.Wc;
#pragma unroll
      for (int c = 0; c < 3; c++)
        v[c] = ok ? *(const f32x2*)(p.x + (((long)n * 3 + c) * p.H + 2 * si + bh) * p.W + 2 * sj) : f32x2{0.f, 0.f};
      uint2 o0[2], o1[2];
      split4h(f32x4{v[0][0], v[1][0], v[2][0], 0.f}, sx, o0);     // bw = 0: c0 c1 c2 0
      split4h(f32x4{v[0][1], v[1][1], v[2][1], 0.f}, sx, o1);     // bw = 1
      char* const dst = sm + pix * 32 + ((bh ^ (pix >> 3)) & 1) * 16;
      *(uint4*)dst = uint4{o0[0].x, o0[0].y, o1[0].x, o1[0].y};
      *(uint4*)(dst + SPL) = uint4{o0[1].x, o0[1].y, o1[1].x, o1[1].y};
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- main loop: wave w owns output rows 2 w, 2 w + 1 of the tile (a = 0, 1) x 64 channels (b = 0, 1)
  const int lr = lane & 31, kh2 = lane >> 5;
  const int boff = SB_OFF + lr * 32 + (((kh2 ^ (lr >> 3)) & 1) << 4);
  auto a_addr = [&](int a, int ta, int tb) {
    const int pidx = (2 * wave + a + ta) * SPW + lr + tb;
    return pidx * 32 + (((kh2 ^ (pidx >> 3)) & 1) << 4);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  auto fread = [&](int s, f16x8 (&fa)[2][2], f16x8 (&fb)[2][2]) {
    const int ta = s >> 2, tb = s & 3;
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int ad = a_addr(a, ta, tb);
      fa[0][a] = *(const f16x8*)(sm + ad);
      fa[1][a] = *(const f16x8*)(sm + ad + SPL);
    }
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int b = 0; b < 2; b++) fb[q][b] = *(const f16x8*)(sm + boff + ((q * 16 + s) * 2 + b) * 1024);
  };
  auto mma = [&](const f16x8 (&fa)[2][2], const f16x8 (&fb)[2][2]) {   // products (h, l), (l, h), (h, h): the tiled kernel's order
#pragma unroll
    for (int pr = 0; pr < 3; pr++) {
      const int qa = pr == 1 ? 1 : 0, qb = pr == 0 ? 1 : 0;
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
    }
  };
  {
    f16x8 faP[2][2], fbP[2][2], faQ[2][2], fbQ[2][2];
    fread(0, faP, fbP);
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
      fread(s + 1, faQ, fbQ);
      mma(faP, fbP);
      if (s + 2 < 16) fread(s + 2, faP, fbP);
      mma(faQ, fbQ);
    }
  }
  __syncthreads();   // everybody is done with the patch and the weights: LDS becomes the pooling stage

  // ---- epilogue.  acc[a][b][r]: output row 2 w + a, column 8 (r / 4) + r % 4 + 4 (lane / 32), channel 32 b + lane % 32
  const int col_l = lane & 31, rq = lane >> 5;
  const float inv = 1.f / (sx * *p.s_w);
  float sc[2], sh[2];
#pragma unroll
  for (int b = 0; b < 2; b++) { sc[b] = p.scale[b * 32 + col_l] * inv; sh[b] = p.shift[b * 32 + col_l]; }
  f32x16 m01[2];
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int ox = ox0 + 8 * (r >> 2) + (r & 3) + 4 * rq;
      const bool cok = (unsigned)ox < (unsigned)p.Wc;
      float v[2];
#pragma unroll
      for (int a = 0; a < 2; a++) {
        const int oy = oy0 + 2 * wave + a;
        const float t = fmaxf(acc[a][b][r] * sc[b] + sh[b], 0.f);
        v[a] = (cok && (unsigned)oy < (unsigned)p.Hc) ? t : 0.f;   // outside the image: no candidate (every window holds a value >= 0)
      }
      acc[0][b][r] = v[0];
      m01[b][r] = fmaxf(v[0], v[1]);
    }
  float* const xr = lds;                 // [wave][32 columns][64 channels]: a wave's FIRST row, for the wave above it
  float* const hv = lds + 8 * 32 * 64;   // [7 pooled rows][32 columns][64 channels]: vertical maxima
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int r = 0; r < 16; r++)
      xr[(wave * 32 + 8 * (r >> 2) + (r & 3) + 4 * rq) * 64 + b * 32 + col_l] = acc[0][b][r];
  __syncthreads();
  if (wave < 7) {
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int col = 8 * (r >> 2) + (r & 3) + 4 * rq;
        hv[(wave * 32 + col) * 64 + b * 32 + col_l] = fmaxf(m01[b][r], xr[((wave + 1) * 32 + col) * 64 + b * 32 + col_l]);
      }
  }
  __syncthreads();
  float amx = 0.f, asum = 0.f, acnt = 0.f;
  for (int o = tid; o < 7 * 15 * 16; o += 512) {
    const int c4 = o & 15, q = (o >> 4) % 15, j = (o >> 4) / 15;
    const int py = py0 + j, px = px0 + q;
    if (py >= p.Hp || px >= p.Wp) continue;
    const f32x4* const row = (const f32x4*)(hv + (j * 32 + 2 * q) * 64) + c4;
    const f32x4 t0 = row[0], t1 = row[16], t2 = row[32];
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      m[e] = fmaxf(fmaxf(t0[e], t1[e]), t2[e]);
      amx = fmaxf(amx, m[e]);
      asum += m[e];
    }
    acnt += 4.f;
    *(f32x4*)(p.y + (((long)n * p.Hp + py) * p.Wp + px) * 64 + c4 * 4) = m;
  }
  if (p.amax_out) {
    __syncthreads();
    float* const red = lds;
    const bool stats = p.amax_stats && (blockIdx.x & 63) == 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
    if (stats) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { asum += __shfl_xor(asum, o, 64); acnt += __shfl_xor(acnt, o, 64); }
    }
    if (lane == 0) { red[wave] = amx; red[16 + wave] = asum; red[32 + wave] = acnt; }
    __syncthreads();
    if (tid == 0) {
      float m = red[0], sm_ = red[16], cn = red[32];
      for (int i = 1; i < 8; i++) { m = fmaxf(m, red[i]); sm_ += red[16 + i]; cn += red[32 + i]; }
      const unsigned bits = __builtin_bit_cast(unsigned, m);
      if (m > 0.f && bits > __hip_atomic_load(p.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p.amax_out, bits);
      if (stats && cn > 0.f) {
        const int k = (blockIdx.x >> 6) & 15;
        atomicAdd((float*)p.amax_out + 1 + k, sm_);
        atomicAdd((float*)p.amax_out + 17 + k, cn);
      }
    }
  }
}


// ------------------------------------------------------------------------------------ 3x3, 64 -> 64 channels (layer1's conv2)
// The 3x3 convolutions of layer1 (backbone/resnet.py:254-274 with 64 bottleneck channels, frozen: forward only, the teacher's
// 8 x 256^2 and the student's 4 x 256^2 maps) ran on the tiled kernel's 128 x 64 tile at 3.5 x their HBM time (188 us for 268 MB):
// K = 576 is 36 steps, each re-fetching its rows' pixels through the L2 -> LDS path nine times over, behind a per-tile prologue.
// Same recipe as the stem: a block owns 8 x 16 output pixels, gathers their 10 x 18 input patch ONCE (fp32, 16-byte loads,
// coalesced), scales and splits it into the two fp16 terms on the way into LDS ([16-channel slab][pixel][32 B] per plane), and the
// nine taps read their fragments from the patch at pixel offset 18 kh + kw.  The weight planes stream through two 16 KB buffers,
// one tap (four steps) ahead.  4 waves (2 output rows x 16 columns x 64 channels each), 80 KB of LDS: two blocks per CU.
// Same products, same order as the tiled kernel: bit-identical (tests/test_stem_gpu.py).
constexpr int C64_PW = 18, C64_PIX = 180, C64_SLAB = 192 * 32, C64_PL = 4 * C64_SLAB, C64_BOFF = 2 * C64_PL;

__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(const ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const sm = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.Wo + 15) >> 4, tiles_y = (p.Ho + 7) >> 3;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int img = bid / tiles_y;
  const int oy0 = ty * 8, ox0 = tx * 16;
  const F16Guard guard = f16_guard_load(p.guard_x);
  const float sx = f16_scale_of(*p.f16_sx);
  if (f16_guard_bad(guard)) {   // (uniform over the grid) exact fp32 products; pixels of a ragged tile that wrap are written twice
    conv_slow_tile((img * p.Ho + oy0) * p.Wo + ox0, 16, p.Wo, 128, 0, 64, nullptr, false, tid, 256, blockIdx.x);
    return;
  }
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpl, 0, 0x7ffffff0, 0x00020000);
  auto copy_chunk = [&](int t, int buf) {   // the weight planes of tap t: 4 steps x 2 planes x 2 blocks of 1 KiB, four per wave
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int item = wave * 4 + i, st = item >> 2, q = (item >> 1) & 1, blk = item & 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(sm + C64_BOFF + buf * 16384 + ((st * 2 + q) * 2 + blk) * 1024), 16,
                                               (int)((q * p.wpl_stride + (long)((t * 4 + st) * 2 + blk) * 512 + lane * 8) * 2), 0, 0, 0);
    }
  };
  copy_chunk(0, 0);
  // ---- the patch: item = (pixel, 4-channel quad): one 16-byte load, two 8-byte LDS stores (h, l)
  {
    f32x4 v[12];
#pragma unroll
    for (int rnd = 0; rnd < 12; rnd++) {
      const int item = tid + 256 * rnd;
      const int pix = item >> 4, cq = item & 15;
      const int pr = pix / C64_PW, pc = pix - pr * C64_PW;
      const int iy = oy0 - 1 + pr, ix = ox0 - 1 + pc;
      const bool ok = pix < C64_PIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      v[rnd] = ok ? ldg4(p.x + (((long)img * p.H + iy) * p.W + ix) * 64 + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int rnd = 0; rnd < 12; rnd++) {
      const int item = tid + 256 * rnd;
      const int pix = item >> 4, cq = item & 15;
      if (pix < C64_PIX) {
        uint2 o[2];
        split4h(v[rnd], sx, o);
        char* const dst = sm + (cq >> 2) * C64_SLAB + pix * 32 + (((((cq & 3) >> 1) ^ (pix >> 3)) & 1) << 4) + (cq & 1) * 8;
        *(uint2*)dst = o[0];
        *(uint2*)(dst + C64_PL) = o[1];
      }
    }
  }
  // ---- main loop.  Wave w: output rows 2 w, 2 w + 1 of the tile; MFMA row i = (i / 16, i % 16); 2 column tiles of 32 channels
  const int lr = lane & 31, kh2 = lane >> 5;
  const int prow = 2 * wave + (lr >> 4), pcol = lr & 15;
  const int boff = C64_BOFF + lr * 32 + (((kh2 ^ (lr >> 3)) & 1) << 4);
  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
#pragma unroll 1
  for (int t = 0; t < 9; t++) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tap t's planes (and, first time, the patch) are in LDS
    __syncthreads();
    if (t + 1 < 9) copy_chunk(t + 1, (t + 1) & 1);                // into the buffer everybody finished reading before this barrier
    const int kh = t / 3, kw = t - kh * 3;
    const int pidx = (prow + kh) * C64_PW + pcol + kw;
    const char* const pa = sm + pidx * 32 + (((kh2 ^ (pidx >> 3)) & 1) << 4);
    const char* const pb = sm + boff + (t & 1) * 16384;
#pragma unroll
    for (int st = 0; st < 4; st++) {
      f16x8 fa[2], fb[2][2];
      fa[0] = *(const f16x8*)(pa + st * C64_SLAB);
      fa[1] = *(const f16x8*)(pa + st * C64_SLAB + C64_PL);
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int b = 0; b < 2; b++) fb[q][b] = *(const f16x8*)(pb + ((st * 2 + q) * 2 + b) * 1024);
#pragma unroll
      for (int pr = 0; pr < 3; pr++) {   // products (h, l), (l, h), (h, h): the tiled kernel's order
        const int qa = pr == 1 ? 1 : 0, qb = pr == 0 ? 1 : 0;
#pragma unroll
        for (int b = 0; b < 2; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[qa], fb[qb][b], acc[b], 0, 0, 0);
      }
    }
  }
  // ---- epilogue from the registers: acc[b][r] = MFMA row i = 8 (r / 4) + r % 4 + 4 (lane / 32), channel 32 b + lane % 32
  const int col_l = lane & 31, rq = lane >> 5;
  const float inv = 1.f / (sx * *p.f16_sw);
  const int ybytes = (int)((long)p.M * 64 * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, ybytes, 0x00020000);
  float amx = 0.f, asum = 0.f, acnt = 0.f;
#pragma unroll
  for (int b = 0; b < 2; b++) {
    const float sc = (p.scale ? p.scale[b * 32 + col_l] : 1.f) * inv, sh = p.shift ? p.shift[b * 32 + col_l] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int i = 8 * (r >> 2) + (r & 3) + 4 * rq;
      const int oy = oy0 + 2 * wave + (i >> 4), ox = ox0 + (i & 15);
      float v = acc[b][r] * sc + sh;
      if (p.relu) v = fmaxf(v, 0.f);
      const bool ok = oy < p.Ho && ox < p.Wo;
      const float av = ok ? fabsf(v) : 0.f;
      amx = fmaxf(amx, av);
      asum += av;
      acnt += ok ? 1.f : 0.f;
      const unsigned off = ok ? (unsigned)(((img * p.Ho + oy) * p.Wo + ox) * 64 + b * 32 + col_l) * 4u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)off, 0, 0);
    }
  }
  if (p.amax_out) {
    __syncthreads();
    float* const red = lds;
    const bool stats = p.amax_stats && (blockIdx.x & 63) == 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
    if (stats) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { asum += __shfl_xor(asum, o, 64); acnt += __shfl_xor(acnt, o, 64); }
    }
    if (lane == 0) { red[wave] = amx; red[16 + wave] = asum; red[32 + wave] = acnt; }
    __syncthreads();
    if (tid == 0) {
      float m = red[0], sm_ = red[16], cn = red[32];
      for (int i = 1; i < 4; i++) { m = fmaxf(m, red[i]); sm_ += red[16 + i]; cn += red[32 + i]; }
      const unsigned bits = __builtin_bit_cast(unsigned, m);
      if (m > 0.f && bits > __hip_atomic_load(p.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p.amax_out, bits);
      if (stats && cn > 0.f) {
        const int k = (blockIdx.x >> 6) & 15;
        atomicAdd((float*)p.amax_out + 1 + k, sm_);
        atomicAdd((float*)p.amax_out + 17 + k, cn);
      }
    }
  }
}


}  // namespace

namespace mmtconv {
// 3x3 / stride 1 / pad 1 with 64 input and 64 output channels on fp32 tensors, plain affine (+ ReLU) epilogue: layer1's conv2
bool c64_shape(const ConvP& p) {
  const char* e = getenv("MMT_C64");   // read per call (A/B timing, parity tests)
  if (e && atoi(e) == 0) return false;
  return p.Cin == 64 && p.Cout == 64 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Ho == p.H && p.Wo == p.W && !p.io &&
         !p.ypl && !p.mul && !p.mask && p.res_mode == 0 && p.out_stride == 1 && p.wpl && p.w && p.f16_sx && p.f16_ax && p.f16_sw &&
         (long)p.M * 64 * 4 < (1L << 31);
}

int launch_c64(const ConvP& p, hipStream_t s) {
  constexpr int LDS = C64_BOFF + 2 * 16384;
  static bool done = false;
  if (!done) {
    const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  const int tiles = p.N * ((p.Ho + 7) >> 3) * ((p.Wo + 15) >> 4);
  hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(tiles), dim3(256), LDS, s, p);
  MMT_LAUNCH_CHECK();
  return 0;
}
}  // namespace mmtconv

extern "C" int mmt_stem_fused(const float* x, int N, int H, int W, const float* w_s2d, const void* w_planes, long w_plane_stride,
                              const float* s_w, const float* scale, const float* shift, const float* x_slot, float* y, float* y_slot,
                              void* stream) {
  if (!x || !w_s2d || !w_planes || !s_w || !scale || !shift || !x_slot || !y || N <= 0 || H < 8 || W < 8 || (H & 3) || (W & 3) ||
      ((size_t)w_planes & 15) || (w_plane_stride & 7) || ((size_t)x & 7) || ((size_t)y & 15) || precision() != 3)
    return MMT_EINVAL;
  if ((long)N * 3 * H * W >= (1L << 31)) return MMT_EINVAL;
  StemP p;
  p.x = x; p.wpl = (const unsigned short*)w_planes; p.wpl_stride = w_plane_stride; p.w = w_s2d;
  p.scale = scale; p.shift = shift; p.y = y; p.x_slot = x_slot; p.s_w = s_w;
  p.amax_out = (unsigned*)y_slot; p.amax_stats = y_slot ? 1 : 0;
  p.N = N; p.H = H; p.W = W; p.Hc = H / 2; p.Wc = W / 2; p.Hp = H / 4; p.Wp = W / 4;
  p.tiles_y = mmt_cdiv(p.Hp, 7); p.tiles_x = mmt_cdiv(p.Wp, 15);
  static bool done = false;
  if (!done) {
    const hipError_t e = hipFuncSetAttribute((const void*)stem_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(stem_fused_kernel, dim3(N * p.tiles_y * p.tiles_x), dim3(512), STEM_LDS, (hipStream_t)stream, p);
  MMT_LAUNCH_CHECK();
  return 0;
}
