// Weight gradient of a stride-1 convolution from ROW-BLOCKED fp16 planes of both operands (round 5):
//   dW[co][tap][ci] += rowscale[co] / (s_x s_dy) * sum over pixels  dy[px][co] * x[px + shift(tap)][ci]
// Replaces autograd's conv2d weight gradient behind maskrcnn_benchmark.layers.Conv2d (layers/misc.py:30-43) for the 3x3 layers whose
// input planes the forward pass and whose gradient planes the data-gradient launch have already produced (mmt_split_planes_f16_rb:
// [N H][C / 16][W][16]) -- conv_wgrad_pipe_kernel (conv_wgrad.hip) reads the fp32 tensors, splits them in registers and transposes
// them through LDS with vector stores: 115 vector instructions per 12 MFMAs, MFMA-busy 0.35.
//
// The reduction index of this GEMM is the PIXEL, the strided index of both operands.  Row-blocked planes make it cheap twice:
//   * 32 consecutive pixels of an image row x one 16-channel block are ONE run of 1 KiB in memory: one buffer_load ... lds per
//     (plane, channel block, 32-pixel super-step), lane-linear into LDS as a [32 pixels][16 channels] block;
//   * that block is exactly what ds_read_b64_tr_b16 wants (tools/micro/trread.hip: lane i of a 16-lane group receives column i of a
//     [4 rows][16 columns] block, rows any stride apart): two of them give a lane the 8 consecutive pixels of its channel = its
//     operand of v_mfma_f32_32x32x16_f16.  No vector arithmetic, no LDS stores in the loop.
// Block = 8 matrix waves + 4 copy waves (the recipe of conv_pg_kernel): two GROUPS of 2 x 2 waves work on two consecutive pixel
// ranges of one 128 (co) x 128 (tap, ci) tile and add their accumulators through LDS; pixel ranges across blocks go to slabs of the
// caller's workspace that a second launch (wgrad_pl_reduce_kernel) sums (in-launch reduction was measured slower for the weight
// gradient's short ranges: profiles/r05_history.md).  Cin % 128 == 0 puts a column tile inside ONE tap: the x offset of a block is fixed.
// Products and their order per 16-pixel step: (h l), (l h), (h h), fp32 accumulation -- the arithmetic of the default mode.
// Roofline: MFMA, 833 TFLOP/s algorithmic; copies 32 KB per 32-pixel super-step and group = 96 MFMAs.
#include "conv_shared.h"

namespace {

constexpr unsigned WG_OOB = 0x80000000u;
constexpr int WG_BLK = 1024 + 128;      // bytes of one [32 pixels][16 channels] block in LDS: the pad puts the blocks of the two 16-lane
                                        // groups of a half wave (channel blocks cb, cb + 1) on different halves of the banks
constexpr int WG_GSTAGE = 32 * WG_BLK;  // a group's stage: A (dy) planes h, l x 8 channel blocks, then B (x) the same
constexpr int WG_STAGE = 2 * WG_GSTAGE, WG_S = 2;

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 tr_pair(const char* p) {   // 8 consecutive pixels of this lane's channel: keys 0..3 and 4..7
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 128));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(f16x8, v);
}

#ifdef MMT_PG_ABLATE
#define WGPL_DBG(p) ((p).dbg)
#else
#define WGPL_DBG(p) 0
#endif

struct WgP {
  const float* x; const float* dy;                       // fp32 tensors (the slow, exact path)
  const unsigned short* xpl; long xpl_stride;            // row-blocked planes of x * s_x
  const unsigned short* dpl; long dpl_stride;            // ... of dy * s_dy
  const float* s_x; const float* s_dy;                   // device scalars: the planes' scales
  const float* guard_x; const float* guard_dy;           // statistics slots (range guard) or null
  const float* rowscale; float* dw; float* ws; float* dbias;
  int N, H, W, Cin, Cout, KH, KW, pad, ksplit;
  int dbg;   // (tools build only, -DMMT_PG_ABLATE = `make ablate`: 1 = every copy out of range -- zeros, no memory traffic; 2 = no fragment
             // reads.  Wrong results; the product library compiles the arms out, WGPL_DBG below)
  int lag;   // round 6: bit 0 / bit 1 -- the planes of x / of dy were written by their producer's epilogue with a scale fixed beforehand
};

// bid_in / nwg: this block's index among the blocks of its launch -- or, in a grouped launch (wgrad_pl_group_kernel), of its item
__device__ __forceinline__ void wgrad_pl_body(const WgP& p, const int bid_in, const int nwg_in) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ring = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool copier = wave >= 8;
  const int grp = copier ? (wave - 8) >> 1 : wave >> 2;   // pixel range (K group) this wave works on / serves
  const int gw = wave & 3, wm = gw >> 1, wn = gw & 1;     // matrix waves: 2 x 2 waves of 64 x 64
  const int NP = p.KH * p.KW * p.Cin, tiles_n = NP >> 7;
  int bid = bid_in;
  {  // XCD-aware order: the ksplit blocks of a tile read the same weight-sized output region, neighbours read the same pixels
    const int nwg = nwg_in, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile = bid % ((p.Cout >> 7) * tiles_n), ks = bid / ((p.Cout >> 7) * tiles_n);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int co0 = tile_m * 128, n0 = tile_n * 128;
  const int tap = n0 / p.Cin, ci0 = n0 - tap * p.Cin;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int dkh = kh - p.pad, dkw = kw - p.pad;
  const F16Guard gx = f16_guard_load(p.guard_x), gd = f16_guard_load(p.guard_dy);
  // super-steps (32 pixels of one image row) of this wave's group: range r = ks * 2 + grp of R = 2 * ksplit
  const int T = (p.N * p.H * p.W) >> 5, R = 2 * p.ksplit, rng = ks * 2 + grp;
  const int t0 = (int)((long)rng * T / R), nt = (int)((long)(rng + 1) * T / R) - t0;
  const int nt_max = (T + R - 1) / R;
  const int segs = p.W >> 5;
  char* const gring = ring + grp * WG_GSTAGE;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  float bsum[2] = {0.f, 0.f};
  const bool do_bias = p.dbias != nullptr && tile_n == 0 && wn == 0 && !copier;
  const bool slow = ((p.lag & 1) ? f16_guard_bad_lag(gx, *p.s_x) : f16_guard_bad(gx)) ||
                    ((p.lag & 2) ? f16_guard_bad_lag(gd, *p.s_dy) : f16_guard_bad(gd));

  if (copier && !slow) {
    // ================================================================ copy wave: 16 of its group's 32 blocks per super-step
    const int sub = (wave - 8) & 1;                    // 0: the A (dy) blocks, 1: the B (x) blocks
    const long n_d = (long)p.N * p.H * p.W * p.Cout, n_x = (long)p.N * p.H * p.W * p.Cin;
    // (x: the descriptor starts 64 bytes in front of the planes so that the tap's shift of -1 pixel is a non-negative offset)
    const __amdgpu_buffer_rsrc_t rs0 = sub == 0 ? __builtin_amdgcn_make_buffer_rsrc((void*)p.dpl, 0, (int)(n_d * 2), 0x00020000)
                                                : __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.xpl - 64), 0, (int)(n_x * 2 + 64), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = sub == 0 ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.dpl + p.dpl_stride), 0, (int)(n_d * 2), 0x00020000)
                                                : __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(p.xpl + p.xpl_stride) - 64), 0, (int)(n_x * 2 + 64), 0x00020000);
    const int cbn = (sub == 0 ? p.Cout : p.Cin) >> 4;  // channel blocks per pixel row of the operand
    const int cb0 = (sub == 0 ? co0 : ci0) >> 4;
    const int px = lane >> 1;
    int f_t = 0;                                       // super-steps issued so far
    auto copy_step = [&](int stage) {
      const bool real = f_t < nt && !(WGPL_DBG(p) & 1);
      const int t = t0 + (real ? f_t : 0);
      const int row = t / segs, w0 = (t - row * segs) << 5;   // image row (n H + h), first pixel
      unsigned vo = (unsigned)lane * 16u;
      int soff;
      if (sub == 0) {
        soff = (row * cbn + cb0) * (p.W * 32) + w0 * 32;
        if (!real) vo = WG_OOB;
      } else {
        const int h = row % p.H, ih = h + dkh, iw = w0 + px + dkw;
        const bool ok = real && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        soff = ((row + dkh) * cbn + cb0) * (p.W * 32) + (w0 + dkw) * 32 + 64;
        if (!ok) { vo = WG_OOB; }
        if (!(real && (unsigned)ih < (unsigned)p.H)) soff = 64;   // (no lane reads: keep the scalar offset inside the tensor)
      }
      char* const st = gring + stage * WG_STAGE + sub * 16 * WG_BLK;
#pragma unroll
      for (int cb = 0; cb < 8; cb++) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(st + cb * WG_BLK), 16, (int)vo, soff + cb * (p.W * 32), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(st + (8 + cb) * WG_BLK), 16, (int)vo, soff + cb * (p.W * 32), 0, 0);
      }
      f_t++;
    };
    copy_step(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // (0) super-step 0 has landed
    copy_step(1);
    for (int t = 0; t < nt_max; t++) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // super-step t + 1 has landed ...
      __builtin_amdgcn_s_barrier();                      // ... and every matrix wave has finished reading stage t % 2
      copy_step(t & 1);                                  // super-step t + 2 into it
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (!slow) {
    // ================================================================ matrix wave
    const int i16 = lane & 15, g1 = (lane >> 4) & 1, khalf = lane >> 5;
    const int loff = khalf * 256 + (i16 >> 2) * 32 + (i16 & 3) * 8;   // inside a block: pixel 8 khalf + (i16 >> 2), columns 4 (i16 & 3) ..
    // block of plane q, row / column block a of this wave: A: q 8 + (wm 4 + a 2 + g1), B: 16 + q 8 + (wn 4 + b 2 + g1)
    const char* const fA = gring + (wm * 4 + g1) * WG_BLK + loff;
    const char* const fB = gring + (16 + wn * 4 + g1) * WG_BLK + loff;
    __builtin_amdgcn_s_barrier();                      // (0)
    for (int t = 0; t < nt_max; t++) {
      const char* const sa = fA + (t & 1) * WG_STAGE;
      const char* const sb = fB + (t & 1) * WG_STAGE;
      if (t < nt) {
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {   // the two 16-pixel steps of the super-step
          f16x8 fa[2][2], fb[2][2];
#pragma unroll
          for (int q = 0; q < 2; q++)
#pragma unroll
            for (int a = 0; a < 2; a++) {
              if ((WGPL_DBG(p) & 2) && t > 0) continue;
              fa[q][a] = tr_pair(sa + (q * 8 + a * 2) * WG_BLK + k2 * 512);
              fb[q][a] = tr_pair(sb + (q * 8 + a * 2) * WG_BLK + k2 * 512);
            }
          if (do_bias) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
              for (int e = 0; e < 8; e++) bsum[a] += (float)fa[0][a][e] + (float)fa[1][a][e];
          }
#pragma unroll
          for (int pr = 0; pr < 3; pr++) {
            const int qa = pr == 1 ? 1 : 0, qb = pr == 0 ? 1 : 0;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
              for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  const float sxy = *p.s_x * *p.s_dy;
  if (slow) {
    // ---- an operand's dynamic range defeats fp16 (f16_guard_bad): exact fp32 products from the fp32 tensors for this wave's
    // part of the tile over its group's pixels, scaled like the fast path's sums (slow: a handful of launches until the host has
    // moved the site to the 3-term bf16 split)
    if (!copier) {
      const int HW = p.H * p.W;
      for (int m = t0 * 32; m < (t0 + nt) * 32; m++) {
        const int img = m / HW, rem = m - img * HW, h = rem / p.W, w = rem - h * p.W;
        const int ih = h + dkh, iw = w + dkw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const float xv = ok ? p.x[((long)(img * p.H + ih) * p.W + iw) * p.Cin + ci0 + wn * 64 + b * 32 + (lane & 31)] : 0.f;
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
              const int co = co0 + wm * 64 + a * 32 + 8 * (r >> 2) + (r & 3) + 4 * (lane >> 5);
              acc[a][b][r] = fmaf(p.dy[(long)m * p.Cout + co], xv, acc[a][b][r]);
            }
        }
        if (do_bias && lane < 32) {
#pragma unroll
          for (int a = 0; a < 2; a++) bsum[a] += p.dy[(long)m * p.Cout + co0 + wm * 64 + a * 32 + lane] * *p.s_dy;
        }
      }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[a][b][r] *= sxy;
    }
  }
  __syncthreads();

  // ---- bias gradient (column tile 0 only): this lane's channel over its half of every 16-pixel step; the other half sits 32 lanes on
  if (do_bias) {
    const float inv_d = 1.f / *p.s_dy;
#pragma unroll
    for (int a = 0; a < 2; a++) {
      float v = bsum[a];
      if (!slow) v += __shfl_xor(v, 32, 64);
      if (lane < 32) atomicAdd(p.dbias + co0 + wm * 64 + a * 32 + lane, v * inv_d);
    }
  }

  // ---- the two groups' accumulators meet in LDS: sub-tile i = (a, b) of every wave is finished by group i & 1 (its own sum from
  // registers, the other group's from LDS, added in range order)
  {
    f32x4* const xch = (f32x4*)ring;
    const int gtid = tid & 255;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (!copier && (i & 1) != grp) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++)
          xch[(i * 4 + r4) * 256 + gtid] = f32x4{acc[i >> 1][i & 1][4 * r4], acc[i >> 1][i & 1][4 * r4 + 1],
                                                 acc[i >> 1][i & 1][4 * r4 + 2], acc[i >> 1][i & 1][4 * r4 + 3]};
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (!copier && (i & 1) == grp) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
          const f32x4 v = xch[(i * 4 + r4) * 256 + gtid];
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const float mine = acc[i >> 1][i & 1][4 * r4 + c];
            acc[i >> 1][i & 1][4 * r4 + c] = grp == 0 ? mine + v[c] : v[c] + mine;   // range 2 ks first, then 2 ks + 1
          }
        }
      }
    }
  }
  if (copier) return;

  // ---- out: straight into dw (one pixel range per tile: dw += rowscale * sum / (s_x s_dy)) or into this range's slab (the sum /
  // (s_x s_dy); wgrad_reduce_kernel adds the slabs and applies rowscale).  An accumulator register = one co row x 32 columns per
  // half wave = whole 128-byte lines.
  const float inv = 1.f / sxy;
  float* const dst = p.ksplit > 1 ? p.ws + (long)ks * p.Cout * NP : p.dw;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if ((i & 1) != grp) continue;
    const int a = i >> 1, b = i & 1;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int co = co0 + wm * 64 + a * 32 + 8 * (r >> 2) + (r & 3) + 4 * (lane >> 5);
      float* const q = dst + (long)co * NP + n0 + wn * 64 + b * 32 + (lane & 31);
      if (p.ksplit > 1) *q = acc[a][b][r] * inv;
      else *q = *q + acc[a][b][r] * ((p.rowscale ? p.rowscale[co] : 1.f) * inv);
    }
  }
}

__global__ __launch_bounds__(768) void wgrad_pl_kernel(const WgP p) { wgrad_pl_body(p, blockIdx.x, gridDim.x); }

// ---- round 6: the weight gradients of SEVERAL layers in one launch.  A backward pass hands its weight-gradient
// jobs to the side stream a batch at a time (layers/fused.py::flush_wgrads); launched one by one, a 3x3 layer at N = 2 has 36 tiles and
// is cut into 7 pixel ranges to fill the chip -- 18 super-steps per range between a prologue, an LDS exchange, a 64 KB slab store and a
// reduce launch.  In a group the tiles of all its layers fill the chip together: few or no pixel ranges per layer, no slabs, and what
// is left of the reduce work is one launch for the whole group.  A block finds its item by its index (items' block ranges start at
// multiples of 8: the XCD mapping of wgrad_pl_body stays what it is for a single launch).
constexpr int WG_MAXG = 12;
struct WgGroup { WgP it[WG_MAXG]; int first[WG_MAXG + 1]; int n; };
static_assert(sizeof(WgGroup) <= 3584, "kernel-argument segment");
__global__ __launch_bounds__(768) void wgrad_pl_group_kernel(const WgGroup g) {
  int i = 0;
  for (int k = 1; k < g.n; k++) i = (int)blockIdx.x >= g.first[k] ? k : i;
  const int local = (int)blockIdx.x - g.first[i];
  const WgP p = g.it[i];
  const int real = (p.Cout >> 7) * ((p.KH * p.KW * p.Cin) >> 7) * p.ksplit;
  if (local >= real) return;   // (padding up to the next multiple of 8)
  wgrad_pl_body(p, local, real);
}

// dw_i[co][n] += rowscale_i[co] * sum_s ws_i[s][co][n] for every item of a group in ONE launch (the slabs hold true partial sums)
constexpr int WG_MAXR = 24;
struct WgReduceGroup { mmtconv::WgReduceItem it[WG_MAXR]; int first[WG_MAXR + 1]; int n; };
static_assert(sizeof(WgReduceGroup) <= 3584, "kernel-argument segment");
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const WgReduceGroup g) {
  int k = 0;
  for (int j = 1; j < g.n; j++) k = (int)blockIdx.x >= g.first[j] ? j : k;
  const mmtconv::WgReduceItem it = g.it[k];
  const int nb = g.first[k + 1] - g.first[k], lb = (int)blockIdx.x - g.first[k];
  const long n4 = (long)it.Cout * it.NP / 4, slab = (long)it.Cout * it.NP;
  for (long i = (long)lb * 256 + threadIdx.x; i < n4; i += (long)nb * 256) {
    f32x4 o = ((f32x4*)it.dw)[i];
    f32x4 a = ((const f32x4*)it.ws)[i];
    int s = 1;
    for (; s + 8 <= it.splits; s += 8) {
      f32x4 b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) b[u] = *(const f32x4*)(it.ws + (s + u) * slab + i * 4);
#pragma unroll
      for (int u = 0; u < 8; u++) a += b[u];
    }
    for (; s < it.splits; s++) a += *(const f32x4*)(it.ws + s * slab + i * 4);
    const float sc = it.rowscale ? it.rowscale[(int)((i * 4) / it.NP)] : 1.f;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] += a[e] * sc;
    ((f32x4*)it.dw)[i] = o;
  }
}

// dw[co][n] += rowscale[co] * sum_s ws[s][co][n]   (the slabs hold sums already divided by s_x s_dy)
__global__ __launch_bounds__(256) void wgrad_pl_reduce_kernel(const float* __restrict__ ws, int splits, int Cout, int NP,
                                                              const float* __restrict__ rowscale, float* __restrict__ dw) {
  const long n4 = (long)Cout * NP / 4, slab = (long)Cout * NP;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 o = ((f32x4*)dw)[i];
    f32x4 a = ((const f32x4*)ws)[i];
    int s = 1;
    for (; s + 8 <= splits; s += 8) {
      f32x4 b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) b[u] = *(const f32x4*)(ws + (s + u) * slab + i * 4);
#pragma unroll
      for (int u = 0; u < 8; u++) a += b[u];
    }
    for (; s < splits; s++) a += *(const f32x4*)(ws + s * slab + i * 4);
    const float sc = rowscale ? rowscale[(int)((i * 4) / NP)] : 1.f;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] += a[e] * sc;
    ((f32x4*)dw)[i] = o;
  }
}

// does the plane-fed weight gradient take this layer, and with how many pixel ranges across blocks?  0 = no
int wgpl_splits(const mmt_conv_args* a) {
  if (precision() != 3 || a->stride != 1 || a->Ho != a->H || a->Wo != a->W || (a->W & 31) || (a->Cin & 127) || (a->Cout & 127) ||
      a->KH != a->KW || a->pad != (a->KH - 1) / 2 || !(a->KH & 1) || a->N < 1 || a->io_bf16 ||
      (long)a->N * a->H * a->W * (a->Cin > a->Cout ? a->Cin : a->Cout) >= (1L << 29))
    return 0;
  const char* e = getenv("MMT_WGRAD_PLANES");   // read per call (A/B timing, parity tests)
  if (e && atoi(e) == 0) return 0;
  const long T = ((long)a->N * a->H * a->W) >> 5;
  const long tiles = (long)(a->Cout >> 7) * ((long)a->KH * a->KW * a->Cin >> 7);
  static const long target = getenv("MMT_WGPL_BLOCKS") ? atol(getenv("MMT_WGPL_BLOCKS")) : 256;   // (blocks per launch the pixel ranges aim at; the switch: profiles/r06_history.md)
  long ks = target / tiles;
  if (ks > T / 8) ks = T / 8;     // >= 4 super-steps per group and range
  if (ks < 1) ks = 1;
  if (T < 2) return 0;
  return (int)ks;
}

}  // namespace

namespace mmtconv {
int wgpl_eligible_splits(const mmt_conv_args* a) { return a ? wgpl_splits(a) : 0; }
long wgpl_super_steps(const mmt_conv_args* a) { return ((long)a->N * a->H * a->W) >> 5; }

int launch_wgrad_reduce_group(const WgReduceItem* items, int n, hipStream_t s) {
  for (int i0 = 0; i0 < n; i0 += WG_MAXR) {
    WgReduceGroup g;
    g.n = n - i0 < WG_MAXR ? n - i0 : WG_MAXR;
    int nb = 0;
    for (int i = 0; i < g.n; i++) {
      g.it[i] = items[i0 + i];
      g.first[i] = nb;
      long b = ((long)items[i0 + i].Cout * items[i0 + i].NP / 4 + 255) / 256;
      nb += (int)(b > 1024 ? 1024 : b);
    }
    g.first[g.n] = nb;
    hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3(nb), dim3(256), 0, s, g);
    MMT_LAUNCH_CHECK();
  }
  return 0;
}

// the kernel launch of a group of <= WG_MAXG plane-fed jobs with the pixel ranges the caller chose (job.ksplit; slabs job.ws when > 1)
int launch_wgpl_group(const WgPlJob* jobs, int n, hipStream_t s) {
  if (n < 1 || n > WG_MAXG) return MMT_EINVAL;
  WgGroup g;
  g.n = n;
  int nb = 0;
  for (int i = 0; i < n; i++) {
    const WgPlJob& j = jobs[i];
    const mmt_conv_args* a = j.a;
    if (!a->x || !j.dy || !j.xpl || !j.dpl || !j.s_x || !j.s_dy || !j.dw || ((size_t)j.xpl & 15) || ((size_t)j.dpl & 15) ||
        (j.xpl_stride & 7) || (j.dpl_stride & 7) || j.ksplit < 1 || (j.ksplit > 1 && !j.ws))
      return MMT_EINVAL;
    WgP& p = g.it[i];
    p.x = a->x; p.dy = j.dy;
    p.xpl = (const unsigned short*)j.xpl; p.xpl_stride = j.xpl_stride;
    p.dpl = (const unsigned short*)j.dpl; p.dpl_stride = j.dpl_stride;
    p.s_x = j.s_x; p.s_dy = j.s_dy;
    p.guard_x = (const float*)a->f16_guard_x; p.guard_dy = (const float*)a->f16_guard_dy;
    p.rowscale = j.rowscale; p.dw = j.dw; p.ws = j.ws; p.dbias = j.dbias;
    p.dbg = 0; p.lag = a->x_planes_lag;
    p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW; p.pad = a->pad; p.ksplit = j.ksplit;
    g.first[i] = nb;
    const int blocks = (a->Cout >> 7) * ((a->KH * a->KW * a->Cin) >> 7) * j.ksplit;
    nb += (blocks + 7) & ~7;
  }
  g.first[n] = nb;
  constexpr size_t lds = (size_t)WG_S * WG_STAGE;
  static bool done = false;
  if (!done) {
    const hipError_t e = hipFuncSetAttribute((const void*)wgrad_pl_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(wgrad_pl_group_kernel, dim3(nb), dim3(768), lds, s, g);
  MMT_LAUNCH_CHECK();
  return 0;
}
}  // namespace mmtconv

extern "C" int mmt_conv_wgrad_planes_splits(const mmt_conv_args* a) { return a ? wgpl_splits(a) : 0; }

extern "C" int mmt_conv_wgrad_planes(const mmt_conv_args* a, const float* dy, const void* x_planes, long x_plane_stride,
                                     const void* dy_planes, long dy_plane_stride, const float* s_x, const float* s_dy,
                                     const float* rowscale, float* dw, float* dbias, float* workspace, void* stream) {
  if (!a) return MMT_EINVAL;
  const int ks = wgpl_splits(a);
  if (ks == 0) return 1;   // not taken: the caller runs mmt_conv_wgrad
  if (!a->x || !dy || !x_planes || !dy_planes || !s_x || !s_dy || !dw || ((size_t)x_planes & 15) || ((size_t)dy_planes & 15) ||
      (x_plane_stride & 7) || (dy_plane_stride & 7) || (ks > 1 && !workspace))
    return MMT_EINVAL;
  WgP p;
  p.x = a->x; p.dy = dy;
  p.xpl = (const unsigned short*)x_planes; p.xpl_stride = x_plane_stride;
  p.dpl = (const unsigned short*)dy_planes; p.dpl_stride = dy_plane_stride;
  p.s_x = s_x; p.s_dy = s_dy;
  p.guard_x = (const float*)a->f16_guard_x; p.guard_dy = (const float*)a->f16_guard_dy;
  p.rowscale = rowscale; p.dw = dw; p.ws = workspace; p.dbias = dbias;
#ifdef MMT_PG_ABLATE
  p.dbg = getenv("MMT_WGPL_DBG") ? atoi(getenv("MMT_WGPL_DBG")) : 0;
#else
  p.dbg = 0;
#endif
  p.lag = a->x_planes_lag;
  p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW; p.pad = a->pad; p.ksplit = ks;
  const int NP = a->KH * a->KW * a->Cin;
  const int tiles = (a->Cout >> 7) * (NP >> 7);
  constexpr size_t lds = (size_t)WG_S * WG_STAGE;
  static_assert(lds <= 160 * 1024 && lds >= 16 * 256 * 16, "LDS");
  static bool done = false;
  if (!done) {
    const hipError_t e = hipFuncSetAttribute((const void*)wgrad_pl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(wgrad_pl_kernel, dim3(tiles * ks), dim3(768), lds, s, p);
  MMT_LAUNCH_CHECK();
  if (ks > 1) {
    const long n4 = (long)a->Cout * NP / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_pl_reduce_kernel, dim3(blocks), dim3(256), 0, s, workspace, ks, a->Cout, NP, rowscale, dw);
    MMT_LAUNCH_CHECK();
  }
  return 0;
}
