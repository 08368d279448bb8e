// Plane-fed implicit-GEMM convolution for gfx950 (round 5): the general kernel of the default arithmetic (two-term fp16 split, 3
// matrix products per multiply) for the shapes conv3x3_strip_kernel does not take -- 3x3 on small maps (layer3 / layer4 conv2, the
// small pyramid levels, the mask head's 14 x 14 maps), strided and 1x1 layers with long K, the fc layers; forward and data gradient.
//
// Replaces, like conv_igemm.hip, the ATen / cuDNN calls behind maskrcnn_benchmark.layers.Conv2d / nn.Linear (layers/misc.py:30-43,
// backbone/resnet.py:254-274, backbone/fpn.py:43-69, rpn/rpn.py:39-46, roi_heads/mask_head/roi_mask_feature_extractors.py:131-146,
// box_head/roi_box_feature_extractors.py:97-98) with the FrozenBN scale/shift (layers/batch_norm.py:19-24), ReLU, residual add and
// the ReLU mask of a data gradient fused into the epilogue.
//
// Why a new kernel (profiles/r04_pmc_tiled_vs_strip.txt): conv_fwd_glds_kernel spends 4.6 vector + 5 scalar instructions
// per MFMA splitting x in registers and computing copy addresses, reads every B fragment in all four waves (4 x 1 wave grid) and
// leaves its tile through an LDS-staged epilogue whose residual / mask loads the compiler waits for one by one: MFMA-busy 0.36.
// conv3x3_strip_kernel showed what works on this part (busy 0.67): both operands as pre-split fp16 planes copied global -> LDS by
// buffer_load ... lds with per-lane offsets that are fixed for a whole tap and ONE scalar offset per step, 64 x 64 wave tiles, no
// vector arithmetic in the loop.  This kernel is that recipe for any (KH, KW, stride, pad):
//   * GEMM view  C[m][n] = sum_k A[m][k] B[n][k],  m = (img, ho, wo), n = cout, k = (kh, kw, ci); K order = the packed weight
//     planes' (mmt_pack_weight_f16 / _flipped_f16: [K/16][Cout/32][32][2][8] fp16 = the LDS image, 1 KiB per copy instruction);
//   * A = the two fp16 planes of x * s_x with x's own NHWC indexing (mmt_split_planes_f16): a copy instruction fills one 32-row
//     block of one plane for one 16-channel step (lane = (row, 16-byte half); the half-swizzle half ^= (row >> 3) & 1 is applied on
//     the source channel offset and again on the fragment read).  The per-lane offset (pixel of the row for the current tap, or
//     out of range = zeros for halo / rows past M) is recomputed when the filled step enters a new tap -- Cin / 16 steps apart;
//   * a block = 8 matrix waves + 4 copy waves (768 threads, one block per CU).  The matrix waves are KG groups of WM x 2 waves of
//     64 x 64: (WM, KG) = (4, 1): one 256 x 128 tile; (2, 2): a 128 x 128 tile over two K ranges side by side; (1, 4): a 64 x 128
//     tile over four -- split-K INSIDE the block, summed through LDS in a fixed order, so a few-tile shape still has two waves on
//     every SIMD.  The copy waves issue every buffer_load ... lds (no MFMA wave computes an address or waits for vmcnt); ring of S
//     stages, one raw s_barrier per 16-k step, counted vmcnt in the copy waves: the copies of step t + S go out while the matrix
//     waves are in step t, the fragments of step t + 1 are read behind the MFMAs of step t (8 ds_read_b128 per 12 MFMAs per wave).
//     What the loop waits for (tools/bench_pg.py --ablate, profiles/r05_pg_ablate.txt): copy time ADDS to fragment-read time on
//     the LDS port whoever issues it (about 100 GB/s per CU of DMA writes) -- the bound of every plane-fed loop on this part;
//   * epilogue straight from the accumulator registers (an accumulator register = one output row x 32 consecutive channels per
//     half wave = complete 128-byte lines): no LDS staging, every residual / mask value of a 32 x 32 sub-tile requested before the
//     previous sub-tile is finished, unconditional buffer operations (absent operand = zero-sized buffer, row past M / column past
//     Cout = offset beyond the buffer);
//   * split-K for few-tile shapes in ONE launch: every block parks its accumulators in the per-stream workspace in register
//     order (coalesced 16-byte stores), releases, draws a ticket; the last arriver of a tile acquires, adds the partial tiles in the
//     fixed order 0 .. ksplit - 1 and runs the epilogue (cdna_hip_programming.md section 5: in-launch split-K reduction).
// Arithmetic, product order (h l, l h, h h per 16-k step) and split-K ranges are those of conv_fwd_glds_kernel<.., 2, 3, true>:
// results are bit-identical to it for an equal number of K ranges (tests/test_pgemm_gpu.py).
//
// Roofline: MFMA, 2500 / 3 = 833 TFLOP/s algorithmic (3 products per multiply); LDS port per 16-k step and CU at WM = 4: 64 KB of
// fragment reads + 24 KB of copies = 704 of the 768 cycles its 96 MFMAs take.
#include <map>
#include <mutex>
#include "conv_shared.h"

namespace {

constexpr unsigned PG_OOB = 0x80000000u;

template <int I, int U, class F>
__device__ __forceinline__ void pg_unroll(F&& f) {
  if constexpr (I < U) {
    f(std::integral_constant<int, I>{});
    pg_unroll<I + 1, U>(f);
  }
}

// WM x 2 waves of 64 x 64 form a GROUP that owns the (64 WM) x 128 tile over one K range; KG groups (8 waves in all: (WM, KG) = (4, 1),
// (2, 2), (1, 4)) work on KG consecutive K ranges of the same tile side by side and add their accumulators through LDS at the end --
// split-K INSIDE the block: a few-tile shape keeps two waves on every SIMD and its tile count is 256 / (64 WM) per 256 rows without a
// trip through memory.  ksplit > 1 adds K ranges across blocks (R = ksplit * KG ranges in all, range r = ks * KG + group).
// AF: the A operand comes from the fp32 tensor itself (no plane-split pass in front of the launch): the copy waves fetch the raw rows
// into registers, split them into the two fp16 terms of x * s_x there -- vector work on waves that have nothing else to do -- and
// store the same LDS image; p.f16_sx then points to max |x| (p.f16_ax)
template <int WM, int KG, int S, int DBG = 0, bool AF = false>   // DBG (tools/bench_pg.py --ablate): 1 A copies re-read one step, 2 B copies, 4 no copies, 8 no fragment reads
__global__ __launch_bounds__(768) void conv_pg_kernel(const ConvP p, const int ksplit, float* __restrict__ ws,
                                                      unsigned* __restrict__ tickets) {
  constexpr int BM = 64 * WM, BN = 128, GW = 2 * WM, NT = 512, GT = 64 * GW, NTALL = 768;
  static_assert(GW * KG == 8, "eight matrix waves");
  constexpr int PA = BM * 32, PB = BN * 32;            // bytes of one plane of the A / B tile of a 16-k step
  constexpr int A_BYTES = 2 * PA, GSTAGE = A_BYTES + 2 * PB, STAGE = KG * GSTAGE;
  // Waves 0 .. 7 multiply; waves 8 .. 11 (one per SIMD) do nothing but copy.  An LDS-DMA instruction occupies its wave's issue
  // port for 60 - 185 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), during which the MFMAs queued behind it wait:
  // issued by the matrix waves the copies' time ADDED to the loop's (profiles/r05_pg_ablate.txt: 289 us with, 195 us without).
  // Copy wave c serves group c / PPG: two 32-row blocks of A (both planes) and NBP of the group's eight B items per step.
  constexpr int PPG = 4 / KG, NBP = 8 / PPG, NI = 4 + NBP;   // copy waves per group; B items / all copy instructions per copy wave and step
  constexpr int U = (S % 2) ? 2 * S : S;               // steps per loop body: stages rotate mod S, register sets mod 2
  constexpr int OWN = 4 / KG;                          // 32 x 32 sub-tiles of a wave's 64 x 64 that its group finishes
  static_assert(S >= 3 && (S - 2) * NI < 64, "ring depth / vmcnt range");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ring = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool copier = wave >= 8;
  const int cw = wave - 8;                             // copy wave index
  const int grp = copier ? cw / PPG : wave / GW;       // K group this wave belongs to / serves
  const int gwave = wave % GW;                         // matrix waves: wave inside the group
  const int wm = gwave >> 1, wn = gwave & 1;
  const int gtid = tid - grp * GT;

  const int tiles_n = (p.Cout + BN - 1) / BN;
  int bid = blockIdx.x;
  {  // XCD-aware order: an XCD owns a contiguous range of (tile, K range) units -- the rows of A cross the fabric once
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ks = bid % ksplit;
  bid /= ksplit;
  const int tile_lin = bid;
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int HoWo = p.Ho * p.Wo;
  const F16Guard guard = f16_guard_load(p.guard_x);   // issued here, tested behind the prologue's copies
  const float s_lag = p.xpl_lag ? *p.f16_sx : 0.f;    // (round 6) planes written by the producer with a scale fixed beforehand
  auto guard_bad = [&]() { return p.xpl_lag ? f16_guard_bad_lag(guard, s_lag) : f16_guard_bad(guard); };

  // ---- K range of this wave's group
  const int nkt_all = p.K >> 4, spt = p.Cin >> 4;
  const int R = ksplit * KG, rng = ks * KG + grp;
  const int kt0 = (int)((long)rng * nkt_all / R), nkt = (int)((long)(rng + 1) * nkt_all / R) - kt0;
  const int nkt_max = (nkt_all + R - 1) / R;           // steps of the longest range of the launch: every wave runs that many barriers
  char* const gring = ring + grp * GSTAGE;             // the group's part of stage 0

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  if (copier) {
    // ================================================================ copy wave
    const int sub = cw % PPG;                          // which part of the group's items
    // A: row blocks 2 sub, 2 sub + 1 of the group's tile (32 rows each), both planes; lane = (row r, physical 16-byte half)
    // (DBG & 32 / & 64, timing only: 4 / 8 lanes share a row -- 64- / 128-byte contiguous pieces instead of 32-byte ones)
    const int ar = (DBG & 64) ? lane >> 3 : (DBG & 32) ? lane >> 2 : lane >> 1;
    const int alh = (DBG & 64) ? (lane & 7) : (DBG & 32) ? (lane & 3) : (lane & 1) ^ ((ar >> 3) & 1);      // logical 8-channel half this lane fetches
    int aih0[2], aiw0[2];
    unsigned apix[2];                                  // pixel index of the image's first pixel
    bool aok[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int m = m0 + (2 * sub + t) * 32 + ar;
      aok[t] = m < p.M;
      const int mm = aok[t] ? m : 0;
      const int img = mm / HoWo, rem = mm - img * HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      aih0[t] = ho * p.stride - p.pad;
      aiw0[t] = wo * p.stride - p.pad;
      apix[t] = (unsigned)img * (unsigned)(p.H * p.W);
    }
    const long n_x = (long)p.N * p.H * p.W * p.Cin;
    const __amdgpu_buffer_rsrc_t rs_a0 = AF ? __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(n_x * 4), 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)p.xpl, 0, (int)(n_x * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a1 = AF ? rs_a0
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)(p.xpl + p.xpl_stride), 0, (int)(n_x * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpl, 0, 0x7ffffff0, 0x00020000);
    // B: items NBP sub .. NBP sub + NBP - 1 of the group's eight: item t -> plane t / 4, 32-channel block t % 4 of the tile
    const int nb32 = (p.Cout + 31) >> 5;
    const int b_step = nb32 * 1024;                    // bytes per 16-k step of a packed plane
    unsigned vo_b[NBP];
    int dst_b[NBP];
#pragma unroll
    for (int i = 0; i < NBP; i++) {
      const int t = NBP * sub + i, q = t >> 2, cb = t & 3;
      int nb = n0 / 32 + cb;
      if (nb >= nb32) nb = nb32 - 1;                   // tile hanging over Cout: any valid block (those columns are never stored)
      vo_b[i] = (unsigned)((q * p.wpl_stride + (long)nb * 512 + lane * 8) * 2);
      dst_b[i] = A_BYTES + q * PB + cb * 1024;
    }
    // fill state: (tap, 16-channel step inside the tap) of the NEXT step to copy
    int f_kh, f_kw, f_ci, f_n = 0;                     // f_n: steps issued so far
    {
      const int tap0 = kt0 / spt;
      f_ci = kt0 - tap0 * spt;
      f_kh = tap0 / p.KW;
      f_kw = tap0 - f_kh * p.KW;
    }
    // bytes from one 16-channel step to the next inside a tap: 32 with planes indexed like x, one image row of the block (W x 32)
    // with row-blocked planes [N H][Cin / 16][W][16]
    const int a_step = AF ? 64 : (p.xpl_rb ? p.W * 32 : 32);   // (AF: 16 fp32 channels)
    int soff_a = f_ci * a_step, soff_b = kt0 * b_step;
    unsigned vo_a[2] = {PG_OOB, PG_OOB};
    auto enter_tap = [&]() {   // per-lane offset of (row's pixel for tap (f_kh, f_kw), this lane's 8 channels); halo / past M: zeros
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int ih = aih0[t] + f_kh, iw = aiw0[t] + f_kw;
        const bool ok = aok[t] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        vo_a[t] = !ok ? PG_OOB
                  : AF ? ((apix[t] + (unsigned)(ih * p.W + iw)) * (unsigned)p.Cin + (unsigned)(alh * 8)) * 4u
                  : p.xpl_rb ? (((apix[t] + (unsigned)(ih * p.W)) * (unsigned)spt + (unsigned)iw) * 16u + (unsigned)(alh * 8)) * 2u
                             : ((apix[t] + (unsigned)(ih * p.W + iw)) * (unsigned)p.Cin + (unsigned)(alh * 8)) * 2u;
      }
    };
    enter_tap();
    auto advance = [&]() {
      f_n++;
      if constexpr (DBG & 1) { soff_b += (DBG & 2) ? 0 : b_step; return; }
      soff_a += a_step;
      soff_b += (DBG & 2) ? 0 : b_step;
      if (++f_ci == spt) {
        f_ci = 0;
        soff_a = 0;
        if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
        enter_tap();
      }
    };
    // the copies of one step into `stage`; steps past the end of the range copy zeros (every lane out of range): the counted waits
    // stay uniform
    auto copy_step = [&](int stage) {
      const bool real = (DBG & 4) ? false : f_n < nkt;
      char* const st = gring + stage * STAGE;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int va = (int)(real ? vo_a[t] : PG_OOB);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, (lds_ptr_t)(st + (2 * sub + t) * 1024), 16, va, soff_a, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (lds_ptr_t)(st + PA + (2 * sub + t) * 1024), 16, va, soff_a, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NBP; i++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(st + dst_b[i]), 16, (int)(real ? vo_b[i] : PG_OOB), soff_b, 0, 0);
      advance();
    };
    auto wait_copies = [&]() {   // the newest S - 2 steps of this wave's copies may stay pending
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * NI) : "memory");
    };
    if constexpr (AF || (DBG & 16)) {
      // ---- register-staged copies (experiment): the same LDS image, filled by buffer_load_b128 into a ring of S register sets and
      // ds_write_b128 one step later instead of by LDS-DMA.  pump(n): store step n (set n % S) into stage n % S, then request step
      // n + S into the freed set
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 rg[S][NI];
      const float s_a = AF ? f16_scale_of(*p.f16_sx) : 1.f;
      auto load_step = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
        const bool real = f_n < nkt;
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int va = (int)(real ? vo_a[t] : PG_OOB);
          rg[set][2 * t] = __builtin_amdgcn_raw_buffer_load_b128(rs_a0, va, soff_a, 0);
          rg[set][2 * t + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs_a1, AF ? va + 16 : va, soff_a, 0);   // (AF: the lane's channels 4 .. 7)
        }
#pragma unroll
        for (int i = 0; i < NBP; i++) rg[set][4 + i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)(real ? vo_b[i] : PG_OOB), soff_b, 0);
        advance();
      };
      auto store_step = [&](auto setc, int stage) {
        constexpr int set = decltype(setc)::value;
        char* const st = gring + stage * STAGE + lane * 16;
#pragma unroll
        for (int t = 0; t < 2; t++) {
          if constexpr (AF) {   // eight fp32 channels of the lane's row -> their (h, l) terms: the split of mmt_split_planes_f16, here
            uint2 o0[2], o1[2];
            split4h(__builtin_bit_cast(f32x4, rg[set][2 * t]), s_a, o0);
            split4h(__builtin_bit_cast(f32x4, rg[set][2 * t + 1]), s_a, o1);
            *(u32x4*)(st + (2 * sub + t) * 1024) = u32x4{o0[0].x, o0[0].y, o1[0].x, o1[0].y};
            *(u32x4*)(st + PA + (2 * sub + t) * 1024) = u32x4{o0[1].x, o0[1].y, o1[1].x, o1[1].y};
          } else {
            *(u32x4*)(st + (2 * sub + t) * 1024) = rg[set][2 * t];
            *(u32x4*)(st + PA + (2 * sub + t) * 1024) = rg[set][2 * t + 1];
          }
        }
#pragma unroll
        for (int i = 0; i < NBP; i++) *(u32x4*)(st + dst_b[i]) = rg[set][4 + i];
      };
      auto pump = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * NI) : "memory");   // the set's loads (S slots old) have arrived
        store_step(setc, set);
        load_step(setc);
      };
      pg_unroll<0, S>([&](auto ic) { load_step(ic); });
      if (!guard_bad()) {
        pg_unroll<0, S - 1>([&](auto ic) { pump(ic); });   // steps 0 .. S - 2 into stages 0 .. S - 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // (0) step 0 has landed
        pump(std::integral_constant<int, S - 1>{});
        for (int kt = 0; kt < nkt_max; kt += S) {
          pg_unroll<0, S>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (i == 0 || kt + i < nkt_max) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stores of the previous slot are in LDS
              __builtin_amdgcn_s_barrier();
              pump(ic);
            }
          });
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
#pragma unroll
    for (int t = 0; t < S - 1; t++) copy_step(t);      // prologue: steps 0 .. S - 2 into stages 0 .. S - 2
    if (!guard_bad()) {
      wait_copies();
      __builtin_amdgcn_s_barrier();                    // (0) step 0 has landed
      copy_step(S - 1);
      for (int kt = 0; kt < nkt_max; kt += S) {
        pg_unroll<0, S>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if (i == 0 || kt + i < nkt_max) {
            // this wave's copies of step kt + i + 1 have landed (those of the S - 2 steps behind it may stay in flight); behind the
            // barrier every matrix wave has finished reading stage i, which is filled with step kt + i + S
            wait_copies();
            __builtin_amdgcn_s_barrier();
            copy_step(i);
          }
        });
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero copies of the tail steps land before LDS is reused
    }
  } else {
    // ================================================================ matrix wave
    // fragment reads: row = lane & 31 of a 32-row block, 16-byte half (lane >> 5) ^ ((row >> 3) & 1)
    const int lr = lane & 31, kh2 = lane >> 5;
    const int foff = lr * 32 + (((kh2 ^ (lr >> 3)) & 1) << 4);
    const char* const fA = gring + (wm * 2) * 1024 + foff;
    const char* const fB = gring + A_BYTES + (wn * 2) * 1024 + foff;
    // i-th fragment read of a step, in the order the MFMAs of that step consume them: products (h, l), (l, h), (h, h)
    auto fread = [&](int i, int stage, f16x8 (&fa)[2][2], f16x8 (&fb)[2][2]) {
      const char* const a = fA + stage * STAGE;
      const char* const b = fB + stage * STAGE;
      switch (i) {
        case 0: fa[0][0] = *(const f16x8*)(a); break;
        case 1: fb[1][0] = *(const f16x8*)(b + PB); break;
        case 2: fb[1][1] = *(const f16x8*)(b + PB + 1024); break;
        case 3: fa[0][1] = *(const f16x8*)(a + 1024); break;
        case 4: fa[1][0] = *(const f16x8*)(a + PA); break;
        case 5: fb[0][0] = *(const f16x8*)(b); break;
        case 6: fb[0][1] = *(const f16x8*)(b + 1024); break;
        default: fa[1][1] = *(const f16x8*)(a + PA + 1024); break;
      }
    };
    // one 16-k step: 12 MFMAs on (fa, fb); behind the first eight the fragment reads of step kt + 1 (stage st_next, into fan / fbn)
    auto step = [&](int kt, int st_next, const f16x8 (&fa)[2][2], const f16x8 (&fb)[2][2], f16x8 (&fan)[2][2], f16x8 (&fbn)[2][2]) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of step kt are complete ...
      __builtin_amdgcn_s_barrier();                        // ... everybody's are, and step kt + 1 has landed
      if (kt < nkt) {
        int j = 0;
#pragma unroll
        for (int pr = 0; pr < 3; pr++) {
          const int qa = pr == 1 ? 1 : 0, qb = pr == 0 ? 1 : 0;
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[qa][a], fb[qb][b], acc[a][b], 0, 0, 0);
              if (j < 8) { if constexpr (!(DBG & 8)) fread(j, st_next, fan, fbn); }
              j++;
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
    };
    if (!guard_bad()) {
      __builtin_amdgcn_s_barrier();                    // (0) step 0 has landed
      f16x8 faP[2][2], fbP[2][2], faQ[2][2], fbQ[2][2];
#pragma unroll
      for (int i = 0; i < 8; i++) fread(i, 0, faP, fbP);
      if constexpr (DBG & 8) {
#pragma unroll
        for (int i = 0; i < 8; i++) fread(i, 1, faQ, fbQ);
      }
      for (int kt = 0; kt < nkt_max; kt += U) {
        pg_unroll<0, U>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if (i == 0 || kt + i < nkt_max) {
            if constexpr (i % 2 == 0) step(kt + i, (i + 1) % S, faP, fbP, faQ, fbQ);
            else step(kt + i, (i + 1) % S, faQ, fbQ, faP, fbP);
          }
        });
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (guard_bad()) {   // this tensor's dynamic range defeats fp16 (uniform over the grid): exact fp32 products, whole K, by
    if (ks == 0) conv_slow_tile(m0, BM, 0, BM, n0, BN, nullptr, false, tid, NTALL, blockIdx.x);   // the tile's first block
    return;
  }
  __syncthreads();

  // ---- the KG groups' accumulators meet in LDS: sub-tile i = (a, b) of every wave is finished by group i % KG, which adds the
  // groups' partial sums in range order (its own from registers).  Exchange element: 16 bytes at ((i KG + src) 4 + r4) GT + gtid.
  if constexpr (KG > 1) {
    f32x4* const xch = (f32x4*)ring;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (!copier && i % KG != grp) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++)
          xch[((i * KG + grp) * 4 + r4) * GT + gtid] = f32x4{acc[i >> 1][i & 1][4 * r4], acc[i >> 1][i & 1][4 * r4 + 1],
                                                             acc[i >> 1][i & 1][4 * r4 + 2], acc[i >> 1][i & 1][4 * r4 + 3]};
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (!copier && i % KG == grp) {
        f32x16 sum;
#pragma unroll
        for (int src = 0; src < KG; src++) {
          f32x16 t;
          if (src == grp) {
            t = acc[i >> 1][i & 1];
          } else {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
              const f32x4 v = xch[((i * KG + src) * 4 + r4) * GT + gtid];
              t[4 * r4] = v[0]; t[4 * r4 + 1] = v[1]; t[4 * r4 + 2] = v[2]; t[4 * r4 + 3] = v[3];
            }
          }
          if (src == 0) sum = t; else sum += t;
        }
        acc[i >> 1][i & 1] = sum;
      }
    }
    __syncthreads();   // (the exchange area is reused below)
  }

  // ---- split-K across blocks: every block parks the sub-tiles it finished (register order: 16 bytes at (o 4 + r4) NT + tid,
  // write-through stores), draws a ticket; the last arriver of the tile adds the ksplit partial tiles in the order 0 .. ksplit - 1
  // (cdna_hip_programming.md section 5, "in-launch split-K reduction", the sc1 form: no cache write-back, no fence)
  if (ksplit > 1) {
    constexpr int SLAB16 = OWN * 4 * NT;   // 16-byte elements per partial tile
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void*)ws, 0, (int)SPLITK_WS_BYTES, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned tile_off = (unsigned)tile_lin * (unsigned)ksplit * (unsigned)(SLAB16 * 16);
    if (!copier) {
      const unsigned slab_off = tile_off + (unsigned)ks * (unsigned)(SLAB16 * 16) + (unsigned)tid * 16u;
      int o = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (i % KG == grp) {
#pragma unroll
          for (int r4 = 0; r4 < 4; r4++) {
            const f32x4 v = {acc[i >> 1][i & 1][4 * r4], acc[i >> 1][i & 1][4 * r4 + 1], acc[i >> 1][i & 1][4 * r4 + 2],
                             acc[i >> 1][i & 1][4 * r4 + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rws, (int)(slab_off + (unsigned)((o * 4 + r4) * NT * 16)), 0, 16);
          }
          o++;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* const flag = (unsigned*)ring;
    if (tid == 0) *flag = __hip_atomic_fetch_add(tickets + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != (unsigned)(ksplit - 1)) return;
    if (tid == 0) __hip_atomic_store(tickets + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch
    if (!copier) {   // (copy waves stay with the block to its last barrier)
      // the partial tiles are added straight into the accumulator registers (this block's own partial is in its slab like everybody
      // else's, so the order is 0 .. ksplit - 1 whoever arrives last), four 16-byte loads in flight per owned sub-tile: a separate
      // sum[OWN] beside acc[2][2] and sixteen loads in flight was 192 live registers of the 168 a 768-thread block has -- the
      // 16 spilled registers / 107 scratch instructions VERDICT r5 found in <4, 1, 4>
      for (int j = 0; j < ksplit; j++) {
        const unsigned slab_off = tile_off + (unsigned)j * (unsigned)(SLAB16 * 16) + (unsigned)tid * 16u;
        int o = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (i % KG == grp) {
            u32x4 t[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) t[r4] = __builtin_amdgcn_raw_buffer_load_b128(rws, (int)(slab_off + (unsigned)((o * 4 + r4) * NT * 16)), 0, 16);
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
              const f32x4 v = __builtin_bit_cast(f32x4, t[r4]);
#pragma unroll
              for (int c = 0; c < 4; c++) {
                if (j == 0) acc[i >> 1][i & 1][4 * r4 + c] = v[c]; else acc[i >> 1][i & 1][4 * r4 + c] += v[c];
              }
            }
            o++;
          }
        }
      }
    }
  }

  // ---- epilogue from the accumulator registers (conv_shared.h): each group finishes its own sub-tiles
  {
    const int mrow0 = m0 + wm * 64 + 4 * (lane >> 5);
    conv_epilogue_direct<2, 2>(p, acc, (unsigned)mrow0 * ((unsigned)p.Cout * 4u), p.M - mrow0, n0 + wn * 64 + (lane & 31),
                         copier ? 0 : (KG == 1 ? 15 : (KG == 2 ? (grp ? 10 : 5) : (1 << grp))), (float*)ring + 64, wave, lane, tid, blockIdx.x);
  }
}

// x (NHWC fp32, R = N H image rows of W pixels, C channels) -> two fp16 planes of x * s in the ROW-BLOCKED order
// [R][C / 16][W][16]: what a copy instruction of conv_pg_kernel wants for 32 consecutive pixels of an image row and one 16-channel
// step is then ONE run of 1 KiB instead of 32 pieces of 32 bytes in 32 cache lines (the vector memory path handles a 64-byte
// sector per clock whatever part of it is asked for: profiles/r05_pg_ablate.txt, "A pieces of 64 / 128 B").  One block = 32 pixels
// of a row x up to 256 channels: coalesced float4 reads, transposed through LDS, written as runs of (pixels x 32) bytes.
__global__ __launch_bounds__(256) void split_planes_f16_rb_kernel(const float* __restrict__ x, unsigned short* __restrict__ pl,
                                                                  const long plane_stride, const int W, const int C, const int segs,
                                                                  const float* __restrict__ amax, float* __restrict__ s_out) {
  constexpr int CBS = 1024 + 32;   // bytes of one 16-channel block of the tile in LDS (32 pixels x 32 bytes, padded)
  __shared__ __attribute__((aligned(16))) char tile[2 * 16 * CBS];
  const float s = f16_scale_of(*amax);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && s_out) *s_out = s;
  const int row = blockIdx.x / segs, w0 = (blockIdx.x % segs) * 32, c0 = blockIdx.y * 256;
  const int cc = min(256, C - c0), c4n = cc >> 2, ncb = cc >> 4, npx = min(32, W - w0);
  const float* const src = x + ((long)row * W + w0) * C + c0;
  const int total = npx * c4n;   // <= 2048 float4: eight per thread, all requested before the first is used
  f32x4 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int i = threadIdx.x + 256 * u;
    const int px = i / c4n, c4 = i - px * c4n;
    v[u] = i < total ? ldg4(src + (long)px * C + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int i = threadIdx.x + 256 * u;
    if (i < total) {
      const int px = i / c4n, c4 = i - px * c4n;
      uint2 o[2];
      split4h(v[u], s, o);
#pragma unroll
      for (int q = 0; q < 2; q++) *(uint2*)(tile + (q * 16 + (c4 >> 2)) * CBS + px * 32 + (c4 & 3) * 8) = o[q];
    }
  }
  __syncthreads();
  const int per_cb = npx * 2;   // 16-byte pieces of one (plane, block) run
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int j = threadIdx.x + 256 * u;
    if (j < 2 * ncb * per_cb) {
      const int q = j / (ncb * per_cb), r = j - q * (ncb * per_cb), cb = r / per_cb, k = r - cb * per_cb;
      const uint4 t = *(const uint4*)(tile + (q * 16 + cb) * CBS + k * 16);
      *(uint4*)(pl + q * plane_stride + (((long)row * (C >> 4) + (c0 >> 4) + cb) * W + w0) * 16 + k * 8) = t;
    }
  }
}

// the same re-ordering for small tensors (a few MB: L2-resident, the pass is two memory round trips long whatever it does): every
// thread splits four channels of a pixel and stores its two 8-byte pieces straight to their row-blocked places -- no LDS, no barrier
__global__ __launch_bounds__(256) void split_planes_f16_rb_small_kernel(const float* __restrict__ x, unsigned short* __restrict__ pl,
                                                                        const long plane_stride, const int W, const int C, const long n4,
                                                                        const float* __restrict__ amax, float* __restrict__ s_out) {
  const float s = f16_scale_of(*amax);
  if (blockIdx.x == 0 && threadIdx.x == 0 && s_out) *s_out = s;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c4n = C >> 2;
  const long pix = i / c4n;
  const int c4 = (int)(i - pix * c4n);
  const long row = pix / W;
  const int w = (int)(pix - row * W);
  uint2 o[2];
  split4h(ldg4(x + i * 4), s, o);
  const long dst = ((row * (C >> 4) + (c4 >> 2)) * W + w) * 16 + (c4 & 3) * 4;
#pragma unroll
  for (int q = 0; q < 2; q++) *(uint2*)(pl + q * plane_stride + dst) = o[q];
}

// ---- round 6: y = a + b (+ c) (+ d) (the gradient of a tensor with several consumers, mmt_sum_stats) that ALSO leaves y as row-blocked
// fp16 planes of y * *scale for the plane-fed data-gradient launch that consumes it: the split pass of that launch disappears.  Same
// block shape as split_planes_f16_rb_kernel (32 pixels of an image row x up to 256 channels, planes transposed through LDS); the
// statistics slot gets max |y| from every block and sum / count from every 16th (as sum_stats_kernel), `next` the same maximum.
__global__ __launch_bounds__(256) void sum_stats_rb_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                           const float* __restrict__ d, float* __restrict__ y, unsigned short* __restrict__ pl,
                                                           const long plane_stride, const int W, const int C, const int segs,
                                                           const float* __restrict__ scale, unsigned* __restrict__ out,
                                                           unsigned* __restrict__ next) {
  constexpr int CBS = 1024 + 32;
  __shared__ __attribute__((aligned(16))) char tile[2 * 16 * CBS];
  __shared__ float wred[3][4];
  const float s = *scale;
  const int row = blockIdx.x / segs, w0 = (blockIdx.x % segs) * 32, c0 = blockIdx.y * 256;
  const int cc = min(256, C - c0), c4n = cc >> 2, ncb = cc >> 4, npx = min(32, W - w0);
  const long base = ((long)row * W + w0) * C + c0;
  const int total = npx * c4n;
  f32x4 v[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int i = threadIdx.x + 256 * u;
    const int px = i / c4n, c4 = i - px * c4n;
    const long o = base + (long)px * C + c4 * 4;
    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < total) {
      v[u] = ldg4(a + o) + ldg4(b + o);
      if (c) v[u] += ldg4(c + o);
      if (d) v[u] += ldg4(d + o);
    }
  }
  float m = 0.f, sum = 0.f, cnt = 0.f;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int i = threadIdx.x + 256 * u;
    if (i < total) {
      const int px = i / c4n, c4 = i - px * c4n;
      *(f32x4*)(y + base + (long)px * C + c4 * 4) = v[u];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
      sum += (fabsf(v[u][0]) + fabsf(v[u][1])) + (fabsf(v[u][2]) + fabsf(v[u][3]));
      cnt += 4.f;
      if (pl) {
        uint2 o[2];
        split4h(v[u], s, o);
#pragma unroll
        for (int q = 0; q < 2; q++) *(uint2*)(tile + (q * 16 + (c4 >> 2)) * CBS + px * 32 + (c4 & 3) * 8) = o[q];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); sum += __shfl_xor(sum, o, 64); cnt += __shfl_xor(cnt, o, 64); }
  if ((threadIdx.x & 63) == 0) { wred[0][threadIdx.x >> 6] = m; wred[1][threadIdx.x >> 6] = sum; wred[2][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  const int per_cb = npx * 2;
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int j = threadIdx.x + 256 * u;
    if (pl && j < 2 * ncb * per_cb) {
      const int q = j / (ncb * per_cb), r = j - q * (ncb * per_cb), cb = r / per_cb, k = r - cb * per_cb;
      const uint4 t = *(const uint4*)(tile + (q * 16 + cb) * CBS + k * 16);
      *(uint4*)(pl + q * plane_stride + (((long)row * (C >> 4) + (c0 >> 4) + cb) * W + w0) * 16 + k * 8) = t;
    }
  }
  if (threadIdx.x == 0) {
    const float mm = fmaxf(fmaxf(wred[0][0], wred[0][1]), fmaxf(wred[0][2], wred[0][3]));
    const unsigned bits = __builtin_bit_cast(unsigned, mm);
    if (mm > 0.f && bits > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, bits);
    if (next && mm > 0.f && bits > __hip_atomic_load(next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(next, bits);
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if ((lin & 15) == 0) {
      const float sm = (wred[1][0] + wred[1][1]) + (wred[1][2] + wred[1][3]), cn = (wred[2][0] + wred[2][1]) + (wred[2][2] + wred[2][3]);
      if (cn > 0.f) {
        const int k = (lin >> 4) & 15;
        atomicAdd((float*)out + 1 + k, sm);
        atomicAdd((float*)out + 17 + k, cn);
      }
    }
  }
}

// the producing sites' scales for the NEXT step (one launch per step): state[i] = {scale, pending maximum (float bits)}; a site that
// recorded a maximum since the last call gets the power of two that puts 2 x that maximum into [2^13, 2^14) -- one binade of
// head-room over the largest value any call of the site produced -- and its pending maximum is cleared; sites nobody called keep theirs
__global__ void rb_scales_update_kernel(float* __restrict__ state, const int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float pend = state[2 * i + 1];
  if (pend > 0.f && pend < 3.0e38f) state[2 * i] = f16_scale_of(pend) * 0.5f;
  state[2 * i + 1] = 0.f;
}

// is this call one the plane-fed kernel takes?  (shape / epilogue form only; the caller checked planes and arithmetic)
bool pg_shape(const ConvP& p) {
  return p.xpl && p.wpl && !p.io && !p.ypl && !p.mul && p.out_stride == 1 && p.res_mode <= 1 && (p.Cin & 15) == 0 && p.Cout > 32 &&
         (long)p.N * p.H * p.W * p.Cin < (1L << 29) && (long)p.M * p.Cout * 4 < (1L << 31) && ((size_t)p.xpl & 15) == 0 &&
         (p.xpl_stride & 7) == 0 && ((size_t)p.wpl & 15) == 0 && (p.wpl_stride & 7) == 0;
}

// tile height (64 / 128 / 256 rows: 4 / 2 / 1 K groups inside the block) and K ranges across blocks.  Every block is 8 waves and
// takes a CU for itself: the tallest tile that still gives (nearly) every CU a block; K ranges across blocks only when even the
// 64-row tiles leave half the chip empty and K is long
void pg_plan(const ConvP& p, int& rows, int& ksplit) {
  const long tn = mmt_cdiv(p.Cout, 128);
  const long t256 = (long)mmt_cdiv(p.M, 256) * tn, t128 = (long)mmt_cdiv(p.M, 128) * tn, t64 = (long)mmt_cdiv(p.M, 64) * tn;
  const int nkt = p.K >> 4;
  rows = t256 >= 448 ? 256 : (t128 >= 224 ? 128 : 64);
  ksplit = 1;
  if (rows == 64 && t64 <= 128) {
    int ks = (int)(256 / t64);
    if (ks > nkt / 64) ks = nkt / 64;     // >= 16 steps per group and range
    if (ks > 8) ks = 8;
    if (ks >= 2) ksplit = ks;
  }
  const char* e = getenv("MMT_SPLITK");   // read per call: the schedule-equivalence test switches it
  if (e && atoi(e) == 0) ksplit = 1;
}

template <int WM, int KG, int S, int DBG = 0, bool AF = false>
int launch_pg(const ConvP& p, hipStream_t s, int ksplit) {
  constexpr int BM = 64 * WM;
  const int tiles = mmt_cdiv(p.M, BM) * mmt_cdiv(p.Cout, 128);
  SplitWs w{nullptr, nullptr};
  if (ksplit > 1) {
    w = split_workspace(s);
    if (!w.ws || !w.tickets || tiles > SPLITK_TICKETS || (size_t)tiles * ksplit * BM * 128 * 4 > SPLITK_WS_BYTES) return MMT_EINVAL;
  }
  constexpr size_t ring = (size_t)S * KG * (BM * 64 + 2 * 128 * 32), xch = KG > 1 ? (size_t)4 * (KG - 1) * 4 * (128 * WM) * 16 : 0;
  constexpr size_t lds = ring > xch ? ring : xch;
  static_assert(lds <= 160 * 1024, "LDS");
  auto kern = conv_pg_kernel<WM, KG, S, DBG, AF>;
  static bool done = false;   // per instantiation
  if (!done) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles * ksplit), dim3(768), lds, s, p, ksplit, w.ws, w.tickets);
  MMT_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int mmt_conv_pg_plan(const mmt_conv_args* a, int* tile_rows, int* ksplit) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  int rows = 0, ks = 0;
  if (precision() == 3 && p.wpl && p.M > 0 && p.Cout > 0) {
    const unsigned short* keep = p.xpl;
    if (!p.xpl) p.xpl = (const unsigned short*)16;   // shape question: the planes need not exist yet
    if (pg_shape(p)) pg_plan(p, rows, ks);
    p.xpl = keep;
  }
  if (tile_rows) *tile_rows = rows;
  if (ksplit) *ksplit = ks;
  return 0;
}

// does the library want this call on the plane-fed kernel (with a plane-split pass of x in front)?  Measured on the step's shapes
// (profiles/r05_bench_pg.txt): yes for the 3x3 (and larger) convolutions the tap-strip kernel does not take; no for 1x1 / fc, whose
// inputs are large against their work -- the split pass costs what the leaner loop returns
extern "C" int mmt_conv_pg_wanted(const mmt_conv_args* a) {
  ConvP p;
  if (fill(p, a)) return 0;
  if (precision() != 3 || !p.wpl || p.M <= 0 || p.Cout < 96 || p.KH * p.KW < 4) return 0;   // (Cout = 64: half of every 128-column tile would be zeros)
  const char* e = getenv("MMT_PG");   // read per call (A/B timing, parity tests)
  if (e && atoi(e) == 0) return 0;
  if (!p.xpl) p.xpl = (const unsigned short*)16;
  return pg_shape(p) ? 1 : 0;
}

extern "C" int mmt_split_planes_f16_rb(const float* x, void* planes, long plane_stride, int rows, int W, int C, const float* amax,
                                       float* scale_out, void* stream) {
  const long n = (long)rows * W * C;
  if (!x || !planes || !amax || rows < 0 || W < 1 || C < 16 || (C & 15) || plane_stride < n || (plane_stride & 7) || ((size_t)planes & 15) ||
      ((size_t)x & 15) || n >= (1L << 30))
    return MMT_EINVAL;
  if (n == 0) return 0;
  if (n <= (1L << 22)) {   // <= 16 MB of fp32: the one-trip form (7.5 -> ~4 us on the student's 64 x 64 maps)
    hipLaunchKernelGGL(split_planes_f16_rb_small_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short*)planes, plane_stride, W, C, n / 4, amax, scale_out);
    MMT_LAUNCH_CHECK();
    return 0;
  }
  const int segs = mmt_cdiv(W, 32);
  hipLaunchKernelGGL(split_planes_f16_rb_kernel, dim3(rows * segs, mmt_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)planes, plane_stride, W, C, segs, amax, scale_out);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_sum_stats_rb(const float* a, const float* b, const float* c, const float* d, float* y, int rows, int W, int C,
                                float* slot, void* planes, long plane_stride, const float* scale, float* amax_next, void* stream) {
  const long n = (long)rows * W * C;
  if (!a || !b || !y || !slot || !scale || (!c && d) || rows < 0 || W < 1 || C < 16 || (C & 15) || (planes && plane_stride < n) ||
      (plane_stride & 7) || ((size_t)planes & 15) || (((size_t)a | (size_t)b | (size_t)c | (size_t)d | (size_t)y) & 15) || n >= (1L << 30))
    return MMT_EINVAL;
  if (n == 0) return 0;
  const int segs = mmt_cdiv(W, 32);
  hipLaunchKernelGGL(sum_stats_rb_kernel, dim3(rows * segs, mmt_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, d, y,
                     (unsigned short*)planes, plane_stride, W, C, segs, scale, (unsigned*)slot, (unsigned*)amax_next);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_rb_scales_update(float* state, int n, void* stream) {
  if (!state || n < 0) return MMT_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(rb_scales_update_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, state, n);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_conv_forward_pg(const mmt_conv_args* a, const float* s_x, const float* s_w, int tile_rows, int ksplit, void* stream) {
  ConvP p;
  int e = fill(p, a);
  if (e) return e;
  // x_planes null: the A operand is read from x itself and split by the kernel's copy waves; s_x then points to max |x|
  const bool af = p.xpl == nullptr;
  if (af) {
    if (!p.x || ((size_t)p.x & 15)) return MMT_EINVAL;
    p.xpl = (const unsigned short*)p.x; p.xpl_stride = 0; p.xpl_rb = 0;   // (the shape test below wants an aligned pointer)
  }
  if (!p.y || !s_x || !s_w || precision() != 3 || !pg_shape(p)) return MMT_EINVAL;
  if (p.M == 0 || p.Cout == 0) return 0;
  p.f16_sx = s_x; p.f16_sw = s_w; p.f16_ax = af ? 1 : 0;
  int rows, ks;
  pg_plan(p, rows, ks);
  if (tile_rows == 64 || tile_rows == 128 || tile_rows == 256) rows = tile_rows; else if (tile_rows != 0) return MMT_EINVAL;
  if (ksplit > 0) ks = ksplit;
  if (ks < 1 || ks * (256 / rows) > (p.K >> 4)) return MMT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
#ifdef MMT_PG_ABLATE   // (tools only: `make ablate` -> libmmtpsm_ablate.so for tools/bench_pg.py --ablate; the product library carries no DBG arm)
  if (const char* d = af ? nullptr : getenv("MMT_PG_DBG")) {   // ablations of the main loop (wrong results; tools/bench_pg.py --ablate)
    const int dbg = atoi(d);
    if (dbg == 32 || dbg == 64) {
      if (rows == 256) return dbg == 32 ? launch_pg<4, 1, 4, 32>(p, s, ks) : launch_pg<4, 1, 4, 64>(p, s, ks);
      if (rows == 64) return dbg == 32 ? launch_pg<1, 4, 3, 32>(p, s, ks) : launch_pg<1, 4, 3, 64>(p, s, ks);
    }
    if (dbg == 16) {
      if (rows == 256) return launch_pg<4, 1, 4, 16>(p, s, ks);
      if (rows == 128) return launch_pg<2, 2, 4, 16>(p, s, ks);
      return launch_pg<1, 4, 3, 16>(p, s, ks);
    }
    if (rows == 64) switch (dbg) {
      case 1: return launch_pg<1, 4, 3, 1>(p, s, ks);
      case 2: return launch_pg<1, 4, 3, 2>(p, s, ks);
      case 3: return launch_pg<1, 4, 3, 3>(p, s, ks);
      case 4: return launch_pg<1, 4, 3, 4>(p, s, ks);
      case 8: return launch_pg<1, 4, 3, 8>(p, s, ks);
      case 12: return launch_pg<1, 4, 3, 12>(p, s, ks);
      default: break;
    }
    if (rows == 256) switch (dbg) {
      case 1: return launch_pg<4, 1, 4, 1>(p, s, ks);
      case 2: return launch_pg<4, 1, 4, 2>(p, s, ks);
      case 4: return launch_pg<4, 1, 4, 4>(p, s, ks);
      case 8: return launch_pg<4, 1, 4, 8>(p, s, ks);
      case 12: return launch_pg<4, 1, 4, 12>(p, s, ks);
      default: break;
    }
  }
#endif
  if (af) {
#ifdef MMT_PG_ABLATE   // (round-5 experiment, slower than planes + a split pass and 144 live registers of copy ring: tools build only)
    if (rows == 256) return launch_pg<4, 1, 4, 0, true>(p, s, ks);
    if (rows == 128) return launch_pg<2, 2, 4, 0, true>(p, s, ks);
    return launch_pg<1, 4, 3, 0, true>(p, s, ks);
#else
    return MMT_EINVAL;
#endif
  }
  if (rows == 256) return launch_pg<4, 1, 4>(p, s, ks);
  if (rows == 128) return launch_pg<2, 2, 4>(p, s, ks);
  return launch_pg<1, 4, 3>(p, s, ks);
}
