// Batched greedy NMS for gfx950 with the sweep on the device.
//
// Replaces cuda/nms.cu:23-131 of the reference (64x64 IoU bit-mask tiles, then a D2H copy of the whole
// mask and a serial sweep on the host, one call per (image, level)).  Here B independent, score-sorted
// segments are handled by one mask launch and one sweep launch, nothing returns to the host.
// Semantics are the reference CPU path's (cpu/nms_cpu.cpp:22,56-60): "+1" areas, suppress when
// IoU >= thr -- the CUDA file uses '>' (SURVEY.md D9).  Built with -ffp-contract=off so that the IoU
// arithmetic rounds exactly like the CPU code.
//
// Stage 1 (nms_mask_kernel): block (cb, rb, seg) of 64 threads; thread i owns row box rb*64+i and
// produces the 64-bit word of suppressions against column boxes cb*64..cb*64+63 (only j > i).
// Stage 2 (nms_sweep_kernel): one sweeping wave per segment (+ three staging waves).  Lane w owns word w of the
// "removed" bitmap (n <= 64*64 boxes).  Rows are staged through LDS 64 at a time; inside a 64-row chunk the serial
// keep/suppress recurrence runs on the diagonal word with v_readlane broadcasts, then the kept rows of
// the chunk are OR-ed into every lane's word with independent (pipelined) LDS reads.
#include "common.h"

__device__ __forceinline__ bool iou_ge(const float* a, const float* b, float thr) {
  // a, b: x1,y1,x2,y2 ; areas with +1 (nms_cpu.cpp:22)
  const float aa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  const float ab = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
  const float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
  const float w = fmaxf(0.f, xx2 - xx1 + 1), h = fmaxf(0.f, yy2 - yy1 + 1);
  const float inter = w * h;
  const float ovr = inter / (aa + ab - inter);
  return ovr >= thr;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes,
                                                      const int* __restrict__ seg_off, int max_n, int words,
                                                      float thr, unsigned long long* __restrict__ mask) {
  const int seg = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  const int s0 = seg_off[seg], n = seg_off[seg + 1] - s0;
  if (rb * 64 >= n || cb * 64 >= n || cb < rb) return;  // upper triangle only; lower words are never read
  __shared__ float cbx[64 * 4];
  const int t = threadIdx.x;
  const int cj = cb * 64 + t;
  if (cj < n) {
    const f32x4 v = *(const f32x4*)(boxes + (long)(s0 + cj) * 4);
    cbx[t * 4 + 0] = v[0]; cbx[t * 4 + 1] = v[1]; cbx[t * 4 + 2] = v[2]; cbx[t * 4 + 3] = v[3];
  }
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  float a[4];
  {
    const f32x4 v = *(const f32x4*)(boxes + (long)(s0 + i) * 4);
    a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
  }
  unsigned long long bits = 0;
  const int jn = min(64, n - cb * 64);
  const int jstart = (cb == rb) ? t + 1 : 0;
  for (int j = jstart; j < jn; j++)
    if (iou_ge(a, &cbx[j * 4], thr)) bits |= 1ULL << j;
  mask[((long)seg * max_n + i) * words + cb] = bits;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// OR of a 32-bit value over the 64 lanes (uniform result): inclusive OR-scan inside each row of 16 lanes with DPP
// row shifts, then the four row totals
__device__ __forceinline__ unsigned wave_or32(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
  return __builtin_amdgcn_readlane(v, 15) | __builtin_amdgcn_readlane(v, 31) | __builtin_amdgcn_readlane(v, 47) |
         __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
  return ((unsigned long long)wave_or32((unsigned)(v >> 32)) << 32) | wave_or32((unsigned)v);
}

__global__ __launch_bounds__(256) void nms_sweep_kernel(const int* __restrict__ seg_off, int max_n, int words,
                                                        const unsigned long long* __restrict__ mask,
                                                        int* __restrict__ keep, int* __restrict__ keep_cnt) {
  // wave 0 sweeps chunk c out of LDS buffer c & 1 while waves 1..3 stage the rows of chunk c + 1 into the other one:
  // the sweep is a latency chain on a single wave, so the global-load latency of the staging must not sit in it
  extern __shared__ __attribute__((aligned(16))) unsigned long long rows[];  // [2][64][words]
  const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = seg_off[seg + 1] - seg_off[seg];
  const unsigned long long* M = mask + (long)seg * max_n * words;
  int* kp = keep + (long)seg * max_n;
  unsigned long long removed = 0;  // wave 0, lane w: word w
  int cnt = 0;
  const int nchunks = (n + 63) / 64;
  // rows r0..r0+rn-1 of chunk c, words c..nchunks-1 (lower words are irrelevant from there on)
  auto stage = [&](int c, int t, int nt) {
    const int r0 = c * 64, rn = min(64, n - r0), wn = nchunks - c;
    unsigned long long* dst = rows + (c & 1) * 64 * words;
    const int tot = rn * wn;
    for (int base = t; base < tot; base += 8 * nt) {  // eight loads in flight per thread, then the LDS stores
      unsigned long long v[8];
      int at[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int idx = base + u * nt;
        const int rr = idx / wn, ww = c + idx - rr * wn;
        at[u] = idx < tot ? rr * words + ww : -1;
        v[u] = idx < tot ? M[(long)r0 * words + at[u]] : 0ULL;
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (at[u] >= 0) dst[at[u]] = v[u];
    }
  };
  if (nchunks > 0) stage(0, tid, 256);
  __syncthreads();
  for (int c = 0; c < nchunks; c++) {
    if (wave > 0) {
      if (c + 1 < nchunks) stage(c + 1, tid - 64, 192);
    } else {
      const unsigned long long* buf = rows + (c & 1) * 64 * words;
      const int r0 = c * 64, rn = min(64, n - r0);
      // serial recurrence on the diagonal word; lane rr holds diag of row rr
      const unsigned long long diag = lane < rn ? buf[lane * words + c] : 0ULL;
      // keep_i = alive_i && no kept j < i suppresses i.  Instead of walking the 64 boxes one by one (a ~100-cycle
      // scalar step each), iterate K <- alive & ~OR_{j in K} diag_j from K = alive: position i is final after i + 1
      // rounds at the latest (diag_j only has bits > j), the fixpoint is unique, and in practice a handful of rounds
      // reach it; one round is a 64-bit OR-reduction over the wave (DPP row scans + four readlanes).
      const unsigned long long rem = readlane64(removed, c);
      const unsigned long long alive = ~rem & (rn == 64 ? ~0ULL : ((1ULL << rn) - 1ULL));
      unsigned long long keptbits = alive;
      for (int round = 0; round < 64; round++) {
        const bool in = (keptbits >> lane) & 1ULL;
        const unsigned long long sup = wave_or64(in ? diag : 0ULL);
        const unsigned long long next = alive & ~sup;
        if (next == keptbits) break;
        keptbits = next;
      }
      // emit kept indices (ascending) and fold kept rows into the bitmap
      {
        const bool mine = (keptbits >> lane) & 1ULL;
        const int before = __popcll(keptbits & ((1ULL << lane) - 1ULL));
        if (mine) kp[cnt + before] = r0 + lane;
        cnt += __popcll(keptbits);
      }
      if (lane >= c && lane < nchunks) {
        unsigned long long acc = removed;
#pragma unroll 8
        for (int rr = 0; rr < rn; rr++) {  // branch-free: independent, pipelined LDS reads
          const unsigned long long v = buf[rr * words + lane];
          acc |= ((keptbits >> rr) & 1ULL) ? v : 0ULL;
        }
        removed = acc;
      }
    }
    __syncthreads();
  }
  if (tid == 0) keep_cnt[seg] = cnt;
}

extern "C" int mmt_nms_batched(const float* boxes, const int32_t* seg_off, int B, int max_n, float thr,
                               uint64_t* mask_ws, int32_t* keep, int32_t* keep_cnt, void* stream) {
  if (B <= 0) return 0;
  if (max_n <= 0 || max_n > 64 * 64) return MMT_EINVAL;
  const int words = (max_n + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, B), dim3(64), 0, (hipStream_t)stream, boxes, seg_off, max_n,
                     words, thr, (unsigned long long*)mask_ws);
  MMT_LAUNCH_CHECK();
  const size_t lds = (size_t)2 * 64 * words * sizeof(unsigned long long);
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, seg_off, max_n, words,
                     (const unsigned long long*)mask_ws, keep, keep_cnt);
  MMT_LAUNCH_CHECK();
  return 0;
}
