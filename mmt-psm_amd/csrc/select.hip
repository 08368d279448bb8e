// Selection kernels of the proposal pipeline: what the reference does with sigmoid / topk / index /
// cat / nonzero / randperm tensor calls around its NMS (rpn/inference.py:78-243, balanced_positive_negative_sampler.py:20-72),
// each of them a handful of launches per (image, level), as three launches per call here:
//
//   rpn_gather_kernel   after the per-level top-k of the objectness: gather of the logit, box deltas and anchors of every
//                       selected index, sigmoid, BoxCoder.decode (weights 1), clip_to_image -> candidate boxes / scores
//                       in the (image, level, rank) order the batched NMS wants
//   rpn_post_kernel     after the NMS: the per-level POST_NMS_TOP_N cut, then FPN_POST_NMS_TOP_N over the whole batch
//                       (training) or per image in score order (inference), ground-truth boxes appended (training):
//                       fixed-capacity proposal tensors + counts on the device
//   sample_kernel       BalancedPositiveNegativeSampler: the num_pos / num_neg candidates with the smallest random keys
//
// The last two are built on one primitive, an exact block-wide radix SELECT over 64-bit keys that are unique by construction
// (value bits in the high word, element index in the low word): "the k largest keys" is then a well-defined set and order,
// with the reference's unspecified order between equal values resolved as "lower index first".  Integer / comparison work
// only besides the decode; built with -ffp-contract=off like targets.hip.
#include <string.h>
#include "common.h"

namespace {

constexpr int NT = 1024;

__device__ __forceinline__ unsigned f2ord(float f) {  // order-preserving float -> unsigned
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SelShared {
  unsigned hist[256];
  unsigned long long prefix, mask;
  int desired, count;
};

// Threshold T such that exactly min(k, #valid) of the keys key(i), i in [0, n), are >= T.  key(i) == 0 means "not a
// candidate"; candidate keys are unique and non-zero.  Returns 1 (everything non-zero) when there are <= k candidates.
// All NT threads call it; `sh` is block-shared.
template <class KeyFn>
__device__ unsigned long long block_select(KeyFn key, const int n, const int k, SelShared& sh) {
  const int tid = threadIdx.x;
  if (tid == 0) { sh.prefix = 0; sh.mask = 0; sh.desired = k; sh.count = 0; }
  __syncthreads();
  // count the candidates once: with <= k of them there is nothing to select
  {
    int c = 0;
    for (int i = tid; i < n; i += 4 * NT) {
      unsigned long long kk[4];
#pragma unroll
      for (int u = 0; u < 4; u++) kk[u] = i + u * NT < n ? key(i + u * NT) : 0ull;
#pragma unroll
      for (int u = 0; u < 4; u++) c += kk[u] != 0ull;
    }
    if (c) atomicAdd(&sh.count, c);
    __syncthreads();
    if (sh.count <= k) return 1ull;
  }
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (tid < 256) sh.hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = sh.prefix, mask = sh.mask;
    for (int i = tid; i < n; i += 4 * NT) {   // four keys in flight per thread (keys usually come from global memory)
      unsigned long long kk[4];
#pragma unroll
      for (int u = 0; u < 4; u++) kk[u] = i + u * NT < n ? key(i + u * NT) : 0ull;
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (kk[u] != 0ull && (kk[u] & mask) == prefix) atomicAdd(&sh.hist[(unsigned)(kk[u] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int want = sh.desired, b = 255;
      for (; b > 0; b--) {
        const int c = (int)sh.hist[b];
        if (c >= want) break;
        want -= c;
      }
      sh.desired = want;
      sh.prefix = prefix | ((unsigned long long)b << shift);
      sh.mask = mask | (0xffull << shift);
    }
    __syncthreads();
  }
  return sh.prefix;  // all 64 bits fixed: the key of the k-th largest candidate
}

// descending bitonic sort of P (power of two, <= 2048) 64-bit keys in LDS
__device__ void block_sort_desc(unsigned long long* a, const int P) {
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < P / 2; t += NT) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long x = a[lo], y = a[hi];
        if ((x < y) == desc) { a[lo] = y; a[hi] = x; }
      }
    }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ RPN candidates
// The top-k itself stays a library call per level (torch.topk over [N, H*W*A] logits: a multi-block radix select; one
// block per segment would be bound by what ONE CU can pull from L2, ~10 passes x 12 MB for the 196k-anchor level).
// Everything after it, for all levels and images at once: gather of the logit, the 4 deltas and the anchor of every
// selected index, sigmoid, BoxCoder.decode (weights 1,1,1,1), clip_to_image -- one thread per candidate.
struct RpnLevel { const float* head; const float* anchors; const long* topk; int HW; int k; int out_off; int pad; };
struct RpnSelArgs {
  RpnLevel lv[8];
  int L, N, A, C, sumk;      // C = channels of the fused head output (A logits + 4A deltas), NHWC
  float clipv;               // BoxCoder's dw / dh clip
  const float* lim;          // [N][2] = (width - 1, height - 1)
  float* boxes; float* scores; long* idx; float* box_reg;
};

__global__ __launch_bounds__(256) void rpn_gather_kernel(const RpnSelArgs a) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.N * a.sumk) return;
  const int img = t / a.sumk, r = t - img * a.sumk;
  int l = 0;
  while (l + 1 < a.L && r >= a.lv[l + 1].out_off) l++;
  const RpnLevel lv = a.lv[l];
  const int j = r - lv.out_off;
  const int i = (int)lv.topk[(long)img * lv.k + j];
  const int px = i / a.A, an = i - px * a.A;
  const float* hd = lv.head + ((long)img * lv.HW + px) * a.C;
  const float logit = hd[an];
  const float* d = hd + a.A + an * 4;
  const f32x4 b = *(const f32x4*)(lv.anchors + (long)i * 4);
  const float w = b[2] - b[0] + 1.f, h = b[3] - b[1] + 1.f;
  const float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
  const float dw = fminf(d[2], a.clipv), dh = fminf(d[3], a.clipv);
  const float pcx = d[0] * w + cx, pcy = d[1] * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  const float lw = a.lim[2 * img], lh = a.lim[2 * img + 1];
  f32x4 bx = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw - 1.f, pcy + 0.5f * ph - 1.f};
  bx[0] = fminf(fmaxf(bx[0], 0.f), lw); bx[1] = fminf(fmaxf(bx[1], 0.f), lh);
  bx[2] = fminf(fmaxf(bx[2], 0.f), lw); bx[3] = fminf(fmaxf(bx[3], 0.f), lh);
  *(f32x4*)(a.boxes + (long)t * 4) = bx;
  a.scores[t] = 1.f / (1.f + expf(-logit));  // torch.sigmoid
  a.idx[t] = i;
  *(f32x4*)(a.box_reg + (long)t * 4) = f32x4{d[0], d[1], d[2], d[3]};
}

// ------------------------------------------------------------------------------------------------ per-level top-k
// torch.topk(objectness, k, sorted=True) of every (image, level) segment (rpn/inference.py:94-96) for ALL levels and
// images of a call in five launches (a library top-k is ~8 launches PER LEVEL): a multi-block radix select over the
// order-preserving 32-bit image of the logits, 11 + 11 + 10 bits,
//   hist<0>   keys of every element written densely to scratch (the head output is read once, strided), histogram of the
//             top 11 bits per segment (LDS histogram per 8192-element chunk, then global atomics on the non-empty bins)
//   hist<1|2> every block re-derives the selected bin(s) of its segment from the earlier histograms (a 2048-bin scan) and
//             histograms the next digit of the matching keys
//   compact   threshold key T = value of the k-th largest: every element with key >= T becomes a candidate (value ||
//             ~index as one 64-bit word; equal values -> lower index first, as everywhere in this file)
//   final     one block per segment: bitonic sort of the <= 4096 candidates, the first k indices are the result.  More
//             than 4096 candidates = thousands of logits EQUAL to the threshold (blank images): exact 64-bit select over
//             the whole segment instead (slow, correct).
constexpr int TK_BINS = 2048, TK_CHUNK = 8192, TK_CAP = 4096;
struct TopkLevel { const float* head; long* out; int HW; int k; int chunk0; int key_off; };   // key_off: offset of the level in an image's dense keys
struct TopkArgs {
  TopkLevel lv[8];
  int L, N, A, C, chunks, keys_per_image;
  unsigned* hist;              // [N*L][3][TK_BINS], zeroed
  unsigned* keys;              // [N][keys_per_image]
  unsigned long long* cand;    // [N*L][TK_CAP]
  int* cnt;                    // [N*L], zeroed
};

// largest bin b with sum_{j >= b} h[j] >= want; want <- want - sum_{j > b} h[j].  256 threads, h in global memory.
__device__ __forceinline__ void tk_scan(const unsigned* __restrict__ h, int& bin, int& want, int* sh /*[8]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned v[8];
  int mine = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { v[j] = h[TK_BINS - 1 - (tid * 8 + j)]; mine += (int)v[j]; }   // thread t: bins descending
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  __syncthreads();
  if (lane == 63) sh[wv] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wv; w++) base += sh[w];
  const int before = base + incl - mine;   // elements in bins above this thread's
  __syncthreads();
  if (before < want && before + mine >= want) {
    int acc = before;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (acc < want && acc + (int)v[j] >= want) { sh[4] = TK_BINS - 1 - (tid * 8 + j); sh[5] = want - acc; }
      acc += (int)v[j];
    }
  }
  __syncthreads();
  bin = sh[4];
  want = sh[5];
  __syncthreads();
}

__device__ __forceinline__ void tk_locate(const TopkArgs& a, int& l, int& img, int& chunk) {
  const int b = blockIdx.x % a.chunks;
  img = blockIdx.x / a.chunks;
  l = 0;
  while (l + 1 < a.L && b >= a.lv[l + 1].chunk0) l++;
  chunk = b - a.lv[l].chunk0;
}

template <int PASS>  // 0, 1, 2: histogram of digit PASS; 3: candidate compaction
__global__ __launch_bounds__(256) void topk_pass_kernel(const TopkArgs a) {
  __shared__ unsigned lh[TK_BINS];
  __shared__ int sh[8];
  int l, img, chunk;
  tk_locate(a, l, img, chunk);
  const TopkLevel lv = a.lv[l];
  const int n = lv.HW * a.A, seg = img * a.L + l, tid = threadIdx.x;
  const int i0 = chunk * TK_CHUNK, i1 = min(i0 + TK_CHUNK, n);
  unsigned* keys = a.keys + (long)img * a.keys_per_image + lv.key_off;
  const unsigned* h = a.hist + (long)seg * 3 * TK_BINS;
  unsigned prefix = 0, mask = 0;
  int want = min(lv.k, n);
  if (PASS >= 1) { int b; tk_scan(h, b, want, sh); prefix = (unsigned)b << 21; mask = 0xffe00000u; }
  if (PASS >= 2) { int b; tk_scan(h + TK_BINS, b, want, sh); prefix |= (unsigned)b << 10; mask = 0xfffffc00u; }
  if (PASS == 3) {
    int b;
    tk_scan(h + 2 * TK_BINS, b, want, sh);
    const unsigned T = prefix | (unsigned)b;
    for (int i = i0 + tid; i < i1; i += 256) {
      const unsigned kk = keys[i];
      if (kk >= T) {
        const int p = atomicAdd(a.cnt + seg, 1);
        if (p < TK_CAP) a.cand[(long)seg * TK_CAP + p] = ((unsigned long long)kk << 32) | (unsigned)(0xffffffffu - (unsigned)i);
      }
    }
    return;
  }
  for (int j = tid; j < TK_BINS; j += 256) lh[j] = 0;
  __syncthreads();
  if (PASS == 0) {
    const float* hd = lv.head + (long)img * lv.HW * a.C;
    for (int i = i0 + tid; i < i1; i += 256) {
      const int px = i / a.A, an = i - px * a.A;
      const unsigned kk = f2ord(hd[(long)px * a.C + an]);
      keys[i] = kk;
      atomicAdd(&lh[kk >> 21], 1u);
    }
  } else {
    const int shift = PASS == 1 ? 10 : 0;
    const unsigned dm = PASS == 1 ? 0x7ffu : 0x3ffu;
    for (int i = i0 + tid; i < i1; i += 256) {
      const unsigned kk = keys[i];
      if ((kk & mask) == prefix) atomicAdd(&lh[(kk >> shift) & dm], 1u);
    }
  }
  __syncthreads();
  unsigned* gh = a.hist + ((long)seg * 3 + PASS) * TK_BINS;
  for (int j = tid; j < TK_BINS; j += 256)
    if (lh[j]) atomicAdd(gh + j, lh[j]);
}

__global__ __launch_bounds__(NT) void topk_final_kernel(const TopkArgs a) {
  __shared__ SelShared sh;
  __shared__ unsigned long long list[TK_CAP];
  __shared__ int n_list;
  const int seg = blockIdx.x, img = seg / a.L, l = seg % a.L, tid = threadIdx.x;
  const TopkLevel lv = a.lv[l];
  const int n = lv.HW * a.A, k = min(lv.k, n);
  int c = a.cnt[seg];
  if (c > TK_CAP) {   // massive ties at the threshold: exact select with the index in the key, over the whole segment
    const unsigned* keys = a.keys + (long)img * a.keys_per_image + lv.key_off;
    auto k64 = [&](int i) -> unsigned long long { return ((unsigned long long)keys[i] << 32) | (unsigned)(0xffffffffu - (unsigned)i); };
    const unsigned long long T = block_select(k64, n, k, sh);
    if (tid == 0) n_list = 0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
      const unsigned long long kk = k64(i);
      if (kk >= T) { const int p = atomicAdd(&n_list, 1); if (p < TK_CAP) list[p] = kk; }
    }
    __syncthreads();
    c = n_list;
  } else {
    for (int j = tid; j < c; j += NT) list[j] = a.cand[(long)seg * TK_CAP + j];
  }
  int P = 2;
  while (P < c) P <<= 1;
  for (int j = c + tid; j < P; j += NT) list[j] = 0ull;
  block_sort_desc(list, P);
  long* out = lv.out + (long)img * lv.k;
  for (int j = tid; j < k; j += NT) out[j] = (long)(0xffffffffu - (unsigned)list[j]);
}

// ------------------------------------------------------------------------------------------------ after the NMS
struct RpnPostArgs {
  const float* boxes; const float* scores; const long* idx; const float* box_reg;   // candidates [N][sumk]
  const int* keep; const int* keep_cnt;   // NMS result per segment (image-major, level-minor): [N*L][kmax], [N*L]
  int seg_off[9], own_pre[8];             // level offsets inside an image's sumk candidates; this selector's pre-NMS prefix
  int L, N, sumk, kmax, post_n, fpn_post_n, training, cap, min_size_filter, pad;
  const float* gt; const int* gt_off;     // training: ground-truth boxes appended after the selection (or null)
  float* out_boxes; float* out_scores; long* out_idx; float* out_reg; int* out_level; int* out_cnt;  // [N][cap] ..., [N]
  unsigned long long* key_scratch;        // [N * L * min(post_n + 1, kmax)] words of workspace
};

// entry e of image n = (level l, rank j < min(post_n, kmax)): valid when j < keep_cnt and its position lies in the
// selector's own pre-NMS prefix.  Training: ONE block ranks the valid entries of the whole batch, keeps the best
// fpn_post_n and writes them per image in (level, rank) order.  Inference: one block per image, best fpn_post_n of the
// image in descending score order.
__global__ __launch_bounds__(NT) void rpn_post_kernel(const RpnPostArgs a) {
  __shared__ SelShared sh;
  __shared__ unsigned long long list[2048];
  __shared__ int n_list, base;
  __shared__ int wsum[NT / 64];
  __shared__ int bad_rank[64];   // per segment: rank of the one min-size-removed box the NMS kept (they all sit on the same
                                 // far-away spot, so the first survives and suppresses the rest), or INT_MAX
  const int tid = threadIdx.x;
  const int post = a.post_n > 0 && a.post_n < a.kmax ? a.post_n : a.kmax;   // valid ranks per segment that can survive
  const int per = a.min_size_filter ? min(post + 1, a.kmax) : post;         // list ranks to look at
  const int n_img = a.training ? a.N : 1, img0 = a.training ? 0 : blockIdx.x;
  const int E = n_img * a.L * per;
  if (tid < 64) bad_rank[tid] = 0x7fffffff;
  __syncthreads();
  if (a.min_size_filter) {
    for (int sg = tid >> 6; sg < n_img * a.L; sg += NT / 64) {   // one wave per segment
      const int img = img0 + sg / a.L, l = sg % a.L, seg = img * a.L + l;
      const int c = min(a.keep_cnt[seg], per);
      int best = 0x7fffffff;
      for (int j = tid & 63; j < c; j += 64) {
        const int p = a.keep[(long)seg * a.kmax + j];
        if (a.scores[(long)img * a.sumk + a.seg_off[l] + p] < 0.f) best = min(best, j);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
      if ((tid & 63) == 0) bad_rank[sg] = best;
    }
    __syncthreads();
  }
  auto entry = [&](int e, int& img, int& pos) -> bool {   // -> valid; pos = candidate index inside the image
    const int s = e / per, j = e - s * per;
    img = img0 + s / a.L;
    const int l = s % a.L, seg = img * a.L + l;
    if (j >= a.keep_cnt[seg]) return false;
    const int br = bad_rank[s];
    if (j == br || j - (j > br) >= post) return false;   // the removed box holds no slot; ranks behind it move up by one
    const int p = a.keep[(long)seg * a.kmax + j];
    if (p >= a.own_pre[l]) return false;
    pos = a.seg_off[l] + p;
    return true;
  };
  // the key of every entry is computed ONCE (three dependent loads: count, kept position, score) into the caller's scratch
  // (this block's slice of it); the select and the compaction then stream over that array
  unsigned long long* const kbuf = a.key_scratch + (long)(a.training ? 0 : blockIdx.x) * a.L * per;
  for (int e = tid; e < E; e += NT) {
    int img, pos;
    kbuf[e] = entry(e, img, pos)
                  ? (((unsigned long long)f2ord(a.scores[(long)img * a.sumk + pos]) << 32) | (unsigned)(0xffffffffu - (unsigned)e))
                  : 0ull;
  }
  __syncthreads();
  auto key = [&](int e) -> unsigned long long { return kbuf[e]; };
  const unsigned long long T = block_select(key, E, a.fpn_post_n, sh);
  if (!a.training) {
    // descending score order (rpn/inference.py:235-242: topk(sorted=True) then index)
    if (tid == 0) n_list = 0;
    int P = 2;
    while (P < a.fpn_post_n && P < 2048) P <<= 1;
    for (int i = tid; i < P; i += NT) list[i] = 0ull;
    __syncthreads();
    for (int e = tid; e < E; e += NT) {
      const unsigned long long kk = key(e);
      if (kk != 0ull && kk >= T) list[atomicAdd(&n_list, 1)] = kk;
    }
    block_sort_desc(list, P);
    const int cnt = n_list;
    for (int j = tid; j < cnt; j += NT) {
      const int e = (int)(0xffffffffu - (unsigned)list[j]);
      int img, pos;
      entry(e, img, pos);
      const long src = (long)img * a.sumk + pos, dst = (long)img * a.cap + j;
      *(f32x4*)(a.out_boxes + dst * 4) = *(const f32x4*)(a.boxes + src * 4);
      a.out_scores[dst] = a.scores[src];
      a.out_idx[dst] = a.idx[src];
      *(f32x4*)(a.out_reg + dst * 4) = *(const f32x4*)(a.box_reg + src * 4);
      a.out_level[dst] = (e / per) % a.L;
    }
    if (tid == 0) a.out_cnt[img0] = cnt;
    return;
  }
  // training: selected entries keep their (image, level, rank) order: an ordered compaction per image.  Each thread owns
  // a contiguous run of entries, so a block scan of the run counts gives every selected entry its slot.
  for (int img = 0; img < a.N; img++) {
    const int e0 = img * a.L * per, e1 = e0 + a.L * per;
    const int chunk = (a.L * per + NT - 1) / NT;
    const int lo = e0 + tid * chunk, hi = min(lo + chunk, e1);
    int c = 0;
    for (int e = lo; e < hi; e++) { const unsigned long long kk = key(e); c += (kk != 0ull && kk >= T); }
    // exclusive scan of c over the block
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < (tid >> 6); w++) wbase += wsum[w];
    int slot = wbase + incl - c;
    if (tid == NT - 1) base = wbase + incl;   // total of the image
    for (int e = lo; e < hi; e++) {
      const unsigned long long kk = key(e);
      if (kk == 0ull || kk < T) continue;
      int im, pos;
      entry(e, im, pos);
      const long src = (long)img * a.sumk + pos, dst = (long)img * a.cap + slot++;
      *(f32x4*)(a.out_boxes + dst * 4) = *(const f32x4*)(a.boxes + src * 4);
      a.out_scores[dst] = a.scores[src];
      a.out_idx[dst] = a.idx[src];
      *(f32x4*)(a.out_reg + dst * 4) = *(const f32x4*)(a.box_reg + src * 4);
      a.out_level[dst] = (e / per) % a.L;
    }
    __syncthreads();
    int total = base;
    if (a.gt) {  // add_gt_proposals (rpn/inference.py:55-76): the image's gt boxes with objectness 1
      const int g0 = a.gt_off[img], g1 = a.gt_off[img + 1];
      for (int g = g0 + tid; g < g1; g += NT) {
        const long dst = (long)img * a.cap + total + (g - g0);
        *(f32x4*)(a.out_boxes + dst * 4) = *(const f32x4*)(a.gt + (long)g * 4);
        a.out_scores[dst] = 1.f;
        a.out_idx[dst] = -1;
        *(f32x4*)(a.out_reg + dst * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        a.out_level[dst] = -1;
      }
      total += g1 - g0;
    }
    if (tid == 0) a.out_cnt[img] = total;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ fg / bg sampler
// labels: >= 1 positive, 0 negative, < 0 ignored; per image the num_pos = min(#pos, P) positives and
// num_neg = min(#neg, B - num_pos) negatives with the SMALLEST keys (uniform random keys -> a uniform sample without
// replacement, balanced_positive_negative_sampler.py:40-60); equal keys: lower index first.
// Uniform keys make this cheap: the num smallest of m keys all lie below tau = 8 num / m (expected 8 num members there),
// so ONE pass over the labels collects the members below tau into LDS (<= 4096) and the exact select runs on that list.
// Any key distribution stays correct: if fewer than num or more than 4096 land below tau the select runs over the whole
// vector instead (many passes: slow, never seen with uniform keys).
template <typename LT>
__global__ __launch_bounds__(NT) void sample_kernel(const LT* __restrict__ labels, const float* __restrict__ keys,
                                                    const int* __restrict__ off, const int batch, const int max_pos,
                                                    unsigned char* __restrict__ pos_mask, unsigned char* __restrict__ neg_mask,
                                                    int* __restrict__ counts) {
  constexpr int CAP = 4096;
  __shared__ SelShared sh;
  __shared__ unsigned long long list[CAP];
  __shared__ int npos_s, nneg_s, nl_s;
  __shared__ unsigned long long lmin_s;
  const int img = blockIdx.x, tid = threadIdx.x;
  const int o0 = off[img], n = off[img + 1] - o0;
  const LT* lab = labels + o0;
  const float* ky = keys + o0;
  if (tid == 0) { npos_s = 0; nneg_s = 0; }
  __syncthreads();
  // (every pass below keeps FOUR independent loads per thread in flight: one block walks 262 k anchors per image, and with
  // one load per thread per iteration a pass was a chain of 256 memory latencies -- 0.6 ms per call for the RPN sampler)
  int cp = 0, cn = 0;
  for (int i = tid; i < n; i += 4 * NT) {
    LT l[4];
#pragma unroll
    for (int u = 0; u < 4; u++) l[u] = i + u * NT < n ? lab[i + u * NT] : (LT)-1;
#pragma unroll
    for (int u = 0; u < 4; u++) { cp += l[u] >= (LT)1; cn += l[u] == (LT)0; }
  }
  if (cp) atomicAdd(&npos_s, cp);
  if (cn) atomicAdd(&nneg_s, cn);
  __syncthreads();
  const int npos = npos_s, nneg = nneg_s;
  const int num_pos = min(npos, max_pos), num_neg = min(nneg, batch - num_pos);
  // smallest key first = largest inverted key; the index in the low word makes keys unique (equal keys: lower index first)
  auto k64 = [&](int i) -> unsigned long long {
    return ((unsigned long long)(~f2ord(ky[i])) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
  };
  // threshold key of one class (want_pos: labels >= 1, else labels == 0): members with k64 >= T are sampled
  auto threshold = [&](const bool want_pos, const int members, const int num) -> unsigned long long {
    if (num <= 0) return ~0ull;          // nothing
    if (members <= num) return 1ull;     // everything
    auto member = [&](int i) { return want_pos ? lab[i] >= (LT)1 : lab[i] == (LT)0; };
    const float tau = fminf(1.0f, 8.0f * (float)num / (float)members);
    __syncthreads();
    if (tid == 0) nl_s = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 4 * NT) {
      LT l[4];
      float kf[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool in = i + u * NT < n;
        l[u] = in ? lab[i + u * NT] : (LT)-1;
        kf[u] = in ? ky[i + u * NT] : 2.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if ((want_pos ? l[u] >= (LT)1 : l[u] == (LT)0) && kf[u] < tau) {
          const int p = atomicAdd(&nl_s, 1);
          if (p < CAP) list[p] = ((unsigned long long)(~f2ord(kf[u])) << 32) | (unsigned)(0xffffffffu - (unsigned)(i + u * NT));
        }
    }
    __syncthreads();
    const int nl = nl_s;
    if (nl > num && nl <= CAP) {
      auto lk = [&](int j) -> unsigned long long { return list[j]; };
      return block_select(lk, nl, num, sh);
    }
    if (nl == num && nl <= CAP) {
      // exactly `num` members below tau: they ARE the sample, and the threshold is the smallest key of the list
      // (block_select's "<= k candidates: everything" answer would be read against ALL members of the class below)
      if (tid == 0) lmin_s = ~0ull;
      __syncthreads();
      unsigned long long m = ~0ull;
      for (int j = tid; j < nl; j += NT) m = list[j] < m ? list[j] : m;
      if (m != ~0ull) atomicMin(&lmin_s, m);
      __syncthreads();
      return lmin_s;
    }
    auto gk = [&](int i) -> unsigned long long { return member(i) ? k64(i) : 0ull; };
    return block_select(gk, n, num, sh);
  };
  const unsigned long long Tp = threshold(true, npos, num_pos);
  const unsigned long long Tn = threshold(false, nneg, num_neg);
  __syncthreads();
  for (int i = tid; i < n; i += 4 * NT) {
    LT l[4];
    float kf[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool in = i + u * NT < n;
      l[u] = in ? lab[i + u * NT] : (LT)-1;
      kf[u] = in ? ky[i + u * NT] : 2.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = i + u * NT;
      if (j >= n) break;
      const unsigned long long k = ((unsigned long long)(~f2ord(kf[u])) << 32) | (unsigned)(0xffffffffu - (unsigned)j);
      pos_mask[o0 + j] = (l[u] >= (LT)1 && k >= Tp) ? 1 : 0;
      neg_mask[o0 + j] = (l[u] == (LT)0 && k >= Tn) ? 1 : 0;
    }
  }
  if (tid == 0) { counts[2 * img] = num_pos; counts[2 * img + 1] = num_neg; }
}

// The same sampler for LONG label vectors (the RPN's 262 k anchors per image): the single-block kernel above walks the vector
// four times with one block per image -- 4 x 64 rounds of memory latency, 0.15 - 0.46 ms per call on the main stream's chain.
// Here the three streaming passes are many blocks wide (SW_CHUNK labels per block) and only the exact select on the short list
// stays a one-block-per-image step: count -> collect both classes' members below their tau into global lists -> thresholds
// (same `block_select`, same fall-backs: a list that overflowed or came up short selects over the whole vector) -> masks.
// The order in which the lists fill is arbitrary, the keys are unique: the result is the single-block kernel's, bit for bit.
constexpr int SW_CAP = 4096;
constexpr int SW_CHUNK = 16384;
struct SampleWs {                 // per image
  int npos, nneg, nl[2];
  unsigned long long thr[2];
  unsigned long long list[2][SW_CAP];
};

template <typename LT>
__global__ __launch_bounds__(NT) void sample_count_kernel(const LT* __restrict__ labels, const int* __restrict__ off,
                                                          SampleWs* __restrict__ ws) {
  const int img = blockIdx.y, tid = threadIdx.x;
  const int o0 = off[img], n = off[img + 1] - o0;
  const int c0 = blockIdx.x * SW_CHUNK, c1 = min(n, c0 + SW_CHUNK);
  if (c0 >= n) return;
  const LT* lab = labels + o0;
  int cp = 0, cn = 0;
  for (int i = c0 + tid; i < c1; i += 4 * NT) {
    LT l[4];
#pragma unroll
    for (int u = 0; u < 4; u++) l[u] = i + u * NT < c1 ? lab[i + u * NT] : (LT)-1;
#pragma unroll
    for (int u = 0; u < 4; u++) { cp += l[u] >= (LT)1; cn += l[u] == (LT)0; }
  }
  __shared__ int sp, sn;
  if (tid == 0) { sp = 0; sn = 0; }
  __syncthreads();
  if (cp) atomicAdd(&sp, cp);
  if (cn) atomicAdd(&sn, cn);
  __syncthreads();
  if (tid == 0) {
    if (sp) atomicAdd(&ws[img].npos, sp);
    if (sn) atomicAdd(&ws[img].nneg, sn);
  }
}

template <typename LT>
__global__ __launch_bounds__(NT) void sample_collect_kernel(const LT* __restrict__ labels, const float* __restrict__ keys,
                                                            const int* __restrict__ off, const int batch, const int max_pos,
                                                            SampleWs* __restrict__ ws) {
  const int img = blockIdx.y, tid = threadIdx.x;
  const int o0 = off[img], n = off[img + 1] - o0;
  const int c0 = blockIdx.x * SW_CHUNK, c1 = min(n, c0 + SW_CHUNK);
  if (c0 >= n) return;
  const LT* lab = labels + o0;
  const float* ky = keys + o0;
  SampleWs& w = ws[img];
  const int npos = w.npos, nneg = w.nneg;
  const int num_pos = min(npos, max_pos), num_neg = min(nneg, batch - num_pos);
  // a class needs a list only when it is actually cut (threshold(): 0 < num < members)
  const bool need_p = num_pos > 0 && npos > num_pos, need_n = num_neg > 0 && nneg > num_neg;
  if (!need_p && !need_n) return;
  const float tau_p = need_p ? fminf(1.0f, 8.0f * (float)num_pos / (float)npos) : -1.f;
  const float tau_n = need_n ? fminf(1.0f, 8.0f * (float)num_neg / (float)nneg) : -1.f;
  for (int i = c0 + tid; i < c1; i += 4 * NT) {
    LT l[4];
    float kf[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool in = i + u * NT < c1;
      l[u] = in ? lab[i + u * NT] : (LT)-1;
      kf[u] = in ? ky[i + u * NT] : 2.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool isp = l[u] >= (LT)1, isn = l[u] == (LT)0;
      if ((isp && kf[u] < tau_p) || (isn && kf[u] < tau_n)) {
        const int cls = isp ? 0 : 1;
        const int p = atomicAdd(&w.nl[cls], 1);
        if (p < SW_CAP) w.list[cls][p] = ((unsigned long long)(~f2ord(kf[u])) << 32) | (unsigned)(0xffffffffu - (unsigned)(i + u * NT));
      }
    }
  }
}

template <typename LT>
__global__ __launch_bounds__(NT) void sample_threshold_kernel(const LT* __restrict__ labels, const float* __restrict__ keys,
                                                              const int* __restrict__ off, const int batch, const int max_pos,
                                                              SampleWs* __restrict__ ws, int* __restrict__ counts) {
  __shared__ SelShared sh;
  __shared__ unsigned long long list[SW_CAP];
  __shared__ unsigned long long lmin_s;
  const int img = blockIdx.x, tid = threadIdx.x;
  const int o0 = off[img], n = off[img + 1] - o0;
  const LT* lab = labels + o0;
  const float* ky = keys + o0;
  SampleWs& w = ws[img];
  const int npos = w.npos, nneg = w.nneg;
  const int num_pos = min(npos, max_pos), num_neg = min(nneg, batch - num_pos);
  auto k64 = [&](int i) -> unsigned long long {
    return ((unsigned long long)(~f2ord(ky[i])) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
  };
  auto threshold = [&](const int cls, const int members, const int num) -> unsigned long long {
    if (num <= 0) return ~0ull;          // nothing
    if (members <= num) return 1ull;     // everything
    const int nl = w.nl[cls];
    __syncthreads();
    if (nl >= num && nl <= SW_CAP)
      for (int j = tid; j < nl; j += NT) list[j] = w.list[cls][j];
    __syncthreads();
    if (nl > num && nl <= SW_CAP) {
      auto lk = [&](int j) -> unsigned long long { return list[j]; };
      return block_select(lk, nl, num, sh);
    }
    if (nl == num && nl <= SW_CAP) {     // exactly `num` members below tau: they are the sample
      if (tid == 0) lmin_s = ~0ull;
      __syncthreads();
      unsigned long long m = ~0ull;
      for (int j = tid; j < nl; j += NT) m = list[j] < m ? list[j] : m;
      if (m != ~0ull) atomicMin(&lmin_s, m);
      __syncthreads();
      return lmin_s;
    }
    auto gk = [&](int i) -> unsigned long long { return (cls == 0 ? lab[i] >= (LT)1 : lab[i] == (LT)0) ? k64(i) : 0ull; };
    return block_select(gk, n, num, sh);
  };
  const unsigned long long Tp = threshold(0, npos, num_pos);
  const unsigned long long Tn = threshold(1, nneg, num_neg);
  if (tid == 0) {
    w.thr[0] = Tp;
    w.thr[1] = Tn;
    counts[2 * img] = num_pos;
    counts[2 * img + 1] = num_neg;
  }
}

template <typename LT>
__global__ __launch_bounds__(NT) void sample_mask_kernel(const LT* __restrict__ labels, const float* __restrict__ keys,
                                                         const int* __restrict__ off, const SampleWs* __restrict__ ws,
                                                         unsigned char* __restrict__ pos_mask, unsigned char* __restrict__ neg_mask) {
  const int img = blockIdx.y, tid = threadIdx.x;
  const int o0 = off[img], n = off[img + 1] - o0;
  const int c0 = blockIdx.x * SW_CHUNK, c1 = min(n, c0 + SW_CHUNK);
  if (c0 >= n) return;
  const LT* lab = labels + o0;
  const float* ky = keys + o0;
  const unsigned long long Tp = ws[img].thr[0], Tn = ws[img].thr[1];
  for (int i = c0 + tid; i < c1; i += 4 * NT) {
    LT l[4];
    float kf[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool in = i + u * NT < c1;
      l[u] = in ? lab[i + u * NT] : (LT)-1;
      kf[u] = in ? ky[i + u * NT] : 2.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = i + u * NT;
      if (j >= c1) break;
      const unsigned long long k = ((unsigned long long)(~f2ord(kf[u])) << 32) | (unsigned)(0xffffffffu - (unsigned)j);
      pos_mask[o0 + j] = (l[u] >= (LT)1 && k >= Tp) ? 1 : 0;
      neg_mask[o0 + j] = (l[u] == (LT)0 && k >= Tn) ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------ detections
// PostProcessor.filter_results (box_head/inference.py:91-160) for a batch without a host round trip in the middle: per
// (image, foreground class) the candidates with score > thresh in stable descending score order (det_sort_kernel: one
// sorting block per segment, segments laid out back to back -- every block counts the candidates of the segments before
// it itself, B is a handful), mmt_nms_batched over the segments, then per image (det_finish_kernel) the survivors of
// every class in ascending ORIGINAL row order (`_C.nms` returns ascending indices, cpu/nms_cpu.cpp:64), classes
// concatenated, and the DETECTIONS_PER_IMG cut `score >= kthvalue(scores, n - D + 1)` (ties kept, list order kept).
struct DetArgs {
  const float* prob;      // [rows][nc] class probabilities
  const float* boxes;     // [rows][nc * 4] decoded boxes
  const int* row_off;     // [N + 1] rows of image n
  int N, nc, max_n, D, capo;
  float thresh;
  float* sboxes; float* sscores; int* srow; int* seg_off;   // sorted candidates [<= rows * (nc - 1)], seg_off [B + 1]
  const int* keep; const int* keep_cnt;                      // NMS result [B][max_n], [B]
  int* cand;                                                 // [N][capo] scratch
  float* out_boxes; float* out_scores; long* out_labels; int* out_cnt;   // [N][capo] ..., [N]
};

__global__ __launch_bounds__(NT) void det_sort_kernel(const DetArgs a) {
  __shared__ unsigned long long list[2048];
  __shared__ int n_before, n_valid;
  const int tid = threadIdx.x, seg = blockIdx.x, F = a.nc - 1;
  const int img = seg / F, j = seg % F + 1;
  if (tid == 0) { n_before = 0; n_valid = 0; }
  __syncthreads();
  {  // candidates of the segments before this one
    int c = 0;
    for (int s2 = 0; s2 < seg; s2++) {
      const int im = s2 / F, jj = s2 % F + 1, r0 = a.row_off[im], R = a.row_off[im + 1] - r0;
      for (int t = tid; t < R; t += NT) c += a.prob[(long)(r0 + t) * a.nc + jj] > a.thresh;
    }
    if (c) atomicAdd(&n_before, c);
  }
  const int r0 = a.row_off[img], R = a.row_off[img + 1] - r0;
  int P = 2;
  while (P < R) P <<= 1;
  int c = 0;
  for (int t = tid; t < P; t += NT) {
    float sc = -1.f;
    if (t < R) sc = a.prob[(long)(r0 + t) * a.nc + j];
    const bool ok = t < R && sc > a.thresh;
    list[t] = ok ? (((unsigned long long)f2ord(sc) << 32) | (unsigned)(0xffffffffu - (unsigned)t)) : 0ull;
    c += ok;
  }
  if (c) atomicAdd(&n_valid, c);
  block_sort_desc(list, P);    // (starts and ends with a barrier) equal scores: lower row first, as the stable sort
  const int off = n_before, nv = n_valid;
  for (int pos = tid; pos < nv; pos += NT) {
    const int row = (int)(0xffffffffu - (unsigned)list[pos]);
    *(f32x4*)(a.sboxes + (long)(off + pos) * 4) = *(const f32x4*)(a.boxes + ((long)(r0 + row) * a.nc + j) * 4);
    a.sscores[off + pos] = a.prob[(long)(r0 + row) * a.nc + j];
    a.srow[off + pos] = row;
  }
  if (tid == 0) {
    a.seg_off[seg] = off;
    if (seg == gridDim.x - 1) a.seg_off[seg + 1] = off + nv;
  }
}

__global__ __launch_bounds__(NT) void det_finish_kernel(const DetArgs a) {
  __shared__ SelShared sh;
  __shared__ unsigned long long list[2048];
  __shared__ int wsum[NT / 64];
  __shared__ int cum[65], total_sel;
  const int tid = threadIdx.x, img = blockIdx.x, F = a.nc - 1;
  int* const cand = a.cand + (long)img * a.capo;
  int total = 0;
  for (int j = 1; j <= F; j++) {
    const int seg = img * F + j - 1, c = min(a.keep_cnt[seg], 2048), s0 = a.seg_off[seg];
    int P = 2;
    while (P < c) P <<= 1;
    __syncthreads();
    for (int t = tid; t < P; t += NT) {
      unsigned long long k = 0ull;
      if (t < c) {
        const int pos = a.keep[(long)seg * a.max_n + t];
        k = ((unsigned long long)(0xffffffffu - (unsigned)a.srow[s0 + pos]) << 32) | (unsigned)pos;
      }
      list[t] = k;
    }
    block_sort_desc(list, P);   // descending in ~row = ascending original row
    for (int t = tid; t < c; t += NT) cand[total + t] = s0 + (int)(unsigned)list[t];
    if (tid == 0) cum[j] = total + c;
    total += c;
  }
  if (tid == 0) cum[0] = 0;
  __syncthreads();   // cand (global, written by this block) and cum are visible to every thread of the block
  const int n = total;
  unsigned thr = 0u;
  if (a.D > 0 && n > a.D) {
    auto key = [&](int e) -> unsigned long long {
      return ((unsigned long long)f2ord(a.sscores[cand[e]]) << 32) | (unsigned)(0xffffffffu - (unsigned)e);
    };
    thr = (unsigned)(block_select(key, n, a.D, sh) >> 32);   // score of the D-th largest: everything >= it stays (ties too)
  }
  // ordered compaction: each thread owns a contiguous run of the list
  const int chunk = (n + NT - 1) / NT, lo = min(tid * chunk, n), hi = min(lo + chunk, n);
  int c = 0;
  for (int e = lo; e < hi; e++) c += f2ord(a.sscores[cand[e]]) >= thr;
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < (tid >> 6); w++) wbase += wsum[w];
  int slot = wbase + incl - c;
  if (tid == NT - 1) total_sel = wbase + incl;
  for (int e = lo; e < hi; e++) {
    const int src = cand[e];
    const float sc = a.sscores[src];
    if (f2ord(sc) < thr) continue;
    int j = 1;
    while (e >= cum[j]) j++;
    const long dst = (long)img * a.capo + slot++;
    *(f32x4*)(a.out_boxes + dst * 4) = *(const f32x4*)(a.sboxes + (long)src * 4);
    a.out_scores[dst] = sc;
    a.out_labels[dst] = j;
  }
  __syncthreads();
  if (tid == 0) a.out_cnt[img] = total_sel;
}


}  // namespace

extern "C" int mmt_rpn_gather_decode(const mmt_rpn_select_args* a, void* stream) {
  if (!a || a->L < 1 || a->L > 8 || a->N < 1 || !a->boxes || !a->scores || !a->idx || !a->box_reg || !a->lim) return MMT_EINVAL;
  static_assert(sizeof(mmt_rpn_select_args) == sizeof(RpnSelArgs), "argument layout");
  RpnSelArgs k;
  memcpy(&k, a, sizeof(k));
  for (int l = 0; l < k.L; l++)
    if (!k.lv[l].head || !k.lv[l].anchors || !k.lv[l].topk || k.lv[l].k < 1) return MMT_EINVAL;
  hipLaunchKernelGGL(rpn_gather_kernel, dim3(mmt_cdiv((long)k.N * k.sumk, 256)), dim3(256), 0, (hipStream_t)stream, k);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_rpn_post_select(const mmt_rpn_post_args* a, void* stream) {
  if (!a || a->L < 1 || a->L > 8 || a->N < 1 || !a->keep || !a->keep_cnt || !a->out_boxes || !a->out_cnt || !a->key_scratch)
    return MMT_EINVAL;
  static_assert(sizeof(mmt_rpn_post_args) == sizeof(RpnPostArgs), "argument layout");
  RpnPostArgs k;
  memcpy(&k, a, sizeof(k));
  // bad_rank[] holds one entry per segment of a block: all N * L segments in training (one block), L per image in inference
  // capacity: an image can receive at most L * post entries of the selection (the caller adds room for its gt boxes)
  const int post = k.post_n > 0 && k.post_n < k.kmax ? k.post_n : k.kmax;
  const int per_image = k.fpn_post_n < k.L * post ? k.fpn_post_n : k.L * post;
  if (k.fpn_post_n < 1 || (!k.training && k.fpn_post_n > 2048) || k.cap < per_image || (k.training && k.N * k.L > 64))
    return MMT_EINVAL;
  hipLaunchKernelGGL(rpn_post_kernel, dim3(k.training ? 1 : k.N), dim3(NT), 0, (hipStream_t)stream, k);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_sample_fg_bg(const void* labels, int labels_are_float, const float* keys, const int32_t* off, int n_images,
                                int batch_size_per_image, int max_pos, uint8_t* pos_mask, uint8_t* neg_mask, int32_t* counts,
                                void* stream) {
  if (!labels || !keys || !off || !pos_mask || !neg_mask || !counts || n_images < 1) return MMT_EINVAL;
  if (labels_are_float)
    hipLaunchKernelGGL(sample_kernel<float>, dim3(n_images), dim3(NT), 0, (hipStream_t)stream, (const float*)labels, keys, off,
                       batch_size_per_image, max_pos, pos_mask, neg_mask, counts);
  else
    hipLaunchKernelGGL(sample_kernel<long>, dim3(n_images), dim3(NT), 0, (hipStream_t)stream, (const long*)labels, keys, off,
                       batch_size_per_image, max_pos, pos_mask, neg_mask, counts);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" long mmt_sample_fg_bg_workspace_bytes(int n_images) { return n_images < 1 ? -1 : (long)n_images * (long)sizeof(SampleWs); }

template <typename LT>
static int sample_wide(const LT* labels, const float* keys, const int32_t* off, int n_images, int max_n, int batch, int max_pos,
                       uint8_t* pos_mask, uint8_t* neg_mask, int32_t* counts, SampleWs* ws, hipStream_t s) {
  // only the counters need zeroing (the head of every per-image record); the lists are read up to their counters
  for (int i = 0; i < n_images; i++)
    if (hipMemsetAsync(&ws[i], 0, 4 * sizeof(int), s) != hipSuccess) return MMT_EINVAL;
  const dim3 wide(mmt_cdiv(max_n, SW_CHUNK), n_images);
  hipLaunchKernelGGL(sample_count_kernel<LT>, wide, dim3(NT), 0, s, labels, off, ws);
  MMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(sample_collect_kernel<LT>, wide, dim3(NT), 0, s, labels, keys, off, batch, max_pos, ws);
  MMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(sample_threshold_kernel<LT>, dim3(n_images), dim3(NT), 0, s, labels, keys, off, batch, max_pos, ws, counts);
  MMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(sample_mask_kernel<LT>, wide, dim3(NT), 0, s, labels, keys, off, (const SampleWs*)ws, pos_mask, neg_mask);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_sample_fg_bg_wide(const void* labels, int labels_are_float, const float* keys, const int32_t* off, int n_images,
                                     int max_n, int batch_size_per_image, int max_pos, uint8_t* pos_mask, uint8_t* neg_mask,
                                     int32_t* counts, void* workspace, void* stream) {
  if (!labels || !keys || !off || !pos_mask || !neg_mask || !counts || !workspace || ((size_t)workspace & 7) || n_images < 1 ||
      n_images > 64 || max_n < 1)
    return MMT_EINVAL;
  if (labels_are_float)
    return sample_wide((const float*)labels, keys, off, n_images, max_n, batch_size_per_image, max_pos, pos_mask, neg_mask, counts,
                       (SampleWs*)workspace, (hipStream_t)stream);
  return sample_wide((const long*)labels, keys, off, n_images, max_n, batch_size_per_image, max_pos, pos_mask, neg_mask, counts,
                     (SampleWs*)workspace, (hipStream_t)stream);
}

extern "C" long mmt_rpn_topk_workspace_bytes(int N, int L, long anchors_per_image) {
  if (N < 1 || L < 1 || L > 8 || anchors_per_image < 1) return -1;
  const long segs = (long)N * L;
  return segs * 3 * TK_BINS * 4 + segs * 4 + ((long)N * anchors_per_image * 4 + 15) / 16 * 16 + segs * TK_CAP * 8 + 64;
}

extern "C" int mmt_rpn_topk(const mmt_rpn_topk_level* levels, int L, int N, int A, void* workspace, void* stream) {
  if (!levels || L < 1 || L > 8 || N < 1 || A < 1 || !workspace || ((size_t)workspace & 15)) return MMT_EINVAL;
  TopkArgs a;
  a.L = L; a.N = N; a.A = A; a.C = 5 * A;
  int chunks = 0, koff = 0;
  for (int l = 0; l < L; l++) {
    if (!levels[l].head || !levels[l].topk || levels[l].HW < 1 || levels[l].k < 1 || levels[l].k > TK_CAP / 2) return MMT_EINVAL;
    a.lv[l].head = levels[l].head; a.lv[l].out = (long*)levels[l].topk; a.lv[l].HW = levels[l].HW; a.lv[l].k = levels[l].k;
    a.lv[l].chunk0 = chunks; a.lv[l].key_off = koff;
    chunks += mmt_cdiv((long)levels[l].HW * A, TK_CHUNK);
    koff += levels[l].HW * A;
  }
  a.chunks = chunks; a.keys_per_image = koff;
  const long segs = (long)N * L;
  char* w = (char*)workspace;
  a.hist = (unsigned*)w;                      w += segs * 3 * TK_BINS * 4;
  a.cnt = (int*)w;                            w += (segs * 4 + 15) / 16 * 16;
  a.keys = (unsigned*)w;                      w += ((long)N * koff * 4 + 15) / 16 * 16;
  a.cand = (unsigned long long*)w;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(workspace, 0, (size_t)(segs * 3 * TK_BINS * 4 + (segs * 4 + 15) / 16 * 16), s) != hipSuccess) return MMT_EINVAL;
  const dim3 grid(chunks * N);
  hipLaunchKernelGGL(topk_pass_kernel<0>, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(topk_pass_kernel<1>, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(topk_pass_kernel<2>, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(topk_pass_kernel<3>, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(topk_final_kernel, dim3((int)segs), dim3(NT), 0, s, a);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" long mmt_det_workspace_bytes(int rows, int N, int nc) {
  if (rows < 0 || N < 1 || nc < 2) return -1;
  const long F = nc - 1, cap = (long)rows * F;
  // sboxes, sscores, srow | seg_off [N F + 1], keep_cnt [N F] | keep [N F][max_n <= rows] | NMS mask words | cand [N][capo]
  long b = cap * 16 + cap * 4 + cap * 4 + (N * F + 1) * 4 + N * F * 4 + N * F * (long)rows * 4 +
           N * F * (long)rows * ((rows + 63) / 64) * 8 + (long)N * cap * 4;
  return b + 256;
}

// prob [rows][nc] / boxes [rows][nc * 4] of N images (row_off [N + 1], host copy row_off_host for the bounds) ->
// out_boxes [N][capo][4], out_scores [N][capo], out_labels int64 [N][capo], out_cnt [N] with capo = max rows per image x (nc - 1).
// Three kernels + the two of mmt_nms_batched, nothing read back.  Images of at most 2048 rows, nc <= 64.
extern "C" int mmt_det_postprocess(const float* prob, const float* boxes, const int32_t* row_off, const int32_t* row_off_host,
                                   int N, int nc, float score_thresh, float nms_thresh, int detections_per_img, void* workspace,
                                   float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* out_cnt, void* stream) {
  if (!prob || !boxes || !row_off || !row_off_host || !workspace || !out_boxes || !out_scores || !out_labels || !out_cnt ||
      N < 1 || nc < 2 || nc > 64 || ((size_t)boxes & 15) || ((size_t)workspace & 15) || ((size_t)out_boxes & 15))
    return MMT_EINVAL;
  int rmax = 0;
  for (int n = 0; n < N; n++) {
    const int R = row_off_host[n + 1] - row_off_host[n];
    if (R < 0 || R > 2048) return MMT_EINVAL;
    rmax = R > rmax ? R : rmax;
  }
  const int rows = row_off_host[N] - row_off_host[0], F = nc - 1, B = N * F;
  hipStream_t s = (hipStream_t)stream;
  if (rmax == 0) return hipMemsetAsync(out_cnt, 0, (size_t)N * 4, s) == hipSuccess ? 0 : MMT_EINVAL;
  const long cap = (long)rows * F;
  char* w = (char*)workspace;
  DetArgs a;
  memset(&a, 0, sizeof(a));
  a.prob = prob; a.boxes = boxes; a.row_off = row_off;
  a.N = N; a.nc = nc; a.max_n = rmax; a.D = detections_per_img; a.capo = rmax * F; a.thresh = score_thresh;
  a.sboxes = (float*)w; w += cap * 16;
  a.sscores = (float*)w; w += cap * 4;
  a.srow = (int*)w; w += cap * 4;
  a.seg_off = (int*)w; w += (long)(B + 1) * 4;
  int* keep_cnt = (int*)w; w += (long)B * 4;
  int* keep = (int*)w; w += (long)B * rmax * 4;
  w = (char*)(((size_t)w + 15) & ~(size_t)15);
  unsigned long long* mask = (unsigned long long*)w; w += (long)B * rmax * ((rmax + 63) / 64) * 8;
  a.cand = (int*)w;
  a.keep = keep; a.keep_cnt = keep_cnt;
  a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_labels = (long*)out_labels; a.out_cnt = out_cnt;
  hipLaunchKernelGGL(det_sort_kernel, dim3(B), dim3(NT), 0, s, a);
  MMT_LAUNCH_CHECK();
  if (hipMemsetAsync(keep_cnt, 0, (size_t)B * 4, s) != hipSuccess) return MMT_EINVAL;
  const int e = mmt_nms_batched(a.sboxes, a.seg_off, B, rmax, nms_thresh, (uint64_t*)mask, keep, keep_cnt, stream);
  if (e) return e;
  hipLaunchKernelGGL(det_finish_kernel, dim3(N), dim3(NT), 0, s, a);
  MMT_LAUNCH_CHECK();
  return 0;
}
