// IR-Net attention kernels of libmmtpsm.so (include/mmtpsm.h): the multi-head geometric relation attention of the
// relation NMS (reference modeling/relation/relation_module.py:33-90, RelationModule.forward) and the cross-instance
// attention of the mask refinement (reference modeling/relation/mask_relation_module.py:199-242, CIAM_Module.forward),
// each as one forward and two backward launches instead of ~15 + ~30 library launches (batched GEMMs, top-k, softmax,
// scatter, permutes).  The problems are tiny (<= 128 boxes x 16 heads x 64 dims; <= a few hundred instances x 16
// channels x 196 pixels): they are latency-bound, not matrix-bound -- fp32 FMAs on the vector ALU, operands staged in LDS,
// one workgroup per (class, head) / per row block; nothing here is shaped for the MFMA units on purpose.
#include "common.h"

namespace {

constexpr int RA_MAXN = 128;    // boxes per class (FIRST_N = 90 in the shipped recipe): two columns per lane of a wave
constexpr int RA_MAXDV = 16;

// lanes of ONE wave exchange values through LDS: LDS operations of a wave complete in order, the fences keep the compiler from
// moving the accesses across the exchange point
#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------- relation attention, forward
// per class c and head g (N boxes, DQ = query / key dims per head, DV = value dims per head):
//   S[n][m] = scale * <Q[c,n,g,:], K[c,m,g,:]> + log(max(WG[c,n,m,g], 1e-6))
//   P[n][:] = softmax over the top-k entries of S[n][:] (zero elsewhere; ties broken towards the lower index)
//   out[n][c][g*DV + o] = bias[g*DV + o] + sum_m P[n][m] V[c,m,g,o]
// V = (appearance features) x (conv1 weight of head g)^T is a plain Linear done by the caller: the reference's
// `conv1(bmm(w, f_a))` (grouped 1x1 over the 16 x feat_dim stacked head outputs) is the same bilinear form, summed in the
// other order.  The problem is a chain of latencies, not of arithmetic: grid (G, C, row chunks of RA_RB rows) -- ~770
// workgroups for the shipped shape -- and every global operand of a workgroup (key slice, value slice, its query rows, its
// strided w_g rows) is requested in ONE round up front into LDS; a wave then owns RA_RB / 4 rows, its lanes the columns
// m = lane, lane + 64.
constexpr int RA_RB = 8;

__global__ __launch_bounds__(256) void relation_attention_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                     const float* __restrict__ wg, const float* __restrict__ v,
                                                                     const float* __restrict__ bias, int C, int N, int G, int DQ,
                                                                     int DV, int topk, float scale, float* __restrict__ P,
                                                                     float* __restrict__ out) {
  extern __shared__ float lds[];
  const int g = blockIdx.x, c = blockIdx.y, n0 = blockIdx.z * RA_RB, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nr = min(RA_RB, N - n0);
  const int KP = DQ + 1;                       // padded key rows: lanes walk m, the inner loop walks d
  float* Ks = lds;                             // [N][DQ + 1]
  float* Vs = Ks + N * KP;                     // [N][DV]
  float* Qs = Vs + N * DV;                     // [RA_RB][DQ]
  float* Ws = Qs + RA_RB * DQ;                 // [RA_RB][N]     log(max(w_g, 1e-6)) of the chunk's rows
  float* Sr = Ws + RA_RB * N;                  // [4][128]       the wave's score row
  const long ldq = (long)G * DQ, ldv = (long)G * DV;
  for (int e = tid; e < N * DQ; e += 256) {
    const int m = e / DQ, d = e - m * DQ;
    Ks[m * KP + d] = k[((long)c * N + m) * ldq + (long)g * DQ + d];
  }
  for (int e = tid; e < N * DV; e += 256) {
    const int m = e / DV, o = e - m * DV;
    Vs[e] = v[((long)c * N + m) * ldv + (long)g * DV + o];
  }
  for (int e = tid; e < nr * DQ; e += 256) {
    const int r = e / DQ, d = e - r * DQ;
    Qs[e] = q[((long)c * N + n0 + r) * ldq + (long)g * DQ + d];
  }
  for (int e = tid; e < nr * N; e += 256) Ws[e] = logf(fmaxf(wg[(((long)c * N + n0) * N + e) * G + g], 1e-6f));
  __syncthreads();
  float* sr = Sr + wave * 128;
  const int m0 = lane, m1 = lane + 64;
  for (int r = wave; r < nr; r += 4) {
    const int n = n0 + r;
    const float* qr = Qs + r * DQ;
    float s0 = 0.f, s1 = 0.f;
    if (m0 < N) {
      const float* kr = Ks + m0 * KP;
      for (int d = 0; d < DQ; ++d) s0 = fmaf(qr[d], kr[d], s0);
      s0 = s0 * scale + Ws[r * N + m0];
      sr[m0] = s0;
    }
    if (m1 < N) {
      const float* kr = Ks + m1 * KP;
      for (int d = 0; d < DQ; ++d) s1 = fmaf(qr[d], kr[d], s1);
      s1 = s1 * scale + Ws[r * N + m1];
      sr[m1] = s1;
    }
    WAVE_SYNC();
    // rank of the lane's two entries in the row: how many entries come before them in (value descending, index ascending)
    int r0 = 0, r1 = 0;
    for (int j = 0; j < N; ++j) {
      const float sj = sr[j];
      r0 += (sj > s0) || (sj == s0 && j < m0);
      r1 += (sj > s1) || (sj == s1 && j < m1);
    }
    const bool in0 = m0 < N && r0 < topk, in1 = m1 < N && r1 < topk;
    const float mx = wave_max(fmaxf(in0 ? s0 : -INFINITY, in1 ? s1 : -INFINITY));
    const float e0 = in0 ? expf(s0 - mx) : 0.f, e1 = in1 ? expf(s1 - mx) : 0.f;
    const float inv = 1.f / wave_sum(e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    float* prow = P + (((long)c * G + g) * N + n) * N;
    if (m0 < N) prow[m0] = p0;
    if (m1 < N) prow[m1] = p1;
    for (int o = 0; o < DV; ++o) {
      float a = 0.f;
      if (m0 < N) a = p0 * Vs[m0 * DV + o];
      if (m1 < N) a = fmaf(p1, Vs[m1 * DV + o], a);
      a = wave_sum(a);
      if (lane == 0) out[((long)n * C + c) * ldv + (long)g * DV + o] = a + bias[g * DV + o];
    }
    WAVE_SYNC();
  }
}

// ---------------------------------------------------------------- relation attention, backward
//   dP[n][m] = <dOut[n,c,g,:], V[c,m,g,:]>;  dS = P (dP - sum_m P dP)  (zero outside the top-k: P is zero there)
//   dQ[c,n,g,:] = scale sum_m dS[n][m] K[c,m,g,:];   dK[c,m,g,:] = scale sum_n dS[n][m] Q[c,n,g,:]
//   dV[c,m,g,:] = sum_n P[n][m] dOut[n,c,g,:];       dWG[c,n,m,g] = dS[n][m] / WG[c,n,m,g] where WG >= 1e-6 (clamp), else 0
// Two launches, both over (G, C, chunks of RA_RB) like the forward pass: the ROW pass owns rows of dS (softmax backward, dWG,
// dQ; dS is also written out), the COLUMN pass owns columns (dK, dV: sums over all rows).  Every output element belongs to
// exactly one workgroup: no atomics.
__global__ __launch_bounds__(256) void relation_attention_bwd_rows_kernel(const float* __restrict__ k, const float* __restrict__ wg,
                                                                          const float* __restrict__ v, const float* __restrict__ P,
                                                                          const float* __restrict__ dout, int C, int N, int G,
                                                                          int DQ, int DV, float scale, float* __restrict__ dS,
                                                                          float* __restrict__ dq, float* __restrict__ dwg) {
  extern __shared__ float lds[];
  const int g = blockIdx.x, c = blockIdx.y, n0 = blockIdx.z * RA_RB, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nr = min(RA_RB, N - n0);
  const int KP = DQ + 1;
  float* Ks = lds;                 // [N][DQ + 1]
  float* Vs = Ks + N * KP;         // [N][DV]
  float* Gs = Vs + N * DV;         // [RA_RB][DV]   dOut[n, c, g, :]
  float* Ps = Gs + RA_RB * DV;     // [RA_RB][N]    P rows, then dS rows
  float* Wr = Ps + RA_RB * N;      // [RA_RB][N]    w_g rows
  const long ldq = (long)G * DQ, ldv = (long)G * DV;
  const float* Pcg = P + ((long)c * G + g) * N * N;
  float* dScg = dS + ((long)c * G + g) * N * N;
  for (int e = tid; e < N * DQ; e += 256) {
    const int m = e / DQ, d = e - m * DQ;
    Ks[m * KP + d] = k[((long)c * N + m) * ldq + (long)g * DQ + d];
  }
  for (int e = tid; e < N * DV; e += 256) {
    const int m = e / DV, o = e - m * DV;
    Vs[e] = v[((long)c * N + m) * ldv + (long)g * DV + o];
  }
  for (int e = tid; e < nr * DV; e += 256) {
    const int r = e / DV, o = e - r * DV;
    Gs[e] = dout[((long)(n0 + r) * C + c) * ldv + (long)g * DV + o];
  }
  for (int e = tid; e < nr * N; e += 256) {
    Ps[e] = Pcg[(long)n0 * N + e];
    Wr[e] = wg[(((long)c * N + n0) * N + e) * G + g];
  }
  __syncthreads();
  const int m0 = lane, m1 = lane + 64;
  for (int r = wave; r < nr; r += 4) {
    const float p0 = m0 < N ? Ps[r * N + m0] : 0.f, p1 = m1 < N ? Ps[r * N + m1] : 0.f;
    float d0 = 0.f, d1 = 0.f;
    for (int o = 0; o < DV; ++o) {
      const float go = Gs[r * DV + o];
      if (m0 < N) d0 = fmaf(go, Vs[m0 * DV + o], d0);
      if (m1 < N) d1 = fmaf(go, Vs[m1 * DV + o], d1);
    }
    const float rs = wave_sum(p0 * d0 + p1 * d1);
    const float t0 = p0 * (d0 - rs), t1 = p1 * (d1 - rs);
    WAVE_SYNC();                   // every lane has read its P entries of the row before the row is overwritten with dS
    if (m0 < N) {
      Ps[r * N + m0] = t0;
      dScg[(long)(n0 + r) * N + m0] = t0;
      const float w = Wr[r * N + m0];
      dwg[(((long)c * N + n0 + r) * N + m0) * G + g] = w >= 1e-6f ? t0 / w : 0.f;
    }
    if (m1 < N) {
      Ps[r * N + m1] = t1;
      dScg[(long)(n0 + r) * N + m1] = t1;
      const float w = Wr[r * N + m1];
      dwg[(((long)c * N + n0 + r) * N + m1) * G + g] = w >= 1e-6f ? t1 / w : 0.f;
    }
  }
  __syncthreads();
  for (int e = tid; e < nr * DQ; e += 256) {   // dQ: lanes along d (Ks rows conflict-free, the dS entry is a broadcast)
    const int r = e / DQ, d = e - r * DQ;
    float a = 0.f;
    for (int m = 0; m < N; ++m) a = fmaf(Ps[r * N + m], Ks[m * KP + d], a);
    dq[((long)c * N + n0 + r) * ldq + (long)g * DQ + d] = a * scale;
  }
}

__global__ __launch_bounds__(256) void relation_attention_bwd_cols_kernel(const float* __restrict__ q, const float* __restrict__ P,
                                                                          const float* __restrict__ dS, const float* __restrict__ dout,
                                                                          int C, int N, int G, int DQ, int DV, float scale,
                                                                          float* __restrict__ dk, float* __restrict__ dv) {
  extern __shared__ float lds[];
  const int g = blockIdx.x, c = blockIdx.y, m0 = blockIdx.z * RA_RB, tid = threadIdx.x;
  const int nc = min(RA_RB, N - m0);
  float* Qs = lds;                 // [N][DQ]
  float* Gs = Qs + N * DQ;         // [N][DV]      dOut[n, c, g, :]
  float* Dc = Gs + N * DV;         // [N][RA_RB]   dS columns of the chunk
  float* Pc = Dc + N * RA_RB;      // [N][RA_RB]   P columns of the chunk
  const long ldq = (long)G * DQ, ldv = (long)G * DV;
  const float* Pcg = P + ((long)c * G + g) * N * N;
  const float* dScg = dS + ((long)c * G + g) * N * N;
  for (int e = tid; e < N * DQ; e += 256) {
    const int n = e / DQ, d = e - n * DQ;
    Qs[e] = q[((long)c * N + n) * ldq + (long)g * DQ + d];
  }
  for (int e = tid; e < N * DV; e += 256) {
    const int n = e / DV, o = e - n * DV;
    Gs[e] = dout[((long)n * C + c) * ldv + (long)g * DV + o];
  }
  for (int e = tid; e < N * RA_RB; e += 256) {
    const int n = e / RA_RB, j = e - n * RA_RB;
    const bool ok = j < nc;
    Dc[e] = ok ? dScg[(long)n * N + m0 + j] : 0.f;
    Pc[e] = ok ? Pcg[(long)n * N + m0 + j] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < nc * DQ; e += 256) {   // dK: lanes along d
    const int j = e / DQ, d = e - j * DQ;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a = fmaf(Dc[n * RA_RB + j], Qs[n * DQ + d], a);
    dk[((long)c * N + m0 + j) * ldq + (long)g * DQ + d] = a * scale;
  }
  for (int e = tid; e < nc * DV; e += 256) {
    const int j = e / DV, o = e - j * DV;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a = fmaf(Pc[n * RA_RB + j], Gs[n * DV + o], a);
    dv[((long)c * N + m0 + j) * ldv + (long)g * DV + o] = a;
  }
}

// ---------------------------------------------------------------- CIAM, forward
// x [n][C][HW] (NCHW-dense instances of all (image, class) groups, `grp[i]` = group id, sorted so that a group is a
// contiguous run -- the kernels only need equality):
//   E[c][i][j] = <x[i,c,:], x[j,c,:]>   within a group;   M[i][j] = mean_c (max_j' E[c][i][j'] - E[c][i][j])
//   A[i][:] = softmax_j M[i][j] over the group;   out[i] = gamma * sum_j A[i][j] x[j] + x[i]
// One workgroup per instance i: the row of E for every channel lives in LDS ([C][nj] with nj = the group's size), x[i] in
// LDS; the group's x[j] stream through L2 twice (energies, then the mix).  Saved for the backward: A (dense n x n, zero
// outside the group) and the per-channel arg-max J[c][i].
// the contiguous run of instances with instance i's group id: every thread looks at a strided share of the ids (one round of
// loads; a scan outwards from i would be a chain of up to n dependent global loads), [lo, hi) by LDS atomics.  Ends with a
// barrier.
__device__ __forceinline__ void group_bounds(const int64_t* __restrict__ grp, int n, int i, int* s_lo, int* s_hi) {
  if (threadIdx.x == 0) {
    *s_lo = i;
    *s_hi = i + 1;
  }
  __syncthreads();
  const int64_t gi = grp[i];
  int lo = i, hi = i + 1;
  for (int j = threadIdx.x; j < n; j += 256)
    if (grp[j] == gi) {
      lo = min(lo, j);
      hi = max(hi, j + 1);
    }
  if (lo < i) atomicMin(s_lo, lo);
  if (hi > i + 1) atomicMax(s_hi, hi);
  __syncthreads();
}

constexpr int CI_MAXG = 512;     // instances of one (image, class) group
constexpr int CI_MAXC = 16;

__global__ __launch_bounds__(256) void ciam_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ grp, int n, int C,
                                                       int HW, const float* __restrict__ gamma, float* __restrict__ A,
                                                       int* __restrict__ J, float* __restrict__ out) {
  extern __shared__ float lds[];
  const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ int s_lo, s_hi;
  group_bounds(grp, n, i, &s_lo, &s_hi);
  float* xi = lds;                      // [C][HW]
  const int CH = C * HW;
  for (int e = tid; e < CH; e += 256) xi[e] = x[(long)i * CH + e];
  __syncthreads();
  const int lo = s_lo, nj = s_hi - s_lo;
  float* E = xi + CH;                   // [C][nj]
  float* Mr = E + C * nj;               // [nj]
  // energies: one thread per (c, j) pair walks the HW pixels (consecutive lanes = consecutive j of one channel: x[i, c, h] is an
  // LDS broadcast, every lane streams its own 4 HW-byte run of x[j, c, :]; no cross-lane reduction per pair)
  for (int t = tid; t < C * nj; t += 256) {
    const int c = t / nj, j = t - c * nj;
    const float* xj = x + (long)(lo + j) * CH + (long)c * HW;
    const float* xc = xi + c * HW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int h = 0;
    if ((HW & 3) == 0) {
      for (; h < HW; h += 4) {
        const float4 u = *(const float4*)(xj + h);
        a0 = fmaf(xc[h], u.x, a0);
        a1 = fmaf(xc[h + 1], u.y, a1);
        a2 = fmaf(xc[h + 2], u.z, a2);
        a3 = fmaf(xc[h + 3], u.w, a3);
      }
    }
    for (; h < HW; ++h) a0 = fmaf(xc[h], xj[h], a0);
    E[c * nj + j] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  // per channel: max over j and its FIRST index (what torch.max's gradient follows on the device is one index; see the backward)
  __shared__ float s_mx[CI_MAXC];
  for (int c = wave; c < C; c += 4) {
    float best = -INFINITY;
    int bj = 0x7fffffff;
    for (int j = lane; j < nj; j += 64) {
      const float e = E[c * nj + j];
      if (e > best) { best = e; bj = j; }
    }
    const float mx = wave_max(best);
    int cand = best == mx ? bj : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (lane == 0) {
      s_mx[c] = mx;
      J[(long)c * n + i] = lo + cand;
    }
  }
  __syncthreads();
  const float invC = 1.f / (float)C;
  for (int j = tid; j < nj; j += 256) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += s_mx[c] - E[c * nj + j];
    Mr[j] = a * invC;
  }
  __syncthreads();
  // softmax over the group (block-wide: nj may exceed a wave)
  __shared__ float s_red[4];
  float mx = -INFINITY;
  for (int j = tid; j < nj; j += 256) mx = fmaxf(mx, Mr[j]);
  mx = wave_max(mx);
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float sm = 0.f;
  for (int j = tid; j < nj; j += 256) {
    const float e = expf(Mr[j] - mx);
    Mr[j] = e;
    sm += e;
  }
  sm = wave_sum(sm);
  if (lane == 0) s_red[wave] = sm;
  __syncthreads();
  const float inv = 1.f / (s_red[0] + s_red[1] + s_red[2] + s_red[3]);
  for (int j = tid; j < n; j += 256) {
    const int jj = j - lo;
    const float a = (jj >= 0 && jj < nj) ? Mr[jj] * inv : 0.f;
    A[(long)i * n + j] = a;
    if (jj >= 0 && jj < nj) Mr[jj] = a;
  }
  __syncthreads();
  const float gm = gamma[0];
  for (int e = tid; e < CH; e += 256) {
    float a = 0.f;
    for (int j = 0; j < nj; ++j) a = fmaf(Mr[j], x[(long)(lo + j) * CH + e], a);
    out[(long)i * CH + e] = gm * a + xi[e];
  }
}

// ---------------------------------------------------------------- CIAM, backward
// dOut -> dx, dgamma.  With O = A X (the mix), out = gamma O + x:
//   dgamma = <dOut, O>;  dO = gamma dOut;  dA[i][j] = <dO[i], x[j]>;  dM = A (dA - sum_j A dA);
//   dE[c][i][j] = (dM_rowsum[i] [j == J[c][i]] - dM[i][j]) / C;
//   dx[i] = dOut[i] + sum_j A[j][i] dO[j]  (mix)  +  per channel sum_j (dE[c][i][j] + dE[c][j][i]) x[j,c,:]  (energies)
// pass 1 (one workgroup per i): dM row -> T[i][j] = dM[i][j], R[i] = sum_j dM[i][j]; dgamma by one atomic per workgroup
// pass 2 (one workgroup per i): dx[i] from T, R, J, A (all n x n or C x n, L2-resident)
__global__ __launch_bounds__(256) void ciam_bwd_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ grp, int n, int C,
                                                            int HW, const float* __restrict__ gamma, const float* __restrict__ A,
                                                            const float* __restrict__ dout, float* __restrict__ T,
                                                            float* __restrict__ R, float* __restrict__ dgamma) {
  extern __shared__ float lds[];
  const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ int s_lo, s_hi;
  __shared__ float s_red[4];
  group_bounds(grp, n, i, &s_lo, &s_hi);
  const int CH = C * HW;
  float* gi_ = lds;                      // [CH] dOut[i]
  for (int e = tid; e < CH; e += 256) gi_[e] = dout[(long)i * CH + e];
  __syncthreads();
  const int lo = s_lo, nj = s_hi - s_lo;
  float* dA = gi_ + CH;                  // [nj]  <dOut[i], x[j]>  (gamma applied below)
  for (int j = wave; j < nj; j += 4) {
    const float* xj = x + (long)(lo + j) * CH;
    float a = 0.f;
    for (int e = lane; e < CH; e += 64) a = fmaf(gi_[e], xj[e], a);
    a = wave_sum(a);
    if (lane == 0) dA[j] = a;
  }
  __syncthreads();
  // <dOut[i], O[i]> = sum_j A[i][j] dA[j]  (O = A X): dgamma's share of this row, and the softmax's row term
  float s = 0.f;
  for (int j = tid; j < nj; j += 256) s = fmaf(A[(long)i * n + lo + j], dA[j], s);
  s = wave_sum(s);
  if (lane == 0) s_red[wave] = s;
  __syncthreads();
  const float dot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const float gm = gamma[0];
  if (tid == 0) atomicAdd(dgamma, dot);
  __syncthreads();
  float rs = 0.f;
  for (int j = tid; j < n; j += 256) {
    const int jj = j - lo;
    float t = 0.f;
    if (jj >= 0 && jj < nj) t = gm * A[(long)i * n + j] * (dA[jj] - dot);
    T[(long)i * n + j] = t;
    rs += t;
  }
  rs = wave_sum(rs);
  if (lane == 0) s_red[wave] = rs;
  __syncthreads();
  if (tid == 0) R[i] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ __launch_bounds__(256) void ciam_bwd_dx_kernel(const float* __restrict__ x, const int64_t* __restrict__ grp, int n, int C,
                                                          int HW, const float* __restrict__ gamma, const float* __restrict__ A,
                                                          const int* __restrict__ J, const float* __restrict__ T,
                                                          const float* __restrict__ R, const float* __restrict__ dout,
                                                          float* __restrict__ dx) {
  extern __shared__ float lds[];
  const int i = blockIdx.x, tid = threadIdx.x;
  __shared__ int s_lo, s_hi;
  group_bounds(grp, n, i, &s_lo, &s_hi);
  __syncthreads();
  const int lo = s_lo, nj = s_hi - s_lo;
  const int CH = C * HW;
  float* W = lds;                        // [C][nj]  coefficient of x[j, c, :] in dx[i, c, :] from the energies
  float* Am = W + C * nj;                // [nj]     gamma A[j][i]
  const float invC = 1.f / (float)C, gm = gamma[0];
  for (int t = tid; t < C * nj; t += 256) {
    const int c = t / nj, j = lo + (t - c * nj);
    // dE[c][i][j] + dE[c][j][i]
    float w = -(T[(long)i * n + j] + T[(long)j * n + i]);
    if (J[(long)c * n + i] == j) w += R[i];
    if (J[(long)c * n + j] == i) w += R[j];
    W[t] = w * invC;
  }
  for (int j = tid; j < nj; j += 256) Am[j] = gm * A[(long)(lo + j) * n + i];
  __syncthreads();
  for (int e = tid; e < CH; e += 256) {
    const int c = e / HW;
    float a = dout[(long)i * CH + e];
    for (int j = 0; j < nj; ++j) {
      a = fmaf(Am[j], dout[(long)(lo + j) * CH + e], a);
      a = fmaf(W[c * nj + j], x[(long)(lo + j) * CH + e], a);
    }
    dx[(long)i * CH + e] = a;
  }
}

}  // namespace

static int ra_set_lds(const void* kern, size_t lds) {
  if (lds > 160 * 1024) return MMT_EINVAL;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

extern "C" int mmt_relation_attention_fwd(const float* q, const float* k, const float* wg, const float* v, const float* bias, int C,
                                          int N, int G, int DQ, int DV, int topk, float scale, float* P, float* out, void* stream) {
  if (!q || !k || !wg || !v || !bias || !P || !out || C < 1 || N < 1 || N > RA_MAXN || G < 1 || DQ < 1 || DQ > 128 || DV < 1 ||
      DV > RA_MAXDV || topk < 1)
    return MMT_EINVAL;
  const size_t lds = sizeof(float) * ((size_t)N * (DQ + 1) + (size_t)N * DV + RA_RB * DQ + RA_RB * (size_t)N + 4 * 128);
  if (int e = ra_set_lds((const void*)relation_attention_fwd_kernel, lds)) return e;
  hipLaunchKernelGGL(relation_attention_fwd_kernel, dim3(G, C, (N + RA_RB - 1) / RA_RB), dim3(256), lds, (hipStream_t)stream, q, k, wg,
                     v, bias, C, N, G, DQ, DV, topk < N ? topk : N, scale, P, out);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_relation_attention_bwd(const float* q, const float* k, const float* wg, const float* v, const float* P,
                                          const float* dout, int C, int N, int G, int DQ, int DV, float scale, float* dS, float* dq,
                                          float* dk, float* dwg, float* dv, void* stream) {
  if (!q || !k || !wg || !v || !P || !dout || !dS || !dq || !dk || !dwg || !dv || C < 1 || N < 1 || N > RA_MAXN || G < 1 || DQ < 1 ||
      DQ > 128 || DV < 1 || DV > RA_MAXDV)
    return MMT_EINVAL;
  const dim3 grid(G, C, (N + RA_RB - 1) / RA_RB);
  const size_t lds1 = sizeof(float) * ((size_t)N * (DQ + 1) + (size_t)N * DV + RA_RB * DV + 2 * RA_RB * (size_t)N);
  const size_t lds2 = sizeof(float) * ((size_t)N * DQ + (size_t)N * DV + 2 * RA_RB * (size_t)N);
  if (int e = ra_set_lds((const void*)relation_attention_bwd_rows_kernel, lds1)) return e;
  if (int e = ra_set_lds((const void*)relation_attention_bwd_cols_kernel, lds2)) return e;
  hipLaunchKernelGGL(relation_attention_bwd_rows_kernel, grid, dim3(256), lds1, (hipStream_t)stream, k, wg, v, P, dout, C, N, G, DQ, DV,
                     scale, dS, dq, dwg);
  MMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(relation_attention_bwd_cols_kernel, grid, dim3(256), lds2, (hipStream_t)stream, q, P, dS, dout, C, N, G, DQ, DV,
                     scale, dk, dv);
  MMT_LAUNCH_CHECK();
  return 0;
}

static size_t ciam_lds(int C, int HW, int maxg) { return sizeof(float) * ((size_t)C * HW + (size_t)(C + 1) * maxg); }

extern "C" int mmt_ciam_fwd(const float* x, const int64_t* group, int n, int C, int HW, int max_group, const float* gamma, float* A,
                            int* J, float* out, void* stream) {
  if (!x || !group || !gamma || !A || !J || !out || n < 1 || C < 1 || C > CI_MAXC || HW < 1 || max_group < 1 || max_group > CI_MAXG)
    return MMT_EINVAL;
  const size_t lds = ciam_lds(C, HW, max_group);
  if (lds > 64 * 1024) return MMT_EINVAL;
  hipLaunchKernelGGL(ciam_fwd_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, x, group, n, C, HW, gamma, A, J, out);
  MMT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmt_ciam_bwd(const float* x, const int64_t* group, int n, int C, int HW, int max_group, const float* gamma,
                            const float* A, const int* J, const float* dout, float* T, float* R, float* dx, float* dgamma,
                            void* stream) {
  if (!x || !group || !gamma || !A || !J || !dout || !T || !R || !dx || !dgamma || n < 1 || C < 1 || C > CI_MAXC || HW < 1 ||
      max_group < 1 || max_group > CI_MAXG)
    return MMT_EINVAL;
  const size_t lds1 = sizeof(float) * ((size_t)C * HW + max_group), lds2 = sizeof(float) * ((size_t)(C + 1) * max_group);
  if (lds1 > 64 * 1024 || lds2 > 64 * 1024) return MMT_EINVAL;
  if (hipMemsetAsync(dgamma, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return MMT_EINVAL;
  hipLaunchKernelGGL(ciam_bwd_rows_kernel, dim3(n), dim3(256), lds1, (hipStream_t)stream, x, group, n, C, HW, gamma, A, dout, T, R,
                     dgamma);
  MMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(ciam_bwd_dx_kernel, dim3(n), dim3(256), lds2, (hipStream_t)stream, x, group, n, C, HW, gamma, A, J, T, R, dout,
                     dx);
  MMT_LAUNCH_CHECK();
  return 0;
}
