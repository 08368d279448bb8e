// L2 -> LDS fill rate per CU by data path (no arithmetic): (0) global_load_lds_dwordx4 into a 3-stage ring with counted waits and a
// barrier per step, (1) global_load_dwordx4 -> VGPR -> ds_write_b128, two steps of loads in flight.  A step = 8 KB per block of
// 256 threads; source = 128 rows x ROWB bytes per block (16 MB over the chip: L2 resident), 64 B per row per step.
// build: hipcc --offload-arch=gfx950 -O3 -o ldsfill ldsfill.hip ; run: ./ldsfill [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWB = 512, ROWS = 128, STEPS_PER_PASS = ROWB / 64;

template <int MODE, int CONTIG>
__global__ __launch_bounds__(256, 2) void fill(const char* __restrict__ src, float* __restrict__ out, int steps) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + (size_t)blockIdx.x * ROWS * ROWB;
  // instruction i of wave w covers rows (w + 4 i) * 16 + (lane >> 2), chunk lane & 3 (CONTIG: 1 KB of consecutive bytes instead)
  const char* g[2];
  for (int i = 0; i < 2; i++) {
    const int row = (wave + 4 * i) * 16 + (lane >> 2);
    g[i] = CONTIG ? base + (wave + 4 * i) * 1024 + lane * 16 : base + (size_t)row * ROWB + (lane & 3) * 16;
  }
  float acc = 0.f;
  if (MODE == 0) {
    auto issue = [&](int s, int stage) {
      const int off = CONTIG ? (s % STEPS_PER_PASS) * 8192 : (s % STEPS_PER_PASS) * 64;
      for (int i = 0; i < 2; i++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g[i] + off),
                                         (lds_ptr_t)(ring + stage * 8192 + (wave + 4 * i) * 1024), 16, 0, 0);
    };
    issue(0, 0); issue(1, 1);
    int stage = 0;
    for (int s = 0; s < steps; s++) {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int s2 = stage == 0 ? 2 : stage - 1;  // free stage
      issue(s + 2, s2);
      acc += *(const float*)(ring + stage * 8192 + tid * 16);
      stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    f32x4 r[2][2];
    auto load = [&](int s, f32x4 (&v)[2]) {
      const int off = CONTIG ? (s % STEPS_PER_PASS) * 8192 : (s % STEPS_PER_PASS) * 64;
      for (int i = 0; i < 2; i++) v[i] = *(const f32x4*)(g[i] + off);
    };
    load(0, r[0]); load(1, r[1]);
    for (int s = 0; s < steps; s += 2) {
      for (int h = 0; h < 2; h++) {
        char* st = ring + ((s + h) & 1) * 8192;
        for (int i = 0; i < 2; i++) *(f32x4*)(st + (wave + 4 * i) * 1024 + lane * 16) = r[h][i];
        load(s + h + 2, r[h]);
        __syncthreads();
        acc += *(const float*)(st + tid * 16);
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int CONTIG>
static void run(const char* name, const char* src, float* out, int blocks) {
  const int steps = 8192;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<MODE, CONTIG>), dim3(blocks), dim3(256), 3 * 8192, 0, src, out, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-28s blocks %4d: %.3f ms  %.1f GB/s per CU, chip %.2f TB/s\n", name, blocks, ms,
                    8192.0 * steps * (blocks / 256.0) / ms / 1e6, 8192.0 * steps * blocks / ms / 1e9);
  }
}
int main(int argc, char** argv) {
  char* src; float* out;
  const size_t bytes = (size_t)1024 * ROWS * ROWB + (1 << 20);
  hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&out, 64);
  for (int blocks : {256, 512}) {
    run<0, 0>("dma, 64 B x 16 rows", src, out, blocks);
    run<0, 1>("dma, 1 KB contiguous", src, out, blocks);
    run<1, 0>("vgpr+ds_write, 64 B rows", src, out, blocks);
    run<1, 1>("vgpr+ds_write, contiguous", src, out, blocks);
  }
  return 0;
}
