// ds_read_b64_tr_b16 on gfx950: what does lane l get?  (round 5: the layout question behind a plane-fed weight-gradient kernel)
//   hipcc --offload-arch=gfx950 -O3 trread.hip -o /tmp/trread && /tmp/trread
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, int row_stride) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
  __syncthreads();
  // lane l: 16-lane group g = l >> 4, i = l & 15 supplies the address of (row i >> 2, columns 4 (i & 3) ..) of the group's
  // [4 rows][16 columns] block; rows row_stride elements apart; group g's block starts 4 rows further down
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  const unsigned short* p = lds + (g * 4 + (i >> 2)) * row_stride + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; j++) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[8192], o[256], *di, *dout;
  for (int i = 0; i < 8192; i++) h[i] = (unsigned short)i;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rs : {16, 128}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, rs);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
      for (int j = 0; j < 4; j++) {
        const int g = l >> 4, i = l & 15, want = (g * 4 + j) * rs + i;   // column i of the block, row j
        bad += o[l * 4 + j] != want;
      }
    printf("row stride %d elements: lane 0 -> %d %d %d %d, lane 1 -> %d %d %d %d, lane 17 -> %d %d %d %d; 'column i of the [4][16] block' %s\n", rs,
           o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[68], o[69], o[70], o[71], bad ? "DOES NOT HOLD" : "holds for all 64 lanes");
  }
  return 0;
}
