// Census of HW_REG_XCC_ID against blockIdx % 8 for a grid that fills the chip several times over.
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/xcc_census.hip -o scripts/probes/xcc_census
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void census(int* hist /* [8][8] xcc x (block % 8) */, int* raw) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) {
    atomicAdd(&hist[(x & 7) * 8 + (blockIdx.x & 7)], 1);
    if (blockIdx.x < 16) raw[blockIdx.x] = (int)x;
  }
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);
}
int main() {
  int *h, *r;
  hipMalloc(&h, 64 * sizeof(int)); hipMalloc(&r, 16 * sizeof(int));
  hipMemset(h, 0, 64 * sizeof(int));
  hipLaunchKernelGGL(census, dim3(4096), dim3(256), 65536, 0, h, r);
  int hh[64], rr[16];
  hipMemcpy(hh, h, sizeof(hh), hipMemcpyDeviceToHost); hipMemcpy(rr, r, sizeof(rr), hipMemcpyDeviceToHost);
  printf("raw XCC_ID of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" 0x%x", rr[i]); printf("\n");
  printf("rows = XCC_ID & 7, cols = blockIdx %% 8 (4096 blocks, 64 KB LDS each)\n");
  for (int x = 0; x < 8; ++x) { for (int b = 0; b < 8; ++b) printf("%6d", hh[x * 8 + b]); printf("\n"); }
  return 0;
}
