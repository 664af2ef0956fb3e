// Probe: which XCDs / CUs does a CU-masked stream run on?  (bit i of hipExtStreamCreateWithCUMask -> ?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
__global__ void where(uint32_t* out) {
  uint32_t xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // burn a little time so that all CUs in the mask get work
  float x = threadIdx.x;
  for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid + (x == 12345.f); }
}
int main() {
  const int NB = 4096;
  uint32_t* d; hipMalloc(&d, NB * 8);
  std::vector<uint32_t> h(NB * 2);
  for (int test = 0; test < 4; ++test) {
    uint32_t mask[8] = {0};
    if (test == 0) mask[0] = 0xFFFFFFFFu;                                   // bits 0..31
    if (test == 1) for (int i = 0; i < 256; i += 8) mask[i / 32] |= 1u << (i % 32);  // every 8th bit
    if (test == 2) for (int w = 0; w < 8; ++w) mask[w] = 0xFFFFFFFCu;        // all but bits 32w, 32w+1
    if (test == 3) for (int i = 16; i < 256; ++i) mask[i / 32] |= 1u << (i % 32);    // all but bits 0..15
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("create failed: %s\n", hipGetErrorString(e)); return 1; }
    hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
    int per_xcc[16] = {0};
    std::set<uint32_t> cus[16];
    for (int b = 0; b < NB; ++b) {
      const uint32_t xcc = h[2 * b] & 0xF, hw = h[2 * b + 1];
      per_xcc[xcc]++;
      cus[xcc].insert((hw >> 8) & 0xFFFF);   // CU_ID / SH_ID / SE_ID fields
    }
    printf("test %d:", test);
    for (int x = 0; x < 8; ++x) printf("  xcc%d: %d blk on %zu CUs", x, per_xcc[x], cus[x].size());
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
