// Which SIMD does wave k of a workgroup land on? (gfx9 HW_ID: SIMD_ID = bits 5:4, WAVE_ID = bits 3:0, CU_ID = bits 11:8)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
  for (int nw : {4, 8}) {
    unsigned* d; hipMalloc(&d, 64 * 16 * 4);
    k<<<4, nw * 64>>>(d);
    unsigned h[64]; hipMemcpy(h, d, 4 * nw * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) { printf("block %d (%d waves): ", b, nw); for (int w = 0; w < nw; ++w) printf("w%d->simd%u(cu%u) ", w, (h[b * nw + w] >> 4) & 3, (h[b * nw + w] >> 8) & 15); printf("\n"); }
    hipFree(d);
  }
}
