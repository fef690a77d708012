// Cost of s_barrier per iteration for 4- and 8-wave workgroups (one workgroup per CU, 256 workgroups), gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int n, unsigned* out) {
  unsigned acc = 0;
  for (int i = 0; i < n; ++i) {
    asm volatile("s_barrier" ::: "memory");
    acc += i;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (threadIdx.x == 0 && acc == 0x12345) out[0] = acc;
}
int main() {
  unsigned* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nw : {4, 8, 16}) for (int wg : {256, 512}) {
    const int n = 20000;
    k<<<wg, nw * 64>>>(100, d);
    hipEventRecord(e0); k<<<wg, nw * 64>>>(n, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%2d waves/workgroup, %3d workgroups: %.1f ns per iteration (2 barriers)\n", nw, wg, ms * 1e6 / n);
  }
}
