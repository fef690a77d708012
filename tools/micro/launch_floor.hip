// Back-to-back launch floor on one stream: empty kernel, and a kernel that reads a 256-byte by-value argument block (as gemm_f16_kernel does)
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { long a[32]; };
__global__ void empty_k(unsigned* out) { if (threadIdx.x == 12345) out[0] = 1; }
__global__ void args_k(Big b, unsigned* out) { long s = 0; for (int i = 0; i < 32; ++i) s += b.a[i]; if (s == 12345) out[0] = 1; }
int main() {
  unsigned* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  Big b{}; const int n = 2000;
  for (int wg : {1, 256, 640, 1280}) for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 20; ++i) { if (mode) args_k<<<wg, 256>>>(b, d); else empty_k<<<wg, 256>>>(d); }
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) { if (mode) args_k<<<wg, 256>>>(b, d); else empty_k<<<wg, 256>>>(d); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%4d workgroups x 256 threads, %s: %.2f us per launch\n", wg, mode ? "256-byte args" : "empty", ms * 1e3 / n);
  }
}
