// Does kernarg preloading (first 16 dwords pushed into SGPRs by the command processor) remove the cold kernarg read from a launch?
// Build twice: plain and with -mllvm -amdgpu-kernarg-preload-count=16.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Rest { long a[24]; };
__global__ void scal_k(const long* p0, const long* p1, unsigned* out, int a, int b, int c, int d, int e, int f, int g, int h, int i, int j) {
  long s = (long)p0 + (long)p1 + a + b + c + d + e + f + g + h + i + j;
  if (s == 12345) out[0] = 1;
}
__global__ void mixed_k(const long* p0, const long* p1, unsigned* out, int a, int b, int c, int d, int e, int f, int g, int h, int i, int j, Rest r) {
  long s = (long)p0 + (long)p1 + a + b + c + d + e + f + g + h + i + j;
  if (s == 12345) { for (int k = 0; k < 24; ++k) s += r.a[k]; out[0] = (unsigned)s; }   // the struct is only read on a path never taken
}
int main() {
  unsigned* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  Rest r{}; const int n = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < n + 20; ++i) {
      if (i == 20) hipEventRecord(e0);
      if (mode) mixed_k<<<640, 256>>>((long*)d, (long*)d, d, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, r); else scal_k<<<640, 256>>>((long*)d, (long*)d, d, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("640 workgroups, %s: %.2f us per launch\n", mode ? "16 scalar dwords + 192-byte struct (unread)" : "16 scalar dwords", ms * 1e3 / n);
  }
}
