// Does a launch with hipExtAnyOrderLaunch (AQL barrier bit clear) start while its predecessor on the SAME stream is still running on gfx950,
// and what is a dependent chain worth when the dependency moves from the kernel boundary into per-workgroup flags?
//   T1  overlap probe: A = 256 workgroups x ~200 us of spinning; B = one workgroup that stamps its start. B.start << A.end  =>  overlapped.
//   T2  back-to-back chain of N short kernels (every workgroup spins `work` us), ordered launches vs any-order launches (no dependencies at all).
//   T3  the same chain with REAL dependencies carried in flags: workgroup i of kernel k waits for flag[k-1][i] (relaxed sc1 poll, one lane, then an
//       agent-scope acquire, cdna_hip_programming.md Guideline 16), spins `work` us (+ a per-(k,i) skew), stores a line write-through and publishes
//       flag[k][i]. Ordered launches need no flags: they pay max-over-workgroups of every kernel plus the boundary.
// Built twice: as a program on the system HIP runtime (hipcc anyorder.hip -o anyorder) and as libanyorder.so loaded behind `import torch`
// (tools/anyorder_probe.py), because the product library runs on the runtime the PyTorch wheel bundles.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz
__device__ __forceinline__ void spin_ticks(unsigned long long t) { const unsigned long long t0 = rt(); while (rt() - t0 < t) __builtin_amdgcn_s_sleep(2); }

__global__ void long_k(unsigned long long* stamps, int ticks) {
  spin_ticks(ticks);
  if (threadIdx.x == 0) stamps[blockIdx.x] = rt();
}
__global__ void stamp_k(unsigned long long* out) { if (threadIdx.x == 0) out[0] = rt(); }

__global__ void work_k(int ticks, unsigned* sink) {
  spin_ticks(ticks);
  if (threadIdx.x == 12345) sink[0] = 1;
}

// chain step: wait (optional) -> spin -> publish (optional)
__global__ void chain_k(const unsigned* wait_flags, unsigned wait_epoch, unsigned* my_flags, unsigned epoch, int ticks, int skew_ticks, int kidx, float* payload, unsigned* timeouts) {
  const int b = blockIdx.x;
  if (wait_flags) {
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load((const gu32*)(wait_flags + b * 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != wait_epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 17)) { atomicAdd(timeouts, 1u); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  const unsigned h = (unsigned)(b * 2654435761u) ^ (unsigned)(kidx * 40503u);
  spin_ticks(ticks + (skew_ticks ? (int)(h % (unsigned)skew_ticks) : 0));
  if (my_flags) {
    // payload: one 16-B write-through store per thread, drained, then ONE lane publishes
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)payload, 0, 0x7ffffff0, 0x00020000);
    typedef unsigned u4v __attribute__((__vector_size__(16)));
    u4v v = {epoch, (unsigned)b, (unsigned)kidx, threadIdx.x};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (b * 256 + threadIdx.x) * 16, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((gu32*)(my_flags + b * 32), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename F> static float time_ms(F f, hipStream_t s, int reps = 5) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) { hipEventRecord(e0, s); f(); hipEventRecord(e1, s); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  hipEventDestroy(e0); hipEventDestroy(e1);
  return t[t.size() / 2];
}

extern "C" int anyorder_run(int verbose) {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long* stamps; CK(hipMalloc(&stamps, 4096 * 8));
  unsigned* sink; CK(hipMalloc(&sink, 64));
  int rtver = 0; hipRuntimeGetVersion(&rtver);
  printf("HIP runtime version %d\n", rtver);
  // ---- T1
  for (int flag = 0; flag < 2; ++flag) {
    CK(hipMemsetAsync(stamps, 0, 4096 * 8, s));
    CK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(long_k, dim3(256), dim3(256), 0, s, stamps, 20000);
    hipExtLaunchKernelGGL(stamp_k, dim3(1), dim3(64), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, stamps + 1024);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(1025);
    CK(hipMemcpy(h.data(), stamps, 1025 * 8, hipMemcpyDeviceToHost));
    unsigned long long amax = 0, amin = ~0ull;
    for (int i = 0; i < 256; ++i) { amax = std::max(amax, h[i]); amin = std::min(amin, h[i]); }
    printf("T1 flags=%d: B.start - A.last_end = %+.2f us (A first..last end spread %.2f us)  => %s\n", flag, ((double)h[1024] - (double)amax) / 100.0,
           (double)(amax - amin) / 100.0, h[1024] < amax ? "OVERLAPPED" : "serialised");
  }
  // ---- T2
  const int N = 200;
  for (int wg : {256, 640}) for (int work : {0, 500, 1500}) {
    float t[2];
    for (int flag = 0; flag < 2; ++flag)
      t[flag] = time_ms([&] { for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(work_k, dim3(wg), dim3(256), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, work, sink); }, s);
    printf("T2 %4d wgs, work %5.1f us: ordered %.2f us/launch, any-order %.2f us/launch\n", wg, work / 100.0, t[0] * 1e3 / N, t[1] * 1e3 / N);
  }
  // ---- T3
  {
    const int NW = 512;
    unsigned* flags; CK(hipMalloc(&flags, (size_t)2 * NW * 32 * 4)); CK(hipMemset(flags, 0, (size_t)2 * NW * 32 * 4));
    float* payload; CK(hipMalloc(&payload, (size_t)NW * 256 * 16));
    unsigned* tmo; CK(hipMalloc(&tmo, 64)); CK(hipMemset(tmo, 0, 64));
    unsigned epoch = 0;
    for (int wg : {256, 512}) for (int work : {500, 1500}) for (int skew : {0, 300}) {
      float t_ord = time_ms([&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(chain_k, dim3(wg), dim3(256), 0, s, (const unsigned*)nullptr, 0u, (unsigned*)nullptr, 0u, work, skew, i, payload, tmo); }, s);
      float t_pub = time_ms([&] { for (int i = 0; i < N; ++i) { ++epoch; hipLaunchKernelGGL(chain_k, dim3(wg), dim3(256), 0, s, (const unsigned*)nullptr, 0u, flags + (i & 1) * NW * 32, epoch, work, skew, i, payload, tmo); } }, s);
      float t_any = time_ms([&] {
        for (int i = 0; i < N; ++i) {
          ++epoch;
          const unsigned* wf = i ? flags + ((i - 1) & 1) * NW * 32 : nullptr;
          // first kernel of the chain is an ordered launch (everything before it must be complete); the rest may start early
          hipExtLaunchKernelGGL(chain_k, dim3(wg), dim3(256), 0, s, nullptr, nullptr, i ? hipExtAnyOrderLaunch : 0, wf, epoch - 1, flags + (i & 1) * NW * 32, epoch, work, skew, i, payload, tmo);
        } }, s);
      // NOTE: flag[k] is overwritten by kernel k+2 while kernel k+1 may still poll it for `epoch(k)` -- epochs are distinct per kernel and kernel k+2's
      // workgroup i publishes only after kernel k+1's workgroup i has seen epoch(k), so the two-deep flag ring is safe for this per-workgroup chain.
      unsigned h_tmo = 0; CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost));
      printf("T3 %3d wgs, work %4.1f us, skew 0..%3.1f us: ordered %.2f | ordered+publish %.2f | any-order+flags %.2f us/kernel  (timeouts %u)\n", wg, work / 100.0, skew / 100.0,
             t_ord * 1e3 / N, t_pub * 1e3 / N, t_any * 1e3 / N, h_tmo);
    }
    hipFree(flags); hipFree(payload); hipFree(tmo);
  }
  hipFree(stamps); hipFree(sink); hipStreamDestroy(s);
  return 0;
}

#ifdef STANDALONE
int main() { return anyorder_run(1); }
#endif
