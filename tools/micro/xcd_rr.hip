// Two questions about kernel boundaries on MI355X (round 4):
//  (1) Dispatch: blocks b and b + 8 share an XCD -- but does block 0 of launch i+1 land on the same XCD as block 0 of launch i? How does the start move with the
//      grid size of the launch before it?
//  (2) L2 retention: do clean lines a kernel read stay in the XCD's L2 across a kernel boundary, i.e. is a re-read by the SAME XCD in the next launch faster than a
//      re-read by a different XCD (Infinity-Cache-served)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_rr.hip -o tools/micro/xcd_rr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }      // HW_REG_XCC_ID[3:0]

__global__ void who(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// block b reads part (b / 8) of slice ((b % 8) + rot) % 8: 16 B per lane, whole lines
__global__ __launch_bounds__(256) void rd(const uint4* x, size_t slice_vec, int parts, int rot, unsigned* sink) {
  const int sl = ((blockIdx.x & 7) + rot) & 7, part = blockIdx.x >> 3;
  const size_t per = slice_vec / parts;
  const uint4* p = x + (size_t)sl * slice_vec + (size_t)part * per;
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i < per; i += 256 * 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p[i + u * 256 < per ? i + u * 256 : i];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void fill(uint4* p, size_t n, unsigned v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v + 1, v + 2, v + 3);
}

int main() {
  int* d; hipMalloc(&d, 4096 * sizeof(int));
  std::vector<int> h(4096);
  const int grids[] = {480, 480, 485, 485, 512, 100, 100, 643, 640, 8, 9, 256, 255, 256};
  printf("# (1) dispatch: XCC id of block 0, and whether block b sits on (xcc0 + b) %% 8 for every b\n");
  int prev0 = -1, prevg = 0;
  for (int g : grids) {
    who<<<g, 64>>>(d);
    hipMemcpy(h.data(), d, g * sizeof(int), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < g; ++b) bad += ((h[b] - h[0]) & 7) != (b & 7);
    printf("grid %4d: block 0 on XCC %d (previous launch: block 0 on %d, grid %d -> predicted by a running pointer: %d), %d of %d blocks off the round-robin\n", g, h[0], prev0, prevg,
           prev0 < 0 ? -1 : (prev0 + prevg) & 7, bad, g);
    prev0 = h[0]; prevg = g;
  }
  // interleave another kernel type between two `who` launches
  unsigned* sink; hipMalloc(&sink, 64);
  const size_t slice_bytes = 3u << 20, slice_vec = slice_bytes / 16;
  uint4* x; hipMalloc(&x, 8 * slice_bytes);
  uint4* big; const size_t big_bytes = (size_t)600 << 20; hipMalloc(&big, big_bytes);
  fill<<<2048, 256>>>(x, 8 * slice_vec, 7);
  who<<<480, 64>>>(d); hipMemcpy(h.data(), d, 4, hipMemcpyDeviceToHost); const int a0 = h[0];
  rd<<<256, 256>>>(x, slice_vec, 32, 0, sink);
  who<<<480, 64>>>(d); hipMemcpy(h.data(), d, 4, hipMemcpyDeviceToHost);
  printf("who(480) -> block 0 on %d; rd(256 blocks); who(480) -> block 0 on %d\n", a0, h[0]);

  printf("# (2) L2 retention across a kernel boundary: 8 slices of 3 MiB, 256 blocks; block b reads slice ((b %% 8) + rot) %% 8\n");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timed = [&](int rot) {
    hipEventRecord(e0); rd<<<256, 256>>>(x, slice_vec, 32, rot, sink); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
  };
  for (int rep = 0; rep < 3; ++rep) {
    fill<<<2048, 256>>>(big, big_bytes / 16, rep);      // 600 MB written: L2s and Infinity Cache hold none of x
    hipDeviceSynchronize();
    const float t_cold = timed(0), t_same = timed(0), t_same2 = timed(0), t_rot = timed(1), t_rot2 = timed(1), t_back = timed(0);
    printf("rot 0 from HBM %.1f us | rot 0 again %.1f, %.1f | rot 1 (other XCD's slice: Infinity Cache) %.1f | rot 1 again %.1f | rot 0 %.1f\n", t_cold, t_same, t_same2, t_rot, t_rot2, t_back);
  }
  // with a different kernel in between (another grid size): is the placement still the same?
  for (int g : {480, 485, 100}) {
    timed(0); timed(0);
    who<<<g, 64>>>(d);
    const float t = timed(0);
    printf("rot 0 warm, then who(%d), then rot 0: %.1f us\n", g, t);
  }
  return 0;
}
