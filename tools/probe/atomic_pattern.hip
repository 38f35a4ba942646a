// Probe: what do the decoder's dz float atomics cost against the lane -> address map?  dz^T is [64 k][N] fp32; a chain
// wave adds a [64 k][32 n] tile per row tile and pair of groups.
//   pattern A (32x32 accumulators, dec_x3_64g): one instruction = 2 rows k x 32 consecutive n (128-byte runs)
//   pattern B (16x16 accumulators):             one instruction = 4 rows k x 16 consecutive n (64-byte runs)
// Same grid as the kernel's adds at the scale-up shape: 512 workgroups x 4 waves, every wave walks 512 of the 2048 row
// tiles, 32 adds per lane and tile = 8.6 GB of adds per launch.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/atomic_pattern.hip -o tools/probe/atomic_pattern && tools/probe/atomic_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int PAT>
__global__ __launch_bounds__(256) void adds(float* dzT, int N) {
  const int lane = threadIdx.x & 63, sw = threadIdx.x >> 6;
  const int ntiles = N / 32;
  const float v = 1.0f + lane;
  for (int tile = sw; tile < ntiles; tile += 4) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      int k, n;
      if (PAT == 0) { k = 32 * (r >> 4) + 8 * ((r >> 2) & 3) + 4 * (lane >> 5) + (r & 3); n = tile * 32 + (lane & 31); }
      else { k = 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3); n = tile * 32 + 16 * ((r >> 2) & 1) + (lane & 15); }
      atomicAdd(dzT + (int64_t)k * N + n, v);
    }
  }
}
int main() {
  const int N = 65536;
  float* d;
  hipMalloc(&d, (size_t)64 * N * 4);
  hipMemset(d, 0, (size_t)64 * N * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int pat = 0; pat < 2; ++pat) {
      hipEventRecord(e0);
      if (pat == 0) hipLaunchKernelGGL(adds<0>, dim3(512), dim3(256), 0, 0, d, N);
      else hipLaunchKernelGGL(adds<1>, dim3(512), dim3(256), 0, 0, d, N);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 512.0 * 2048 * 32 * 64 * 4;
      printf("pattern %c: %.3f ms  %.2f TB/s of adds\n", pat ? 'B' : 'A', ms, bytes / ms / 1e9);
    }
  return 0;
}
