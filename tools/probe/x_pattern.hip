// Probe: what does HBM give the two ways of walking X [4096 rows][6600 columns] (the MEG-like shape, 44 groups of 150)?
//   pattern A (dec_single today): workgroup = (group, row chunk), 64-row tiles, 64-column chunks: a wave's load
//     instruction covers 64 columns of ONE row (256-byte runs, rows 26 KB apart), 16 loads in flight per thread;
//   pattern B (a workgroup that owns rows): workgroup = 32 rows x a quarter of the groups, thread = column of the
//     current group, 32 loads in flight per thread: a load instruction covers the 150 columns of one row (600 bytes),
//     consecutive groups continue the same rows.
// Both only sum what they read.   hipcc -O3 --offload-arch=gfx950 tools/probe/x_pattern.hip -o tools/probe/x_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int B = 4096, G = 44, DG = 150, D = G * DG;
__global__ __launch_bounds__(256) void pat_a(const float* X, float* out, int nchunks) {
  const int g = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x, dd = tid & 63, rq = tid >> 6;
  float acc = 0.f;
  for (int tile = chunk; tile < B / 64; tile += nchunks)
    for (int c0 = 0; c0 < DG; c0 += 64) {
      float xv[16];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = tile * 64 + 16 * rq + rr, d = c0 + dd;
        xv[rr] = d < DG ? X[(int64_t)row * D + g * DG + d] : 0.f;
      }
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc += xv[rr];
    }
  if (acc == 12345.678f) out[0] = acc;
}
template <int R>
__global__ __launch_bounds__(256) void pat_b(const float* X, float* out, int gper) {
  const int rt = blockIdx.x, gq = blockIdx.y, tid = threadIdx.x;
  float acc = 0.f;
  for (int g = gq * gper; g < (gq + 1) * gper && g < G; ++g) {
    float xv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) xv[r] = tid < DG ? X[(int64_t)(rt * R + r) * D + g * DG + tid] : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) acc += xv[r];
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  float *X, *out;
  hipMalloc(&X, (size_t)B * D * 4);
  hipMalloc(&out, 4);
  hipMemset(X, 0, (size_t)B * D * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, (double)B * D * 4 / (ms / 20 * 1e-3) / 1e12);
  };
  time("A: (group, chunk) x 64-col chunks, 35 chunks", [&] { hipLaunchKernelGGL(pat_a, dim3(G, 35), dim3(256), 0, 0, X, out, 35); });
  time("A: 64 chunks (one tile each)", [&] { hipLaunchKernelGGL(pat_a, dim3(G, 64), dim3(256), 0, 0, X, out, 64); });
  time("B: 32 rows x 4 group quarters (512 wgs)", [&] { hipLaunchKernelGGL(pat_b<32>, dim3(B / 32, 4), dim3(256), 0, 0, X, out, 11); });
  time("B: 32 rows x 2 group halves (256 wgs)", [&] { hipLaunchKernelGGL(pat_b<32>, dim3(B / 32, 2), dim3(256), 0, 0, X, out, 22); });
  time("B: 16 rows x 4 quarters (1024 wgs)", [&] { hipLaunchKernelGGL(pat_b<16>, dim3(B / 16, 4), dim3(256), 0, 0, X, out, 11); });
  time("B: 16 rows x all groups (256 wgs)", [&] { hipLaunchKernelGGL(pat_b<16>, dim3(B / 16, 1), dim3(256), 0, 0, X, out, 44); });
  time("B: 32 rows x 11 (4 groups each, 1408 wgs)", [&] { hipLaunchKernelGGL(pat_b<32>, dim3(B / 32, 11), dim3(256), 0, 0, X, out, 4); });
  return 0;
}
