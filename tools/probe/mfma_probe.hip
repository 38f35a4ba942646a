// Micro-probe: does VALU / LDS work issued between v_mfma_f32_32x32x2_f32 instructions of ONE wave per SIMD
// hide under the 64-cycle matrix instruction on gfx950?  Prints cycles per MFMA slot for several fillers.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_probe.hip -o tools/probe/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NV, int NL, int NACC, int NT>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters, long long* cyc) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x16)(0.f);
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
  float a = in[threadIdx.x], b = in[threadIdx.x + 64];
  const float* lp = lds + (threadIdx.x & 63);
  float ls = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      float lv[NL > 0 ? NL : 1];
#pragma unroll
      for (int q = 0; q < NL; ++q) lv[q] = lp[(s * NL + q) * 64 % 4032];
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
#pragma unroll
      for (int q = 0; q < NT; ++q) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]);
      acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s % NACC], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NL; ++q) ls += lv[q];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = ls;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NL, int NACC, int NT>
void run(const char* name, float* out, float* in, long long* cyc, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NV, NL, NACC, NT><<<blocks, 256>>>(out, in, 10, cyc);
  hipEventRecord(e0);
  probe<NV, NL, NACC, NT><<<blocks, 256>>>(out, in, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double slots = (double)iters * 32;
  printf("%-44s wall %.3f ms  memtime/slot %.1f  (wall cycles/slot @2.4GHz %.1f)\n", name, ms,
         (double)h[0] / slots, ms * 1e-3 * 2.4e9 / slots / ((blocks + 255) / 256));
}

int main() {
  float *out, *in; long long* cyc;
  const int blocks = 256;  // one workgroup per CU: one wave per SIMD
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 8192 * 4); hipMalloc(&cyc, blocks * 8);
  float* h = (float*)malloc(8192 * 4);
  for (int i = 0; i < 8192; ++i) h[i] = (float)rand() / RAND_MAX;
  hipMemcpy(in, h, 8192 * 4, hipMemcpyHostToDevice);
  run<0, 0, 1, 0>("chain, no filler", out, in, cyc, blocks);
  run<0, 0, 2, 0>("2 accumulators, no filler", out, in, cyc, blocks);
  run<2, 0, 1, 0>("chain + 2 v_fma", out, in, cyc, blocks);
  run<4, 0, 1, 0>("chain + 4 v_fma", out, in, cyc, blocks);
  run<8, 0, 1, 0>("chain + 8 v_fma", out, in, cyc, blocks);
  run<12, 0, 1, 0>("chain + 12 v_fma", out, in, cyc, blocks);
  run<16, 0, 1, 0>("chain + 16 v_fma", out, in, cyc, blocks);
  run<8, 0, 2, 0>("2 acc + 8 v_fma", out, in, cyc, blocks);
  run<0, 0, 1, 2>("chain + 2 v_exp", out, in, cyc, blocks);
  run<0, 0, 1, 4>("chain + 4 v_exp", out, in, cyc, blocks);
  run<0, 1, 1, 0>("chain + 1 ds_read_b32", out, in, cyc, blocks);
  run<0, 2, 1, 0>("chain + 2 ds_read_b32", out, in, cyc, blocks);
  run<0, 4, 1, 0>("chain + 4 ds_read_b32", out, in, cyc, blocks);
  run<4, 2, 1, 0>("chain + 4 v_fma + 2 ds_read", out, in, cyc, blocks);
  return 0;
}
