// On a power-limited chip, which shape of the fp16 matrix instruction delivers more flops: v_mfma_f32_32x32x16_f16
// (8 passes, 32768 flops) or v_mfma_f32_16x16x32_f16 (4 passes, 16384 flops)?  All 1024 SIMDs run a chain over
// independent accumulators for ~30 ms on random operands; reports wall time per instruction and TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const unsigned* in, int iters, float* out) {
  u32x4 a[4], b[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 4; ++e) {
      a[j][e] = in[(threadIdx.x * 16 + j * 4 + e) & 4095];
      b[j][e] = in[(threadIdx.x * 16 + j * 4 + e + 2048) & 4095];
    }
  float r = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = (f32x16)(0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[(j + s) & 3]), "v"(b[(j + 2 * s) & 3]));
    }
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) r += acc[j][e];
  } else {
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4)(0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[(j + s) & 3]), "v"(b[(j + 2 * s) & 3]));
    }
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) r += acc[j][e];
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
  unsigned* in; float* o;
  (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&o, 1024 * 256 * 4);
  unsigned h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) { unsigned lo = 0x3800u | (rand() & 0x83ff), hi = 0x3800u | (rand() & 0x83ff); h[i] = lo | (hi << 16); }
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 100000;
  for (int rep = 0; rep < 3; ++rep)
    for (int shape = 0; shape < 2; ++shape) {
      (void)hipEventRecord(e0, 0);
      if (shape == 0) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, in, iters, o);
      else hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, in, iters, o);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double n = (shape == 0 ? 16.0 : 32.0) * iters, fl = shape == 0 ? 32768.0 : 16384.0;
      printf("%-26s rep %d: %6.2f ns per instruction, %.0f TFLOP/s\n", shape == 0 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", rep,
             ms * 1e6 / n, 1024 * fl / (ms * 1e6 / n) / 1e3);
    }
  return 0;
}
