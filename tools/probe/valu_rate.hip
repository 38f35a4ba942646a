// Issue rate of plain VALU instructions from ONE wave per SIMD on gfx950: independent destinations vs a dependent chain,
// alone and between bf16 MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, long long* cyc) {
  float v[8], x = in[threadIdx.x], y = in[threadIdx.x + 1];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
  f32x16 acc = (f32x16)(0.f);
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[threadIdx.x + j]; b[j] = (__bf16)in[threadIdx.x + 8 + j]; }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (MODE == 0 || MODE == 2) {  // 8 independent
        asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                     "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(x));
      }
      if (MODE == 1 || MODE == 3) {  // 8 dependent
        asm volatile("v_sub_f32 %0, %1, %0\n v_sub_f32 %0, %1, %0\n v_sub_f32 %0, %1, %0\n v_sub_f32 %0, %1, %0\n"
                     "v_sub_f32 %0, %1, %0\n v_sub_f32 %0, %1, %0\n v_sub_f32 %0, %1, %0\n v_sub_f32 %0, %1, %0" : "+v"(y) : "v"(x));
      }
      if (MODE >= 2) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = y;
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int j = 0; j < 16; ++j) r += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> double run(float* o, float* in, long long* c) {
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, o, in, 2000, c);
  (void)hipDeviceSynchronize();
  long long h[256]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  return s / 256 / (2000.0 * 16);
}
int main() {
  float *o, *in; long long* c;
  (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&c, 256 * 8);
  (void)hipMemset(in, 0, 4096 * 4);
  printf("cycles per group of 8 v_sub_f32: independent %.1f, dependent chain %.1f\n", run<0>(o, in, c), run<1>(o, in, c));
  printf("cycles per (8 v_sub_f32 + 1 bf16 MFMA): independent %.1f, dependent %.1f\n", run<2>(o, in, c), run<3>(o, in, c));
  return 0;
}
