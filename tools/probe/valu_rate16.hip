// How many of a wave's own plain vector instructions hide behind one fp16 matrix instruction, by shape: cycles per
// (1 MFMA + N independent v_sub_f32), one wave per SIMD, for v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16
// (two of the latter do the work of one of the former).  Independent accumulators (4 in rotation).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/valu_rate16.hip -o tools/probe/valu_rate16
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int SHAPE, int NV>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, long long* cyc) {
  float v[8], x = in[threadIdx.x];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
  f32x16 A[4]; f32x4 C[4];
  for (int i = 0; i < 4; ++i) { A[i] = (f32x16)(0.f); C[i] = (f32x4)(0.f); }
  h16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)in[threadIdx.x + j]; b[j] = (_Float16)in[threadIdx.x + 8 + j]; }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int n = 0; n < NV; ++n) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[n & 7]) : "v"(x));
      if (SHAPE == 32) A[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, A[s & 3], 0, 0, 0);
      else C[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, C[s & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 16; ++j) r += A[i][j]; for (int j = 0; j < 4; ++j) r += C[i][j]; }
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int SHAPE, int NV> double run(float* o, float* in, long long* c) {
  hipLaunchKernelGGL((k<SHAPE, NV>), dim3(256), dim3(256), 0, 0, o, in, 2000, c);
  (void)hipDeviceSynchronize();
  long long h[256]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  return s / 256 / (2000.0 * 16);
}
int main() {
  float *o, *in; long long* c;
  (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&c, 256 * 8);
  (void)hipMemset(in, 0, 4096 * 4);
  printf("s_memtime ticks per (1 MFMA + N v_sub_f32)\n N   32x32x16  16x16x32\n");
  printf(" 0   %7.1f  %7.1f\n", run<32, 0>(o, in, c), run<16, 0>(o, in, c));
  printf(" 1   %7.1f  %7.1f\n", run<32, 1>(o, in, c), run<16, 1>(o, in, c));
  printf(" 2   %7.1f  %7.1f\n", run<32, 2>(o, in, c), run<16, 2>(o, in, c));
  printf(" 3   %7.1f  %7.1f\n", run<32, 3>(o, in, c), run<16, 3>(o, in, c));
  printf(" 4   %7.1f  %7.1f\n", run<32, 4>(o, in, c), run<16, 4>(o, in, c));
  printf(" 6   %7.1f  %7.1f\n", run<32, 6>(o, in, c), run<16, 6>(o, in, c));
  printf(" 8   %7.1f  %7.1f\n", run<32, 8>(o, in, c), run<16, 8>(o, in, c));
  printf("12   %7.1f  %7.1f\n", run<32, 12>(o, in, c), run<16, 12>(o, in, c));
  return 0;
}
