// Do one wave's VALU instructions overlap ANOTHER wave's bf16 MFMAs on the same SIMD?  Workgroups of 8 waves (two per
// SIMD): waves 0-3 run a dependent MFMA chain, waves 4-7 a stream of independent v_sub_f32; each role is timed alone
// and with the other running.  Also: two waves per SIMD that each mix one MFMA with 8 VALU (the decoder's ratio).
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int MIX>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int it_m, int it_v, long long* cyc) {
  float v[8], x = in[threadIdx.x];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
  f32x16 acc = (f32x16)(0.f);
  u32x4 a, b;
  for (int j = 0; j < 4; ++j) { a[j] = __float_as_uint(in[threadIdx.x + j]); b[j] = __float_as_uint(in[threadIdx.x + 8 + j]); }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (MIX) {
    const int iters = wave < 4 ? it_m : it_v;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
        for (int q = 0; q < MIX; ++q) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[q & 7]) : "v"(x));
      }
    }
  } else if (wave < 4) {
    for (int it = 0; it < it_m; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
  } else {
    for (int it = 0; it < it_v; ++it) {
#pragma unroll
      for (int s = 0; s < 128; ++s) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[s & 7]) : "v"(x));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = x;
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int j = 0; j < 16; ++j) r += acc[j];
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
static float *o, *in; static long long* c;
template <int MIX> void run(const char* what, int it_m, int it_v) {
  hipLaunchKernelGGL((k<MIX>), dim3(256), dim3(512), 0, 0, o, in, it_m, it_v, c);
  (void)hipDeviceSynchronize();
  static long long h[256 * 8]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double sm = 0, sv = 0;
  for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += h[i * 8 + w];
  printf("%-58s waves 0-3: %8.0f cycles   waves 4-7: %8.0f cycles\n", what, sm / 1024, sv / 1024);
}
int main() {
  (void)hipMalloc(&o, 256 * 512 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&c, 256 * 8 * 8);
  (void)hipMemset(in, 0, 4096 * 4);
  run<0>("warm-up", 2000, 2000);
  run<0>("MFMA chain alone (2000 x 16 MFMA = 1.02M cycles at 32)", 2000, 0);
  run<0>("VALU stream alone (2000 x 128 v_sub = 1.02M cycles at 4)", 0, 2000);
  run<0>("both, one role per wave of a SIMD", 2000, 2000);
  run<8>("1 MFMA + 8 VALU per slot, one wave per SIMD", 2000, 0);
  run<8>("1 MFMA + 8 VALU per slot, two waves per SIMD", 2000, 2000);
  run<4>("1 MFMA + 4 VALU per slot, two waves per SIMD", 2000, 2000);
  run<2>("1 MFMA + 2 VALU per slot, two waves per SIMD", 2000, 2000);
  return 0;
}
