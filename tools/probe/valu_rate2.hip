// Does a wave's own VALU work overlap its bf16 MFMAs?  Slots of one MFMA (dependent chain) + NV independent VALU
// instructions of several kinds; accumulator in VGPRs or in AGPRs.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

enum { V_SUB, V_CVT, V_EXP, V_ACCRD, V_LSHL };

template <int KIND, int NV, bool AGPR>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, long long* cyc) {
  float v[8], x = in[threadIdx.x];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
  f32x16 acc = (f32x16)(0.f), acc2 = (f32x16)(1.f);
  u32x4 a, b;
  for (int j = 0; j < 4; ++j) { a[j] = __float_as_uint(in[threadIdx.x + j]); b[j] = __float_as_uint(in[threadIdx.x + 8 + j]); }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        if (KIND == V_SUB) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[q & 7]) : "v"(x));
        if (KIND == V_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %0" : "+v"(v[q & 7]) : "v"(x));
        if (KIND == V_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[q & 7]));
        if (KIND == V_LSHL) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[q & 7]));
        if (KIND == V_ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[q & 7]) : "a"(acc2[q & 15]));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = x;
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int j = 0; j < 16; ++j) r += acc[j] + acc2[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static float *o, *in; static long long* c;
template <int KIND, int NV, bool AGPR> double run() {
  hipLaunchKernelGGL((k<KIND, NV, AGPR>), dim3(256), dim3(256), 0, 0, o, in, 2000, c);
  (void)hipDeviceSynchronize();
  long long h[256]; (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  return s / 256 / (2000.0 * 16);
}
template <int KIND, bool AGPR> void row(const char* n) {
  printf("%-18s acc in %s: cycles per slot with 0/2/4/6/8/12 VALU: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f\n", n, AGPR ? "AGPR" : "VGPR",
         run<KIND, 0, AGPR>(), run<KIND, 2, AGPR>(), run<KIND, 4, AGPR>(), run<KIND, 6, AGPR>(), run<KIND, 8, AGPR>(), run<KIND, 12, AGPR>());
}
int main() {
  (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&c, 256 * 8);
  (void)hipMemset(in, 0, 4096 * 4);
  row<V_SUB, false>("v_sub_f32"); row<V_SUB, true>("v_sub_f32");
  row<V_CVT, true>("v_cvt_pk_bf16_f32"); row<V_LSHL, true>("v_lshlrev_b32");
  row<V_EXP, true>("v_exp_f32"); row<V_ACCRD, true>("v_accvgpr_read");
  return 0;
}
