// Does a chain of matrix instructions into ONE fp32 accumulator round to nearest or truncate?  Same-sign products
// (a constant operand against positive random ones) accumulated over n steps, against the exact sum in double.
// A rounding accumulator errs like sqrt(n) * 2^-25 with either sign; a truncating one like -n * 2^-24 / 2, always low.
// hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_accum_round.hip -o tools/probe/mfma_accum_round
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// mode 0: 16x16x32 f16; 1: 32x32x16 f16; 2: 32x32x16 bf16; 3: 32x32x2 f32
__global__ void probe(const float* vals, int n, int mode, float* out, double* exact) {
  const int lane = threadIdx.x;
  f4 acc4 = {0, 0, 0, 0};
  f16v acc16 = (f16v)(0.f);
  double ex = 0.0;   // exact value of element [0][0]... every element sees the same slots: A rows equal, B cols equal
  for (int s = 0; s < n; ++s) {
    // A[row][k] = vals[s*64 + k] for every row; B[k][col] = vals[s*64 + 32 + k] for every col: full-width products, all positive
    if (mode == 0) {
      h8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)vals[s * 64 + (lane >> 4) * 8 + e]; b[e] = (_Float16)vals[s * 64 + 32 + ((lane >> 4) * 8 + e) % 32]; }
      acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4, 0, 0, 0);
      if (lane == 0) for (int k = 0; k < 32; ++k) ex += (double)(float)(_Float16)vals[s * 64 + k] * (double)(float)(_Float16)vals[s * 64 + 32 + k];
    } else if (mode == 1) {
      h8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)vals[s * 64 + (lane >> 5) * 8 + e]; b[e] = (_Float16)vals[s * 64 + 32 + (lane >> 5) * 8 + e]; }
      acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc16, 0, 0, 0);
      if (lane == 0) for (int k = 0; k < 16; ++k) ex += (double)(float)(_Float16)vals[s * 64 + k] * (double)(float)(_Float16)vals[s * 64 + 32 + k];
    } else if (mode == 2) {
      b8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)vals[s * 64 + (lane >> 5) * 8 + e]; b[e] = (__bf16)vals[s * 64 + 32 + (lane >> 5) * 8 + e]; }
      acc16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16, 0, 0, 0);
      if (lane == 0) for (int k = 0; k < 16; ++k) ex += (double)(float)(__bf16)vals[s * 64 + k] * (double)(float)(__bf16)vals[s * 64 + 32 + k];
    } else {
      const float a = vals[s * 64 + (lane >> 5)], b = vals[s * 64 + 32 + (lane >> 5)];
      acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16, 0, 0, 0);
      if (lane == 0) for (int k = 0; k < 2; ++k) ex += (double)vals[s * 64 + k] * (double)vals[s * 64 + 32 + k];
    }
  }
  if (lane == 0) { out[0] = (mode == 0) ? acc4[0] : acc16[0]; exact[0] = ex; }
}

int main() {
  const int NMAX = 4096;
  std::vector<float> v(NMAX * 64);
  uint32_t st = 12345;
  for (auto& x : v) { st = st * 1664525u + 1013904223u; x = 0.5f + (st >> 8) * (1.0f / 16777216.0f); }   // [0.5, 1.5)
  float *dv, *dout; double* dex;
  hipMalloc(&dv, v.size() * 4); hipMalloc(&dout, 4); hipMalloc(&dex, 8);
  hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
  const char* names[4] = {"f16 16x16x32", "f16 32x32x16", "bf16 32x32x16", "f32 32x32x2"};
  for (int mode = 0; mode < 4; ++mode)
    for (int n : {1, 8, 64, 512, 2048, 4096}) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dv, n, mode, dout, dex);
      float o; double e;
      hipMemcpy(&o, dout, 4, hipMemcpyDeviceToHost); hipMemcpy(&e, dex, 8, hipMemcpyDeviceToHost);
      printf("%-14s n=%5d  acc=%.9g exact=%.12g  rel err=%+.3e  (n*2^-25 = %.2e)\n", names[mode], n, o, e, (o - e) / e, n * ldexp(1.0, -25));
    }
  return 0;
}
