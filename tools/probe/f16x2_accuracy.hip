// Accuracy of fp32 products carried on the 16-bit matrix instructions, against float64:
//   fp32   v_mfma_f32_32x32x2_f32 (what the x3 kernels are measured against)
//   b3x6   three bf16 terms per operand, six products (the x3 scheme of this repository)
//   h2x4   two fp16 terms per operand, four products      h2x3: the same without lo*lo
// C[32][32] = A[32][K] * B[K][32], K = 64 (one decoder product) and 4096, entries ~ U(-1,1) * scale; the fp16 variants
// are NOT range-safe (no scaling here): `scale` shows where they break.  One wave.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using h16x2 = __attribute__((ext_vector_type(2))) _Float16;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

#define ROW(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))

__global__ __launch_bounds__(64) void k(const float* A, const float* B, int K, float* C) {
  const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
  f32x16 c32 = (f32x16)(0.f), cb = (f32x16)(0.f), ch4 = (f32x16)(0.f), ch3 = (f32x16)(0.f);
  for (int k0 = 0; k0 < K; k0 += 16) {
    float a[8], b[8];
    for (int e = 0; e < 8; ++e) { a[e] = A[l31 * K + k0 + 8 * hh + e]; b[e] = B[(k0 + 8 * hh + e) * 32 + l31]; }
    // fp32 instruction: 32x32x2, lane half = k parity
    for (int s = 0; s < 8; ++s) {
      const float av = A[l31 * K + k0 + 2 * s + hh], bv = B[(k0 + 2 * s + hh) * 32 + l31];
      c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c32, 0, 0, 0);
    }
    bf16x8 ab[3], bb[3];
    h16x8 ah[2], bh[2];
    for (int e = 0; e < 8; ++e) {
      float ra = a[e], rb = b[e];
      for (int p = 0; p < 3; ++p) {
        const __bf16 qa = (__bf16)ra, qb = (__bf16)rb;
        ab[p][e] = qa; bb[p][e] = qb;
        ra -= (float)qa; rb -= (float)qb;
      }
      ra = a[e]; rb = b[e];
      for (int p = 0; p < 2; ++p) {
        const _Float16 qa = (_Float16)ra, qb = (_Float16)rb;
        ah[p][e] = qa; bh[p][e] = qb;
        ra -= (float)qa; rb -= (float)qb;
      }
    }
    const int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    for (int q = 0; q < 6; ++q) cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[PA[q]], bb[PB[q]], cb, 0, 0, 0);
    ch4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[1], ch4, 0, 0, 0);
    ch4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[0], ch4, 0, 0, 0);
    ch4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[1], ch4, 0, 0, 0);
    ch4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], ch4, 0, 0, 0);
    ch3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[0], ch3, 0, 0, 0);
    ch3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[1], ch3, 0, 0, 0);
    ch3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], ch3, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = ROW(r, hh);
    C[0 * 1024 + row * 32 + l31] = c32[r];
    C[1 * 1024 + row * 32 + l31] = cb[r];
    C[2 * 1024 + row * 32 + l31] = ch4[r];
    C[3 * 1024 + row * 32 + l31] = ch3[r];
  }
}
int main() {
  const char* names[4] = {"fp32", "b3x6", "h2x4", "h2x3"};
  const int Ks[2] = {64, 4096};
  const float scales[6] = {1.f, 1e-3f, 1e-5f, 1e-7f, 300.f, 1e5f};
  float *dA, *dB, *dC;
  (void)hipMalloc(&dA, 32 * 4096 * 4); (void)hipMalloc(&dB, 32 * 4096 * 4); (void)hipMalloc(&dC, 4 * 1024 * 4);
  static float A[32 * 4096], B[32 * 4096], C[4 * 1024];
  for (int ki = 0; ki < 2; ++ki)
    for (int si = 0; si < 6; ++si) {
      const int K = Ks[ki];
      srand(7);
      for (int i = 0; i < 32 * K; ++i) { A[i] = (2.f * rand() / RAND_MAX - 1.f) * scales[si]; B[i] = (2.f * rand() / RAND_MAX - 1.f); }
      (void)hipMemcpy(dA, A, 32 * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, 32 * K * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
      (void)hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
      double err[4] = {0, 0, 0, 0}, ref_max = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double r = 0;
          for (int kk = 0; kk < K; ++kk) r += (double)A[i * K + kk] * (double)B[kk * 32 + j];
          ref_max = fmax(ref_max, fabs(r));
          for (int m = 0; m < 4; ++m) err[m] = fmax(err[m], fabs((double)C[m * 1024 + i * 32 + j] - r));
        }
      printf("K=%4d  A ~ %.0e * U(-1,1):  max|err|/max|ref| ", K, scales[si]);
      for (int m = 0; m < 4; ++m) printf(" %s %.2e", names[m], err[m] / ref_max);
      printf("\n");
    }
  return 0;
}
