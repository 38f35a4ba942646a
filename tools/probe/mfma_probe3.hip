// Micro-probe 3: v_mfma_f32_32x32x16_bf16 on gfx950 -- issue rate of a chain (one or two accumulators), and the marginal
// cost of the instructions an fp32 -> 3 x bf16 operand split needs when they are issued between the MFMAs, with one
// and with two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

enum { F_NONE, F_VMOV, F_PKADD, F_CVT, F_AND, F_PERM, F_LSHL, F_DSR128, F_DSW64, F_DSW128, F_GLOAD, F_SPLIT };

template <int F, int N, int NACC, int TPB>
__global__ __launch_bounds__(TPB) void probe(float* out, const float* in, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += TPB) lds[i] = in[i];
  __syncthreads();
  f32x16 acc0 = (f32x16)(0.f), acc1 = (f32x16)(0.f);
  float v0 = in[threadIdx.x], v1 = in[threadIdx.x + 1], v2 = in[threadIdx.x + 2], v3 = in[threadIdx.x + 3];
  f32x2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v1, v2};
  f32x4 q0 = {v0, v1, v2, v3};
  unsigned u0 = __float_as_uint(v0), u1 = __float_as_uint(v1), u2 = 0x07060302u;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[threadIdx.x + j]; b[j] = (__bf16)in[threadIdx.x + 8 + j]; }
  unsigned laddr = (unsigned)(size_t)(lds) + (threadIdx.x & 63) * 16;
  const float* gp = in + (threadIdx.x & 63) * 4;
  unsigned long long gaddr = (unsigned long long)gp;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        if (F == F_VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v1) : "v"(v0));
        if (F == F_PKADD) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p0) : "v"(p1), "v"(p2));
        if (F == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u1) : "v"(v0), "v"(v2));
        if (F == F_AND) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u1) : "v"(u0));
        if (F == F_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u1) : "v"(u0), "v"(v2), "v"(u2));
        if (F == F_LSHL) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u1) : "v"(u0));
        if (F == F_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(laddr));
        if (F == F_DSW64) asm volatile("ds_write_b64 %0, %1" : : "v"(laddr), "v"(p1));
        if (F == F_DSW128) asm volatile("ds_write_b128 %0, %1" : : "v"(laddr), "v"(q0));
        if (F == F_GLOAD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q0) : "v"(gaddr));
        if (F == F_SPLIT) {  // the full split of one pair of floats into three packed bf16 dwords (9 instructions)
          unsigned hp, mp, lp; f32x2 h2, r1, m2, r2;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hp) : "v"(p1[0]), "v"(p1[1]));
          asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(h2[0]) : "v"(hp));
          asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h2[1]) : "v"(hp));
          asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r1) : "v"(p1), "v"(h2));
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(mp) : "v"(r1[0]), "v"(r1[1]));
          asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(m2[0]) : "v"(mp));
          asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(m2[1]) : "v"(mp));
          asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "v"(r1), "v"(m2));
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lp) : "v"(r2[0]), "v"(r2[1]));
          u1 = hp ^ mp ^ lp;
        }
      }
      if (NACC == 1 || (s & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = v1 + p0[0] + p0[1] + q0[0] + q0[3] + __uint_as_float(u1) + (float)(gaddr & 1);
  for (int j = 0; j < 16; ++j) r += acc0[j] + acc1[j];
  out[blockIdx.x * TPB + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float *g_out, *g_in;
static long long* g_cyc;

static double g_mhz = 0;  // shader clock of the last run: s_memtime cycles / hipEvent time

template <int F, int N, int NACC, int TPB>
double run(int iters = 2000) {
  const int blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<F, N, NACC, TPB>), dim3(blocks), dim3(TPB), 0, 0, g_out, g_in, 10, g_cyc);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<F, N, NACC, TPB>), dim3(blocks), dim3(TPB), 0, 0, g_out, g_in, iters, g_cyc);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < 256; ++i) sum += (double)h[i];
  g_mhz = sum / 256 / (ms * 1e3);
  return sum / 256 / ((double)iters * 16);
}

template <int F>
void row(const char* name) {
  printf("%-22s 1 wave/SIMD  +1: %6.1f +2: %6.1f +4: %6.1f +8: %6.1f | 2 waves/SIMD (cycles per MFMA pair) +1: %6.1f +4: %6.1f +8: %6.1f\n", name,
         run<F, 1, 2, 256>(), run<F, 2, 2, 256>(), run<F, 4, 2, 256>(), run<F, 8, 2, 256>(),
         run<F, 1, 2, 512>(), run<F, 4, 2, 512>(), run<F, 8, 2, 512>());
}

int main() {
  (void)hipMalloc(&g_out, 256 * 512 * 4); (void)hipMalloc(&g_in, 16384 * 4); (void)hipMalloc(&g_cyc, 256 * 8);
  float* h = (float*)malloc(16384 * 4);
  for (int i = 0; i < 16384; ++i) h[i] = (float)rand() / RAND_MAX;
  (void)hipMemcpy(g_in, h, 16384 * 4, hipMemcpyHostToDevice);
  printf("bf16 32x32x16 chain, cycles per MFMA: one accumulator %.1f, two accumulators %.1f; two waves per SIMD (per pair) %.1f\n",
         run<F_NONE, 1, 1, 256>(), run<F_NONE, 1, 2, 256>(), run<F_NONE, 1, 2, 512>());
  for (int rep = 0; rep < 3; ++rep) {
    const double c1 = run<F_NONE, 1, 2, 256>(200000);
    const double f1 = g_mhz;
    const double c2 = run<F_NONE, 1, 2, 512>(100000);
    printf("sustained bf16 chain (~0.1-0.2 s): 1 wave/SIMD %.1f cycles/MFMA at %.0f MHz; 2 waves/SIMD %.1f per pair at %.0f MHz\n", c1, f1, c2, g_mhz);
  }
  row<F_VMOV>("v_mov_b32");
  row<F_PKADD>("v_pk_add_f32");
  row<F_CVT>("v_cvt_pk_bf16_f32");
  row<F_AND>("v_and_b32");
  row<F_PERM>("v_perm_b32");
  row<F_LSHL>("v_lshlrev_b32");
  row<F_SPLIT>("split pair (9 instr)");
  row<F_DSR128>("ds_read_b128");
  row<F_DSW64>("ds_write_b64");
  row<F_DSW128>("ds_write_b128");
  row<F_GLOAD>("global_load_dwordx4");
  return 0;
}
