// Micro-probe 2: marginal cost (cycles per 64-cycle v_mfma_f32_32x32x2_f32 slot) of each instruction class issued
// between the MFMAs of ONE wave per SIMD on gfx950.  Fillers are inline asm so that exactly what is named is issued.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

enum { F_NONE, F_VMOV, F_VFMA, F_PKFMA, F_PKADD, F_SNOP, F_SALU, F_WAIT, F_DSR32, F_DSR2, F_DSR64, F_DSR128, F_DSW32, F_DSW128,
       F_ACCRD, F_ACCWR, F_EXP, F_RCP, F_GLOAD, F_VADD64 };

template <int F, int N>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x16 acc = (f32x16)(0.f);
  float v0 = in[threadIdx.x], v1 = in[threadIdx.x + 1], v2 = in[threadIdx.x + 2], v3 = in[threadIdx.x + 3];
  f32x2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v1, v2};
  f32x4 q0 = {v0, v1, v2, v3};
  float a = in[threadIdx.x], b = in[threadIdx.x + 64];
  unsigned laddr = (unsigned)(size_t)(lds) + (threadIdx.x & 63) * 16;  // 16-byte aligned per-lane address
  const float* gp = in + (threadIdx.x & 63) * 4;
  unsigned long long gaddr = (unsigned long long)gp;
  unsigned s0 = 1;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        if (F == F_VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v1) : "v"(v0));
        if (F == F_VFMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v1) : "v"(v0), "v"(v2));
        if (F == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p2));
        if (F == F_PKADD) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p0) : "v"(p1), "v"(p2));
        if (F == F_SNOP) asm volatile("s_nop 0");
        if (F == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0));
        if (F == F_WAIT) asm volatile("s_waitcnt lgkmcnt(0)");
        if (F == F_DSR32) asm volatile("ds_read_b32 %0, %1" : "=v"(v1) : "v"(laddr));
        if (F == F_DSR2) asm volatile("ds_read2_b32 %0, %1 offset1:65" : "=v"(p0) : "v"(laddr));
        if (F == F_DSR64) asm volatile("ds_read_b64 %0, %1" : "=v"(p0) : "v"(laddr));
        if (F == F_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(laddr));
        if (F == F_DSW32) asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(v0));
        if (F == F_DSW128) asm volatile("ds_write_b128 %0, %1" : : "v"(laddr), "v"(q0));
        if (F == F_ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v1) : "a"(v3));
        if (F == F_ACCWR) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(v3) : "v"(v0));
        if (F == F_EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(v1) : "v"(v0));
        if (F == F_RCP) asm volatile("v_rcp_f32 %0, %1" : "=v"(v1) : "v"(v0));
        if (F == F_GLOAD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q0) : "v"(gaddr));
        if (F == F_VADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(gaddr) : "s"((unsigned long long)0));
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = v1 + v3 + p0[0] + p0[1] + q0[0] + q0[3] + (float)s0 + (float)(gaddr & 1);
  for (int j = 0; j < 16; ++j) r += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float *g_out, *g_in;
static long long* g_cyc;

template <int F, int N>
double run() {
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL((probe<F, N>), dim3(blocks), dim3(256), 0, 0, g_out, g_in, 10, g_cyc);
  hipLaunchKernelGGL((probe<F, N>), dim3(blocks), dim3(256), 0, 0, g_out, g_in, iters, g_cyc);
  (void)hipDeviceSynchronize();
  long long h[256];
  (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < 256; ++i) sum += (double)h[i];
  return sum / 256 / ((double)iters * 16);
}

template <int F>
void row(const char* name) {
  const double c1 = run<F, 1>(), c2 = run<F, 2>(), c4 = run<F, 4>(), c8 = run<F, 8>();
  printf("%-22s +1: %6.1f  +2: %6.1f  +4: %6.1f  +8: %6.1f   per-instr at 8: %5.2f\n", name, c1, c2, c4, c8, (c8 - 64.0) / 8);
}

int main() {
  (void)hipMalloc(&g_out, 256 * 256 * 4); (void)hipMalloc(&g_in, 16384 * 4); (void)hipMalloc(&g_cyc, 256 * 8);
  float* h = (float*)malloc(16384 * 4);
  for (int i = 0; i < 16384; ++i) h[i] = (float)rand() / RAND_MAX;
  (void)hipMemcpy(g_in, h, 16384 * 4, hipMemcpyHostToDevice);
  printf("cycles per MFMA slot (bare chain = %.1f)\n", run<F_NONE, 1>());
  row<F_VMOV>("v_mov_b32");
  row<F_VFMA>("v_fma_f32");
  row<F_PKFMA>("v_pk_fma_f32");
  row<F_PKADD>("v_pk_add_f32");
  row<F_EXP>("v_exp_f32");
  row<F_RCP>("v_rcp_f32");
  row<F_VADD64>("v_lshl_add_u64");
  row<F_ACCRD>("v_accvgpr_read");
  row<F_ACCWR>("v_accvgpr_write");
  row<F_SNOP>("s_nop 0");
  row<F_SALU>("s_add_u32");
  row<F_WAIT>("s_waitcnt lgkmcnt(0)");
  row<F_DSR32>("ds_read_b32");
  row<F_DSR2>("ds_read2_b32");
  row<F_DSR64>("ds_read_b64");
  row<F_DSR128>("ds_read_b128");
  row<F_DSW32>("ds_write_b32");
  row<F_DSW128>("ds_write_b128");
  row<F_GLOAD>("global_load_dwordx4");
  return 0;
}
