// What the decoder translation units share: the kernel argument block, the small device helpers, the LDS hand-over
// primitives with their bounded waits, the end-of-kernel reduction of the all-64 kernels and the LDS budgets.
//   mfma_decoder.hip      the kernels that run by default (dec_x3_64s, dec_mfma<K,H,M,DG>), set-up kernels, guard, launch
//   xcheck_decoders.hip   earlier generations of the all-64 kernel, kept as cross-checks behind OIVAE_DEC=... and as the
//                         fallbacks for launches the default cannot address (dz beyond 32-bit byte offsets, unaligned
//                         parameters)
#pragma once
#include "common.h"
#include "x3_split.h"

#include <stdlib.h>
#include <type_traits>

namespace oivae {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct MfmaDecArgs {
  const float *W1, *b1, *W2, *b2, *ls, *Ws;
  float *gW1, *gb1, *gW2, *gb2, *gls, *gWs;  // chunk-0 destinations
  int64_t chunk_stride;
  const float* X;
  const float* z;
  float* dzT;      // [K, N]
  int64_t B, N;
  int D;
  float inv;       // 1/(S*B)
  float sig_off;
  double* ll_part;
  const uint32_t* zimg;  // dec_x3_64: split image of z (x3_split_z), else unused
  const uint32_t* rng;   // dec_x3_64g: [0] max|X| the scales assume, [5] max|z|
  uint32_t* rng_seen;    // dec_x3_64g: max|x| of what the kernel read (atomicMax), or nullptr
  const uint32_t* gate;  // gated launch: returns at once when *gate == 0 (nullptr: always runs)
  ErrWords err;          // the library's error words (host-mapped + device-resident), raised by a timed-out LDS wait
  int G;
  float* acc2;           // dec_x3_64s: second-level slabs of the weight-gradient waves (nullptr: short launches, no second level)
  int xg;                // dec_x3_64s<true>: columns of X per group (< 64, a multiple of 4); 0 otherwise
};

__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); exp2/rcp are ~1 ulp, absolute error ~2e-7
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);  // 2*log2(e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// row index inside a 32x32 accumulator tile held by register r of lane half hh
#define OIVAE_ROW(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))

#define OIVAE_SB() __builtin_amdgcn_sched_barrier(0)
#define OIVAE_ROWC(r) (((r) & 3) + 8 * ((r) >> 2))

__device__ __forceinline__ f32x16 mf32(float a_, float b_, f32x16 c_) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, c_, 0, 0, 0);
}

// ---- end of an all-64 decoder kernel: the four waves' partials of dW2 and P are summed in a fixed order through
// LDS, dW1 = P Ws and dWs = P^T W1 are formed (two 64^3 fp32 products per group), and the per-column sums become
// db2, dlog_sigma, db1 and the group's log-likelihood partial.  `stage`: >= 4*4096 + 64*65 floats; sP, uW1: 64*65 each.
__device__ __forceinline__ void dec64_finish(const MfmaDecArgs& a, int g, int chunk, int tid, int wave, int lane,
                                             f32x16 (&dW2a)[2][2], f32x16 (&Pa)[2][2], float cs_gmu, float cs_gmu2,
                                             float cs_da1, float* stage, float* sP, float* uW1, bool act = true) {
  // `act`: waves that hold partials (all of them in the 4-wave kernels; the weight-gradient waves of dec_x3_64p,
  // numbered 0..3 with tid 0..255 by the caller); the others only pass the barriers
  constexpr int K = 64, H = 64, M = 64, DG = 64, LW = 65;
  const int l31 = lane & 31, hh = lane >> 5;
  const float* Wsg = a.Ws + (int64_t)g * K * H;
  const float* W1g = a.W1 + (int64_t)g * M * H;

  // ---- reduce the four waves' partials in a fixed order (as v3) ----
  const int64_t cs = a.chunk_stride * chunk;
  auto stage_acc = [&](f32x16 (&acc)[2][2]) {
    __syncthreads();
    if (act) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[wave * 4096 + (32 * i + OIVAE_ROW(r, hh)) * 64 + 32 * j + l31] = acc[i][j][r];
    }
    __syncthreads();
  };
  stage_acc(dW2a);
  if (act) {
    float* dst = a.gW2 + cs + (int64_t)g * DG * M;
    for (int idx = tid; idx < 4096; idx += 256)
      dst[idx] = ((stage[idx] + stage[4096 + idx]) + stage[2 * 4096 + idx]) + stage[3 * 4096 + idx];
  }
  stage_acc(Pa);
  float* uWs = stage + 4 * 4096;
  for (int idx = tid; act && idx < 4096; idx += 256) {
    const int o = (idx >> 6) * LW + (idx & 63);
    sP[o] = ((stage[idx] + stage[4096 + idx]) + stage[2 * 4096 + idx]) + stage[3 * 4096 + idx];
    uW1[o] = W1g[idx];
    uWs[o] = Wsg[idx];
  }
  __syncthreads();
  if (act) {
    const int it = wave >> 1, ht = wave & 1;
    f32x16 acc = (f32x16)(0.f);  // dW1[m][h] = sum_k P[m][k] Ws[k][h]
    const float* pa = sP + (32 * it + l31) * LW + hh;
    const float* pb = uWs + hh * LW + 32 * ht + l31;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) acc = mf32(pa[2 * s], pb[2 * s * LW], acc);
    float* d1 = a.gW1 + cs + (int64_t)g * M * H;
#pragma unroll
    for (int r = 0; r < 16; ++r) d1[(32 * it + OIVAE_ROW(r, hh)) * H + 32 * ht + l31] = acc[r];
    acc = (f32x16)(0.f);         // dWs[k][h] = sum_m P[m][k] W1[m][h]
    pa = sP + hh * LW + 32 * it + l31;
    pb = uW1 + hh * LW + 32 * ht + l31;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) acc = mf32(pa[2 * s * LW], pb[2 * s * LW], acc);
    float* d2 = a.gWs + cs + (int64_t)g * K * H;
#pragma unroll
    for (int r = 0; r < 16; ++r) d2[(32 * it + OIVAE_ROW(r, hh)) * H + 32 * ht + l31] = acc[r];
  }
  if (act) {
    stage[wave * 192 + lane] = cs_gmu;
    stage[wave * 192 + 64 + lane] = cs_gmu2;
    stage[wave * 192 + 128 + lane] = cs_da1;
  }
  __syncthreads();
  float* sred = stage + 1024;
  if (act && tid < 64) {
    const float s_gmu = ((stage[tid] + stage[192 + tid]) + stage[384 + tid]) + stage[576 + tid];
    const float s_gmu2 = ((stage[64 + tid] + stage[256 + tid]) + stage[448 + tid]) + stage[640 + tid];
    const float s_da1 = ((stage[128 + tid] + stage[320 + tid]) + stage[512 + tid]) + stage[704 + tid];
    const int64_t d = (int64_t)g * DG + tid;
    const float e = expf(a.ls[d]);
    const float sg = e + a.sig_off;
    const float c2 = a.inv / sg;
    const float sumq2 = s_gmu2 / (c2 * c2);
    const float nrows = (chunk == 0) ? (float)a.N : 0.f;
    a.gb2[cs + d] = s_gmu;
    a.gls[cs + d] = (nrows - sumq2) * (e * a.inv / sg);
    a.gb1[cs + (int64_t)g * M + tid] = s_da1;
    sred[tid] = nrows * (-0.5f * kLog2Pi - logf(sg)) - 0.5f * sumq2;
  }
  __syncthreads();
  if (act && tid == 0) {
    double tot = 0.0;
    for (int i = 0; i < 64; ++i) tot += (double)sred[i];
    a.ll_part[(int64_t)chunk * gridDim.x + g] = tot;
  }
}

constexpr int kX3DecLds = (4 * 6144 + 192 + 3 * 64 * 65) * 4;  // images | b1, b2, c3 | fp32 staging

__device__ __forceinline__ int x3_slot_feature(int ks, int hh, int e) { return 16 * ks + 8 * (e >> 2) + 4 * hh + (e & 3); }

constexpr int kX3DecLdsP = kX3DecLds + 64;

// Bounded wait on an LDS counter.  The bound (2^22 polls with s_sleep: > 0.1 s, five orders of magnitude above a
// hand-over) only ends a wave whose partner died; the wave then raises the library's error word (host-mapped memory,
// api.hip: the next entry-point call on this device fails with OIVAE_E_HIP) and carries on with whatever the tile
// holds, so the grid still drains.  OIVAE_TEST_WAIT_LIMIT: scratch build of the test suite that forces the time-out.
#ifndef OIVAE_TEST_WAIT_LIMIT
#define OIVAE_TEST_WAIT_LIMIT (1 << 22)
#endif
__device__ __forceinline__ void lds_wait_ge(const volatile uint32_t* f, uint32_t v, const ErrWords& err) {
  int guard = 0;
  // the counter lives in LDS: read it as such (through the generic pointer it is a flat load, which also waits for
  // every global load in flight)
  const volatile __attribute__((address_space(3))) uint32_t* fl = (const volatile __attribute__((address_space(3))) uint32_t*)f;
  while (__builtin_amdgcn_readfirstlane(*fl) < v) {
    if (++guard >= OIVAE_TEST_WAIT_LIMIT) {
      if (err.host != nullptr && (threadIdx.x & 63) == 0) {   // "LDSW"
        *(volatile uint32_t*)err.dev = 0x4c445357u;    // first the word the update kernels read, then the host's
        *(volatile uint32_t*)err.host = 0x4c445357u;
      }
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_post(volatile uint32_t* f, uint32_t v) {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's tile accesses are done
  *(volatile __attribute__((address_space(3))) uint32_t*)f = v;
  __builtin_amdgcn_sched_barrier(0);   // or the scheduler sinks the store below the MFMAs that follow: the partner waits
}


// timing-only ablation (tools/dec_ablate.py, never in the shipped library): the dz adds compiled out behind a
// condition the compiler cannot resolve, so the product itself stays
#ifdef OIVAE_ABL_NOATOMIC
#define OIVAE_DZ_ADD(p, v) do { if (a.N < 0) *(p) = (v); } while (0)
#else
#define OIVAE_DZ_ADD(p, v) atomicAdd((p), (v))
#endif

constexpr int kPairImg = 2 * 3 * 4096;   // dwords: [group][A | W2 | W2^T][2 rb][4 ks][2 terms][64 lanes][4]
constexpr int kPairVec = 2 * 192;        // floats: per group b1, b2, c3 in scaled units
constexpr int kPairTile = 32 * 68;       // floats: one chain wave's published tile
constexpr int kPairScr = 3 * 64 * 65 + 64 + 512;   // set-up staging 3 x [64][65] + bounds + partials; row loop: four tiles
constexpr int kPairLds = (kPairImg + kPairVec + kPairScr + 64) * 4;
static_assert(kPairScr >= 4 * kPairTile, "tiles");
static_assert(kPairImg >= 2 * 4096 + 3 * 64 * 65 + 2 * 192, "end-of-kernel staging");

typedef short x3_s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t x3_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ x3_u32x2 lds_read_tr16(const char* p) {
  return __builtin_bit_cast(x3_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) x3_s16x4*)(const_cast<char*>(p))));
}

// Scratch builds only (tools/build_ablation.sh TIMES, tools/dec_probe.py): wall-clock stamps of wave 0 of every
// workgroup at the phase boundaries of dec_x3_64g, read back through oivae_probe_read.  (The stamps cost registers:
// that build spills in the row loop, so only its set-up and end-of-kernel phases are representative.)
#ifdef OIVAE_ABL_TIMES
static __device__ unsigned long long g_probe[8192 * 8];
#define OIVAE_PROBE(i) \
  do { if (threadIdx.x == 0) g_probe[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 8 + (i)] = wall_clock64(); } while (0)
#define OIVAE_PROBE_SETUP(i) do {} while (0)
#elif defined(OIVAE_ABL_TIMES2)   // the sub-phases of the set-up of a pair's SECOND group (dec_x3_64s; tools/dec_probe.py --setup)
static __device__ unsigned long long g_probe[8192 * 8];
#define OIVAE_PROBE(i) do {} while (0)
#define OIVAE_PROBE_SETUP(i) \
  do { if (threadIdx.x == 0 && q == 1) g_probe[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define OIVAE_PROBE(i) do {} while (0)
#define OIVAE_PROBE_SETUP(i) do {} while (0)
#endif
#ifdef OIVAE_ABL_TIMES3   // the sub-phases of the end of the kernel for a pair's second group (tools/dec_probe.py --end)
static __device__ unsigned long long g_probe[8192 * 8];
#define OIVAE_PROBE_END(i) \
  do { if (threadIdx.x == 256 && q == 1) g_probe[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define OIVAE_PROBE_END(i) do {} while (0)
#endif

// timing-only stand-in (tools/dec_ablate.py C16 / W16): the flops of one 32x32x16 product issued as two 16x16x32
// instructions on the same operand registers (results are wrong)
__device__ __forceinline__ void h2mm_as16(f32x16& acc, const Frag2& A, const Frag2& B) {
  x3_f32x4 c0 = __builtin_shufflevector(acc, acc, 0, 1, 2, 3), c1 = __builtin_shufflevector(acc, acc, 4, 5, 6, 7);
  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_h16x8, A.p[PA[q]]), __builtin_bit_cast(x3_h16x8, B.p[PB[q]]), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_h16x8, A.p[PA[q]]), __builtin_bit_cast(x3_h16x8, B.p[PB[q]]), c1, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i] = c0[i]; acc[4 + i] = c1[i]; }
}


// xcheck_decoders.hip: `which` = '4' dec_mfma64_v4 (fp32 matrix instruction, composed first layer), 'x' dec_x3_64
// (three bf16 terms, one wave per SIMD), 'p' dec_x3_64p (bf16 terms, role-split pair of waves; also the fallback for
// launches whose dz image exceeds 32-bit byte offsets and for unaligned parameters), 'g' dec_x3_64g (fp16 terms on
// 32x32x16 instructions, two groups per workgroup).  Sets the LDS attribute and launches; the caller has prepared
// dz, the z image and the ranges as for the default kernel.
int launch_xcheck_decoder(char which, const MfmaDecArgs& a, int G, int chunks, hipStream_t st);

}  // namespace oivae
