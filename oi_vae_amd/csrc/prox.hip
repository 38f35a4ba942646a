// Group-lasso proximal step (block soft-threshold of every row of Ws) for gfx950.
//
// Replaces BayesianGroupLassoGenerator.proximal_step
// (/root/reference/lib/models.py:106-111), optionally fused with the plain SGD
// update of Ws that precedes it (/root/reference/lib/oivae.py:76-79).
//
// HBM-bound: one read and one write of Ws (8 B/element; 12 with the fused SGD,
// which also reads dWs) plus one mask byte per row.  The arithmetic is pinned
// bit for bit to the reference's CPU path -- see oracle/prox_oracle.c for the
// derivation of each rule:
//   * w*w and every partial sum rounded to fp32 (this TU is built with
//     -ffp-contract=off), summed in the order of torch 2.10's CPU sum kernel;
//   * the square root is MKL-VML's vmsSqrt(HA): rsqrt14 table lookup plus one
//     fused Newton step, not the correctly rounded root;
//   * divide and multiply correctly rounded, two roundings per element.
#include "common.h"
#include "prox_dev.h"

namespace oivae {

// ---- path A: any H, one thread per row, scalar emulation of the sum order ----
template <bool SGD>
__global__ __launch_bounds__(256) void prox_rows_scalar(float* __restrict__ Ws,
                                                        const float* __restrict__ dWs,
                                                        int64_t rows, int H, float neg_lr, float t,
                                                        int do_prox,
                                                        uint8_t* __restrict__ mask,
                                                        float* __restrict__ norms,
                                                        const uint16_t* __restrict__ tab) {
  if (*reinterpret_cast<const uint32_t*>(tab + kBadWordOffsetU16) != 0) return;   // a decoder launch of this step failed: no update (common.h ErrWords)
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  prox_row_scalar_dev<SGD>(Ws, dWs, r, H, neg_lr, t, do_prox, mask, norms, tab);
}

// ---- path B: H = 8*V, two threads per row, 16-byte accesses, row in registers ----
// Thread `hh` of a row owns lanes 4hh..4hh+3 of every 8-float vector, which is
// exactly what one lane group of the CPU kernel's vector accumulators sees.
// The arithmetic of one row held by its two threads (v: this thread's V 16-byte quarters after the SGD update):
// returns the row norm, leaves the thresholded row in v and says whether this thread's half is all zero.
template <int V, typename TabPtr>
__device__ __forceinline__ float prox_row_pair(float4 (&v)[V], int hh, float t, int do_prox, TabPtr tab, bool& allzero) {
  float4 P[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) P[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int NQ = V / 4;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int k = (i < 4 * NQ) ? (i & 3) : 0;
    P[k].x = P[k].x + v[i].x * v[i].x;
    P[k].y = P[k].y + v[i].y * v[i].y;
    P[k].z = P[k].z + v[i].z * v[i].z;
    P[k].w = P[k].w + v[i].w * v[i].w;
  }
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    P[0].x = P[0].x + P[k].x; P[0].y = P[0].y + P[k].y;
    P[0].z = P[0].z + P[k].z; P[0].w = P[0].w + P[k].w;
  }
  // lanes 0..7 left to right: lanes 0-3 live in the even thread, 4-7 in the odd one
  const float o0 = __shfl_xor(P[0].x, 1), o1 = __shfl_xor(P[0].y, 1);
  const float o2 = __shfl_xor(P[0].z, 1), o3 = __shfl_xor(P[0].w, 1);
  float l0, l1, l2, l3, l4, l5, l6, l7;
  if (hh == 0) { l0 = P[0].x; l1 = P[0].y; l2 = P[0].z; l3 = P[0].w; l4 = o0; l5 = o1; l6 = o2; l7 = o3; }
  else         { l0 = o0; l1 = o1; l2 = o2; l3 = o3; l4 = P[0].x; l5 = P[0].y; l6 = P[0].z; l7 = P[0].w; }
  float acc = 0.f;
  acc = acc + l0; acc = acc + l1; acc = acc + l2; acc = acc + l3;
  acc = acc + l4; acc = acc + l5; acc = acc + l6; acc = acc + l7;
  const float n = mkl_sqrtf_dev(acc, tab);
  float den, sc;
  prox_scales(n, t, den, sc);
  allzero = true;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    if (do_prox) {
      v[i].x = v[i].x / den; v[i].x = v[i].x * sc;
      v[i].y = v[i].y / den; v[i].y = v[i].y * sc;
      v[i].z = v[i].z / den; v[i].z = v[i].z * sc;
      v[i].w = v[i].w / den; v[i].w = v[i].w * sc;
    }
    allzero = allzero && (v[i].x == 0.f) && (v[i].y == 0.f) && (v[i].z == 0.f) && (v[i].w == 0.f);
  }
  return n;
}

template <int V, bool SGD>
__device__ __forceinline__ void prox_sgd_update(float4 (&v)[V], const float4 (&g)[V], float neg_lr) {
  if (!SGD) return;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    v[i].x = __builtin_fmaf(neg_lr, g[i].x, v[i].x);
    v[i].y = __builtin_fmaf(neg_lr, g[i].y, v[i].y);
    v[i].z = __builtin_fmaf(neg_lr, g[i].z, v[i].z);
    v[i].w = __builtin_fmaf(neg_lr, g[i].w, v[i].w);
  }
}

template <int V, bool SGD>
__global__ __launch_bounds__(256) void prox_rows_vec(float* __restrict__ Ws,
                                                     const float* __restrict__ dWs,
                                                     int64_t rows, float neg_lr, float t, int do_prox,
                                                     uint8_t* __restrict__ mask,
                                                     float* __restrict__ norms,
                                                     const uint16_t* __restrict__ tab) {
  if (*reinterpret_cast<const uint32_t*>(tab + kBadWordOffsetU16) != 0) return;   // a decoder launch of this step failed: no update (common.h ErrWords)
  constexpr int H = 8 * V;
  const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = gt >> 1;
  const int hh = (int)(gt & 1);
  const bool live = r < rows;
  float4 v[V], g[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { v[i] = make_float4(0.f, 0.f, 0.f, 0.f); g[i] = v[i]; }
  if (live) {
    const float4* wp = reinterpret_cast<const float4*>(Ws + r * H) + hh;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = wp[2 * i];
    if (SGD) {
      const float4* gp = reinterpret_cast<const float4*>(dWs + r * H) + hh;
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = gp[2 * i];
    }
  }
  prox_sgd_update<V, SGD>(v, g, neg_lr);
  bool allzero;
  const float n = prox_row_pair<V>(v, hh, t, do_prox, tab, allzero);
  const int other = __shfl_xor((int)allzero, 1);
  if (live) {
    float4* wp = reinterpret_cast<float4*>(Ws + r * H) + hh;
#pragma unroll
    for (int i = 0; i < V; ++i) wp[2 * i] = v[i];
    if (hh == 0) {
      if (mask) mask[r] = (allzero && other) ? 1 : 0;
      if (norms) norms[r] = n;
    }
  }
}

// ---- path C: path B for large launches, persistent, with the rsqrt14 table in LDS ----
// Path B's threads live for two dependent memory round trips -- the row, then the table entry its norm selects (a
// 64-address gather into a 128 KB table: L2 at best) -- and the wide-prox launch (2 M rows of 32 bytes) runs eight
// generations of them per CU: 4.7 TB/s where a plain elementwise kernel over the same bytes reaches 6.2.  Here one
// workgroup of 1024 threads per CU copies the table into LDS once (128 KB) and walks its share of the rows, 512 per
// pass, with the next pass's rows requested before this pass's are worked on: one round trip per pass, and it is
// hidden (requesting two passes ahead changes nothing).  Same arithmetic, same bits.  Wide-prox launch with the fused
// SGD update: 43.5 -> 36.4 us = 5.5 TB/s.
constexpr int kProxLdsThreads = 1024;
template <int V, bool SGD>
__global__ __launch_bounds__(kProxLdsThreads, 1) void prox_rows_vec_lds(float* __restrict__ Ws,
                                                                        const float* __restrict__ dWs,
                                                                        int64_t rows, float neg_lr, float t, int do_prox,
                                                                        uint8_t* __restrict__ mask,
                                                                        float* __restrict__ norms,
                                                                        const uint16_t* __restrict__ tab) {
  if (*reinterpret_cast<const uint32_t*>(tab + kBadWordOffsetU16) != 0) return;   // a decoder launch of this step failed: no update (common.h ErrWords)
  constexpr int H = 8 * V;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint16_t* ltab = reinterpret_cast<uint16_t*>(smem_raw);   // [65536]
  {
    const uint4* src = reinterpret_cast<const uint4*>(tab);
    uint4* dst = reinterpret_cast<uint4*>(ltab);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[threadIdx.x + i * kProxLdsThreads] = src[threadIdx.x + i * kProxLdsThreads];
  }
  const int hh = (int)(threadIdx.x & 1);
  const int64_t stride = (int64_t)gridDim.x * (kProxLdsThreads / 2);
  int64_t r = (int64_t)blockIdx.x * (kProxLdsThreads / 2) + (threadIdx.x >> 1);
  const int64_t passes = (rows + stride - 1) / stride;   // the same for every thread: the pair shuffles stay uniform
  float4 v[V], g[V], vn[V], gn[V];
  auto request = [&](int64_t row, float4 (&a)[V], float4 (&b)[V]) {
#pragma unroll
    for (int i = 0; i < V; ++i) { a[i] = make_float4(0.f, 0.f, 0.f, 0.f); b[i] = a[i]; }
    if (row < rows) {
      const float4* wp = reinterpret_cast<const float4*>(Ws + row * H) + hh;
#pragma unroll
      for (int i = 0; i < V; ++i) a[i] = wp[2 * i];
      if (SGD) {
        const float4* gp = reinterpret_cast<const float4*>(dWs + row * H) + hh;
#pragma unroll
        for (int i = 0; i < V; ++i) b[i] = gp[2 * i];
      }
    }
  };
  request(r, vn, gn);
  __syncthreads();   // the table is in place
  for (int64_t p = 0; p < passes; ++p, r += stride) {
#pragma unroll
    for (int i = 0; i < V; ++i) { v[i] = vn[i]; g[i] = gn[i]; }
    if (p + 1 < passes) request(r + stride, vn, gn);
    prox_sgd_update<V, SGD>(v, g, neg_lr);
    bool allzero;
    const float n = prox_row_pair<V>(v, hh, t, do_prox, (const uint16_t*)ltab, allzero);
    const int other = __shfl_xor((int)allzero, 1);
    if (r < rows) {
      float4* wp = reinterpret_cast<float4*>(Ws + r * H) + hh;
#pragma unroll
      for (int i = 0; i < V; ++i) wp[2 * i] = v[i];
      if (hh == 0) {
        if (mask) mask[r] = (allzero && other) ? 1 : 0;
        if (norms) norms[r] = n;
      }
    }
  }
}

// ---- path D: H >= 512, where ATen's row sum cascades (oracle/prox_oracle.c: oivae_oracle_row_sumsq) ----
// The CPU kernel's sum is 32 independent chains -- 4 vector accumulators x 8 lanes, chain (k, l) taking elements
// 32 i + 8 k + l, i = 0, 1, ... -- each with the same cascade: every 16 steps level 0 is added into level 1 and cleared,
// every 256 steps level 1 into level 2, every 4096 level 2 into level 3.  32 threads share a row, one chain each (the
// reads of a step are one contiguous 128 bytes); then the levels fold into level 0, the vectors left over behind the
// last full step go to accumulator 0's lanes, accumulators 1..3 are added to accumulator 0 in order, and one thread
// adds the scalar tail and the eight lanes left to right.  A second pass over the row applies the threshold.
template <bool SGD>
__global__ __launch_bounds__(256) void prox_rows_wide(float* __restrict__ Ws, const float* __restrict__ dWs,
                                                      int64_t rows, int H, float neg_lr, float t, int do_prox,
                                                      uint8_t* __restrict__ mask, float* __restrict__ norms,
                                                      const uint16_t* __restrict__ tab) {
  if (*reinterpret_cast<const uint32_t*>(tab + kBadWordOffsetU16) != 0) return;   // a decoder launch of this step failed: no update (common.h ErrWords)
  const int sub = threadIdx.x & 31;                                    // chain: accumulator k = sub / 8, lane l = sub % 8
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const bool live = r < rows;                                           // uniform per group of 32
  const int64_t rr = live ? r : rows - 1;
  float* w = Ws + rr * H;
  const float* g = SGD ? dWs + rr * H : nullptr;
  auto ld = [&](int i) -> float {
    float v = w[i];
    if (SGD) v = __builtin_fmaf(neg_lr, g[i], v);
    return v;
  };
  const int V = H / 8, nq = V / 4;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  int i = 0;
  while (i + 16 <= nq) {
    for (int j = 0; j < 16; ++j, ++i) { const float a = ld(32 * i + sub); p0 = p0 + a * a; }
    p1 = p1 + p0; p0 = 0.f;
    if ((i & (15 << 4)) == 0) {
      p2 = p2 + p1; p1 = 0.f;
      if ((i & (15 << 8)) == 0) { p3 = p3 + p2; p2 = 0.f; }
    }
  }
  for (; i < nq; ++i) { const float a = ld(32 * i + sub); p0 = p0 + a * a; }
  p0 = p0 + p1; p0 = p0 + p2; p0 = p0 + p3;
  if (sub < 8)
    for (int v = nq * 4; v < V; ++v) { const float a = ld(v * 8 + sub); p0 = p0 + a * a; }
  const int base = (threadIdx.x & 63) & 32;   // first lane of this row's 32 inside the wave
  {
    const float q1 = __shfl(p0, base + 8 + (sub & 7)), q2 = __shfl(p0, base + 16 + (sub & 7)), q3 = __shfl(p0, base + 24 + (sub & 7));
    p0 = p0 + q1; p0 = p0 + q2; p0 = p0 + q3;   // meaningful in the threads of accumulator 0 (sub < 8)
  }
  float acc = 0.f;
  for (int h = V * 8; h < H; ++h) { const float a = ld(h); acc = acc + a * a; }   // (every thread: the tail is under 8 elements)
#pragma unroll
  for (int l = 0; l < 8; ++l) acc = acc + __shfl(p0, base + l);
  const float n = mkl_sqrtf_dev(acc, tab);
  float den, sc;
  prox_scales(n, t, den, sc);
  bool allzero = true;
  if (live) {
    for (int h = sub; h < H; h += 32) {
      float v = ld(h);
      if (do_prox) { v = v / den; v = v * sc; }
      w[h] = v;
      allzero = allzero && (v == 0.0f);
    }
  }
  int az = allzero ? 1 : 0;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) az &= __shfl_xor(az, o);
  if (live && sub == 0) {
    if (mask) mask[r] = (uint8_t)az;
    if (norms) norms[r] = n;
  }
}

// ---- column form: FirstLayerSparseDecoder.proximal_step (/root/reference/lib/models.py:65-71) ----
// The G latent-to-group maps (nn.Linear(K, H_g).weight, [H_g, K]) are stacked on rows; a column of one map
// is soft-thresholded by its norm over dim 0.  One thread per (map, column): consecutive threads read
// consecutive k, so every row of a map is one coalesced access.  The sum order is torch 2.10's CPU
// outer-reduction kernel, restated rule by rule in oracle/prox_oracle.c (multi_row_sumsq / row_sum_sumsq).
__device__ __forceinline__ float cols_multi_row(const float* __restrict__ w, int64_t stride, int n) {
  // aten multi_row_sum: level 0 flushed into level 1 every 16 rows, level 1 into level 2 every 256, level 2 into
  // level 3 every 4096 (two levels are the same sum below 272 rows, where the first recorded vectors ended)
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int i = 0;
  while (i + 16 <= n) {
    for (int j = 0; j < 16; ++j, ++i) { const float v = w[(int64_t)i * stride]; a0 = a0 + v * v; }
    a1 = a1 + a0;
    a0 = 0.0f;
    if ((i & (15 << 4)) == 0) {
      a2 = a2 + a1;
      a1 = 0.0f;
      if ((i & (15 << 8)) == 0) { a3 = a3 + a2; a2 = 0.0f; }
    }
  }
  for (; i < n; ++i) { const float v = w[(int64_t)i * stride]; a0 = a0 + v * v; }
  a0 = a0 + a1;
  a0 = a0 + a2;
  a0 = a0 + a3;
  return a0;
}

__global__ __launch_bounds__(256) void prox_cols(float* __restrict__ W, const int32_t* __restrict__ group_off,
                                                 int G, int K, float t, uint8_t* __restrict__ mask,
                                                 float* __restrict__ norms, const uint16_t* __restrict__ tab) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)G * K) return;
  const int g = (int)(idx / K), k = (int)(idx - (int64_t)g * K);
  const int r0 = group_off[g], H = group_off[g + 1] - r0;
  float* w = W + (int64_t)r0 * K + k;
  float acc;
  if (K == 1) {
    // contiguous inner reduction: the row kernel's order (8-lane vectors, 4 interleaved accumulators)
    if (H < 8) {
      float p[4] = {0.f, 0.f, 0.f, 0.f};
      const int nq = H / 4;
      for (int i = 0; i < nq; ++i)
        for (int q = 0; q < 4; ++q) { const float v = w[4 * i + q]; p[q] = p[q] + v * v; }
      for (int i = nq * 4; i < H; ++i) { const float v = w[i]; p[0] = p[0] + v * v; }
      p[0] = p[0] + p[1]; p[0] = p[0] + p[2]; p[0] = p[0] + p[3];
      acc = p[0];
    } else {
      float P[4][8];
      for (int q = 0; q < 4; ++q)
        for (int l = 0; l < 8; ++l) P[q][l] = 0.f;
      const int V = H / 8, nq = V / 4;
      for (int i = 0; i < nq; ++i)
        for (int q = 0; q < 4; ++q)
          for (int l = 0; l < 8; ++l) { const float v = w[(4 * i + q) * 8 + l]; P[q][l] = P[q][l] + v * v; }
      for (int i = nq * 4; i < V; ++i)
        for (int l = 0; l < 8; ++l) { const float v = w[i * 8 + l]; P[0][l] = P[0][l] + v * v; }
      for (int q = 1; q < 4; ++q)
        for (int l = 0; l < 8; ++l) P[0][l] = P[0][l] + P[q][l];
      acc = 0.f;
      for (int i = V * 8; i < H; ++i) { const float v = w[i]; acc = acc + v * v; }
      for (int l = 0; l < 8; ++l) acc = acc + P[0][l];
    }
  } else {
    const int n_seq = (K >= 8) ? (K / 32) * 32 : (K / 4) * 4;
    if (k < n_seq) {
      acc = cols_multi_row(w, K, H);
    } else {
      const int n4 = H / 4;
      float p[4];
      for (int q = 0; q < 4; ++q) p[q] = cols_multi_row(w + (int64_t)q * K, 4 * (int64_t)K, n4);
      for (int i = 4 * n4; i < H; ++i) { const float v = w[(int64_t)i * K]; p[0] = p[0] + v * v; }
      p[0] = p[0] + p[1]; p[0] = p[0] + p[2]; p[0] = p[0] + p[3];
      acc = p[0];
    }
  }
  const float n = mkl_sqrtf_dev(acc, tab);
  float den, sc;
  prox_scales(n, t, den, sc);
  bool allzero = true;
  for (int h = 0; h < H; ++h) {
    float v = w[(int64_t)h * K];
    v = v / den;
    v = v * sc;
    w[(int64_t)h * K] = v;
    allzero = allzero && (v == 0.0f);
  }
  if (mask) mask[idx] = allzero ? 1 : 0;
  if (norms) norms[idx] = n;
}

template <int VV, bool SGD>
static int launch_prox_vec(float* Ws, const float* dWs, int64_t rows, float neg_lr, float t, int do_prox, uint8_t* mask,
                           float* norms, const uint16_t* tab, hipStream_t st) {
  // path C from 4 passes of 256 workgroups up (512 rows each per pass), for rows of at most 128 bytes (two rows per
  // thread in registers under 1024 threads per workgroup; at 256 bytes the fused-SGD form spills)
  if constexpr (VV <= 4) {
    if (rows >= (int64_t)4 * 256 * (kProxLdsThreads / 2)) {
      OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&prox_rows_vec_lds<VV, SGD>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
      hipLaunchKernelGGL((prox_rows_vec_lds<VV, SGD>), dim3(256), dim3(kProxLdsThreads), 131072, st, Ws, dWs, rows,
                         neg_lr, t, do_prox, mask, norms, tab);
      return OIVAE_OK;
    }
  }
  const unsigned blocks = (unsigned)((rows * 2 + 255) / 256);
  hipLaunchKernelGGL((prox_rows_vec<VV, SGD>), dim3(blocks), dim3(256), 0, st, Ws, dWs, rows, neg_lr, t, do_prox, mask,
                     norms, tab);
  return OIVAE_OK;
}

template <bool SGD>
static int launch_prox(float* Ws, const float* dWs, int64_t rows, int H, float lr, float t,
                       int do_prox, uint8_t* mask, float* norms, hipStream_t st) {
  const uint16_t* tab = rsqrt14_table_device();
  if (!tab) return fail(OIVAE_E_NOINIT, "oivae_init has not been called on this device");
  if (rows == 0) return OIVAE_OK;
  const float neg_lr = -lr;
  const bool aligned = ((reinterpret_cast<uintptr_t>(Ws) & 15) == 0) &&
                       (!SGD || (reinterpret_cast<uintptr_t>(dWs) & 15) == 0);
#define OIVAE_PROX_VEC(VV)                                                                        \
  { if (int rc = launch_prox_vec<VV, SGD>(Ws, dWs, rows, neg_lr, t, do_prox, mask, norms, tab, st)) return rc; }
  if (aligned && H == 8) OIVAE_PROX_VEC(1)
  else if (aligned && H == 16) OIVAE_PROX_VEC(2)
  else if (aligned && H == 32) OIVAE_PROX_VEC(4)
  else if (aligned && H == 64) OIVAE_PROX_VEC(8)
  else if (aligned && H == 128) OIVAE_PROX_VEC(16)
  else if (H >= 512) {
    const unsigned blocks = (unsigned)((rows * 32 + 255) / 256);
    hipLaunchKernelGGL((prox_rows_wide<SGD>), dim3(blocks), dim3(256), 0, st, Ws, dWs, rows, H,
                       neg_lr, t, do_prox, mask, norms, tab);
  } else {
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    hipLaunchKernelGGL((prox_rows_scalar<SGD>), dim3(blocks), dim3(256), 0, st, Ws, dWs, rows, H,
                       neg_lr, t, do_prox, mask, norms, tab);
  }
#undef OIVAE_PROX_VEC
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

}  // namespace oivae

using namespace oivae;

extern "C" int oivae_prox_f32(float* Ws, int64_t rows, int32_t H, float t, uint8_t* zero_mask,
                              float* row_norms, void* stream) {
  if (!Ws || rows < 0) return fail(OIVAE_E_ARG, "oivae_prox_f32: null Ws or negative rows");
  if (H < 1 || H > (1 << 20))
    return fail(OIVAE_E_SHAPE, "oivae_prox_f32: H=%d outside [1, 2^20] (the reference's row-sum order is pinned only there)", H);
  if (rows * 2 / 256 > 0x7fffffffLL) return fail(OIVAE_E_SHAPE, "oivae_prox_f32: too many rows");
  if (int rc = poll_error_word("oivae_prox_f32", (hipStream_t)stream)) return rc;
  return launch_prox<false>(Ws, nullptr, rows, H, 0.f, t, 1, zero_mask, row_norms, (hipStream_t)stream);
}

extern "C" int oivae_sgd_prox_f32(float* Ws, const float* dWs, int64_t rows, int32_t H, float lr,
                                  float t, int32_t do_prox, uint8_t* zero_mask, float* row_norms,
                                  void* stream) {
  if (!Ws || !dWs || rows < 0) return fail(OIVAE_E_ARG, "oivae_sgd_prox_f32: null pointer or negative rows");
  if (H < 1 || H > (1 << 20))
    return fail(OIVAE_E_SHAPE, "oivae_sgd_prox_f32: H=%d outside [1, 2^20]", H);
  if (rows * 2 / 256 > 0x7fffffffLL) return fail(OIVAE_E_SHAPE, "oivae_sgd_prox_f32: too many rows");
  if (int rc = poll_error_word("oivae_sgd_prox_f32", (hipStream_t)stream)) return rc;
  return launch_prox<true>(Ws, dWs, rows, H, lr, t, do_prox ? 1 : 0, zero_mask, row_norms, (hipStream_t)stream);
}

extern "C" int oivae_prox_cols_f32(float* W, const int32_t* group_off, int32_t G, int32_t K, int32_t max_group_rows,
                                   float t, uint8_t* zero_mask, float* col_norms, void* stream) {
  if (!W || !group_off || G < 0 || K < 1) return fail(OIVAE_E_ARG, "oivae_prox_cols_f32: null pointer or bad sizes");
  if (max_group_rows < 1 || (int64_t)max_group_rows * K >= 32768 || (K == 1 && max_group_rows >= 512))
    return fail(OIVAE_E_SHAPE, "oivae_prox_cols_f32: H_g=%d, K=%d outside the range where the reference's column-sum "
                               "order is pinned (H_g*K < 32768; K = 1: H_g < 512)", max_group_rows, K);
  const uint16_t* tab = rsqrt14_table_device();
  if (!tab) return fail(OIVAE_E_NOINIT, "oivae_init has not been called on this device");
  if (G == 0) return OIVAE_OK;
  const int64_t n = (int64_t)G * K;
  hipLaunchKernelGGL(prox_cols, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, group_off, G, K, t,
                     zero_mask, col_norms, tab);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}
