// Device helpers of the proximal step shared between prox.hip and small_step.hip (both built with -ffp-contract=off:
// the arithmetic below is pinned bit for bit to the reference's CPU path, see prox.hip's header).
#pragma once
#include "common.h"

namespace oivae {

template <typename TabPtr>
__device__ __forceinline__ float mkl_sqrtf_dev(float x, TabPtr tab) {
  if (!(x > 0.0f) || x == __builtin_inff()) return x;  // 0, NaN, +inf pass through (x is a sum of squares)
  float back = 1.0f;
  if (x < 0x1p-100f) { x *= 0x1p64f; back = 0x1p-32f; }
  const uint32_t u = __float_as_uint(x);
  const int e = (int)(u >> 23) - 127;
  const uint32_t m = u & 0x7fffffu;
  const int par = e & 1;
  const int half = (e - par) >> 1;
  float y;
  if (par == 0 && m == 0) {
    y = __uint_as_float((uint32_t)(127 - half) << 23);
  } else {
    const uint32_t frac = tab[par * 32768 + (m >> 8)];
    y = __uint_as_float(((uint32_t)(126 - half) << 23) | (frac << 7));
  }
  const float s = x * y;
  const float h = 0.5f * y;
  const float d = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(d, h, s) * back;
}

__device__ __forceinline__ void prox_scales(float n, float t, float& den, float& sc) {
  den = (n < 1e-16f) ? 1e-16f : n;  // clamp(n, min=1e-16); NaN passes through
  sc = n - t;
  if (sc < 0.0f) sc = 0.0f;         // clamp(n - t, min=0); NaN passes through
}

// One row of the proximal step by one thread, any H: the reference's fp32 row-sum order restated scalar by scalar
// (derivation: the header of prox.hip), optionally behind the SGD update w = fma(-lr, g, w).  Shared by prox_rows_scalar (prox.hip)
// and the small-model step kernel (small_step.hip); both translation units are built with -ffp-contract=off.
template <bool SGD, typename TabPtr>
__device__ __forceinline__ void prox_row_scalar_dev(float* __restrict__ Ws, const float* __restrict__ dWs, int64_t r, int H,
                                                    float neg_lr, float t, int do_prox, uint8_t* __restrict__ mask,
                                                    float* __restrict__ norms, TabPtr tab) {
  float* w = Ws + r * H;
  const float* g = SGD ? dWs + r * H : nullptr;
  auto ld = [&](int i) -> float {
    float v = w[i];
    if (SGD) v = __builtin_fmaf(neg_lr, g[i], v);
    return v;
  };
  float acc;
  if (H < 8) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    const int nq = H / 4;
    for (int i = 0; i < nq; ++i) {
      float a0 = ld(4 * i), a1 = ld(4 * i + 1), a2 = ld(4 * i + 2), a3 = ld(4 * i + 3);
      p0 = p0 + a0 * a0; p1 = p1 + a1 * a1; p2 = p2 + a2 * a2; p3 = p3 + a3 * a3;
    }
    for (int i = nq * 4; i < H; ++i) { float a = ld(i); p0 = p0 + a * a; }
    p0 = p0 + p1; p0 = p0 + p2; p0 = p0 + p3;
    acc = p0;
  } else {
    float P[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int l = 0; l < 8; ++l) P[k][l] = 0.f;
    const int V = H / 8, nq = V / 4;
    for (int i = 0; i < nq; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 8; ++l) { float a = ld((4 * i + k) * 8 + l); P[k][l] = P[k][l] + a * a; }
    }
    for (int i = nq * 4; i < V; ++i)
#pragma unroll
      for (int l = 0; l < 8; ++l) { float a = ld(i * 8 + l); P[0][l] = P[0][l] + a * a; }
#pragma unroll
    for (int k = 1; k < 4; ++k)
#pragma unroll
      for (int l = 0; l < 8; ++l) P[0][l] = P[0][l] + P[k][l];
    acc = 0.f;
    for (int i = V * 8; i < H; ++i) { float a = ld(i); acc = acc + a * a; }
#pragma unroll
    for (int l = 0; l < 8; ++l) acc = acc + P[0][l];
  }
  const float n = mkl_sqrtf_dev(acc, tab);
  float den, sc;
  prox_scales(n, t, den, sc);
  bool allzero = true;
  for (int i = 0; i < H; ++i) {
    float v = ld(i);
    if (do_prox) { v = v / den; v = v * sc; }
    w[i] = v;
    allzero = allzero && (v == 0.0f);
  }
  if (mask) mask[r] = allzero ? 1 : 0;
  if (norms) norms[r] = n;
}

}  // namespace oivae
