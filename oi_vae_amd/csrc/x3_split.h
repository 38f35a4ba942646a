// fp32 operands as three bf16 terms for v_mfma_f32_32x32x16_bf16 (gfx950): shared by x3_encoder.hip and the
// x3 decoder kernel.  x = hi + mid + lo exactly (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), RNE);
// a product of two such operands is accumulated in fp32 from the six partial products of weight >= 2^-16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oivae {

using x3_f32x16 = __attribute__((ext_vector_type(16))) float;
using x3_f32x4 = __attribute__((ext_vector_type(4))) float;
using x3_u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using x3_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// low half = bf16(a), high half = bf16(b), round to nearest even: one v_cvt_pk_bf16_f32.  Written as a vector
// conversion, not inline asm, so that the compiler knows the instruction (hazards against the MFMAs that read it).
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// (a, b) -> packed bf16 pairs of the three terms
struct Split3 { uint32_t h, m, l; };
__device__ __forceinline__ Split3 split_pair(float a, float b) {
  Split3 r;
  r.h = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(r.h << 16), rb = b - __uint_as_float(r.h & 0xffff0000u);
  r.m = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(r.m << 16), sb = rb - __uint_as_float(r.m & 0xffff0000u);
  r.l = cvt_pk_bf16(sa, sb);
  return r;
}

// one MFMA operand (8 reduction slots per lane) in its three terms
struct Frag3 { x3_u32x4 p[3]; };

__device__ __forceinline__ Frag3 split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  const Split3 a = split_pair(v0, v1), b = split_pair(v2, v3), c = split_pair(v4, v5), d = split_pair(v6, v7);
  Frag3 f;
  f.p[0] = x3_u32x4{a.h, b.h, c.h, d.h};
  f.p[1] = x3_u32x4{a.m, b.m, c.m, d.m};
  f.p[2] = x3_u32x4{a.l, b.l, c.l, d.l};
  return f;
}

__device__ __forceinline__ x3_f32x16 mfma_bf16(x3_u32x4 a, x3_u32x4 b, x3_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3_bf16x8, a), __builtin_bit_cast(x3_bf16x8, b), c, 0, 0, 0);
}

// acc += A * B over 16 reduction slots: the six partial products, small terms first
__device__ __forceinline__ void x3mm(x3_f32x16& acc, const Frag3& A, const Frag3& B) {
  acc = mfma_bf16(A.p[2], B.p[0], acc);
  acc = mfma_bf16(A.p[0], B.p[2], acc);
  acc = mfma_bf16(A.p[1], B.p[1], acc);
  acc = mfma_bf16(A.p[1], B.p[0], acc);
  acc = mfma_bf16(A.p[0], B.p[1], acc);
  acc = mfma_bf16(A.p[0], B.p[0], acc);
}

// ---- two fp16 terms per operand ("h2"): x * scale = hi + lo with hi = fp16(x s), lo = fp16(x s - hi) (11 + 11
// significand bits); a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulation.  As accurate
// as the fp32 matrix instruction WHEN max|operand| * scale sits in [2^8, 2^15] (tools/probe/f16x2_accuracy.hip):
// the caller owns the scale (a power of two, so scaling is exact).
typedef _Float16 x3_h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 x3_h16x8 __attribute__((ext_vector_type(8)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
struct Split2 { uint32_t h, l; };
__device__ __forceinline__ Split2 split2_pair(float a, float b) {   // a, b already scaled
  Split2 r;
  const x3_h16x2 h = __builtin_convertvector(x3_f32x2{a, b}, x3_h16x2);
  const x3_f32x2 hf = __builtin_convertvector(h, x3_f32x2);
  const x3_h16x2 l = __builtin_convertvector(x3_f32x2{a - hf[0], b - hf[1]}, x3_h16x2);
  r.h = __builtin_bit_cast(uint32_t, h);
  r.l = __builtin_bit_cast(uint32_t, l);
  return r;
}
// The same split in four instructions per pair instead of six, for callers whose a, b are not themselves the result of
// a multiplication the compiler can fuse: a - float(h) is written as a * one - float(h) with a 1.0 the compiler cannot
// see through, which is one v_fma_mix_f32 (fp16 operand converted on the fly) instead of a conversion and a
// subtraction, and as exact.  (The decoder's chain wave is bound by its vector instructions: -1 ms per launch.)
// (Tried: the low term written element by element, which the compiler turns into v_fma_mixlo_f16 / v_fma_mixhi_f16 --
// three instructions per pair, same bits, 168 fewer vector instructions in dec_x3_64s: 11.48 against 11.34 ms on one
// box, SLOWER -- the second one waits for the first one's partial write of the same register.)
__device__ __forceinline__ Split2 split2_pair_mix(float a, float b) {
  Split2 r;
  const x3_h16x2 h = __builtin_convertvector(x3_f32x2{a, b}, x3_h16x2);
  float one = 1.0f;
  asm("" : "+s"(one));
  const x3_h16x2 l = __builtin_convertvector(x3_f32x2{__builtin_fmaf(a, one, -(float)h[0]), __builtin_fmaf(b, one, -(float)h[1])}, x3_h16x2);
  r.h = __builtin_bit_cast(uint32_t, h);
  r.l = __builtin_bit_cast(uint32_t, l);
  return r;
}
__device__ __forceinline__ x3_f32x16 mfma_f16(x3_u32x4 a, x3_u32x4 b, x3_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_h16x8, a), __builtin_bit_cast(x3_h16x8, b), c, 0, 0, 0);
}
// power-of-two scale that brings a maximum with biased exponent E to [2^13, 2^14); 1 for zero, denormal, tiny (< 2^-113)
// or non-finite maxima.  `inv` is its reciprocal.
__device__ __forceinline__ void h2_scale_of(uint32_t absmax_bits, float* scale, float* inv) {
  const uint32_t E = (absmax_bits >> 23) & 0xffu;
  const bool ok = E >= 14u && E <= 254u;
  *scale = ok ? __uint_as_float((267u - E) << 23) : 1.0f;
  *inv = ok ? __uint_as_float((E - 13u) << 23) : 1.0f;
}

// one MFMA operand in its two fp16 terms (inputs already scaled)
struct Frag2 { x3_u32x4 p[2]; };
__device__ __forceinline__ Frag2 split8_h2(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  const Split2 a = split2_pair_mix(v0, v1), b = split2_pair_mix(v2, v3), c = split2_pair_mix(v4, v5), d = split2_pair_mix(v6, v7);
  Frag2 f;
  f.p[0] = x3_u32x4{a.h, b.h, c.h, d.h};
  f.p[1] = x3_u32x4{a.l, b.l, c.l, d.l};
  return f;
}
// acc += A * B over 16 reduction slots: lo*hi + hi*lo + hi*hi
__device__ __forceinline__ void h2mm(x3_f32x16& acc, const Frag2& A, const Frag2& B) {
  acc = mfma_f16(A.p[1], B.p[0], acc);
  acc = mfma_f16(A.p[0], B.p[1], acc);
  acc = mfma_f16(A.p[0], B.p[0], acc);
}

}  // namespace oivae
