// Encoder GEMMs with fp32 operands carried as three bf16 terms each ("x3" kernels).
//
// Same products as mfma_encoder.hip -- forward pre[B, 2K] = X * [Wmu; Wsig]^T
// (/root/reference/lib/models.py:15-16), backward [dWmu; dWsig] = dpre^T * X
// (lib/oivae.py:75) -- but issued on the bf16 matrix instruction
// v_mfma_f32_32x32x16_bf16, which retires 16x the flops per cycle of
// v_mfma_f32_32x32x2_f32 on gfx950.  Every fp32 operand x is written EXACTLY as
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// (round to nearest even; the two subtractions are exact in fp32, 3 x 8 significand
// bits cover fp32's 24), and a product a*b is accumulated in fp32 from the six
// partial products of weight >= 2^-16:
//     a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid .
// The three dropped ones (mid*lo, lo*mid, lo*lo) are below 2^-24 |a b|, i.e. below
// the rounding of the fp32 product itself; each partial product of two bf16 numbers
// is exact in fp32 and the matrix unit accumulates in fp32, so the result carries the
// same error as the fp32 MFMA chain (measured: tests/test_gpu_scale.py).
// Non-finite inputs give NaN where the fp32 kernels give +-inf (inf - inf in the split).
//
// The streamed operand (X, 17 GB) is split in registers on its way into LDS; the small
// operand (the weights / the pre-activation gradients) is split once per step into an
// image laid out the way LDS wants it.  Roofline: 6/16 of the fp32 MFMA time, which at
// the scale-up shape puts both GEMMs at the HBM time of reading X once.
#include "common.h"
#include "x3_split.h"

#include <stdlib.h>

namespace oivae {

using f32x16 = x3_f32x16;
using f32x4 = x3_f32x4;
using u32x4 = x3_u32x4;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

#define OIVAE_ROW(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))

// six partial products of one 32x32x16 step; fa/fb = {hi, mid, lo}
template <int NI, int NJ>
__device__ __forceinline__ void x3_step(f32x16 (&acc)[NI][NJ], const u32x4 (&fa)[NI][3], const u32x4 (&fb)[NJ][3]) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // small terms first
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_bf16(fa[i][PA[q]], fb[j][PB[q]], acc[i][j]);
}

// ------------------------------------------------------------------------------
// Split image of a row-major operand W[rows][Kd] (reduction index contiguous):
//   img[row][kt][s][piece][16]  (bf16), kt = k / 32, s = (k / 16) & 1, inside: k & 15
// rows beyond `rows` and columns beyond Kd are zero.  One thread per 8 consecutive k.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void x3_split_rows(const float* __restrict__ W0, const float* __restrict__ W1,
                                                     int rows0, int rows, int Kd, int rows_pad, int KT,
                                                     uint32_t* __restrict__ img, const uint32_t* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;  // the fp16 forward kernel's result stands (h2_fwd_guard)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)rows_pad * KT * 4;  // 8-wide chunks
  if (i >= total) return;
  const int row = (int)(i / (KT * 4)), c = (int)(i % (KT * 4));
  const int kt = c >> 2, s = (c >> 1) & 1, h = c & 1;
  const int k = 8 * c;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (row < rows) {
    const float* src = (row < rows0 ? W0 + (int64_t)row * Kd : W1 + (int64_t)(row - rows0) * Kd);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (k + e < Kd) v[e] = src[k + e];
  }
  u32x4 P[3];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const Split3 q = split_pair(v[2 * e], v[2 * e + 1]);
    P[0][e] = q.h; P[1][e] = q.m; P[2][e] = q.l;
  }
  uint32_t* dst = img + (((int64_t)row * KT + kt) * 2 + s) * 24 + 4 * h;  // 24 dwords per (row, kt, s)
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + 8 * p) = P[p];
}

// ------------------------------------------------------------------------------
// forward: pre[b][j] = sum_d X[b][d] * W[j][d];  workgroup tile 128 rows x 128 columns, reduction in macro tiles
// of 32 (two MFMA steps of 16).  Twelve waves with three roles (one of each per SIMD):
//   waves 0-3   consumers: 64 x 64 of the tile each; LDS operand reads and the 48 MFMAs of a macro tile, nothing else;
//   waves 4-7   X producers: global loads of X four macro tiles ahead (HBM latency under load is ~3 us), the
//               fp32 -> 3 x bf16 split, LDS writes;
//   waves 8-11  W producers: the weight image from L2, four tiles ahead, copied to LDS.  (Loads return in order per
//               wave: in one queue the L2 hits would wait behind the HBM loads.)
// The producers' VALU/LDS/VMEM instructions issue in the shadow of the consumers' MFMAs (tools/probe/mfma_probe3).
// LDS holds two macro tiles (113 KB); ONE barrier per macro tile: before it the consumers have read tile t into
// registers and the producers have written tile t+1, after it the consumers' remaining MFMAs of tile t run from
// registers while tile t+1 is read, so no LDS latency is exposed either.
// LDS row r of sub-step s = 28 dwords: piece p at +8p, lane-half h at +4h (8 bf16 = one MFMA operand).
// ------------------------------------------------------------------------------
constexpr int kX3Row = 28;                      // dwords per LDS row: 24 + 4 pad (28*l mod 64: 16 distinct multiples of 4)
constexpr int kX3Sub = 128 * kX3Row + 32;         // one sub-step of one operand; +32 dwords so that the two sub-steps a
                                                 // producer store straddles fall on opposite halves of the 64 banks
constexpr int kX3Buf = 4 * kX3Sub;              // dwords per macro tile: X [2][kX3Sub] | W [2][kX3Sub]
constexpr int kX3StagesX = 4;                   // X macro tiles in flight per producer
constexpr int kX3StagesW = 4;

// BWD = false: out0 = pre [B, J] (+ split-K images), M side = rows of X (k = d contiguous), N side = weight image.
// BWD = true : out0/out1 = dWmu/dWsig [K, D]; the reduction runs over the batch: M side = 128 columns d of X
//              (transposed on the way into LDS: bf16 pairs are formed from two consecutive rows; LDS row
//              rho = 32 (d & 3) + (d >> 2) keeps those writes conflict-free), N side = the image of dpre^T.
//              B is a multiple of 32 and D of 128 on this path.
template <bool BWD>
__global__ __launch_bounds__(768, 1) void enc_x3_gemm(const float* __restrict__ X, int64_t B, int D, int J,
                                                      const uint32_t* __restrict__ Wimg, int KT,
                                                      float* __restrict__ out0, float* __restrict__ out1, int klen,
                                                      uint32_t* __restrict__ rng, const uint32_t* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;  // uniform, before any barrier: the fp16 forward kernel's result stands
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem_raw);  // [2][kX3Buf]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * 128;  // first row of X (forward) / first column d (backward)
  const int j0 = blockIdx.y * 128;
  const int kbeg = BWD ? 0 : blockIdx.z * klen;
  const int kend = BWD ? (int)B : ((kbeg + klen < D) ? kbeg + klen : D);
  const int T = (kend - kbeg) / 32;  // D (forward) / B (backward) is a multiple of 32 on this path
  // every role passes the same number of barriers: T rounded up to a multiple of 4 (the producers' loops are
  // unrolled over their register rings and contain no branch: a branch around a memory instruction makes the
  // compiler drain the prefetch queue; tiles past the end are copies of the last one, written to a buffer
  // that nobody reads any more)
  const int Tr = (T + 3) & ~3;
  float* pre = out0;
  if (!BWD) pre += (int64_t)blockIdx.z * B * J;

  if (wave >= 8) {
    // ---------------- W producers ----------------
    const int tid = threadIdx.x - 512;
    const uint32_t* wptr[6];
    int woff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int f = tid + 256 * i, row = f / 12, ch = f % 12;
      const int s = ch / 6, p = (ch % 6) >> 1, h = ch & 1;
      wptr[i] = Wimg + ((int64_t)(j0 + row) * KT + (kbeg >> 5)) * 48 + 4 * ch;
      woff[i] = 2 * kX3Sub + s * kX3Sub + row * kX3Row + 8 * p + 4 * h;
    }
    u32x4 rw[kX3StagesW][6];
    auto wload = [&](int t, u32x4 (&r)[6]) {
      const int64_t kt = (t < T ? t : T - 1);
#pragma unroll
      for (int i = 0; i < 6; ++i) r[i] = *reinterpret_cast<const u32x4*>(wptr[i] + kt * 48);
    };
    auto wstore = [&](int buf, const u32x4 (&w)[6]) {
#pragma unroll
      for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4*>(lds + buf * kX3Buf + woff[i]) = w[i];
    };
    // tile u lives in rw[u % 4]
    wload(0, rw[0]); wload(1, rw[1]); wload(2, rw[2]); wload(3, rw[3]);
    wstore(0, rw[0]);
    wload(4, rw[0]);
    __syncthreads();  // P_{-1}: tile 0 is in LDS
    auto step = [&](int t, u32x4 (&w)[6]) {  // writes tile t+1, refills with tile t+5
      wstore((t + 1) & 1, w);
      wload(t + 5, w);
      __syncthreads();  // P_t
    };
    for (int t = 0; t < Tr; t += 4) {
      step(t, rw[1]); step(t + 1, rw[2]); step(t + 2, rw[3]); step(t + 3, rw[0]);
    }
  } else if (wave >= 4) {
    // ---------------- X producers ----------------
    const int tid = threadIdx.x - 256;
    const float* xptr[4];
    int xoff;
    if (!BWD) {
      // rows of a 16-lane group two apart (low two bits of the row index swapped): with consecutive rows the 8-byte
      // LDS stores of the two sub-steps collide pairwise (64 instead of 32 cycles per store, tools/probe/lds_b128.hip)
      const int xr0 = tid >> 3, xc = tid & 7;
      const int xr = (xr0 & ~3) | ((xr0 & 1) << 1) | ((xr0 >> 1) & 1);
      // rows beyond B re-read row B-1 (their results are never stored): no load is conditional
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + xr + 32 * i;
        xptr[i] = X + (row < B ? row : B - 1) * (int64_t)D + kbeg + 4 * xc;
      }
      xoff = (xc >> 2) * kX3Sub + xr * kX3Row + 4 * ((xc >> 1) & 1) + 2 * (xc & 1);
    } else {
      const int c = tid & 31, rg = tid >> 5;  // columns d = m0 + 4c..4c+3, rows 4 rg..4 rg+3 of the macro tile
#pragma unroll
      for (int i = 0; i < 4; ++i) xptr[i] = X + (int64_t)(4 * rg + i) * D + m0 + 4 * c;
      xoff = (rg >> 2) * kX3Sub + c * kX3Row + 4 * ((rg >> 1) & 1) + 2 * (rg & 1);
    }
    const int64_t xstep = BWD ? 32 * (int64_t)D : 32;
    f32x4 rx[kX3StagesX][4];
    auto xload = [&](int t, f32x4 (&r)[4]) {  // tiles past the end re-read the last one (never used)
      const int64_t k0 = xstep * (t < T ? t : T - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xptr[i] + k0));
    };
    float xmax = 0.f;  // forward: max |x| seen by this lane, for the fp16 backward kernel's scale
    auto xstore = [&](int buf, const f32x4 (&r)[4]) {
      uint32_t* base = lds + buf * kX3Buf;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!BWD) xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(r[i][0]), fabsf(r[i][1]))), fmaxf(fabsf(r[i][2]), fabsf(r[i][3])));
        u32x2 P[3];
        // forward: r[i] = 4 consecutive k of row i; backward: column i of 4 consecutive rows (k) -> LDS row 32 i + c
        const Split3 q0 = BWD ? split_pair(r[0][i], r[1][i]) : split_pair(r[i][0], r[i][1]);
        const Split3 q1 = BWD ? split_pair(r[2][i], r[3][i]) : split_pair(r[i][2], r[i][3]);
        P[0] = u32x2{q0.h, q1.h}; P[1] = u32x2{q0.m, q1.m}; P[2] = u32x2{q0.l, q1.l};
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(base + xoff + 32 * i * kX3Row + 8 * p) = P[p];
      }
    };
    // tile u lives in rx[u % 4]
    xload(0, rx[0]); xload(1, rx[1]); xload(2, rx[2]); xload(3, rx[3]);
    xstore(0, rx[0]);
    xload(4, rx[0]);
    __syncthreads();  // P_{-1}
    auto step = [&](int t, f32x4 (&r)[4]) {  // writes tile t+1, refills with tile t+5
      xstore((t + 1) & 1, r);
      xload(t + 5, r);
      __syncthreads();  // P_t
    };
    for (int t = 0; t < Tr; t += 4) {
      step(t, rx[1]); step(t + 1, rx[2]); step(t + 2, rx[3]); step(t + 3, rx[0]);
    }
    if (!BWD && rng != nullptr) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o));
      if (lane == 0) atomic_max_if_larger(rng, __float_as_uint(xmax));  // non-negative floats order like their bit patterns
    }
  } else {
    // ---------------- consumers ----------------
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t* a0 = lds + (wm * 64 + l31) * kX3Row + 4 * hh;
    const uint32_t* b0 = lds + 2 * kX3Sub + (wn * 64 + l31) * kX3Row + 4 * hh;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);
    u32x4 fa[2][2][3], fb[2][2][3];  // [sub-step][tile][piece]
    auto lread = [&](int buf, int s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[s][i][p] = *reinterpret_cast<const u32x4*>(a0 + buf * kX3Buf + s * kX3Sub + 32 * i * kX3Row + 8 * p);
          fb[s][i][p] = *reinterpret_cast<const u32x4*>(b0 + buf * kX3Buf + s * kX3Sub + 32 * i * kX3Row + 8 * p);
        }
    };
    __syncthreads();  // P_{-1}
    for (int t = 0; t < T; ++t) {
      lread(t & 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t > 0) x3_step<2, 2>(acc, fa[1], fb[1]);  // second half of tile t-1, from registers
      __builtin_amdgcn_sched_barrier(0);
      lread(t & 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      x3_step<2, 2>(acc, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // P_t: tile t is in registers, tile t+1 is in LDS
    }
    x3_step<2, 2>(acc, fa[1], fb[1]);
    for (int t = T; t < Tr; ++t) __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = j0 + (wn * 2 + j) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mrow = (wm * 2 + i) * 32 + OIVAE_ROW(r, hh);
          if (!BWD) {
            const int64_t row = m0 + mrow;
            if (row < B && col < J) pre[row * J + col] = acc[i][j][r];
          } else {
            const int64_t d = m0 + 4 * (mrow & 31) + (mrow >> 5);  // LDS row rho -> column of X
            const int Kh = J / 2;
            if (col < J) (col < Kh ? out0 + (int64_t)col * D : out1 + (int64_t)(col - Kh) * D)[d] = acc[i][j][r];
          }
        }
      }
  }
}

// ------------------------------------------------------------------------------
// Backward GEMM on two fp16 terms per operand ("h2", x3_split.h): [dWmu; dWsig] = dpre^T * X with THREE fp16 matrix
// products per fp32 product instead of six bf16 ones -- half the matrix energy on a chip that runs these kernels at
// its power limit.  fp16 has no exponent range to spare, so both operands are scaled by powers of two
// taken from their true maxima: max|X| was recorded by the forward kernel's producers while they streamed the same X
// (rng[0]), max|dpre| by absmax_f32 below (rng[2]); the result is scaled back in the epilogue.  Same tile, roles and
// barrier schedule as enc_x3_gemm<true>; LDS rows are 20 dwords (2 pieces x 8 + 4 pad: conflict-free 16-byte reads).
// ------------------------------------------------------------------------------
constexpr int kH2Row = 20;
constexpr int kH2Sub = 128 * kH2Row + 16;   // sub-step images 16 banks apart (stores bank over 32 dwords): the producers' stores stay conflict-free
constexpr int kH2Buf = 4 * kH2Sub;
// Second level of the consumers' running sums (H2Consumer16): 64 registers x 64 lanes of fp32 per consumer wave behind
// the two macro-tile buffers, and the number of macro tiles (reduction steps of 32) between two moves into it.
constexpr int kH2Acc2Floats = 4 * 64 * 64;
constexpr int kH2FlushTiles = 64;
constexpr size_t kH2LdsBytes = sizeof(uint32_t) * 2 * kH2Buf + sizeof(float) * kH2Acc2Floats;

// max|v| over one array, or over two of the same length (blockIdx.y), into one word
__global__ __launch_bounds__(256) void absmax_f32(const float* __restrict__ v0, const float* __restrict__ v1, int64_t n,
                                                  uint32_t* __restrict__ out) {
  const float* __restrict__ v = blockIdx.y ? v1 : v0;
  float m = 0.f;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  if ((reinterpret_cast<uintptr_t>(v) & 15) == 0) {   // 16 bytes per lane, scalar tail
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t n4 = n >> 2;
    const f4* __restrict__ v4 = reinterpret_cast<const f4*>(v);
    int64_t i = tid;
    for (; i + 7 * nth < n4; i += 8 * nth) {   // eight loads in flight per lane: few workgroups must cover the latency
      f4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = v4[i + u * nth];
#pragma unroll
      for (int u = 0; u < 8; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(q[u][0]), fabsf(q[u][1]))), fmaxf(fabsf(q[u][2]), fabsf(q[u][3])));
    }
    for (; i < n4; i += nth) {
      const f4 q = v4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(q[0]), fabsf(q[1]))), fmaxf(fabsf(q[2]), fabsf(q[3])));
    }
    for (int64_t i = 4 * n4 + tid; i < n; i += nth) m = fmaxf(m, fabsf(v[i]));
  } else {
    for (int64_t i = tid; i < n; i += nth) m = fmaxf(m, fabsf(v[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float part[4];   // at most one atomic per workgroup: thousands of atomics on one address take 14 ns each
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    // ... and none when the word already holds as much (a plain coherent load: of n workgroups with like maxima only
    // about ln n raise the word)
    const uint32_t mine = __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
    if (__hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(out, mine);
  }
}

// max|P[:, j]| of P [B, J] (J a multiple of 128) for every column j, into colmax[j] (zeroed by the caller): the scales
// of the dpre^T image are per COLUMN of dpre -- per row of the weight gradient, i.e. per encoder unit -- so that a
// unit whose gradient lies far below the others' (a saturated sigma head: 1e-15 beside 1e+7, seen in
// tests/test_gpu_scale.py's wide-prox state) keeps its own 22 bits instead of being flushed by the matrix maximum.
// Thread = column (tid mod 128) + 128 c, two row lanes per workgroup; one atomic per column and workgroup, skipped
// when the word already holds as much.
__global__ __launch_bounds__(256) void absmax_cols_f32(const float* __restrict__ P, int64_t B, int J,
                                                       uint32_t* __restrict__ colmax) {
  __shared__ float part[128];
  const int jl = threadIdx.x & 127, rl = threadIdx.x >> 7;
  const int64_t rpb = (B + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = (r0 + rpb < B) ? r0 + rpb : B;
  for (int jc = 0; jc < J; jc += 128) {
    float m = 0.f;
    int64_t r = r0 + rl;
    for (; r + 6 < r1; r += 8) {   // four rows in flight per lane
      const float a0 = P[r * J + jc + jl], a1 = P[(r + 2) * J + jc + jl], a2 = P[(r + 4) * J + jc + jl], a3 = P[(r + 6) * J + jc + jl];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(a0), fabsf(a1))), fmaxf(fabsf(a2), fabsf(a3)));
    }
    for (; r < r1; r += 2) m = fmaxf(m, fabsf(P[r * J + jc + jl]));
    __syncthreads();
    if (rl == 1) part[jl] = m;
    __syncthreads();
    if (rl == 0) atomic_max_if_larger(colmax + jc + jl, __float_as_uint(fmaxf(m, part[jl])));
  }
}

// image of dpre^T in two fp16 terms: img[j][bt][s][piece][h][4 dwords], row j scaled by the power of two of colmax[j]
__global__ __launch_bounds__(256) void h2_split_cols(const float* __restrict__ P, int64_t B, int J, int rows_pad, int BT,
                                                     uint32_t* __restrict__ img, const uint32_t* __restrict__ colmax) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)rows_pad * BT * 4;
  if (i >= total) return;
  const int j = (int)(i % rows_pad);
  float sc, inv;
  h2_scale_of(colmax[j], &sc, &inv);
  const int64_t c = i / rows_pad;
  const int64_t bt = c >> 2;
  const int s = (int)(c >> 1) & 1, h = (int)c & 1;
  const int64_t b = 8 * c;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (j < J && b + e < B) ? P[(b + e) * J + j] * sc : 0.f;
  u32x4 Q[2];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const Split2 q = split2_pair(v[2 * e], v[2 * e + 1]);
    Q[0][e] = q.h; Q[1][e] = q.l;
  }
  uint32_t* dst = img + (((int64_t)j * BT + bt) * 2 + s) * 16 + 4 * h;
#pragma unroll
  for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(dst + 8 * p) = Q[p];
}

template <int NI, int NJ>
__device__ __forceinline__ void h2_step(f32x16 (&acc)[NI][NJ], const u32x4 (&fa)[NI][2], const u32x4 (&fb)[NJ][2]) {
  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};  // lo*hi, hi*lo, hi*hi
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_f16(fa[i][PA[q]], fb[j][PB[q]], acc[i][j]);
}

// ------------------------------------------------------------------------------
// Consumer wave of the h2 kernels on v_mfma_f32_16x16x32_f16.  On this power-limited chip the 16x16x32 shape delivers
// 13 % more flops than 32x32x16 on random operands (tools/probe/mfma_shape_power.hip: 2006 against 1772 TFLOP/s).
// A wave owns 64 x 64 outputs = 4 x 4 tiles of 16 x 16 (64 accumulator registers, as before); one macro tile of the
// operands is one reduction step of 32.  Lane (l15 = lane & 15, kb = lane >> 4) holds row pi(l15) of a 16-row block and
// the 8 reduction slots of chunk kb: chunks 0..3 are (sub-step image, half) = (0,0), (1,0), (0,1), (1,1), and pi
// interleaves the rows so that every 16-lane group of a ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) meets
// sixteen different 4-bank groups: rows with r mod 4 in {0,1} for lanes 0-3 and 12-15, {2,3} for lanes 4-11; the two
// chunks a group mixes lie 4 bank groups apart (kH2Sub), which those two row sets are closed under.
// Operand registers: rows halves A0, A1 and column halves B0, B1, 16 registers each (64, as before).  The four
// quadrant products of a macro tile run in a cyclic order -- (0,0) (0,1) (1,1) (1,0) on even tiles, (0,1) (0,0) (1,0)
// (1,1) on odd ones -- so that the last product of a tile, which is still to be issued when the tile's barrier comes,
// leaves free exactly the two operand sets the next tile's first product needs.
// ------------------------------------------------------------------------------
__device__ __forceinline__ int h2_pi16(int l) {
  return l < 4 ? (l & 1) + 4 * (l >> 1) : l < 12 ? 2 + ((l - 4) & 1) + 4 * ((l - 4) >> 1) : 8 + ((l - 12) & 1) + 4 * ((l - 12) >> 1);
}
__device__ __forceinline__ f32x4 mfma16_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_h16x8, a), __builtin_bit_cast(x3_h16x8, b), c, 0, 0, 0);
}
struct H2Consumer16 {
  f32x4 acc[4][4];
  u32x4 pa[2][2][2], pb[2][2][2];   // [half][16-row block][term]
  const uint32_t* a0;
  const uint32_t* b0;
  // The matrix instruction aligns its products to the largest addend, the accumulator, and drops what falls below its
  // window: over a reduction of 65536 (2048 macro tiles) the lo x hi terms are partly lost against the grown running
  // sum (dec_x3_64s.h has the measurement).  Every kH2FlushTiles tiles the 64 accumulators are added -- by vector
  // adds, rounded to nearest -- into a wave-private image in LDS and start from zero again.
  float* l2;
  bool has2;
  __device__ __forceinline__ void flush() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4* q = reinterpret_cast<f32x4*>(l2) + (i * 4 + j) * 64;
        *q = has2 ? *q + acc[i][j] : acc[i][j];
        acc[i][j] = (f32x4)(0.f);
      }
    has2 = true;
  }
  __device__ __forceinline__ void init(uint32_t* lds, int wm, int wn, int lane) {
    const int l15 = lane & 15, kb = lane >> 4;
    l2 = reinterpret_cast<float*>(lds + 2 * kH2Buf) + (wm * 2 + wn) * (64 * 64) + lane * 4;
    has2 = false;
    const int off = (kb & 1) * kH2Sub + 4 * (kb >> 1);
    a0 = lds + (wm * 64 + h2_pi16(l15)) * kH2Row + off;
    b0 = lds + 2 * kH2Sub + (wn * 64 + h2_pi16(l15)) * kH2Row + off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);
  }
  template <int RH> __device__ __forceinline__ void rdA(int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) pa[RH][i][p] = *reinterpret_cast<const u32x4*>(a0 + buf * kH2Buf + (32 * RH + 16 * i) * kH2Row + 8 * p);
  }
  template <int CH> __device__ __forceinline__ void rdB(int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) pb[CH][i][p] = *reinterpret_cast<const u32x4*>(b0 + buf * kH2Buf + (32 * CH + 16 * i) * kH2Row + 8 * p);
  }
  template <int RH, int CH> __device__ __forceinline__ void mm() {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};  // lo*hi, hi*lo, hi*hi
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[2 * RH + i][2 * CH + j] = mfma16_f16(pa[RH][i][PA[q]], pb[CH][j][PB[q]], acc[2 * RH + i][2 * CH + j]);
  }
  // all T macro tiles; every consumer passes barriers P_{-1} .. P_{Tr-1}
  __device__ __forceinline__ void even_tile(int buf) {   // (0,0) (0,1) (1,1); (1,0) stays to be issued
    rdA<0>(buf); rdB<0>(buf);
    __builtin_amdgcn_sched_barrier(0);
    mm<1, 1>();                       // the previous (odd) tile's last product (zero operands before the first tile)
    __builtin_amdgcn_sched_barrier(0);
    rdB<1>(buf);
    __builtin_amdgcn_sched_barrier(0);
    mm<0, 0>();
    __builtin_amdgcn_sched_barrier(0);
    rdA<1>(buf);
    __builtin_amdgcn_sched_barrier(0);
    mm<0, 1>();
    __builtin_amdgcn_sched_barrier(0);
    mm<1, 1>();
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void odd_tile(int buf) {    // (0,1) (0,0) (1,0); (1,1) stays to be issued
    rdA<0>(buf); rdB<1>(buf);
    __builtin_amdgcn_sched_barrier(0);
    mm<1, 0>();                       // the previous (even) tile's last product
    __builtin_amdgcn_sched_barrier(0);
    rdB<0>(buf);
    __builtin_amdgcn_sched_barrier(0);
    mm<0, 1>();
    __builtin_amdgcn_sched_barrier(0);
    rdA<1>(buf);
    __builtin_amdgcn_sched_barrier(0);
    mm<0, 0>();
    __builtin_amdgcn_sched_barrier(0);
    mm<1, 0>();
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void run(int T, int Tr) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) { pa[1][i][p] = (u32x4)(0u); pb[1][i][p] = (u32x4)(0u); }
    __syncthreads();  // P_{-1}
    int t = 0;
    for (; t + 1 < T; t += 2) {
      even_tile(0);
      __syncthreads();  // P_t
      odd_tile(1);
      __syncthreads();  // P_{t+1}
      if (((t + 2) & (kH2FlushTiles - 1)) == 0 && t + 2 < T) flush();   // (the tile's last product, still to be issued, joins the next sum)
    }
    if (t < T) {
      even_tile(0);
      __syncthreads();
      mm<1, 0>();
    } else if (T > 0) {
      mm<1, 1>();
    }
    for (int u = T; u < Tr; ++u) __syncthreads();
    if (has2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += reinterpret_cast<const f32x4*>(l2)[(i * 4 + j) * 64];
    }
  }
  // output (16-row block I, 16-column block Jb, register i) of lane: tile row / tile column inside the wave's 64 x 64
  __device__ __forceinline__ static int out_row(int I, int i, int lane) { return 16 * I + h2_pi16(4 * (lane >> 4) + i); }
  __device__ __forceinline__ static int out_col(int Jb, int lane) { return 16 * Jb + h2_pi16(lane & 15); }
};

__global__ __launch_bounds__(768, 1) void enc_h2_bwd_gemm(const float* __restrict__ X, int64_t B, int D, int J,
                                                          const uint32_t* __restrict__ Pimg, int BT,
                                                          float* __restrict__ out0, float* __restrict__ out1,
                                                          const uint32_t* __restrict__ rng, uint32_t* __restrict__ rng_seen,
                                                          const uint32_t* __restrict__ gate,
                                                          const uint32_t* __restrict__ colmax) {
  if (gate != nullptr && *gate == 0) return;  // second launch: only if the guard rejected the first one's scale of X
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem_raw);  // [2][kH2Buf]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t m0 = (int64_t)blockIdx.x * 128;  // first column d
  const int j0 = blockIdx.y * 128;
  const int T = (int)(B / 32);
  const int Tr = (T + 3) & ~3;
  float sx, ix;
  h2_scale_of(rng[0], &sx, &ix);

  if (wave >= 8) {
    // ---------------- producers of the dpre^T image ----------------
    const int tid = threadIdx.x - 512;
    const uint32_t* wptr[4];
    int woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i, row = f >> 3, ch = f & 7;
      const int s = ch >> 2, p = (ch >> 1) & 1, h = ch & 1;
      wptr[i] = Pimg + (int64_t)(j0 + row) * BT * 32 + 4 * ch;
      woff[i] = 2 * kH2Sub + s * kH2Sub + row * kH2Row + 8 * p + 4 * h;
    }
    u32x4 rw[kX3StagesW][4];
    auto wload = [&](int t, u32x4 (&r)[4]) {
      const int64_t kt = (t < T ? t : T - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const u32x4*>(wptr[i] + kt * 32);
    };
    auto wstore = [&](int buf, const u32x4 (&w)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(lds + buf * kH2Buf + woff[i]) = w[i];
    };
    wload(0, rw[0]); wload(1, rw[1]); wload(2, rw[2]); wload(3, rw[3]);
    wstore(0, rw[0]);
    wload(4, rw[0]);
    __syncthreads();  // P_{-1}
    auto step = [&](int t, u32x4 (&w)[4]) {
      wstore((t + 1) & 1, w);
      wload(t + 5, w);
      __syncthreads();  // P_t
    };
    for (int t = 0; t < Tr; t += 4) {
      step(t, rw[1]); step(t + 1, rw[2]); step(t + 2, rw[3]); step(t + 3, rw[0]);
    }
  } else if (wave >= 4) {
    // ---------------- X producers: columns d = m0 + 4c..4c+3, rows 4 rg..4 rg+3 of the macro tile ----------------
    const int tid = threadIdx.x - 256;
    const int c = tid & 31, rg = tid >> 5;
    const float* xptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xptr[i] = X + (int64_t)(4 * rg + i) * D + m0 + 4 * c;
    const int xoff = (rg >> 2) * kH2Sub + c * kH2Row + 4 * ((rg >> 1) & 1) + 2 * (rg & 1);
    const int64_t xstep = 32 * (int64_t)D;
    f32x4 rx[kX3StagesX][4];
    auto xload = [&](int t, f32x4 (&r)[4]) {
      const int64_t k0 = xstep * (t < T ? t : T - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xptr[i] + k0));
    };
    float xmax = 0.f;
    auto xstore = [&](int buf, const f32x4 (&r)[4]) {
      uint32_t* base = lds + buf * kH2Buf;
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // column i of four consecutive rows (the reduction index) -> LDS row 32 i + c
        xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(r[i][0]), fabsf(r[i][1]))), fmaxf(fabsf(r[i][2]), fabsf(r[i][3])));
        const Split2 q0 = split2_pair(r[0][i] * sx, r[1][i] * sx);
        const Split2 q1 = split2_pair(r[2][i] * sx, r[3][i] * sx);
        *reinterpret_cast<u32x2*>(base + xoff + 32 * i * kH2Row) = u32x2{q0.h, q1.h};
        *reinterpret_cast<u32x2*>(base + xoff + 32 * i * kH2Row + 8) = u32x2{q0.l, q1.l};
      }
    };
    xload(0, rx[0]); xload(1, rx[1]); xload(2, rx[2]); xload(3, rx[3]);
    xstore(0, rx[0]);
    xload(4, rx[0]);
    __syncthreads();  // P_{-1}
    auto step = [&](int t, f32x4 (&r)[4]) {
      xstore((t + 1) & 1, r);
      xload(t + 5, r);
      __syncthreads();  // P_t
    };
    for (int t = 0; t < Tr; t += 4) {
      step(t, rx[1]); step(t + 1, rx[2]); step(t + 2, rx[3]); step(t + 3, rx[0]);
    }
    if (rng_seen != nullptr) {   // max|x| of what this wave streamed (every tile is stored at least once)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o));
      if (lane == 0) atomic_max_if_larger(rng_seen, __float_as_uint(xmax));
    }
  } else {
    // ---------------- consumers (H2Consumer16) ----------------
    const int wm = wave >> 1, wn = wave & 1;
    H2Consumer16 cw;
    cw.init(lds, wm, wn, lane);
    cw.run(T, Tr);
    const int Kh = J / 2;
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int Jb = 0; Jb < 4; ++Jb) {
        const int col = j0 + wn * 64 + H2Consumer16::out_col(Jb, lane);
        float sp, ip;
        h2_scale_of(colmax[col], &sp, &ip);   // (the image and colmax have rows_pad >= j0 + 128 rows)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int mrow = wm * 64 + H2Consumer16::out_row(I, i, lane);
          const int64_t d = m0 + 4 * (mrow & 31) + (mrow >> 5);  // LDS row rho -> column of X
          if (col < J) (col < Kh ? out0 + (int64_t)col * D : out1 + (int64_t)(col - Kh) * D)[d] = (cw.acc[I][Jb][i] * ix) * ip;
        }
      }
  }
}

// ------------------------------------------------------------------------------
// Forward GEMM on two fp16 terms per operand.  The weights' scale comes from a reduction over them (rng[1]).  X is seen
// here for the first time in the step, so the kernel runs with the power of two of the PREVIOUS forward call's max|X|
// (rng[4], kept across calls) while its producers record this call's maximum (rng[0]); h2_fwd_guard accepts the result
// only if both give the same scale -- otherwise it raises rng[3] and a second launch of the same kernel, which returns
// at once while the flag is down, repeats the product with the right scale.  Either way the result is the one the
// scale of X's own maximum gives: a function of the inputs alone, bit-reproducible, whatever came before.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void h2_split_rows(const float* __restrict__ W0, const float* __restrict__ W1,
                                                     int rows0, int rows, int Kd, int rows_pad, int KT,
                                                     uint32_t* __restrict__ img, const uint32_t* __restrict__ rng) {
  const int64_t total = (int64_t)rows_pad * KT * 4;
  const int64_t half = (total + 1) >> 1;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i0 >= half) return;
  float sc, inv;
  h2_scale_of(rng[1], &sc, &inv);
  // two items per thread (i0 and i0 + half), both loads requested before either is split: 32 bytes per thread in
  // flight left the kernel at 2.5 TB/s
  float v[2][8];
  int64_t dsto[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int64_t i = i0 + u * half;
    const bool on = i < total;
    const int row = on ? (int)(i / (KT * 4)) : 0, c = on ? (int)(i % (KT * 4)) : 0;
    const int kt = c >> 2, s = (c >> 1) & 1, h = c & 1;
    const int k = 8 * c;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
    if (on && row < rows) {
      const float* src = (row < rows0 ? W0 + (int64_t)row * Kd : W1 + (int64_t)(row - rows0) * Kd);
      if (k + 8 <= Kd && ((reinterpret_cast<uintptr_t>(src + k) & 15) == 0)) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + k), b = *reinterpret_cast<const f32x4*>(src + k + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[u][e] = a[e]; v[u][4 + e] = b[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k + e < Kd) v[u][e] = src[k + e];
      }
    }
    dsto[u] = on ? (((int64_t)row * KT + kt) * 2 + s) * 16 + 4 * h : -1;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (dsto[u] < 0) continue;
    u32x4 Q[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const Split2 q = split2_pair(v[u][2 * e] * sc, v[u][2 * e + 1] * sc);
      Q[0][e] = q.h; Q[1][e] = q.l;
    }
    uint32_t* dst = img + dsto[u];
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(dst + 8 * p) = Q[p];
  }
}

__global__ void h2_fwd_guard(uint32_t* rng) {
  float sp, ip, sc, ic;
  h2_scale_of(rng[4], &sp, &ip);   // the scale the kernel used
  h2_scale_of(rng[0], &sc, &ic);   // the scale of the maximum it saw
  rng[3] = (sp == sc) ? 0u : 1u;   // accepted only if they are the same: the result is then a function of X and W alone
  rng[4] = rng[0];
}

__global__ __launch_bounds__(768, 1) void enc_h2_fwd_gemm(const float* __restrict__ X, int64_t B, int D, int J,
                                                          const uint32_t* __restrict__ Wimg, int KT,
                                                          float* __restrict__ out0, int klen, uint32_t* __restrict__ rng,
                                                          const uint32_t* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;  // second launch: only if the guard rejected the first one's scale
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem_raw);  // [2][kH2Buf]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t m0 = (int64_t)blockIdx.x * 128;
  const int j0 = blockIdx.y * 128;
  const int kbeg = blockIdx.z * klen;
  const int kend = (kbeg + klen < D) ? kbeg + klen : D;
  const int T = (kend - kbeg) / 32;
  const int Tr = (T + 3) & ~3;
  float* pre = out0 + (int64_t)blockIdx.z * B * J;
  float sx, ix, sw, iw;
  h2_scale_of(rng[4], &sx, &ix);
  h2_scale_of(rng[1], &sw, &iw);

  if (wave >= 8) {
    // ---------------- producers of the weight image ----------------
    const int tid = threadIdx.x - 512;
    const uint32_t* wptr[4];
    int woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i, row = f >> 3, ch = f & 7;
      const int s = ch >> 2, p = (ch >> 1) & 1, h = ch & 1;
      wptr[i] = Wimg + ((int64_t)(j0 + row) * KT + (kbeg >> 5)) * 32 + 4 * ch;
      woff[i] = 2 * kH2Sub + s * kH2Sub + row * kH2Row + 8 * p + 4 * h;
    }
    u32x4 rw[kX3StagesW][4];
    auto wload = [&](int t, u32x4 (&r)[4]) {
      const int64_t kt = (t < T ? t : T - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const u32x4*>(wptr[i] + kt * 32);
    };
    auto wstore = [&](int buf, const u32x4 (&w)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(lds + buf * kH2Buf + woff[i]) = w[i];
    };
    wload(0, rw[0]); wload(1, rw[1]); wload(2, rw[2]); wload(3, rw[3]);
    wstore(0, rw[0]);
    wload(4, rw[0]);
    __syncthreads();  // P_{-1}
    auto step = [&](int t, u32x4 (&w)[4]) {
      wstore((t + 1) & 1, w);
      wload(t + 5, w);
      __syncthreads();  // P_t
    };
    for (int t = 0; t < Tr; t += 4) {
      step(t, rw[1]); step(t + 1, rw[2]); step(t + 2, rw[3]); step(t + 3, rw[0]);
    }
  } else if (wave >= 4) {
    // ---------------- X producers ----------------
    const int tid = threadIdx.x - 256;
    const int xr0 = tid >> 3, xc = tid & 7;
    const int xr = (xr0 & ~3) | ((xr0 & 1) << 1) | ((xr0 >> 1) & 1);  // rows of a 16-lane group two apart (see enc_x3_gemm)
    const float* xptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = m0 + xr + 32 * i;
      xptr[i] = X + (row < B ? row : B - 1) * (int64_t)D + kbeg + 4 * xc;
    }
    const int xoff = (xc >> 2) * kH2Sub + xr * kH2Row + 4 * ((xc >> 1) & 1) + 2 * (xc & 1);
    f32x4 rx[kX3StagesX][4];
    auto xload = [&](int t, f32x4 (&r)[4]) {
      const int64_t k0 = 32 * (int64_t)(t < T ? t : T - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xptr[i] + k0));
    };
    float xmax = 0.f;
    auto xstore = [&](int buf, const f32x4 (&r)[4]) {
      uint32_t* base = lds + buf * kH2Buf;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(r[i][0]), fabsf(r[i][1]))), fmaxf(fabsf(r[i][2]), fabsf(r[i][3])));
        const Split2 q0 = split2_pair(r[i][0] * sx, r[i][1] * sx);
        const Split2 q1 = split2_pair(r[i][2] * sx, r[i][3] * sx);
        *reinterpret_cast<u32x2*>(base + xoff + 32 * i * kH2Row) = u32x2{q0.h, q1.h};
        *reinterpret_cast<u32x2*>(base + xoff + 32 * i * kH2Row + 8) = u32x2{q0.l, q1.l};
      }
    };
    xload(0, rx[0]); xload(1, rx[1]); xload(2, rx[2]); xload(3, rx[3]);
    xstore(0, rx[0]);
    xload(4, rx[0]);
    __syncthreads();  // P_{-1}
    auto step = [&](int t, f32x4 (&r)[4]) {
      xstore((t + 1) & 1, r);
      xload(t + 5, r);
      __syncthreads();  // P_t
    };
    for (int t = 0; t < Tr; t += 4) {
      step(t, rx[1]); step(t + 1, rx[2]); step(t + 2, rx[3]); step(t + 3, rx[0]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o));
    if (lane == 0) atomic_max_if_larger(rng, __float_as_uint(xmax));
  } else {
    // ---------------- consumers (H2Consumer16) ----------------
    const int wm = wave >> 1, wn = wave & 1;
    H2Consumer16 cw;
    cw.init(lds, wm, wn, lane);
    cw.run(T, Tr);
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int Jb = 0; Jb < 4; ++Jb) {
        const int col = j0 + wn * 64 + H2Consumer16::out_col(Jb, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t row = m0 + wm * 64 + H2Consumer16::out_row(I, i, lane);
          if (row < B && col < J) pre[row * J + col] = (cw.acc[I][Jb][i] * ix) * iw;
        }
      }
  }
}

// Split image of dpre^T: img[j][bt][s][piece][16 b] from dpre [B, J]; rows j >= J and b >= B are zero.
// One thread per (j, 8 consecutive b); consecutive threads take consecutive j (coalesced reads).
__global__ __launch_bounds__(256) void x3_split_cols(const float* __restrict__ P, int64_t B, int J, int rows_pad, int BT,
                                                     uint32_t* __restrict__ img) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)rows_pad * BT * 4;
  if (i >= total) return;
  const int j = (int)(i % rows_pad);
  const int64_t c = i / rows_pad;  // 8-chunk of b
  const int64_t bt = c >> 2;
  const int s = (int)(c >> 1) & 1, h = (int)c & 1;
  const int64_t b = 8 * c;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (j < J && b + e < B) ? P[(b + e) * J + j] : 0.f;
  u32x4 Q[3];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const Split3 q = split_pair(v[2 * e], v[2 * e + 1]);
    Q[0][e] = q.h; Q[1][e] = q.m; Q[2][e] = q.l;
  }
  uint32_t* dst = img + (((int64_t)j * BT + bt) * 2 + s) * 24 + 4 * h;
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + 8 * p) = Q[p];
}

// ---- host side ------------------------------------------------------------------
bool mfma_encoder_applies(const oivae_shape_t& s);  // mfma_encoder.hip

bool x3_encoder_applies(const oivae_shape_t& s) {
  if (!mfma_encoder_applies(s)) return false;
  if (test_hooks().enc == 'f') return false;  // test hook: OIVAE_ENC=fp32 keeps the fp32 matrix instructions
  // 2K a multiple of 64: the kernels' column tile is 128 wide and the images are padded to it with zero rows, so
  // 2K = 64, 192, ... leave half a tile's matrix work unused -- under kernels that wait for HBM either way
  return (2 * s.K) % 64 == 0 && s.D % 32 == 0;
}

static inline int x3_rows_pad(int J) { return (J + 127) / 128 * 128; }
static inline int x3_ktiles(int Kd) { return (Kd + 31) / 32; }

bool x3_encoder_bwd_applies(const oivae_shape_t& s, int64_t B) {
  return x3_encoder_applies(s) && s.D % 128 == 0 && B % 32 == 0;
}

int64_t x3_encoder_ws_bytes(const oivae_shape_t& s, int64_t B) {
  if (!x3_encoder_applies(s)) return 0;
  const int64_t kt = x3_ktiles(s.D), bt = x3_encoder_bwd_applies(s, B) ? (B + 31) / 32 : 0;
  return (int64_t)x3_rows_pad(2 * s.K) * (kt > bt ? kt : bt) * 48 * 4;  // 3 x bf16 per element of the larger image
}

// the per-column maxima of dpre (rows_pad words) sit behind the fp16 image of dpre^T (rows_pad * BT * 32 words) inside
// the image's allocation, which is sized for the three-term bf16 image (48 words per row and 32-row tile)
uint32_t* h2_bwd_colmax(uint32_t* img, int J, int64_t B) {
  return img + (int64_t)x3_rows_pad(J) * ((B + 31) / 32) * 32;
}
int64_t x3_encoder_bwd_img_bytes(const oivae_shape_t& s, int64_t B) {
  if (!x3_encoder_bwd_applies(s, B)) return 0;
  return (int64_t)x3_rows_pad(2 * s.K) * ((B + 31) / 32) * 48 * 4;
}

int x3_encoder_fwd_splits(const oivae_shape_t& s, int64_t B) {
  const int64_t tiles = ((B + 127) / 128) * (x3_rows_pad(2 * s.K) / 128);
  int splits = 1;
  const int ktiles = x3_ktiles(s.D);
  while (splits < kMaxEncSplits && tiles * splits < 512 && ktiles / (splits * 2) >= 16) splits *= 2;
  return splits;
}

// Ranges of the fp16 kernels: 64 words of the workspace (Workspace::ranges), all maxima as fp32 bit patterns:
//   [0] max|X| of the last pass over X (forward encoder, decoder or backward encoder call, whichever ran last)
//   [1] max|W| of the encoder weights   [2] max|dpre|   [3] forward redo flag   [4] max|X| the forward kernel assumed
//   [5] max|z|   [6] max|x| the decoder read, [7] its redo flag   [8] max|x| the backward kernel read, [9] its redo flag
// A kernel that scales X by [0] records what it reads itself; range_guard (mfma_decoder.hip) compares the two powers
// of two and a gated second launch repeats the kernel when they differ.  No host-side memory of earlier calls: every
// entry point is a function of its arguments and computes with the scale of its own X's maximum.
// test hook: OIVAE_ENC=bf16 keeps the three-term bf16 kernels
// few workgroups (they all finish together and each ends in an atomic on one word: 2048 of them took 27 us where 256
// take 10), many loads in flight per lane instead
static inline unsigned absmax_blocks(int64_t n, int arrays) { const int64_t b = (n + 8191) / 8192, cap = 256 / arrays; return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); }
static inline bool h2_enabled() { return test_hooks().enc != 'b'; }
void launch_range_guard(uint32_t* rng, int seen, int flag, hipStream_t st);  // mfma_decoder.hip

void launch_absmax_f32(const float* v, int64_t n, uint32_t* out, hipStream_t st) {
  hipLaunchKernelGGL(absmax_f32, dim3(absmax_blocks(n, 1), 1), dim3(256), 0, st, v, (const float*)nullptr, n, out);
}

int launch_encoder_fwd_x3(const oivae_shape_t& s, const oivae_tensors_t* params, const float* X, int64_t B,
                          float* pre, uint32_t* wimg, uint32_t* rng, int splits, hipStream_t st, bool image_ready) {
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(wimg) & 15))
    return fail(OIVAE_E_ARG, "x3 encoder: X and the workspace must be 16-byte aligned");
  const int J = 2 * s.K, Jp = x3_rows_pad(J), KT = x3_ktiles(s.D);
  const int64_t chunks = (int64_t)Jp * KT * 4;
  const size_t lds = sizeof(uint32_t) * 2 * kX3Buf;
  OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_x3_gemm<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int klen = ((KT + splits - 1) / splits) * 32;
  const dim3 grid((unsigned)((B + 127) / 128), (unsigned)(Jp / 128), (unsigned)splits);
  if (image_ready && !h2_enabled()) return fail(OIVAE_E_ARG, "x3 encoder: a prepared image needs the fp16 kernels");
  // [0] max|X| of this call, [1] max|W|, [2] unused here, [3] redo flag (a prepared image comes with its rng[1])
  OIVAE_HIP_CHECK(hipMemsetAsync(rng, 0, image_ready ? 4 : 16, st));
  if (h2_enabled()) {
    if (!image_ready) {
      hipLaunchKernelGGL(absmax_f32, dim3(absmax_blocks((int64_t)s.K * s.D, 2), 2), dim3(256), 0, st, params->enc_Wmu,
                         params->enc_Wsig, (int64_t)s.K * s.D, rng + 1);
      OIVAE_LAUNCH_CHECK();
      hipLaunchKernelGGL(h2_split_rows, dim3((unsigned)(((chunks + 1) / 2 + 255) / 256)), dim3(256), 0, st, params->enc_Wmu,
                         params->enc_Wsig, s.K, J, s.D, Jp, KT, wimg, rng);
      OIVAE_LAUNCH_CHECK();
    }
    const size_t lds2 = kH2LdsBytes;
    OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_h2_fwd_gemm),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    hipLaunchKernelGGL(enc_h2_fwd_gemm, grid, dim3(768), lds2, st, X, B, s.D, J, wimg, KT, pre, klen, rng,
                       (const uint32_t*)nullptr);
    OIVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(h2_fwd_guard, dim3(1), dim3(1), 0, st, rng);
    OIVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(enc_h2_fwd_gemm, grid, dim3(768), lds2, st, X, B, s.D, J, wimg, KT, pre, klen, rng,
                       (const uint32_t*)(rng + 3));
    OIVAE_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(x3_split_rows, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, params->enc_Wmu,
                       params->enc_Wsig, s.K, J, s.D, Jp, KT, wimg, (const uint32_t*)nullptr);
    OIVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(enc_x3_gemm<false>, grid, dim3(768), lds, st, X, B, s.D, J, wimg, KT, pre, (float*)nullptr, klen,
                       rng, (const uint32_t*)nullptr);
    OIVAE_LAUNCH_CHECK();
  }
  return OIVAE_OK;
}

int launch_encoder_bwd_x3(const oivae_shape_t& s, const oivae_tensors_t* grads, const float* X, const float* dpre,
                          int64_t B, uint32_t* img, uint32_t* rng, hipStream_t st, bool colmax_ready) {
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(img) & 15))
    return fail(OIVAE_E_ARG, "x3 encoder: X and the workspace must be 16-byte aligned");
  const int J = 2 * s.K, Jp = x3_rows_pad(J);
  const int BT = (int)((B + 31) / 32);
  const int64_t chunks = (int64_t)Jp * BT * 4;
  if (h2_enabled()) {
    // rng[8], rng[9] were zeroed by the caller, and the Jp column maxima of dpre behind the image either hold their
    // values already (colmax_ready: generic.hip encoder_bwd_run finds them where dpre is made, enc_bwd_pre +
    // enc_bwd_bias_final) or were zeroed for the pass over dpre below (flat_decoder.hip: memsets)
    uint32_t* colmax = h2_bwd_colmax(img, J, B);
    if (!colmax_ready) {
      int64_t nb = (B + 15) / 16;   // a chain of dependent loads per lane: short chains, many workgroups
      if (nb > 512) nb = 512;
      hipLaunchKernelGGL(absmax_cols_f32, dim3((unsigned)nb), dim3(256), 0, st, dpre, B, J, colmax);
    }
    OIVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(h2_split_cols, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, dpre, B, J, Jp, BT, img,
                       (const uint32_t*)colmax);
    OIVAE_LAUNCH_CHECK();
    const size_t lds = kH2LdsBytes;
    OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_h2_bwd_gemm),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)(s.D / 128), (unsigned)(Jp / 128), 1);
    // X is scaled by the max|X| the ranges hold (the forward call of this step recorded it); the kernel records what
    // it streams itself and the guard repeats it with that maximum if the two powers of two differ
    hipLaunchKernelGGL(enc_h2_bwd_gemm, grid, dim3(768), lds, st, X, B, s.D, J, img, BT, grads->enc_Wmu, grads->enc_Wsig,
                       (const uint32_t*)rng, rng + 8, (const uint32_t*)nullptr, (const uint32_t*)colmax);
    OIVAE_LAUNCH_CHECK();
    launch_range_guard(rng, 8, 9, st);
    OIVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(enc_h2_bwd_gemm, grid, dim3(768), lds, st, X, B, s.D, J, img, BT, grads->enc_Wmu, grads->enc_Wsig,
                       (const uint32_t*)rng, (uint32_t*)nullptr, (const uint32_t*)(rng + 9), (const uint32_t*)colmax);
    OIVAE_LAUNCH_CHECK();
    return OIVAE_OK;
  }
  hipLaunchKernelGGL(x3_split_cols, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, dpre, B, J, Jp, BT, img);
  OIVAE_LAUNCH_CHECK();
  const size_t lds = sizeof(uint32_t) * 2 * kX3Buf;
  OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_x3_gemm<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((unsigned)(s.D / 128), (unsigned)(Jp / 128), 1);
  hipLaunchKernelGGL(enc_x3_gemm<true>, grid, dim3(768), lds, st, X, B, s.D, J, img, BT, grads->enc_Wmu,
                     grads->enc_Wsig, 0, (uint32_t*)nullptr, (const uint32_t*)nullptr);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

}  // namespace oivae
