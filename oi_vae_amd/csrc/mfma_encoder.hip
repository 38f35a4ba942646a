// Encoder GEMMs on the fp32 matrix cores of gfx950.
//
// Forward: pre[B, 2K] = X[B, D] * [Wmu; Wsig]^T  -- both heads of the encoder
// (/root/reference/lib/models.py:15-16 with the two nn.Linear of
// bars_data_oivae.py:66-87) in ONE pass over X, instead of the reference's two
// separate Linear calls that each re-read X.
// Backward: [dWmu; dWsig][2K, D] = dpre^T[2K, B] * X[B, D]  -- the encoder weight
// gradient of loss.backward() (lib/oivae.py:75); no dX is needed.
//
// Both are LDS-tiled v_mfma_f32_32x32x2_f32 kernels (exact fp32: the result is an
// fmaf chain), 4 waves per workgroup, register-staged double buffering with one
// barrier per 32-deep K-tile.  The MFMA reduction index may be visited in any
// order, which the forward kernel uses to read both operands with conflict-free
// 16-byte LDS loads: lane half hh takes the odd/even 4-float chunk of each row.
// Roofline: MFMA-bound at the scale-up shape (1.1 TFLOP each over 17.2 GB of X).
#include "common.h"

#include <stdlib.h>

namespace oivae {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

#define OIVAE_ROW(r, hh) (((r) & 3) + 8 * ((r) >> 2) + 4 * (hh))
using ldsv_t = const volatile float __attribute__((address_space(3)))*;

// ------------------------------------------------------------------------------
// forward: pre[b][j] = sum_d X[b][d] * Wcat[j][d]
//   workgroup tile = (32*WM*WGM) rows x (32*WN*WGN) columns, WGM*WGN = 4 waves
// ------------------------------------------------------------------------------
template <int WM, int WN, int WGM, int WGN>
__global__ __launch_bounds__(256) void enc_fwd_gemm(const float* __restrict__ X, int64_t B, int D, int K,
                                                    const float* __restrict__ Wmu,
                                                    const float* __restrict__ Wsig,
                                                    float* __restrict__ pre, int klen) {
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN, LD = 36;
  constexpr int A4 = BM * 8 / 256, B4 = BN * 8 / 256;  // float4 loads per thread per tile
  static_assert(BM * 8 % 256 == 0 && BN * 8 % 256 == 0, "tile/threads");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sA = reinterpret_cast<float*>(smem_raw);  // [2][BM][LD]
  float* sB = sA + 2 * BM * LD;                    // [2][BN][LD]
  const int J = 2 * K;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int j0 = blockIdx.y * BN;

  const float* aptr[A4];
  bool aok[A4];
  int aoff[A4];
#pragma unroll
  for (int i = 0; i < A4; ++i) {
    const int f = tid + 256 * i, row = f >> 3, c = f & 7;
    aok[i] = (m0 + row) < B;
    aptr[i] = X + (aok[i] ? (m0 + row) : 0) * (int64_t)D + 4 * c;
    aoff[i] = row * LD + 4 * c;
  }
  const float* bptr[B4];
  bool bok[B4];
  int boff[B4];
#pragma unroll
  for (int i = 0; i < B4; ++i) {
    const int f = tid + 256 * i, row = f >> 3, c = f & 7;
    const int j = j0 + row;
    bok[i] = j < J;
    bptr[i] = (j < K ? Wmu + (int64_t)j * D : Wsig + (int64_t)(bok[i] ? j - K : 0) * D) + 4 * c;
    boff[i] = row * LD + 4 * c;
  }
  f32x4 ra[A4], rb[B4];
  // split-K: blockIdx.z owns columns [kbeg, kend) of X and writes its own partial image of pre
  const int kbeg = blockIdx.z * klen;
  const int kend = (kbeg + klen < D) ? kbeg + klen : D;
  const int c4 = 4 * (tid & 7);  // this thread's column inside a 32-wide K tile
  auto gload = [&](int k0) {     // columns at or beyond kend read as zero (D need only be a multiple of 4)
    const bool cok = (k0 + c4) < kend;
#pragma unroll
    for (int i = 0; i < A4; ++i) ra[i] = (aok[i] && cok) ? *reinterpret_cast<const f32x4*>(aptr[i] + k0) : (f32x4)(0.f);
#pragma unroll
    for (int i = 0; i < B4; ++i) rb[i] = (bok[i] && cok) ? *reinterpret_cast<const f32x4*>(bptr[i] + k0) : (f32x4)(0.f);
  };
  auto lstore = [&](int buf) {
    float* a = sA + buf * BM * LD;
    float* b = sB + buf * BN * LD;
#pragma unroll
    for (int i = 0; i < A4; ++i) *reinterpret_cast<f32x4*>(a + aoff[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < B4; ++i) *reinterpret_cast<f32x4*>(b + boff[i]) = rb[i];
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x16)(0.f);

  const int T = (kend - kbeg + 31) / 32;
  pre += (int64_t)blockIdx.z * B * J;
  gload(kbeg);
  lstore(0);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) gload(kbeg + (t + 1) * 32);
    const float* a = sA + (t & 1) * BM * LD + (wm * WM * 32 + l31) * LD + 4 * hh;
    const float* b = sB + (t & 1) * BN * LD + (wn * WN * 32 + l31) * LD + 4 * hh;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LD + 8 * c);
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LD + 8 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < T) lstore((t + 1) & 1);
    __syncthreads();
  }
  // C[i = row b][j]: lane holds column j, registers hold rows
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = j0 + (wn * WN + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wm * WM + i) * 32 + OIVAE_ROW(r, hh);
        if (row < B && col < J) pre[row * J + col] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------
// backward: dW[j][d] = sum_b dpre[b][j] * X[b][d]
//   workgroup tile = (32*TJ) rows j x 256 columns d; each wave owns all rows x 64 columns
// ------------------------------------------------------------------------------
template <int TJ, int ND>
__global__ __launch_bounds__(256, (ND == 1 ? 2 : 1)) void enc_bwd_gemm(const float* __restrict__ X, const float* __restrict__ dpre,
                                                    int64_t B, int D, int K, float* __restrict__ dWmu,
                                                    float* __restrict__ dWsig, int64_t tiles_per_split,
                                                    int64_t split_stride) {
  // gridDim.z > 1: the batch rows are cut into gridDim.z contiguous slices of tiles_per_split 32-row tiles and slice z
  // writes its partial image at +z*split_stride floats (summed in slice order by the caller): shapes whose [2K, D]
  // output has too few tiles to fill the chip (MEG-like: 2K = 20, D = 6600 -> 26 workgroups)
  // workgroup tile = (32*TJ) rows j x (128*ND) columns d; each wave owns all rows x 32*ND columns.
  // ND == 1 halves the LDS footprint so that two workgroups share a CU: one computes while the other
  // sits at its barrier or waits for its tile loads.
  constexpr int BJ = 32 * TJ, BD = 128 * ND;
  constexpr int X4 = 32 * BD / 4 / 256;  // float4 per thread for the X tile
  constexpr int P4 = 32 * BJ / 4 / 256;  // float4 per thread for the dpre tile (BJ=128: 4, 64: 2, 32: 1)
  static_assert(32 * BJ / 4 % 256 == 0, "tile/threads");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sX = reinterpret_cast<float*>(smem_raw);  // [2][32][BD]
  float* sP = sX + 2 * 32 * BD;                    // [2][32][BJ]
  const int J = 2 * K;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int d0 = blockIdx.x * BD, j0 = blockIdx.y * BJ;

  f32x4 rx[X4], rp[P4];
  auto gload = [&](int64_t b0) {
#pragma unroll
    for (int i = 0; i < X4; ++i) {
      const int f = tid + 256 * i, row = f / (BD / 4), c = f % (BD / 4);
      const int64_t b = b0 + row;
      const int d = d0 + 4 * c;
      rx[i] = (b < B && d < D) ? *reinterpret_cast<const f32x4*>(X + b * (int64_t)D + d) : (f32x4)(0.f);
    }
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      const int f = tid + 256 * i, row = f / (BJ / 4), c = f % (BJ / 4);
      const int64_t b = b0 + row;
      const int j = j0 + 4 * c;
      rp[i] = (b < B && j < J) ? *reinterpret_cast<const f32x4*>(dpre + b * (int64_t)J + j) : (f32x4)(0.f);
    }
  };
  auto lstore = [&](int buf) {
    float* x = sX + buf * 32 * BD;
    float* p = sP + buf * 32 * BJ;
#pragma unroll
    for (int i = 0; i < X4; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<f32x4*>(x + 4 * f) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<f32x4*>(p + 4 * f) = rp[i];
    }
  };

  f32x16 acc[TJ][ND];
#pragma unroll
  for (int i = 0; i < TJ; ++i)
#pragma unroll
    for (int q = 0; q < ND; ++q) acc[i][q] = (f32x16)(0.f);

  const int64_t Tall = (B + 31) / 32;
  const int64_t tbeg = (int64_t)blockIdx.z * tiles_per_split;
  const int64_t T = (tbeg + tiles_per_split < Tall) ? tbeg + tiles_per_split : Tall;
  dWmu += (int64_t)blockIdx.z * split_stride;
  dWsig += (int64_t)blockIdx.z * split_stride;
  gload(tbeg * 32);
  lstore(tbeg & 1);
  __syncthreads();
  for (int64_t t = tbeg; t < T; ++t) {
    if (t + 1 < T) gload((t + 1) * 32);
    // Operands of k-step s+1 are read from LDS before the MFMAs of k-step s (two register sets), so the matrix
    // pipe never waits for an LDS round trip inside a tile.  Volatile LDS pointers keep the reads in this order
    // and out of ds_read2_b32 (whose 8-bit offsets would cost a VALU add per pair; a VALU instruction costs ~8
    // cycles of fp32 matrix time, an LDS read ~1-2: DESIGN.md 3.1).
    const ldsv_t p = (ldsv_t)(sP + (t & 1) * 32 * BJ + hh * BJ + l31);
    const ldsv_t x = (ldsv_t)(sX + (t & 1) * 32 * BD + hh * BD + wave * (32 * ND) + l31);
    float fa[2][TJ], fb[2][ND];
#pragma unroll
    for (int i = 0; i < TJ; ++i) fa[0][i] = p[32 * i];
#pragma unroll
    for (int q = 0; q < ND; ++q) fb[0][q] = x[32 * q];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) {
#pragma unroll
        for (int i = 0; i < TJ; ++i) fa[(s + 1) & 1][i] = p[2 * (s + 1) * BJ + 32 * i];
#pragma unroll
        for (int q = 0; q < ND; ++q) fb[(s + 1) & 1][q] = x[2 * (s + 1) * BD + 32 * q];
      }
      __builtin_amdgcn_sched_barrier(0);  // the reads stay AHEAD of this step's MFMAs
#pragma unroll
      for (int i = 0; i < TJ; ++i)
#pragma unroll
        for (int q = 0; q < ND; ++q)
          acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][i], fb[s & 1][q], acc[i][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t + 1 < T) lstore((t + 1) & 1);
    __syncthreads();
  }
  // C[i = j][col = d]
#pragma unroll
  for (int i = 0; i < TJ; ++i)
#pragma unroll
    for (int q = 0; q < ND; ++q) {
      const int d = d0 + wave * (32 * ND) + 32 * q + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + 32 * i + OIVAE_ROW(r, hh);
        if (d < D && j < J) {
          if (j < K) dWmu[(int64_t)j * D + d] = acc[i][q][r];
          else dWsig[(int64_t)(j - K) * D + d] = acc[i][q][r];
        }
      }
    }
}

bool mfma_encoder_applies(const oivae_shape_t& s) {
  if (test_hooks().force_generic) return false;  // test hook
  // 16-byte alignment of the rows of X ([B, D]) and of dpre ([B, 2K]); partial tiles are masked
  return (s.D % 4 == 0) && s.D >= 32 && (s.K % 2 == 0);
}

// tile shape (rows x cols of pre per workgroup) and K-splits for a given problem: big batches use
// 256-row tiles (X read once, weights re-read B/256 times from L2); small batches (data-parallel
// shards) use 64-row tiles and split the reduction over D so that >= 256 workgroups exist.
struct FwdPlan { int cfg; int splits; };
static FwdPlan plan_fwd(int J, int D, int64_t B) {
  FwdPlan p;
  // 256-row tiles whenever split-K (at most kMaxEncSplits slices of D) can still put >= 256 workgroups on the chip:
  // they issue 128 MFMAs per wave per barrier and re-read the weights B/256 times instead of B/64
  const bool big = B >= 256 * (256 / kMaxEncSplits);
  int bm, bn;
  if (J % 128 == 0) {
    // >= 512 tiles of 128 rows: two 128 x 128 workgroups per CU without split-K; else 256-row tiles (+ split-K)
    const bool two_per_cu = B >= 128 * 512;
    p.cfg = two_per_cu ? 5 : (big ? 0 : 1);
    bm = two_per_cu ? 128 : (big ? 256 : 64);
    bn = 128;
  }
  else if (J % 64 == 0) { p.cfg = big ? 2 : 3; bm = big ? 256 : 64; bn = 64; }
  else { p.cfg = 4; bm = 128; bn = 32; }
  const int64_t tiles = ((B + bm - 1) / bm) * ((J + bn - 1) / bn);
  int splits = 1;
  const int ktiles = (D + 31) / 32;
  while (splits < kMaxEncSplits && tiles * splits < 256 && ktiles / (splits * 2) >= 16) splits *= 2;
  p.splits = splits;
  return p;
}

int mfma_encoder_fwd_splits(const oivae_shape_t& s, int64_t B) {
  if (!mfma_encoder_applies(s)) return 1;
  return plan_fwd(2 * s.K, s.D, B).splits;
}

template <int WM, int WN, int WGM, int WGN>
static int launch_fwd(const float* X, int64_t B, int D, int K, const float* Wmu, const float* Wsig, float* pre,
                      int splits, hipStream_t st) {
  constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN;
  const size_t lds = sizeof(float) * 2 * (BM + BN) * 36;
  OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_fwd_gemm<WM, WN, WGM, WGN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int ktiles = (D + 31) / 32;
  const int klen = ((ktiles + splits - 1) / splits) * 32;
  const dim3 grid((unsigned)((B + BM - 1) / BM), (unsigned)((2 * K + BN - 1) / BN), (unsigned)splits);
  hipLaunchKernelGGL((enc_fwd_gemm<WM, WN, WGM, WGN>), grid, dim3(256), lds, st, X, B, D, K, Wmu, Wsig, pre, klen);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

// C[M, J] = A[M, Kd] * W[J, Kd]^T on the fp32 matrix cores (J even; rows of A and W 16-byte aligned).  With
// `splits` > 1 (mfma_gemm_nt_splits) the reduction is cut into slices and C receives `splits` partial images
// [splits][M][J] for the caller to add in index order.
int mfma_gemm_nt_splits(int64_t M, int Kd, int J) { return plan_fwd(J, Kd, M).splits; }

int mfma_gemm_nt(const float* A, int64_t M, int Kd, const float* W, int J, float* C, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (Kd % 4) || (J % 2))
    return fail(OIVAE_E_ARG, "mfma_gemm_nt: operands must be 16-byte aligned, Kd a multiple of 4, J even");
  const FwdPlan p = plan_fwd(J, Kd, M);
  const int Kh = J / 2;
  const float *Wm = W, *Wg = W + (int64_t)Kh * Kd;
  switch (p.cfg) {
    // 128 x 128 tiles at two workgroups per CU beat 256 x 128 at one (8.43 vs 8.7 ms at the scale-up shape,
    // same box): one workgroup computes while the other sits at its barrier; the weights' extra L2 re-reads are free
    case 5: return launch_fwd<2, 2, 2, 2>(A, M, Kd, Kh, Wm, Wg, C, p.splits, st);
    case 0: return launch_fwd<2, 4, 4, 1>(A, M, Kd, Kh, Wm, Wg, C, p.splits, st);
    case 1: return launch_fwd<1, 2, 2, 2>(A, M, Kd, Kh, Wm, Wg, C, p.splits, st);
    case 2: return launch_fwd<2, 2, 4, 1>(A, M, Kd, Kh, Wm, Wg, C, p.splits, st);
    case 3: return launch_fwd<1, 1, 2, 2>(A, M, Kd, Kh, Wm, Wg, C, p.splits, st);
    default: return launch_fwd<1, 1, 4, 1>(A, M, Kd, Kh, Wm, Wg, C, p.splits, st);
  }
}

int launch_encoder_fwd_mfma(const oivae_shape_t& s, const oivae_tensors_t* params, const float* X, int64_t B,
                            float* pre, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(params->enc_Wmu) & 15) ||
      (reinterpret_cast<uintptr_t>(params->enc_Wsig) & 15))
    return fail(OIVAE_E_ARG, "MFMA encoder: X, Wmu, Wsig must be 16-byte aligned");
  const FwdPlan p = plan_fwd(2 * s.K, s.D, B);
  const float *Wm = params->enc_Wmu, *Wg = params->enc_Wsig;
  switch (p.cfg) {
    case 5: return launch_fwd<2, 2, 2, 2>(X, B, s.D, s.K, Wm, Wg, pre, p.splits, st);
    case 0: return launch_fwd<2, 4, 4, 1>(X, B, s.D, s.K, Wm, Wg, pre, p.splits, st);
    case 1: return launch_fwd<1, 2, 2, 2>(X, B, s.D, s.K, Wm, Wg, pre, p.splits, st);
    case 2: return launch_fwd<2, 2, 4, 1>(X, B, s.D, s.K, Wm, Wg, pre, p.splits, st);
    case 3: return launch_fwd<1, 1, 2, 2>(X, B, s.D, s.K, Wm, Wg, pre, p.splits, st);
    default: return launch_fwd<1, 1, 4, 1>(X, B, s.D, s.K, Wm, Wg, pre, p.splits, st);
  }
}

template <int TJ, int ND>
static int launch_bwd(const float* X, const float* dpre, int64_t B, int D, int K, float* dWmu, float* dWsig,
                      hipStream_t st, int splits = 1, float* part = nullptr) {
  constexpr int BD = 128 * ND;
  const size_t lds = sizeof(float) * 2 * 32 * (BD + 32 * TJ);
  OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&enc_bwd_gemm<TJ, ND>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t T = (B + 31) / 32;
  if (splits > 1 && part == nullptr) splits = 1;
  const int64_t per = (T + splits - 1) / splits;
  const int nz = (int)((T + per - 1) / per);   // every slice holds at least one tile
  const dim3 grid((unsigned)((D + BD - 1) / BD), (unsigned)((2 * K + 32 * TJ - 1) / (32 * TJ)), (unsigned)nz);
  const int64_t JD = (int64_t)2 * K * D;
  if (nz > 1) {
    hipLaunchKernelGGL((enc_bwd_gemm<TJ, ND>), grid, dim3(256), lds, st, X, dpre, B, D, K, part, part + (int64_t)K * D, per, JD);
    OIVAE_LAUNCH_CHECK();
    launch_partial_sum2(part, JD, (int64_t)K * D, nz, dWmu, dWsig, st);   // both heads' images in one launch
  } else {
    hipLaunchKernelGGL((enc_bwd_gemm<TJ, ND>), grid, dim3(256), lds, st, X, dpre, B, D, K, dWmu, dWsig, T, (int64_t)0);
  }
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

// slices of the batch for the backward GEMM: only when its [2K, D] output has too few tiles for the chip
int mfma_encoder_bwd_splits(const oivae_shape_t& s, int64_t B) {
  if (!mfma_encoder_applies(s)) return 1;
  const int J = 2 * s.K;
  const int bj = (J % 128 == 0) ? 128 : ((J % 64 == 0) ? 64 : 32), bd = (J % 128 == 0) ? 128 : 256;
  const int64_t wgs = (int64_t)((s.D + bd - 1) / bd) * ((J + bj - 1) / bj);
  const int64_t T = (B + 31) / 32;
  int splits = 1;
  while (splits < 16 && wgs * splits < 256 && T / (splits * 2) >= 8) splits *= 2;
  return splits;
}

int launch_encoder_bwd_mfma(const oivae_shape_t& s, const oivae_tensors_t* grads, const float* X,
                            const float* dpre, int64_t B, float* part, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(dpre) & 15))
    return fail(OIVAE_E_ARG, "MFMA encoder: X and dpre must be 16-byte aligned");
  const int J = 2 * s.K;
  const int splits = mfma_encoder_bwd_splits(s, B);
  // 128 x 128 tiles, two workgroups per CU: 8.84 ms against 8.98 ms for 128 x 256 at the scale-up shape
  if (J % 128 == 0) return launch_bwd<4, 1>(X, dpre, B, s.D, s.K, grads->enc_Wmu, grads->enc_Wsig, st, splits, part);
  if (J % 64 == 0) return launch_bwd<2, 2>(X, dpre, B, s.D, s.K, grads->enc_Wmu, grads->enc_Wsig, st, splits, part);
  return launch_bwd<1, 2>(X, dpre, B, s.D, s.K, grads->enc_Wmu, grads->enc_Wsig, st, splits, part);
}

// C[J, Dn] = P[M, J]^T * X[M, Dn] on the fp32 matrix cores, rows of M added in order (deterministic).
// J a multiple of 4, Dn a multiple of 4, operands 16-byte aligned.
int mfma_gemm_tn(const float* P, const float* X, int64_t M, int J, int Dn, float* C, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(P) & 15) || (reinterpret_cast<uintptr_t>(X) & 15) || (J % 4) || (Dn % 4))
    return fail(OIVAE_E_ARG, "mfma_gemm_tn: operands must be 16-byte aligned, J and Dn multiples of 4");
  const int Kh = J / 2;
  float* C2 = C + (int64_t)Kh * Dn;
  if (J % 128 == 0) return launch_bwd<4, 1>(X, P, M, Dn, Kh, C, C2, st);
  if (J % 64 == 0) return launch_bwd<2, 2>(X, P, M, Dn, Kh, C, C2, st);
  return launch_bwd<1, 2>(X, P, M, Dn, Kh, C, C2, st);
}

}  // namespace oivae
