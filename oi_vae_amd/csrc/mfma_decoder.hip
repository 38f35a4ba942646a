// Grouped decoder forward + backward on the fp32 matrix cores of gfx950.
//
// Replaces, for two-layer generators with every dim a multiple of 32 (the
// BASELINE "synthetic scale-up" shape: K = H = M = D_g = 64, G = 1024), the
// reference's per-group Python loop (/root/reference/lib/models.py:91-95), the
// Gaussian log-likelihood (lib/distributions.py:40-45, lib/oivae.py:55-59) and
// the decoder half of loss.backward() (lib/oivae.py:75).
//
// Design (MI355X-first, not a port):
//   * one workgroup (4 waves, one per SIMD) owns one group g: its three weight
//     matrices (Ws[g], W1[g], W2[g]) sit in LDS once, with odd row strides so a
//     single copy serves both the forward (lanes along rows) and the backward
//     (lanes along columns) operand reads conflict-free;
//   * every wave walks its own 32-row tiles independently (no workgroup barrier
//     inside the row loop).  All products are computed transposed,
//     out^T[feature, row] = W[feature, k] * act^T[k, row], with
//     v_mfma_f32_32x32x2_f32: the accumulator of one layer (feature in the
//     registers, row on the lane) is then directly the B operand of the next
//     layer, so the forward chain and the three data-gradient products never
//     touch LDS for activations;
//   * the three weight-gradient products sum over rows, which live on lanes, so
//     their operands take one wave-private LDS round trip ([row][feature] image,
//     stride 68 floats: 16-byte stores and 4-byte operand reads conflict-free);
//     each wave keeps its 64x64 partial of dW2, dW1 and dWs in 192 accumulator
//     registers across the whole row loop (exact f32 MFMA = fmaf chain);
//   * mu and sigma [N, D] are never materialised; X is read once, as 16-byte
//     pieces, at the start of each tile so the epilogue finds it in registers;
//   * dz is summed over the G groups with float atomics into a transposed
//     [K, N] image (two 128-byte segments per wave instruction).  That sum is the
//     only order-dependent reduction; every weight gradient is reproducible.
// Roofline: MFMA-bound.  9 products x 2*32*64*64 flops per 32-row tile = 576
// v_mfma_f32_32x32x2_f32 per wave-tile.
#include "dec_shared.h"

namespace oivae {

template <int K, int H, int M, int DG>
__global__ __launch_bounds__(256) void dec_mfma(MfmaDecArgs a) {
  constexpr int KT = K / 32, HT = H / 32, MT = M / 32, DT = DG / 32;
  constexpr int LW_S = H + 1, LW_1 = H + 1, LW_2 = M + 1;  // odd weight strides
  constexpr int FMAX = (K > H ? (K > M ? (K > DG ? K : DG) : (M > DG ? M : DG)) : (H > M ? (H > DG ? H : DG) : (M > DG ? M : DG)));
  constexpr int LDB = FMAX + 4;                            // activation image stride
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sWs = reinterpret_cast<float*>(smem_raw);
  float* sW1 = sWs + K * LW_S;
  float* sW2 = sW1 + M * LW_1;
  float* sb1 = sW2 + DG * LW_2;
  float* sb2 = sb1 + M;
  float* sIsig = sb2 + DG;   // 1/sigma_d
  float* sC2 = sIsig + DG;   // inv/sigma_d
  float* sAct = sC2 + DG;    // 4 waves x 2 images x [32][LDB], 16-byte aligned (see host check)

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int g = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;

  // ---- stage the group's weights ----
  {
    const float* Wsg = a.Ws + (int64_t)g * K * H;
    for (int i = tid; i < K * H; i += 256) sWs[(i / H) * LW_S + (i % H)] = Wsg[i];
    const float* W1g = a.W1 + (int64_t)g * M * H;
    for (int i = tid; i < M * H; i += 256) sW1[(i / H) * LW_1 + (i % H)] = W1g[i];
    const float* W2g = a.W2 + (int64_t)g * DG * M;
    for (int i = tid; i < DG * M; i += 256) sW2[(i / M) * LW_2 + (i % M)] = W2g[i];
    for (int i = tid; i < M; i += 256) sb1[i] = a.b1[(int64_t)g * M + i];
    for (int i = tid; i < DG; i += 256) {
      const float sg = expf(a.ls[(int64_t)g * DG + i]) + a.sig_off;
      sb2[i] = a.b2[(int64_t)g * DG + i];
      sIsig[i] = 1.0f / sg;
      sC2[i] = a.inv / sg;
    }
  }
  __syncthreads();

  float* bufA = sAct + wave * (2 * 32 * LDB);
  float* bufB = bufA + 32 * LDB;

  f32x16 dW2acc[DT][MT], dW1acc[MT][HT], dWsacc[KT][HT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) dW2acc[i][j] = (f32x16)(0.f);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < HT; ++j) dW1acc[i][j] = (f32x16)(0.f);
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int j = 0; j < HT; ++j) dWsacc[i][j] = (f32x16)(0.f);
  float cs_gmu = 0.f, cs_gmu2 = 0.f, cs_da1 = 0.f;  // column sums; lane = feature

  const int64_t ntiles = (a.N + 31) / 32;
  for (int64_t tile = (int64_t)chunk * 4 + wave; tile < ntiles; tile += 4 * (int64_t)nchunks) {
    const int64_t n = tile * 32 + l31;
    const bool live = n < a.N;
    // ---- loads: this lane's half row of z, and its pieces of x ----
    f32x4 zreg[K / 8];
    {
      const f32x4* zp = reinterpret_cast<const f32x4*>(a.z + (live ? n : 0) * K + hh * (K / 2));
#pragma unroll
      for (int v = 0; v < K / 8; ++v) zreg[v] = live ? zp[v] : (f32x4)(0.f);
    }
    f32x4 xreg[DT][4];
    {
      const float* xp = a.X + ((live ? n : 0) % a.B) * (int64_t)a.D + (int64_t)g * DG + 4 * hh;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) xreg[dt][q] = *reinterpret_cast<const f32x4*>(xp + 32 * dt + 8 * q);
    }
    // ---- G1: h^T = Ws^T z^T ----
    f32x16 hacc[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) hacc[t] = (f32x16)(0.f);
#pragma unroll
    for (int s = 0; s < K / 2; ++s) {
      const float b = zreg[s >> 2][s & 3];
      const float* wp = sWs + (hh * (K / 2) + s) * LW_S + l31;
#pragma unroll
      for (int t = 0; t < HT; ++t) hacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[32 * t], b, hacc[t], 0, 0, 0);
    }
    // ---- G2: a1^T = W1 h^T + b1; t = tanh(a1) ----
    f32x16 tacc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(sb1 + 32 * t + 8 * q + 4 * hh);
        tacc[t][4 * q + 0] = bq[0]; tacc[t][4 * q + 1] = bq[1]; tacc[t][4 * q + 2] = bq[2]; tacc[t][4 * q + 3] = bq[3];
      }
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = hacc[ht][r];
        const float* wp = sW1 + l31 * LW_1 + 32 * ht + OIVAE_ROW(r, hh);
#pragma unroll
        for (int t = 0; t < MT; ++t) tacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[32 * t * LW_1], b, tacc[t], 0, 0, 0);
      }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) tacc[t][r] = fast_tanh(tacc[t][r]);
    // ---- G3: mu^T = W2 t^T + b2 ----
    f32x16 gacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(sb2 + 32 * t + 8 * q + 4 * hh);
        gacc[t][4 * q + 0] = bq[0]; gacc[t][4 * q + 1] = bq[1]; gacc[t][4 * q + 2] = bq[2]; gacc[t][4 * q + 3] = bq[3];
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = tacc[mt][r];
        const float* wp = sW2 + l31 * LW_2 + 32 * mt + OIVAE_ROW(r, hh);
#pragma unroll
        for (int t = 0; t < DT; ++t) gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[32 * t * LW_2], b, gacc[t], 0, 0, 0);
      }
    // ---- epilogue: residual -> dL/dmu (in place) ----
    const float livef = live ? 1.0f : 0.0f;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 is = *reinterpret_cast<const f32x4*>(sIsig + 32 * t + 8 * q + 4 * hh);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(sC2 + 32 * t + 8 * q + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float qv = (xreg[t][q][e] - gacc[t][4 * q + e]) * is[e];
          gacc[t][4 * q + e] = -(qv * c2[e]) * livef;
        }
      }
    // ---- images for dW2: gmu -> bufA, t -> bufB ----
    lds_fence();
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {gacc[t][4 * q], gacc[t][4 * q + 1], gacc[t][4 * q + 2], gacc[t][4 * q + 3]};
        *reinterpret_cast<f32x4*>(bufA + l31 * LDB + 32 * t + 8 * q + 4 * hh) = v;
      }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {tacc[t][4 * q], tacc[t][4 * q + 1], tacc[t][4 * q + 2], tacc[t][4 * q + 3]};
        *reinterpret_cast<f32x4*>(bufB + l31 * LDB + 32 * t + 8 * q + 4 * hh) = v;
      }
    lds_fence();
    if (lane < DG) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) { const float v = bufA[r * LDB + lane]; s1 += v; s2 = fmaf(v, v, s2); }
      cs_gmu += s1; cs_gmu2 += s2;
    }
    // ---- W1: dW2 += gmu^T t ----
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = 2 * s + hh;
      float av[DT], bv[MT];
#pragma unroll
      for (int t = 0; t < DT; ++t) av[t] = bufA[row * LDB + 32 * t + l31];
#pragma unroll
      for (int t = 0; t < MT; ++t) bv[t] = bufB[row * LDB + 32 * t + l31];
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) dW2acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], dW2acc[i][j], 0, 0, 0);
    }
    // ---- G5: dt^T = W2^T gmu^T; da1 = dt (1 - t^2) ----
    f32x16 dacc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) dacc[t] = (f32x16)(0.f);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = gacc[dt][r];
        const float* wp = sW2 + (32 * dt + OIVAE_ROW(r, hh)) * LW_2 + l31;
#pragma unroll
        for (int t = 0; t < MT; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[32 * t], b, dacc[t], 0, 0, 0);
      }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[t][r] = dacc[t][r] * (1.0f - tacc[t][r] * tacc[t][r]);
    // ---- images for dW1: da1 -> bufA, h -> bufB ----
    lds_fence();
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {dacc[t][4 * q], dacc[t][4 * q + 1], dacc[t][4 * q + 2], dacc[t][4 * q + 3]};
        *reinterpret_cast<f32x4*>(bufA + l31 * LDB + 32 * t + 8 * q + 4 * hh) = v;
      }
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {hacc[t][4 * q], hacc[t][4 * q + 1], hacc[t][4 * q + 2], hacc[t][4 * q + 3]};
        *reinterpret_cast<f32x4*>(bufB + l31 * LDB + 32 * t + 8 * q + 4 * hh) = v;
      }
    lds_fence();
    if (lane < M) {
      float s1 = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) s1 += bufA[r * LDB + lane];
      cs_da1 += s1;
    }
    // ---- W2: dW1 += da1^T h ----
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = 2 * s + hh;
      float av[MT], bv[HT];
#pragma unroll
      for (int t = 0; t < MT; ++t) av[t] = bufA[row * LDB + 32 * t + l31];
#pragma unroll
      for (int t = 0; t < HT; ++t) bv[t] = bufB[row * LDB + 32 * t + l31];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < HT; ++j) dW1acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], dW1acc[i][j], 0, 0, 0);
    }
    // ---- G6: dh^T = W1^T da1^T ----
    f32x16 eacc[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) eacc[t] = (f32x16)(0.f);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = dacc[mt][r];
        const float* wp = sW1 + (32 * mt + OIVAE_ROW(r, hh)) * LW_1 + l31;
#pragma unroll
        for (int t = 0; t < HT; ++t) eacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[32 * t], b, eacc[t], 0, 0, 0);
      }
    // ---- images for dWs: z -> bufA, dh -> bufB ----
    lds_fence();
#pragma unroll
    for (int v = 0; v < K / 8; ++v) *reinterpret_cast<f32x4*>(bufA + l31 * LDB + hh * (K / 2) + 4 * v) = zreg[v];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {eacc[t][4 * q], eacc[t][4 * q + 1], eacc[t][4 * q + 2], eacc[t][4 * q + 3]};
        *reinterpret_cast<f32x4*>(bufB + l31 * LDB + 32 * t + 8 * q + 4 * hh) = v;
      }
    lds_fence();
    // ---- W3: dWs += z^T dh ----
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = 2 * s + hh;
      float av[KT], bv[HT];
#pragma unroll
      for (int t = 0; t < KT; ++t) av[t] = bufA[row * LDB + 32 * t + l31];
#pragma unroll
      for (int t = 0; t < HT; ++t) bv[t] = bufB[row * LDB + 32 * t + l31];
#pragma unroll
      for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int j = 0; j < HT; ++j) dWsacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], dWsacc[i][j], 0, 0, 0);
    }
    // ---- G7: dz^T = Ws dh^T, summed over groups ----
    f32x16 zacc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) zacc[t] = (f32x16)(0.f);
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = eacc[ht][r];
        const float* wp = sWs + l31 * LW_S + 32 * ht + OIVAE_ROW(r, hh);
#pragma unroll
        for (int t = 0; t < KT; ++t) zacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[32 * t * LW_S], b, zacc[t], 0, 0, 0);
      }
    if (live) {
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          atomicAdd(a.dzT + (int64_t)(32 * t + OIVAE_ROW(r, hh)) * a.N + n, zacc[t][r]);
    }
  }

  // ---- reduce the four waves' partials in a fixed order and store ----
  const int64_t cs = a.chunk_stride * chunk;
  float* stage = sAct;  // 4 x 4096 floats needed; the activation images hold 4*2*32*LDB >= 4*4096 for FMAX=64
  auto reduce_store = [&](auto& acc, auto TI, auto TJ, float* dst, int ldd) {
    constexpr int ti_n = decltype(TI)::value, tj_n = decltype(TJ)::value;
    constexpr int rows = 32 * ti_n, cols = 32 * tj_n;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ti_n; ++i)
#pragma unroll
      for (int j = 0; j < tj_n; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[wave * (rows * cols) + (32 * i + OIVAE_ROW(r, hh)) * cols + 32 * j + l31] = acc[i][j][r];
    __syncthreads();
    for (int idx = tid; idx < rows * cols; idx += 256) {
      float s = stage[idx];
      s += stage[rows * cols + idx];
      s += stage[2 * rows * cols + idx];
      s += stage[3 * rows * cols + idx];
      dst[(idx / cols) * ldd + (idx % cols)] = s;
    }
  };
  reduce_store(dW2acc, std::integral_constant<int, DT>{}, std::integral_constant<int, MT>{},
               a.gW2 + cs + (int64_t)g * DG * M, M);
  reduce_store(dW1acc, std::integral_constant<int, MT>{}, std::integral_constant<int, HT>{},
               a.gW1 + cs + (int64_t)g * M * H, H);
  reduce_store(dWsacc, std::integral_constant<int, KT>{}, std::integral_constant<int, HT>{},
               a.gWs + cs + (int64_t)g * K * H, H);
  __syncthreads();
  stage[wave * 192 + lane] = cs_gmu;
  stage[wave * 192 + 64 + lane] = cs_gmu2;
  stage[wave * 192 + 128 + lane] = cs_da1;
  __syncthreads();
  float* sred = stage + 1024;
  if (tid < 64) {
    const float s_gmu = ((stage[tid] + stage[192 + tid]) + stage[384 + tid]) + stage[576 + tid];
    const float s_gmu2 = ((stage[64 + tid] + stage[256 + tid]) + stage[448 + tid]) + stage[640 + tid];
    const float s_da1 = ((stage[128 + tid] + stage[320 + tid]) + stage[512 + tid]) + stage[704 + tid];
    float llv = 0.f;
    if (tid < DG) {
      const int64_t d = (int64_t)g * DG + tid;
      const float e = expf(a.ls[d]);
      const float sg = e + a.sig_off;
      const float c2 = a.inv / sg;
      const float sumq2 = s_gmu2 / (c2 * c2);           // sum over this block's rows of ((x-mu)/sigma)^2
      const float nrows = (chunk == 0) ? (float)a.N : 0.f;  // the constant terms are added once per group
      a.gb2[cs + d] = s_gmu;
      a.gls[cs + d] = (nrows - sumq2) * (e * a.inv / sg);
      llv = nrows * (-0.5f * kLog2Pi - logf(sg)) - 0.5f * sumq2;
    }
    if (tid < M) a.gb1[cs + (int64_t)g * M + tid] = s_da1;
    sred[tid] = llv;
  }
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int i = 0; i < 64; ++i) tot += (double)sred[i];
    a.ll_part[(int64_t)chunk * gridDim.x + g] = tot;
  }
}



__global__ __launch_bounds__(256) void dec_absmax_f32(const float* __restrict__ v, int64_t n, uint32_t* __restrict__ out,
                                                      int per_block) {
  float m = 0.f;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  if ((reinterpret_cast<uintptr_t>(v) & 15) == 0) {   // 16 bytes per lane, scalar tail
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const f4 q = reinterpret_cast<const f4*>(v)[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(q[0]), fabsf(q[1]))), fmaxf(fabsf(q[2]), fabsf(q[3])));
    }
    for (int64_t i = 4 * n4 + tid; i < n; i += nth) m = fmaxf(m, fabsf(v[i]));
  } else {
    for (int64_t i = tid; i < n; i += nth) m = fmaxf(m, fabsf(v[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float part[4];   // one atomic per workgroup: thousands of atomics on one address take 14 ns each
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t mx = __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
    if (per_block) out[blockIdx.x] = mx;   // one slot per workgroup, every slot written: nothing to zero first
    else atomicMax(out, mx);
  }
}

// Split image of z for dec_x3_64, shared by all groups: per 32-row tile 24 fragments of 1 KB,
//   [0..11]  (ks, term): transposed orientation, lane (row l31, half hh), slot e <-> latent x3_slot_feature(ks, hh, e)
//   [12..23] (kb, q, term): plain orientation, lane (latent 32 kb + l31, half hh), slot e <-> row 16 q + 8 (e/4) + 4 hh + e%4
// rows past N repeat row N-1 (the decoder zeroes their gradients).
// lay16 (dec_x3_64s, 16x16x32 instructions; two fp16 terms only): lane (c = lane & 15, q = lane >> 4),
//   [0..3]  (kk, nb): row 16 nb + c, slot e <-> latent 32 kk + 8 q + e
//   [4..7]  kb: latent 16 kb + c, slot e <-> row 8 q + e
// dec_x3_64g / dec_x3_64s (rng != nullptr) fold two more jobs into this launch (each was a 5 us memset before): the
// maximum of z is reduced here from the kZmaxParts per-workgroup maxima dec_absmax_f32 left (every workgroup reduces
// them for itself; workgroup 0 also writes rng[5] = max|z| and lowers rng[6], rng[7] for the decoder kernel that
// follows), and dz -- N x 64 floats, one float4 per thread of this launch -- is zeroed.
constexpr int kZmaxParts = 256;
__global__ __launch_bounds__(256) void x3_split_z(const float* __restrict__ z, int64_t N, uint32_t* __restrict__ img,
                                                  uint32_t* __restrict__ rng, int lay16,
                                                  const uint32_t* __restrict__ zmax_parts, float* __restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tiles = (N + 31) / 32;
  uint32_t zmax = 0;
  if (rng != nullptr) {
    const int l = threadIdx.x & 63;
    zmax = max(max(zmax_parts[l], zmax_parts[64 + l]), max(zmax_parts[128 + l], zmax_parts[192 + l]));   // bit patterns of non-negative floats order like the floats
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zmax = max(zmax, (uint32_t)__shfl_xor((int)zmax, o));
    if (i == 0) { rng[5] = zmax; rng[6] = 0; rng[7] = 0; }
    if (i < N * 16) reinterpret_cast<x3_f32x4*>(dz)[i] = (x3_f32x4)(0.f);
  }
  if (i >= tiles * 512) return;
  const int64_t tile = i >> 9;
  const int grp = (int)(i >> 6) & 7, lane = (int)i & 63, l31 = lane & 31, hh = lane >> 5;
  float v[8];
  if (lay16) {
    const int lc = lane & 15, lq = lane >> 4;
    if (grp < 4) {
      int64_t n = tile * 32 + 16 * (grp & 1) + lc;
      n = n < N ? n : N - 1;
      const x3_f32x4 lo = *reinterpret_cast<const x3_f32x4*>(z + n * 64 + 32 * (grp >> 1) + 8 * lq);
      const x3_f32x4 hi = *reinterpret_cast<const x3_f32x4*>(z + n * 64 + 32 * (grp >> 1) + 8 * lq + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int64_t n = tile * 32 + 8 * lq + e;
        n = n < N ? n : N - 1;
        v[e] = z[n * 64 + 16 * (grp - 4) + lc];
      }
    }
  } else if (grp < 4) {
    int64_t n = tile * 32 + l31;
    n = n < N ? n : N - 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = z[n * 64 + x3_slot_feature(grp, hh, e)];
  } else {
    const int kb = (grp - 4) >> 1, q = (grp - 4) & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int64_t n = tile * 32 + 16 * q + 8 * (e >> 2) + 4 * hh + (e & 3);
      n = n < N ? n : N - 1;
      v[e] = z[n * 64 + 32 * kb + l31];
    }
  }
  if (rng != nullptr) {   // dec_x3_64g: two fp16 terms of z * 2^e in fragments 2 grp, 2 grp + 1 (both orientations)
    float sz, iz;
    h2_scale_of(zmax, &sz, &iz);
    const Frag2 f2 = split8_h2(v[0] * sz, v[1] * sz, v[2] * sz, v[3] * sz, v[4] * sz, v[5] * sz, v[6] * sz, v[7] * sz);
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<x3_u32x4*>(img + ((tile * 24 + grp * 2 + p) * 64 + lane) * 4) = f2.p[p];
    return;
  }
  const Frag3 f = split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<x3_u32x4*>(img + ((tile * 24 + grp * 3 + p) * 64 + lane) * 4) = f.p[p];
}


#include "dec_x3_64s.h"

// After a kernel that built its fp16 scales from rng[0] (the max|X| an earlier pass recorded) and recorded the
// max|x| it read itself in rng[seen]: the result stands only if both maxima have the same power of two, i.e. if the
// scales are the ones X's own maximum gives.  Otherwise flag rng[flag]: a gated second launch repeats the kernel with
// rng[0] corrected.  Either way the result is a function of the call's inputs alone.
__global__ void range_guard(uint32_t* rng, int seen, int flag) {
  float s0, i0, s1, i1;
  h2_scale_of(rng[0], &s0, &i0);
  h2_scale_of(rng[seen], &s1, &i1);
  rng[flag] = (s0 == s1) ? 0u : 1u;
  rng[0] = rng[seen];
}
void launch_range_guard(uint32_t* rng, int seen, int flag, hipStream_t st) {
  hipLaunchKernelGGL(range_guard, dim3(1), dim3(1), 0, st, rng, seen, flag);
}

__global__ __launch_bounds__(256) void gated_zero_f32(float* __restrict__ p, int64_t n, const uint32_t* __restrict__ gate) {
  if (*gate == 0) return;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    reinterpret_cast<f32x4*>(p)[i] = (f32x4)(0.f);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[4 * n4 + threadIdx.x] = 0.f;
}

bool mfma_decoder_applies(const oivae_shape_t& s) {
  if (test_hooks().force_generic) return false;  // test hook: cross-check the two kernel families
  if (s.gen_type != OIVAE_GEN_MLP) return false;
  // Two-layer generators no wider than 64 anywhere, one output width for all groups (a multiple of 4: X is read 16
  // bytes per lane): the all-64 shape has its own kernels (dec_x3_64s and its cross-checks), every other one runs
  // dec_x3_64s on zero-padded or pair-packed copies (dec_pad64.hip; dec_mfma<K, H, M, DG> on the fp32 matrix instruction
  // is the cross-check where every dimension is 32 or 64).
  const int dg = s.uniform_group_dim;
  return s.K >= 1 && s.K <= 64 && s.H >= 1 && s.H <= 64 && s.M >= 1 && s.M <= 64 && dg >= 4 && dg <= 64 && (dg & 3) == 0;
}
// ... and whether it pays: a padded launch costs what the all-64 shape costs whatever it holds (0.06 ms + 0.18 us per
// 1024 row-groups), the shape-generic kernel's time follows the shape's own work.  tools/narrow_dims.py: at 4.2 M
// row-groups the padded launch wins for every shape tried (0.75 ms against 1.05 ms at K = H = M = D_g = 8, 75 ms at
// 64/64/64/48); at 2048 row-groups from K H + H M + M D_g of about 1500 on.  Shapes of 32s and 64s always take it.
bool mfma_decoder_pays(const oivae_shape_t& s, int64_t N) {
  const int dg = s.uniform_group_dim;
  auto ok = [](int v) { return v == 32 || v == 64; };
  if (ok(s.K) && ok(s.H) && ok(s.M) && ok(dg)) return true;
  return s.K * s.H + s.H * s.M + s.M * dg >= kPadMinWork || N * (int64_t)s.G >= kPadMinRowGroups;
}
static inline bool dec_all64(const oivae_shape_t& s) {
  return s.K == 64 && s.H == 64 && s.M == 64 && s.uniform_group_dim == 64;
}

template <int K, int H, int M, int DG>
static int launch_one(const MfmaDecArgs& a, int G, int chunks, hipStream_t st) {
  constexpr int fm = (K > H ? (K > M ? (K > DG ? K : DG) : (M > DG ? M : DG)) : (H > M ? (H > DG ? H : DG) : (M > DG ? M : DG)));
  constexpr int LDB = fm + 4;
  size_t floats = (size_t)K * (H + 1) + (size_t)M * (H + 1) + (size_t)DG * (M + 1) + M + 3 * DG;
  floats = (floats + 3) / 4 * 4;  // keep the activation images 16-byte aligned
  size_t act = (size_t)4 * 2 * 32 * LDB;
  if (act < (size_t)4 * fm * fm) act = (size_t)4 * fm * fm;  // staging for the final 4-wave reduction
  const size_t lds = (floats + act) * sizeof(float);
  OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_mfma<K, H, M, DG>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((dec_mfma<K, H, M, DG>), dim3((unsigned)G, (unsigned)chunks), dim3(256), lds, st, a);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

// dec_x3_64s: bytes of second-level slabs (one of 128 x 64 floats per weight-gradient wave) -- 0 when a launch has no
// more rounds than one flush interval, or when another kernel runs
int64_t mfma_decoder_acc2_bytes(const oivae_shape_t& s, int64_t N, int chunks) {
  if (!dec_all64(s) && !dec_pad64_applies(s)) return 0;   // (the padded shapes run as all-64)
  const int64_t ntiles = (N + 31) / 32;
  const int64_t nrounds = (ntiles + 4 * (int64_t)chunks - 1) / (4 * (int64_t)chunks);
  if (nrounds <= kFlushRounds) return 0;
  return (int64_t)((s.G + 1) / 2) * chunks * 4 * 8192 * (int64_t)sizeof(float);
}

int mfma_decoder_chunks(const oivae_shape_t& s, int64_t N) {
  const int64_t wave_tiles = (N + 31) / 32;
  int64_t c = (1024 + s.G - 1) / s.G;               // aim for >= 1024 workgroups
  const int64_t cmax = (wave_tiles + 3) / 4;         // at least one tile per wave
  if (c > cmax) c = cmax;
  if (c < 1) c = 1;
  if ((int64_t)s.G * c > kMaxPartials) c = kMaxPartials / s.G > 0 ? kMaxPartials / s.G : 1;
  return (int)c;
}

int launch_decoder_mfma(const oivae_shape_t& s, const oivae_tensors_t* params, const oivae_tensors_t* grads,
                        const float* X, int64_t B, int32_t S, const float* z, float gen_sig_off, float* dz,
                        Workspace& w, hipStream_t st) {
  const int64_t N = B * (int64_t)S;
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(z) & 15) || (s.D % 4))
    return fail(OIVAE_E_ARG, "MFMA decoder: X and z must be 16-byte aligned");
  const char* dec_env = test_hooks().dec[0] ? test_hooks().dec : nullptr;
  // K, H, M in {32, 64} with 64 outputs per group: the all-64 fp16-term kernel on zero-padded copies (dec_pad64.hip),
  // unless a test hook asks for another family (OIVAE_DEC=v1: dec_mfma<K, H, M, DG> below, their cross-check)
  if (dec_pad64_applies(s) && w.dec_pad != nullptr &&
      (dec_env == nullptr || dec_env[0] == 's' || (dec_env[0] == 'g' && !dec_pad64_xhalf(s))))
    return launch_decoder_pad64(s, params, grads, X, B, S, z, gen_sig_off, dz, w, st);
  const DecLayout L = dec_layout(s);
  const int chunks = w.n_chunks;
  MfmaDecArgs a;
  a.W1 = params->gen_W1; a.b1 = params->gen_b1; a.W2 = params->gen_W2; a.b2 = params->gen_b2;
  a.ls = params->log_sigma; a.Ws = params->Ws;
  if (chunks > 1) {
    a.gW1 = w.slab + L.oW1; a.gb1 = w.slab + L.ob1; a.gW2 = w.slab + L.oW2; a.gb2 = w.slab + L.ob2;
    a.gls = w.slab + L.oLs; a.gWs = w.slab + L.oWs;
    a.chunk_stride = L.total;
  } else {
    a.gW1 = grads->gen_W1; a.gb1 = grads->gen_b1; a.gW2 = grads->gen_W2; a.gb2 = grads->gen_b2;
    a.gls = grads->log_sigma; a.gWs = grads->Ws;
    a.chunk_stride = 0;
  }
  // (dec_pad64.hip, D_g < 64 padded to 64 outputs: the virtual shape has 64 G rows of parameters, X has D_g G columns)
  const bool xhalf = w.dec_xhalf != 0;
  a.X = X; a.z = z; a.dzT = dz; a.B = B; a.N = N; a.D = xhalf ? s.G * w.dec_xhalf : s.D;
  a.xg = w.dec_xhalf;
  a.inv = 1.0f / ((float)S * (float)B);
  a.sig_off = gen_sig_off;
  a.ll_part = w.ll_part;
  a.zimg = nullptr;
  a.rng = nullptr;
  a.acc2 = nullptr;
  int rc;
  // test hooks: OIVAE_DEC=v1 the layer-by-layer fp32 template kernel (which also serves launches whose dz image
  // exceeds the 32-bit byte offsets of dec_mfma64_v4's atomics, N*K*4 >= 2^32), OIVAE_DEC=v4 the fp32
  // composed-first-layer kernel, OIVAE_DEC=x3 the single-wave-per-SIMD split-bf16 kernel, OIVAE_DEC=p the all-bf16
  // pair of waves, OIVAE_DEC=g dec_x3_64g (32x32x16 fp16 instructions; round 2's default until dec_x3_64s), OIVAE_DEC=s
  // (= unset) dec_x3_64s, the default below
  const bool force_v1 = dec_env != nullptr && dec_env[0] == 'v' && dec_env[1] == '1';
  const bool use_x3 = (dec_env != nullptr && (dec_env[0] == 'x' || dec_env[0] == 'p')) ||
                      (dec_env == nullptr && N * (int64_t)s.K * 4 >= ((int64_t)1 << 32));
  const bool use_pair = (dec_env != nullptr && dec_env[0] == 'p') || (dec_env == nullptr && N * (int64_t)s.K * 4 >= ((int64_t)1 << 32));
  // default for the all-64 shape: dec_x3_64s / dec_x3_64g (two groups per workgroup on fp16 terms); its scales assume the max|X|
  // that the workspace's ranges hold from the last pass over X (the forward encoder call of this step, or the
  // previous decoder call) and it checks that assumption against what it reads (range_guard): no host-side memory
  // of earlier calls.
  // (its atomics address dz with 32-bit byte offsets: larger launches take the bf16 pair)
  const bool fits32 = N * (int64_t)s.K * 4 < ((int64_t)1 << 32);
  // (it also fetches the generator matrices 16 bytes per lane: parameters that are not 16-byte aligned take the bf16 pair)
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(params->gen_W1) | reinterpret_cast<uintptr_t>(params->gen_W2) |
                           reinterpret_cast<uintptr_t>(params->Ws)) & 15) == 0;
  const bool all64 = dec_all64(s);
  const bool use_g = all64 && fits32 && aligned16 && (dec_env == nullptr || dec_env[0] == 'g' || dec_env[0] == 's');
  const bool use_s = use_g && (dec_env == nullptr || dec_env[0] == 's');   // dec_x3_64s: the same kernel on 16x16x32 instructions (default)
  if (xhalf && !use_s) return fail(OIVAE_E_SHAPE, "MFMA decoder: X by halves is dec_x3_64s's alone");
  uint32_t* rng = w.ranges;
  a.G = s.G;
  a.err = error_words_device();
  a.gate = nullptr;
  a.rng_seen = nullptr;
  if (all64 && (use_x3 || use_g)) {
    if (!w.x3_z) return fail(OIVAE_E_WORKSPACE, "x3 decoder: no z image in the workspace");
    const int64_t zt = (N + 31) / 32;
    if (use_g) {
      // [5] max|z|, [6] max|x| the kernel reads, [7] redo flag: set by x3_split_z from the per-workgroup maxima
      if (!w.dec_zmax) return fail(OIVAE_E_WORKSPACE, "x3 decoder: no room for the maxima of z in the workspace");
      hipLaunchKernelGGL(dec_absmax_f32, dim3(kZmaxParts), dim3(256), 0, st, z, N * 64, w.dec_zmax, 1);
      OIVAE_LAUNCH_CHECK();
    } else {
      OIVAE_HIP_CHECK(hipMemsetAsync(dz, 0, sizeof(float) * N * s.K, st));
    }
    hipLaunchKernelGGL(x3_split_z, dim3((unsigned)((zt * 512 + 255) / 256)), dim3(256), 0, st, z, N, w.x3_z,
                       use_g ? rng : (uint32_t*)nullptr, use_s ? 1 : 0, (const uint32_t*)w.dec_zmax, dz);
    OIVAE_LAUNCH_CHECK();
    a.zimg = w.x3_z;
    a.rng = rng;
    if (use_g) {
      if (use_s)
        OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xhalf ? &dec_x3_64s<true> : &dec_x3_64s<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
      const dim3 grid((unsigned)((s.G + 1) / 2), (unsigned)chunks);
      auto run = [&](const MfmaDecArgs& aa) -> int {
        if (!use_s) return launch_xcheck_decoder('g', aa, s.G, chunks, st);
        if (xhalf) hipLaunchKernelGGL(dec_x3_64s<true>, grid, dim3(512), (size_t)kPairLds, st, aa);
        else hipLaunchKernelGGL(dec_x3_64s<false>, grid, dim3(512), (size_t)kPairLds, st, aa);
        OIVAE_LAUNCH_CHECK();
        return OIVAE_OK;
      };
      a.rng_seen = rng + 6;
      a.acc2 = use_s ? w.dec_acc2 : nullptr;
      if (int rcl = run(a)) return rcl;
      launch_range_guard(rng, 6, 7, st);
      OIVAE_LAUNCH_CHECK();
      // the repeat, gated on the guard's flag: dz starts from zero again, every other output is overwritten
      int64_t zb = (N * 64 / 4 + 255) / 256;
      if (zb > 2048) zb = 2048;
      hipLaunchKernelGGL(gated_zero_f32, dim3((unsigned)zb), dim3(256), 0, st, dz, N * 64, (const uint32_t*)(rng + 7));
      a.rng_seen = nullptr;
      a.gate = rng + 7;
      if (int rcl = run(a)) return rcl;
    } else {
      if (int rcl = launch_xcheck_decoder(use_pair ? 'p' : 'x', a, s.G, chunks, st)) return rcl;
    }
    rc = OIVAE_OK;
  } else if (all64 && !force_v1 && N * (int64_t)s.K * 4 < (int64_t)1 << 32) {
    OIVAE_HIP_CHECK(hipMemsetAsync(dz, 0, sizeof(float) * N * s.K, st));
    rc = launch_xcheck_decoder('4', a, s.G, chunks, st);
  } else {
    // dimension-generic kernel: one instantiation per combination of 32 / 64
    OIVAE_HIP_CHECK(hipMemsetAsync(dz, 0, sizeof(float) * N * s.K, st));
    rc = fail(OIVAE_E_SHAPE, "MFMA decoder: unsupported dims K=%d H=%d M=%d D_g=%d", s.K, s.H, s.M, s.uniform_group_dim);
#define OIVAE_DEC_ONE(KK, HH, MM, DD) \
    if (s.K == KK && s.H == HH && s.M == MM && s.uniform_group_dim == DD) rc = launch_one<KK, HH, MM, DD>(a, s.G, chunks, st);
#define OIVAE_DEC_KH(KK, HH) OIVAE_DEC_ONE(KK, HH, 32, 32) OIVAE_DEC_ONE(KK, HH, 32, 64) OIVAE_DEC_ONE(KK, HH, 64, 32) OIVAE_DEC_ONE(KK, HH, 64, 64)
    OIVAE_DEC_KH(32, 32) OIVAE_DEC_KH(32, 64) OIVAE_DEC_KH(64, 32) OIVAE_DEC_KH(64, 64)
#undef OIVAE_DEC_KH
#undef OIVAE_DEC_ONE
  }
  if (rc) return rc;
  if (chunks > 1)
    if (int rc2 = launch_slab_reduce(w.slab, L, chunks, grads, st)) return rc2;
  return OIVAE_OK;
}

}  // namespace oivae

#if defined(OIVAE_ABL_TIMES) || defined(OIVAE_ABL_TIMES2) || defined(OIVAE_ABL_TIMES3)
extern "C" int oivae_probe_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(oivae::g_probe), sizeof(unsigned long long) * (size_t)n);
}
#endif

