// Decoder forward + backward for MANY SMALL groups (the BASELINE "wide-prox" shape: G = 8192 groups of 4 outputs,
// K = 256, H = 8): same mathematics as dec_generic (/root/reference/lib/models.py:91-95, lib/distributions.py:40-45,
// lib/oivae.py:55-59,75), different decomposition.
//
// One workgroup per group wastes the chip when a group is a 256x8 matrix and four outputs: its dz contribution alone
// is N*K float atomics per group.  Here the G latent-to-group maps are ONE matrix product over the concatenated
// inputs of all generators, which is affordable exactly when groups are small (N*G*H floats of workspace):
//     Hc [N, G*H] = z [N, K] * Wc^T          Wc[(g,h), k] = Ws[g, k, h]          (mfma_gemm_nt)
//     per (row, group): generator Linear(H, D_g) (+ Tanh), Gaussian log-likelihood, dmu, dh; dW2, db2, dlog_sigma
//     Ct [K, G*H] = z^T * dH                 -> dWs[g, k, h]                     (mfma_gemm_tn, rows in order)
//     dz [N, K]   = dH * Wk^T                Wk[k, (g,h)] = Ws[g, k, h]          (mfma_gemm_nt, split over G*H)
// so the three big products run on the fp32 matrix cores and dz needs no atomics (the step becomes bit-reproducible).
// Roofline: the GEMMs are MFMA-bound (3 * 2*N*K*G*H flops), the group kernel HBM-bound (one read and one write of Hc).
#include "common.h"
#include "x3_split.h"

#include <stdlib.h>

namespace oivae {

int mfma_gemm_nt_splits(int64_t M, int Kd, int J);
int mfma_gemm_nt(const float* A, int64_t M, int Kd, const float* W, int J, float* C, hipStream_t st);
int mfma_gemm_tn(const float* P, const float* X, int64_t M, int J, int Dn, float* C, hipStream_t st);

constexpr int kFlatGB = 16;  // groups per block of the group kernel
constexpr int kFlatRL = 16;  // row lanes per block

bool flat_decoder_applies(const oivae_shape_t& s, int64_t N) {
  if (test_hooks().force_generic) return false;  // test hook
  if (s.gen_type == OIVAE_GEN_MLP) return false;
  const int dg = s.uniform_group_dim;
  if (!((dg == 4 || dg == 8) && (s.H == 4 || s.H == 8))) return false;
  if (s.K % 4 != 0 || s.K < 32 || s.G < 64) return false;
  const int64_t gh = (int64_t)s.G * s.H;
  if (gh > (1 << 24)) return false;
  return N * gh * 4 <= ((int64_t)1 << 30);  // Hc fits 1 GiB of workspace
}

// The three products on two fp16 terms (round 2, last part): they have the encoder GEMMs' shapes -- a long-reduction
// "NT" product (dz: N x K over G*H, = the encoder forward with X := dH, W := Wk), a weight-gradient "TN" product
// (Ct: K x G*H over N, = the encoder backward with dpre := z) and Hc = z Wc^T (the forward kernel again, with the
// latent dimension as its short reduction and the G*H columns as its 128-wide column tiles) -- so they run on
// enc_h2_fwd_gemm / enc_h2_bwd_gemm (x3_encoder.hip) with their scale / guard / gated-repeat protocol, each product
// family on its own 16 words of the workspace's ranges.  Wide-prox step: the three fp32-instruction GEMMs took 0.95
// of 1.73 ms.  Shapes outside the h2 kernels' tiles, and OIVAE_ENC=bf16 / fp32, keep the fp32 matrix instruction.
static bool flat_h2_applies(const oivae_shape_t& s, int64_t N) {
  if (test_hooks().enc != 0) return false;
  const int64_t gh = (int64_t)s.G * s.H;
  return s.K % 128 == 0 && gh % 128 == 0 && N % 32 == 0;
}
static int flat_h2_dz_splits(const oivae_shape_t& s, int64_t N) {
  const int64_t tiles = ((N + 127) / 128) * (s.K / 128);
  const int64_t ktiles = (int64_t)s.G * s.H / 32;
  int sp = 1;
  while (sp < 32 && tiles * sp < 256 && ktiles / (sp * 2) >= 16) sp *= 2;
  return sp;
}

// floats of workspace the flat path needs: Hc | Wc | Wk | Ct | dz partial images | h2 operand images
int64_t flat_decoder_floats(const oivae_shape_t& s, int64_t N, FlatLayout* L) {
  const int64_t gh = (int64_t)s.G * s.H;
  FlatLayout l;
  int64_t off = 0;
  auto take = [&](int64_t n) { const int64_t o = off; off += (n + 63) / 64 * 64; return o; };
  l.h2 = flat_h2_applies(s, N);
  l.oHc = take(N * gh);
  l.oWc = take(gh * s.K);
  l.oWk = take(gh * s.K);
  l.oCt = take(gh * s.K);
  l.splits = l.h2 ? flat_h2_dz_splits(s, N) : mfma_gemm_nt_splits(N, (int)gh, s.K);
  l.oDzp = take((l.splits > 1 || l.h2) ? (int64_t)l.splits * N * s.K : 0);
  // images in two fp16 terms: 4 bytes per element of the operand (rows padded to 128); z^T sized as the encoder sizes dpre^T
  l.oImgA = take(l.h2 ? gh * s.K : 0);
  l.oImgC = take(l.h2 ? gh * s.K : 0);
  l.oImgB = take(l.h2 ? (int64_t)s.K * ((N + 31) / 32) * 48 : 0);
  l.total = off;
  if (L) *L = l;
  return off;
}

// Wc[(g*H+h)*K + k] = Wk[k*G*H + g*H + h] = Ws[(g*K+k)*H + h]
__global__ __launch_bounds__(256) void flat_permute_ws(const float* __restrict__ Ws, int G, int K, int H,
                                                       float* __restrict__ Wc, float* __restrict__ Wk) {
  const int64_t n = (int64_t)G * K * H;
  const int64_t gh = (int64_t)G * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i % H);
    const int64_t gk = i / H;
    const int k = (int)(gk % K);
    const int64_t g = gk / K;
    const float v = Ws[i];
    Wc[(g * H + h) * K + k] = v;
    Wk[(int64_t)k * gh + g * H + h] = v;
  }
}

// Both weight images of the h2 path straight from Ws [G, K, H], in the layout h2_split_rows (x3_encoder.hip) gives an
// image: per (row, 32 reduction slots) 32 dwords = two sub-steps of [hi 8 halves | hi 8 halves | lo | lo]:
//   imgA: row k, reduction over the flattened (g, h) (the dz product's W := Wk),
//   imgC: row (g, h), reduction over k (the h product's W := Wc),
// scaled by the power of two of max|Ws| (maxw[0]; also copied to maxw_copy[0] for the other product's ranges).
// A workgroup owns 32 / H consecutive groups = one 32-slot reduction tile of imgA for every k, and 32 rows of imgC.
template <int H>
__global__ __launch_bounds__(256) void flat_split_ws(const float* __restrict__ Ws, int G, int K,
                                                     uint32_t* __restrict__ imgA, uint32_t* __restrict__ imgC,
                                                     const uint32_t* __restrict__ maxw, uint32_t* __restrict__ maxw_copy) {
  constexpr int GB = 32 / H, LH = H + 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);   // [GB][K][H + 1]
  const int tid = threadIdx.x;
  const int g0 = blockIdx.x * GB;
  if (blockIdx.x == 0 && tid == 0) maxw_copy[0] = maxw[0];
  float sc, inv;
  h2_scale_of(maxw[0], &sc, &inv);
  const int per = K * H;
  for (int i = tid; i < GB * per; i += 256) {
    const int gl = i / per, rem = i - gl * per;
    const int k = rem / H, h = rem - k * H;
    tile[(gl * K + k) * LH + h] = (g0 + gl < G) ? Ws[(int64_t)(g0 + gl) * per + rem] * sc : 0.f;
  }
  __syncthreads();
  const int KT_A = G * H / 32, KT_C = K / 32;
  auto put = [](uint32_t* dst, const float (&v)[8]) {
    x3_u32x4 Q[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const Split2 q = split2_pair(v[2 * e], v[2 * e + 1]);
      Q[0][e] = q.h; Q[1][e] = q.l;
    }
    *reinterpret_cast<x3_u32x4*>(dst) = Q[0];
    *reinterpret_cast<x3_u32x4*>(dst + 8) = Q[1];
  };
  for (int it = tid; it < K * 4; it += 256) {   // imgA: (row k, chunk c of this workgroup's reduction tile)
    const int k = it >> 2, c = it & 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * c + e, gl = r / H, h = r - gl * H;
      v[e] = tile[(gl * K + k) * LH + h];
    }
    put(imgA + (((int64_t)k * KT_A + blockIdx.x) * 2 + (c >> 1)) * 16 + 4 * (c & 1), v);
  }
  const int cpr = K / 8;   // chunks per row of imgC
  for (int it = tid; it < 32 * cpr; it += 256) {   // imgC: (row rl of this workgroup's 32, chunk c over k)
    const int rl = it / cpr, c = it - rl * cpr;
    const int gl = rl / H, h = rl - gl * H;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(gl * K + 8 * c + e) * LH + h];
    const int64_t row = (int64_t)blockIdx.x * 32 + rl;
    put(imgC + ((row * KT_C + (c >> 2)) * 2 + ((c >> 1) & 1)) * 16 + 4 * (c & 1), v);
  }
}

// dWs[(g*K+k)*H + h] = Ct[k*G*H + g*H + h]
__global__ __launch_bounds__(256) void flat_unpermute_dws(const float* __restrict__ Ct, int G, int K, int H,
                                                          float* __restrict__ dWs) {
  const int64_t n = (int64_t)G * K * H;
  const int64_t gh = (int64_t)G * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i % H);
    const int64_t gk = i / H;
    const int k = (int)(gk % K);
    const int64_t g = gk / K;
    dWs[i] = Ct[(int64_t)k * gh + g * H + h];
  }
}

// One thread per (row lane, group); 16 consecutive groups per block so that a row's 16*H floats are one contiguous
// read.  The block walks all N rows of its groups, so every weight gradient is complete (and deterministic) here.
template <int H, int DG, bool TANH>
__global__ __launch_bounds__(256) void flat_group_kernel(float* __restrict__ Hc, const float* __restrict__ X, int64_t B,
                                                         int64_t N, int D, int G, const float* __restrict__ W2,
                                                         const float* __restrict__ b2, const float* __restrict__ ls,
                                                         float inv, float sig_off, float* __restrict__ gW2,
                                                         float* __restrict__ gb2, float* __restrict__ gls,
                                                         double* __restrict__ ll_part) {
  constexpr int NACC = DG * H + 2 * DG + 1;  // dW2 | db2 | sum(1 - q^2) | log-likelihood
  __shared__ float red[kFlatRL][kFlatGB][NACC + 1];
  __shared__ double llred[kFlatGB];
  const int tid = threadIdx.x;
  const int gl = tid % kFlatGB, rl = tid / kFlatGB;
  const int g = blockIdx.x * kFlatGB + gl;
  const bool gok = g < G;
  const int gg = gok ? g : G - 1;
  float w[DG][H], bias[DG], isg[DG], lsg[DG];
#pragma unroll
  for (int d = 0; d < DG; ++d) {
#pragma unroll
    for (int h = 0; h < H; ++h) w[d][h] = W2[((int64_t)gg * DG + d) * H + h];
    bias[d] = b2[(int64_t)gg * DG + d];
    const float sg = expf(ls[(int64_t)gg * DG + d]) + sig_off;
    isg[d] = sg;
    lsg[d] = logf(sg);
  }
  float aw[DG][H], ab[DG], ar[DG];
  float ll = 0.f;
#pragma unroll
  for (int d = 0; d < DG; ++d) {
    ab[d] = 0.f; ar[d] = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) aw[d][h] = 0.f;
  }
  const int64_t gh = (int64_t)G * H;
  for (int64_t n = rl; n < N; n += kFlatRL) {
    float* hp = Hc + n * gh + (int64_t)gg * H;
    float hv[H], act[H];
#pragma unroll
    for (int q = 0; q < H / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(hp + 4 * q);
      hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int h = 0; h < H; ++h) act[h] = TANH ? tanhf(hv[h]) : hv[h];
    const float* xp = X + (n % B) * (int64_t)D + (int64_t)gg * DG;
    float gm[DG];
#pragma unroll
    for (int d = 0; d < DG; ++d) {
      float mu = bias[d];
#pragma unroll
      for (int h = 0; h < H; ++h) mu = fmaf(act[h], w[d][h], mu);
      const float q = (xp[d] - mu) / isg[d];
      ll += -0.5f * kLog2Pi - lsg[d] - 0.5f * q * q;
      gm[d] = -(q / isg[d]) * inv;
      ar[d] += 1.f - q * q;
      ab[d] += gm[d];
#pragma unroll
      for (int h = 0; h < H; ++h) aw[d][h] = fmaf(gm[d], act[h], aw[d][h]);
    }
    float dh[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < DG; ++d) a = fmaf(gm[d], w[d][h], a);
      dh[h] = TANH ? a * (1.f - act[h] * act[h]) : a;
    }
    if (gok) {
#pragma unroll
      for (int q = 0; q < H / 4; ++q)
        *reinterpret_cast<float4*>(hp + 4 * q) = make_float4(dh[4 * q], dh[4 * q + 1], dh[4 * q + 2], dh[4 * q + 3]);
    }
  }
  // fold the 16 row lanes of each group in a fixed order
  float* my = red[rl][gl];
#pragma unroll
  for (int d = 0; d < DG; ++d) {
#pragma unroll
    for (int h = 0; h < H; ++h) my[d * H + h] = aw[d][h];
    my[DG * H + d] = ab[d];
    my[DG * H + DG + d] = ar[d];
  }
  my[NACC - 1] = ll;
  __syncthreads();
  for (int i = tid; i < kFlatGB * NACC; i += 256) {
    const int g2 = i / NACC, j = i - g2 * NACC;
    const int gq = blockIdx.x * kFlatGB + g2;
    if (gq >= G) continue;
    float a = 0.f;
    double ad = 0.0;
    for (int r = 0; r < kFlatRL; ++r) { a += red[r][g2][j]; ad += (double)red[r][g2][j]; }
    if (j < DG * H) gW2[(int64_t)gq * DG * H + j] = a;
    else if (j < DG * H + DG) gb2[(int64_t)gq * DG + (j - DG * H)] = a;
    else if (j < DG * H + 2 * DG) {
      const int d = j - DG * H - DG;
      const float e = expf(ls[(int64_t)gq * DG + d]);
      gls[(int64_t)gq * DG + d] = a * (e * inv / (e + sig_off));
    } else llred[g2] = ad;
  }
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int i = 0; i < kFlatGB; ++i)
      if (blockIdx.x * kFlatGB + i < G) tot += llred[i];
    ll_part[blockIdx.x] = tot;
  }
}

int flat_decoder_blocks(const oivae_shape_t& s) { return (s.G + kFlatGB - 1) / kFlatGB; }

int launch_decoder_flat(const oivae_shape_t& s, const oivae_tensors_t* params, const oivae_tensors_t* grads,
                        const float* X, int64_t B, int32_t S, const float* z, float gen_sig_off, float* dz,
                        Workspace& w, hipStream_t st) {
  const int64_t N = B * (int64_t)S;
  const int GH = s.G * s.H;
  FlatLayout L;
  flat_decoder_floats(s, N, &L);
  float* Hc = w.flat + L.oHc;
  float* Wc = w.flat + L.oWc;
  float* Wk = w.flat + L.oWk;
  float* Ct = w.flat + L.oCt;
  float* dzp = w.flat + L.oDzp;
  if (reinterpret_cast<uintptr_t>(z) & 15) return fail(OIVAE_E_ARG, "flat decoder: z must be 16-byte aligned");
  if (!L.h2) {
    hipLaunchKernelGGL(flat_permute_ws, dim3(2048), dim3(256), 0, st, params->Ws, s.G, s.K, s.H, Wc, Wk);
    OIVAE_LAUNCH_CHECK();
  }
  uint32_t* rngA = w.ranges + 16;   // dz and Ct products (X := dH)
  uint32_t* rngC = w.ranges + 32;   // h product (X := z)
  oivae_shape_t sa = s, sc = s;      // the encoder shapes the h2 launchers see: J = 2 K', reduction D'
  sa.K = s.K / 2; sa.D = GH;
  sc.K = GH / 2; sc.D = s.K;
  if (L.h2) {
    if ((reinterpret_cast<uintptr_t>(w.flat) & 15) != 0) return fail(OIVAE_E_ARG, "flat decoder: workspace must be 16-byte aligned");
    // max|Ws| once, both weight images in one pass over Ws (no fp32 copies of the permuted matrices on this path)
    uint32_t* imgA = reinterpret_cast<uint32_t*>(w.flat + L.oImgA);
    uint32_t* imgC = reinterpret_cast<uint32_t*>(w.flat + L.oImgC);
    OIVAE_HIP_CHECK(hipMemsetAsync(rngC + 1, 0, 4, st));
    launch_absmax_f32(params->Ws, (int64_t)s.G * s.K * s.H, rngC + 1, st);
    OIVAE_LAUNCH_CHECK();
    {
      const int gb = 32 / s.H;
      const unsigned blocks = (unsigned)((s.G + gb - 1) / gb);
      const size_t lds = sizeof(float) * (size_t)gb * s.K * (s.H + 1);
      if (s.H == 8) {
        OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&flat_split_ws<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(flat_split_ws<8>, dim3(blocks), dim3(256), lds, st, params->Ws, s.G, s.K, imgA, imgC, (const uint32_t*)(rngC + 1), rngA + 1);
      } else {
        OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&flat_split_ws<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(flat_split_ws<4>, dim3(blocks), dim3(256), lds, st, params->Ws, s.G, s.K, imgA, imgC, (const uint32_t*)(rngC + 1), rngA + 1);
      }
      OIVAE_LAUNCH_CHECK();
    }
    oivae_tensors_t pc = {};
    if (int rc = launch_encoder_fwd_x3(sc, &pc, z, N, Hc, imgC, rngC, 1, st, true)) return rc;
  } else {
    // Hc = z Wc^T   (one partial image: the reduction length K is short)
    if (mfma_gemm_nt_splits(N, s.K, GH) != 1) return fail(OIVAE_E_SHAPE, "flat decoder: unexpected split of the h product");
    if (int rc = mfma_gemm_nt(z, N, s.K, Wc, GH, Hc, st)) return rc;
  }
  const float inv = 1.0f / ((float)S * (float)B);
  const dim3 grid((unsigned)flat_decoder_blocks(s));
  const bool th = s.gen_type == OIVAE_GEN_TANH_LINEAR;
#define OIVAE_FLAT(HH, DD)                                                                                            \
  if (s.H == HH && s.uniform_group_dim == DD) {                                                                       \
    if (th) hipLaunchKernelGGL((flat_group_kernel<HH, DD, true>), grid, dim3(256), 0, st, Hc, X, B, N, s.D, s.G,      \
                               params->gen_W2, params->gen_b2, params->log_sigma, inv, gen_sig_off, grads->gen_W2,    \
                               grads->gen_b2, grads->log_sigma, w.ll_part);                                           \
    else hipLaunchKernelGGL((flat_group_kernel<HH, DD, false>), grid, dim3(256), 0, st, Hc, X, B, N, s.D, s.G,        \
                            params->gen_W2, params->gen_b2, params->log_sigma, inv, gen_sig_off, grads->gen_W2,       \
                            grads->gen_b2, grads->log_sigma, w.ll_part);                                              \
  }
  OIVAE_FLAT(4, 4) OIVAE_FLAT(4, 8) OIVAE_FLAT(8, 4) OIVAE_FLAT(8, 8)
#undef OIVAE_FLAT
  OIVAE_LAUNCH_CHECK();
  if (L.h2) {
    // dz first: its forward-type kernel sees dH for the first time, verifies the scale it assumed and leaves max|dH|
    // in its ranges; the weight-gradient kernel that follows scales dH by that
    oivae_tensors_t pa = {};
    if (int rc = launch_encoder_fwd_x3(sa, &pa, Hc, N, dzp, reinterpret_cast<uint32_t*>(w.flat + L.oImgA), rngA, L.splits, st, true)) return rc;
    launch_partial_sum(dzp, N * s.K, N * s.K, L.splits, dz, st);
    OIVAE_LAUNCH_CHECK();
    OIVAE_HIP_CHECK(hipMemsetAsync(rngA + 8, 0, 8, st));   // the backward kernel's own maximum of dH and its flag
    uint32_t* imgB = reinterpret_cast<uint32_t*>(w.flat + L.oImgB);
    OIVAE_HIP_CHECK(hipMemsetAsync(h2_bwd_colmax(imgB, s.K, N), 0, sizeof(uint32_t) * s.K, st));   // column maxima of z
    oivae_tensors_t ga = {};
    ga.enc_Wmu = Ct; ga.enc_Wsig = Ct + (int64_t)(s.K / 2) * GH;
    if (int rc = launch_encoder_bwd_x3(sa, &ga, Hc, z, N, imgB, rngA, st)) return rc;
    hipLaunchKernelGGL(flat_unpermute_dws, dim3(2048), dim3(256), 0, st, Ct, s.G, s.K, s.H, grads->Ws);
    OIVAE_LAUNCH_CHECK();
    return OIVAE_OK;
  }
  // dWs: Ct[K, GH] = z^T dH, then back to [G, K, H]
  if (int rc = mfma_gemm_tn(z, Hc, N, s.K, GH, Ct, st)) return rc;
  hipLaunchKernelGGL(flat_unpermute_dws, dim3(2048), dim3(256), 0, st, Ct, s.G, s.K, s.H, grads->Ws);
  OIVAE_LAUNCH_CHECK();
  // dz[N, K] = dH Wk^T, reduction over G*H split into L.splits slices summed in index order
  if (L.splits > 1) {
    if (int rc = mfma_gemm_nt(Hc, N, GH, Wk, s.K, dzp, st)) return rc;
    launch_partial_sum(dzp, N * s.K, N * s.K, L.splits, dz, st);
    OIVAE_LAUNCH_CHECK();
  } else {
    if (int rc = mfma_gemm_nt(Hc, N, GH, Wk, s.K, dz, st)) return rc;
  }
  return OIVAE_OK;
}

}  // namespace oivae
