// Shape-generic kernels of the oi-VAE step (any K, H, M, ragged group dims).
//
// These are the correctness-first paths that serve the small, launch-bound
// configurations (bars, mocap, MEG: SURVEY.md section 8 C1-C3, C5) and every shape
// the MFMA fast paths (mfma_*.hip) do not cover.  All reductions that feed
// parameter gradients run in a fixed order (per-block slabs, summed in index
// order), so decoder and encoder weight gradients are reproducible run to run;
// only dz, summed over groups with float atomics, is not.
#include "common.h"

#include <stdlib.h>

namespace oivae {

int64_t single_decoder_lds_bytes(const oivae_shape_t& s);
bool single_decoder_applies(const oivae_shape_t& s);

// ------------------------------------------------------------------------------
// block reduction helper (double), result valid in thread 0
// ------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_double(double v, double* sm /*[256]*/) {
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  return sm[0];
}

// ------------------------------------------------------------------------------
// encoder forward: two linear heads + exp/affine + reparameterisation + KL
// (lib/models.py:15-16, bars_data_oivae.py:66-87, lib/utils.py:9-18,
//  lib/distributions.py:36-38)
// ------------------------------------------------------------------------------
constexpr int kEncRows = 16;   // batch rows per block
constexpr int kEncDC = 256;    // X columns staged per step

__global__ __launch_bounds__(256) void enc_fwd_generic(
    const float* __restrict__ X, int64_t B, int D, int K, const float* __restrict__ Wmu,
    const float* __restrict__ Wsig, float* __restrict__ pre) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sX = reinterpret_cast<float*>(smem_raw);            // [kEncRows][kEncDC+1]
  const int J = 2 * K;
  const int tid = threadIdx.x;
  const int r = tid >> 4, jj = tid & 15;
  const int64_t b0 = (int64_t)blockIdx.x * kEncRows;
  // output columns are processed 8*16 = 128 at a time
  for (int jb = 0; jb < J; jb += 128) {
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    const float* wrow[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = jb + jj + 16 * q;
      wrow[q] = (j < K) ? (Wmu + (int64_t)j * D) : (j < J ? Wsig + (int64_t)(j - K) * D : nullptr);
    }
    for (int d0 = 0; d0 < D; d0 += kEncDC) {
      const int dn = min(kEncDC, D - d0);
      __syncthreads();
      for (int rr = 0; rr < kEncRows; ++rr) {
        const int64_t b = b0 + rr;
        if (tid < dn) sX[rr * (kEncDC + 1) + tid] = (b < B) ? X[b * D + d0 + tid] : 0.f;
      }
      __syncthreads();
      const float* xr = sX + r * (kEncDC + 1);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (wrow[q]) {
          const float* w = wrow[q] + d0;
          float a = acc[q];
          for (int dd = 0; dd < dn; ++dd) a = fmaf(xr[dd], w[dd], a);
          acc[q] = a;
        }
      }
    }
    const int64_t b = b0 + r;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = jb + jj + 16 * q;
      if (j < J && b < B) pre[b * J + j] = acc[q];
    }
  }
}

// bias, exp/affine head, reparameterisation and KL over pre[B, 2K]
__global__ __launch_bounds__(256) void enc_fwd_epilogue(
    const float* __restrict__ pre, int nsplit, int64_t B, int K, const float* __restrict__ bmu,
    const float* __restrict__ bsig, int S, const float* __restrict__ eps, float c, float sig_off,
    const float* __restrict__ prior_mu, const float* __restrict__ prior_sigma, int prior_stride,
    float* __restrict__ mu_q, float* __restrict__ sig_q, float* __restrict__ ea, float* __restrict__ z,
    double* __restrict__ kl_part) {
  __shared__ double sred[256];
  const int J = 2 * K;
  const int64_t total = B * K;
  double kl = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / K;
    const int k = (int)(i - b * K);
    float pm = pre[b * J + k], pa = pre[b * J + K + k];
    for (int s0 = 1; s0 < nsplit; s0 += 7) {  // split-K partial images, summed in index order; requested seven at a time
      float vm[7], va[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const bool ok = s0 + u < nsplit;
        vm[u] = ok ? pre[(int64_t)(s0 + u) * B * J + b * J + k] : 0.f;
        va[u] = ok ? pre[(int64_t)(s0 + u) * B * J + b * J + K + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 7; ++u) { pm += vm[u]; pa += va[u]; }
    }
    const float mu = pm + bmu[k];
    const float a = pa + bsig[k];
    const float e = expf(a);
    const float sg = e * c + sig_off;
    const float mp = prior_mu[b * prior_stride + k];
    const float sp = prior_sigma[b * prior_stride + k];
    const float dm = mu - mp;
    kl += (double)(-0.5f + (sg * sg + dm * dm) / (2.f * sp * sp) + logf(sp) - logf(sg));
    mu_q[i] = mu;
    sig_q[i] = sg;
    ea[i] = e;
    for (int s = 0; s < S; ++s) {
      const int64_t n = (int64_t)s * B + b;
      z[n * K + k] = mu + sg * eps[n * K + k];
    }
  }
  const double tot = block_sum_double(kl, sred);
  if (threadIdx.x == 0) kl_part[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------
// grouped decoder forward + backward, generic shapes
// (lib/models.py:91-95, lib/distributions.py:40-45,88-93, lib/oivae.py:54-59,75)
// ------------------------------------------------------------------------------
struct DecArgs {
  int D, K, G, H, M, Hin, gen_type;
  const int32_t* group_off;
  const float *W1, *b1, *W2, *b2, *ls, *Ws;  // parameters
  float *gW1, *gb1, *gW2, *gb2, *gls, *gWs;  // destinations of chunk 0
  int64_t chunk_stride;                      // floats between chunks' destinations (0 if one chunk)
  const float* X;
  const float* z;
  float* dz;
  int64_t B, N;
  float inv;          // 1/(S*B)
  float sig_off;
  int DC;             // output-column chunk
  double* ll_part;
};

template <int R>
__global__ __launch_bounds__(256) void dec_generic(DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ double sred[256];
  const int tid = threadIdx.x;
  const int g = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
  const int K = a.K, H = a.H, M = a.M, Hin = a.Hin, typ = a.gen_type, DC = a.DC;
  const int off = a.group_off[g];
  const int Dg = a.group_off[g + 1] - off;
  const int HP = H | 1, HinP = Hin | 1;
  // ---- carve LDS ----
  float* p = reinterpret_cast<float*>(smem_raw);
  float* sWs = p;  p += K * HP;
  float* sW1 = p;  p += (typ == OIVAE_GEN_MLP) ? M * HP : 0;
  float* sb1 = p;  p += (typ == OIVAE_GEN_MLP) ? M : 0;
  float* sW2 = p;  p += DC * HinP;           // one column chunk of W2 at a time
  float* sb2 = p;  p += DC;
  float* sSig = p; p += DC;
  float* sLogSig = p; p += DC;
  float* sCoef = p; p += DC;
  float* sz = p;   p += R * K;
  float* sh = p;   p += R * H;
  float* sact = (typ == OIVAE_GEN_LINEAR) ? sh : p;  p += (typ == OIVAE_GEN_LINEAR) ? 0 : R * Hin;
  float* sdact = p; p += R * Hin;
  float* sdh = (typ == OIVAE_GEN_MLP) ? p : sdact;   p += (typ == OIVAE_GEN_MLP) ? R * H : 0;
  float* sgmu = p; p += R * DC;
  float* sres = p; p += R * DC;

  // ---- group weights -> LDS (once) ----
  const float* Wsg = a.Ws + (int64_t)g * K * H;
  for (int i = tid; i < K * H; i += 256) sWs[(i / H) * HP + (i % H)] = Wsg[i];
  if (typ == OIVAE_GEN_MLP) {
    const float* W1g = a.W1 + (int64_t)g * M * H;
    for (int i = tid; i < M * H; i += 256) sW1[(i / H) * HP + (i % H)] = W1g[i];
    for (int i = tid; i < M; i += 256) sb1[i] = a.b1[(int64_t)g * M + i];
  }
  // destinations for this chunk
  const int64_t cs = a.chunk_stride * chunk;
  float* gW2 = a.gW2 + cs + (int64_t)off * Hin;
  float* gb2 = a.gb2 + cs + off;
  float* gls = a.gls + cs + off;
  float* gWs = a.gWs + cs + (int64_t)g * K * H;
  float* gW1 = (typ == OIVAE_GEN_MLP) ? a.gW1 + cs + (int64_t)g * M * H : nullptr;
  float* gb1 = (typ == OIVAE_GEN_MLP) ? a.gb1 + cs + (int64_t)g * M : nullptr;
  const float* W2g = a.W2 + (int64_t)off * Hin;

  double ll_acc = 0.0;
  const int64_t ntiles = (a.N + R - 1) / R;
  for (int64_t tile = chunk; tile < ntiles; tile += nchunks) {
    const int64_t n0 = tile * R;
    __syncthreads();
    // A1: z tile
    for (int i = tid; i < R * K; i += 256) {
      const int64_t n = n0 + i / K;
      sz[i] = (n < a.N) ? a.z[n * K + (i % K)] : 0.f;
    }
    __syncthreads();
    // A2: h = z Ws[g]
    for (int i = tid; i < R * H; i += 256) {
      const int r = i / H, j = i - r * H;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(sz[r * K + k], sWs[k * HP + j], acc);
      sh[i] = acc;
    }
    __syncthreads();
    // A3: activation feeding the last linear layer
    if (typ == OIVAE_GEN_MLP) {
      for (int i = tid; i < R * M; i += 256) {
        const int r = i / M, m = i - r * M;
        float acc = sb1[m];
        for (int j = 0; j < H; ++j) acc = fmaf(sh[r * H + j], sW1[m * HP + j], acc);
        sact[i] = tanhf(acc);
      }
    } else if (typ == OIVAE_GEN_TANH_LINEAR) {
      for (int i = tid; i < R * H; i += 256) sact[i] = tanhf(sh[i]);
    }
    for (int i = tid; i < R * Hin; i += 256) sdact[i] = 0.f;
    __syncthreads();
    // ---- loop over output-column chunks of this group ----
    float ll_tile = 0.f;
    for (int dc0 = 0; dc0 < Dg; dc0 += DC) {
      const int dn = min(DC, Dg - dc0);
      // stage this chunk of the last layer
      for (int i = tid; i < dn * Hin; i += 256) sW2[(i / Hin) * HinP + (i % Hin)] = W2g[(int64_t)dc0 * Hin + i];
      for (int i = tid; i < dn; i += 256) {
        const float lsv = a.ls[off + dc0 + i];
        const float e = expf(lsv);
        const float sg = e + a.sig_off;
        sb2[i] = a.b2[off + dc0 + i];
        sSig[i] = sg;
        sLogSig[i] = logf(sg);
        sCoef[i] = e * a.inv / sg;   // d loss / d log_sigma = sum_n (1 - ((x-mu)/sigma)^2) * e*inv/sigma
      }
      __syncthreads();
      // B1: mu, residual, log-likelihood, dL/dmu
      for (int i = tid; i < R * dn; i += 256) {
        const int r = i / dn, dd = i - r * dn;
        const int64_t n = n0 + r;
        float gm = 0.f, res = 0.f;
        if (n < a.N) {
          float mu = sb2[dd];
          for (int j = 0; j < Hin; ++j) mu = fmaf(sact[r * Hin + j], sW2[dd * HinP + j], mu);
          const float x = a.X[(n % a.B) * (int64_t)a.D + off + dc0 + dd];
          const float sg = sSig[dd];
          const float q = (x - mu) / sg;
          ll_tile += -0.5f * kLog2Pi - sLogSig[dd] - 0.5f * q * q;
          gm = -(q / sg) * a.inv;
          res = 1.f - q * q;
        }
        sgmu[r * DC + dd] = gm;
        sres[r * DC + dd] = res;
      }
      __syncthreads();
      // B3: dact += gmu W2   (rows x Hin)
      for (int i = tid; i < R * Hin; i += 256) {
        const int r = i / Hin, j = i - r * Hin;
        float acc = sdact[i];
        for (int dd = 0; dd < dn; ++dd) acc = fmaf(sgmu[r * DC + dd], sW2[dd * HinP + j], acc);
        sdact[i] = acc;
      }
      // B4: last-layer weight grads for this chunk (fixed owner thread per element)
      for (int i = tid; i < dn * Hin; i += 256) {
        const int dd = i / Hin, j = i - dd * Hin;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(sgmu[r * DC + dd], sact[r * Hin + j], acc);
        gW2[(int64_t)(dc0 + dd) * Hin + j] += acc;
      }
      for (int i = tid; i < dn; i += 256) {
        float accb = 0.f, accs = 0.f;
        for (int r = 0; r < R; ++r) { accb += sgmu[r * DC + i]; accs += sres[r * DC + i]; }
        gb2[dc0 + i] += accb;
        gls[dc0 + i] += accs * sCoef[i];
      }
      __syncthreads();
    }
    ll_acc += (double)ll_tile;
    // C1: back through the activation / hidden layer
    if (typ == OIVAE_GEN_MLP) {
      for (int i = tid; i < R * M; i += 256) { const float t = sact[i]; sdact[i] = sdact[i] * (1.f - t * t); }
      __syncthreads();
      for (int i = tid; i < R * H; i += 256) {
        const int r = i / H, j = i - r * H;
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc = fmaf(sdact[r * M + m], sW1[m * HP + j], acc);
        sdh[i] = acc;
      }
      // C2: hidden-layer weight grads
      for (int i = tid; i < M * H; i += 256) {
        const int m = i / H, j = i - m * H;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(sdact[r * M + m], sh[r * H + j], acc);
        gW1[i] += acc;
      }
      for (int i = tid; i < M; i += 256) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += sdact[r * M + i];
        gb1[i] += acc;
      }
    } else if (typ == OIVAE_GEN_TANH_LINEAR) {
      for (int i = tid; i < R * H; i += 256) { const float t = sact[i]; sdact[i] = sdact[i] * (1.f - t * t); }
    }
    __syncthreads();
    // C3: dWs[g] += z^T dh
    for (int i = tid; i < K * H; i += 256) {
      const int k = i / H, j = i - k * H;
      float acc = 0.f;
      for (int r = 0; r < R; ++r) acc = fmaf(sz[r * K + k], sdh[r * H + j], acc);
      gWs[i] += acc;
    }
    // C4: dz += dh Ws[g]^T   (summed over groups: float atomics)
    for (int i = tid; i < R * K; i += 256) {
      const int r = i / K, k = i - r * K;
      const int64_t n = n0 + r;
      if (n < a.N) {
        float acc = 0.f;
        for (int j = 0; j < H; ++j) acc = fmaf(sdh[r * H + j], sWs[k * HP + j], acc);
        atomicAdd(a.dz + n * K + k, acc);
      }
    }
  }
  const double tot = block_sum_double(ll_acc, sred);
  if (tid == 0) a.ll_part[(int64_t)chunk * gridDim.x + g] = tot;
}

// ------------------------------------------------------------------------------
// Single-layer generators, mu_net = Linear(H, D_g) or Tanh -> Linear(H, D_g) with a skinny H (the reference's own
// scripts: H = 1 for bars and MEG, bars_data_oivae.py:96-97, meg_oi_vae.py:178; H = 8 for mocap,
// mo_mo_mocap_oivae.py:203-207), any K, ragged D_g.  With H this small there is no matrix to multiply: per group
// the decoder is a rank-H update of a [rows, D_g] slab of X, HBM-bound on reading X once.  dec_generic spends its
// time in phases that only R*H or D_g*H threads can share (and in 64-bit index arithmetic per element); here every
// phase keeps all 256 threads busy by giving a thread a COLUMN when the sum runs over rows and a ROW when it runs
// over columns, with one [64][65] LDS tile of dL/dmu between the two:
//   phase 1, thread = (column dd, row quarter): x loaded coalesced, mu, log-likelihood, gm = dL/dmu -> tile;
//            db2, dlog_sigma and dW2[d][:] are sums over rows = private registers of the column's thread;
//   phase 2, thread = (row r, column quarter): dact[r][:] += gm[r][d] W2[d][:], private over all column chunks;
//   then dh, dWs[g] = z^T dh (LDS accumulators over the workgroup's row tiles), dz += dh Ws[g]^T (float atomics).
// Weight gradients are summed in a fixed order and written once per (group, row chunk): reproducible, as in dec_generic.
// ------------------------------------------------------------------------------
template <int H, bool TANH>
__global__ __launch_bounds__(256) void dec_single(DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ double sred[256];
  __shared__ int64_t sxrow[64];
  constexpr int CP = H + 4;   // column parameters: W2[d][0..H), b2, 1/sigma, log sigma, d loss/d log_sigma coefficient
  constexpr int CA = H + 2;   // column accumulators: dW2[d][0..H), db2, sum of 1 - q^2
  const int tid = threadIdx.x;
  const int g = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
  const int K = a.K, KP = K | 1;
  const int off = a.group_off[g];
  const int Dg = a.group_off[g + 1] - off;
  float* p = reinterpret_cast<float*>(smem_raw);
  float* sWs = p;      p += K * H;
  float* accWs = p;    p += K * H;
  float* sz = p;       p += 64 * KP;
  float* sact = p;     p += 64 * H;
  float* sdh = p;      p += 64 * H;
  float* sdap = p;     p += 4 * 64 * H;
  float* tileg = p;    p += 64 * 65;
  float* colpart = p;  p += 4 * 64 * CA;
  float* colpar = p;   p += a.DC * CP;     // DC = the largest group's width here
  float* colacc = p;

  const float* Wsg = a.Ws + (int64_t)g * K * H;
  for (int i = tid; i < K * H; i += 256) { sWs[i] = Wsg[i]; accWs[i] = 0.f; }
  for (int d = tid; d < Dg; d += 256) {
    const float e = expf(a.ls[off + d]);
    const float sg = e + a.sig_off;
#pragma unroll
    for (int j = 0; j < H; ++j) colpar[d * CP + j] = a.W2[(int64_t)(off + d) * H + j];
    colpar[d * CP + H] = a.b2[off + d];
    colpar[d * CP + H + 1] = 1.0f / sg;
    colpar[d * CP + H + 2] = logf(sg);
    colpar[d * CP + H + 3] = e * a.inv / sg;
#pragma unroll
    for (int v = 0; v < CA; ++v) colacc[d * CA + v] = 0.f;
  }
  const int dd = tid & 63, rq = tid >> 6;   // phase 1: column of the chunk, quarter of the rows
  const int r2 = tid & 63, cq = tid >> 6;   // phase 2: row, quarter of the chunk's columns
  double ll_acc = 0.0;
  const int64_t ntiles = (a.N + 63) / 64;
  for (int64_t tile = chunk; tile < ntiles; tile += nchunks) {
    const int64_t n0 = tile * 64;
    __syncthreads();
    for (int i = tid; i < 64 * K; i += 256) {
      const int r = i / K, k = i - r * K;
      const int64_t n = n0 + r;
      sz[r * KP + k] = (n < a.N) ? a.z[n * K + k] : 0.f;
    }
    if (tid < 64) {
      const int64_t n = n0 + tid;
      sxrow[tid] = ((n < a.N ? n : a.N - 1) % a.B) * (int64_t)a.D + off;
    }
    __syncthreads();
    for (int i = tid; i < 64 * H; i += 256) {
      const int r = i / H, j = i % H;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(sz[r * KP + k], sWs[k * H + j], acc);
      sact[i] = TANH ? tanhf(acc) : acc;
    }
    float da[H];
#pragma unroll
    for (int j = 0; j < H; ++j) da[j] = 0.f;
    float ll_tile = 0.f;
    __syncthreads();
    // x of this thread's column for its 16 rows, requested one column chunk ahead: the loads of chunk c0 + 64 fly
    // while phase 2 of chunk c0 and the barriers run (issued one by one in front of their use, each cost an HBM round trip)
    float xv[16];
    auto load_x = [&](int c0) {
      const int d = c0 + dd;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = 16 * rq + rr;
        xv[rr] = (d < Dg && n0 + row < a.N) ? a.X[sxrow[row] + d] : 0.f;
      }
    };
    load_x(0);
    for (int c0 = 0; c0 < Dg; c0 += 64) {
      // ---- phase 1: this thread's column over its 16 rows ----
      const int d = c0 + dd;
      const bool dvalid = d < Dg;
      float w2r[H], accw[H], b2r = 0.f, isig = 0.f, lsig = 0.f, accb = 0.f, accs = 0.f;
#pragma unroll
      for (int j = 0; j < H; ++j) { w2r[j] = dvalid ? colpar[d * CP + j] : 0.f; accw[j] = 0.f; }
      if (dvalid) { b2r = colpar[d * CP + H]; isig = colpar[d * CP + H + 1]; lsig = colpar[d * CP + H + 2]; }
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = 16 * rq + rr;
        const bool live = dvalid && (n0 + row < a.N);
        float gm = 0.f;
        if (live) {
          const float x = xv[rr];
          float mu = b2r;
#pragma unroll
          for (int j = 0; j < H; ++j) mu = fmaf(sact[row * H + j], w2r[j], mu);
          const float q = (x - mu) * isig;
          ll_tile += -0.5f * kLog2Pi - lsig - 0.5f * q * q;
          gm = -(q * isig) * a.inv;
          accb += gm;
          accs += 1.f - q * q;
#pragma unroll
          for (int j = 0; j < H; ++j) accw[j] = fmaf(gm, sact[row * H + j], accw[j]);
        }
        tileg[row * 65 + dd] = gm;
      }
      if (c0 + 64 < Dg) load_x(c0 + 64);
#pragma unroll
      for (int j = 0; j < H; ++j) colpart[(rq * 64 + dd) * CA + j] = accw[j];
      colpart[(rq * 64 + dd) * CA + H] = accb;
      colpart[(rq * 64 + dd) * CA + H + 1] = accs;
      __syncthreads();
      // ---- phase 2: this thread's row over its 16 columns; the four row quarters' column sums folded in order ----
#pragma unroll 4
      for (int q = 0; q < 16; ++q) {
        const int c = 16 * cq + q;
        if (c0 + c < Dg) {
          const float gmv = tileg[r2 * 65 + c];
#pragma unroll
          for (int j = 0; j < H; ++j) da[j] = fmaf(gmv, colpar[(c0 + c) * CP + j], da[j]);
        }
      }
      for (int i = tid; i < 64 * CA; i += 256) {
        const int c = i / CA, v = i - c * CA;
        if (c0 + c < Dg)
          colacc[(c0 + c) * CA + v] += ((colpart[c * CA + v] + colpart[(64 + c) * CA + v]) + colpart[(128 + c) * CA + v]) +
                                       colpart[(192 + c) * CA + v];
      }
      __syncthreads();
    }
    ll_acc += (double)ll_tile;
    // ---- back through the activation; dWs and dz ----
#pragma unroll
    for (int j = 0; j < H; ++j) sdap[(cq * 64 + r2) * H + j] = da[j];
    __syncthreads();
    for (int i = tid; i < 64 * H; i += 256) {
      const int r = i / H, j = i % H;
      float v = ((sdap[(0 * 64 + r) * H + j] + sdap[(1 * 64 + r) * H + j]) + sdap[(2 * 64 + r) * H + j]) + sdap[(3 * 64 + r) * H + j];
      if (TANH) { const float t = sact[i]; v *= 1.f - t * t; }
      sdh[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < K * H; i += 256) {
      const int k = i / H, j = i % H;
      float acc = 0.f;
      for (int r = 0; r < 64; ++r) acc = fmaf(sz[r * KP + k], sdh[r * H + j], acc);
      accWs[i] += acc;
    }
    for (int i = tid; i < 64 * K; i += 256) {
      const int r = i / K, k = i - r * K;
      const int64_t n = n0 + r;
      if (n < a.N) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < H; ++j) acc = fmaf(sdh[r * H + j], sWs[k * H + j], acc);
        atomicAdd(a.dz + n * K + k, acc);
      }
    }
  }
  __syncthreads();
  const int64_t cs = a.chunk_stride * chunk;
  for (int d = tid; d < Dg; d += 256) {
#pragma unroll
    for (int j = 0; j < H; ++j) a.gW2[cs + (int64_t)(off + d) * H + j] = colacc[d * CA + j];
    a.gb2[cs + off + d] = colacc[d * CA + H];
    a.gls[cs + off + d] = colacc[d * CA + H + 1] * colpar[d * CP + H + 3];
  }
  for (int i = tid; i < K * H; i += 256) a.gWs[cs + (int64_t)g * K * H + i] = accWs[i];
  const double tot = block_sum_double(ll_acc, sred);
  if (tid == 0) a.ll_part[(int64_t)chunk * gridDim.x + g] = tot;
}

bool single_decoder_applies(const oivae_shape_t& s) {
  if (test_hooks().force_generic) return false;
  if (s.gen_type == OIVAE_GEN_MLP) return false;
  if (!(s.H == 1 || s.H == 2 || s.H == 4 || s.H == 8) || s.K > 128) return false;
  return single_decoder_lds_bytes(s) <= 120 * 1024;
}

int64_t single_decoder_lds_bytes(const oivae_shape_t& s) {
  const int64_t H = s.H, KP = s.K | 1;
  return 4 * (2 * s.K * H + 64 * KP + 2 * 64 * H + 4 * 64 * H + 64 * 65 + 4 * 64 * (H + 2) + (int64_t)s.max_group_dim * (2 * H + 6));
}

// grads[i] = sum_c slab[c][i], c in index order; the six decoder parameter segments in one launch (blockIdx.y)
__global__ __launch_bounds__(256) void slab_reduce(const float* __restrict__ slab, int64_t P, int chunks, SlabSegs sg) {
  const int q = blockIdx.y;
  float* __restrict__ dst = sg.dst[q];
  const int64_t seg_off = sg.off[q], seg_n = sg.n[q];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < seg_n; i += stride) {   // eight chunks requested at a time (one by one, each cost a trip to memory); same order of addition
    float acc = 0.f;
    for (int c0 = 0; c0 < chunks; c0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (c0 + u < chunks) ? slab[(int64_t)(c0 + u) * P + seg_off + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    dst[i] = acc;
  }
}

// dst[i] = sum_c parts[c * P + i], i < n, c in index order (one segment)
void launch_partial_sum(const float* parts, int64_t P, int64_t n, int chunks, float* dst, hipStream_t st) {
  if (n == 0) return;
  SlabSegs sg = {};
  sg.dst[0] = dst; sg.off[0] = 0; sg.n[0] = n;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce, dim3((unsigned)blocks, 1), dim3(256), 0, st, parts, P, chunks, sg);
}

// two consecutive segments of n floats each (the encoder backward's two heads): dst0[i] = sum_c parts[c * P + i],
// dst1[i] = sum_c parts[c * P + n + i], one launch
void launch_partial_sum2(const float* parts, int64_t P, int64_t n, int chunks, float* dst0, float* dst1, hipStream_t st) {
  if (n == 0) return;
  SlabSegs sg = {};
  sg.dst[0] = dst0; sg.off[0] = 0; sg.n[0] = n;
  sg.dst[1] = dst1; sg.off[1] = n; sg.n[1] = n;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce, dim3((unsigned)blocks, 2), dim3(256), 0, st, parts, P, chunks, sg);
}

int launch_slab_reduce(const float* slab, const DecLayout& L, int chunks, const oivae_tensors_t* grads, hipStream_t st) {
  SlabSegs sg;
  float* dst[6] = {grads->gen_W1, grads->gen_b1, grads->gen_W2, grads->gen_b2, grads->log_sigma, grads->Ws};
  const int64_t off[7] = {L.oW1, L.ob1, L.oW2, L.ob2, L.oLs, L.oWs, L.total};
  int64_t most = 0;
  for (int q = 0; q < 6; ++q) {
    sg.dst[q] = dst[q]; sg.off[q] = off[q]; sg.n[q] = off[q + 1] - off[q];
    if (sg.n[q] != 0 && !dst[q]) return fail(OIVAE_E_ARG, "decoder: null gradient tensor %d", q);
    if (sg.n[q] > most) most = sg.n[q];
  }
  int64_t blocks = (most + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(slab_reduce, dim3((unsigned)blocks, 6), dim3(256), 0, st, slab, L.total, chunks, sg);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

// ------------------------------------------------------------------------------
// encoder backward (lib/oivae.py:75, encoder half)
// ------------------------------------------------------------------------------
// dz element (n, k) lives at dz[n*dz_sn + k*dz_sk]: [N,K] from the generic decoder,
// [K,N] from the MFMA decoder.
// One block owns a contiguous slab of rows: it writes dpre [B, 2K] for them and the slab's column sums
// (the encoder bias gradients' partials, summed over blocks in index order by enc_bwd_bias_final).
__global__ __launch_bounds__(256) void enc_bwd_pre(int64_t B, int K, int S, const float* __restrict__ dz,
                                                   int64_t dz_sn, int64_t dz_sk,
                                                   const float* __restrict__ eps, float c,
                                                   const float* __restrict__ prior_mu,
                                                   const float* __restrict__ prior_sigma, int prior_stride,
                                                   const float* __restrict__ mu_q,
                                                   const float* __restrict__ sig_q,
                                                   const float* __restrict__ ea, float* __restrict__ dpre,
                                                   double* __restrict__ bias_part, uint32_t* __restrict__ max_part) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* col = reinterpret_cast<double*>(smem_raw);  // [2K] column sums of this block
  const int tid = threadIdx.x;
  // max_part (the fp16 backward GEMM follows): this block's max|dpre[:, j]| per column, as bit patterns -- the scales
  // of the dpre^T image are per column (x3_encoder.hip h2_split_cols); found here, where dpre is made, instead of by a
  // pass of its own over dpre (20 us at 8192 rows)
  uint32_t* cmax = reinterpret_cast<uint32_t*>(col + 2 * K + 512) + ((dz_sn == 1 && S == 1) ? 65 * K : 0);   // [2K]
  for (int j = tid; j < 2 * K; j += 256) { col[j] = 0.0; if (max_part) cmax[j] = 0u; }
  __syncthreads();
  const int64_t rpb = (B + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = (r0 + rpb < B) ? r0 + rpb : B;
  const float invB = 1.0f / (float)B;
  const bool fixed_col = (256 % K) == 0;  // then thread t always meets column t % K: sums stay in registers
  double smu = 0.0, sda = 0.0;
  float mmu = 0.f, mda = 0.f;
  auto element = [&](int64_t i, int64_t b, int k, float sdz, float sdze) {
    const float mp = prior_mu[b * prior_stride + k], sp = prior_sigma[b * prior_stride + k];
    const float sq = sig_q[i];
    const float dmu = sdz + ((mu_q[i] - mp) / (sp * sp)) * invB;
    const float dsig = sdze + (sq / (sp * sp) - 1.0f / sq) * invB;
    const float da = dsig * c * ea[i];
    dpre[b * 2 * K + k] = dmu;
    dpre[b * 2 * K + K + k] = da;
    if (fixed_col) { smu += (double)dmu; sda += (double)da; mmu = fmaxf(mmu, fabsf(dmu)); mda = fmaxf(mda, fabsf(da)); }
    else {
      atomicAdd(&col[k], (double)dmu); atomicAdd(&col[K + k], (double)da);  // LDS; odd K only (order-dependent in the last bits)
      if (max_part) { atomicMax(&cmax[k], __float_as_uint(fabsf(dmu))); atomicMax(&cmax[K + k], __float_as_uint(fabsf(da))); }
    }
  };
  if (dz_sn == 1 && S == 1) {
    // dz arrives transposed ([K, N], the matrix-core decoders): consecutive threads of the element loop below meet
    // consecutive k, i.e. addresses N floats apart.  Tiles of 64 rows are therefore read along n (256-byte runs) into
    // LDS first (round 2: 84 -> 25 us at 65536 rows).
    float* sT = reinterpret_cast<float*>(col + 2 * K + 512);   // [K][65]
    for (int64_t t0 = r0; t0 < r1; t0 += 64) {
      const int nr = (int)((r1 - t0 < 64) ? r1 - t0 : 64);
      __syncthreads();   // the tile before this one has been consumed
      for (int k = tid >> 6; k < K; k += 4) {
        const int r = tid & 63;
        sT[k * 65 + r] = (r < nr) ? dz[(t0 + r) + k * dz_sk] : 0.f;
      }
      __syncthreads();
      for (int64_t i = t0 * K + tid; i < (t0 + nr) * K; i += 256) {
        const int64_t b = i / K;
        const int k = (int)(i - b * K);
        const float d = sT[k * 65 + (int)(b - t0)];
        element(i, b, k, d, d * eps[i]);
      }
    }
  } else {
    for (int64_t i = r0 * K + tid; i < r1 * K; i += 256) {
      const int64_t b = i / K;
      const int k = (int)(i - b * K);
      float sdz = 0.f, sdze = 0.f;
      for (int s = 0; s < S; ++s) {
        const int64_t n = (int64_t)s * B + b;
        const float d = dz[n * dz_sn + k * dz_sk];
        sdz += d;
        sdze = fmaf(d, eps[n * K + k], sdze);
      }
      element(i, b, k, sdz, sdze);
    }
  }
  if (fixed_col) {
    // threads t, t+K, t+2K, ... hold the same column: fold them in a fixed order
    double* sm = col + 2 * K;  // [2][256]
    sm[tid] = smu;
    sm[256 + tid] = sda;
    __syncthreads();
    if (tid < K) {
      double a = 0.0, d = 0.0;
      for (int j = tid; j < 256; j += K) { a += sm[j]; d += sm[256 + j]; }
      col[tid] = a;
      col[K + tid] = d;
    }
    if (max_part) {   // a maximum does not care for the order (NaN: fmaxf drops it here as absmax_cols_f32 did)
      atomicMax(&cmax[tid % K], __float_as_uint(mmu));
      atomicMax(&cmax[K + tid % K], __float_as_uint(mda));
    }
  }
  __syncthreads();
  for (int j = tid; j < 2 * K; j += 256) {
    bias_part[(int64_t)blockIdx.x * 2 * K + j] = col[j];
    if (max_part) max_part[(int64_t)blockIdx.x * 2 * K + j] = cmax[j];
  }
}

// one block per column j: the per-block partials are folded by a fixed-shape tree (deterministic)
__global__ __launch_bounds__(256) void enc_bwd_bias_final(const double* __restrict__ bias_part, int nblk, int K,
                                                          float* __restrict__ dbmu, float* __restrict__ dbsig,
                                                          uint32_t* __restrict__ rng, uint32_t* __restrict__ colmax,
                                                          const uint32_t* __restrict__ max_part, int colmax_n) {
  __shared__ double sm[256];
  __shared__ uint32_t mx[256];
  const int j = blockIdx.x;
  // the ranges the fp16 backward GEMM's helpers accumulate into with atomicMax start at zero (x3_encoder.hip:
  // [2] max|dpre|, [8] max|x| the kernel reads, [9] its redo flag); done here, one kernel ahead of them, instead of
  // two memsets of their own
  if (rng != nullptr && j == 0 && threadIdx.x == 0) { rng[2] = 0u; rng[8] = 0u; rng[9] = 0u; }
  // ... and column j's maximum of |dpre| over the blocks' partial maxima (bit patterns of non-negative floats order
  // like the floats)
  double acc = 0.0;
  uint32_t m = 0u;
  for (int b = threadIdx.x; b < nblk; b += 256) {
    acc += bias_part[(int64_t)b * 2 * K + j];
    if (max_part != nullptr) { const uint32_t v = max_part[(int64_t)b * 2 * K + j]; m = v > m ? v : m; }
  }
  sm[threadIdx.x] = acc;
  mx[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sm[threadIdx.x] += sm[threadIdx.x + s];
      mx[threadIdx.x] = mx[threadIdx.x + s] > mx[threadIdx.x] ? mx[threadIdx.x + s] : mx[threadIdx.x];
    }
    __syncthreads();
  }
  if (colmax != nullptr && threadIdx.x == 0) colmax[j] = mx[0];
  // (the image's padding rows 2K .. colmax_n - 1 hold zeros whatever their scale: give them a defined one)
  if (colmax != nullptr && j == 0)
    for (int i = 2 * K + threadIdx.x; i < colmax_n; i += 256) colmax[i] = 0u;
  if (threadIdx.x == 0) {
    if (j < K) dbmu[j] = (float)sm[0]; else dbsig[j - K] = (float)sm[0];
  }
}

// dW[j][d] = sum_b dpre[b][j] X[b][d]; block = 32 j x 64 d, loops over all b in order
__global__ __launch_bounds__(256) void enc_bwd_wgrad_generic(const float* __restrict__ X,
                                                             const float* __restrict__ dpre, int64_t B,
                                                             int D, int K, float* __restrict__ dWmu,
                                                             float* __restrict__ dWsig) {
  __shared__ float sX[64][65];
  __shared__ float sP[64][33];
  const int J = 2 * K;
  const int tid = threadIdx.x;
  const int dd = tid & 63, jj = tid >> 6;
  const int d0 = blockIdx.x * 64, j0 = blockIdx.y * 32;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.f;
  for (int64_t bb0 = 0; bb0 < B; bb0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
      const int rr = i >> 6, cc = i & 63;
      const int64_t b = bb0 + rr;
      sX[rr][cc] = (b < B && d0 + cc < D) ? X[b * D + d0 + cc] : 0.f;
    }
    for (int i = tid; i < 64 * 32; i += 256) {
      const int rr = i >> 5, cc = i & 31;
      const int64_t b = bb0 + rr;
      sP[rr][cc] = (b < B && j0 + cc < J) ? dpre[b * J + j0 + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < 64; ++rr) {
      const float xv = sX[rr][dd];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = fmaf(sP[rr][jj + 4 * q], xv, acc[q]);
    }
  }
  if (d0 + dd < D) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = j0 + jj + 4 * q;
      if (j < K) dWmu[(int64_t)j * D + d0 + dd] = acc[q];
      else if (j < J) dWsig[(int64_t)(j - K) * D + d0 + dd] = acc[q];
    }
  }
}

// ------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------
static int64_t dec_lds_floats(const oivae_shape_t& s, int R, int DC) {
  const int Hin = (s.gen_type == OIVAE_GEN_MLP) ? s.M : s.H;
  const int HP = s.H | 1, HinP = Hin | 1;
  int64_t n = (int64_t)s.K * HP;
  if (s.gen_type == OIVAE_GEN_MLP) n += (int64_t)s.M * HP + s.M;
  n += (int64_t)DC * HinP + 4 * DC;
  n += (int64_t)R * s.K + (int64_t)R * s.H;
  if (s.gen_type != OIVAE_GEN_LINEAR) n += (int64_t)R * Hin;
  n += (int64_t)R * Hin;
  if (s.gen_type == OIVAE_GEN_MLP) n += (int64_t)R * s.H;
  n += 2 * (int64_t)R * DC;
  return n;
}

bool mfma_decoder_applies(const oivae_shape_t& s);          // mfma_decoder.hip
bool mfma_decoder_pays(const oivae_shape_t& s, int64_t N);  // mfma_decoder.hip
int mfma_decoder_chunks(const oivae_shape_t& s, int64_t N);  // mfma_decoder.hip
bool mfma_encoder_applies(const oivae_shape_t& s);           // mfma_encoder.hip
int launch_encoder_fwd_mfma(const oivae_shape_t& s, const oivae_tensors_t* params, const float* X, int64_t B,
                            float* pre, hipStream_t st);
int launch_encoder_bwd_mfma(const oivae_shape_t& s, const oivae_tensors_t* grads, const float* X,
                            const float* dpre, int64_t B, float* part, hipStream_t st);

DecPlan plan_decoder(const oivae_shape_t& s, int64_t N) {
  DecPlan pl;
  pl.mfma = mfma_decoder_applies(s) && mfma_decoder_pays(s, N);
  pl.flat = !pl.mfma && flat_decoder_applies(s, N);
  pl.single = !pl.mfma && !pl.flat && single_decoder_applies(s);
  pl.DC = s.max_group_dim < 64 ? s.max_group_dim : 64;
  pl.R = 0;
  pl.lds_bytes = 0;
  const int cand[3] = {64, 32, 16};
  for (int i = 0; i < 3; ++i) {
    const int64_t bytes = dec_lds_floats(s, cand[i], pl.DC) * 4;
    if (bytes <= 150 * 1024) { pl.R = cand[i]; pl.lds_bytes = bytes; break; }
  }
  const int64_t R = pl.R ? pl.R : 16;
  const int64_t tiles = (N + R - 1) / R;
  int64_t c = (2048 + s.G - 1) / s.G;
  if (c > tiles) c = tiles;
  if (c < 1) c = 1;
  if ((int64_t)s.G * c > kMaxPartials) c = kMaxPartials / s.G > 0 ? kMaxPartials / s.G : 1;
  pl.chunks = (int32_t)c;
  if (pl.mfma) pl.chunks = mfma_decoder_chunks(s, N);
  if (pl.flat) pl.chunks = 1;
  if (pl.single) {   // 64-row tiles; several workgroups per CU (their loads of X overlap) and at least one tile each
    const int64_t t64 = (N + 63) / 64;
    // as many workgroups as the chip holds at once (four per CU by registers at H <= 2, three above: 1024 / 768), not
    // more: a second, partial round of workgroups costs a whole workgroup time (MEG-like decoder: 74 us at 1536
    // workgroups, 68 at 1024, 78 at 768 and at 2816)
    int64_t cc = ((s.H <= 2 ? 1024 : 768) + s.G - 1) / s.G;
    if (cc > t64) cc = t64;
    if (cc < 1) cc = 1;
    if ((int64_t)s.G * cc > kMaxPartials) cc = kMaxPartials / s.G > 0 ? kMaxPartials / s.G : 1;
    pl.chunks = (int32_t)cc;
  }
  return pl;
}

int launch_decoder_mfma(const oivae_shape_t& s, const oivae_tensors_t* params, const oivae_tensors_t* grads,
                        const float* X, int64_t B, int32_t S, const float* z, float gen_sig_off, float* dz,
                        Workspace& w, hipStream_t st);  // mfma_decoder.hip
int launch_decoder_flat(const oivae_shape_t& s, const oivae_tensors_t* params, const oivae_tensors_t* grads,
                        const float* X, int64_t B, int32_t S, const float* z, float gen_sig_off, float* dz,
                        Workspace& w, hipStream_t st);  // flat_decoder.hip

}  // namespace oivae

using namespace oivae;

extern "C" int oivae_encoder_fwd_f32(const oivae_shape_t* shape, const oivae_tensors_t* params,
                                     const float* X, int64_t B, int32_t S, const float* eps, float c,
                                     float sig_off, const float* prior_mu, const float* prior_sigma,
                                     int32_t prior_rows, float* mu_q, float* sig_q, float* ea, float* z,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_shape(shape)) return rc;
  if (!params || !X || !eps || !prior_mu || !prior_sigma || !mu_q || !sig_q || !ea || !z || !workspace)
    return fail(OIVAE_E_ARG, "oivae_encoder_fwd_f32: null pointer");
  if (B < 1 || S < 1) return fail(OIVAE_E_ARG, "oivae_encoder_fwd_f32: B and S must be >= 1");
  if (prior_rows != 1 && prior_rows != B)
    return fail(OIVAE_E_ARG, "oivae_encoder_fwd_f32: prior_rows must be 1 or B (got %d, B=%lld)", prior_rows, (long long)B);
  if (int rc = poll_error_word("oivae_encoder_fwd_f32", (hipStream_t)stream)) return rc;
  Workspace w;
  if (int rc = carve_workspace(*shape, B, S, workspace, workspace_bytes, &w)) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* pre = w.dpre;  // [B, 2K]; the backward pass reuses the buffer for dpre
  if (x3_encoder_applies(*shape)) {
    if (int rc = launch_encoder_fwd_x3(*shape, params, X, B, pre, w.x3_img, w.ranges, w.enc_splits, st)) return rc;
  } else if (mfma_encoder_applies(*shape)) {
    if (int rc = launch_encoder_fwd_mfma(*shape, params, X, B, pre, st)) return rc;
  } else {
    const int64_t blocks = (B + kEncRows - 1) / kEncRows;
    const size_t lds = sizeof(float) * (size_t)kEncRows * (kEncDC + 1);
    hipLaunchKernelGGL(enc_fwd_generic, dim3((unsigned)blocks), dim3(256), lds, st, X, B, shape->D, shape->K,
                       params->enc_Wmu, params->enc_Wsig, pre);
    OIVAE_LAUNCH_CHECK();
  }
  const int64_t eblocks = w.n_kl;  // every block writes its KL partial
  hipLaunchKernelGGL(enc_fwd_epilogue, dim3((unsigned)eblocks), dim3(256), 0, st, pre, w.enc_splits, B, shape->K,
                     params->enc_bmu, params->enc_bsig, S, eps, c, sig_off, prior_mu, prior_sigma,
                     prior_rows == 1 ? 0 : shape->K, mu_q, sig_q, ea, z, w.kl_part);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

extern "C" int oivae_decoder_fwd_bwd_f32(const oivae_shape_t* shape, const oivae_tensors_t* params,
                                         const oivae_tensors_t* grads, const float* X, int64_t B, int32_t S,
                                         const float* z, float gen_sig_off, float* dz, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (int rc = check_shape(shape)) return rc;
  if (!params || !grads || !X || !z || !dz || !workspace)
    return fail(OIVAE_E_ARG, "oivae_decoder_fwd_bwd_f32: null pointer");
  if (B < 1 || S < 1) return fail(OIVAE_E_ARG, "oivae_decoder_fwd_bwd_f32: B and S must be >= 1");
  const oivae_shape_t& s = *shape;
  if (int rc = poll_error_word("oivae_decoder_fwd_bwd_f32", (hipStream_t)stream)) return rc;
  Workspace w;
  if (int rc = carve_workspace(s, B, S, workspace, workspace_bytes, &w)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = B * (int64_t)S;
  const DecPlan pl = plan_decoder(s, N);
  if (pl.mfma) return launch_decoder_mfma(s, params, grads, X, B, S, z, gen_sig_off, dz, w, st);   // (zeroes dz itself)
  OIVAE_HIP_CHECK(hipMemsetAsync(dz, 0, sizeof(float) * N * s.K, st));
  if (pl.flat) return launch_decoder_flat(s, params, grads, X, B, S, z, gen_sig_off, dz, w, st);
  if (pl.R == 0 && !pl.single)
    return fail(OIVAE_E_SHAPE, "oivae_decoder_fwd_bwd_f32: group weights do not fit LDS (K=%d H=%d M=%d)", s.K, s.H, s.M);
  const DecLayout L = dec_layout(s);
  const bool mlp = s.gen_type == OIVAE_GEN_MLP;
  DecArgs a;
  a.D = s.D; a.K = s.K; a.G = s.G; a.H = s.H; a.M = s.M; a.Hin = mlp ? s.M : s.H; a.gen_type = s.gen_type;
  a.group_off = s.group_off;
  a.W1 = params->gen_W1; a.b1 = params->gen_b1; a.W2 = params->gen_W2; a.b2 = params->gen_b2;
  a.ls = params->log_sigma; a.Ws = params->Ws;
  if (pl.chunks > 1) {
    if (!pl.single)   // dec_generic accumulates into its destinations; dec_single writes each exactly once
      OIVAE_HIP_CHECK(hipMemsetAsync(w.slab, 0, sizeof(float) * L.total * pl.chunks, st));
    a.gW1 = w.slab + L.oW1; a.gb1 = w.slab + L.ob1; a.gW2 = w.slab + L.oW2; a.gb2 = w.slab + L.ob2;
    a.gls = w.slab + L.oLs; a.gWs = w.slab + L.oWs;
    a.chunk_stride = L.total;
  } else {
    if (mlp) {
      OIVAE_HIP_CHECK(hipMemsetAsync(grads->gen_W1, 0, sizeof(float) * (L.ob1 - L.oW1), st));
      OIVAE_HIP_CHECK(hipMemsetAsync(grads->gen_b1, 0, sizeof(float) * (L.oW2 - L.ob1), st));
    }
    if (!pl.single) {
      OIVAE_HIP_CHECK(hipMemsetAsync(grads->gen_W2, 0, sizeof(float) * (L.ob2 - L.oW2), st));
      OIVAE_HIP_CHECK(hipMemsetAsync(grads->gen_b2, 0, sizeof(float) * (L.oLs - L.ob2), st));
      OIVAE_HIP_CHECK(hipMemsetAsync(grads->log_sigma, 0, sizeof(float) * (L.oWs - L.oLs), st));
      OIVAE_HIP_CHECK(hipMemsetAsync(grads->Ws, 0, sizeof(float) * (L.total - L.oWs), st));
    }
    a.gW1 = grads->gen_W1; a.gb1 = grads->gen_b1; a.gW2 = grads->gen_W2; a.gb2 = grads->gen_b2;
    a.gls = grads->log_sigma; a.gWs = grads->Ws;
    a.chunk_stride = 0;
  }
  a.X = X; a.z = z; a.dz = dz; a.B = B; a.N = N;
  a.inv = 1.0f / ((float)S * (float)B);
  a.sig_off = gen_sig_off;
  a.DC = pl.DC;
  a.ll_part = w.ll_part;
  const dim3 grid((unsigned)s.G, (unsigned)pl.chunks);
  if (pl.single) {
    a.DC = s.max_group_dim;
    const size_t lds = (size_t)single_decoder_lds_bytes(s);
#define OIVAE_SINGLE_LAUNCH(HH, TT)                                                                       \
  {                                                                                                      \
    OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_single<HH, TT>),               \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
    hipLaunchKernelGGL((dec_single<HH, TT>), grid, dim3(256), lds, st, a);                                \
  }
    const bool th = s.gen_type == OIVAE_GEN_TANH_LINEAR;
    if (s.H == 1) { if (th) OIVAE_SINGLE_LAUNCH(1, true) else OIVAE_SINGLE_LAUNCH(1, false) }
    else if (s.H == 2) { if (th) OIVAE_SINGLE_LAUNCH(2, true) else OIVAE_SINGLE_LAUNCH(2, false) }
    else if (s.H == 4) { if (th) OIVAE_SINGLE_LAUNCH(4, true) else OIVAE_SINGLE_LAUNCH(4, false) }
    else { if (th) OIVAE_SINGLE_LAUNCH(8, true) else OIVAE_SINGLE_LAUNCH(8, false) }
#undef OIVAE_SINGLE_LAUNCH
  } else {
#define OIVAE_DEC_LAUNCH(RR)                                                                             \
  {                                                                                                      \
    OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_generic<RR>),                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes)); \
    hipLaunchKernelGGL((dec_generic<RR>), grid, dim3(256), (size_t)pl.lds_bytes, st, a);                  \
  }
  if (pl.R == 64) OIVAE_DEC_LAUNCH(64)
  else if (pl.R == 32) OIVAE_DEC_LAUNCH(32)
  else OIVAE_DEC_LAUNCH(16)
#undef OIVAE_DEC_LAUNCH
  }
  OIVAE_LAUNCH_CHECK();
  if (pl.chunks > 1)
    if (int rc = launch_slab_reduce(w.slab, L, pl.chunks, grads, st)) return rc;
  return OIVAE_OK;
}

extern "C" int oivae_dz_layout(const oivae_shape_t* shape, int64_t B, int32_t S) {
  if (check_shape(shape) != OIVAE_OK || B < 1 || S < 1) return -1;
  return plan_decoder(*shape, B * (int64_t)S).mfma ? 1 : 0;
}

extern "C" int oivae_encoder_bwd_f32(const oivae_shape_t* shape, const oivae_tensors_t* grads, const float* X,
                                     int64_t B, int32_t S, const float* eps, float c, const float* prior_mu,
                                     const float* prior_sigma, int32_t prior_rows, const float* mu_q,
                                     const float* sig_q, const float* ea, const float* dz, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  if (int rc = check_shape(shape)) return rc;
  if (!grads || !X || !eps || !prior_mu || !prior_sigma || !mu_q || !sig_q || !ea || !dz || !workspace)
    return fail(OIVAE_E_ARG, "oivae_encoder_bwd_f32: null pointer");
  if (B < 1 || S < 1) return fail(OIVAE_E_ARG, "oivae_encoder_bwd_f32: B and S must be >= 1");
  if (prior_rows != 1 && prior_rows != B) return fail(OIVAE_E_ARG, "oivae_encoder_bwd_f32: prior_rows must be 1 or B");
  const oivae_shape_t& s = *shape;
  if (int rc = poll_error_word("oivae_encoder_bwd_f32", (hipStream_t)stream)) return rc;
  Workspace w;
  if (int rc = carve_workspace(s, B, S, workspace, workspace_bytes, &w)) return rc;
  return encoder_bwd_run(s, grads, X, B, S, eps, c, prior_mu, prior_sigma, prior_rows, mu_q, sig_q, ea, dz, w.dpre,
                         w.bias_part, w.n_bias, w.x3_img, w.ranges, w.enc_bwd_part, (hipStream_t)stream);
}

int oivae::encoder_bwd_run(const oivae_shape_t& s, const oivae_tensors_t* grads, const float* X, int64_t B, int32_t S,
                           const float* eps, float c, const float* prior_mu, const float* prior_sigma, int32_t prior_rows,
                           const float* mu_q, const float* sig_q, const float* ea, const float* dz, float* dpre,
                           double* bias_part, int n_bias, uint32_t* x3_img, uint32_t* rng, float* bwd_part, hipStream_t st) {
  const bool dz_transposed = plan_decoder(s, B * (int64_t)S).mfma;
  const int64_t dz_sn = dz_transposed ? 1 : s.K, dz_sk = dz_transposed ? B * (int64_t)S : 1;
  const bool h2 = x3_encoder_bwd_applies(s, B);
  const size_t pre_lds = sizeof(double) * (2 * (size_t)s.K + 512) + ((dz_sn == 1 && S == 1) ? sizeof(float) * 65 * (size_t)s.K : 0) +
                         sizeof(uint32_t) * 2 * (size_t)s.K;
  // the blocks' column maxima of |dpre| sit behind their column sums (enc_bias_part_bytes)
  uint32_t* max_part = h2 ? reinterpret_cast<uint32_t*>(bias_part + (int64_t)n_bias * 2 * s.K) : (uint32_t*)nullptr;
  hipLaunchKernelGGL(enc_bwd_pre, dim3((unsigned)n_bias), dim3(256), pre_lds, st, B, s.K, S, dz, dz_sn, dz_sk, eps, c,
                     prior_mu, prior_sigma, prior_rows == 1 ? 0 : s.K, mu_q, sig_q, ea, dpre, bias_part, max_part);
  OIVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(enc_bwd_bias_final, dim3((unsigned)(2 * s.K)), dim3(256), 0, st, bias_part, n_bias,
                     s.K, grads->enc_bmu, grads->enc_bsig, rng,
                     h2 ? h2_bwd_colmax(x3_img, 2 * s.K, B) : (uint32_t*)nullptr, (const uint32_t*)max_part,
                     (2 * s.K + 127) / 128 * 128);
  OIVAE_LAUNCH_CHECK();
  if (h2) {
    if (int rc = launch_encoder_bwd_x3(s, grads, X, dpre, B, x3_img, rng, st, /*colmax_ready=*/true)) return rc;
  } else if (mfma_encoder_applies(s)) {
    if (int rc = launch_encoder_bwd_mfma(s, grads, X, dpre, B, bwd_part, st)) return rc;
  } else {
    const dim3 grid((unsigned)((s.D + 63) / 64), (unsigned)((2 * s.K + 31) / 32));
    hipLaunchKernelGGL(enc_bwd_wgrad_generic, grid, dim3(256), 0, st, X, dpre, B, s.D, s.K, grads->enc_Wmu,
                       grads->enc_Wsig);
    OIVAE_LAUNCH_CHECK();
  }
  return OIVAE_OK;
}
