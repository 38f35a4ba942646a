// Two-layer generators up to 64 wide (K, H, M <= 64, one output width D_g <= 64 for all groups; not all 64) on the
// fp16-term decoder.
//
// dec_x3_64s (dec_x3_64s.h) is written for K = H = M = D_g = 64: its LDS plan, operand fragments and register
// allocation are fixed to that shape.  A generator whose latent, group-input or hidden width is narrower is the same
// computation with some operand rows equal to zero, and zeros are exact in every term of the split arithmetic
// (x3_split.h) -- so these shapes run through that kernel on ZERO-PADDED copies of the parameters and of z:
//   Ws[g] [K, H] -> [64, 64],  W1[g] [M, H] -> [64, 64],  b1[g] [M] -> [64],  W2 [D, M] -> [D, 64],  z [N, K] -> [N, 64]
// (a padded hidden unit has a1 = 0, t = tanh 0 = 0 and a zero column in W2; a padded latent has z = 0 and a zero row in
// Ws), the gradients come back in the padded layout and their valid blocks are copied out; dz [K, N] is the first K
// rows of the padded [64, N].  Cost: the padded shape's matrix work (K = 32: 1.33x the shape's own flops, M = 32: 2x)
// at 16x the instruction rate of the fp32 matrix instruction these shapes ran on before (dec_mfma<K, H, M, DG>, still
// their cross-check: OIVAE_DEC=v1), plus two passes over the parameters (50 MB at G = 1024) and z.
// D_g = 32: the kernel reads X in 64-column tiles per group, and X cannot be padded (17 GB).  Two ways:
// (a) the OUTPUTS padded: W2[g] [D_g, M] -> [64, 64] with zero rows (b2: zeros, log_sigma: zeros) and the kernel told
//     that groups own D_g columns of X (dec_x3_64s<true>: the padded outputs' lanes read the group's first columns
//     again, so every address stays inside the rows of X, and the padded outputs stay out of the log-likelihood sum).  A padded output d has mu = 0 and a
//     residual of its own, but its column of W2^T is zero: nothing of it reaches dt, dz or the live gradients; its own
//     gradient rows are dropped.  Any G, any K, H, M, any D_g that is a multiple of 4 (X is read 16 bytes per lane).
// (b) cheaper where it applies, PAIRING: with H = M = D_g = 32 two neighbouring groups g, g + 1 ARE one all-64 group -- group input
// [Ws_g | Ws_g+1] (K x 64), first layer blockdiag(W1_g, W1_g+1) (64 hidden units, the two groups' side by side), second
// layer blockdiag(W2_g, W2_g+1) onto the 64 contiguous columns of X the two groups own.  The zero blocks are exact, so
// every product of the virtual group restricted to its diagonal blocks is the pair's own; its weight gradients hold
// the pair's in their diagonal blocks (dWs side by side), the off-diagonal blocks are dropped.  G must be even (an odd
// G takes (a)); K = 32 is zero-padded as above.
#include "common.h"

namespace oivae {

bool mfma_decoder_applies(const oivae_shape_t& s);   // mfma_decoder.hip
int launch_decoder_mfma(const oivae_shape_t& s, const oivae_tensors_t* params, const oivae_tensors_t* grads,
                        const float* X, int64_t B, int32_t S, const float* z, float gen_sig_off, float* dz,
                        Workspace& w, hipStream_t st);   // mfma_decoder.hip

// two 32-wide groups as one all-64 group (see the head of this file)
static bool pair32(const oivae_shape_t& s) {
  return s.uniform_group_dim == 32 && s.H == 32 && s.M == 32 && (s.G % 2) == 0;
}

bool dec_pad64_applies(const oivae_shape_t& s) {
  if (!mfma_decoder_applies(s)) return false;   // two layers, K, H, M, D_g <= 64, D_g uniform and a multiple of 4
  return !(s.K == 64 && s.H == 64 && s.M == 64 && s.uniform_group_dim == 64);
}
// 32 outputs per group padded to 64 ((a) at the head of this file)
bool dec_pad64_xhalf(const oivae_shape_t& s) {
  return dec_pad64_applies(s) && s.uniform_group_dim != 64 && !pair32(s);
}
enum { kPadZero = 0, kPadPair = 1, kPadXHalf = 2 };
static int pad_mode(const oivae_shape_t& s) { return pair32(s) ? kPadPair : (s.uniform_group_dim != 64 ? kPadXHalf : kPadZero); }

static oivae_shape_t padded_shape(const oivae_shape_t& s) {
  oivae_shape_t p = s;
  if (pair32(s)) { p.G = s.G / 2; p.uniform_group_dim = 64; p.max_group_dim = 64; }
  else if (s.uniform_group_dim != 64) { p.D = 64 * s.G; p.uniform_group_dim = 64; p.max_group_dim = 64; }
  p.K = 64; p.H = 64; p.M = 64;
  return p;
}

// floats: padded parameters | padded gradients | padded slabs (chunks > 1) | z [N, 64] | dz [64, N]
int64_t dec_pad64_bytes(const oivae_shape_t& s, int64_t N, int chunks) {
  if (!dec_pad64_applies(s)) return 0;
  const DecLayout L = dec_layout(padded_shape(s));
  const int64_t al = 64;   // floats: every region 256-byte aligned
  auto up = [&](int64_t n) { return (n + al - 1) / al * al; };
  return (int64_t)sizeof(float) * (2 * up(L.total) + (chunks > 1 ? up(L.total * chunks) : 0) + 2 * up(N * 64));
}

struct PadArgs {
  const float *W1, *b1, *W2, *b2, *ls, *Ws, *z;      // the shape's own arrays
  float *pW1, *pb1, *pW2, *pb2, *pls, *pWs, *pz;     // padded ones
  int64_t nW1, nb1, nW2, nD, nWs, nz;                // element counts of the PADDED arrays (b2 and ls: nD each)
  int K, H, M;
  int mode;                                          // kPadPair: two 32-wide groups per virtual group; kPadXHalf: xg < 64 outputs per group padded to 64
  int xg;
};

__global__ __launch_bounds__(256) void pad64_kernel(PadArgs a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int64_t i = tid; i < a.nW1; i += stride) {   // [G][64][64] <- [G][M][H]; pairs: blockdiag(W1_2v, W1_2v+1)
    const int h = (int)(i & 63), m = (int)((i >> 6) & 63);
    const int64_t g = i >> 12;
    if (a.mode == kPadPair) a.pW1[i] = ((m >> 5) == (h >> 5)) ? a.W1[((2 * g + (m >> 5)) * 32 + (m & 31)) * 32 + (h & 31)] : 0.f;
    else a.pW1[i] = (m < a.M && h < a.H) ? a.W1[(g * a.M + m) * a.H + h] : 0.f;
  }
  for (int64_t i = tid; i < a.nb1; i += stride) {   // [G][64] <- [G][M]; pairs: the two groups' biases side by side
    const int m = (int)(i & 63);
    if (a.mode == kPadPair) a.pb1[i] = a.b1[i];                 // [Gv][64] = [G][32] element for element
    else a.pb1[i] = (m < a.M) ? a.b1[(i >> 6) * a.M + m] : 0.f;
  }
  for (int64_t i = tid; i < a.nW2; i += stride) {   // [D][64] <- [D][M]; pairs: row d keeps its group's half of the columns
    const int m = (int)(i & 63);
    const int64_t d = i >> 6;
    if (a.mode == kPadPair) a.pW2[i] = ((m >> 5) == (int)((d >> 5) & 1)) ? a.W2[d * 32 + (m & 31)] : 0.f;
    else if (a.mode == kPadXHalf) a.pW2[i] = ((int)(d & 63) < a.xg && m < a.M) ? a.W2[((d >> 6) * a.xg + (d & 63)) * a.M + m] : 0.f;
    else a.pW2[i] = (m < a.M) ? a.W2[d * a.M + m] : 0.f;
  }
  for (int64_t i = tid; i < a.nD; i += stride) {
    if (a.mode == kPadXHalf) {   // virtual output 64 g + r: the group's output r for r < xg
      const bool live = (int)(i & 63) < a.xg;
      const int64_t d = live ? (i >> 6) * a.xg + (i & 63) : 0;
      a.pb2[i] = live ? a.b2[d] : 0.f;
      a.pls[i] = live ? a.ls[d] : 0.f;
    } else {
      a.pb2[i] = a.b2[i];
      a.pls[i] = a.ls[i];
    }
  }
  for (int64_t i = tid; i < a.nWs; i += stride) {   // [G][64][64] <- [G][K][H]; pairs: [Ws_2v | Ws_2v+1]
    const int h = (int)(i & 63), k = (int)((i >> 6) & 63);
    const int64_t g = i >> 12;
    if (a.mode == kPadPair) a.pWs[i] = (k < a.K) ? a.Ws[((2 * g + (h >> 5)) * a.K + k) * 32 + (h & 31)] : 0.f;
    else a.pWs[i] = (k < a.K && h < a.H) ? a.Ws[(g * a.K + k) * a.H + h] : 0.f;
  }
  for (int64_t i = tid; i < a.nz; i += stride) {    // [N][64] <- [N][K]
    const int k = (int)(i & 63);
    a.pz[i] = (k < a.K) ? a.z[(i >> 6) * a.K + k] : 0.f;
  }
}

struct UnpadArgs {
  const float *pW1, *pb1, *pW2, *pb2, *pls, *pWs, *pdz;   // padded gradients, dz [64, N]
  float *W1, *b1, *W2, *b2, *ls, *Ws, *dz;                // the shape's own gradient arrays, dz [K, N]
  int64_t nW1, nb1, nW2, nD, nWs, ndz;                    // element counts of the shape's OWN arrays
  int K, H, M;
  int mode;
  int xg;
};

__global__ __launch_bounds__(256) void unpad64_kernel(UnpadArgs a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int64_t i = tid; i < a.nW1; i += stride) {   // [G][M][H] <- [G][64][64]; pairs: the diagonal blocks
    const int h = (int)(i % a.H);
    const int64_t r = i / a.H;
    const int m = (int)(r % a.M);
    const int64_t g = r / a.M;
    if (a.mode == kPadPair) a.W1[i] = a.pW1[((g >> 1) * 64 + 32 * (g & 1) + m) * 64 + 32 * (g & 1) + h];
    else a.W1[i] = a.pW1[(g * 64 + m) * 64 + h];
  }
  for (int64_t i = tid; i < a.nb1; i += stride) a.b1[i] = (a.mode == kPadPair) ? a.pb1[i] : a.pb1[(i / a.M) * 64 + i % a.M];
  for (int64_t i = tid; i < a.nW2; i += stride) {
    const int64_t d = i / a.M;
    const int m = (int)(i % a.M);
    const int64_t dv = (a.mode == kPadXHalf) ? (d / a.xg) * 64 + d % a.xg : d;   // the virtual output row
    a.W2[i] = (a.mode == kPadPair) ? a.pW2[d * 64 + 32 * ((d >> 5) & 1) + m] : a.pW2[dv * 64 + m];
  }
  for (int64_t i = tid; i < a.nD; i += stride) {
    const int64_t dv = (a.mode == kPadXHalf) ? (i / a.xg) * 64 + i % a.xg : i;
    a.b2[i] = a.pb2[dv];
    a.ls[i] = a.pls[dv];
  }
  for (int64_t i = tid; i < a.nWs; i += stride) {   // [G][K][H] <- [G][64][64]; pairs: the two groups' columns side by side
    const int h = (int)(i % a.H);
    const int64_t r = i / a.H;
    const int k = (int)(r % a.K);
    const int64_t g = r / a.K;
    if (a.mode == kPadPair) a.Ws[i] = a.pWs[((g >> 1) * 64 + k) * 64 + 32 * (g & 1) + h];
    else a.Ws[i] = a.pWs[(g * 64 + k) * 64 + h];
  }
  for (int64_t i = tid; i < a.ndz; i += stride) a.dz[i] = a.pdz[i];   // the first K rows of [64, N]
}

int launch_decoder_pad64(const oivae_shape_t& s, const oivae_tensors_t* params, const oivae_tensors_t* grads,
                         const float* X, int64_t B, int32_t S, const float* z, float gen_sig_off, float* dz,
                         Workspace& w, hipStream_t st) {
  if (!w.dec_pad) return fail(OIVAE_E_WORKSPACE, "padded decoder: no room for the padded arrays in the workspace");
  const int64_t N = B * (int64_t)S;
  const oivae_shape_t s64 = padded_shape(s);
  const DecLayout L = dec_layout(s), P = dec_layout(s64);
  const int chunks = w.n_chunks;
  auto up = [](int64_t n) { return (n + 63) / 64 * 64; };
  float* pp = w.dec_pad;
  float* pg = pp + up(P.total);
  float* pslab = pg + up(P.total);
  float* pz = pslab + (chunks > 1 ? up(P.total * chunks) : 0);
  float* pdz = pz + up(N * 64);
  oivae_tensors_t p64 = *params, g64 = *grads;   // (the encoder's entries are not the decoder's business)
  p64.gen_W1 = pp + P.oW1; p64.gen_b1 = pp + P.ob1; p64.gen_W2 = pp + P.oW2; p64.gen_b2 = pp + P.ob2;
  p64.log_sigma = pp + P.oLs; p64.Ws = pp + P.oWs;
  g64.gen_W1 = pg + P.oW1; g64.gen_b1 = pg + P.ob1; g64.gen_W2 = pg + P.oW2; g64.gen_b2 = pg + P.ob2;
  g64.log_sigma = pg + P.oLs; g64.Ws = pg + P.oWs;

  PadArgs a;
  a.W1 = params->gen_W1; a.b1 = params->gen_b1; a.W2 = params->gen_W2; a.b2 = params->gen_b2;
  a.ls = params->log_sigma; a.Ws = params->Ws; a.z = z;
  a.pW1 = p64.gen_W1; a.pb1 = p64.gen_b1; a.pW2 = p64.gen_W2; a.pb2 = p64.gen_b2; a.pls = p64.log_sigma; a.pWs = p64.Ws;
  a.pz = pz;
  a.nW1 = P.ob1 - P.oW1; a.nb1 = P.oW2 - P.ob1; a.nW2 = P.ob2 - P.oW2; a.nD = s64.D; a.nWs = P.total - P.oWs; a.nz = N * 64;
  a.K = s.K; a.H = s.H; a.M = s.M; a.mode = pad_mode(s); a.xg = s.uniform_group_dim;
  int64_t most = a.nW1 > a.nz ? a.nW1 : a.nz;
  if (a.nW2 > most) most = a.nW2;
  int64_t blocks = (most + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pad64_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  OIVAE_LAUNCH_CHECK();

  // the virtual launch writes G / 2 log-likelihood partials per chunk where finalize adds up G of them: the rest are zero
  if (pair32(s)) OIVAE_HIP_CHECK(hipMemsetAsync(w.ll_part, 0, sizeof(double) * (size_t)w.n_ll, st));
  Workspace w64 = w;
  w64.slab = pslab;
  w64.p_dec = P.total;
  w64.dec_pad = nullptr;
  w64.dec_xhalf = pad_mode(s) == kPadXHalf ? s.uniform_group_dim : 0;
  if (int rc = launch_decoder_mfma(s64, &p64, &g64, X, B, S, pz, gen_sig_off, pdz, w64, st)) return rc;

  UnpadArgs u;
  u.pW1 = g64.gen_W1; u.pb1 = g64.gen_b1; u.pW2 = g64.gen_W2; u.pb2 = g64.gen_b2; u.pls = g64.log_sigma; u.pWs = g64.Ws;
  u.pdz = pdz;
  u.W1 = grads->gen_W1; u.b1 = grads->gen_b1; u.W2 = grads->gen_W2; u.b2 = grads->gen_b2; u.ls = grads->log_sigma;
  u.Ws = grads->Ws; u.dz = dz;
  u.nW1 = L.ob1 - L.oW1; u.nb1 = L.oW2 - L.ob1; u.nW2 = L.ob2 - L.oW2; u.nD = s.D; u.nWs = L.total - L.oWs;
  u.ndz = N * (int64_t)s.K;
  u.K = s.K; u.H = s.H; u.M = s.M; u.mode = pad_mode(s); u.xg = s.uniform_group_dim;
  hipLaunchKernelGGL(unpad64_kernel, dim3((unsigned)blocks), dim3(256), 0, st, u);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

}  // namespace oivae
