// Shared host/device helpers of the oi-VAE HIP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/oivae_hip.h"

namespace oivae {

// fp32 value of log(fp32(2*pi)) as the reference computes it
// (/root/reference/lib/distributions.py:9).
constexpr float kLog2Pi = 1.8378770351409912f;

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define OIVAE_HIP_CHECK(expr)                                                        \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess)                                                            \
      return ::oivae::fail(OIVAE_E_HIP, "%s failed: %s (%s:%d)", #expr,              \
                           hipGetErrorString(_e), __FILE__, __LINE__);               \
  } while (0)

#define OIVAE_LAUNCH_CHECK() OIVAE_HIP_CHECK(hipGetLastError())

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Device pointer to the 65536-entry rsqrt14 table of the current device
// (uploaded by oivae_init), or nullptr.
const uint16_t* rsqrt14_table_device();
// The current device's error words, raised together by a kernel whose bounded wait runs out (dec_shared.h
// lds_wait_ge): `host` is one word of host-mapped memory the library polls, `dev` its device-resident twin (behind the
// rsqrt14 table, the allocation every proximal kernel already holds a pointer to) that the UPDATE kernels -- Adam,
// SGD, the proximal step -- read first: with it raised they return without touching a parameter, so gradients of a
// failed launch are never applied, however far ahead of the device the host has queued.
// error_words_device(): what kernels take (null pointers before oivae_init).  poll_error_word(): called first by
// every entry point -- a raised word fails THAT call with OIVAE_E_HIP, clears the host word at once and the device
// word on `stream` (behind everything already queued there).
struct ErrWords { uint32_t* host; uint32_t* dev; };
ErrWords error_words_device();
const uint32_t* bad_word_device();   // the device-resident word alone (nullptr before oivae_init)
constexpr int kBadWordOffsetU16 = 65536;   // its place behind the 65536-entry table, in table entries
int poll_error_word(const char* who, hipStream_t stream = nullptr);

// ---- workspace layout -------------------------------------------------------
// One flat device buffer, carved identically by every entry point.
struct Workspace {
  // reduction partials (double), fixed slot counts
  double* kl_part;      int32_t n_kl;      // one per encoder-forward block
  double* ll_part;      int32_t n_ll;      // one per decoder block
  double* theta_part;   int32_t n_theta;   // weight-prior partials
  double* pen_part;     int32_t n_pen;     // penalty partials
  float*  dpre;                            // [enc_splits, B, 2K] pre-activation partials (fwd), then [B, 2K] their grads (bwd)
  int32_t enc_splits;
  double* bias_part;    int32_t n_bias;    // [n_bias, 2K] per-block column sums of dpre (encoder bias gradients)
  float*  slab;         int32_t n_chunks;  // [n_chunks, P_dec] decoder grad partials (n_chunks > 1)
  int64_t p_dec;                           // floats per decoder-grad slab
  float*  flat;                            // flat (many-small-groups) decoder: Hc | Wc | Wk | Ct | dz partials
  uint32_t* x3_z;                          // x3 decoder: split image of z, 24 KB per 32 rows
  uint32_t* x3_img;                        // x3 encoder: split image of the weights (fwd) / of dpre (bwd)
  uint32_t* ranges;                        // 64 words: operand maxima of the fp16 kernels (x3_encoder.hip "ranges")
  float*  enc_bwd_part;                    // fp32 matrix-core encoder backward: [slices][2K][D] partial images (few-tile shapes)
  uint32_t* dec_zmax;                      // dec_x3_64s/g: 256 per-workgroup maxima of |z| (dec_absmax_f32 -> x3_split_z)
  char*   small;                           // small_step.hip: per-block gradient slabs, KL / log-lik partials, ticket counter
  float*  dec_acc2;                        // dec_x3_64s: second-level slabs of its weight-gradient waves (nullptr: not needed)
  float*  dec_pad;                         // dec_pad64.hip: zero-padded parameters, gradients, slabs, z and dz of the {32,64} mixes (nullptr: other shapes)
  int32_t dec_xhalf;                       // dec_pad64.hip's virtual launch for D_g < 64: the columns of X per group (0 everywhere else)
  int64_t total_bytes;
};

constexpr int kMaxPartials = 16384;
constexpr int kEncEpiBlocks = 2048;  // cap on enc_fwd_epilogue blocks (= KL partials)
// bytes of the encoder backward's per-block partials: [n_bias, 2K] column sums of dpre (doubles) and, behind them,
// [n_bias, 2K] column maxima of |dpre| (bit patterns) for the fp16 backward GEMM's per-column scales
inline int64_t enc_bias_part_bytes(int n_bias, int K) { return (int64_t)(sizeof(double) + sizeof(uint32_t)) * n_bias * 2 * K; }
constexpr int kBiasBlocks = 1024;    // cap on enc_bwd_pre blocks (= bias-gradient partials)
constexpr int kMaxEncSplits = 8;     // split-K partial images of the encoder pre-activations  // cap on per-kernel block counts that emit partials

// sizes of the decoder parameter block, in floats, in slab order:
// [gen_W1 | gen_b1 | gen_W2 | gen_b2 | log_sigma | Ws]
struct DecLayout {
  int64_t oW1, ob1, oW2, ob2, oLs, oWs, total;
};

inline DecLayout dec_layout(const oivae_shape_t& s) {
  DecLayout L;
  int64_t hin = (s.gen_type == OIVAE_GEN_MLP) ? s.M : s.H;
  int64_t nW1 = (s.gen_type == OIVAE_GEN_MLP) ? (int64_t)s.G * s.M * s.H : 0;
  int64_t nb1 = (s.gen_type == OIVAE_GEN_MLP) ? (int64_t)s.G * s.M : 0;
  L.oW1 = 0;
  L.ob1 = L.oW1 + nW1;
  L.oW2 = L.ob1 + nb1;
  L.ob2 = L.oW2 + (int64_t)s.D * hin;
  L.oLs = L.ob2 + s.D;
  L.oWs = L.oLs + s.D;
  L.total = L.oWs + (int64_t)s.G * s.K * s.H;
  return L;
}

// rows per decoder tile and number of row chunks for the generic decoder
struct DecPlan {
  int32_t R;        // rows per tile (32 or 64)
  int32_t DC;       // output-column chunk
  int32_t chunks;   // row chunks (grid.y)
  int64_t lds_bytes;
  bool mfma;        // fast path applies
  bool flat;        // many-small-groups path applies
  bool single;      // single-layer generators with a skinny H: dec_single (generic.hip)
};

DecPlan plan_decoder(const oivae_shape_t& s, int64_t N);
int64_t mfma_decoder_acc2_bytes(const oivae_shape_t& s, int64_t N, int chunks);   // mfma_decoder.hip
// dec_pad64.hip: K, H, M in {32, 64} (not all 64) with 64 outputs per group run through the all-64 fp16-term kernel on
// zero-padded copies
bool dec_pad64_applies(const oivae_shape_t& s);
bool dec_pad64_xhalf(const oivae_shape_t& s);   // ... with fewer than 64 outputs per group padded to 64 (dec_x3_64s<true> only)
// shapes outside {32, 64}^4 take the padded launch from this much per-group work on, K H + H M + M D_g (all-64: 12288),
// or from this many (row, group) pairs on (mfma_decoder_pays, mfma_decoder.hip)
constexpr int kPadMinWork = 1536;
constexpr int64_t kPadMinRowGroups = 524288;
int64_t dec_pad64_bytes(const oivae_shape_t& s, int64_t N, int chunks);
int launch_decoder_pad64(const oivae_shape_t& s, const oivae_tensors_t* params, const oivae_tensors_t* grads,
                         const float* X, int64_t B, int32_t S, const float* z, float gen_sig_off, float* dz,
                         Workspace& w, hipStream_t st);
// small_step.hip: the whole step of a small model in one launch
struct SmallPlan { bool ok; int RB, NR, GR, nblk; int64_t P; int64_t ws_bytes; size_t lds_bytes; };
SmallPlan plan_small_step(const oivae_shape_t& s, int64_t B, int32_t S);

// flat (many-small-groups) decoder, flat_decoder.hip
struct FlatLayout { int64_t oHc, oWc, oWk, oCt, oDzp, oImgA, oImgB, oImgC, total; int splits; bool h2; };
bool flat_decoder_applies(const oivae_shape_t& s, int64_t N);
int64_t flat_decoder_floats(const oivae_shape_t& s, int64_t N, FlatLayout* L);
int flat_decoder_blocks(const oivae_shape_t& s);
int carve_workspace(const oivae_shape_t& s, int64_t B, int32_t S, void* base, int64_t bytes,
                    Workspace* out);
int check_shape(const oivae_shape_t* s);
int mfma_encoder_fwd_splits(const oivae_shape_t& s, int64_t B);  // mfma_encoder.hip
int mfma_encoder_bwd_splits(const oivae_shape_t& s, int64_t B);  // mfma_encoder.hip: batch slices of the backward GEMM
inline int64_t enc_bwd_part_bytes(const oivae_shape_t& s, int64_t B, bool x3_bwd) {
  const int sp = x3_bwd ? 1 : mfma_encoder_bwd_splits(s, B);
  return sp > 1 ? (int64_t)sizeof(float) * sp * 2 * s.K * s.D : 0;
}
// x3_encoder.hip: fp32 GEMMs as six bf16 matrix products
// Test hooks (kernel-family switches for cross-checks): OIVAE_FORCE_GENERIC, OIVAE_ENC, OIVAE_DEC are read from the
// environment by oivae_init and by oivae_reload_test_hooks, never per call: the kernel choice -- and with it dz's
// layout -- cannot change between the decoder call and the encoder backward call of a step behind the host's back.
struct TestHooks {
  bool force_generic;   // OIVAE_FORCE_GENERIC set: shape-generic kernels everywhere
  char enc;             // OIVAE_ENC: 0 default (fp16 terms), 'b' bf16 terms, 'f' fp32 matrix instructions
  char dec[2];          // OIVAE_DEC: first two characters (v1, v4, x3, p, g), 0 when unset
};
const TestHooks& test_hooks();
void reload_test_hooks();
bool x3_encoder_applies(const oivae_shape_t& s);
int64_t x3_encoder_ws_bytes(const oivae_shape_t& s, int64_t B);
uint32_t* h2_bwd_colmax(uint32_t* img, int J, int64_t B);   // the per-column maxima of dpre behind its image (rows_pad words)
int64_t x3_encoder_bwd_img_bytes(const oivae_shape_t& s, int64_t B);  // image of dpre^T alone (0 when the x3 backward kernels do not apply)
int x3_encoder_fwd_splits(const oivae_shape_t& s, int64_t B);
bool x3_encoder_bwd_applies(const oivae_shape_t& s, int64_t B);
int launch_encoder_bwd_x3(const oivae_shape_t& s, const oivae_tensors_t* grads, const float* X, const float* dpre,
                          int64_t B, uint32_t* img, uint32_t* rng, hipStream_t st, bool colmax_ready = false);
int launch_encoder_fwd_x3(const oivae_shape_t& s, const oivae_tensors_t* params, const float* X, int64_t B,
                          float* pre, uint32_t* wimg, uint32_t* rng, int splits, hipStream_t st,
                          bool image_ready = false);   // image_ready: wimg and rng[1] (max|W|) were prepared by the caller
void launch_absmax_f32(const float* v, int64_t n, uint32_t* out, hipStream_t st);   // atomicMax into *out (caller zeroes it)
// the kernel sequence of oivae_encoder_bwd_f32 on explicit scratch regions (generic.hip): dpre [B, 2K] floats,
// bias_part [n_bias, 2K] doubles, x3_img x3_encoder_bwd_img_bytes (or the workspace's larger image), rng 64 words
int encoder_bwd_run(const oivae_shape_t& s, const oivae_tensors_t* grads, const float* X, int64_t B, int32_t S,
                    const float* eps, float c, const float* prior_mu, const float* prior_sigma, int32_t prior_rows,
                    const float* mu_q, const float* sig_q, const float* ea, const float* dz, float* dpre,
                    double* bias_part, int n_bias, uint32_t* x3_img, uint32_t* rng, float* bwd_part, hipStream_t st);
// atomicMax on a word that thousands of workgroups raise at about the same time (the maxima the fp16 kernels record
// while they stream X): same-address atomics are served one by one, ~14 ns each, so the ones that cannot change the
// word are skipped after a plain coherent load of it.  The word ends at the same maximum either way.
#if defined(__HIPCC__)
__device__ __forceinline__ void atomic_max_if_larger(uint32_t* p, uint32_t v) {
  if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) atomicMax(p, v);
}
#endif
#if defined(__HIPCC__)
// torch.optim.Adam's element update (default options) as adam_kernel (api.hip) and the small-model step kernel
// (small_step.hip) apply it; contraction as in api.hip's translation unit whatever the including one's flags are.
__device__ __forceinline__ void adam_update_dev(float& p, float g, float& m, float& v, float one_minus_b1, float b2,
                                                float one_minus_b2, float step_size, float inv_bc2_sqrt, float eps) {
#pragma clang fp contract(fast)
  float mi = m, vi = v;
  mi = mi + one_minus_b1 * (g - mi);           // exp_avg.lerp_(grad, 1-beta1)
  vi = b2 * vi + one_minus_b2 * g * g;         // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
  const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
  p = p - step_size * (mi / denom);
  m = mi;
  v = vi;
}
#endif
inline int enc_bias_blocks(int64_t B) { const int64_t bb = (B + 15) / 16; return (int)(bb > kBiasBlocks ? kBiasBlocks : bb); }
// noalloc.hip: resolves the device address of the static scratch array for `device` (which must be current)
int noalloc_scratch_setup(int device);
// decoder gradient partials [chunks][DecLayout::total] -> the six gradient tensors, summed in chunk order (one launch)
struct SlabSegs { float* dst[6]; int64_t off[6], n[6]; };
int launch_slab_reduce(const float* slab, const DecLayout& L, int chunks, const oivae_tensors_t* grads, hipStream_t st);
void launch_partial_sum(const float* parts, int64_t P, int64_t n, int chunks, float* dst, hipStream_t st);
void launch_partial_sum2(const float* parts, int64_t P, int64_t n, int chunks, float* dst0, float* dst1, hipStream_t st);

}  // namespace oivae
