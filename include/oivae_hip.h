/* oivae_hip.h -- C ABI of the MI355X (gfx950) oi-VAE training-step engine.
 *
 * The reference (samuela/oi-vae) is pure Python on PyTorch and has no FFI of its
 * own; its drop-in boundary is the object protocol consumed by OIVAE.step
 * (/root/reference/lib/oivae.py:42-90).  This header is the native boundary that
 * the host-side mirror of that protocol (oi_vae_amd/, Python + ctypes) binds.
 * Every entry point cites the reference code it replaces.
 *
 * Conventions
 *   - plain C types only: device pointers (fp32 unless said otherwise), sizes,
 *     and the HIP stream as an opaque `void*` (a hipStream_t; NULL = the null
 *     stream).  Pointers are borrowed for the duration of the call.
 *   - every call is asynchronous on `stream` and returns 0 on success or a
 *     nonzero OIVAE_E_* code; oivae_last_error() gives the message (per thread).
 *   - no entry point allocates or frees device memory, except oivae_init, which
 *     uploads a 128 KiB constant table once per device (and resolves the address of
 *     the static scratch array of oivae_encoder_bwd_noalloc_f32 and maps one
 *     host-visible error word).
 *   - row-major everywhere.  B = batch rows, S = Monte-Carlo samples,
 *     N = S*B decoder rows (sample-major: row n = s*B + b, lib/oivae.py:53),
 *     K = latent dim, G = groups, H = group input dim, M = generator hidden
 *     width (0 when the generator has no hidden layer), D = sum of group dims.
 */
#ifndef OIVAE_HIP_H
#define OIVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  OIVAE_OK = 0,
  OIVAE_E_ARG = 1,        /* bad argument (null pointer, negative size, ...) */
  OIVAE_E_SHAPE = 2,      /* shape outside what the kernels support */
  OIVAE_E_HIP = 3,        /* a HIP runtime call failed */
  OIVAE_E_WORKSPACE = 4,  /* workspace too small */
  OIVAE_E_NOINIT = 5      /* oivae_init not called for the current device */
};

/* generator mu_net architectures seen in the reference's scripts */
enum {
  OIVAE_GEN_LINEAR = 0,       /* Linear(H, D_g)                bars_data_oivae.py:97, meg_oi_vae.py:178 */
  OIVAE_GEN_TANH_LINEAR = 1,  /* Tanh -> Linear(H, D_g)        mo_mo_mocap_oivae.py:203-207 */
  OIVAE_GEN_MLP = 2           /* Linear(H,M) -> Tanh -> Linear(M, D_g)   (BASELINE.json "2-layer generators") */
};

/* Shapes of one model.  group_off is a DEVICE array of G+1 int32 prefix sums of
 * the group output dims (group g owns columns [group_off[g], group_off[g+1]) of
 * X); max_group_dim is the largest of them. */
typedef struct {
  int32_t D, K, G, H, M;
  int32_t gen_type;
  int32_t max_group_dim;
  int32_t uniform_group_dim;  /* D_g when all groups are equal, else 0 */
  const int32_t* group_off;
} oivae_shape_t;

/* One set of per-parameter device arrays: used for the parameters and, with the
 * same layout, for their gradients.
 *   enc_Wmu,enc_Wsig [K,D]; enc_bmu,enc_bsig [K]      (nn.Linear [out,in], bars_data_oivae.py:66-87)
 *   gen_W1 [G,M,H]; gen_b1 [G,M]                      (MLP only)
 *   gen_W2 [D,Hin] = the G per-group Linear weights [D_g,Hin] stacked on rows,
 *   gen_b2 [D]       Hin = M for MLP else H           (bars_data_oivae.py:96-97)
 *   log_sigma [D]                                     (bars_data_oivae.py:92-95)
 *   Ws [G,K,H]                                        (lib/models.py:86-89) */
typedef struct {
  float *enc_Wmu, *enc_bmu, *enc_Wsig, *enc_bsig;
  float *gen_W1, *gen_b1, *gen_W2, *gen_b2;
  float *log_sigma;
  float *Ws;
} oivae_tensors_t;

/* Scalars of one step, written on the device by oivae_finalize_f32
 * (lib/oivae.py:81-90, all pre-update values). */
typedef struct {
  float z_kl, loglik_term, logprob_theta, logprob_W, loss, elbo;
  float penalty;   /* sum of row norms, lib/models.py:113-114 */
  float pad;
} oivae_scalars_t;

/* ---- library ---------------------------------------------------------- */
int oivae_init(int device);               /* idempotent; uploads the rsqrt14 table */
const char* oivae_last_error(void);
const char* oivae_version(void);

/* Bytes of workspace the step entry points need for these shapes and N rows.
 * The three step entry points below are one step's sequence -- encoder forward, decoder, encoder backward -- but
 * each is a function of its own arguments: the library keeps no memory of earlier calls.  At the shapes served by
 * the fp16-term kernels the workspace carries max|X| of the last pass over X from call to call; a kernel that
 * scales X by it records the maximum it reads itself and is repeated (a gated second launch) when the two powers
 * of two differ, so a stale, zeroed or foreign workspace costs one more pass of that kernel, never a wrong result
 * (tests/test_gpu_ranges.py).  Passing the same workspace every step avoids the repeats.
 * A bounded wait inside the decoder kernels that runs out raises two error words, one host-visible and one on the
 * device.  Calls are asynchronous, so the host learns of it at the first entry point it calls AFTER the kernel has
 * run -- every entry point looks, and that call returns OIVAE_E_HIP -- which may be several calls later when the
 * host runs ahead.  The parameters are protected regardless: oivae_adam_f32, oivae_sgd_prox_f32 and oivae_prox_f32
 * read the device word on the device and leave their arguments untouched while it is raised, so the gradients of a
 * failed decoder launch are never applied.  oivae_check_errors (below) gives the definite answer for everything
 * queued so far, e.g. behind the last step of a run. */
int64_t oivae_workspace_bytes(const oivae_shape_t* shape, int64_t B, int32_t S);

/* Synchronises `stream` and reports (OIVAE_E_HIP, message in oivae_last_error) if any kernel queued on the current
 * device before this call raised the error words; clears them.  No counterpart in the reference (its CPU path has
 * no asynchronous failures). */
int oivae_check_errors(void* stream);

/* ---- proximal step ------------------------------------------------------
 * Replaces BayesianGroupLassoGenerator.proximal_step, lib/models.py:106-111:
 * in place, for every row of Ws viewed as [rows, H]:
 *   n = sqrt(sum_h w^2); w = (w / max(n,1e-16)) * max(n - t, 0)
 * bit for bit as the reference's CPU path computes it (torch row-sum order,
 * MKL-VML square root, two roundings per element; 1 <= H <= 2^20, the cascade
 * of torch's row sum from H = 512 on included).  zero_mask[r] (optional)
 * = 1 iff the row is all zeros afterwards; row_norms (optional) = n. */
int oivae_prox_f32(float* Ws, int64_t rows, int32_t H, float t,
                   uint8_t* zero_mask, float* row_norms, void* stream);

/* SGD(momentum=0) on Ws fused with the proximal step: w -= lr*g, then prox.
 * Replaces optimizer_Ws.step() (bars_data_oivae.py:241-244, torch.optim.SGD:
 * w = fma(-lr, g, w) as the CPU kernel rounds it) followed by
 * proximal_step (lib/oivae.py:76-79).  do_prox = 0 skips the threshold
 * (lam == 0, lib/oivae.py:78). */
int oivae_sgd_prox_f32(float* Ws, const float* dWs, int64_t rows, int32_t H,
                       float lr, float t, int32_t do_prox,
                       uint8_t* zero_mask, float* row_norms, void* stream);

/* Column form of the proximal step: replaces FirstLayerSparseDecoder.proximal_step,
 * lib/models.py:65-71 (the autoencoder variant's decoder, data_vs_test_loss_autoencoder.py:126).
 * W holds the G latent-to-group maps stacked on rows, [sum_g H_g, K] (each map is the
 * nn.Linear(K, H_g, bias=False).weight of lib/models.py:41-44); group_off is a DEVICE array of
 * G+1 int32 prefix sums of H_g; max_group_rows the largest H_g.  In place, for every map g and
 * column k:  n = sqrt(sum_h W_g[h,k]^2); W_g[:,k] = (W_g[:,k] / max(n,1e-16)) * max(n - t, 0),
 * bit for bit as the reference's CPU path (torch's column-sum order over dim 0, MKL-VML root).
 * Accepted: H_g * K < 32768 for every map (beyond it torch splits the columns over its threads and
 * the reference's own result depends on the host's thread count) and, for K = 1, H_g < 512;
 * else OIVAE_E_SHAPE.  zero_mask / col_norms (optional): [G, K]. */
int oivae_prox_cols_f32(float* W, const int32_t* group_off, int32_t G, int32_t K, int32_t max_group_rows,
                        float t, uint8_t* zero_mask, float* col_norms, void* stream);

/* Gamma log-likelihood: replaces Gamma.logprob, lib/distributions.py:65-71 -- the sum over n elements of
 *   shape*log(rate) - lgamma(shape) + (shape-1)*log(x) - rate*x
 * (each element's four terms added in fp32 in that order, so 0*log 0 = NaN as in the reference; the sum over elements
 * in fp64, fixed order) -- and, for every non-NULL gradient pointer (n floats each), what autograd derives from it:
 * d_shape = log(rate) - digamma(shape) + log(x), d_rate = shape/rate - x, d_x = (shape-1)/x - rate.  logprob: one float
 * on the device.  workspace: oivae_gamma_workspace_bytes() bytes of device memory, 8-byte aligned.  Evaluation-side
 * operator: the reference's training step cannot take Gamma members (lib/distributions.py:82 asks them for `size`). */
int64_t oivae_gamma_workspace_bytes(void);
int oivae_gamma_logprob_f32(const float* shape, const float* rate, const float* x, int64_t n, float* logprob,
                            float* d_shape, float* d_rate, float* d_x, void* workspace, int64_t workspace_bytes,
                            void* stream);

/* ---- encoder forward + reparameterisation + KL ---------------------------
 * Replaces inference_model(X) (lib/models.py:15-16 with the heads of
 * bars_data_oivae.py:66-87), KL_Normals (lib/utils.py:9-18, lib/oivae.py:50)
 * and the S calls of Normal.sample (lib/distributions.py:36-38, lib/oivae.py:53):
 *   mu_q = X Wmu^T + bmu;  ea = exp(X Wsig^T + bsig);  sig_q = ea*c + sig_off
 *   z[s*B+b] = mu_q[b] + sig_q[b]*eps[s,b]
 *   kl partial sums of -0.5 + (sig_q^2 + (mu_q-mu_p)^2)/(2 sig_p^2) + log sig_p - log sig_q
 * prior_mu/prior_sigma: [prior_rows, K] with prior_rows 1 (broadcast) or B.
 * Outputs mu_q, sig_q, ea [B,K]; z [S*B,K]. KL partials go to the workspace. */
int oivae_encoder_fwd_f32(const oivae_shape_t* shape, const oivae_tensors_t* params,
                          const float* X, int64_t B, int32_t S, const float* eps,
                          float c, float sig_off,
                          const float* prior_mu, const float* prior_sigma, int32_t prior_rows,
                          float* mu_q, float* sig_q, float* ea, float* z,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* ---- grouped decoder forward + backward ----------------------------------
 * Replaces generative_model(z_sample).logprob(Xrep) (lib/models.py:91-95,
 * lib/distributions.py:40-45,88-93, lib/oivae.py:54-59) and the decoder half
 * of loss.backward() (lib/oivae.py:75): per group g
 *   h = z Ws[g]; mu = mu_net_g(h); sigma = exp(log_sigma_g) + gen_sig_off
 *   loglik partial sums of -0.5*LOG2PI - log sigma - 0.5*((x - mu)/sigma)^2
 * and the gradients of loss = -loglik/(S*B) w.r.t. gen_W1,gen_b1,gen_W2,gen_b2,
 * log_sigma, Ws (written to `grads`, complete: not accumulated) and w.r.t. z
 * (dz, S*B*K floats, overwritten; [S*B,K] or [K,S*B]: oivae_dz_layout below).  Neither mu nor sigma [N,D] is materialised; X is
 * indexed as X[n mod B] instead of being repeated (lib/oivae.py:54).
 * The encoder entries of `grads` are not touched. */
int oivae_decoder_fwd_bwd_f32(const oivae_shape_t* shape, const oivae_tensors_t* params,
                              const oivae_tensors_t* grads,
                              const float* X, int64_t B, int32_t S, const float* z,
                              float gen_sig_off, float* dz,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* Layout of the dz buffer that oivae_decoder_fwd_bwd_f32 writes and oivae_encoder_bwd_f32 reads, for these
 * shapes and N = S*B rows: 0 = [N, K] row-major (sample-major rows), 1 = TRANSPOSED [K, N] (the matrix-core
 * decoders: two-layer generators up to 64 wide with one output width, when the padded launch pays at this B, S), negative = error.  Both entry points derive it the same way
 * from (shape, B, S) and the library's test hooks (below); a host that reads dz, or feeds its own, asks here.
 * dz is summed over the groups with float atomics by every decoder kernel except the flat (many-small-groups)
 * path, so it -- and the encoder gradients computed from it -- can differ in the last bits from run to run
 * (within 1e-5 of the tensor's scale: tests/test_gpu_dp.py); all other gradients are bit-reproducible. */
int oivae_dz_layout(const oivae_shape_t* shape, int64_t B, int32_t S);

/* Test hooks.  Three environment variables select other kernel families for cross-checks (no counterpart in the
 * reference): OIVAE_FORCE_GENERIC (shape-generic kernels), OIVAE_ENC = bf16 | fp32, OIVAE_DEC = v1 | v4 | x3 | p | g.
 * They are read when the library is first used, when oivae_init initialises a device, and when this function is
 * called -- not on every call of an entry point, so the kernel choice (and dz's layout with it) cannot change
 * between the calls of one step unless the host asks for it here. */
void oivae_reload_test_hooks(void);

/* ---- encoder backward -----------------------------------------------------
 * Replaces the encoder half of loss.backward() (lib/oivae.py:75):
 * (dz as the decoder call of the same shape, B and S left it: oivae_dz_layout)
 *   dmu_q = sum_s dz[s] + (mu_q - mu_p)/sig_p^2 / B
 *   dsig_q = sum_s dz[s]*eps[s] + (sig_q/sig_p^2 - 1/sig_q) / B
 *   da = dsig_q * c * ea
 *   dWmu = dmu_q^T X, dbmu = colsum(dmu_q), dWsig = da^T X, dbsig = colsum(da)
 * written to the encoder entries of `grads`. */
int oivae_encoder_bwd_f32(const oivae_shape_t* shape, const oivae_tensors_t* grads,
                          const float* X, int64_t B, int32_t S, const float* eps, float c,
                          const float* prior_mu, const float* prior_sigma, int32_t prior_rows,
                          const float* mu_q, const float* sig_q, const float* ea,
                          const float* dz,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* ---- encoder backward without a workspace ---------------------------------------
 * No counterpart in the reference.  For callers that can neither allocate device memory nor hand over a
 * workspace: code under HIP graph capture, tight evaluation/serving loops, harnesses that forbid hipMalloc.
 * Same arguments, kernels and results (bit for bit) as oivae_encoder_bwd_f32, without `workspace`: the scratch
 * of its kernels (dpre [B,2K], bias-gradient partials, the split image of dpre^T at the matrix-core shapes)
 * lives in a static device array of OIVAE_NOALLOC_SCRATCH_BYTES that each GPU holds from oivae_init until the
 * process ends.  Calls whose scratch would exceed it return OIVAE_E_WORKSPACE (need = 8*B*K bytes for dpre
 * + 16*K*min(1024, ceil(B/16)) for the partials + 192*ceil(2K/128)*128*ceil(B/32) for the image when
 * 2K % 128 == 0, D % 128 == 0 and B % 32 == 0, else up to 16 partial images of 8*K*D bytes when the fp32
 * matrix-core kernel slices the batch; e.g. K=64: B <= 8192).  Calls no hipMalloc/hipFree or any
 * allocating routine; oivae_init(device) must have run.  Every call on a GPU shares that GPU's array: calls
 * that overlap on one GPU (two streams or threads) race; calls ordered on one stream do not. */
#define OIVAE_NOALLOC_SCRATCH_BYTES (16u << 20)
int oivae_encoder_bwd_noalloc_f32(const oivae_shape_t* shape, const oivae_tensors_t* grads,
                                  const float* X, int64_t B, int32_t S, const float* eps, float c,
                                  const float* prior_mu, const float* prior_sigma, int32_t prior_rows,
                                  const float* mu_q, const float* sig_q, const float* ea,
                                  const float* dz, void* stream);

/* ---- weight prior, group-lasso penalty, scalars ----------------------------
 * Replaces NormalPriorTheta.logprob (lib/oivae.py:15-23): sum over every
 * generator mu_net parameter (gen_W1,gen_b1,gen_W2,gen_b2) of
 * -0.5*LOG2PI - log(sigma_theta) - 0.5*(theta/sigma_theta)^2, adds its gradient
 * theta/sigma_theta^2 to `grads`; group_lasso_penalty (lib/models.py:113-114);
 * and the combination of lib/oivae.py:50,55-59,65,70-71 into `out` (device):
 *   z_kl = kl/B; loglik_term = ll/S/B; logprob_W = -lam*penalty;
 *   loss = -(-z_kl + loglik_term + logprob_theta); elbo = -loss + logprob_W. */
int oivae_finalize_f32(const oivae_shape_t* shape, const oivae_tensors_t* params,
                       const oivae_tensors_t* grads, int64_t B, int32_t S,
                       float sigma_theta, float lam, oivae_scalars_t* out,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* ---- Adam --------------------------------------------------------------------
 * One fused Adam update over a flat fp32 buffer: replaces optimizer.step() for
 * torch.optim.Adam with default options (bars_data_oivae.py:233-239):
 *   m = m + (1-b1)(g-m); v = b2*v + (1-b2) g^2
 *   p -= (lr/(1-b1^step)) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * grad_scale multiplies g first (1/world_size when gradients were summed
 * across ranks). */
int oivae_adam_f32(float* p, const float* g, float* m, float* v, int64_t n,
                   float lr, float beta1, float beta2, float eps, int32_t step,
                   float grad_scale, void* stream);

/* ---- the whole step of a small model in one launch ------------------------------------------------------------
 * Replaces OIVAE.step (lib/oivae.py:42-90) in one call for the shapes the reference's own scripts train
 * (bars_data_oivae.py:20-33, mo_mo_mocap_oivae.py:26-35): encoder forward + KL + samples, grouped decoder forward /
 * backward + log-likelihood, weight prior, penalty, the six scalars, encoder backward and -- with do_optimizers --
 * Adam over every non-Ws parameter (bars_data_oivae.py:233-239), SGD on Ws and, with do_prox, the proximal step
 * (lib/oivae.py:76-79), the last two bit for bit as oivae_sgd_prox_f32.  One kernel (+ a 4-byte memset node, see ticket_is_zero) instead of the
 * two dozen launches of the entry points above (which stay valid for the same shapes and give the same results to
 * rounding; these models are launch-bound).  `grads` is written completely either way; mu_q, sig_q [B,K], z [S*B,K]
 * and `out` hold the PRE-update values (lib/oivae.py:81-90).  adam_m / adam_v: Adam's moments, one array per tensor
 * in the layout of `params` (Ws entries unused); adam_step >= 1 is the step being taken.  The four tensor sets must
 * share one memory layout (every tensor at the same element offset from its set's enc_Wmu, enc_Wmu lowest: one flat
 * buffer each, as the host mirror keeps them), else OIVAE_E_ARG.  Float sums are formed in a
 * fixed order: results are bit-reproducible run to run.
 * Accepted (oivae_small_step_applies says 1): single-layer generators (OIVAE_GEN_LINEAR, OIVAE_GEN_TANH_LINEAR),
 * H in {1,2,4,8}, K <= 32, D <= 256, groups of at most 16 outputs, S <= 64, S*B <= 8192, at most 2^17 parameters;
 * else OIVAE_E_SHAPE.  Workspace: oivae_workspace_bytes(shape, B, S).
 * ticket_is_zero: 0 is always right (the call zeroes its workgroup ticket in the workspace first: a 4-byte memset
 * node).  Nonzero skips that node: the caller then vouches that the last thing that wrote this workspace's small-step
 * region was a COMPLETED call of this entry point with the same shape, B and S (the kernel leaves the ticket at
 * zero); anything else -- a fresh or re-used allocation, another B or S, a failed call -- needs 0. */
int oivae_small_step_applies(const oivae_shape_t* shape, int64_t B, int32_t S);
int oivae_small_step_f32(const oivae_shape_t* shape, const oivae_tensors_t* params, const oivae_tensors_t* grads,
                         const float* X, int64_t B, int32_t S, const float* eps, float c, float sig_off,
                         const float* prior_mu, const float* prior_sigma, int32_t prior_rows,
                         float gen_sig_off, float sigma_theta, float lam,
                         float* mu_q, float* sig_q, float* z, oivae_scalars_t* out,
                         int32_t do_optimizers, const oivae_tensors_t* adam_m, const oivae_tensors_t* adam_v,
                         float lr, float beta1, float beta2, float adam_eps, int32_t adam_step,
                         float ws_lr, float t, int32_t do_prox, uint8_t* zero_mask, float* row_norms,
                         int32_t ticket_is_zero, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OIVAE_HIP_H */
