// oivae_encoder_bwd_noalloc_f32: the encoder backward entry point for callers that cannot allocate device memory
// and cannot hand over a workspace either (code under HIP graph capture, tight evaluation/serving loops, harnesses
// that forbid hipMalloc).  Same kernels as oivae_encoder_bwd_f32 (generic.hip: encoder_bwd_run); the three scratch
// regions those kernels need -- the pre-activation gradients dpre [B, 2K], the per-block bias-gradient partials and,
// at the matrix-core shapes, the split image of dpre^T -- live in a file-scope device array instead of a workspace.
//
// The array's NAME is a host-side stand-in, not a device address: every use below goes through the per-device
// pointer that hipGetSymbolAddress returned at oivae_init (each GPU holds its own copy at its own address).
// All calls on one GPU share that GPU's array and it keeps the last call's contents; every region is completely
// rewritten by its producer kernel before it is read (nothing has to start at zero), but two calls that overlap
// on one GPU (two streams, two threads) race on it.
#include "common.h"

namespace oivae {

static __device__ __align__(256) unsigned char scratch[OIVAE_NOALLOC_SCRATCH_BYTES];

constexpr int kMaxDev = 64;
static unsigned char* g_scratch_ptr[kMaxDev] = {nullptr};

int noalloc_scratch_setup(int device) {
  if (device < 0 || device >= kMaxDev) return fail(OIVAE_E_ARG, "noalloc scratch: bad device %d", device);
  if (g_scratch_ptr[device]) return OIVAE_OK;
  void* p = nullptr;
  OIVAE_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  if (!p) return fail(OIVAE_E_HIP, "noalloc scratch: hipGetSymbolAddress returned null on device %d", device);
  g_scratch_ptr[device] = static_cast<unsigned char*>(p);
  return OIVAE_OK;
}

}  // namespace oivae

using namespace oivae;

extern "C" int oivae_encoder_bwd_noalloc_f32(const oivae_shape_t* shape, const oivae_tensors_t* grads, const float* X,
                                             int64_t B, int32_t S, const float* eps, float c, const float* prior_mu,
                                             const float* prior_sigma, int32_t prior_rows, const float* mu_q,
                                             const float* sig_q, const float* ea, const float* dz, void* stream) {
  if (int rc = check_shape(shape)) return rc;
  if (!grads || !X || !eps || !prior_mu || !prior_sigma || !mu_q || !sig_q || !ea || !dz)
    return fail(OIVAE_E_ARG, "oivae_encoder_bwd_noalloc_f32: null pointer");
  if (B < 1 || S < 1) return fail(OIVAE_E_ARG, "oivae_encoder_bwd_noalloc_f32: B and S must be >= 1");
  if (prior_rows != 1 && prior_rows != B)
    return fail(OIVAE_E_ARG, "oivae_encoder_bwd_noalloc_f32: prior_rows must be 1 or B");
  if (int rc = poll_error_word("oivae_encoder_bwd_noalloc_f32", (hipStream_t)stream)) return rc;
  const oivae_shape_t& s = *shape;
  int dev = -1;
  OIVAE_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev || !g_scratch_ptr[dev])
    return fail(OIVAE_E_NOINIT, "oivae_encoder_bwd_noalloc_f32: oivae_init(%d) has not run", dev);
  // carve: dpre | bias partials | image of dpre^T | partial images of the fp32 backward GEMM | ranges
  const int n_bias = enc_bias_blocks(B);
  const int64_t b_dpre = align_up((int64_t)sizeof(float) * B * 2 * s.K, 256);
  const int64_t b_bias = align_up(enc_bias_part_bytes(n_bias, s.K), 256);
  const int64_t b_img = align_up(x3_encoder_bwd_img_bytes(s, B), 256);
  const int64_t b_part = align_up(enc_bwd_part_bytes(s, B, x3_encoder_bwd_applies(s, B)), 256);
  const int64_t need = b_dpre + b_bias + b_img + b_part + 256;
  if (need > (int64_t)OIVAE_NOALLOC_SCRATCH_BYTES)
    return fail(OIVAE_E_WORKSPACE,
                "oivae_encoder_bwd_noalloc_f32: B=%lld, K=%d need %lld bytes of scratch, the static array holds %lld; "
                "use oivae_encoder_bwd_f32 with a workspace", (long long)B, s.K, (long long)need,
                (long long)OIVAE_NOALLOC_SCRATCH_BYTES);
  unsigned char* base = g_scratch_ptr[dev];   // the device address, never the array's name
  float* dpre = reinterpret_cast<float*>(base);
  double* bias_part = reinterpret_cast<double*>(base + b_dpre);
  uint32_t* img = reinterpret_cast<uint32_t*>(base + b_dpre + b_bias);
  float* part = reinterpret_cast<float*>(base + b_dpre + b_bias + b_img);
  uint32_t* rng = reinterpret_cast<uint32_t*>(base + b_dpre + b_bias + b_img + b_part);
  return encoder_bwd_run(s, grads, X, B, S, eps, c, prior_mu, prior_sigma, prior_rows, mu_q, sig_q, ea, dz, dpre,
                         bias_part, n_bias, img, rng, part, (hipStream_t)stream);
}
