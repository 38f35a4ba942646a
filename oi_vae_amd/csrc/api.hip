// Library plumbing of the oi-VAE HIP engine: errors, init, workspace carving,
// the fused Adam update and the scalar finalisation.
#include "common.h"

#include <string.h>
#include <mutex>

// The rsqrt14 table (tools/gen_rsqrt14_table.c) is linked into the library.
#ifndef OIVAE_RSQRT14_BIN
#error "build with -DOIVAE_RSQRT14_BIN=\"/abs/path/to/rsqrt14_u16.bin\""
#endif
__asm__(
    ".section .rodata\n"
    ".balign 16\n"
    ".global oivae_rsqrt14_blob\n"
    "oivae_rsqrt14_blob:\n"
    ".incbin \"" OIVAE_RSQRT14_BIN "\"\n"
    ".global oivae_rsqrt14_blob_end\n"
    "oivae_rsqrt14_blob_end:\n"
    ".previous\n");
extern "C" const unsigned char oivae_rsqrt14_blob[];
extern "C" const unsigned char oivae_rsqrt14_blob_end[];

namespace oivae {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

constexpr int kMaxDevices = 64;
static uint16_t* g_tab[kMaxDevices] = {nullptr};
static std::mutex g_init_mu;

static uint32_t* g_err_host[kMaxDevices] = {nullptr};
static uint32_t* g_err_dev[kMaxDevices] = {nullptr};

ErrWords error_words_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || !g_tab[dev]) return ErrWords{nullptr, nullptr};
  return ErrWords{g_err_dev[dev], reinterpret_cast<uint32_t*>(g_tab[dev] + kBadWordOffsetU16)};
}

const uint32_t* bad_word_device() { return error_words_device().dev; }

int poll_error_word(const char* who, hipStream_t stream) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || !g_err_host[dev]) return OIVAE_OK;
  volatile uint32_t* w = g_err_host[dev];
  const uint32_t code = *w;
  if (code == 0) return OIVAE_OK;
  *w = 0;
  // the device-resident twin: lowered behind what is already queued on the caller's stream (the update kernels of
  // the failed step, if the host ran ahead, still find it raised)
  (void)hipMemsetAsync(g_tab[dev] + kBadWordOffsetU16, 0, 4, stream);
  return fail(OIVAE_E_HIP, "%s: an earlier decoder launch on device %d ran out of its bounded wait for an LDS hand-over "
              "(code %08x); the results of that step are invalid and its optimizer / proximal kernels did not run", who, dev, code);
}

const uint16_t* rsqrt14_table_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  return g_tab[dev];
}

int check_shape(const oivae_shape_t* s) {
  if (!s) return fail(OIVAE_E_ARG, "null shape");
  if (s->D < 1 || s->K < 1 || s->G < 1 || s->H < 1 || s->M < 0)
    return fail(OIVAE_E_SHAPE, "bad dims D=%d K=%d G=%d H=%d M=%d", s->D, s->K, s->G, s->H, s->M);
  if (s->gen_type < 0 || s->gen_type > 2) return fail(OIVAE_E_SHAPE, "bad gen_type %d", s->gen_type);
  if (s->gen_type == OIVAE_GEN_MLP && s->M < 1) return fail(OIVAE_E_SHAPE, "MLP generator needs M >= 1");
  if (s->gen_type != OIVAE_GEN_MLP && s->M != 0) return fail(OIVAE_E_SHAPE, "M must be 0 without a hidden layer");
  if (!s->group_off) return fail(OIVAE_E_ARG, "null group_off");
  if (s->max_group_dim < 1 || s->max_group_dim > s->D) return fail(OIVAE_E_SHAPE, "bad max_group_dim");
  if (s->K > 512) return fail(OIVAE_E_SHAPE, "K=%d > 512 unsupported", s->K);
  return OIVAE_OK;
}

int carve_workspace(const oivae_shape_t& s, int64_t B, int32_t S, void* base, int64_t bytes,
                    Workspace* w) {
  const int64_t N = B * (int64_t)S;
  DecPlan plan = plan_decoder(s, N);
  DecLayout L = dec_layout(s);
  char* p = static_cast<char*>(base);
  int64_t off = 0;
  auto take = [&](int64_t nbytes) -> char* {
    char* r = p ? p + off : nullptr;
    off += align_up(nbytes, 256);
    return r;
  };
  // every slot below n_* is written by exactly one block of its producer kernel on every step (no memset needed)
  {
    int64_t eb = (B * s.K + 255) / 256;
    w->n_kl = (int32_t)(eb > kEncEpiBlocks ? kEncEpiBlocks : eb);
  }
  w->kl_part = reinterpret_cast<double*>(take(sizeof(double) * kEncEpiBlocks));
  w->n_ll = plan.flat ? flat_decoder_blocks(s) : s.G * plan.chunks;
  w->ll_part = reinterpret_cast<double*>(take(sizeof(double) * kMaxPartials));
  w->n_bias = enc_bias_blocks(B);  // at least 16 rows per block
  w->bias_part = reinterpret_cast<double*>(take(enc_bias_part_bytes(kBiasBlocks, s.K)));
  w->n_theta = 2048;  // 512 blocks per generator parameter array; every slot written each step
  w->theta_part = reinterpret_cast<double*>(take(sizeof(double) * 2048));
  {
    int64_t pb = ((int64_t)s.G * s.K + 63) / 64;   // few passes per thread: the kernel is a chain of dependent loads
    w->n_pen = (int32_t)(pb > 1024 ? 1024 : pb);  // = penalty_kernel's grid: every slot written
  }
  w->pen_part = reinterpret_cast<double*>(take(sizeof(double) * 1024));
  w->enc_splits = x3_encoder_applies(s) ? x3_encoder_fwd_splits(s, B) : mfma_encoder_fwd_splits(s, B);
  w->dpre = reinterpret_cast<float*>(take(sizeof(float) * B * 2 * s.K * w->enc_splits));
  w->n_chunks = plan.chunks;
  w->p_dec = L.total;
  w->slab = reinterpret_cast<float*>(take(plan.chunks > 1 ? sizeof(float) * L.total * plan.chunks : 0));
  w->flat = reinterpret_cast<float*>(take(plan.flat ? sizeof(float) * flat_decoder_floats(s, N, nullptr) : 0));
  w->x3_img = reinterpret_cast<uint32_t*>(take(x3_encoder_ws_bytes(s, B)));
  w->ranges = reinterpret_cast<uint32_t*>(take(256));
  w->enc_bwd_part = reinterpret_cast<float*>(take(enc_bwd_part_bytes(s, B, x3_encoder_bwd_applies(s, B))));
  const bool z_img = plan.mfma && (s.K == 64 || dec_pad64_applies(s));   // (the padded shapes run with K = 64)
  w->x3_z = reinterpret_cast<uint32_t*>(take(z_img ? ((N + 31) / 32) * 24 * 1024 : 0));
  w->small = take(plan_small_step(s, B, S).ws_bytes);
  w->dec_zmax = reinterpret_cast<uint32_t*>(take(z_img ? 1024 : 0));
  {
    const int64_t a2 = plan.mfma ? mfma_decoder_acc2_bytes(s, N, plan.chunks) : 0;
    w->dec_acc2 = a2 > 0 ? reinterpret_cast<float*>(take(a2)) : nullptr;
  }
  {
    const int64_t pb = plan.mfma ? dec_pad64_bytes(s, N, plan.chunks) : 0;
    w->dec_pad = pb > 0 ? reinterpret_cast<float*>(take(pb)) : nullptr;
    w->dec_xhalf = 0;
  }
  w->total_bytes = off;
  if (p && bytes < off)
    return fail(OIVAE_E_WORKSPACE, "workspace too small: %lld < %lld bytes", (long long)bytes, (long long)off);
  return OIVAE_OK;
}

// ---- Adam ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float one_minus_b1, float b2,
                                                   float one_minus_b2, float step_size,
                                                   float inv_bc2_sqrt, float eps, float gscale,
                                                   const uint32_t* __restrict__ bad) {
  if (bad != nullptr && *bad != 0) return;   // a decoder launch of this step failed (common.h ErrWords): no update
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_update_dev(pi, g[i] * gscale, mi, vi, one_minus_b1, b2, one_minus_b2, step_size, inv_bc2_sqrt, eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// ---- weight prior + penalty + scalar combination ---------------------------------
// theta prior over the four generator parameter arrays (blockIdx.y): sum of theta^2 (double) and
// grad += theta/sigma^2.  Every block writes its partial slot (zero for an empty array): no memset.
struct ThetaSegs {
  const float* th[4];
  float* gr[4];
  int64_t n[4];
};

__device__ __forceinline__ void theta_prior_block(const ThetaSegs& sg, float inv_var, double* __restrict__ part,
                                                  int bx, int nbx, double* sm) {
  const int q = blockIdx.y;
  const float* __restrict__ th = sg.th[q];
  float* __restrict__ gr = sg.gr[q];
  const int64_t n = sg.n[q];
  double acc = 0.0;
  int64_t i = ((int64_t)bx * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)nbx * blockDim.x * 4;
  const bool vec = ((reinterpret_cast<uintptr_t>(th) | reinterpret_cast<uintptr_t>(gr)) & 15) == 0;
  for (; i < n; i += stride) {
    if (vec && i + 4 <= n) {
      const float4 x = *reinterpret_cast<const float4*>(th + i);
      float4 g = *reinterpret_cast<const float4*>(gr + i);
      acc += (double)x.x * x.x + (double)x.y * x.y + (double)x.z * x.z + (double)x.w * x.w;
      g.x += x.x * inv_var; g.y += x.y * inv_var; g.z += x.z * inv_var; g.w += x.w * inv_var;
      *reinterpret_cast<float4*>(gr + i) = g;
    } else {
      for (int64_t j = i; j < n && j < i + 4; ++j) {
        const float x = th[j];
        acc += (double)x * (double)x;
        gr[j] = gr[j] + x * inv_var;
      }
    }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(int64_t)q * nbx + bx] = sm[0];
}

// sum over rows of ||Ws[row, :]||_2.  L lanes per row (a power of two, 16 bytes per lane and pass) when rows are
// 16-byte aligned -- a wave then reads whole rows back to back instead of 64 rows 4 H bytes apart -- else one thread
// per row.
__device__ __forceinline__ void penalty_block(const float* __restrict__ Ws, int64_t rows, int H, int L,
                                              double* __restrict__ part, int bx, int nbx, double* sm) {
  double acc = 0.0;
  if (L > 0) {
    const int sub = threadIdx.x & (L - 1);
    int64_t r = ((int64_t)bx * blockDim.x + threadIdx.x) / L;
    const int64_t stride = (int64_t)nbx * blockDim.x / L;
    const int64_t rounds = (rows + stride - 1) / stride;   // every lane of a row group runs the same number of passes
    for (int64_t it = 0; it < rounds; ++it, r += stride) {
      float s = 0.f;
      if (r < rows) {
        const float4* w = reinterpret_cast<const float4*>(Ws + r * H);
        for (int h = sub; h < (H >> 2); h += L) {
          const float4 q = w[h];
          s += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
        }
      }
      for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (sub == 0 && r < rows) acc += (double)sqrtf(s);
    }
  } else {
    int64_t r = (int64_t)bx * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)nbx * blockDim.x;
    for (; r < rows; r += stride) {
      const float* w = Ws + r * H;
      float s = 0.f;
      for (int h = 0; h < H; ++h) s += w[h] * w[h];
      acc += (double)sqrtf(s);
    }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[bx] = sm[0];
}

// One launch for both (they were two; at an 8-GPU shard every launch is 5 us of a 2.4 ms step): rows 0..3 of the grid
// are the weight prior over the four generator parameter arrays, row 4 the penalty; a row uses the first nbx_* of the
// grid's columns.
__global__ __launch_bounds__(256) void prior_penalty_kernel(ThetaSegs sg, float inv_var, double* __restrict__ theta_part,
                                                            int nbx_theta, const float* __restrict__ Ws, int64_t rows,
                                                            int H, int L, double* __restrict__ pen_part, int nbx_pen) {
  __shared__ double sm[256];
  if (blockIdx.y < 4) {
    if ((int)blockIdx.x < nbx_theta) theta_prior_block(sg, inv_var, theta_part, blockIdx.x, nbx_theta, sm);
  } else if ((int)blockIdx.x < nbx_pen) {
    penalty_block(Ws, rows, H, L, pen_part, blockIdx.x, nbx_pen, sm);
  }
}

// sums the partial arrays in index order (deterministic) and writes the scalars
__global__ __launch_bounds__(256) void finalize_kernel(const double* kl, int n_kl, const double* ll, int n_ll,
                                                       const double* th, int n_th, const double* pen, int n_pen,
                                                       double n_theta_elems, float sigma_theta, float lam,
                                                       float B, float S, oivae_scalars_t* out) {
  __shared__ double sm[4][256];
  const double* src[4] = {kl, ll, th, pen};
  const int cnt[4] = {n_kl, n_ll, n_th, n_pen};
  for (int q = 0; q < 4; ++q) {   // eight loads in flight per thread (one by one, each cost a trip to memory); same order of addition
    double a = 0.0;
    for (int i0 = threadIdx.x; i0 < cnt[q]; i0 += 8 * 256) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (i0 + u * 256 < cnt[q]) ? src[q][i0 + u * 256] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    sm[q][threadIdx.x] = a;
  }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int q = 0; q < 4; ++q) sm[q][threadIdx.x] += sm[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float kl_sum = (float)sm[0][0];
    const float ll_sum = (float)sm[1][0];
    const double th_sq = sm[2][0];
    const float penalty = (float)sm[3][0];
    // lib/oivae.py:15-23: sum of -0.5*LOG2PI - log(sigma) - 0.5*(theta/sigma)^2
    const double st = (double)sigma_theta;
    const float lp_theta = (float)(n_theta_elems * (-0.5 * (double)kLog2Pi - log(st)) - 0.5 * th_sq / (st * st));
    const float z_kl = kl_sum / B;                 // lib/oivae.py:50
    const float loglik = (ll_sum / S) / B;         // lib/oivae.py:55-59
    const float lp_W = -lam * penalty;             // lib/oivae.py:65
    const float loss = -1.0f * (-z_kl + loglik + lp_theta);  // lib/oivae.py:70
    const float elbo = -loss + lp_W;               // lib/oivae.py:71
    out->z_kl = z_kl;
    out->loglik_term = loglik;
    out->logprob_theta = lp_theta;
    out->logprob_W = lp_W;
    out->loss = loss;
    out->elbo = elbo;
    out->penalty = penalty;
    out->pad = 0.f;
  }
}

}  // namespace oivae

using namespace oivae;

extern "C" const char* oivae_last_error(void) { return g_err; }
extern "C" const char* oivae_version(void) { return "oivae_hip 0.1 (gfx950)"; }

// ---- test hooks: read from the environment at first use, at each device's oivae_init and on request ----
namespace oivae {
static TestHooks read_test_hooks() {
  TestHooks h{};
  h.force_generic = getenv("OIVAE_FORCE_GENERIC") != nullptr;
  const char* e = getenv("OIVAE_ENC");
  h.enc = (e && (e[0] == 'b' || e[0] == 'f')) ? e[0] : 0;
  const char* d = getenv("OIVAE_DEC");
  h.dec[0] = d ? d[0] : 0;
  h.dec[1] = (d && d[0]) ? d[1] : 0;
  return h;
}
static TestHooks& hooks_storage() {
  static TestHooks h = read_test_hooks();
  return h;
}
const TestHooks& test_hooks() { return hooks_storage(); }
void reload_test_hooks() { hooks_storage() = read_test_hooks(); }
}  // namespace oivae

extern "C" void oivae_reload_test_hooks(void) { reload_test_hooks(); }

extern "C" int oivae_init(int device) {
  if (device < 0 || device >= kMaxDevices) return fail(OIVAE_E_ARG, "oivae_init: bad device %d", device);
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (g_tab[device]) return OIVAE_OK;
  const size_t nbytes = (size_t)(oivae_rsqrt14_blob_end - oivae_rsqrt14_blob);
  if (nbytes != 65536 * sizeof(uint16_t))
    return fail(OIVAE_E_ARG, "oivae_init: embedded rsqrt14 table has %zu bytes", nbytes);
  int prev = -1;
  OIVAE_HIP_CHECK(hipGetDevice(&prev));
  OIVAE_HIP_CHECK(hipSetDevice(device));
  uint16_t* d = nullptr;
  hipError_t e = hipMalloc(&d, nbytes + 256);   // + the device-resident error word (common.h ErrWords)
  if (e == hipSuccess) e = hipMemset(reinterpret_cast<char*>(d) + nbytes, 0, 256);
  if (e == hipSuccess) e = hipMemcpy(d, oivae_rsqrt14_blob, nbytes, hipMemcpyHostToDevice);
  int rc = OIVAE_OK;
  if (e != hipSuccess) rc = fail(OIVAE_E_HIP, "oivae_init: uploading the rsqrt14 table failed: %s", hipGetErrorString(e));
  if (rc == OIVAE_OK) rc = noalloc_scratch_setup(device);   // `device` is current here
  if (rc == OIVAE_OK && !g_err_host[device]) {
    void* h = nullptr;
    void* dp = nullptr;
    e = hipHostMalloc(&h, 64, hipHostMallocMapped);
    if (e == hipSuccess) { memset(h, 0, 64); e = hipHostGetDevicePointer(&dp, h, 0); }
    if (e != hipSuccess) rc = fail(OIVAE_E_HIP, "oivae_init: error word: %s", hipGetErrorString(e));
    else { g_err_host[device] = static_cast<uint32_t*>(h); g_err_dev[device] = static_cast<uint32_t*>(dp); }
  }
  if (rc == OIVAE_OK) { g_tab[device] = d; reload_test_hooks(); }
  else if (d) (void)hipFree(d);
  (void)hipSetDevice(prev);   // also on the error paths
  return rc;
}

extern "C" int oivae_check_errors(void* stream) {
  OIVAE_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return poll_error_word("oivae_check_errors", (hipStream_t)stream);
}

extern "C" int64_t oivae_workspace_bytes(const oivae_shape_t* shape, int64_t B, int32_t S) {
  if (check_shape(shape) != OIVAE_OK || B < 1 || S < 1) return -1;
  Workspace w;
  carve_workspace(*shape, B, S, nullptr, 0, &w);
  return w.total_bytes;
}

extern "C" int oivae_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                              float beta1, float beta2, float eps, int32_t step, float grad_scale,
                              void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return fail(OIVAE_E_ARG, "oivae_adam_f32: bad argument");
  if (n == 0) return OIVAE_OK;
  if (int rc = poll_error_word("oivae_adam_f32", (hipStream_t)stream)) return rc;
  // scalar prep in double like torch's Python-side _single_tensor_adam
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                     1.0f - beta1, beta2, 1.0f - beta2, step_size, inv_bc2_sqrt, eps, grad_scale, bad_word_device());
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

extern "C" int oivae_finalize_f32(const oivae_shape_t* shape, const oivae_tensors_t* params,
                                  const oivae_tensors_t* grads, int64_t B, int32_t S, float sigma_theta,
                                  float lam, oivae_scalars_t* out, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  if (int rc = check_shape(shape)) return rc;
  if (!params || !grads || !out || !workspace || B < 1 || S < 1)
    return fail(OIVAE_E_ARG, "oivae_finalize_f32: bad argument");
  if (!(sigma_theta > 0.f)) return fail(OIVAE_E_ARG, "oivae_finalize_f32: sigma_theta must be > 0");
  if (int rc = poll_error_word("oivae_finalize_f32", (hipStream_t)stream)) return rc;
  Workspace w;
  if (int rc = carve_workspace(*shape, B, S, workspace, workspace_bytes, &w)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const oivae_shape_t& s = *shape;
  const int64_t hin = (s.gen_type == OIVAE_GEN_MLP) ? s.M : s.H;
  // the four generator parameter arrays, each with a quarter of the partial slots
  const float* th[4] = {params->gen_W1, params->gen_b1, params->gen_W2, params->gen_b2};
  float* gr[4] = {grads->gen_W1, grads->gen_b1, grads->gen_W2, grads->gen_b2};
  const int64_t cnt[4] = {s.gen_type == OIVAE_GEN_MLP ? (int64_t)s.G * s.M * s.H : 0,
                          s.gen_type == OIVAE_GEN_MLP ? (int64_t)s.G * s.M : 0, (int64_t)s.D * hin,
                          (int64_t)s.D};
  const float inv_var = 1.0f / (sigma_theta * sigma_theta);
  double n_theta_elems = 0;
  ThetaSegs sg;
  for (int q = 0; q < 4; ++q) {
    if (cnt[q] != 0 && (!th[q] || !gr[q])) return fail(OIVAE_E_ARG, "oivae_finalize_f32: null generator tensor %d", q);
    sg.th[q] = th[q]; sg.gr[q] = gr[q]; sg.n[q] = cnt[q];
    n_theta_elems += (double)cnt[q];
  }
  const int64_t rows = (int64_t)s.G * s.K;
  int pen_lanes = 0;   // lanes per row: the largest power of two <= min(64, H / 4)
  if (s.H % 4 == 0 && (reinterpret_cast<uintptr_t>(params->Ws) & 15) == 0) {
    pen_lanes = 1;
    while (pen_lanes * 2 <= s.H / 4 && pen_lanes < 64) pen_lanes *= 2;
  }
  const int nbx_theta = w.n_theta / 4;
  hipLaunchKernelGGL(prior_penalty_kernel, dim3((unsigned)(nbx_theta > w.n_pen ? nbx_theta : w.n_pen), 5), dim3(256), 0, st,
                     sg, inv_var, w.theta_part, nbx_theta, params->Ws, rows, s.H, pen_lanes, w.pen_part, w.n_pen);
  OIVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, w.kl_part, w.n_kl, w.ll_part, w.n_ll,
                     w.theta_part, w.n_theta, w.pen_part, w.n_pen, n_theta_elems, sigma_theta, lam,
                     (float)B, (float)S, out);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}
