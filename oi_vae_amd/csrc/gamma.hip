// Gamma log-likelihood of non-negative data, forward and gradients in one pass.
//
// Replaces Gamma.logprob (/root/reference/lib/distributions.py:65-71): the sum over all elements of
//   shape * log(rate) - lgamma(shape) + (shape - 1) * log(x) - rate * x
// and, when the gradient pointers are given, what autograd derives from that expression:
//   d/dshape = log(rate) - digamma(shape) + log(x),   d/drate = shape / rate - x,   d/dx = (shape - 1) / x - rate.
// The reference evaluates the expression with ~10 elementwise torch kernels and as many temporaries of the data's
// size; here the three inputs are read once and up to three gradients written once (HBM-bound: 12 B read + 12 B written
// per element).  Each element's four terms are formed and added in fp32 in the reference's order (this translation
// unit is built with -ffp-contract=off), so special values come out as there: x = 0 with shape = 1 is 0 * log 0 = NaN,
// not the 0 torch.distributions' xlogy would give.  The sum over elements is accumulated in fp64 in a fixed order
// (per thread, per workgroup, then one workgroup over the partials): reproducible, and within fp32 rounding of the
// reference's fp32 pairwise sum.
//
// Scope: the reference's training step cannot run with Gamma members -- _DistributionCat.logprob needs
// `size`, which the reference's Gamma lacks (lib/distributions.py:82) -- so this is the evaluation-side operator,
// reached from oi_vae_amd.distributions.Gamma.logprob on device tensors, not an epilogue of the fused step.
#include "common.h"

namespace oivae {

constexpr int kGammaThreads = 256;
constexpr int kGammaMaxBlocks = 1024;

// digamma(a) for a > 0 in fp64: upward recurrence to a >= 6, then the asymptotic series (error < 1e-12 there)
__device__ __forceinline__ double digamma_pos(double a) {
  double r = 0.0;
  while (a < 6.0) { r -= 1.0 / a; a += 1.0; }
  const double i = 1.0 / a, i2 = i * i;
  const double series = i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0 - i2 * (1.0 / 132.0)))));
  return r + log(a) - 0.5 * i - series;
}

__global__ __launch_bounds__(kGammaThreads) void gamma_logprob_kernel(const float* __restrict__ shape, const float* __restrict__ rate,
                                                                     const float* __restrict__ x, int64_t n, double* __restrict__ part,
                                                                     float* __restrict__ d_shape, float* __restrict__ d_rate,
                                                                     float* __restrict__ d_x) {
  __shared__ double red[kGammaThreads];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kGammaThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kGammaThreads) {
    const float a = shape[i], b = rate[i], v = x[i];
    const float lb = logf(b), lv = logf(v);
    // the reference's association: ((a log b - lgamma a) + (a - 1) log x) - b x, every operation rounded to fp32
    float t = a * lb;
    t = t - lgammaf(a);
    t = t + (a - 1.0f) * lv;
    t = t - b * v;
    acc += (double)t;
    if (d_shape) d_shape[i] = (a > 0.f) ? (float)((double)lb - digamma_pos((double)a) + (double)lv) : __builtin_nanf("");
    if (d_rate) d_rate[i] = a / b - v;
    if (d_x) d_x[i] = (a - 1.0f) / v - b;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kGammaThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(kGammaThreads) void gamma_logprob_final(const double* __restrict__ part, int nparts, float* __restrict__ out) {
  __shared__ double red[kGammaThreads];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kGammaThreads) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kGammaThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)red[0];
}

}  // namespace oivae

using namespace oivae;

extern "C" int64_t oivae_gamma_workspace_bytes(void) { return (int64_t)sizeof(double) * kGammaMaxBlocks; }

extern "C" int oivae_gamma_logprob_f32(const float* shape, const float* rate, const float* x, int64_t n, float* logprob,
                                       float* d_shape, float* d_rate, float* d_x, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
  if (!logprob || n < 0 || (n > 0 && (!shape || !rate || !x)))
    return fail(OIVAE_E_ARG, "oivae_gamma_logprob_f32: null pointer or negative size");
  if (!workspace || workspace_bytes < oivae_gamma_workspace_bytes() || (reinterpret_cast<uintptr_t>(workspace) & 7))
    return fail(OIVAE_E_WORKSPACE, "oivae_gamma_logprob_f32: workspace of %lld bytes (8-byte aligned) needed",
                (long long)oivae_gamma_workspace_bytes());
  hipStream_t st = (hipStream_t)stream;
  if (int rc = poll_error_word("oivae_gamma_logprob_f32", st)) return rc;
  int64_t blocks = (n + kGammaThreads * 4 - 1) / (kGammaThreads * 4);   // about four elements per thread
  if (blocks < 1) blocks = 1;
  if (blocks > kGammaMaxBlocks) blocks = kGammaMaxBlocks;
  double* part = static_cast<double*>(workspace);
  hipLaunchKernelGGL(gamma_logprob_kernel, dim3((unsigned)blocks), dim3(kGammaThreads), 0, st, shape, rate, x, n, part, d_shape,
                     d_rate, d_x);
  OIVAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(gamma_logprob_final, dim3(1), dim3(kGammaThreads), 0, st, (const double*)part, (int)blocks, logprob);
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}
