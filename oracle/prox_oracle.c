/* CPU restatement of the reference's proximal step -- TEST INFRASTRUCTURE ONLY
 * (checker for the HIP kernel and the cpu_baseline leg of bench.py; the product
 * never links or loads this).
 *
 * Restates /root/reference/lib/models.py:106-111:
 *     n  = sqrt(sum(pow(Ws, 2), dim=2, keepdim=True))
 *     Ws = Ws / clamp(n, min=1e-16)
 *     Ws = Ws * clamp(n - t, min=0)
 * with the arithmetic of the reference's CPU path spelled out so the result is
 * the same bits on any host:
 *   - pow(w,2) = w*w rounded to fp32; the row sum in the order of torch 2.10's
 *     CPU kernel (see row_sumsq below);
 *   - sqrt = Intel MKL VML vmsSqrt(HA) as torch 2.10 calls it on an AVX-512
 *     host: y = rsqrt14(x); s = x*y; h = y/2; r = fma(fma(-s,s,x), h, s), with
 *     rsqrt14 restated from a table (tools/gen_rsqrt14_table.c);
 *   - divide and multiply correctly rounded, two roundings per element.
 * Pinned by tests/golden/prox_vectors.npz and the pre_prox/post Ws pairs of
 * every tests/golden/c*.npz (all recorded from the real reference).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared oracle/prox_oracle.c -o oracle/_build/libprox_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* VRSQRT14SS restated.  tab: 65536 uint16 (tools/gen_rsqrt14_table.c). */
float oivae_oracle_rsqrt14(float x, const uint16_t* tab) {
  uint32_t u = f2u(x);
  if (x != x) return x;
  if ((u << 1) == 0) return u2f((u & 0x80000000u) | 0x7f800000u); /* +-0 -> +-inf */
  if (u >> 31) return u2f(0xffc00000u);                            /* negative -> NaN */
  if (u == 0x7f800000u) return 0.0f;
  int e = (int)(u >> 23) - 127;
  uint32_t m = u & 0x7fffffu;
  if ((u >> 23) == 0) { /* denormal: normalise */
    e = -126;
    while (!(m & 0x800000u)) { m <<= 1; e--; }
    m &= 0x7fffffu;
  }
  int par = e & 1;          /* x = 2^(e-par) * [1,4) */
  int half = (e - par) / 2; /* exact: e-par is even */
  if (par == 0 && m == 0) return ldexpf(1.0f, -half);
  uint32_t frac = tab[par * 32768 + (m >> 8)];
  float r = u2f((126u << 23) | (frac << 7)); /* in [0.5,1) */
  return ldexpf(r, -half);
}

/* MKL VML vmsSqrt (HA) on AVX-512 as torch.sqrt reaches it. */
float oivae_oracle_mkl_sqrtf(float x, const uint16_t* tab) {
  if (x != x || x == 0.0f || x == INFINITY) return x;
  if (x < 0.0f) return u2f(0xffc00000u);
  /* Below 2^-104 the residual fma would go denormal; MKL rescales by an even
   * power of two there (any even power >= 2^24 reproduces it, denormal inputs
   * included), i.e. the formula is evaluated free of exponent-range effects. */
  float back = 1.0f;
  if (x < 0x1p-100f) { x *= 0x1p64f; back = 0x1p-32f; }
  float y = oivae_oracle_rsqrt14(x, tab);
  float s = x * y;
  float h = 0.5f * y;
  float d = fmaf(-s, s, x);
  return fmaf(d, h, s) * back;
}

void oivae_oracle_mkl_sqrtf_array(const float* in, float* out, long n, const uint16_t* tab) {
  for (long i = 0; i < n; i++) out[i] = oivae_oracle_mkl_sqrtf(in[i], tab);
}

void oivae_oracle_rsqrt14_array(const float* in, float* out, long n, const uint16_t* tab) {
  for (long i = 0; i < n; i++) out[i] = oivae_oracle_rsqrt14(in[i], tab);
}

/* sum_h w[h]^2 in torch 2.10's CPU order for a contiguous fp32 inner dim
 * (aten SumKernel: vectorized_inner_sum -> row_sum -> multi_row_sum, 8-lane vectors,
 * 4 interleaved vector accumulators = one "step" of 32 floats).  From 16 steps on
 * (H >= 512) multi_row_sum cascades: every 16 steps the level-0 accumulators are added
 * into level 1 and cleared, every 256 steps level 1 into level 2, every 4096 level 2
 * into level 3 (level_power = max(4, ceil_log2(steps) / 4) = 4 below 2^20 steps);
 * the steps left over after the last full block of 16 go to level 0, then levels
 * 1..3 are added to level 0 in that order. */
float oivae_oracle_row_sumsq(const float* w, int H) {
  if (H < 8) {
    float p[4] = {0, 0, 0, 0};
    int nq = H / 4;
    for (int i = 0; i < nq; i++)
      for (int k = 0; k < 4; k++) p[k] = p[k] + w[4 * i + k] * w[4 * i + k];
    for (int i = nq * 4; i < H; i++) p[0] = p[0] + w[i] * w[i];
    for (int k = 1; k < 4; k++) p[0] = p[0] + p[k];
    return p[0];
  }
  float P[4][4][8];   /* [level][accumulator][lane] */
  memset(P, 0, sizeof(P));
  int V = H / 8, nq = V / 4;
  int i = 0;
  while (i + 16 <= nq) {
    for (int j = 0; j < 16; j++, i++)
      for (int k = 0; k < 4; k++)
        for (int l = 0; l < 8; l++) { float a = w[(4 * i + k) * 8 + l]; P[0][k][l] = P[0][k][l] + a * a; }
    for (int lev = 1; lev < 4; lev++) {
      for (int k = 0; k < 4; k++)
        for (int l = 0; l < 8; l++) { P[lev][k][l] = P[lev][k][l] + P[lev - 1][k][l]; P[lev - 1][k][l] = 0.0f; }
      if ((i & (15 << (4 * lev))) != 0) break;
    }
  }
  for (; i < nq; i++)
    for (int k = 0; k < 4; k++)
      for (int l = 0; l < 8; l++) { float a = w[(4 * i + k) * 8 + l]; P[0][k][l] = P[0][k][l] + a * a; }
  for (int lev = 1; lev < 4; lev++)
    for (int k = 0; k < 4; k++)
      for (int l = 0; l < 8; l++) P[0][k][l] = P[0][k][l] + P[lev][k][l];
  for (int v = nq * 4; v < V; v++)
    for (int l = 0; l < 8; l++) { float a = w[v * 8 + l]; P[0][0][l] = P[0][0][l] + a * a; }
  for (int k = 1; k < 4; k++)
    for (int l = 0; l < 8; l++) P[0][0][l] = P[0][0][l] + P[0][k][l];
  float acc = 0.0f;
  for (int h = V * 8; h < H; h++) acc = acc + w[h] * w[h];
  for (int l = 0; l < 8; l++) acc = acc + P[0][0][l];
  return acc;
}

/* In-place proximal step over Ws[rows][H]; mask[r] = 1 iff the row is all zero
 * afterwards; norms_out (optional) receives the pre-update row norms.
 * Returns 0, or 1 if H is outside the pinned range. */
int oivae_oracle_prox_f32(float* Ws, long rows, int H, float t, const uint16_t* tab,
                          uint8_t* mask, float* norms_out) {
  if (H < 1 || H > (1 << 20)) return 1;
  for (long r = 0; r < rows; r++) {
    float* w = Ws + r * (long)H;
    float n = oivae_oracle_mkl_sqrtf(oivae_oracle_row_sumsq(w, H), tab);
    float den = n < 1e-16f ? 1e-16f : n;   /* clamp(n, min=1e-16); NaN passes through */
    if (n != n) den = n;
    float sc = n - t;
    if (sc < 0.0f) sc = 0.0f;              /* clamp(n - t, min=0); NaN passes through */
    int allzero = 1;
    for (int h = 0; h < H; h++) {
      float v = w[h] / den;
      v = v * sc;
      w[h] = v;
      if (!(v == 0.0f)) allzero = 0;
    }
    if (mask) mask[r] = (uint8_t)allzero;
    if (norms_out) norms_out[r] = n;
  }
  return 0;
}

/* ---------------------------------------------------------------------------
 * FirstLayerSparseDecoder.proximal_step, /root/reference/lib/models.py:65-71:
 * per latent-to-group map W_g [H_g, K] (nn.Linear(K, H_g).weight):
 *     n = sqrt(sum(pow(W_g, 2), dim=0));  W_g /= clamp(n, 1e-16);  W_g *= clamp(n - t, 0)
 * The column sum over dim 0 follows torch 2.10's CPU outer-reduction kernel
 * (aten SumKernel: vectorized_outer_sum / scalar_outer_sum), probed in the build
 * container and pinned by tests/golden/fls_prox_vectors.npz:
 *   - columns are handled in chunks of 32 (4 vectors of 8) while 32 remain -- for
 *     K < 8 in chunks of 4 -- by multi_row_sum: rows added in order into one
 *     accumulator per column, flushed into a second-level accumulator every 16 rows;
 *   - the remaining columns by row_sum: four interleaved accumulators over rows
 *     (row r -> accumulator r mod 4, each a multi_row_sum over H/4 terms), leftover
 *     rows into accumulator 0, then accumulators 1..3 added in order;
 *   - K == 1 is a contiguous inner reduction: the order of oivae_oracle_row_sumsq.
 * Pinned for H_g*K < 32768 (beyond it torch splits the columns over its threads, and which columns take which of
 * the two paths above then depends on the host's thread count) and, for K == 1, H_g < 512.
 * ------------------------------------------------------------------------- */
static float multi_row_sumsq(const float* w, long stride, int n) {
  /* aten multi_row_sum, level step 16 for n < 2^20: level 0 is flushed into level 1 every 16 terms, level 1 into
   * level 2 every 256, level 2 into level 3 every 4096.  (Until round 2's last part this kept two levels only, which
   * is the same sum below 272 terms -- the recorded vectors went to 255 -- and a different one from there on.) */
  float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  int i = 0;
  while (i + 16 <= n) {
    for (int j = 0; j < 16; j++, i++) { float v = w[(long)i * stride]; a[0] = a[0] + v * v; }
    for (int lev = 1; lev < 4; lev++) {
      a[lev] = a[lev] + a[lev - 1];
      a[lev - 1] = 0.0f;
      if ((i & (15 << (4 * lev))) != 0) break;
    }
  }
  for (; i < n; i++) { float v = w[(long)i * stride]; a[0] = a[0] + v * v; }
  a[0] = a[0] + a[1];
  a[0] = a[0] + a[2];
  a[0] = a[0] + a[3];
  return a[0];
}

static float row_sum_sumsq(const float* w, long stride, int n) {
  int n4 = n / 4;
  float p[4];
  for (int k = 0; k < 4; k++) p[k] = multi_row_sumsq(w + (long)k * stride, 4 * stride, n4);
  for (int i = 4 * n4; i < n; i++) { float v = w[(long)i * stride]; p[0] = p[0] + v * v; }
  for (int k = 1; k < 4; k++) p[0] = p[0] + p[k];
  return p[0];
}

float oivae_oracle_col_sumsq(const float* W, int H, int K, int k) {
  if (K == 1) {
    float tmp[512];
    for (int h = 0; h < H; h++) tmp[h] = W[h];
    return oivae_oracle_row_sumsq(tmp, H);
  }
  int n_seq = (K >= 8) ? (K / 32) * 32 : (K / 4) * 4;
  return (k < n_seq) ? multi_row_sumsq(W + k, K, H) : row_sum_sumsq(W + k, K, H);
}

/* W: the G maps stacked on rows, [sum H_g, K]; group_off: G+1 prefix sums of H_g.
 * mask/norms [G, K].  Returns 0, or 1 if a shape is outside the pinned range. */
int oivae_oracle_prox_cols_f32(float* W, const int32_t* group_off, int G, int K, float t,
                               const uint16_t* tab, uint8_t* mask, float* norms_out) {
  for (int g = 0; g < G; g++) {
    int H = group_off[g + 1] - group_off[g];
    if (H < 1 || (long)H * K >= 32768 || (K == 1 && H >= 512)) return 1;
  }
  for (int g = 0; g < G; g++) {
    float* Wg = W + (long)group_off[g] * K;
    int H = group_off[g + 1] - group_off[g];
    for (int k = 0; k < K; k++) {
      float n = oivae_oracle_mkl_sqrtf(oivae_oracle_col_sumsq(Wg, H, K, k), tab);
      float den = n < 1e-16f ? 1e-16f : n;
      if (n != n) den = n;
      float sc = n - t;
      if (sc < 0.0f) sc = 0.0f;
      int allzero = 1;
      for (int h = 0; h < H; h++) {
        float v = Wg[(long)h * K + k] / den;
        v = v * sc;
        Wg[(long)h * K + k] = v;
        if (!(v == 0.0f)) allzero = 0;
      }
      if (mask) mask[(long)g * K + k] = (uint8_t)allzero;
      if (norms_out) norms_out[(long)g * K + k] = n;
    }
  }
  return 0;
}
