// The weight-gradient accumulation of dec_x3_64s in isolation: acc += split(g)^T split(t) over n tiles of 32 rows with
// the library's own split (x3_split.h), t constant over rows (a dead group: h = 0, t = tanh(b1)), g random of one sign.
// Against the exact sum in double.  Variants: RNE split as shipped; lo*lo term added; accumulate in two levels.
// hipcc -O3 --offload-arch=gfx950 -I oi_vae_amd/csrc tools/probe/h2_accum_bias.hip -o tools/probe/h2_accum_bias
#include "x3_split.h"
#include <stdio.h>
#include <math.h>
#include <vector>
using namespace oivae;
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mm(x3_u32x4 a, x3_u32x4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_h16x8, a), __builtin_bit_cast(x3_h16x8, b), c, 0, 0, 0);
}
// g: [n*32 rows][16 features], t: [16 features] (constant over rows).  out[16][16] = sum_rows g[r][d] t[m]
__global__ void probe(const float* g, const float* t, int n, int variant, float* out) {
  const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
  f4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  for (int s = 0; s < n; ++s) {
    float gv[8], tv[8];
    for (int e = 0; e < 8; ++e) { gv[e] = g[(s * 32 + 8 * q + e) * 16 + c]; tv[e] = t[c]; }
    const Frag2 G = split8_h2(gv[0], gv[1], gv[2], gv[3], gv[4], gv[5], gv[6], gv[7]);
    const Frag2 T = split8_h2(tv[0], tv[1], tv[2], tv[3], tv[4], tv[5], tv[6], tv[7]);
    if (variant == 1) acc = mm(G.p[1], T.p[1], acc);
    acc = mm(G.p[1], T.p[0], acc);
    acc = mm(G.p[0], T.p[1], acc);
    acc = mm(G.p[0], T.p[0], acc);
    if (variant == 2 && (s & 15) == 15) { acc2 += acc; acc = (f4){0, 0, 0, 0}; }
  }
  if (variant == 2) acc += acc2;
  for (int i = 0; i < 4; ++i) out[(4 * q + i) * 16 + c] = acc[i];
}
int main() {
  const int NMAX = 2048;
  std::vector<float> g((size_t)NMAX * 32 * 16), t(16);
  uint32_t st = 777;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (st >> 8) * (1.0f / 16777216.0f); };
  for (auto& x : g) x = (0.05f + rnd()) * 4096.0f;       // positive, scaled like gmu
  for (auto& x : t) x = (0.02f + 0.25f * rnd()) * 8192.0f;  // 2^13 t
  float *dg, *dt, *dout;
  hipMalloc(&dg, g.size() * 4); hipMalloc(&dt, 64); hipMalloc(&dout, 1024);
  hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt, t.data(), 64, hipMemcpyHostToDevice);
  const char* names[3] = {"as shipped (3 products)", "+ lo*lo (4 products)", "3 products, flushed every 16 tiles"};
  for (int variant = 0; variant < 3; ++variant)
    for (int n : {16, 128, 512, 1024, 2048}) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dg, dt, n, variant, dout);
      float o[256];
      hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
      double worst = 0, mean = 0;
      for (int d = 0; d < 16; ++d) {
        double sg = 0;
        for (int r = 0; r < n * 32; ++r) sg += g[(size_t)r * 16 + d];
        for (int m = 0; m < 16; ++m) {
          const double ex = sg * t[m], e = (o[d * 16 + m] - ex) / ex;
          mean += e / 256; if (fabs(e) > fabs(worst)) worst = e;
        }
      }
      printf("%-36s tiles=%5d  mean rel err %+.3e  worst %+.3e\n", names[variant], n, mean, worst);
    }
  return 0;
}
