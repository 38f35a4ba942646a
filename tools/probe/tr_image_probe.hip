// Probe: can the chain decoder's FORWARD weight image (lane = output row m, 8 slots = k in accumulator order, two
// fp16 terms) also serve the backward product (lane = k, slots = m) through ds_read_b64_tr_b16?  Builds both images
// of a random 64 x 64 matrix the way round 1's dec_x3_64h did and compares, for every (rb, ks, term), the backward fragment
// read with one ds_read_b128 from the transposed image against two transposed reads from the forward image.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/tr_image_probe.hip -o tools/probe/tr_image_probe && tools/probe/tr_image_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ int slot_feature(int ks, int hh, int e) { return 16 * ks + 8 * (e >> 2) + 4 * hh + (e & 3); }
__device__ void split2(float a, float b, uint32_t& h, uint32_t& l) {
  const h16x2 hv = __builtin_convertvector(f32x2{a, b}, h16x2);
  const f32x2 hf = __builtin_convertvector(hv, f32x2);
  const h16x2 lv = __builtin_convertvector(f32x2{a - hf[0], b - hf[1]}, h16x2);
  h = __builtin_bit_cast(uint32_t, hv); l = __builtin_bit_cast(uint32_t, lv);
}
__global__ __launch_bounds__(512) void probe(const float* A, uint32_t* out_old, uint32_t* out_new) {
  __shared__ __attribute__((aligned(16))) uint32_t img[2 * 4096];   // [0] forward image of A, [1] image of A^T
  const int tid = threadIdx.x;
  {
    const int r = tid >> 3, ks = (tid >> 1) & 3, h2 = tid & 1;
    for (int q = 0; q < 2; ++q) {
      float v[8];
      for (int e = 0; e < 8; ++e) { const int c = slot_feature(ks, h2, e); v[e] = q == 0 ? A[r * 64 + c] : A[c * 64 + r]; }
      u32x4 H, L;
      for (int e = 0; e < 4; ++e) { uint32_t h, l; split2(v[2 * e], v[2 * e + 1], h, l); H[e] = h; L[e] = l; }
      *reinterpret_cast<u32x4*>(img + q * 4096 + ((((r >> 5) * 4 + ks) * 2 + 0) * 64 + h2 * 32 + (r & 31)) * 4) = H;
      *reinterpret_cast<u32x4*>(img + q * 4096 + ((((r >> 5) * 4 + ks) * 2 + 1) * 64 + h2 * 32 + (r & 31)) * 4) = L;
    }
  }
  __syncthreads();
  if (tid >= 64) return;
  const int lane = tid, gidx = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3, hh = gidx >> 1;
  for (int rb = 0; rb < 2; ++rb)
    for (int ks = 0; ks < 4; ++ks)
      for (int p = 0; p < 2; ++p) {
        const u32x4 old = *reinterpret_cast<const u32x4*>(img + 4096 + ((rb * 4 + ks) * 2 + p) * 256 + lane * 4);
        u32x4 nw;
        for (int j = 0; j < 2; ++j) {
          const int m = 16 * ks + 8 * j + 4 * hh + qq;
          const int ksk = 2 * rb + (gidx & 1);
          const int byte = (((m >> 5) * 4 + ksk) * 2 + p) * 1024 + ((pp & 1) * 32 + (m & 31)) * 16 + (pp >> 1) * 8;
          const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(reinterpret_cast<char*>(img) + byte));
          nw[2 * j] = (uint32_t)(uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16);
          nw[2 * j + 1] = (uint32_t)(uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
        }
        const int o = (((rb * 4 + ks) * 2 + p) * 64 + lane) * 4;
        for (int e = 0; e < 4; ++e) { out_old[o + e] = old[e]; out_new[o + e] = nw[e]; }
      }
}
int main() {
  const int n = 2 * 4 * 2 * 64 * 4;
  float hA[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) hA[i] = 8192.0f * ((float)rand() / RAND_MAX * 2 - 1);
  float* dA; uint32_t *d0, *d1;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, dA, d0, d1);
  static uint32_t h0[n], h1[n];
  hipMemcpy(h0, d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (h0[i] != h1[i]) { if (bad < 8) printf("mismatch at %d: %08x vs %08x\n", i, h0[i], h1[i]); ++bad; }
  printf("tr_image_probe: %d of %d dwords differ -> %s\n", bad, n, bad ? "FAIL" : "ok");
  return bad != 0;
}
