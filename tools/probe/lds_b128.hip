// ds_read_b128 / ds_write_b64 cost against the lane -> address map: which lanes does the LDS serve in one cycle?
// One wave per SIMD (256 threads), every lane reads 16 bytes at dword address map(lane); cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

__global__ __launch_bounds__(256) void k(const int* addr, int wr, int iters, unsigned* out, long long* cyc) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned* p = lds + addr[lane];
  u32x4 s = {0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned ba = (unsigned)(addr[lane] * 4);
  for (int it = 0; it < iters; ++it) {
    if (wr == 2) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (u & 1) asm volatile("ds_write_b128 %0, %1 offset:32768" :: "v"(ba), "v"(s) : "memory");
        else asm volatile("ds_write_b128 %0, %1" :: "v"(ba), "v"(s) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (wr) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        u32x2 d = {s[0], s[1]};
        if (u & 1) asm volatile("ds_write_b64 %0, %1 offset:32768" :: "v"(ba), "v"(d) : "memory");
        else asm volatile("ds_write_b64 %0, %1" :: "v"(ba), "v"(d) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      u32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (u & 1) asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(v[u]) : "v"(ba) : "memory");
        else asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(ba) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static int* d_addr; static unsigned* o; static long long* c;
static void run(const char* what, int wr, int (*map)(int)) {
  int h[64];
  for (int l = 0; l < 64; ++l) h[l] = map(l);
  (void)hipMemcpy(d_addr, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, d_addr, wr, 500, o, c);
  (void)hipDeviceSynchronize();
  long long hc[256]; (void)hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += hc[i];
  printf("%-72s %6.1f cycles per instruction (4 waves per CU issuing)\n", what, s / 256 / (500.0 * 16));
}
int main() {
  (void)hipMalloc(&d_addr, 256); (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&c, 256 * 8);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run("read b128, lane-linear (16 B x lane)", 0, [](int l) { return 4 * l; });
  run("read b128, row = lane & 31, stride 28, half +4   (encoder today)", 0, [](int l) { return (l & 31) * 28 + 4 * (l >> 5); });
  run("read b128, row = lane & 31, stride 24, half +4", 0, [](int l) { return (l & 31) * 24 + 4 * (l >> 5); });
  run("read b128, row = lane & 31, stride 20, half +4", 0, [](int l) { return (l & 31) * 20 + 4 * (l >> 5); });
  run("read b128, row = lane & 31, stride 36, half +4", 0, [](int l) { return (l & 31) * 36 + 4 * (l >> 5); });
  run("read b128, row = lane & 31, stride 8 (two halves interleaved: 8 row + 4 half)", 0, [](int l) { return (l & 31) * 8 + 4 * (l >> 5); });
  run("read b128, row = lane & 31, stride 40, half +4", 0, [](int l) { return (l & 31) * 40 + 4 * (l >> 5); });
  run("read b128, row = lane & 31, stride 28, half +8", 0, [](int l) { return (l & 31) * 28 + 8 * (l >> 5); });
  run("read b128, row = lane & 31, stride 28, half +16", 0, [](int l) { return (l & 31) * 28 + 16 * (l >> 5); });
  run("read b128, all lanes one address", 0, [](int) { return 0; });
  run("read b128, stride 64 dwords (same bank)", 0, [](int l) { return (l & 31) * 64 + 4 * (l >> 5); });
  run("write b64, X producer today (2 rows x 8 cols per 16 lanes, sub-step pad 32)", 1, [](int l) { int xr = l >> 3, xc = l & 7; return (xc >> 2) * 3616 + xr * 28 + 4 * ((xc >> 1) & 1) + 2 * (xc & 1); });
  run("write b64, same with rows r, r+2 in a 16-lane group", 1, [](int l) { int xr = l >> 3, xc = l & 7; xr = (xr & ~3) | ((xr & 1) << 1) | ((xr >> 1) & 1); return (xc >> 2) * 3616 + xr * 28 + 4 * ((xc >> 1) & 1) + 2 * (xc & 1); });
  run("write b64, sub-step pad 8", 1, [](int l) { int xr = l >> 3, xc = l & 7; return (xc >> 2) * 3592 + xr * 28 + 4 * ((xc >> 1) & 1) + 2 * (xc & 1); });
  run("write b64, sub-step pad 16", 1, [](int l) { int xr = l >> 3, xc = l & 7; return (xc >> 2) * 3600 + xr * 28 + 4 * ((xc >> 1) & 1) + 2 * (xc & 1); });
  run("write b64, sub-step pad 0", 1, [](int l) { int xr = l >> 3, xc = l & 7; return (xc >> 2) * 3584 + xr * 28 + 4 * ((xc >> 1) & 1) + 2 * (xc & 1); });
  run("write b64, one row per lane, stride 28 (backward X producer)", 1, [](int l) { return (l & 31) * 28 + 2 * (l >> 5); });
  run("write b64, lane-linear", 1, [](int l) { return 2 * l; });
  run("write b128, lane-linear", 2, [](int l) { return 4 * l; });
  run("write b128, W producer today (12 chunks per row, pad 32)", 2, [](int l) { int row = l / 12, ch = l % 12; return row * 28 + (ch >= 6 ? 3616 : 0) + (ch % 6) * 4; });
  run("write b128, one row per lane, stride 28", 2, [](int l) { return (l & 31) * 28 + 4 * (l >> 5); });
  run("write b64, 4 rows x 4 col-pairs per 16 lanes (xc low 2 bits)", 1, [](int l) { int xr = l >> 2, xc = l & 3; return xr * 28 + 4 * ((xc >> 1) & 1) + 2 * (xc & 1); });
  return 0;
}
