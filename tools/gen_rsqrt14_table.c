/* Tabulate the x86 VRSQRT14SS instruction (AVX-512F) over one even and one odd
 * binade.  Why: the reference's proximal step calls torch.sqrt
 * (/root/reference/lib/models.py:107), which on the reference's CPU path is
 * Intel MKL VML vmsSqrt (HA mode); on AVX-512 hosts that routine is
 *     y = rsqrt14(x); s = x*y; h = 0.5*y; r = fma(fma(-s, s, x), h, s)
 * (established empirically, 0 mismatches on 2e6 inputs), which is NOT the
 * correctly rounded root for ~0.6 % of inputs.  To reproduce the reference's
 * row norms bit for bit on the GPU we need rsqrt14 itself.  Its result is
 * architecturally defined (identical on every AVX-512 CPU) and depends only on
 * the exponent parity and the top 15 mantissa bits, plus the special case of an
 * exact power of four -> exact power of two.
 *
 * Output: 65536 uint16 entries.  entry[parity*32768 + (mantissa>>8)] = the top
 * 16 mantissa bits of rsqrt14(x) for x in [1,2) (parity 0) or [2,4) (parity 1);
 * the result's exponent is 126 (value in [0.5,1)) for every entry.  The lone
 * exception rsqrt14(1.0) = 1.0 is handled in code, not in the table.
 *
 * Build+run (AVX-512 host only):
 *   gcc -O2 -mavx512f -mavx512vl tools/gen_rsqrt14_table.c -o /tmp/gen && /tmp/gen oi_vae_amd/data/rsqrt14_u16.bin
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static uint32_t r14(uint32_t u) {
  float x; memcpy(&x, &u, 4);
  __m128 v = _mm_set_ss(x);
  v = _mm_rsqrt14_ss(v, v);
  float r = _mm_cvtss_f32(v);
  uint32_t ru; memcpy(&ru, &r, 4);
  return ru;
}

int main(int argc, char** argv) {
  static uint16_t tab[65536];
  for (int par = 0; par < 2; par++) {
    uint32_t base = par ? 0x40000000u : 0x3f800000u;
    for (uint32_t i = 0; i < 32768; i++) {
      /* representative: low bits 0x80 (never the exact power of four) */
      uint32_t r = r14(base | (i << 8) | 0x80u);
      if ((r >> 23) != 126 || (r & 0x7f)) { fprintf(stderr, "unexpected shape at %u/%u: %08x\n", par, i, r); return 1; }
      tab[par * 32768 + i] = (uint16_t)((r & 0x7fffff) >> 7);
      /* every input of the bucket must agree (except exact 1.0) */
      for (uint32_t lo = 0; lo < 256; lo += 51) {
        if (par == 0 && i == 0 && lo == 0) continue;
        if (r14(base | (i << 8) | lo) != r) { fprintf(stderr, "bucket not constant at %u/%u\n", par, i); return 1; }
      }
    }
  }
  if (r14(0x3f800000u) != 0x3f800000u) { fprintf(stderr, "rsqrt14(1) != 1\n"); return 1; }
  FILE* f = fopen(argc > 1 ? argv[1] : "rsqrt14_u16.bin", "wb");
  fwrite(tab, 2, 65536, f);
  fclose(f);
  return 0;
}
