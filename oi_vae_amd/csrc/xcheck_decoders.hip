// Earlier generations of the all-64 grouped decoder (K = H = M = D_g = 64), kept OUT of the default path:
//   dec_mfma64_v4  OIVAE_DEC=v4   fp32 matrix instruction, first two linear maps composed
//   dec_x3_64      OIVAE_DEC=x3   three bf16 terms per operand (six products), one wave per SIMD
//   dec_x3_64p     OIVAE_DEC=p    the same arithmetic, chain wave + weight-gradient wave per SIMD; also what a launch
//                                 falls back to when dz exceeds 32-bit byte offsets or the parameters are unaligned
//   dec_x3_64g     OIVAE_DEC=g    two fp16 terms, 32x32x16 instructions, two groups per workgroup (round 2's default
//                                 until dec_x3_64s moved it to 16x16x32 instructions)
// Same mathematics as dec_x3_64s on other instructions and operand widths: the parity tests run several of them
// against the oracle and against each other (tests/test_gpu_parity.py, test_gpu_x3.py, test_gpu_scale.py).  Their
// design notes and measurements: HISTORY.md 3.0.
#include "dec_shared.h"

namespace oivae {

// ==============================================================================
// v4 of the all-64 kernel: v3's six-product data flow with the instruction stream
// rebuilt around one measured fact (tools/probe/mfma_probe2.hip on MI355X): next to
// v_mfma_f32_32x32x2_f32 every vector-ALU instruction of the same wave costs ~8
// cycles of matrix time (a transcendental ~12), packed f32 forms cost the same as
// scalar ones, while scalar ALU, s_waitcnt, s_nop and LDS reads cost ~0-2.  fp32
// "matrix" work shares the vector ALU, so the lever is the NUMBER of VALU
// instructions per tile (v3: 910, of which 256 were AGPR<->VGPR copies):
//   * accumulators that element-wise code touches (a1/t, mu/gmu, dt/da1, dz) use the
//     VGPR form of the MFMA (inline asm: the compiler only emits the AGPR form in a
//     512-register kernel), so no v_accvgpr copies; the two weight-gradient
//     accumulators stay in AGPRs through the builtin;
//   * biases are folded into the accumulator's initial value (loaded from LDS),
//     the tanh argument scale 2*log2(e) into a second copy of A;
//   * all element-wise math is written on register pairs (v_pk_*_f32);
//   * column sums are accumulated as pairs and folded once at the end;
//   * tile bookkeeping and the atomics' row bases are scalar (wave index through
//     readfirstlane, 32-bit per-lane offsets).
// Hazards of the asm MFMAs (the compiler does not pad inline asm): every asm MFMA is
// preceded by s_nop 1 (VALU-written operand -> MFMA), and every consumer of an asm
// MFMA's result is placed at least two 64-cycle MFMAs later in program order,
// pinned by sched_barrier(0) after every slot.
// ==============================================================================
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ void mfv(f32x16& acc, float a_, float b_) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a_), "v"(b_));
}
__device__ __forceinline__ void mfv0(f32x16& acc, float a_, float b_) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a_), "v"(b_));
}

// AGPR form (weight-gradient accumulators, never touched by element-wise code).  Also inline asm: with no
// MFMA visible to it, the compiler's gfx950 peephole that splits v_pk_*_f32 "in the shadow of an MFMA" into two
// scalar VALU ops (right for bf16 MFMAs, wrong beside fp32 ones) leaves the packed element-wise math alone.
__device__ __forceinline__ void mfa(f32x16& acc, float a_, float b_) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a_), "v"(b_));
}

// The three phase shapes below share one convention: the operands of a phase's first k-steps arrive in `w0`,
// already in flight (issued two slots before the previous phase ended), and each phase issues the first operands
// of the NEXT phase (`nf(i)`, i < NN, into `wn`) two slots before its own end, so that no phase starts by waiting
// for an LDS round trip.
//
// weight-gradient step on two AGPR tiles that share the A operand: acc0 += af(s)*bf0(s), acc1 += af(s)*bf1(s)
// w0 = {af(0), bf0(0), bf1(0), af(1), bf0(1), bf1(1)}
template <int NK, int NN, class AF, class BF0, class BF1, class VF, class NF>
__device__ __forceinline__ void aphasewg(f32x16& acc0, f32x16& acc1, const float (&w0)[6], AF af, BF0 bf0, BF1 bf1, VF vf,
                                         NF nf, float (&wn)[NN]) {
  float wa[NK], wb0[NK], wb1[NK];
  wa[0] = w0[0]; wb0[0] = w0[1]; wb1[0] = w0[2]; wa[1] = w0[3]; wb0[1] = w0[4]; wb1[1] = w0[5];
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    if (s + 2 < NK) { wa[s + 2] = af(s + 2); wb0[s + 2] = bf0(s + 2); wb1[s + 2] = bf1(s + 2); }
    if (s == NK - 2) {
#pragma unroll
      for (int i = 0; i < NN; ++i) wn[i] = nf(i);
    }
    vf(2 * s);
    mfa(acc0, wa[s], wb0[s]);
    vf(2 * s + 1);
    mfa(acc1, wa[s], wb1[s]);
    OIVAE_SB();
  }
}

// one output tile held in VGPRs, NK k-steps.  w0 = {wf(0), wf(1), wf(2)}
template <int NK, bool FRESH, int NN, class WF, class BF, class VF, class NF>
__device__ __forceinline__ void vphase1(f32x16& acc, const float (&w0)[3], WF wf, BF bf, VF vf, NF nf, float (&wn)[NN]) {
  float w[NK];
  w[0] = w0[0]; w[1] = w0[1]; w[2] = w0[2];
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    if (s + 3 < NK) w[s + 3] = wf(s + 3);
    if (s == NK - 2) {
#pragma unroll
      for (int i = 0; i < NN; ++i) wn[i] = nf(i);
    }
    vf(s);
    if (FRESH && s == 0) mfv0(acc, w[s], bf(s)); else mfv(acc, w[s], bf(s));
    OIVAE_SB();
  }
}

// two VGPR output tiles sharing the register operand (accumulating in place); MFMA slot u = 2s+t
// w0 = {wf(0,0), wf(0,1), wf(1,0), wf(1,1)}
template <int NK, int NN, class WF, class BF, class VF, class NF>
__device__ __forceinline__ void vphase2w(f32x16& acc0, f32x16& acc1, const float (&w0)[4], WF wf, BF bf, VF vf, NF nf,
                                         float (&wn)[NN]) {
  float w[NK][2];
  w[0][0] = w0[0]; w[0][1] = w0[1]; w[1][0] = w0[2]; w[1][1] = w0[3];
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    if (s + 2 < NK) { w[s + 2][0] = wf(s + 2, 0); w[s + 2][1] = wf(s + 2, 1); }
    if (s == NK - 2) {
#pragma unroll
      for (int i = 0; i < NN; ++i) wn[i] = nf(i);
    }
    const float b = bf(s);
    vf(2 * s);
    mfv(acc0, w[s][0], b);
    vf(2 * s + 1);
    mfv(acc1, w[s][1], b);
    OIVAE_SB();
  }
}

// Operand reads of the matrix instructions go through LDS-address-space volatile pointers: a volatile read is
// never merged into ds_read2_b32, whose 8-bit offsets would cost one VALU add (8 matrix cycles) per merged pair.
using ldsv_t = const volatile float __attribute__((address_space(3)))*;

// Pins a value pair in a 64-bit register pair: the optimizer cannot scalarize arithmetic on it, so
// the following vector ops select v_pk_*_f32 (one VALU slot for two floats).  Emits no instruction.
__device__ __forceinline__ f32x2 pin2(f32x2 v) { asm("" : "+v"(v)); return v; }
// a - b on a register pair in ONE instruction (the compiler expands a v2f32 subtraction into two scalar ones)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a_, f32x2 b_) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a_), "v"(b_));
  return r;
}

__global__ __launch_bounds__(256) void dec_mfma64_v4(MfmaDecArgs a) {
  constexpr int K = 64, H = 64, M = 64, DG = 64;
  constexpr int LW = 65, LDB = 68;
  constexpr float kC = 2.885390081777927f;  // 2*log2(e): tanh(x) = 1 - 2/(exp2(kC x) + 1)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sAf = reinterpret_cast<float*>(smem_raw);  // kC * A[m][k]   (forward)
  float* sAb = sAf + M * LW;                         // A[m][k]        (backward)
  float* sW2 = sAb + M * LW;
  float* sb1 = sW2 + DG * LW;  // kC * b1
  float* sb2 = sb1 + M;
  float* sC3 = sb2 + DG;       // inv / sigma_d^2
  float* sPad = sC3 + DG;
  float* sAct = sPad + 64;     // 4 waves x 3 images x [32][LDB]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int g = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
  const float* Wsg = a.Ws + (int64_t)g * K * H;
  const float* W1g = a.W1 + (int64_t)g * M * H;
  float* tW1 = sAct;
  float* tWs = sAct + 64 * LW;
  {
    const float* W2g = a.W2 + (int64_t)g * DG * M;
    for (int i = tid; i < 4096; i += 256) {
      const int o = (i >> 6) * LW + (i & 63);
      tW1[o] = W1g[i];
      tWs[o] = Wsg[i];
      sW2[o] = W2g[i];
    }
    if (tid < 64) {
      const float sg = expf(a.ls[(int64_t)g * DG + tid]) + a.sig_off;
      sb1[tid] = kC * a.b1[(int64_t)g * M + tid];
      sb2[tid] = a.b2[(int64_t)g * DG + tid];
      sC3[tid] = (a.inv / sg) / sg;
    }
  }
  __syncthreads();
  {
    const int mt = wave >> 1, kt = wave & 1;
    f32x16 acc = (f32x16)(0.f);
    const float* pa = tW1 + (32 * mt + l31) * LW + hh;
    const float* pb = tWs + (32 * kt + l31) * LW + hh;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) acc = mf32(pa[2 * s], pb[2 * s], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (32 * mt + OIVAE_ROW(r, hh)) * LW + 32 * kt + l31;
      sAb[o] = acc[r];
      sAf[o] = kC * acc[r];
    }
  }
  __syncthreads();

  float* bufA = sAct + wave * (3 * 32 * LDB);
  float* bufB = bufA + 32 * LDB;
  float* bufZ = bufB + 32 * LDB;
  const ldsv_t pA_r = (ldsv_t)(sAf + l31 * LW + 32 * hh);       // P1: + 32t*LW + s
  const ldsv_t pW2_c = (ldsv_t)(sW2 + l31 * LW + 4 * hh);       // P3: + 32t*LW + 32mt + ROWC
  const ldsv_t pW2_r = (ldsv_t)(sW2 + (4 * hh) * LW + l31);     // P5: + (32dt+ROWC)*LW + 32t
  const ldsv_t pA_c = (ldsv_t)(sAb + (4 * hh) * LW + l31);      // P7: + (32mt+ROWC)*LW + 32t
  const ldsv_t imA = (ldsv_t)(bufA + hh * LDB + l31);
  const ldsv_t imB = (ldsv_t)(bufB + hh * LDB + l31);
  const ldsv_t imZ = (ldsv_t)(bufZ + hh * LDB + l31);
  float* wrA = bufA + l31 * LDB + 4 * hh;
  float* wrB = bufB + l31 * LDB + 4 * hh;
  float* wrZ = bufZ + l31 * LDB + 32 * hh;
  const float* pb1 = sb1 + 4 * hh;
  const float* pb2 = sb2 + 4 * hh;
  const float* pc3 = sC3 + 4 * hh;
  const ldsv_t csA = (ldsv_t)(bufA + lane);             // column sums: + row*LDB

  f32x16 dW2a[2][2], Pa[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { dW2a[i][j] = (f32x16)(0.f); Pa[i][j] = (f32x16)(0.f); }
  f32x2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f}, s3p = {0.f, 0.f};  // pair accumulators of the column sums

  const int N32 = (int)a.N;
  const int ntiles = (N32 + 31) / 32;
  const int tstride = 4 * nchunks;
  int tile = chunk * 4 + wave;
  const bool had_tiles = tile < ntiles;
  const bool one_sample = (a.N == a.B);
  f32x4 zreg[8], xreg[2][4];
  f32x16 tacc[2], gacc[2], dacc[2], zacc[2];
  zacc[0] = (f32x16)(0.f);
  zacc[1] = (f32x16)(0.f);
  auto row_of = [&](int t) { const int nn = t * 32 + l31; return nn < N32 ? nn : N32 - 1; };
  auto z_ptr = [&](int nn) { return reinterpret_cast<const f32x4*>(a.z + (int64_t)nn * K + hh * 32); };
  auto x_ptr = [&](int nn) {
    const int64_t xr = one_sample ? (int64_t)nn : (int64_t)nn % a.B;
    return a.X + xr * (int64_t)a.D + (int64_t)g * DG + 4 * hh;
  };
  auto load_bias = [&](f32x16& acc, const float* pb, int t) {  // accumulator := bias in accumulator layout
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pb + 32 * t + 8 * q);
      acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
    }
  };
  int n = row_of(had_tiles ? tile : 0);
  {
    const f32x4* zp = z_ptr(n);
#pragma unroll
    for (int v = 0; v < 8; ++v) zreg[v] = zp[v];
    const float* xp = x_ptr(n);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) xreg[dt][q] = *reinterpret_cast<const f32x4*>(xp + 32 * dt + 8 * q);
  }
  load_bias(tacc[0], pb1, 0); load_bias(tacc[1], pb1, 1);
  load_bias(gacc[0], pb2, 0); load_bias(gacc[1], pb2, 1);
  // dz^T[k][n] lives at dzT + k*N + n: uniform row base (scalar) + 32-bit per-lane byte offset
  char* const dzb = reinterpret_cast<char*>(a.dzT);
  const int64_t rowbytes = a.N * 4;
  uint32_t voff_prev = (uint32_t)((int64_t)(4 * hh) * a.N + n) * 4u;
  float w1a[3] = {pA_r[0], pA_r[1], pA_r[2]};  // first operands of the first phase, carried around the loop

  for (; tile < ntiles; tile += tstride) {
    const int tnext = (tile + tstride < ntiles) ? tile + tstride : tile;
    const int n_next = row_of(tnext);
    const bool full = (tile * 32 + 32 <= N32);  // wave-uniform: every row of the tile exists (all tiles but possibly the last)
    const float livef = (tile * 32 + l31 < N32) ? 1.0f : 0.0f;
    const f32x4* zp_next = z_ptr(n_next);
    const float* xp_next = x_ptr(n_next);

    auto img_store = [&](float* base, const f32x16& v, int t, int q) {
      f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      *reinterpret_cast<f32x4*>(base + 32 * t + 8 * q) = o;
    };
    auto imrow = [&](ldsv_t im, int col) { return [=](int s) { return im[2 * s * LDB + col]; }; };
    // t = 1 - 2/(exp2(arg)+1) on elements 2p, 2p+1 of an accumulator that holds kC*(a1+b1)
    auto tanh_pair = [&](f32x16& acc, int p) {
      f32x2 e = pin2(f32x2{__builtin_amdgcn_exp2f(acc[2 * p]), __builtin_amdgcn_exp2f(acc[2 * p + 1])});
      e = pin2(e + 1.0f);
      f32x2 r = pin2(f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])});
      r = pin2(r * (-2.0f) + 1.0f);
      acc[2 * p] = r[0]; acc[2 * p + 1] = r[1];
    };
    // gmu = (mu - x) * inv/sigma^2 on quad q (elements 4q..4q+3) of tile t, then its 16 bytes of the gmu image
    f32x2 csv = {0.f, 0.f};  // column-sum pair in flight
    f32x4 c3q[4];  // inv/sigma^2 of the four quads of one tile, fetched one slot before their use
    auto epi_load = [&](int t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) c3q[q] = *reinterpret_cast<const f32x4*>(pc3 + 32 * t + 8 * q);
    };
    auto epi_quad = [&](int t, int q) {
      f32x4 c3 = c3q[q];
      if (!full) {  // a real (scalar) branch: taken by at most one tile of the launch
        asm volatile("" ::: "memory");
        c3 = c3 * livef;
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        f32x2 mu = pin2(f32x2{gacc[t][4 * q + 2 * h2], gacc[t][4 * q + 2 * h2 + 1]});
        const f32x2 x = pin2(f32x2{xreg[t][q][2 * h2], xreg[t][q][2 * h2 + 1]});
        const f32x2 c = pin2(f32x2{c3[2 * h2], c3[2 * h2 + 1]});
        mu = pin2(pk_sub(mu, x) * c);
        gacc[t][4 * q + 2 * h2] = mu[0]; gacc[t][4 * q + 2 * h2 + 1] = mu[1];
      }
      img_store(wrA, gacc[t], t, q);
    };
    // da1 = dt * (1 - t^2) on quad q of tile t, then its 16 bytes of the da1 image
    auto da1_quad = [&](int t, int q) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const f32x2 tt = pin2(f32x2{tacc[t][4 * q + 2 * h2], tacc[t][4 * q + 2 * h2 + 1]});
        f32x2 d = pin2(f32x2{dacc[t][4 * q + 2 * h2], dacc[t][4 * q + 2 * h2 + 1]});
        const f32x2 u = pin2(1.0f - tt * tt);
        d = pin2(d * u);
        dacc[t][4 * q + 2 * h2] = d[0]; dacc[t][4 * q + 2 * h2 + 1] = d[1];
      }
      img_store(wrA, dacc[t], t, q);
    };

    // operand functors of the thirteen phases (k-step s -> LDS operand), and the first loads of each as i -> value
    auto w_p1 = [&](int t) { return [=](int s) { return pA_r[32 * t * LW + s]; }; };
    auto w_p3a = [&](int s, int t) { return pW2_c[32 * t * LW + OIVAE_ROWC(s)]; };
    auto w_p3b = [&](int s) { return pW2_c[32 + OIVAE_ROWC(s)]; };
    auto w_p3c = [&](int s) { return pW2_c[32 * LW + 32 + OIVAE_ROWC(s)]; };
    auto w_p5 = [&](int t) { return [=](int s) { return pW2_r[(32 * (s >> 4) + OIVAE_ROWC(s & 15)) * LW + 32 * t]; }; };
    auto w_p7 = [&](int t) { return [=](int s) { return pA_c[(32 * (s >> 4) + OIVAE_ROWC(s & 15)) * LW + 32 * t]; }; };
    auto wg_first = [&](ldsv_t ia, int ca, ldsv_t ib) {  // {af(0), bf0(0), bf1(0), af(1), bf0(1), bf1(1)}
      return [=](int i) { const int s = i / 3, j = i % 3; return j == 0 ? ia[2 * s * LDB + ca] : ib[2 * s * LDB + (j == 1 ? 0 : 32)]; };
    };
    auto b_z = [&](int s) { return zreg[s >> 2][s & 3]; };
    auto b_g = [&](int s) { return gacc[s >> 4][s & 15]; };
    auto b_d = [&](int s) { return dacc[s >> 4][s & 15]; };
    float c3[3], c4[4], c6[6];  // operand carries between phases (c3 enters holding P1a's first operands)
    c3[0] = w1a[0]; c3[1] = w1a[1]; c3[2] = w1a[2];

    // ---- P1  kC*(a1+b1)^T = (kC A) z^T + kC b1, tile by tile.  Under tile 0: dz atomics of the previous row tile, z image ----
    // (one atomic every other slot over both tiles: at most ~16 stay in flight per wave, below the issue-stall range)
    auto dz_atomic = [&](int t, int r) {
      atomicAdd(reinterpret_cast<float*>(dzb + (int64_t)(32 * t + OIVAE_ROWC(r)) * rowbytes + voff_prev), zacc[t][r]);
    };
    vphase1<32, false>(tacc[0], c3, w_p1(0), b_z, [&](int s) {
      if (!(s & 1)) dz_atomic(0, s >> 1);
      if ((s & 3) == 3) *reinterpret_cast<f32x4*>(wrZ + 4 * (s >> 2)) = zreg[s >> 2];
    }, w_p1(1), c3);
    vphase1<32, false>(tacc[1], c3, w_p1(1), b_z, [&](int s) {
      if (s & 1) dz_atomic(1, s >> 1);
      if (s >= 2 && s < 10) tanh_pair(tacc[0], s - 2);
    }, [&](int i) { return w_p3a(i >> 1, i & 1); }, c4);
    // ---- P3  mu^T = W2 t^T + b2.  k over t tile 0 first (tanh of tile 1 under it) ----
    vphase2w<16>(gacc[0], gacc[1], c4, w_p3a, [&](int s) { return tacc[0][s]; },
                 [&](int u) { if (u >= 2 && u < 10) tanh_pair(tacc[1], u - 2); }, w_p3b, c3);
    vphase1<16, false>(gacc[0], c3, w_p3b, [&](int s) { return tacc[1][s]; },
                       [&](int s) { if (s >= 2 && s < 10) img_store(wrB, tacc[(s - 2) >> 2], (s - 2) >> 2, (s - 2) & 3); },  // t image
                       w_p3c, c3);
    vphase1<16, false>(gacc[1], c3, w_p3c, [&](int s) { return tacc[1][s]; },
                       [&](int s) { if (s == 0) epi_load(0); if (s >= 2 && s < 6) epi_quad(0, s - 2); },
                       wg_first(imA, 0, imB), c6);
    // ---- P4  dW2 += gmu^T t.  rows i=0 first; epilogue of tile 1 and the z prefetch under them ----
    aphasewg<16>(dW2a[0][0], dW2a[0][1], c6, imrow(imA, 0), imrow(imB, 0), imrow(imB, 32),
                 [&](int u) {
                   if (u == 0) epi_load(1);
                   if (u >= 2 && u < 6) epi_quad(1, u - 2);
                   else if (u >= 8 && u < 16) zreg[u - 8] = zp_next[u - 8];
                 }, wg_first(imA, 32, imB), c6);
    aphasewg<16>(dW2a[1][0], dW2a[1][1], c6, imrow(imA, 32), imrow(imB, 0), imrow(imB, 32),
                 [&](int u) {
                   if ((!(u & 1) && u >= 2) || u == 31) { const f32x2 v = pin2(csv); s1p = pin2(s1p + v); s2p = pin2(v * v + s2p); }
                   if (!(u & 1)) csv = f32x2{csA[u * LDB], csA[(u + 1) * LDB]};  // not pinned here: a pin waits for the load
                 }, w_p5(0), c3);
    // ---- P5  dt^T = W2^T gmu^T tile by tile; x prefetch under tile 0, da1 of tile 0 under tile 1 ----
    vphase1<32, true>(dacc[0], c3, w_p5(0), b_g, [&](int s) {
      if (s >= 2 && s < 10) xreg[(s - 2) >> 2][(s - 2) & 3] = *reinterpret_cast<const f32x4*>(xp_next + 32 * ((s - 2) >> 2) + 8 * ((s - 2) & 3));
    }, w_p5(1), c3);
    vphase1<32, true>(dacc[1], c3, w_p5(1), b_g, [&](int s) { if (s >= 2 && s < 6) da1_quad(0, s - 2); },
                      wg_first(imA, 0, imZ), c6);
    // ---- P6  P += da1^T z.  rows m 0..31 first; da1 of tile 1 under them ----
    aphasewg<16>(Pa[0][0], Pa[0][1], c6, imrow(imA, 0), imrow(imZ, 0), imrow(imZ, 32),
                 [&](int u) { if (u >= 2 && u < 6) da1_quad(1, u - 2); }, wg_first(imA, 32, imZ), c6);
    aphasewg<16>(Pa[1][0], Pa[1][1], c6, imrow(imA, 32), imrow(imZ, 0), imrow(imZ, 32),
                 [&](int u) {
                   if ((!(u & 1) && u >= 2) || u == 31) s3p = pin2(s3p + pin2(csv));
                   if (!(u & 1)) csv = f32x2{csA[u * LDB], csA[(u + 1) * LDB]};
                 }, w_p7(0), c3);
    // ---- P7  dz^T = A^T da1^T tile by tile (atomics deferred); the next tile's bias images under them ----
    voff_prev = (uint32_t)((int64_t)(4 * hh) * a.N + n) * 4u;
    vphase1<32, true>(zacc[0], c3, w_p7(0), b_d, [&](int s) {
      if (s == 2) load_bias(gacc[0], pb2, 0);
      if (s == 3) load_bias(gacc[1], pb2, 1);
    }, w_p7(1), c3);
    vphase1<32, true>(zacc[1], c3, w_p7(1), b_d, [&](int s) {
      if (s == 2) load_bias(tacc[0], pb1, 0);
      if (s == 3) load_bias(tacc[1], pb1, 1);
    }, w_p1(0), w1a);
    n = n_next;
  }
  if (had_tiles) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        atomicAdd(reinterpret_cast<float*>(dzb + (int64_t)(32 * t + OIVAE_ROWC(r)) * rowbytes + voff_prev), zacc[t][r]);
  }
  const float cs_gmu = s1p[0] + s1p[1], cs_gmu2 = s2p[0] + s2p[1], cs_da1 = s3p[0] + s3p[1];
  // A's two copies and W2 are dead: P and W1 take their places
  dec64_finish(a, g, chunk, tid, wave, lane, dW2a, Pa, cs_gmu, cs_gmu2, cs_da1, sAct, sAf, sW2);
}

// ------------------------------------------------------------------------------------------------------------
// dec_x3_64: the all-64 decoder on the bf16 matrix instruction, every fp32 operand carried as three bf16 terms
// (x3_split.h; six partial products per fp32 product, fp32 accumulation: fp32-grade results at 6/16 of the fp32
// MFMA time).  Same mathematics as dec_mfma64_v4 (composed first layer A = W1 Ws^T), different data flow:
//   * weights sit in LDS as MFMA-ready split images (one 1 KB fragment per (row block, 16 reduction slots, term)):
//     kC A (rows m), A^T (rows k), W2 (rows d), W2^T (rows m);
//   * the chain runs transposed as in v4 (lane = batch row, registers = features: an accumulator becomes the next
//     layer's operand after a split in registers); for the weight-gradient products, whose reduction runs over
//     batch rows, each activation takes a wave-private LDS round trip as fp32 ([row][feature] tile written 16 bytes
//     at a time, read back with lane = feature) and is split again in that orientation; no workgroup barrier in
//     the row loop;
//   * column sums (db1, db2, sum of squares for log-lik and dlog_sigma) are plain per-lane adds on the values
//     read back in the feature-on-lane orientation.
// The reduction slot order of a chained operand follows the accumulator layout: slot e of lane half hh in step
// ks is feature 16 ks + 8 (e / 4) + 4 hh + e % 4; the weight images are built in that order.
// ------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void dec_x3_64(MfmaDecArgs a) {
  constexpr int K = 64, H = 64, M = 64, DG = 64, LW = 65;
  constexpr float kC = 2.885390081777927f;  // 2*log2(e): tanh(x) = 1 - 2/(exp2(kC x) + 1)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint32_t* img = reinterpret_cast<uint32_t*>(smem_raw);   // [4][2 rb][4 ks][3 terms][64 lanes][4 dwords]
  float* sb1 = reinterpret_cast<float*>(img + 4 * 6144);   // kC * b1
  float* sb2 = sb1 + 64;
  float* sC3 = sb2 + 64;                                   // inv / sigma_d^2
  float* scr = sC3 + 64;                                   // 3 x [64][65] fp32 staging
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int g = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
  {
    float* tW1 = scr;
    float* tWs = scr + 64 * LW;
    float* tW2 = scr + 2 * 64 * LW;
    const float* Wsg = a.Ws + (int64_t)g * K * H;
    const float* W1g = a.W1 + (int64_t)g * M * H;
    const float* W2g = a.W2 + (int64_t)g * DG * M;
    for (int i = tid; i < 4096; i += 256) {
      const int o = (i >> 6) * LW + (i & 63);
      tW1[o] = W1g[i];
      tWs[o] = Wsg[i];
      tW2[o] = W2g[i];
    }
    if (tid < 64) {
      const float sg = expf(a.ls[(int64_t)g * DG + tid]) + a.sig_off;
      sb1[tid] = kC * a.b1[(int64_t)g * M + tid];
      sb2[tid] = a.b2[(int64_t)g * DG + tid];
      sC3[tid] = (a.inv / sg) / sg;
    }
    __syncthreads();
    // A[m][k] = sum_h W1[m][h] Ws[k][h] on the fp32 matrix instruction (an fmaf chain, as in dec_mfma64_v4): one
    // 32 x 32 block per wave, then in place of W1
    {
      const int mt = wave >> 1, kt = wave & 1;
      f32x16 acc = (f32x16)(0.f);
      const float* pa = tW1 + (32 * mt + l31) * LW + hh;
      const float* pb = tWs + (32 * kt + l31) * LW + hh;
#pragma unroll 8
      for (int s2 = 0; s2 < 32; ++s2) acc = mf32(pa[2 * s2], pb[2 * s2], acc);
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) tW1[(32 * mt + OIVAE_ROW(r, hh)) * LW + 32 * kt + l31] = acc[r];
    }
    __syncthreads();
    const float* A = tW1;
    // the four split images: item = (row r, step ks, lane half h2) -> 8 slots -> three 16-byte terms
    for (int item = tid; item < 512; item += 256) {
      const int r = item >> 3, ks = (item >> 1) & 3, h2 = item & 1;
      float v[4][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = x3_slot_feature(ks, h2, e);
        v[0][e] = kC * A[r * LW + c];   // kC A   : rows m, reduction over k
        v[1][e] = A[c * LW + r];        // A^T    : rows k, reduction over m
        v[2][e] = tW2[r * LW + c];      // W2     : rows d, reduction over m
        v[3][e] = tW2[c * LW + r];      // W2^T   : rows m, reduction over d
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const Frag3 f = split8(v[q][0], v[q][1], v[q][2], v[q][3], v[q][4], v[q][5], v[q][6], v[q][7]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<x3_u32x4*>(img + q * 6144 + ((((r >> 5) * 4 + ks) * 3 + p) * 64 + h2 * 32 + (r & 31)) * 4) = f.p[p];
      }
    }
    __syncthreads();
  }
  const uint32_t* imgl = img + lane * 4;
  auto wfrag = [&](int which, int rb, int ks) {
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = *reinterpret_cast<const x3_u32x4*>(imgl + which * 6144 + ((rb * 4 + ks) * 3 + p) * 256);
    return f;
  };
  // operand of 16 reduction slots from accumulator registers 8q..8q+7
  auto afrag = [&](const f32x16& v, int q) {
    return split8(v[8 * q], v[8 * q + 1], v[8 * q + 2], v[8 * q + 3], v[8 * q + 4], v[8 * q + 5], v[8 * q + 6], v[8 * q + 7]);
  };
  // accumulator := vector in transposed layout (register r <-> feature 32 rb + OIVAE_ROW(r, hh))
  auto init_t = [&](f32x16& acc, const float* vec, int rb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(vec + 32 * rb + 8 * q + 4 * hh);
      acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
    }
  };

  f32x16 dW2a[2][2], Pa[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { dW2a[i][j] = (f32x16)(0.f); Pa[i][j] = (f32x16)(0.f); }
  float sg1[2] = {0.f, 0.f}, sg2[2] = {0.f, 0.f}, sd1[2] = {0.f, 0.f};  // per-lane column sums (feature 32 b + l31)

  const int N32 = (int)a.N;
  const int ntiles = (N32 + 31) / 32;
  const bool one_sample = (a.N == a.B);
  // wave-private [32 rows][68] fp32 tile in the (idle during the row loop) staging area: every activation is
  // computed in the transposed orientation (lane = batch row), written here 16 bytes at a time and read back in
  // the plain one (lane = feature, registers = batch rows) for the weight-gradient products and the column sums
  constexpr int LT = 68;
  float* tr = scr + wave * (32 * LT);
  float* trw = tr + l31 * LT + 4 * hh;                 // this lane's row
  const float* trr = tr + (4 * hh) * LT + l31;         // plain orientation: register r <-> row OIVAE_ROW(r, hh)
  auto tr_write = [&](const f32x16& v, int rb) {       // accumulator block rb (features 32 rb + OIVAE_ROW(r, hh))
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f32x4*>(trw + 32 * rb + 8 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  };
  auto tr_read = [&](int cb) {                         // column block cb: 16 rows of column 32 cb + l31
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = trr[(8 * (r >> 2) + (r & 3)) * LT + 32 * cb];
    return v;
  };
  // z's split fragments come ready-made from the image every group shares.  Global-memory latency is never waited
  // for in the row loop: a tile's transposed-orientation fragments are fetched during the previous tile, its
  // plain-orientation ones and its x rows at the top of the tile, two and four products ahead of their use.
  auto zfrag_of = [&](int t, int f) {
    Frag3 r;
    const uint32_t* zi = a.zimg + ((int64_t)t * 24 * 64 + lane) * 4;
#pragma unroll
    for (int p = 0; p < 3; ++p) r.p[p] = *reinterpret_cast<const x3_u32x4*>(zi + (f * 3 + p) * 256);
    return r;
  };
  int tile = chunk * 4 + wave;
  Frag3 zf[4];
  if (tile < ntiles) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) zf[ks] = zfrag_of(tile, ks);
  }
  for (; tile < ntiles; tile += 4 * nchunks) {
    const int r0 = tile * 32;
    const int nT = (r0 + l31 < N32) ? r0 + l31 : N32 - 1;   // this lane's batch row
    const float liveT = (r0 + l31 < N32) ? 1.0f : 0.0f;
    const float* xT = a.X + (one_sample ? (int64_t)nT : (int64_t)nT % a.B) * (int64_t)a.D + (int64_t)g * DG + 4 * hh;
    const int tnext = (tile + 4 * nchunks < ntiles) ? tile + 4 * nchunks : tile;
    auto zfrag = [&](int f) { return zfrag_of(tile, f); };
    f32x4 xq[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int q = 0; q < 4; ++q) xq[rb][q] = *reinterpret_cast<const f32x4*>(xT + 32 * rb + 8 * q);
    __builtin_amdgcn_sched_barrier(0);
    // ---- kC (a1 + b1); t = tanh; afterwards only 1 - t^2 stays in the accumulators ----
    f32x16 uT[2];
    Frag3 tf[4], tNf[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      init_t(uT[rb], sb1, rb);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) x3mm(uT[rb], wfrag(0, rb, ks), zf[ks]);
#pragma unroll
      for (int r = 0; r < 16; ++r) uT[rb][r] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(uT[rb][r]) + 1.0f);
      tr_write(uT[rb], rb);
      tf[2 * rb] = afrag(uT[rb], 0); tf[2 * rb + 1] = afrag(uT[rb], 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) uT[rb][r] = 1.0f - uT[rb][r] * uT[rb][r];
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const f32x16 tn = tr_read(mb);
      tNf[mb][0] = afrag(tn, 0);
      tNf[mb][1] = afrag(tn, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) zf[ks] = zfrag_of(tnext, ks);   // the next tile's first operand
    Frag3 zn[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) zn[f] = zfrag(4 + f);              // this tile's last operand
    __builtin_amdgcn_sched_barrier(0);
    // ---- mu = W2 t + b2; gmu = (mu - x) inv / sigma^2 ----
    Frag3 gf[4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 gT;
      init_t(gT, sb2, rb);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) x3mm(gT, wfrag(2, rb, ks), tf[ks]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 x = xq[rb][q];
        const f32x4 c3 = *reinterpret_cast<const f32x4*>(sC3 + 32 * rb + 8 * q + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) gT[4 * q + i] = (gT[4 * q + i] - x[i]) * (c3[i] * liveT);
      }
      tr_write(gT, rb);
      gf[2 * rb] = afrag(gT, 0); gf[2 * rb + 1] = afrag(gT, 1);
    }
    // ---- dW2[d][m] += sum_n gmu[n][d] t[n][m]; column sums of gmu ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x16 gn = tr_read(i);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s1 += gn[r]; s2 = fmaf(gn[r], gn[r], s2); }
      sg1[i] += s1;
      sg2[i] += s2;
      const Frag3 g0 = afrag(gn, 0), g1 = afrag(gn, 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        x3mm(dW2a[i][j], g0, tNf[j][0]);
        x3mm(dW2a[i][j], g1, tNf[j][1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dt = W2^T gmu; da1 = dt (1 - t^2) ----
    Frag3 df[4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      f32x16 dT = (f32x16)(0.f);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) x3mm(dT, wfrag(3, rb, ks), gf[ks]);
#pragma unroll
      for (int r = 0; r < 16; ++r) dT[r] *= uT[rb][r];
      tr_write(dT, rb);
      df[2 * rb] = afrag(dT, 0); df[2 * rb + 1] = afrag(dT, 1);
    }
    // ---- P[m][k] += sum_n da1[n][m] z[n][k]; column sums of da1 ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x16 dn = tr_read(i);
      float sd = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sd += dn[r];
      sd1[i] += sd;
      const Frag3 d0 = afrag(dn, 0), d1 = afrag(dn, 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        x3mm(Pa[i][j], d0, zn[2 * j]);
        x3mm(Pa[i][j], d1, zn[2 * j + 1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dz^T[k][n] += sum_m A[m][k] da1[n][m] ----
    float* dzp = a.dzT + (int64_t)(4 * hh) * a.N + nT;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 dz = (f32x16)(0.f);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) x3mm(dz, wfrag(1, kb, ks), df[ks]);
      if (r0 + l31 < N32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(dzp + (int64_t)(32 * kb + 8 * (r >> 2) + (r & 3)) * a.N, dz[r]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // per-column sums: the two lane halves of a column are added through LDS, lane L then holds column L
  __syncthreads();  // every wave is done with the staging area's previous contents (the images stay untouched)
  float* tmp = scr + 2 * 64 * LW + wave * 384;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    tmp[(0 * 2 + b) * 64 + hh * 32 + l31] = sg1[b];
    tmp[(1 * 2 + b) * 64 + hh * 32 + l31] = sg2[b];
    tmp[(2 * 2 + b) * 64 + hh * 32 + l31] = sd1[b];
  }
  __syncthreads();
  const int cb = lane >> 5;
  const float cs_gmu = tmp[(0 * 2 + cb) * 64 + l31] + tmp[(0 * 2 + cb) * 64 + 32 + l31];
  const float cs_gmu2 = tmp[(1 * 2 + cb) * 64 + l31] + tmp[(1 * 2 + cb) * 64 + 32 + l31];
  const float cs_da1 = tmp[(2 * 2 + cb) * 64 + l31] + tmp[(2 * 2 + cb) * 64 + 32 + l31];
  // the images are dead: staging for the four-wave reduction; P and W1 go to the fp32 staging slots
  dec64_finish(a, g, chunk, tid, wave, lane, dW2a, Pa, cs_gmu, cs_gmu2, cs_da1, reinterpret_cast<float*>(img), scr, scr + 64 * LW);
}


// ------------------------------------------------------------------------------------------------------------
// dec_x3_64p: dec_x3_64 with TWO waves per SIMD.  dec_x3_64 is instruction-issue-bound in its single wave per SIMD
// (2600 instructions per 32-row tile against 288 MFMAs), and its registers (128 persistent accumulators + the chain)
// and LDS leave no room for a second copy of that wave.  Here the work of one tile stream is split by ROLE instead:
//   chain wave  (waves 0-3): the four chained products, tanh / gmu / da1, the splits in the transposed orientation,
//                            the dz atomics; publishes t, gmu and da1 as fp32 tiles in the stream's LDS tile;
//   weight-gradient wave (waves 4-7, same stream = wave & 3): reads each tile back with lane = feature, column sums,
//                            splits in that orientation, dW2 += gmu^T t and P += da1^T z with the 128 persistent
//                            accumulators.
// Each role fits 256 registers, the LDS footprint is unchanged, and the two issue streams of a SIMD interleave in
// hardware (one wave's vector work under the other's MFMAs).  Hand-over through two counters per stream in LDS
// (tiles published / tiles consumed, single buffer): a wave's LDS instructions complete in order, so "write tile,
// s_waitcnt lgkmcnt(0), bump counter" and "see counter, read tile" need no barrier; every wait loop is bounded.
// ------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(512, 1) void dec_x3_64p(MfmaDecArgs a) {
  constexpr int K = 64, H = 64, M = 64, DG = 64, LW = 65;
  constexpr float kC = 2.885390081777927f;  // 2*log2(e): tanh(x) = 1 - 2/(exp2(kC x) + 1)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint32_t* img = reinterpret_cast<uint32_t*>(smem_raw);   // [4][2 rb][4 ks][3 terms][64 lanes][4 dwords]
  float* sb1 = reinterpret_cast<float*>(img + 4 * 6144);   // kC * b1
  float* sb2 = sb1 + 64;
  float* sC3 = sb2 + 64;                                   // inv / sigma_d^2
  float* scr = sC3 + 64;                                   // 3 x [64][65] fp32 staging
  volatile uint32_t* flags = reinterpret_cast<volatile uint32_t*>(scr + 3 * 64 * LW);  // [4 streams][published, consumed]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int g = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
  const bool chainw = wave < 4;
  const int sw = wave & 3;
  {
    float* tW1 = scr;
    float* tWs = scr + 64 * LW;
    float* tW2 = scr + 2 * 64 * LW;
    const float* Wsg = a.Ws + (int64_t)g * K * H;
    const float* W1g = a.W1 + (int64_t)g * M * H;
    const float* W2g = a.W2 + (int64_t)g * DG * M;
    for (int i = tid; i < 4096; i += 512) {
      const int o = (i >> 6) * LW + (i & 63);
      tW1[o] = W1g[i];
      tWs[o] = Wsg[i];
      tW2[o] = W2g[i];
    }
    if (tid < 64) {
      const float sg = expf(a.ls[(int64_t)g * DG + tid]) + a.sig_off;
      sb1[tid] = kC * a.b1[(int64_t)g * M + tid];
      sb2[tid] = a.b2[(int64_t)g * DG + tid];
      sC3[tid] = (a.inv / sg) / sg;
    }
    if (tid < 16) flags[tid] = 0;
    __syncthreads();
    // A[m][k] = sum_h W1[m][h] Ws[k][h] on the fp32 matrix instruction, one 32 x 32 block per chain wave
    f32x16 acc = (f32x16)(0.f);
    const int mt = sw >> 1, kt = sw & 1;
    if (chainw) {
      const float* pa = tW1 + (32 * mt + l31) * LW + hh;
      const float* pb = tWs + (32 * kt + l31) * LW + hh;
#pragma unroll 8
      for (int s2 = 0; s2 < 32; ++s2) acc = mf32(pa[2 * s2], pb[2 * s2], acc);
    }
    __syncthreads();
    if (chainw) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tW1[(32 * mt + OIVAE_ROW(r, hh)) * LW + 32 * kt + l31] = acc[r];
    }
    __syncthreads();
    const float* A = tW1;
    {  // the four split images: item = (row r, step ks, lane half h2) -> 8 slots -> three 16-byte terms; 512 items
      const int item = tid;
      const int r = item >> 3, ks = (item >> 1) & 3, h2 = item & 1;
      float v[4][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = x3_slot_feature(ks, h2, e);
        v[0][e] = kC * A[r * LW + c];
        v[1][e] = A[c * LW + r];
        v[2][e] = tW2[r * LW + c];
        v[3][e] = tW2[c * LW + r];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const Frag3 f = split8(v[q][0], v[q][1], v[q][2], v[q][3], v[q][4], v[q][5], v[q][6], v[q][7]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<x3_u32x4*>(img + q * 6144 + ((((r >> 5) * 4 + ks) * 3 + p) * 64 + h2 * 32 + (r & 31)) * 4) = f.p[p];
      }
    }
    __syncthreads();
  }
  const uint32_t* imgl = img + lane * 4;
  auto wfrag = [&](int which, int rb, int ks) {
    Frag3 f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = *reinterpret_cast<const x3_u32x4*>(imgl + which * 6144 + ((rb * 4 + ks) * 3 + p) * 256);
    return f;
  };
  auto afrag = [&](const f32x16& v, int q) {
    return split8(v[8 * q], v[8 * q + 1], v[8 * q + 2], v[8 * q + 3], v[8 * q + 4], v[8 * q + 5], v[8 * q + 6], v[8 * q + 7]);
  };
  // One chained product, both 32-row output blocks, as eight pinned units (u = 4 rb + ks) of six MFMAs: the weight
  // fragments of unit u+2 are requested and the operand of step u+1 is split inside unit u's region, so that LDS
  // latency and the split's vector work sit under MFMAs already issued (left to itself the scheduler requests a
  // fragment one MFMA before its use and waits for it).  `side(u)`: independent work to place in unit u.
  auto chain_mm = [&](f32x16* acc, int which, Frag3* bf, auto&& make, auto&& side) {
    Frag3 w[3];
    w[0] = wfrag(which, 0, 0);
    w[1] = wfrag(which, 0, 1);
    bf[0] = make(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u + 2 < 8) w[(u + 2) % 3] = wfrag(which, (u + 2) >> 2, (u + 2) & 3);
      if (u + 1 < 4) bf[u + 1] = make(u + 1);
      side(u);
      x3mm(acc[u >> 2], w[u % 3], bf[u & 3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto init_t = [&](f32x16& acc, const float* vec, int rb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(vec + 32 * rb + 8 * q + 4 * hh);
      acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
    }
  };
  auto zfrag_of = [&](int t, int f) {
    Frag3 r;
    const uint32_t* zi = a.zimg + ((int64_t)t * 24 * 64 + lane) * 4;
#pragma unroll
    for (int p = 0; p < 3; ++p) r.p[p] = *reinterpret_cast<const x3_u32x4*>(zi + (f * 3 + p) * 256);
    return r;
  };
  // the stream's fp32 tile [32 rows][68]: written 16 bytes at a time with lane = batch row, read with lane = feature
  constexpr int LT = 68;
  float* tr = scr + sw * (32 * LT);
  float* trw = tr + l31 * LT + 4 * hh;
  const float* trr = tr + (4 * hh) * LT + l31;
  auto tr_write = [&](const f32x16& v, int rb) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f32x4*>(trw + 32 * rb + 8 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  };
  auto tr_read = [&](int cb) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = trr[(8 * (r >> 2) + (r & 3)) * LT + 32 * cb];
    return v;
  };
  volatile uint32_t* f_pub = flags + 2 * sw;
  volatile uint32_t* f_con = flags + 2 * sw + 1;

  const int N32 = (int)a.N;
  const int ntiles = (N32 + 31) / 32;
  const bool one_sample = (a.N == a.B);
  f32x16 dW2a[2][2], Pa[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { dW2a[i][j] = (f32x16)(0.f); Pa[i][j] = (f32x16)(0.f); }
  float sg1[2] = {0.f, 0.f}, sg2[2] = {0.f, 0.f}, sd1[2] = {0.f, 0.f};

  if (chainw) {
    // =============================== chain wave ===============================
    __builtin_amdgcn_s_setprio(3);   // the critical path of the pair: issue before the other wave of the SIMD
    uint32_t pub = 0;
    Frag3 zf[4];
    if (chunk * 4 + sw < ntiles) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) zf[ks] = zfrag_of(chunk * 4 + sw, ks);
    }
    for (int tile = chunk * 4 + sw; tile < ntiles; tile += 4 * nchunks) {
      const int tnext = (tile + 4 * nchunks < ntiles) ? tile + 4 * nchunks : tile;
      const int r0 = tile * 32;
      const int nT = (r0 + l31 < N32) ? r0 + l31 : N32 - 1;
      const float liveT = (r0 + l31 < N32) ? 1.0f : 0.0f;
      const float* xT = a.X + (one_sample ? (int64_t)nT : (int64_t)nT % a.B) * (int64_t)a.D + (int64_t)g * DG + 4 * hh;
      // ---- kC (a1 + b1); z's fragments were fetched during the previous tile, x is needed two products from here ----
      f32x4 xq[2][4];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) xq[rb][q] = *reinterpret_cast<const f32x4*>(xT + 32 * rb + 8 * q);
      f32x16 uT[2];
      init_t(uT[0], sb1, 0);
      init_t(uT[1], sb1, 1);
      auto none = [&](int) {};
      chain_mm(uT, 0, zf, [&](int ks) { return zf[ks]; }, none);
      // ---- t = tanh: published first, split during the next product ----
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) uT[rb][r] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(uT[rb][r]) + 1.0f);
      lds_wait_ge(f_con, pub, a.err);
      tr_write(uT[0], 0);
      tr_write(uT[1], 1);
      lds_post(f_pub, ++pub);
      // ---- mu = W2 t + b2; gmu = (mu - x) inv / sigma^2; afterwards uT holds 1 - t^2 ----
      Frag3 tf[4];
      f32x16 gT[2];
      init_t(gT[0], sb2, 0);
      init_t(gT[1], sb2, 1);
      auto gmu_quarter = [&](int rb, int q) {
        const f32x4 c3 = *reinterpret_cast<const f32x4*>(sC3 + 32 * rb + 8 * q + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) gT[rb][4 * q + i] = (gT[rb][4 * q + i] - xq[rb][q][i]) * (c3[i] * liveT);
      };
      chain_mm(gT, 2, tf,
               [&](int ks) {
                 const Frag3 f = afrag(uT[ks >> 1], ks & 1);
                 if (ks & 1) {
#pragma unroll
                   for (int r = 0; r < 16; ++r) uT[ks >> 1][r] = 1.0f - uT[ks >> 1][r] * uT[ks >> 1][r];
                 }
                 return f;
               },
               [&](int u) { if (u >= 4) gmu_quarter(0, u - 4); });   // block 0 is complete from unit 4 on
#pragma unroll
      for (int q = 0; q < 4; ++q) gmu_quarter(1, q);
      lds_wait_ge(f_con, pub, a.err);
      tr_write(gT[0], 0);
      tr_write(gT[1], 1);
      lds_post(f_pub, ++pub);
      // ---- dt = W2^T gmu; da1 = dt (1 - t^2) ----
      Frag3 gf[4];
      f32x16 dT[2];
      dT[0] = (f32x16)(0.f);
      dT[1] = (f32x16)(0.f);
      chain_mm(dT, 3, gf, [&](int ks) { return afrag(gT[ks >> 1], ks & 1); },
               [&](int u) {
                 if (u >= 4) {
#pragma unroll
                   for (int r = 4 * (u - 4); r < 4 * (u - 4) + 4; ++r) dT[0][r] *= uT[0][r];
                 }
               });
#pragma unroll
      for (int r = 0; r < 16; ++r) dT[1][r] *= uT[1][r];
      lds_wait_ge(f_con, pub, a.err);
      tr_write(dT[0], 0);
      tr_write(dT[1], 1);
      lds_post(f_pub, ++pub);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) zf[ks] = zfrag_of(tnext, ks);   // the next tile's first operand, one product ahead
      // ---- dz^T[k][n] += sum_m A[m][k] da1[n][m] ----
      Frag3 df[4];
      f32x16 dz[2];
      dz[0] = (f32x16)(0.f);
      dz[1] = (f32x16)(0.f);
      float* dzp = a.dzT + (int64_t)(4 * hh) * a.N + nT;
      const bool liveRow = r0 + l31 < N32;
      chain_mm(dz, 1, df, [&](int ks) { return afrag(dT[ks >> 1], ks & 1); },
               [&](int u) {
                 if (u >= 4 && liveRow) {
#pragma unroll
                   for (int r = 4 * (u - 4); r < 4 * (u - 4) + 4; ++r)
                     atomicAdd(dzp + (int64_t)(8 * (r >> 2) + (r & 3)) * a.N, dz[0][r]);
                 }
               });
      if (liveRow) {
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(dzp + (int64_t)(32 + 8 * (r >> 2) + (r & 3)) * a.N, dz[1][r]);
      }
    }
  } else {
    // =============================== weight-gradient wave ===============================
    uint32_t seen = 0;
    for (int tile = chunk * 4 + sw; tile < ntiles; tile += 4 * nchunks) {
      f32x16 rd[2];
      Frag3 tNf[2][2];
      // ---- t ----
      lds_wait_ge(f_pub, seen + 1, a.err);
      rd[0] = tr_read(0);
      rd[1] = tr_read(1);
      lds_post(f_con, ++seen);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) { tNf[mb][0] = afrag(rd[mb], 0); tNf[mb][1] = afrag(rd[mb], 1); }
      // ---- gmu: column sums, dW2[d][m] += sum_n gmu[n][d] t[n][m] ----
      lds_wait_ge(f_pub, seen + 1, a.err);
      rd[0] = tr_read(0);
      rd[1] = tr_read(1);
      lds_post(f_con, ++seen);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1 += rd[i][r]; s2 = fmaf(rd[i][r], rd[i][r], s2); }
        sg1[i] += s1;
        sg2[i] += s2;
        const Frag3 g0 = afrag(rd[i], 0), g1 = afrag(rd[i], 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          x3mm(dW2a[i][j], g0, tNf[j][0]);
          x3mm(dW2a[i][j], g1, tNf[j][1]);
        }
      }
      // ---- da1: column sums, P[m][k] += sum_n da1[n][m] z[n][k] (z's fragments requested before the wait) ----
      Frag3 zn[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) zn[f] = zfrag_of(tile, 4 + f);
      lds_wait_ge(f_pub, seen + 1, a.err);
      rd[0] = tr_read(0);
      rd[1] = tr_read(1);
      lds_post(f_con, ++seen);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float sd = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sd += rd[i][r];
        sd1[i] += sd;
        const Frag3 d0 = afrag(rd[i], 0), d1 = afrag(rd[i], 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          x3mm(Pa[i][j], d0, zn[2 * j]);
          x3mm(Pa[i][j], d1, zn[2 * j + 1]);
        }
      }
    }
  }
  // per-column sums: the two lane halves of a column are added through LDS, lane L then holds column L
  __syncthreads();
  float* tmp = scr + 2 * 64 * LW + sw * 384;
  if (!chainw) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      tmp[(0 * 2 + b) * 64 + hh * 32 + l31] = sg1[b];
      tmp[(1 * 2 + b) * 64 + hh * 32 + l31] = sg2[b];
      tmp[(2 * 2 + b) * 64 + hh * 32 + l31] = sd1[b];
    }
  }
  __syncthreads();
  const int cb = lane >> 5;
  const float cs_gmu = tmp[(0 * 2 + cb) * 64 + l31] + tmp[(0 * 2 + cb) * 64 + 32 + l31];
  const float cs_gmu2 = tmp[(1 * 2 + cb) * 64 + l31] + tmp[(1 * 2 + cb) * 64 + 32 + l31];
  const float cs_da1 = tmp[(2 * 2 + cb) * 64 + l31] + tmp[(2 * 2 + cb) * 64 + 32 + l31];
  dec64_finish(a, g, chunk, tid & 255, sw, lane, dW2a, Pa, cs_gmu, cs_gmu2, cs_da1, reinterpret_cast<float*>(img), scr,
               scr + 64 * LW, !chainw);
}


// ------------------------------------------------------------------------------------------------------------
// dec_x3_64g: dec_x3_64p on two fp16 terms per operand, with TWO groups per workgroup so that the sum over groups of
// dz starts on chip.  (Its one-group predecessor dec_x3_64h, round 1's default, is gone: git history)
// fp16 terms: every chain operand is scaled by a power of two taken from its maximum -- the weight images from
// reductions at set-up, z from a reduction before its image is built (rng[5]), t by 2^13 -- or from a bound of it:
// |mu - x| <= max_d(|b2| + sum_m |W2|) + max|X| (rng[0]), |gmu| <= max c3 times that, |da1| <= max_m sum_d |W2| times
// that.  A bound may be loose by 2^13 before precision suffers (the scaled maximum only has to stay above 2^0).
// 96 chain and 48 weight-gradient matrix instructions per row tile and group, a 6-instruction split per operand pair;
// the weight-gradient accumulators hold sums in scaled units and are scaled back once before the end of the kernel.
// dz = sum_g is the one reduction of the decoder that crosses workgroups; as float atomics it was 17.2 GB of adds
// per launch at the chip-wide atomic rate (1.27 of 1.3 TB/s: tools/dec_ablate.py, 13.5 ms with the adds, 10.8
// without).  Moving a group's dz tile between waves costs as much as the atomics it saves (measured: a rendezvous
// of two chain waves through LDS 1.2 ms per launch, a fourth published tile drained by the weight-gradient waves
// 3 ms, ds_add_f32 into a shared LDS tile 17 ms), so here nothing moves: ONE chain wave runs the chains of both
// groups of its workgroup for the same row tile, one after the other, and the second group's dz product
// accumulates on top of the first's in the same registers (rescaled by the exact power of two between the two
// groups' units).  32 atomics per tile and PAIR of groups: half the atomic bytes, no extra traffic.
//   chain wave c (0..3): row tiles chunk*4 + c, ...; groups in the order (c & 1), 1 - (c & 1); six published tiles
//     per row tile (t, gmu, da1 of each group) through its one tile buffer;
//   weight-gradient wave (p, grp) = 2 p + grp: group grp's tiles of chain waves 2 p + grp (which does grp first)
//     and 2 p + 1 - grp (second), alternately, so that both consumers of a chain pair are busy all the time; it holds
//     group grp's dW2 and P partials over those two row streams.
// LDS for two groups' weights: the image of A^T is gone, the dz product reads the forward image of A with
// ds_read_b64_tr_b16 (tools/probe/tr_image_probe.hip: same fragments bit for bit; no measurable cost), which leaves
// 3 x 16 KB per group; kC moves from the image into the tanh argument's scale.
// The chain waves also record max|x| of what they read (a.rng_seen): the host's guard compares its power of two
// with the one the scales were built from (a.rng[0]) and repeats the launch when they differ, so the result never
// depends on a stale range (launch_decoder_mfma).
// ------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(512, 1) void dec_x3_64g(MfmaDecArgs a) {
  constexpr int K = 64, H = 64, M = 64, DG = 64, LW = 65;
  constexpr float kC = 2.885390081777927f;  // 2*log2(e): tanh(x) = 1 - 2/(exp2(kC x) + 1)
  constexpr float kSt = 8192.0f;            // t lies in (-1, 1)
  if (a.gate != nullptr && *a.gate == 0) return;
  OIVAE_PROBE(0);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint32_t* img = reinterpret_cast<uint32_t*>(smem_raw);
  float* vec = reinterpret_cast<float*>(img + kPairImg);
  float* scr = vec + kPairVec;
  volatile uint32_t* flags = reinterpret_cast<volatile uint32_t*>(scr + kPairScr);
  // flags: [2 c], [2 c + 1] tiles published / consumed of chain wave c; [16 + 5 q ..] group q's maxima;
  // [26] max|x| seen by this workgroup
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int pairIdx = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
  const bool chainw = wave < 4;
  const int sw = wave & 3;
  const int ngrp = (2 * pairIdx + 1 < a.G) ? 2 : 1;   // G odd: the last workgroup has one group
  // per group of the pair (index q): scales of the chain (chain waves) and of the accumulated sums (weight-gradient waves)
  float inv1[2] = {1.f, 1.f}, inv2[2] = {1.f, 1.f}, inv4[2] = {1.f, 1.f}, omtC[2] = {1.f, 1.f};
  float invW[2] = {1.f, 1.f}, invP[2] = {1.f, 1.f}, invG[2] = {1.f, 1.f}, invD[2] = {1.f, 1.f};
  if (tid < 64) flags[tid] = 0;
  // the three weight matrices of both groups: requested up front, 16 bytes per lane (the latency of global memory is
  // paid once per workgroup, not once per loop iteration and group)
  f32x4 pre[2][3][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int gq = (2 * pairIdx + q < a.G) ? 2 * pairIdx + q : 2 * pairIdx;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = 4 * (tid + 512 * j);
      pre[q][0][j] = *reinterpret_cast<const f32x4*>(a.W1 + (int64_t)gq * M * H + e);
      pre[q][1][j] = *reinterpret_cast<const f32x4*>(a.Ws + (int64_t)gq * K * H + e);
      pre[q][2][j] = *reinterpret_cast<const f32x4*>(a.W2 + (int64_t)gq * DG * M + e);
    }
  }
  for (int q = 0; q < 2; ++q) {   // set-up of the pair's groups, one after the other through the same staging
    const int gq = 2 * pairIdx + q;
    if (gq >= a.G) break;
    float* sb1 = vec + 192 * q;
    float* sb2 = sb1 + 64;
    float* sC3 = sb2 + 64;
    float* tW1 = scr;
    float* tWs = scr + 64 * LW;
    float* tW2 = scr + 2 * 64 * LW;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = 4 * (tid + 512 * j);
      const int o = (e >> 6) * LW + (e & 63);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tW1[o + i] = q ? pre[1][0][j][i] : pre[0][0][j][i];
        tWs[o + i] = q ? pre[1][1][j][i] : pre[0][1][j][i];
        tW2[o + i] = q ? pre[1][2][j][i] : pre[0][2][j][i];
      }
    }
    if (tid < 64) {
      const float sig = expf(a.ls[(int64_t)gq * DG + tid]) + a.sig_off;
      sb1[tid] = a.b1[(int64_t)gq * M + tid];
      sb2[tid] = a.b2[(int64_t)gq * DG + tid];
      sC3[tid] = (a.inv / sig) / sig;
    }
    __syncthreads();
    // A[m][k] = sum_h W1[m][h] Ws[k][h] on the fp32 matrix instruction, one 32 x 32 block per chain wave
    f32x16 acc = (f32x16)(0.f);
    const int mt = sw >> 1, kt = sw & 1;
    if (chainw) {
      const float* pa = tW1 + (32 * mt + l31) * LW + hh;
      const float* pb = tWs + (32 * kt + l31) * LW + hh;
#pragma unroll 8
      for (int k2 = 0; k2 < 32; ++k2) acc = mf32(pa[2 * k2], pb[2 * k2], acc);
    }
    __syncthreads();
    if (chainw) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tW1[(32 * mt + OIVAE_ROW(r, hh)) * LW + 32 * kt + l31] = acc[r];
    }
    __syncthreads();
    const float* A = tW1;
    // ranges: [0] max|A|  [1] max|W2|  [2] bound of |gmu|  [3] bound of |da1|.  The bounds are formed dimension by
    // dimension -- |gmu_d| <= c3_d (|b2_d| + sum_m |W2_dm| + max|X|), |da1_m| <= sum_d |W2_dm| (bound of gmu_d) -- and not
    // as products of maxima over different dimensions: one output dimension with a huge weight row and a wide sigma
    // beside ordinary ones made that product loose by 2^19, more than the 2^13 the fp16 terms can absorb
    // (tests/test_gpu_x3.py, case huge_w2_row).
    uint32_t* red = const_cast<uint32_t*>(flags) + 16 + 5 * q;
    float* gbound = scr + 3 * 64 * LW;   // [64]
    {
      float mA = 0.f, mW = 0.f;
      for (int i = tid; i < 4096; i += 512) {
        const int o = (i >> 6) * LW + (i & 63);
        mA = fmaxf(mA, fabsf(A[o]));
        mW = fmaxf(mW, fabsf(tW2[o]));
      }
      float gmx = 0.f;
      {   // eight lanes per output dimension d: sum_m |W2[d][m]|
        const int d = tid >> 3, part = tid & 7;
        float rs = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) rs += fabsf(tW2[d * LW + 8 * m + part]);
        rs += __shfl_xor(rs, 1);
        rs += __shfl_xor(rs, 2);
        rs += __shfl_xor(rs, 4);
        if (part == 0) {
          // max|X| enters as the upper edge of its binade, 2^(E+1): that is all the range guard verifies about the
          // recorded maximum, so the scales -- and with them the result's bits -- depend on nothing else of it
          const float xedge = __uint_as_float(((a.rng[0] & 0x7f800000u) < 0x7f000000u ? (a.rng[0] & 0x7f800000u) + 0x00800000u : 0x7f000000u));
          const float gb = sC3[d] * (rs + fabsf(sb2[d]) + xedge);
          gbound[d] = gb;
          gmx = gb;
        }
      }
      // one LDS atomic per wave and maximum (see dec_x3_64s.h: 512 lanes on one LDS word cost 8 us per group)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        mA = fmaxf(mA, __shfl_xor(mA, o));
        mW = fmaxf(mW, __shfl_xor(mW, o));
        gmx = fmaxf(gmx, __shfl_xor(gmx, o));
      }
      if (lane == 0) {
        atomicMax(red + 0, __float_as_uint(mA));
        atomicMax(red + 1, __float_as_uint(mW));
        atomicMax(red + 2, __float_as_uint(gmx));
      }
    }
    __syncthreads();
    {   // eight lanes per hidden unit m: sum_d |W2[d][m]| gbound[d]
      const int m = tid & 63, part = tid >> 6;
      float cs = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) cs += fabsf(tW2[(8 * part + d) * LW + m]) * gbound[8 * part + d];
      float* csp = scr + 3 * 64 * LW + 64;   // [8][64] partials behind gbound
      csp[part * 64 + m] = cs;
    }
    __syncthreads();
    if (tid < 64) {
      const float* csp = scr + 3 * 64 * LW + 64;
      float cs = 0.f;
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8) cs += csp[p8 * 64 + tid];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cs = fmaxf(cs, __shfl_xor(cs, o));
      if (lane == 0) atomicMax(red + 3, __float_as_uint(cs));
    }
    __syncthreads();
    float q1, q2, qz, qg, qd, i1, i2, iz, ig, id;
    {
      const float mA = __uint_as_float(red[0]), mW = __uint_as_float(red[1]);
      const float gmax = __uint_as_float(red[2]), dmax = __uint_as_float(red[3]);
      h2_scale_of(__float_as_uint(mA), &q1, &i1);
      h2_scale_of(__float_as_uint(mW), &q2, &i2);
      h2_scale_of(a.rng[5], &qz, &iz);
      h2_scale_of(__float_as_uint(gmax), &qg, &ig);
      h2_scale_of(__float_as_uint(dmax), &qd, &id);
    }
    auto uni = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };   // scalar registers
    inv1[q] = uni(kC * (i1 * iz)); inv2[q] = uni(i2 * (1.0f / kSt)); inv4[q] = uni(i1 * id);
    invW[q] = uni(ig * (1.0f / kSt)); invP[q] = uni(id * iz);
    invG[q] = uni(ig); invD[q] = uni(id); omtC[q] = uni((i2 * ig) * qd);
    __syncthreads();   // every thread has read the maxima before the biases change
    if (tid < 64) {
      sb1[tid] *= q1 * qz;    // accumulator initial values in the products' scaled units
      sb2[tid] *= q2 * kSt;
      sC3[tid] *= -qg;        // gmu is produced in its operand's scaled units; the chain multiplies (x - mu) by - c3
    }
    {  // the three images in two fp16 terms: item = (row r, step ks, lane half h2) -> 8 slots -> two 16-byte terms
      const int r = tid >> 3, ks = (tid >> 1) & 3, h2 = tid & 1;
      float v[3][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = x3_slot_feature(ks, h2, e);
        v[0][e] = q1 * A[r * LW + c];
        v[1][e] = q2 * tW2[r * LW + c];
        v[2][e] = q2 * tW2[c * LW + r];
      }
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const Frag2 f = split8_h2(v[w][0], v[w][1], v[w][2], v[w][3], v[w][4], v[w][5], v[w][6], v[w][7]);
#pragma unroll
        for (int p = 0; p < 2; ++p)
          *reinterpret_cast<x3_u32x4*>(img + (q * 3 + w) * 4096 + ((((r >> 5) * 4 + ks) * 2 + p) * 64 + h2 * 32 + (r & 31)) * 4) = f.p[p];
      }
    }
    OIVAE_PROBE(1 + q);
  }
  __syncthreads();
  OIVAE_PROBE(3);
  auto wfrag = [&](const uint32_t* imgl, int which, int rb, int ks) {
    Frag2 f;
#pragma unroll
    for (int p = 0; p < 2; ++p) f.p[p] = *reinterpret_cast<const x3_u32x4*>(imgl + which * 4096 + ((rb * 4 + ks) * 2 + p) * 256);
    return f;
  };
  // The dz product needs A with lane = latent k and slots = hidden m.  Transposed reads of the forward image: per 16
  // lanes a block of 4 rows (m) x 16 columns (k); lane 4 qq + pp of the group addresses (row qq, columns 4 pp .. +3),
  // lane i receives column i.  Group (lane >> 4): k = 32 rb + 16 (group & 1) + i, m = 16 ks + 8 j + 4 hh + qq.
  const int tr_lane = ((lane >> 4) & 1) * 2048 + ((lane & 1) * 32 + 4 * hh + ((lane >> 2) & 3)) * 16 + ((lane >> 1) & 1) * 8;
  auto wfrag_tr = [&](const char* trb, int rb, int ks) {
    Frag2 f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const x3_u32x2 v = lds_read_tr16(trb + ((((ks >> 1) * 4 + 2 * rb) * 2 + p) * 1024 + (16 * (ks & 1) + 8 * j) * 16));
        f.p[p][2 * j] = v[0];
        f.p[p][2 * j + 1] = v[1];
      }
    return f;
  };
  auto afrag2n = [&](const f32x16& v, int q) {   // values already in scaled units
#ifdef OIVAE_ABL_NOCSPLIT   // timing only: the splits replaced by one bit operation per pair
    Frag2 f;
    for (int e = 0; e < 4; ++e) { f.p[0][e] = __float_as_uint(v[8 * q + e]) & 0x3bff3bffu; f.p[1][e] = __float_as_uint(v[8 * q + 4 + e]) & 0x3bff3bffu; }
    return f;
#else
    return split8_h2(v[8 * q], v[8 * q + 1], v[8 * q + 2], v[8 * q + 3], v[8 * q + 4], v[8 * q + 5], v[8 * q + 6], v[8 * q + 7]);
#endif
  };
  // One chained product, both 32-row output blocks, as eight pinned units (u = 4 rb + ks) of three MFMAs: the weight
  // fragments of unit u+2 are requested and the operand of step u+1 is split inside unit u's region (dec_x3_64p).
  auto chain_mm = [&](f32x16* acc, auto&& wf, Frag2* bf, auto&& make, auto&& side) {
    Frag2 w[3];
    w[0] = wf(0, 0);
    w[1] = wf(0, 1);
    bf[0] = make(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u + 2 < 8) w[(u + 2) % 3] = wf((u + 2) >> 2, (u + 2) & 3);
      if (u + 1 < 4) bf[u + 1] = make(u + 1);
      side(u);
#ifdef OIVAE_ABL_NOCMFMA   // timing only: the chain wave without its matrix instructions
      acc[u >> 2][u & 15] += __uint_as_float((w[u % 3].p[0][0] ^ bf[u & 3].p[1][0] ^ w[u % 3].p[1][3] ^ bf[u & 3].p[0][3]) & 0x3fffffffu);
#elif defined(OIVAE_ABL_C16)
      h2mm_as16(acc[u >> 2], w[u % 3], bf[u & 3]);
#else
      h2mm(acc[u >> 2], w[u % 3], bf[u & 3]);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#ifdef OIVAE_ABL_CHAIN2
  // variant: the two accumulators of a product alternate, so that consecutive matrix instructions are independent
  auto chain_mm2 = [&](f32x16* acc, auto&& wf, Frag2* bf, auto&& make, auto&& side) {
    Frag2 w[2][2];
    w[0][0] = wf(0, 0);
    w[0][1] = wf(1, 0);
    bf[0] = make(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) {
        w[(ks + 1) & 1][0] = wf(0, ks + 1);
        w[(ks + 1) & 1][1] = wf(1, ks + 1);
        bf[ks + 1] = make(ks + 1);
      }
      const Frag2& a0 = w[ks & 1][0];
      const Frag2& a1 = w[ks & 1][1];
      acc[0] = mfma_f16(a0.p[1], bf[ks].p[0], acc[0]);
      acc[1] = mfma_f16(a1.p[1], bf[ks].p[0], acc[1]);
      acc[0] = mfma_f16(a0.p[0], bf[ks].p[1], acc[0]);
      acc[1] = mfma_f16(a1.p[0], bf[ks].p[1], acc[1]);
      acc[0] = mfma_f16(a0.p[0], bf[ks].p[0], acc[0]);
      acc[1] = mfma_f16(a1.p[0], bf[ks].p[0], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 4; u < 8; ++u) side(u);
  };
#define chain_mm chain_mm2
#endif
  auto init_t = [&](f32x16& acc, const float* v0, int rb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(v0 + 32 * rb + 8 * q + 4 * hh);
      acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
    }
  };
  auto zfrag2_of = [&](int t, int f) {   // two-term fragments of z * sz (x3_split_z with ranges)
    Frag2 r;
    const uint32_t* zi = a.zimg + ((int64_t)t * 24 * 64 + lane) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) r.p[p] = *reinterpret_cast<const x3_u32x4*>(zi + (f * 2 + p) * 256);
    return r;
  };
  // a chain wave's fp32 tile [32 rows][68]: written 16 bytes at a time with lane = batch row, read with lane = feature
  constexpr int LT = 68;

  const int N32 = (int)a.N;
  const int ntiles = (N32 + 31) / 32;
  const bool one_sample = (a.N == a.B);
  f32x16 dW2a[2][2], Pa[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { dW2a[i][j] = (f32x16)(0.f); Pa[i][j] = (f32x16)(0.f); }
  float sg1[2] = {0.f, 0.f}, sg2[2] = {0.f, 0.f}, sd1[2] = {0.f, 0.f};
  float xm = 0.f;
  const int wgrp = sw & 1, wp = sw >> 1;   // weight-gradient wave (wp, wgrp): group wgrp of chain waves 2 wp, 2 wp + 1
  const bool active = chainw || wgrp < ngrp;

  if (chainw) {
    // =============================== chain wave ===============================
    __builtin_amdgcn_s_setprio(3);   // the critical path of the pair: issue before the other wave of the SIMD
    float* trw = scr + sw * kPairTile + l31 * LT + 4 * hh;
    auto tr_write = [&](const f32x16& v, int rb) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(trw + 32 * rb + 8 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    };
    volatile uint32_t* f_pub = flags + 2 * sw;
    volatile uint32_t* f_con = flags + 2 * sw + 1;
    uint32_t pub = 0;
    // Hand-over of a tile, chain side.  The "consumed" counter is requested BEFORE the elementwise work that produces
    // the tile and looked at after it (its LDS latency is hidden; the polling loop only runs when the other wave is
    // late), and the "published" counter is bumped right behind the tile's stores with no wait in between: a wave's
    // LDS instructions are performed in order, so whoever sees the counter sees the tile.
    typedef const volatile __attribute__((address_space(3))) uint32_t* lds_cu32;
    typedef volatile __attribute__((address_space(3))) uint32_t* lds_u32;
    auto con_peek = [&]() { return *(lds_cu32)f_con; };
    auto publish = [&](const f32x16& v0, const f32x16& v1, uint32_t peeked) {
      if ((uint32_t)__builtin_amdgcn_readfirstlane(peeked) < pub) lds_wait_ge(f_con, pub, a.err);
      tr_write(v0, 0);
      tr_write(v1, 1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      *(lds_u32)f_pub = ++pub;
      __builtin_amdgcn_sched_barrier(0);   // or the scheduler sinks the store below the MFMAs that follow: the partner waits
    };
    char* const dzb = reinterpret_cast<char*>(a.dzT);
    const int64_t rowbytes = a.N * 4;
    Frag2 zf[4];
    if (chunk * 4 + sw < ntiles) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) zf[ks] = zfrag2_of(chunk * 4 + sw, ks);
    }
    for (int tile = chunk * 4 + sw; tile < ntiles; tile += 4 * nchunks) {
      const int tnext = (tile + 4 * nchunks < ntiles) ? tile + 4 * nchunks : tile;
      const int r0 = tile * 32;
      const int nT = (r0 + l31 < N32) ? r0 + l31 : N32 - 1;
      const float liveT = (r0 + l31 < N32) ? 1.0f : 0.0f;
      const bool partial = r0 + 32 > N32;
#ifdef OIVAE_ABL_NOX    // timing only: every tile reads the same rows of X (L2 hits)
      const float* xrow = a.X + (int64_t)l31 * (int64_t)a.D + 4 * hh;
#else
      const float* xrow = a.X + (one_sample ? (int64_t)nT : (int64_t)nT % a.B) * (int64_t)a.D + 4 * hh;
#endif
      f32x16 dz[2];
      dz[0] = (f32x16)(0.f);
      dz[1] = (f32x16)(0.f);
#pragma unroll 1
      for (int h = 0; h < ngrp; ++h) {
        const int q = (ngrp == 2) ? ((sw & 1) ^ h) : 0;   // the group of the pair this half of the tile works on
        const uint32_t* imgl = img + q * 3 * 4096 + lane * 4;
        const char* trb = reinterpret_cast<const char*>(img + q * 3 * 4096) + tr_lane;
        const float* sb1 = vec + 192 * q;
        const float* sb2 = sb1 + 64;
        const float* sC3 = sb2 + 64;
        const float k_inv1 = q ? inv1[1] : inv1[0], k_ninv2 = -(q ? inv2[1] : inv2[0]), k_omtC = q ? omtC[1] : omtC[0];
        const float k_nomt2 = -(k_omtC / (kSt * kSt));
        const float* xT = xrow + (int64_t)(2 * pairIdx + q) * DG;
        f32x4 xq[2][4];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) xq[rb][qq] = *reinterpret_cast<const f32x4*>(xT + 32 * rb + 8 * qq);
        // ---- a1 + b1 in units of 1 / (s1 sz) ----
        f32x16 uT[2];
        init_t(uT[0], sb1, 0);
        init_t(uT[1], sb1, 1);
        auto none = [&](int) {};
        chain_mm(uT, [&](int rb, int ks) { return wfrag(imgl, 0, rb, ks); }, zf, [&](int ks) { return zf[ks]; }, none);
        // ---- t = tanh: published first, split during the next product ----
        // (whole-vector expressions: the compiler turns them into packed fp32 instructions, two elements each)
        uint32_t peek = con_peek();
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#ifdef OIVAE_ABL_NOTANH
          uT[rb] = uT[rb] * k_inv1 + kSt;
#else
          f32x16 e = uT[rb] * k_inv1;
#pragma unroll
          for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(e[r]);
          e = e + 1.0f;
#pragma unroll
          for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_rcpf(e[r]);
          uT[rb] = e * (-2.0f * kSt) + kSt;   // 2^13 t
#endif
        }
#ifndef OIVAE_ABL_NOPUB
        publish(uT[0], uT[1], peek);
#endif
        // ---- mu = W2 t + b2; gmu = (mu - x) inv / sigma^2; afterwards uT holds inv3 sd (1 - t^2) ----
        Frag2 tf[4];
        f32x16 gT[2];
        init_t(gT[0], sb2, 0);
        init_t(gT[1], sb2, 1);
        auto gmu_quarter = [&](int rb, int qq) {
          f32x4 c3n = *reinterpret_cast<const f32x4*>(sC3 + 32 * rb + 8 * qq + 4 * hh);   // - c3 in scaled units
          if (partial) {   // rows past the end of the batch contribute nothing (a real, uniform branch: last tile only)
            asm volatile("");
            c3n *= liveT;
          }
          const f32x4 x4 = xq[rb][qq];
          asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(xm) : "v"(x4[0]), "v"(x4[1]));
          asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(xm) : "v"(x4[2]), "v"(x4[3]));
          if (rb == 0) {   // inside the product: scalar instructions (see below)
#pragma unroll
            for (int i = 0; i < 4; ++i) gT[rb][4 * qq + i] = fmaf(gT[rb][4 * qq + i], k_ninv2, x4[i]) * c3n[i];
            return;
          }
          f32x4 gq4 = {gT[rb][4 * qq], gT[rb][4 * qq + 1], gT[rb][4 * qq + 2], gT[rb][4 * qq + 3]};
          gq4 = (gq4 * k_ninv2 + x4) * c3n;   // (mu - x) c3 = (x - mu) (- c3)
#pragma unroll
          for (int i = 0; i < 4; ++i) gT[rb][4 * qq + i] = gq4[i];
        };
        chain_mm(gT, [&](int rb, int ks) { return wfrag(imgl, 1, rb, ks); }, tf,
                 [&](int ks) {
                   const Frag2 f = afrag2n(uT[ks >> 1], ks & 1);
                   if (ks & 1) {   // (scalar on purpose: packed fp32 instructions between the matrix instructions of a product cost time)
#pragma unroll
                     for (int r = 0; r < 16; ++r) uT[ks >> 1][r] = fmaf(k_nomt2 * uT[ks >> 1][r], uT[ks >> 1][r], k_omtC);
                   }
                   return f;
                 },
                 [&](int u) { if (u >= 4) gmu_quarter(0, u - 4); });
        peek = con_peek();
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) gmu_quarter(1, qq);
#ifndef OIVAE_ABL_NOPUB
        publish(gT[0], gT[1], peek);
#endif
        // ---- dt = W2^T gmu; da1 = dt (1 - t^2) ----
        Frag2 gf[4];
        f32x16 dT[2];
        dT[0] = (f32x16)(0.f);
        dT[1] = (f32x16)(0.f);
        chain_mm(dT, [&](int rb, int ks) { return wfrag(imgl, 2, rb, ks); }, gf, [&](int ks) { return afrag2n(gT[ks >> 1], ks & 1); },
                 [&](int u) {
                   if (u >= 4) {
#pragma unroll
                     for (int r = 4 * (u - 4); r < 4 * (u - 4) + 4; ++r) dT[0][r] *= uT[0][r];
                   }
                 });
        peek = con_peek();
        dT[1] *= uT[1];
#ifndef OIVAE_ABL_NOPUB
        publish(dT[0], dT[1], peek);
#endif
        {   // the next first product's operand, one product ahead: the same row tile's for the pair's second group
          const int tz = (h + 1 < ngrp) ? tile : tnext;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) zf[ks] = zfrag2_of(tz, ks);
        }
        // ---- dz^T[k][n] += sum_m A[m][k] da1[n][m], on top of the pair's first group (brought to this group's units) ----
        if (h == 1) {
          const float ratio = (q ? inv4[0] : inv4[1]) / (q ? inv4[1] : inv4[0]);   // powers of two: exact
          dz[0] *= ratio;
          dz[1] *= ratio;
        }
        Frag2 df[4];
        const bool last = h + 1 == ngrp;
        const float k_inv4 = q ? inv4[1] : inv4[0];
        // dz^T[k][n] lives at dzT + k N + n: uniform row base (scalar registers) + 32-bit per-lane byte offset
        // (the row bases are formed next to their use: hoisted out of this loop, the 32 addresses were spilled)
        uint32_t voff = (uint32_t)((int64_t)(4 * hh) * a.N + nT) * 4u;
        asm volatile("" : "+v"(voff));
        const bool liveRow = r0 + l31 < N32;
        auto dz_row = [&](int row) { return reinterpret_cast<float*>(dzb + (int64_t)row * rowbytes + voff); };
        chain_mm(dz, [&](int rb, int ks) { return wfrag_tr(trb, rb, ks); }, df, [&](int ks) { return afrag2n(dT[ks >> 1], ks & 1); },
                 [&](int u) {
                   if (u >= 4 && last && liveRow) {
#pragma unroll
                     for (int r = 4 * (u - 4); r < 4 * (u - 4) + 4; ++r) OIVAE_DZ_ADD(dz_row(8 * (r >> 2) + (r & 3)), dz[0][r] * k_inv4);
                   }
                 });
        if (last && liveRow) {
          const f32x16 o16 = dz[1] * k_inv4;
#pragma unroll
          for (int r = 0; r < 16; ++r) OIVAE_DZ_ADD(dz_row(32 + 8 * (r >> 2) + (r & 3)), o16[r]);
        }
      }
    }
  } else if (active) {
    // =============================== weight-gradient wave ===============================
    // producer j = 0: chain wave 2 wp + wgrp (this group is the first half of its row tile), j = 1: the other one
    uint32_t seen[2] = {0, 0};
    const int npub = 3 * ngrp;
    for (int t0 = chunk * 4 + 2 * wp; t0 < ntiles; t0 += 4 * nchunks) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 2 * wp + (j ^ wgrp);           // the chain wave
        const int tile = t0 - 2 * wp + c;             // its row tile of this round
        if (tile >= ntiles) continue;
        const float* trr = scr + c * kPairTile + (4 * hh) * LT + l31;
        auto tr_read = [&](int cb) {
          f32x16 v;
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = trr[(8 * (r >> 2) + (r & 3)) * LT + 32 * cb];
          return v;
        };
        volatile uint32_t* f_pub = flags + 2 * c;
        volatile uint32_t* f_con = flags + 2 * c + 1;
        uint32_t n = seen[j] + ((ngrp == 2 && j == 1) ? 3 : 0);   // publishes of this chain wave before this group's three
        seen[j] += npub;
        f32x16 rd[2];
        Frag2 tNf[2][2];
#ifdef OIVAE_ABL_NOWSPLIT   // timing only: the weight-gradient wave's splits replaced by one bit operation per pair
        auto afrag2n = [&](const f32x16& v, int q) {
          Frag2 f;
          for (int e = 0; e < 4; ++e) { f.p[0][e] = __float_as_uint(v[8 * q + e]) & 0x3bff3bffu; f.p[1][e] = __float_as_uint(v[8 * q + 4 + e]) & 0x3bff3bffu; }
          return f;
        };
#endif
#ifdef OIVAE_ABL_W16
        auto h2mm = [&](f32x16& acc, const Frag2& A, const Frag2& B) { h2mm_as16(acc, A, B); };
#endif
#ifdef OIVAE_ABL_NOWMFMA   // timing only: the weight-gradient wave without its matrix instructions
        auto h2mm = [&](f32x16& acc, const Frag2& A, const Frag2& B) {
          acc[0] += __uint_as_float((A.p[0][0] ^ B.p[1][0] ^ A.p[1][3] ^ B.p[0][3] ^ A.p[0][1] ^ A.p[0][2] ^ A.p[1][1] ^ A.p[1][2] ^ B.p[0][1] ^ B.p[0][2] ^ B.p[1][1] ^ B.p[1][2]) & 0x3fffffffu);
        };
#endif
        // ---- t ----
#ifndef OIVAE_ABL_NOPUB
        lds_wait_ge(f_pub, n + 1, a.err);
#endif
        rd[0] = tr_read(0);
        rd[1] = tr_read(1);
#ifndef OIVAE_ABL_NOPUB
        lds_post(f_con, ++n);
#endif
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) { tNf[mb][0] = afrag2n(rd[mb], 0); tNf[mb][1] = afrag2n(rd[mb], 1); }
        // ---- gmu: column sums, dW2[d][m] += sum_n gmu[n][d] t[n][m] ----
#ifndef OIVAE_ABL_NOPUB
        lds_wait_ge(f_pub, n + 1, a.err);
#endif
        rd[0] = tr_read(0);
        rd[1] = tr_read(1);
#ifndef OIVAE_ABL_NOPUB
        lds_post(f_con, ++n);
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float c1 = 0.f, c2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { c1 += rd[i][r]; c2 = fmaf(rd[i][r], rd[i][r], c2); }
          sg1[i] += c1;
          sg2[i] += c2;
          const Frag2 g0 = afrag2n(rd[i], 0), g1 = afrag2n(rd[i], 1);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            h2mm(dW2a[i][jj], g0, tNf[jj][0]);
            h2mm(dW2a[i][jj], g1, tNf[jj][1]);
          }
        }
        // ---- da1: column sums, P[m][k] += sum_n da1[n][m] z[n][k] (z's fragments requested before the wait) ----
        Frag2 zn[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) zn[f] = zfrag2_of(tile, 4 + f);
#ifndef OIVAE_ABL_NOPUB
        lds_wait_ge(f_pub, n + 1, a.err);
#endif
        rd[0] = tr_read(0);
        rd[1] = tr_read(1);
#ifndef OIVAE_ABL_NOPUB
        lds_post(f_con, ++n);
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float csum = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) csum += rd[i][r];
          sd1[i] += csum;
          const Frag2 d0 = afrag2n(rd[i], 0), d1 = afrag2n(rd[i], 1);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            h2mm(Pa[i][jj], d0, zn[2 * jj]);
            h2mm(Pa[i][jj], d1, zn[2 * jj + 1]);
          }
        }
      }
    }
    {   // the accumulated weight gradients back into true units
      const float kW = wgrp ? invW[1] : invW[0], kP = wgrp ? invP[1] : invP[0];
      const float kG = wgrp ? invG[1] : invG[0], kD = wgrp ? invD[1] : invD[0];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { dW2a[i][j] *= kW; Pa[i][j] *= kP; }
#pragma unroll
      for (int i = 0; i < 2; ++i) { sg1[i] *= kG; sg2[i] *= kG * kG; sd1[i] *= kD; }   // the tiles were in scaled units
    }
  }
  // ---- end of the kernel ----
  OIVAE_PROBE(4);
  __syncthreads();
  OIVAE_PROBE(5);
  // max|x| of the rows and columns this workgroup read: one atomic per workgroup
  if (chainw) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xm = fmaxf(xm, __shfl_xor(xm, o));
    if (lane == 0) atomicMax(const_cast<uint32_t*>(flags) + 26, __float_as_uint(xm));
  }
  // per-column sums: the two lane halves of a column are added through LDS, lane L then holds column L
  float* tmp = scr + sw * 384;
  if (!chainw) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      tmp[(0 * 2 + b) * 64 + hh * 32 + l31] = sg1[b];
      tmp[(1 * 2 + b) * 64 + hh * 32 + l31] = sg2[b];
      tmp[(2 * 2 + b) * 64 + hh * 32 + l31] = sd1[b];
    }
  }
  __syncthreads();
  if (tid == 0 && a.rng_seen != nullptr) atomicMax(a.rng_seen, flags[26]);
  const int cb = lane >> 5;
  const float cs_gmu = tmp[(0 * 2 + cb) * 64 + l31] + tmp[(0 * 2 + cb) * 64 + 32 + l31];
  const float cs_gmu2 = tmp[(1 * 2 + cb) * 64 + l31] + tmp[(1 * 2 + cb) * 64 + 32 + l31];
  const float cs_da1 = tmp[(2 * 2 + cb) * 64 + l31] + tmp[(2 * 2 + cb) * 64 + 32 + l31];
  // Per group: the two streams' partials of dW2 and P summed in a fixed order, dW1 = P Ws and dWs = P^T W1 on the
  // fp32 matrix instruction (the four weight-gradient waves take a quadrant each), column sums -> db2, dlog_sigma,
  // db1 and the group's log-likelihood partial.  Staging in the image area: [2][4096] partials | Ws | P | W1 | sums.
  float* stage = reinterpret_cast<float*>(img);
  float* uWs = stage + 2 * 4096;
  float* sP = uWs + 64 * LW;
  float* uW1 = sP + 64 * LW;
  float* ssum = uW1 + 64 * LW;   // [2 streams][3][64]
  const bool worker = !chainw;
  const int wtid = tid & 255;
  const int64_t cs = a.chunk_stride * chunk;
  for (int q = 0; q < 2; ++q) {
    const int gq = 2 * pairIdx + q;
    if (gq >= a.G) break;
    const bool mine = worker && wgrp == q && active;
    const float* Wsg = a.Ws + (int64_t)gq * K * H;
    const float* W1g = a.W1 + (int64_t)gq * M * H;
    auto stage_acc = [&](f32x16 (&acc)[2][2]) {
      __syncthreads();
      if (mine) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              stage[wp * 4096 + (32 * i + OIVAE_ROW(r, hh)) * 64 + 32 * j + l31] = acc[i][j][r];
      }
      __syncthreads();
    };
    stage_acc(dW2a);
    if (worker) {
      float* dst = a.gW2 + cs + (int64_t)gq * DG * M;
      for (int idx = wtid; idx < 4096; idx += 256) dst[idx] = stage[idx] + stage[4096 + idx];
    }
    stage_acc(Pa);
    if (worker) {
      for (int idx = wtid; idx < 4096; idx += 256) {
        const int o = (idx >> 6) * LW + (idx & 63);
        sP[o] = stage[idx] + stage[4096 + idx];
        uW1[o] = W1g[idx];
        uWs[o] = Wsg[idx];
      }
    }
    if (mine) {
      ssum[wp * 192 + lane] = cs_gmu;
      ssum[wp * 192 + 64 + lane] = cs_gmu2;
      ssum[wp * 192 + 128 + lane] = cs_da1;
    }
    __syncthreads();
    if (worker) {
      const int it = sw >> 1, ht = sw & 1;
      f32x16 acc = (f32x16)(0.f);  // dW1[m][h] = sum_k P[m][k] Ws[k][h]
      const float* pa = sP + (32 * it + l31) * LW + hh;
      const float* pb = uWs + hh * LW + 32 * ht + l31;
#pragma unroll 8
      for (int k2 = 0; k2 < 32; ++k2) acc = mf32(pa[2 * k2], pb[2 * k2 * LW], acc);
      float* d1 = a.gW1 + cs + (int64_t)gq * M * H;
#pragma unroll
      for (int r = 0; r < 16; ++r) d1[(32 * it + OIVAE_ROW(r, hh)) * H + 32 * ht + l31] = acc[r];
      acc = (f32x16)(0.f);         // dWs[k][h] = sum_m P[m][k] W1[m][h]
      pa = sP + hh * LW + 32 * it + l31;
      pb = uW1 + hh * LW + 32 * ht + l31;
#pragma unroll 8
      for (int k2 = 0; k2 < 32; ++k2) acc = mf32(pa[2 * k2 * LW], pb[2 * k2 * LW], acc);
      float* d2 = a.gWs + cs + (int64_t)gq * K * H;
#pragma unroll
      for (int r = 0; r < 16; ++r) d2[(32 * it + OIVAE_ROW(r, hh)) * H + 32 * ht + l31] = acc[r];
    }
    float* sred = stage;   // the partials were consumed before the last barrier
    if (worker && wtid < 64) {
      const float s_gmu = ssum[wtid] + ssum[192 + wtid];
      const float s_gmu2 = ssum[64 + wtid] + ssum[256 + wtid];
      const float s_da1 = ssum[128 + wtid] + ssum[320 + wtid];
      const int64_t d = (int64_t)gq * DG + wtid;
      const float e = expf(a.ls[d]);
      const float sig = e + a.sig_off;
      const float c2 = a.inv / sig;
      const float sumq2 = s_gmu2 / (c2 * c2);
      const float nrows = (chunk == 0) ? (float)a.N : 0.f;
      a.gb2[cs + d] = s_gmu;
      a.gls[cs + d] = (nrows - sumq2) * (e * a.inv / sig);
      a.gb1[cs + (int64_t)gq * M + wtid] = s_da1;
      sred[wtid] = nrows * (-0.5f * kLog2Pi - logf(sig)) - 0.5f * sumq2;
    }
    __syncthreads();
    if (worker && wtid < 64) {   // one wave: 64 terms summed in double, in a fixed tree order
      double tot = (double)sred[wtid];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
      if (wtid == 0) a.ll_part[(int64_t)chunk * a.G + gq] = tot;
    }
  }
  OIVAE_PROBE(6);
}

int launch_xcheck_decoder(char which, const MfmaDecArgs& a, int G, int chunks, hipStream_t st) {
  switch (which) {
    case '4': {
      const size_t lds = (size_t)(3 * 64 * 65 + 4 * 64 + 4 * 3 * 32 * 68) * sizeof(float);
      OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_mfma64_v4),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(dec_mfma64_v4, dim3((unsigned)G, (unsigned)chunks), dim3(256), lds, st, a);
      break;
    }
    case 'x':
      OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_x3_64),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kX3DecLds));
      hipLaunchKernelGGL(dec_x3_64, dim3((unsigned)G, (unsigned)chunks), dim3(256), (size_t)kX3DecLds, st, a);
      break;
    case 'p':
      OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_x3_64p),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kX3DecLdsP));
      hipLaunchKernelGGL(dec_x3_64p, dim3((unsigned)G, (unsigned)chunks), dim3(512), (size_t)kX3DecLdsP, st, a);
      break;
    case 'g':
      OIVAE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_x3_64g),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds));
      hipLaunchKernelGGL(dec_x3_64g, dim3((unsigned)((G + 1) / 2), (unsigned)chunks), dim3(512), (size_t)kPairLds, st, a);
      break;
    default:
      return fail(OIVAE_E_ARG, "launch_xcheck_decoder: unknown kernel '%c'", which);
  }
  OIVAE_LAUNCH_CHECK();
  return OIVAE_OK;
}

}  // namespace oivae
