// dec_x3_64s: dec_x3_64g (xcheck_decoders.hip) on v_mfma_f32_16x16x32_f16.  Included by mfma_decoder.hip, inside
// namespace oivae; dec_x3_64g's arguments, LDS plan, set-up, hand-over protocol and end of kernel are shared (dec_shared.h).
//
// Why another instruction shape: the chip is power-limited under these kernels, and per flop the 16x16x32 instruction
// draws less than 32x32x16 (tools/probe/mfma_shape_power.hip: 2006 against 1772 TFLOP/s on random operands).  A timing
// stand-in that issued dec_x3_64g's matrix flops as pairs of 16x16x32 instructions on the same registers ran 10.5
// against 11.8 ms (tools/dec_ablate.py C16+W16); this kernel is that change done properly, which means new layouts
// for everything that follows an accumulator's shape.
//
// Lane (c = lane & 15, q = lane >> 4).  One instruction: D[16 x 16] += A[16 x 32] B[32 x 16]; A: lane holds row c, the
// 8 reduction slots 8 q .. 8 q + 7; B: column c, same slots; D: rows 4 q .. 4 q + 3 of column c.
//   chain (out^T[feature][row] = W[feature][k] act^T[k][row], as before): a 64-feature x 32-row result is eight
//     accumulators acc[fb][nb] of four registers: feature 16 fb + 4 q + i, row 16 nb + c.  The next product's B
//     operand for reduction step kk (32 slots) and row block nb is acc[2 kk][nb] | acc[2 kk + 1][nb] split in
//     registers, i.e. slot e of lane quarter q in step kk <-> feature 32 kk + 16 (e / 4) + 4 q + e % 4
//     (s16_slot_feature); the images of W2 and W2^T are built in that order.  The image of A = W1 Ws^T keeps the plain
//     order (slot <-> latent 32 kk + 8 q + e: z's fragments are built to match), so that the dz product can read it
//     transposed with ds_read_b64_tr_b16: per 16 lanes a 4 x 16 block, conflict-free here.
//     A product is eight pinned units (kk, fb) of six instructions (both row blocks, three terms).
//   weight gradients (sums over the 32 rows of a tile = ONE reduction step): the published fp32 tile [32 rows][68] is
//     read with lane = feature 16 fb + c and slots = rows 8 q + e, for both operands; the 16-column blocks of the tile
//     are swapped in rows 8..15 and 24..31 (column ^ 16) so that those reads meet 32 different banks per half wave
//     while the chain wave's 16-byte stores stay conflict-free.  dW2 and P partials: acc[db][mb] of four registers
//     (row 16 db + 4 q + i, column 16 mb + c), 128 persistent registers as before.
//     (Tried and dropped: handing over the chain wave's fp16 FRAGMENTS instead of the fp32 tile, read transposed by the
//     weight-gradient wave, which then neither splits nor reads fp32 and sums columns with v_dot2_f32_f16: correct,
//     11.31 against 11.09 ms -- the fragment stores tie the chain wave's splits to its LDS queue, and the
//     weight-gradient wave's vector work was never the bound.  Also measured against this form, all within 0.2 ms and
//     none faster: feature block outer / reduction step inner units, no tile swizzle, packed fp32 side work inside
//     products, one operand fragment split ahead of a product instead of two.)
//   z image (x3_split_z, lay16): per 32-row tile fragments [0..3] = (kk, nb) in the chain's orientation,
//     [4..7] = latent block kb in the weight-gradient orientation (lane = latent 16 kb + c, slots = rows 8 q + e).
//   dz adds: one instruction covers 4 latent rows x 16 consecutive batch rows (64-byte runs); the chip-wide rate of
//     float atomics is the same as for 128-byte runs (tools/probe/atomic_pattern.hip: 1.33 TB/s both ways).

__device__ __forceinline__ int s16_slot_feature(int kk, int q, int e) { return 32 * kk + 16 * (e >> 2) + 4 * q + (e & 3); }

__device__ __forceinline__ f32x4 mfma16_h(x3_u32x4 a, x3_u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_h16x8, a), __builtin_bit_cast(x3_h16x8, b), c, 0, 0, 0);
}
// acc_j += A * B_j over 32 reduction slots (lo*hi + hi*lo + hi*hi), the NB accumulators taking turns so that
// consecutive instructions are independent
template <int NB>
__device__ __forceinline__ void h2mm16(f32x4* acc, const Frag2& A, const Frag2* B) {
  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = mfma16_h(A.p[PA[t]], B[j].p[PB[t]], acc[j]);
}
__device__ __forceinline__ Frag2 split8_h2v(const f32x4& lo, const f32x4& hi) {
  return split8_h2(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
}

constexpr int kFlushRounds = 64;   // rounds (two tiles each per weight-gradient wave) between two moves to the second level (32: +0.13 ms at 65536 rows, dead groups 2.4e-6; 64: half of that, 5e-6)
#define OIVAE_ACC2_ADD(p, v) (void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// XH (dec_pad64.hip): the groups own a.xg < 64 columns of X each (a multiple of 4), their second layers are zero-padded
// to 64 outputs.  The kernel then reads group g's columns at g * a.xg; the lanes of the padded outputs read the
// group's first columns again (real values, inside the rows of X; what they produce is dropped), and the padded
// outputs stay out of the log-likelihood sum.
template <bool XH>
__global__ __launch_bounds__(512, 1) void dec_x3_64s(MfmaDecArgs a) {
  constexpr int K = 64, H = 64, M = 64, DG = 64, LW = 65;
  const int XG = XH ? a.xg : 64;   // columns of X per group
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
  const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;   // 32 x 32 fp32 products of set-up and end
  const int lc = lane & 15, lq = lane >> 4;                      // 16 x 16 x 32 products of the row loop
  const int pairIdx = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
  const bool chainw = wave < 4;
  const int sw = wave & 3;
  const int ngrp = (2 * pairIdx + 1 < a.G) ? 2 : 1;   // G odd: the last workgroup has one group
  float inv1[2] = {1.f, 1.f}, inv2[2] = {1.f, 1.f}, inv4[2] = {1.f, 1.f}, omtC[2] = {1.f, 1.f};
  float invW[2] = {1.f, 1.f}, invP[2] = {1.f, 1.f}, invG[2] = {1.f, 1.f}, invD[2] = {1.f, 1.f};
  if (tid < 64) flags[tid] = 0;
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
  for (int q = 0; q < 2; ++q) {   // set-up of the pair's groups (dec_x3_64g's, up to the layout of the images)
    const int gq = 2 * pairIdx + q;
    if (gq >= a.G) break;
    float* sb1 = vec + 192 * q;
    float* sb2 = sb1 + 64;
    float* sC3 = sb2 + 64;
    float* tW1 = scr;
    float* tWs = scr + 64 * LW;
    float* tW2 = scr + 2 * 64 * LW;
    __syncthreads();
    OIVAE_PROBE_SETUP(0);
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
    OIVAE_PROBE_SETUP(1);
    f32x16 acc = (f32x16)(0.f);   // A[m][k] = sum_h W1[m][h] Ws[k][h] on the fp32 matrix instruction
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
    OIVAE_PROBE_SETUP(2);
    const float* A = tW1;
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
      // one LDS atomic per wave and maximum: 512 lanes on one LDS word took 6 of the set-up's 14 us per group
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
    OIVAE_PROBE_SETUP(3);
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
    OIVAE_PROBE_SETUP(4);
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
    auto uni = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    inv1[q] = uni(kC * (i1 * iz)); inv2[q] = uni(i2 * (1.0f / kSt)); inv4[q] = uni(i1 * id);
    invW[q] = uni(ig * (1.0f / kSt)); invP[q] = uni(id * iz);
    invG[q] = uni(ig); invD[q] = uni(id); omtC[q] = uni((i2 * ig) * qd);
    __syncthreads();   // every thread has read the maxima before the biases change
    OIVAE_PROBE_SETUP(5);
    if (tid < 64) {
      sb1[tid] *= q1 * qz;    // accumulator initial values in the products' scaled units
      sb2[tid] *= q2 * kSt;
      sC3[tid] *= -qg;        // gmu is produced in its operand's scaled units; the chain multiplies (x - mu) by - c3
    }
    {  // the three images in two fp16 terms: item = (row r, step kk, lane quarter iq) -> 8 slots -> two 16-byte terms
      const int r = tid >> 3, kk = (tid >> 2) & 1, iq = tid & 3;
      float v[3][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = s16_slot_feature(kk, iq, e);
        v[0][e] = q1 * A[r * LW + 32 * kk + 8 * iq + e];   // plain order (transposed reads, z's fragments)
        v[1][e] = q2 * tW2[r * LW + c];
        v[2][e] = q2 * tW2[c * LW + r];
      }
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const Frag2 f = split8_h2(v[w][0], v[w][1], v[w][2], v[w][3], v[w][4], v[w][5], v[w][6], v[w][7]);
#pragma unroll
        for (int p = 0; p < 2; ++p)
          *reinterpret_cast<x3_u32x4*>(img + (q * 3 + w) * 4096 + ((((r >> 4) * 2 + kk) * 2 + p) * 64 + iq * 16 + (r & 15)) * 4) = f.p[p];
      }
    }
    OIVAE_PROBE_SETUP(6);
    OIVAE_PROBE(1 + q);
  }
  __syncthreads();
  OIVAE_PROBE(3);
  // fragment (16-row block fb, step kk, term p) of image `which`: 1 KB, lane-major
  auto wfrag = [&](const uint32_t* imgl, int which, int fb, int kk) {
    Frag2 f;
#pragma unroll
    for (int p = 0; p < 2; ++p) f.p[p] = *reinterpret_cast<const x3_u32x4*>(imgl + which * 4096 + ((fb * 2 + kk) * 2 + p) * 256);
    return f;
  };
  // The dz product needs A^T: lane = latent k = 16 kb + c, slots = hidden m in s16_slot_feature order.  Transposed
  // reads of the forward image (element A[m][k]: fragment (m / 16, k / 32), lane ((k / 8) % 4, m % 16), half k % 8):
  // per 16 lanes (quarter q) a block of 4 rows m = 32 kk + 16 j + 4 q + {0..3} x 16 columns k = 16 kb + {0..15}; lane
  // 4 qq + pp of the quarter addresses (row qq, columns 4 pp .. + 3), lane i receives column i.
  const int tr_lane = (((lane & 3) >> 1) * 16 + 4 * lq + ((lane >> 2) & 3)) * 16 + (lane & 1) * 8;
  auto wfrag_tr = [&](const char* trb, int kb, int kk) {
    Frag2 f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const x3_u32x2 v = lds_read_tr16(trb + ((((2 * kk + j) * 2 + (kb >> 1)) * 2 + p) * 1024 + (kb & 1) * 512));
        f.p[p][2 * j] = v[0];
        f.p[p][2 * j + 1] = v[1];
      }
    return f;
  };
  // One chained product as eight pinned units u = 4 kk + fb of six instructions: the weight fragments of unit u + 2 are
  // requested and the operands of the second reduction step are split inside units 0 and 1.
  auto chain_mm = [&](f32x4 (*acc)[2], auto&& wf, Frag2* bf, auto&& make, auto&& side) {
    Frag2 w[3];
    w[0] = wf(0, 0);
    w[1] = wf(1, 0);
    bf[0] = make(0);
    bf[1] = make(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u + 2 < 8) w[(u + 2) % 3] = wf((u + 2) & 3, (u + 2) >> 2);
      if (u < 2) bf[2 + u] = make(2 + u);
      if (u >= 5) { side(u - 5, 0); side(u - 5, 1); }
      h2mm16<2>(acc[u & 3], w[u % 3], bf + 2 * (u >> 2));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto init_t = [&](f32x4 (*acc)[2], const float* v0) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(v0 + 16 * fb + 4 * lq);
      acc[fb][0] = v;
      acc[fb][1] = v;
    }
  };
  auto zfrag2_of = [&](int t, int f) {
    Frag2 r;
    const uint32_t* zi = a.zimg + ((int64_t)t * 24 * 64 + lane) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) r.p[p] = *reinterpret_cast<const x3_u32x4*>(zi + (f * 2 + p) * 256);
    return r;
  };
  constexpr int LT = 68;   // a chain wave's fp32 tile [32 rows][68]
  constexpr bool kInsideScalar = true;   // side work between a product's matrix instructions: scalar fp32 (packed forms cost time there)

  const int N32 = (int)a.N;
  const int ntiles = (N32 + 31) / 32;
  const bool one_sample = (a.N == a.B);
  f32x4 dW2a[4][4], Pa[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { dW2a[i][j] = (f32x4)(0.f); Pa[i][j] = (f32x4)(0.f); }
  float sg1[4] = {0.f, 0.f, 0.f, 0.f}, sg2[4] = {0.f, 0.f, 0.f, 0.f}, sd1[4] = {0.f, 0.f, 0.f, 0.f};
  float xm = 0.f;
  const int wgrp = sw & 1, wp = sw >> 1;   // weight-gradient wave (wp, wgrp): group wgrp of chain waves 2 wp, 2 wp + 1
  const bool active = chainw || wgrp < ngrp;
  // Rounds of four row tiles (one per chain wave).  Every workgroup walks them from its own starting round, so that
  // the workgroups running at one time add to different rows of dz: at a small batch (an 8-GPU shard) they otherwise
  // stay in step and their float atomics meet on the same lines.
  const int nrounds = (ntiles + 4 * nchunks - 1) / (4 * nchunks);
  // (tools/dec_ablate.py on one box, decoder ms at 65536 / 8192 rows: no rotation 11.08 / 1.670, window of 8 rounds
  // 11.11 / 1.669, 32 rounds 11.08 / 1.542, 128 rounds 11.24 / 1.553, all rounds 11.58 / 1.538: a wide window loses the
  // L2 hits on z's image, which every workgroup reads)
  constexpr int kRotW = 32;
  const int rot = pairIdx % (nrounds < kRotW ? nrounds : kRotW);
  auto tile_of = [&](int rnd, int c) {
    int pr = rnd + rot;
    if (pr >= nrounds) pr -= nrounds;
    return (pr * nchunks + chunk) * 4 + c;
  };

  if (chainw) {
    // =============================== chain wave ===============================
    __builtin_amdgcn_s_setprio(3);
    const int sx = (lc >> 3) & 1;   // rows 8..15 of a 16-row block store their 16-column blocks swapped
    float* trw0 = scr + sw * kPairTile + lc * LT + 4 * lq + 16 * sx;        // even feature blocks
    float* trw1 = scr + sw * kPairTile + lc * LT + 4 * lq + 16 - 16 * sx;   // odd feature blocks
    auto tr_write = [&](const f32x4 (*v)[2]) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
          *reinterpret_cast<f32x4*>(((fb & 1) ? trw1 : trw0) + 16 * nb * LT + 32 * (fb >> 1)) = v[fb][nb];
    };
    volatile uint32_t* f_pub = flags + 2 * sw;
    volatile uint32_t* f_con = flags + 2 * sw + 1;
    uint32_t pub = 0;
    typedef const volatile __attribute__((address_space(3))) uint32_t* lds_cu32;
    typedef volatile __attribute__((address_space(3))) uint32_t* lds_u32;
    auto con_peek = [&]() { return *(lds_cu32)f_con; };
    auto publish = [&](const f32x4 (*v)[2], uint32_t peeked) {
      if ((uint32_t)__builtin_amdgcn_readfirstlane(peeked) < pub) lds_wait_ge(f_con, pub, a.err);
      tr_write(v);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      *(lds_u32)f_pub = ++pub;
      __builtin_amdgcn_sched_barrier(0);
    };
    char* const dzb = reinterpret_cast<char*>(a.dzT);
    const int64_t rowbytes = a.N * 4;
    int xoffs[4] = {0, 0, 0, 0};   // XH: this lane's four columns inside a group (the padded outputs' lanes: the first columns)
    if (XH) {
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) xoffs[fb] = (16 * fb + 4 * lq < XG) ? 16 * fb + 4 * lq : 0;
    }
    Frag2 zf[4];
    {
      int t1 = tile_of(0, sw);
      if (t1 >= ntiles) t1 = tile_of(1, sw);   // (at most one round of a wave is past the end)
      if (t1 < ntiles) {
#pragma unroll
        for (int f = 0; f < 4; ++f) zf[f] = zfrag2_of(t1, f);
      }
    }
    for (int rnd = 0; rnd < nrounds; ++rnd) {
      const int tile = tile_of(rnd, sw);
      if (tile >= ntiles) continue;
      int tnext = rnd + 1 < nrounds ? tile_of(rnd + 1, sw) : tile;
      if (tnext >= ntiles) tnext = rnd + 2 < nrounds ? tile_of(rnd + 2, sw) : tile;
      const int r0 = tile * 32;
      const bool partial = r0 + 32 > N32;
      const bool live0 = r0 + lc < N32, live1 = r0 + 16 + lc < N32;
      const int nT0 = live0 ? r0 + lc : N32 - 1, nT1 = live1 ? r0 + 16 + lc : N32 - 1;
      const float liveT[2] = {live0 ? 1.0f : 0.0f, live1 ? 1.0f : 0.0f};
      const float* xrow[2];
      xrow[0] = a.X + (one_sample ? (int64_t)nT0 : (int64_t)nT0 % a.B) * (int64_t)a.D + (XH ? 0 : 4 * lq);
      xrow[1] = a.X + (one_sample ? (int64_t)nT1 : (int64_t)nT1 % a.B) * (int64_t)a.D + (XH ? 0 : 4 * lq);
      f32x4 dz[4][2];
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) { dz[fb][0] = (f32x4)(0.f); dz[fb][1] = (f32x4)(0.f); }
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
        f32x4 xq[4][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int fb = 0; fb < 4; ++fb)
            xq[fb][nb] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xrow[nb] + (int64_t)(2 * pairIdx + q) * XG + (XH ? xoffs[fb] : 16 * fb)));
        // ---- a1 + b1 in units of 1 / (s1 sz) ----
        f32x4 uT[4][2];
        init_t(uT, sb1);
        auto none = [&](int, int) {};
        chain_mm(uT, [&](int fb, int kk) { return wfrag(imgl, 0, fb, kk); }, zf, [&](int f) { return zf[f]; }, none);
        // ---- t = tanh: published first, split during the next product ----
        uint32_t peek = con_peek();
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 e = uT[fb][nb] * k_inv1;
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(e[r]);
            e = e + 1.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_rcpf(e[r]);
            uT[fb][nb] = e * (-2.0f * kSt) + kSt;   // 2^13 t
          }
        publish(uT, peek);
        // ---- mu = W2 t + b2; gmu = (mu - x) inv / sigma^2; afterwards uT holds inv3 sd (1 - t^2) ----
        Frag2 tf[4];
        f32x4 gT[4][2];
        init_t(gT, sb2);
        auto gmu_block = [&](int fb, int nb, bool inside) {
          f32x4 c3n = *reinterpret_cast<const f32x4*>(sC3 + 16 * fb + 4 * lq);   // - c3 in scaled units
          if (partial) {   // rows past the end of the batch contribute nothing (a real, uniform branch: last tile only)
            asm volatile("");
            c3n *= liveT[nb];
          }
          const f32x4 x4 = xq[fb][nb];
          asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(xm) : "v"(x4[0]), "v"(x4[1]));
          asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(xm) : "v"(x4[2]), "v"(x4[3]));
          if (inside) {   // inside the product: scalar instructions (packed ones between matrix instructions cost time)
#pragma unroll
            for (int i = 0; i < 4; ++i) gT[fb][nb][i] = fmaf(gT[fb][nb][i], k_ninv2, x4[i]) * c3n[i];
          } else {
            gT[fb][nb] = (gT[fb][nb] * k_ninv2 + x4) * c3n;   // (mu - x) c3 = (x - mu) (- c3)
          }
        };
        chain_mm(gT, [&](int fb, int kk) { return wfrag(imgl, 1, fb, kk); }, tf,
                 [&](int f) {
                   const int kk = f >> 1, nb = f & 1;
                   const Frag2 fr = split8_h2v(uT[2 * kk][nb], uT[2 * kk + 1][nb]);
#pragma unroll
                   for (int j = 0; j < 2; ++j)
#pragma unroll
                     for (int r = 0; r < 4; ++r)
                       uT[2 * kk + j][nb][r] = fmaf(k_nomt2 * uT[2 * kk + j][nb][r], uT[2 * kk + j][nb][r], k_omtC);
                   return fr;
                 },
                 [&](int fb, int nb) { gmu_block(fb, nb, kInsideScalar); });
        peek = con_peek();
        gmu_block(3, 0, false);
        gmu_block(3, 1, false);
        publish(gT, peek);
        // ---- dt = W2^T gmu; da1 = dt (1 - t^2) ----
        Frag2 gf[4];
        f32x4 dT[4][2];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) { dT[fb][0] = (f32x4)(0.f); dT[fb][1] = (f32x4)(0.f); }
        chain_mm(dT, [&](int fb, int kk) { return wfrag(imgl, 2, fb, kk); }, gf,
                 [&](int f) { return split8_h2v(gT[2 * (f >> 1)][f & 1], gT[2 * (f >> 1) + 1][f & 1]); },
                 [&](int fb, int nb) {
#pragma unroll
                   for (int r = 0; r < 4; ++r) dT[fb][nb][r] *= uT[fb][nb][r];
                 });
        peek = con_peek();
        dT[3][0] *= uT[3][0];
        dT[3][1] *= uT[3][1];
        publish(dT, peek);
        {   // the next first product's operand, one product ahead: the same row tile's for the pair's second group
          const int tz = (h + 1 < ngrp) ? tile : tnext;
#pragma unroll
          for (int f = 0; f < 4; ++f) zf[f] = zfrag2_of(tz, f);
        }
        // ---- dz^T[k][n] += sum_m A[m][k] da1[n][m], on top of the pair's first group (brought to this group's units) ----
        if (h == 1) {
          const float ratio = (q ? inv4[0] : inv4[1]) / (q ? inv4[1] : inv4[0]);   // powers of two: exact
#pragma unroll
          for (int fb = 0; fb < 4; ++fb) { dz[fb][0] *= ratio; dz[fb][1] *= ratio; }
        }
        Frag2 df[4];
        const bool last = h + 1 == ngrp;
        const float k_inv4 = q ? inv4[1] : inv4[0];
        // dz^T[k][n] at dzT + k N + n: uniform row base (scalar registers) + 32-bit per-lane byte offset
        uint32_t voff0 = (uint32_t)((int64_t)(4 * lq) * a.N + nT0) * 4u, voff1 = (uint32_t)((int64_t)(4 * lq) * a.N + nT1) * 4u;
        asm volatile("" : "+v"(voff0), "+v"(voff1));
        auto dz_adds = [&](int fb, int nb, bool scalar_mul) {
          f32x4 o = dz[fb][nb];
          if (scalar_mul) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= k_inv4;
          } else {
            o *= k_inv4;
          }
          if (nb ? live1 : live0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              OIVAE_DZ_ADD(reinterpret_cast<float*>(dzb + (int64_t)(16 * fb + r) * rowbytes + (nb ? voff1 : voff0)), o[r]);
          }
        };
        chain_mm(dz, [&](int kb, int kk) { return wfrag_tr(trb, kb, kk); }, df,
                 [&](int f) { return split8_h2v(dT[2 * (f >> 1)][f & 1], dT[2 * (f >> 1) + 1][f & 1]); },
                 [&](int fb, int nb) { if (last) dz_adds(fb, nb, kInsideScalar); });
        if (last) { dz_adds(3, 0, false); dz_adds(3, 1, false); }
      }
    }
  } else if (active) {
    // =============================== weight-gradient wave ===============================
    // producer j = 0: chain wave 2 wp + wgrp (this group is the first half of its row tile), j = 1: the other one
    uint32_t seen[2] = {0, 0};
    const int npub = 3 * ngrp;
    // Second level of the two running sums.  The matrix instruction aligns its 32 products to the largest addend -- the
    // accumulator -- and drops what falls below its window: once a running sum has grown over some hundred tiles, the
    // lo x hi terms (2^-11 of the products) are partly lost, systematically when an operand hardly changes from row to
    // row (a dead group: t = tanh(b1) on every row; dW2 off by 4e-5 of its scale at 65536 rows against float64,
    // tools/probe/h2_accum_bias.hip, found by tests/test_gpu_fullsize.py).  So every kFlushRounds rounds (128 tiles) the
    // wave moves its 128 accumulators into a slab of its own in the workspace -- a store the first time, float adds
    // without return afterwards: no temporaries, nothing to wait for -- and starts them from zero; the slab comes back
    // once, behind the row loop.  (Tried first: each tile's products into zeroed temporaries, added by vector
    // instructions -- exact, but 77 spilled registers in this wave and 13.8 against 10.2 ms.)
    // (uniform slab base in scalar registers + one 32-bit lane offset, made opaque at each use: formed ahead of the row
    // loop, the 256 addresses would cost as many spilled registers)
    char* const my2 = a.acc2 ? reinterpret_cast<char*>(a.acc2 + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + sw) * 8192) : nullptr;
    bool flushed = false;
    auto flush = [&]() {
      uint32_t vo = (uint32_t)lane * 4u;
      asm volatile("" : "+v"(vo));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* pw = reinterpret_cast<float*>(my2 + vo + ((i * 4 + j) * 4 + r) * 256);
            float* pp = reinterpret_cast<float*>(my2 + 16384 + vo + ((i * 4 + j) * 4 + r) * 256);
            if (!flushed) { *pw = dW2a[i][j][r]; *pp = Pa[i][j][r]; }
            else { OIVAE_ACC2_ADD(pw, dW2a[i][j][r]); OIVAE_ACC2_ADD(pp, Pa[i][j][r]); }
          }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dW2a[i][j] = (f32x4)(0.f); Pa[i][j] = (f32x4)(0.f); }
      flushed = true;
    };
    const int sr = lq & 1;   // rows 8 q + e: their 16-column blocks are swapped for odd q
    for (int rnd = 0; rnd < nrounds; ++rnd) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 2 * wp + (j ^ wgrp);           // the chain wave
        const int tile = tile_of(rnd, c);             // its row tile of this round
        if (tile >= ntiles) continue;
        const float* trr0 = scr + c * kPairTile + 8 * lq * LT + lc + 16 * sr;
        const float* trr1 = scr + c * kPairTile + 8 * lq * LT + lc + 16 - 16 * sr;
        float rd[4][8];   // [feature block][row slot]
        auto tr_read = [&]() {
#pragma unroll
          for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int e = 0; e < 8; ++e) rd[fb][e] = ((fb & 1) ? trr1 : trr0)[e * LT + 32 * (fb >> 1)];
        };
        auto frag_of = [&](int fb) {
          return split8_h2(rd[fb][0], rd[fb][1], rd[fb][2], rd[fb][3], rd[fb][4], rd[fb][5], rd[fb][6], rd[fb][7]);
        };
        volatile uint32_t* f_pub = flags + 2 * c;
        volatile uint32_t* f_con = flags + 2 * c + 1;
        uint32_t n = seen[j] + ((ngrp == 2 && j == 1) ? 3 : 0);   // publishes of this chain wave before this group's three
        seen[j] += npub;
        Frag2 tNf[4];
        // ---- t ----
        lds_wait_ge(f_pub, n + 1, a.err);
        tr_read();
        lds_post(f_con, ++n);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) tNf[mb] = frag_of(mb);
        // ---- gmu: column sums, dW2[d][m] += sum_n gmu[n][d] t[n][m] ----
        lds_wait_ge(f_pub, n + 1, a.err);
        tr_read();
        lds_post(f_con, ++n);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          float c1 = 0.f, c2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { c1 += rd[db][e]; c2 = fmaf(rd[db][e], rd[db][e], c2); }
          sg1[db] += c1;
          sg2[db] += c2;
          const Frag2 g = frag_of(db);
          h2mm16<4>(dW2a[db], g, tNf);
        }
        // ---- da1: column sums, P[m][k] += sum_n da1[n][m] z[n][k] (z's fragments requested before the wait) ----
        Frag2 zn[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) zn[kb] = zfrag2_of(tile, 4 + kb);
        lds_wait_ge(f_pub, n + 1, a.err);
        tr_read();
        lds_post(f_con, ++n);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          float csum = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) csum += rd[mb][e];
          sd1[mb] += csum;
          const Frag2 d = frag_of(mb);
          h2mm16<4>(Pa[mb], d, zn);
        }
      }
      if (my2 != nullptr && (rnd + 1) % kFlushRounds == 0 && rnd + 1 < nrounds) flush();
    }
    if (flushed) {   // the slab back on top of the last rounds' sums (this wave's own adds: complete once vmcnt drains)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      uint32_t vo = (uint32_t)lane * 4u;
      asm volatile("" : "+v"(vo));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* pw = reinterpret_cast<const float*>(my2 + vo + ((i * 4 + j) * 4 + r) * 256);
            const float* pp = reinterpret_cast<const float*>(my2 + 16384 + vo + ((i * 4 + j) * 4 + r) * 256);
            dW2a[i][j][r] += __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            Pa[i][j][r] += __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
    }
    {   // the accumulated weight gradients back into true units
      const float kW = wgrp ? invW[1] : invW[0], kP = wgrp ? invP[1] : invP[0];
      const float kG = wgrp ? invG[1] : invG[0], kD = wgrp ? invD[1] : invD[0];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dW2a[i][j] *= kW; Pa[i][j] *= kP; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { sg1[i] *= kG; sg2[i] *= kG * kG; sd1[i] *= kD; }   // the tiles were in scaled units
    }
  }
  // ---- end of the kernel (dec_x3_64g's, up to the accumulators' layout) ----
  OIVAE_PROBE(4);
  __syncthreads();
  OIVAE_PROBE(5);
  if (chainw) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xm = fmaxf(xm, __shfl_xor(xm, o));
    if (lane == 0) atomicMax(const_cast<uint32_t*>(flags) + 26, __float_as_uint(xm));
  }
  // per-column sums: the four lane quarters of a column are added through LDS, lane L then holds column L
  float* tmp = scr + sw * 768;   // [3 sums][4 feature blocks][64 lanes]
  if (!chainw) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      tmp[(0 * 4 + b) * 64 + lane] = sg1[b];
      tmp[(1 * 4 + b) * 64 + lane] = sg2[b];
      tmp[(2 * 4 + b) * 64 + lane] = sd1[b];
    }
  }
  __syncthreads();
  if (tid == 0 && a.rng_seen != nullptr) atomic_max_if_larger(a.rng_seen, flags[26]);
  float cs_gmu = 0.f, cs_gmu2 = 0.f, cs_da1 = 0.f;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {   // fixed order
    cs_gmu += tmp[(0 * 4 + lq) * 64 + 16 * qq + lc];
    cs_gmu2 += tmp[(1 * 4 + lq) * 64 + 16 * qq + lc];
    cs_da1 += tmp[(2 * 4 + lq) * 64 + 16 * qq + lc];
  }
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
    auto stage_acc = [&](f32x4 (&acc)[4][4]) {
      __syncthreads();
      if (mine) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              stage[wp * 4096 + (16 * i + 4 * lq + r) * 64 + 16 * j + lc] = acc[i][j][r];
      }
      __syncthreads();
    };
    OIVAE_PROBE_END(0);
    stage_acc(dW2a);
    OIVAE_PROBE_END(1);
    if (worker) {
      float* dst = a.gW2 + cs + (int64_t)gq * DG * M;
      for (int idx = wtid; idx < 4096; idx += 256) dst[idx] = stage[idx] + stage[4096 + idx];
    }
    OIVAE_PROBE_END(2);
    stage_acc(Pa);
    OIVAE_PROBE_END(3);
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
    OIVAE_PROBE_END(4);
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
    OIVAE_PROBE_END(5);
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
      sred[wtid] = (XH && wtid >= XG) ? 0.f : nrows * (-0.5f * kLog2Pi - logf(sig)) - 0.5f * sumq2;
    }
    __syncthreads();
    if (worker && wtid < 64) {   // one wave: 64 terms summed in double, in a fixed tree order
      double tot = (double)sred[wtid];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
      if (wtid == 0) a.ll_part[(int64_t)chunk * a.G + gq] = tot;
    }
    OIVAE_PROBE_END(6);
  }
  OIVAE_PROBE(6);
}
