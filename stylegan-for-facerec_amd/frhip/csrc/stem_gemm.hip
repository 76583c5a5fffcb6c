// frhip -- the input-layer GEMMs (bf16, gfx950): y0 = X0 * W0^T and dW0 = g_y0^T * X0.
//
// The stem convolution Conv2d(3|6, 64, 3x3) of Backbone.input_layer (backbone/model_irse.py:140) runs as a GEMM over
// the im2col rows X0 [B*112*112][K] (K = 32 or 64: 9*Cin zero-padded, written by fr_stem_im2col) -- 3.2 M rows at
// batch 256, but only 64 columns and one or two 32-deep MFMA steps.  A tiled GEMM kernel spends its time in tile
// prologues and epilogues there; these two kernels are shaped for it instead: no operand staging for the forward (a
// fragment is one coalesced 16-byte global load: 16 rows x 64 B are contiguous), weights live in registers, every
// wave streams 16-row tiles, and both are HBM-bound by construction (forward: 64 B in, 128 B out per row).
#include "common.h"
#include "frhip_internal.h"

namespace {

constexpr int SN = 64;  // output channels of the stem

// (Round 4 also carried these GEMMs on IMPLICIT rows -- built in registers / from LDS-staged image rows out of the fp32 batch,
// bit-identical, 205 MB less memory and +0.05-0.09 ms per step: profiles/r04_ab_stem_implicit.txt.  Removed in round 5 with
// ABI v5; git tag r05-before-prune.)

// ------------------------------------------------------------------------------------------ forward + BN statistics
// out[m][n] = sum_k X[m][k] W[n][k];  part[blk][0][n] = sum_m out, part[blk][1][n] = sum_m out^2 (of the rounded bf16)
// MODE 0: out = y, statistics of y.  MODE 1: statistics of y only, nothing stored (the first of two passes over the rows:
// the GEMM is 13 GFLOP at batch 256, its output 411 MB -- recomputing it is cheaper than writing and re-reading it).
// MODE 2: z = PReLU(BN(y)) with the coefficients the statistics of pass 1 gave (the arithmetic of fr_bn_apply on the rounded
// y, element for element), stored to zout together with y (out, optional: the backward pass reads it), statistics of the
// rounded z for the BatchNorm of the first residual unit: the fr_bn_apply pass over the stem output (822 MB) is gone.
struct StemAct {
  const float *scale, *shift, *slope;
  bf16_t* zout;
};
template <int K, int MODE = 0>
__global__ __launch_bounds__(256) void stem_gemm_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp,
                                                        bf16_t* __restrict__ out, float* __restrict__ part, int M,
                                                        const StemAct act) {
  constexpr int KS = K / 32;
  constexpr int OSTR = SN * 2 + 16;                    // per-wave transpose tile [16 rows][64 ch], padded rows
  __shared__ __attribute__((aligned(16))) char tiles[(MODE == 2 ? 8 : 4) * 16 * OSTR];
  constexpr int NV = 2;  // vectors per partial row
  __shared__ float red[4 * NV * SN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  char* tile = tiles + wave * 16 * OSTR;
  // weights as the MFMA A operand (rows = output channels): a lane then owns four consecutive channels of one row of X
  s16x8 wf[4][KS];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      wf[j][kk] = *reinterpret_cast<const s16x8*>(Wp + (size_t)(j * 16 + fr) * K + kk * 32 + fq * 8);
  float s0[4][4], s1[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) s0[j][r] = s1[j][r] = 0.f;
  constexpr bool ACT = MODE == 2;
  float asc[ACT ? 4 : 1][4], ash[ACT ? 4 : 1][4], asl[ACT ? 4 : 1][4];
  char* ztile = tiles + ((MODE == 2 ? 4 : 0) + wave) * 16 * OSTR;
  if (ACT) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = j * 16 + fq * 4 + r;
        asc[j][r] = act.scale[n];
        ash[j][r] = act.shift[n];
        asl[j][r] = act.slope[n];
      }
  }
  const int ntiles = (M + 15) / 16;
  // the waves of the whole grid stride over the rows
  // the statistics-only mode has no barrier in the loop: several tiles per trip, all their loads requested before the first MFMA
  constexpr int NT = MODE == 1 ? 4 : 1;
  const int tbase = blockIdx.x * 4 + wave;
  const int tmul = gridDim.x * 4;
  const int tlim = ntiles;
  const int trips = (ntiles + tmul - 1) / tmul;  // same trip count for every wave: the loop body has barriers
  for (int it = 0; it < trips; it += NT) {
    s16x8 afs[NT][KS];
    bool oks[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = tbase + (it + u) * tmul;
      const int row = t * 16 + fr;
      oks[u] = it + u < trips && t < tlim && row < M;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        afs[u][kk] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (oks[u]) afs[u][kk] = *reinterpret_cast<const s16x8*>(X + (size_t)row * K + kk * 32 + fq * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
    const int t = tbase + (it + u) * tmul;
    const bool ok = oks[u];
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], afs[u][kk], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint2 o;
      o.x = pack2bf(acc[j][0], acc[j][1]);
      o.y = pack2bf(acc[j][2], acc[j][3]);
      if (MODE != 1) *reinterpret_cast<uint2*>(tile + fr * OSTR + (j * 16 + fq * 4) * 2) = o;
      if (MODE == 2) {  // z = PReLU(BN(y)) of the ROUNDED y, as fr_bn_apply computes it from the stored tensor
        float uq[4] = {__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xFFFF0000u), __uint_as_float(o.y << 16),
                       __uint_as_float(o.y & 0xFFFF0000u)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          uq[r] = fmaf(uq[r], asc[j][r], ash[j][r]);
          uq[r] = uq[r] > 0.f ? uq[r] : uq[r] * asl[j][r];
        }
        o.x = pack2bf(uq[0], uq[1]);
        o.y = pack2bf(uq[2], uq[3]);
        *reinterpret_cast<uint2*>(ztile + fr * OSTR + (j * 16 + fq * 4) * 2) = o;
      }
      if (ok) {  // statistics of what the next layer will actually read
        const float q0 = __uint_as_float(o.x << 16), q1 = __uint_as_float(o.x & 0xFFFF0000u);
        const float q2 = __uint_as_float(o.y << 16), q3 = __uint_as_float(o.y & 0xFFFF0000u);
        s0[j][0] += q0;
        s0[j][1] += q1;
        s0[j][2] += q2;
        s0[j][3] += q3;
        s1[j][0] = fmaf(q0, q0, s1[j][0]);
        s1[j][1] = fmaf(q1, q1, s1[j][1]);
        s1[j][2] = fmaf(q2, q2, s1[j][2]);
        s1[j][3] = fmaf(q3, q3, s1[j][3]);
      }
    }
    if (MODE == 1) continue;  // statistics only: no tile, no barrier
    __syncthreads();
    // the wave's 16 x 128 B tile leaves as 16-byte stores: 8 lanes per row
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int c = lane + v * 64, r = c >> 3, c8 = c & 7;
      const int orow = t * 16 + r;
      if (t < tlim && orow < M) {
        if (MODE != 2 || out) st16(out + (size_t)orow * SN + c8 * 8, ld16(tile + r * OSTR + c8 * 16));
        if (MODE == 2) st16(act.zout + (size_t)orow * SN + c8 * 8, ld16(ztile + r * OSTR + c8 * 16));
      }
    }
    __syncthreads();
    }
  }
  // column sums: fold the 16 row lanes, then the 4 waves
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = s0[j][r], c = s1[j][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a += __shfl_xor(a, o, 64);
        c += __shfl_xor(c, o, 64);
      }
      if (fr == 0) {
        red[(wave * NV + 0) * SN + j * 16 + fq * 4 + r] = a;
        red[(wave * NV + 1) * SN + j * 16 + fq * 4 + r] = c;
      }
    }
  __syncthreads();
  if (tid < NV * SN) {
    const int k = tid / SN, n = tid - k * SN;
    st_part(part + ((size_t)blockIdx.x * NV + k) * SN + n, red[(0 * NV + k) * SN + n] + red[(1 * NV + k) * SN + n] +
                                                              red[(2 * NV + k) * SN + n] + red[(3 * NV + k) * SN + n]);
  }
}

// ------------------------------------------------------------------------------------------ BN / PReLU backward sums
// Backward of BatchNorm2d -> PReLU behind the stem GEMM WITHOUT the stored GEMM output: y is recomputed from the rows (rounded
// as the forward pass rounded it) and the sums of fr_bn_bwd_reduce(slope) over (g, y) leave as rows part[blk][3][64]:
//   [0] sum g', [1] sum g' * xhat, [2] sum g * u * [u <= 0];  u = y*scale + shift, g' = g * prelu'(u), xhat = (y - mean) * invstd
// -- the expressions of bn_bwd_reduce_lean_kernel<true>, element for element.  A wave streams 16-row tiles like the forward
// kernel; its y tile goes through a per-wave LDS tile so that the sums run in the [row][8-channel chunk] layout of the
// coalesced 16-byte gradient loads: a lane meets 8 channels (their coefficients: 40 registers), not the 16 of the MFMA
// layout (first version: 200 VGPRs, two waves per SIMD, 157 us; one channel tile per wave with 32-byte gradient pieces: 215 us).
template <int K>
__global__ __launch_bounds__(256) void stem_bwd_sums_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp,
                                                            const bf16_t* __restrict__ G, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ slope, float* __restrict__ part,
                                                            int M) {
  constexpr int KS = K / 32, NT = 4;
  constexpr int OSTR = SN * 2 + 16;
  __shared__ __attribute__((aligned(16))) char tiles[4 * 16 * OSTR];
  __shared__ float red[4 * 3 * SN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int c8 = lane & 7, r8 = lane >> 3;  // sums layout: rows r8 and r8 + 8 of a tile, channels 8*c8 .. +7
  char* tile = tiles + wave * 16 * OSTR;
  s16x8 wf[4][KS];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      wf[j][kk] = *reinterpret_cast<const s16x8*>(Wp + (size_t)(j * 16 + fr) * K + kk * 32 + fq * 8);
  float sc[8], sh[8], sl[8], mu[8], is[8], a0[8], a1[8], a2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    sc[q] = scale[c8 * 8 + q];
    sh[q] = shift[c8 * 8 + q];
    sl[q] = slope[c8 * 8 + q];
    mu[q] = mean[c8 * 8 + q];
    is[q] = invstd[c8 * 8 + q];
    a0[q] = a1[q] = a2[q] = 0.f;
  }
  const int ntiles = (M + 15) / 16;
  const int tstep = gridDim.x * 4;
  for (int t0 = blockIdx.x * 4 + wave; t0 < ntiles; t0 += tstep * NT) {
    s16x8 af[NT][KS];
    U128 gv[NT][2];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = t0 + u * tstep, row = t * 16 + fr;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        af[u][kk] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (t < ntiles && row < M) af[u][kk] = *reinterpret_cast<const s16x8*>(X + (size_t)row * K + kk * 32 + fq * 8);
      }
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int grow = t * 16 + r8 + 8 * v;
        gv[u][v] = zero16();
        if (t < ntiles && grow < M) gv[u][v] = ld16(G + (size_t)grow * SN + c8 * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = t0 + u * tstep;
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], af[u][kk], acc[j], 0, 0, 0);
      __builtin_amdgcn_wave_barrier();  // the previous tile's reads of this wave's LDS tile are issued
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint2 o;
        o.x = pack2bf(acc[j][0], acc[j][1]);
        o.y = pack2bf(acc[j][2], acc[j][3]);
        *reinterpret_cast<uint2*>(tile + fr * OSTR + (j * 16 + fq * 4) * 2) = o;
      }
      __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order: its own tile needs no s_barrier
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int rr = r8 + 8 * v;
        if (t < ntiles && t * 16 + rr < M) {
          float yv[8], gg[8];
          unpack16<bf16_t>(ld16(tile + rr * OSTR + c8 * 16), yv);
          unpack16<bf16_t>(gv[u][v], gg);
#pragma unroll
          for (int q = 0; q < 8; ++q) {  // operation order of bn_bwd_reduce_lean_kernel<true>
            float gq = gg[q];
            const float uu = fmaf(yv[q], sc[q], sh[q]);
            const bool pos = uu > 0.f;
            a2[q] += pos ? 0.f : gq * uu;
            gq = pos ? gq : gq * sl[q];
            a0[q] += gq;
            a1[q] = fmaf(gq, (yv[q] - mu[q]) * is[q], a1[q]);
          }
        }
      }
    }
  }
  // fold the 8 row lanes of a wave (lane bits 3..5), then the 4 waves
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float a = a0[q], c = a1[q], e = a2[q];
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      a += __shfl_xor(a, o, 64);
      c += __shfl_xor(c, o, 64);
      e += __shfl_xor(e, o, 64);
    }
    if (r8 == 0) {
      red[(wave * 3 + 0) * SN + c8 * 8 + q] = a;
      red[(wave * 3 + 1) * SN + c8 * 8 + q] = c;
      red[(wave * 3 + 2) * SN + c8 * 8 + q] = e;
    }
  }
  __syncthreads();
  if (tid < 3 * SN) {
    const int k = tid / SN, n = tid - k * SN;
    part[((size_t)blockIdx.x * 3 + k) * SN + n] =
        red[(0 * 3 + k) * SN + n] + red[(1 * 3 + k) * SN + n] + red[(2 * 3 + k) * SN + n] + red[(3 * 3 + k) * SN + n];
  }
}

// ------------------------------------------------------------------------------------------ ... fed by the first unit
// Round 6.  The gradient the sums above read is the LAST thing the first residual unit writes: gx = BN1-backward(g) [+ the
// shortcut gradient], fr_bn_bwd_apply over 3.2 M rows x 64 channels at batch 256 -- 411 MB written, and read straight back
// by the kernel above (both HBM-bound; the 256-MB memory-side cache holds none of it).  Here the kernel above forms gx
// itself: it loads the unit's (g, x [, add]) chunks where it used to load G, applies the expressions of
// bn_bwd_apply_lean_kernel<ADD> element for element, stores the rounded gx (the weight-gradient kernel reads it next) and
// feeds the ROUNDED values to its sums -- bit-identical gx and partial rows, one pass of 411 MB less.
struct StemFromUnit {
  const bf16_t* g;      // BN1-backward input of the unit [M][64]
  const bf16_t* x;      // the unit's input (= the stem output z) [M][64]
  const bf16_t* add;    // ADD 1: [M][64]; ADD 2: [B][H/2][W/2][64], lands on the pixels with even h and w
  bf16_t* gx;           // out [M][64]
  const float* mean;    // BN1 of the unit
  const float* invstd;
  const float* gamma;   // may be NULL (1)
  const float* s0;
  const float* s1;
  float inv_count;
  int W, HW;            // ADD 2: image width, pixels per image
  float invW, invHW;
};

// RX: the unit's input x IS the stem's output z = round(PReLU(BN0(y))) -- recomputed here from the y tile the sums need anyway
// (the expression of stem_gemm_kernel<K, 2>, element for element: same bits) instead of read: another 411 MB less.
template <int K, int ADD, bool RX>
__global__ __launch_bounds__(256) void stem_bwd_sums_from_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp,
                                                                 const StemFromUnit un, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ slope, float* __restrict__ part,
                                                                 int M) {
  constexpr int KS = K / 32, NT = 2;
  constexpr int OSTR = SN * 2 + 16;
  __shared__ __attribute__((aligned(16))) char tiles[4 * 16 * OSTR];
  __shared__ float red[4 * 3 * SN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int c8 = lane & 7, r8 = lane >> 3;  // rows r8 and r8 + 8 of a tile, channels 8*c8 .. +7
  char* tile = tiles + wave * 16 * OSTR;
  s16x8 wf[4][KS];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      wf[j][kk] = *reinterpret_cast<const s16x8*>(Wp + (size_t)(j * 16 + fr) * K + kk * 32 + fq * 8);
  float sc[8], sh[8], sl[8], mu[8], is[8], a0[8], a1[8], a2[8];
  float umu[8], uis[8], ucoef[8], ua[8], ub[8];  // the unit's BN1: names of bn_bwd_apply_lean_kernel
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    sc[q] = scale[c8 * 8 + q];
    sh[q] = shift[c8 * 8 + q];
    sl[q] = slope[c8 * 8 + q];
    mu[q] = mean[c8 * 8 + q];
    is[q] = invstd[c8 * 8 + q];
    a0[q] = a1[q] = a2[q] = 0.f;
    umu[q] = un.mean[c8 * 8 + q];
    uis[q] = un.invstd[c8 * 8 + q];
    ucoef[q] = (un.gamma ? un.gamma[c8 * 8 + q] : 1.f) * uis[q];
    ua[q] = un.s0[c8 * 8 + q] * un.inv_count;
    ub[q] = un.s1[c8 * 8 + q] * un.inv_count;
  }
  const unsigned Wd = (unsigned)un.W, HW = (unsigned)un.HW, Wh = Wd >> 1, HWq = HW >> 2;
  const int ntiles = (M + 15) / 16;
  const int tstep = gridDim.x * 4;
  for (int t0 = blockIdx.x * 4 + wave; t0 < ntiles; t0 += tstep * NT) {  // the tile order of stem_bwd_sums_kernel
    s16x8 af[NT][KS];
    U128 gv[NT][2], xv[RX ? 1 : NT][2], ev[NT][2];
    bool hit[NT][2];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = t0 + u * tstep, row = t * 16 + fr;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        af[u][kk] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (t < ntiles && row < M) af[u][kk] = *reinterpret_cast<const s16x8*>(X + (size_t)row * K + kk * 32 + fq * 8);
      }
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int grow = t * 16 + r8 + 8 * v;
        gv[u][v] = ev[u][v] = zero16();
        if (!RX) xv[u][v] = zero16();
        hit[u][v] = false;
        if (t < ntiles && grow < M) {
          gv[u][v] = ld16(un.g + (size_t)grow * SN + c8 * 8);
          if (!RX) xv[u][v] = ld16(un.x + (size_t)grow * SN + c8 * 8);
          if (ADD == 1) ev[u][v] = ld16(un.add + (size_t)grow * SN + c8 * 8);
          if (ADD == 2) {
            unsigned b, rem, h, w;
            fast_divmod((unsigned)grow, HW, un.invHW, b, rem);
            fast_divmod(rem, Wd, un.invW, h, w);
            hit[u][v] = ((h | w) & 1u) == 0u;
            if (hit[u][v]) ev[u][v] = ld16(un.add + ((size_t)b * HWq + (h >> 1) * Wh + (w >> 1)) * SN + c8 * 8);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = t0 + u * tstep;
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], af[u][kk], acc[j], 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint2 o;
        o.x = pack2bf(acc[j][0], acc[j][1]);
        o.y = pack2bf(acc[j][2], acc[j][3]);
        *reinterpret_cast<uint2*>(tile + fr * OSTR + (j * 16 + fq * 4) * 2) = o;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int rr = r8 + 8 * v;
        if (t < ntiles && t * 16 + rr < M) {
          float yv[8], gg[8], xx[8], ee[8];
          unpack16<bf16_t>(ld16(tile + rr * OSTR + c8 * 16), yv);
          unpack16<bf16_t>(gv[u][v], gg);
          if (RX) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {  // stem_gemm_kernel<K, 2>: z of the rounded y
              float uq = fmaf(yv[q], sc[q], sh[q]);
              xx[q] = uq > 0.f ? uq : uq * sl[q];
            }
            unpack16<bf16_t>(pack16<bf16_t>(xx), xx);
          } else {
            unpack16<bf16_t>(xv[u][v], xx);
          }
          unpack16<bf16_t>(ev[u][v], ee);
#pragma unroll
          for (int q = 0; q < 8; ++q) {  // bn_bwd_apply_lean_kernel<ADD>, element for element
            float o = ucoef[q] * (gg[q] - ua[q] - (xx[q] - umu[q]) * uis[q] * ub[q]);
            if (ADD == 1 || (ADD == 2 && hit[u][v])) o += ee[q];
            gg[q] = o;
          }
          const U128 packed = pack16<bf16_t>(gg);
          st16(un.gx + (size_t)(t * 16 + rr) * SN + c8 * 8, packed);
          unpack16<bf16_t>(packed, gg);  // the sums see what the stored tensor holds
#pragma unroll
          for (int q = 0; q < 8; ++q) {  // operation order of bn_bwd_reduce_lean_kernel<true>
            float gq = gg[q];
            const float uu = fmaf(yv[q], sc[q], sh[q]);
            const bool pos = uu > 0.f;
            a2[q] += pos ? 0.f : gq * uu;
            gq = pos ? gq : gq * sl[q];
            a0[q] += gq;
            a1[q] = fmaf(gq, (yv[q] - mu[q]) * is[q], a1[q]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float a = a0[q], c = a1[q], e = a2[q];
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      a += __shfl_xor(a, o, 64);
      c += __shfl_xor(c, o, 64);
      e += __shfl_xor(e, o, 64);
    }
    if (r8 == 0) {
      red[(wave * 3 + 0) * SN + c8 * 8 + q] = a;
      red[(wave * 3 + 1) * SN + c8 * 8 + q] = c;
      red[(wave * 3 + 2) * SN + c8 * 8 + q] = e;
    }
  }
  __syncthreads();
  if (tid < 3 * SN) {
    const int k = tid / SN, n = tid - k * SN;
    part[((size_t)blockIdx.x * 3 + k) * SN + n] =
        red[(0 * 3 + k) * SN + n] + red[(1 * 3 + k) * SN + n] + red[(2 * 3 + k) * SN + n] + red[(3 * 3 + k) * SN + n];
  }
}

// ------------------------------------------------------------------------------------------ weight gradient
// slab[blk][co][k] = sum over the workgroup's rows of g[m][co] * X[m][k]; fr_reduce_parts adds the slabs.
typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;

__device__ __forceinline__ s16x8 tr_frag2(const char* p0, const char* p1) {
  const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p0);
  const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p1);
  const s16x4 ai = __builtin_bit_cast(s16x4, a), bi = __builtin_bit_cast(s16x4, b);
  return (s16x8){ai[0], ai[1], ai[2], ai[3], bi[0], bi[1], bi[2], bi[3]};
}

// BN: G is the gradient at the OUTPUT of BatchNorm -> PReLU (model_irse.py:141-142) and the BN backward is applied
// while the rows are staged: g_y = gamma*invstd * (g*prelu'(u) - s0/n - xhat*s1/n), u = y*scale + shift -- the same
// fp32 expression as bn_bwd_apply_kernel, rounded to bf16 once, so the result equals the unfused path bit for bit,
// but the 411 MB gradient tensor at the stem output is never written or re-read.
// RECOMP (round 4): y is not read -- it is recomputed per 64-row trip from the staged X rows (64 x 64 x K on the MFMAs, the
// weights Wp in registers), rounded to bf16 as the forward pass rounded it, and the staged gradient rows are transformed in
// place in LDS with the same expression: bit-identical to the variant that reads y, 411 MB less traffic at batch 256.
struct StemBn {
  const bf16_t* y;  // BN input = stem GEMM output [M][64] (NULL with RECOMP)
  const float *mean, *invstd, *scale, *shift, *slope, *gamma, *s0, *s1;
  float inv_count;
  const bf16_t* wp;  // RECOMP: the packed stem weight [64][K]
};

template <int K, bool BN, bool RECOMP = false>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                         float* __restrict__ slab, int M, const StemBn bn) {
  constexpr int RB = 64;                     // rows staged per trip (two 32-deep MFMA steps)
  constexpr int GSTR = SN * 2 + 32;          // conflict-free row strides for the transposing reads
  constexpr int XSTR = K * 2 + 32;
  constexpr int KT = K / 16;                 // 16-wide k tiles of the result
  constexpr int GCH = RB * 8, XCH = RB * (K / 8), NCH = GCH + XCH;
  constexpr int NLD = (NCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) char Gs[RB * GSTR];
  __shared__ __attribute__((aligned(16))) char Xs[RB * XSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;  // wave = co tile
  const int li = lane & 15, lq = lane >> 4;
  const int colb = (4 * (li & 3)) * 2;
  f32x4 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // BN: per-channel coefficients of this thread's 8-channel chunk (256 % 8 == 0: the chunk never changes)
  float bsc[8], bsh[8], bsl[8], bco[8], ba[8], bmu[8], bis[8], bbb[8];
  // RECOMP: wave w recomputes the 16-channel tile w of y for all 64 rows of a trip, so a lane only ever meets channels
  // 16w + 4*lq .. +3: their coefficients (8 vectors x 4 channels) and the weight fragments of the tile stay in registers
  constexpr int KS = K / 32;
  s16x8 wf[RECOMP ? KS : 1];
  float cf[RECOMP ? 8 : 1][4];
  if (RECOMP) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wave * 16 + lq * 4 + r;
      const float is_ = bn.invstd[n];
      cf[0][r] = bn.scale[n];
      cf[1][r] = bn.shift[n];
      cf[2][r] = bn.slope[n];
      cf[3][r] = bn.mean[n];
      cf[4][r] = is_;
      cf[5][r] = bn.gamma[n] * is_;
      cf[6][r] = bn.s0[n] * bn.inv_count;
      cf[7][r] = bn.s1[n] * bn.inv_count;
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      wf[kk] = *reinterpret_cast<const s16x8*>(bn.wp + (size_t)(wave * 16 + li) * K + kk * 32 + lq * 8);
  }
  if (BN && !RECOMP) {
    const int c0 = (tid & 7) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bsc[j] = bn.scale[c0 + j];
      bsh[j] = bn.shift[c0 + j];
      bsl[j] = bn.slope[c0 + j];
      bmu[j] = bn.mean[c0 + j];
      bis[j] = bn.invstd[c0 + j];
      bco[j] = bn.gamma[c0 + j] * bis[j];
      ba[j] = bn.s0[c0 + j] * bn.inv_count;
      bbb[j] = bn.s1[c0 + j] * bn.inv_count;
    }
  }
  const int nchunks = (M + RB - 1) / RB;
  for (int cblk = blockIdx.x; cblk < nchunks; cblk += gridDim.x) {
    const int row0 = cblk * RB;
    U128 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = u * 256 + tid;
      v[u] = zero16();
      if (idx < GCH) {
        const int r = idx >> 3, c = idx & 7;
        if (row0 + r < M) {
          v[u] = ld16(G + (size_t)(row0 + r) * SN + c * 8);
          if (BN && !RECOMP) {
            float g[8], y[8];
            unpack16<bf16_t>(v[u], g);
            unpack16<bf16_t>(ld16(bn.y + (size_t)(row0 + r) * SN + c * 8), y);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float uu = fmaf(y[j], bsc[j], bsh[j]);
              const float gp = uu > 0.f ? g[j] : g[j] * bsl[j];
              g[j] = bco[j] * (gp - ba[j] - (y[j] - bmu[j]) * bis[j] * bbb[j]);
            }
            v[u] = pack16<bf16_t>(g);
          }
        }
      } else if (idx < NCH) {
        const int a = idx - GCH, r = a / (K / 8), c = a - r * (K / 8);
        if (row0 + r < M) v[u] = ld16(X + (size_t)(row0 + r) * K + c * 8);
      }
    }
    __syncthreads();  // the previous trip's fragments have been read
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = u * 256 + tid;
      if (idx < GCH) {
        st16(Gs + (idx >> 3) * GSTR + (idx & 7) * 16, v[u]);
      } else if (idx < NCH) {
        const int a = idx - GCH, r = a / (K / 8), c = a - r * (K / 8);
        st16(Xs + r * XSTR + c * 16, v[u]);
      }
    }
    __syncthreads();
    if (RECOMP) {
      // y of this trip, channel tile `wave`, four 16-row tiles: lane = (row li of the tile, k chunk lq) for the MFMA and ends up
      // with channels 16*wave + 4*lq .. +3 of row 16*t + li -- the cell of Gs it transforms in place
      const int n0 = wave * 16 + lq * 4;
#pragma unroll
      for (int t = 0; t < RB / 16; ++t) {
        const int rr = t * 16 + li;
        f32x4 ya = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          const s16x8 af = *reinterpret_cast<const s16x8*>(Xs + rr * XSTR + (kk * 32 + lq * 8) * 2);
          ya = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk], af, ya, 0, 0, 0);
        }
        if (row0 + rr < M) {
          uint2* cell = reinterpret_cast<uint2*>(Gs + rr * GSTR + n0 * 2);
          const uint2 gq = *cell;
          uint2 yq;
          yq.x = pack2bf(ya[0], ya[1]);
          yq.y = pack2bf(ya[2], ya[3]);
          const float y[4] = {__uint_as_float(yq.x << 16), __uint_as_float(yq.x & 0xFFFF0000u), __uint_as_float(yq.y << 16),
                              __uint_as_float(yq.y & 0xFFFF0000u)};
          float g[4] = {__uint_as_float(gq.x << 16), __uint_as_float(gq.x & 0xFFFF0000u), __uint_as_float(gq.y << 16),
                        __uint_as_float(gq.y & 0xFFFF0000u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {  // the expression of the BN variant above, operand for operand
            const float uu = fmaf(y[r], cf[0][r], cf[1][r]);
            const float gp = uu > 0.f ? g[r] : g[r] * cf[2][r];
            g[r] = cf[5][r] * (gp - cf[6][r] - (y[r] - cf[3][r]) * cf[4][r] * cf[7][r]);
          }
          uint2 o;
          o.x = pack2bf(g[0], g[1]);
          o.y = pack2bf(g[2], g[3]);
          *cell = o;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int ks = 0; ks < RB / 32; ++ks) {
      const int m0 = ks * 32 + 4 * lq + (li >> 2), m1 = m0 + 16;
      const s16x8 gf = tr_frag2(Gs + m0 * GSTR + (wave * 16) * 2 + colb, Gs + m1 * GSTR + (wave * 16) * 2 + colb);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const s16x8 xf = tr_frag2(Xs + m0 * XSTR + (k * 16) * 2 + colb, Xs + m1 * XSTR + (k * 16) * 2 + colb);
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, xf, acc[k], 0, 0, 0);
      }
    }
  }
  float* __restrict__ o = slab + (size_t)blockIdx.x * SN * K;
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(size_t)(wave * 16 + lq * 4 + r) * K + k * 16 + li] = acc[k][r];
}

}  // namespace

extern "C" int fr_stem_gemm(const void* X, const void* Wp, void* out, float* part, long long M, int K, int nblocks,
                            void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 31) - 64 || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_gemm: K must be 32 or 64, 0 < M < 2^31, nblocks >= 1");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const StemAct noact = {};
  if (!part) FR_UNSUPPORTED("fr_stem_gemm: part is required (the statistics rows)");
#define STEM_FWD(KK, MD)                                                                                           \
  hipLaunchKernelGGL((stem_gemm_kernel<KK, MD>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)Wp, \
                     (bf16_t*)out, part, (int)M, noact)
  if (K == 32) {
    if (out) STEM_FWD(32, 0);
    else STEM_FWD(32, 1);
  } else {
    if (out) STEM_FWD(64, 0);
    else STEM_FWD(64, 1);
  }
#undef STEM_FWD
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_gemm_bn_prelu(const void* X, const void* Wp, const float* scale, const float* shift, const float* slope,
                                     void* y, void* z, float* part, long long M, int K, int nblocks, void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 31) - 64 || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_gemm_bn_prelu: K must be 32 or 64, 0 < M < 2^31, nblocks >= 1");
  if (!X || !Wp || !scale || !shift || !slope || !z) FR_UNSUPPORTED("fr_stem_gemm_bn_prelu: X, Wp, scale, shift, slope, z are required");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const StemAct act = {scale, shift, slope, (bf16_t*)z};
  if (K == 32)
    hipLaunchKernelGGL((stem_gemm_kernel<32, 2>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)Wp,
                       (bf16_t*)y, part, (int)M, act);
  else
    hipLaunchKernelGGL((stem_gemm_kernel<64, 2>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)Wp,
                       (bf16_t*)y, part, (int)M, act);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_bwd_sums(const void* X, const void* Wp, const void* G, const float* mean, const float* invstd,
                                const float* scale, const float* shift, const float* slope, float* part, long long M, int K,
                                int nblocks, void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 31) - 64 || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_bwd_sums: K must be 32 or 64, 0 < M < 2^31, nblocks >= 1");
  if (!X || !Wp || !G || !mean || !invstd || !scale || !shift || !slope || !part)
    FR_UNSUPPORTED("fr_stem_bwd_sums: X, Wp, G, the five coefficient vectors and part are required");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (K == 32)
    hipLaunchKernelGGL((stem_bwd_sums_kernel<32>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)Wp,
                       (const bf16_t*)G, mean, invstd, scale, shift, slope, part, (int)M);
  else
    hipLaunchKernelGGL((stem_bwd_sums_kernel<64>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)Wp,
                       (const bf16_t*)G, mean, invstd, scale, shift, slope, part, (int)M);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_bwd_sums_from(const FrBnBwdArgs* unit, const void* X, const void* Wp, const float* mean,
                                     const float* invstd, const float* scale, const float* shift, const float* slope,
                                     float* part, long long M, int K, int nblocks, void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 24) || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_bwd_sums_from: K must be 32 or 64, 0 < M < 2^24, nblocks >= 1");
  if (!unit || !X || !Wp || !mean || !invstd || !scale || !shift || !slope || !part)
    FR_UNSUPPORTED("fr_stem_bwd_sums_from: unit, X, Wp, the five coefficient vectors and part are required");
  const FrBnBwdArgs& u = *unit;
  if (u.C != 64 || u.rows != M || !u.g || !u.gx || !u.mean || !u.invstd || !u.s0 || !u.s1 || u.se || u.slope)
    FR_UNSUPPORTED("fr_stem_bwd_sums_from: unit = the arguments of a plain fr_bn_bwd_apply over [M][64] (no slope, no gate)");
  if (u.add_kind < 0 || u.add_kind > 2 || (u.add_kind && !u.add))
    FR_UNSUPPORTED("fr_stem_bwd_sums_from: add_kind 0, 1 or 2 (with add)");
  if (u.add_kind == 2 && (u.add_stride != 2 || u.H < 2 || u.W < 2 || (u.H & 1) || (u.W & 1) || u.rows_per_image != u.H * u.W ||
                          M % u.rows_per_image))
    FR_UNSUPPORTED("fr_stem_bwd_sums_from: add_kind 2 needs add_stride 2 and even H, W with rows_per_image = H * W");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  StemFromUnit un = {(const bf16_t*)u.g, (const bf16_t*)u.x, (const bf16_t*)u.add, (bf16_t*)u.gx, u.mean, u.invstd, u.gamma,
                     u.s0, u.s1, u.inv_count, u.add_kind == 2 ? u.W : 1, u.add_kind == 2 ? u.rows_per_image : 1,
                     u.add_kind == 2 ? 1.0f / (float)u.W : 1.f, u.add_kind == 2 ? 1.0f / (float)u.rows_per_image : 1.f};
#define FROM(KK, AA)                                                                                                   \
  do {                                                                                                                 \
    if (u.x)                                                                                                           \
      hipLaunchKernelGGL((stem_bwd_sums_from_kernel<KK, AA, false>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X, \
                         (const bf16_t*)Wp, un, mean, invstd, scale, shift, slope, part, (int)M);                      \
    else                                                                                                               \
      hipLaunchKernelGGL((stem_bwd_sums_from_kernel<KK, AA, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X,  \
                         (const bf16_t*)Wp, un, mean, invstd, scale, shift, slope, part, (int)M);                      \
  } while (0)
  if (K == 32) {
    if (u.add_kind == 0) FROM(32, 0);
    else if (u.add_kind == 1) FROM(32, 1);
    else FROM(32, 2);
  } else {
    if (u.add_kind == 0) FROM(64, 0);
    else if (u.add_kind == 1) FROM(64, 1);
    else FROM(64, 2);
  }
#undef FROM
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_wgrad_bn_r(const void* G, const void* X, const void* Wp, const float* mean, const float* invstd,
                                  const float* scale, const float* shift, const float* slope, const float* gamma,
                                  const float* s0, const float* s1, float inv_count, float* slab, long long M, int K,
                                  int nblocks, void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 31) - 64 || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_wgrad_bn_r: K must be 32 or 64, 0 < M < 2^31, nblocks >= 1");
  if (!Wp || !mean || !invstd || !scale || !shift || !slope || !gamma || !s0 || !s1)
    FR_UNSUPPORTED("fr_stem_wgrad_bn_r: the packed weight and every BatchNorm / PReLU coefficient vector are required");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const StemBn bn = {nullptr, mean, invstd, scale, shift, slope, gamma, s0, s1, inv_count, (const bf16_t*)Wp};
  if (K == 32)
    hipLaunchKernelGGL((stem_wgrad_kernel<32, true, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G,
                       (const bf16_t*)X, slab, (int)M, bn);
  else
    hipLaunchKernelGGL((stem_wgrad_kernel<64, true, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G,
                       (const bf16_t*)X, slab, (int)M, bn);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_wgrad(const void* G, const void* X, float* slab, long long M, int K, int nblocks, void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 31) - 64 || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_wgrad: K must be 32 or 64, 0 < M < 2^31, nblocks >= 1");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const StemBn none = {};
  if (K == 32)
    hipLaunchKernelGGL((stem_wgrad_kernel<32, false>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X,
                       slab, (int)M, none);
  else
    hipLaunchKernelGGL((stem_wgrad_kernel<64, false>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X,
                       slab, (int)M, none);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_wgrad_bn(const void* G, const void* Y, const void* X, const float* mean, const float* invstd,
                                const float* scale, const float* shift, const float* slope, const float* gamma,
                                const float* s0, const float* s1, float inv_count, float* slab, long long M, int K,
                                int nblocks, void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 31) - 64 || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_wgrad_bn: K must be 32 or 64, 0 < M < 2^31, nblocks >= 1");
  if (!Y || !mean || !invstd || !scale || !shift || !slope || !gamma || !s0 || !s1)
    FR_UNSUPPORTED("fr_stem_wgrad_bn: every BatchNorm / PReLU coefficient vector is required");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const StemBn bn = {(const bf16_t*)Y, mean, invstd, scale, shift, slope, gamma, s0, s1, inv_count, nullptr};
  if (K == 32)
    hipLaunchKernelGGL((stem_wgrad_kernel<32, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X,
                       slab, (int)M, bn);
  else
    hipLaunchKernelGGL((stem_wgrad_kernel<64, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X,
                       slab, (int)M, bn);
  FR_LAUNCH_CHECK();
}
