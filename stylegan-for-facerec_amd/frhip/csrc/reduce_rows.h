// frhip -- the one summation order of per-channel partial rows (fr_bn_finalize, fr_bn_finalize_res, fr_reduce_parts).
//
// Every kernel that forms per-channel sums (BatchNorm statistics, BatchNorm-backward sums, the PReLU slope gradient) leaves
// one row of partial sums per workgroup, part[row][K][C]; a second small launch adds the rows in double.  (Round 4 also
// carried an in-launch form of these reductions -- FrTail, last-arriver tickets in 13 producers: bit-identical, 136 launches
// fewer and 0.35-0.75 ms per step SLOWER, profiles/r04_ab_tail_*.txt; removed in round 5 with ABI v5, the
// code is in the git tag r05-before-prune.)
//
// Reference arithmetic: the batch statistics of nn.BatchNorm2d / BatchNorm1d in train mode and their autograd sums
// (backbone/model_irse.py:57,60,141,144,148), the PReLU slope gradient (:58,142).
#pragma once
#include "common.h"
#include "frhip.h"

constexpr int FR_RT = 256;  // threads of one reduction block (the workgroup of the reduction kernels)

// part is [nparts][KC] fp32.  A block of FR_RT threads owns 8 consecutive columns (per column set): thread (row-lane
// rl = vt/8, column cl = vt%8) strides over the rows (32-B coalesced segments), accumulates in double, and the
// row-lanes are combined with wave shuffles + one LDS step.  The totals are valid in the threads with vt < 8
// (column vt).  Fixed summation order: deterministic for a given nparts.  vt = thread index inside the (virtual) block;
// lds = [NCOLSETS][FR_RT / 64][8] doubles of that block.  Every thread of the WORKGROUP must call (one barrier inside).
template <int NCOLSETS>
__device__ __forceinline__ void fr_reduce_rows8(const float* part, int nparts, int KC, const int (&col0)[NCOLSETS],
                                                double (&out)[NCOLSETS], double* lds, int vt) {
  // NCOLSETS column groups are reduced in the same sweep so that all their loads are in flight together (pure latency: a
  // few hundred KB per block)
  const int cl = vt & 7, rl = vt >> 3;
  constexpr int RL = FR_RT / 8, RW = FR_RT / 64;
  double s[NCOLSETS];
  const float* colp[NCOLSETS];  // out-of-range columns read the last valid one (branch-free loads) and are zeroed below
#pragma unroll
  for (int k = 0; k < NCOLSETS; ++k) {
    s[k] = 0.0;
    const int c = col0[k] + cl;
    colp[k] = part + (c < KC ? c : KC - 1);
  }
  int r = rl;
  auto trip = [&](auto utag) {  // U x NCOLSETS independent loads per trip
    constexpr int U = decltype(utag)::value;
    for (; r + (U - 1) * RL < nparts; r += U * RL) {
      float v[NCOLSETS][U];
#pragma unroll
      for (int k = 0; k < NCOLSETS; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u) v[k][u] = colp[k][(size_t)(r + u * RL) * KC];
#pragma unroll
      for (int k = 0; k < NCOLSETS; ++k) {
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) t += (double)v[k][u];
        s[k] += t;
      }
    }
  };
  trip(std::integral_constant<int, 8>{});
  trip(std::integral_constant<int, 4>{});
  trip(std::integral_constant<int, 2>{});
  for (; r < nparts; r += RL)
#pragma unroll
    for (int k = 0; k < NCOLSETS; ++k) s[k] += (double)colp[k][(size_t)r * KC];
#pragma unroll
  for (int k = 0; k < NCOLSETS; ++k)
    if (col0[k] + cl >= KC) s[k] = 0.0;
  // lanes of a wave: 8 row-lanes x 8 columns -> fold the row-lane bits (lane bits 3..5)
  const int wave = vt >> 6;
#pragma unroll
  for (int k = 0; k < NCOLSETS; ++k) {
    double t = s[k];
    t += __shfl_xor(t, 8, 64);
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if ((vt & 63) < 8) lds[(k * RW + wave) * 8 + cl] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NCOLSETS; ++k) {
    double t = 0.0;
    if (vt < 8)
      for (int w = 0; w < RW; ++w) t += lds[(k * RW + w) * 8 + vt];
    out[k] = t;
  }
}

// the per-channel arithmetic of fr_bn_finalize (threads vt < 8 of a block that reduced channels c0 .. c0 + 7)
struct FrBnFin {
  double count;
  const float* gamma;
  const float* beta;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  long long* nbt;
  float* mean;
  float* invstd;
  float* scale;
  float* shift;
};
// Floating-point contraction is OFF in this function, and it uses plain operators only: hipcc fuses a*b+c into an fma or not
// depending on the code around it, and this function must give the same bits wherever it is inlined.
// ROCm's __fmul_rn / __fadd_rn do NOT help: they are inline wrappers around * and + compiled with contraction allowed, and
// fuse with each other after inlining whatever the caller's pragma says.
// (mean, biased variance) of a channel -> everything fr_bn_finalize writes for it
__device__ __forceinline__ void fr_bn_from_moments(const FrBnFin& f, int c, double m, double var) {
#pragma clang fp contract(off)
  const float is = (float)(1.0 / sqrt(var + (double)f.eps));
  const float g = f.gamma ? f.gamma[c] : 1.f, bt = f.beta ? f.beta[c] : 0.f;
  const float mf = (float)m;
  f.mean[c] = mf;
  f.invstd[c] = is;
  f.scale[c] = g * is;
  const float gm = g * mf;
  f.shift[c] = __builtin_fmaf(-gm, is, bt);
  if (f.running_mean) {
    const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
    const float keep = 1.f - f.momentum;
    const float om = keep * f.running_mean[c], nm = f.momentum * mf;
    f.running_mean[c] = om + nm;
    const float ov = keep * f.running_var[c];
    f.running_var[c] = __builtin_fmaf(f.momentum, (float)unbiased, ov);
  }
  if (f.nbt && c == 0) *f.nbt += 1;
}
__device__ __forceinline__ void fr_bn_finalize_channel(const FrBnFin& f, int c, double s, double q) {
#pragma clang fp contract(off)
  // (the fused operations here and in fr_bn_from_moments are the ones rounds 1-3 shipped -- then chosen by the compiler, now
  // written out -- so that the fp32 parity fixtures see the same bits: batch-4 ... 16 networks at random init amplify a
  // last-bit change of a shift)
  const double m = s / f.count;
  double var = __builtin_fma(-m, m, q / f.count);
  if (var < 0.0) var = 0.0;
  fr_bn_from_moments(f, c, m, var);
}

