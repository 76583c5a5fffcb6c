// frhip -- the ONE summation order of weight-gradient slabs, shared by the stand-alone sum (reduce_slabs_kernel,
// conv_wgrad_strip.hip) and the sum folded into the next weight-gradient launch (conv_wgrad_roll.hip):
//
//   out[i] = ((c_0 + c_1) + c_2) + ...        c_j = ((s[16j][i] + s[16j+1][i]) + ...) over the <= 16 slabs of chunk j
//
// Fixed by the slab count alone, so every path gives the same bits, run after run.  A chunk is one batch of <= 16
// independent 16-byte loads per lane (one memory round trip); chunks of one element sit on LPE = 1 .. 16 adjacent lanes
// and are combined by lane 0 of the group in chunk order.
#pragma once
#include "common.h"

__device__ __forceinline__ int slab_lanes_per_element(int groups) {
  const int chunks = (groups + 15) >> 4;
  int lpe = 1;
  while (lpe < chunks) lpe <<= 1;
  return lpe;  // <= 16 for groups <= 256
}

// sum of slabs [g0, g1) (g1 - g0 <= 16) of element i, in slab order
__device__ __forceinline__ f32x4 slab_chunk_sum(const f32x4* __restrict__ slab, long long n4, long long i, int g0, int g1) {
  f32x4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int g = g0 + u < g1 ? g0 + u : g1 - 1;  // clamped: the batch stays 16 unconditional loads
    v[u] = slab[(long long)g * n4 + i];
  }
  f32x4 s = v[0];
#pragma unroll
  for (int u = 1; u < 16; ++u)
    if (g0 + u < g1) s += v[u];
  return s;
}

// Sums elements [e_begin, e_end) with NTH cooperating threads (all of them must call; tid = 0 .. NTH-1, NTH a multiple of
// 64).  EPT = elements per thread and pass: their chunk loads are all issued before the first addition, so a pass is ONE
// memory round trip of EPT x 16 loads per lane (EPT x 64 registers).
template <int NTH, int EPT = 1>
__device__ __forceinline__ void slab_sum_range(const float* __restrict__ slab_f, int groups, long long n4, long long e_begin,
                                               long long e_end, float* __restrict__ out_f, int tid) {
  const f32x4* __restrict__ slab = reinterpret_cast<const f32x4*>(slab_f);
  f32x4* __restrict__ out = reinterpret_cast<f32x4*>(out_f);
  const int lpe = slab_lanes_per_element(groups);
  const int chunks = (groups + 15) >> 4;
  const int epp = NTH / lpe;  // elements per pass and sub-pass
  const int sub = tid & (lpe - 1), el = tid / lpe;
  const int g0 = sub < chunks ? sub * 16 : 0;
  const int g1 = g0 + 16 < groups ? g0 + 16 : groups;
  for (long long e0 = e_begin; e0 < e_end; e0 += (long long)epp * EPT) {
    f32x4 v[EPT][16];
    long long ev[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const long long e = e0 + (long long)q * epp + el;
      ev[q] = e;
      const long long ec = e < e_end ? e : e_end - 1;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int g = g0 + u < g1 ? g0 + u : g1 - 1;  // clamped: the batch stays unconditional loads
        v[q][u] = slab[(long long)g * n4 + ec];
      }
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      f32x4 c = v[q][0];
#pragma unroll
      for (int u = 1; u < 16; ++u)
        if (g0 + u < g1) c += v[q][u];
      f32x4 s = c;
      // lane 0 of the element's lane group adds the chunks in order (every lane executes the shuffles)
      for (int j = 1; j < lpe; ++j) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __shfl(c[r], (tid & 63 & ~(lpe - 1)) + j, 64);
        if (j < chunks) s += o;
      }
      if (ev[q] < e_end && sub == 0) out[ev[q]] = s;
    }
  }
}
