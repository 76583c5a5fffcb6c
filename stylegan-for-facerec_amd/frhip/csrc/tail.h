// frhip -- in-launch reduction of the partial rows a launch writes (FrTail, ABI v4).
//
// Every kernel that forms per-channel sums (BatchNorm statistics, BatchNorm-backward sums, the PReLU slope gradient) leaves
// one row of partial sums per workgroup, part[row][K][C], and until round 3 a second tiny launch (fr_bn_finalize /
// fr_reduce_parts) added the rows in double: 136 launches per IR-50 step that move a few hundred KB each, and -- in the
// backward pass, where they have to find a slot beside the resident weight-gradient workgroups of the side stream -- take
// 10-50 us in situ for 5 us of work.  With a tail the producing launch finishes the job itself:
//
//   every workgroup   stores its rows WRITE-THROUGH (st_part: global_store_dword sc1), every wave drains its stores
//                     (s_waitcnt vmcnt(0)), the workgroup's barrier, then ONE lane adds 1 to ticket[0] (agent scope)
//   the last S to arrive ("reducers"; S = tail.nred) wait until the count is complete (relaxed sc1 poll + s_sleep by one
//                     lane, bounded), ONE agent-scope acquire, barrier, and add the rows -- reducer s takes the 8-column
//                     groups s, s + S, ... -- with the very code of the stand-alone kernels (fr_reduce_rows8 below, one
//                     256-thread "virtual block" per group: same rows per thread, same shuffle / LDS tree => bit-identical
//                     sums), then write the outputs of fr_reduce_parts (FR_TAIL_SUMS) or fr_bn_finalize (FR_TAIL_BN)
//   the last reducer to finish zeroes the ticket again.
//
// This is the publish / consume recipe of cdna_hip_programming.md section 6, Guideline 16 in its counter form (sc1 payload,
// drained, agent-scope atomic add; consumer: poll or returned add, acquire, barrier, plain vector loads).  At most S - 1
// workgroups ever wait, and only for workgroups that are running or will be dispatched as others retire, so there is no
// residency requirement and no deadlock for any grid size (S <= 32 << the workgroup slots of the chip).
//
// Reference arithmetic: the batch statistics of nn.BatchNorm2d / BatchNorm1d in train mode and their autograd sums
// (backbone/model_irse.py:57,60,141,144,148), the PReLU slope gradient (:58,142).
#pragma once
#include "common.h"
#include "frhip.h"

constexpr int FR_RT = 256;  // threads of one reduction block (the stand-alone kernels' workgroup; a "virtual block" of a tail)

// write-through store of one partial sum: visible to every other workgroup of the launch once the storing wave has drained
__device__ __forceinline__ void st_part(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
// depending on the code around it, and this function must give the same bits inlined into a producer's tail as in the
// stand-alone kernel (round 4: the stand-alone kernel fused running_var's update, the tail did not -- last-bit differences).
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

// the per-channel arithmetic behind FR_TAIL_BNBWD / fr_bn_bwd_coeffs: the reduced sums and the BatchNorm backward as an affine
// map of (g, x):  gx = k*(g - s0/n - xhat*s1/n), xhat = (x - mean)*invstd, k = gamma*invstd  ==  ca*g + cb*x + cc
struct FrBnBwdCo {
  double count;
  const float* gamma;
  const float* mean;
  const float* invstd;
  int bn_eval;
  float* o0;
  float* o1;
  float* ca;
  float* cb;
  float* cc;
};
__device__ __forceinline__ void fr_bnbwd_channel(const FrBnBwdCo& f, int c, double d0, double d1) {
#pragma clang fp contract(off)
  const float s0 = (float)d0, s1 = (float)d1;
  if (f.o0) f.o0[c] = s0;
  if (f.o1) f.o1[c] = s1;
  const float inv_count = (float)(1.0 / f.count);
  const float is = f.invstd[c], mu = f.mean[c];
  const float k = (f.gamma ? f.gamma[c] : 1.f) * is;
  const float a = f.bn_eval ? 0.f : s0 * inv_count, bb = f.bn_eval ? 0.f : s1 * inv_count;
  const float ib = is * bb;
  const float kib = k * ib;
  const float ibm = ib * mu;
  const float d = ibm - a;
  f.ca[c] = k;
  f.cb[c] = -kib;
  f.cc[c] = k * d;
}

// bytes of LDS scratch fr_tail needs (it may alias anything that is dead once the partial rows are stored)
template <int NTH>
struct FrTailLds {
  static constexpr int BYTES = (NTH / FR_RT) * 2 * (FR_RT / 64) * 8 * 8;
};

// The tail of a producing launch.  Call from EVERY thread of EVERY workgroup, after the workgroup's partial rows have been
// stored with st_part (any thread may have stored them; this function drains and synchronises).  nparts = rows the whole
// launch writes, narrive = workgroups of the launch.  NTH = threads per workgroup (a multiple of 256); lds: FrTailLds<NTH>
// bytes, 8-byte aligned.  No effect when t.ticket == NULL.
template <int NTH>
__device__ __forceinline__ void fr_tail(const FrTail& t, const float* part, int nparts, unsigned narrive, void* lds_raw,
                                        int tid) {
  static_assert(NTH % FR_RT == 0, "tail: workgroups of whole 256-thread reduction blocks");
  if (t.ticket == nullptr) return;  // uniform
  unsigned* ldsu = reinterpret_cast<unsigned*>(lds_raw);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through rows have left
  __syncthreads();
  if (tid == 0) ldsu[0] = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned arrived = ldsu[0];
  const unsigned S = (unsigned)t.nred < narrive ? (unsigned)(t.nred > 0 ? t.nred : 1) : narrive;
  if (arrived + S < narrive) return;  // not one of the last S: done (uniform)
  const int slice = (int)(arrived - (narrive - S));
  if (tid == 0) {
    if (arrived + 1 != narrive) {  // the count is not complete yet: one lane polls, relaxed, write-through loads
      unsigned spins = 0;
      while (__hip_atomic_load(t.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < narrive) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 24)) {  // ~seconds: a producer died -- say so instead of hanging the queue
          __hip_atomic_store(t.ticket + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // ---- reduce: virtual block vb of reducer `slice` takes the 8-column groups (slice * NVB + vb) + i * S * NVB
  constexpr int NVB = NTH / FR_RT;
  const int vb = tid / FR_RT, vt = tid - vb * FR_RT;
  double* lds = reinterpret_cast<double*>(lds_raw) + vb * 2 * (FR_RT / 64) * 8;
  const bool bn = t.kind == FR_TAIL_BN, bwd = t.kind == FR_TAIL_BNBWD;
  const int C = t.C;
  const int KC = bn ? 2 * C : t.K * C;
  const int cg = (C + 7) / 8;
  // ---- pass 1 (BN, BNBWD): 8-channel groups with TWO column sets -- (sum, sum of squares) resp. (sum g', sum g' xhat) --
  // in one sweep; the thread that holds a channel's two totals writes everything derived from them
  if (bn || bwd) {
    for (int g0 = slice * NVB; g0 < cg; g0 += (int)S * NVB) {
      __syncthreads();  // the previous trip's LDS totals have been read
      const int g = g0 + vb;
      const bool valid = g < cg;
      const int gc = valid ? g : cg - 1;
      const int cols[2] = {gc * 8, C + gc * 8};
      double sq[2];
      fr_reduce_rows8<2>(part, nparts, KC, cols, sq, lds, vt);
      const int c = gc * 8 + vt;
      if (valid && vt < 8 && c < C) {
        if (bn) {
          FrBnFin f;
          f.count = t.count;
          f.gamma = t.gamma;
          f.beta = t.beta;
          f.eps = t.eps;
          f.momentum = t.momentum;
          f.running_mean = t.running_mean;
          f.running_var = t.running_var;
          f.nbt = reinterpret_cast<long long*>(t.nbt);
          f.mean = t.mean;
          f.invstd = t.invstd;
          f.scale = t.scale;
          f.shift = t.shift;
          fr_bn_finalize_channel(f, c, sq[0], sq[1]);
        } else {
          FrBnBwdCo f;
          f.count = t.count;
          f.gamma = t.gamma;
          f.mean = t.in_mean;
          f.invstd = t.in_invstd;
          f.bn_eval = t.bn_eval;
          f.o0 = t.o0;
          f.o1 = t.o1;
          f.ca = t.ca;
          f.cb = t.cb;
          f.cc = t.cc;
          fr_bnbwd_channel(f, c, sq[0], sq[1]);
        }
      }
    }
    __syncthreads();
  }
  // ---- pass 2 (SUMS; the third vector of BNBWD): single column sets.  Only the vectors somebody wants (C % 8 == 0: an
  // 8-column group never straddles two vectors).  Written with selects, not arrays: hipcc promotes small private arrays to LDS.
  if (!bn) {
    float* const out0 = bwd ? nullptr : t.o0;
    float* const out1 = (bwd || t.K < 2) ? nullptr : t.o1;
    float* const out2 = t.K > 2 ? t.o2 : nullptr;
    const bool whole = (C & 7) == 0;
    // active vector number ka -> vector index: km0 <= km1 <= km2 over the non-NULL outputs
    const int has0 = out0 != nullptr, has1 = out1 != nullptr, has2 = out2 != nullptr;
    const int nact = has0 + has1 + has2;
    const int km0 = has0 ? 0 : (has1 ? 1 : 2);
    const int km1 = has0 ? (has1 ? 1 : 2) : 2;
    const int G = whole ? nact * cg : (bwd ? 0 : (KC + 7) / 8);
    for (int g0 = slice * NVB; g0 < G; g0 += (int)S * NVB) {
      __syncthreads();  // the previous trip's LDS totals have been read
      const int g = g0 + vb;
      const bool valid = g < G;
      const int gc = valid ? g : G - 1;
      int col0 = gc * 8;
      if (whole) {
        const int ka = gc / cg;
        col0 = (ka == 0 ? km0 : (ka == 1 ? km1 : 2)) * C + (gc - ka * cg) * 8;
      }
      const int cols[1] = {col0};
      double tot[1];
      fr_reduce_rows8<1>(part, nparts, KC, cols, tot, lds, vt);
      const int idx = col0 + vt;
      if (valid && vt < 8 && idx < KC) {
        const int k = idx / C, c = idx - k * C;
        float* o = k == 0 ? out0 : (k == 1 ? out1 : out2);
        if (o) o[c] = (float)tot[0];
      }
    }
  }
  // ---- leave the ticket zero for the next launch that uses it
  if (tid == 0) {
    const unsigned done = __hip_atomic_fetch_add(t.ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1 == S) {
      __hip_atomic_store(t.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// host side: the validated tail a launcher hands to its kernel for a launch that writes rows of K x C columns from
// workgroups of nvb 256-thread blocks (nred chosen here when the caller left it 0).  Returns 0, or -1 with the error set.
int fr_tail_default_nred();
inline int fr_tail_prepare(const FrTail& in, int K, int C, int nvb, FrTail* out, bool has_part) {
  *out = in;
  if (in.ticket == nullptr || in.kind == FR_TAIL_NONE) {
    *out = FrTail{};
    return 0;
  }
  if (!has_part) FR_UNSUPPORTED("tail: the launch writes no partial rows (part == NULL or an epilogue without sums)");
  if (in.C != C) FR_UNSUPPORTED("tail: C must equal the channel count of the launch's partial rows");
  int groups;
  if (in.kind == FR_TAIL_BN) {
    if (K != 2) FR_UNSUPPORTED("tail: FR_TAIL_BN needs (sum, sum of squares) rows");
    if (!in.mean || !in.invstd || !in.scale || !in.shift || !(in.count > 0.0))
      FR_UNSUPPORTED("tail: FR_TAIL_BN needs count, mean, invstd, scale, shift");
    out->K = 2;
    groups = (C + 7) / 8;
  } else if (in.kind == FR_TAIL_SUMS) {
    if (in.K < 1 || in.K > K) FR_UNSUPPORTED("tail: FR_TAIL_SUMS K must be 1 .. the vectors per partial row");
    out->K = K;  // the row pitch; outputs beyond the caller's K are NULL
    if (in.K < 3) out->o2 = nullptr;
    if (in.K < 2) out->o1 = nullptr;
    groups = (K * C + 7) / 8;
  } else if (in.kind == FR_TAIL_BNBWD) {
    if (K != 3 || in.K != 3) FR_UNSUPPORTED("tail: FR_TAIL_BNBWD reduces the three-vector rows of fr_bn_bwd_reduce (K = 3)");
    if ((C & 7) != 0) FR_UNSUPPORTED("tail: FR_TAIL_BNBWD needs C % 8 == 0");
    if (!in.in_mean || !in.in_invstd || !in.ca || !in.cb || !in.cc || !(in.count > 0.0))
      FR_UNSUPPORTED("tail: FR_TAIL_BNBWD needs count, in_mean, in_invstd, ca, cb, cc");
    out->K = 3;
    groups = (C + 7) / 8;
  } else {
    FR_UNSUPPORTED("tail: unknown kind");
  }
  if (out->nred <= 0) {
    const int want = (groups + nvb - 1) / nvb;  // one 8-column group per virtual block
    const int cap = fr_tail_default_nred();
    out->nred = want < 1 ? 1 : (want > cap ? cap : want);
  }
  if (out->nred > 64) out->nred = 64;
  return 0;
}
