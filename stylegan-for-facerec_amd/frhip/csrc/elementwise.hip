// frhip -- HBM-bound channel-wise passes of the IR / IR-SE residual units on gfx950.
//
// All tensors are "pixel rows x channels" (NHWC) in the compute dtype; every thread moves 16-byte chunks
// (8 bf16 / 4 f32 channels), a block's threads are laid out [row-thread][channel-chunk] so global accesses
// are fully coalesced along the channel axis and per-channel reductions stay in registers until one
// LDS combine per block.  Partial sums leave the kernel as one row per block (part[blk][k][C]); a tiny
// second kernel adds them in double precision -- deterministic, no global float atomics.
//
// Reference arithmetic replaced (paths under /root/reference):
//   BatchNorm2d train-mode statistics / apply / backward   backbone/model_irse.py:57,60,141,144
//   PReLU (stem) fwd / bwd                                 backbone/model_irse.py:142
//   residual add with MaxPool2d(1,stride) or conv shortcut backbone/model_irse.py:52-66
//   SEModule squeeze / excite                              backbone/model_irse.py:23-46
//   Dropout(0.5) + Flatten of the output layer             backbone/model_irse.py:145-146
#include <type_traits>

#include "common.h"
#include "frhip_internal.h"
#include "reduce_rows.h"

namespace {

constexpr int NT = 256;

// combine per-thread column partials acc[K][VEC] across the row-threads of a block; the first row-thread
// group writes part[blk][k][C].  cpr = chunks per row (C / VEC), must divide NT... or cpr >= NT handled by caller.
template <int K, int VEC>
__device__ __forceinline__ void block_col_reduce(float (&acc)[K][VEC], float* red, float* part_blk, int C,
                                                 int cpr, int tid) {
  // red: [NT][K*VEC]
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[tid * (K * VEC) + k * VEC + j] = acc[k][j];
  __syncthreads();
  const int rt_count = NT / cpr;
  for (int e = tid; e < cpr * K * VEC; e += NT) {
    const int cc = e / (K * VEC), kj = e - cc * (K * VEC);
    float s = 0.f;
    for (int r = 0; r < rt_count; ++r) s += red[(r * cpr + cc) * (K * VEC) + kj];
    const int k = kj / VEC, j = kj - k * VEC;
    st_part(part_blk + (size_t)k * C + cc * VEC + j, s);
  }
}

// ------------------------------------------------------------------------------------------ stem im2col
template <typename T>
__global__ void stem_im2col_kernel(const float* __restrict__ x, const float* __restrict__ avg, T* __restrict__ out,
                                   int B, int H, int W, int C, int Cavg, int ldk) {
  // one thread per (pixel, tap): writes Ct = C + Cavg consecutive k entries; tap 9 = zero tail
  const int Ct = C + Cavg;
  const long long total = (long long)B * H * W * 10;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 10);
    const long long pix = i / 10;
    T* o = out + pix * ldk;
    if (tap == 9) {
      for (int k = 9 * Ct; k < ldk; ++k) Elt<T>::st(o + k, 0.f);
      continue;
    }
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int b = (int)(pix / ((long long)W * H));
    const int kh = tap / 3, kw = tap - kh * 3;
    const int sh = h + kh - 1, sw = w + kw - 1;
    const bool ok = (unsigned)sh < (unsigned)H && (unsigned)sw < (unsigned)W;
    for (int c = 0; c < Ct; ++c) {
      float v = 0.f;
      if (ok) {
        if (c < C) v = x[(((long long)b * C + c) * H + sh) * W + sw];
        else v = avg[((long long)(c - C) * H + sh) * W + sw];
      }
      Elt<T>::st(o + tap * Ct + c, v);
    }
  }
}

// bf16 rows of the two shipped stems (3 image channels -> K = 32; + 3 average-image channels -> K = 64): one thread builds
// one whole row in registers (27 / 54 cached scalar reads, neighbours share them) and writes it with 16-byte stores; a wave
// writes 4 / 8 KB contiguous.  (The per-(pixel, tap) kernel above moves 2 bytes per store: 0.19 ms for 243 MB at B = 256.)
template <int CT, int LDK>
__global__ __launch_bounds__(256) void stem_im2col_rows_kernel(const float* __restrict__ x, const float* __restrict__ avg,
                                                               bf16_t* __restrict__ out, int B, int H, int W) {
  constexpr int C = 3;
  const int total = B * H * W;
  for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += gridDim.x * blockDim.x) {
    const int w = pix % W, h = (pix / W) % H, b = pix / (W * H);
    float v[LDK];
#pragma unroll
    for (int k = 0; k < LDK; ++k) v[k] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int sh = h + tap / 3 - 1, sw = w + tap % 3 - 1;
      const bool ok = (unsigned)sh < (unsigned)H && (unsigned)sw < (unsigned)W;
      const int shc = ok ? sh : h, swc = ok ? sw : w;  // always a valid address; the value is dropped
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float t = c < C ? x[(((size_t)b * C + c) * H + shc) * W + swc] : avg[((size_t)(c - C) * H + shc) * W + swc];
        v[tap * CT + c] = ok ? t : 0.f;
      }
    }
    bf16_t* o = out + (size_t)pix * LDK;
#pragma unroll
    for (int q = 0; q < LDK / 8; ++q) st16(o + q * 8, pack16<bf16_t>(v + q * 8));
  }
}

// ------------------------------------------------------------------------------------------ partial-row reduction
// fr_reduce_rows8 (reduce_rows.h): one summation order everywhere.
constexpr int RT = FR_RT;  // threads of the partial-sum reduction kernels: small workgroups, so that they still find
                           // a slot on CUs that hold a resident strip workgroup of the side stream (512: +0.1 ms / step)

// ------------------------------------------------------------------------------------------ BN finalize
__global__ __launch_bounds__(RT) void bn_finalize_kernel(const float* __restrict__ part, int nparts, int C, FrBnFin f) {
  __shared__ double lds[2 * (RT / 64) * 8];
  // part row layout [2][C]: columns c (sum) and C + c (sum of squares); a block finalises 8 channels
  const int c0 = blockIdx.x * 8;
  const int cols[2] = {c0, C + c0};
  double sq[2];
  fr_reduce_rows8<2>(part, nparts, 2 * C, cols, sq, lds, threadIdx.x);
  const int c = c0 + threadIdx.x;
  if (threadIdx.x < 8 && c < C) fr_bn_finalize_channel(f, c, sq[0], sq[1]);
}

// Statistics of a residual sum without a pass over it (round 4).  A unit's output is  o' = a*y + b + o  per channel (y =
// conv2's output, (a, b) = BN2's coefficients, o = the unit's input).  Its batch moments follow from the moments of y (the
// rows conv2 writes anyway), the moments of o (the statistics BN1 of this unit normalised with) and ONE cross moment
// sum(y*o) (third row vector, FR_EPI_STATS_X):  mean' = a*my + b + mo,  var' = a^2*vy + vo + 2a*(E[y*o] - my*mo).
// So BN2's coefficients and the next unit's BN1 coefficients come out of the same launch, and the BN-apply pass that
// materialised o' only to measure it is gone (its consumer forms o' itself: FR_PRO_RESBN).  Centred form in double: the
// variances add, no difference of large sums beyond the ones fr_bn_finalize already takes.
static FrBnFin fin_of(const FrBnFinArgs& t) {
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
  return f;
}
__global__ __launch_bounds__(RT) void bn_finalize_res_kernel(const float* __restrict__ part, int nparts, int C, FrBnFin f,
                                                             const float* __restrict__ in_mean,
                                                             const float* __restrict__ in_invstd, float in_eps, FrBnFin fn,
                                                             bool has_next) {
  __shared__ double lds[3 * (RT / 64) * 8];
  const int c0 = blockIdx.x * 8;
  const int cols[3] = {c0, C + c0, 2 * C + c0};
  double sq[3];
  fr_reduce_rows8<3>(part, nparts, 3 * C, cols, sq, lds, threadIdx.x);
  const int c = c0 + threadIdx.x;
  if (threadIdx.x < 8 && c < C) {
    const double my = sq[0] / f.count;
    double vy = __builtin_fma(-my, my, sq[1] / f.count);
    if (vy < 0.0) vy = 0.0;
    fr_bn_from_moments(f, c, my, vy);  // BN2 exactly as fr_bn_finalize leaves it
    if (!has_next) return;
    const double a = (double)f.scale[c], b = (double)f.shift[c];  // the rounded coefficients the prologue multiplies with
    const double mo = (double)in_mean[c], io = (double)in_invstd[c];
    double vo = 1.0 / (io * io) - (double)in_eps;
    if (vo < 0.0) vo = 0.0;
    const double cov = sq[2] / f.count - my * mo;
    double vn = a * a * vy + vo + 2.0 * a * cov;
    if (vn < 0.0) vn = 0.0;
    fr_bn_from_moments(fn, c, a * my + b + mo, vn);
  }
}

// eval-mode coefficients from running statistics
__global__ void bn_eval_coeffs_kernel(const float* rm, const float* rv, const float* gamma, const float* beta,
                                      float eps, int C, float* mean, float* invstd, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rv[c] + eps);
  mean[c] = rm[c];
  invstd[c] = is;
  scale[c] = gamma[c] * is;
  shift[c] = beta[c] - rm[c] * gamma[c] * is;
}

__global__ __launch_bounds__(RT) void reduce_parts_kernel(const float* __restrict__ part, int nparts, int K, int C,
                                                          float* o0, float* o1, float* o2) {
  __shared__ double lds[(RT / 64) * 8];
  const int col0 = blockIdx.x * 8;  // over the K*C columns of a partial row
  const int cols[1] = {col0};
  double tot[1];
  fr_reduce_rows8<1>(part, nparts, K * C, cols, tot, lds, threadIdx.x);
  const double s = tot[0];
  const int idx = col0 + threadIdx.x;
  if (threadIdx.x < 8 && idx < K * C) {
    const int k = idx / C, c = idx - k * C;
    float* o = k == 0 ? o0 : (k == 1 ? o1 : o2);
    if (o) o[c] = (float)s;
  }
}

// ------------------------------------------------------------------------------------------ channel stats
template <typename T>
__global__ __launch_bounds__(NT) void channel_stats_kernel(const T* __restrict__ x, long long rows, int C,
                                                           float* __restrict__ part) {
  constexpr int VEC = Elt<T>::VEC;
  __shared__ float red[NT * 2 * VEC];
  const int cpr = C / VEC, tid = threadIdx.x;
  const int cc = tid % cpr, rt = tid / cpr, rtc = NT / cpr;
  float acc[2][VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = 0.f;
  const int nrows = (int)rows, rstep = gridDim.x * rtc;
  for (int r = blockIdx.x * rtc + rt; r < nrows; r += rstep) {
    float f[VEC];
    unpack16<T>(ld16(x + (size_t)r * C + cc * VEC), f);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      acc[0][j] += f[j];
      acc[1][j] = fmaf(f[j], f[j], acc[1][j]);
    }
  }
  block_col_reduce<2, VEC>(acc, red, part + (size_t)blockIdx.x * 2 * C, C, cpr, tid);
}

// ------------------------------------------------------------------------------------------ BN apply (+SE, +PReLU, +residual)
template <typename T>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const FrApplyArgs p) {
  constexpr int VEC = Elt<T>::VEC;
  __shared__ float red[NT * 2 * VEC];
  const int C = p.C, cpr = C / VEC, tid = threadIdx.x;
  const int cc = tid % cpr, rt = tid / cpr, rtc = NT / cpr;
  const int c0 = cc * VEC;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  float sc[VEC], sh[VEC], sl[VEC], rs[VEC], rh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = p.scale[c0 + j];
    sh[j] = p.shift[c0 + j];
    sl[j] = p.slope ? p.slope[c0 + j] : 1.f;
    rs[j] = p.res_kind == 2 ? p.rscale[c0 + j] : 1.f;
    rh[j] = p.res_kind == 2 ? p.rshift[c0 + j] : 0.f;
  }
  float acc[2][VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = 0.f;
  const long long rows = (long long)p.B * p.H * p.W;
  const int HW = p.H * p.W;
  const bool need_b = p.se != nullptr || (p.res_kind == 1 && p.res_stride > 1);
  const int nrows = (int)rows, rstep = gridDim.x * rtc;  // rows < 2^31: 32-bit row arithmetic (64-bit divides are slow)
  for (int r = blockIdx.x * rtc + rt; r < nrows; r += rstep) {
    float f[VEC];
    unpack16<T>(ld16(x + (size_t)r * C + c0), f);
    const int b = need_b ? r / HW : 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) f[j] = fmaf(f[j], sc[j], sh[j]);
    if (p.se) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] *= p.se[(size_t)b * C + c0 + j];
    }
    if (p.slope) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = f[j] > 0.f ? f[j] : f[j] * sl[j];
    }
    if (p.res_kind != 0) {
      long long rr = r;
      if (p.res_kind == 1 && p.res_stride > 1) {
        const int rem = r - b * HW;
        const int h = rem / p.W, w = rem - h * p.W;
        rr = ((long long)b * (p.H * p.res_stride) + (long long)h * p.res_stride) * (p.W * p.res_stride) +
             (long long)w * p.res_stride;
      }
      float g[VEC];
      unpack16<T>(ld16(res + rr * C + c0), g);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] += fmaf(g[j], rs[j], rh[j]);
    }
    const U128 o = pack16<T>(f);
    st16(out + (size_t)r * C + c0, o);
    if (p.part) {
      float q[VEC];
      unpack16<T>(o, q);  // statistics of what the next layer will actually read
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        acc[0][j] += q[j];
        acc[1][j] = fmaf(q[j], q[j], acc[1][j]);
      }
    }
  }
  if (p.part) block_col_reduce<2, VEC>(acc, red, p.part + (size_t)blockIdx.x * 2 * C, C, cpr, tid);
}

// ------------------------------------------------------------------------------------------ BN backward
// g' = g                                  plain
//    = g * prelu'(u), u = x*scale+shift   when slope != NULL (stem: BN -> PReLU)
//    = g * se[b][c] + gse[b][c]           when se != NULL (IR-SE: BN -> SE excite)
template <typename T>
__device__ __forceinline__ void bn_bwd_gprime(const FrBnBwdArgs& p, const float* g, const float* xv, int b, int c0,
                                              const float* sc, const float* sh, const float* sl, float* gp,
                                              float* slope_term) {
  constexpr int VEC = Elt<T>::VEC;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float v = g[j];
    if (p.slope) {
      const float u = fmaf(xv[j], sc[j], sh[j]);
      const bool pos = u > 0.f;
      if (slope_term) slope_term[j] = pos ? 0.f : v * u;
      v = pos ? v : v * sl[j];
    }
    if (p.se) v = fmaf(v, p.se[(size_t)b * p.C + c0 + j], p.gse ? p.gse[(size_t)b * p.C + c0 + j] : 0.f);
    gp[j] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const FrBnBwdArgs p) {
  constexpr int VEC = Elt<T>::VEC;
  __shared__ float red[NT * 3 * VEC];
  const int C = p.C, cpr = C / VEC, tid = threadIdx.x;
  const int cc = tid % cpr, rt = tid / cpr, rtc = NT / cpr, c0 = cc * VEC;
  const T* __restrict__ g = reinterpret_cast<const T*>(p.g);
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  float mu[VEC], is[VEC], sc[VEC], sh[VEC], sl[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = p.mean[c0 + j];
    is[j] = p.invstd[c0 + j];
    sc[j] = p.scale ? p.scale[c0 + j] : 1.f;
    sh[j] = p.shift ? p.shift[c0 + j] : 0.f;
    sl[j] = p.slope ? p.slope[c0 + j] : 1.f;
  }
  float acc[3][VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.f;
  const int nrows = (int)p.rows, rstep = gridDim.x * rtc;
  for (int r = blockIdx.x * rtc + rt; r < nrows; r += rstep) {
    float gv[VEC], xv[VEC], gp[VEC], st[VEC];
    unpack16<T>(ld16(g + (size_t)r * C + c0), gv);
    unpack16<T>(ld16(x + (size_t)r * C + c0), xv);
    const int b = p.se ? r / p.rows_per_image : 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) st[j] = 0.f;
    bn_bwd_gprime<T>(p, gv, xv, b, c0, sc, sh, sl, gp, st);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      acc[0][j] += gp[j];
      acc[1][j] = fmaf(gp[j], (xv[j] - mu[j]) * is[j], acc[1][j]);
      acc[2][j] += st[j];
    }
  }
  block_col_reduce<3, VEC>(acc, red, p.part + (size_t)blockIdx.x * 3 * C, C, cpr, tid);
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const FrBnBwdArgs p) {
  constexpr int VEC = Elt<T>::VEC;
  const int C = p.C, cpr = C / VEC, tid = threadIdx.x;
  const int cc = tid % cpr, rt = tid / cpr, rtc = NT / cpr, c0 = cc * VEC;
  const T* __restrict__ g = reinterpret_cast<const T*>(p.g);
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ add = reinterpret_cast<const T*>(p.add);
  T* __restrict__ gx = reinterpret_cast<T*>(p.gx);
  float mu[VEC], is[VEC], sc[VEC], sh[VEC], sl[VEC], coef[VEC], a[VEC], bb[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = p.mean[c0 + j];
    is[j] = p.invstd[c0 + j];
    sc[j] = p.scale ? p.scale[c0 + j] : 1.f;
    sh[j] = p.shift ? p.shift[c0 + j] : 0.f;
    sl[j] = p.slope ? p.slope[c0 + j] : 1.f;
    coef[j] = (p.gamma ? p.gamma[c0 + j] : 1.f) * is[j];
    a[j] = p.s0[c0 + j] * p.inv_count;
    bb[j] = p.s1[c0 + j] * p.inv_count;
  }
  const int nrows = (int)p.rows, rstep = gridDim.x * rtc;
  const bool need_b = p.se != nullptr || p.add_kind == 2;
  for (int r = blockIdx.x * rtc + rt; r < nrows; r += rstep) {
    float gv[VEC], xv[VEC], gp[VEC];
    unpack16<T>(ld16(g + (size_t)r * C + c0), gv);
    unpack16<T>(ld16(x + (size_t)r * C + c0), xv);
    const int b = need_b ? r / p.rows_per_image : 0;
    bn_bwd_gprime<T>(p, gv, xv, b, c0, sc, sh, sl, gp, nullptr);
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = coef[j] * (gp[j] - a[j] - (xv[j] - mu[j]) * is[j] * bb[j]);
    if (p.add_kind == 1) {
      float e[VEC];
      unpack16<T>(ld16(add + (size_t)r * C + c0), e);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += e[j];
    } else if (p.add_kind == 2) {
      // identity shortcut MaxPool2d(1, s): the gradient lands on pixels with h % s == 0 and w % s == 0
      const int rem = r - b * p.rows_per_image;
      const int h = rem / p.W, w = rem - h * p.W;
      const int s = p.add_stride;
      if (h % s == 0 && w % s == 0) {
        const long long rr = ((long long)b * (p.H / s) + h / s) * (p.W / s) + w / s;
        float e[VEC];
        unpack16<T>(ld16(add + rr * C + c0), e);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] += e[j];
      }
    }
    st16(gx + (size_t)r * C + c0, pack16<T>(o));
  }
}

// ------------------------------------------------------------------------------------------ lean bf16 variants
// The plain cases of the three kernels above (no PReLU slope, no SE gate, no strided identity scatter) are what the
// bf16 training step launches ~110 times.  They are latency x concurrency bound at the 14x14 / 7x7 stages (25-MB
// tensors, ~6 grid-stride iterations per thread), so these variants trade vector width for rows in flight: a thread
// owns 4 channels (8-byte loads) of UNR rows per iteration.  The per-channel coefficients shrink to 4 registers per
// array and are shared by all rows in flight, the kernels fit 64 VGPRs (8 waves per SIMD, and a wave of them fits
// beside a resident strip workgroup of the side stream), and 4x the bytes are in flight per CU.  Same arithmetic,
// same operation order, same part[blk][k][C] layout as the general kernels.
constexpr int LV = 4;    // channels per thread
constexpr int LUNR = 4;  // rows in flight per thread (forward kernels: they run alone on the chip)
// The two backward kernels run BESIDE the weight-gradient kernels of the side stream (2 waves x ~200 registers per SIMD
// there): two rows in flight keep them at 36 - 60 registers, so that two of their waves fit a SIMD's free registers
// instead of one (step: -0.1 ms; alone they are ~5 % slower).  Three rows since the weight-gradient launches leave 32 CUs
// without a persistent workgroup (engine.py, wgrad_wgs): 15.33-15.35 against 15.36-15.42 ms per step (four rows: 15.35-15.39).
constexpr int LUNRB = 3;

__device__ __forceinline__ uint2 pack4bf(const float* f) {
  uint2 u;
  u.x = pack2bf(f[0], f[1]);
  u.y = pack2bf(f[2], f[3]);
  return u;
}
__device__ __forceinline__ void unpack4bf(const uint2& u, float* f) {
  f[0] = __uint_as_float(u.x << 16);
  f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16);
  f[3] = __uint_as_float(u.y & 0xFFFF0000u);
}

__device__ __forceinline__ uint2 ld8(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }

// RES: 0 none, 1 identity (same geometry), 2 conv shortcut with its BN folded (rscale/rshift)
// SE (round 3): the squeeze-excite gate se[b][c] (fp32, a few KB per image: L2 hits) between the BN and the residual add --
// bottleneck_IR_SE (model_irse.py:86-91); the IR-SE nets ran the general kernel here (19 us instead of 8 per 14x14 unit at
// 128 images: one row in flight per thread).
template <int RES, bool STATS, bool SE = false>
__global__ __launch_bounds__(NT, 6) void bn_apply_lean_kernel(const FrApplyArgs p) {
  __shared__ float red[STATS ? NT * 2 * LV : 1];
  const int C = p.C, cpr = C / LV, tid = threadIdx.x;
  const int cc = tid % cpr, rt = tid / cpr, rtc = NT / cpr, c0 = cc * LV;
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(p.x) + c0;
  const bf16_t* __restrict__ res = reinterpret_cast<const bf16_t*>(p.res) + c0;
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out) + c0;
  float sc[LV], sh[LV], rs[LV], rh[LV];
#pragma unroll
  for (int j = 0; j < LV; ++j) {
    sc[j] = p.scale[c0 + j];
    sh[j] = p.shift[c0 + j];
    rs[j] = RES == 2 ? p.rscale[c0 + j] : 1.f;
    rh[j] = RES == 2 ? p.rshift[c0 + j] : 0.f;
  }
  float acc[2][LV];
#pragma unroll
  for (int j = 0; j < LV; ++j) acc[0][j] = acc[1][j] = 0.f;
  const int nrows = p.B * p.H * p.W, rstep = gridDim.x * rtc * LUNR;
  const int HW = p.H * p.W;
  for (int r0 = blockIdx.x * rtc * LUNR + rt; r0 < nrows; r0 += rstep) {
    uint2 xr[LUNR], gr[LUNR];  // raw (packed) rows in flight; unpacked one row at a time
#pragma unroll
    for (int u = 0; u < LUNR; ++u) {
      const int r = r0 + u * rtc;
      if (r < nrows) {
        xr[u] = ld8(x + (size_t)r * C);
        if (RES != 0) gr[u] = ld8(res + (size_t)r * C);
      }
    }
#pragma unroll
    for (int u = 0; u < LUNR; ++u) {
      const int r = r0 + u * rtc;
      if (r < nrows) {
        float f[LV], g[LV];
        unpack4bf(xr[u], f);
        if (RES != 0) unpack4bf(gr[u], g);
        const float4 gate = SE ? *reinterpret_cast<const float4*>(p.se + (size_t)(r / HW) * C + c0)
                               : make_float4(1.f, 1.f, 1.f, 1.f);
        const float gt[4] = {gate.x, gate.y, gate.z, gate.w};
        static_assert(LV == 4, "the gate is read as one float4");
#pragma unroll
        for (int j = 0; j < LV; ++j) {
          f[j] = fmaf(f[j], sc[j], sh[j]);
          // (gate and residual as ONE fused multiply-add, written out: FR_PRO_RESBN_SE forms the same tensor elsewhere and
          // must round the same way whatever the compiler contracts)
          if (SE && RES != 0) f[j] = fmaf(f[j], gt[j], fmaf(g[j], rs[j], rh[j]));
          else if (SE) f[j] *= gt[j];
          else if (RES != 0) f[j] += fmaf(g[j], rs[j], rh[j]);
        }
        const uint2 o = pack4bf(f);
        *reinterpret_cast<uint2*>(out + (size_t)r * C) = o;
        if (STATS) {
          float q[LV];
          unpack4bf(o, q);  // statistics of what the next layer will actually read
#pragma unroll
          for (int j = 0; j < LV; ++j) {
            acc[0][j] += q[j];
            acc[1][j] = fmaf(q[j], q[j], acc[1][j]);
          }
        }
      }
    }
  }
  if (STATS) block_col_reduce<2, LV>(acc, red, p.part + (size_t)blockIdx.x * 2 * C, C, cpr, tid);
}

// SLOPE: BN followed by PReLU (the stem, model_irse.py:141-142): g' = g * prelu'(u), u = x*scale + shift, and the third
// partial row collects the slope gradient sum g*u*[u <= 0].  SE: BN followed by the SE excite (bottleneck_IR_SE,
// model_irse.py:86-87): g' = g * se[b][c] + gse[b][c], both [B][C] fp32 (a few KB per image, L2 hits).
__device__ __forceinline__ void se_gprime(const FrBnBwdArgs& p, int r, int c0, float* v) {
  const int b = r / p.rows_per_image;
  const float4 s4 = *reinterpret_cast<const float4*>(p.se + (size_t)b * p.C + c0);
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.gse) g4 = *reinterpret_cast<const float4*>(p.gse + (size_t)b * p.C + c0);
  v[0] = fmaf(v[0], s4.x, g4.x);
  v[1] = fmaf(v[1], s4.y, g4.y);
  v[2] = fmaf(v[2], s4.z, g4.z);
  v[3] = fmaf(v[3], s4.w, g4.w);
}

template <bool SLOPE, bool SE = false>
__global__ __launch_bounds__(NT, 6) void bn_bwd_reduce_lean_kernel(const FrBnBwdArgs p) {
  __shared__ float red[NT * 3 * LV];
  const int C = p.C, cpr = C / LV, tid = threadIdx.x;
  const int cc = tid % cpr, rt = tid / cpr, rtc = NT / cpr, c0 = cc * LV;
  const bf16_t* __restrict__ g = reinterpret_cast<const bf16_t*>(p.g) + c0;
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(p.x) + c0;
  float mu[LV], is[LV], sc[LV], sh[LV], sl[LV];
#pragma unroll
  for (int j = 0; j < LV; ++j) {
    mu[j] = p.mean[c0 + j];
    is[j] = p.invstd[c0 + j];
    sc[j] = SLOPE ? p.scale[c0 + j] : 1.f;
    sh[j] = SLOPE ? p.shift[c0 + j] : 0.f;
    sl[j] = SLOPE ? p.slope[c0 + j] : 1.f;
  }
  float acc[3][LV];
#pragma unroll
  for (int j = 0; j < LV; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.f;
  const int nrows = (int)p.rows, rstep = gridDim.x * rtc * LUNRB;
  for (int r0 = blockIdx.x * rtc * LUNRB + rt; r0 < nrows; r0 += rstep) {
    uint2 gr[LUNRB], xr[LUNRB];
#pragma unroll
    for (int u = 0; u < LUNRB; ++u) {
      const int r = r0 + u * rtc;
      if (r < nrows) {
        gr[u] = ld8(g + (size_t)r * C);
        xr[u] = ld8(x + (size_t)r * C);
      }
    }
#pragma unroll
    for (int u = 0; u < LUNRB; ++u) {
      if (r0 + u * rtc < nrows) {
        float gv[LV], xv[LV];
        unpack4bf(gr[u], gv);
        unpack4bf(xr[u], xv);
        if (SE) se_gprime(p, r0 + u * rtc, c0, gv);  // same expression as bn_bwd_gprime
#pragma unroll
        for (int j = 0; j < LV; ++j) {
          float v = gv[j];
          if (SLOPE) {  // same operation order as bn_bwd_gprime
            const float uu = fmaf(xv[j], sc[j], sh[j]);
            const bool pos = uu > 0.f;
            acc[2][j] += pos ? 0.f : v * uu;
            v = pos ? v : v * sl[j];
          }
          acc[0][j] += v;
          acc[1][j] = fmaf(v, (xv[j] - mu[j]) * is[j], acc[1][j]);
        }
      }
    }
  }
  block_col_reduce<3, LV>(acc, red, p.part + (size_t)blockIdx.x * 3 * C, C, cpr, tid);
}

// ADD: 0 none, 1 a tensor of the same geometry, 2 (round 4) the strided scatter of a stride-2 shortcut gradient -- add[b, h/2,
// w/2, c] lands on the pixels with even h and w (MaxPool2d(1, 2) / the 1x1 stride-2 shortcut convolution, model_irse.py:52-56).
// The four stage-entry units ran the general kernel for it: 387 us at 112x112 (1.2 GB of traffic, 200 us at HBM rate), 45-66
// us at the other three.
// NEXT (round 6): the rows of bn_bwd_reduce_lean_kernel<false> for the BatchNorm in front (input p.nx), from the rounded gx, in
// the same pass.  Rounds 2-5 built this three times and dropped it: at 14x14 the separate kernels run two waves per SIMD where
// this one (more registers) runs one, and they are latency-bound there.  On tensors that stream through HBM it is a pass less.
template <int ADD, bool SE = false, bool NEXT = false>
__global__ __launch_bounds__(NT, NEXT ? 4 : 6) void bn_bwd_apply_lean_kernel(const FrBnBwdArgs p) {
  __shared__ float nred[NEXT ? NT * 2 * LV : 1];
  const int C = p.C, cpr = C / LV, tid = threadIdx.x;
  const int cc = tid % cpr, rt = tid / cpr, rtc = NT / cpr, c0 = cc * LV;
  const bf16_t* __restrict__ g = reinterpret_cast<const bf16_t*>(p.g) + c0;
  const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(p.x) + c0;
  const bf16_t* __restrict__ add = reinterpret_cast<const bf16_t*>(p.add) + c0;
  bf16_t* __restrict__ gx = reinterpret_cast<bf16_t*>(p.gx) + c0;
  float mu[LV], is[LV], coef[LV], a[LV], bb[LV];
#pragma unroll
  for (int j = 0; j < LV; ++j) {
    mu[j] = p.mean[c0 + j];
    is[j] = p.invstd[c0 + j];
    coef[j] = (p.gamma ? p.gamma[c0 + j] : 1.f) * is[j];
    a[j] = p.s0[c0 + j] * p.inv_count;
    bb[j] = p.s1[c0 + j] * p.inv_count;
  }
  const bf16_t* __restrict__ nx = NEXT ? reinterpret_cast<const bf16_t*>(p.nx) + c0 : nullptr;
  float nmu[NEXT ? LV : 1], nis[NEXT ? LV : 1], nacc[2][LV];
  if (NEXT) {
#pragma unroll
    for (int j = 0; j < LV; ++j) {
      nmu[j] = p.nmean[c0 + j];
      nis[j] = p.ninvstd[c0 + j];
      nacc[0][j] = nacc[1][j] = 0.f;
    }
  }
  // (Round 2-3 carried a NEXT variant here that also formed the backward sums of the BatchNorm in front from the rounded
  // gx -- bit-identical, one pass less, and 0.04-0.6 ms SLOWER per step beside the weight gradients of the side stream: 80
  // registers, one wave per SIMD where the two separate kernels run two.  Removed in round 4, ABI v4.  Round 5 rebuilt it once
  // more for the 128-workgroup weight-gradient schedule, where half of the CUs carry no side-stream workgroup: bit-identical
  // again, 14.98-15.03 against 14.51-14.59 ms per step, three same-box alternations; not kept.)
  const int nrows = (int)p.rows, rstep = gridDim.x * rtc * LUNRB;
  const unsigned W = (unsigned)p.W, HW = (unsigned)p.rows_per_image, Wh = W >> 1, HWq = HW >> 2;  // ADD == 2
  const float invW = 1.0f / (float)p.W, invHW = 1.0f / (float)p.rows_per_image;
  for (int r0 = blockIdx.x * rtc * LUNRB + rt; r0 < nrows; r0 += rstep) {
    uint2 gr[LUNRB], xr[LUNRB], er[LUNRB], nr[NEXT ? LUNRB : 1];
    bool hit[LUNRB];
#pragma unroll
    for (int u = 0; u < LUNRB; ++u) {
      const int r = r0 + u * rtc;
      hit[u] = false;
      if (r < nrows) {
        gr[u] = ld8(g + (size_t)r * C);
        xr[u] = ld8(x + (size_t)r * C);
        if (NEXT) nr[u] = ld8(nx + (size_t)r * C);
        if (ADD == 1) er[u] = ld8(add + (size_t)r * C);
        if (ADD == 2) {
          unsigned b, rem, h, w;
          fast_divmod((unsigned)r, HW, invHW, b, rem);  // rows < 2^24 per launch is checked by the launcher
          fast_divmod(rem, W, invW, h, w);
          hit[u] = ((h | w) & 1u) == 0u;
          if (hit[u]) er[u] = ld8(add + ((size_t)b * HWq + (h >> 1) * Wh + (w >> 1)) * C);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < LUNRB; ++u) {
      const int r = r0 + u * rtc;
      if (r < nrows) {
        float gv[LV], xv[LV], e[LV], o[LV];
        unpack4bf(gr[u], gv);
        unpack4bf(xr[u], xv);
        if (ADD == 1 || (ADD == 2 && hit[u])) unpack4bf(er[u], e);
        if (SE) se_gprime(p, r, c0, gv);
#pragma unroll
        for (int j = 0; j < LV; ++j) {
          o[j] = coef[j] * (gv[j] - a[j] - (xv[j] - mu[j]) * is[j] * bb[j]);
          if (ADD == 1 || (ADD == 2 && hit[u])) o[j] += e[j];
        }
        const uint2 packed = pack4bf(o);
        *reinterpret_cast<uint2*>(gx + (size_t)r * C) = packed;
        if (NEXT) {  // bn_bwd_reduce_lean_kernel<false> on what the tensor now holds
          float gq[LV], xn[LV];
          unpack4bf(packed, gq);
          unpack4bf(nr[u], xn);
#pragma unroll
          for (int j = 0; j < LV; ++j) {
            nacc[0][j] += gq[j];
            nacc[1][j] = fmaf(gq[j], (xn[j] - nmu[j]) * nis[j], nacc[1][j]);
          }
        }
      }
    }
  }
  if (NEXT) block_col_reduce<2, LV>(nacc, nred, p.npart + (size_t)blockIdx.x * 2 * C, C, cpr, tid);
}

// ------------------------------------------------------------------------------------------ SE
// pooled[b][c] = mean_hw(x*scale+shift) = scale*mean_hw(x)+shift : one block per (image, 64-channel... ) simple:
// grid = B blocks, threads [row-thread][chunk] over the image's HW rows.
// SNT threads per image: 1024 for the gradient squeeze (GS; round 3) -- with 256 a thread walked 25 rows of a 14x14 image in
// four dependent trips, and at 128 images per GPU only half the CUs had a block at all; 32 row-threads per channel chunk
// have every row in flight at once (IR-SE-101 bs 128: 18.8 -> see DESIGN section 3 per fused launch).
constexpr int SNT_GS = 1024;
template <typename T, bool GS, int SNT = NT>
__global__ __launch_bounds__(SNT) void se_pool_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ out,
                                                     int HW, int C) {
  constexpr int VEC = Elt<T>::VEC;
  __shared__ float red[SNT * VEC];
  const int cpr = C / VEC, tid = threadIdx.x, b = blockIdx.x;
  for (int cbase = 0; cbase < cpr; cbase += SNT) {  // C/VEC <= 256 always here, loop kept for generality
    const int cc = (tid % (cpr < SNT ? cpr : SNT)) + cbase;
    const int rtc = cpr < SNT ? SNT / cpr : 1, rt = cpr < SNT ? tid / cpr : 0;
    const int c0 = cc * VEC;
    float acc[1][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[0][j] = 0.f;
    // eight rows of a thread in flight, added in the same order as before (one row per trip cost one HBM round trip per
    // row: 22.8 us per 14x14 unit at 128 images for 25.7 MB)
    constexpr int UN = 8;
    float scv[VEC], shv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      scv[j] = GS ? scale[c0 + j] : 0.f;
      shv[j] = GS ? shift[c0 + j] : 0.f;
    }
    for (int r0 = rt; r0 < HW; r0 += rtc * UN) {
      U128 xv[UN], gq[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int r = r0 + u * rtc;
        if (r < HW) {
          xv[u] = ld16(x + ((size_t)b * HW + r) * C + c0);
          if (GS) gq[u] = ld16(g + ((size_t)b * HW + r) * C + c0);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int r = r0 + u * rtc;
        if (r < HW) {
          float f[VEC];
          unpack16<T>(xv[u], f);
          if (GS) {
            float gv[VEC];
            unpack16<T>(gq[u], gv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[0][j] = fmaf(gv[j], fmaf(f[j], scv[j], shv[j]), acc[0][j]);
          } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[0][j] += f[j];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[tid * VEC + j] = acc[0][j];
    __syncthreads();
    if (rt == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float s = 0.f;
        for (int r = 0; r < rtc; ++r) s += red[(r * cpr + (cc - cbase)) * VEC + j];
        if (GS) out[(size_t)b * C + c0 + j] = s;
        else out[(size_t)b * C + c0 + j] = fmaf(s / (float)HW, scale[c0 + j], shift[c0 + j]);
      }
    }
    __syncthreads();
  }
}

// The same squeeze from per-strip column sums a strip convolution's STATS epilogue already produced: part rows are
// [strip][2][C] with the strips of image b at rows b*NS .. b*NS+NS-1, so the activation is not read again.
__global__ void se_pool_parts_kernel(const float* __restrict__ part, int NS, const float* __restrict__ scale,
                                     const float* __restrict__ shift, float* __restrict__ out, int C, float inv_hw) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < NS; ++k) s += part[((size_t)(b * NS + k) * 2) * C + c];
    out[(size_t)b * C + c] = fmaf(s * inv_hw, scale[c], shift[c]);
  }
}

// s = sigmoid(W2 relu(W1 pooled)); one block per image.  PARTS: the squeeze of se_pool_parts_kernel (same arithmetic, same
// order) happens here and `pooled` is an OUTPUT (the weight gradient reads it) -- one launch less per IR-SE unit.
template <bool PARTS>
__global__ void se_mlp_fwd_kernel(const float* __restrict__ pooled_in, const float* __restrict__ w1,
                                  const float* __restrict__ w2, float* __restrict__ hidden, float* __restrict__ s,
                                  int C, int R, const float* __restrict__ part, int NS,
                                  const float* __restrict__ scale, const float* __restrict__ shift, float inv_hw,
                                  float* __restrict__ pooled_out, int NV = 2, const float* __restrict__ xm = nullptr,
                                  float* __restrict__ om = nullptr) {
  extern __shared__ float sm[];  // [C] pooled, [R] hidden
  float* pv = sm;
  float* hv = sm + C;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += blockDim.x) {
    if (PARTS) {
      float t = 0.f;
      for (int k = 0; k < NS; ++k) t += part[((size_t)(b * NS + k) * NV) * C + c];
      const float v = fmaf(t * inv_hw, scale[c], shift[c]);
      pv[c] = v;
      pooled_out[(size_t)b * C + c] = v;
    } else {
      pv[c] = pooled_in[(size_t)b * C + c];
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
  for (int r0 = wave; r0 < R; r0 += 4 * nw) {  // four hidden units of a wave in flight (se_mlp_bwd_body); bit-identical sums
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = lane; c < C; c += 64) {
      const float pc = pv[c];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * nw;
        acc[u] = fmaf(w1[(size_t)(r < R ? r : r0) * C + c], pc, acc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * nw;
      const float t = wave_sum(acc[u]);
      if (lane == 0 && r < R) {
        const float h = t > 0.f ? t : 0.f;
        hv[r] = h;
        hidden[(size_t)b * R + r] = h;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int r = 0; r < R; ++r) acc = fmaf(w2[(size_t)c * R + r], hv[r], acc);
    const float gate = 1.0f / (1.0f + __expf(-acc));
    s[(size_t)b * C + c] = gate;
    if (PARTS && om) {
      // per-image moments of the unit's output o = (a*y + bb)*g + x from the moments of y (this image's strip rows), the
      // cross moment sum(y*x) and the moments of x: no pass over o (fr_se_pool_parts_mlp_fwd_res)
      double Sy = 0.0, Syy = 0.0, Syx = 0.0;
      for (int k = 0; k < NS; ++k) {
        const float* row = part + ((size_t)(b * NS + k) * NV) * C + c;
        Sy += (double)row[0];
        Syy += (double)row[C];
        Syx += (double)row[2 * (size_t)C];
      }
      const double a = (double)scale[c], bb = (double)shift[c], g = (double)gate, hw = 1.0 / (double)inv_hw;
      const double Sx = (double)xm[((size_t)b * 2 + 0) * C + c], Sxx = (double)xm[((size_t)b * 2 + 1) * C + c];
      om[((size_t)b * 2 + 0) * C + c] = (float)(g * (a * Sy + bb * hw) + Sx);
      om[((size_t)b * 2 + 1) * C + c] =
          (float)(g * g * (a * a * Syy + 2.0 * a * bb * Sy + bb * bb * hw) + 2.0 * g * (a * Syx + bb * Sx) + Sxx);
    }
  }
}

// per-image (sum, sum of squares) of an NHWC bf16 tensor: the head of a chain of fr_se_pool_parts_mlp_fwd_res launches.
// One block per image; thread = (row group, 8-channel chunk).
__global__ __launch_bounds__(256) void image_moments_kernel(const bf16_t* __restrict__ x, int HW, int C,
                                                            float* __restrict__ out) {
  __shared__ float red[256 * 16];
  const int b = blockIdx.x, tid = threadIdx.x, cpr = C / 8, cc = tid % cpr, rt = tid / cpr, rtc = 256 / cpr;
  float a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a0[j] = a1[j] = 0.f;
  const bf16_t* base = x + (size_t)b * HW * C + cc * 8;
  for (int r = rt; r < HW; r += rtc) {
    float f[8];
    unpack16<bf16_t>(ld16(base + (size_t)r * C), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0[j] += f[j];
      a1[j] = fmaf(f[j], f[j], a1[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[tid * 16 + j] = a0[j];
    red[tid * 16 + 8 + j] = a1[j];
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) {
    const int k = i / C, c = i - k * C;
    float t = 0.f;
    for (int g = 0; g < rtc; ++g) t += red[(g * cpr + c / 8) * 16 + k * 8 + (c & 7)];
    out[((size_t)b * 2 + k) * C + c] = t;
  }
}

// backward of the MLP, image part: one block per image.  gz (gradient at the fc2 output) and gh (at the fc1 output)
// are written out for the weight-gradient kernel below -- accumulating dW1/dW2 with atomics from 256 blocks onto the
// same few thousand addresses took 110 us per SE module (and an unordered sum); gz and gh are [B][C] / [B][R] floats.
// gs_row: this image's gs[C] -- global memory, or LDS when the squeeze ran in the same block (se_gscale_mlp_bwd_kernel)
__device__ __forceinline__ void se_mlp_bwd_body(const float* gs_row, const float* __restrict__ s,
                                                const float* __restrict__ hidden, const float* __restrict__ w1,
                                                const float* __restrict__ w2, float* __restrict__ gpooled,
                                                float* __restrict__ gz_out, float* __restrict__ gh_out, int C, int R,
                                                float inv_hw, float* gz, float* gh) {
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += blockDim.x) {
    const float sv = s[(size_t)b * C + c];
    const float v = gs_row[c] * sv * (1.f - sv);
    gz[c] = v;
    gz_out[(size_t)b * C + c] = v;
  }
  __syncthreads();
  // Round 6: a wave's hidden units four at a time -- the loads of four units in flight together, then their reductions (one
  // unit per trip was load -> 6 shuffles -> store, R / waves dependent trips: 15 us per launch for a 256 x 16 product).
  // Same partial sums, same order: bit-identical.
  const int wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
  for (int r0 = wave; r0 < R; r0 += 4 * nw) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = lane; c < C; c += 64) {
      const float zc = gz[c];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * nw;
        acc[u] = fmaf(w2[(size_t)c * R + (r < R ? r : r0)], zc, acc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * nw;
      const float t = wave_sum(acc[u]);
      if (lane == 0 && r < R) {
        const float v = hidden[(size_t)b * R + r] > 0.f ? t : 0.f;
        gh[r] = v;
        gh_out[(size_t)b * R + r] = v;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int r = 0; r < R; ++r) acc = fmaf(w1[(size_t)r * C + c], gh[r], acc);
    gpooled[(size_t)b * C + c] = acc * inv_hw;
  }
}

__global__ void se_mlp_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ s,
                                  const float* __restrict__ hidden, const float* __restrict__ w1,
                                  const float* __restrict__ w2, float* __restrict__ gpooled, float* __restrict__ gz_out,
                                  float* __restrict__ gh_out, int C, int R, float inv_hw) {
  extern __shared__ float sm[];  // [C] gz (grad at fc2 output), [R] gh
  se_mlp_bwd_body(gs + (size_t)blockIdx.x * C, s, hidden, w1, w2, gpooled, gz_out, gh_out, C, R, inv_hw, sm, sm + C);
}

// fr_se_gscale + the image part of fr_se_mlp_bwd in one launch (both are one block per image): gs[b][:] never leaves LDS.
// The squeeze is se_pool_kernel<T, true>'s loop, row for row (C / VEC <= NT: one pass), so the pair is bit-identical.
template <typename T>
__global__ __launch_bounds__(SNT_GS) void se_gscale_mlp_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ s,
                                                               const float* __restrict__ hidden,
                                                               const float* __restrict__ w1,
                                                               const float* __restrict__ w2, float* __restrict__ gpooled,
                                                               float* __restrict__ gz_out, float* __restrict__ gh_out,
                                                               int HW, int C, int R, float inv_hw) {
  constexpr int VEC = Elt<T>::VEC;
  __shared__ float red[SNT_GS * VEC];
  extern __shared__ float sm[];  // [C] gs, [C] gz, [R] gh
  const int cpr = C / VEC, tid = threadIdx.x, b = blockIdx.x;
  const int cc = tid % cpr, rtc = SNT_GS / cpr, rt = tid / cpr, c0 = cc * VEC;
  float acc[VEC], scv[VEC], shv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    acc[j] = 0.f;
    scv[j] = scale[c0 + j];
    shv[j] = shift[c0 + j];
  }
  constexpr int UN = 8;
  for (int r0 = rt; r0 < HW; r0 += rtc * UN) {
    U128 xv[UN], gq[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int r = r0 + u * rtc;
      if (r < HW) {
        xv[u] = ld16(x + ((size_t)b * HW + r) * C + c0);
        gq[u] = ld16(g + ((size_t)b * HW + r) * C + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int r = r0 + u * rtc;
      if (r < HW) {
        float f[VEC], gv[VEC];
        unpack16<T>(xv[u], f);
        unpack16<T>(gq[u], gv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(gv[j], fmaf(f[j], scv[j], shv[j]), acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[tid * VEC + j] = acc[j];
  __syncthreads();
  if (rt == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = 0.f;
      for (int r = 0; r < rtc; ++r) t += red[(r * cpr + cc) * VEC + j];
      sm[c0 + j] = t;
    }
  }
  __syncthreads();
  se_mlp_bwd_body(sm, s, hidden, w1, w2, gpooled, gz_out, gh_out, C, R, inv_hw, sm + C, sm + 2 * C);
}

// Round 5: the gradient squeeze over S row slices per image, 256-thread workgroups (B x S of them).  The fused kernel above is
// one 1024-thread workgroup per image: at batch 128 (IR-SE-101) half the CUs have none, and with four waves per SIMD it finds
// no room beside the resident weight-gradient workgroups of the side stream (18 us per unit in the step for 25 MB of reads).
// part[(b*S + s)][c] = sum over the rows of slice s of g * (x*scale + shift); se_mlp_bwd_parts_kernel adds the S rows of an
// image in slice order and runs the MLP part.
// Round 6, SUMS: the same pass also leaves, per slice, what the backward of BN2 needs of (g, x): with the excite gate the
// BatchNorm sees g' = g * s[b][c] + gse[b][c] (se_gprime), both constant over an image, so
//   sum g'        = sum_b ( s_b * G_b  + HW * gse_b ),          G_b  = sum over the image of g
//   sum g' * xhat = sum_b ( s_b * GX_b + gse_b * XH_b ),        GX_b = sum g * xhat,  XH_b = sum xhat,  xhat = (x - mean) * invstd
// and fr_bn_bwd_reduce -- a third pass over (g, x) behind the MLP, 14-19 us per squeeze-excite unit on the main stream's
// chain -- disappears: part[(b*S + s)][4][C] = gs, G, GX, XH.
template <typename T, bool SUMS = false>
__global__ __launch_bounds__(256) void se_gsq_part_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ part, int HW, int C, int S,
                                                          const float* __restrict__ mean = nullptr,
                                                          const float* __restrict__ invstd = nullptr) {
  constexpr int VEC = Elt<T>::VEC;
  constexpr int NV = SUMS ? 4 : 1;
  __shared__ float red[256 * VEC];
  const int cpr = C / VEC, tid = threadIdx.x, b = blockIdx.x, sl = blockIdx.y;
  const int cc = tid % cpr, rtc = 256 / cpr, rt = tid / cpr, c0 = cc * VEC;
  const int r_lo = (int)((long long)HW * sl / S), r_hi = (int)((long long)HW * (sl + 1) / S);
  float acc[NV][VEC], scv[VEC], shv[VEC], muv[SUMS ? VEC : 1], isv[SUMS ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k][j] = 0.f;
    scv[j] = scale[c0 + j];
    shv[j] = shift[c0 + j];
    if (SUMS) {
      muv[j] = mean[c0 + j];
      isv[j] = invstd[c0 + j];
    }
  }
  constexpr int UN = SUMS ? 4 : 8;
  for (int r0 = r_lo + rt; r0 < r_hi; r0 += rtc * UN) {
    U128 xv[UN], gq[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      int r = r0 + u * rtc;
      r = r < r_hi ? r : r_hi - 1;  // clamp: the batch of loads stays branch-free (the duplicate is not added)
      xv[u] = ld16(x + ((size_t)b * HW + r) * C + c0);
      gq[u] = ld16(g + ((size_t)b * HW + r) * C + c0);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (r0 + u * rtc < r_hi) {
        float f[VEC], gv[VEC];
        unpack16<T>(xv[u], f);
        unpack16<T>(gq[u], gv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          acc[0][j] = fmaf(gv[j], fmaf(f[j], scv[j], shv[j]), acc[0][j]);
          if (SUMS) {
            const float xh = (f[j] - muv[j]) * isv[j];
            acc[1][j] += gv[j];
            acc[2][j] = fmaf(gv[j], xh, acc[2][j]);
            acc[3][j] += xh;
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (k) __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[tid * VEC + j] = acc[k][j];
    __syncthreads();
    if (rt == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = 0.f;
        for (int r = 0; r < rtc; ++r) t += red[(r * cpr + cc) * VEC + j];
        part[(((size_t)b * S + sl) * NV + k) * C + c0 + j] = t;
      }
    }
  }
}

__global__ __launch_bounds__(256) void se_mlp_bwd_parts_kernel(const float* __restrict__ part, int S,
                                                               const float* __restrict__ s, const float* __restrict__ hidden,
                                                               const float* __restrict__ w1, const float* __restrict__ w2,
                                                               float* __restrict__ gpooled, float* __restrict__ gz_out,
                                                               float* __restrict__ gh_out, int C, int R, float inv_hw,
                                                               float* __restrict__ bn_part = nullptr) {
  extern __shared__ float sm[];  // [C] gs, [C] gz, [R] gh
  const int b = blockIdx.x;
  const int NV = bn_part ? 4 : 1;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float t = 0.f;
    for (int k = 0; k < S; ++k) t += part[(((size_t)b * S + k) * NV) * C + c];
    sm[c] = t;
  }
  __syncthreads();
  se_mlp_bwd_body(sm, s, hidden, w1, w2, gpooled, gz_out, gh_out, C, R, inv_hw, sm + C, sm + 2 * C);
  if (bn_part) {  // this image's rows of the BN2-backward sums (se_gsq_part_kernel<SUMS>): [B][2][C], added by fr_reduce_parts
    const float hw = 1.0f / inv_hw;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {  // the thread that wrote gpooled[b][c] above reads it back
      float G = 0.f, GX = 0.f, XH = 0.f;
      for (int k = 0; k < S; ++k) {
        const float* row = part + (((size_t)b * S + k) * 4) * C + c;
        G += row[C];
        GX += row[2 * (size_t)C];
        XH += row[3 * (size_t)C];
      }
      const float sv = s[(size_t)b * C + c], gse = gpooled[(size_t)b * C + c];
      bn_part[((size_t)b * 2 + 0) * C + c] = fmaf(sv, G, hw * gse);
      bn_part[((size_t)b * 2 + 1) * C + c] = fmaf(sv, GX, gse * XH);
    }
  }
}

// weight part: dW1[r][c] = sum_b gh[b][r] * pooled[b][c],  dW2[c][r] = sum_b gz[b][c] * hidden[b][r].  Block = 64
// channels x 4 batch quarters of one r; each thread walks its quarter with 8 independent loads in flight, the four
// partial sums are added in quarter order through LDS (reproducible); overwrites.
__global__ __launch_bounds__(256) void se_mlp_wgrad_kernel(const float* __restrict__ gz, const float* __restrict__ gh,
                                                           const float* __restrict__ hidden,
                                                           const float* __restrict__ pooled, float* __restrict__ dw1,
                                                           float* __restrict__ dw2, int B, int C, int R) {
  __shared__ float part[2][4][64];
  const int r = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const int per = (B + 3) / 4, b0 = q * per, b1 = b0 + per < B ? b0 + per : B;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float g_[8], p_[8], z_[8], h_[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        g_[u] = gh[(size_t)(b + u) * R + r];
        p_[u] = pooled[(size_t)(b + u) * C + c];
        z_[u] = gz[(size_t)(b + u) * C + c];
        h_[u] = hidden[(size_t)(b + u) * R + r];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a1 = fmaf(g_[u], p_[u], a1);
        a2 = fmaf(z_[u], h_[u], a2);
      }
    }
    for (; b < b1; ++b) {
      a1 = fmaf(gh[(size_t)b * R + r], pooled[(size_t)b * C + c], a1);
      a2 = fmaf(gz[(size_t)b * C + c], hidden[(size_t)b * R + r], a2);
    }
  }
  part[0][q][threadIdx.x & 63] = a1;
  part[1][q][threadIdx.x & 63] = a2;
  __syncthreads();
  if (q == 0 && c < C) {
    const int t = threadIdx.x;
    dw1[(size_t)r * C + c] = ((part[0][0][t] + part[0][1][t]) + part[0][2][t]) + part[0][3][t];
    dw2[(size_t)c * R + r] = ((part[1][0][t] + part[1][1][t]) + part[1][2][t]) + part[1][3][t];
  }
}

// ------------------------------------------------------------------------------------------ dropout
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, float p) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f) >= p;
}

// out[b][(hw)*C + c] = keep ? (x*scale+shift)/(1-p) : 0 ; mask index = reference flatten order b*(C*HW)+c*HW+hw
template <typename T, bool BWD>
__global__ void bn_dropout_kernel(const T* __restrict__ x, T* __restrict__ out, const float* __restrict__ scale,
                                  const float* __restrict__ shift, long long rows, int C, int HW, float p,
                                  uint64_t seed) {
  constexpr int VEC = Elt<T>::VEC;
  const int cpr = C / VEC;
  const long long total = rows * cpr;
  const float keep_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cpr;
    const int c0 = (int)(i - r * cpr) * VEC;
    const long long b = r / HW;
    const int hw = (int)(r - b * HW);
    float f[VEC];
    unpack16<T>(ld16(x + r * C + c0), f);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = BWD ? f[j] : fmaf(f[j], scale[c0 + j], shift[c0 + j]);
      if (p > 0.f) {
        const uint64_t idx = (uint64_t)b * (uint64_t)(C * HW) + (uint64_t)(c0 + j) * HW + hw;
        v = drop_keep(seed, idx, p) ? v * keep_scale : 0.f;
      }
      f[j] = v;
    }
    st16(out + r * C + c0, pack16<T>(f));
  }
}

// The same with the activation of the Linear layer in the REFERENCE's flatten order (round 4): a[b][c*HW + hw], what
// Flatten() of the NCHW tensor gives (model_irse.py:146), so that Linear(25088, 512) runs on the master weight as it lies
// (linear_gemm.hip).  One workgroup per (image, 64 channels): the [HW][64] tile goes through LDS and leaves as one contiguous
// 64*HW-element run (forward), or arrives as one and leaves as HW rows of 128 B (backward, out of place: g_cm -> NHWC).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_dropout_cm_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int C, int HW, float p,
                                                            uint64_t seed) {
  constexpr int VEC = Elt<T>::VEC;
  constexpr int CPR = 64 / VEC;  // 16-byte chunks per pixel of this workgroup's 64 channels
  extern __shared__ __attribute__((aligned(16))) char smem_cm[];
  float* tile = reinterpret_cast<float*>(smem_cm);  // [64][HW + 1]
  const int LD = HW + 1;
  const int b = blockIdx.x, cg = blockIdx.y, tid = threadIdx.x;
  const float keep_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const size_t seg = (size_t)b * C * HW + (size_t)cg * 64 * HW;  // first element of the contiguous c-major run
  if (!BWD) {
    for (int e = tid; e < HW * CPR; e += 256) {
      const int hw = e / CPR, ch = e - hw * CPR;
      const int c0 = cg * 64 + ch * VEC;
      float f[VEC];
      unpack16<T>(ld16(in + ((size_t)b * HW + hw) * C + c0), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = fmaf(f[j], scale[c0 + j], shift[c0 + j]);
        if (p > 0.f) {
          const uint64_t idx = (uint64_t)b * (uint64_t)(C * HW) + (uint64_t)(c0 + j) * HW + hw;
          v = drop_keep(seed, idx, p) ? v * keep_scale : 0.f;
        }
        tile[(ch * VEC + j) * LD + hw] = v;
      }
    }
    __syncthreads();
    for (int e = tid; e * VEC < 64 * HW; e += 256) {
      float f[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int idx = e * VEC + j, cl = idx / HW;
        f[j] = tile[cl * LD + (idx - cl * HW)];
      }
      st16(out + seg + (size_t)e * VEC, pack16<T>(f));
    }
  } else {
    for (int e = tid; e * VEC < 64 * HW; e += 256) {
      float f[VEC];
      unpack16<T>(ld16(in + seg + (size_t)e * VEC), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int idx = e * VEC + j, cl = idx / HW;
        float v = f[j];
        if (p > 0.f) v = drop_keep(seed, (uint64_t)seg + (uint64_t)idx, p) ? v * keep_scale : 0.f;
        tile[cl * LD + (idx - cl * HW)] = v;
      }
    }
    __syncthreads();
    for (int e = tid; e < HW * CPR; e += 256) {
      const int hw = e / CPR, ch = e - hw * CPR;
      float f[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = tile[(ch * VEC + j) * LD + hw];
      st16(out + ((size_t)b * HW + hw) * C + cg * 64 + ch * VEC, pack16<T>(f));
    }
  }
}

// ------------------------------------------------------------------------------------------ weight packing
// src fp32 [Cout][taps][Cin]; wp [Cout][taps][Cin] (T), wt [Cin][taps][Cout] (T)
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wp, T* __restrict__ wt, int Cout,
                                   int taps, int Cin) {
  __shared__ float tile[32][33];
  // grid: x over Cin/32, y over Cout/32, z over taps
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, tap = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
      v = w[((size_t)co * taps + tap) * Cin + ci];
      if (wp) Elt<T>::st(wp + ((size_t)co * taps + tap) * Cin + ci, v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (wt) {
    for (int r = ty; r < 32; r += 8) {
      const int ci = ci0 + r, co = co0 + tx;
      if (co < Cout && ci < Cin) Elt<T>::st(wt + ((size_t)ci * taps + tap) * Cout + co, tile[tx][r]);
    }
  }
}

// all convolution weights of the network in ONE launch: chunk = (tensor index, 32x32 tile index)
// (round 4: tensors whose channel counts are multiples of 64 -- every 3x3 and shortcut convolution of the IR nets -- are moved
// by 64 x 64 tiles with 16-byte accesses on all three sides; chunk.y < 0 marks such a tile: -(index + 1).  The 32 x 32 tiles
// with 4-byte loads and 2-byte stores ran the 350 MB of a step at 3 TB/s.)
// element offset of the 8 values (n, tap, k .. k + 7), k % 8 == 0, in MFMA-fragment order (FrConvArgs.w_frag)
__device__ __forceinline__ size_t frag_offset(int n, int tap, int k, int taps, int K) {
  return ((((size_t)(n >> 4) * taps + tap) * (K >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (n & 15)) * 8;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const FrPackTensor* __restrict__ table,
                                                                 const int2* __restrict__ chunks) {
  __shared__ float tile[32][33];
  __shared__ __attribute__((aligned(16))) unsigned short big[64 * 72];  // [co][ci] bf16, rows padded to 144 B
  const int2 ch = chunks[blockIdx.x];
  const FrPackTensor t = table[ch.x];
  if (ch.y < 0) {
    if constexpr (sizeof(T) == 2) {
      const int idx = -ch.y - 1;
      const int tx_n = t.Cin / 64, ty_n = t.Cout / 64;
      const int tap = idx / (tx_n * ty_n), rem = idx - tap * (tx_n * ty_n);
      const int ci0 = (rem % tx_n) * 64, co0 = (rem / tx_n) * 64;
      const int r = threadIdx.x >> 2, q = (threadIdx.x & 3) * 16;
      const float* src = t.w + ((size_t)(co0 + r) * t.taps + tap) * t.Cin + ci0 + q;
      float f[16];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * k);
        f[4 * k] = v.x;
        f[4 * k + 1] = v.y;
        f[4 * k + 2] = v.z;
        f[4 * k + 3] = v.w;
      }
      if (t.oscale) {
        const float sc = t.oscale[co0 + r];
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] *= sc;
      }
      const U128 lo = pack16<bf16_t>(f), hi = pack16<bf16_t>(f + 8);
      if (t.wp && (t.frag & 1)) {
        bf16_t* dst = reinterpret_cast<bf16_t*>(t.wp);
        st16(dst + frag_offset(co0 + r, tap, ci0 + q, t.taps, t.Cin), lo);
        st16(dst + frag_offset(co0 + r, tap, ci0 + q + 8, t.taps, t.Cin), hi);
      } else if (t.wp) {
        bf16_t* dst = reinterpret_cast<bf16_t*>(t.wp) + ((size_t)(co0 + r) * t.taps + tap) * t.Cin + ci0 + q;
        st16(dst, lo);
        st16(dst + 8, hi);
      }
      if (t.wt) {
        st16(big + r * 72 + q, lo);
        st16(big + r * 72 + q + 8, hi);
        __syncthreads();
        // row r of the transposed tile = input channel ci0 + r, output channels co0 + q .. + 15
        unsigned short g[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) g[k] = big[(q + k) * 72 + r];
        uint32_t w32[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w32[k] = (uint32_t)g[2 * k] | ((uint32_t)g[2 * k + 1] << 16);
        U128 a, b;
        a.x = w32[0]; a.y = w32[1]; a.z = w32[2]; a.w = w32[3];
        b.x = w32[4]; b.y = w32[5]; b.z = w32[6]; b.w = w32[7];
        if (t.frag & 2) {
          bf16_t* dst = reinterpret_cast<bf16_t*>(t.wt);
          st16(dst + frag_offset(ci0 + r, tap, co0 + q, t.taps, t.Cout), a);
          st16(dst + frag_offset(ci0 + r, tap, co0 + q + 8, t.taps, t.Cout), b);
        } else {
          bf16_t* dst = reinterpret_cast<bf16_t*>(t.wt) + ((size_t)(ci0 + r) * t.taps + tap) * t.Cout + co0 + q;
          st16(dst, a);
          st16(dst + 8, b);
        }
      }
    }
    return;
  }
  const int tx_n = (t.Cin + 31) / 32, ty_n = (t.Cout + 31) / 32;
  const int tap = ch.y / (tx_n * ty_n), rem = ch.y - tap * (tx_n * ty_n);
  const int ci0 = (rem % tx_n) * 32, co0 = (rem / tx_n) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  T* wp = reinterpret_cast<T*>(t.wp);
  T* wt = reinterpret_cast<T*>(t.wt);
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < t.Cout && ci < t.Cin) {
      v = t.w[((size_t)co * t.taps + tap) * t.Cin + ci];
      if (t.oscale) v *= t.oscale[co];  // inference: BatchNorm scale of the layer's output folded into the weights
      if (wp) Elt<T>::st(wp + ((size_t)co * t.taps + tap) * t.Cin + ci, v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (wt) {
    for (int r = ty; r < 32; r += 8) {
      const int ci = ci0 + r, co = co0 + tx;
      if (co < t.Cout && ci < t.Cin) Elt<T>::st(wt + ((size_t)ci * t.taps + tap) * t.Cout + co, tile[tx][r]);
    }
  }
}

// Linear(25088,512): dir 0: torch fp32 [O][C*HW] -> packed T [O][HW*C] (+ transposed T [HW*C][O])
//                    dir 1: packed fp32 grad [O][HW*C] -> torch fp32 grad [O][C*HW]
template <typename T>
__global__ void permute_linear_kernel(const float* __restrict__ in, T* __restrict__ out, T* __restrict__ wt,
                                      float* __restrict__ gout, int O, int C, int HW, int dir) {
  const long long total = (long long)O * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    if (dir == 0) {
      // i indexes the packed layout (o, hw, c; c fastest): coalesced packed writes, HW-strided fp32 reads
      const int c = (int)(i % C);
      const int hw = (int)((i / C) % HW);
      const int o = (int)(i / ((long long)C * HW));
      const float v = in[(long long)o * C * HW + (long long)c * HW + hw];
      Elt<T>::st(out + i, v);
      if (wt) Elt<T>::st(wt + ((long long)hw * C + c) * O + o, v);
    } else {
      // i indexes the packed layout (o, hw, c; c fastest): coalesced reads of the packed gradient
      const int c = (int)(i % C);
      const int hw = (int)((i / C) % HW);
      const int o = (int)(i / ((long long)C * HW));
      gout[(long long)o * C * HW + (long long)c * HW + hw] = in[i];
    }
  }
}

// Tiled forms (C % 64 == 0): a block moves one (o, 64-channel) tile through LDS so that both the torch-layout side
// (64*HW contiguous floats) and the packed side (64 contiguous channels per pixel) are accessed in full segments.
// The tile is [64][HW | 1] floats: an odd row length keeps both access patterns bank-conflict free.
template <typename T>
__global__ __launch_bounds__(256) void permute_linear_tile_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                                  float* __restrict__ gout, int C, int HW, int dir) {
  extern __shared__ float ptile[];
  const int HWP = HW | 1;
  const int o = blockIdx.y, c0 = blockIdx.x * 64, n = 64 * HW;
  const size_t torch_base = ((size_t)o * C + c0) * HW;  // [o][c0..c0+63][hw]: n contiguous floats
  const size_t packed_base = (size_t)o * HW * C + c0;   // [o][hw][c0..c0+63]
  if (dir == 0) {
    for (int i = threadIdx.x; i < n; i += 256) ptile[(i / HW) * HWP + i % HW] = in[torch_base + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
      const int hw = i >> 6, c = i & 63;
      Elt<T>::st(out + packed_base + (size_t)hw * C + c, ptile[c * HWP + hw]);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) {
      const int hw = i >> 6, c = i & 63;
      ptile[c * HWP + hw] = in[packed_base + (size_t)hw * C + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) gout[torch_base + i] = ptile[(i / HW) * HWP + i % HW];
  }
}

// out[k][r] = in[r][k] for an [R][K] matrix, 64 x 64 tiles
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int K) {
  __shared__ T tt[64][66];
  const int k0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4)
    if (r0 + j < R && k0 + tx < K) tt[j][tx] = in[(size_t)(r0 + j) * K + k0 + tx];
  __syncthreads();
  for (int j = ty; j < 64; j += 4)
    if (k0 + j < K && r0 + tx < R) out[(size_t)(k0 + j) * R + r0 + tx] = tt[tx][j];
}

template <typename T>
__global__ void pack_stem_kernel(const float* __restrict__ w, long long s_o, long long s_c, long long s_h,
                                 long long s_w, T* __restrict__ wp, int Cout, int C, int ldk) {
  const int total = Cout * ldk;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int o = i / ldk, k = i - o * ldk;
    float v = 0.f;
    if (k < 9 * C) {
      const int tap = k / C, c = k - tap * C;
      v = w[o * s_o + c * s_c + (tap / 3) * s_h + (tap % 3) * s_w];
    }
    Elt<T>::st(wp + i, v);
  }
}
__global__ void unpack_stem_grad_kernel(const float* __restrict__ gp, float* __restrict__ gw, long long s_o,
                                        long long s_c, long long s_h, long long s_w, int Cout, int C, int ldk) {
  const int total = Cout * 9 * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int o = i / (9 * C), k = i - o * 9 * C;
    const int tap = k / C, c = k - tap * C;
    gw[o * s_o + c * s_c + (tap / 3) * s_h + (tap % 3) * s_w] = gp[(size_t)o * ldk + k];
  }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    Elt<TO>::st(out + i, Elt<TI>::ld(in + i));
}

inline int grid_for(long long work_items, int per_block, int cap = 2048) {
  long long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16, name)          \
  if ((dtype) == FR_F32) {                                    \
    CALL_F32;                                                 \
  } else if ((dtype) == FR_BF16) {                            \
    CALL_BF16;                                                \
  } else {                                                    \
    FR_UNSUPPORTED(name ": dtype must be FR_F32 or FR_BF16"); \
  }

inline bool chan_ok(int C, int dtype) {
  const int vec = dtype == FR_F32 ? 4 : 8;
  const int cpr = C / vec;
  return C % vec == 0 && cpr <= NT && NT % cpr == 0;
}

inline bool lean_ok(int C) {  // 4-channel threads: C/4 threads per row must tile the 256-thread block
  const int cpr = C / 4;
  return C % 4 == 0 && cpr <= NT && NT % cpr == 0;
}

}  // namespace

extern "C" int fr_stem_im2col(const float* x, const float* avg, void* out, int B, int H, int W, int C, int Cavg,
                              int ldk, int dtype, void* stream) {
  if (9 * (C + Cavg) > ldk) FR_UNSUPPORTED("fr_stem_im2col: ldk too small");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FR_BF16 && C == 3 && (long long)B * H * W < (1ll << 31) &&
      ((Cavg == 0 && ldk == 32) || (Cavg == 3 && ldk == 64 && avg))) {
    const int g = grid_for((long long)B * H * W, 256, 1 << 16);
    if (Cavg == 0)
      hipLaunchKernelGGL((stem_im2col_rows_kernel<3, 32>), dim3(g), dim3(256), 0, st, x, avg, (bf16_t*)out, B, H, W);
    else
      hipLaunchKernelGGL((stem_im2col_rows_kernel<6, 64>), dim3(g), dim3(256), 0, st, x, avg, (bf16_t*)out, B, H, W);
    FR_LAUNCH_CHECK();
  }
  const int grid = grid_for((long long)B * H * W * 10, 256, 1 << 16);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(grid), dim3(256), 0, st, x, avg, (float*)out, B, H,
                                W, C, Cavg, ldk),
             hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, x, avg, (bf16_t*)out, B,
                                H, W, C, Cavg, ldk),
             "fr_stem_im2col");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_bn_finalize(const float* part, int nparts, int C, double count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, int64_t* nbt, float* mean, float* invstd, float* scale,
                              float* shift, void* stream) {
  FrBnFin f;
  f.count = count;
  f.gamma = gamma;
  f.beta = beta;
  f.eps = eps;
  f.momentum = momentum;
  f.running_mean = running_mean;
  f.running_var = running_var;
  f.nbt = (long long*)nbt;
  f.mean = mean;
  f.invstd = invstd;
  f.scale = scale;
  f.shift = shift;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(RT), 0, (hipStream_t)stream, part, nparts, C, f);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_bn_finalize_res(const float* part, int nparts, int C, const FrBnFinArgs* bn, const float* in_mean,
                                  const float* in_invstd, float in_eps, const FrBnFinArgs* next, void* stream) {
  if (!part || nparts < 1 || C < 1 || !bn || (next && (!in_mean || !in_invstd)))
    FR_UNSUPPORTED("fr_bn_finalize_res: part and bn are required, in_mean / in_invstd with next");
  if (!(bn->count > 0.0) || !bn->mean || !bn->invstd || !bn->scale || !bn->shift ||
      (next && (!next->mean || !next->invstd || !next->scale || !next->shift)))
    FR_UNSUPPORTED("fr_bn_finalize_res: count and the four coefficient vectors of both BatchNorms are required");
  FrBnFin f = fin_of(*bn), fn = next ? fin_of(*next) : f;
  fn.count = bn->count;  // same pixels
  hipLaunchKernelGGL(bn_finalize_res_kernel, dim3((C + 7) / 8), dim3(RT), 0, (hipStream_t)stream, part, nparts, C, f,
                     in_mean, in_invstd, in_eps, fn, next != nullptr);
  FR_LAUNCH_CHECK();
}

namespace {
__global__ void bn_eval_coeffs_multi_kernel(const FrBnEvalEntry* __restrict__ table) {
  const FrBnEvalEntry e = table[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
    const float is = 1.0f / sqrtf(e.rv[c] + e.eps);
    e.mean[c] = e.rm[c];
    e.invstd[c] = is;
    e.scale[c] = e.gamma[c] * is;
    e.shift[c] = e.beta[c] - e.rm[c] * e.gamma[c] * is;
  }
}
}  // namespace

extern "C" int fr_bn_eval_coeffs_multi(const FrBnEvalEntry* table_dev, int n, void* stream) {
  if (n < 1) return 0;
  hipLaunchKernelGGL(bn_eval_coeffs_multi_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table_dev);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_bn_eval_coeffs(const float* rm, const float* rv, const float* gamma, const float* beta, float eps,
                                 int C, float* mean, float* invstd, float* scale, float* shift, void* stream) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, rm, rv, gamma,
                     beta, eps, C, mean, invstd, scale, shift);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_reduce_parts(const float* part, int nparts, int K, int C, float* o0, float* o1, float* o2,
                               void* stream) {
  if (K < 1 || K > 3) FR_UNSUPPORTED("fr_reduce_parts: K must be 1..3");
  FR_LAUNCH_KERNEL(reduce_parts_kernel, dim3((K * C + 7) / 8), dim3(RT), 0, (hipStream_t)stream, part, nparts, K,
                     C, o0, o1, o2);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_channel_stats(const void* x, long long rows, int C, float* part, int nblocks, int dtype,
                                void* stream) {
  if (!chan_ok(C, dtype)) FR_UNSUPPORTED("fr_channel_stats: unsupported channel count");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(channel_stats_kernel<float>, dim3(nblocks), dim3(NT), 0, st, (const float*)x, rows,
                                C, part),
             hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, dim3(nblocks), dim3(NT), 0, st, (const bf16_t*)x,
                                rows, C, part),
             "fr_channel_stats");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_bn_apply(const FrApplyArgs* args, int dtype, void* stream) {
  if (!chan_ok(args->C, dtype)) FR_UNSUPPORTED("fr_bn_apply: unsupported channel count");
  if (args->nblocks < 1) FR_UNSUPPORTED("fr_bn_apply: nblocks < 1");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FR_BF16 && lean_ok(args->C) && !args->slope &&
      !(args->res_kind == 1 && args->res_stride > 1) && (long long)args->B * args->H * args->W < (1ll << 31)) {
    const dim3 grid(args->nblocks), blk(NT);
#define LEAN_APPLY(RES, ST)                                                                          \
  do {                                                                                               \
    if (args->se) hipLaunchKernelGGL((bn_apply_lean_kernel<RES, ST, true>), grid, blk, 0, st, *args); \
    else hipLaunchKernelGGL((bn_apply_lean_kernel<RES, ST>), grid, blk, 0, st, *args);               \
  } while (0)
    if (args->part) {
      if (args->res_kind == 0) LEAN_APPLY(0, true);
      else if (args->res_kind == 1) LEAN_APPLY(1, true);
      else LEAN_APPLY(2, true);
    } else {
      if (args->res_kind == 0) LEAN_APPLY(0, false);
      else if (args->res_kind == 1) LEAN_APPLY(1, false);
      else LEAN_APPLY(2, false);
    }
#undef LEAN_APPLY
    FR_LAUNCH_CHECK();
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(args->nblocks), dim3(NT), 0, st, *args),
             hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(args->nblocks), dim3(NT), 0, st, *args),
             "fr_bn_apply");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_bn_bwd_reduce(const FrBnBwdArgs* args, int dtype, void* stream) {
  if (!chan_ok(args->C, dtype)) FR_UNSUPPORTED("fr_bn_bwd_reduce: unsupported channel count");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FR_BF16 && lean_ok(args->C) && args->rows < (1ll << 31) && !(args->se && args->slope) &&
      (!args->slope || (args->scale && args->shift))) {
    const dim3 grid(args->nblocks), blk(NT);
    if (args->se) hipLaunchKernelGGL((bn_bwd_reduce_lean_kernel<false, true>), grid, blk, 0, st, *args);
    else if (args->slope) hipLaunchKernelGGL((bn_bwd_reduce_lean_kernel<true, false>), grid, blk, 0, st, *args);
    else hipLaunchKernelGGL((bn_bwd_reduce_lean_kernel<false, false>), grid, blk, 0, st, *args);
    FR_LAUNCH_CHECK();
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(args->nblocks), dim3(NT), 0, st, *args),
             hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(args->nblocks), dim3(NT), 0, st, *args),
             "fr_bn_bwd_reduce");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_bn_bwd_apply(const FrBnBwdArgs* args, int dtype, void* stream) {
  if (!chan_ok(args->C, dtype)) FR_UNSUPPORTED("fr_bn_bwd_apply: unsupported channel count");
  hipStream_t st = (hipStream_t)stream;
  // add_kind 2 on the lean kernel: stride 2, even geometry, row indices that the float-reciprocal division handles exactly
  const bool scatter_ok = args->add_kind != 2 || (args->add_stride == 2 && !args->se && args->H % 2 == 0 && args->W % 2 == 0 &&
                                                  args->rows_per_image == args->H * args->W && args->rows < (1ll << 24));
  if (dtype == FR_BF16 && lean_ok(args->C) && !args->slope && scatter_ok && args->rows < (1ll << 31)) {
    const dim3 grid(args->nblocks), blk(NT);
    if (args->se) {
      if (args->add_kind == 1) hipLaunchKernelGGL((bn_bwd_apply_lean_kernel<1, true>), grid, blk, 0, st, *args);
      else hipLaunchKernelGGL((bn_bwd_apply_lean_kernel<0, true>), grid, blk, 0, st, *args);
    } else if (args->nx) {
      if (!args->nmean || !args->ninvstd || !args->npart) FR_UNSUPPORTED("fr_bn_bwd_apply: nx needs nmean, ninvstd and npart");
      if (args->add_kind == 1) hipLaunchKernelGGL((bn_bwd_apply_lean_kernel<1, false, true>), grid, blk, 0, st, *args);
      else if (args->add_kind == 2) hipLaunchKernelGGL((bn_bwd_apply_lean_kernel<2, false, true>), grid, blk, 0, st, *args);
      else hipLaunchKernelGGL((bn_bwd_apply_lean_kernel<0, false, true>), grid, blk, 0, st, *args);
    } else {
      if (args->add_kind == 1) hipLaunchKernelGGL((bn_bwd_apply_lean_kernel<1, false>), grid, blk, 0, st, *args);
      else if (args->add_kind == 2) hipLaunchKernelGGL((bn_bwd_apply_lean_kernel<2, false>), grid, blk, 0, st, *args);
      else hipLaunchKernelGGL((bn_bwd_apply_lean_kernel<0, false>), grid, blk, 0, st, *args);
    }
    FR_LAUNCH_CHECK();
  }
  if (args->nx) FR_UNSUPPORTED("fr_bn_bwd_apply: nx (the sums of the BatchNorm in front) is served by the bf16 lean kernel only");
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(args->nblocks), dim3(NT), 0, st, *args),
             hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(args->nblocks), dim3(NT), 0, st, *args),
             "fr_bn_bwd_apply");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_pool(const void* x, const float* scale, const float* shift, float* pooled, int B, int HW, int C,
                          int dtype, void* stream) {
  if (!chan_ok(C, dtype)) FR_UNSUPPORTED("fr_se_pool: unsupported channel count");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((se_pool_kernel<float, false>), dim3(B), dim3(NT), 0, st, (const float*)x,
                                (const float*)nullptr, scale, shift, pooled, HW, C),
             hipLaunchKernelGGL((se_pool_kernel<bf16_t, false>), dim3(B), dim3(NT), 0, st, (const bf16_t*)x,
                                (const bf16_t*)nullptr, scale, shift, pooled, HW, C),
             "fr_se_pool");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_pool_parts(const float* part, int rows_per_image, const float* scale, const float* shift,
                                float* pooled, int B, int HW, int C, void* stream) {
  if (rows_per_image < 1 || B < 1 || C < 1) FR_UNSUPPORTED("fr_se_pool_parts: bad geometry");
  hipLaunchKernelGGL(se_pool_parts_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, part, rows_per_image, scale, shift,
                     pooled, C, 1.0f / (float)HW);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_gscale(const void* g, const void* x, const float* scale, const float* shift, float* gs, int B,
                            int HW, int C, int dtype, void* stream) {
  if (!chan_ok(C, dtype)) FR_UNSUPPORTED("fr_se_gscale: unsupported channel count");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((se_pool_kernel<float, true, SNT_GS>), dim3(B), dim3(SNT_GS), 0, st, (const float*)x,
                                (const float*)g, scale, shift, gs, HW, C),
             hipLaunchKernelGGL((se_pool_kernel<bf16_t, true, SNT_GS>), dim3(B), dim3(SNT_GS), 0, st, (const bf16_t*)x,
                                (const bf16_t*)g, scale, shift, gs, HW, C),
             "fr_se_gscale");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_mlp_fwd(const float* pooled, const float* w1, const float* w2, float* hidden, float* s, int B,
                             int C, int R, void* stream) {
  hipLaunchKernelGGL(se_mlp_fwd_kernel<false>, dim3(B), dim3(256), (C + R) * sizeof(float), (hipStream_t)stream, pooled,
                     w1, w2, hidden, s, C, R, (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, 0.f,
                     (float*)nullptr);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_pool_parts_mlp_fwd(const float* part, int rows_per_image, const float* scale, const float* shift,
                                        const float* w1, const float* w2, float* pooled, float* hidden, float* s, int B,
                                        int HW, int C, int R, void* stream) {
  if (rows_per_image < 1 || B < 1 || C < 1 || R < 1 || !part || !pooled) FR_UNSUPPORTED("fr_se_pool_parts_mlp_fwd: bad arguments");
  hipLaunchKernelGGL(se_mlp_fwd_kernel<true>, dim3(B), dim3(256), (C + R) * sizeof(float), (hipStream_t)stream,
                     (const float*)nullptr, w1, w2, hidden, s, C, R, part, rows_per_image, scale, shift,
                     1.0f / (float)HW, pooled);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_pool_parts_mlp_fwd_res(const float* part, int rows_per_image, int nv, const float* scale,
                                            const float* shift, const float* w1, const float* w2, float* pooled,
                                            float* hidden, float* s, int B, int HW, int C, int R, const float* xm, float* om,
                                            void* stream) {
  if (rows_per_image < 1 || B < 1 || C < 1 || R < 1 || !part || !pooled || nv != 3 || !xm || !om)
    FR_UNSUPPORTED("fr_se_pool_parts_mlp_fwd_res: rows of 3 vectors (FR_EPI_STATS_X), xm and om are required");
  hipLaunchKernelGGL(se_mlp_fwd_kernel<true>, dim3(B), dim3(256), (C + R) * sizeof(float), (hipStream_t)stream,
                     (const float*)nullptr, w1, w2, hidden, s, C, R, part, rows_per_image, scale, shift,
                     1.0f / (float)HW, pooled, nv, xm, om);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_image_moments(const void* x, int B, int HW, int C, float* out, void* stream) {
  if (!x || !out || B < 1 || HW < 1 || C < 8 || C % 8 || C / 8 > 256 || 256 % (C / 8))
    FR_UNSUPPORTED("fr_image_moments: bf16 [B][HW][C] with C / 8 a divisor of 256");
  hipLaunchKernelGGL(image_moments_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, HW, C, out);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_mlp_bwd(const float* gs, const float* s, const float* hidden, const float* pooled,
                             const float* w1, const float* w2, float* gpooled, float* dw1, float* dw2, float* gz,
                             float* gh, int B, int C, int R, int HW, void* stream) {
  if (!gz || !gh) FR_UNSUPPORTED("fr_se_mlp_bwd: gz [B][C] and gh [B][R] scratch are required");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(B), dim3(256), (C + R) * sizeof(float), st, gs, s, hidden, w1, w2, gpooled,
                     gz, gh, C, R, 1.0f / (float)HW);
  hipLaunchKernelGGL(se_mlp_wgrad_kernel, dim3((C + 63) / 64, R), dim3(256), 0, st, gz, gh, hidden, pooled, dw1, dw2, B,
                     C, R);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_gscale_slices(int B, int HW) {
  int S = (512 + B - 1) / B;  // >= 512 workgroups of one wave per SIMD: they fit beside a resident weight-gradient workgroup
  if (S > 8) S = 8;
  if (S > HW) S = HW;
  return S < 1 ? 1 : S;
}

extern "C" int fr_se_gscale_mlp_bwd(const void* g, const void* x, const float* scale, const float* shift, const float* s,
                                    const float* hidden, const float* pooled, const float* w1, const float* w2,
                                    float* gpooled, float* dw1, float* dw2, float* gz, float* gh, float* gs_part, int B, int C,
                                    int R, int HW, int dtype, void* stream) {
  if (!gz || !gh) FR_UNSUPPORTED("fr_se_gscale_mlp_bwd: gz [B][C] and gh [B][R] scratch are required");
  if (!chan_ok(C, dtype)) FR_UNSUPPORTED("fr_se_gscale_mlp_bwd: unsupported channel count");
  const int vec = dtype == FR_BF16 ? 8 : 4;
  if (C % vec || C / vec > SNT_GS || SNT_GS % (C / vec)) FR_UNSUPPORTED("fr_se_gscale_mlp_bwd: C / vector width must divide the block");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(2 * C + R) * sizeof(float);
  if (gs_part && C / vec <= 256 && 256 % (C / vec) == 0) {  // round 5: the squeeze over row slices (scratch [B][slices][C])
    const int S = fr_se_gscale_slices(B, HW);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(se_gsq_part_kernel<float>, dim3(B, S), dim3(256), 0, st, (const float*)x, (const float*)g,
                                  scale, shift, gs_part, HW, C, S),
               hipLaunchKernelGGL(se_gsq_part_kernel<bf16_t>, dim3(B, S), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)g,
                                  scale, shift, gs_part, HW, C, S),
               "fr_se_gscale_mlp_bwd");
    hipLaunchKernelGGL(se_mlp_bwd_parts_kernel, dim3(B), dim3(256), lds, st, gs_part, S, s, hidden, w1, w2, gpooled, gz, gh, C,
                       R, 1.0f / (float)HW);
    if (dw1 || dw2)
      hipLaunchKernelGGL(se_mlp_wgrad_kernel, dim3((C + 63) / 64, R), dim3(256), 0, st, gz, gh, hidden, pooled, dw1, dw2, B, C,
                         R);
    FR_LAUNCH_CHECK();
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(se_gscale_mlp_bwd_kernel<float>, dim3(B), dim3(SNT_GS), lds, st, (const float*)x,
                                (const float*)g, scale, shift, s, hidden, w1, w2, gpooled, gz, gh, HW, C, R,
                                1.0f / (float)HW),
             hipLaunchKernelGGL(se_gscale_mlp_bwd_kernel<bf16_t>, dim3(B), dim3(SNT_GS), lds, st, (const bf16_t*)x,
                                (const bf16_t*)g, scale, shift, s, hidden, w1, w2, gpooled, gz, gh, HW, C, R,
                                1.0f / (float)HW),
             "fr_se_gscale_mlp_bwd");
  if (dw1 || dw2)
    hipLaunchKernelGGL(se_mlp_wgrad_kernel, dim3((C + 63) / 64, R), dim3(256), 0, st, gz, gh, hidden, pooled, dw1, dw2, B,
                       C, R);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_mlp_wgrad(const float* gz, const float* gh, const float* hidden, const float* pooled, float* dw1,
                               float* dw2, int B, int C, int R, void* stream) {
  if (!gz || !gh || !hidden || !pooled || !dw1 || !dw2) FR_UNSUPPORTED("fr_se_mlp_wgrad: every pointer is required");
  hipLaunchKernelGGL(se_mlp_wgrad_kernel, dim3((C + 63) / 64, R), dim3(256), 0, (hipStream_t)stream, gz, gh, hidden, pooled,
                     dw1, dw2, B, C, R);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_se_gscale_mlp_bwd_sums(const void* g, const void* x, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, const float* s, const float* hidden,
                                         const float* w1, const float* w2, float* gpooled, float* gz, float* gh, float* gs_part,
                                         float* bn_part, int B, int C, int R, int HW, int dtype, void* stream) {
  if (!g || !x || !scale || !shift || !mean || !invstd || !s || !hidden || !w1 || !w2 || !gpooled || !gz || !gh || !gs_part ||
      !bn_part)
    FR_UNSUPPORTED("fr_se_gscale_mlp_bwd_sums: every pointer is required");
  if (!chan_ok(C, dtype)) FR_UNSUPPORTED("fr_se_gscale_mlp_bwd_sums: unsupported channel count");
  const int vec = dtype == FR_BF16 ? 8 : 4;
  if (C % vec || C / vec > 256 || 256 % (C / vec)) FR_UNSUPPORTED("fr_se_gscale_mlp_bwd_sums: C / vector width must divide 256");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(2 * C + R) * sizeof(float);
  const int S = fr_se_gscale_slices(B, HW);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((se_gsq_part_kernel<float, true>), dim3(B, S), dim3(256), 0, st, (const float*)x,
                                (const float*)g, scale, shift, gs_part, HW, C, S, mean, invstd),
             hipLaunchKernelGGL((se_gsq_part_kernel<bf16_t, true>), dim3(B, S), dim3(256), 0, st, (const bf16_t*)x,
                                (const bf16_t*)g, scale, shift, gs_part, HW, C, S, mean, invstd),
             "fr_se_gscale_mlp_bwd_sums");
  hipLaunchKernelGGL(se_mlp_bwd_parts_kernel, dim3(B), dim3(256), lds, st, gs_part, S, s, hidden, w1, w2, gpooled, gz, gh, C, R,
                     1.0f / (float)HW, bn_part);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_bn_dropout(const void* x, void* out, const float* scale, const float* shift, long long rows, int C,
                             int HW, float p, uint64_t seed, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == FR_F32 ? 4 : 8;
  if (C % vec) FR_UNSUPPORTED("fr_bn_dropout: unsupported channel count");
  const int grid = grid_for(rows * (C / vec), 256, 4096);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((bn_dropout_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)x,
                                (float*)out, scale, shift, rows, C, HW, p, seed),
             hipLaunchKernelGGL((bn_dropout_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x,
                                (bf16_t*)out, scale, shift, rows, C, HW, p, seed),
             "fr_bn_dropout");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_dropout_bwd(void* g, long long rows, int C, int HW, float p, uint64_t seed, int dtype,
                              void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == FR_F32 ? 4 : 8;
  if (C % vec) FR_UNSUPPORTED("fr_dropout_bwd: unsupported channel count");
  const int grid = grid_for(rows * (C / vec), 256, 4096);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((bn_dropout_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)g,
                                (float*)g, (const float*)nullptr, (const float*)nullptr, rows, C, HW, p, seed),
             hipLaunchKernelGGL((bn_dropout_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)g,
                                (bf16_t*)g, (const float*)nullptr, (const float*)nullptr, rows, C, HW, p, seed),
             "fr_dropout_bwd");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_bn_dropout_cm(const void* x, void* out, const float* scale, const float* shift, int B, int C, int HW,
                                float p, uint64_t seed, int dtype, void* stream) {
  if (B < 1 || C % 64 || HW < 1 || (64 * (HW + 1) * 4) > 64 * 1024) FR_UNSUPPORTED("fr_bn_dropout_cm: C % 64 == 0, HW <= 255");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(B, C / 64);
  const size_t lds = (size_t)64 * (HW + 1) * 4;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((bn_dropout_cm_kernel<float, false>), grid, dim3(256), lds, st, (const float*)x, (float*)out,
                                scale, shift, C, HW, p, seed),
             hipLaunchKernelGGL((bn_dropout_cm_kernel<bf16_t, false>), grid, dim3(256), lds, st, (const bf16_t*)x,
                                (bf16_t*)out, scale, shift, C, HW, p, seed),
             "fr_bn_dropout_cm");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_dropout_bwd_cm(const void* g_cm, void* out, int B, int C, int HW, float p, uint64_t seed, int dtype,
                                 void* stream) {
  if (B < 1 || C % 64 || HW < 1 || (64 * (HW + 1) * 4) > 64 * 1024) FR_UNSUPPORTED("fr_dropout_bwd_cm: C % 64 == 0, HW <= 255");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(B, C / 64);
  const size_t lds = (size_t)64 * (HW + 1) * 4;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((bn_dropout_cm_kernel<float, true>), grid, dim3(256), lds, st, (const float*)g_cm,
                                (float*)out, nullptr, nullptr, C, HW, p, seed),
             hipLaunchKernelGGL((bn_dropout_cm_kernel<bf16_t, true>), grid, dim3(256), lds, st, (const bf16_t*)g_cm,
                                (bf16_t*)out, nullptr, nullptr, C, HW, p, seed),
             "fr_dropout_bwd_cm");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_pack_weight(const float* w, void* wp, void* wt, int Cout, int taps, int Cin, int dtype,
                              void* stream) {
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, taps);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_weight_kernel<float>, grid, dim3(256), 0, st, w, (float*)wp, (float*)wt, Cout,
                                taps, Cin),
             hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, grid, dim3(256), 0, st, w, (bf16_t*)wp, (bf16_t*)wt,
                                Cout, taps, Cin),
             "fr_pack_weight");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_pack_weights_multi(const FrPackTensor* table_dev, const int32_t* chunks_dev, int nchunks, int dtype,
                                     void* stream) {
  if (nchunks <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_weights_multi_kernel<float>, dim3(nchunks), dim3(256), 0, st, table_dev,
                                (const int2*)chunks_dev),
             hipLaunchKernelGGL(pack_weights_multi_kernel<bf16_t>, dim3(nchunks), dim3(256), 0, st, table_dev,
                                (const int2*)chunks_dev),
             "fr_pack_weights_multi");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_permute_linear(const float* in, void* out, void* wt, int O, int C, int HW, int dir, int dtype,
                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (C % 64 == 0 && 64 * (HW | 1) * 4 <= 64 * 1024) {
    const dim3 grid(C / 64, O), tgrid((C * HW + 63) / 64, (O + 63) / 64);
    const size_t lds = (size_t)64 * (HW | 1) * sizeof(float);
    if (dir == 1) {
      hipLaunchKernelGGL(permute_linear_tile_kernel<float>, grid, dim3(256), lds, st, in, (float*)nullptr, (float*)out,
                         C, HW, 1);
    } else if (dtype == FR_F32) {
      hipLaunchKernelGGL(permute_linear_tile_kernel<float>, grid, dim3(256), lds, st, in, (float*)out, (float*)nullptr,
                         C, HW, 0);
      if (wt) hipLaunchKernelGGL(transpose2d_kernel<float>, tgrid, dim3(256), 0, st, (const float*)out, (float*)wt, O, C * HW);
    } else if (dtype == FR_BF16) {
      hipLaunchKernelGGL(permute_linear_tile_kernel<bf16_t>, grid, dim3(256), lds, st, in, (bf16_t*)out,
                         (float*)nullptr, C, HW, 0);
      if (wt) hipLaunchKernelGGL(transpose2d_kernel<bf16_t>, tgrid, dim3(256), 0, st, (const bf16_t*)out, (bf16_t*)wt, O, C * HW);
    } else {
      FR_UNSUPPORTED("fr_permute_linear: bad dtype");
    }
    FR_LAUNCH_CHECK();
  }
  const int grid = grid_for((long long)O * C * HW, 256, 8192);
  if (dir == 1) {
    hipLaunchKernelGGL(permute_linear_kernel<float>, dim3(grid), dim3(256), 0, st, in, (float*)nullptr,
                       (float*)nullptr, (float*)out, O, C, HW, 1);
    FR_LAUNCH_CHECK();
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(permute_linear_kernel<float>, dim3(grid), dim3(256), 0, st, in, (float*)out,
                                (float*)wt, (float*)nullptr, O, C, HW, 0),
             hipLaunchKernelGGL(permute_linear_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, in, (bf16_t*)out,
                                (bf16_t*)wt, (float*)nullptr, O, C, HW, 0),
             "fr_permute_linear");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_pack_stem(const float* w, long long s_o, long long s_c, long long s_h, long long s_w, void* wp,
                            int Cout, int C, int ldk, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for((long long)Cout * ldk, 256);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_stem_kernel<float>, dim3(grid), dim3(256), 0, st, w, s_o, s_c, s_h, s_w,
                                (float*)wp, Cout, C, ldk),
             hipLaunchKernelGGL(pack_stem_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w, s_o, s_c, s_h, s_w,
                                (bf16_t*)wp, Cout, C, ldk),
             "fr_pack_stem");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_unpack_stem_grad(const float* gp, float* gw, long long s_o, long long s_c, long long s_h,
                                   long long s_w, int Cout, int C, int ldk, void* stream) {
  const int grid = grid_for((long long)Cout * 9 * C, 256);
  hipLaunchKernelGGL(unpack_stem_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gp, gw, s_o, s_c, s_h,
                     s_w, Cout, C, ldk);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_cast(const void* in, void* out, long long n, int dtype_in, int dtype_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(n, 256 * 4, 4096);
  if (dtype_in == FR_F32 && dtype_out == FR_BF16) {
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)in, (bf16_t*)out,
                       n);
  } else if (dtype_in == FR_BF16 && dtype_out == FR_F32) {
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)in, (float*)out,
                       n);
  } else if (dtype_in == FR_F32 && dtype_out == FR_F32) {
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)in, (float*)out, n);
  } else {
    FR_UNSUPPORTED("fr_cast: unsupported dtype pair");
  }
  FR_LAUNCH_CHECK();
}
