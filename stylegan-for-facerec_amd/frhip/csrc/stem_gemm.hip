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
#include "tail.h"

namespace {

constexpr int SN = 64;  // output channels of the stem

// Round 4: the im2col rows need not exist.  X0 [B*112*112][K] is 205 MB written by fr_stem_im2col and read back by both
// GEMMs (410 MB + a launch per step in the forward pass, 205 MB more in the backward pass) to carry 38.5 MB of image.  With a
// StemSrc the two kernels build their 16-byte row chunks from the fp32 NCHW batch (and pSp's constant average image) in
// registers -- the same values, rounded to bf16 the same way, so the results are bit-identical to the X0 path.
struct StemSrc {
  const float* x;    // [B][C][H][W] fp32, or NULL: read the materialised rows
  const float* avg;  // [Cavg][H][W] or NULL
  int H, W, C, Cavg;
  float inv_w, inv_hw;
};

// im2col row `row` = pixel (b, h, w), elements k0 .. k0 + 7 (k = tap*CT + c, zero beyond 9*CT and outside the image)
template <int CT>
__device__ __forceinline__ U128 stem_row_chunk(const StemSrc& s, int row, int k0) {
  uint32_t b, rem, h, w;
  fast_divmod((uint32_t)row, (uint32_t)(s.H * s.W), s.inv_hw, b, rem);
  fast_divmod(rem, (uint32_t)s.W, s.inv_w, h, w);
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    const int tap = k / CT, c = k - tap * CT;
    const int th = tap / 3;
    const int sh = (int)h + th - 1, sw = (int)w + (tap - th * 3) - 1;
    const bool ok = k < 9 * CT && (unsigned)sh < (unsigned)s.H && (unsigned)sw < (unsigned)s.W;
    const int shc = ok ? sh : (int)h, swc = ok ? sw : (int)w;  // always a valid address; the value is dropped
    const int cc = k < 9 * CT ? c : 0;
    const float* p = (CT > 3 && cc >= s.C) ? s.avg + ((size_t)(cc - s.C) * s.H + shc) * s.W + swc
                                           : s.x + (((size_t)b * s.C + cc) * s.H + shc) * s.W + swc;
    const float v = *p;
    f[j] = ok ? v : 0.f;
  }
  return pack16<bf16_t>(f);
}

// ------------------------------------------------------------------------------------------ forward + BN statistics
// out[m][n] = sum_k X[m][k] W[n][k];  part[blk][0][n] = sum_m out, part[blk][1][n] = sum_m out^2 (of the rounded bf16)
template <int K, bool IMPL>
__global__ __launch_bounds__(256) void stem_gemm_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp,
                                                        bf16_t* __restrict__ out, float* __restrict__ part, int M,
                                                        const FrTail tail, const StemSrc src) {
  constexpr int CT = K == 32 ? 3 : 6;
  constexpr int KS = K / 32;
  constexpr int OSTR = SN * 2 + 16;                    // per-wave transpose tile [16 rows][64 ch], padded rows
  __shared__ __attribute__((aligned(16))) char tiles[4 * 16 * OSTR];
  __shared__ float red[4 * 2 * SN];
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
  const int ntiles = (M + 15) / 16;
  const int tstep = gridDim.x * 4;
  const int trips = (ntiles + tstep - 1) / tstep;  // same trip count for every wave: the loop body has barriers
  for (int it = 0; it < trips; ++it) {
    const int t = it * tstep + blockIdx.x * 4 + wave;
    const int row = t * 16 + fr;
    const bool ok = t < ntiles && row < M;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      s16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) {
        if (IMPL) af = __builtin_bit_cast(s16x8, stem_row_chunk<CT>(src, row, kk * 32 + fq * 8));
        else af = *reinterpret_cast<const s16x8*>(X + (size_t)row * K + kk * 32 + fq * 8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], af, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint2 o;
      o.x = pack2bf(acc[j][0], acc[j][1]);
      o.y = pack2bf(acc[j][2], acc[j][3]);
      *reinterpret_cast<uint2*>(tile + fr * OSTR + (j * 16 + fq * 4) * 2) = o;
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
    __syncthreads();
    // the wave's 16 x 128 B tile leaves as 16-byte stores: 8 lanes per row
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = lane + u * 64, r = c >> 3, c8 = c & 7;
      const int orow = t * 16 + r;
      if (t < ntiles && orow < M) st16(out + (size_t)orow * SN + c8 * 8, ld16(tile + r * OSTR + c8 * 16));
    }
    __syncthreads();
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
        red[(wave * 2 + 0) * SN + j * 16 + fq * 4 + r] = a;
        red[(wave * 2 + 1) * SN + j * 16 + fq * 4 + r] = c;
      }
    }
  __syncthreads();
  if (tid < 2 * SN) {
    const int k = tid / SN, n = tid - k * SN;
    st_part(part + ((size_t)blockIdx.x * 2 + k) * SN + n,
            red[(0 * 2 + k) * SN + n] + red[(1 * 2 + k) * SN + n] + red[(2 * 2 + k) * SN + n] + red[(3 * 2 + k) * SN + n]);
  }
  fr_tail<256>(tail, part, gridDim.x, gridDim.x, tiles, tid);  // in-launch BatchNorm statistics (tail.h)
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
struct StemBn {
  const bf16_t* y;  // BN input = stem GEMM output [M][64]
  const float *mean, *invstd, *scale, *shift, *slope, *gamma, *s0, *s1;
  float inv_count;
};

template <int K, bool BN, bool IMPL = false>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                         float* __restrict__ slab, int M, const StemBn bn,
                                                         const StemSrc src) {
  constexpr int CT = K == 32 ? 3 : 6;
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
  if (BN) {
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
          if (BN) {
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
        if (row0 + r < M) v[u] = IMPL ? stem_row_chunk<CT>(src, row0 + r, c * 8) : ld16(X + (size_t)(row0 + r) * K + c * 8);
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

namespace {
int stem_src(const float* x, const float* avg, int B, int H, int W, int C, int Cavg, int K, StemSrc* s, long long* M) {
  if (!x || B < 1 || H < 1 || W < 1) FR_UNSUPPORTED("stem (implicit im2col): x and a positive geometry are required");
  if (!((K == 32 && C == 3 && Cavg == 0) || (K == 64 && C + Cavg == 6 && C >= 1 && (Cavg == 0 || avg))))
    FR_UNSUPPORTED("stem (implicit im2col): 3 image channels (K = 32) or 6 channels in all, image + average image (K = 64)");
  *M = (long long)B * H * W;
  if (*M >= (1ll << 24)) FR_UNSUPPORTED("stem (implicit im2col): fewer than 2^24 pixels per launch");
  s->x = x;
  s->avg = avg;
  s->H = H;
  s->W = W;
  s->C = C;
  s->Cavg = Cavg;
  s->inv_w = 1.0f / (float)W;
  s->inv_hw = 1.0f / (float)(H * W);
  return 0;
}
}  // namespace

extern "C" int fr_stem_gemm(const void* X, const void* Wp, void* out, float* part, long long M, int K, int nblocks,
                            const FrTail* tail, void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 31) - 64 || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_gemm: K must be 32 or 64, 0 < M < 2^31, nblocks >= 1");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  FrTail t;
  if (fr_tail_prepare(tail ? *tail : FrTail{}, 2, SN, 1, &t, part != nullptr)) return -1;
  const StemSrc none = {};
  if (K == 32)
    hipLaunchKernelGGL((stem_gemm_kernel<32, false>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)Wp,
                       (bf16_t*)out, part, (int)M, t, none);
  else
    hipLaunchKernelGGL((stem_gemm_kernel<64, false>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)X, (const bf16_t*)Wp,
                       (bf16_t*)out, part, (int)M, t, none);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_gemm_x(const float* x, const float* avg, const void* Wp, void* out, float* part, int B, int H, int W,
                              int C, int Cavg, int K, int nblocks, const FrTail* tail, void* stream) {
  StemSrc src;
  long long M;
  if (stem_src(x, avg, B, H, W, C, Cavg, K, &src, &M)) return -1;
  if (nblocks < 1) FR_UNSUPPORTED("fr_stem_gemm_x: nblocks >= 1");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  FrTail t;
  if (fr_tail_prepare(tail ? *tail : FrTail{}, 2, SN, 1, &t, part != nullptr)) return -1;
  if (K == 32)
    hipLaunchKernelGGL((stem_gemm_kernel<32, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)nullptr,
                       (const bf16_t*)Wp, (bf16_t*)out, part, (int)M, t, src);
  else
    hipLaunchKernelGGL((stem_gemm_kernel<64, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)nullptr,
                       (const bf16_t*)Wp, (bf16_t*)out, part, (int)M, t, src);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_wgrad(const void* G, const void* X, float* slab, long long M, int K, int nblocks, void* stream) {
  if ((K != 32 && K != 64) || M < 1 || M >= (1ll << 31) - 64 || nblocks < 1)
    FR_UNSUPPORTED("fr_stem_wgrad: K must be 32 or 64, 0 < M < 2^31, nblocks >= 1");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const StemBn none = {};
  const StemSrc nosrc = {};
  if (K == 32)
    hipLaunchKernelGGL((stem_wgrad_kernel<32, false>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X,
                       slab, (int)M, none, nosrc);
  else
    hipLaunchKernelGGL((stem_wgrad_kernel<64, false>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X,
                       slab, (int)M, none, nosrc);
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
  const StemBn bn = {(const bf16_t*)Y, mean, invstd, scale, shift, slope, gamma, s0, s1, inv_count};
  const StemSrc nosrc = {};
  if (K == 32)
    hipLaunchKernelGGL((stem_wgrad_kernel<32, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X,
                       slab, (int)M, bn, nosrc);
  else
    hipLaunchKernelGGL((stem_wgrad_kernel<64, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X,
                       slab, (int)M, bn, nosrc);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_stem_wgrad_bn_x(const void* G, const void* Y, const float* x, const float* avg, const float* mean,
                                  const float* invstd, const float* scale, const float* shift, const float* slope,
                                  const float* gamma, const float* s0, const float* s1, float inv_count, float* slab, int B,
                                  int H, int W, int C, int Cavg, int K, int nblocks, void* stream) {
  StemSrc src;
  long long M;
  if (stem_src(x, avg, B, H, W, C, Cavg, K, &src, &M)) return -1;
  if (nblocks < 1) FR_UNSUPPORTED("fr_stem_wgrad_bn_x: nblocks >= 1");
  if (!Y || !mean || !invstd || !scale || !shift || !slope || !gamma || !s0 || !s1)
    FR_UNSUPPORTED("fr_stem_wgrad_bn_x: every BatchNorm / PReLU coefficient vector is required");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const StemBn bn = {(const bf16_t*)Y, mean, invstd, scale, shift, slope, gamma, s0, s1, inv_count};
  if (K == 32)
    hipLaunchKernelGGL((stem_wgrad_kernel<32, true, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G,
                       (const bf16_t*)nullptr, slab, (int)M, bn, src);
  else
    hipLaunchKernelGGL((stem_wgrad_kernel<64, true, true>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)G,
                       (const bf16_t*)nullptr, slab, (int)M, bn, src);
  FR_LAUNCH_CHECK();
}
