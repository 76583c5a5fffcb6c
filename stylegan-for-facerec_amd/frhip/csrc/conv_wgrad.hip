// frhip -- convolution / Linear weight gradient on MFMA for gfx950.
//
//   dw[co][tap][ci] += sum_{p in slice}  g[p][co] * pro( src[pixel(p, tap)][ci] )
//
// p enumerates the conv's output pixels (b, gh, gw); src is the conv input (NHWC), gathered with the same
// tap/stride/pad map as the forward pass.  taps = 1 with GH = GW = 1 is the dense  dW = g^T * a  of
// Linear(25088,512) and of the 512 x N_classes margin head (head/metrics.py:103 under autograd).
//
// GEMM view: M = Cout, N = Cin (per tap), K = pixels.  Both operands arrive "K-major" (a pixel row holds
// contiguous channels) while MFMA wants 8 consecutive k per lane, so the two [32 pixels][channels] LDS
// tiles are read *transposed*: bf16 through ds_read_b64_tr_b16 (two per fragment), f32 through eight
// ds_read_b32.  The k order inside a 32-pixel step is permuted identically for both operands (lane group
// q, element j -> pixel 4q+j for j<4, 16+4q+j-4 otherwise) which makes every transposed read bank-conflict
// free with rows padded by 32 B (bf16) / 16 B (f32).  Pixel slices (gridDim.y) are combined with fp32
// atomics, issued as whole 256-B row segments out of an LDS copy of the accumulator tile.
#include "common.h"
#include "frhip_internal.h"

namespace {

constexpr int NT = 256;
constexpr int KP = 32;  // pixels per K step

template <typename T, int BC>
struct TileRow {  // padded LDS row length in elements for a [KP][BC] tile
  static constexpr int value = BC + (sizeof(T) == 2 ? 16 : 4);
};

template <typename T>
__device__ __forceinline__ void read_frag_T(Frag<T>& f, const T* tile, int ldrow, int cbase, int lane);

// bf16: hardware transposing read.  16-lane group q reads the 4x16 block rows [r0, r0+4) x cols [cbase, cbase+16):
// lane i of the group supplies &tile[r0 + (i>>2)][cbase + 4*(i&3)] and receives column cbase+i of the 4 rows.
template <>
__device__ __forceinline__ void read_frag_T<bf16_t>(Frag<bf16_t>& f, const bf16_t* tile, int ldrow, int cbase,
                                                    int lane) {
  const int i = lane & 15, q = lane >> 4;
  const bf16_t* p0 = tile + (4 * q + (i >> 2)) * ldrow + cbase + 4 * (i & 3);
  const bf16_t* p1 = p0 + 16 * ldrow;
  typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;
  const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p0);
  const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p1);
  const s16x4 ai = __builtin_bit_cast(s16x4, a), bi = __builtin_bit_cast(s16x4, b);
  f.v = (s16x8){ai[0], ai[1], ai[2], ai[3], bi[0], bi[1], bi[2], bi[3]};
}
template <>
__device__ __forceinline__ void read_frag_T<float>(Frag<float>& f, const float* tile, int ldrow, int cbase,
                                                   int lane) {
  const int i = lane & 15, q = lane >> 4;
  const float* p0 = tile + (4 * q) * ldrow + cbase + i;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = p0[j * ldrow];
    f.v[4 + j] = p0[(16 + j) * ldrow];
  }
}

template <typename T, int BCO, int BCI, int PRO>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const FrWgradArgs p) {
  constexpr int VEC = Elt<T>::VEC;
  constexpr int LG = TileRow<T, BCO>::value, LA = TileRow<T, BCI>::value;
  constexpr int G_CPR = BCO / VEC, A_CPR = BCI / VEC;           // 16-B chunks per tile row
  constexpr int NG = (KP * G_CPR + NT - 1) / NT, NA = (KP * A_CPR + NT - 1) / NT;
  constexpr int WM = BCO / 2, WN = BCI / 2, TM = WM / 16, TN = WN / 16;
  constexpr int STAGE = KP * (LG + LA);                         // elements per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* S = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int co_tiles = (p.Cout + BCO - 1) / BCO, ci_tiles = (p.SC + BCI - 1) / BCI;
  int t = blockIdx.x;
  const int cot = t % co_tiles;
  t /= co_tiles;
  const int cit = t % ci_tiles;
  const int tap = t / ci_tiles;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int co0 = cot * BCO, ci0 = cit * BCI;

  const long long P = (long long)p.B * p.GH * p.GW;
  long long per = (P + p.nsplit - 1) / p.nsplit;
  per = (per + KP - 1) / KP * KP;
  const long long pbeg = per * blockIdx.y;
  long long pend = pbeg + per;
  if (pend > P) pend = P;
  const int nsteps = pbeg < pend ? (int)((pend - pbeg + KP - 1) / KP) : 0;

  const T* __restrict__ G = reinterpret_cast<const T*>(p.g);
  const T* __restrict__ X = reinterpret_cast<const T*>(p.src);
  const float inv_gw = 1.0f / (float)p.GW, inv_gh = 1.0f / (float)p.GH;

  U128 rg[NG], ra[NA];
  bool va[NA];
  float pa[VEC], pb[VEC];
  const int a_cc = tid % A_CPR;
  if (PRO != FR_PRO_NONE) {
    const int c = ci0 + a_cc * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      pa[j] = (c + j < p.SC) ? p.pro_a[c + j] : 0.f;
      pb[j] = (PRO == FR_PRO_BN && c + j < p.SC) ? p.pro_b[c + j] : 0.f;
    }
  }

  auto gload = [&](int step) {
    const long long pix0 = pbeg + (long long)step * KP;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int c = tid + i * NT;
      const int row = c / G_CPR, cc = c - row * G_CPR;
      const long long pp = pix0 + row;
      const int co = co0 + cc * VEC;
      if (c < KP * G_CPR && pp < pend && co < p.Cout) rg[i] = ld16(G + (size_t)pp * (size_t)p.ldg + co);
      else rg[i] = zero16();
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + i * NT;
      const int row = c / A_CPR;
      const long long pp = pix0 + row;
      const int ci = ci0 + a_cc * VEC;
      bool ok = (c < KP * A_CPR) && pp < pend && ci < p.SC;
      size_t off = 0;
      if (ok) {
        uint32_t q1, gw, b, gh;
        fast_divmod((uint32_t)pp, (uint32_t)p.GW, inv_gw, q1, gw);
        fast_divmod(q1, (uint32_t)p.GH, inv_gh, b, gh);
        const int sh = (int)gh * p.stride + kh - p.pad, sw = (int)gw * p.stride + kw - p.pad;
        ok = (unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW;
        off = ((size_t)b * p.SH * p.SW + (size_t)sh * p.SW + sw) * (size_t)p.lda + ci;
      }
      va[i] = ok;
      ra[i] = ok ? ld16(X + off) : zero16();
    }
  };
  auto lstore = [&](int stage) {
    T* gs = S + stage * STAGE;
    T* as = gs + KP * LG;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int c = tid + i * NT;
      const int row = c / G_CPR, cc = c - row * G_CPR;
      if (c < KP * G_CPR) st16(gs + row * LG + cc * VEC, rg[i]);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + i * NT;
      const int row = c / A_CPR;
      U128 v = ra[i];
      if (PRO != FR_PRO_NONE && va[i]) {
        float f[VEC];
        unpack16<T>(v, f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (PRO == FR_PRO_BN) f[j] = fmaf(f[j], pa[j], pb[j]);
          else f[j] = f[j] > 0.f ? f[j] : f[j] * pa[j];
        }
        v = pack16<T>(f);
      }
      if (c < KP * A_CPR) st16(as + row * LA + a_cc * VEC, v);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int stage = s & 1;
    const bool more = s + 1 < nsteps;
    if (more) gload(s + 1);
    const T* gs = S + stage * STAGE;
    const T* as = gs + KP * LG;
    Frag<T> fa[TM], fb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) read_frag_T<T>(fa[i], gs, LG, wm * WM + i * 16, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) read_frag_T<T>(fb[j], as, LA, wn * WN + j * 16, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mma16(fa[i], fb[j], acc[i][j]);
    if (more) lstore(stage ^ 1);
    __syncthreads();
  }
  if (nsteps == 0 && !p.slab) return;  // (an empty pixel slice still has to zero its slab)

  // ---- epilogue: accumulators -> LDS (half of the co rows at a time) -> row-contiguous fp32 atomics
  constexpr int CROW = BCI + 4;
  float* Cs = reinterpret_cast<float*>(smem);  // [BCO/2][CROW]
  const int frow = lane & 15, fq = lane >> 4;
  const int taps = p.KH * p.KW;
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    if (wm == h) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cs[(i * 16 + fq * 4 + r) * CROW + wn * WN + j * 16 + frow] = acc[i][j][r];
    }
    __syncthreads();
    for (int e = tid; e < WM * BCI; e += NT) {
      const int r = e / BCI, c = e - r * BCI;
      const int co = co0 + h * WM + r, ci = ci0 + c;
      if (co < p.Cout && ci < p.SC) {
        const size_t o = ((size_t)co * taps + tap) * (size_t)p.SC + ci;
        if (p.slab) p.slab[(size_t)blockIdx.y * ((size_t)p.Cout * taps * p.SC) + o] = Cs[r * CROW + c];  // own slab
        else atomicAdd(p.dw + o, Cs[r * CROW + c]);
      }
    }
  }
}

template <typename T, int BCO, int BCI>
constexpr int lds_bytes() {
  constexpr int stage = KP * (TileRow<T, BCO>::value + TileRow<T, BCI>::value) * (int)sizeof(T);
  constexpr int epi = (BCO / 2) * (BCI + 4) * 4;
  return 2 * stage > epi ? 2 * stage : epi;
}

template <typename T, int BCO, int BCI, int PRO>
int launch(const FrWgradArgs& a, hipStream_t st) {
  const int co_tiles = (a.Cout + BCO - 1) / BCO, ci_tiles = (a.SC + BCI - 1) / BCI;
  dim3 grid(co_tiles * ci_tiles * a.KH * a.KW, a.nsplit, 1);
  constexpr int LDS = lds_bytes<T, BCO, BCI>();
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<T, BCO, BCI, PRO>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    fr_attr_done(attr_done);
  }
  hipLaunchKernelGGL((conv_wgrad_kernel<T, BCO, BCI, PRO>), grid, dim3(NT), LDS, st, a);
  if (a.slab) {  // reproducible mode: every pixel slice wrote its own slab; add them in a fixed order into dw
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      fr_set_error(hipGetErrorString(e));
      return (int)e;
    }
    return fr_launch_reduce_slabs(a.slab, a.nsplit, (long long)a.Cout * a.KH * a.KW * a.SC, a.dw, st);
  }
  FR_LAUNCH_CHECK();
}

template <typename T, int PRO>
int dispatch_tile(const FrWgradArgs& a, hipStream_t st) {
  if (a.Cout >= 128) {
    if (a.SC >= 128) return launch<T, 128, 128, PRO>(a, st);
    return launch<T, 128, 64, PRO>(a, st);
  }
  if (a.SC >= 64) return launch<T, 64, 64, PRO>(a, st);
  return launch<T, 64, 32, PRO>(a, st);
}

template <typename T>
int dispatch(const FrWgradArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return dispatch_tile<T, FR_PRO_NONE>(a, st);
    case FR_PRO_BN: return dispatch_tile<T, FR_PRO_BN>(a, st);
    case FR_PRO_PRELU: return dispatch_tile<T, FR_PRO_PRELU>(a, st);
  }
  FR_UNSUPPORTED("fr_conv_wgrad: unknown prologue");
}

}  // namespace

extern "C" int fr_conv_wgrad(const FrWgradArgs* args, int dtype, void* stream) {
  const FrWgradArgs& a = *args;
  const int esz = dtype == FR_F32 ? 4 : 2;
  const int vec = 16 / esz;
  if (a.SC % vec || a.Cout % vec) FR_UNSUPPORTED("fr_conv_wgrad: channel counts must be multiples of 16 bytes");
  if ((a.ldg * esz) % 16 || (a.lda * esz) % 16) FR_UNSUPPORTED("fr_conv_wgrad: row strides must be 16-byte multiples");
  if ((long long)a.B * a.GH * a.GW >= (1ll << 24)) FR_UNSUPPORTED("fr_conv_wgrad: more than 2^24 pixels");
  if (a.nsplit < 1 || a.nsplit > 65535) FR_UNSUPPORTED("fr_conv_wgrad: bad nsplit");
  if (a.slab && ((long long)a.Cout * a.KH * a.KW * a.SC) % 4) FR_UNSUPPORTED("fr_conv_wgrad: slab mode needs Cout*taps*Cin % 4 == 0");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == FR_F32) return dispatch<float>(a, st);
  if (dtype == FR_BF16) return dispatch<bf16_t>(a, st);
  FR_UNSUPPORTED("fr_conv_wgrad: dtype must be FR_F32 or FR_BF16");
}
