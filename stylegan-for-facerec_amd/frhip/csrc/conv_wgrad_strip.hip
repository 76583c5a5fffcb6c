// frhip -- weight gradient of the stride-1 3x3 convolutions with both operand strips resident in LDS (bf16).
//
//   dw[co][tap][ci] = sum_{pixels p}  g[p][co] * pro(x[p + tap][ci])
//
// A workgroup owns one 64(co) x 64(ci) x 9(taps) block of dW and a *group* of image strips.  Per strip it loads
// g[rows][64 co] and the haloed input x[(rows+2) x (W+2)][64 ci] into LDS once (prologue applied once), then runs
// K = pixels through MFMA with all nine taps served from the same LDS image: the tap only shifts the row address of
// the transposing LDS read (ds_read_b64_tr_b16), a compile-time offset.  Each g fragment is reused for 9 taps, so the
// kernel needs ~1.2 LDS reads per MFMA and no barrier inside a strip.  Accumulators (2 x 9 tiles per wave) stay in
// registers across all strips of the group; the group's partial dW goes to a slab and a second kernel adds the slabs
// (deterministic, no float atomics).  Workgroups of one group sit on one XCD (block-id remap), so the strips they
// share are fetched from HBM once and re-read from that XCD's L2.
//
// Reference arithmetic: autograd weight gradient of Conv2d(c, d, (3,3), (1,1), 1) in bottleneck_IR
// (backbone/model_irse.py:57-59) with BN apply (:57) / PReLU (:58) folded into the input load.
#include <stdlib.h>

#include "common.h"
#include "frhip_internal.h"
#include "slab_sum.h"

namespace {

constexpr int CT = 64;               // co and ci tile
constexpr int TSTR = CT * 2 + 32;    // LDS row stride (bytes) of both tiles: conflict-free transposed reads

// S2: the stride-2 convolution of a stage transition (model_irse.py:59).  W / ROWS then describe the LOW-resolution
// grid (= the gradient g); the input tile holds the four parity planes x[2i+ph][2j+pw] of the strip, each
// (ROWS+1) x (W+1) pixels with one halo row / column (plane row -1, column -1), and tap (kh, kw) reads plane
// (kh != 1, kw != 1) at row offset (kh > 0), column offset (kw > 0) -- still a compile-time address offset.
// RK ("row-aligned K"): the pixel (= K) axis of both tiles is laid out in rows of RW = 16 / 32 slots (W = 14 / 28; the
// surplus slots of the g tile hold zeros), so that a tap's row shift kh is a shift by whole 16-pixel fragment halves: the
// half-fragments of the input tile read for tap (0, kw) at K step s ARE the ones tap (kh, kw) needs at an earlier step, and
// a step reads 6 new input half-fragments instead of 18 (10 instead of 22 LDS reads per 18 MFMAs: MFMA- instead of
// LDS-bound: 8 waves x 22 x 512 B = 704 LDS clocks against 576 MFMA clocks per step before).
template <int W, int ROWS, int NIMG, int NW, bool S2 = false, bool RK = false>
struct WC {
  static constexpr int NTH = NW * 64;                 // 8 waves (2 co halves x 4 ci tiles) or 4 waves (4 ci tiles, all co)
  static constexpr int TCO = 4 / (NW / 4);            // co tiles (16 wide) per wave
  static constexpr int H = W;
  static constexpr int GW = S2 ? W + 1 : W + 2, GH = S2 ? ROWS + 1 : ROWS + 2;
  static constexpr int PLANE = GH * GW;               // S2: pixels of one parity plane
  static constexpr int MI = ROWS * W;                 // output pixels per image strip
  static constexpr int M = MI * NIMG;                 // pixels per fill
  static constexpr int RW = RK ? (W <= 14 ? 16 : 32) : GW;  // RK: K slots per image row
  static constexpr int NKS = RK ? ROWS * RW / 32 : (M + 31) / 32;
  static constexpr int PXP = NKS * 32;
  static constexpr int G_BYTES = PXP * TSTR;
  static constexpr int APIX = NIMG * GH * GW * (S2 ? 4 : 1);
  static constexpr int APOS = RK ? GH * RW + 16 : APIX;  // RK: tile positions (+ the tail a kw-shifted read of the last row touches)
  static constexpr int A_BYTES = APOS * TSTR;
  static constexpr int LDS = G_BYTES + A_BYTES;
  static constexpr int NS = H / ROWS;                 // strips per image
  static constexpr int GCH = M * 8, ACH = APIX * 8;   // 16-B chunks per fill
  static constexpr int NLD = (GCH + ACH + NTH - 1) / NTH;
  static constexpr bool PF = NW == 8;                 // 8 waves: register-prefetch the next strip; 4 waves: two
                                                      // resident workgroups overlap each other instead
  static constexpr int KUNR = (NLD > 8 || !PF) ? 1 : NKS;  // the wide strips keep more prefetch registers live
  static_assert(H % ROWS == 0, "strip rows must divide the image");
  static_assert(NIMG == 1 || ROWS == H, "several images per fill only for whole-image strips");
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(!RK || (!S2 && NIMG == 1 && W + 2 <= RW && (ROWS * RW) % 32 == 0), "row-aligned K: stride 1, one strip");
};

typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;

// pixel offset of tap (kh, kw) inside the input tile, relative to the tile position of output pixel (h, w)
template <class C, bool S2>
__device__ __forceinline__ constexpr int tap_pix(int tap) {
  const int kh = tap / 3, kw = tap % 3;
  if (!S2) return kh * C::GW + kw;
  const int plane = (kh != 1 ? 2 : 0) + (kw != 1 ? 1 : 0);
  return plane * C::PLANE + (kh > 0 ? 1 : 0) * C::GW + (kw > 0 ? 1 : 0);
}

__device__ __forceinline__ s16x4 tr_half(const char* p0) {
  return __builtin_bit_cast(s16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p0));
}

__device__ __forceinline__ s16x8 tr_frag(const char* p0, const char* p1) {
  const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p0);
  const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p1);
  const s16x4 ai = __builtin_bit_cast(s16x4, a), bi = __builtin_bit_cast(s16x4, b);
  return (s16x8){ai[0], ai[1], ai[2], ai[3], bi[0], bi[1], bi[2], bi[3]};
}

template <int W, int ROWS, int NIMG, int NW, int PRO, bool S2 = false, bool RK = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_wgrad_strip_kernel(const FrWgradArgs p) {
  using C = WC<W, ROWS, NIMG, NW, S2, RK>;
  constexpr int NTH = C::NTH;
  constexpr int TCO = C::TCO;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Gs = smem;
  char* As = smem + C::G_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wci = wave & 3, wco = wave >> 2;  // wco is 0 when NW == 4 (the wave then owns all four co tiles)

  // block -> (group, tile): consecutive logical ids (= all tiles of a group) share an XCD
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cit_n = p.SC / CT, tiles = (p.Cout / CT) * cit_n;
  const int group = bid / tiles, tile = bid - group * tiles;
  const int cot = tile / cit_n, cit = tile - cot * cit_n;
  const int total_fills = (p.B * C::NS + NIMG - 1) / NIMG;
  const int per = (total_fills + p.nsplit - 1) / p.nsplit;
  const int f_begin = group * per;
  int f_end = f_begin + per;
  if (f_end > total_fills) f_end = total_fills;

  const bf16_t* __restrict__ G = reinterpret_cast<const bf16_t*>(p.g);
  const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(p.src);

  // zero the padded pixel rows of the g tile once (rows >= M never get written again); RK: the surplus slots are spread
  // over both tiles (and must be finite in the input tile, where they meet the zeros of the g tile): clear everything
  if (RK) {
    for (int idx = tid; idx < (C::PXP + C::APOS) * 8; idx += NTH) st16(smem + (idx / 8) * TSTR + (idx & 7) * 16, zero16());
  } else {
    for (int idx = tid; idx < (C::PXP - C::M) * 8; idx += NTH)
      st16(Gs + (C::M + idx / 8) * TSTR + (idx & 7) * 16, zero16());
  }
  // tile position of g pixel c / of input-tile pixel a
  auto gpos = [](int c) -> int { return RK ? (c / W) * C::RW + c % W : c; };
  auto apos = [](int a) -> int { return RK ? (a / C::GW) * C::RW + a % C::GW : a; };

  const int ch = tid & 7;  // NTH % 8 == 0: a thread always handles the same 8-channel chunk
  float pa[8], pb[8];
  if (PRO != FR_PRO_NONE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pa[j] = p.pro_a[cit * CT + ch * 8 + j];
      pb[j] = PRO == FR_PRO_BN ? p.pro_b[cit * CT + ch * 8 + j] : 0.f;
    }
  }

  // input-tile pixel a of the fill starting at (img0, row0) -> source address / validity
  auto xsrc = [&](int a, int img0, int row0, const bf16_t*& sp) -> bool {
    constexpr int IPIX = C::GH * C::GW * (S2 ? 4 : 1);
    const int im = a / IPIX;
    int r = a - im * IPIX;
    const int b = img0 + im;
    if (S2) {
      const int plane = r / C::PLANE;
      r -= plane * C::PLANE;
      const int gh = r / C::GW, gw = r - gh * C::GW;
      const int i = row0 + gh - 1, j = gw - 1;  // plane coordinates; high-res pixel (2i + ph, 2j + pw)
      const int h = 2 * i + (plane >> 1), w = 2 * j + (plane & 1);
      sp = X + ((size_t)(b * 2 * C::H + h) * (2 * W) + w) * (size_t)p.lda + cit * CT + ch * 8;
      return b < p.B && i >= 0 && j >= 0;
    }
    const int gh = r / C::GW, gw = r - gh * C::GW;
    const int h = row0 + gh - 1, w = gw - 1;
    sp = X + ((size_t)(b * C::H + h) * W + w) * (size_t)p.lda + cit * CT + ch * 8;
    return b < p.B && (unsigned)h < (unsigned)C::H && (unsigned)w < (unsigned)W;
  };

  constexpr int NPF = C::PF ? C::NLD : 1;
  U128 ld[NPF];
  bool okv[NPF];
  auto issue = [&](int f) {
    // fill f covers images [img0, img0+NIMG) (whole images) or one strip of one image
    const int img0 = NIMG > 1 ? f * NIMG : f / C::NS;
    const int row0 = NIMG > 1 ? 0 : (f - img0 * C::NS) * ROWS;
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int idx = u * NTH + tid;
      const int c = idx >> 3;
      bool ok = false;
      const bf16_t* src = G;
      if (idx < C::GCH) {  // g tile row c
        const int im = c / C::MI, r = c - im * C::MI;
        const int b = img0 + im;
        ok = b < p.B;
        src = G + ((size_t)(b * C::H + row0) * W + r) * (size_t)p.ldg + cot * CT + ch * 8;
      } else if (idx < C::GCH + C::ACH) {
        const int a = c - C::M;
        const bf16_t* sp;
        ok = xsrc(a, img0, row0, sp);
        src = sp;
      }
      okv[u] = ok;
      ld[u] = ok ? ld16(src) : zero16();
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int idx = u * NTH + tid;
      const int c = idx >> 3;
      if (idx < C::GCH) {
        st16(Gs + gpos(c) * TSTR + ch * 16, ld[u]);
      } else if (idx < C::GCH + C::ACH) {
        U128 x = ld[u];
        if (PRO != FR_PRO_NONE && okv[u]) {
          float f[8];
          unpack16<bf16_t>(x, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (PRO == FR_PRO_BN) f[j] = fmaf(f[j], pa[j], pb[j]);
            else f[j] = f[j] > 0.f ? f[j] : f[j] * pa[j];
          }
          x = pack16<bf16_t>(f);
        }
        st16(As + apos(c - C::M) * TSTR + ch * 16, x);
      }
    }
  };

  static_assert(!RK || C::PF, "row-aligned K rides on the register-prefetch loader");
  auto load_now = [&](int f) {  // !PF: stream the strip through 8 registers at a time
    const int img0 = NIMG > 1 ? f * NIMG : f / C::NS;
    const int row0 = NIMG > 1 ? 0 : (f - img0 * C::NS) * ROWS;
    constexpr int UNR = 8;
    for (int base = 0; base < C::GCH + C::ACH; base += NTH * UNR) {
      U128 v[UNR];
      bool ok[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = base + u * NTH + tid;
        const int c = idx >> 3;
        ok[u] = false;
        const bf16_t* src = G;
        if (idx < C::GCH) {
          const int im = c / C::MI, r = c - im * C::MI;
          const int b = img0 + im;
          ok[u] = b < p.B;
          src = G + ((size_t)(b * C::H + row0) * W + r) * (size_t)p.ldg + cot * CT + ch * 8;
        } else if (idx < C::GCH + C::ACH) {
          const int a = c - C::M;
          const bf16_t* sp;
          ok[u] = xsrc(a, img0, row0, sp);
          src = sp;
        }
        v[u] = ok[u] ? ld16(src) : zero16();
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = base + u * NTH + tid;
        const int c = idx >> 3;
        if (idx < C::GCH) {
          st16(Gs + c * TSTR + ch * 16, v[u]);
        } else if (idx < C::GCH + C::ACH) {
          U128 x = v[u];
          if (PRO != FR_PRO_NONE && ok[u]) {
            float fl[8];
            unpack16<bf16_t>(x, fl);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (PRO == FR_PRO_BN) fl[j] = fmaf(fl[j], pa[j], pb[j]);
              else fl[j] = fl[j] > 0.f ? fl[j] : fl[j] * pa[j];
            }
            x = pack16<bf16_t>(fl);
          }
          st16(As + (c - C::M) * TSTR + ch * 16, x);
        }
      }
    }
  };

  f32x4 acc[TCO][9];
#pragma unroll
  for (int t = 0; t < TCO; ++t)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, lq = lane >> 4;
  const int colb = (4 * (li & 3)) * 2;  // byte offset of this lane's 4-channel group inside a 16-channel tile
  if (C::PF && f_begin < f_end) issue(f_begin);
  if (p.prev_n) {
    // deferred slab sum of the previous weight-gradient launch of this stream (FrWgradArgs.prev_*), added while the
    // first strip of this one is in flight
    const long long n4 = p.prev_n >> 2;
    const long long per_wg = (n4 + gridDim.x - 1) / gridDim.x;
    const long long e0 = (long long)blockIdx.x * per_wg;
    long long e1 = e0 + per_wg;
    if (e1 > n4) e1 = n4;
    if (e0 < e1) slab_sum_range<NTH>(p.prev_slab, p.prev_groups, n4, e0, e1, p.prev_dw, tid);
  }
  for (int f = f_begin; f < f_end; ++f) {
    __syncthreads();  // previous strip fully consumed
    if (C::PF) commit();
    else load_now(f);
    __syncthreads();
    if (C::PF && f + 1 < f_end) issue(f + 1);  // next strip's loads fly under this strip's MFMAs
    if constexpr (RK) {
      // half-fragment j = tile positions [16 j, 16 j + 16) shifted by kw; tap (kh, kw) of step ks pairs the halves
      // 2 ks + kh RW / 16 and the next one.  Everything below is unrolled: hg[][] lives in registers, indices are static.
      constexpr int R16 = C::RW / 16, NJ = 2 * C::NKS + 2 * R16;
      const char* ab = As + (4 * lq + (li >> 2)) * TSTR + (wci * 16) * 2 + colb;
      const char* gb = Gs + (4 * lq + (li >> 2)) * TSTR + (wco * 32) * 2 + colb;
      s16x4 hg[NJ][3];
#pragma unroll
      for (int j = 0; j < 2 * R16; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) hg[j][kw] = tr_half(ab + (16 * j + kw) * TSTR);
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
#pragma unroll
        for (int j = 2 * ks + 2 * R16; j < 2 * ks + 2 * R16 + 2; ++j)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) hg[j][kw] = tr_half(ab + (16 * j + kw) * TSTR);
        s16x8 gf[TCO];
#pragma unroll
        for (int t = 0; t < TCO; ++t) gf[t] = tr_frag(gb + (32 * ks) * TSTR + t * 32, gb + (32 * ks + 16) * TSTR + t * 32);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int j0 = 2 * ks + (tap / 3) * R16;
          const s16x4 lo = hg[j0][tap % 3], hi = hg[j0 + 1][tap % 3];
          const s16x8 af = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int t = 0; t < TCO; ++t)
            acc[t][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[t], af, acc[t][tap], 0, 0, 0);
        }
      }
      continue;
    }
    // fully unrolled where registers allow (reads of step k+1 are then scheduled under the MFMAs of step k)
#pragma unroll C::KUNR
    for (int ks = 0; ks < C::NKS; ++ks) {
      const int m0 = ks * 32 + 4 * lq + (li >> 2), m1 = m0 + 16;
      const char* g0 = Gs + m0 * TSTR + (wco * 32) * 2 + colb;
      const char* g1 = Gs + m1 * TSTR + (wco * 32) * 2 + colb;
      s16x8 gf[TCO];
#pragma unroll
      for (int t = 0; t < TCO; ++t) gf[t] = tr_frag(g0 + t * 32, g1 + t * 32);
      const char* a01[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int m = e ? m1 : m0;
        m = m < C::M ? m : C::M - 1;
        const int im = m / C::MI, r = m - im * C::MI;
        const int h = r / W, w = r - h * W;
        a01[e] = As + ((im * (S2 ? 4 : 1) * C::GH + h) * C::GW + w) * TSTR + (wci * 16) * 2 + colb;
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int off = tap_pix<C, S2>(tap) * TSTR;
        const s16x8 af = tr_frag(a01[0] + off, a01[1] + off);
#pragma unroll
        for (int t = 0; t < TCO; ++t)
          acc[t][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[t], af, acc[t][tap], 0, 0, 0);
      }
    }
  }

  // slab[group][co][tap][ci]
  float* __restrict__ slab = p.slab + (size_t)group * (size_t)p.Cout * 9 * (size_t)p.SC;
#pragma unroll
  for (int t = 0; t < TCO; ++t)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * CT + wco * 32 + t * 16 + lq * 4 + r;
        const int ci = cit * CT + wci * 16 + li;
        slab[((size_t)co * 9 + tap) * (size_t)p.SC + ci] = acc[t][tap][r];
      }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, int groups, long long n4,
                                                           float* __restrict__ out) {
  // n4 = elements / 4; every workgroup takes a contiguous share (slab_sum.h fixes the order of the additions)
  const long long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long long e0 = (long long)blockIdx.x * per;
  long long e1 = e0 + per;
  if (e1 > n4) e1 = n4;
  if (e0 < e1) slab_sum_range<256>(slab, groups, n4, e0, e1, out, threadIdx.x);
}

}  // namespace

int fr_launch_reduce_slabs(const float* slab, int groups, long long n, float* out, hipStream_t st) {
  if (groups > 256) FR_UNSUPPORTED("weight-gradient slabs: at most 256 groups");
  const long long n4 = n / 4;
  const int lpe = groups <= 16 ? 1 : (groups <= 32 ? 2 : (groups <= 64 ? 4 : (groups <= 128 ? 8 : 16)));
  long long g = (n4 * lpe + 255) / 256;  // one pass per workgroup where the grid allows
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((int)g), dim3(256), 0, st, slab, groups, n4, out);
  FR_LAUNCH_CHECK();
}

namespace {

template <int W, int ROWS, int NIMG, int NW, int PRO, bool S2 = false, bool RK = false>
int launch(const FrWgradArgs& a, hipStream_t st) {
  using C = WC<W, ROWS, NIMG, NW, S2, RK>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_strip_kernel<W, ROWS, NIMG, NW, PRO, S2, RK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    fr_attr_done(attr_done);
  }
  const int tiles = (a.Cout / CT) * (a.SC / CT);
  hipLaunchKernelGGL((conv_wgrad_strip_kernel<W, ROWS, NIMG, NW, PRO, S2, RK>), dim3(tiles * a.nsplit), dim3(C::NTH), C::LDS,
                     st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fr_set_error(hipGetErrorString(e));
    return (int)e;
  }
  if (a.defer) return 0;  // the caller sums the slabs (prev_* of a later launch, or fr_reduce_slabs)
  return fr_launch_reduce_slabs(a.slab, a.nsplit, (long long)a.Cout * 9 * a.SC, a.dw, st);
}

template <int W, int ROWS, int NIMG, int NW, bool S2 = false, bool RK = false>
int by_pro(const FrWgradArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return launch<W, ROWS, NIMG, NW, FR_PRO_NONE, S2, RK>(a, st);
    case FR_PRO_BN: return launch<W, ROWS, NIMG, NW, FR_PRO_BN, S2, RK>(a, st);
    case FR_PRO_PRELU: return launch<W, ROWS, NIMG, NW, FR_PRO_PRELU, S2, RK>(a, st);
  }
  FR_UNSUPPORTED("fr_conv_wgrad_strip: unknown prologue");
}

// the 28x28 / 14x14 instances use the row-aligned K layout (round 2; its A/B switch left in round 5)
bool row_k() { return true; }

}  // namespace

// 1 when (W) is in the strip table and the channel counts are multiples of 64
extern "C" int fr_conv_wgrad_strip_supported(int Cout, int Cin, int W) {
  if (Cout % CT || Cin % CT) return 0;
  return W == 112 || W == 56 || W == 28 || W == 14 || W == 7;
}

// every shape fr_conv_wgrad_strip serves honours defer / prev_* (FRHIP_WGRAD_DEFER=0 turns the answer off: A/B switch)
extern "C" int fr_conv_wgrad_strip_defers(const FrWgradArgs* args) {
  static const int* on = fr_option_slot("FRHIP_WGRAD_DEFER", 1);
  const FrWgradArgs& a = *args;
  if (!*on || a.ldg % 8 || a.lda % 8 || !a.slab || a.nsplit < 1 || a.nsplit > 256 || a.KH != 3 || a.KW != 3 || a.pad != 1 ||
      a.Cout % CT || a.SC % CT)
    return 0;
  if (a.stride == 2) return a.GH == a.GW && a.SH == 2 * a.GH && a.SW == 2 * a.GW && (a.GW == 56 || a.GW == 28 || a.GW == 14 || a.GW == 7);
  return a.stride == 1 && a.GH == a.SH && a.GW == a.SW && a.SH == a.SW && fr_conv_wgrad_strip_supported(a.Cout, a.SC, a.SW);
}

extern "C" int fr_reduce_slabs(const float* slab, int groups, long long n, float* out, void* stream) {
  if (!slab || !out || groups < 1 || n < 4 || n % 4) FR_UNSUPPORTED("fr_reduce_slabs: bad arguments");
  return fr_launch_reduce_slabs(slab, groups, n, out, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int fr_conv_wgrad_strip(const FrWgradArgs* args, void* stream) {
  const FrWgradArgs& a = *args;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.ldg % 8 || a.lda % 8 || !a.slab || a.nsplit < 1) FR_UNSUPPORTED("fr_conv_wgrad_strip: bad strides / slab");
  if (a.prev_n && (!a.prev_slab || !a.prev_dw || a.prev_groups < 1 || a.prev_groups > 256 || a.prev_n % 4 ||
                   a.prev_slab == a.slab))
    FR_UNSUPPORTED("fr_conv_wgrad_strip: bad prev_* (deferred slab sum)");
  if (a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.GH == a.GW && a.SH == 2 * a.GH && a.SW == 2 * a.GW &&
      a.Cout % CT == 0 && a.SC % CT == 0) {
    if (fr_wgrad_s2roll_serves(a)) return fr_wgrad_s2roll_launch(a, st);
    // stride 2: the input tile holds the four parity planes of the strip (see WC)
    switch (a.GW) {
      case 56: return by_pro<56, 2, 1, 8, true>(a, st);
      case 28: return by_pro<28, 4, 1, 8, true>(a, st);
      case 14: return by_pro<14, 7, 1, 8, true>(a, st);
      case 7: return by_pro<7, 7, 2, 8, true>(a, st);
    }
    FR_UNSUPPORTED("fr_conv_wgrad_strip: stride-2 width not in the strip table");
  }
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.GH != a.SH || a.GW != a.SW || a.SH != a.SW ||
      !fr_conv_wgrad_strip_supported(a.Cout, a.SC, a.SW))
    FR_UNSUPPORTED("fr_conv_wgrad_strip: square 3x3 bf16 convolutions, stride 1 or 2, 64-multiple channels");
  if (fr_wgrad_roll_serves(a)) return fr_wgrad_roll_launch(a, st);
  // 4-wave workgroups (two resident per CU, NW = 4) were measured at about half the throughput of the 8-wave form
  // with register prefetch (tools/kbench.py, B = 256) and are not instantiated.
  switch (a.SW) {
    case 112: return by_pro<112, 2, 1, 8>(a, st);
    case 56: return by_pro<56, 4, 1, 8>(a, st);
    case 28: return row_k() ? by_pro<28, 7, 1, 8, false, true>(a, st) : by_pro<28, 7, 1, 8>(a, st);
    case 14: return row_k() ? by_pro<14, 14, 1, 8, false, true>(a, st) : by_pro<14, 14, 1, 8>(a, st);
    case 7: return by_pro<7, 7, 4, 8>(a, st);
  }
  FR_UNSUPPORTED("fr_conv_wgrad_strip: width not in the strip table");
}
