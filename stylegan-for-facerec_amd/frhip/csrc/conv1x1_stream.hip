// frhip -- 1x1 convolutions as row-streaming GEMMs (bf16, gfx950): the convolved shortcut of a stage entry.
//
// shortcut_layer = Conv2d(in, depth, (1, 1), stride) -> BatchNorm2d of bottleneck_IR / bottleneck_IR_SE
// (backbone/model_irse.py:52-56) is a GEMM with a tiny reduction: K = 64 / 128 / 256 input channels, N = 2K outputs, one row per
// OUTPUT pixel (every other pixel of every other input row).  2.2-3.3 GFLOP per launch at batch 256 and 16-77 MB of traffic:
// HBM-bound, 4-15 us at the achievable rate -- on the tiled implicit GEMM these three launches took 24-51 us each (two or
// four 32-deep K steps per 128 x 128 tile: the tile prologue / epilogue is the kernel), their data gradients the same.
// Shaped like the stem GEMM instead (stem_gemm.hip): the WEIGHTS are the stationary MFMA A operand, held in registers -- the
// output channels are split over the NW waves of a workgroup so that a wave's share fits (N / NW channels x K) --, every wave
// streams the same 16-row tiles (a row fragment is one 16-byte load per lane; the rows of a tile are shared through the
// cache), the waves' slices meet in one LDS tile and leave as whole rows of 16-byte stores, BatchNorm statistics of the
// fp32 accumulators in the epilogue (one partial row per workgroup; a wave owns its channels: no exchange between waves).
// The data gradient of the same layer is the same GEMM with the transposed weight and dense rows.
#include "common.h"
#include "frhip_internal.h"

namespace {

struct C1Geo {
  int M;            // rows = B * RH * RW
  int RW, RHW;      // output grid (row -> (b, oh, ow))
  int SW, SHW;      // source grid
  int stride, lda, ldc;
  float inv_rw, inv_rhw;
};

// K: reduction (source channels), N: outputs, NW: waves per workgroup; a wave owns N / NW output channels
template <int K, int N, int NW, bool STATS>
__global__ __launch_bounds__(NW * 64) void conv1x1_stream_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp,
                                                                 bf16_t* __restrict__ out, float* __restrict__ part,
                                                                 const C1Geo g) {
  constexpr int KS = K / 32, NPW = N / NW, NTL = NPW / 16;  // K steps, channels and 16-channel tiles per wave
  constexpr int OSTR = N * 2 + 16;                           // workgroup tile [16 rows][N channels], padded rows
  constexpr int NT = 512 / K;                                // row tiles per trip: 16 KB of rows in flight per wave
  constexpr int NTH = NW * 64;
  static_assert(K % 32 == 0 && NPW % 16 == 0 && NTL * KS * 4 <= 160, "a wave's weights must fit its registers");
  // the waves' channel slices of a row tile meet in ONE LDS tile so that the rows leave whole (N * 2 bytes contiguous per row:
  // with per-wave tiles every wave wrote 32-128-byte pieces of a row at another time, and the kernel was no faster than the
  // tiled GEMM it replaces)
  __shared__ __attribute__((aligned(16))) char tiles[NT * 16 * OSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = wave * NPW;
  s16x8 wf[NTL][KS];
#pragma unroll
  for (int j = 0; j < NTL; ++j)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      wf[j][kk] = *reinterpret_cast<const s16x8*>(Wp + (size_t)(n0 + j * 16 + fr) * K + kk * 32 + fq * 8);
  float s0[STATS ? NTL : 1][4], s1[STATS ? NTL : 1][4];
  if (STATS) {
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s0[j][r] = s1[j][r] = 0.f;
  }
  const int ntiles = (g.M + 15) / 16;
  constexpr int CPR = N / 8;  // 16-byte chunks per output row
  const int trips = (ntiles + gridDim.x * NT - 1) / (gridDim.x * NT);  // same for every thread: barriers in the loop
  for (int it = 0; it < trips; ++it) {
    const int t0 = it * gridDim.x * NT + blockIdx.x;
    s16x8 af[NT][KS];
    bool ok[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = t0 + u * gridDim.x, row = t * 16 + fr;
      ok[u] = t < ntiles && row < g.M;
      size_t off = (size_t)row * g.lda;
      if (g.stride != 1) {  // row = (b, oh, ow) of the output grid -> source pixel (b, oh*stride, ow*stride)
        uint32_t b, rem, oh, ow;
        fast_divmod((uint32_t)(ok[u] ? row : 0), (uint32_t)g.RHW, g.inv_rhw, b, rem);
        fast_divmod(rem, (uint32_t)g.RW, g.inv_rw, oh, ow);
        off = ((size_t)b * g.SHW + (size_t)oh * g.stride * g.SW + (size_t)ow * g.stride) * g.lda;
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        af[u][kk] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (ok[u]) af[u][kk] = *reinterpret_cast<const s16x8*>(X + off + kk * 32 + fq * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      f32x4 acc[NTL];
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], af[u][kk], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        uint2 o;
        o.x = pack2bf(acc[j][0], acc[j][1]);
        o.y = pack2bf(acc[j][2], acc[j][3]);
        *reinterpret_cast<uint2*>(tiles + u * 16 * OSTR + fr * OSTR + (n0 + j * 16 + fq * 4) * 2) = o;
        if (STATS && ok[u]) {  // the sums of fr_conv_igemm's FR_EPI_STATS: of the fp32 accumulators
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s0[j][r] += acc[j][r];
            s1[j][r] = fmaf(acc[j][r], acc[j][r], s1[j][r]);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = t0 + u * gridDim.x;
      for (int c = tid; c < 16 * CPR; c += NTH) {
        const int r = c / CPR, c8 = c - r * CPR;
        const int orow = t * 16 + r;
        if (t < ntiles && orow < g.M) st16(out + (size_t)orow * g.ldc + c8 * 8, ld16(tiles + u * 16 * OSTR + r * OSTR + c8 * 16));
      }
    }
    __syncthreads();
  }
  if (STATS) {  // fold the 16 row lanes; every (wave, tile, fq, r) is a channel of its own
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s0[j][r], c = s1[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          c += __shfl_xor(c, o, 64);
        }
        if (fr == 0) {
          float* row = part + (size_t)blockIdx.x * 2 * N + n0 + j * 16 + fq * 4 + r;
          row[0] = a;
          row[N] = c;
        }
      }
  }
}

// a workgroup loads its weights once (up to 256 KB at 256 -> 512): at least two trips of 512 / K row tiles each, so that the
// load is paid for (one tile per workgroup at 7x7 -- 784 workgroups x 256 KB -- ran 55 us against 24 on the tiled GEMM)
int c1_grid(long long M, int K) {
  const long long tiles = (M + 15) / 16;
  const long long per = 2 * (512 / K);
  const long long wgs = (tiles + per - 1) / per;
  return (int)(wgs < 1024 ? wgs : 1024);
}

template <int K, int N, int NW>
int c1_launch(const FrConvArgs& a, const C1Geo& g, hipStream_t st) {
  const int grid = c1_grid(g.M, K);
  if (a.epi == FR_EPI_STATS)
    FR_LAUNCH_KERNEL((conv1x1_stream_kernel<K, N, NW, true>), dim3(grid), dim3(NW * 64), 0, st, (const bf16_t*)a.src,
                       (const bf16_t*)a.w, (bf16_t*)a.out, a.part, g);
  else
    FR_LAUNCH_KERNEL((conv1x1_stream_kernel<K, N, NW, false>), dim3(grid), dim3(NW * 64), 0, st, (const bf16_t*)a.src,
                       (const bf16_t*)a.w, (bf16_t*)a.out, a.part, g);
  FR_LAUNCH_CHECK();
}

bool c1_shape(int K, int N) {
  return (K == 64 && N == 128) || (K == 128 && N == 256) || (K == 256 && N == 512) || (K == 128 && N == 64) ||
         (K == 256 && N == 128) || (K == 512 && N == 256);
}

}  // namespace

// Partial rows ([2][N] each) a FR_EPI_STATS launch writes for B * RH * RW rows, 0 when the shape is not served.
extern "C" int fr_conv1x1_stream_parts(int B, int RH, int RW, int K, int N) {
  if (!c1_shape(K, N) || B < 1 || RH < 1 || RW < 1) return 0;
  const long long M = (long long)B * RH * RW;
  if (M >= (1ll << 24)) return 0;
  return c1_grid(M, K);
}

extern "C" int fr_conv1x1_stream(const FrConvArgs* args, void* stream) {
  const FrConvArgs& a = *args;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.KH != 1 || a.KW != 1 || a.pad != 0 || a.mode != 0 || a.pro != FR_PRO_NONE || a.out_f32 || a.splitk > 1 || a.bias ||
      (a.epi != FR_EPI_STORE && a.epi != FR_EPI_STATS))
    FR_UNSUPPORTED("fr_conv1x1_stream: 1x1, no padding / prologue / bias, epilogue STORE or STATS");
  if (a.w_frag) FR_UNSUPPORTED("fr_conv1x1_stream: fragment-order weights (w_frag) are read by the LDS-strip kernels only");
  if (!c1_shape(a.SC, a.N)) FR_UNSUPPORTED("fr_conv1x1_stream: shape not served (64->128, 128->256, 256->512 and their transposes)");
  if ((a.stride != 1 && a.stride != 2) || a.SH != a.RH * a.stride || a.SW != a.RW * a.stride)
    FR_UNSUPPORTED("fr_conv1x1_stream: stride 1 or 2 with SH = RH * stride, SW = RW * stride");
  if (a.lda % 8 || a.ldc % 8 || a.lda < a.SC || a.ldc < a.N) FR_UNSUPPORTED("fr_conv1x1_stream: strides must be 16-byte multiples");
  if (a.epi == FR_EPI_STATS && !a.part) FR_UNSUPPORTED("fr_conv1x1_stream: FR_EPI_STATS needs part");
  C1Geo g;
  const long long M = (long long)a.B * a.RH * a.RW;
  if (M < 1 || M >= (1ll << 24)) FR_UNSUPPORTED("fr_conv1x1_stream: fewer than 2^24 rows per launch");
  g.M = (int)M;
  g.RW = a.RW;
  g.RHW = a.RH * a.RW;
  g.SW = a.SW;
  g.SHW = a.SH * a.SW;
  g.stride = a.stride;
  g.lda = a.lda;
  g.ldc = a.ldc;
  g.inv_rw = 1.0f / (float)a.RW;
  g.inv_rhw = 1.0f / (float)(a.RH * a.RW);
  if (a.SC == 64 && a.N == 128) return c1_launch<64, 128, 4>(a, g, st);
  if (a.SC == 128 && a.N == 256) return c1_launch<128, 256, 4>(a, g, st);
  if (a.SC == 256 && a.N == 512) return c1_launch<256, 512, 8>(a, g, st);
  if (a.SC == 128 && a.N == 64) return c1_launch<128, 64, 4>(a, g, st);
  if (a.SC == 256 && a.N == 128) return c1_launch<256, 128, 4>(a, g, st);
  return c1_launch<512, 256, 8>(a, g, st);
}
