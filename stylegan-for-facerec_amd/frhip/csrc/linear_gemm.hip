// frhip -- Linear(25088, 512) of the output layer as weight-streaming GEMMs on the fp32 master weight (bf16 MFMA, gfx950).
//
// output_layer = BatchNorm2d -> Dropout -> Flatten -> Linear(512*7*7, 512) -> BatchNorm1d (backbone/model_irse.py:143-148).
// At batch 256 the Linear is 6.6 GFLOP against a 51-MB fp32 weight: an HBM-bound op (the weight once = 8 us at 6.3 TB/s)
// that rounds 1-3 ran through the generic implicit-GEMM kernels behind a per-step re-layout of the weight: the master is
// [512][C*HW] in the reference's Flatten order (c-major) while the activations here are NHWC, so every step permuted +
// transposed + cast the weight (45 us), ran forward / data gradient on 392 small tiles (43 + 29 us), and wrote the weight
// gradient in the packed order to permute it back (11 + 65 + 23 us): 0.23 ms per step for 0.3 % of the FLOPs.
//
// Here the ACTIVATION takes the reference's order instead (fr_bn_dropout_cm writes a[b][c*HW + hw], 12.8 MB, through an LDS
// transpose), and the fp32 master is the GEMM operand as it lies, converted to bf16 in registers (the rounding the per-step
// cast did):
//   fr_linear_fwd   : slab[s][b][o] = sum_{k in slice s} a[b][k] W[o][k]  (+ bias in slice 0).  Both operands are
//                     k-contiguous, so every MFMA fragment is a plain global load (32 B of fp32 / 16 B of bf16 per lane): no
//                     LDS at all.  A workgroup owns 64 output features x 256 rows x one K slice (K / 32 / slices steps);
//                     its four waves split the rows, all read the same weight fragments (L1 / L2 hits), two steps of
//                     fragments in flight in registers.  224 workgroups at O = 512, 28 slices; the slices are added in a
//                     fixed order by fr_reduce_parts (reproducible; 14.7 MB of slabs).
//   fr_linear_dgrad : ga[b][k] = sum_o g[b][o] W[o][k].  The reduction runs over the ROW index of W: 32 x 128 tiles of W
//                     are staged in LDS as bf16 and read back transposed (ds_read_b64_tr_b16, the idiom of conv_wgrad.hip);
//                     the gradient rows g (256 KB, L2) are loaded as fragments with the matching k permutation.  One
//                     workgroup per (128 columns of K, 128 batch rows): 392 workgroups, two per CU, four steps in flight.
// The weight gradient dW[o][k] = sum_b g[b][o] a[b][k] is fr_conv_wgrad on the same a: it lands in the master's own layout.
#include "common.h"
#include "frhip_internal.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ s16x8 cvt8(const float4& lo, const float4& hi) {
  U128 u;
  u.x = pack2bf(lo.x, lo.y);
  u.y = pack2bf(lo.z, lo.w);
  u.z = pack2bf(hi.x, hi.y);
  u.w = pack2bf(hi.z, hi.w);
  return __builtin_bit_cast(s16x8, u);
}

// ------------------------------------------------------------------------------------------ forward
// grid (O / 64, slices, ceil(B / 256)); block 256.  Wave w: rows b0 + 64 w .. + 64 (4 tiles of 16), features o0 .. o0 + 64
// (4 tiles).  MFMA A = weight fragment (rows = features), B = activation fragment (columns = batch rows): a lane ends up
// with four consecutive features of one batch row -> 16-byte slab stores.
__global__ __launch_bounds__(NT) void linear_fwd_kernel(const bf16_t* __restrict__ a, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ slab, int B,
                                                        int O, int K, int ksteps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int o0 = blockIdx.x * 64, s = blockIdx.y;
  const int b0 = blockIdx.z * 256 + wave * 64;
  const int k0 = s * ksteps * 32;
  const float* wp[4];
  const bf16_t* ap[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wp[i] = W + (size_t)(o0 + i * 16 + fr) * K + k0 + fq * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int b = b0 + j * 16 + fr;
    b = b < B ? b : B - 1;  // rows past the batch repeat the last one; their results are not stored
    ap[j] = a + (size_t)b * K + k0 + fq * 8;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // two steps of fragments in flight: (4 x 32 B + 4 x 16 B) per lane and step
  float4 wl[2][4], wh[2][4];
  s16x8 af[2][4];
  auto issue = [&](int slot, int step) {
    const int st = step < ksteps ? step : ksteps - 1;  // clamp: the count of loads in flight stays static
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wl[slot][i] = *reinterpret_cast<const float4*>(wp[i] + st * 32);
      wh[slot][i] = *reinterpret_cast<const float4*>(wp[i] + st * 32 + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) af[slot][j] = *reinterpret_cast<const s16x8*>(ap[j] + st * 32);
  };
  auto consume = [&](int slot, int next_step) {
    // pin this slot's registers here and fence the scheduler: hipcc otherwise converts a slot's weights right behind its loads
    // and shuffles the two slots' requests, and since loads return in order the waits then cover BOTH slots (one memory
    // latency per trip instead of two steps of overlap)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("" : "+v"(wl[slot][i].x), "+v"(wl[slot][i].y), "+v"(wl[slot][i].z), "+v"(wl[slot][i].w));
      asm volatile("" : "+v"(wh[slot][i].x), "+v"(wh[slot][i].y), "+v"(wh[slot][i].z), "+v"(wh[slot][i].w));
    }
    s16x8 wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = cvt8(wl[slot][i], wh[slot][i]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[slot][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // refill the slot two steps ahead, BEHIND the MFMAs that read its activation fragments (requested in front of them, the
    // new values had to land in other registers and be copied over at the end of the trip -- behind a vmcnt(0))
    issue(slot, next_step);
    __builtin_amdgcn_sched_barrier(0);
  };
  issue(0, 0);
  issue(1, 1);
  // The steady-state loop is ONE basic block (two steps per trip, no branch inside): hipcc's counted vmcnt waits are exact
  // only on straight-line code -- with a per-step `if` the first version waited for EVERY load in flight at every step
  // (vmcnt(0)) and ran at one memory latency per step: 52 us for 78 MB.
  const int pairs = ksteps >> 1;
  for (int t = 0; t < pairs; ++t) {
    consume(0, 2 * t + 2);
    consume(1, 2 * t + 3);
  }
  if (ksteps & 1) consume(0, ksteps);
  // D[o][b]: lane holds features o0 + i*16 + fq*4 + r (r = 0..3) of batch row b0 + j*16 + fr
  float* out = slab + (size_t)s * B * O;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = b0 + j * 16 + fr;
    if (b >= B) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = o0 + i * 16 + fq * 4;
      f32x4 v = acc[i][j];
      if (s == 0 && bias) {
        v[0] += bias[o];
        v[1] += bias[o + 1];
        v[2] += bias[o + 2];
        v[3] += bias[o + 3];
      }
      *reinterpret_cast<f32x4*>(out + (size_t)b * O + o) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------ data gradient
// grid (K / 128, ceil(B / 128)); block 256, two workgroups per CU.  Wave w: batch rows b0 + 32 w .. + 32 (2 tiles), columns
// k0 .. k0 + 128 (8 tiles).  Per step of 32 reduction rows (o): W[o .. o + 32][k0 .. k0 + 128] fp32 -> bf16 LDS tile,
// transposing reads give the fragment "row = column of W, 8 reduction elements" = the MFMA A operand; g fragments (B operand)
// come from global with the same permutation of the 32 reduction elements (lane group q, element j -> 4q + j, 16 + 4q + (j - 4)).
// The op is pure memory latency (16 MFMAs per wave and step against a ~2-us round trip), so FOUR steps of W tiles and g
// fragments are requested together (64 KB of W per workgroup, ~100 KB per CU with two workgroups resident); the first
// version (one step ahead, a branch per step) ran 43 us for 64 MB.
constexpr int DG_LD = 128 + 16;  // LDS row length in bf16 elements (+32 B: conflict-free transposing reads)
constexpr int DG_DEPTH = 4;

__global__ __launch_bounds__(NT, 2) void linear_dgrad_kernel(const bf16_t* __restrict__ g, const float* __restrict__ W,
                                                             bf16_t* __restrict__ ga, int B, int O, int K) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[2][32 * DG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int k0 = blockIdx.x * 128;
  const int b0 = blockIdx.y * 128 + wave * 32;
  // staging: thread t moves row t / 8 of the 32 x 128 tile, 16 consecutive columns (four float4 loads)
  const int srow = tid >> 3, scol = (tid & 7) * 16;
  const float* wsrc = W + (size_t)srow * K + k0 + scol;
  const int nsteps = O / 32;
  const bf16_t* gp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int b = b0 + j * 16 + fr;
    b = b < B ? b : B - 1;
    gp[j] = g + (size_t)b * O + 4 * fq;
  }
  float4 ring[DG_DEPTH][4];
  uint2 gring[DG_DEPTH][2][2];
  auto issue = [&](int slot, int step) {
    const int st = step < nsteps ? step : nsteps - 1;  // clamp: a static number of loads in flight
    const float* p = wsrc + (size_t)st * 32 * K;
#pragma unroll
    for (int u = 0; u < 4; ++u) ring[slot][u] = *reinterpret_cast<const float4*>(p + u * 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      gring[slot][j][0] = *reinterpret_cast<const uint2*>(gp[j] + st * 32);
      gring[slot][j][1] = *reinterpret_cast<const uint2*>(gp[j] + st * 32 + 16);
    }
  };
  auto stage_store = [&](int slot, int buf) {
    // pin the tile's registers HERE: hipcc otherwise converts each tile to bf16 right behind its loads (fewer live registers)
    // and with that waits for every request before it issues the next one -- four round trips per trip instead of one
#pragma unroll
    for (int u = 0; u < 4; ++u)
      asm volatile("" : "+v"(ring[slot][u].x), "+v"(ring[slot][u].y), "+v"(ring[slot][u].z), "+v"(ring[slot][u].w));
    U128 lo, hi;
    lo.x = pack2bf(ring[slot][0].x, ring[slot][0].y);
    lo.y = pack2bf(ring[slot][0].z, ring[slot][0].w);
    lo.z = pack2bf(ring[slot][1].x, ring[slot][1].y);
    lo.w = pack2bf(ring[slot][1].z, ring[slot][1].w);
    hi.x = pack2bf(ring[slot][2].x, ring[slot][2].y);
    hi.y = pack2bf(ring[slot][2].z, ring[slot][2].w);
    hi.z = pack2bf(ring[slot][3].x, ring[slot][3].y);
    hi.w = pack2bf(ring[slot][3].z, ring[slot][3].w);
    bf16_t* d = &tile[buf][srow * DG_LD + scol];
    st16(d, lo);
    st16(d + 8, hi);
  };
  f32x4 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;
  // Trips of four steps, each self-contained: request the four tiles + g fragments, then per step commit the tile to one of
  // two LDS buffers, barrier, multiply.  (Keeping loads in flight ACROSS the loop's back edge made hipcc's wait-count pass
  // fall back to vmcnt(0) at three of the four steps; inside one straight-line trip its counted waits are exact -- 24, 16, 8,
  // 0 outstanding loads -- and the co-resident workgroup covers the bubble between trips.)
  auto do_step = [&](int slot, int step) {
    const int buf = step & 1;
    stage_store(slot, buf);
    __syncthreads();
    s16x8 gf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      U128 u;
      u.x = gring[slot][j][0].x;
      u.y = gring[slot][j][0].y;
      u.z = gring[slot][j][1].x;
      u.w = gring[slot][j][1].y;
      gf[j] = __builtin_bit_cast(s16x8, u);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16_t* p0 = &tile[buf][(4 * fq + (fr >> 2)) * DG_LD + i * 16 + 4 * (fr & 3)];
      const bf16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p0);
      const bf16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(p0 + 16 * DG_LD));
      const s16x4 a0 = __builtin_bit_cast(s16x4, x0), a1 = __builtin_bit_cast(s16x4, x1);
      const s16x8 wf = (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, gf[j], acc[i][j], 0, 0, 0);
    }
  };
  for (int step = 0; step < nsteps; step += DG_DEPTH) {  // nsteps % 4 == 0 (O % 128 == 0)
#pragma unroll
    for (int d = 0; d < DG_DEPTH; ++d) {
      issue(d, step + d);
      __builtin_amdgcn_sched_barrier(0);  // requests leave in step order: loads return in order, the waits count on it
    }
#pragma unroll
    for (int d = 0; d < DG_DEPTH; ++d) {
      do_step(d, step + d);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D[k][b]: lane holds columns k0 + i*16 + fq*4 + r of batch row b0 + j*16 + fr -> 8-byte bf16 stores
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int b = b0 + j * 16 + fr;
    if (b >= B) continue;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint2 o;
      o.x = pack2bf(acc[i][j][0], acc[i][j][1]);
      o.y = pack2bf(acc[i][j][2], acc[i][j][3]);
      *reinterpret_cast<uint2*>(ga + (size_t)b * K + k0 + i * 16 + fq * 4) = o;
    }
  }
}

}  // namespace

extern "C" int fr_linear_slices(int O, int K) {
  // K steps of 32 split over enough slices that (O / 64) x slices fills ~7/8 of the 256 CUs; every slice the same length
  if (O < 64 || O % 64 || K < 32 || K % 32) return 0;
  const int steps = K / 32, tiles = O / 64;
  int best = 1;
  for (int s = 1; s <= steps && s * tiles <= 256; ++s)
    if (steps % s == 0) best = s;
  return best;
}

extern "C" int fr_linear_fwd(const void* a, const float* W, const float* bias, float* slab, int B, int O, int K, int slices,
                             void* stream) {
  if (B < 1 || !a || !W || !slab) FR_UNSUPPORTED("fr_linear_fwd: a, W, slab and B >= 1 are required");
  if (O < 64 || O % 64 || K % 32 || slices < 1 || (K / 32) % slices)
    FR_UNSUPPORTED("fr_linear_fwd: O % 64 == 0, K % 32 == 0, slices must divide K / 32");
  const dim3 grid(O / 64, slices, (B + 255) / 256);
  hipLaunchKernelGGL(linear_fwd_kernel, grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)a, W, bias, slab, B, O, K,
                     K / 32 / slices);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_linear_dgrad(const void* g, const float* W, void* ga, int B, int O, int K, void* stream) {
  if (B < 1 || !g || !W || !ga) FR_UNSUPPORTED("fr_linear_dgrad: g, W, ga and B >= 1 are required");
  if (O % 128 || K % 128) FR_UNSUPPORTED("fr_linear_dgrad: O % 128 == 0 and K % 128 == 0");
  const dim3 grid(K / 128, (B + 127) / 128);
  hipLaunchKernelGGL(linear_dgrad_kernel, grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)g, W, (bf16_t*)ga, B, O, K);
  FR_LAUNCH_CHECK();
}
