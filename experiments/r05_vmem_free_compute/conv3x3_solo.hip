// frhip -- stride-1 3x3 convolution, whole image per workgroup, ONE wave per SIMD (bf16, gfx950; round 5).
//
// The 256 -> 256 @14x14 layers are half of the IR-50 FLOPs.  conv3x3_strip.hip runs them as 8 waves x
// (13 x 2) accumulator tiles: two waves share every SIMD, each streams its own weight fragments and re-reads every pixel
// fragment from LDS, and the K loop ends with a barrier + an LDS transpose of the output tile (stamps, round 3: loop at
// ~65 % of the MFMA rate, 7.4 us in which the older wave of a SIMD waits for its partner, 6-10 us of epilogue).  Here a
// workgroup is FOUR waves, one per SIMD, each with the whole 512-entry register file of its SIMD:
//   * wave w owns output channels [64 w, 64 w + 64) of ALL 196 pixels: 13 x 4 accumulator tiles (208 registers), so a
//     pixel fragment read from LDS feeds 4 MFMAs instead of 2 (half the LDS reads per MFMA, 25 % of the LDS peak) and
//     there is no SIMD partner to arbitrate with: MFMAs issue back to back, one ds_read_b128 per four of them;
//   * latency is covered by depth, not by a partner: pixel fragments through a 13-deep register ring (one whole tap
//     ahead), weight fragments 3 taps (~2500 cycles) ahead;
//   * the epilogue works on the accumulators where they are: a lane holds 4 consecutive channels of one pixel (weights
//     are the MFMA A operand), so the fused cells (BN statistics, PReLU backward, BN-backward sums, cross moment) read
//     their aux operand with 8-byte loads in the accumulator layout and the result leaves with 8-byte stores (a wave
//     writes whole 128-byte lines with 4 consecutive store instructions).  No output tile in LDS, no barrier after the K
//     loop, per-channel sums never cross a wave (a wave owns its channels): shuffles only.
// Same arithmetic, K order and summation order as conv3x3_strip_kernel<256,256,14,14,8,8>: outputs and partial rows are
// bit-identical to it (tests/test_gpu_kernels.py::test_conv3x3_solo_is_bit_identical_to_the_strip_kernel).
//
// Reference arithmetic: Conv2d 3x3 s1 of bottleneck_IR (backbone/model_irse.py:57-59) with BN apply (:57) or PReLU (:58)
// on the input, and its autograd data gradient (mode 1: taps mirrored, weights given as [Cin][tap][Cout]).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "frhip_internal.h"

#ifdef FRHIP_STAMPS
__device__ unsigned long long* fr_stamp_buf_solo = nullptr;
extern "C" int fr_debug_set_stamp_buffer_solo(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fr_stamp_buf_solo), &dev_ptr, sizeof(dev_ptr));
}
#define FR_STAMP(k)                                                                                             \
  do {                                                                                                          \
    if (tid == 0 && fr_stamp_buf_solo) fr_stamp_buf_solo[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define FR_STAMP_CLK(k)                                                                                         \
  do {                                                                                                          \
    if (tid == 0 && fr_stamp_buf_solo) fr_stamp_buf_solo[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define FR_STAMP(k)
#define FR_STAMP_CLK(k)
#endif

#ifndef FRHIP_SOLO_D
#define FRHIP_SOLO_D 13  // pixel-fragment ring depth (must divide 9 * 13)
#endif
#ifndef FRHIP_SOLO_AUXT
#define FRHIP_SOLO_AUXT 2  // aux requests per tap
#endif
#ifndef FRHIP_SOLO_WSPREAD
#define FRHIP_SOLO_WSPREAD 1  // weight requests of a tap spread over the MFMAs of the tap two ahead (0: in one batch, three taps ahead)
#endif
#ifndef FRHIP_SOLO_ABL
#define FRHIP_SOLO_ABL 0  // timing ablations (diagnostic builds only; results are wrong): 1 no lgkmcnt waits, 2 no LDS reads, 4 no weight stream, 8 no vmcnt waits
#endif
#ifndef FRHIP_SOLO_AUX_EARLY
#define FRHIP_SOLO_AUX_EARLY 0  // 1: aux operand of the fused epilogues requested under the last channel chunks of the K loop -- WRONG RESULTS as built by hipcc 7.2 (it copies the destination registers before the loads land); kept for the timing it gave
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<0, N>(f);
}
// the instructions of the K loop (see the comment at the loop): asm volatile, operands by constraint
template <int OFF>
__device__ __forceinline__ void ds_read128(i32x4& dst, int addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void gload128(i32x4& dst, int voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
template <int OFF>
__device__ __forceinline__ void gload64(uint2& dst, int voff, const void* sbase) {
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(OFF));
}
__device__ __forceinline__ void mfma_a(f32x4& acc, const i32x4& w, const i32x4& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N));
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
}
__device__ __forceinline__ void pin_a(f32x4& acc) { asm volatile("" : "+a"(acc)); }
__device__ __forceinline__ void pin_v(uint2& v) { asm volatile("" : "+v"(v)); }

// aux requests issued behind the weight request of peeled tap g
template <int NAUXL, int AUXT, int AUX_TAPS, int T0>
constexpr int aux_at(int g) {
  const int k = g - T0;
  if (g < 0 || k < 0 || k >= AUX_TAPS) return 0;
  const int left = NAUXL - k * AUXT;
  return left < AUXT ? left : AUXT;
}

template <int CIN, int COUT, int W>
struct SO {
  static constexpr int NTH = 256;
  static constexpr int H = W, GW = W + 2, GH = W + 2;
  static constexpr int CH = CIN / 8;                 // 16-byte chunks per pixel
  static constexpr int PSTR = CIN * 2 + 32;          // padded pixel stride (conflict-free ds_read_b128, tools/lds_probe.hip)
  static constexpr int RSTR = GW * PSTR + 192;       // slot index keeps counting across an image-row wrap
  static constexpr int LDS = GH * RSTR + 128;        // + slack for the ring's reads past the last chunk
  static constexpr int M = W * W;
  static constexpr int TM = (M + 15) / 16;
  static constexpr int TN = COUT / 64;               // 16-channel tiles per wave (4 waves)
  static constexpr int NVAL = M * CH;                // chunks of the image proper
  static constexpr int PER = (NVAL + NTH - 1) / NTH;
  static constexpr int NHALO = (GH * GW - M) * CH;
  static_assert(NTH % CH == 0 && CIN % 32 == 0 && COUT % 64 == 0, "bad channel counts");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int CIN, int COUT, int W, int NSPL, int PRO, bool AUXK>
__global__ __launch_bounds__(256, 1) void conv3x3_solo_kernel(const FrConvArgs p, const int xcd) {
  using C = SO<CIN, COUT, W>;
  constexpr int NTH = C::NTH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool RES = PRO == FR_PRO_RESBN || PRO == FR_PRO_RESBN_SE;
  const bf16_t* __restrict__ wgt = reinterpret_cast<const bf16_t*>(p.w);

  const int lb = xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int nh = NSPL > 1 ? lb % NSPL : 0;
  const int ncol0 = nh * COUT;
  const int b = NSPL > 1 ? lb / NSPL : lb;  // image
  const size_t rowbase = (size_t)b * C::M;

  FR_STAMP(0);
  // ---------------------------------------------------------------- image -> LDS (prologue applied once)
  {
    const bf16_t* __restrict__ img = reinterpret_cast<const bf16_t*>(p.src) + rowbase * (size_t)p.lda;
    const bf16_t* __restrict__ img2 = RES ? reinterpret_cast<const bf16_t*>(p.src2) + rowbase * (size_t)p.lda : nullptr;
    bf16_t* __restrict__ pro_out = RES && p.pro_out ? reinterpret_cast<bf16_t*>(p.pro_out) + rowbase * (size_t)p.lda : nullptr;
    const int ch = tid % C::CH;
    float pa[8], pb[8], pc[RES ? 8 : 1], pd[RES ? 8 : 1], pg[PRO == FR_PRO_RESBN_SE ? 8 : 1];
    if (PRO != FR_PRO_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pa[j] = p.pro_a[ch * 8 + j];
        pb[j] = (PRO == FR_PRO_BN || RES) ? p.pro_b[ch * 8 + j] : 0.f;
        if (RES) pc[j] = p.pro_c[ch * 8 + j];
        if (RES) pd[j] = p.pro_d[ch * 8 + j];
        if (PRO == FR_PRO_RESBN_SE) pg[j] = p.pro_g[(size_t)b * p.SC + ch * 8 + j];
      }
    }
    // halo: zero rows / columns around the image (never touched again)
    for (int q = tid; q < C::NHALO; q += NTH) {
      const int hp = q / C::CH, c = q - hp * C::CH;
      int gh, gw;
      if (hp < C::GW) {
        gh = 0, gw = hp;
      } else if (hp < 2 * C::GW) {
        gh = C::GH - 1, gw = hp - C::GW;
      } else {
        const int r = hp - 2 * C::GW;
        gh = 1 + (r >> 1), gw = (r & 1) ? C::GW - 1 : 0;
      }
      st16(smem + gh * C::RSTR + gw * C::PSTR + c * 16, zero16());
    }
    // the image proper: every load of a thread in flight at once (two batches with two sources)
    constexpr int NB = RES ? 2 : 1;
    constexpr int UNR = (C::PER + NB - 1) / NB;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
      U128 v[UNR], v2[RES ? UNR : 1];
      int off[UNR], lo[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int idx = (bt * UNR + u) * NTH + tid;
        idx = idx < C::NVAL ? idx : C::NVAL - C::CH + ch;  // clamp (the duplicate rewrites the same bytes)
        const int px = idx / C::CH;
        const int h = px / W, w = px - h * W;
        off[u] = px * p.lda + ch * 8;
        lo[u] = (h + 1) * C::RSTR + (w + 1) * C::PSTR + ch * 16;
        v[u] = ld16(img + off[u]);
        if (RES) v2[u] = ld16(img2 + off[u]);
      }
      __builtin_amdgcn_sched_barrier(0);  // the whole batch in flight before the first prologue instruction
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if ((bt * UNR + u) * NTH >= C::NVAL) continue;  // compile-time: rounds past the image
        U128 x = v[u];
        if (PRO != FR_PRO_NONE) {
          float f[8];
          unpack16<bf16_t>(x, f);
          if (RES) {
            float f2[8];
            unpack16<bf16_t>(v2[u], f2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // the arithmetic of fr_bn_apply (res_kind 1 [, se])
              f[j] = fmaf(f[j], pa[j], pb[j]);
              f[j] = PRO == FR_PRO_RESBN_SE ? fmaf(f[j], pg[j], f2[j]) : f[j] + f2[j];
            }
            x = pack16<bf16_t>(f);
            if (nh == 0 && pro_out) st16(pro_out + off[u], x);
            unpack16<bf16_t>(x, f);  // BN1 normalises what the residual stream holds
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(f[j], pc[j], pd[j]);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (PRO == FR_PRO_BN) f[j] = fmaf(f[j], pa[j], pb[j]);
              else f[j] = f[j] > 0.f ? f[j] : f[j] * pa[j];
            }
          }
          x = pack16<bf16_t>(f);
        }
        st16(smem + lo[u], x);
      }
    }
  }
  FR_STAMP(1);
  __syncthreads();
  FR_STAMP(2);
  FR_STAMP_CLK(5);

  // ---------------------------------------------------------------- main loop: 9 taps x CIN/32, no barriers
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = wave * C::TN * 16;
  const int flip = p.mode;
  const bf16_t* wrow[C::TN];
#pragma unroll
  for (int j = 0; j < C::TN; ++j) wrow[j] = wgt + (size_t)(ncol0 + n0 + j * 16 + fr) * 9 * CIN + fq * 8;
  f32x4 acc[C::TM][C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int abase[C::TM];
#pragma unroll
  for (int i = 0; i < C::TM; ++i) {
    int m = i * 16 + fr;
    m = m < C::M ? m : 0;
    const int h = m / W, w = m - h * W;
    abase[i] = h * C::RSTR + w * C::PSTR + fq * 16;
  }
  // The K loop is written as a straight line of `asm volatile` statements (program order = source order: volatile asms are
  // never reordered against each other), operands allocated by the compiler through constraints.  Why not builtins: with 512
  // registers hipcc splits the file 256 VGPR / 256 AGPR, rotated the accumulators through the MFMAs' separate dst / src C
  // and then repaired the rotation at the loop edge with 224 v_accvgpr moves per channel chunk, each a read of a fresh MFMA
  // result.  Here the accumulators are pinned to AGPRs ("+a": dst == src C), every fragment to VGPRs, and the waits are
  // counted by hand (the compiler's waitcnt pass cannot see into asm): NO compiler-generated memory operation may sit inside
  // the loop (a spill would break the vmcnt arithmetic -- check `scratch` = 0 in the listing).
  //
  // aux operand of the fused epilogues, in the accumulator layout: lane (fr, fq) of tile (i, j) holds 4 consecutive channels
  // of pixel i*16 + fr.  Requested under the LAST channel chunks of the K loop, AUXT loads behind every weight request
  // (vector-memory results return in order: a request in front of a weight fragment delays it by its own latency, so they
  // are spread, never batched in front of the stream).
  constexpr int NAUXL = C::TM * C::TN;
  uint2 av[AUXK ? C::TM : 1][AUXK ? C::TN : 1];
  const int aux_v = AUXK ? (fr * p.ldaux + fq * 4) * 2 : 0;  // byte offset of this lane inside a 16-pixel tile
  int aux_vl = aux_v;                                         // ... of the last tile: pixels past the image read the last one
  if (AUXK && (C::TM - 1) * 16 + fr >= C::M) aux_vl = ((C::M - 1 - (C::TM - 1) * 16) * p.ldaux + fq * 4) * 2;
  const bf16_t* __restrict__ auxb = AUXK ? reinterpret_cast<const bf16_t*>(p.aux) + rowbase * (size_t)p.ldaux + ncol0 + n0 : nullptr;
  auto aux_load = [&](auto ktag) {  // k in [0, NAUXL)
    if constexpr (AUXK) {
      constexpr int k = decltype(ktag)::value;
      constexpr int i = k / C::TN, j = k - i * C::TN;
      const bf16_t* sb = auxb + (size_t)(i * 16) * p.ldaux;  // uniform
      gload64<j * 32>(av[i][j], i == C::TM - 1 ? aux_vl : aux_v, sb);
    }
  };

  i32x4 bq[3][C::TN];
  int woff[C::TN];  // byte offset of this lane's weight row inside the tap slice
#pragma unroll
  for (int j = 0; j < C::TN; ++j) woff[j] = ((ncol0 + n0 + j * 16 + fr) * 9 * CIN + fq * 8) * 2;
  auto load_b = [&](auto slottag, int c0, int tap) {
    constexpr int slot = decltype(slottag)::value;
    const int wt = flip ? 8 - tap : tap;
    const bf16_t* sb = wgt + (wt * CIN + c0);  // uniform
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
      gload128(bq[slot][j], woff[j], sb);
  };
  auto load_b1 = [&](auto slottag, auto jtag, int c0, int tap) {  // one fragment
    constexpr int slot = decltype(slottag)::value;
    constexpr int j = decltype(jtag)::value;
    const int wt = flip ? 8 - tap : tap;
    gload128(bq[slot][j], woff[j], wgt + (wt * CIN + c0));
  };
  constexpr bool SPREAD = FRHIP_SOLO_WSPREAD != 0;
  constexpr int WAHEAD = SPREAD ? 2 : 3;           // taps between a weight request and its use
  constexpr int WSTEP = C::TM / C::TN;             // spread: one request every WSTEP steps
  constexpr int NSTEP = 9 * C::TM;
  constexpr int D = FRHIP_SOLO_D;
  static_assert(NSTEP % D == 0 && D <= 15, "ring depth must divide the steps of a channel chunk (and fit lgkmcnt)");
  i32x4 ring[D];
  auto a_read = [&](auto sttag) {  // ring[step % D] <- fragment of step; step in [0, 2*NSTEP): second half = next 32 input channels
    constexpr int step = decltype(sttag)::value;
    constexpr int cadd = step >= NSTEP ? 64 : 0;
    constexpr int st = step >= NSTEP ? step - NSTEP : step;
    constexpr int tap = st / C::TM, i = st - tap * C::TM;
    ds_read128<(tap / 3) * C::RSTR + (tap % 3) * C::PSTR + cadd>(ring[step % D], abase[i]);
  };
  load_b(std::integral_constant<int, 0>{}, 0, 0);
  load_b(std::integral_constant<int, 1>{}, 0, 1);
  if constexpr (!SPREAD) load_b(std::integral_constant<int, 2>{}, 0, 2);
  static_for<D>([&](auto d) { a_read(d); });
  // aux requests: AUXT per tap, at compile-time positions inside the last NQ (peeled) channel chunks.  The last AUX_FREE taps
  // carry none.
  constexpr int AUXT = FRHIP_SOLO_AUXT;
  constexpr int AUX_TAPS = (NAUXL + AUXT - 1) / AUXT;
  constexpr int AUX_FREE = 6;
  constexpr int NQ = (AUXK && FRHIP_SOLO_AUX_EARLY) ? (AUX_TAPS + AUX_FREE + 8) / 9 : 0;
  constexpr int T0 = NQ * 9 - AUX_FREE - AUX_TAPS;  // first tap (of the peeled range) that carries aux requests
  static_assert(NQ * 32 <= CIN, "not enough channel chunks to spread the aux requests");
  auto chunk = [&](const int c0, auto qtag) {
    constexpr int Q = decltype(qtag)::value;  // -1: no aux requests in this chunk
    static_for<NSTEP>([&](auto sttag) {
      constexpr int st = decltype(sttag)::value;
      constexpr int tap = st / C::TM, i = st - tap * C::TM;
      constexpr int slot = tap % 3;
      if constexpr (i == 0) {
        // weights of this tap: requested three taps ago; younger = two taps of weights + the aux requests behind the last three
        constexpr int g = Q < 0 ? -100 : Q * 9 + tap;
        constexpr int younger = (WAHEAD - 1) * C::TN + (SPREAD ? 0 : aux_at<NAUXL, AUXT, AUX_TAPS, T0>(g - 3)) +
                                aux_at<NAUXL, AUXT, AUX_TAPS, T0>(g - 2) + aux_at<NAUXL, AUXT, AUX_TAPS, T0>(g - 1);
        if constexpr (!(FRHIP_SOLO_ABL & 12)) wait_vm<younger>();
      }
      if constexpr (!(FRHIP_SOLO_ABL & 3)) wait_lgkm<D - 1>();
#pragma unroll
      for (int j = 0; j < C::TN; ++j)
        mfma_a(acc[i][j], bq[slot][j], ring[st % D]);  // = (W X^T) tile
      if constexpr (!(FRHIP_SOLO_ABL & 2)) a_read(std::integral_constant<int, st + D>{});  // past the last channel chunk this reads (never used) bytes inside LDS
      if constexpr (SPREAD && i % WSTEP == 1 && i / WSTEP < C::TN && !(FRHIP_SOLO_ABL & 4)) {
        // fragment i / WSTEP of the tap two ahead, into the slot the previous tap has just left
        int nt = tap + 2, nc = c0;
        if (nt >= 9) {
          nt -= 9;
          nc += 32;
        }
        nc = nc < CIN ? nc : CIN - 32;
        load_b1(std::integral_constant<int, (tap + 2) % 3>{}, std::integral_constant<int, i / WSTEP>{}, nc, nt);
      }
      if constexpr (i == C::TM - 1) {
        int nt = tap + 3, nc = c0;
        if (nt >= 9) {
          nt -= 9;
          nc += 32;
        }
        nc = nc < CIN ? nc : CIN - 32;  // clamp instead of branching: the count of loads in flight stays static
        if constexpr (!(FRHIP_SOLO_ABL & 4) && !SPREAD) load_b(std::integral_constant<int, slot>{}, nc, nt);
        if constexpr (Q >= 0) {
          constexpr int k = Q * 9 + tap - T0;
          if constexpr (k >= 0 && k < AUX_TAPS) {
            static_for<AUXT>([&](auto q) {
              if constexpr (k * AUXT + decltype(q)::value < NAUXL) aux_load(std::integral_constant<int, k * AUXT + decltype(q)::value>{});
            });
          }
        }
      }
    });
#pragma unroll
    for (int i = 0; i < C::TM; ++i) abase[i] += 64;  // next 32 input channels
  };
  {
    int c0 = 0;
#pragma unroll 1
    for (; c0 < CIN - NQ * 32; c0 += 32) chunk(c0, std::integral_constant<int, -1>{});
    if constexpr (NQ >= 1) chunk(CIN - NQ * 32 + 0, std::integral_constant<int, 0>{});
    if constexpr (NQ >= 2) chunk(CIN - NQ * 32 + 32, std::integral_constant<int, 1>{});
    if constexpr (NQ >= 3) chunk(CIN - NQ * 32 + 64, std::integral_constant<int, 2>{});
    if constexpr (NQ >= 4) chunk(CIN - NQ * 32 + 96, std::integral_constant<int, 3>{});
    static_assert(NQ <= 4, "peel more chunks");
  }
  if constexpr (AUXK && !FRHIP_SOLO_AUX_EARLY) static_for<NAUXL>([&](auto k) { aux_load(k); });
  // everything in flight has landed; the last MFMAs have left the pipe before a vector instruction reads an accumulator
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  FR_STAMP(4);
  FR_STAMP_CLK(7);
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
      pin_a(acc[i][j]);
      if constexpr (AUXK) pin_v(av[i][j]);
    }
  FR_STAMP(3);

  // ---------------------------------------------------------------- epilogue, on the accumulators where they are
  const int epi = p.epi;
  bf16_t* __restrict__ outp = reinterpret_cast<bf16_t*>(p.out) + rowbase * (size_t)p.ldc + ncol0 + n0 + fq * 4;
  auto cells = [&](auto tag) {
    constexpr int E = decltype(tag)::value;
    constexpr bool AUX = E == FR_EPI_PRELU_BWD || E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES || E == FR_EPI_STATS_X;
    constexpr bool SUMS = E != FR_EPI_STORE && E != FR_EPI_BIAS_RES;
    constexpr int V = E == FR_EPI_STATS_X ? 3 : 2;
    float ea[C::TN][4], eb[C::TN][4], s0[C::TN][4], s1[C::TN][4], s2[E == FR_EPI_STATS_X ? C::TN : 1][4];
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = ncol0 + n0 + j * 16 + fq * 4 + r;
        ea[j][r] = (E == FR_EPI_PRELU_BWD || E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES) ? p.epi_a[n] : 0.f;
        eb[j][r] = (E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES) ? p.epi_b[n] : 0.f;
        s0[j][r] = s1[j][r] = 0.f;
        if (E == FR_EPI_STATS_X) s2[j][r] = 0.f;
      }
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
      const int m = i * 16 + fr;
      if (m >= C::M) continue;
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        float v[4], x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        if constexpr (AUX && AUXK) {
          const uint2 u = av[i][j];
          x[0] = __uint_as_float(u.x << 16);
          x[1] = __uint_as_float(u.x & 0xFFFF0000u);
          x[2] = __uint_as_float(u.y << 16);
          x[3] = __uint_as_float(u.y & 0xFFFF0000u);
        } else {
          x[0] = x[1] = x[2] = x[3] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (E == FR_EPI_STATS) {
            s0[j][r] += v[r];
            s1[j][r] = fmaf(v[r], v[r], s1[j][r]);
          } else if (E == FR_EPI_STATS_X) {  // + the cross moment with the residual input (fr_bn_finalize_res)
            s0[j][r] += v[r];
            s1[j][r] = fmaf(v[r], v[r], s1[j][r]);
            s2[j][r] = fmaf(v[r], x[r], s2[j][r]);
          } else if (E == FR_EPI_PRELU_BWD) {
            const bool pos = x[r] > 0.f;
            s0[j][r] += pos ? 0.f : v[r] * x[r];
            v[r] = pos ? v[r] : v[r] * ea[j][r];
          } else if (E == FR_EPI_BNBWD) {
            s0[j][r] += v[r];
            s1[j][r] = fmaf(v[r], (x[r] - ea[j][r]) * eb[j][r], s1[j][r]);
          } else if (E == FR_EPI_BIAS_RES) {
            v[r] += ea[j][r] + eb[j][r] + x[r];
          }
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(outp + (size_t)m * p.ldc + j * 16) = o;
      }
    }
    if (SUMS && p.part) {  // fold the 16 pixel lanes (fr); the wave owns these channels: nothing to combine across waves
      float* prow = p.part + (size_t)b * V * (COUT * NSPL) + ncol0 + n0 + fq * 4;
#pragma unroll
      for (int j = 0; j < C::TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = s0[j][r], c = s1[j][r], d = E == FR_EPI_STATS_X ? s2[j][r] : 0.f;
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            a += __shfl_xor(a, o, 64);
            c += __shfl_xor(c, o, 64);
            if (E == FR_EPI_STATS_X) d += __shfl_xor(d, o, 64);
          }
          if (fr == 0) {
            // (the strip kernel adds its WM = 1 row group to 0.f: the same bits)
            prow[0 * (COUT * NSPL) + j * 16 + r] = 0.f + a;
            prow[1 * (COUT * NSPL) + j * 16 + r] = 0.f + c;
            if (E == FR_EPI_STATS_X) prow[2 * (COUT * NSPL) + j * 16 + r] = 0.f + d;
          }
        }
    }
  };
  switch (epi) {
    case FR_EPI_STATS: cells(std::integral_constant<int, FR_EPI_STATS>{}); break;
    case FR_EPI_STATS_X: cells(std::integral_constant<int, FR_EPI_STATS_X>{}); break;
    case FR_EPI_PRELU_BWD: cells(std::integral_constant<int, FR_EPI_PRELU_BWD>{}); break;
    case FR_EPI_BNBWD: cells(std::integral_constant<int, FR_EPI_BNBWD>{}); break;
    case FR_EPI_BIAS_RES: cells(std::integral_constant<int, FR_EPI_BIAS_RES>{}); break;
    default: cells(std::integral_constant<int, FR_EPI_STORE>{}); break;
  }
  FR_STAMP(6);
}

template <int CIN, int COUT, int W, int NSPL, int PRO, bool AUXK>
int launch(const FrConvArgs& a, hipStream_t st) {
  using C = SO<CIN, COUT, W>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_solo_kernel<CIN, COUT, W, NSPL, PRO, AUXK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
  }
  static const int* xcd = fr_option_slot("FRHIP_XCD_ORDER", 1);
  hipLaunchKernelGGL((conv3x3_solo_kernel<CIN, COUT, W, NSPL, PRO, AUXK>), dim3(a.B * NSPL), dim3(C::NTH), C::LDS, st, a, *xcd);
  FR_LAUNCH_CHECK();
}

template <int CIN, int COUT, int W, int NSPL, bool AUXK>
int by_pro(const FrConvArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return launch<CIN, COUT, W, NSPL, FR_PRO_NONE, AUXK>(a, st);
    case FR_PRO_BN: return launch<CIN, COUT, W, NSPL, FR_PRO_BN, AUXK>(a, st);
    case FR_PRO_PRELU: return launch<CIN, COUT, W, NSPL, FR_PRO_PRELU, AUXK>(a, st);
    case FR_PRO_RESBN:
      if (!a.src2 || !a.pro_a || !a.pro_b || !a.pro_c || !a.pro_d || !a.pro_out || a.mode != 0)
        FR_UNSUPPORTED("fr_conv3x3_strip: FR_PRO_RESBN needs src2, pro_a ... pro_d, pro_out and mode 0");
      return launch<CIN, COUT, W, NSPL, FR_PRO_RESBN, AUXK>(a, st);
    case FR_PRO_RESBN_SE:
      if (!a.src2 || !a.pro_a || !a.pro_b || !a.pro_c || !a.pro_d || !a.pro_g || !a.pro_out || a.mode != 0)
        FR_UNSUPPORTED("fr_conv3x3_strip: FR_PRO_RESBN_SE needs src2, pro_a ... pro_d, pro_g, pro_out and mode 0");
      return launch<CIN, COUT, W, NSPL, FR_PRO_RESBN_SE, AUXK>(a, st);
  }
  FR_UNSUPPORTED("fr_conv3x3_strip: prologue not served by the one-wave-per-SIMD instances");
}

}  // namespace

// FRHIP_SOLO=0: the 8-wave strip instances again (A/B switch)
bool fr_solo_enabled() {
  static const int* v = fr_option_slot("FRHIP_SOLO", 0);
  return *v != 0;
}

bool fr_solo_serves(const FrConvArgs& a) {
  if (!fr_solo_enabled()) return false;
  // the aux-operand epilogues are unfinished here (requested under the K loop: wrong results, see FRHIP_SOLO_AUX_EARLY;
  // requested after it: memory fault) -- the strip kernel keeps them; the K-loop measurements do not depend on them
  if (a.epi == FR_EPI_PRELU_BWD || a.epi == FR_EPI_BNBWD || a.epi == FR_EPI_BIAS_RES || a.epi == FR_EPI_STATS_X) return false;
  return a.SC == 256 && a.N == 256 && a.SW == 14 && a.B > 160;
}

int fr_solo_launch(const FrConvArgs& a, hipStream_t st) {
  const bool aux = a.epi == FR_EPI_PRELU_BWD || a.epi == FR_EPI_BNBWD || a.epi == FR_EPI_BIAS_RES || a.epi == FR_EPI_STATS_X;
  if (aux && !a.aux) FR_UNSUPPORTED("fr_conv3x3_strip: this epilogue needs aux");
  if (a.ldc % 4 || (a.aux && a.ldaux % 4)) FR_UNSUPPORTED("fr_conv3x3_strip: strides must be 8-byte multiples");
  return aux ? by_pro<256, 256, 14, 1, true>(a, st) : by_pro<256, 256, 14, 1, false>(a, st);
}
