// frhip -- stride-1 3x3 convolution, whole image per workgroup, computing waves that issue NO vector-memory instruction
// (bf16, gfx950; round 5).
//
// Why (profiles/r05_solo_stamps_and_ablations.txt): in the strip kernels every wave streams its own weight fragments
// L2 -> registers between its MFMAs.  Timing ablations of the K loop on one wave per SIMD say what that costs: MFMAs alone
// 16.96 cycles each, + all LDS reads 17.32 (free), + the weight stream 20.84 -- 45 cycles of lost MFMA issue per
// global_load_dwordx4, whether the loads are waited for or not, batched or spread.  A vector-memory instruction in a wave
// that issues MFMAs stalls the matrix pipe; an LDS read does not.
//
// So here the weights reach the MFMAs through LDS as well.  512 threads = two groups of four waves, one wave of each group
// per SIMD:
//   * group A (waves 0-3) only computes: ds_read_b128 + MFMA, nothing else in the K loop.  Wave w owns 32 output channels
//     (2 tiles) x all 196 pixels (13 tiles) of the current PASS; the 256 output channels are two passes of 128 over the
//     same LDS-resident image.
//   * group B (waves 4-7) moves data: weight fragments global -> registers (P steps ahead) -> a two-slot LDS ring, one
//     slot = one (tap, 32 input channels) step of the pass = 8 KB in fragment order (a fragment is one linear, conflict-free
//     ds_read_b128).  Hand-off by two monotonic LDS counters per wave pair (FULL: steps written, FREE: steps read), polled one
//     step ahead so that a computing wave never waits in steady state.
// LDS: image 142,464 B (layout of conv3x3_strip.hip) + ring 16,384 B + counters.
// The K loop of the computing waves is a straight line of asm volatile statements with hand-counted lgkmcnt waits (see
// conv3x3_solo.hip for why); the data-moving waves count vmcnt by hand.
//
// Same arithmetic, K order and summation order as conv3x3_strip_kernel<256,256,14,14,8,8>: bit-identical outputs and
// partial rows.  Reference: Conv2d 3x3 s1 of bottleneck_IR (backbone/model_irse.py:57-59) + its data gradient.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "frhip_internal.h"

#ifdef FRHIP_STAMPS
__device__ unsigned long long* fr_stamp_buf_relay = nullptr;
extern "C" int fr_debug_set_stamp_buffer_relay(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fr_stamp_buf_relay), &dev_ptr, sizeof(dev_ptr));
}
#define FR_STAMP(k)                                                                                               \
  do {                                                                                                            \
    if (tid == 0 && fr_stamp_buf_relay) fr_stamp_buf_relay[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define FR_STAMP_CLK(k)                                                                                           \
  do {                                                                                                            \
    if (tid == 0 && fr_stamp_buf_relay) fr_stamp_buf_relay[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define FR_STAMP(k)
#define FR_STAMP_CLK(k)
#endif

#ifndef FRHIP_RELAY_ABL
#define FRHIP_RELAY_ABL 0  // timing ablations (diagnostic builds; results wrong): 1 computing waves do not poll, 2 no data-moving work, 4 data-moving waves at s_setprio 3
#endif
#ifndef FRHIP_RELAY_P
#define FRHIP_RELAY_P 6  // steps of weight fragments a data-moving wave keeps in flight (even)
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

// ---- the instructions of the hand-scheduled loops
template <int OFF>
__device__ __forceinline__ void ds_read128(i32x4& dst, int addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void ds_read128v(i32x4& dst, int addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
__device__ __forceinline__ void ds_write128(int addr, const i32x4& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ds_read32(int& dst, int addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr)); }
// one lane adds 1 to an LDS counter (every lane of these waves is active: EXEC is all ones before and after)
__device__ __forceinline__ void ds_inc(int addr, int one) {
  asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(one) : "memory");
}
__device__ __forceinline__ void gload128(i32x4& dst, int voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void mfma_v(f32x4& acc, const i32x4& w, const i32x4& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N));
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
}
// the counter value requested N DS operations ago (in vflag) must be >= need; otherwise poll until it is
template <int N>
__device__ __forceinline__ void poll_ge(int& vflag, int addr, int need) {
  int s, spins;
  asm volatile(
      "s_waitcnt lgkmcnt(%5)\n\t"
      "v_readfirstlane_b32 %1, %0\n\t"
      "s_cmp_ge_u32 %1, %4\n\t"
      "s_cbranch_scc1 1f\n\t"
      "s_mov_b32 %2, 0\n"
      "0:\n\t"
      "ds_read_b32 %0, %3\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %1, %0\n\t"
      "s_add_u32 %2, %2, 1\n\t"
      "s_cmp_ge_u32 %2, 0x80000\n\t"  // bounded (a protocol bug must not hang the device), then wrong results
      "s_cbranch_scc1 1f\n\t"
      "s_cmp_ge_u32 %1, %4\n\t"
      "s_cbranch_scc0 0b\n"
      "1:\n\t"
      : "+v"(vflag), "=&s"(s), "=&s"(spins)
      : "v"(addr), "s"(need), "i"(N)
      : "scc", "memory");
}
__device__ __forceinline__ void pin_f(f32x4& acc) { asm volatile("" : "+v"(acc)); }

constexpr int CIN = 256, COUT = 256, W = 14;
constexpr int NTH = 512;
constexpr int GW = W + 2, GH = W + 2;
constexpr int CH = CIN / 8;
constexpr int PSTR = CIN * 2 + 32;
constexpr int RSTR = GW * PSTR + 192;
constexpr int IMG = GH * RSTR + 256;            // + slack for the ring's reads past the last chunk
constexpr int M = W * W, TM = (M + 15) / 16;    // 196 pixels, 13 tiles
constexpr int TN = 2;                           // 16-channel tiles per computing wave and pass
constexpr int PASSN = 4 * TN * 16;              // 128 output channels per pass
constexpr int NPASS = COUT / PASSN;
constexpr int SLOT = PASSN * 32 * 2;            // 8 KB: one (tap, 32 input channels) step of a pass
constexpr int WR0 = IMG;                        // weight ring
constexpr int FL0 = WR0 + 2 * SLOT;             // counters: FULL at FL0, FREE at FL0 + 64
constexpr int LDS = FL0 + 128;
constexpr int NVAL = M * CH, PER = (NVAL + NTH - 1) / NTH, NHALO = (GH * GW - M) * CH;
constexpr int KSTEPS = 9 * (CIN / 32);          // 72 steps per pass
constexpr int D = 9;                            // pixel-fragment ring depth
static_assert(LDS <= 160 * 1024, "LDS budget");

// DS operations of a computing wave per step, in issue order: refill(0..2), wread0, wread1, add, refill(3..8), flagread,
// refill(9..12) -- position of the refill behind tile i
constexpr int pos_refill(int i) { return i + (i >= 3 ? 3 : 0) + (i >= 9 ? 1 : 0); }
// DS operations issued after the producer of tile i's pixel fragment (refill behind tile seq - D) when tile i waits for it
constexpr int ring_younger(int i) {
  return i >= D ? pos_refill(i) - 1 - pos_refill(i - D) : 16 - pos_refill(i + TM - D) + pos_refill(i);
}

template <int PRO>
__global__ __launch_bounds__(512, 2) void conv3x3_relay_kernel(const FrConvArgs p, const int xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  constexpr bool RES = PRO == FR_PRO_RESBN || PRO == FR_PRO_RESBN_SE;
  const bf16_t* __restrict__ wgt = reinterpret_cast<const bf16_t*>(p.w);
  const int b = xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;  // image
  const size_t rowbase = (size_t)b * M;
  const int fr = lane & 15, fq = lane >> 4;
  const int flip = p.mode;

  FR_STAMP(0);
  if (tid < 32) reinterpret_cast<int*>(smem + FL0)[tid] = 0;
  // ---------------------------------------------------------------- image -> LDS (prologue applied once), all 8 waves
  {
    const bf16_t* __restrict__ img = reinterpret_cast<const bf16_t*>(p.src) + rowbase * (size_t)p.lda;
    const bf16_t* __restrict__ img2 = RES ? reinterpret_cast<const bf16_t*>(p.src2) + rowbase * (size_t)p.lda : nullptr;
    bf16_t* __restrict__ pro_out = RES && p.pro_out ? reinterpret_cast<bf16_t*>(p.pro_out) + rowbase * (size_t)p.lda : nullptr;
    const int ch = tid % CH;
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
    for (int q = tid; q < NHALO; q += NTH) {  // halo: zero rows / columns around the image
      const int hp = q / CH, c = q - hp * CH;
      int gh, gw;
      if (hp < GW) {
        gh = 0, gw = hp;
      } else if (hp < 2 * GW) {
        gh = GH - 1, gw = hp - GW;
      } else {
        const int r = hp - 2 * GW;
        gh = 1 + (r >> 1), gw = (r & 1) ? GW - 1 : 0;
      }
      st16(smem + gh * RSTR + gw * PSTR + c * 16, zero16());
    }
    constexpr int NB = RES ? 2 : 1;
    constexpr int UNR = (PER + NB - 1) / NB;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
      U128 v[UNR], v2[RES ? UNR : 1];
      int off[UNR], lo[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int idx = (bt * UNR + u) * NTH + tid;
        idx = idx < NVAL ? idx : NVAL - CH + ch;  // clamp (the duplicate rewrites the same bytes)
        const int px = idx / CH;
        const int h = px / W, w = px - h * W;
        off[u] = px * p.lda + ch * 8;
        lo[u] = (h + 1) * RSTR + (w + 1) * PSTR + ch * 16;
        v[u] = ld16(img + off[u]);
        if (RES) v2[u] = ld16(img2 + off[u]);
      }
      __builtin_amdgcn_sched_barrier(0);  // the whole batch in flight before the first prologue instruction
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if ((bt * UNR + u) * NTH >= NVAL) continue;  // compile-time: rounds past the image
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
            if (pro_out) st16(pro_out + off[u], x);
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
  FR_STAMP_CLK(8);

  // one FULL / FREE counter pair per (data-moving wave w4, computing wave w4): a computing wave reads only what its own
  // partner writes, so the four channels are independent (a SUM over the four would not say that the slowest has arrived)
  const int a_full = FL0 + w4 * 16, a_free = FL0 + 64 + w4 * 16;
  const int one = 1;
  if (grp == 1) {
    if (FRHIP_RELAY_ABL & 4) __builtin_amdgcn_s_setprio(3);
    if (FRHIP_RELAY_ABL & 2) return;
    // ================================================================ data-moving waves: weights -> LDS ring
    // wave w4 moves the fragments of computing wave w4: per step two 1-KB fragments (tile j = 0, 1), lane (fr, fq) holds
    // weight row (pass*128 + w4*32 + j*16 + fr), input channels [c0 + 8 fq, + 8) of tap `tap`
    constexpr int P = FRHIP_RELAY_P;
    static_assert(P % 2 == 0, "the slot of a step must be a compile-time parity inside the unrolled trip");
    int woff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) woff[j] = ((w4 * TN * 16 + j * 16 + fr) * 9 * CIN + fq * 8) * 2;
    const int wdst = WR0 + (w4 * TN) * 1024 + lane * 16;  // + slot * SLOT + j * 1024
    i32x4 pf[P][TN];
    auto issue = [&](auto stag, int g) {  // global step g -> (pass, chunk, tap); steps past the end repeat the last one
      constexpr int s = decltype(stag)::value;
      g = g < NPASS * KSTEPS ? g : NPASS * KSTEPS - 1;
      const int pass = g / KSTEPS, k = g - pass * KSTEPS;
      const int c0 = (k / 9) * 32, tap = k % 9;
      const int wt = flip ? 8 - tap : tap;
      const bf16_t* sb = wgt + (size_t)pass * PASSN * 9 * CIN + (wt * CIN + c0);
#pragma unroll
      for (int j = 0; j < TN; ++j) gload128(pf[s][j], woff[j], sb);
    };
    static_for<P>([&](auto s) { issue(s, decltype(s)::value); });
    int vflag = 0;
    constexpr int NG = NPASS * KSTEPS + 1;  // one dummy step past the end: the computing waves read one step ahead
    for (int g0 = 0; g0 < NG; g0 += P) {
      static_for<P>([&](auto stag) {
        constexpr int s = decltype(stag)::value;
        const int g = g0 + s;
        if (g < NG) {
          // slot g % 2 was read last for step g - 2: the partner has issued g - 1 reads (the counter only grows: a value seen
          // earlier stays valid; LDS executes a wave's operations in order, so the add behind the writes publishes them)
          if (g >= 2) poll_ge<0>(vflag, a_free, g - 1);
          wait_vm<TN*(P - 1)>();
          ds_write128(wdst + (s % 2) * SLOT, pf[s][0]);
          ds_write128(wdst + (s % 2) * SLOT + 1024, pf[s][1]);
          ds_inc(a_full, one);
          issue(stag, g + P);
        }
      });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  } else {
    // ================================================================ computing waves: ds_read_b128 + MFMA only
    int abase0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int m = i * 16 + fr;
      m = m < M ? m : 0;
      const int h = m / W, w = m - h * W;
      abase0[i] = h * RSTR + w * PSTR + fq * 16;
    }
    const int wsrc = WR0 + (w4 * TN) * 1024 + lane * 16;
    const int epi = p.epi;
    i32x4 wreg[2][TN];  // weight fragments of the current / the next step (live across the pass boundary)
    for (int pass = 0; pass < NPASS; ++pass) {
      const int n0 = pass * PASSN + w4 * TN * 16;
      f32x4 acc[TM][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int abase[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) abase[i] = abase0[i];
      i32x4 ring[D];
      int vflag = 0;
      int g = pass * KSTEPS;  // global step
      // pipeline fill: the first D pixel fragments, the weights of the first step, the counter for the second
      auto a_read = [&](auto sttag) {  // ring[seq % D] <- fragment of tile seq (seq in [0, 19 * TM): two chunks + 1 step)
        constexpr int seq = decltype(sttag)::value;
        constexpr int st = seq / TM, i = seq - st * TM;  // local step (0..18), tile
        constexpr int cadd = (st / 9) * 64;
        constexpr int tap = st % 9;
        ds_read128<(tap / 3) * RSTR + (tap % 3) * PSTR + cadd>(ring[seq % D], abase[i]);
      };
      static_for<D>([&](auto d) { a_read(d); });
      if (pass == 0) {  // (later passes: the last step of the pass before has read them)
        if (!(FRHIP_RELAY_ABL & 1)) poll_ge<0>(vflag, a_full, g + 1);
        ds_read128v(wreg[0][0], wsrc);
        ds_read128v(wreg[0][1], wsrc + 1024);
        ds_inc(a_free, one);
      }
      ds_read32(vflag, a_full);
      wait_lgkm<0>();
      // main loop: two 32-channel chunks (18 steps) per trip
      for (int cc = 0; cc < CIN / 64; ++cc) {
        static_for<18 * TM>([&](auto sttag) {
          constexpr int seq = decltype(sttag)::value;
          constexpr int st = seq / TM, i = seq - st * TM;
          // (KSTEPS is even and a trip is 18 steps: the slot of a step is its local parity)
          wait_lgkm<ring_younger(i)>();  // at i == 0 this also covers the weight fragments of the step (12 younger)
#pragma unroll
          for (int j = 0; j < TN; ++j) mfma_v(acc[i][j], wreg[st % 2][j], ring[seq % D]);
          a_read(std::integral_constant<int, seq + D>{});
          if constexpr (i == 2) {
            // the next step's weights: its slot must be FULL (counter requested behind tile 8 of the previous step)
            if constexpr (!(FRHIP_RELAY_ABL & 1)) poll_ge<7>(vflag, a_full, g + st + 2);
            ds_read128v(wreg[(st + 1) % 2][0], wsrc + ((st + 1) % 2) * SLOT);
            ds_read128v(wreg[(st + 1) % 2][1], wsrc + ((st + 1) % 2) * SLOT + 1024);
            ds_inc(a_free, one);
          }
          if constexpr (i == 8) ds_read32(vflag, a_full);
        });
        g += 18;
#pragma unroll
        for (int i = 0; i < TM; ++i) abase[i] += 128;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) pin_f(acc[i][j]);
      if (pass == 0) {
        FR_STAMP(3);
        FR_STAMP_CLK(11);
      } else {
        FR_STAMP(5);
      }

      // ---------------------------------------------------------------- epilogue of the pass, on the accumulators
      bf16_t* __restrict__ outp = reinterpret_cast<bf16_t*>(p.out) + rowbase * (size_t)p.ldc + n0 + fq * 4;
      const bf16_t* __restrict__ auxp = p.aux ? reinterpret_cast<const bf16_t*>(p.aux) + rowbase * (size_t)p.ldaux + n0 + fq * 4 : nullptr;
      auto cells = [&](auto tag) {
        constexpr int E = decltype(tag)::value;
        constexpr bool AUX = E == FR_EPI_PRELU_BWD || E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES || E == FR_EPI_STATS_X;
        constexpr bool SUMS = E != FR_EPI_STORE && E != FR_EPI_BIAS_RES;
        constexpr int V = E == FR_EPI_STATS_X ? 3 : 2;
        float ea[TN][4], eb[TN][4], s0[TN][4], s1[TN][4], s2[E == FR_EPI_STATS_X ? TN : 1][4];
        uint2 av[AUX ? TM : 1][AUX ? TN : 1];
        if constexpr (AUX) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            int m = i * 16 + fr;
            m = m < M ? m : M - 1;
#pragma unroll
            for (int j = 0; j < TN; ++j) av[i][j] = *reinterpret_cast<const uint2*>(auxp + (size_t)m * p.ldaux + j * 16);
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + j * 16 + fq * 4 + r;
            ea[j][r] = (E == FR_EPI_PRELU_BWD || E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES) ? p.epi_a[n] : 0.f;
            eb[j][r] = (E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES) ? p.epi_b[n] : 0.f;
            s0[j][r] = s1[j][r] = 0.f;
            if (E == FR_EPI_STATS_X) s2[j][r] = 0.f;
          }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = i * 16 + fr;
          if (m >= M) continue;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v[4], x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
            if constexpr (AUX) {
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
              } else if (E == FR_EPI_STATS_X) {
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
        if (SUMS && p.part) {  // fold the 16 pixel lanes (fr); a wave owns its channels: nothing to combine across waves
          float* prow = p.part + (size_t)b * V * COUT + n0 + fq * 4;
#pragma unroll
          for (int j = 0; j < TN; ++j)
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
                prow[0 * COUT + j * 16 + r] = 0.f + a;  // (the strip kernel adds its one row group to 0.f: the same bits)
                prow[1 * COUT + j * 16 + r] = 0.f + c;
                if (E == FR_EPI_STATS_X) prow[2 * COUT + j * 16 + r] = 0.f + d;
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
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the epilogue's own loads: nothing of it in flight in the next pass
      if (pass == 0) {
        FR_STAMP(4);
        FR_STAMP_CLK(9);
      }
    }
    FR_STAMP(6);
    FR_STAMP_CLK(10);
  }
}

template <int PRO>
int launch(const FrConvArgs& a, hipStream_t st) {
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_relay_kernel<PRO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS);
  }
  static const int* xcd = fr_option_slot("FRHIP_XCD_ORDER", 1);
  hipLaunchKernelGGL((conv3x3_relay_kernel<PRO>), dim3(a.B), dim3(NTH), LDS, st, a, *xcd);
  FR_LAUNCH_CHECK();
}

}  // namespace

// FRHIP_RELAY=0: the 8-wave strip instance again (A/B switch)
bool fr_relay_serves(const FrConvArgs& a) {
  static const int* on = fr_option_slot("FRHIP_RELAY", 0);
  if (!*on) return false;
  return a.SC == 256 && a.N == 256 && a.SW == 14 && a.B > 160;
}

int fr_relay_launch(const FrConvArgs& a, hipStream_t st) {
  const bool aux = a.epi == FR_EPI_PRELU_BWD || a.epi == FR_EPI_BNBWD || a.epi == FR_EPI_BIAS_RES || a.epi == FR_EPI_STATS_X;
  if (aux && !a.aux) FR_UNSUPPORTED("fr_conv3x3_strip: this epilogue needs aux");
  if (a.ldc % 4 || (a.aux && a.ldaux % 4)) FR_UNSUPPORTED("fr_conv3x3_strip: strides must be 8-byte multiples");
  switch (a.pro) {
    case FR_PRO_NONE: return launch<FR_PRO_NONE>(a, st);
    case FR_PRO_BN: return launch<FR_PRO_BN>(a, st);
    case FR_PRO_PRELU: return launch<FR_PRO_PRELU>(a, st);
    case FR_PRO_RESBN:
      if (!a.src2 || !a.pro_a || !a.pro_b || !a.pro_c || !a.pro_d || !a.pro_out || a.mode != 0)
        FR_UNSUPPORTED("fr_conv3x3_strip: FR_PRO_RESBN needs src2, pro_a ... pro_d, pro_out and mode 0");
      return launch<FR_PRO_RESBN>(a, st);
    case FR_PRO_RESBN_SE:
      if (!a.src2 || !a.pro_a || !a.pro_b || !a.pro_c || !a.pro_d || !a.pro_g || !a.pro_out || a.mode != 0)
        FR_UNSUPPORTED("fr_conv3x3_strip: FR_PRO_RESBN_SE needs src2, pro_a ... pro_d, pro_g, pro_out and mode 0");
      return launch<FR_PRO_RESBN_SE>(a, st);
  }
  FR_UNSUPPORTED("fr_conv3x3_strip: prologue not served by the relay instance");
}
