// frhip -- weight gradient of the stride-1 3x3 convolutions at 14x14 / 28x28, warp-specialised (bf16, round 3).
//
//   dw[co][tap][ci] = sum_{pixels p}  g[p][co] * pro(x[p + tap][ci])
//
// Same decomposition as conv_wgrad_strip.hip (a workgroup owns one 64 co x 64 ci x 9 tap block of dW and a group of
// images; the group's partial goes to a slab), different machinery.  The strip kernel spent 8.5 vector instructions per
// MFMA (address arithmetic of the register prefetch, the PReLU / BN prologue, 276 v_mov per image to pair half-fragments)
// in the SAME waves that issue the MFMAs, and stopped the MFMAs for every commit between two barriers: its K loop ran at
// ~60 % of the MFMA rate.  Here the eight waves of a workgroup have two roles (as in conv3x3_roll64.hip):
//
//   waves 0-3 (one per SIMD)  MFMA only.  Wave w owns ci tile w (16 channels) x all 64 co x 9 taps = 36 accumulator tiles
//                             (144 VGPRs), so a g fragment serves 9 taps and an input fragment serves 4 co tiles: 26
//                             ds_read_b64_tr_b16 per 36 MFMAs, no vector ALU instruction in the loop (every address is
//                             lane base + immediate: the K axis of both tiles is laid out in rows of RW = 16 / 32 slots,
//                             so tap (kh, kw) of K slot k sits at tile position k + RW kh + kw).
//   waves 4-7                 move the data: global loads two phases ahead into two register sets, BN / PReLU prologue
//                             on the registers, ds_write into the buffer the computing waves are NOT reading.
//
// An image is cut into phases of whole K steps (14x14: rows 0-7 = 4 steps, rows 8-13 = 3 steps; 28x28: 4 rows = 4 steps)
// that alternate between two LDS buffers; one barrier per phase.  The barrier sits two MFMA groups BEFORE the end of a
// phase's arithmetic: at that point the computing waves have every fragment of the phase in registers, so behind the
// barrier they finish the phase while the first fragments of the next one are already on their way -- the MFMA pipe
// never drains at a phase boundary.  Halo rows / columns of the input tiles and the surplus K slots of the g tiles are
// zeroed once and never written again.
//
// Reference arithmetic: autograd weight gradient of Conv2d(c, d, (3,3), (1,1), 1) in bottleneck_IR
// (backbone/model_irse.py:57-59) with BN apply (:57) / PReLU (:58) folded into the input load.
#include <stdlib.h>

#include "common.h"
#include "frhip_internal.h"
#include "slab_sum.h"

namespace {

constexpr int CT = 64;             // co and ci tile
constexpr int TSTR = CT * 2 + 32;  // LDS row stride (bytes): conflict-free transposing reads (see conv_wgrad_strip.hip)
constexpr int NLT = 256;           // data-moving threads (waves 4-7)

typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;

__device__ __forceinline__ s16x8 tr_frag(const char* p0, const char* p1) {
  const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p0);
  const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)p1);
  const s16x4 ai = __builtin_bit_cast(s16x4, a), bi = __builtin_bit_cast(s16x4, b);
  return (s16x8){ai[0], ai[1], ai[2], ai[3], bi[0], bi[1], bi[2], bi[3]};
}

// Geometry of one width.  A phase = PR output rows = PR * RW / 32 K steps; phase kinds alternate per image as listed.
template <int W_>
struct RC;
// krows = rows of K slots the phase's arithmetic covers (a multiple of 32 / RW); > rows only at 7x7, where the image's 7
// rows of 8 slots are followed by one row of zero g slots -- the same row of zeros is the halo below the image.
template <>
struct RC<14> {
  static constexpr int W = 14, RW = 16, NPH = 2;          // phases per image
  static constexpr int rows(int ph) { return ph == 0 ? 8 : 6; }
  static constexpr int krows(int ph) { return rows(ph); }
  static constexpr int row0(int ph) { return ph == 0 ? 0 : 8; }
};
template <>
struct RC<28> {
  static constexpr int W = 28, RW = 32, NPH = 7;
  static constexpr int rows(int) { return 4; }
  static constexpr int krows(int) { return 4; }
  static constexpr int row0(int ph) { return 4 * ph; }
};
template <>
struct RC<7> {  // uniform schedule, one phase (two K steps) per image (pairs of images per phase -- four K steps, only real
                // pixels moved, no row clamping: 0.085 against 0.083 ms, measured and dropped)
  static constexpr int W = 7, RW = 8, NPH = 1;
  static constexpr int rows(int) { return 7; }
  static constexpr int krows(int) { return 8; }
  static constexpr int row0(int) { return 0; }
};

template <int W>
struct RL {  // LDS layout: buffer 0 holds the even phases of the schedule, buffer 1 the odd ones
  using C = RC<W>;
  static constexpr int pr(int buf) { return C::NPH == 2 ? C::krows(buf) : C::krows(0); }
  static constexpr int g_bytes(int buf) { return pr(buf) * C::RW * TSTR; }                 // K slots of the phase
  static constexpr int a_bytes(int buf) { return ((pr(buf) + 2) * C::RW + 16) * TSTR; }    // + the tail a kw-shifted read touches
  static constexpr int BUF0 = g_bytes(0) + a_bytes(0);
  static constexpr int LDS = BUF0 + g_bytes(1) + a_bytes(1);
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// pro2<PRO>: the BN / PReLU prologue on one dword (two bf16), written as instructions -- frhip_internal.h

#define LDS_FENCE_BARRIER_RAW() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Diagnostic build (make stamps; never loaded by the product): per workgroup, cycles (s_memtime) of the whole kernel and
// of the image loop, and the cycles wave 0 (computing) / wave 4 (data-moving) spend inside the hand-over barriers.
#ifdef FRHIP_STAMPS
__device__ unsigned long long* fr_stamp_buf_wgr = nullptr;
#define TSTAMP() __builtin_amdgcn_s_memtime()
// one region of 4096 workgroups x 16 qwords per width (14, 28, 56, 112), so that a whole training step leaves the stamps
// of the LAST launch of every width
#define STAMP_SLOT(W) ((size_t)((W) == 14 ? 0 : (W) == 28 ? 1 : (W) == 56 ? 2 : 3) * 4096 * 16 + (size_t)blockIdx.x * 16)
#define LDS_FENCE_BARRIER()                         \
  do {                                              \
    const unsigned long long t0__ = TSTAMP();       \
    LDS_FENCE_BARRIER_RAW();                        \
    bar_wait += TSTAMP() - t0__;                    \
  } while (0)
#else
#define LDS_FENCE_BARRIER() LDS_FENCE_BARRIER_RAW()
#endif

// This workgroup's share of the deferred slab sum of the PREVIOUS weight-gradient launch (FrWgradArgs.prev_*), split between
// the wave roles of the three kernels below: [e0, cut) by the computing waves (tid < 256, two elements per thread and
// pass), [cut, e1) by the data-moving waves (one element).
__device__ __forceinline__ void fold_prev_share(const FrWgradArgs& p, bool movers, int tid) {
  const long long n4 = p.prev_n >> 2;
  const long long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long long e0 = (long long)blockIdx.x * per;
  long long e1 = e0 + per;
  if (e1 > n4) e1 = n4;
  if (e0 >= e1) return;
  const long long cut = e0 + ((e1 - e0) * 2 + 2) / 3;
  if (!movers) slab_sum_range<256, 2>(p.prev_slab, p.prev_groups, n4, e0, cut, p.prev_dw, tid);
  else if (cut < e1) slab_sum_range<256, 1>(p.prev_slab, p.prev_groups, n4, cut, e1, p.prev_dw, tid - 256);
}

template <int W, int PRO>
__global__ __launch_bounds__(512, 2) void conv_wgrad_roll_kernel(const FrWgradArgs p) {
  using C = RC<W>;
  using L = RL<W>;
  constexpr int RW = C::RW, NPH = C::NPH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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
  // work of a group: whole images (two-phase schedule) or a run of phases that may start inside an image (uniform)
  const int units = NPH == 2 ? p.B : p.B * NPH;
  const int per = (units + p.nsplit - 1) / p.nsplit;
  const int b_begin = group * per;
  int b_end = b_begin + per;
  if (b_end > units) b_end = units;
  const int nimg = b_end > b_begin ? b_end - b_begin : 0;  // images / phases of this workgroup

#ifdef FRHIP_STAMPS
  unsigned long long bar_wait = 0;
  const unsigned long long t_start = TSTAMP(), rt_start = __builtin_amdgcn_s_memrealtime();
#endif
  // zero both buffers once: halo rows / columns, surplus K slots and tails stay zero for the whole launch
  for (int idx = tid; idx < L::LDS / 16; idx += 512) st16(smem + idx * 16, zero16());
  __syncthreads();

  // Deferred slab sum of the PREVIOUS weight-gradient launch of this stream (FrWgradArgs.prev_*): every workgroup adds its
  // share of that launch's slabs in the fixed order g = 0, 1, ... while its own first tiles are in flight -- the sum no
  // longer costs a launch (and a trip of 2 x groups x |dW| bytes through a kernel of its own).
  // The computing waves (tid < 256; their accumulator registers are not live yet) take two thirds of this workgroup's
  // share with two elements per thread and pass, the data-moving waves one third with one element, squeezed in between
  // their first two tile requests -- one memory round trip for the whole sum at every shape of the step (three sequential
  // ones cost 8.5-9.9 us per launch in situ, tools/stamps_step.py).
  auto fold_prev = [&](bool movers) { fold_prev_share(p, movers, tid); };

#ifdef ROLL_ALL_PRIO  // experiment: the weight-gradient waves win every issue arbitration against co-resident channel-wise kernels
  __builtin_amdgcn_s_setprio(ROLL_ALL_PRIO);
#endif
  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------- data-moving waves
#ifdef ROLL_LOADER_PRIO
    __builtin_amdgcn_s_setprio(ROLL_LOADER_PRIO);
#endif
    const int lt = tid - 256;
    const int ch = lt & 7;  // the 8-channel chunk of a pixel this thread always handles
    // uniform base (scalar registers) + 32-bit per-lane byte offset: the loads take the saddr form, no 64-bit vector
    // address arithmetic per load and image
    const char* __restrict__ G = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.g) + cot * CT);
    const char* __restrict__ X = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.src) + cit * CT);
    float pa[8], pb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pa[j] = PRO != FR_PRO_NONE ? p.pro_a[cit * CT + ch * 8 + j] : 0.f;
      pb[j] = PRO == FR_PRO_BN ? p.pro_b[cit * CT + ch * 8 + j] : 0.f;
    }
    // Phase ph moves G rows [row0, row0+PR) and input rows [row0-1, row0+PR+1) clipped to the image.  Slots of 256
    // threads; a slot is wholly g or wholly input (compile time), lanes past the end repeat the last chunk (same bytes
    // to the same address: no predication anywhere, the waitcnt bookkeeping of the compiler stays exact).
    auto g_chunks = [](int ph) constexpr { return C::rows(ph) * W * 8; };
    auto a_r0 = [](int ph) constexpr { return C::row0(ph) > 0 ? C::row0(ph) - 1 : 0; };
    auto a_r1 = [](int ph) constexpr { return C::row0(ph) + C::rows(ph) + 1 < W ? C::row0(ph) + C::rows(ph) + 1 : W; };
    auto a_chunks = [=](int ph) constexpr { return (a_r1(ph) - a_r0(ph)) * W * 8; };
    constexpr int NGS = (C::rows(0) * W * 8 + NLT - 1) / NLT;                         // g slots (largest phase)
    constexpr int NAS = ((C::rows(0) + 2) * W * 8 + NLT - 1) / NLT;                   // input slots (largest phase)
    constexpr int NS = NGS + NAS;

    // register set of one phase in flight
    struct Set {
      U128 v[NS];
    };
    auto issue = [&](Set& s, int b, auto phc) {
      constexpr int ph = decltype(phc)::value;
      const char* gi = G + (size_t)b * (W * W) * (size_t)p.ldg * 2;
      const char* xi = X + (size_t)b * (W * W) * (size_t)p.lda * 2;
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        if (u < NGS) {
          int q = u * NLT + lt;
          if (u * NLT >= g_chunks(ph)) continue;  // slot unused in this phase kind
          q = q < g_chunks(ph) ? q : g_chunks(ph) - 1;
          const int px = q >> 3;  // pixel inside the phase's rows (ch == q & 7 because NLT % 8 == 0 and the clamp keeps ch)
          s.v[u] = ld16(gi + (unsigned)(((C::row0(ph) * W + px) * p.ldg + ch * 8) * 2));
        } else {
          int q = (u - NGS) * NLT + lt;
          if ((u - NGS) * NLT >= a_chunks(ph)) continue;
          q = q < a_chunks(ph) ? q : a_chunks(ph) - 1;
          const int px = q >> 3;
          s.v[u] = ld16(xi + (unsigned)(((a_r0(ph) * W + px) * p.lda + ch * 8) * 2));
        }
      }
    };
    auto commit = [&](Set& s, char* buf, auto phc) {
      constexpr int ph = decltype(phc)::value;
      char* Gs = buf;
      char* As = buf + L::g_bytes(ph & 1);
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        if (u < NGS) {
          int q = u * NLT + lt;
          if (u * NLT >= g_chunks(ph)) continue;
          q = q < g_chunks(ph) ? q : g_chunks(ph) - 1;
          const int px = q >> 3, r = px / W, c = px - r * W;
          st16(Gs + (r * RW + c) * TSTR + ch * 16, s.v[u]);
        } else {
          int q = (u - NGS) * NLT + lt;
          if ((u - NGS) * NLT >= a_chunks(ph)) continue;
          q = q < a_chunks(ph) ? q : a_chunks(ph) - 1;
          const int px = q >> 3, r = px / W, c = px - r * W;
          U128 x = s.v[u];
          if (PRO != FR_PRO_NONE) {
            x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
            x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
            x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
            x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
          }
          // tile row of image row h in this phase: h - (row0 - 1); column c + 1 (column 0 / W + 1 = zero halo)
          const int tr = a_r0(ph) + r - (C::row0(ph) - 1);
          st16(As + (tr * RW + c + 1) * TSTR + ch * 16, x);
        }
      }
    };

    char* const buf0 = smem;
    char* const buf1 = smem + L::BUF0;
    if (nimg == 0) {  // (never with the group counts the host computes; both roles then skip every barrier)
      if (p.prev_n) fold_prev(true);
      return;
    }
    if constexpr (NPH == 2) {
      // 14x14: phase kind == buffer.  set0 <-> (phase 0, buf0), set1 <-> (phase 1, buf1)
      Set s0, s1;
      issue(s0, b_begin, std::integral_constant<int, 0>{});
      if (p.prev_n) fold_prev(true);
      __builtin_amdgcn_sched_barrier(0);
      issue(s1, b_begin, std::integral_constant<int, 1>{});
      commit(s0, buf0, std::integral_constant<int, 0>{});
      {
        const int bn = b_begin + 1 < b_end ? b_begin + 1 : b_end - 1;
        issue(s0, bn, std::integral_constant<int, 0>{});
      }
      LDS_FENCE_BARRIER();  // B0: phase 0 of the first image is in buf0
#ifdef FRHIP_STAMPS
      const unsigned long long t_loop = TSTAMP();
      bar_wait = 0;
#endif
#pragma unroll 1
      for (int i = 0; i < nimg; ++i) {
        const int b = b_begin + i;
        const int b1 = b + 1 < b_end ? b + 1 : b_end - 1, b2 = b + 2 < b_end ? b + 2 : b_end - 1;
        // computing waves: phase 0 of image i (buf0).  buf1 is free.  The sched_barriers keep the order commit -> issue:
        // hoisted above the commit, the new requests would be the youngest loads in flight at its wait.
        commit(s1, buf1, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        issue(s1, b1, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        LDS_FENCE_BARRIER();  // phase 1 of image i is in buf1; buf0 is free
        commit(s0, buf0, std::integral_constant<int, 0>{});  // phase 0 of image i + 1 (past the end: a repeat nobody reads)
        __builtin_amdgcn_sched_barrier(0);
        issue(s0, b2, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        LDS_FENCE_BARRIER();  // phase 0 of image i + 1 is in buf0; buf1 is free
      }
#ifdef FRHIP_STAMPS
      if (tid == 256 && fr_stamp_buf_wgr) {
        fr_stamp_buf_wgr[STAMP_SLOT(W) + 3] = bar_wait;
        fr_stamp_buf_wgr[STAMP_SLOT(W) + 4] = TSTAMP() - t_loop;
      }
#endif
    } else {
      // Uniform schedule (28x28: 7 phases of 4 rows per image).  Phase k of the workgroup lives in buffer k & 1 and in
      // register set k & 1; a phase moves g rows [r0, r0+4) and input rows [r0-1, r0+5) -- the two halo rows are data of
      // the neighbouring phases here, or zeros at the image border (loaded from a clamped row, replaced on the way to LDS).
      constexpr int PR = C::rows(0), GCH = PR * W * 8, ACH = (PR + 2) * W * 8;
      constexpr int NGS = (GCH + NLT - 1) / NLT, NAS = (ACH + NLT - 1) / NLT;
      struct Set {
        U128 g[NGS], a[NAS];
        int img_row0;  // first output row of the phase (uniform)
      };
      // per-thread constants of the slots
      int goff[NGS], gdst[NGS], aoff_c[NAS], atr[NAS], adst[NAS];
#pragma unroll
      for (int u = 0; u < NGS; ++u) {
        int q = u * NLT + lt;
        q = q < GCH ? q : GCH - 1;
        const int px = q >> 3, r = px / W, c = px - r * W;
        goff[u] = (px * p.ldg + ch * 8) * 2;
        gdst[u] = (r * RW + c) * TSTR + ch * 16;
      }
#pragma unroll
      for (int u = 0; u < NAS; ++u) {
        int q = u * NLT + lt;
        q = q < ACH ? q : ACH - 1;
        const int px = q >> 3, r = px / W, c = px - r * W;
        atr[u] = r;                                  // tile row 0 .. PR+1 <-> image row r0 - 1 + r
        aoff_c[u] = (c * p.lda + ch * 8) * 2;        // + clamped image row * W * lda * 2
        adst[u] = (r * RW + c + 1) * TSTR + ch * 16;
      }
      auto issue = [&](Set& s, int f) {
        const int img = f / NPH, ph = f - img * NPH, r0 = ph * PR;
        s.img_row0 = r0;
        const char* gi = G + ((size_t)img * (W * W) + (size_t)r0 * W) * (size_t)p.ldg * 2;
        const char* xi = X + (size_t)img * (W * W) * (size_t)p.lda * 2;
#pragma unroll
        for (int u = 0; u < NGS; ++u) s.g[u] = ld16(gi + (unsigned)goff[u]);
#pragma unroll
        for (int u = 0; u < NAS; ++u) {
          int row = r0 - 1 + atr[u];
          row = row < 0 ? 0 : (row > W - 1 ? W - 1 : row);
          s.a[u] = ld16(xi + (unsigned)(row * W * p.lda * 2 + aoff_c[u]));
        }
      };
      auto commit = [&](Set& s, char* buf) {
        char* Gs = buf;
        char* As = buf + L::g_bytes(0);
#pragma unroll
        for (int u = 0; u < NGS; ++u) st16(Gs + gdst[u], s.g[u]);
#pragma unroll
        for (int u = 0; u < NAS; ++u) {
          U128 x = s.a[u];
          if (PRO != FR_PRO_NONE) {
            x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
            x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
            x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
            x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
          }
          // slots that can hold a row outside the image (tile row 0 of the first phase, row PR+1 of the last)
          if (u * NLT < W * 8 || (u + 1) * NLT > (PR + 1) * W * 8) {
            const int row = s.img_row0 - 1 + atr[u];
            const bool out = row < 0 || row > W - 1;
            x.x = out ? 0u : x.x;
            x.y = out ? 0u : x.y;
            x.z = out ? 0u : x.z;
            x.w = out ? 0u : x.w;
          }
          st16(As + adst[u], x);
        }
      };
      Set s0, s1;
      const int fl = b_end - 1;  // requests past the run are clamped to its last phase and never read
      issue(s0, b_begin);
      if (p.prev_n) fold_prev(true);
      __builtin_amdgcn_sched_barrier(0);
      issue(s1, b_begin + 1 < b_end ? b_begin + 1 : fl);
      commit(s0, buf0);
      __builtin_amdgcn_sched_barrier(0);
      issue(s0, b_begin + 2 < b_end ? b_begin + 2 : fl);
      __builtin_amdgcn_sched_barrier(0);
      LDS_FENCE_BARRIER();  // B0: phase 0 is in buffer 0
#ifdef FRHIP_STAMPS
      const unsigned long long t_loop = TSTAMP();
      bar_wait = 0;
#endif
#pragma unroll 1
      for (int k = 0; k < nimg; k += 2) {
        const int f = b_begin + k;
        // computing waves: phase k (buffer 0); buffer 1 is free
        commit(s1, buf1);
        __builtin_amdgcn_sched_barrier(0);
        issue(s1, f + 3 < b_end ? f + 3 : fl);
        __builtin_amdgcn_sched_barrier(0);
        LDS_FENCE_BARRIER();  // phase k + 1 is in buffer 1; buffer 0 is free
        commit(s0, buf0);
        __builtin_amdgcn_sched_barrier(0);
        issue(s0, f + 4 < b_end ? f + 4 : fl);
        __builtin_amdgcn_sched_barrier(0);
        LDS_FENCE_BARRIER();  // phase k + 2 is in buffer 0; buffer 1 is free
      }
#ifdef FRHIP_STAMPS
      if (tid == 256 && fr_stamp_buf_wgr) {
        fr_stamp_buf_wgr[STAMP_SLOT(W) + 3] = bar_wait;
        fr_stamp_buf_wgr[STAMP_SLOT(W) + 4] = TSTAMP() - t_loop;
      }
#endif
    }
    return;
  }

  // ------------------------------------------------------------------------------------------------ computing waves
  const int wci = wave;  // ci tile of this wave
  const int li = lane & 15, lq = lane >> 4;
  const int colb = (4 * (li & 3)) * 2;  // byte offset of this lane's 4-channel group inside a 16-channel tile
  const int lrow = (4 * lq + (li >> 2)) * TSTR;
#ifdef FRHIP_STAMPS
  const unsigned long long t_zero = TSTAMP();
#endif
  if (p.prev_n) fold_prev(false);  // before the accumulators exist: the sum may use their registers
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[4][9];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef FRHIP_STAMPS
  const unsigned long long t_fold = TSTAMP();
  if (tid == 0 && fr_stamp_buf_wgr) {
    fr_stamp_buf_wgr[STAMP_SLOT(W) + 8] = t_zero - t_start;   // LDS zero fill
    fr_stamp_buf_wgr[STAMP_SLOT(W) + 9] = t_fold - t_zero;    // slab sum of the previous launch
  }
#endif
  if (nimg > 0) {
    const char* const gb0 = smem + lrow + colb;
    const char* const ab0 = smem + L::g_bytes(0) + lrow + (wci * 16) * 2 + colb;
    const char* const gb1 = smem + L::BUF0 + lrow + colb;
    const char* const ab1 = smem + L::BUF0 + L::g_bytes(1) + lrow + (wci * 16) * 2 + colb;
    // One "item" = one tap of one K step: an input fragment (2 reads) and 4 MFMAs (one per co tile).  Flat software
    // pipeline over the items of a phase: the fragment of item it + LA is requested behind the MFMAs of item it into a ring
    // of RS = LA + 1 = 3 (the slot of an item is its index mod 3 in every phase: 36 and 27 items); the g fragments of the
    // next step (8 reads) are requested over taps 4-7 into a second register set.  Register budget: 144 accumulators + 12
    // ring + 32 g = 194 with addresses, i.e. 112 registers per SIMD stay free beside the two waves of this kernel -- room
    // for BatchNorm waves of the main stream, whose speed beside this kernel decides the step (a 9-deep ring, 222
    // registers, is no faster alone and serialises them; two co tiles per item, 190 registers, costs 70 % more LDS
    // instructions: 1.35 per MFMA by PMC, a quarter of the wave cycles in LDS-issue stalls).
    // The barrier that hands the buffers over sits at the last item of a phase: its fragment was requested two items
    // earlier; the first fragments of the next phase are requested right behind the barrier, under that item's MFMAs.
#ifndef ROLL_LA
#define ROLL_LA 2
#endif
    constexpr int LA = ROLL_LA, RS = LA + 1;  // ring slots
    auto a_frag = [&](const char* ab, int item) -> s16x8 {
      const int ks = item / 9, tap = item % 9;
      const int off = 32 * ks + RW * (tap / 3) + tap % 3;
      return tr_frag(ab + off * TSTR, ab + (off + 16) * TSTR);
    };
    auto g_frag = [&](const char* gb, int ks, int t) -> s16x8 {
      return tr_frag(gb + (32 * ks) * TSTR + t * 32, gb + (32 * ks + 16) * TSTR + t * 32);
    };
    s16x8 ring[RS];
    s16x8 gf[4], gn[4];

    auto run_phase = [&](const char* gb, const char* ab, const char* gb_next, const char* ab_next, auto nstep_c) {
      constexpr int NSTEP = decltype(nstep_c)::value;
      constexpr int NI = NSTEP * 9, IB = NI - LA + 1;
      static_assert(NI % RS == 0, "the ring slot of an item must not depend on the phase");
      static_assert(IB / 9 == NSTEP - 1 && IB % 9 >= 4, "the barrier sits in the last step, not before its g prefetch taps");
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int ks = it / 9, tap = it % 9;
        if (it == IB) {
          LDS_FENCE_BARRIER();  // every fragment of this phase is in registers; the next buffer is complete
#pragma unroll
          for (int t = 0; t < 4; ++t) gn[t] = g_frag(gb_next, 0, t);
#pragma unroll
          for (int jt = NI - LA; jt < IB; ++jt) ring[(jt + LA) % RS] = a_frag(ab_next, jt + LA - NI);  // deferred requests
          __builtin_amdgcn_sched_barrier(0);
        }
        const s16x8 af = ring[it % RS];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[t], af, acc[t][tap], 0, 0, 0);
        if (it + LA < NI) ring[(it + LA) % RS] = a_frag(ab, it + LA);
        else if (it >= IB) ring[(it + LA) % RS] = a_frag(ab_next, it + LA - NI);
        if (ks + 1 < NSTEP && tap >= 4 && tap < 8) gn[tap - 4] = g_frag(gb, ks + 1, tap - 4);
        if (tap == 8) {
#pragma unroll
          for (int t = 0; t < 4; ++t) gf[t] = gn[t];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    constexpr int NS0 = C::krows(0) * RW / 32, NS1 = C::krows(NPH == 2 ? 1 : 0) * RW / 32;
    LDS_FENCE_BARRIER();  // B0: phase 0 of the first image is in buffer 0
#ifdef FRHIP_STAMPS
    const unsigned long long t_loop = TSTAMP();
    bar_wait = 0;
    if (tid == 0 && fr_stamp_buf_wgr)
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 10] = t_loop - t_fold;  // wait for the first tiles
#endif
#pragma unroll
    for (int t = 0; t < 4; ++t) gf[t] = g_frag(gb0, 0, t);
#pragma unroll
    for (int it = 0; it < LA; ++it) ring[it] = a_frag(ab0, it);
    if constexpr (NPH == 2) {
#pragma unroll 1
      for (int i = 0; i < nimg; ++i) {
        // behind the last image the "next phase" fragments are read from whatever buffer 0 holds and never used
        run_phase(gb0, ab0, gb1, ab1, std::integral_constant<int, NS0>{});
        run_phase(gb1, ab1, gb0, ab0, std::integral_constant<int, NS1>{});
      }
    } else {
      // uniform schedule: the data-moving waves run their loop two phases at a time; an odd run leaves one barrier over
#pragma unroll 1
      for (int k = 0; k < nimg; k += 2) {
        run_phase(gb0, ab0, gb1, ab1, std::integral_constant<int, NS0>{});
        if (k + 1 < nimg) run_phase(gb1, ab1, gb0, ab0, std::integral_constant<int, NS0>{});
        else LDS_FENCE_BARRIER();
      }
    }
#ifdef FRHIP_STAMPS
    if (tid == 0 && fr_stamp_buf_wgr) {
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 1] = TSTAMP() - t_loop;
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 2] = bar_wait;
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 6] = nimg;
    }
#endif
  }

  // slab[group][co][tap][ci]
  float* __restrict__ slab = p.slab + (size_t)group * (size_t)p.Cout * 9 * (size_t)p.SC;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * CT + t * 16 + lq * 4 + r;
        const int ci = cit * CT + wci * 16 + li;
        slab[((size_t)co * 9 + tap) * (size_t)p.SC + ci] = acc[t][tap][r];
      }
#ifdef FRHIP_STAMPS
  if (tid == 0 && fr_stamp_buf_wgr) {
    fr_stamp_buf_wgr[STAMP_SLOT(W) + 0] = TSTAMP() - t_start;
    fr_stamp_buf_wgr[STAMP_SLOT(W) + 5] = __builtin_amdgcn_s_memrealtime() - rt_start;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// 56x56 / 112x112 (the 64-channel stages: ONE or two dW tiles, 128 - 256 groups): "virtual rows".  A workgroup walks a run
// of image rows; every input row is fetched ONCE into a ring of 4 LDS rows (the strip kernel re-fetched the halo rows of
// every strip: 1.5 - 2 x the input bytes of layers that are HBM-bound to begin with).  An image is VH = W + 2 virtual rows:
// its W rows between a zero row above and below; phase V (one virtual row = RW / 32 K steps) multiplies the g row of
// virtual row V with the input rows V-1, V, V+1.  The two halo rows of an image are phases too -- with a g row of zeros
// (2 / (W + 2) of the arithmetic wasted) -- which keeps the schedule uniform: one new input row and one g row per phase,
// whatever the position in the image, and a run may start and end anywhere.  Ring slots are relative to the run's first
// row, so four phases unrolled give compile-time LDS addresses again.  Same roles, barriers, fragment pipeline and
// register budget as the kernel above.
template <int W_>
struct VC {
  static constexpr int W = W_, RW = W_ <= 64 ? 64 : 128, VH = W_ + 2, NSTEP = RW / 32;
  static constexpr int AROW = (RW + 16) * TSTR;  // positions 0 .. RW+15: column c sits at position c + 1
  static constexpr int GROW = RW * TSTR;
  static constexpr int A_OFF = 2 * GROW;         // [g row ring of 2][input row ring of 4]
  static constexpr int LDS = A_OFF + 4 * AROW;
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int W, int PRO>
__global__ __launch_bounds__(512, 2) void conv_wgrad_vr_kernel(const FrWgradArgs p) {
  using C = VC<W>;
  constexpr int RW = C::RW, VH = C::VH, NSTEP = C::NSTEP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cit_n = p.SC / CT, tiles = (p.Cout / CT) * cit_n;
  const int group = bid / tiles, tile = bid - group * tiles;
  const int cot = tile / cit_n, cit = tile - cot * cit_n;
  const int total = p.B * VH;  // virtual rows of the launch
  const int per = (total + p.nsplit - 1) / p.nsplit;
  const int v_begin = group * per;
  int v_end = v_begin + per;
  if (v_end > total) v_end = total;
  const int nph = v_end > v_begin ? v_end - v_begin : 0;  // phases of this workgroup

#ifdef FRHIP_STAMPS
  unsigned long long bar_wait = 0;
  const unsigned long long t_start = TSTAMP(), rt_start = __builtin_amdgcn_s_memrealtime();
#endif
  for (int idx = tid; idx < C::LDS / 16; idx += 512) st16(smem + idx * 16, zero16());
  __syncthreads();

  // deferred slab sum of the previous launch, split between the wave roles (see conv_wgrad_roll_kernel)
  auto fold_prev = [&](bool movers) { fold_prev_share(p, movers, tid); };

#ifdef ROLL_ALL_PRIO  // experiment: the weight-gradient waves win every issue arbitration against co-resident channel-wise kernels
  __builtin_amdgcn_s_setprio(ROLL_ALL_PRIO);
#endif
  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------- data-moving waves
#ifdef ROLL_LOADER_PRIO
    __builtin_amdgcn_s_setprio(ROLL_LOADER_PRIO);
#endif
    if (nph == 0) {
      if (p.prev_n) fold_prev(true);
      return;
    }
    const int lt = tid - 256;
    const int ch = lt & 7;
    const char* __restrict__ G = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.g) + cot * CT);
    const char* __restrict__ X = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.src) + cit * CT);
    float pa[8], pb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pa[j] = PRO != FR_PRO_NONE ? p.pro_a[cit * CT + ch * 8 + j] : 0.f;
      pb[j] = PRO == FR_PRO_BN ? p.pro_b[cit * CT + ch * 8 + j] : 0.f;
    }
    constexpr int RCH = W * 8;                          // 16-byte chunks of one row
    constexpr int NRS = (RCH + NLT - 1) / NLT;          // slots per row
    int goff[NRS], xoff[NRS], gdst[NRS], adst[NRS];
#pragma unroll
    for (int u = 0; u < NRS; ++u) {
      int q = u * NLT + lt;
      q = q < RCH ? q : RCH - 1;  // lanes past the row repeat its last chunk (same bytes, same address)
      const int c = q >> 3;
      goff[u] = (c * p.ldg + ch * 8) * 2;
      xoff[u] = (c * p.lda + ch * 8) * 2;
      gdst[u] = c * TSTR + ch * 16;
      adst[u] = (c + 1) * TSTR + ch * 16;
    }
    struct Row {
      U128 v[NRS];
      bool zero;  // uniform: a halo row, or a row outside the tensor
    };
    // virtual row V (any int): image row or zero row
    auto row_of = [&](int V, int& img, int& r) -> bool {
      if (V < 0 || V >= total) return false;
      img = V / VH;
      r = V - img * VH - 1;
      return r >= 0 && r < W;
    };
    auto issue_a = [&](Row& s, int V) {
      int img = 0, r = 0;
      const bool ok = row_of(V, img, r);
      s.zero = !ok;
      const char* xi = X + ((size_t)(ok ? img : 0) * (W * W) + (size_t)(ok ? r : 0) * W) * (size_t)p.lda * 2;
#pragma unroll
      for (int u = 0; u < NRS; ++u) s.v[u] = ld16(xi + (unsigned)xoff[u]);
    };
    auto issue_g = [&](Row& s, int V) {
      int img = 0, r = 0;
      const bool ok = row_of(V, img, r);
      s.zero = !ok;
      const char* gi = G + ((size_t)(ok ? img : 0) * (W * W) + (size_t)(ok ? r : 0) * W) * (size_t)p.ldg * 2;
#pragma unroll
      for (int u = 0; u < NRS; ++u) s.v[u] = ld16(gi + (unsigned)goff[u]);
    };
    auto commit_a = [&](Row& s, int slot) {
      char* As = smem + C::A_OFF + slot * C::AROW;
#pragma unroll
      for (int u = 0; u < NRS; ++u) {
        U128 x = s.v[u];
        if (PRO != FR_PRO_NONE) {
          x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
          x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
          x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
          x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
        }
        if (s.zero) x = zero16();
        st16(As + adst[u], x);
      }
    };
    auto commit_g = [&](Row& s, int slot) {
      char* Gs = smem + slot * C::GROW;
#pragma unroll
      for (int u = 0; u < NRS; ++u) st16(Gs + gdst[u], s.zero ? zero16() : s.v[u]);
    };
    // phase j (relative to the run) = virtual row v_begin + j: g slot j & 1, input rows j-1, j, j+1 in slots j, j+1, j+2 (mod 4)
    {
      // the three input rows and the g row of phase 0 in ONE round trip, the slab sum's loads queued behind them
      Row t0, t1, t2, tg;
      issue_a(t0, v_begin - 1);
      issue_a(t1, v_begin);
      issue_a(t2, v_begin + 1);
      issue_g(tg, v_begin);
      if (p.prev_n) fold_prev(true);
      __builtin_amdgcn_sched_barrier(0);
      commit_a(t0, 0);
      commit_a(t1, 1);
      commit_a(t2, 2);
      commit_g(tg, 0);
    }
    Row a0, g0, a1, g1;  // set 0: rows for the commits of even phases, set 1: odd phases
    issue_a(a0, v_begin + 2);
    issue_g(g0, v_begin + 1);
    issue_a(a1, v_begin + 3);
    issue_g(g1, v_begin + 2);
    LDS_FENCE_BARRIER();  // B0: phase 0 is resident
#ifdef FRHIP_STAMPS
    const unsigned long long t_loop = TSTAMP();
    bar_wait = 0;
#endif
#pragma unroll 1
    for (int j = 0; j < nph; j += 4) {
      const int V = v_begin + j;
      // computing waves: phase j.  Row j + 2 goes to slot (j + 3) & 3, the g row of phase j + 1 to slot (j + 1) & 1; then
      // the same register set requests the rows of two phases later.  (Past the run: rows nobody reads.)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Row& ra = (s & 1) ? a1 : a0;
        Row& rg = (s & 1) ? g1 : g0;
        commit_a(ra, (s + 3) & 3);
        commit_g(rg, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        issue_a(ra, V + s + 4);
        issue_g(rg, V + s + 3);
        __builtin_amdgcn_sched_barrier(0);
        LDS_FENCE_BARRIER();
      }
    }
#ifdef FRHIP_STAMPS
    if (tid == 256 && fr_stamp_buf_wgr) {
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 3] = bar_wait;
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 4] = TSTAMP() - t_loop;
    }
#endif
    return;
  }

  // ------------------------------------------------------------------------------------------------ computing waves
  const int wci = wave;
  const int li = lane & 15, lq = lane >> 4;
  const int colb = (4 * (li & 3)) * 2;
  const int lrow = (4 * lq + (li >> 2)) * TSTR;

  if (p.prev_n) fold_prev(false);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[4][9];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (nph > 0) {
    const char* const gbase = smem + lrow + colb;
    const char* const abase = smem + C::A_OFF + lrow + (wci * 16) * 2 + colb;
    constexpr int LA = 2, RS = 3;  // tap-major items, ring of 3: see the kernel above
    constexpr int NI = NSTEP * 9, IB = NI - LA + 1;
    static_assert(NI % RS == 0 && IB / 9 == NSTEP - 1 && IB % 9 >= 4, "item pipeline");
    // input fragment of item (step ks = it / 9, tap): row slot of kh, position 32 ks + kw
    auto a_frag = [&](int slot0, int item) -> s16x8 {  // slot0 = ring slot of the phase's row V - 1
      const int ks = item / 9, tap = item % 9;
      const char* ab = abase + ((slot0 + tap / 3) & 3) * C::AROW + (32 * ks + tap % 3) * TSTR;
      return tr_frag(ab, ab + 16 * TSTR);
    };
    auto g_frag = [&](int gslot, int ks, int t) -> s16x8 {
      const char* gb = gbase + gslot * C::GROW + (32 * ks) * TSTR + t * 32;
      return tr_frag(gb, gb + 16 * TSTR);
    };
    s16x8 ring[RS];
    s16x8 gf[4], gn[4];
    // phase with relative index s (mod 4): g slot s & 1, input rows in slots s, s+1, s+2; `live` = false past the run:
    // only the barrier (the data-moving waves run their loop in fours)
    auto run_phase = [&](auto sc, bool live) {
      constexpr int s = decltype(sc)::value;
      if (!live) {
        LDS_FENCE_BARRIER();
        return;
      }
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int ks = it / 9, tap = it % 9;
        if (it == IB) {
          LDS_FENCE_BARRIER();
#pragma unroll
          for (int t = 0; t < 4; ++t) gn[t] = g_frag((s + 1) & 1, 0, t);
#pragma unroll
          for (int jt = NI - LA; jt < IB; ++jt) ring[(jt + LA) % RS] = a_frag((s + 1) & 3, jt + LA - NI);
          __builtin_amdgcn_sched_barrier(0);
        }
        const s16x8 af = ring[it % RS];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[t], af, acc[t][tap], 0, 0, 0);
        if (it + LA < NI) ring[(it + LA) % RS] = a_frag(s, it + LA);
        else if (it >= IB) ring[(it + LA) % RS] = a_frag((s + 1) & 3, it + LA - NI);
        if (ks + 1 < NSTEP && tap >= 4 && tap < 8) gn[tap - 4] = g_frag(s & 1, ks + 1, tap - 4);
        if (tap == 8) {
#pragma unroll
          for (int t = 0; t < 4; ++t) gf[t] = gn[t];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    LDS_FENCE_BARRIER();  // B0
#ifdef FRHIP_STAMPS
    const unsigned long long t_loop = TSTAMP();
    bar_wait = 0;
#endif
#pragma unroll
    for (int t = 0; t < 4; ++t) gf[t] = g_frag(0, 0, t);
#pragma unroll
    for (int it = 0; it < LA; ++it) ring[it] = a_frag(0, it);
#pragma unroll 1
    for (int j = 0; j < nph; j += 4) {
      run_phase(std::integral_constant<int, 0>{}, true);
      run_phase(std::integral_constant<int, 1>{}, j + 1 < nph);
      run_phase(std::integral_constant<int, 2>{}, j + 2 < nph);
      run_phase(std::integral_constant<int, 3>{}, j + 3 < nph);
    }
#ifdef FRHIP_STAMPS
    if (tid == 0 && fr_stamp_buf_wgr) {
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 1] = TSTAMP() - t_loop;
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 2] = bar_wait;
      fr_stamp_buf_wgr[STAMP_SLOT(W) + 6] = nph;
    }
#endif
  }

  float* __restrict__ slab = p.slab + (size_t)group * (size_t)p.Cout * 9 * (size_t)p.SC;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * CT + t * 16 + lq * 4 + r;
        const int ci = cit * CT + wci * 16 + li;
        slab[((size_t)co * 9 + tap) * (size_t)p.SC + ci] = acc[t][tap][r];
      }
#ifdef FRHIP_STAMPS
  if (tid == 0 && fr_stamp_buf_wgr) {
    fr_stamp_buf_wgr[STAMP_SLOT(W) + 0] = TSTAMP() - t_start;
    fr_stamp_buf_wgr[STAMP_SLOT(W) + 5] = __builtin_amdgcn_s_memrealtime() - rt_start;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride 2 (the second convolution of the first unit of a stage, model_irse.py:59; 128 / 256 / 512 channels):
//   dw[co][kh][kw][ci] = sum over low-res pixels (i, j) of g[i][j][co] * pro(x[2i + kh - 1][2j + kw - 1][ci])
// The K axis is the LOW-resolution pixel, in rows of RW slots as above.  The data-moving waves split the high-res input
// rows of a phase into the four parity planes x[2r + a][2c + b] on their way into LDS; inside a plane tap (kh, kw) is again
// K slot + immediate: plane (a, b) = (kh != 1, kw != 1), one plane row further for kh = 2, one position further for kw >= 1
// (position 0 of a plane row is plane column -1: the zero halo of the odd-column planes).  A phase is two K steps (PRK
// low-res rows): 2 rows at 28x28, 4 at 14x14 (16 virtual rows per image, the last two all zero), the whole image + one
// zero row at 7x7 -- the four planes of more rows do not fit twice into LDS.  Per K step the data-moving waves carry 4x the
// input bytes of a stride-1 layer (9 of their 11 slots per phase are input chunks, each with the BN / PReLU prologue), so
// this kernel is paced by them, not by the MFMA waves; same roles, barriers, fragment pipeline, accumulator layout, slab
// format and deferred slab sum as conv_wgrad_roll_kernel.
template <int WL_>
struct SC2;
template <>
struct SC2<56> {  // 64 channels, ONE dW tile: 256 groups of 56 phases; a phase re-reads the last input row of the previous one
  static constexpr int WL = 56, RW = 64, PRK = 1, NPH = 56;
};
template <>
struct SC2<28> {
  static constexpr int WL = 28, RW = 32, PRK = 2, NPH = 14;
};
template <>
struct SC2<14> {
  static constexpr int WL = 14, RW = 16, PRK = 4, NPH = 4;
};
template <>
struct SC2<7> {
  static constexpr int WL = 7, RW = 8, PRK = 8, NPH = 1;
};

template <int WL>
struct SL2 {
  using C = SC2<WL>;
  static constexpr int G_BYTES = C::PRK * C::RW * TSTR;
  // tail: the positions past the last row a shifted fragment read touches (one at most; 16 kept where LDS allows)
  static constexpr int TAIL = C::RW >= 64 ? 2 : 16;
  static constexpr int P1 = ((C::PRK + 1) * C::RW + TAIL) * TSTR;  // odd high-res rows (a = 1): plane rows r0-1 .. r0+PRK-1
  static constexpr int P0 = (C::PRK * C::RW + TAIL) * TSTR;        // even rows (a = 0): plane rows r0 .. r0+PRK-1
  static constexpr int plane_off(int a, int b) { return G_BYTES + (a ? (b ? 0 : P1) : 2 * P1 + (b ? 0 : P0)); }
  static constexpr int BUF = G_BYTES + 2 * P1 + 2 * P0;
  static constexpr int LDS = 2 * BUF;
  // tile offset (bytes) of tap (kh, kw) relative to the K slot
  static constexpr int tap_off(int tap) {
    const int kh = tap / 3, kw = tap % 3;
    return plane_off(kh != 1, kw != 1) + ((kh == 2 ? C::RW : 0) + (kw == 0 ? 0 : 1)) * TSTR;
  }
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int WL, int PRO>
__global__ __launch_bounds__(512, 2) void conv_wgrad_s2roll_kernel(const FrWgradArgs p) {
  using C = SC2<WL>;
  using L = SL2<WL>;
  constexpr int RW = C::RW, NPH = C::NPH, PRK = C::PRK, WH = 2 * WL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cit_n = p.SC / CT, tiles = (p.Cout / CT) * cit_n;
  const int group = bid / tiles, tile = bid - group * tiles;
  const int cot = tile / cit_n, cit = tile - cot * cit_n;
  const int units = p.B * NPH;  // phases of the launch
  const int per = (units + p.nsplit - 1) / p.nsplit;
  const int b_begin = group * per;
  int b_end = b_begin + per;
  if (b_end > units) b_end = units;
  const int nph = b_end > b_begin ? b_end - b_begin : 0;  // phases of this workgroup

  for (int idx = tid; idx < L::LDS / 16; idx += 512) st16(smem + idx * 16, zero16());
  __syncthreads();

  auto fold_prev = [&](bool movers) { fold_prev_share(p, movers, tid); };

#ifdef ROLL_ALL_PRIO  // experiment: the weight-gradient waves win every issue arbitration against co-resident channel-wise kernels
  __builtin_amdgcn_s_setprio(ROLL_ALL_PRIO);
#endif
  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------- data-moving waves
    if (nph == 0) {
      if (p.prev_n) fold_prev(true);
      return;
    }
    const int lt = tid - 256;
    const int ch = lt & 7;
    const char* __restrict__ G = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.g) + cot * CT);
    const char* __restrict__ X = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(p.src) + cit * CT);
    float pa[8], pb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pa[j] = PRO != FR_PRO_NONE ? p.pro_a[cit * CT + ch * 8 + j] : 0.f;
      pb[j] = PRO == FR_PRO_BN ? p.pro_b[cit * CT + ch * 8 + j] : 0.f;
    }
    // a phase moves g rows [r0, r0 + PRK) and high-res input rows [2 r0 - 1, 2 (r0 + PRK) - 1]; rows outside the image
    // (above the first, below the last, the all-zero virtual rows) are read from a clamped row and replaced by zeros
    constexpr int GCH = PRK * WL * 8, ACH = (2 * PRK + 1) * WH * 8;
    constexpr int NGS = (GCH + NLT - 1) / NLT, NAS = (ACH + NLT - 1) / NLT;
    struct Set {
      U128 g[NGS], a[NAS];
      int r0;  // first low-res row of the phase (uniform)
    };
    int gcol[NGS], grow[NGS], gdst[NGS], acol[NAS], ae[NAS], adst[NAS];
#pragma unroll
    for (int u = 0; u < NGS; ++u) {
      int q = u * NLT + lt;
      q = q < GCH ? q : GCH - 1;
      const int px = q >> 3, r = px / WL, c = px - r * WL;
      grow[u] = r;
      gcol[u] = (c * p.ldg + ch * 8) * 2;
      gdst[u] = (r * RW + c) * TSTR + ch * 16;
    }
#pragma unroll
    for (int u = 0; u < NAS; ++u) {
      int q = u * NLT + lt;
      q = q < ACH ? q : ACH - 1;
      const int px = q >> 3, e = px / WH, hc = px - e * WH;  // e: high-res row 2 r0 - 1 + e
      ae[u] = e;
      acol[u] = (hc * p.lda + ch * 8) * 2;
      const int a = (e & 1) ? 0 : 1, b = hc & 1;
      const int trow = a ? (e >> 1) : ((e - 1) >> 1);
      adst[u] = L::plane_off(a, b) + (trow * RW + (hc >> 1) + 1) * TSTR + ch * 16;
    }
    auto issue = [&](Set& s, int f) {
      const int img = f / NPH, ph = f - img * NPH, r0 = ph * PRK;
      s.r0 = r0;
      const char* gi = G + (size_t)img * (WL * WL) * (size_t)p.ldg * 2;
      const char* xi = X + (size_t)img * (WH * WH) * (size_t)p.lda * 2;
#pragma unroll
      for (int u = 0; u < NGS; ++u) {
        int row = r0 + grow[u];
        row = row > WL - 1 ? WL - 1 : row;
        s.g[u] = ld16(gi + (unsigned)(row * WL * p.ldg * 2 + gcol[u]));
      }
#pragma unroll
      for (int u = 0; u < NAS; ++u) {
        int row = 2 * r0 - 1 + ae[u];
        row = row < 0 ? 0 : (row > WH - 1 ? WH - 1 : row);
        s.a[u] = ld16(xi + (unsigned)(row * WH * p.lda * 2 + acol[u]));
      }
    };
    auto commit = [&](Set& s, char* buf) {
#pragma unroll
      for (int u = 0; u < NGS; ++u) {
        U128 x = s.g[u];
        if (WL % PRK != 0 || NPH * PRK > WL) {  // virtual rows below the image
          const bool out = s.r0 + grow[u] > WL - 1;
          x.x = out ? 0u : x.x;
          x.y = out ? 0u : x.y;
          x.z = out ? 0u : x.z;
          x.w = out ? 0u : x.w;
        }
        st16(buf + gdst[u], x);
      }
#pragma unroll
      for (int u = 0; u < NAS; ++u) {
        U128 x = s.a[u];
        if (PRO != FR_PRO_NONE) {
          x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
          x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
          x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
          x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
        }
        {
          const int row = 2 * s.r0 - 1 + ae[u];
          const bool out = row < 0 || row > WH - 1;
          x.x = out ? 0u : x.x;
          x.y = out ? 0u : x.y;
          x.z = out ? 0u : x.z;
          x.w = out ? 0u : x.w;
        }
        st16(buf + adst[u], x);
      }
    };
    char* const buf0 = smem;
    char* const buf1 = smem + L::BUF;
    Set s0, s1;
    const int fl = b_end - 1;  // requests past the run are clamped to its last phase and never read
    issue(s0, b_begin);
    if (p.prev_n) fold_prev(true);
    __builtin_amdgcn_sched_barrier(0);
    issue(s1, b_begin + 1 < b_end ? b_begin + 1 : fl);
    commit(s0, buf0);
    __builtin_amdgcn_sched_barrier(0);
    issue(s0, b_begin + 2 < b_end ? b_begin + 2 : fl);
    __builtin_amdgcn_sched_barrier(0);
    LDS_FENCE_BARRIER_RAW();  // B0: phase 0 is in buffer 0
#pragma unroll 1
    for (int k = 0; k < nph; k += 2) {
      const int f = b_begin + k;
      commit(s1, buf1);
      __builtin_amdgcn_sched_barrier(0);
      issue(s1, f + 3 < b_end ? f + 3 : fl);
      __builtin_amdgcn_sched_barrier(0);
      LDS_FENCE_BARRIER_RAW();  // phase k + 1 is in buffer 1; buffer 0 is free
      commit(s0, buf0);
      __builtin_amdgcn_sched_barrier(0);
      issue(s0, f + 4 < b_end ? f + 4 : fl);
      __builtin_amdgcn_sched_barrier(0);
      LDS_FENCE_BARRIER_RAW();  // phase k + 2 is in buffer 0; buffer 1 is free
    }
    return;
  }

  // ------------------------------------------------------------------------------------------------ computing waves
  const int wci = wave;
  const int li = lane & 15, lq = lane >> 4;
  const int colb = (4 * (li & 3)) * 2;
  const int lrow = (4 * lq + (li >> 2)) * TSTR;
  if (p.prev_n) fold_prev(false);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[4][9];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (nph > 0) {
    const char* const gb0 = smem + lrow + colb;
    const char* const ab0 = smem + lrow + (wci * 16) * 2 + colb;  // + tap_off (which includes the g tile and the plane)
    const char* const gb1 = gb0 + L::BUF;
    const char* const ab1 = ab0 + L::BUF;
    constexpr int LA = 2, RS = 3, NSTEP = PRK * RW / 32, NI = NSTEP * 9, IB = NI - LA + 1;
    static_assert(NI % RS == 0 && IB / 9 == NSTEP - 1 && IB % 9 >= 4, "phase shape (see conv_wgrad_roll_kernel)");
    auto a_frag = [&](const char* ab, int item) -> s16x8 {
      const int ks = item / 9, tap = item % 9;
      const int off = L::tap_off(tap) + 32 * ks * TSTR;
      return tr_frag(ab + off, ab + off + 16 * TSTR);
    };
    auto g_frag = [&](const char* gb, int ks, int t) -> s16x8 {
      return tr_frag(gb + (32 * ks) * TSTR + t * 32, gb + (32 * ks + 16) * TSTR + t * 32);
    };
    s16x8 ring[RS];
    s16x8 gf[4], gn[4];
    auto run_phase = [&](const char* gb, const char* ab, const char* gb_next, const char* ab_next) {
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int ks = it / 9, tap = it % 9;
        if (it == IB) {
          LDS_FENCE_BARRIER_RAW();  // every fragment of this phase is in registers; the next buffer is complete
#pragma unroll
          for (int t = 0; t < 4; ++t) gn[t] = g_frag(gb_next, 0, t);
#pragma unroll
          for (int jt = NI - LA; jt < IB; ++jt) ring[(jt + LA) % RS] = a_frag(ab_next, jt + LA - NI);
          __builtin_amdgcn_sched_barrier(0);
        }
        const s16x8 af = ring[it % RS];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[t], af, acc[t][tap], 0, 0, 0);
        if (it + LA < NI) ring[(it + LA) % RS] = a_frag(ab, it + LA);
        else if (it >= IB) ring[(it + LA) % RS] = a_frag(ab_next, it + LA - NI);
        if (ks + 1 < NSTEP && tap >= 4 && tap < 8) gn[tap - 4] = g_frag(gb, ks + 1, tap - 4);
        if (tap == 8) {
#pragma unroll
          for (int t = 0; t < 4; ++t) gf[t] = gn[t];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    LDS_FENCE_BARRIER_RAW();  // B0
#pragma unroll
    for (int t = 0; t < 4; ++t) gf[t] = g_frag(gb0, 0, t);
#pragma unroll
    for (int it = 0; it < LA; ++it) ring[it] = a_frag(ab0, it);
#pragma unroll 1
    for (int k = 0; k < nph; k += 2) {
      run_phase(gb0, ab0, gb1, ab1);
      if (k + 1 < nph) run_phase(gb1, ab1, gb0, ab0);
      else LDS_FENCE_BARRIER_RAW();
    }
  }

  float* __restrict__ slab = p.slab + (size_t)group * (size_t)p.Cout * 9 * (size_t)p.SC;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * CT + t * 16 + lq * 4 + r;
        const int ci = cit * CT + wci * 16 + li;
        slab[((size_t)co * 9 + tap) * (size_t)p.SC + ci] = acc[t][tap][r];
      }
}

template <int WL, int PRO>
int launch_s2(const FrWgradArgs& a, hipStream_t st) {
  using L = SL2<WL>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_s2roll_kernel<WL, PRO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS);
    fr_attr_done(attr_done);
  }
  const int tiles = (a.Cout / CT) * (a.SC / CT);
  hipLaunchKernelGGL((conv_wgrad_s2roll_kernel<WL, PRO>), dim3(tiles * a.nsplit), dim3(512), L::LDS, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fr_set_error(hipGetErrorString(e));
    return (int)e;
  }
  if (a.defer) return 0;
  return fr_launch_reduce_slabs(a.slab, a.nsplit, (long long)a.Cout * 9 * a.SC, a.dw, st);
}

template <int WL>
int by_pro_s2(const FrWgradArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return launch_s2<WL, FR_PRO_NONE>(a, st);
    case FR_PRO_BN: return launch_s2<WL, FR_PRO_BN>(a, st);
    case FR_PRO_PRELU: return launch_s2<WL, FR_PRO_PRELU>(a, st);
  }
  FR_UNSUPPORTED("fr_conv_wgrad_strip: unknown prologue");
}

template <int W, int PRO>
int launch_vr(const FrWgradArgs& a, hipStream_t st) {
  using C = VC<W>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_vr_kernel<W, PRO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    fr_attr_done(attr_done);
  }
  const int tiles = (a.Cout / CT) * (a.SC / CT);
  hipLaunchKernelGGL((conv_wgrad_vr_kernel<W, PRO>), dim3(tiles * a.nsplit), dim3(512), C::LDS, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fr_set_error(hipGetErrorString(e));
    return (int)e;
  }
  if (a.defer) return 0;
  return fr_launch_reduce_slabs(a.slab, a.nsplit, (long long)a.Cout * 9 * a.SC, a.dw, st);
}

template <int W>
int by_pro_vr(const FrWgradArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return launch_vr<W, FR_PRO_NONE>(a, st);
    case FR_PRO_BN: return launch_vr<W, FR_PRO_BN>(a, st);
    case FR_PRO_PRELU: return launch_vr<W, FR_PRO_PRELU>(a, st);
  }
  FR_UNSUPPORTED("fr_conv_wgrad_strip: unknown prologue");
}

template <int W, int PRO>
int launch(const FrWgradArgs& a, hipStream_t st) {
  using L = RL<W>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_roll_kernel<W, PRO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS);
    fr_attr_done(attr_done);
  }
  const int tiles = (a.Cout / CT) * (a.SC / CT);
  hipLaunchKernelGGL((conv_wgrad_roll_kernel<W, PRO>), dim3(tiles * a.nsplit), dim3(512), L::LDS, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fr_set_error(hipGetErrorString(e));
    return (int)e;
  }
  if (a.defer) return 0;  // the caller sums the slabs (prev_* of a later launch, or fr_reduce_slabs)
  return fr_launch_reduce_slabs(a.slab, a.nsplit, (long long)a.Cout * 9 * a.SC, a.dw, st);
}

template <int W>
int by_pro(const FrWgradArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return launch<W, FR_PRO_NONE>(a, st);
    case FR_PRO_BN: return launch<W, FR_PRO_BN>(a, st);
    case FR_PRO_PRELU: return launch<W, FR_PRO_PRELU>(a, st);
  }
  FR_UNSUPPORTED("fr_conv_wgrad_strip: unknown prologue");
}

}  // namespace

#ifdef FRHIP_STAMPS
extern "C" int fr_debug_set_stamp_buffer_wgr(void* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fr_stamp_buf_wgr), &dev_ptr, sizeof(dev_ptr));
}
#endif

// FRHIP_WGRAD_ROLL=0: back to the strip kernel for every shape (A/B switch; rounds 3-4 also had one switch per shape family:
// profiles/r03_switch_matrix.txt, r04_switch_matrix.txt)
bool fr_wgrad_roll_enabled() {
  static const int* on = fr_option_slot("FRHIP_WGRAD_ROLL", 1);
  return *on != 0;
}

// stride-1 3x3 at 7x7 / 14x14 / 28x28 / 56x56 / 112x112, channel counts multiples of 64, at least one image (14x14) / phase (28x28) per group
bool fr_wgrad_roll_serves(const FrWgradArgs& a) {
  if (!(fr_wgrad_roll_enabled() && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.GH == a.SH &&
        a.GW == a.SW && a.SH == a.SW && a.Cout % CT == 0 && a.SC % CT == 0 && a.nsplit >= 1))
    return false;
  if (a.SW == 14) return a.nsplit <= a.B;
  if (a.SW == 28) return a.nsplit <= a.B * RC<28>::NPH;
  if (a.SW == 7) return a.nsplit <= a.B;
  if (a.SW == 56 || a.SW == 112) return a.nsplit <= a.B * (a.SW + 2);
  return false;
}

// stride-2 3x3 with a 28 / 14 / 7 wide gradient, channel counts multiples of 64, at least one phase per group
bool fr_wgrad_s2roll_serves(const FrWgradArgs& a) {
  if (!(fr_wgrad_roll_enabled() && a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.GH == a.GW &&
        a.SH == 2 * a.GH && a.SW == 2 * a.GW && a.Cout % CT == 0 && a.SC % CT == 0 && a.nsplit >= 1))
    return false;
  if (a.GW == 56) return a.nsplit <= a.B * SC2<56>::NPH;
  if (a.GW == 28) return a.nsplit <= a.B * SC2<28>::NPH;
  if (a.GW == 14) return a.nsplit <= a.B * SC2<14>::NPH;
  if (a.GW == 7) return a.nsplit <= a.B * SC2<7>::NPH;
  return false;
}

int fr_wgrad_s2roll_launch(const FrWgradArgs& a, hipStream_t st) {
  switch (a.GW) {
    case 56: return by_pro_s2<56>(a, st);
    case 28: return by_pro_s2<28>(a, st);
    case 14: return by_pro_s2<14>(a, st);
    case 7: return by_pro_s2<7>(a, st);
  }
  FR_UNSUPPORTED("fr_conv_wgrad_strip: stride-2 width not served by the warp-specialised kernel");
}

int fr_wgrad_roll_launch(const FrWgradArgs& a, hipStream_t st) {
  switch (a.SW) {
    case 7: return by_pro<7>(a, st);
    case 14: return by_pro<14>(a, st);
    case 28: return by_pro<28>(a, st);
    case 56: return by_pro_vr<56>(a, st);
    case 112: return by_pro_vr<112>(a, st);
  }
  FR_UNSUPPORTED("fr_conv_wgrad_strip: width not served by the warp-specialised kernel");
}
