// frhip -- the stride-2 3x3 convolution of the first unit of every stage (bottleneck_IR conv2 with stride 2,
// backbone/model_irse.py:57-59) and its data gradient, as LDS-resident strips (bf16, gfx950).
//
// Both directions are written on the LOW-resolution grid (HL x WL = the output of the forward convolution).  A
// stride-2 window touches the four parity planes x[2i+ph][2j+pw] of the high-resolution image with 1, 2, 2 and 4 of
// its 9 taps, and inside one plane those taps are plain neighbour offsets (0 / -1 rows and columns).  So
//   forward (KIND 0):   y[i][j] = sum over the 4 planes of a small stride-1 "convolution" of that plane -- a workgroup
//                       owning ROWS low-res rows of one image stages one plane at a time in LDS (prologue applied once)
//                       and accumulates all four into the same MFMA accumulators;
//   data gradient (KIND 1): dx[2i+ph][2j+pw] = the same kind of sum over the low-res gradient g (offsets 0 / +1) with
//                       the taps of class (ph, pw) -- the g strip is staged once and the four output classes are
//                       produced one after the other, each with its own fused epilogue (PReLU backward of
//                       model_irse.py:58 / BN-backward sums) and 9/4 taps per pixel instead of 9.
// Everything else follows conv3x3_strip.hip: A fragments are single ds_read_b128 at (row register + immediate tap
// offset) out of a conflict-free padded [pixel][channel] image, weights stream L2 -> registers (2-deep ring, requested
// two tap-steps ahead), no barrier inside a tap list, bf16 results leave through an LDS tile as row-contiguous
// 16-byte stores.  Same contracts as fr_conv_igemm (mode 0 stride 2 / mode 2): bit-compatible layouts of src, w,
// out, aux and the column partial sums.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "frhip_internal.h"

#ifdef FRHIP_STAMPS
// Diagnostic build only (make stamps -> libfrhip_stamps.so; never the product library): wave 0 of every workgroup writes
// s_memrealtime (100 MHz) at its phase boundaries to a buffer of its own (16 slots per workgroup), read by tools/stamps_s2.py.
__device__ unsigned long long* fr_stamp_buf_s2 = nullptr;
extern "C" int fr_debug_set_stamp_buffer_s2(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fr_stamp_buf_s2), &dev_ptr, sizeof(dev_ptr));
}
#define S2_STAMP(k)                                                                                          \
  do {                                                                                                       \
    if (threadIdx.x == 0 && fr_stamp_buf_s2) fr_stamp_buf_s2[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define S2_STAMP(k)
#endif

namespace {

// COUT is the width ONE workgroup computes (NSPL workgroups share a strip, conv3x3_strip.hip); NIMG > 1: a workgroup owns
// NIMG whole images (ROWS == WL) stacked in LDS, so that a weight fragment feeds NIMG x ROWS x WL pixels.
template <int CIN, int COUT, int WL, int ROWS, int WN, int NW, int KIND, int NIMG = 1>
struct S2 {
  static constexpr int NTH = NW * 64;
  static constexpr int HL = WL;
  static constexpr int GW = WL + 1;                        // one halo column (left: forward, right: gradient)
  static constexpr int GH = ROWS + 1;                      // one halo row
  static constexpr int CH = CIN / 8;
  static constexpr int PPAD = CIN == 64 ? 16 : 32;         // see conv3x3_strip.hip: conflict-free pixel stride
  static constexpr int PSTR = CIN * 2 + PPAD;
  // row stride: the 16-byte slot index keeps counting across a row wrap of an M tile (slot(h+1, 0) == slot(h, WL) mod 16)
  // (multi-image instances: the same residue with the next row starting INSIDE the last pixel's 32 padding bytes -- 256 B
  // less per row, which is what lets two 512-channel gradient images sit beside the output tile)
  static constexpr bool TIGHT = NIMG > 1 && PSTR % 256 <= PPAD;
  static constexpr int RSTR = TIGHT ? GW * PSTR - PSTR % 256 : GW * PSTR + (256 - PSTR % 256) % 256;
  static constexpr int ISTR = GH * RSTR;                   // one image (NIMG > 1)
  static constexpr int IMG_BYTES = NIMG * ISTR + 512;      // + slack for ring reads past the last channel chunk
  static constexpr int M = NIMG * ROWS * WL;
  static constexpr int MT = (M + 15) / 16;
  static constexpr int WM = NW / WN;
  static constexpr int TN = COUT / 16 / WN;
  static constexpr int TM = (MT + WM - 1) / WM;
  static constexpr int OSTR = COUT * 2 + 16;
  static constexpr int OUT_BYTES = (M * OSTR + 15) / 16 * 16;
  static constexpr int RED_BYTES = WM * 2 * COUT * 4;
  // forward: the output tile reuses the (dead) plane image; gradient: the g strip stays live across the four classes
  static constexpr int OUT_OFF = KIND == 0 ? 0 : (IMG_BYTES + 15) / 16 * 16;
  static constexpr int SYNC_OFF = OUT_OFF + OUT_BYTES + RED_BYTES;  // gradient: counters of the half-workgroup barriers
  static constexpr int LDS = KIND == 0 ? (IMG_BYTES > OUT_BYTES + RED_BYTES ? IMG_BYTES : OUT_BYTES + RED_BYTES)
                                       : SYNC_OFF + 64;
  static constexpr int NS = HL / ROWS;
  static_assert(HL % ROWS == 0, "strip rows must divide the low-res image");
  static_assert(NIMG == 1 || ROWS == WL, "multi-image workgroups own whole images");
  static_assert(NTH % CH == 0, "threads must be a multiple of the chunks per pixel");
  static_assert(COUT % (16 * WN) == 0 && NW % WN == 0, "bad wave split");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// Tap list of plane / class P = 2*ph + pw: NT = (ph ? 2 : 1) * (pw ? 2 : 1) taps; tap t -> (kernel tap index, LDS row
// and column offset).  Forward: kernel row kh reads high-res row 2i+kh-1 = plane row i-1 (kh = 0, ph = 1), i (kh = 1,
// ph = 0) or i (kh = 2, ph = 1); the image holds rows row0-1 .. row0+ROWS-1, so the LDS row offset is 0 or 1.
// Gradient: dx row 2i+ph collects g rows i+1 (kh = 0) and i (kh = 2) for ph = 1, g row i (kh = 1) for ph = 0; the
// image holds rows row0 .. row0+ROWS.
template <int KIND, int P>
struct Taps {
  static constexpr int PH = P >> 1, PW = P & 1;
  static constexpr int NH = PH ? 2 : 1, NWD = PW ? 2 : 1, NT = NH * NWD;
  static constexpr int kh(int t) { return PH ? 2 * (t / NWD) : 1; }
  static constexpr int kw(int t) { return PW ? 2 * (t % NWD) : 1; }
  static constexpr int ktap(int t) { return kh(t) * 3 + kw(t); }
  static constexpr int roff(int t) { return KIND == 0 ? (PH ? t / NWD : 1) : (PH ? 1 - t / NWD : 0); }
  static constexpr int coff(int t) { return KIND == 0 ? (PW ? t % NWD : 1) : (PW ? 1 - t % NWD : 0); }
};

// acc += sum over the taps of (KIND, P) and all CIN channels, A from the LDS image, B from global/L2.
template <class C, int CIN, int KIND, int P>
__device__ __forceinline__ void mma_taps(const char* smem, const int (&abase0)[C::TM], f32x4 (&acc)[C::TM][C::TN],
                                         const bf16_t* const (&wrow)[C::TN], const int wsh) {
  using T = Taps<KIND, P>;
  constexpr int NT = T::NT;
  constexpr int UC = NT == 1 ? 2 : 1;  // 32-channel chunks per unrolled body (so that a body holds >= 2 tap-steps)
  constexpr int QB = UC * NT;          // tap-steps per body: 2, 2, 2, 4
  constexpr int DB = 2;                // weight ring depth == request distance in tap-steps
  constexpr int NSTEP = QB * C::TM;
  // A ring depth (divides NSTEP): one slot per M tile; 2 in the whole-image 256-channel instance (13 slots: 8-76 bytes of
  // scratch, same time)
  constexpr int D = (C::TM == 13 && CIN >= 256) ? 2 : C::TM;
  static_assert(D * 4 + C::TM * C::TN * 4 <= 140, "fragment ring + accumulators: register budget");
  static_assert((CIN / 32) % UC == 0 && QB % DB == 0 && NSTEP % D == 0, "bad body shape");
  int abase[C::TM];
#pragma unroll
  for (int i = 0; i < C::TM; ++i) abase[i] = abase0[i];
  s16x8 bq[DB][C::TN];
  s16x8 ring[D];
  auto load_b = [&](int slot, int c0, int q) {
    const int t = q % NT, u = q / NT;
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
      bq[slot][j] = *reinterpret_cast<const s16x8*>(wrow[j] + ((T::ktap(t) * CIN + c0 + u * 32) << wsh));
  };
  auto a_addr = [&](int step) -> const s16x8* {  // step in [0, 2*NSTEP): second half = next body
    const int wrap = step >= NSTEP ? 1 : 0;
    const int st = wrap ? step - NSTEP : step;
    const int q = st / C::TM, i = st - q * C::TM;
    const int t = q % NT, u = q / NT;
    return reinterpret_cast<const s16x8*>(smem + abase[i] + T::roff(t) * C::RSTR + T::coff(t) * C::PSTR + u * 64 +
                                          wrap * UC * 64);
  };
#pragma unroll
  for (int d = 0; d < DB; ++d) load_b(d, 0, d);
#pragma unroll
  for (int d = 0; d < D; ++d) ring[d] = *a_addr(d);
  for (int c0 = 0; c0 < CIN; c0 += 32 * UC) {
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int q = st / C::TM, i = st - q * C::TM;
      const int slot = q % DB;
      const s16x8 a = ring[st % D];
#pragma unroll
      for (int j = 0; j < C::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[slot][j], a, acc[i][j], 0, 0, 0);  // = (W X^T) tile
      ring[st % D] = *a_addr(st + D);  // past the last chunk this reads (never used) bytes inside the LDS slack
      if (i == C::TM - 1) {
        int nq = q + DB, nc = c0;
        if (nq >= QB) {
          nq -= QB;
          nc += 32 * UC;
        }
        nc = nc < CIN ? nc : CIN - 32 * UC;  // clamp instead of branching: the count of loads in flight stays static
        load_b(slot, nc, nq);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, C::TN, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
      if (i == C::TM - 1) __builtin_amdgcn_sched_barrier(0);  // keep the weight requests two tap-steps ahead
    }
#pragma unroll
    for (int i = 0; i < C::TM; ++i) abase[i] += 64 * UC;
  }
}

template <int CIN, int COUT, int WL, int ROWS, int WN, int NW, int KIND, int PRO, int NSPL = 1, int NIMG = 1>
__global__ __launch_bounds__(NW * 64, 2) void conv3x3_s2_kernel(const FrConvArgs p, const int xcd) {
  using C = S2<CIN, COUT, WL, ROWS, WN, NW, KIND, NIMG>;
  constexpr int NTH = C::NTH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const otile = smem + C::OUT_OFF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const bf16_t* __restrict__ src = reinterpret_cast<const bf16_t*>(p.src);
  const bf16_t* __restrict__ wgt = reinterpret_cast<const bf16_t*>(p.w);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out);

  const int lb = (xcd & 1) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;  // halo-sharing strips (and the NSPL parts of one
                                                                         // strip) meet in one XCD's L2
  const int nh = NSPL > 1 ? lb % NSPL : 0;
  const int ncol0 = nh * COUT;                                           // first output channel of this workgroup
  const int sblk = NSPL > 1 ? lb / NSPL : lb;
  const int nstrips = gridDim.x / NSPL;
  const int b = NIMG > 1 ? sblk * NIMG : sblk / C::NS;                   // (first) image of the strip
  const int row0 = NIMG > 1 ? 0 : (sblk - b * C::NS) * ROWS;
  constexpr int PLANE = C::GH * C::GW;                                   // staged pixels per image

  // ------------------------------------------------------------------ image loader (prologue applied once)
  constexpr int TOTAL = NIMG * C::GH * C::GW * C::CH;
  const int ch = tid % C::CH;
  float pa[8], pb[8];
  if (PRO != FR_PRO_NONE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pa[j] = p.pro_a[ch * 8 + j];
      pb[j] = PRO == FR_PRO_BN ? p.pro_b[ch * 8 + j] : 0.f;
    }
  }
  auto load_image = [&](int ph, int pw) {
    // one batch when it is at most 16 loads per thread: a second trip of this loop costs a second HBM round trip, and the
    // compiler answered the two-trip form with kilobytes of scratch in the whole-image 256-channel instance
    constexpr int PER = (TOTAL + NTH - 1) / NTH;
    constexpr int UNR = PER <= 16 ? PER : 8;
    for (int base = 0; base < TOTAL; base += NTH * UNR) {
      U128 v[UNR];
      bool ok[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = base + u * NTH + tid;
        int pc = idx / C::CH;
        const int img = NIMG > 1 ? pc / PLANE : 0;
        pc -= img * PLANE;
        const int gh = pc / C::GW, gw = pc - gh * C::GW;
        size_t pix;
        if (KIND == 0) {  // plane (ph, pw) of the high-res input: low-res rows row0-1 .. row0+ROWS-1, columns -1 .. WL-1
          const int i = row0 + gh - 1, j = gw - 1;
          ok[u] = idx < TOTAL && i >= 0 && j >= 0;
          pix = ((size_t)((b + img) * 2 * C::HL + 2 * i + ph) * (2 * WL) + 2 * j + pw);
        } else {          // low-res gradient rows row0 .. row0+ROWS, columns 0 .. WL
          const int i = row0 + gh, j = gw;
          ok[u] = idx < TOTAL && i < C::HL && j < WL;
          pix = ((size_t)((b + img) * C::HL + i) * WL + j);
        }
        v[u] = ok[u] ? ld16(src + pix * (size_t)p.lda + ch * 8) : zero16();
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = base + u * NTH + tid;
        if (idx < TOTAL) {
          int pc = idx / C::CH;
          const int img = NIMG > 1 ? pc / PLANE : 0;
          pc -= img * PLANE;
          const int gh = pc / C::GW, gw = pc - gh * C::GW;
          U128 x = v[u];
          if (PRO != FR_PRO_NONE && ok[u]) {  // pro2 (frhip_internal.h): the prologue on packed pairs
            x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
            x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
            x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
            x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
          }
          st16(smem + img * C::ISTR + gh * C::RSTR + gw * C::PSTR + ch * 16, x);
        }
      }
    }
  };

  // Forward: the NEXT plane is requested into registers before the taps of the current one run and committed after them
  // (one workgroup per CU at 256 / 512 channels: nothing else hides the load latency of the four phases).
  // Only where the register budget is spent anyway: at 64 / 128 channels the 20-32 extra registers cost a resident workgroup
  // (128 -> 128 @28: 0.166 -> 0.196 ms with the prefetch).
  // Round 3, measured again with 7-row strips (tools/ab_libs.sh, one box): the prefetch at 128 channels 0.1355 -> 0.1372 ms;
  // two plane buffers (commit into the buffer the taps are not reading, one barrier per plane instead of two)
  // 128 ch 0.1355 -> 0.137, 256 ch 0.1066 -> 0.118 (costs the second resident workgroup), 512 ch 0.148 -> 0.147: the plane
  // loads are not what these kernels wait for.  4-wave workgroups at 128 channels (13 M tiles per wave, half the weight
  // stream): 170-3000 bytes of scratch per lane in every instance.
  constexpr int NPF = (TOTAL + NTH - 1) / NTH;
#ifndef FRHIP_S2_PFP_BUDGET
#define FRHIP_S2_PFP_BUDGET 150
#endif
  // ... and only where prefetch registers + accumulators + fragment ring fit (the whole-image 256-channel instance: 15
  // prefetch registers x 4 + 52 + 52 -> 370-520 bytes of scratch with it)
  constexpr bool PFP = KIND == 0 && CIN >= 256 && NPF * 4 + C::TM * C::TN * 4 + C::TM * 4 <= FRHIP_S2_PFP_BUDGET;
  U128 pf[PFP ? NPF : 1];
  unsigned pfok = 0;
  auto issue_plane = [&](int ph, int pw) {
    pfok = 0;
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int idx = u * NTH + tid;
      int pc = idx / C::CH;
      const int img = NIMG > 1 ? pc / PLANE : 0;
      pc -= img * PLANE;
      const int gh = pc / C::GW, gw = pc - gh * C::GW;
      const int i = row0 + gh - 1, j = gw - 1;
      const bool ok = idx < TOTAL && i >= 0 && j >= 0;
      const size_t pix = ((size_t)((b + img) * 2 * C::HL + 2 * i + ph) * (2 * WL) + 2 * j + pw);
      pf[PFP ? u : 0] = ok ? ld16(src + pix * (size_t)p.lda + ch * 8) : zero16();
      pfok |= ok ? (1u << u) : 0u;
    }
  };
  auto commit_plane = [&]() {
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int idx = u * NTH + tid;
      if (idx < TOTAL) {
        int pc = idx / C::CH;
        const int img = NIMG > 1 ? pc / PLANE : 0;
        pc -= img * PLANE;
        const int gh = pc / C::GW, gw = pc - gh * C::GW;
        U128 x = pf[PFP ? u : 0];
        if (PRO != FR_PRO_NONE && ((pfok >> u) & 1u)) {
          x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
          x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
          x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
          x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
        }
        st16(smem + img * C::ISTR + gh * C::RSTR + gw * C::PSTR + ch * 16, x);
      }
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = wn * C::TN * 16;
  const bf16_t* wrow[C::TN];
#pragma unroll
  for (int j = 0; j < C::TN; ++j) wrow[j] = wgt + (size_t)(ncol0 + n0 + j * 16 + fr) * 9 * CIN + fq * 8;
  const int wsh = p.w_frag ? 4 : 0;  // weights in MFMA-fragment order (FrConvArgs.w_frag; conv3x3_strip.hip): 1024 contiguous bytes per load
  if (wsh) {
#pragma unroll
    for (int j = 0; j < C::TN; ++j) wrow[j] = wgt + (size_t)((ncol0 + n0) / 16 + j) * 16 * 9 * CIN + lane * 8;
  }
  int abase[C::TM];
#pragma unroll
  for (int i = 0; i < C::TM; ++i) {
    int m = (wm * C::TM + i) * 16 + fr;
    m = m < C::M ? m : 0;
    const int img = NIMG > 1 ? m / (ROWS * WL) : 0;
    m -= img * (ROWS * WL);
    const int h = m / WL, w = m - h * WL;
    abase[i] = img * C::ISTR + h * C::RSTR + w * C::PSTR + fq * 16;
  }
  const int epi = p.epi;
  const bool stats = epi == FR_EPI_STATS || epi == FR_EPI_PRELU_BWD || epi == FR_EPI_BNBWD;
  constexpr int OCH = COUT / 8;
  // Round 6, data gradient: the two HALVES of the workgroup run half a class out of step.  A wave owns 16 (or 32) output
  // channels of every pixel, waves w and w + 4 share a SIMD; with all eight in lock-step the SIMD's matrix pipe idles through
  // every epilogue (aux tile -> LDS, PReLU-backward cells, tile -> global: 5.2 of the 9.5 us a class takes, tools/stamps_s2.py)
  // and its vector pipe through every tap list.  The halves touch disjoint columns of the output tile and of the reduction
  // scratch, so each gets barriers of its own (an LDS counter, 4 waves) and waves 4-7 start their first tap list when waves
  // 0-3 have finished theirs: from then on one wave of a SIMD issues MFMAs while the other runs its epilogue.  Same
  // arithmetic, same order: bit-identical.  (xcd bit 1: FRHIP_S2_STAGGER, default on.)
  constexpr bool STG = KIND == 1 && NW == 8 && WN == 8;
  const bool stg = STG && (xcd & 2);
  const int half = stg ? wave >> 2 : 0;
  const int htid = stg ? (tid & 255) : tid;
  const int hnth = stg ? 256 : NTH;
  const int hoch = stg ? OCH / 2 : OCH;       // 16-byte chunks of a tile row this half moves
  const int hc0 = half * hoch;
  volatile unsigned* hctr = reinterpret_cast<volatile unsigned*>(smem + C::SYNC_OFF) + half;  // [0], [1]: barrier counters; [2]: start flag
  unsigned hgen = 0;
  auto hsync = [&]() {
    if (!stg) {
      __syncthreads();
      return;
    }
    hgen += 4;  // four waves arrive per barrier; the counter only grows
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(hctr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (*hctr < hgen) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  f32x4 acc[C::TM][C::TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // ------------------------------------------------------------------ epilogue of one output class
  // cls < 0: forward (output pixel = low-res pixel); cls = 2*ph + pw: gradient class (output pixel (2i+ph, 2j+pw))
  // aux cells of one output class: requested before the taps of that class run (data gradient), written into the output
  // tile behind them
  constexpr int NAX = (C::M * OCH + NTH - 1) / NTH;
  U128 axr[KIND == 1 ? NAX : 1];  // forward (inference epilogue only): loaded inside the epilogue, no registers held across the taps
  const bool has_aux = epi == FR_EPI_PRELU_BWD || epi == FR_EPI_BNBWD || epi == FR_EPI_BIAS_RES;
  auto out_pix = [&](int cls, int r) -> size_t {
    const int ph = cls < 0 ? 0 : cls >> 1, pw = cls < 0 ? 0 : cls & 1;
    const int img = NIMG > 1 ? r / (ROWS * WL) : 0;
    r -= img * (ROWS * WL);
    const int h = r / WL, w = r - h * WL;
    if (KIND == 0) return (size_t)((b + img) * C::HL + row0 + h) * WL + w;
    return (size_t)((b + img) * 2 * C::HL + 2 * (row0 + h) + ph) * (2 * WL) + 2 * w + pw;
  };
  auto issue_aux = [&](int cls) {
    if (KIND == 1 && has_aux) {
      const bf16_t* __restrict__ aux = reinterpret_cast<const bf16_t*>(p.aux);
#pragma unroll
      for (int u = 0; u < NAX; ++u) {
        int idx = u * hnth + htid;
        idx = idx < C::M * hoch ? idx : C::M * hoch - 1;
        const int r = idx / hoch, c8 = hc0 + idx - r * hoch;
        axr[KIND == 1 ? u : 0] = ld16(aux + out_pix(cls, r) * (size_t)p.ldaux + ncol0 + c8 * 8);
      }
    }
  };
  auto epilogue = [&](int cls) {
    auto dst_pix = [&](int r) -> size_t { return out_pix(cls, r); };
    if (has_aux) {
      if (KIND == 1) {
#pragma unroll
        for (int u = 0; u < NAX; ++u) {
          const int idx = u * hnth + htid;
          if (idx < C::M * hoch) {
            const int r = idx / hoch, c8 = hc0 + idx - r * hoch;
            st16(otile + r * C::OSTR + c8 * 16, axr[KIND == 1 ? u : 0]);
          }
        }
      } else {
        const bf16_t* __restrict__ aux = reinterpret_cast<const bf16_t*>(p.aux);
        for (int idx = tid; idx < C::M * OCH; idx += NTH) {
          const int r = idx / OCH, c8 = idx - r * OCH;
          st16(otile + r * C::OSTR + c8 * 16, ld16(aux + dst_pix(r) * (size_t)p.ldaux + ncol0 + c8 * 8));
        }
      }
      hsync();
    }
    // weights were the MFMA A operand: a lane holds four consecutive channels (fq*4 + r) of one pixel (fr) per tile
    float* red = reinterpret_cast<float*>(otile + C::OUT_BYTES);  // [WM][2][COUT] column sums, behind the output tile
    // The epilogue kind is a run-time argument, but inside the per-element loops it must be a compile-time constant:
    // with `epi` tested per element the compiler emitted a scalar branch per accumulator (8000 instructions, ~10 us).
    auto cells = [&](auto tag) {
      constexpr int E = decltype(tag)::value;
      float ea[C::TN][4], eb[C::TN][4], s0[C::TN][4], s1[C::TN][4];
  #pragma unroll
      for (int j = 0; j < C::TN; ++j)
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = ncol0 + n0 + j * 16 + fq * 4 + r;
          ea[j][r] = (E == FR_EPI_PRELU_BWD || E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES) ? p.epi_a[n] : 0.f;
          eb[j][r] = (E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES) ? p.epi_b[n] : 0.f;
          s0[j][r] = s1[j][r] = 0.f;
        }
  #pragma unroll
      for (int i = 0; i < C::TM; ++i) {
        const int m = (wm * C::TM + i) * 16 + fr;
        if (wm * C::TM + i >= C::MT || m >= C::M) continue;
  #pragma unroll
        for (int j = 0; j < C::TN; ++j) {
          uint2* cell = reinterpret_cast<uint2*>(otile + m * C::OSTR + (n0 + j * 16 + fq * 4) * 2);
          float v[4], x[4];
  #pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
          if (E == FR_EPI_PRELU_BWD || E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES) {
            const uint2 u = *cell;
            x[0] = __uint_as_float(u.x << 16);
            x[1] = __uint_as_float(u.x & 0xFFFF0000u);
            x[2] = __uint_as_float(u.y << 16);
            x[3] = __uint_as_float(u.y & 0xFFFF0000u);
          }
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (E == FR_EPI_STATS) {
              s0[j][r] += v[r];
              s1[j][r] = fmaf(v[r], v[r], s1[j][r]);
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
          *cell = o;
        }
      }
      if (E != FR_EPI_STORE && E != FR_EPI_BIAS_RES) {
  #pragma unroll
        for (int j = 0; j < C::TN; ++j)
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            float a = s0[j][r], c = s1[j][r];
  #pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
              a += __shfl_xor(a, o, 64);
              c += __shfl_xor(c, o, 64);
            }
            if (fr == 0) {
              red[(wm * 2 + 0) * COUT + n0 + j * 16 + fq * 4 + r] = a;
              red[(wm * 2 + 1) * COUT + n0 + j * 16 + fq * 4 + r] = c;
            }
          }
      }
    };
    switch (epi) {
      case FR_EPI_STATS: cells(std::integral_constant<int, FR_EPI_STATS>{}); break;
      case FR_EPI_PRELU_BWD: cells(std::integral_constant<int, FR_EPI_PRELU_BWD>{}); break;
      case FR_EPI_BNBWD: cells(std::integral_constant<int, FR_EPI_BNBWD>{}); break;
      case FR_EPI_BIAS_RES: cells(std::integral_constant<int, FR_EPI_BIAS_RES>{}); break;
      default: cells(std::integral_constant<int, FR_EPI_STORE>{}); break;
    }
    hsync();
    for (int idx = htid; idx < C::M * hoch; idx += hnth) {
      const int r = idx / hoch, c8 = hc0 + idx - r * hoch;
      st16(out + dst_pix(r) * (size_t)p.ldc + ncol0 + c8 * 8, ld16(otile + r * C::OSTR + c8 * 16));
    }
    if (stats) {
      const size_t prow = (size_t)(cls < 0 ? 0 : cls) * nstrips + sblk;  // gradient: rows ordered [class][strip]
      const int hcout = stg ? COUT / 2 : COUT;
      for (int c = htid; c < 2 * hcout; c += hnth) {
        const int k = c / hcout, n = half * hcout + c - k * hcout;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < C::WM; ++g) t += red[(g * 2 + k) * COUT + n];
        st_part(p.part + (prow * 2 + k) * (COUT * NSPL) + ncol0 + n, t);
      }
    }
    hsync();  // the tile and the reduction scratch (this half's columns) are free again
  };

  if (KIND == 0 && !PFP) {
    zero_acc();
    S2_STAMP(0);
    load_image(0, 0);
    S2_STAMP(1);
    __syncthreads();
    S2_STAMP(2);
    mma_taps<C, CIN, 0, 0>(smem, abase, acc, wrow, wsh);
    S2_STAMP(3);
    __syncthreads();  // every wave is done reading the plane
    S2_STAMP(4);
    load_image(0, 1);
    __syncthreads();
    S2_STAMP(5);
    mma_taps<C, CIN, 0, 1>(smem, abase, acc, wrow, wsh);
    __syncthreads();
    S2_STAMP(6);
    load_image(1, 0);
    __syncthreads();
    S2_STAMP(7);
    mma_taps<C, CIN, 0, 2>(smem, abase, acc, wrow, wsh);
    __syncthreads();
    S2_STAMP(8);
    load_image(1, 1);
    S2_STAMP(9);
    __syncthreads();
    S2_STAMP(10);
    issue_aux(-1);
    mma_taps<C, CIN, 0, 3>(smem, abase, acc, wrow, wsh);
    S2_STAMP(11);
    __syncthreads();  // LDS is now the output tile
    S2_STAMP(12);
    epilogue(-1);
    S2_STAMP(13);
  } else if (KIND == 0) {
    zero_acc();
    issue_plane(0, 0);
    commit_plane();
    __syncthreads();
    issue_plane(0, 1);
    mma_taps<C, CIN, 0, 0>(smem, abase, acc, wrow, wsh);
    __syncthreads();  // every wave is done reading the plane
    commit_plane();
    __syncthreads();
    issue_plane(1, 0);
    mma_taps<C, CIN, 0, 1>(smem, abase, acc, wrow, wsh);
    __syncthreads();
    commit_plane();
    __syncthreads();
    issue_plane(1, 1);
    mma_taps<C, CIN, 0, 2>(smem, abase, acc, wrow, wsh);
    __syncthreads();
    commit_plane();
    __syncthreads();
    issue_aux(-1);
    mma_taps<C, CIN, 0, 3>(smem, abase, acc, wrow, wsh);
    __syncthreads();  // LDS is now the output tile
    epilogue(-1);
  } else {
    S2_STAMP(0);
    if (STG && tid < 4) reinterpret_cast<unsigned*>(smem + C::SYNC_OFF)[tid] = 0u;
    load_image(0, 0);
    S2_STAMP(1);
    __syncthreads();
    S2_STAMP(2);
    zero_acc();
    issue_aux(0);
    if (stg) {
      volatile unsigned* start = reinterpret_cast<volatile unsigned*>(smem + C::SYNC_OFF) + 2;
      if (half == 1) {  // half a class behind: wait for waves 0-3 to finish their first tap list
        while (*start < 4u) __builtin_amdgcn_s_sleep(2);
      }
      mma_taps<C, CIN, 1, 0>(smem, abase, acc, wrow, wsh);
      if (half == 0 && lane == 0)
        __hip_atomic_fetch_add(const_cast<unsigned*>(start), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      mma_taps<C, CIN, 1, 0>(smem, abase, acc, wrow, wsh);
    }
    S2_STAMP(3);
    epilogue(0);
    S2_STAMP(4);
    zero_acc();
    issue_aux(1);
    mma_taps<C, CIN, 1, 1>(smem, abase, acc, wrow, wsh);
    S2_STAMP(5);
    epilogue(1);
    S2_STAMP(6);
    zero_acc();
    issue_aux(2);
    mma_taps<C, CIN, 1, 2>(smem, abase, acc, wrow, wsh);
    S2_STAMP(7);
    epilogue(2);
    S2_STAMP(8);
    zero_acc();
    issue_aux(3);
    mma_taps<C, CIN, 1, 3>(smem, abase, acc, wrow, wsh);
    S2_STAMP(9);
    epilogue(3);
    S2_STAMP(10);
  }
}

static int s2_xcd_order() {  // bit 0: FRHIP_XCD_ORDER=0: strips in dispatch order; bit 1: FRHIP_S2_STAGGER=0: halves in lock-step
  static const int* v = fr_option_slot("FRHIP_XCD_ORDER", 1);
  static const int* g = fr_option_slot("FRHIP_S2_STAGGER", 1);
  return (*v != 0 ? 1 : 0) | (*g != 0 ? 2 : 0);
}

template <int CIN, int COUT, int WL, int ROWS, int WN, int NW, int KIND, int PRO, int NSPL, int NIMG>
int launch(const FrConvArgs& a, hipStream_t st) {
  using C = S2<CIN, COUT, WL, ROWS, WN, NW, KIND, NIMG>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv3x3_s2_kernel<CIN, COUT, WL, ROWS, WN, NW, KIND, PRO, NSPL, NIMG>),
        hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    fr_attr_done(attr_done);
  }
  FR_LAUNCH_KERNEL((conv3x3_s2_kernel<CIN, COUT, WL, ROWS, WN, NW, KIND, PRO, NSPL, NIMG>),
                     dim3(a.B * C::NS / NIMG * NSPL), dim3(C::NTH), C::LDS, st, a, s2_xcd_order());
  FR_LAUNCH_CHECK();
}

template <int CIN, int COUT, int WL, int ROWS, int WN, int NW, int KIND, int NSPL = 1, int NIMG = 1>
int by_pro(const FrConvArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return launch<CIN, COUT, WL, ROWS, WN, NW, KIND, FR_PRO_NONE, NSPL, NIMG>(a, st);
    case FR_PRO_BN: return launch<CIN, COUT, WL, ROWS, WN, NW, KIND, FR_PRO_BN, NSPL, NIMG>(a, st);
    case FR_PRO_PRELU: return launch<CIN, COUT, WL, ROWS, WN, NW, KIND, FR_PRO_PRELU, NSPL, NIMG>(a, st);
  }
  FR_UNSUPPORTED("fr_conv3x3_s2_strip: unknown prologue");
}

// The table (round 3; its round-2 predecessor behind FRHIP_S2_VARIANT=0 was removed in round 4): instances in which ONE
// weight fragment feeds as many M tiles as the registers allow -- a wave's cost per MFMA is its private weight stream from L2
// (conv3x3_strip.hip):
//   128 @28: 8 waves x (13 tiles x 1 column) instead of 4 x 2 waves x (7 x 2): 0.139 -> 0.116 ms forward, 0.193 -> 0.168 gradient
//   256 @14 forward: the whole image (13 tiles) x 1 column, output channels over two workgroups (123 KB plane);
//            the gradient keeps 7-row strips: its g strip stays resident beside the output tile (178 KB otherwise)
//   512 @7:  two images per workgroup (7 tiles x 1 column), output channels over four workgroups; odd batches keep the
//            one-image instance

// strips (= partial-sum rows per output class) of a served shape, 0 = not served: the four IR stage transitions
int s2_strips(int B, int C, int WL, int mode) {
  if (C == 64 && WL == 56) return B * 28;   // 2 low-res rows per workgroup
  if (C == 128 && WL == 28) return B * 4;   // 7 rows
  if (C == 256 && WL == 14) return mode == 0 ? B : B * 2;
  if (C == 512 && WL == 7) return B % 2 == 0 ? B / 2 : B;
  return 0;
}

}  // namespace

// Partial-sum rows the kernel writes for a supported (B, C -> C, low-res width) problem; 0 when not served.
// mode 0: forward (one row per workgroup); mode 2: data gradient (4 classes x workgroups, ordered [class][workgroup]).
extern "C" int fr_conv3x3_s2_strip_parts(int B, int Cin, int Cout, int WL, int mode) {
  if (Cin != Cout) return 0;
  // 64 channels: one row per work item of the rolling-window kernel.  NOTE: the caller's epilogue decides whether that kernel
  // serves the launch (forward STORE / STATS, gradient PReLU backward); the combinations it does not serve write no sums.
  if (Cin == 64 && WL == 56 && fr_s2roll_enabled()) return fr_s2roll_parts(B);
  if (mode == 0) {  // the warp-specialised forward kernel has its own strip count at 7x7 (four images per workgroup)
    const int ws = fr_s2ws_strips(B, Cin, WL);
    if (ws) return ws;
  }
  const int strips = s2_strips(B, Cin, WL, mode);
  return mode == 2 ? 4 * strips : strips;
}

// 1 when the launch reads fragment-order weights (FrConvArgs.w_frag): every served shape but the 64-channel layer, whose
// rolling-window kernel keeps its weights resident
extern "C" int fr_conv3x3_s2_strip_takes_frag(int B, int C, int WL, int mode) {
  if (C == 64) return 0;
  return fr_conv3x3_s2_strip_parts(B, C, C, WL, mode) > 0 ? 1 : 0;
}

extern "C" int fr_conv3x3_s2_strip(const FrConvArgs* args, void* stream) {
  const FrConvArgs& a = *args;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.KH != 3 || a.KW != 3 || a.stride != 2 || a.pad != 1 || a.out_f32 || a.splitk > 1 || a.bias ||
      a.epi == FR_EPI_MARGIN || a.epi == FR_EPI_ATOMIC || (a.mode != 0 && a.mode != 2))
    FR_UNSUPPORTED("fr_conv3x3_s2_strip: stride-2 3x3 pad-1 bf16 convolution (mode 0) or its data gradient (mode 2)");
  if (a.mode == 2 && (a.par_h >= 0 || a.par_w >= 0))
    FR_UNSUPPORTED("fr_conv3x3_s2_strip: the data gradient produces all four parity classes (par_h = par_w = -1)");
  if (a.lda % 8 || a.ldc % 8 || (a.aux && a.ldaux % 8)) FR_UNSUPPORTED("fr_conv3x3_s2_strip: strides must be 16-byte multiples");
  // low-res grid: forward output / gradient input
  const int WLo = a.mode == 0 ? a.RW : a.SW, HLo = a.mode == 0 ? a.RH : a.SH;
  const int WHi = a.mode == 0 ? a.SW : a.RW, HHi = a.mode == 0 ? a.SH : a.RH;
  if (WLo != HLo || WHi != 2 * WLo || HHi != 2 * HLo || a.SC != a.N)
    FR_UNSUPPORTED("fr_conv3x3_s2_strip: square images, high-res side = 2 x low-res side, Cin == Cout");
  if (fr_s2roll_serves(a)) {
    if (a.w_frag) FR_UNSUPPORTED("fr_conv3x3_s2_strip: the 64-channel rolling-window kernel takes the plain weight layout (w_frag)");
    return fr_s2roll_launch(a, st);
  }
  if (fr_s2ws_serves(a)) return fr_s2ws_launch(a, st);  // forward, 128 / 256 / 512 channels (round 6)
  // fr_conv3x3_s2_strip_parts() has no epilogue argument: for the 64-channel layer it answers with the row count of the
  // rolling-window kernel.  A summing epilogue that kernel does not serve would make the strip kernel below write a
  // DIFFERENT number of partial rows into a buffer sized from that answer -- refuse instead of overrunning it.
  if (a.SC == 64 && WLo == 56 && fr_s2roll_enabled() && a.part &&
      (a.epi == FR_EPI_STATS || a.epi == FR_EPI_PRELU_BWD || a.epi == FR_EPI_BNBWD))
    FR_UNSUPPORTED("fr_conv3x3_s2_strip: this prologue / epilogue combination of the 64-channel layer writes partial rows "
                   "in the strip kernel's layout, not the one fr_conv3x3_s2_strip_parts() reports (set FRHIP_S2ROLL=0 or "
                   "use fr_conv_igemm)");
#define SHAPE(c, wl, rows, wn, nw)                                   \
  if (a.SC == c && WLo == wl) {                                      \
    if (a.mode == 0) return by_pro<c, c, wl, rows, wn, nw, 0>(a, st); \
    return by_pro<c, c, wl, rows, wn, nw, 1>(a, st);                  \
  }
  SHAPE(64, 56, 2, 2, 4)
  // 128 -> 128: 7-row strips (196 pixels per weight pass) although only one workgroup then fits a CU: 0.166 -> 0.140 ms
  // forward, 0.221 -> 0.206 ms gradient against the 4-row strips (the kernel is bound by the weight stream)
  SHAPE(128, 28, 7, 8, 8)
  if (a.SC == 256 && WLo == 14 && a.mode == 0) return by_pro<256, 128, 14, 14, 8, 8, 0, 2, 1>(a, st);
  if (a.SC == 512 && WLo == 7 && a.B % 2 == 0) {
    if (a.mode == 0) return by_pro<512, 128, 7, 7, 8, 8, 0, 4, 2>(a, st);
    return by_pro<512, 128, 7, 7, 8, 8, 1, 4, 2>(a, st);
  }
  SHAPE(256, 14, 7, 8, 8)
  SHAPE(512, 7, 7, 8, 8)
#undef SHAPE
  FR_UNSUPPORTED("fr_conv3x3_s2_strip: shape not in the table");
}
