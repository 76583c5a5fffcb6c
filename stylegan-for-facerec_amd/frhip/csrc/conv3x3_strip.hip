// frhip -- stride-1 3x3 convolution with the input strip resident in LDS (bf16, gfx950).
//
// The 3x3 stride-1 convolutions are >85 % of the IR-50 FLOPs.  In the generic implicit-GEMM
// kernel every input element is gathered from L2 nine times (once per tap) and the BatchNorm / PReLU prologue
// is re-applied each time; at 128x128 tiles that gather alone saturates the per-CU vector-memory path.  Here a
// workgroup owns ROWS output rows of ONE image: it loads the (ROWS+2) x (W+2) x CIN input strip once, applies the
// prologue once, and keeps it in LDS (up to 153 KB of the 160 KB) as a [pixel][channel] image whose pixel stride is
// padded by 16 B.  The nine taps are then nine shifted views of the same LDS image: an A fragment is ONE
// ds_read_b128 at (row base register + compile-time tap offset) -- no per-tap address arithmetic at all -- and the
// odd 16-B-slot stride makes 16 consecutive pixels hit 16 different bank groups.  The main loop has NO barrier.  Weights never
// touch LDS: every wave owns a private slice of output channels and streams its B fragments straight from
// L2 into registers (16 B per lane, 3-deep register ring), so all 256 workgroups read the same 0.1-2.4 MB of
// weights out of their XCD's L2.
//
// Epilogue: accumulators (+ fused PReLU-backward / BN-backward sums / BN statistics, same contracts as
// fr_conv_igemm) are written as bf16 into an LDS tile [rows][COUT] (aliasing the dead input strip) and leave
// as row-contiguous 16-B stores.
//
// Same reference arithmetic as fr_conv_igemm: Conv2d 3x3 s1 of bottleneck_IR (backbone/model_irse.py:57-59)
// with BN apply (:57) or PReLU (:58) on the input, and its autograd data gradient (flip = 1: taps mirrored,
// weights given as [Cin][tap][Cout]).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "frhip_internal.h"

#ifdef FRHIP_STAMPS
// Diagnostic build only (make stamps -> libfrhip_stamps.so; never the product library): wave 0 of every workgroup
// writes s_memrealtime (100 MHz) at its phase boundaries + its hardware id to a buffer of its own, read by
// tools/stamps.py.  No output value depends on a stamp.
__device__ unsigned long long* fr_stamp_buf = nullptr;
extern "C" int fr_debug_set_stamp_buffer(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fr_stamp_buf), &dev_ptr, sizeof(dev_ptr));
}
// (the shader-clock counter s_memtime goes to a second plane of the buffer, 65536 x 8 entries behind the first: the clock the
// loop holds is delta s_memtime / delta s_memrealtime x 100 MHz)
#define FR_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (tid == 0 && fr_stamp_buf) {                                                                   \
      fr_stamp_buf[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();                  \
      fr_stamp_buf[(size_t)(65536 + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memtime();            \
    }                                                                                                 \
  } while (0)
#else
#define FR_STAMP(k)
#endif

namespace {


// NIMG > 1: a workgroup owns NIMG whole images (ROWS == W), stacked in LDS with their own halos -- M = NIMG*W*W rows share
// every weight fragment, which is what the 7x7 stage needs (49 pixels per image: 4.7 MB of weights per 0.23 GFLOP
// otherwise).  KSPL > 1: only CIN/KSPL input channels are resident at a time; the strip is staged KSPL times and the
// accumulators persist across the stages (halves the LDS image so that NIMG = 2 fits at 512 channels).
template <int CIN, int COUT, int W, int ROWS, int WN, int NW, int NIMG = 1, int KSPL = 1>
struct SC {
  static constexpr int NTH = NW * 64;                      // 8 waves: one workgroup per CU; 4 waves: two co-resident
                                                           // workgroups whose load / epilogue phases overlap each other's MFMAs
  static constexpr int H = W;
  static constexpr int GW = W + 2;
  static constexpr int GH = ROWS + 2;
  static constexpr int CK = CIN / KSPL;                    // input channels resident in LDS
  static constexpr int CH = CK / 8;                        // 16-B chunks per pixel
  // Pixel stride = CK*2 + 32 B: an even number of 16-B slots per pixel, so within every hardware lane group of a
  // ds_read_b128 the 8 rows that read k-chunk q=0/2 land on even slots and the 8 rows reading q=1/3 (+16 B) on odd
  // slots -- measured conflict-free (tools/lds_probe.hip: 6.9 cycles vs 8.0 for CIN*2+16 and 32 for CIN*2).
  // (64-channel strips keep the 16-B pad: the extra 16 B per pixel would push the 56x56 strip past 80 KB and cost the
  // second resident workgroup, which is worth far more there than the 15 % on LDS reads.)
  static constexpr int PPAD = CK == 64 ? 16 : 32;
  static constexpr int PSTR = CK * 2 + PPAD;
  // Row stride: + 192 B so that the slot index keeps counting across an image-row wrap (slot(h+1, 0) == slot(h, W)
  // mod 16): the 16 consecutive output pixels of an M tile behave like 16 consecutive pixels of one row.
  static constexpr int RSTR = GW * PSTR + (PPAD == 32 ? 192 : 224);
  // one image (NIMG > 1).  Round 6: padded so that the slot index also keeps counting across an IMAGE wrap (the pixel behind
  // (ROWS - 1, W - 1) of image n is (0, 0) of image n + 1: an M tile that straddles two 7x7 images reads 16 consecutive slots)
  static constexpr int IWRAP = (GH * RSTR - (ROWS - 1) * RSTR - W * PSTR) % 256;
  // (512 -> 512 @7, four images per workgroup, same box: forward 0.0583 -> 0.0566 ms, data gradient 0.0669 -> 0.0645)
  static constexpr int ISTR = GH * RSTR + (NIMG > 1 ? (256 - IWRAP) % 256 : 0);
  static constexpr int IMG_BYTES = NIMG * ISTR + 128;       // + slack for the ring's reads past the last chunk
  static constexpr int M = NIMG * ROWS * W;
  static constexpr int MT = (M + 15) / 16;
  static constexpr int WM = NW / WN;
  static constexpr int TN = COUT / 16 / WN;
  static constexpr int TM = (MT + WM - 1) / WM;
  static constexpr int OSTR = COUT * 2 + 16;               // out-tile row stride, bytes
  static constexpr int OUT_BYTES = (M * OSTR + 15) / 16 * 16;
  static constexpr int RED_BYTES = WM * 3 * COUT * 4;        // [WM][<= 3][COUT] column sums
  static constexpr int LDS = IMG_BYTES > OUT_BYTES + RED_BYTES ? IMG_BYTES : OUT_BYTES + RED_BYTES;
  static constexpr int NS = H / ROWS;                      // strips per image
  static_assert(H % ROWS == 0, "strip rows must divide the image");
  static_assert(NIMG == 1 || ROWS == W, "multi-image workgroups own whole images");
  static_assert(NTH % CH == 0, "threads must be a multiple of the chunks per pixel");
  static_assert(COUT % (16 * WN) == 0 && NW % WN == 0, "bad wave split");
  static_assert(CIN % KSPL == 0 && CK % 32 == 0, "bad channel split");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// (Round 3 carried a PAIR variant -- conv1 -> PReLU -> conv2 of a unit in one launch, the bf16 tile of conv1 turned into
// conv2's haloed LDS image -- bit-identical to two launches and measured no faster as a kernel (0.1156-0.1215 ms against
// 0.1114-0.1259) and 0.1-0.2 ms slower per step: the 4.5-us strip load it saves is paid back by the tile -> image pass, and
// two launches let conv2's workgroups start on the CUs that finish conv1 first.  Removed in round 4 with ABI v4.)
template <int CIN, int COUT, int W, int ROWS, int WN, int NW, int NSPL, int PRO, int NIMG = 1, int KSPL = 1>
__global__ __launch_bounds__(NW * 64, 2) void conv3x3_strip_kernel(const FrConvArgs p, const int xcd) {
  using C = SC<CIN, COUT, W, ROWS, WN, NW, NIMG, KSPL>;
  constexpr int CK = C::CK;
  constexpr int NTH = C::NTH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: keeps the tile loops branch-free
  const int wn = wave % WN, wm = wave / WN;
  constexpr bool TWO = PRO == FR_PRO_RESBN || PRO == FR_PRO_RESBN_SE;  // two-source prologues
  const bf16_t* __restrict__ src = reinterpret_cast<const bf16_t*>(p.src);
  const bf16_t* __restrict__ wgt = reinterpret_cast<const bf16_t*>(p.w);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out);

  // One strip per workgroup.  (Walking several strips with the next one register-prefetched under the MFMAs was
  // measured slower -- spills, and the strips of a workgroup serialise -- and is gone.)
  // NSPL > 1: the output channels are split over NSPL workgroups per strip (COUT is the per-workgroup width)
  // logical block: strips that share halo rows (and the NSPL halves of one strip) are neighbours on one XCD
  const int lb = xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int nh = NSPL > 1 ? lb % NSPL : 0;
  const int ncol0 = nh * COUT;
  const int sblk = NSPL > 1 ? lb / NSPL : lb;
  const int s = sblk;  // strip index; a multi-image strip = NIMG consecutive images

  constexpr int WP = W + 2;
  constexpr int TOTAL = NIMG * C::GH * WP * C::CH;
  const int ch = tid % C::CH;  // NTH is a multiple of CH: a thread always handles the same channel chunk
  constexpr bool RES = PRO == FR_PRO_RESBN || PRO == FR_PRO_RESBN_SE;
  float pa[8], pb[8], pc[8], pd[RES ? 8 : 1], pg[PRO == FR_PRO_RESBN_SE ? 8 : 1];  // prologue coefficients of this thread's channel chunk: re-read per strip (L1 hits) rather
                              // than kept live across the MFMA loop
  int kc = 0;  // channel stage (KSPL > 1): input channels [kc*CK, (kc+1)*CK) are resident
  auto kco = [&]() -> int { return KSPL > 1 ? kc * CK : 0; };  // literally 0 for the single-stage instances
  auto load_pro = [&](int s) {
    if (PRO != FR_PRO_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pa[j] = p.pro_a[kco() + ch * 8 + j];
        pb[j] = (PRO == FR_PRO_BN || TWO) ? p.pro_b[kco() + ch * 8 + j] : 0.f;
        pc[j] = TWO ? p.pro_c[kco() + ch * 8 + j] : 0.f;
        if (RES) pd[j] = p.pro_d[kco() + ch * 8 + j];
        if (PRO == FR_PRO_RESBN_SE) pg[j] = p.pro_g[(size_t)(s / C::NS) * p.SC + kco() + ch * 8 + j];  // this image's gates
      }
    }
  };
  // (Round 4 also carried FR_PRO_BNBWD2 here -- the backward of BN2 applied while conv2's data gradient loads its strip: -0.19 ms
  // per step on one stream, nothing on two, profiles/r04_ab_fuse_bn2_schedule.txt; removed in round 5 with ABI v5.)
  // FR_PRO_RESBN (forward): the operand is BN1 of o = round(a*y + b + x2) -- the output of the unit in front, formed here from
  // its conv2 output y and its input x2 instead of by a BN-apply pass; o goes to p.pro_out, the residual stream.
  const bf16_t* __restrict__ src2 = reinterpret_cast<const bf16_t*>(p.src2);
  bf16_t* __restrict__ pro_out = TWO ? reinterpret_cast<bf16_t*>(p.pro_out) : nullptr;
  // chunk u of this thread (idx = u*NTH + tid) -> source validity / element offset / LDS address
  auto chunk_off = [&](int s, int idx, bool& ok, bool& own) -> size_t {
    int b = s / C::NS;
    const int row0 = (s - b * C::NS) * ROWS;
    int pc_ = idx / C::CH;
    if (NIMG > 1) {
      const int img = pc_ / (C::GH * WP);
      pc_ -= img * (C::GH * WP);
      b = s * NIMG + img;
    }
    const int gh = pc_ / WP, gw = pc_ - gh * WP;
    const int h = row0 + gh - 1, w = gw - 1;
    ok = idx < TOTAL && (unsigned)h < (unsigned)C::H && (unsigned)w < (unsigned)W;
    own = ok && gh >= 1 && gh <= ROWS;  // a pixel of this strip's own rows (halo rows belong to the neighbours)
    return ((size_t)(b * C::H + h) * W + w) * (size_t)p.lda + kco() + ch * 8;
  };
  auto chunk_store = [&](int idx, U128 x, U128 x2, bool ok, bool own, size_t off) {
    if (idx < TOTAL) {
      int pc_ = idx / C::CH;
      int ioff = 0;
      if (NIMG > 1) {
        const int img = pc_ / (C::GH * WP);
        pc_ -= img * (C::GH * WP);
        ioff = img * C::ISTR;
      }
      const int gh = pc_ / WP, gw = pc_ - gh * WP;
      if ((PRO == FR_PRO_BN || PRO == FR_PRO_PRELU) && ok) {
        // pro2 (frhip_internal.h): the prologue on packed pairs, 5 / 7 vector instructions per dword where the C form below
        // compiles to 11-12 (round 6: the strip's vector work is what its load phase costs beside the other waves)
        constexpr int P2 = PRO == FR_PRO_BN ? FR_PRO_BN : FR_PRO_PRELU;
        x.x = pro2<P2>(x.x, pa[0], pb[0], pa[1], pb[1]);
        x.y = pro2<P2>(x.y, pa[2], pb[2], pa[3], pb[3]);
        x.z = pro2<P2>(x.z, pa[4], pb[4], pa[5], pb[5]);
        x.w = pro2<P2>(x.w, pa[6], pb[6], pa[7], pb[7]);
      } else if (PRO != FR_PRO_NONE && ok) {
        float f[8];
        unpack16<bf16_t>(x, f);
        if (RES) {
          float f2[8];
          unpack16<bf16_t>(x2, f2);
#pragma unroll
          for (int j = 0; j < 8; ++j) {  // the arithmetic of fr_bn_apply (res_kind 1 [, se])
            f[j] = fmaf(f[j], pa[j], pb[j]);
            f[j] = PRO == FR_PRO_RESBN_SE ? fmaf(f[j], pg[j], f2[j]) : f[j] + f2[j];
          }
          x = pack16<bf16_t>(f);
          if (own && nh == 0 && pro_out) st16(pro_out + off, x);
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
      st16(smem + ioff + gh * C::RSTR + gw * C::PSTR + ch * 16, x);
    }
  };
  auto load_now = [&](int s) {  // stream the strip through UNR registers at a time
    // 4-wave workgroups (two per CU): the whole strip in ONE batch of <= 18 loads per thread.  In-kernel stamps
    // (tools/stamps.py) showed the two batches of 8 of a 58-KB strip taking 5.4 us -- two HBM round trips with too
    // few bytes in flight per CU -- next to 5.6 us of MFMA loop.
    constexpr int PER = (TOTAL + NTH - 1) / NTH;
#ifndef FRHIP_STRIP_LOAD_BATCH
#define FRHIP_STRIP_LOAD_BATCH 18
#endif
    // (channel stages: the accumulators are live across the load -- one batch only beside small accumulator tiles)
    constexpr bool ONE = (NW == 4 && PER <= 18) || (PER <= FRHIP_STRIP_LOAD_BATCH && (KSPL == 1 || C::TM * C::TN * 4 <= 64));
    constexpr int UNR1 = ONE ? PER : 8;
    constexpr int UNR = (TWO && UNR1 > 10) ? (UNR1 + 1) / 2 : UNR1;  // two sources: half the chunks per batch
    load_pro(s);
    for (int base = 0; base < TOTAL; base += NTH * UNR) {
      U128 v[UNR], v2[TWO ? UNR : 1];
      bool ok[UNR], own[UNR];
      size_t off[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        off[u] = chunk_off(s, base + u * NTH + tid, ok[u], own[u]);
        v[u] = ok[u] ? ld16(src + off[u]) : zero16();
        if (TWO) v2[u] = ok[u] ? ld16(src2 + off[u]) : zero16();
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        chunk_store(base + u * NTH + tid, v[u], TWO ? v2[u] : v[u], ok[u], own[u], off[u]);
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = wn * C::TN * 16;
  const bf16_t* wrow[C::TN];
  const int flip = p.mode;
  constexpr int OCH = COUT / 8;

  {
#pragma unroll
    for (int j = 0; j < C::TN; ++j) wrow[j] = wgt + (size_t)(ncol0 + n0 + j * 16 + fr) * 9 * CIN + fq * 8;
    // w_frag (round 6, ABI v7): the weights in MFMA-fragment order [N / 16][tap][K / 32][lane][8] -- the 64 lanes of a weight
    // load read 1024 contiguous bytes (eight whole 128-byte lines) instead of 64 bytes of each of 16 rows 9 * CIN * 2 bytes
    // apart (half of every line fetched, the other half wanted nine taps later and long evicted): 0.0609 -> 0.0560 ms at
    // 256 -> 256 @14, 0.0701 -> 0.0641 at 128 @28, 0.0662 -> 0.0632 at 512 @7 (tools/kbench.py, B = 256)
    const int wsh = p.w_frag ? 4 : 0;
    if (wsh) {
#pragma unroll
      for (int j = 0; j < C::TN; ++j) wrow[j] = wgt + (size_t)((ncol0 + n0) / 16 + j) * 16 * 9 * CIN + lane * 8;
    }
    const int epi = p.epi;
    const bool stats = epi == FR_EPI_STATS || epi == FR_EPI_PRELU_BWD || epi == FR_EPI_BNBWD || epi == FR_EPI_STATS_X;
    const int b = NIMG > 1 ? s * NIMG : s / C::NS;  // first image of the strip
    const int row0 = NIMG > 1 ? 0 : (s - b * C::NS) * ROWS;
    f32x4 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (kc = 0; kc < KSPL; ++kc) {
    // ---------------------------------------------------------------- strip -> LDS (prologue applied once)
    __syncthreads();  // the previous strip's output tile has left LDS / the previous channel stage has been consumed
    FR_STAMP(0);
    load_now(s);
    FR_STAMP(1);
    __syncthreads();
    FR_STAMP(2);

    // ---------------------------------------------------------------- main loop: 9 taps x CK/32, no barriers
    int abase[C::TM];  // LDS byte address of this lane's fragment for tap (0,0), channel chunk c0 = 0
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
      int m = (wm * C::TM + i) * 16 + fr;
      m = m < C::M ? m : 0;
      int ioff = 0;
      if (NIMG > 1) {
        const int img = m / (ROWS * W);
        m -= img * (ROWS * W);
        ioff = img * C::ISTR;
      }
      const int h = m / W, w = m - h * W;
      abase[i] = ioff + h * C::RSTR + w * C::PSTR + fq * 16;
    }

    // B fragments: 3-deep register ring: tap t's weights are requested during tap t-3 (address clamped instead of
    // branching at the end, so the body stays one basic block and hipcc's counted vmcnt waits stay exact).  A
    // fragments: D-deep ring of ds_read_b128, refilled D (tap, tile) steps ahead of use; sched_group_barrier pins
    // the 1 read : TN MFMA interleave.
    s16x8 bq[3][C::TN];
    auto load_b = [&](int slot, int c0, int tap) {
      const int wt = flip ? 8 - tap : tap;
#pragma unroll
      for (int j = 0; j < C::TN; ++j) bq[slot][j] = *reinterpret_cast<const s16x8*>(wrow[j] + ((wt * CIN + kco() + c0) << wsh));
    };
    constexpr int NSTEP = 9 * C::TM;
    // ring depth must divide NSTEP (slots line up across the channel loop): 9 when registers allow, else 3
    constexpr int D = (KSPL > 1 || C::TM * C::TN * 4 + 3 * C::TN * 4 > 100) ? 3 : 9;
    s16x8 ring[D];
    auto a_addr = [&](int step) -> const s16x8* {  // step in [0, 2*NSTEP): second half = next 32 input channels
      const int cadd = step >= NSTEP ? 64 : 0;
      const int st = step >= NSTEP ? step - NSTEP : step;
      const int tap = st / C::TM, i = st - tap * C::TM;
      return reinterpret_cast<const s16x8*>(smem + abase[i] + (tap / 3) * C::RSTR + (tap % 3) * C::PSTR + cadd);
    };
#ifdef FRHIP_STRIP_YOUNG_PRIO
    // the second-dispatched half of an 8-wave workgroup loses every issue arbitration against its older SIMD partner
    // (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): it finishes the K loop ~7 us after it, alone on its SIMD
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(FRHIP_STRIP_YOUNG_PRIO);
#endif
    load_b(0, 0, 0);
    load_b(1, 0, 1);
    load_b(2, 0, 2);
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = *a_addr(d);
    for (int c0 = 0; c0 < CK; c0 += 32) {
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int tap = st / C::TM, i = st - tap * C::TM;
        const int slot = tap % 3;
        const s16x8 a = ring[st % D];
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[slot][j], a, acc[i][j], 0, 0, 0);  // = (W X^T) tile
        ring[st % D] = *a_addr(st + D);  // past the last channel chunk this reads (never used) bytes inside LDS
        if (i == C::TM - 1) {
          int nt = tap + 3, nc = c0;
          if (nt >= 9) {
            nt -= 9;
            nc += 32;
          }
          nc = nc < CK ? nc : CK - 32;  // clamp instead of branching: the count of loads in flight stays static
#ifndef FRHIP_EXP_NOW  // timing ablation (tools/stamps.py): the loop without its weight stream (results are wrong)
          load_b(slot, nc, nt);
#endif
        }
        __builtin_amdgcn_sched_group_barrier(0x008, C::TN, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
        // pin the weight requests three taps ahead of their use: without the fence the scheduler sinks the loads
        // down to their consumer (shorter live ranges) and every tap then waits a full L2 round trip
        if (i == C::TM - 1) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < C::TM; ++i) abase[i] += 64;  // next 32 input channels
    }
    }  // channel stages

    // ---------------------------------------------------------------- epilogue
    FR_STAMP(3);
    const size_t rowbase = (size_t)(b * C::H + row0) * W;
    // The aux tile (the PReLU input / BN input of the fused backward epilogues) is requested in ONE batch, before the
    // barrier that ends the K loop: as a load -> wait -> LDS-store loop it cost one HBM round trip per 16 bytes of a
    // thread (13 of them, ~7 us of a 55-us 14x14 launch); the registers are those of the dead fragment rings.
    constexpr int NAUX = (C::M * OCH + NTH - 1) / NTH;
    const bool has_aux = epi == FR_EPI_PRELU_BWD || epi == FR_EPI_BNBWD || epi == FR_EPI_BIAS_RES || epi == FR_EPI_STATS_X;
    U128 av[NAUX];
    if (has_aux) {
      const bf16_t* __restrict__ aux = reinterpret_cast<const bf16_t*>(p.aux);
#pragma unroll
      for (int u = 0; u < NAUX; ++u) {
        int idx = u * NTH + tid;
        idx = idx < C::M * OCH ? idx : C::M * OCH - 1;  // clamp: the batch stays one basic block
        const int r = idx / OCH, c8 = idx - r * OCH;
        av[u] = ld16(aux + (rowbase + r) * (size_t)p.ldaux + ncol0 + c8 * 8);
      }
    }
    __syncthreads();  // every wave is done with the input strip; LDS is now the output tile
    FR_STAMP(4);
    if (has_aux) {
#pragma unroll
      for (int u = 0; u < NAUX; ++u) {
        const int idx = u * NTH + tid;
        if (idx < C::M * OCH) {
          const int r = idx / OCH, c8 = idx - r * OCH;
          st16(smem + r * C::OSTR + c8 * 16, av[u]);
        }
      }
      __syncthreads();
    }
    // The MFMAs ran with the weights as the A operand, so a lane holds, per 16x16 tile, FOUR CONSECUTIVE CHANNELS
    // (n = fq*4 + r) of ONE pixel (m = fr): one 8-byte LDS access per tile instead of four 2-byte ones, and the
    // 16 lanes of a ds_write_b64 lane group hit 16 different rows on disjoint banks (row stride = 4 banks mod 64).
    float* red = reinterpret_cast<float*>(smem + C::OUT_BYTES);  // [WM][NV][COUT] column sums, behind the output tile
    const int NV = epi == FR_EPI_STATS_X ? 3 : 2;
    // The epilogue kind is a run-time argument, but inside the per-element loops it must be a compile-time constant:
    // with `epi` tested per element the compiler emitted a scalar branch per accumulator (8000 instructions, ~10 us).
    auto cells = [&](auto tag) {
      constexpr int E = decltype(tag)::value;
      constexpr bool AUX = E == FR_EPI_PRELU_BWD || E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES || E == FR_EPI_STATS_X;
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
        const int m = (wm * C::TM + i) * 16 + fr;
        if (wm * C::TM + i >= C::MT || m >= C::M) continue;
  #pragma unroll
        for (int j = 0; j < C::TN; ++j) {
          uint2* cell = reinterpret_cast<uint2*>(smem + m * C::OSTR + (n0 + j * 16 + fq * 4) * 2);
          float v[4], x[4];
  #pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
          if (AUX) {
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
          *cell = o;
        }
      }
      // column sums: fold the 16 pixel lanes (fr), then the WM row groups through LDS (behind the output tile)
      if (E != FR_EPI_STORE && E != FR_EPI_BIAS_RES) {
  #pragma unroll
        for (int j = 0; j < C::TN; ++j)
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            constexpr int V = E == FR_EPI_STATS_X ? 3 : 2;
            float a = s0[j][r], c = s1[j][r], d = E == FR_EPI_STATS_X ? s2[j][r] : 0.f;
  #pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
              a += __shfl_xor(a, o, 64);
              c += __shfl_xor(c, o, 64);
              if (E == FR_EPI_STATS_X) d += __shfl_xor(d, o, 64);
            }
            if (fr == 0) {
              red[(wm * V + 0) * COUT + n0 + j * 16 + fq * 4 + r] = a;
              red[(wm * V + 1) * COUT + n0 + j * 16 + fq * 4 + r] = c;
              if (E == FR_EPI_STATS_X) red[(wm * V + 2) * COUT + n0 + j * 16 + fq * 4 + r] = d;
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
    __syncthreads();
    FR_STAMP(5);
    {  // all LDS reads of a thread first, then its stores (as a read -> wait -> store loop: one LDS latency per 16 bytes)
      U128 ov[NAUX];
#pragma unroll
      for (int u = 0; u < NAUX; ++u) {
        int idx = u * NTH + tid;
        idx = idx < C::M * OCH ? idx : C::M * OCH - 1;
        const int r = idx / OCH, c8 = idx - r * OCH;
        ov[u] = ld16(smem + r * C::OSTR + c8 * 16);
      }
#pragma unroll
      for (int u = 0; u < NAUX; ++u) {
        const int idx = u * NTH + tid;
        if (idx < C::M * OCH) {
          const int r = idx / OCH, c8 = idx - r * OCH;
          st16(out + (rowbase + r) * (size_t)p.ldc + ncol0 + c8 * 8, ov[u]);
        }
      }
    }
    FR_STAMP(6);
#ifdef FRHIP_STAMPS
    if (tid == 0 && fr_stamp_buf) {
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      fr_stamp_buf[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    if (stats) {
      for (int c = tid; c < NV * COUT; c += NTH) {
        const int k = c / COUT, n = c - k * COUT;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < C::WM; ++g) t += red[(g * NV + k) * COUT + n];
        st_part(p.part + ((size_t)sblk * NV + k) * (COUT * NSPL) + ncol0 + n, t);
      }
    }
  }
}

// FRHIP_XCD_ORDER=0: workgroups take strips in dispatch order (A/B switch for tools/kbench.py)
static int xcd_order() {
  static const int* v = fr_option_slot("FRHIP_XCD_ORDER", 1);
  return *v != 0;
}

template <int CIN, int COUT, int W, int ROWS, int WN, int NW, int NSPL, int PRO, int NIMG = 1, int KSPL = 1>
int launch(const FrConvArgs& a, hipStream_t st) {
  using C = SC<CIN, COUT, W, ROWS, WN, NW, NIMG, KSPL>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv3x3_strip_kernel<CIN, COUT, W, ROWS, WN, NW, NSPL, PRO, NIMG, KSPL>),
        hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    fr_attr_done(attr_done);
  }
  const int strips = a.B * C::NS / NIMG;
  if (a.epi == FR_EPI_STATS_X && (!a.part || !a.aux))
    FR_UNSUPPORTED("fr_conv3x3_strip: FR_EPI_STATS_X needs part and aux (its rows go to fr_bn_finalize_res)");
  FR_LAUNCH_KERNEL((conv3x3_strip_kernel<CIN, COUT, W, ROWS, WN, NW, NSPL, PRO, NIMG, KSPL>),
                     dim3(strips * NSPL), dim3(C::NTH), C::LDS, st, a, xcd_order());
  FR_LAUNCH_CHECK();
}

template <int CIN, int COUT, int W, int ROWS, int WN, int NW = 8, int NSPL = 1, int NIMG = 1, int KSPL = 1>
int by_pro(const FrConvArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return launch<CIN, COUT, W, ROWS, WN, NW, NSPL, FR_PRO_NONE, NIMG, KSPL>(a, st);
    case FR_PRO_BN: return launch<CIN, COUT, W, ROWS, WN, NW, NSPL, FR_PRO_BN, NIMG, KSPL>(a, st);
    case FR_PRO_PRELU: return launch<CIN, COUT, W, ROWS, WN, NW, NSPL, FR_PRO_PRELU, NIMG, KSPL>(a, st);
    case FR_PRO_RESBN:  // conv1 of a unit behind an identity unit (both sources have this layer's input channels)
      if (!a.src2 || !a.pro_a || !a.pro_b || !a.pro_c || !a.pro_d || !a.pro_out || a.mode != 0)
        FR_UNSUPPORTED("fr_conv3x3_strip: FR_PRO_RESBN needs src2, pro_a ... pro_d, pro_out and mode 0");
      return launch<CIN, COUT, W, ROWS, WN, NW, NSPL, FR_PRO_RESBN, NIMG, KSPL>(a, st);
    case FR_PRO_RESBN_SE:
      if constexpr (NIMG == 1) {
        if (!a.src2 || !a.pro_a || !a.pro_b || !a.pro_c || !a.pro_d || !a.pro_g || !a.pro_out || a.mode != 0)
          FR_UNSUPPORTED("fr_conv3x3_strip: FR_PRO_RESBN_SE needs src2, pro_a ... pro_d, pro_g, pro_out and mode 0");
        return launch<CIN, COUT, W, ROWS, WN, NW, NSPL, FR_PRO_RESBN_SE, NIMG, KSPL>(a, st);
      } else {
        FR_UNSUPPORTED("fr_conv3x3_strip: FR_PRO_RESBN_SE is served by the one-image-per-workgroup instances only");
      }
  }
  FR_UNSUPPORTED("fr_conv3x3_strip: unknown prologue");
}

}  // namespace

// Shape table.  (Rounds 1-3 carried five generations of it behind FRHIP_STRIP_VARIANT=0..4 for same-box A/B runs: the
// one-workgroup-per-CU instances of the 64- and 128-channel layers, the 7 x 2 tile instances of the stage-entry shapes and
// the one- / two-image 7x7 instances as the DEFAULT.  Table 4 -- 4-wave workgroups, two resident per CU, at 64 / 128
// channels; one tile column per wave at the stage entries; 2 - 4 images per workgroup at 7x7 -- ran 0.8 ms per step faster
// than table 0 and has a round of green tests behind it; the older generations were removed in round 4, their measurements
// stay in the comments below and in DESIGN.md section 3.)
// fewer whole-image workgroups than ~3/4 of the CUs: prefer the instances that split an image over two workgroups
// FRHIP_SPLIT_STRIPS=1 forces them at any batch: 2 x B workgroups of half the output channels instead of B whole-image
// ones, so that CUs taken by another resident kernel (RCCL's all-reduce under data parallelism) cost a proportional
// share instead of a whole second round of workgroups (A/B switch for multi-GPU runs; slower stand-alone).
static bool small_batch(int B) {
  static const int* force = fr_option_slot("FRHIP_SPLIT_STRIPS", 0);
  return B <= 160 || *force == 1;
}

// rows per strip for a shape (0 = not served)
static int strip_rows(int Cin, int Cout, int W) {
#define SHAPE(ci, co, w, rows) \
  if (Cin == ci && Cout == co && W == w) return rows;
  SHAPE(64, 64, 112, 2)
  SHAPE(64, 64, 56, 7)
  SHAPE(64, 128, 56, 4)
  SHAPE(128, 64, 56, 7)
  SHAPE(128, 128, 28, 7)
  SHAPE(128, 256, 28, 7)
  SHAPE(256, 128, 28, 7)
  SHAPE(256, 256, 14, 14)
  SHAPE(256, 512, 14, 14)
  SHAPE(512, 256, 14, 14)
  SHAPE(512, 512, 7, 7)
#undef SHAPE
  return 0;
}

// The two-source prologues (FR_PRO_RESBN[_SE]) and the cross-moment epilogue (FR_EPI_STATS_X) exist on the LDS-strip instances
// of the square layers; the 64-channel layers run on the rolling-window kernel, which takes neither.
extern "C" int fr_conv3x3_strip_serves_resbn(int B, int C, int W) {
  if (C == 64 && (W == 112 || W == 56) && fr_roll64_enabled()) return 0;
  return fr_conv3x3_strip_parts(B, C, C, W, FR_EPI_PRELU_BWD) > 0 ? 1 : 0;
}

extern "C" int fr_conv3x3_strip_takes_frag(int B, int Cin, int Cout, int W) {
  if (Cin == 64 && Cout == 64 && (W == 112 || W == 56) && fr_roll64_enabled()) return 0;
  return fr_conv3x3_strip_parts(B, Cin, Cout, W, FR_EPI_STORE) > 0 ? 1 : 0;
}

// Number of partial rows the kernel writes into `part` (= workgroups) for a supported shape, 0 if unsupported.
extern "C" int fr_conv3x3_strip_parts(int B, int Cin, int Cout, int W, int epi) {
  if (Cin == 256 && Cout == 512 && W == 14 && epi != FR_EPI_STORE) return 0;
  if (Cin == 64 && Cout == 64 && (W == 112 || W == 56) && fr_roll64_enabled()) return fr_roll64_parts(B, W);
  const int rows = strip_rows(Cin, Cout, W);
  if (Cin == 512 && Cout == 512 && W == 7 && B % 2 == 0 && small_batch(B)) return B / 2;
  if (Cin == 512 && Cout == 512 && W == 7 && B % 4 == 0) return B / 4;  // four images per strip
  if (Cin == 512 && Cout == 512 && W == 7 && B % 2 == 0) return B / 2;  // two images per strip
  return rows ? B * (W / rows) : 0;
}

extern "C" int fr_conv3x3_strip(const FrConvArgs* args, void* stream) {
  const FrConvArgs& a = *args;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.RH != a.SH || a.RW != a.SW || a.SH != a.SW ||
      a.out_f32 || a.splitk > 1 || a.bias || a.epi == FR_EPI_MARGIN || a.epi == FR_EPI_ATOMIC)
    FR_UNSUPPORTED("fr_conv3x3_strip: only square stride-1 3x3 bf16 convolutions");
  if (a.lda % 8 || a.ldc % 8 || (a.aux && a.ldaux % 8)) FR_UNSUPPORTED("fr_conv3x3_strip: strides must be 16-byte multiples");
  if (a.SC == 64 && a.N == 64 && (a.SW == 112 || a.SW == 56) && fr_roll64_enabled()) {
    if (a.w_frag) FR_UNSUPPORTED("fr_conv3x3_strip: the 64-channel rolling-window kernel takes the plain weight layout (w_frag)");
    return fr_roll64_launch(a, st);
  }
#define SHAPE(ci, co, w, rows, wn, nw) \
  if (a.SC == ci && a.N == co && a.SW == w) return by_pro<ci, co, w, rows, wn, nw>(a, st);
  // (round 3: 128 -> 128 @28 as 8 waves x (25 tiles x 1 column) on half images, one workgroup per CU, half the weight stream:
  // 0.143 / 0.151 ms forward / data gradient against 0.068 / 0.070 for the two resident 4-wave workgroups below)
  // 4-wave workgroups, two resident per CU -- measured (tools/kbench.py, B=256) 1.1-1.45x over the 8-wave one-per-CU
  // instances <64,64,112,4,2,8>, <64,64,56,7,2,8>, <128,128,28,14,4,8> of round 1 (removed in round 4)
  SHAPE(64, 64, 112, 2, 2, 4)
  SHAPE(64, 64, 56, 7, 2, 4)
  SHAPE(128, 128, 28, 7, 4, 4)
  // 256 -> 256 @14 (half of all FLOPs): 8 waves x (13 x 2) accumulator tiles on the whole image.  Measured and rejected
  // at B=256 (tools/kbench.py): 4-wave half-height strips 0.101 ms, the same with the channels split over two
  // workgroups (NSPL = 2, no spills) 0.093 ms, 8 waves x (7 x 4) tiles 0.114 ms (spills) -- against 0.062 ms here.
  // Round 3: two 4-wave workgroups per image and CU (<256, 128, 14, 14, 4, 4, 2, 1, 2>: 128 output channels each, 128
  // input channels resident at a time, 77 KB of LDS, so that one's strip load / epilogue overlaps the other's MFMAs):
  // 0.068 / 0.071 / 0.075 ms (forward BN / forward PReLU / data gradient) against 0.057 / 0.061 / 0.064 -- the strip is
  // loaded and its prologue applied twice per image, 176-200 B of scratch; a 9-deep fragment ring here: 0.0623 vs 0.0596
  // (256 registers + scratch); a 6-deep one over two channel chunks per trip: 84 B of scratch; s_setprio 1 for the younger
  // half of the waves: no effect.  None instantiated.
  // Round 6 (fragment-order weights; stamps: wave 0's K loop 24.5 us, then 14.4 us waiting for its SIMD partner -- 38.9 us for
  // 27-31 us of MFMA work at the clock the loop holds): FOUR waves, one per SIMD, 13 x 4 tiles each (208 accumulator registers,
  // A ring 13 deep, <256, 256, 14, 14, 4, 4> with __launch_bounds__(256, 1)): hipcc splits the 512 registers 256 + 256, moves
  // 228 accumulator registers per 468 MFMAs through v_accvgpr_* and spills 136-604 B; K loop 41.8 us, load 6.2 (4.4), cells
  // 4.7-26 (1.5-5.9): 0.0586 / 0.0713 / 0.0847 ms (forward BN / PReLU+stats / data gradient) against 0.0501 / 0.0538 / 0.0583.
  // Not instantiated.
  // small batches (fewer whole-image workgroups than CUs): split the output channels over two workgroups per image
  // (measured at B = 128: 0.034 instead of 0.049 ms per launch -- IR-SE-101 trains at 128 images per GPU)
  if (a.SC == 256 && a.N == 256 && a.SW == 14 && small_batch(a.B)) return by_pro<256, 128, 14, 14, 8, 8, 2>(a, st);
  SHAPE(256, 256, 14, 14, 8, 8)
  // 64 -> 128 @56 (one forward launch per step): 4-row strips, two resident 4-wave workgroups per CU: 0.170-0.174 -> 0.159-0.160 ms
  // against the 7-row 8-wave instance <64,128,56,7,4,8> (removed)
  SHAPE(64, 128, 56, 4, 4, 4)
  // Round 3: what a wave pays per MFMA is its private weight stream (16 B per lane from L2 for every (tap, 32 channels,
  // 16 output channels)): an instance is fast when one weight fragment feeds ~13 M tiles.  The data gradients of the three
  // stage-entry convolutions ran 7 tiles x 2 columns per wave (<128,64,56,7,2,8>, <256,128,28,7,4,8>, <512,256,14,7,8,8>:
  // removed in round 4); with ONE 16-channel column per wave and 13 tiles (128 -> 64: 4 x 2 waves; 256 -> 128: 8 x 1;
  // 512 -> 256: the whole image in two channel stages, output channels over two workgroups) the same launches take
  // 0.254 -> 0.202, 0.202 -> 0.150 and 0.177 -> 0.132 ms (same box), the step 0.2 ms less.
  SHAPE(128, 64, 56, 7, 4, 8)
  SHAPE(256, 128, 28, 7, 8, 8)
  if (a.SC == 512 && a.N == 256 && a.SW == 14) return by_pro<512, 128, 14, 14, 8, 8, 2, 1, 2>(a, st);
  SHAPE(128, 256, 28, 7, 8, 8)
  // 512 -> 512 @7: two images per workgroup, 256 resident input channels at a time, output channels split over two
  // workgroups: every weight fragment now feeds 98 pixels instead of 49 and the M tiles are 12 % instead of 23 % padding
  // Round 3: FOUR images per workgroup, 128 resident input channels at a time (four stages of the strip), output channels
  // split over four workgroups: 196 pixels (13 M tiles, 6 % padding) per weight fragment, 13 MFMAs per 16-byte weight
  // load instead of 7 x 2 per 2 loads, half the weight stream per launch (302 MB): 0.092 -> 0.073 ms forward, 0.095 -> 0.070
  // data gradient at B = 256 (tools/kbench.py, same box), no scratch, 100 KB of LDS.
  // ... at small batches (IR-SE-101 trains at 128 images per GPU: 32 four-image strips x 4 = 128 workgroups would leave half
  // the CUs idle) two images per workgroup with the same split: B / 2 x 4 workgroups
  if (a.SC == 512 && a.N == 512 && a.SW == 7 && a.B % 2 == 0 && small_batch(a.B))
    return by_pro<512, 128, 7, 7, 8, 8, 4, 2, 2>(a, st);
  if (a.SC == 512 && a.N == 512 && a.SW == 7 && a.B % 4 == 0) return by_pro<512, 128, 7, 7, 8, 8, 4, 4, 4>(a, st);
  if (a.SC == 512 && a.N == 512 && a.SW == 7 && a.B % 2 == 0) return by_pro<512, 256, 7, 7, 8, 8, 2, 2, 2>(a, st);
  SHAPE(512, 512, 7, 7, 8, 8)
#undef SHAPE
  if (a.SC == 256 && a.N == 512 && a.SW == 14 && a.epi == FR_EPI_STORE) {
    // 256 -> 512 @14 (one layer per network): two passes over 256 output channels each reuse the 256x256 instance
    // (a 512-wide accumulator tile would spill); the strip is simply loaded twice
    FrConvArgs h = a;
    h.N = 256;
    for (int half = 0; half < 2; ++half) {
      h.w = reinterpret_cast<const bf16_t*>(a.w) + (size_t)half * 256 * 9 * 256;
      h.out = reinterpret_cast<bf16_t*>(a.out) + half * 256;
      if ((a.pro == FR_PRO_RESBN || a.pro == FR_PRO_RESBN_SE) && half == 1) {  // the first pass has materialised the residual sum: plain BN1 on it
        h.pro = FR_PRO_BN;
        h.src = a.pro_out;
        h.pro_a = a.pro_c;
        h.pro_b = a.pro_d;
      }
      const int rc = small_batch(a.B) ? by_pro<256, 128, 14, 14, 8, 8, 2>(h, st) : by_pro<256, 256, 14, 14, 8, 8>(h, st);
      if (rc) return rc;
    }
    return 0;
  }
  FR_UNSUPPORTED("fr_conv3x3_strip: shape not in the strip table");
}
