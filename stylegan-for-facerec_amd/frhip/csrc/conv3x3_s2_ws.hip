// frhip -- the stride-2 3x3 convolution of a stage entry (bottleneck_IR conv2 with stride 2, backbone/model_irse.py:57-59),
// FORWARD, 128 / 256 / 512 channels, warp-specialised (bf16, gfx950; round 6).
//
// conv3x3_s2_strip.hip stages the four parity planes of the high-resolution input one after the other and runs the taps of
// a plane between two barriers; its in-kernel stamps (tools/stamps_s2.py, 128 -> 128 @28, 27.2 us per workgroup) read
//   plane loads 11.0 us (40 %)   tap lists 13.2 us (48 %)   epilogue 2.9 us (11 %)
// -- eight waves that all load, all wait, all compute: nothing runs under the HBM round trip of a plane, and a register
// prefetch from the computing waves does not help (vector-memory operations of a wave return in order: a plane load in flight
// stalls every later weight wait, the finding behind conv3x3_roll64.hip).  Here the roles are split, as in the rolling-window
// kernels:
//   waves 0-3 (one per SIMD) issue nothing but ds_read_b128 / MFMA / weight loads: wave w owns 32 of the workgroup's 128
//             output channels for ALL 196 pixels -- 13 x 2 accumulator tiles, so a pixel fragment read from LDS feeds two MFMAs
//             (the strip instances read one per MFMA: LDS-bound at the MFMA rate);
//   waves 4-7 move the data: while the computing waves run the taps of phase p out of one plane buffer, they load phase p + 1
//             (one batch of <= 16 16-byte loads per thread), apply the BN / PReLU prologue once per element and commit it to
//             the OTHER buffer.  One workgroup barrier per phase.
// A phase = (parity plane, 128 input channels): 4 / 8 / 16 phases at 128 / 256 / 512 channels, every plane image is
// <= 74 KB, two of them fit the CU's LDS at every width, and every instance has M = 196 pixels per weight fragment
// (7 rows @28, the whole image @14, four images @7) with the output channels split over 1 / 2 / 4 workgroups.
// Same layouts, partial-sum rows and arithmetic (K order: plane, channel stage, tap, 32-channel chunk) as
// fr_conv3x3_s2_strip mode 0; dispatched from there.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "frhip_internal.h"

#ifdef FRHIP_STAMPS
// Diagnostic build only (make stamps): per workgroup 48 slots of s_memrealtime (100 MHz) -- slots 0..23: computing wave 0
// (0 start, 1 phase 0 resident, 2 + ph taps of phase ph done, then cells done, stores done); slots 24..47: data-moving wave 4
// (24 start, 25 + ph phase ph committed).  tools/stamps_s2.py --ws.
__device__ unsigned long long* fr_stamp_buf_ws = nullptr;
extern "C" int fr_debug_set_stamp_buffer_ws(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fr_stamp_buf_ws), &dev_ptr, sizeof(dev_ptr));
}
#define WS_STAMP(cond, k)                                                                                       \
  do {                                                                                                          \
    if ((cond) && fr_stamp_buf_ws) fr_stamp_buf_ws[(size_t)blockIdx.x * 48 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define WS_STAMP(cond, k)
#endif

namespace {

constexpr int CK = 128;    // input channels per phase
constexpr int CW = 128;    // output channels per workgroup
constexpr int NCW = 4;     // computing waves
constexpr int NLT = 256;   // data-moving threads

template <int WL, int ROWS, int NIMG>
struct WS {
  static constexpr int HL = WL;
  static constexpr int GW = WL + 1, GH = ROWS + 1;          // one halo column (left) and row (top), see conv3x3_s2_strip.hip
  static constexpr int CH = CK / 8;
  static constexpr int PSTR = CK * 2 + 32;                  // conflict-free pixel stride (conv3x3_strip.hip)
  static constexpr bool TIGHT = NIMG > 1;                   // rows of stacked images start inside the last pixel's padding
  static constexpr int RSTR = TIGHT ? GW * PSTR - PSTR % 256 : GW * PSTR + (256 - PSTR % 256) % 256;
  static constexpr int ISTR = GH * RSTR;
  static constexpr int BUF = (NIMG * ISTR + 1024 + 1023) / 1024 * 1024;  // + slack for ring reads past the last chunk
  static constexpr int M = NIMG * ROWS * WL;
  static constexpr int TM = (M + 15) / 16, TN = CW / 16 / NCW;
  static constexpr int OSTR = CW * 2 + 16;
  static constexpr int OUT_BYTES = (M * OSTR + 15) / 16 * 16;
  static constexpr int LDS = 2 * BUF;
  static constexpr int NS = HL / ROWS;
  static constexpr int PLANE = GH * GW;
  static constexpr int TOTAL = NIMG * PLANE * CH;
  static constexpr int PER = (TOTAL + NLT - 1) / NLT;
  static_assert(HL % ROWS == 0 && (NIMG == 1 || ROWS == WL), "strips divide the image; stacked images are whole");
  static_assert(OUT_BYTES <= BUF && LDS <= 160 * 1024 && PER <= 16, "LDS / register budget");
  static_assert(TM == 13 && TN == 2, "196 pixels x 32 channels per computing wave");
};

// taps of parity plane P = 2*ph + pw (conv3x3_s2_strip.hip, Taps<0, P>)
template <int P>
struct Taps {
  static constexpr int PH = P >> 1, PW = P & 1;
  static constexpr int NWD = PW ? 2 : 1, NT = (PH ? 2 : 1) * NWD;
  static constexpr int kh(int t) { return PH ? 2 * (t / NWD) : 1; }
  static constexpr int kw(int t) { return PW ? 2 * (t % NWD) : 1; }
  static constexpr int ktap(int t) { return kh(t) * 3 + kw(t); }
  static constexpr int roff(int t) { return PH ? t / NWD : 1; }
  static constexpr int coff(int t) { return PW ? t % NWD : 1; }
};

// acc += taps of plane P over the CK resident channels; A from the plane buffer, B (weights) from global / L2
template <class C, int CIN, int P>
__device__ __forceinline__ void mma_taps(const char* buf, const int (&abase0)[C::TM], f32x4 (&acc)[C::TM][C::TN],
                                         const bf16_t* const (&wrow)[C::TN], const int wsh) {
  using T = Taps<P>;
  constexpr int NT = T::NT;
#ifndef FRHIP_S2WS_DB
#define FRHIP_S2WS_DB 4
#endif
  constexpr int DB = FRHIP_S2WS_DB;    // weight ring: requested DB tap-steps (26 MFMAs each) ahead -- one wave per SIMD has no
                                       // partner to hide an L2 round trip behind
  constexpr int UC = DB % NT == 0 ? DB / NT : 1;  // 32-channel chunks per unrolled body: a body holds a multiple of DB tap-steps
  constexpr int QB = UC * NT;
  constexpr int NSTEP = QB * C::TM;
#ifndef FRHIP_S2WS_D
#define FRHIP_S2WS_D 13
#endif
  constexpr int D = FRHIP_S2WS_D;      // pixel-fragment ring: one wave per SIMD has nobody to hide an LDS round trip behind
  static_assert((CK / 32) % UC == 0 && QB % DB == 0 && DB <= QB && NSTEP % D == 0, "bad body shape");
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
  auto a_addr = [&](int step) -> const s16x8* {
    const int wrap = step >= NSTEP ? 1 : 0;
    const int st = wrap ? step - NSTEP : step;
    const int q = st / C::TM, i = st - q * C::TM;
    const int t = q % NT, u = q / NT;
    return reinterpret_cast<const s16x8*>(buf + abase[i] + T::roff(t) * C::RSTR + T::coff(t) * C::PSTR + u * 64 +
                                          wrap * UC * 64);
  };
#pragma unroll
  for (int d = 0; d < DB; ++d) load_b(d, 0, d);
#pragma unroll
  for (int d = 0; d < D; ++d) ring[d] = *a_addr(d);
  for (int c0 = 0; c0 < CK; c0 += 32 * UC) {
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int q = st / C::TM, i = st - q * C::TM;
      const int slot = q % DB;
      const s16x8 a = ring[st % D];
#pragma unroll
      for (int j = 0; j < C::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[slot][j], a, acc[i][j], 0, 0, 0);  // = (W X^T) tile
      ring[st % D] = *a_addr(st + D);  // past the last chunk: (never used) bytes inside the buffer's slack
      if (i == C::TM - 1) {
        int nq = q + DB, nc = c0;
        if (nq >= QB) {
          nq -= QB;
          nc += 32 * UC;
        }
        nc = nc < CK ? nc : CK - 32 * UC;  // clamp instead of branching: the count of loads in flight stays static
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

template <int CIN, int WL, int ROWS, int NIMG, int PRO>
__global__ __launch_bounds__(512) void conv3x3_s2_ws_kernel(const FrConvArgs p, const int xcd) {
  using C = WS<WL, ROWS, NIMG>;
  constexpr int NSPL = CIN / CW, KSPL = CIN / CK, NPH = 4 * KSPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool computing = wave < NCW;
  const bf16_t* __restrict__ src = reinterpret_cast<const bf16_t*>(p.src);
  const bf16_t* __restrict__ wgt = reinterpret_cast<const bf16_t*>(p.w);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(p.out);

  const int lb = xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;  // the NSPL parts of a strip meet in one XCD's L2
  const int nh = NSPL > 1 ? lb % NSPL : 0;
  const int ncol0 = nh * CW;
  const int sblk = NSPL > 1 ? lb / NSPL : lb;
  const int b = NIMG > 1 ? sblk * NIMG : sblk / C::NS;
  const int row0 = NIMG > 1 ? 0 : (sblk - b * C::NS) * ROWS;
  const int epi = p.epi;
  constexpr int OCH = CW / 8;

  if (!computing) {
    // ---------------------------------------------------------------- data-moving waves
#ifndef FRHIP_S2WS_LPRIO
#define FRHIP_S2WS_LPRIO 2
#endif
    // the data-moving waves are the second-dispatched half of the workgroup: at equal priority their vector instructions get
    // the issue slots the computing wave of the SIMD leaves (MI355X_MICROARCH.md, "Two waves per SIMD", items 2 and 4)
    if (FRHIP_S2WS_LPRIO) __builtin_amdgcn_s_setprio(FRHIP_S2WS_LPRIO);
    const int lt = tid - NCW * 64;
    const int ch = lt % C::CH;
    float pa[8], pb[8];
    if (PRO != FR_PRO_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pa[j] = pb[j] = 0.f;
    }
    // this thread's chunks: source offset for plane (0, 0) / channel stage 0, LDS offset, validity -- the same for every phase.
    // Everything below is branch-free (the first version tested `ok` per chunk: hipcc turned every load and every commit into
    // an exec-mask branch with a vmcnt(0) behind it -- 135 branches, one HBM round trip per chunk, 4.5 us per stage): a halo
    // chunk loads the image's first pixel and is zeroed by a select, chunks past the end repeat the last one (same bytes to
    // the same LDS address).
    size_t soff[C::PER];
    int loff[C::PER];
    unsigned okm = 0;
#pragma unroll
    for (int u = 0; u < C::PER; ++u) {
      int idx = u * NLT + lt;
      idx = idx < C::TOTAL ? idx : C::TOTAL - C::CH + ch;  // clamp: the last pixel, this thread's own channel chunk
      int pc = idx / C::CH;
      const int img = NIMG > 1 ? pc / C::PLANE : 0;
      pc -= img * C::PLANE;
      const int gh = pc / C::GW, gw = pc - gh * C::GW;
      const int i = row0 + gh - 1, j = gw - 1;
      const bool ok = i >= 0 && j >= 0;
      okm |= ok ? (1u << u) : 0u;
      const int ii = ok ? i : 0, jj = ok ? j : 0;
      soff[u] = ((size_t)((b + img) * 2 * C::HL + 2 * ii) * (2 * WL) + 2 * jj) * (size_t)p.lda + ch * 8;
      loff[u] = img * C::ISTR + gh * C::RSTR + gw * C::PSTR + ch * 16;
    }
    // Phase ph = (plane, channel stage) -> buffer ph & 1.  The loads of phase ph + 2 are ISSUED (into the register set the
    // commit of phase ph has just freed) before phase ph + 1 is committed: a stage is then prologue + LDS writes only, its HBM
    // round trip has had a whole phase to come back.
    auto issue = [&](int ph, U128 (&v)[C::PER]) {
      const int plane = ph / KSPL, kc = ph - plane * KSPL;
      const size_t poff = ((size_t)(plane >> 1) * (2 * WL) + (plane & 1)) * (size_t)p.lda + kc * CK;
#pragma unroll
      for (int u = 0; u < C::PER; ++u) v[u] = ld16(src + soff[u] + poff);
    };
    // prologue coefficients of phase ph's channel stage: requested BEFORE the next batch of plane loads (vector-memory
    // operations of a wave return in order: behind them, the wait for 16 floats would be a wait for the whole batch)
    auto coefs = [&](int ph) {
      const int kc = ph % KSPL;
      if (PRO != FR_PRO_NONE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pa[j] = p.pro_a[kc * CK + ch * 8 + j];
          if (PRO == FR_PRO_BN) pb[j] = p.pro_b[kc * CK + ch * 8 + j];
        }
      }
    };
    auto commit = [&](int ph, U128 (&v)[C::PER]) {
      char* dst = smem + (ph & 1) * C::BUF;
#pragma unroll
      for (int u = 0; u < C::PER; ++u) {
        U128 x = v[u];
        if (PRO != FR_PRO_NONE) {
          // pro2 (frhip_internal.h): the prologue on packed pairs, written as instructions -- 5 (BN) / 7 (PReLU) vector
          // instructions per dword where the C form compiles to 11-12 (16-bit compares + selects + permutes).  The loader's
          // vector instructions are what a stage costs beside the computing wave of its SIMD (file header).
          x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
          x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
          x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
          x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
        }
        const unsigned keep = ((okm >> u) & 1u) ? 0xFFFFFFFFu : 0u;  // zero padding stays zero (BN would turn it into the shift)
        x.x &= keep;
        x.y &= keep;
        x.z &= keep;
        x.w &= keep;
        st16(dst + loff[u], x);
      }
    };
    U128 va[C::PER], vb[C::PER];
    WS_STAMP(lt == 0, 24);
    coefs(0);
    issue(0, va);
    issue(1, vb);
    commit(0, va);
    WS_STAMP(lt == 0, 25);
    __syncthreads();
#pragma unroll 1
    for (int ph = 0; ph < NPH; ph += 2) {  // NPH is even: two phases per trip, so that the register sets are compile-time names
      if (KSPL > 1) coefs(ph + 1);
      if (ph + 2 < NPH) issue(ph + 2, va);
      commit(ph + 1, vb);
      WS_STAMP(lt == 0, 26 + ph);
      __syncthreads();  // the taps of phase ph are done / phase ph + 1 is resident
      if (KSPL > 1 && ph + 2 < NPH) coefs(ph + 2);
      if (ph + 3 < NPH) issue(ph + 3, vb);
      if (ph + 2 < NPH) commit(ph + 2, va);
      WS_STAMP(lt == 0, 27 + ph);
      __syncthreads();
    }
  } else {
    // ---------------------------------------------------------------- computing waves
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = wave * C::TN * 16;
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
      int m = i * 16 + fr;
      m = m < C::M ? m : 0;
      const int img = NIMG > 1 ? m / (ROWS * WL) : 0;
      m -= img * (ROWS * WL);
      const int h = m / WL, w = m - h * WL;
      abase[i] = img * C::ISTR + h * C::RSTR + w * C::PSTR + fq * 16;
    }
    f32x4 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
      for (int j = 0; j < C::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    WS_STAMP(tid == 0, 0);
    __syncthreads();  // phase 0 is resident
    WS_STAMP(tid == 0, 1);
    auto plane_phases = [&](auto tag) {
      constexpr int P = decltype(tag)::value;
#pragma unroll 1
      for (int kc = 0; kc < KSPL; ++kc) {
        const int ph = P * KSPL + kc;
        const bf16_t* wk[C::TN];
#pragma unroll
        for (int j = 0; j < C::TN; ++j) wk[j] = wrow[j] + ((kc * CK) << wsh);
        mma_taps<C, CIN, P>(smem + (ph & 1) * C::BUF, abase, acc, wk, wsh);
        WS_STAMP(tid == 0, 2 + ph);
        __syncthreads();
      }
    };
    plane_phases(std::integral_constant<int, 0>{});
    plane_phases(std::integral_constant<int, 1>{});
    plane_phases(std::integral_constant<int, 2>{});
    plane_phases(std::integral_constant<int, 3>{});
    // ---------------------------------------------------------------- epilogue cells (output tile = buffer 0: the last
    // phase read buffer 1, nobody writes buffer 0 any more).  Weights were the MFMA A operand: a lane holds four
    // consecutive channels (fq*4 + r) of one pixel (fr) per tile.
    const bool stats = epi == FR_EPI_STATS;
    float s0[C::TN][4], s1[C::TN][4];
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s0[j][r] = s1[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
      const int m = i * 16 + fr;
      if (m >= C::M) continue;
#pragma unroll
      for (int j = 0; j < C::TN; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r];
          s0[j][r] += v[r];
          s1[j][r] = fmaf(v[r], v[r], s1[j][r]);
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(smem + m * C::OSTR + (n0 + j * 16 + fq * 4) * 2) = o;
      }
    }
    if (stats) {  // a wave owns its 32 columns: fold the 16 pixel lanes, one partial row per workgroup
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
            const int n = ncol0 + n0 + j * 16 + fq * 4 + r;
            st_part(p.part + ((size_t)sblk * 2 + 0) * (CW * NSPL) + n, a);
            st_part(p.part + ((size_t)sblk * 2 + 1) * (CW * NSPL) + n, c);
          }
        }
    }
  }
  WS_STAMP(tid == 0, 2 + NPH);
  __syncthreads();  // the output tile is complete
  WS_STAMP(tid == 0, 3 + NPH);
  for (int idx = tid; idx < C::M * OCH; idx += 512) {
    int r = idx / OCH;
    const int c8 = idx - r * OCH;
    const U128 v = ld16(smem + r * C::OSTR + c8 * 16);
    const int img = NIMG > 1 ? r / (ROWS * WL) : 0;
    r -= img * (ROWS * WL);
    const int h = r / WL, w = r - h * WL;
    const size_t pix = (size_t)((b + img) * C::HL + row0 + h) * WL + w;
    st16(out + pix * (size_t)p.ldc + ncol0 + c8 * 8, v);
  }
  WS_STAMP(tid == 0, 4 + NPH);
}

int* ws_switch() {
  static int* v = fr_option_slot("FRHIP_S2_WS", 1);
  return v;
}

template <int CIN, int WL, int ROWS, int NIMG, int PRO>
int launch(const FrConvArgs& a, hipStream_t st) {
  using C = WS<WL, ROWS, NIMG>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_s2_ws_kernel<CIN, WL, ROWS, NIMG, PRO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    fr_attr_done(attr_done);
  }
  static const int* order = fr_option_slot("FRHIP_XCD_ORDER", 1);
  FR_LAUNCH_KERNEL((conv3x3_s2_ws_kernel<CIN, WL, ROWS, NIMG, PRO>), dim3(a.B * C::NS / NIMG * (CIN / CW)), dim3(512), C::LDS,
                   st, a, *order != 0);
  FR_LAUNCH_CHECK();
}

template <int CIN, int WL, int ROWS, int NIMG>
int by_pro(const FrConvArgs& a, hipStream_t st) {
  switch (a.pro) {
    case FR_PRO_NONE: return launch<CIN, WL, ROWS, NIMG, FR_PRO_NONE>(a, st);
    case FR_PRO_BN: return launch<CIN, WL, ROWS, NIMG, FR_PRO_BN>(a, st);
    case FR_PRO_PRELU: return launch<CIN, WL, ROWS, NIMG, FR_PRO_PRELU>(a, st);
  }
  FR_UNSUPPORTED("fr_conv3x3_s2_strip (warp-specialised forward): unknown prologue");
}

}  // namespace

// images per workgroup of the warp-specialised forward kernel for a (batch, channels, low-res width), 0 = not served
int fr_s2ws_nimg(int B, int C, int WL) {
  if (!*ws_switch()) return 0;
  if (C == 128 && WL == 28) return 1;
  if (C == 256 && WL == 14) return 1;
  if (C == 512 && WL == 7 && B % 4 == 0) return 4;
  return 0;
}

// strips (= partial-sum rows) of a served forward launch
int fr_s2ws_strips(int B, int C, int WL) {
  const int n = fr_s2ws_nimg(B, C, WL);
  if (!n) return 0;
  return C == 128 ? B * 4 : B / n;
}

bool fr_s2ws_serves(const FrConvArgs& a) {
  if (a.mode != 0 || a.SC != a.N || a.RW != a.RH || a.SW != 2 * a.RW || a.SH != 2 * a.RH) return false;
  if (a.epi != FR_EPI_STORE && a.epi != FR_EPI_STATS) return false;
  if (a.epi == FR_EPI_STATS && !a.part) return false;
  if (a.pro != FR_PRO_NONE && a.pro != FR_PRO_BN && a.pro != FR_PRO_PRELU) return false;
  return fr_s2ws_nimg(a.B, a.SC, a.RW) != 0;
}

int fr_s2ws_launch(const FrConvArgs& a, hipStream_t st) {
  if (a.SC == 128) return by_pro<128, 28, 7, 1>(a, st);
  if (a.SC == 256) return by_pro<256, 14, 14, 1>(a, st);
  return by_pro<512, 7, 7, 4>(a, st);
}
