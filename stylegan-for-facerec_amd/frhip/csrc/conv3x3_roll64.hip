// frhip -- 64 -> 64 channel stride-1 3x3 convolution as a ROLLING window over image rows (bf16, gfx950).
//
// The 64-channel layers (112x112 and 56x56: backbone/model_irse.py:57-59 in units 0-2) do 576 MACs per input element:
// at B = 256 a 3.2-M-pixel tensor has to stream through the CUs in about the time the MFMAs need (HBM floor 130 us,
// MFMA floor 128 us for 64 -> 64 @112).  The LDS-strip kernel (conv3x3_strip.hip) spends 5 us loading a 58-KB strip
// (address arithmetic, prologue, HBM latency) synchronously in front of 5.6 us of MFMA loop and 3 us of epilogue per
// 224 pixels, half of the loaded rows being halo (tools/stamps.py) -- 3.4x its HBM floor.  Here ONE 8-wave workgroup per
// CU is PERSISTENT over a column band of an image and walks it top to bottom, with the waves split by role:
//
//   * waves 0-3 (one per SIMD) only compute: MFMA loop over the 6 resident rows of the iteration (RI = 4 output rows =
//     224 pixels = 14 M tiles, 7 x 2 per wave), weight fragments streamed L2 -> registers through a cyclic 3-tap ring,
//     then the epilogue cells (BN statistics / PReLU backward / BN-backward sums, carried in registers over the whole walk)
//     into an LDS output tile.  Their only vector-memory operations are the weight loads -- which matters: such
//     operations return in order, so one HBM load or store in flight in the same wave stalls every later weight wait
//     (a first version that prefetched the next rows from the computing waves ran its MFMA loop at half speed).
//   * waves 4-7 move the data: they drain the previous iteration's output tile to global memory, commit the NEXT
//     iteration's 4 new rows (requested one iteration earlier; BN-apply / PReLU prologue applied on the way) into a
//     10-row LDS ring, stage the aux tile of the fused backward epilogues, and request the rows after that.  No halo row
//     is read twice along the walk.
//   * one workgroup barrier per iteration; output tiles are double-buffered; statistics leave as ONE partial row per
//     workgroup (256-512 per launch instead of 14 336).
//
// Same contracts as fr_conv3x3_strip (prologues, epilogues, flip = mirrored taps for the data gradient).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "frhip_internal.h"

#ifdef FRHIP_STAMPS
// diagnostic build only (make stamps): per-workgroup phase times, accumulated over the walk (tools/stamps.py --roll)
__device__ unsigned long long* fr_stamp_buf = nullptr;
extern "C" int fr_debug_set_stamp_buffer_roll(unsigned long long* dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(fr_stamp_buf), &dev_ptr, sizeof(dev_ptr));
}
#define RSTAMP(k)                                             \
  do {                                                        \
    const unsigned long long t__ = __builtin_amdgcn_s_memrealtime(); \
    if (k > 0) tacc[k - 1] += t__ - tlast;                    \
    tlast = t__;                                              \
  } while (0)
#else
#define RSTAMP(k)
#endif

namespace {

struct R64 {
  static constexpr int C = 64;                  // CIN == COUT
  static constexpr int CH = 8;                  // 16-B chunks per pixel
  static constexpr int BW = 56;                 // band width
  static constexpr int GW = BW + 2;             // + halo columns
  static constexpr int RI = 4;                  // output rows per iteration
  static constexpr int NR = 10;                 // ring rows: 6 live + the 4 of the next iteration
  static constexpr int PSTR = C * 2 + 16;       // 144: odd number of 16-B slots per pixel (conv3x3_strip.hip)
  static constexpr int RSTR = GW * PSTR + 224;  // row stride: the slot index keeps counting across a row wrap
  static constexpr int RING = NR * RSTR;
  static constexpr int M = RI * BW;             // 224 pixels per iteration
  static constexpr int WM = 2, WN = 2, TM = 7, TN = 2;
  static constexpr int OSTR = C * 2 + 16;
  static constexpr int OUT_BYTES = M * OSTR;
  static constexpr int TILE_OFF = RING;                   // output tile (also carries the aux cells in)
  static constexpr int W1_OFF = TILE_OFF + OUT_BYTES;     // weights of input channels 32..63 as MFMA fragments:
  static constexpr int W1_BYTES = 9 * 4 * 64 * 16;        //   [tap][16-channel tile][lane] x 16 B = 36 KB
  static constexpr int RED_OFF = W1_OFF + W1_BYTES;
  static constexpr int RED_BYTES = WM * 2 * C * 4;
  static constexpr int PRO_OFF = RED_OFF + RED_BYTES;     // [4][64] prologue / epilogue coefficients
  static constexpr int LDS = PRO_OFF + 4 * C * 4;
  static constexpr int NTH = 512;                         // 4 computing + 4 data-moving waves
  static constexpr int ROWCHUNKS = GW * CH;     // 464 chunks per ring row
  static_assert(M == WM * TM * 16, "tiles must cover the iteration exactly");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int W, int PRO, bool AUX>
__global__ __launch_bounds__(512, 2) void conv3x3_roll64_kernel(const FrConvArgs p, const int nseg, const int nitems) {
  using K = R64;
  constexpr int H = W, NB = W / K::BW, NIT_IMG = H / K::RI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nit = NIT_IMG / nseg;
  const int epi = p.epi;

  // per-channel coefficients live in LDS: [pro_a | pro_b | epi_a | epi_b] x 64
  float* const coef = reinterpret_cast<float*>(smem + K::PRO_OFF);
  if (tid < K::C) {
    coef[tid] = PRO != FR_PRO_NONE ? p.pro_a[tid] : 0.f;
    coef[K::C + tid] = PRO == FR_PRO_BN ? p.pro_b[tid] : 0.f;
    coef[2 * K::C + tid] = AUX ? p.epi_a[tid] : 0.f;
    coef[3 * K::C + tid] = (AUX && (epi == FR_EPI_BNBWD || epi == FR_EPI_BIAS_RES)) ? p.epi_b[tid] : 0.f;
  }
  __syncthreads();

  if (wave >= 4) {
    // ================================================================================ data-moving waves
    const int ptid = tid - 256;
    const int ch = ptid & 7;   // a thread always handles the same 8 channels (16 bytes)
    const int t5 = ptid >> 3;  // 0..31: pixel slot within a group of 32
    // Everything that does not change along the walk is computed once (these waves share their SIMDs' issue slots with
    // the MFMA streams: per-iteration address arithmetic was what made them the slower side).
    // row loader: 8 chunks per thread = 4 ring rows of 58 pixel slots
    int rrow[8], rlds[8];
    unsigned rgoff[8], colok = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pc = u * 32 + t5;
      rrow[u] = pc / K::GW;
      const int gw = pc - rrow[u] * K::GW;
      rlds[u] = gw * K::PSTR + ch * 16;
      colok |= (rrow[u] < 4 ? 1u : 0u) << u;
      rgoff[u] = 0;  // + band column, below
    }
    // the iteration's 224 output pixels: 7 chunks per thread, pixel m = u*32 + t5 -> (row lr, column cc) of the band
    int ptile[7], prow[7], pcol[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int m = u * 32 + t5;
      prow[u] = m / K::BW;
      pcol[u] = m - prow[u] * K::BW;
      ptile[u] = K::TILE_OFF + m * K::OSTR + ch * 16;
    }
    float pa[8], pb[8];
    if (PRO != FR_PRO_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pa[j] = coef[ch * 8 + j];
        pb[j] = coef[K::C + ch * 8 + j];
      }
    }
    const int ldrow = W * p.lda * 2, auxrow = AUX ? W * p.ldaux * 2 : 0, outrow = W * p.ldc * 2;  // bytes per image row

#pragma unroll 1
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
      const int seg = item % nseg;
      const int ib = item / nseg;
      const int band = ib % NB, b = ib / NB;
      const int row_first = seg * nit * K::RI;
      const int col0 = band * K::BW;
      // wave-uniform 64-bit bases + 32-bit per-lane byte offsets (an image is < 4 GB)
      const char* const src_img = reinterpret_cast<const char*>(p.src) + (size_t)b * H * ldrow;
      const char* const aux_img = AUX ? reinterpret_cast<const char*>(p.aux) + (size_t)b * H * auxrow : nullptr;
      char* const out_img = reinterpret_cast<char*>(p.out) + (size_t)b * H * outrow;
      unsigned cmask = 0;  // chunks whose column lies inside the image
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pc = u * 32 + t5;
        const int gw = pc - rrow[u] * K::GW;
        const int gc = col0 + gw - 1;
        const int gcc = gc < 0 ? 0 : (gc >= W ? W - 1 : gc);
        rgoff[u] = (unsigned)(gcc * p.lda + ch * 8) * 2u;
        cmask |= ((unsigned)gc < (unsigned)W ? 1u : 0u) << u;
      }
      U128 st[8];
      unsigned okmask = 0;
      auto issue_rows = [&](int first_row, int nrows) {  // image rows first_row .. first_row + nrows - 1 -> st[]
        okmask = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ra = first_row + rrow[u];
          const bool ok = rrow[u] < nrows && (unsigned)ra < (unsigned)H && ((cmask >> u) & 1u);
          // always load (from a clamped, valid address): a static number of operations in flight; padding is zeroed
          // when the row is committed
          const int rac = ra < 0 ? 0 : (ra >= H ? H - 1 : ra);
          st[u] = ld16(src_img + ((unsigned)(rac * ldrow) + rgoff[u]));
          okmask |= ok ? (1u << u) : 0u;
        }
      };
      auto commit_rows = [&](int first_row, int nrows) {  // st[] -> prologue -> ring slots of those rows
        const int s0 = (first_row + 1 + 10 * K::NR) % K::NR;  // slot of the first row (first_row >= -1)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (rrow[u] < nrows) {
            const bool ok = (okmask >> u) & 1u;
            U128 x = ok ? st[u] : zero16();
            if (PRO != FR_PRO_NONE && ok) {  // pro2 (frhip_internal.h): 5 / 7 vector instructions per dword instead of 11-12
              x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
              x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
              x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
              x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
            }
            int slot = s0 + rrow[u];
            slot = slot >= K::NR ? slot - K::NR : slot;
            st16(smem + slot * K::RSTR + rlds[u], x);
          }
        }
      };
      U128 ax[AUX ? 7 : 1];
      auto issue_aux = [&](int r0) {  // aux rows r0 .. r0+3 (clamped past the image: never used)
        if (AUX) {
          const int rc = r0 + K::RI <= H ? r0 : H - K::RI;
          const char* rowb = aux_img + (size_t)rc * auxrow;
#pragma unroll
          for (int u = 0; u < 7; ++u)
            ax[AUX ? u : 0] = ld16(rowb + ((unsigned)(prow[u] * auxrow) + (unsigned)((col0 + pcol[u]) * p.ldaux + ch * 8) * 2u));
        }
      };
      auto stage_aux = [&]() {  // ax[] -> aux cells of the output tile
        if (AUX) {
#pragma unroll
          for (int u = 0; u < 7; ++u) st16(smem + ptile[u], ax[AUX ? u : 0]);
        }
      };
      auto drain_tile = [&](int r0) {  // output tile -> global rows r0 .. r0+3
        U128 o[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) o[u] = ld16(smem + ptile[u]);
        char* rowb = out_img + (size_t)r0 * outrow;
#pragma unroll
        for (int u = 0; u < 7; ++u)
          st16(rowb + ((unsigned)(prow[u] * outrow) + (unsigned)((col0 + pcol[u]) * p.ldc + ch * 8) * 2u), o[u]);
      };

      // prime: rows row_first-1 .. row_first+4 resident, rows of iteration 1 and the aux tile of iteration 0 requested
      issue_rows(row_first - 1, 4);
      commit_rows(row_first - 1, 4);
      issue_rows(row_first + 3, 2);
      commit_rows(row_first + 3, 2);
      issue_rows(row_first + K::RI + 1, K::RI);
      issue_aux(row_first);
      __syncthreads();
#pragma unroll 1
      for (int k = 0; k < nit; ++k) {
        const int r0 = row_first + k * K::RI;
        // first half of the iteration (the computing waves run channel chunk 0): empty the tile of iteration k-1 and
        // put the aux cells of iteration k in its place
        if (k > 0) drain_tile(r0 - K::RI);
        stage_aux();
        __syncthreads();
        // second half: the ring rows of iteration k+1 (requested one iteration ago), then the requests for k+2
        if (k + 1 < nit) commit_rows(r0 + K::RI + 1, K::RI);
        issue_rows(r0 + 2 * K::RI + 1, K::RI);  // past the walk: clamped, never committed
        issue_aux(r0 + K::RI);
        __syncthreads();  // iteration k is computed: the tile is complete, the ring rows of k+1 are visible
      }
      drain_tile(row_first + (nit - 1) * K::RI);
      __syncthreads();  // end of item: the tile is empty, the ring may be primed again (+ the statistics hand-over)
    }
    return;
  }

  // ================================================================================== computing waves
  const int wn = wave & 1, wm = wave >> 1;
  const bf16_t* __restrict__ wgt = reinterpret_cast<const bf16_t*>(p.w);
  const int flip = p.mode;
  // The weights never change along the walk and must not be streamed: 4 waves x 36.9 KB per iteration is 2.5x the
  // activation bytes and held the MFMA loop at 5.5 us per iteration (3.0 us with the loads removed), however deep the
  // register ring.  Input channels 0..31 stay in registers (72 VGPRs per computing wave), channels 32..63 in LDS in
  // fragment order: one conflict-free ds_read_b128 per fragment, each reused for TM = 7 tiles.  Staged by these waves
  // while the data-moving waves prime the ring.
  for (int idx = tid; idx < 9 * 4 * 64; idx += 256) {
    const int tap = idx >> 8, nt = (idx >> 6) & 3, ln = idx & 63;
    const int wt = flip ? 8 - tap : tap;
    st16(smem + K::W1_OFF + idx * 16, ld16(wgt + (size_t)(nt * 16 + (ln & 15)) * 9 * K::C + wt * K::C + 32 + (ln >> 4) * 8));
  }
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = wn * K::TN * 16;
  // input channels 0..31: 9 taps x TN fragments stationary in registers for the whole launch
  s16x8 bq[9][K::TN];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int wt = flip ? 8 - t : t;
#pragma unroll
    for (int j = 0; j < K::TN; ++j)
      bq[t][j] = *reinterpret_cast<const s16x8*>(wgt + (size_t)(n0 + j * 16 + fr) * 9 * K::C + fq * 8 + wt * K::C);
  }
  // input channels 32..63: fragments from LDS, two taps ahead
  s16x8 b1[2][K::TN];
  const char* const w1_lane = smem + K::W1_OFF + (wn * K::TN * 64 + lane) * 16;
  auto load_b1 = [&](int slot, int tap) {
#pragma unroll
    for (int j = 0; j < K::TN; ++j) b1[slot][j] = *reinterpret_cast<const s16x8*>(w1_lane + (tap * 4 + j) * 64 * 16);
  };
  // per-lane pixel geometry of the TM tiles of this wave: local output row, column
  int plr[K::TM], pcc[K::TM];
#pragma unroll
  for (int i = 0; i < K::TM; ++i) {
    const int m = (wm * K::TM + i) * 16 + fr;
    plr[i] = m / K::BW;
    pcc[i] = (m - plr[i] * K::BW) * K::PSTR + fq * 16;
  }

#pragma unroll 1
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int seg = item % nseg;
    const int row_first = seg * nit * K::RI;
    // column sums carried over the whole walk (statistics / slope / BN-backward epilogues)
    float s0[K::TN][4], s1[K::TN][4];
#pragma unroll
    for (int j = 0; j < K::TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s0[j][r] = s1[j][r] = 0.f;
    __syncthreads();  // the ring is primed

#ifdef FRHIP_STAMPS
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
    const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
      RSTAMP(0);
      const int r0 = row_first + it * K::RI;
      // ring addresses of this iteration: input row (r0 + lr + dy - 1) sits in slot (r0 + lr + dy) % 10
      const int sb = r0 % K::NR;
      int abase[K::TM][3];
#pragma unroll
      for (int i = 0; i < K::TM; ++i)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          int slot = sb + plr[i] + dy;
          slot = slot >= K::NR ? slot - K::NR : slot;
          abase[i][dy] = slot * K::RSTR + pcc[i];
        }

      f32x4 acc[K::TM][K::TN];
#pragma unroll
      for (int i = 0; i < K::TM; ++i)
#pragma unroll
        for (int j = 0; j < K::TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

      // -------------------------------------------------------------- main loop: 2 channel chunks x 9 taps x TM tiles
      constexpr int NSTEP = 9 * K::TM;
      constexpr int D = 9;  // A-fragment ring depth (steps of LDS-read lead); 63 % D == 0
      s16x8 ring[D];
      auto a_addr = [&](int step) -> const s16x8* {  // step in [0, 2 * NSTEP)
        const int chunk = step >= NSTEP ? 1 : 0;
        const int stp = step - chunk * NSTEP;
        const int tap = stp / K::TM, i = stp - tap * K::TM;
        return reinterpret_cast<const s16x8*>(smem + abase[i][tap / 3] + (tap % 3) * K::PSTR + chunk * 64);
      };
#pragma unroll
      for (int d = 0; d < D; ++d) ring[d] = *a_addr(d);
      // one 32-channel chunk = 63 fully unrolled steps; the chunk index is a compile-time constant of each copy (a
      // runtime chunk loop would make the register arrays dynamically indexed)
      auto run_chunk = [&](auto ctag) {
        constexpr int chunk = decltype(ctag)::value;
#pragma unroll
        for (int stp = 0; stp < NSTEP; ++stp) {
          const int step = chunk * NSTEP + stp;
          const int tap = stp / K::TM, i = stp - tap * K::TM;
          const s16x8 a = ring[step % D];
#pragma unroll
          for (int j = 0; j < K::TN; ++j)
            acc[i][j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(chunk == 0 ? bq[tap][j] : b1[tap & 1][j], a, acc[i][j], 0, 0, 0);
          if (step + D < 2 * NSTEP) ring[step % D] = *a_addr(step + D);
          if (i == K::TM - 1) {  // chunk-1 fragments: two taps ahead
            if (chunk == 0 && tap >= 7) load_b1(tap - 7, tap - 7);
            if (chunk == 1 && tap + 2 < 9) load_b1(tap & 1, tap + 2);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, K::TN, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
          if (i == K::TM - 1) __builtin_amdgcn_sched_barrier(0);
        }
      };
      run_chunk(std::integral_constant<int, 0>{});
      // the data-moving waves have emptied the output tile and staged this iteration's aux cells by now (they only
      // have a tile to drain in this half): nothing of this wave has to be visible, so no wait in front of the barrier
      __builtin_amdgcn_s_barrier();
      run_chunk(std::integral_constant<int, 1>{});
      RSTAMP(1);

      // -------------------------------------------------------------- epilogue cells -> output tile
      char* const otile = smem + K::TILE_OFF;
      auto cells = [&](auto tag) {
        constexpr int E = decltype(tag)::value;
#pragma unroll
        for (int j = 0; j < K::TN; ++j) {
          float ea[4], eb[4];  // from LDS per use rather than live across the MFMA loop
          if (E == FR_EPI_PRELU_BWD || E == FR_EPI_BNBWD || E == FR_EPI_BIAS_RES) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(coef + 2 * K::C + n0 + j * 16 + fq * 4);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(coef + 3 * K::C + n0 + j * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ea[r] = t[r];
              eb[r] = t2[r];
            }
          }
#pragma unroll
          for (int i = 0; i < K::TM; ++i) {
            const int m = (wm * K::TM + i) * 16 + fr;
            uint2* cell = reinterpret_cast<uint2*>(otile + m * K::OSTR + (n0 + j * 16 + fq * 4) * 2);
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
                v[r] = pos ? v[r] : v[r] * ea[r];
              } else if (E == FR_EPI_BNBWD) {
                s0[j][r] += v[r];
                s1[j][r] = fmaf(v[r], (x[r] - ea[r]) * eb[r], s1[j][r]);
              } else if (E == FR_EPI_BIAS_RES) {
                v[r] += ea[r] + eb[r] + x[r];
              }
            }
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *cell = o;
          }
        }
      };
      if (AUX) {
        if (epi == FR_EPI_PRELU_BWD) cells(std::integral_constant<int, FR_EPI_PRELU_BWD>{});
        else if (epi == FR_EPI_BNBWD) cells(std::integral_constant<int, FR_EPI_BNBWD>{});
        else cells(std::integral_constant<int, FR_EPI_BIAS_RES>{});
      } else {
        if (epi == FR_EPI_STATS) cells(std::integral_constant<int, FR_EPI_STATS>{});
        else cells(std::integral_constant<int, FR_EPI_STORE>{});
      }
      RSTAMP(2);
      __syncthreads();  // the tile is complete; the data-moving waves have committed the rows of it + 1
      RSTAMP(3);
    }
#ifdef FRHIP_STAMPS
    if (tid == 0 && fr_stamp_buf) {
      for (int k = 0; k < 6; ++k) fr_stamp_buf[(size_t)item * 8 + k] = tacc[k];
      fr_stamp_buf[(size_t)item * 8 + 6] = __builtin_amdgcn_s_memrealtime() - tstart;
      fr_stamp_buf[(size_t)item * 8 + 7] = nit;
    }
#endif

    // ---------------------------------------------------------------- one partial row per work item
    float* red = reinterpret_cast<float*>(smem + K::RED_OFF);
    const bool sums = epi != FR_EPI_STORE && epi != FR_EPI_BIAS_RES;
    if (sums) {
#pragma unroll
      for (int j = 0; j < K::TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = s0[j][r], c = s1[j][r];
#pragma unroll
          for (int o2 = 1; o2 < 16; o2 <<= 1) {
            a += __shfl_xor(a, o2, 64);
            c += __shfl_xor(c, o2, 64);
          }
          if (fr == 0) {
            red[(wm * 2 + 0) * K::C + n0 + j * 16 + fq * 4 + r] = a;
            red[(wm * 2 + 1) * K::C + n0 + j * 16 + fq * 4 + r] = c;
          }
        }
    }
    __syncthreads();  // end of item (matches the data-moving waves)
    if (sums && tid < 2 * K::C) {
      const int k = tid / K::C, n = tid - k * K::C;
      st_part(p.part + ((size_t)item * 2 + k) * K::C + n, red[(0 * 2 + k) * K::C + n] + red[(1 * 2 + k) * K::C + n]);
    }
  }
}

int roll_nseg(int B, int W) {
  // row segments per (image, band): enough workgroups for one per CU on 256 CUs; H / 4 iterations must divide evenly
  const int bands = W / R64::BW;
  static const int cand[4] = {1, 2, 7, 14};
  static const int* forced = fr_option_slot("FRHIP_ROLL_NSEG", -1);  // test hook: walks of whole images at small batches
  for (int k = 0; k < 4; ++k) {
    if (*forced == cand[k]) return cand[k];
  }
  for (int k = 0; k < 4; ++k) {
    if ((long long)B * bands * cand[k] >= 256) return cand[k];
  }
  return 14;
}

template <int W, int PRO, bool AUX>
int launch(const FrConvArgs& a, hipStream_t st) {
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_roll64_kernel<W, PRO, AUX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, R64::LDS);
    fr_attr_done(attr_done);
  }
  const int nseg = roll_nseg(a.B, W);
  const int items = a.B * (W / R64::BW) * nseg;
  // one workgroup per CU (156 KB of LDS), persistent over its items: weights and coefficients are staged once
  const int grid = items < 256 ? items : 256;
  FR_LAUNCH_KERNEL((conv3x3_roll64_kernel<W, PRO, AUX>), dim3(grid), dim3(R64::NTH), R64::LDS, st, a, nseg, items);
  FR_LAUNCH_CHECK();
}

template <int W>
int by_pro(const FrConvArgs& a, hipStream_t st) {
  if (a.epi == FR_EPI_STATS_X) FR_UNSUPPORTED("rolling-window convolution: FR_EPI_STATS_X is served by the LDS-strip instances only");
  if (a.epi == FR_EPI_PRELU_BWD || a.epi == FR_EPI_BNBWD) {
    // the fused backward epilogues come with the data gradients, which have no prologue
    if (a.pro != FR_PRO_NONE || !a.aux) FR_UNSUPPORTED("rolling-window convolution: data-gradient epilogues take no prologue and need aux");
    return launch<W, FR_PRO_NONE, true>(a, st);
  }
  if (a.epi == FR_EPI_BIAS_RES) {  // inference conv2: PReLU prologue, folded shifts + shortcut epilogue
    if (a.pro != FR_PRO_PRELU || !a.aux || !a.epi_a || !a.epi_b) FR_UNSUPPORTED("rolling-window convolution: the folded epilogue needs the PReLU prologue, aux, epi_a and epi_b");
    return launch<W, FR_PRO_PRELU, true>(a, st);
  }
  switch (a.pro) {
    case FR_PRO_NONE: return launch<W, FR_PRO_NONE, false>(a, st);
    case FR_PRO_BN: return launch<W, FR_PRO_BN, false>(a, st);
    case FR_PRO_PRELU: return launch<W, FR_PRO_PRELU, false>(a, st);
  }
  FR_UNSUPPORTED("rolling-window convolution: prologue / epilogue combination not served");
}

}  // namespace

// FRHIP_ROLL64=0: the 64 -> 64 layers stay on the LDS-strip kernel (A/B switch)
bool fr_roll64_enabled() {
  static const int* v = fr_option_slot("FRHIP_ROLL64", 1);
  return *v != 0;
}

int fr_roll64_parts(int B, int W) { return B * (W / R64::BW) * roll_nseg(B, W); }

int fr_roll64_launch(const FrConvArgs& a, hipStream_t st) {
  if (a.SW == 112) return by_pro<112>(a, st);
  if (a.SW == 56) return by_pro<56>(a, st);
  FR_UNSUPPORTED("rolling-window convolution: prologue / epilogue combination not served");
}
