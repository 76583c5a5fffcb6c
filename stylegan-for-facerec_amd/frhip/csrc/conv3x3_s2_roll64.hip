// frhip -- the 64-channel stride-2 3x3 convolution (unit 0 of the first stage: 112x112 -> 56x56,
// backbone/model_irse.py:57-59 with stride 2) and its data gradient as a ROLLING window over image rows (bf16, gfx950).
//
// Both directions move 9 MACs per low-resolution pixel and channel pair but stream a 112x112x64 tensor (forward: read it;
// gradient: read the PReLU input `aux` and write dx): at B = 256 their HBM floors are 82 us / 147 us against 24 us of MFMA
// work.  The LDS-strip kernel (conv3x3_s2_strip.hip) ran them at 2.4x the HBM floor: 7168 short workgroups, each loading
// four parity planes (or producing four output classes) one after the other with a barrier on either side of every phase
// (rocprofv3: 57-71 % of the wave cycles waiting).  Here, as in conv3x3_roll64.hip, ONE 8-wave workgroup per CU is
// PERSISTENT over an image (or a row segment of it) and walks it top to bottom, one low-resolution row per iteration:
//
//   * waves 0-3 only compute.  ALL weights of the wave's 32 output channels are stationary in registers (9 taps x 64 input
//     channels x 2 N tiles = 36 fragments = 144 VGPRs), the A fragments come from the LDS ring; a wave owns 2 M tiles (of
//     the 4 that cover the 56 low-res columns) x 2 N tiles.  Forward: 18 K steps into one accumulator set.  Gradient: the
//     four output classes (ph, pw) one after the other (1, 2, 2, 4 taps), each followed by its PReLU-backward cells.
//   * waves 4-7 move the data one iteration ahead: ring rows (forward: two high-res rows per iteration, stored as an even-
//     and an odd-column plane so that the stride-2 taps read consecutive LDS pixels; gradient: one g row), the aux cells of
//     the gradient's epilogue into the output tile, and the finished output tile to global memory.
//   * one workgroup barrier per iteration; output tiles double-buffered; statistics leave as one partial row per work item.
//
// Same contracts as fr_conv3x3_s2_strip for the combinations served: forward with any prologue and the STORE / STATS
// epilogues; gradient (mode 2, all four classes) with the PReLU-backward epilogue.  Everything else stays on the strip kernel.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "frhip_internal.h"

namespace {

struct S2R {
  static constexpr int C = 64, CH = 8;
  static constexpr int WL = 56, WH = 112;
  static constexpr int PSTR = C * 2 + 16;  // 144: odd number of 16-B slots per pixel (conflict-free ds_read_b128)
  // forward ring row = one high-res row: [even columns 0, 2 .. 110 | odd columns -1, 1 .. 111]
  static constexpr int F_OOFF = WL * PSTR;
  static constexpr int F_RSTR = (WL + WL + 1) * PSTR;
  static constexpr int F_NR = 6;
  static constexpr int F_TILE = WL * PSTR;
  // gradient ring row = one g row + a zero column on the right
  static constexpr int D_RSTR = (WL + 1) * PSTR;
  static constexpr int D_NR = 4;
  static constexpr int D_TILE = 2 * WH * PSTR;
  static constexpr int NTH = 512;
  static constexpr int NSTEP = 36;  // K steps x M tiles of one iteration of a computing wave
  template <int KIND>
  struct L {
    static constexpr int RING = KIND == 0 ? F_NR * F_RSTR : D_NR * D_RSTR;
    static constexpr int TILE = KIND == 0 ? F_TILE : D_TILE;
    static constexpr int TILE_OFF = RING;
    static constexpr int RED_OFF = TILE_OFF + 2 * TILE;
    static constexpr int COEF_OFF = RED_OFF + 2 * 2 * C * 4;
    static constexpr int LDS = COEF_OFF + 3 * C * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
  };
};

// One (K step, M tile) of a computing wave's iteration, in execution order.
//   forward: 2 channel chunks x 9 taps x 2 tiles.  roff = kernel row (ring row 2r + kh - 1), coff = kernel column.
//   gradient: classes P = 2 ph + pw in turn, each 2 chunks x its taps x 2 tiles.  dx row 2i + ph collects g rows i + 1 (kh = 0)
//   and i (kh = 2) for ph = 1, g row i (kh = 1) for ph = 0 (columns alike): roff / coff = 0 / 1 ring row / column offsets.
struct Step {
  int cls, ktap, roff, coff, chunk, i, last;
};
constexpr int cls_off(int P) { return P == 0 ? 0 : (P == 1 ? 4 : (P == 2 ? 12 : (P == 3 ? 20 : 36))); }
template <int KIND>
constexpr Step step_of(int s) {
  Step r{};
  if (KIND == 0) {
    const int tap = (s % 18) / 2;
    r.cls = 0;
    r.chunk = s / 18;
    r.i = s & 1;
    r.ktap = tap;
    r.roff = tap / 3;
    r.coff = tap % 3;
    r.last = s == 35;
    return r;
  }
  const int P = s < 4 ? 0 : (s < 12 ? 1 : (s < 20 ? 2 : 3));
  const int l = s - cls_off(P);
  const int PH = P >> 1, PW = P & 1, NWD = PW ? 2 : 1, NT = (PH ? 2 : 1) * NWD;
  const int t = (l / 2) % NT;
  const int kh = PH ? 2 * (t / NWD) : 1, kw = PW ? 2 * (t % NWD) : 1;
  r.cls = P;
  r.chunk = l / (NT * 2);
  r.i = l & 1;
  r.ktap = kh * 3 + kw;
  r.roff = PH ? 1 - t / NWD : 0;
  r.coff = PW ? 1 - t % NWD : 0;
  r.last = l == NT * 4 - 1;
  return r;
}

// Pins the uses of a loaded chunk behind this point of the program.  The instruction scheduler otherwise hoists the unpacking
// of the OTHER register set's chunks (committed after the next barrier) in front of the barrier, and with it their
// s_waitcnt: the wait then covers loads that were meant to stay in flight for another iteration.
__device__ __forceinline__ void pin(U128& a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }

template <int S, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (S < N) {
    f(std::integral_constant<int, S>{});
    static_for<S + 1, N>(f);
  }
}

template <int KIND, int PRO>
__global__ __launch_bounds__(512, 2) void conv3x3_s2_roll64_kernel(const FrConvArgs p, const int nseg, const int nitems) {
  using K = S2R;
  using LY = S2R::L<KIND>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nit = K::WL / nseg;  // low-res rows per work item
  const int epi = p.epi;

  // per-channel coefficients: [pro_a | pro_b | epi_a] x 64
  float* const coef = reinterpret_cast<float*>(smem + LY::COEF_OFF);
  if (tid < K::C) {
    coef[tid] = PRO != FR_PRO_NONE ? p.pro_a[tid] : 0.f;
    coef[K::C + tid] = PRO == FR_PRO_BN ? p.pro_b[tid] : 0.f;
    coef[2 * K::C + tid] = KIND == 1 ? p.epi_a[tid] : 0.f;
  }
  // the cells no iteration ever writes: column -1 of the forward rows' odd plane, the column right of a g row
  if (KIND == 0) {
    for (int idx = tid; idx < K::F_NR * K::CH; idx += K::NTH)
      st16(smem + (idx >> 3) * K::F_RSTR + K::F_OOFF + (idx & 7) * 16, zero16());
  } else {
    for (int idx = tid; idx < K::D_NR * K::CH; idx += K::NTH)
      st16(smem + (idx >> 3) * K::D_RSTR + K::WL * K::PSTR + (idx & 7) * 16, zero16());
  }
  __syncthreads();

  if (wave >= 4) {
    // ================================================================================ data-moving waves
    const int ptid = tid - 256;
    const int ch = ptid & 7;   // a thread always handles the same 8 channels (16 bytes)
    const int t5 = ptid >> 3;  // 0..31: pixel slot within a group of 32
    float pa[8], pb[8];
    if (PRO != FR_PRO_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pa[j] = coef[ch * 8 + j];
        pb[j] = coef[K::C + ch * 8 + j];
      }
    }
    // a pair of high-res rows = 224 consecutive pixels = 7 chunks per thread: pixel pc = u*32 + t5
    int hlds[7];          // forward: LDS offset of the chunk inside its ring row (plane split), + the row of the pair
    unsigned hrow = 0;    // forward: bit u = second row of the pair
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int pc = u * 32 + t5;
      const int hr = pc >= K::WH ? 1 : 0, col = pc - hr * K::WH;
      hrow |= (unsigned)hr << u;
      hlds[u] = ((col & 1) ? K::F_OOFF + ((col >> 1) + 1) * K::PSTR : (col >> 1) * K::PSTR) + ch * 16;
    }
    const int hi_row_src = K::WH * p.lda * 2;                          // bytes per high-res row of src (forward)
    const int lo_row_src = K::WL * p.lda * 2;                          // bytes per g row (gradient)
    const int hi_row_aux = KIND == 1 ? K::WH * p.ldaux * 2 : 0;
    const int out_row = (KIND == 0 ? K::WL : K::WH) * p.ldc * 2;       // bytes per output row

#pragma unroll 1
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
      const int seg = item % nseg, b = item / nseg;
      const int r0 = seg * nit;  // first low-res row of the walk
      const char* const src_img =
          reinterpret_cast<const char*>(p.src) + (size_t)b * (KIND == 0 ? K::WH * (size_t)hi_row_src : K::WL * (size_t)lo_row_src);
      const char* const aux_img = KIND == 1 ? reinterpret_cast<const char*>(p.aux) + (size_t)b * K::WH * hi_row_aux : nullptr;
      char* const out_img = reinterpret_cast<char*>(p.out) + (size_t)b * (KIND == 0 ? K::WL : K::WH) * out_row;

      if constexpr (KIND == 0) {
        // two register sets: a pair of rows is requested TWO iterations before it is committed (one iteration of lead left
        // the walk latency-bound: 2.6 us per iteration for 36 KB)
        struct RowSet {
          U128 st[7];
          unsigned okmask;
        } setA, setB;
        auto issue_rows = [&](RowSet& rs, int first) {  // high-res rows first, first + 1 -> rs
          U128(&st)[7] = rs.st;
          unsigned okmask = 0;
#pragma unroll
          for (int u = 0; u < 7; ++u) {
            const int hr = (hrow >> u) & 1;
            const int ra = first + hr;
            const bool ok = (unsigned)ra < (unsigned)K::WH;
            const int rac = ra < 0 ? 0 : (ra >= K::WH ? K::WH - 1 : ra);  // always load: static number of operations in flight
            const int col = u * 32 + t5 - hr * K::WH;
            st[u] = ld16(src_img + ((unsigned)(rac * hi_row_src) + (unsigned)(col * p.lda + ch * 8) * 2u));
            okmask |= ok ? (1u << u) : 0u;
          }
          rs.okmask = okmask;
        };
        auto commit_rows = [&](RowSet& rs, int first) {  // rs -> prologue -> ring slots of rows first, first + 1 (first is even)
          U128(&st)[7] = rs.st;
          const unsigned okmask = rs.okmask;
          const int s0 = (first + 2 + 6 * K::F_NR) % K::F_NR;
#pragma unroll
          for (int u = 0; u < 7; ++u) pin(st[u]);
          // no branch anywhere in the steady state of these waves: the compiler's vmcnt bookkeeping is exact only on
          // straight-line code, and a conservative wait here would also wait for the loads of the OTHER register set
#pragma unroll
          for (int u = 0; u < 7; ++u) {
            const unsigned keep = ((okmask >> u) & 1u) ? 0xFFFFFFFFu : 0u;
            U128 x = st[u];
            if (PRO != FR_PRO_NONE) {  // pro2 (frhip_internal.h): 5 / 7 vector instructions per dword instead of 11-12
              x.x = pro2<PRO>(x.x, pa[0], pb[0], pa[1], pb[1]);
              x.y = pro2<PRO>(x.y, pa[2], pb[2], pa[3], pb[3]);
              x.z = pro2<PRO>(x.z, pa[4], pb[4], pa[5], pb[5]);
              x.w = pro2<PRO>(x.w, pa[6], pb[6], pa[7], pb[7]);
            }
            x.x &= keep;
            x.y &= keep;
            x.z &= keep;
            x.w &= keep;
            st16(smem + (s0 + ((hrow >> u) & 1)) * K::F_RSTR + hlds[u], x);
          }
        };
        auto drain_tile = [&](int k) {  // output tile of iteration k -> low-res row r0 + k
          const char* tile = smem + LY::TILE_OFF + (k & 1) * LY::TILE;
          char* rowb = out_img + (size_t)(r0 + k) * out_row;
          U128 o[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int m = u * 32 + t5, mc = m < K::WL ? m : K::WL - 1;
            o[u] = ld16(tile + mc * K::PSTR + ch * 16);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {  // lanes past the row store pixel 55 again (same bytes): no divergence
            const int m = u * 32 + t5, mc = m < K::WL ? m : K::WL - 1;
            st16(rowb + (unsigned)(mc * p.ldc + ch * 8) * 2u, o[u]);
          }
        };
        // prime: rows 2 r0 - 1 .. 2 r0 + 1 resident (as the pairs (2 r0 - 2, 2 r0 - 1), (2 r0, 2 r0 + 1)), the next pair requested
        issue_rows(setA, 2 * r0 - 2);
        issue_rows(setB, 2 * r0);
        commit_rows(setA, 2 * r0 - 2);
        issue_rows(setA, 2 * r0 + 2);
        commit_rows(setB, 2 * r0);
        issue_rows(setB, 2 * r0 + 4);
        __syncthreads();
        auto iteration = [&](RowSet& rs, int k) {  // rs: the rows of iteration k + 1, requested two iterations ago
          const int r = r0 + k;
          commit_rows(rs, 2 * r + 2);  // (the last iteration commits rows nobody reads into slots nobody reads)
          issue_rows(rs, 2 * r + 6);   // past the walk: clamped
          __syncthreads();             // iteration k is computed, the rows of k + 1 are visible
        };
        iteration(setA, 0);
        drain_tile(0);
        iteration(setB, 1);
#pragma unroll 1
        for (int k = 2; k < nit; k += 2) {  // nit is even
          drain_tile(k - 1);
          iteration(setA, k);
          drain_tile(k);
          iteration(setB, k + 1);
        }
        drain_tile(nit - 1);
      } else {
        struct GSet {
          U128 sg[2], ax[7];
        } setA, setB;
        auto issue_g = [&](GSet& gs, int row) {  // g row -> gs (clamped past the image; zeroed at commit)
          U128(&sg)[2] = gs.sg;
          const int rc = row < K::WL ? row : K::WL - 1;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int m = u * 32 + t5, mc = m < K::WL ? m : K::WL - 1;  // lanes past the row repeat pixel 55
            sg[u] = ld16(src_img + ((unsigned)(rc * lo_row_src) + (unsigned)(mc * p.lda + ch * 8) * 2u));
          }
        };
        auto commit_g = [&](GSet& gs, int row) {
          U128(&sg)[2] = gs.sg;
#pragma unroll
          for (int u = 0; u < 2; ++u) pin(sg[u]);
          const bool ok = row < K::WL;
          char* slot = smem + (row % K::D_NR) * K::D_RSTR;
          const unsigned keep = ok ? 0xFFFFFFFFu : 0u;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int m = u * 32 + t5, mc = m < K::WL ? m : K::WL - 1;
            U128 x = sg[u];
            x.x &= keep;
            x.y &= keep;
            x.z &= keep;
            x.w &= keep;
            st16(slot + mc * K::PSTR + ch * 16, x);
          }
        };
        auto issue_aux = [&](GSet& gs, int r) {  // aux rows 2r, 2r + 1 = 224 consecutive pixels (clamped past the image: never used)
          U128(&ax)[7] = gs.ax;
          const int rc = r < K::WL ? r : K::WL - 1;
          const char* rowb = aux_img + (size_t)(2 * rc) * hi_row_aux;
#pragma unroll
          for (int u = 0; u < 7; ++u) ax[u] = ld16(rowb + (unsigned)((u * 32 + t5) * p.ldaux + ch * 8) * 2u);
        };
        auto stage_aux = [&](GSet& gs, int k) {  // gs -> the aux cells of the output tile of iteration k
          U128(&ax)[7] = gs.ax;
#pragma unroll
          for (int u = 0; u < 7; ++u) pin(ax[u]);
          char* tile = smem + LY::TILE_OFF + (k & 1) * LY::TILE;
#pragma unroll
          for (int u = 0; u < 7; ++u) st16(tile + (u * 32 + t5) * K::PSTR + ch * 16, ax[u]);
        };
        auto drain_tile = [&](int k) {  // output tile of iteration k -> dx rows 2 (r0 + k), + 1
          const char* tile = smem + LY::TILE_OFF + (k & 1) * LY::TILE;
          char* rowb = out_img + (size_t)(2 * (r0 + k)) * out_row;
          U128 o[7];
#pragma unroll
          for (int u = 0; u < 7; ++u) o[u] = ld16(tile + (u * 32 + t5) * K::PSTR + ch * 16);
#pragma unroll
          for (int u = 0; u < 7; ++u) st16(rowb + (unsigned)((u * 32 + t5) * p.ldc + ch * 8) * 2u, o[u]);
        };
        // prime: g rows r0, r0 + 1 resident, row r0 + 2 requested; aux cells of iteration 0 staged, those of iteration 1 requested
        issue_g(setA, r0);
        issue_g(setB, r0 + 1);
        issue_aux(setA, r0);
        commit_g(setA, r0);
        commit_g(setB, r0 + 1);
        stage_aux(setA, 0);
        issue_g(setA, r0 + 2);  // the sets of iterations 1 and 2: requested two iterations before they are staged
        issue_aux(setA, r0 + 1);
        issue_g(setB, r0 + 3);
        issue_aux(setB, r0 + 2);
        __syncthreads();
        auto iteration = [&](GSet& gs, int k) {  // gs: g row r + 2 and the aux cells of iteration k + 1
          const int r = r0 + k;
          stage_aux(gs, k + 1);  // into the tile just drained (after the last iteration: cells nobody reads)
          commit_g(gs, r + 2);
          issue_g(gs, r + 4);
          issue_aux(gs, r + 3);
          __syncthreads();
        };
        iteration(setA, 0);
        drain_tile(0);
        iteration(setB, 1);
#pragma unroll 1
        for (int k = 2; k < nit; k += 2) {  // nit is even
          drain_tile(k - 1);
          iteration(setA, k);
          drain_tile(k);
          iteration(setB, k + 1);
        }
        drain_tile(nit - 1);
      }
      __syncthreads();  // end of item: tiles drained, the ring may be primed again (+ the statistics hand-over)
    }
    return;
  }

  // ================================================================================== computing waves
  const int wn = wave & 1, wm = wave >> 1;
  const bf16_t* __restrict__ wgt = reinterpret_cast<const bf16_t*>(p.w);
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = wn * 32;
  // every weight of this wave's 32 output channels, stationary for the whole launch: [tap][channel chunk][N tile]
  s16x8 wq[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wq[t][c][j] = *reinterpret_cast<const s16x8*>(wgt + (size_t)(n0 + j * 16 + fr) * 9 * K::C + t * K::C + c * 32 + fq * 8);
  // the wave's two M tiles: low-res columns jcol[i] (columns >= 56 of the last tile are padding)
  int pix[2], cellb[2];
  bool valid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = (wm * 2 + i) * 16 + fr;
    valid[i] = j < K::WL;
    const int jc = valid[i] ? j : K::WL - 1;
    pix[i] = jc * K::PSTR + fq * 16;
    cellb[i] = (KIND == 0 ? jc : 2 * jc) * K::PSTR + (n0 + fq * 4) * 2;
  }

#pragma unroll 1
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int seg = item % nseg;
    const int r0 = seg * nit;
    float s0[2][4], s1[2][4];  // column sums carried over the whole walk
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s0[j][r] = s1[j][r] = 0.f;
    __syncthreads();  // the ring is primed

#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
      const int r = r0 + it;
      // ring rows of this iteration
      constexpr int NRO = KIND == 0 ? 3 : 2;
      int abase[2][NRO];
#pragma unroll
      for (int d = 0; d < NRO; ++d) {
        const int slot = KIND == 0 ? (2 * r + d + 1) % K::F_NR : (r + d) % K::D_NR;
#pragma unroll
        for (int i = 0; i < 2; ++i) abase[i][d] = slot * (KIND == 0 ? K::F_RSTR : K::D_RSTR) + pix[i];
      }
      char* const tile = smem + LY::TILE_OFF + (it & 1) * LY::TILE;

      f32x4 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

      constexpr int D = 6;  // A-fragment ring depth (steps of LDS-read lead)
      s16x8 ring[D];
      auto a_addr = [&](auto tag) -> const s16x8* {
        constexpr Step d = step_of<KIND>(decltype(tag)::value);
        // forward: kernel column 0 / 1 / 2 = odd plane [j], even plane [j], odd plane [j + 1]
        constexpr int coloff = KIND == 0 ? (d.coff == 1 ? 0 : (d.coff == 0 ? K::F_OOFF : K::F_OOFF + K::PSTR)) : d.coff * K::PSTR;
        return reinterpret_cast<const s16x8*>(smem + abase[d.i][d.roff] + coloff + d.chunk * 64);
      };
      static_for<0, D>([&](auto tag) { ring[decltype(tag)::value] = *a_addr(tag); });

      // epilogue cells of one class (forward: the only "class")
      auto cells = [&](auto ctag) {
        constexpr int P = decltype(ctag)::value;
        constexpr int cellc = KIND == 0 ? 0 : ((P >> 1) * K::WH + (P & 1)) * K::PSTR;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float ea[4];
          if (KIND == 1) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(coef + 2 * K::C + n0 + j * 16 + fq * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) ea[q] = t[q];
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (!valid[i]) continue;
            uint2* cell = reinterpret_cast<uint2*>(tile + cellb[i] + cellc + j * 32);
            float v[4], x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q];
            if (KIND == 1) {
              const uint2 u = *cell;
              x[0] = __uint_as_float(u.x << 16);
              x[1] = __uint_as_float(u.x & 0xFFFF0000u);
              x[2] = __uint_as_float(u.y << 16);
              x[3] = __uint_as_float(u.y & 0xFFFF0000u);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const bool pos = x[q] > 0.f;
                s0[j][q] += pos ? 0.f : v[q] * x[q];
                v[q] = pos ? v[q] : v[q] * ea[q];
              }
            } else if (epi == FR_EPI_STATS) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                s0[j][q] += v[q];
                s1[j][q] = fmaf(v[q], v[q], s1[j][q]);
              }
            }
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *cell = o;
          }
        }
      };

      static_for<0, K::NSTEP>([&](auto tag) {
        constexpr int s = decltype(tag)::value;
        constexpr Step d = step_of<KIND>(s);
        const s16x8 a = ring[s % D];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[d.i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[d.ktap][d.chunk][j], a, acc[d.i][j], 0, 0, 0);
        if constexpr (s + D < K::NSTEP) ring[s % D] = *a_addr(std::integral_constant<int, s + D>{});
        if constexpr (d.last != 0) {
          cells(std::integral_constant<int, d.cls>{});
          if constexpr (s + 1 < K::NSTEP) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      });
      __syncthreads();  // the tile is complete; the data-moving waves have committed the rows of it + 1
    }

    // ---------------------------------------------------------------- one partial row per work item
    float* red = reinterpret_cast<float*>(smem + LY::RED_OFF);
    const bool sums = KIND == 1 || epi == FR_EPI_STATS;
    if (sums) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = s0[j][q], c = s1[j][q];
#pragma unroll
          for (int o2 = 1; o2 < 16; o2 <<= 1) {
            a += __shfl_xor(a, o2, 64);
            c += __shfl_xor(c, o2, 64);
          }
          if (fr == 0) {
            red[(wm * 2 + 0) * K::C + n0 + j * 16 + fq * 4 + q] = a;
            red[(wm * 2 + 1) * K::C + n0 + j * 16 + fq * 4 + q] = c;
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

int s2roll_nseg(int B) {
  // row segments per image: enough workgroups for one per CU on 256 CUs; 56 rows must divide evenly
  static const int cand[6] = {1, 2, 4, 7, 14, 28};  // rows per walk 56 / nseg: even (the data-moving loop is unrolled by two)
  static const int* fslot = fr_option_slot("FRHIP_S2ROLL_NSEG", -1);  // test hook: the tests walk whole images with small batches
  const int forced = *fslot;
  for (int k = 0; k < 6; ++k)
    if (forced == cand[k]) return cand[k];
  for (int k = 0; k < 6; ++k)
    if ((long long)B * cand[k] >= 256) return cand[k];
  return 28;
}

template <int KIND, int PRO>
int launch(const FrConvArgs& a, hipStream_t st) {
  using LY = S2R::L<KIND>;
  static unsigned long long attr_done = 0;  // one bit per device
  if (fr_attr_needed(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_s2_roll64_kernel<KIND, PRO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LY::LDS);
    fr_attr_done(attr_done);
  }
  const int nseg = s2roll_nseg(a.B);
  const int items = a.B * nseg;
  const int grid = items < 256 ? items : 256;  // persistent: weights are loaded into registers once per workgroup
  FR_LAUNCH_KERNEL((conv3x3_s2_roll64_kernel<KIND, PRO>), dim3(grid), dim3(S2R::NTH), LY::LDS, st, a, nseg, items);
  FR_LAUNCH_CHECK();
}

}  // namespace

// FRHIP_ROLL64=0: the 64-channel layers (this stride-2 one too) stay on the LDS-strip kernels (A/B switch)
bool fr_s2roll_enabled() { return fr_roll64_enabled(); }

// served: 64 -> 64, low-res side 56; forward (mode 0) with the STORE / STATS epilogues, gradient (mode 2) with PReLU backward
bool fr_s2roll_serves(const FrConvArgs& a) {
  if (!fr_s2roll_enabled() || a.SC != 64 || a.N != 64) return false;
  if (a.mode == 0) return a.RW == 56 && (a.epi == FR_EPI_STORE || a.epi == FR_EPI_STATS);
  return a.mode == 2 && a.SW == 56 && a.epi == FR_EPI_PRELU_BWD && a.pro == FR_PRO_NONE && a.aux && a.epi_a;
}

int fr_s2roll_parts(int B) { return B * s2roll_nseg(B); }

int fr_s2roll_launch(const FrConvArgs& a, hipStream_t st) {
  if (a.mode == 2) return launch<1, FR_PRO_NONE>(a, st);
  switch (a.pro) {
    case FR_PRO_NONE: return launch<0, FR_PRO_NONE>(a, st);
    case FR_PRO_BN: return launch<0, FR_PRO_BN>(a, st);
    case FR_PRO_PRELU: return launch<0, FR_PRO_PRELU>(a, st);
  }
  FR_UNSUPPORTED("rolling-window convolution: prologue / epilogue combination not served");
}
