// internal view of the public C ABI (include/frhip.h is on the include path)
#pragma once
#include "frhip.h"

#include <hip/hip_runtime.h>

// run-time switch `name` (api.hip): pointer to its cached value (environment variable of that name, else dflt)
int* fr_option_slot(const char* name, int dflt);

// out[i] = sum_g slab[g][i] in the fixed order g = 0, 1, ... (n elements, n % 4 == 0); conv_wgrad_strip.hip
int fr_launch_reduce_slabs(const float* slab, int groups, long long n, float* out, hipStream_t st);

// 64 -> 64 stride-1 3x3 layers on the rolling-window kernel (conv3x3_roll64.hip); dispatched from fr_conv3x3_strip
bool fr_roll64_enabled();
int fr_roll64_parts(int B, int W);
int fr_roll64_launch(const FrConvArgs& a, hipStream_t st);

// 64-channel stride-2 3x3 layer (112 -> 56) and its data gradient on the rolling-window kernel (conv3x3_s2_roll64.hip);
// dispatched from fr_conv3x3_s2_strip
bool fr_s2roll_serves(const FrConvArgs& a);
int fr_s2roll_parts(int B);
int fr_s2roll_launch(const FrConvArgs& a, hipStream_t st);
bool fr_s2roll_enabled();

// 3x3 stride-1 weight gradients at 14x14 on the warp-specialised kernel (conv_wgrad_roll.hip); dispatched from
// fr_conv_wgrad_strip
bool fr_wgrad_roll_enabled();
bool fr_wgrad_roll_serves(const FrWgradArgs& a);
int fr_wgrad_roll_launch(const FrWgradArgs& a, hipStream_t st);
// ... and the stride-2 ones with a 28 / 14 / 7 wide gradient (conv_wgrad_s2roll_kernel, same file)
bool fr_wgrad_s2roll_serves(const FrWgradArgs& a);
int fr_wgrad_s2roll_launch(const FrWgradArgs& a, hipStream_t st);

// stride-2 3x3 forward at 128 / 256 / 512 channels on the warp-specialised kernel (conv3x3_s2_ws.hip); dispatched from
// fr_conv3x3_s2_strip.  fr_s2ws_strips: partial-sum rows of a served (B, C, low-res width), 0 = not served.
bool fr_s2ws_serves(const FrConvArgs& a);
int fr_s2ws_strips(int B, int C, int WL);
int fr_s2ws_launch(const FrConvArgs& a, hipStream_t st);

#if defined(__HIPCC__)
// prologue on one dword (two bf16): BN apply or PReLU, fp32 arithmetic, one rounding back to bf16
// Written as instructions: from the equivalent C the compiler rebuilds 16-bit compares + v_cndmask + v_perm (49 vector
// instructions per 16-byte chunk instead of 28) -- and these run on the SIMDs whose issue slots the MFMA waves need.
template <int PRO>
__device__ __forceinline__ uint32_t pro2(uint32_t u, float a0, float b0, float a1, float b1) {
  uint32_t lo, hi, r;
  if (PRO == FR_PRO_BN) {
    asm("v_lshlrev_b32 %0, 16, %3\n\t"
        "v_and_b32 %1, 0xffff0000, %3\n\t"
        "v_fma_f32 %0, %0, %4, %5\n\t"
        "v_fma_f32 %1, %1, %6, %7\n\t"
        "v_cvt_pk_bf16_f32 %2, %0, %1"
        : "=&v"(lo), "=&v"(hi), "=v"(r)
        : "v"(u), "v"(a0), "v"(b0), "v"(a1), "v"(b1));
    return r;
  }
  // PReLU: x > 0 ? x : a x.  Both halves scaled and packed; v_pk_ashrrev_i16 spreads the two sign bits into a mask and
  // v_bfi_b32 takes the scaled half where the sign is set, the input half elsewhere.  Same values as the compare form:
  // -0 and negative NaNs go through a x (x > 0 is false for them too), a x of a positive x is never selected.
  uint32_t m;
  asm("v_lshlrev_b32 %0, 16, %4\n\t"
      "v_and_b32 %1, 0xffff0000, %4\n\t"
      "v_mul_f32 %0, %0, %5\n\t"
      "v_mul_f32 %1, %1, %6\n\t"
      "v_cvt_pk_bf16_f32 %0, %0, %1\n\t"
      "v_pk_ashrrev_i16 %2, 15, %4 op_sel_hi:[0,1]\n\t"
      "v_bfi_b32 %3, %2, %0, %4"
      : "=&v"(lo), "=&v"(hi), "=&v"(m), "=v"(r)
      : "v"(u), "v"(a0), "v"(a1));
  return r;
}
#endif
