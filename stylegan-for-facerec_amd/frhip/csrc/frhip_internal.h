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
