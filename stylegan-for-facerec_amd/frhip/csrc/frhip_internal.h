// internal view of the public C ABI (include/frhip.h is on the include path)
#pragma once
#include "frhip.h"

#include <hip/hip_runtime.h>

// out[i] = sum_g slab[g][i] in the fixed order g = 0, 1, ... (n elements, n % 4 == 0); conv_wgrad_strip.hip
int fr_launch_reduce_slabs(const float* slab, int groups, long long n, float* out, hipStream_t st);
