// frhip -- error string + version + capability query of the C ABI.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "frhip_internal.h"

static thread_local char g_err[256] = "";

extern "C" void fr_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* fr_last_error_string(void) { return g_err; }
extern "C" int fr_abi_version(void) { return FR_ABI_VERSION; }
extern "C" int fr_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(FrConvArgs);
    case 1: return (int)sizeof(FrWgradArgs);
    case 2: return (int)sizeof(FrApplyArgs);
    case 3: return (int)sizeof(FrBnBwdArgs);
    case 4: return (int)sizeof(FrSgdTensor);
    case 5: return (int)sizeof(FrPackTensor);
    case 6: return (int)sizeof(FrAdamTensor);
    case 7: return (int)sizeof(FrBnEvalEntry);
    case 8: return (int)sizeof(FrTail);
  }
  return -1;
}

// Workgroups that share the in-launch reduction of a launch's partial rows (tail.h) when the caller leaves FrTail.nred 0:
// enough that every 256-thread reduction block takes one 8-column group, capped here.  FRHIP_TAIL_NRED overrides the cap
// (1 = the last workgroup to arrive adds everything alone).
int fr_tail_default_nred() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("FRHIP_TAIL_NRED");
    v = e ? atoi(e) : 16;
    if (v < 1) v = 1;
    if (v > 64) v = 64;
  }
  return v;
}

namespace {
__global__ void fill_rows_kernel(float* __restrict__ out, const float* __restrict__ bias, long long n, int C) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = bias ? bias[(int)(i % C)] : 0.f;
}
}  // namespace

extern "C" int fr_fill_rows(float* out, const float* bias, long long rows, int C, void* stream) {
  const long long n = rows * C;
  long long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(fill_rows_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, out, bias, n, C);
  FR_LAUNCH_CHECK();
}
