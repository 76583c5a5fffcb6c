// frhip -- error string + version + capability query of the C ABI.
#include <stdlib.h>
#include <string.h>

#include <mutex>

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
    case 8: return (int)sizeof(FrBnFinArgs);
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------------------
// Run-time switches (kernel-family A/B switches and test hooks).  A switch is an int, first read from the environment
// variable of its name (unset = the default its first reader passes), cached in a slot for the life of the process and
// overridable through fr_set_option (tests flip kernel families inside one process).  Launchers keep a pointer to the
// slot: no getenv on the launch path.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct Opt {
  char name[48];
  int value;
};
Opt g_opts[96];
int g_nopts = 0;
std::mutex g_opt_mu;
Opt* find_opt(const char* name) {
  for (int i = 0; i < g_nopts; ++i)
    if (!strcmp(g_opts[i].name, name)) return &g_opts[i];
  return nullptr;
}
}  // namespace

int* fr_option_slot(const char* name, int dflt) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  Opt* o = find_opt(name);
  if (!o) {
    if (g_nopts >= (int)(sizeof(g_opts) / sizeof(g_opts[0]))) abort();
    o = &g_opts[g_nopts++];
    strncpy(o->name, name, sizeof(o->name) - 1);
    const char* e = getenv(name);
    o->value = (e && e[0]) ? atoi(e) : dflt;
  }
  return &o->value;
}

extern "C" int fr_set_option(const char* name, int value) {
  int* slot = fr_option_slot(name, value);
  const int old = *slot;
  *slot = value;
  return old;
}

extern "C" int fr_get_option(const char* name, int dflt) { return *fr_option_slot(name, dflt); }

// ---------------------------------------------------------------------------------------------------------
// Completion event of the next kernel (common.h, FR_LAUNCH_KERNEL)
// ---------------------------------------------------------------------------------------------------------
thread_local hipEvent_t fr_tls_stop_event = nullptr;
thread_local int fr_tls_stop_launches = 0;

extern "C" int fr_arm_stop_event(void* event) {
  fr_tls_stop_event = reinterpret_cast<hipEvent_t>(event);
  fr_tls_stop_launches = 0;
  return 0;
}

extern "C" int fr_finish_stop_event(void* stream) {
  hipEvent_t ev = fr_tls_stop_event;
  const int n = fr_tls_stop_launches;
  fr_tls_stop_event = nullptr;
  fr_tls_stop_launches = 0;
  if (!ev) FR_UNSUPPORTED("fr_finish_stop_event: no event armed on this thread");
  if (n != 1) {  // no launcher took it, or more kernels followed the one that did: a marker behind all of them
    hipError_t e = hipEventRecord(ev, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
      fr_set_error(hipGetErrorString(e));
      return -(1000 + (int)e);
    }
  }
  return n;
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
