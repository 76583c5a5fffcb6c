// tools/cu_hog -- holds K compute units for a bounded time (stands in for a co-resident RCCL kernel on a 1-GPU box).
// One workgroup per held CU: 96 KB of LDS (two cannot share a CU, and a 142-KB strip workgroup does not fit beside one),
// 256 threads spinning on the 100-MHz wall clock until `seconds` have passed.  Bounded: never waits on anything.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/cu_hog.hip -o gpurun_out/libcuhog.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void cu_hog_kernel(unsigned long long ticks, unsigned* sink) {
  extern __shared__ char lds[];
  const unsigned long long t0 = wall_clock64();
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) {
    lds[threadIdx.x] = (char)acc;  // keep the allocation live
    acc += lds[(threadIdx.x + 1) & 255];
    __builtin_amdgcn_s_sleep(16);
  }
  if (acc == 0xFFFFFFFFu && sink) sink[0] = acc;
}

extern "C" int cu_hog_launch(int K, double seconds, void* stream) {
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cu_hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              96 * 1024);
    once = true;
  }
  if (K < 1) return 0;
  const unsigned long long ticks = (unsigned long long)(seconds * 1e8);  // wall_clock64: 100 MHz
  hipLaunchKernelGGL(cu_hog_kernel, dim3(K), dim3(256), 96 * 1024, (hipStream_t)stream, ticks, (unsigned*)nullptr);
  return (int)hipGetLastError();
}
