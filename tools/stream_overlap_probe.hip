// Do two INDEPENDENT kernels enqueued back to back on ONE HIP stream overlap on the GPU?  (VERDICT r4 item 8 assumes the tail
// launch of a split convolution starts on the CUs the first launch leaves free.)
//   hipcc --offload-arch=gfx950 -O3 tools/stream_overlap_probe.hip -o /tmp/sop && /tmp/sop
// A: 240 workgroups x 150 KB of LDS (one per CU) spinning ~60 us; B: 16 such workgroups spinning ~20 us.
// one stream: A then B;  two streams: A || B.  Overlapped = ~60 us, serialised = ~80 us.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void spin(long long ticks, int* sink) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks < 0) sink[0] = smem[0];
}

int main() {
  int* sink;
  hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipStream_t s1, s2;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  hipEvent_t e0, e1, f;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventCreate(&f);
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s1);
      hipLaunchKernelGGL(spin, dim3(240), dim3(512), 150 * 1024, s1, 6000, sink);
      if (mode == 1) hipLaunchKernelGGL(spin, dim3(16), dim3(512), 150 * 1024, s1, 2000, sink);
      if (mode == 2) {
        hipLaunchKernelGGL(spin, dim3(16), dim3(512), 150 * 1024, s2, 2000, sink);
        hipEventRecord(f, s2);
        hipStreamWaitEvent(s1, f, 0);
      }
      hipEventRecord(e1, s1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-44s %.1f us\n", mode == 0 ? "A alone (240 WGs x 60 us)" : mode == 1 ? "A then B (16 WGs x 20 us), ONE stream" : "A on stream 1, B on stream 2", best * 1e3f);
  }
  return 0;
}
