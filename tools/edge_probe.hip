// What a main -> side stream dependency edge costs the main stream, and whether the completion signal of the kernel itself
// (hipExtLaunchKernelGGL's stopEvent) is cheaper than a marker behind it (hipEventRecord).  Round 6.
//   hipcc --offload-arch=gfx950 -O2 tools/edge_probe.hip -o tools/edge_probe && tools/edge_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void busy(float* x, size_t n, float a) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = x[i] * a + 1.f;
}
__global__ void tiny(float* x) { x[threadIdx.x] += 1.f; }

#define CK(x)                                                               \
  do {                                                                      \
    hipError_t err_ = (x);                                                     \
    if (err_ != hipSuccess) {                                                  \
      printf("%s -> %s\n", #x, hipGetErrorString(err_));                       \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 200;
  float *x, *t;
  const size_t n = 16u << 20;
  CK(hipMalloc(&x, n * 4));
  CK(hipMalloc(&t, 4096));
  CK(hipMemset(x, 0, n * 4));
  CK(hipMemset(t, 0, 4096));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, 1));
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  const char* names[] = {"none", "record+wait", "stopEvent+wait", "record only", "stopEvent only", "waitback(record on side)",
                         "waitback(stopEvent on side)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int v = 0; v < 7; ++v) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s0));
      for (int i = 0; i < N; ++i) {
        if (v == 2 || v == 4) {
          hipExtLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, s0, nullptr, ev[i], 0, x, n, 1.0001f);
        } else {
          hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, s0, x, n, 1.0001f);
        }
        if (v == 1 || v == 3) CK(hipEventRecord(ev[i], s0));
        if (v == 1 || v == 2) {
          CK(hipStreamWaitEvent(s1, ev[i], 0));
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, t);
        }
        if (v == 5) {
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, t);
          CK(hipEventRecord(ev[i], s1));
          CK(hipStreamWaitEvent(s0, ev[i], 0));
        }
        if (v == 6) {
          hipExtLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, nullptr, ev[i], 0, t);
          CK(hipStreamWaitEvent(s0, ev[i], 0));
        }
      }
      CK(hipEventRecord(t1, s0));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep) printf("%-30s %.2f us per main-stream kernel\n", names[v], ms * 1e3 / N);
    }
  return 0;
}
