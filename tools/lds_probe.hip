// Micro-probe: LDS cycles per ds_read_b128 wave-instruction for lane->address patterns used by the strip kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;

__global__ __launch_bounds__(256) void probe(const int* __restrict__ lane_addr, long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 40960; i += 256) reinterpret_cast<int*>(smem)[i] = i;
  __syncthreads();
  const int a = lane_addr[threadIdx.x & 63] + (threadIdx.x >> 6) * 16 * 528;  // each wave its own 16 rows
  i32x4 acc = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    i32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (acc[0] == 123456789) out[1] = acc[1];
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  struct Pat { const char* name; int (*f)(int); };
  Pat pats[] = {
      {"linear l*16", [](int l) { return l * 16; }},
      {"row=l&15 stride 528, fq*16", [](int l) { return (l & 15) * 528 + (l >> 4) * 16; }},
      {"row stride 512 (no pad)", [](int l) { return (l & 15) * 512 + (l >> 4) * 16; }},
      {"row stride 528, fq*64", [](int l) { return (l & 15) * 528 + (l >> 4) * 64; }},
      {"row stride 544, fq*16", [](int l) { return (l & 15) * 544 + (l >> 4) * 16; }},
      {"row stride 592 (37 slots), fq*16", [](int l) { return (l & 15) * 592 + (l >> 4) * 16; }},
      {"row stride 144 (C=64), fq*16", [](int l) { return (l & 15) * 144 + (l >> 4) * 16; }},
      {"row stride 272 (C=128), fq*16", [](int l) { return (l & 15) * 272 + (l >> 4) * 16; }},
      {"row stride 528, fq*16, wrap +224 at row 14", [](int l) { int r = l & 15; return r * 528 + (r >= 14 ? 224 + 2 * 528 : 0) + (l >> 4) * 16; }},
      {"row stride 528+64, fq*16", [](int l) { return (l & 15) * 592 + (l >> 4) * 16; }},
      {"row stride 528, fq*16 swizzled ((r>>2)&3)^fq", [](int l) { int r = l & 15; return r * 528 + ((((r >> 2) & 3) ^ (l >> 4)) * 16); }},
  };
  int* d_addr;
  long long* d_out;
  hipMalloc(&d_addr, 64 * 4);
  hipMalloc(&d_out, 16);
  for (auto& p : pats) {
    std::vector<int> h(64);
    for (int l = 0; l < 64; ++l) h[l] = p.f(l);
    hipMemcpy(d_addr, h.data(), 256, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 160 * 1024, 0, d_addr, d_out, iters);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 160 * 1024, 0, d_addr, d_out, iters);
    hipDeviceSynchronize();
    long long cyc;
    hipMemcpy(&cyc, d_out, 8, hipMemcpyDeviceToHost);
    // 4 waves x iters x 16 reads share the CU's LDS
    printf("%-52s %6.2f cycles per wave-instruction (4 waves)\n", p.name, (double)cyc / (iters * 8.0 * 4.0));
  }
  return 0;
}
