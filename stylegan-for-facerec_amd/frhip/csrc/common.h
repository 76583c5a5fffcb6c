// frhip -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes; MFMA 16x16x32 (bf16) / 16x16x4 (f32) fragments; all math accumulates in fp32.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <utility>

typedef uint16_t bf16_t;  // raw bfloat16 storage

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

struct alignas(16) U128 {
  uint32_t x, y, z, w;
};

#define FR_WAVE 64

// ---------------------------------------------------------------------------------------------------------
// error plumbing (host side): 0 = ok, <0 = unsupported argument, >0 = hipError_t
// ---------------------------------------------------------------------------------------------------------
extern "C" void fr_set_error(const char* msg);
#define FR_UNSUPPORTED(msg) \
  do {                      \
    fr_set_error(msg);      \
    return -1;              \
  } while (0)
#define FR_LAUNCH_CHECK()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) {                   \
      fr_set_error(hipGetErrorString(e__));    \
      return (int)e__;                         \
    }                                          \
    return 0;                                  \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// Completion event of a kernel (fr_arm_stop_event, include/frhip.h).  A dependency edge main stream -> weight-gradient
// stream set with hipEventRecord puts a marker packet behind the producer, and the NEXT kernel of the main stream waits for
// that marker: +3.0 us per edge with nobody waiting, +5.1 us with a waiter (tools/edge_probe.hip, profiles/r06_edge_probe.txt).
// The producer's own completion signal does the same job for +0.0 / +1.6 us: hipExtLaunchKernel(..., stopEvent).  The
// launchers of the kernels that precede such edges go through FR_LAUNCH_KERNEL, which hands an armed event to the first
// launch after fr_arm_stop_event (per host thread) and is hipLaunchKernelGGL otherwise.
// ---------------------------------------------------------------------------------------------------------
extern thread_local hipEvent_t fr_tls_stop_event;   // armed event (nullptr: none)
extern thread_local int fr_tls_stop_launches;       // FR_LAUNCH_KERNEL launches since it was armed
template <typename... P, typename... A, size_t... I>
inline void fr_launch_ext_(void (*kern)(P...), dim3 g, dim3 b, unsigned lds, hipStream_t st, hipEvent_t stop,
                           std::index_sequence<I...>, A&&... a) {
  std::tuple<P...> params{static_cast<P>(std::forward<A>(a))...};  // the kernel's own parameter types, by value
  void* ptrs[] = {static_cast<void*>(&std::get<I>(params))...};
  (void)hipExtLaunchKernel(reinterpret_cast<const void*>(kern), g, b, ptrs, lds, st, nullptr, stop, 0);
}
template <typename... P, typename... A>
inline void fr_launch_ext(void (*kern)(P...), dim3 g, dim3 b, unsigned lds, hipStream_t st, hipEvent_t stop, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
  fr_launch_ext_(kern, g, b, lds, st, stop, std::index_sequence_for<P...>{}, std::forward<A>(a)...);
}
#define FR_LAUNCH_KERNEL(kern, grid, block, lds, st, ...)                                    \
  do {                                                                                        \
    if (fr_tls_stop_event && fr_tls_stop_launches++ == 0)                                     \
      fr_launch_ext(kern, grid, block, lds, st, fr_tls_stop_event, __VA_ARGS__);              \
    else                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                            \
  } while (0)

// hipFuncSetAttribute (dynamic LDS above 64 KB) is per function AND per device: launchers keep one bit per device.  The mask
// is updated atomically (host threads driving different devices may launch the same instance for the first time together);
// a thread that loses the race sets the attribute once more, which is harmless.  Devices 64 and above always set it.
inline bool fr_attr_needed(unsigned long long& done_mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 64) return true;
  const unsigned long long bit = 1ull << dev;
  if (__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & bit) return false;
  return true;
}
inline void fr_attr_done(unsigned long long& done_mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 64) __atomic_fetch_or(&done_mask, 1ull << dev, __ATOMIC_RELEASE);
}

// ---------------------------------------------------------------------------------------------------------
// bf16 <-> f32
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// plain cast so hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving -- MI355X_MICROARCH.md correctness table)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

template <typename T>
struct Elt;  // element traits
template <>
struct Elt<float> {
  static constexpr int VEC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <>
struct Elt<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// unpack a 16-byte chunk into VEC floats / pack back
template <typename T>
__device__ __forceinline__ void unpack16(const U128& u, float* f);
template <>
__device__ __forceinline__ void unpack16<float>(const U128& u, float* f) {
  f[0] = __uint_as_float(u.x);
  f[1] = __uint_as_float(u.y);
  f[2] = __uint_as_float(u.z);
  f[3] = __uint_as_float(u.w);
}
template <>
__device__ __forceinline__ void unpack16<bf16_t>(const U128& u, float* f) {
  f[0] = __uint_as_float(u.x << 16);
  f[1] = __uint_as_float(u.x & 0xFFFF0000u);
  f[2] = __uint_as_float(u.y << 16);
  f[3] = __uint_as_float(u.y & 0xFFFF0000u);
  f[4] = __uint_as_float(u.z << 16);
  f[5] = __uint_as_float(u.z & 0xFFFF0000u);
  f[6] = __uint_as_float(u.w << 16);
  f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
template <typename T>
__device__ __forceinline__ U128 pack16(const float* f);
template <>
__device__ __forceinline__ U128 pack16<float>(const float* f) {
  U128 u;
  u.x = __float_as_uint(f[0]);
  u.y = __float_as_uint(f[1]);
  u.z = __float_as_uint(f[2]);
  u.w = __float_as_uint(f[3]);
  return u;
}
template <>
__device__ __forceinline__ U128 pack16<bf16_t>(const float* f) {
  U128 u;
  u.x = pack2bf(f[0], f[1]);
  u.y = pack2bf(f[2], f[3]);
  u.z = pack2bf(f[4], f[5]);
  u.w = pack2bf(f[6], f[7]);
  return u;
}

__device__ __forceinline__ U128 ld16(const void* p) { return *reinterpret_cast<const U128*>(p); }
__device__ __forceinline__ void st16(void* p, const U128& v) { *reinterpret_cast<U128*>(p) = v; }
__device__ __forceinline__ U128 zero16() {
  U128 u;
  u.x = u.y = u.z = u.w = 0u;
  return u;
}

// one partial sum of a workgroup's row of part[row][K][C] (added by a later launch: fr_bn_finalize / fr_reduce_parts)
__device__ __forceinline__ void st_part(float* p, float v) { *p = v; }

// ---------------------------------------------------------------------------------------------------------
// wave / block reductions (64-wide)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// exact-enough unsigned divide for n < 2^24 by a runtime divisor (inv = 1.0f/d): q = n / d, r = n % d
__device__ __forceinline__ void fast_divmod(uint32_t n, uint32_t d, float inv, uint32_t& q, uint32_t& r) {
  q = (uint32_t)((float)n * inv);
  int rr = (int)n - (int)(q * d);
  if (rr < 0) {
    q -= 1;
    rr += (int)d;
  } else if (rr >= (int)d) {
    q += 1;
    rr -= (int)d;
  }
  r = (uint32_t)rr;
}

// Workgroups are dealt round-robin over the 8 XCDs (block b runs on the XCD that also runs b + 8, MI355X_MICROARCH.md,
// "Workgroup dispatch"), each with its own L2.  xcd_remap turns a block id into a LOGICAL work index such that every
// XCD owns one contiguous range of the n items and walks it in order: neighbouring items (strips that share halo
// rows, the two channel halves of one strip) then run on the same XCD at about the same time and meet in its L2.
// Placement only changes speed, never results.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int x = b & 7, k = b >> 3;
  return x * q + (x < r ? x : r) + k;
}

// ---------------------------------------------------------------------------------------------------------
// MFMA wrappers: one 16x16 output tile, K = 32 per call.  Fragment convention (both dtypes):
//   lane l holds A[row = l&15][k = 8*(l>>4) + j] and B[k = 8*(l>>4) + j][col = l&15], j = 0..7
//   C/D: col = l&15, row = 4*(l>>4) + reg                       (cdna_hip_programming.md section 3)
// For f32 the eight k of a lane are fed to eight 16x16x4 MFMAs (MFMA j takes element j of every lane):
// a permutation of k that is identical for A and B, hence the same dot product.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct Frag;
template <>
struct Frag<bf16_t> {
  s16x8 v;
};
template <>
struct Frag<float> {
  float v[8];
};

__device__ __forceinline__ f32x4 mma16(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
  return c;
}

// read a row-major fragment: 8 consecutive elements (k) at `p` (16-byte aligned) from LDS
__device__ __forceinline__ void lds_frag_row(Frag<bf16_t>& f, const bf16_t* p) {
  f.v = *reinterpret_cast<const s16x8*>(p);
}
__device__ __forceinline__ void lds_frag_row(Frag<float>& f, const float* p) {
  const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  f.v[0] = lo[0];
  f.v[1] = lo[1];
  f.v[2] = lo[2];
  f.v[3] = lo[3];
  f.v[4] = hi[0];
  f.v[5] = hi[1];
  f.v[6] = hi[2];
  f.v[7] = hi[3];
}
