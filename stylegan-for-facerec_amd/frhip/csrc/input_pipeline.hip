// GPU-side training-input transform: staged uint8 HWC images -> the float32 NCHW batch the reference's
// host transform produces (train.py:108-116: Resize(128*S/112) -> RandomCrop(S) -> RandomHorizontalFlip -> ToTensor ->
// Normalize), one launch per batch.
//
// The resize is Pillow's 8-bit bilinear resample restated exactly (src/libImaging/Resample.c: integer weights with 22
// fractional bits, horizontal pass rounded to uint8, then the vertical pass), so the result is bit-identical to the host
// path; the weights and windows come precomputed per axis (frhip/input_pipeline.py, the same table Pillow builds).  Only
// the S x S crop is ever computed: every output pixel evaluates its <= ky rows of <= kx horizontal taps straight from the
// staged image (uint8 reads served by L2; 9.6 MB in, 38.5 MB out at B = 256), ToTensor + Normalize are a 256-entry
// table per channel built on the host in float32, so there is no device-side division to disagree about.
//
// HBM-bound byte work: one thread per output pixel, x fastest -> each of the three channel planes is written with
// fully coalesced 256-byte wave stores.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// xtab / ytab rows: [first input index, tap count, k_0 .. k_{K-1}]
__global__ __launch_bounds__(256) void augment_u8_kernel(const unsigned char* __restrict__ src,
                                                         const int* __restrict__ xtab, const int* __restrict__ ytab,
                                                         const int* __restrict__ crop,
                                                         const unsigned char* __restrict__ flip,
                                                         const float* __restrict__ lut, float* __restrict__ out,
                                                         int Hin, int Win, int S, int kx, int ky) {
  __shared__ float slut[768];
  for (int i = threadIdx.x; i < 768; i += 256) slut[i] = lut[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= S * S) return;
  const int y = pix / S, x = pix - y * S;
  const int x0 = crop[2 * b], y0 = crop[2 * b + 1];
  const int rx = x0 + (flip[b] ? S - 1 - x : x), ry = y0 + y;  // position in the resized image
  const int* xr = xtab + (size_t)rx * (kx + 2);
  const int* yr = ytab + (size_t)ry * (ky + 2);
  const int xmin = xr[0], xn = xr[1], ymin = yr[0], yn = yr[1];
  const unsigned char* img = src + (size_t)b * Hin * Win * 3;
  int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
  for (int t = 0; t < yn; ++t) {
    const unsigned char* row = img + ((size_t)(ymin + t) * Win + xmin) * 3;
    int h[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int j = 0; j < xn; ++j) {
      const int k = xr[2 + j];
      h[0] += row[3 * j] * k;
      h[1] += row[3 * j + 1] * k;
      h[2] += row[3 * j + 2] * k;
    }
    const int kv = yr[2 + t];
    acc[0] += clip8(h[0]) * kv;  // the horizontal pass is rounded to uint8 before the vertical pass reads it
    acc[1] += clip8(h[1]) * kv;
    acc[2] += clip8(h[2]) * kv;
  }
  const size_t plane = (size_t)S * S;
  float* o = out + (size_t)b * 3 * plane + pix;
  o[0] = slut[clip8(acc[0]) * 3];
  o[plane] = slut[clip8(acc[1]) * 3 + 1];
  o[2 * plane] = slut[clip8(acc[2]) * 3 + 2];
}

}  // namespace

extern "C" int fr_augment_u8(const uint8_t* src, const int32_t* xtab, const int32_t* ytab, const int32_t* crop,
                             const uint8_t* flip, const float* lut, float* out, int B, int Hin, int Win, int Hr, int Wr,
                             int S, int kx, int ky, void* stream) {
  if (B <= 0) return 0;
  if (Hin <= 0 || Win <= 0 || S <= 0 || S > Hr || S > Wr || kx <= 0 || ky <= 0)
    FR_UNSUPPORTED("fr_augment_u8: sizes (need 0 < S <= resized size, at least one tap per axis)");
  if (B > 65535) FR_UNSUPPORTED("fr_augment_u8: more than 65535 images per launch");
  dim3 grid((S * S + 255) / 256, B);
  hipLaunchKernelGGL(augment_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, xtab, ytab, crop, flip, lut, out,
                     Hin, Win, S, kx, ky);
  FR_LAUNCH_CHECK();
}


// ------------------------------------------------------------------------------------------ bilinear resize (pSp)
// pSp.forward resizes a batch whose side differs from the encoder's input size with
// F.interpolate(x, size, mode='bilinear') (reference backbone/restyle_psp.py:440-443): align_corners = False, no
// antialiasing -- ATen's upsample_bilinear2d restated: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out in
// float, the two taps weighted 1 - l and l in float32.  planes = B * C NCHW planes; one thread per output pixel.
namespace {
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int Hin, int Win, int Hout, int Wout, float sh, float sw) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= Hout * Wout) return;
  const int oy = pix / Wout, ox = pix - oy * Wout;
  float fy = sh * ((float)oy + 0.5f) - 0.5f, fx = sw * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* pl = in + (size_t)blockIdx.y * Hin * Win;
  const float v = hy * (hx * pl[(size_t)y0 * Win + x0] + lx * pl[(size_t)y0 * Win + x1]) +
                  ly * (hx * pl[(size_t)y1 * Win + x0] + lx * pl[(size_t)y1 * Win + x1]);
  out[(size_t)blockIdx.y * Hout * Wout + pix] = v;
}
}  // namespace

extern "C" int fr_resize_bilinear(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout,
                                  void* stream) {
  if (planes < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1 || planes > 65535)
    FR_UNSUPPORTED("fr_resize_bilinear: planes in 1..65535 and positive sizes");
  const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((Hout * Wout + 255) / 256, planes), dim3(256), 0, (hipStream_t)stream, in,
                     out, Hin, Win, Hout, Wout, sh, sw);
  FR_LAUNCH_CHECK();
}
