// frhip -- margin-head row kernels, focal loss on the batch-mean cross entropy, top-k rank, multi-tensor SGD.
//
// Reference arithmetic replaced (paths under /root/reference):
//   F.normalize(x), F.normalize(W) (eps 1e-12) + their autograd      head/metrics.py:103, :167
//   d phi / d cos of the ArcFace / CosFace margin, label select       head/metrics.py:115-138, :181-189
//   FocalLoss on mean CE                                              loss/focal.py:17-21
//   accuracy top-1/5                                                  util/utils.py:343-358
//   optim.SGD(momentum, coupled weight decay on group 0)              train.py:196, :313-316
#include "common.h"
#include "frhip_internal.h"

namespace {

// ------------------------------------------------------------------------------------------ row normalise
template <typename T>
__global__ void row_normalize_kernel(const float* __restrict__ x, T* __restrict__ xn, T* __restrict__ xt,
                                     float* __restrict__ inv, int rows, int rows_pad, int D, int ldt) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows_pad) return;
  if (row >= rows) {
    for (int d = lane; d < D; d += 64) {
      Elt<T>::st(xn + (size_t)row * D + d, 0.f);
      if (xt) Elt<T>::st(xt + (size_t)d * ldt + row, 0.f);
    }
    return;
  }
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float v = x[(size_t)row * D + d];
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  const float iv = 1.0f / fmaxf(nrm, 1e-12f);
  if (lane == 0) inv[row] = iv;
  for (int d = lane; d < D; d += 64) {
    const float v = x[(size_t)row * D + d] * iv;
    Elt<T>::st(xn + (size_t)row * D + d, v);
    if (xt) Elt<T>::st(xt + (size_t)d * ldt + row, v);
  }
}

// xt[d][r] = xn[r][d] through 64 x 64 LDS tiles (both sides coalesced).  The per-row kernel above writes the transposed
// copy as 4-byte stores with a stride of a whole row per lane; for the head weight (N x 512) this second pass is 3x cheaper.
template <typename T>
__global__ __launch_bounds__(256) void transpose_rows_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int D,
                                                             int ldt) {
  __shared__ T tt[64][66];
  const int d0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4)
    if (r0 + j < R && d0 + tx < D) tt[j][tx] = in[(size_t)(r0 + j) * D + d0 + tx];
  __syncthreads();
  for (int j = ty; j < 64; j += 4)
    if (d0 + j < D && r0 + tx < R) out[(size_t)(d0 + j) * ldt + r0 + tx] = tt[tx][j];
}

__global__ void normalize_bwd_kernel(const float* __restrict__ G, const float* __restrict__ x,
                                     const float* __restrict__ inv, float* __restrict__ gx, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float iv = inv[row];
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot = fmaf(x[(size_t)row * D + d] * iv, G[(size_t)row * D + d], dot);
  dot = wave_sum(dot);
  for (int d = lane; d < D; d += 64) {
    const float xh = x[(size_t)row * D + d] * iv;
    gx[(size_t)row * D + d] = (G[(size_t)row * D + d] - xh * dot) * iv;
  }
}

// ------------------------------------------------------------------------------------------ margin backward
template <typename T>
__global__ void margin_bwd_kernel(const float* __restrict__ g, const long long* __restrict__ label,
                                  const float* __restrict__ cos_t, T* __restrict__ gcos, int rows, int N, int ldg,
                                  int kind, int easy, float cos_m, float sin_m, float th, float scale) {
  const int m = blockIdx.y;
  if (m >= rows) return;
  const long long lab = label[m];
  float dphi = 1.f;
  if (kind == 0) {
    const float c = cos_t[m];
    const bool use_phi = easy ? (c > 0.f) : (c > th);
    if (use_phi) {
      const float t = 1.0f - c * c;
      const bool inside = t > 1e-10f && t < 1.0f - 1e-10f;  // clamp passes gradient only strictly inside
      dphi = cos_m + (inside ? sin_m * c / sqrtf(t) : 0.f);
    }
  }
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < ldg; n += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (n < N) v = scale * g[(size_t)m * N + n] * (n == lab ? dphi : 1.f);
    Elt<T>::st(gcos + (size_t)m * ldg + n, v);
  }
}

// ------------------------------------------------------------------------------------------ cross entropy rows
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ z, const long long* __restrict__ label,
                                                      float* __restrict__ lse, float* __restrict__ ce,
                                                      int* __restrict__ rank, int N, int ld) {
  __shared__ float sred[4];
  __shared__ int ired[4];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = z + (size_t)m * ld;
  const long long lab = label[m];
  const float zl = (lab >= 0 && lab < N) ? row[lab] : 0.f;
  float mx = -INFINITY;
  int cnt = 0;
  for (int n = tid; n < N; n += 256) {
    const float v = row[n];
    mx = fmaxf(mx, v);
    cnt += v > zl ? 1 : 0;
  }
  mx = wave_max(mx);
  if (lane == 0) sred[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  __syncthreads();
  float se = 0.f;
  for (int n = tid; n < N; n += 256) se += __expf(row[n] - mx);
  se = wave_sum(se);
  float fc = wave_sum((float)cnt);
  if (lane == 0) {
    sred[wave] = se;
    ired[wave] = (int)fc;
  }
  __syncthreads();
  if (tid == 0) {
    const float tot = sred[0] + sred[1] + sred[2] + sred[3];
    const float l = mx + logf(tot);
    lse[m] = l;
    ce[m] = l - zl;
    rank[m] = ired[0] + ired[1] + ired[2] + ired[3];
  }
}

__global__ void focal_finalize_kernel(const float* __restrict__ ce, const int* __restrict__ rank, int rows,
                                      float gamma, float* __restrict__ scalars) {
  __shared__ double dred[4];
  __shared__ int r1[4], r5[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  int c1 = 0, c5 = 0;
  for (int i = tid; i < rows; i += 256) {
    s += (double)ce[i];
    c1 += rank[i] < 1;
    c5 += rank[i] < 5;
  }
  s = wave_sum_d(s);
  const float f1 = wave_sum((float)c1), f5 = wave_sum((float)c5);
  if (lane == 0) {
    dred[wave] = s;
    r1[wave] = (int)f1;
    r5[wave] = (int)f5;
  }
  __syncthreads();
  if (tid == 0) {
    const float l = (float)((dred[0] + dred[1] + dred[2] + dred[3]) / rows);
    const float p = expf(-l);
    const float omp = 1.0f - p;
    const float w = powf(omp, gamma);
    scalars[0] = w * l;
    // d/dl [(1-p)^g * l] = g (1-p)^(g-1) p l + (1-p)^g          
    scalars[1] = gamma * powf(omp, gamma - 1.0f) * p * l + w;
    scalars[2] = 100.0f * (float)(r1[0] + r1[1] + r1[2] + r1[3]) / rows;
    scalars[3] = 100.0f * (float)(r5[0] + r5[1] + r5[2] + r5[3]) / rows;
    scalars[4] = l;
  }
}

__global__ void focal_bwd_kernel(const float* __restrict__ z, const long long* __restrict__ label,
                                 const float* __restrict__ lse, const float* __restrict__ scalars,
                                 const float* __restrict__ gup, float* __restrict__ grad, int rows, int N, int ld) {
  const int m = blockIdx.y;
  const float k = gup[0] * scalars[1] / (float)rows;
  const float l = lse[m];
  const long long lab = label[m];
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float sm = __expf(z[(size_t)m * ld + n] - l);
    grad[(size_t)m * ld + n] = k * (sm - (n == lab ? 1.f : 0.f));
  }
}

// rank-only (accuracy on arbitrary logits)
__global__ __launch_bounds__(256) void rank_rows_kernel(const float* __restrict__ z, const long long* __restrict__ label,
                                                        int* __restrict__ rank, int N, int ld) {
  __shared__ int ired[4];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = z + (size_t)m * ld;
  const long long lab = label[m];
  const float zl = (lab >= 0 && lab < N) ? row[lab] : INFINITY;
  int cnt = 0;
  for (int n = tid; n < N; n += 256) cnt += row[n] > zl ? 1 : 0;
  const float fc = wave_sum((float)cnt);
  if (lane == 0) ired[wave] = (int)fc;
  __syncthreads();
  if (tid == 0) rank[m] = ired[0] + ired[1] + ired[2] + ired[3];
}

// out[j] = scale * #{m: rank[m] < k_j}: precision@k of util/utils.py:343-358 from the label ranks, one launch instead of
// four small torch kernels per k
__global__ __launch_bounds__(256) void topk_precision_kernel(const int* __restrict__ rank, int rows, int nk, int4 ks,
                                                             float scale, float* __restrict__ out) {
  __shared__ int ired[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk[4] = {ks.x, ks.y, ks.z, ks.w};
  int cnt[4] = {0, 0, 0, 0};
  for (int m = tid; m < rows; m += 256) {
    const int r = rank[m];
#pragma unroll
    for (int j = 0; j < 4; ++j) cnt[j] += (j < nk && r < kk[j]) ? 1 : 0;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float fc = wave_sum((float)cnt[j]);  // exact: counts below 2^24
    if (lane == 0) ired[j][wave] = (int)fc;
  }
  __syncthreads();
  if (tid < nk) out[tid] = (float)(ired[tid][0] + ired[tid][1] + ired[tid][2] + ired[tid][3]) * scale;
}

// ------------------------------------------------------------------------------------------ class-sharded softmax
// One rank holds the logits of its own contiguous class range only.  Per row it contributes (max, sum exp(z - max),
// label logit or 0); the ranks' triples are gathered and combined in rank order (shard_combine_kernel), which gives every
// rank the same log-sum-exp / cross entropy bit for bit.
__global__ __launch_bounds__(256) void shard_row_stats_kernel(const float* __restrict__ z,
                                                              const long long* __restrict__ label,
                                                              float* __restrict__ stats, int rows, int N, int ld) {
  __shared__ float sred[4];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = z + (size_t)m * ld;
  float mx = -INFINITY;
  for (int n = tid; n < N; n += 256) mx = fmaxf(mx, row[n]);
  mx = wave_max(mx);
  if (lane == 0) sred[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  __syncthreads();
  float se = 0.f;
  for (int n = tid; n < N; n += 256) se += __expf(row[n] - mx);
  se = wave_sum(se);
  if (lane == 0) sred[wave] = se;
  __syncthreads();
  if (tid == 0) {
    const long long lab = label[m];
    stats[m] = mx;
    stats[rows + m] = sred[0] + sred[1] + sred[2] + sred[3];
    stats[2 * rows + m] = (lab >= 0 && lab < N) ? row[lab] : 0.f;
  }
}

__global__ void shard_combine_kernel(const float* __restrict__ stats_all, int world, int rows, float* __restrict__ lse,
                                     float* __restrict__ ce, float* __restrict__ tlogit) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= rows) return;
  float gmax = -INFINITY;
  for (int w = 0; w < world; ++w) gmax = fmaxf(gmax, stats_all[(size_t)w * 3 * rows + m]);
  float sum = 0.f, t = 0.f;
  for (int w = 0; w < world; ++w) {
    const float* s = stats_all + (size_t)w * 3 * rows;
    sum += s[rows + m] * expf(s[m] - gmax);
    t += s[2 * rows + m];  // exactly one rank owns the label, the others contribute 0
  }
  const float l = gmax + logf(sum);
  lse[m] = l;
  ce[m] = l - t;
  tlogit[m] = t;
}

// rank[m] = #{n in this shard: z[m][n] > t[m]}; the ranks' counts add up to the global rank of the label
__global__ __launch_bounds__(256) void shard_rank_rows_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                              int* __restrict__ rank, int N, int ld) {
  __shared__ int ired[4];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = z + (size_t)m * ld;
  const float zl = t[m];
  int cnt = 0;
  for (int n = tid; n < N; n += 256) cnt += row[n] > zl ? 1 : 0;
  const float fc = wave_sum((float)cnt);
  if (lane == 0) ired[wave] = (int)fc;
  __syncthreads();
  if (tid == 0) rank[m] = ired[0] + ired[1] + ired[2] + ired[3];
}

// ------------------------------------------------------------------------------------------ SGD
constexpr int SGD_CHUNK = 4096;
__global__ __launch_bounds__(256) void sgd_kernel(const FrSgdTensor* __restrict__ table,
                                                  const int2* __restrict__ chunks, float lr, float momentum) {
  const int2 ch = chunks[blockIdx.x];
  const FrSgdTensor t = table[ch.x];
  const long long base = (long long)ch.y * SGD_CHUNK;
  long long end = base + SGD_CHUNK;
  if (end > t.n) end = t.n;
  for (long long i = base + threadIdx.x; i < end; i += 256) {
    const float pv = t.p[i];
    const float d = fmaf(t.wd, pv, t.g[i]);
    const float b = fmaf(momentum, t.buf[i], d);
    t.buf[i] = b;
    t.p[i] = pv - lr * b;
  }
}

// ------------------------------------------------------------------------------------------ Adam
// Operation order of torch.optim.Adam's single-tensor path (lerp, mul + addcmul, sqrt / sqrt(bc2) + eps, addcdiv); the
// explicit _rn intrinsics keep the compiler from contracting them into different roundings.
__global__ __launch_bounds__(256) void adam_kernel(const FrAdamTensor* __restrict__ table,
                                                   const int2* __restrict__ chunks, float step_size, float w1,
                                                   float beta2, float w2, float eps, float bc2_sqrt) {
  const int2 ch = chunks[blockIdx.x];
  const FrAdamTensor t = table[ch.x];
  const long long base = (long long)ch.y * SGD_CHUNK;
  long long end = base + SGD_CHUNK;
  if (end > t.n) end = t.n;
  const float neg_step = -step_size;
  for (long long i = base + threadIdx.x; i < end; i += 256) {
    const float g = t.g[i];
    const float m = __fadd_rn(t.m[i], __fmul_rn(w1, __fsub_rn(g, t.m[i])));
    const float v = __fadd_rn(__fmul_rn(t.v[i], beta2), __fmul_rn(__fmul_rn(w2, g), g));
    t.m[i] = m;
    t.v[i] = v;
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
    t.p[i] = __fadd_rn(t.p[i], __fdiv_rn(__fmul_rn(neg_step, m), denom));  // addcdiv: self + (value * t1) / t2
  }
}

}  // namespace

extern "C" int fr_row_normalize(const float* x, void* xn, void* xt, float* inv, int rows, int rows_pad, int D,
                                int ldt, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int grid = (rows_pad + 3) / 4;
  const bool tiled = xt && rows_pad >= 256;  // transposed copy by a second, coalesced pass
  const dim3 tgrid((D + 63) / 64, (rows_pad + 63) / 64);
  if (dtype == FR_F32) {
    hipLaunchKernelGGL(row_normalize_kernel<float>, dim3(grid), dim3(256), 0, st, x, (float*)xn,
                       tiled ? (float*)nullptr : (float*)xt, inv, rows, rows_pad, D, ldt);
    if (tiled)
      hipLaunchKernelGGL(transpose_rows_kernel<float>, tgrid, dim3(256), 0, st, (const float*)xn, (float*)xt, rows_pad, D,
                         ldt);
  } else if (dtype == FR_BF16) {
    hipLaunchKernelGGL(row_normalize_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, x, (bf16_t*)xn,
                       tiled ? (bf16_t*)nullptr : (bf16_t*)xt, inv, rows, rows_pad, D, ldt);
    if (tiled)
      hipLaunchKernelGGL(transpose_rows_kernel<bf16_t>, tgrid, dim3(256), 0, st, (const bf16_t*)xn, (bf16_t*)xt, rows_pad,
                         D, ldt);
  } else {
    FR_UNSUPPORTED("fr_row_normalize: dtype");
  }
  FR_LAUNCH_CHECK();
}

extern "C" int fr_normalize_bwd(const float* G, const float* x, const float* inv, float* gx, int rows, int D,
                                void* stream) {
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, G, x, inv, gx,
                     rows, D);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_margin_bwd(const float* g, const int64_t* label, const float* cos_t, void* gcos, int rows, int N,
                             int ldg, int kind, int easy, float cos_m, float sin_m, float th, float scale, int dtype,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((ldg + 1023) / 1024, rows);
  if (dtype == FR_F32)
    hipLaunchKernelGGL(margin_bwd_kernel<float>, grid, dim3(256), 0, st, g, (const long long*)label, cos_t,
                       (float*)gcos, rows, N, ldg, kind, easy, cos_m, sin_m, th, scale);
  else if (dtype == FR_BF16)
    hipLaunchKernelGGL(margin_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, g, (const long long*)label, cos_t,
                       (bf16_t*)gcos, rows, N, ldg, kind, easy, cos_m, sin_m, th, scale);
  else
    FR_UNSUPPORTED("fr_margin_bwd: dtype");
  FR_LAUNCH_CHECK();
}

extern "C" int fr_ce_rows(const float* logits, const int64_t* label, float* lse, float* ce, int32_t* rank, int rows,
                          int N, int ld, void* stream) {
  hipLaunchKernelGGL(ce_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)label,
                     lse, ce, rank, N, ld);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_rank_rows(const float* logits, const int64_t* label, int32_t* rank, int rows, int N, int ld,
                            void* stream) {
  hipLaunchKernelGGL(rank_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                     (const long long*)label, rank, N, ld);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_topk_precision(const int32_t* rank, int rows, int nk, int k0, int k1, int k2, int k3, float scale,
                                 float* out, void* stream) {
  if (nk < 1 || nk > 4 || rows < 0) FR_UNSUPPORTED("fr_topk_precision: 1..4 values of k");
  hipLaunchKernelGGL(topk_precision_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rank, rows, nk,
                     make_int4(k0, k1, k2, k3), scale, out);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_focal_finalize(const float* ce, const int32_t* rank, int rows, float gamma, float* scalars,
                                 void* stream) {
  hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ce, rank, rows, gamma,
                     scalars);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_focal_bwd(const float* logits, const int64_t* label, const float* lse, const float* scalars,
                            const float* gup, float* grad, int rows, int N, int ld, void* stream) {
  dim3 grid((N + 1023) / 1024, rows);
  hipLaunchKernelGGL(focal_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, (const long long*)label,
                     lse, scalars, gup, grad, rows, N, ld);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_shard_row_stats(const float* logits, const int64_t* label_local, float* stats, int rows, int N, int ld,
                                  void* stream) {
  if (rows <= 0 || N <= 0) FR_UNSUPPORTED("fr_shard_row_stats: empty shard");
  hipLaunchKernelGGL(shard_row_stats_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                     (const long long*)label_local, stats, rows, N, ld);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_shard_combine(const float* stats_all, int world, int rows, float* lse, float* ce, float* tlogit,
                                void* stream) {
  if (rows <= 0 || world <= 0) FR_UNSUPPORTED("fr_shard_combine: empty");
  hipLaunchKernelGGL(shard_combine_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats_all,
                     world, rows, lse, ce, tlogit);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_shard_rank_rows(const float* logits, const float* tlogit, int32_t* rank, int rows, int N, int ld,
                                  void* stream) {
  if (rows <= 0 || N <= 0) FR_UNSUPPORTED("fr_shard_rank_rows: empty shard");
  hipLaunchKernelGGL(shard_rank_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, tlogit, rank, N,
                     ld);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_adam_step(const FrAdamTensor* table_dev, const int32_t* chunks_dev, int nchunks, float step_size,
                            float w1, float beta2, float w2, float eps, float bc2_sqrt, void* stream) {
  if (nchunks <= 0) return 0;
  hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table_dev, (const int2*)chunks_dev,
                     step_size, w1, beta2, w2, eps, bc2_sqrt);
  FR_LAUNCH_CHECK();
}

extern "C" int fr_sgd_chunk_elems(void) { return SGD_CHUNK; }

extern "C" int fr_sgd_step(const FrSgdTensor* table_dev, const int32_t* chunks_dev, int nchunks, float lr,
                           float momentum, void* stream) {
  if (nchunks <= 0) return 0;
  hipLaunchKernelGGL(sgd_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table_dev,
                     (const int2*)chunks_dev, lr, momentum);
  FR_LAUNCH_CHECK();
}
