/* frhip -- C ABI of the MI355X (gfx950) kernels behind the Stage-3 face-recognition training step.
 *
 * Drop-in boundary.  The reference has no FFI for this path: it reaches its arithmetic
 * through torch.nn modules (cuDNN/cuBLAS underneath).  Every entry point below therefore cites the
 * reference call site (file:line under /root/reference) whose arithmetic it replaces.  The only native
 * boundary the reference does have (backbone/stylegan2/op/fused_bias_act.cpp:11-21, upfirdn2d.cpp:12-22)
 * sets the conventions copied here: contiguous device buffers, work enqueued on the caller's stream, no
 * host synchronisation, autograd kept on the Python side.
 *
 * Conventions
 *   - every function returns int: 0 ok, <0 unsupported argument (see fr_last_error_string), >0 hipError_t
 *   - pointers are device pointers borrowed for the duration of the enqueue; `stream` is a hipStream_t
 *   - no allocation, no synchronisation, no global mutable state besides the per-thread error string
 *   - activations are NHWC ("pixels x channels") in the compute dtype (FR_F32 or FR_BF16); statistics,
 *     gradients of parameters and optimizer state are always fp32; labels are int64
 */
#ifndef FRHIP_H
#define FRHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_ABI_VERSION 7

enum { FR_F32 = 0, FR_BF16 = 1 };

/* prologue applied to the gathered A operand */
enum {
  FR_PRO_NONE = 0,
  FR_PRO_BN = 1,     /* x*a[c]+b[c] */
  FR_PRO_PRELU = 2,  /* x>0?x:a[c]*x */
  /* 3: FR_PRO_BNBWD2 of ABI v4 (BatchNorm backward inside the data gradient; measured +-0 on two streams, removed in v5; rebuilt
   *    and measured again in round 6 on the fragment-order kernels: +0.1 ms per step, profiles/r06_ab_bn2_in_dgrad.txt) */
  FR_PRO_RESBN = 4   /* BN1 of a residual unit applied to the OUTPUT of the unit in front of it, formed on the way:
                        o = round(a[c]*x + b[c] + x2)  (x = conv2 output of the previous unit, a / b = its BN2 coefficients, x2 =
                        src2 = that unit's input: bottleneck_IR's `res + shortcut`, backbone/model_irse.py:64-66 with the
                        MaxPool2d(1, 1) shortcut), operand = c[c]*o + d[c] (pro_c / pro_d: this unit's BN1).  o is stored to
                        pro_out (interior pixels, once each): the residual stream is materialised by its consumer and the
                        BN-apply pass behind conv2 is gone.  The batch statistics of o come from fr_bn_finalize_res.
                        Served by fr_conv3x3_strip forward launches (bf16). */
  ,
  FR_PRO_RESBN_SE = 5 /* FR_PRO_RESBN behind a squeeze-excite unit (bottleneck_IR_SE, backbone/model_irse.py:84-91):
                         o = round((a[c]*x + b[c]) * g[image][c] + x2) with the excite gates pro_g = [B][SC] fp32.  Instances
                         whose workgroups hold strips of ONE image. */
};

/* epilogue of fr_conv_igemm */
enum {
  FR_EPI_STORE = 0,     /* out = acc (+bias) */
  FR_EPI_STATS = 1,     /* + part[mtile][0][n] = sum_rows out, part[mtile][1][n] = sum_rows out^2 */
  FR_EPI_PRELU_BWD = 2, /* out = acc * (aux>0 ? 1 : epi_a[n]); part[mtile][0][n] = sum acc*aux*[aux<=0] */
  FR_EPI_BNBWD = 3,     /* out = acc; part[.][0] = sum acc; part[.][1] = sum acc*(aux-epi_a[n])*epi_b[n] */
  FR_EPI_MARGIN = 4,    /* out = scale*(n==label[m] ? phi(acc) : acc); cos_t[m] = acc[label] */
  FR_EPI_ATOMIC = 5,    /* atomicAdd(out, acc) fp32; used with splitk > 1 (summation order not reproducible) */
  FR_EPI_BIAS_RES = 7,  /* out = acc + epi_a[n] + epi_b[n] + aux[m][n]: inference with BatchNorm folded into the weights
                           (util/utils.py:254-307 runs the backbone in eval mode): epi_a / epi_b = the folded BN shifts of
                           the residual branch and of the conv shortcut (both required; pass zeros), aux = the shortcut */
  FR_EPI_STATS_X = 8,   /* FR_EPI_STATS + part[.][2][n] = sum_rows out*aux: the cross moment with the residual input (aux), from
                           which fr_bn_finalize_res derives the statistics of a*out + b + aux; part rows are [3][N] */
  FR_EPI_SLAB = 6       /* split-K slice z stores its fp32 partial (+bias in slice 0) to out[z][rows][ldc]: the caller
                           adds the splitk slabs with fr_reduce_parts(out, splitk, 1, rows*ldc, ...) -- reproducible */
};

/* The BatchNorm arguments of fr_bn_finalize as a struct (fr_bn_finalize_res takes two of them). */
typedef struct FrBnFinArgs {
  double count;            /* elements per channel */
  const float* gamma;
  const float* beta;
  float eps, momentum;
  float* running_mean;     /* NULL: no running statistics */
  float* running_var;
  int64_t* nbt;
  float* mean;             /* outputs [C] */
  float* invstd;
  float* scale;
  float* shift;
} FrBnFinArgs;

typedef struct FrConvArgs {
  const void* src; /* A operand: NHWC [B,SH,SW,SC], pixel stride lda (elements) */
  const void* w;   /* B operand: [N][KH*KW][SC] in the compute dtype */
  void* out;       /* [B*RH*RW][ldc], compute dtype or fp32 (out_f32) */
  int32_t B, RH, RW;   /* row space: one GEMM row per (b, rh, rw) */
  int32_t SH, SW, SC;  /* source tensor geometry; SC % 32 == 0 */
  int32_t N;           /* output columns */
  int32_t KH, KW, stride, pad;
  int32_t mode;        /* 0: sh = rh*stride+kh-pad (forward); 1: sh = (rh+pad-kh)/stride (data gradient);
                          2: stride-2 3x3 data gradient for the output pixels (2i+par_h, 2j+par_w) only: rows = B*RH/2*RW/2,
                             the taps that hit no input pixel are skipped; run once per parity class */
  int32_t lda, ldc, ldaux;
  int32_t pro;         /* FR_PRO_* */
  int32_t epi;         /* FR_EPI_* */
  int32_t out_f32;
  int32_t splitk;      /* >1: K loop split over gridDim.z, requires FR_EPI_ATOMIC + out_f32 */
  int32_t stride_log2; /* filled in by the library */
  int32_t par_h, par_w; /* mode 2: parity class of the output pixels; (-1, -1) = all four classes in one launch,
                           part rows ordered [class = 2*par_h + par_w][M tile] */
  int32_t margin_kind; /* 0 ArcFace, 1 CosFace */
  int32_t easy_margin;
  float cos_m, sin_m, th, mm, scale;
  const float* pro_a; /* [SC] */
  const float* pro_b; /* [SC] */
  const float* bias;  /* [N] or NULL */
  const float* epi_a; /* [N] */
  const float* epi_b; /* [N] */
  const void* aux;    /* [rows][ldaux] compute dtype */
  float* part;        /* [ceil(rows/128)][2][N] partial column sums */
  const int64_t* label; /* [rows] */
  float* cos_t;         /* [rows] */
  /* the two-source prologues FR_PRO_RESBN / FR_PRO_RESBN_SE */
  const void* src2;    /* second source, geometry and strides of src */
  const float* pro_c;  /* [SC] */
  void* pro_out;       /* NULL or [B*SH*SW][lda]: the residual sum of every source pixel */
  const float* pro_d;  /* [SC], FR_PRO_RESBN / FR_PRO_RESBN_SE only */
  const float* pro_g;  /* [B][SC] excite gates, FR_PRO_RESBN_SE only */
  /* ABI v7: 1 = w is in MFMA-fragment order, [N / 16][taps][K / 32][64][8] with (n, tap, k) at
   * (((n / 16 * taps + tap) * (K / 32) + k / 32) * 64 + (k % 32) / 8 * 16 + n % 16) * 8 + k % 8  (N = output columns, K = SC;
   * FrPackTensor.frag writes it): the 64 lanes of a weight-fragment load then read 1024 contiguous bytes.  Taken by
   * fr_conv3x3_strip on its LDS-strip instances (fr_conv3x3_strip_takes_frag) and by fr_conv3x3_s2_strip; refused elsewhere. */
  int32_t w_frag;
  int32_t reserved_;
} FrConvArgs;

/* Convolution forward / data gradient / dense GEMM on MFMA.
 * Replaces: Conv2d 3x3 s1/s2 + neighbours in bottleneck_IR (backbone/model_irse.py:56-60), stem conv
 * (:140, through fr_stem_im2col), shortcut conv1x1 (:55), Linear(25088,512) (:147), and
 * F.linear(F.normalize(x), F.normalize(W)) + margin blend of ArcFace/CosFace (head/metrics.py:103,115-138,
 * :167,181-189); their autograd data-gradients with mode = 1. */
int fr_conv_igemm(const FrConvArgs* args, int dtype, void* stream);

/* Stride-1 3x3 convolution (bf16 only) with the input strip resident in LDS: same FrConvArgs contract as
 * fr_conv_igemm (mode 1 = data gradient: mirrored taps, w = [Cin][tap][Cout]); epilogues STORE / STATS / PRELU_BWD /
 * BNBWD.  Partial rows go to part[workgroup][2][N]; fr_conv3x3_strip_parts returns the number of workgroups for a
 * shape and epilogue, or 0 when that combination is not served (use fr_conv_igemm then).
 * Replaces Conv2d(c, d, (3,3), (1,1), 1) of bottleneck_IR (backbone/model_irse.py:57-59) fwd + data gradient. */
int fr_conv3x3_strip(const FrConvArgs* args, void* stream);
int fr_conv3x3_strip_parts(int B, int Cin, int Cout, int W, int epi);
/* 1 when fr_conv3x3_strip serves the two-source prologues (FR_PRO_RESBN[_SE]) and FR_EPI_STATS_X for a C -> C layer of width W
 * at batch B (the LDS-strip instances; not the 64-channel rolling-window kernel) */
int fr_conv3x3_strip_serves_resbn(int B, int C, int W);
/* 1 when fr_conv3x3_strip reads fragment-order weights (FrConvArgs.w_frag) for a Cin -> Cout layer of width W at batch B: the
 * shape is served and not by the 64-channel rolling-window kernel, which keeps its weights resident and takes the plain layout */
int fr_conv3x3_strip_takes_frag(int B, int Cin, int Cout, int W);

/* 1x1 convolution as a row-streaming GEMM (bf16; round 4): the weights stationary in registers, one row per output pixel,
 * stride 1 or 2 (SH = RH * stride), epilogue STORE or STATS (part[workgroup][2][N]; fr_conv1x1_stream_parts returns the number
 * of partial rows, 0 when the shape is not served: 64 -> 128, 128 -> 256, 256 -> 512 and their transposes).  Same FrConvArgs
 * fields as fr_conv_igemm (mode 0, no prologue).  Replaces shortcut_layer's Conv2d(in, depth, (1, 1), stride) of the first unit
 * of a stage (backbone/model_irse.py:52-56) forward and, with the transposed weight on dense rows, its data gradient. */
int fr_conv1x1_stream(const FrConvArgs* args, void* stream);
int fr_conv1x1_stream_parts(int B, int RH, int RW, int K, int N);

/* Stride-2 3x3 convolution (bf16, Cin == Cout: the first unit of every IR stage) on LDS-resident parity planes:
 * mode 0 = forward (SH = 2*RH), mode 2 with par_h = par_w = -1 = data gradient of all four output parity classes
 * (RH = 2*SH, w = [Cin][tap][Cout]).  Same FrConvArgs contract and epilogues as fr_conv_igemm; partial rows:
 * forward part[workgroup][2][N], gradient part[class][workgroup][2][N]; forward rows are image-major (all rows of
 * image b before those of image b + 1), the same number per image.  fr_conv3x3_s2_strip_parts returns the number of partial
 * rows for (B, channels, low-res width WL, mode), 0 when the shape is not served -- callers size and sum `part` with
 * that number and nothing else: the 64-channel layer (112 -> 56) runs on a persistent rolling-window kernel that
 * writes ONE row per work item (image x row segment) in both directions.
 * Replaces Conv2d(d, d, (3,3), stride 2, 1) of bottleneck_IR (backbone/model_irse.py:59) fwd + data gradient. */
int fr_conv3x3_s2_strip(const FrConvArgs* args, void* stream);
int fr_conv3x3_s2_strip_parts(int B, int Cin, int Cout, int WL, int mode);
/* 1 when fr_conv3x3_s2_strip reads fragment-order weights (FrConvArgs.w_frag) for (B, C -> C, low-res width WL, mode): every
 * served shape except the 64-channel layer (rolling-window kernel, weights resident, plain layout) */
int fr_conv3x3_s2_strip_takes_frag(int B, int C, int WL, int mode);

typedef struct FrWgradArgs {
  const void* g;   /* gradient of the conv output: [B*GH*GW][ldg], columns = Cout */
  const void* src; /* conv input, NHWC [B,SH,SW,SC], pixel stride lda */
  float* dw;       /* [Cout][KH*KW][SC] fp32, accumulated with atomicAdd: caller zeroes it first */
  int32_t B, GH, GW, Cout;
  int32_t SH, SW, SC;
  int32_t KH, KW, stride, pad;
  int32_t ldg, lda;
  int32_t pro; /* FR_PRO_* applied to src */
  int32_t nsplit; /* pixel slices (gridDim.y) / strip groups */
  const float* pro_a;
  const float* pro_b;
  float* slab;     /* [nsplit][Cout][taps][SC] fp32 partial gradients: required by fr_conv_wgrad_strip; optional for
                      fr_conv_wgrad (NULL: pixel slices are combined with fp32 atomics onto a zeroed dw; non-NULL: each
                      slice stores its slab and a second launch adds them in a fixed order -- reproducible) */
  /* Deferred slab sum (fr_conv_wgrad_strip only; fr_conv_wgrad ignores these fields).  defer != 0:
   * this launch only writes its slabs; the caller owes them a sum -- either as the prev_* of a later deferring launch
   * on the same stream (whose workgroups add them while their first tiles are in flight: no launch of their own) or
   * through fr_reduce_slabs.  prev_n != 0: before its own work the launch writes
   * prev_dw[i] = sum_{g < prev_groups} prev_slab[g * prev_n + i] in the library's fixed order (bit-identical to
   * fr_reduce_slabs).  prev_slab must not be this launch's slab. */
  const float* prev_slab;
  float* prev_dw;
  int64_t prev_n;
  int32_t prev_groups;
  int32_t defer;
} FrWgradArgs;

/* Weight gradient  dw[co][tap][ci] += sum_p g[p][co] * pro(src[pixel(p,tap)][ci]).
 * Replaces the autograd weight-gradient of every Conv2d / Linear above. */
int fr_conv_wgrad(const FrWgradArgs* args, int dtype, void* stream);

/* Weight gradient of 3x3 convolutions (bf16) with both operand strips resident in LDS; partial results per strip
 * group go to `slab`, a second launch adds them into dw (overwrites; deterministic, no atomics).  Stride 1 (GH == SH):
 * fr_conv_wgrad_strip_supported tells whether a shape is served (else use fr_conv_wgrad).  Stride 2 (SH == 2*GH,
 * GW in {56, 28, 14, 7}, channels multiples of 64): the input tile holds the four parity planes of the strip. */
int fr_conv_wgrad_strip(const FrWgradArgs* args, void* stream);
int fr_conv_wgrad_strip_supported(int Cout, int Cin, int W);
/* 1 when fr_conv_wgrad_strip serves these arguments (every served shape honours defer / prev_*; 0 also when the
 * deferral is switched off with FRHIP_WGRAD_DEFER=0).  fr_reduce_slabs: out[i] = sum over the slabs g = 0 .. groups-1
 * of slab[g*n + i] in the library's fixed order (chunks of 16 slabs, csrc/slab_sum.h; n % 4 == 0, groups <= 256): the
 * sum a deferring launch left to its caller. */
int fr_conv_wgrad_strip_defers(const FrWgradArgs* args);
int fr_reduce_slabs(const float* slab, int groups, long long n, float* out, void* stream);

/* ---- stem: NCHW fp32 images (+ optional constant avg image, restyle_psp.py:445-447) -> im2col rows
 * out[(b,h,w)][(kh*3+kw)*C + c] in the compute dtype, K padded with zeros to ldk (32 or 64).
 * Replaces the unfold half of input_layer Conv2d(3|6,64,3,1,1) (model_irse.py:140, restyle_psp.py:137). */
int fr_stem_im2col(const float* x, const float* avg, void* out, int B, int H, int W, int C, int Cavg, int ldk,
                   int dtype, void* stream);

/* The stem GEMMs over those rows (bf16, K = ldk = 32 or 64, 64 output channels), shaped for 3.2 M rows x 64 columns:
 * fr_stem_gemm : out[M][64] = X[M][K] * Wp[64][K]^T, part[nblocks][2][64] = column sums / sums of squares of out
 *                (the EPI_STATS contract of fr_conv_igemm with nblocks partial rows);
 * fr_stem_wgrad: slab[nblocks][64][K] = per-workgroup partial of g[M][64]^T * X[M][K] (add them with fr_reduce_parts,
 *                K = 1, C = 64*K).
 * Replace the GEMM half of input_layer Conv2d(3|6,64,3,1,1) (model_irse.py:140) forward and its weight gradient. */
int fr_stem_gemm(const void* X, const void* Wp, void* out, float* part, long long M, int K, int nblocks, void* stream);
/* Round 4: the stem forward as TWO passes over the rows instead of GEMM + a BN-apply pass over its 411-MB output:
 * fr_stem_gemm(out = NULL) leaves only the statistics of y = X * Wp^T (nothing stored), and after fr_bn_finalize
 * fr_stem_gemm_bn_prelu recomputes y, stores z = PReLU(BN(y)) -- fr_bn_apply(slope) on the rounded y, element for element -- and
 * y itself (y may be NULL when nothing reads it) and leaves the statistics of the rounded z in part (for the BatchNorm of the
 * first residual unit).  Replaces input_layer = Conv2d -> BatchNorm2d -> PReLU (backbone/model_irse.py:140-142). */
int fr_stem_gemm_bn_prelu(const void* X, const void* Wp, const float* scale, const float* shift, const float* slope, void* y,
                          void* z, float* part, long long M, int K, int nblocks, void* stream);
/* ... and its backward WITHOUT the stored y (y = NULL above: 411 MB less memory and traffic at batch 256): both kernels
 * recompute y = X * Wp^T from the rows, rounded as the forward pass rounded it.
 * fr_stem_bwd_sums  : part[nblocks][3][64] = the rows of fr_bn_bwd_reduce(slope) over (G, y) -- sum g', sum g'*xhat, sum g*u*[u<=0];
 * fr_stem_wgrad_bn_r: fr_stem_wgrad_bn with y recomputed per 64-row trip (bit-identical slabs).
 * Replace the autograd of Conv2d -> BatchNorm2d -> PReLU of input_layer (backbone/model_irse.py:140-142). */
int fr_stem_bwd_sums(const void* X, const void* Wp, const void* G, const float* mean, const float* invstd, const float* scale,
                     const float* shift, const float* slope, float* part, long long M, int K, int nblocks, void* stream);
int fr_stem_wgrad_bn_r(const void* G, const void* X, const void* Wp, const float* mean, const float* invstd,
                       const float* scale, const float* shift, const float* slope, const float* gamma, const float* s0,
                       const float* s1, float inv_count, float* slab, long long M, int K, int nblocks, void* stream);
int fr_stem_wgrad(const void* G, const void* X, float* slab, long long M, int K, int nblocks, void* stream);
/* As fr_stem_wgrad, with the backward of BatchNorm2d(64) -> PReLU(64) (model_irse.py:141-142) applied to the rows of G
 * while they are staged: G = gradient at the PReLU output, Y = BN input (the stem GEMM output), s0/s1 = the reduced
 * sums of fr_bn_bwd_reduce; equals fr_bn_bwd_apply followed by fr_stem_wgrad bit for bit without the 3-pass round trip. */
int fr_stem_wgrad_bn(const void* G, const void* Y, const void* X, const float* mean, const float* invstd,
                     const float* scale, const float* shift, const float* slope, const float* gamma, const float* s0,
                     const float* s1, float inv_count, float* slab, long long M, int K, int nblocks, void* stream);

/* ---- BatchNorm statistics (train mode; torch defaults eps 1e-5, momentum 0.1 -- SURVEY App. B 13)
 * part: [nparts][2][C] partial (sum, sum of squares) rows; count = elements per channel.
 * Writes mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; updates running stats (unbiased var)
 * and num_batches_tracked when those pointers are non-NULL.  Replaces nn.BatchNorm2d/1d statistics
 * (model_irse.py:57,60,141,144,148). */
int fr_bn_finalize(const float* part, int nparts, int C, double count, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var, int64_t* nbt,
                   float* mean, float* invstd, float* scale, float* shift, void* stream);

/* Statistics of a residual sum without a pass over it (round 4): part = [nparts][3][C] rows of a FR_EPI_STATS_X launch
 * (sum y, sum y^2, sum y*o over the conv2 output y and the unit's input o).  One launch finalises `bn` (the BatchNorm behind
 * conv2: exactly what fr_bn_finalize(part rows 0..1) writes; bn->count = elements per channel) AND `next`, the BatchNorm that
 * normalises the unit's output o' = scale*y + shift + o:  mean' = scale*my + shift + mo,  var' = scale^2*vy + vo +
 * 2*scale*cov(y, o), with (mo, vo) from in_mean / in_invstd / in_eps -- the statistics the unit's own BN1 used for o.
 * next == NULL: only `bn` is finalised (rows of three vectors; squeeze-excite units, whose output statistics need the
 * gates: fr_se_pool_parts_mlp_fwd_res).
 * Replaces the statistics pass behind `res + shortcut` of bottleneck_IR (backbone/model_irse.py:64-66) for identity units. */
int fr_bn_finalize_res(const float* part, int nparts, int C, const FrBnFinArgs* bn, const float* in_mean,
                       const float* in_invstd, float in_eps, const FrBnFinArgs* next, void* stream);

/* per-channel (sum, sumsq) partials of an NHWC tensor: part[blk][2][C], blk < nblocks (= grid size) */
int fr_channel_stats(const void* x, long long rows, int C, float* part, int nblocks, int dtype, void* stream);

/* out = [prelu]( x*scale+shift [* se[b][c]] ) [+ res]  with (sum,sumsq) partials of `out` for the next BN.
 *   res_kind 0 none | 1 identity shortcut x_in[b, h*stride, w*stride, c] (MaxPool2d(1,s), model_irse.py:53)
 *            | 2 BN'd conv shortcut: res*rscale+rshift (model_irse.py:55-56)
 * Replaces BN apply + PReLU of the stem (:141-142) and BN + SE excite + residual add of a unit (:60-66). */
typedef struct FrApplyArgs {
  const void* x;       /* [B*H*W][C] */
  void* out;           /* [B*H*W][C] */
  const float* scale;  /* [C] */
  const float* shift;  /* [C] */
  const float* slope;  /* PReLU [C] or NULL */
  const float* se;     /* [B][C] or NULL */
  const void* res;     /* residual source or NULL */
  const float* rscale; /* [C] for res_kind 2 */
  const float* rshift;
  float* part;         /* [nblocks][2][C] or NULL */
  int32_t B, H, W, C;
  int32_t res_kind, res_stride; /* identity: res has geometry [B, H*res_stride, W*res_stride, C] */
  int32_t nblocks;     /* grid size == number of partial rows */
} FrApplyArgs;
int fr_bn_apply(const FrApplyArgs* args, int dtype, void* stream);

/* ---- BatchNorm backward.  g' is the gradient at the BN output:
 *   g' = g                                    plain
 *      = g * prelu'(u),  u = x*scale+shift    slope != NULL  (stem: BN -> PReLU, model_irse.py:141-142)
 *      = g * se[b][c] + gse[b][c]             se != NULL     (IR-SE: BN -> SE excite, model_irse.py:86-87)
 * fr_bn_bwd_reduce: part[blk][3][C]: k=0 sum g', k=1 sum g'*xhat, k=2 sum g*u*[u<=0] (PReLU slope gradient)
 * fr_bn_bwd_apply : gx = gamma*invstd*(g' - s0/n - xhat*s1/n) [+ add]
 *   add_kind 0 none | 1 tensor of the same geometry (conv-shortcut data gradient)
 *            | 2 identity shortcut MaxPool2d(1,s): add[b, h/s, w/s, c] where h%s==0 and w%s==0 */
typedef struct FrBnBwdArgs {
  const void* g;       /* upstream gradient [rows][C] */
  const void* x;       /* BN input [rows][C] */
  void* gx;            /* apply: output gradient [rows][C] */
  const float* mean;
  const float* invstd;
  const float* scale;  /* gamma*invstd, shift: only read when slope != NULL */
  const float* shift;
  const float* slope;
  const float* se;     /* [B][C] */
  const float* gse;    /* [B][C] gradient wrt the pooled BN output, already divided by H*W */
  const float* gamma;  /* apply */
  const float* s0;     /* apply: reduced sums [C] */
  const float* s1;
  const void* add;     /* apply */
  float* part;         /* reduce: [nblocks][3][C] */
  long long rows;
  float inv_count;     /* 1 / rows */
  int32_t C;
  int32_t rows_per_image; /* H*W (se indexing, identity scatter) */
  int32_t add_kind;
  int32_t H, W, add_stride; /* geometry of gx for add_kind 2 */
  int32_t nblocks;
  /* fr_bn_bwd_apply, round 6 (ABI v6), nx != NULL: the launch ALSO leaves the rows of fr_bn_bwd_reduce for the BatchNorm whose
   * output gradient it has just formed -- part rows npart[nblocks][2][C] = (sum gx, sum gx * xhat_n) over this workgroup's rows,
   * xhat_n = (nx - nmean) * ninvstd, gx as rounded and stored: on tensors that only stream through HBM (56x56 x 64 channels at
   * batch 256: 103 MB) the separate reduce pass re-read gx right behind this one.  bf16, no gate; fr_reduce_parts(npart,
   * nblocks, 2, C, ...) adds the rows. */
  const void* nx;
  const float* nmean;
  const float* ninvstd;
  float* npart;
} FrBnBwdArgs;
int fr_bn_bwd_reduce(const FrBnBwdArgs* args, int dtype, void* stream);
int fr_bn_bwd_apply(const FrBnBwdArgs* args, int dtype, void* stream);

/* Round 6 (ABI v6): the sums of fr_stem_bwd_sums when G is the LAST tensor the first residual unit would write -- gx of its BN1 backward
 * (bottleneck_IR: BatchNorm2d(in_channel) at the head of res_layer, backbone/model_irse.py:57, + the shortcut gradient of
 * MaxPool2d(1, stride) / the residual stream, :52-54, :64-66).  `unit` holds exactly the arguments of the fr_bn_bwd_apply
 * call this replaces (g, x, gx, mean, invstd, gamma, s0, s1, inv_count, add, add_kind 0 | 1 | 2, H, W, add_stride = 2,
 * rows_per_image; C = 64, rows = M; no slope, no gate): the kernel forms gx element for element as that call does, stores it
 * (the weight-gradient kernel reads it next) and feeds the rounded values to the sums -- gx and part are bit-identical to
 * fr_bn_bwd_apply followed by fr_stem_bwd_sums, and the 411-MB tensor (batch 256) is read once less.  unit->x == NULL: the
 * unit's input IS the stem's output z = PReLU(BN(X Wp^T)) of this step (fr_stem_gemm_bn_prelu with the same X, Wp, scale, shift,
 * slope): it is recomputed from the y tile the sums need anyway, bit for bit, instead of read (another 411 MB less). */
int fr_stem_bwd_sums_from(const FrBnBwdArgs* unit, const void* X, const void* Wp, const float* mean, const float* invstd,
                          const float* scale, const float* shift, const float* slope, float* part, long long M, int K,
                          int nblocks, void* stream);

/* add partial rows in double: o_k[c] = sum_blk part[blk][k][c], k < K <= 3; NULL outputs are skipped.
 * Used for d gamma (k=1), d beta (k=0), d PReLU slope (k=2) and for the conv-epilogue partials. */
int fr_reduce_parts(const float* part, int nparts, int K, int C, float* o0, float* o1, float* o2, void* stream);

/* eval-mode BatchNorm coefficients from the running statistics */
int fr_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                      float eps, int C, float* mean, float* invstd, float* scale, float* shift, void* stream);
/* the same for every BatchNorm of a network in one launch (inference: util/utils.py:254-307 evaluates the backbone in
 * eval mode once per epoch on ~50 k images; 53 separate coefficient launches per forward were most of its launches) */
typedef struct FrBnEvalEntry {
  const float* rm;
  const float* rv;
  const float* gamma;
  const float* beta;
  float* mean;
  float* invstd;
  float* scale;
  float* shift;
  int32_t C;
  float eps;
} FrBnEvalEntry;
int fr_bn_eval_coeffs_multi(const FrBnEvalEntry* table_dev, int n, void* stream);

/* ---- SE block (model_irse.py:23-46 / restyle_psp_helpers.py:67-83) */
/* pooled[b][c] = mean_hw (x*scale+shift)  */
int fr_se_pool(const void* x, const float* scale, const float* shift, float* pooled, int B, int HW, int C,
               int dtype, void* stream);
/* the same from the per-strip column sums of a strip convolution's STATS epilogue (part[strip][2][C], the strips of
 * image b at rows b*rows_per_image ..): pooled[b][c] = scale[c] * (sum of those rows' [0][c]) / HW + shift[c] */
int fr_se_pool_parts(const float* part, int rows_per_image, const float* scale, const float* shift, float* pooled,
                     int B, int HW, int C, void* stream);
/* s = sigmoid(W2 relu(W1 pooled)); W1 [R][C], W2 [C][R]; saves hidden [B][R] */
int fr_se_mlp_fwd(const float* pooled, const float* w1, const float* w2, float* hidden, float* s, int B, int C,
                  int R, void* stream);
/* fr_se_pool_parts followed by fr_se_mlp_fwd in one launch (same arithmetic, same order; pooled [B][C] is written for the
 * weight gradient): the squeeze-excite branch of bottleneck_IR_SE (model_irse.py:23-46) behind a strip convolution */
int fr_se_pool_parts_mlp_fwd(const float* part, int rows_per_image, const float* scale, const float* shift,
                             const float* w1, const float* w2, float* pooled, float* hidden, float* s, int B, int HW,
                             int C, int R, void* stream);
/* fr_se_pool_parts_mlp_fwd on rows of `nv` vectors (3: FR_EPI_STATS_X rows -- sum y, sum y^2, sum y*x per strip) that also
 * derives, per IMAGE, the moments of the unit's output  o = (scale*y + shift)*s[b][c] + x  without a pass over it:
 *   om[b][0][c] = sum_hw o = s*(scale*Sy + shift*HW) + Sx
 *   om[b][1][c] = sum_hw o^2 = s^2*(scale^2*Syy + 2*scale*shift*Sy + shift^2*HW) + 2*s*(scale*Syx + shift*Sx) + Sxx
 * with (Sx, Sxx) = xm[b][0..1][c], the same per-image moments of the unit's INPUT (the om of the unit in front, or
 * fr_image_moments at the head of a chain).  fr_bn_finalize(om, B, C, B*HW, ..) then gives the next unit's BN1.
 * Replaces the statistics pass behind `res + shortcut` of bottleneck_IR_SE (backbone/model_irse.py:84-91). */
int fr_se_pool_parts_mlp_fwd_res(const float* part, int rows_per_image, int nv, const float* scale, const float* shift,
                                 const float* w1, const float* w2, float* pooled, float* hidden, float* s, int B, int HW,
                                 int C, int R, const float* xm, float* om, void* stream);
/* out[b][0][c] = sum_hw x, out[b][1][c] = sum_hw x^2 of an NHWC bf16 tensor [B][HW][C] (C / 8 a divisor of 256) */
int fr_image_moments(const void* x, int B, int HW, int C, float* out, void* stream);
/* gs[b][c] = sum_hw g * (x*scale+shift)  (gradient wrt the excite scale) */
int fr_se_gscale(const void* g, const void* x, const float* scale, const float* shift, float* gs, int B, int HW,
                 int C, int dtype, void* stream);
/* backward of the MLP: gpooled [B][C] (already divided by HW); dW1 [R][C], dW2 [C][R] OVERWRITTEN with the sums over
 * the batch in image order (reproducible, no atomics); gz [B][C], gh [B][R]: scratch the two launches hand over */
int fr_se_mlp_bwd(const float* gs, const float* s, const float* hidden, const float* pooled, const float* w1,
                  const float* w2, float* gpooled, float* dw1, float* dw2, float* gz, float* gh, int B, int C, int R,
                  int HW, void* stream);
/* fr_se_gscale followed by fr_se_mlp_bwd: the backward of the squeeze-excite branch of bottleneck_IR_SE (model_irse.py:23-46);
 * g = gradient at the unit output branch, x = y2.  gs_part == NULL: one 1024-thread workgroup per image, gs stays in LDS (same
 * arithmetic and order as the two calls).  gs_part != NULL (round 5; scratch [B][fr_se_gscale_slices(B, HW)][C] floats): the
 * squeeze runs over row slices of an image in 256-thread workgroups (B x slices of them: every CU has work at batch 128, and
 * they fit beside resident weight-gradient workgroups), the slices are added in order in front of the MLP part. */
int fr_se_gscale_mlp_bwd(const void* g, const void* x, const float* scale, const float* shift, const float* s,
                         const float* hidden, const float* pooled, const float* w1, const float* w2, float* gpooled,
                         float* dw1, float* dw2, float* gz, float* gh, float* gs_part, int B, int C, int R, int HW, int dtype,
                         void* stream);
/* Round 6 (ABI v6): dw1 = dw2 = NULL in fr_se_gscale_mlp_bwd leaves out its last launch, the weight gradients of the two
 * 1x1 convolutions of SEModule (fc1 / fc2, backbone/model_irse.py:28-35) -- sums over the batch in image order that feed
 * nothing downstream in the backward pass -- and fr_se_mlp_wgrad is that launch on its own, so that the caller can put it on
 * the weight-gradient stream (6-13 us per squeeze-excite unit off the main stream's chain).  gz [B][C], gh [B][R]: what
 * fr_se_gscale_mlp_bwd left; dw1 [R][C], dw2 [C][R] OVERWRITTEN. */
int fr_se_mlp_wgrad(const float* gz, const float* gh, const float* hidden, const float* pooled, float* dw1, float* dw2, int B,
                    int C, int R, void* stream);
/* Round 6 (ABI v6): fr_se_gscale_mlp_bwd (sliced squeeze, no weight-gradient launch) that ALSO leaves the rows of BN2's
 * backward sums, so that the fr_bn_bwd_reduce(se, gse) pass behind it -- a third read of (g, x) -- is not needed: behind the
 * excite gate BatchNorm2d(depth) of res_layer (backbone/model_irse.py:76-80, 86-87) sees g' = g * s[b][c] + gse[b][c], constant
 * over an image, so sum g' and sum g' * xhat follow from per-image sums of g, g * xhat and xhat taken in the squeeze pass:
 *   bn_part[b][0][c] = s * sum g + HW * gse,   bn_part[b][1][c] = s * sum g * xhat + gse * sum xhat   (xhat = (x - mean) * invstd)
 * fr_reduce_parts(bn_part, B, 2, C, dbeta, dgamma) adds the images.  gs_part: scratch [B][fr_se_gscale_slices(B, HW)][4][C]. */
int fr_se_gscale_mlp_bwd_sums(const void* g, const void* x, const float* scale, const float* shift, const float* mean,
                              const float* invstd, const float* s, const float* hidden, const float* w1, const float* w2,
                              float* gpooled, float* gz, float* gh, float* gs_part, float* bn_part, int B, int C, int R, int HW,
                              int dtype, void* stream);
int fr_se_gscale_slices(int B, int HW);

/* ---- output layer pieces (model_irse.py:144-148) */
/* a[b][(h*7+w)*C + c] = dropout(x*scale+shift): mask from a counter hash of (seed, element index in the
 * reference's C-major flatten order), keep prob 1-p, scaled 1/(1-p); p = 0 disables. */
int fr_bn_dropout(const void* x, void* out, const float* scale, const float* shift, long long rows, int C, int HW,
                  float p, uint64_t seed, int dtype, void* stream);
int fr_dropout_bwd(void* g, long long rows, int C, int HW, float p, uint64_t seed, int dtype, void* stream);

/* The same pair with the Linear layer's activation in the reference's own Flatten order (round 4):
 *   fr_bn_dropout_cm : out[b][c*HW + hw] = dropout(x[b][hw][c]*scale[c] + shift[c])     (x NHWC, out = Flatten of NCHW)
 *   fr_dropout_bwd_cm: out[b][hw][c] = mask * g_cm[b][c*HW + hw] / (1 - p)              (out of place, back to NHWC)
 * same counter-hash mask as above.  C % 64 == 0.  With it Linear(512*7*7, 512) (model_irse.py:146-147) runs on the fp32
 * master weight in its own layout: */
int fr_bn_dropout_cm(const void* x, void* out, const float* scale, const float* shift, int B, int C, int HW, float p,
                     uint64_t seed, int dtype, void* stream);
int fr_dropout_bwd_cm(const void* g_cm, void* out, int B, int C, int HW, float p, uint64_t seed, int dtype, void* stream);
/* Linear(K, O) as weight-streaming GEMMs on the fp32 master W [O][K] (csrc/linear_gemm.hip; bf16 activations, the weight
 * rounded to bf16 in registers, fp32 accumulation):
 *   fr_linear_fwd  : slab[s][b][o] = sum_{k in K-slice s} a[b][k]*W[o][k] (+ bias[o] in slice 0), s < slices; add the slabs
 *                    with fr_reduce_parts(slab, slices, 1, B*O, out, ..) -- a fixed order, reproducible.  fr_linear_slices
 *                    returns the slice count the library recommends for (O, K) (0: shape not served); O % 64 == 0, K % 32 == 0
 *   fr_linear_dgrad: ga[b][k] = sum_o g[b][o]*W[o][k], ga in the compute dtype; O % 128 == 0, K % 128 == 0
 * The weight gradient dW[o][k] = sum_b g[b][o] a[b][k] is fr_conv_wgrad (taps = 1) on the same a.
 * Replace nn.Linear(512*7*7, 512) of output_layer (backbone/model_irse.py:147) forward and its autograd data gradient. */
int fr_linear_slices(int O, int K);
int fr_linear_fwd(const void* a, const float* W, const float* bias, float* slab, int B, int O, int K, int slices,
                  void* stream);
int fr_linear_dgrad(const void* g, const float* W, void* ga, int B, int O, int K, void* stream);

/* ---- weight packing: fp32 master [Cout][taps][Cin] (channels-last storage of the OIHW Parameter)
 *   wp [Cout][taps][Cin] compute dtype (NULL to skip), wt [Cin][taps][Cout] compute dtype (NULL to skip) */
int fr_pack_weight(const float* w, void* wp, void* wt, int Cout, int taps, int Cin, int dtype, void* stream);
/* the same for every convolution of the network in one launch: table_dev = device array of records, chunks_dev =
 * device array of (tensor index, tile index) pairs, tile index over taps x ceil(Cout/32) x ceil(Cin/32); bf16 tensors with
 * Cout % 64 == 0 and Cin % 64 == 0 may use 64 x 64 tiles instead (16-byte accesses): tile index -(1 + index over taps x
 * Cout/64 x Cin/64); one tensor's chunks are all of one kind */
typedef struct FrPackTensor {
  const float* w;
  void* wp; /* may be NULL */
  void* wt; /* may be NULL */
  int32_t Cout, taps, Cin;
  int32_t frag; /* ABI v7, bit 0: wp in MFMA-fragment order (FrConvArgs.w_frag; N = Cout, K = Cin), bit 1: wt (N = Cin, K = Cout);
                   bf16, Cout % 64 == 0 and Cin % 64 == 0 (the 64 x 64 tile chunks) */
  const float* oscale; /* [Cout] or NULL: wp = w * oscale[co] (BatchNorm scale folded into the output channels) */
} FrPackTensor;
int fr_pack_weights_multi(const FrPackTensor* table_dev, const int32_t* chunks_dev, int nchunks, int dtype,
                          void* stream);
/* Linear(25088,512): torch layout [O][C*HW] (c-major) <-> NHWC-flatten [O][HW*C]; dir 0: torch->packed
 * (dtype out, optional transposed copy wt [HW*C][O]), dir 1: packed fp32 grad -> torch fp32 grad */
int fr_permute_linear(const float* in, void* out, void* wt, int O, int C, int HW, int dir, int dtype,
                      void* stream);
/* stem weight: torch [64][C][3][3] fp32 (any strides given) <-> packed [64][ldk] with k=(kh*3+kw)*C+c */
int fr_pack_stem(const float* w, long long s_o, long long s_c, long long s_h, long long s_w, void* wp, int Cout,
                 int C, int ldk, int dtype, void* stream);
int fr_unpack_stem_grad(const float* gp, float* gw, long long s_o, long long s_c, long long s_h, long long s_w,
                        int Cout, int C, int ldk, void* stream);
/* generic cast fp32 -> compute dtype (n elements) and back */
int fr_cast(const void* in, void* out, long long n, int dtype_in, int dtype_out, void* stream);

/* ---- margin head (head/metrics.py:97-140 ArcFace, :164-191 CosFace) */
/* row L2 normalise (F.normalize, eps 1e-12): xn [rows][ldn] compute dtype (rows..rows_pad zero filled),
 * optional transposed copy xt [D][ldt], inv [rows] = 1/max(||x||,eps) */
int fr_row_normalize(const float* x, void* xn, void* xt, float* inv, int rows, int rows_pad, int D, int ldt,
                     int dtype, void* stream);
/* gcos[m][n] = scale * g[m][n] * (n==label[m] ? dphi(cos_t[m]) : 1), zero padded to ldg columns */
int fr_margin_bwd(const float* g, const int64_t* label, const float* cos_t, void* gcos, int rows, int N, int ldg,
                  int kind, int easy, float cos_m, float sin_m, float th, float scale, int dtype, void* stream);
/* backward through F.normalize: gx = (G - xhat*(xhat.G)) * inv, xhat = x*inv, rows of D */
int fr_normalize_bwd(const float* G, const float* x, const float* inv, float* gx, int rows, int D, void* stream);

/* ---- focal loss on the batch-mean cross entropy (loss/focal.py:17-21) + top-k (util/utils.py:343-358) */
/* per row: lse[m], ce[m] = lse - z[label], rank[m] = #{n: z[n] > z[label]} */
int fr_ce_rows(const float* logits, const int64_t* label, float* lse, float* ce, int32_t* rank, int rows, int N,
               int ld, void* stream);
/* rank[m] only (accuracy on arbitrary logits, util/utils.py:343-358) */
int fr_rank_rows(const float* logits, const int64_t* label, int32_t* rank, int rows, int N, int ld, void* stream);
/* out[j] = scale * #{m < rows: rank[m] < k_j}, j < nk <= 4: precision@k in percent with scale = 100 / rows rounded to float,
 * the value correct_k.mul_(100.0 / batch_size) of util/utils.py:343-358 gives (counts are exact in fp32) */
int fr_topk_precision(const int32_t* rank, int rows, int nk, int k0, int k1, int k2, int k3, float scale, float* out,
                      void* stream);
/* scalars[0]=loss, [1]=dloss/dmeanCE, [2]=prec@1, [3]=prec@5, [4]=mean CE */
int fr_focal_finalize(const float* ce, const int32_t* rank, int rows, float gamma, float* scalars, void* stream);
/* grad[m][n] = gup[0]*scalars[1]/rows * (exp(z-lse[m]) - [n==label[m]]) */
int fr_focal_bwd(const float* logits, const int64_t* label, const float* lse, const float* scalars,
                 const float* gup, float* grad, int rows, int N, int ld, void* stream);

/* ---- class-sharded softmax (the process-per-GPU form of the class-dimension split of
 *      head/metrics.py:104-113,170-179): every rank holds the logits of a contiguous class range [lo, hi) for ALL rows of
 *      the global batch; label_local = label - lo, or any value outside [0, N) where another rank owns the label.
 *   fr_shard_row_stats : stats[0][m] = max_n z[m][n], stats[1][m] = sum_n exp(z - max), stats[2][m] = z[m][label_local]
 *                        (0 if not owned); stats is [3][rows] fp32
 *   fr_shard_combine   : stats_all = the ranks' stats in rank order, [world][3][rows]; lse[m] = log sum_n exp z over all
 *                        classes, tlogit[m] = the label's logit, ce[m] = lse - tlogit (fixed summation order: every
 *                        rank gets identical bits)
 *   fr_shard_rank_rows : rank[m] = #{n in this shard: z[m][n] > tlogit[m]} (sum over ranks = top-k rank of the label) */
int fr_shard_row_stats(const float* logits, const int64_t* label_local, float* stats, int rows, int N, int ld,
                       void* stream);
int fr_shard_combine(const float* stats_all, int world, int rows, float* lse, float* ce, float* tlogit, void* stream);
int fr_shard_rank_rows(const float* logits, const float* tlogit, int32_t* rank, int rows, int N, int ld, void* stream);

/* ---- GPU-side training-input transform (replaces the per-sample host transform of train.py:108-116
 *      applied in dataset.py:85-88): Resize(Hr x Wr, Pillow 8-bit bilinear, bit-exact) -> crop S x S at crop[b] = (x0, y0)
 *      -> horizontal flip where flip[b] -> ToTensor -> Normalize, for a batch of staged uint8 images.
 *   src  uint8 [B][Hin][Win][3] (HWC, as decoded)          out  float32 [B][3][S][S] (NCHW, what the reference collates)
 *   xtab int32 [Wr][kx+2], ytab int32 [Hr][ky+2]: per resized column / row (first input index, tap count, kx / ky integer
 *        weights with 22 fractional bits) -- Pillow's precompute_coeffs + normalize_coeffs_8bpc (frhip/input_pipeline.py)
 *   lut  float32 [256][3]: ((v / 255) - mean[c]) / std[c], every step rounded to float32
 *   crop offsets must satisfy 0 <= x0 <= Wr - S, 0 <= y0 <= Hr - S (checked by the caller: they live in device memory) */
int fr_augment_u8(const uint8_t* src, const int32_t* xtab, const int32_t* ytab, const int32_t* crop, const uint8_t* flip,
                  const float* lut, float* out, int B, int Hin, int Win, int Hr, int Wr, int S, int kx, int ky,
                  void* stream);
/* F.interpolate(x, size, mode='bilinear') of pSp.forward (backbone/restyle_psp.py:440-443: align_corners = False, no
 * antialias) on fp32 NCHW planes: in [planes][Hin][Win] -> out [planes][Hout][Wout]; planes = B * C <= 65535. */
int fr_resize_bilinear(const float* in, float* out, int planes, int Hin, int Win, int Hout, int Wout, void* stream);

/* ---- multi-tensor SGD with momentum (torch.optim.SGD defaults; train.py:196)
 *   d = g + wd*p ; buf = momentum*buf + d ; p -= lr*buf      (buf starts at 0, so the first step gives buf = d)
 * table_dev: device array of tensor records; chunks_dev: device array of (tensor index, chunk index) pairs,
 * one thread block per chunk of fr_sgd_chunk_elems() elements. */
typedef struct FrSgdTensor {
  float* p;
  const float* g;
  float* buf;
  long long n;
  float wd;
  int32_t pad_;
} FrSgdTensor;
int fr_sgd_chunk_elems(void);
int fr_sgd_step(const FrSgdTensor* table_dev, const int32_t* chunks_dev, int nchunks, float lr, float momentum,
                void* stream);

/* ---- multi-tensor Adam (torch.optim.Adam defaults: no weight decay, no amsgrad; the reference's OPTIMIZER_NAME ==
 *      'Adam' branch, train.py:197-198), same table / chunk scheme as fr_sgd_step:
 *   m += (g - m) * w1 ; v = v * beta2 + (w2 * g) * g ; p += (-step_size * m) / (sqrt(v) / bc2_sqrt + eps)
 * with the scalars torch derives in double precision on the host and rounds to float: w1 = 1 - beta1, w2 = 1 - beta2,
 * step_size = lr / (1 - beta1^step), bc2_sqrt = sqrt(1 - beta2^step).  Operation order = ATen's CPU kernels (lerp,
 * addcmul, addcdiv), so the update matches torch.optim.Adam to the last bits. */
typedef struct FrAdamTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;
} FrAdamTensor;
int fr_adam_step(const FrAdamTensor* table_dev, const int32_t* chunks_dev, int nchunks, float step_size, float w1,
                 float beta2, float w2, float eps, float bc2_sqrt, void* stream);

/* out[r][c] = bias ? bias[c] : 0  (fp32 [rows][C]); seeds the split-K accumulation of Linear(25088,512) */
int fr_fill_rows(float* out, const float* bias, long long rows, int C, void* stream);

/* ---- misc */
/* Capability queries.  SURVEY.md 8(b)-ii sketched ONE `fr_supported(op, dtype, Cin, Cout, H, W, stride)`; the build answers
 * the same question per kernel family instead, because the answer doubles as the launch geometry the caller must size
 * buffers for: fr_conv3x3_strip_parts / fr_conv3x3_s2_strip_parts return the number of partial-sum rows a launch will
 * write (0 = shape not served: the caller uses fr_conv_igemm, which serves every shape), fr_conv_wgrad_strip_supported
 * answers yes / no.  Every entry point additionally returns < 0 for an unsupported argument (text in
 * fr_last_error_string()), never silently computing something else. */
int fr_abi_version(void);
/* Run-time switches (kernel-family A/B switches, test hooks; README "Switches"): an int per name, first read from the
 * environment variable of that name, then cached for the life of the process.  fr_set_option overrides the cached value
 * (returns the previous one); fr_get_option reads it (dflt when neither set nor in the environment). */
int fr_set_option(const char* name, int value);
int fr_get_option(const char* name, int dflt);
/* Completion event of a kernel (ABI v6).  The backward pass hands its weight gradients to a second stream behind dependency
 * edges (the reference runs them inside autograd's single stream: loss.backward(), train.py:314); an edge set with
 * hipEventRecord costs the producing stream a marker packet the next kernel waits for (+3.0 ... 5.1 us per edge, three edges
 * per residual unit; tools/edge_probe.hip).  fr_arm_stop_event(event) makes the NEXT kernel this thread launches through one
 * of fr_conv3x3_strip / fr_conv3x3_s2_strip / fr_conv_igemm / fr_conv1x1_stream / fr_reduce_parts carry `event` (a
 * hipEvent_t) as its own completion signal instead (+0.0 ... 1.6 us).  fr_finish_stop_event(stream) ends the bracket: it
 * returns the number of kernels launched since the event was armed and, unless that is exactly one, records the event on
 * `stream` the ordinary way (entry points that launch no such kernel, or several), so the event is ALWAYS behind the whole
 * call.  Per-thread state, like the error string.  < 0: no event armed / hipEventRecord failed. */
int fr_arm_stop_event(void* event);
int fr_finish_stop_event(void* stream);
/* sizeof() of the argument structs as compiled, for binding self-checks: 0 FrConvArgs, 1 FrWgradArgs,
 * 2 FrApplyArgs, 3 FrBnBwdArgs, 4 FrSgdTensor, 5 FrPackTensor, 6 FrAdamTensor, 7 FrBnEvalEntry, 8 FrBnFinArgs */
int fr_struct_size(int which);
const char* fr_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif
