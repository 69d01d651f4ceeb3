/*
 * s2i_hip.h — C-ABI of the MI355X (gfx950) kernels under the StackGAN-v2 G/D train step.
 *
 * The reference (smallflyingpig/speech-to-image-translation-without-text) has no FFI of its own:
 * its hot path is stock torch.nn modules (StackGAN_v2/model.py:112-551) driven by
 * StackGAN_v2/trainer.py:375-489.  Each entry point below replaces one group of torch ops on that
 * path; the reference site it stands in for is cited next to it.  The Python host
 * (speech_to_image_translation_without_text_amd/ops.py) binds these with ctypes.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer, borrowed for the stream-ordered duration of the call;
 *   - nothing here allocates, frees or synchronises; scratch comes in through (ws, ws_bytes);
 *   - every call returns 0 on success; on failure a non-zero code, and s2i_last_error() holds text;
 *   - `stream` is a hipStream_t passed as void*;
 *   - activations are NHWC fp32, channel counts multiples of 4, spatial extents powers of two;
 *   - conv weights are consumed in the packed layout P[tap][Cin][Coutp] (Coutp = Cout rounded up
 *     to 4) written by s2i_pack_conv_weight from the reference's OIHW parameter tensors.
 */
#ifndef S2I_HIP_H
#define S2I_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2I_ABI_VERSION 4

/* conv geometry kinds */
#define S2I_CONV_K1      0  /* 1x1 / nn.Linear (model.py:179, 217)                              */
#define S2I_CONV_K3S1    1  /* conv3x3 pad 1 (model.py:125-128)                                 */
#define S2I_CONV_K4S2    2  /* Conv2d(k4,s2,p1) (model.py:371, 383-394); also dgrad of upBlock  */
#define S2I_TCONV_K4S2   3  /* 4-phase transposed conv k4 s2 p1: nearest x2 + conv3x3 collapsed
                               (model.py:133-140) and dgrad of Conv2d(k4,s2,p1)                 */

#define S2I_CONV_1D      4  /* (1 x kw) conv along W with stride / padding from the descriptor: the temporal
                               convolutions of the speech encoder (Audio_to_Image/speech_encoder.py:26-37);
                               forward only                                                      */

/* activations */
#define S2I_ACT_NONE     0
#define S2I_ACT_GLU      1  /* model.py:112-122 */
#define S2I_ACT_LRELU    2  /* nn.LeakyReLU(0.2), model.py:363, 373 */
#define S2I_ACT_TANH     3  /* model.py:293 */
#define S2I_ACT_RELU     4  /* speech_encoder.py:12 */

/* element types of activation tensors (bf16 activation mode, BASELINE config 4) */
#define S2I_DT_F32       0
#define S2I_DT_BF16      1

/* weight pack modes */
#define S2I_PACK_PLAIN   0  /* P[t][i][o] = W[o][i][t]                                          */
#define S2I_PACK_UPFOLD  1  /* 3x3 -> effective 4x4 taps of nearest-x2 + conv3x3               */

const char* s2i_last_error(void);
int  s2i_version(void);
/* Integer tuning knobs of the launch planners (tools and tests; every knob has a measured default; a negative value
   restores it, and s2i_get_tuning reports -1 for a knob at its default).  Returns 0, or non-zero for an unknown key.  The library reads no environment variable on a launch path: the ONE variable
   S2I_TUNE="key=value,key=value" is parsed once, when the library is loaded. */
int  s2i_set_tuning(const char* key, int value);
int  s2i_get_tuning(const char* key, int* value);
/* 0 when the current device is gfx950, non-zero (and last_error set) otherwise */
int  s2i_check_device(void);

/* ---- implicit-GEMM convolution (fp32 MFMA v_mfma_f32_32x32x2_f32) ------------------------- */
typedef struct s2i_conv_desc {
  int kind;      /* S2I_CONV_* / S2I_TCONV_K4S2                                                  */
  int B, H, W;   /* batch and spatial extent of the GATHERED tensor x                           */
  int Cx;        /* channels stored in x                                                        */
  int Cc;        /* channels of a per-image vector broadcast over space and concatenated FIRST
                    (torch.cat((c_code, h_code), 1), model.py:277, 434); 0 = none               */
  int N;         /* output channels                                                             */
  int wmode;     /* 0: weights used as P[t][k][n];  1: transposed per tap, P[t][n][k] (dgrad)    */
  int flip;      /* 1: tap t reads P[T-1-t] (dgrad of a stride-1 3x3)                            */
  int wR;        /* rows per tap in P                                                           */
  int ldw;       /* row stride of P (= Coutp of the forward layer)                              */
  int act;       /* epilogue: S2I_ACT_NONE / LRELU / TANH                                        */
  int stats;     /* 1: also emit per-row-tile column sums and sums of squares (BatchNorm)       */
  int ldy;       /* row stride of y                                                             */
  int groups;    /* BatchNorm groups: the rows are `groups` equal, independent batches stacked along
                    the batch axis (real / wrong / fake passes of trainer.py:390-392 in one launch);
                    statistics are kept per group.  0 or 1 = one batch                           */
  int nosplit;   /* 1: never split K (required with a class bias, s2i_conv_forward_cls)           */
  int kw, stride, pad; /* S2I_CONV_1D geometry (ignored by the other kinds)                       */
  int tile_rows; /* output rows per block of the fp32 matrix kernel: 0 = the planner chooses (96 or 128, whichever
                    fills whole rounds of the chip's block slots); 96 or 128 forces it (96 only where N > 64)     */
  int in_act;    /* s2i_conv_forward_in only: activation of the PRODUCING block applied to x while it is gathered
                    (S2I_ACT_LRELU); 0 elsewhere                                                              */
  int in_groups; /* ... and the number of BatchNorm groups of its coefficient table (0 or 1 = one)               */
} s2i_conv_desc;

/* scratch bytes s2i_conv_forward needs for this descriptor (split-K slabs; 0 when not split) */
size_t s2i_conv_workspace_bytes(const s2i_conv_desc* d);
/* number of partial rows the stats epilogue writes: part is [2][nparts][N] floats */
int    s2i_conv_stat_parts(const s2i_conv_desc* d);
/*
 * y[row][n] = act( sum_{t,c} X(row,t,c) * Wt[t][c][n] + bias[n] )
 * X gathers x (and cvec) per `kind`; rows enumerate (b,oy,ox) of the output grid.
 * Replaces F.conv2d / F.linear forward and their input-gradient on the reference path.
 */
int s2i_conv_forward(const s2i_conv_desc* d, const float* x, const float* cvec, const float* w,
                     const float* bias, float* y, float* part, void* ws, size_t ws_bytes,
                     void* stream);

/* Same, plus `cls_bias` [B][9][N]: a per-image, per-border-class term added before the statistics
   (the pre-reduced contribution of the broadcast c_code channels, see s2i_cvec_bias_table). */
int s2i_conv_forward_cls(const s2i_conv_desc* d, const float* x, const float* cvec, const float* w,
                         const float* bias, const float* cls_bias, float* y, float* part, void* ws,
                         size_t ws_bytes, void* stream);

/* Apply-on-load ("BatchNorm + LeakyReLU fused into the consuming convolution", model.py:369-376 followed by :371 / :360
   of the next block): x_raw is the RAW output y of the producing convolution and in_coef its (in_groups, 4, Cx)
   coefficient table [mean, invstd, scale, shift] (s2i_bn_finalize); the gather computes
   LeakyReLU(scale * y + shift) while it stages the operand, padding taps staying zero, so the producer's activated tensor
   is never written or read.  fp32, forward weight layout (wmode 0), 32 | Cx, no broadcast vector, kinds K1 / K3S1 / K4S2;
   with in_groups > 1 the rows of one producer group must be whole row tiles (as for `groups`). */
int s2i_conv_forward_in(const s2i_conv_desc* d, const float* x_raw, const float* in_coef, const float* w, float* y,
                        float* part, void* ws, size_t ws_bytes, void* stream);

/* ---- spatially constant channels of a 3x3 conv (c_code broadcast, model.py:272-279) -----------
 * A channel that is constant over space contributes sum over the IN-BOUNDS taps of c*W: a bias that
 * depends only on the image and on which borders the pixel touches (9 classes).  Forward: table
 * [B][9][N] from c (B,Cc) and the packed weight rows [0,Cc).  Backward: border sums of dY give, per
 * tap, the sum of dY over the pixels where the tap is in bounds; from them dc and dW[:, :Cc]. */
int s2i_cvec_bias_table(const float* cvec, const float* packed, int B, int Cc, int Ip, int Op, int N,
                        float* table, void* ws /* >= B*9*N floats */, size_t ws_bytes, void* stream);
size_t s2i_border_sums_workspace_bytes(int B, int H, int W, int C);
/* tapsum[b][t][c] = sum of dy[b,y,x,c] over pixels where tap t (3x3, pad 1) is in bounds */
int s2i_tap_sums(const float* dy, int B, int H, int W, int C, float* tapsum, void* ws, size_t ws_bytes,
                 void* stream);
/* dc[b][cc] = sum_{t,co} P[t][cc][co]*tapsum[b][t][co];  dW[co][cc][t] (+)= sum_b c[b][cc]*tapsum[b][t][co]
   (dW is the OIHW gradient of a parameter with I_total input channels, channels [0,Cc) written) */
int s2i_cvec_grads(const float* cvec, const float* packed, const float* tapsum, int B, int Cc, int Ip,
                   int Op, int N, int O, int I_total, float* dc, float* dw_oihw, int accumulate,
                   void* stream);

/* ---- weight gradient ------------------------------------------------------------------------ */
typedef struct s2i_wgrad_desc {
  int kind;      /* geometry of the gather applied to `a`                                        */
  int B, H, W;   /* extent of the gathered tensor a                                             */
  int Ca;        /* channels stored in a                                                        */
  int Cc;        /* broadcast-vector channels concatenated first (0 = none)                     */
  int N;         /* channels of the plain (un-gathered) operand g                               */
  int ldg;       /* row stride of g                                                             */
  int swap;      /* 0: result rows (tap,cin) x cols cout;  1: rows (tap,cout) x cols cin        */
  int fold;      /* 1: fold effective 4x4 taps back onto the 3x3 parameter (S2I_PACK_UPFOLD)    */
  int O, I, KH, KW; /* shape of the OIHW gradient tensor written                                */
  int accumulate;   /* 1: grad += result, 0: grad = result                                      */
  int i_off;        /* the I input channels computed here are channels [i_off, i_off+I) of a       */
  int I_total;      /* parameter with I_total input channels (0 = I): the c_code / h_code split    */
  int a_act;        /* s2i_conv_wgrad_in only: activation of the block that produced `a` (S2I_ACT_LRELU)    */
  int a_groups;     /* ... and the BatchNorm groups of its coefficient table (0 or 1 = one; at most 3)      */
} s2i_wgrad_desc;

size_t s2i_wgrad_workspace_bytes(const s2i_wgrad_desc* d);
/*
 * grad_oihw (+)= sum_rows A(row,t,c) * g[row][n]   — the weight-gradient of F.conv2d / F.linear
 * (autograd of model.py:125-128, 179, 217, 371, 383-394), written straight into the reference's
 * OIHW parameter layout.
 */
int s2i_conv_wgrad(const s2i_wgrad_desc* d, const float* a, const float* cvec, const float* g,
                   float* grad_oihw, void* ws, size_t ws_bytes, void* stream);

/* The weight gradient of the same consumer: the gathered operand is the producer's RAW output a_raw with its coefficient
   table, activated while it is staged (s2i_conv_forward_in).  Only where s2i_conv_wgrad_in_eligible(d) says so (the
   generic fp32 128 x 128 plan: every layer of the discriminator towers); otherwise the caller materialises the operand. */
int s2i_conv_wgrad_in_eligible(const s2i_wgrad_desc* d);
int s2i_conv_wgrad_in(const s2i_wgrad_desc* d, const float* a_raw, const float* a_coef, const float* g,
                      float* grad_oihw, void* ws, size_t ws_bytes, void* stream);


/* ---- split-bf16 matrix products (1 / 2 / 3 bf16 planes; opt-in) ---------------------
 * Same convolution as s2i_conv_forward_cls, with every fp32 operand written as a sum of `planes` bf16 numbers and the
 * products taken by v_mfma_f32_32x32x16_bf16 with fp32 accumulation (planes = 2: 3 products, ~2^-16 relative;
 * planes = 3: 6 products, ~2^-23).  wsplit holds the weights pre-split as [plane][tap][np][kp] bf16, n = output column of the
 * GEMM, k = its reduction index (s2i_split_packed_weight: out_cr for the forward, out_rc for the input gradient that
 * the fp32 path expresses with wmode = 1).  Eligible when the gathered channel count (and Cc) are multiples of 32. */
int s2i_conv_split_eligible(const s2i_conv_desc* d);
int s2i_conv_forward_split(const s2i_conv_desc* d, const float* x, const float* cvec,
                           const unsigned short* wsplit, int planes, int np, int kp, const float* bias,
                           const float* cls_bias, float* y, float* part, void* ws, size_t ws_bytes,
                           void* stream);
/* weight gradient with split-bf16 products (both operands are split while they are staged) */
size_t s2i_wgrad_workspace_bytes_split(const s2i_wgrad_desc* d, int planes);
int s2i_conv_wgrad_split(const s2i_wgrad_desc* d, int planes, const float* a, const float* cvec,
                         const float* g, float* grad_oihw, void* ws, size_t ws_bytes, void* stream);
/* packed fp32 weights P[T][R][C] (s2i_pack_conv_weight) -> bf16 planes, both operand layouts from one read:
   out_rc [plane][T][R][C] (input gradient) and out_cr [plane][T][C][R] (forward); either may be NULL */
int s2i_split_packed_weight(const float* packed, int T, int R, int C, int planes, unsigned short* out_rc,
                            unsigned short* out_cr, void* stream);


/* ---- bf16 activation mode (BASELINE config 4: bf16 activations / weights in HBM, bf16 MFMA, fp32 accumulate) -------
 * Activations between the fused blocks are bf16 NHWC; BatchNorm statistics come from the fp32 accumulators; master
 * weights, gradients, Adam and EMA stay fp32 (trainer.py:236-252).  The same convolutions as above
 * (model.py:125-140, 358-398 and their gradients) for layers whose channel count is a multiple of 32:
 *   - the block stages a 2-D input patch with its halo in LDS once per channel chunk and all taps read from it;
 *   - weights arrive pre-arranged by s2i_pack_conv_weight_bf16 as Wb[phase][chunk][tap][Npad][CK] bf16, from the packed
 *     fp32 copy P[t][R][C]: d->wmode = 0 uses P[t][k][n] (forward), 1 uses P[t][n][k] (input gradient), d->flip /
 *     the transposed-conv parity select the source tap; P may point at a row offset inside a tap (channel split).
 * d->Cc must be 0 (a broadcast vector is concatenated by the caller), d->act NONE, N and ldy multiples of 8. */
int    s2i_conv_bf16_eligible(const s2i_conv_desc* d);
size_t s2i_conv_bf16_workspace_bytes(const s2i_conv_desc* d);
int    s2i_conv_bf16_stat_parts(const s2i_conv_desc* d);
size_t s2i_conv_bf16_weight_elems(const s2i_conv_desc* d);
/* Identifies the weight arrangement (CK | Npad << 8) the plan of this descriptor consumes: the same layer at another batch
 * size or map size may be planned onto another kernel, so a cache of packed bf16 weights is keyed by it.  -1 on error. */
int    s2i_conv_bf16_weight_layout(const s2i_conv_desc* d);
int s2i_pack_conv_weight_bf16(const s2i_conv_desc* d, const float* packed, int R, int C, unsigned short* out,
                              void* stream);
/* The bf16 copies of a whole network in one launch (after the fused Adam step).  s2i_pack16_item_fill (host only, no
 * launch) fills the item of one (descriptor, packed fp32 source, destination) exactly as s2i_pack_conv_weight_bf16 would
 * pack it and returns its block count (-1 on error); the caller assigns block0 = running sum of the counts and uploads the
 * array. */
typedef struct s2i_pack16_item {
  const float* P;        /* packed fp32 source (P[t][R][C], possibly offset to a row inside a tap) */
  unsigned short* out;   /* Wb[phase][chunk][tap][Npad][CK] */
  int R, C, kind, flip, transpose, T, nphase, Nn, Npad, Kk, CK, gx, gy, block0;
} s2i_pack16_item;
int s2i_pack16_item_fill(const s2i_conv_desc* d, const float* packed, int R, int C, unsigned short* out,
                         s2i_pack16_item* item);
int s2i_pack_conv_weights_bf16_batched(const s2i_pack16_item* items_dev, int n, int total_blocks, void* stream);
int s2i_conv_forward_bf16(const s2i_conv_desc* d, const unsigned short* x, const unsigned short* w,
                          const float* cls_bias, unsigned short* y, float* part, void* ws, size_t ws_bytes,
                          void* stream);
/* The fp32-MFMA convolution / weight gradient of s2i_conv_forward_cls / s2i_conv_wgrad with x / y (a / g) stored as
   S2I_DT_F32 or S2I_DT_BF16: the edges of the bf16 mode (image tensors, channel counts that are not multiples of 32).
   Two bf16 operands of a weight gradient run on the bf16 matrix cores. */
int s2i_conv_forward_dt(const s2i_conv_desc* d, const void* x, int x_dtype, const float* cvec, const float* w,
                        const float* bias, const float* cls_bias, void* y, int y_dtype, float* part, void* ws,
                        size_t ws_bytes, void* stream);
size_t s2i_wgrad_workspace_bytes_dt(const s2i_wgrad_desc* d, int a_dtype, int g_dtype);
int s2i_conv_wgrad_dt(const s2i_wgrad_desc* d, const void* a, int a_dtype, const float* cvec, const void* g,
                      int g_dtype, float* grad_oihw, void* ws, size_t ws_bytes, void* stream);
/* `_dt` forms of the BatchNorm / activation / layout kernels below: every activation tensor of the call has `dtype` */
int s2i_bn_act_forward_dt(int dtype, const void* y, long long M, int groups, int C, const float* coef4, int act,
                          const void* residual, void* out, void* stream);
int s2i_bn_act_bwd_reduce_dt(int dtype, const void* y, const void* dout, int lddout, long long M, int groups, int C,
                             const float* coef4, int act, float* part, int nparts, void* stream);
int s2i_bn_act_bwd_apply_dt(int dtype, const void* y, const void* dout, int lddout, long long M, int groups, int C,
                            const float* coef4, const float* red2, int act, void* dy, void* stream);
int s2i_act_backward_dt(int dtype, const void* out, const void* dout, int lddout, long long M, int C, int act,
                        void* dy, void* stream);
int s2i_nchw_to_nhwc_dt(int dtype, const float* src, void* dst, int B, int C, int H, int W, int Cp, void* stream);
int s2i_nhwc_to_nchw_dt(int dtype, const void* src, int lds, float* dst, int B, int C, int H, int W, void* stream);
int s2i_spatial_sum_dt(int dtype, const void* src, int ld, int B, int HW, int C, float* dst, void* ws,
                       size_t ws_bytes, void* stream);
int s2i_tap_sums_dt(int dtype, const void* dy, int B, int H, int W, int C, float* tapsum, void* ws, size_t ws_bytes,
                    void* stream);
/* dst[n] = (dst_dtype) src[n] for a contiguous tensor, n % 4 == 0 */
int s2i_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, void* stream);

/* OIHW parameter -> packed P[t][Ip][Op] (Ip >= I, Op = O rounded up to 4; padding zero filled) */
int s2i_pack_conv_weight(const float* w_oihw, float* packed, int O, int I, int KH, int KW,
                         int Ip, int mode, void* stream);
/* The same for a whole network in one launch: `items` is a DEVICE array; item k owns the linear blocks
   [block0, block0 + gx * ceil(Ip / 8)) with gx = ceil(Op / 32), block0 ascending; total_blocks = their sum;
   max_taps = the largest KH*KW.  Used after the fused Adam step (trainer.py:236-252 equivalent). */
typedef struct s2i_pack_item {
  const float* w;   /* OIHW parameter */
  float* packed;    /* P[t][Ip][Op]   */
  int O, I, KH, KW, Ip, mode, gx, block0;
} s2i_pack_item;
int s2i_pack_conv_weights_batched(const s2i_pack_item* items_dev, int n, int total_blocks, int max_taps,
                                  void* stream);

/* ---- BatchNorm (training statistics) + activation ------------------------------------------ */
/*
 * Reduce the conv epilogue's partials to batch statistics (nn.BatchNorm2d/1d in training mode,
 * model.py:137, 147, 158, 161, 218, 361, 372): mean, biased var -> invstd, scale = gamma*invstd,
 * shift = beta - mean*scale; running_mean/var updated with momentum (unbiased var), as torch does.
 * out4 = [mean | invstd | scale | shift], each C floats, once per group (groups x 4 x C); `count` is
 * the number of rows of ONE group; the partial rows are split evenly over the groups, which are
 * processed in order (running statistics receive `groups` successive updates).
 */
int s2i_bn_finalize(const float* part, int nparts, int groups, int C, long long count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var,
                    long long* num_batches_tracked /* int64, += groups (one per stacked batch); may be NULL */, float momentum,
                    float eps, float* out4, void* stream);
/* eval-mode BatchNorm: scale/shift from the running statistics (trainer.py:681-803 path) */
int s2i_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* out4, void* stream);
/*
 * out = act(scale*y + shift) (+ residual).  GLU halves the channel count (C -> C/2).
 * Replaces BatchNorm apply + GLU / LeakyReLU / ResBlock add (model.py:116-122, 165-169).
 */
int s2i_bn_act_forward(const float* y, long long M, int groups, int C, const float* coef4, int act,
                       const float* residual, float* out, void* stream);
/* column sums of the raw tensor when no conv epilogue produced them: part = [2][nparts][C] */
int s2i_colstats(const float* y, long long M, int C, int ldy, float* part, int nparts,
                 void* stream);
/*
 * Backward of bn_act_forward, pass 1: per-channel sums of dz and dz*xhat (dz = gradient w.r.t. the
 * BatchNorm output after un-doing the activation).  part = [2][nparts][C].
 */
int s2i_bn_act_bwd_reduce(const float* y, const float* dout, int lddout, long long M, int groups, int C,
                          const float* coef4, int act, float* part, int nparts, void* stream);
/* finalise pass 1: dgamma, dbeta (accumulated or assigned) and the two means for pass 2.
   red2 = [mean_dz | mean_dz_xhat], each C floats. */
int s2i_bn_bwd_finalize(const float* part, int nparts, int groups, int C, long long count, float* dgamma,
                        float* dbeta, int accumulate, float* red2, void* stream);
/* pass 2: dy = scale * (dz - mean_dz - xhat*mean_dz_xhat) */
int s2i_bn_act_bwd_apply(const float* y, const float* dout, int lddout, long long M, int groups, int C,
                         const float* coef4, const float* red2, int act, float* dy, void* stream);

/* ---- plain activations ---------------------------------------------------------------------- */
/* dy = dout * act'(out) for LRELU / TANH given the forward OUTPUT (sign- / value-recoverable) */
int s2i_act_backward(const float* out, const float* dout, int lddout, long long M, int C, int act,
                     float* dy, void* stream);
/* 2-D GLU without BatchNorm (CA_NET, model.py:183): out[M][C/2] */
int s2i_glu_forward(const float* x, long long M, int C, float* out, void* stream);
int s2i_glu_backward(const float* x, const float* dout, long long M, int C, float* dx,
                     void* stream);

/* ---- layout ---------------------------------------------------------------------------------- */
/* NCHW (C channels) -> NHWC with Cp >= C channels (extra channels zero), and back */
int s2i_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int Cp,
                     void* stream);
int s2i_nhwc_to_nchw(const float* src, int lds, float* dst, int B, int C, int H, int W,
                     void* stream);
/* [-1,1] float image (NHWC, row stride lds, first 3 channels) -> HWC uint8 RGB, the reference's
   `img.add(1).div(2).mul(255).clamp(0, 255).byte()` (trainer.py:676) fused with its permute(1,2,0) */
int s2i_image_to_u8(const float* src, int lds, unsigned char* dst, long long npix, void* stream);
/* HWC uint8 RGB [B][H][W][3] -> normalised NCHW float [B][3][H][W]: the reference's per-sample
   `ToTensor()` (x / 255) followed by `Normalize((.5,.5,.5), (.5,.5,.5))` ((t - 0.5) / 0.5), datasets.py:440-442,
   applied on the device to the collated uint8 batch (same fp32 operations in the same order: bit-identical) */
int s2i_u8_to_image(const unsigned char* src, float* dst, int B, int H, int W, void* stream);
/* sum over the H*W rows of each image of the first C columns of a [B*HW][ld] tensor -> [B][C] */
int s2i_spatial_sum(const float* src, int ld, int B, int HW, int C, float* dst, void* ws,
                    size_t ws_bytes, void* stream);
size_t s2i_spatial_sum_workspace_bytes(int B, int HW, int C);

/* ---- CA_NET reparameterisation + KL (model.py:182-200, trainer.py:54-58) -------------------- */
/* h = GLU output [B][2E] = [mu | logvar];  c = eps*exp(0.5*logvar) + mu */
int s2i_reparam_forward(const float* h, const float* eps, int B, int E, float* c, void* stream);
/* dh[:, :E] = dc + dmu_extra ; dh[:, E:] = dc*eps*0.5*exp(0.5*logvar) + dlogvar_extra */
int s2i_reparam_backward(const float* h, const float* eps, const float* dc, const float* dmu,
                         const float* dlogvar, int B, int E, float* dh, void* stream);
/* kl = -0.5*mean(1 + logvar - mu^2 - exp(logvar));  also the gradient scaled by `gscale` */
int s2i_kl_forward(const float* mu, int ldmu, const float* logvar, int ldlv, int B, int E,
                   float* kl, void* stream);
int s2i_kl_backward(const float* mu, int ldmu, const float* logvar, int ldlv, int B, int E,
                    const float* gout, float* dmu, float* dlogvar, void* stream);

/* ---- logit heads: Conv2d(C,1,k=4,s=4)+Sigmoid on a 4x4 map (model.py:414-422) + BCE --------- */
/* x NHWC [B][16][C]; w OIHW [1][C][4][4]; prob[b] = sigmoid(<x_b,w> + bias) */
int s2i_logit_forward(const float* x, const float* w, const float* bias, int B, int C,
                      float* prob, void* stream);
/* dlogit[b] given dprob; dx (+)= dlogit*w ; dw (+)= sum_b dlogit*x ; dbias (+)= sum dlogit */
int s2i_logit_backward(const float* x, const float* w, const float* prob, const float* dprob,
                       int B, int C, float* dx, int acc_dx, float* dw, float* dbias, int acc_dw,
                       void* stream);
/* nn.BCELoss(mean) with torch's log clamp at -100 (trainer.py:394-409, 439-443):
   loss (+)= weight * mean(-(t*log p + (1-t)*log(1-p))) ; dprob = weight*gout * dL/dp */
int s2i_bce_forward(const float* prob, float target, int B, float weight, float* loss,
                    int accumulate, void* stream);
int s2i_bce_backward(const float* prob, float target, int B, float weight, const float* gout,
                     float* dprob, void* stream);

/* Sum of G*H BCE terms in one launch (the six terms of trainer.py:394-409): probs[h] is head h's
   probabilities for G stacked batches of B rows; term (g,h) uses target[g*H+h] and weight[g*H+h].
   loss = sum_{g,h} weight * mean_b BCE(probs[h][g*B+b], target);  backward writes dprobs[h] likewise. */
int s2i_bce_multi_forward(const float* const* probs_dev, const float* target, const float* weight, int G, int H,
                          int B, float* loss, void* stream);
int s2i_bce_multi_backward(const float* const* probs_dev, const float* target, const float* weight, int G, int H,
                           int B, const float* gout, float* const* dprobs_dev, void* stream);

/* ---- class-aware loss (trainer.py:298-311) ------------------------------------------------------ */
/* scores = X X^T [B][B] (from s2i_conv_forward, K1, wmode 1); labels int32 [B];
   loss = max(0, mean(S) - mean(S[same class, off-diagonal])) / D, 0 when no such pair.
   dscores = d loss / d S (so that dX = (dS + dS^T) X), both scaled by nothing: caller scales. */
int s2i_cal_loss(const float* scores, const int* labels, int B, int D, float* loss, int accumulate,
                 float* dscores_sym, void* stream);

/* ---- speech-encoder front-end (Audio_to_Image/speech_encoder.py:15-97), inference ---------------- */
/* nn.MaxPool2d((1,3), stride (1,2), padding (0,1)) on NHWC [B][H][W][C] -> [B][H][W/2][C] */
int s2i_maxpool_w3s2(const float* x, int B, int H, int W, int C, float* y, void* stream);
/*
 * One nn.LSTM step for every sequence of a packed batch, one direction.
 *   xproj [B][T][ldx] (this direction's 4*Hd gate pre-activations from the input, biases included, at
 *   column offset already applied), hproj [B][4*Hd] = h_prev W_hh^T, gates in torch order (i, f, g, o).
 *   Sequence b has lens[b] valid steps; at step `s` it processes t = s (forward) or lens[b]-1-s (reverse);
 *   finished sequences keep their state and write nothing (padded outputs stay zero).
 *   h, c [B][Hd] are updated in place; out [B][T][ldo] receives h at (b, t) at column offset applied.
 */
int s2i_lstm_cell(const float* xproj, int ldx, const float* hproj, const int* lens, int B, int T, int Hd,
                  int step, int reverse, float* h, float* c, float* out, int ldo, void* stream);
/* The same step for all D directions in one launch, with the recurrent projection fused in: gates = xproj[b][t] +
 * h_in[d][b] . whh_d^T (whh_* are the reference's weight_hh_l0 / weight_hh_l0_reverse, (4*Hd, Hd) row-major);
 * h_in / h_out [D][B][Hd] are distinct buffers (ping-pong), c [D][B][Hd] in place; direction 1 runs reversed. */
int s2i_lstm_step(const float* xproj, int ldx, const float* whh_fwd, const float* whh_rev, const int* lens,
                  int B, int T, int Hd, int D, int step, const float* h_in, float* h_out, float* c, float* out,
                  int ldo, void* stream);
/* y[b][c] = mean over the T rows of x[b][t][c] (sent_emb = output.mean(-2), speech_encoder.py:93) */
int s2i_time_mean(const float* x, int B, int T, int C, float* y, void* stream);

/* ---- optimiser (trainer.py:236-252, 571-572) -------------------------------------------------- */
/* torch.optim.Adam (no weight decay, no amsgrad) on a flat buffer, step = 1-based step count */
int s2i_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                  float beta2, float eps, int step, const int* step_dev, float gscale, void* stream);
/* *counter += 1 on the stream (device-resident Adam step count, so a captured hipGraph of the
   train step replays with the right bias correction) */
int s2i_increment(int* counter, void* stream);
/* avg = decay*avg + (1-decay)*p */
int s2i_ema_update(float* avg, const float* p, long long n, float decay, void* stream);
/* y = x * a_dev[0] (the scalar lives on the device: no host synchronisation) */
int s2i_scale_dev(float* y, const float* x, long long n, const float* a_dev, void* stream);
/* y = a*x (+ y) elementwise helpers used for gradient averaging and accumulation */
int s2i_axpby(float* y, const float* x, long long n, float a, float b, void* stream);

/* ---- launch-plan replay (host-side machinery of this build; the reference's loop body, trainer.py:536-572, is a
 * Python loop over torch ops) -----------------------------------------------------------------------------------------
 * One stream's piece of the train step is recorded ONCE by HIP stream capture (so launches that do not come from this
 * library are recorded too) and re-issued per step from one C call: s2i_plan_create walks the captured hipGraph_t in
 * dependency order and keeps every node's function, geometry and argument pointers; s2i_plan_replay issues them as plain
 * launches on the given stream.  The caller keeps the captured graph (the argument storage belongs to it) and the memory
 * pool the capture allocated from alive for as long as the plan is used.  Only kernel, 1-D memset and memcpy nodes
 * are accepted; anything else fails with a message (the caller then falls back to the eager step).
 * counts, if not NULL, receives [kernels, memsets, memcpys]. */
int s2i_plan_create(void* hip_graph, void** plan_out, int* counts);
int s2i_plan_replay(void* plan, void* stream);
int s2i_plan_destroy(void* plan);

#ifdef __cplusplus
}
#endif
#endif /* S2I_HIP_H */
