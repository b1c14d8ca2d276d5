/*
 * desenet_hip.h -- C ABI of libdesenet_hip.so: the MI355X (gfx950) kernels behind DeSeNet's CNN hot path.
 *
 * The reference (splwany/DeSeNet) has NO native/FFI layer: its hot path is Python nn.Module code that reaches
 * ATen kernels.  The drop-in boundary is therefore the module surface core.models.common / core.models.yolo
 * (mirrored in desenet_amd/core/models/), and THIS header is what those mirrored modules bind (ctypes,
 * desenet_amd/_lib.py).  Each entry point cites the reference statement(s) it replaces; paths are relative
 * to the reference repository root.
 *
 * Conventions
 *  - Activations are NHWC ("channels last") views: element (n,h,w,c) lives at ptr[((n*H + h)*W + w)*ldc + c],
 *    ldc >= C.  ldc > C addresses a channel slice of a wider buffer: producers write straight into concat
 *    buffers (torch.cat of common.py:145,185,545,615,693 and yolo.py:195 never materialises a copy).
 *  - dtype: DSN_F32 (config 2, fp32 inference) or DSN_BF16 (configs 3-5, bf16 storage, fp32 accumulate).
 *    Per-channel vectors (bias, BN scale/shift/statistics, gradients of weights) are always fp32.
 *  - Weights are passed PACKED: conv fwd  [Co][KH][KW][Ci]  (dsn_pack_weight_fwd),
 *                               conv dgrad [Ci][KH][KW][Co] (dsn_pack_weight_dgrad).
 *  - Ownership: the caller (PyTorch caching allocator) owns every buffer; the library allocates nothing,
 *    retains no pointer after return, and needs workspaces to be passed in (sizes from dsn_*_workspace_bytes).
 *  - Streams: every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no hidden sync.
 *  - Errors: 0 = success; < 0 = dsn_status (bad argument / unsupported shape); > 0 = hipError_t.  The message
 *    of the last failure on the calling thread is returned by dsn_last_error().  Nothing aborts.
 */
#ifndef DESENET_HIP_H
#define DESENET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSN_VERSION 100 /* 0.1.0 */

enum { DSN_F32 = 0, DSN_BF16 = 1 };
enum { DSN_OK = 0, DSN_EINVAL = -1, DSN_EUNSUPPORTED = -2, DSN_EWORKSPACE = -3 };
enum { DSN_ACT_NONE = 0, DSN_ACT_SILU = 1, DSN_ACT_SIGMOID = 2 };

typedef struct {
    void*   ptr;
    int32_t dtype;
    int32_t n, h, w, c;
    int64_t ldc;
} dsn_tensor;

typedef struct {
    int32_t kh, kw;      /* kernel size                                   */
    int32_t stride;      /* same in both directions                       */
    int32_t pad;         /* autopad k//2 (common.py:32-39) or explicit    */
    int32_t dil;         /* dilation (RFB2 branch1/2: 2, 3)               */
    int32_t act;         /* DSN_ACT_* applied after bias                  */
    int32_t accumulate;  /* 1: out += result (gradient fan-in)            */
} dsn_conv_params;

/* ---- BatchNorm + activation from accumulators (operand descriptor of dsn_lazy_materialize) ---------------------------------
 * In training a Conv block computes z = act(bn(conv(x))) (common.py:53) with batch statistics: the statistics of the WHOLE
 * tensor must exist before a single z can be formed.  The block leaves its RAW convolution output y plus the per-channel fp64
 * sums its epilogue produced (dsn_conv2d_fwd_bnacc); ONE elementwise launch (dsn_lazy_materialize) folds the sums in its prologue
 * and writes z (+ shortcut).  Its input is described as up to DSN_LAZY_MAXSEG channel segments (one per BatchNorm module whose
 * output lies in the tensor -- a merged C3 cv2|cv1 pair has two):
 *   input channel c in [c0, c1): accumulator channel ch0 + (c - c0), parameter index p0 + (c - c0)
 *   acc != NULL : fold the sums here (count, eps, gamma, beta -- NULL gamma/beta = 1 / 0); else scale/shift arrays (both NULL:
 *   identity).  act = DSN_ACT_* applied after the affine map.  Channels outside every segment pass through unchanged.
 * (Rounds 2-3 also let the CONSUMING convolution apply the transform in its operand loader -- dsn_conv2d_fwd_lazy[_z],
 *  dsn_conv2d_wgrad_plan_lazy; measured slower on MI355X and removed in round 4.) */
#define DSN_LAZY_MAXSEG 6
typedef struct {
    int32_t       c0, c1;       /* channel range of the consumer's input */
    int32_t       ch0, p0;      /* first accumulator channel / first parameter index of the range */
    int32_t       acc_c;        /* channel count of the accumulator buffer ([DSN_BN_NREP][2][acc_c] doubles) */
    int32_t       act;
    const void*   acc;
    const float*  gamma;
    const float*  beta;
    const float*  scale;
    const float*  shift;
    double        count;        /* pixels per channel behind the sums */
    float         eps, _pad;
} dsn_lazy_seg;
typedef struct {
    int32_t      nseg, _pad;
    dsn_lazy_seg seg[DSN_LAZY_MAXSEG];
} dsn_lazy_in;

/* ---- BatchNorm backward sums in the epilogue of the input-gradient convolution that produces dz ----------------------------
 * The backward of z = act(bn(y)) (common.py:53) needs two per-channel sums over the whole tensor, sum g and sum g * yhat with
 * g = dz * act'(y*scale + shift), before a single dy can be formed (dsn_bn_act_bwd_reduce).  dz itself is the output of the
 * input-gradient convolution of the block's consumer: when that launch writes the FINAL value of dz (every other contribution --
 * a shortcut, an earlier consumer -- is already in its epilogue's residual / accumulate operands), its epilogue can form g from
 * the value it is about to store and the producer's y at the same pixel, and add the sums to the producer's accumulators
 * (dsn_bn_workspace_bytes, zero on entry): the reduce launch and its read of y and dz disappear.  Up to DSN_BNRED_MAXSEG channel
 * segments of dx (a concat of two blocks' outputs -- C3's [m(..) | cv2(x)], common.py:145):
 *   dx channel c in [c0, c1): y element (row, c - c0) of `y` (row stride yld elements, same dtype as dx), parameter index
 *   c - c0 of scale / shift / mean / rstd, accumulator channel ch0 + (c - c0) of acc ([DSN_BN_NREP][2][acc_c] doubles). */
#define DSN_BNRED_MAXSEG 2
typedef struct {
    int32_t      c0, c1;
    int32_t      ch0, acc_c;
    int32_t      act, _pad;
    const void*  y;
    int64_t      yld;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* rstd;
    void*        acc;
} dsn_bnred_seg;
typedef struct {
    int32_t       nseg, _pad;
    dsn_bnred_seg seg[DSN_BNRED_MAXSEG];
} dsn_bnred;
/* dsn_conv2d_dgrad[_res] / dsn_conv2d_dgrad_s2 whose launch completes dz for the blocks of `red` (residual may be NULL).
 * DSN_EUNSUPPORTED when the layer cannot take the vectorised store (channel counts / strides that are not multiples of the
 * 16-byte vector): run the plain entry point and dsn_bn_act_bwd_reduce. */
int dsn_conv2d_dgrad_bnred(const dsn_tensor* dy, const void* w_packed, const dsn_tensor* dx, const dsn_conv_params* p,
                           const dsn_tensor* residual, const dsn_bnred* red, void* stream);
int dsn_conv2d_dgrad_s2_bnred(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p,
                              const dsn_bnred* red, void* stream);

/* Kernel selection of the weights-stationary persistent convolution kernels (csrc/conv_ws.hip) behind dsn_conv2d_fwd* /
 * dsn_conv2d_dgrad*: per kernel (1x1, 3x3) 0 = never, 1 = default (per-shape choice measured on MI355X), 2 = without the
 * data-gradient extras variants, 3 = every eligible launch.  A negative argument leaves that mode unchanged.  Returns
 * 16 * mode_1x1 + mode_3x3 after the update.  Results do not depend on the mode beyond fp32 summation order. */
int         dsn_ws_mode(int32_t mode_1x1, int32_t mode_3x3);
/* The same for the big-tile ping-pong 3x3 kernel (csrc/conv_pp.hip: 256 output pixels x 128 / 256 channels per block, bf16, k3 /
 * s1 / d1, input channels in whole 64-channel slabs): 0 = never, 1 = default (layers whose 16 x 16 patches cover >= 80 % of the map
 * and give >= 160 blocks), 2 = every eligible layer, 3 = the same with 256-channel tiles wherever possible, 4 / 5 = every eligible layer on 128-channel tiles with two / one
 * block(s) per CU (tests, A/B runs).  Negative: unchanged.  Returns the mode after the update. */
int         dsn_pp_mode(int32_t mode);
/* The same for the 1x1 form of that kernel (256 consecutive pixels x 128 / 256 channels, bf16, k1 / s1, input channels a multiple
 * of 32): 0 = never, 1 = default (>= 512 input and >= 256 output channels, >= 160 blocks), 2 = every eligible layer on 128-channel
 * tiles, 3 = the same with 256-channel tiles wherever possible.  Negative: unchanged.  Returns the mode after the update. */
int         dsn_pp1_mode(int32_t mode);
/* Epilogue form of the three ping-pong kernels above: 1 = straight from the accumulator registers (MFMA operands swapped, weight
 * rows permuted in LDS so that a lane owns 8 consecutive channels of a pixel: 16-byte stores, no LDS staging), 0 = staged through
 * LDS.  Same stored values either way (BatchNorm sums differ in fp32 summation order).  Negative: unchanged.  Returns the form. */
int         dsn_pp_dir(int32_t on);
/* The same for the ping-pong kernel-row weight-gradient kernel (csrc/wgrad.hip kind 5: 128 x 128 tiles x the three taps of a kernel
 * row per block; bf16, same-size 3x3 / s1 / p1 / d1, whole 128-channel tiles): 0 = never, 1 = default (16 x 16 patches cover >= 80 %
 * of the map, >= 12800 output pixels), 2 = every eligible layer.  Negative: unchanged.  Returns the mode after the update. */
int         dsn_wgrad_pp_mode(int32_t mode);
int         dsn_version(void);
const char* dsn_last_error(void);

/* ---- convolution (implicit GEMM on MFMA) -------------------------------------------------------------------
 * dsn_conv2d_fwd: y = act(conv(x, w) + bias) + residual
 *   replaces nn.Conv2d + folded BN + nn.SiLU of Conv.forward_fuse (common.py:55-56), the Bottleneck shortcut add
 *   (common.py:111), the raw Conv2d of RFB2 (common.py:515-524), FFM's attention 1x1s (+SiLU / Sigmoid,
 *   common.py:227-233), Detect.m[i] (yolo.py:258) and SegMaskPSP's classifier (yolo.py:182).
 *   In training it produces the pre-BN conv output (bias = NULL, act = NONE).
 *   bias, residual may be NULL.  w_packed: [Co][KH][KW][Ci], same dtype as x.
 * dsn_conv2d_dgrad: dx (+)= conv_transpose(dy, w)   -- autograd of the above w.r.t. x (ATen convolution_backward)
 *   w_packed: [Ci][KH][KW][Co].  p describes the FORWARD conv.
 * dsn_conv2d_wgrad: dw[Co][KH][KW][ci_pad] (fp32, packed-fwd layout) (+)= sum_pixels dy (x) x
 *   split-K over pixels with fp32 slabs in `workspace` (dsn_conv2d_wgrad_workspace_bytes), reduced in a fixed order.
 *   The ci_pad - Ci padding lanes of dw are never written (allocate dw zero-filled).
 */
int dsn_conv2d_fwd(const dsn_tensor* x, const void* w_packed, const float* bias, const dsn_tensor* residual,
                   const dsn_tensor* y, const dsn_conv_params* p, void* stream);
int dsn_conv2d_dgrad(const dsn_tensor* dy, const void* w_packed, const dsn_tensor* dx, const dsn_conv_params* p,
                     void* stream);
/* dx (+)= conv_transpose(dy, w) + residual (residual: dx's shape, stride-1 convs): the gradient of the Bottleneck shortcut
 * `x + cv2(cv1(x))` (common.py:111) added in the epilogue of cv1's input gradient. */
int dsn_conv2d_dgrad_res(const dsn_tensor* dy, const void* w_packed, const dsn_tensor* dx, const dsn_conv_params* p,
                         const dsn_tensor* residual, void* stream);
/* Input gradient of a 3x3 / stride 2 / pad 1 conv as one 2x2 stride-1 conv over dy + depth-to-space store (weights in the
 * dsn_pack_desc.out_dgrad_s2 layout).  Same result as dsn_conv2d_dgrad; needs 16-byte-aligned channel counts. */
int dsn_conv2d_dgrad_s2(const dsn_tensor* dy, const void* w_s2, const dsn_tensor* dx, const dsn_conv_params* p,
                        void* stream);
/* Training forward: plain conv whose epilogue ALSO writes per-tile BatchNorm partial sums (sum, sum of squares per output
 * channel, taken from the fp32 accumulators) -> no separate statistics pass over y.  stats: dsn_conv2d_stats_rows(N*Ho*Wo)
 * * 2 * Co floats; *rows_out = rows written; feed both to dsn_bn_finalize. */
int32_t dsn_conv2d_stats_rows(int64_t out_pixels);
int dsn_conv2d_fwd_stats(const dsn_tensor* x, const void* w_packed, const dsn_tensor* y, const dsn_conv_params* p,
                         float* stats, int32_t* rows_out, void* stream);
/* Training forward of Conv2d followed by BatchNorm2d (Conv.forward, common.py:49-53, train mode), statistics fused: same
 * epilogue reduction, but added with fp64 atomics into `acc` (dsn_bn_workspace_bytes(Co) bytes, ZERO ON ENTRY) -- no
 * partial rows and no finalize launch: dsn_bn_act_fwd_acc folds the accumulators in its prologue. */
int dsn_conv2d_fwd_bnacc(const dsn_tensor* x, const void* w_packed, const dsn_tensor* y, const dsn_conv_params* p,
                         void* acc, int64_t acc_bytes, void* stream);
int64_t dsn_conv2d_wgrad_workspace_bytes(const dsn_tensor* x, const dsn_tensor* dy, const dsn_conv_params* p,
                                         int32_t ci_pad);
/* oihw = 0: dw is packed [Co][KH][KW][ci_pad] (ci_pad >= x->c).
 * oihw = 1: dw is the parameter-gradient layout [Co][Ci][KH][KW] with Ci = ci_pad <= x->c (x may carry zero-padded
 *           channels); with p->accumulate the result is ADDED to dw (gradient accumulation straight into .grad). */
int dsn_conv2d_wgrad(const dsn_tensor* x, const dsn_tensor* dy, float* dw, int32_t ci_pad, int32_t oihw,
                     const dsn_conv_params* p, void* workspace, int64_t workspace_bytes, void* stream);
/* Queued form of the same computation.  dW feeds nothing but the optimizer, so a backward pass can PLAN every layer's
 * weight gradient as it goes (no launch) and RUN them all at its end in three launches (per-tap blocks, all-taps blocks,
 * slab reductions): the blocks of ~80 small layers share one grid instead of 150 launches that each under-fill the chip.
 *   dsn_conv2d_wgrad_plan: same arguments as dsn_conv2d_wgrad + job_out (HOST, dsn_wgrad_job_bytes() bytes).  x, dy, dw and
 *     workspace must stay valid and untouched until the run has executed.  DSN_EUNSUPPORTED: use dsn_conv2d_wgrad.
 *   dsn_conv2d_wgrad_plan_finish: jobs_host = n planned jobs, contiguous; assigns block ranges in place and fills
 *     launch_out[10] (doubles).  Copy jobs_host to the device AFTER this call.
 *   dsn_conv2d_wgrad_run: jobs_dev = device copy of the finished plan. */
int64_t dsn_wgrad_job_bytes(void);
int dsn_conv2d_wgrad_plan(const dsn_tensor* x, const dsn_tensor* dy, float* dw, int32_t ci_pad, int32_t oihw,
                          const dsn_conv_params* p, void* workspace, int64_t workspace_bytes, void* job_out);
int dsn_conv2d_wgrad_plan_finish(void* jobs_host, int32_t n, double* launch_out);
int dsn_conv2d_wgrad_run(const void* jobs_dev, int32_t n, const double* launch, void* stream);

/* weight packing: OIHW fp32 master weights (the state_dict layout, `...conv.weight [c2,c1,k,k]`) -> kernel layouts.
 * ci_pad >= ci zero-pads the input-channel axis (Focus: 12 -> 16 for 16-byte bf16 loads).
 * scale (may be NULL) multiplies output channel co: BN folding W' = diag(g/sqrt(var+eps)) W (torch_utils.py:196-216). */
int dsn_pack_weight_fwd(const float* w_oihw, const float* scale, void* out, int32_t dtype, int32_t co, int32_t ci,
                        int32_t kh, int32_t kw, int32_t ci_pad, void* stream);
/* co_pad >= co zero-pads the dgrad layout's K axis ([Ci][KH][KW][co_pad]): Detect's 33 output channels -> 40, so that the
 * input-gradient GEMM reads 16-byte vectors (its dy carries the same zero padding). */
int dsn_pack_weight_dgrad(const float* w_oihw, void* out, int32_t dtype, int32_t co, int32_t ci, int32_t kh,
                          int32_t kw, int32_t co_pad, void* stream);
/* One launch for ALL conv weights of a model (per optimizer step): descs/work live in device memory.
 * work = int32 pairs {tensor id, tile index}, tile index < dsn_pack_tiles(co, ci, kh, kw) (32 x 64 tiles of the [co][ci*kh*kw]
 * matrix).  ci_pad - ci padding lanes of out_fwd are never written: allocate it zero-filled. */
typedef struct {
    const void* w_oihw;   /* fp32 [co][ci][kh][kw] */
    void*       out_fwd;  /* [co][kh][kw][ci_pad] or NULL */
    void*       out_dgrad;/* [ci][kh][kw][co]     or NULL */
    void*       out_dgrad_s2; /* 3x3 stride-2 convs only, or NULL: [4*ci][2][2][co] for dsn_conv2d_dgrad_s2 (zero-filled by
                               * the caller: 7 of the 16 (sub-pixel, tap) blocks stay zero) */
    int32_t     co, ci, kh, kw, ci_pad, co_pad;   /* co_pad: row length of out_dgrad (0 or co = unpadded) */
} dsn_pack_desc;
int32_t dsn_pack_tiles(int32_t co, int32_t ci, int32_t kh, int32_t kw);
int dsn_pack_weights_multi(const dsn_pack_desc* descs_dev, const int32_t* work_dev, int32_t n_work, int32_t dtype,
                           void* stream);
/* ---- optimizer step (scripts/train.py:159-166,376: optim.SGD(momentum, nesterov=True), three parameter groups) ----------
 * ONE launch updates every parameter: descs (device array, sorted by first_chunk) name the fp32 param / grad / momentum
 * buffers; tensor i owns chunks [first_chunk, first_chunk + ceil(numel / dsn_sgd_chunk())).  hyper (DEVICE memory, 8 floats
 * per group: lr, momentum, dampening, weight_decay, nesterov, first_step, 0, 0) is read by the kernel, so a captured graph
 * follows learning-rate schedules.  Math = torch.optim.SGD: g' = g + wd*p; buf = first ? g' : m*buf + (1-damp)*g';
 * p -= lr * (nesterov ? g' + m*buf : buf). */
typedef struct {
    void*       param;
    const void* grad;
    void*       momentum_buf;
    int64_t     numel;
    int32_t     group, first_chunk;
} dsn_sgd_desc;
int32_t dsn_sgd_chunk(void);
int dsn_sgd_step(const dsn_sgd_desc* descs_dev, int32_t n_tensors, int32_t n_chunks, const float* hyper_dev, void* stream);

/* ---- ModelEMA.update (core/utils/torch_utils.py:330-342): ema = d*ema + (1-d)*model over every floating-point state_dict
 * entry in ONE launch; chunking as dsn_sgd_step (dsn_sgd_chunk() elements per block).  coef (DEVICE memory) = {(float)d,
 * (float)(1-d)} with d = decay*(1-exp(-updates/2000)) computed by the caller in double (torch_utils.py:323).  The three fp32
 * operations are rounded separately, as the reference's `v *= d; v += (1. - d) * m`: results are bit-identical. */
typedef struct {
    void*       ema;
    const void* model;
    int64_t     numel;
    int32_t     first_chunk, _pad;
} dsn_ema_desc;
int dsn_ema_step(const dsn_ema_desc* descs_dev, int32_t n_tensors, int32_t n_chunks, const float* coef_dev, void* stream);

/* dw (packed [Co][KH][KW][Ci_pad] fp32) -> OIHW fp32 gradient, grad (+)= dw */
int dsn_unpack_wgrad(const float* dw_packed, float* grad_oihw, int32_t co, int32_t ci, int32_t kh, int32_t kw,
                     int32_t ci_pad, int32_t accumulate, void* stream);

/* ---- BatchNorm (training) + activation ------------------------------------------------------------------------
 * replaces nn.BatchNorm2d in training mode + nn.SiLU inside Conv.forward (common.py:49-53), eps/momentum as set by
 * initialize_weights (torch_utils.py:164-165).
 * dsn_bn_stats: per-channel batch mean / biased variance of y; writes scale = g*rstd, shift = b - mean*scale,
 *   mean, rstd; updates running_mean/var in place ((1-m)*old + m*(mean, unbiased var)).
 *   workspace: dsn_bn_workspace_bytes(c) bytes of fp64 accumulators, zero-filled ONCE by the caller (dsn_bn_stats
 *   restores the zeros; one stream at a time).
 * dsn_bn_act_fwd_acc: training BN + act (+ shortcut) straight from the accumulators dsn_conv2d_fwd_bnacc filled; writes
 *   scale/shift/mean/rstd (saved for the backward pass) and updates the running statistics.  Leaves acc dirty.
 * dsn_bn_act_fwd:  z = act(y*scale + shift) + residual      (also the eval path of un-fused BN: RFB2 quirk Q3)
 * dsn_bn_act_bwd:  given dz, y: dy = BN/act backward, dgamma/dbeta (+)=.  workspace: dsn_bn_workspace_bytes(c) bytes,
 *   ZERO ON ENTRY, left dirty (callers hand out slices of an arena cleared once per step).
 * SyncBatchNorm (scripts/train.py:218-220, torch.nn.SyncBatchNorm semantics): the fp64 accumulators are plain sums, so the
 *   caller all-reduces (SUM) the accumulator buffer between producer and consumer -- between dsn_conv2d_fwd_bnacc and
 *   dsn_bn_act_fwd_acc in the forward pass, between dsn_bn_act_bwd_reduce and dsn_bn_act_bwd_apply in the backward pass --
 *   and passes `count` = the GLOBAL number of pixels (<= 0: this tensor's own) and, for the backward pass,
 *   `pgrad_scale` = 1 / world_size on the dgamma / dbeta contribution (they are summed over ranks by the gradient
 *   all-reduce afterwards).  dsn_bn_act_bwd = reduce + apply with local count and scale 1.
 */
int64_t dsn_bn_workspace_bytes(int32_t c);
int dsn_bn_stats(const dsn_tensor* y, const float* gamma, const float* beta, float* running_mean,
                 float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                 float* rstd, void* workspace, int64_t workspace_bytes, void* stream);
/* out[c] (+)= sum over n,h,w of t: bias gradient of a biased convolution (ATen convolution_backward's bias term for
 * yolo.py:118 Detect.m and the seg classifier yolo.py:186).  workspace: as dsn_bn_stats. */
int dsn_channel_sum(const dsn_tensor* t, float* out, int32_t accumulate, void* workspace, int64_t workspace_bytes,
                    void* stream);
int dsn_bn_finalize(const float* partial, int32_t rows, int32_t c, int64_t count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                    float* mean, float* rstd, void* stream);
int dsn_bn_act_fwd(const dsn_tensor* y, const float* scale, const float* shift, int32_t act,
                   const dsn_tensor* residual, const dsn_tensor* z, void* stream);
/* Two BatchNorm modules over ONE merged tensor (C3's cv2 and cv1 -- same input, common.py:143-145 -- run as a single
 * convolution with 2c_ output channels): channels [split_c, C) belong to the second module, whose vectors below are indexed
 * from 0.  NULL / split_c == 0: a single module. */
typedef struct {
    int32_t      split_c, _pad;
    const float* gamma;
    const float* beta;
    float*       running_mean;
    float*       running_var;
    float*       dgamma;
    float*       dbeta;
} dsn_bn_split;
int dsn_bn_act_fwd_acc(const dsn_tensor* y, const void* acc, int64_t acc_bytes, double count, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                       float* mean, float* rstd, int32_t act, const dsn_tensor* residual, const dsn_tensor* z,
                       const dsn_bn_split* second, void* stream);
int dsn_bn_act_bwd(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                   const float* mean, const float* rstd, int32_t act, const dsn_tensor* dy, float* dgamma,
                   float* dbeta, int32_t accumulate_param_grads, void* workspace, int64_t workspace_bytes,
                   void* stream);
int dsn_bn_act_bwd_reduce(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                          const float* mean, const float* rstd, int32_t act, void* workspace, int64_t workspace_bytes,
                          void* stream);
/* dsn_bn_act_bwd_reduce for a channel slice of a block, added at channel ch0 of a wider accumulator ([DSN_BN_NREP][2][acc_c]
 * doubles) -- the block's other channels received their sums from dsn_conv2d_dgrad_bnred. */
int dsn_bn_act_bwd_reduce_into(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                               const float* mean, const float* rstd, int32_t act, void* acc, int32_t acc_c, int32_t ch0,
                               void* stream);
int dsn_bn_act_bwd_apply(const dsn_tensor* dz, const dsn_tensor* y, const float* scale, const float* shift,
                         const float* mean, const float* rstd, int32_t act, const dsn_tensor* dy, float* dgamma,
                         float* dbeta, int32_t accumulate_param_grads, const void* workspace, int64_t workspace_bytes,
                         double count, float pgrad_scale, const dsn_bn_split* second, void* stream);
/* act backward without BN (conv -> act, quirk Q1 path and FFM attention): dy = dz * act'(y) */
int dsn_act_bwd(const dsn_tensor* dz, const dsn_tensor* y, int32_t act, const dsn_tensor* dy, void* stream);

/* ---- data-movement / pooling / resampling ---------------------------------------------------------------------
 * dsn_focus_s2d: Focus slicing + cat (common.py:626): NCHW fp32 image -> NHWC [N,H/2,W/2,c_pad], channel
 *   g*C + c for g = 0..3 = (row,col) parity (0,0),(1,0),(0,1),(1,1); channels >= 4C are zero.  Bit-exact copy.
 * dsn_maxpool_s1: stride-1 max pool, -inf padding (SPP, common.py:177,185); idx (int32, may be NULL in eval) gets
 *   the flat input pixel index of the arg-max (first maximum in row-major window order, as ATen).
 * dsn_upsample_nearest2x: nn.Upsample(None, 2, 'nearest') (yolov5s_seg.yaml:31,36).
 * dsn_bilinear_ac: bilinear, align_corners=True (yolo.py:170,174,183; common.py:610-613); src = dst*(in-1)/(out-1).
 *   out_nchw != 0 writes y as contiguous NCHW fp32 (the seg logits the caller sees, yolo.py:183).
 * dsn_adaptive_avgpool: nn.AdaptiveAvgPool2d(k) (common.py:226,597-600): bin i = [floor(i*H/k), ceil((i+1)*H/k)).
 * dsn_copy: strided NHWC copy / add (generic Concat fallback, gradient fan-in).
 */
int dsn_focus_s2d(const float* x_nchw, int32_t n, int32_t c, int32_t h, int32_t w, const dsn_tensor* y,
                  void* stream);
/* the same from the data loader's uint8 NCHW batch with `imgs.float() / 255.0` (scripts/train.py:329, val.py:213,
 * detect.py:129) folded in: y = Focus-slice((float)x / 255.0f), correctly rounded division (bit-exact vs ATen). */
int dsn_focus_s2d_u8(const uint8_t* x_nchw, int32_t n, int32_t c, int32_t h, int32_t w, const dsn_tensor* y,
                     void* stream);
/* letterbox (core/utils/mixed_datasets.py:722-752) with the loader's `transpose(2,0,1)[::-1]` (:576) optionally folded in:
 * src uint8 HWC h0 x w0 (3 channels) -> dst uint8 h x w; the window [top, top+new_h) x [left, left+new_w) holds the
 * INTER_LINEAR resize of src to new_h x new_w (a plain copy when the sizes are equal), the rest the border colour
 * (pad0..2 in source channel order).  chw_reversed = 0: HWC, source channel order (letterbox()'s return value);
 * 1: CHW with reversed channels (BGR -> RGB), the network's input layout.  The host computes the geometry (ratio, new_unpad,
 * dw, dh: desenet_amd.core.utils.augmentations.letterbox).  The resize restates OpenCV's 8-bit fixed-point INTER_LINEAR;
 * cv2 is a third-party dependency absent from the reference tree and this image: that part is "parity unpinned". */
int dsn_letterbox_u8(const uint8_t* src_hwc, int32_t h0, int32_t w0, uint8_t* dst, int32_t h, int32_t w, int32_t new_h,
                     int32_t new_w, int32_t top, int32_t left, int32_t pad0, int32_t pad1, int32_t pad2,
                     int32_t chw_reversed, void* stream);
/* PyramidPooling (common.py:597-615) runs four branches over one map; these do each stage of all branches in one launch
 * (+ one finalize launch for the two reductions).  ys / dys / dxs / xs: contiguous arrays of descriptors, at most 4.
 * dsn_adaptive_avgpool_multi: ys[j] = AdaptiveAvgPool2d(ys[j].h)(x); workspace: sum over j of
 *   dsn_window_reduce_workspace_bytes(n*h_j*w_j, x.h, x.c), jobs laid out back to back.
 * dsn_bilinear_ac_multi: ys[j] = bilinear(align_corners=True)(xs[j]), all ys of one size (channel slices of one buffer).
 * dsn_bilinear_ac_bwd_multi: dxs[j] (+)= its backward for sources of <= 64 pixels; workspace: sum over j of
 *   dsn_window_reduce_workspace_bytes(n*h_j*w_j, dys[j].h, c_j). */
int dsn_adaptive_avgpool_multi(const dsn_tensor* x, const dsn_tensor* ys, int32_t n_out, void* workspace,
                               int64_t workspace_bytes, void* stream);
int dsn_bilinear_ac_multi(const dsn_tensor* xs, const dsn_tensor* ys, int32_t n, void* stream);
int dsn_bilinear_ac_bwd_multi(const dsn_tensor* dys, const dsn_tensor* dxs, int32_t n, int32_t accumulate, void* workspace,
                              int64_t workspace_bytes, void* stream);
int dsn_maxpool_s1(const dsn_tensor* x, const dsn_tensor* y, int32_t* idx, int32_t k, void* stream);
/* the same pool at n_out <= 3 window sizes ks[i] of ONE input (SPP: 5, 9, 13) in one launch; ys: contiguous descriptors,
 * idxs (array of int32 pointers, or NULL / NULL entries in eval) */
int dsn_maxpool_s1_multi(const dsn_tensor* x, const dsn_tensor* ys, void* const* idxs, const int32_t* ks, int32_t n_out,
                         void* stream);
int dsn_maxpool_s1_bwd(const dsn_tensor* dy, const int32_t* idx, const dsn_tensor* dx, int32_t k,
                       int32_t accumulate, void* stream);
/* dx (+)= sum_i maxpool_s1_bwd(dys[i], idxs[i], ks[i]), i < n_src <= 3: the three SPP pools (common.py:177-185) in ONE pass
 * (dys: contiguous array of descriptors). */
int dsn_maxpool_s1_bwd_multi(const dsn_tensor* dys, const void* const* idxs, const int32_t* ks, int32_t n_src,
                             const dsn_tensor* dx, int32_t accumulate, void* stream);
int dsn_upsample_nearest2x(const dsn_tensor* x, const dsn_tensor* y, void* stream);
int dsn_upsample_nearest2x_bwd(const dsn_tensor* dy, const dsn_tensor* dx, int32_t accumulate, void* stream);
int dsn_bilinear_ac(const dsn_tensor* x, const dsn_tensor* y, int32_t out_nchw, void* stream);
int dsn_bilinear_ac_bwd(const dsn_tensor* dy, int32_t dy_nchw, const dsn_tensor* dx, int32_t accumulate,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* split reductions (few segments, large windows): workspace = dsn_window_reduce_workspace_bytes(segments, window rows, C);
 * segments = N*k*k (avgpool), N*Hi*Wi (bilinear bwd from a <= 64-pixel source; workspace may be NULL otherwise), N (FFM) */
int64_t dsn_window_reduce_workspace_bytes(int32_t n_segments, int32_t rows, int32_t c);
int dsn_adaptive_avgpool(const dsn_tensor* x, const dsn_tensor* y, void* workspace, int64_t workspace_bytes, void* stream);
int dsn_adaptive_avgpool_bwd(const dsn_tensor* dy, const dsn_tensor* dx, int32_t accumulate, void* stream);
/* dx (+)= sum_i adaptive_avgpool_bwd(dys[i]), i < n_src <= 4: the four PyramidPooling grids (common.py:597-613) in ONE pass */
int dsn_adaptive_avgpool_bwd_multi(const dsn_tensor* dys, int32_t n_src, const dsn_tensor* dx, int32_t accumulate,
                                   void* stream);
int dsn_copy(const dsn_tensor* x, const dsn_tensor* y, int32_t accumulate, void* stream);

/* ---- FFM channel attention (common.py:236-242): out = feat*att + feat, att: [N,1,1,C] ------------------------ */
int dsn_ffm_scale(const dsn_tensor* feat, const dsn_tensor* att, const dsn_tensor* out, void* stream);
/* dfeat (+)= dout*(1+att);  datt[n,c] = sum_hw dout*feat */
int dsn_ffm_scale_bwd(const dsn_tensor* dout, const dsn_tensor* feat, const dsn_tensor* att,
                      const dsn_tensor* dfeat, const dsn_tensor* datt, int32_t accumulate, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* ---- Detect head (yolo.py:255-277) ------------------------------------------------------------------------------
 * t: conv output of level i, NHWC [N,ny,nx,na*no].  raw: contiguous fp32 [N,na,ny,nx,no] (the training output and
 * second eval output).  pred (may be NULL in training): fp32 [N, total, no] rows [row_offset, row_offset+na*ny*nx):
 * sigmoid; xy = (2s - .5 + grid)*stride with grid[...,0]=x, [...,1]=y (yolo.py:279-282); wh = (2s)^2*anchor_px.
 * dsn_detect_raw_bwd: d(conv output) from d(raw) (the permute's transpose).
 */
int dsn_detect_decode(const dsn_tensor* t, float* raw, float* pred, int64_t pred_total_rows, int64_t row_offset,
                      int32_t na, int32_t no, float stride, const float* anchors_px /* [na][2] device */,
                      void* stream);
/* zero_pad_to (0 or in [c, ldc]): channels c .. zero_pad_to-1 of every pixel row are written as zeros (row padding that
 * lets the consumers read 16-byte vectors; never set it on a channel SLICE of a wider tensor). */
int dsn_detect_raw_bwd(const float* draw, const dsn_tensor* dt, int32_t na, int32_t no, int32_t zero_pad_to, void* stream);
/* All Detect levels in one launch each way (the reference loops over its three heads, yolo.py:258-276): ts / dts are
 * contiguous descriptor arrays (nl <= 4), raws / draws arrays of fp32 [N][na][ny][nx][no] buffers, row_offs[l] the first row
 * of level l in pred (NULL in training), strides[l] host floats, anchors_px DEVICE [nl][na][2].  The backward also adds the
 * heads' bias gradients (per-channel sums of draws[l]) into bias_grads[l] (NULL: skip); workspace: nl * 512 * na*no floats
 * of per-block partial rows (no initialisation needed), folded in a fixed order. */
int dsn_detect_decode_multi(const dsn_tensor* ts, float* const* raws, int32_t nl, float* pred, int64_t pred_rows,
                            const int64_t* row_offs, int32_t na, int32_t no, const float* strides, const float* anchors_px,
                            void* stream);
/* Training forward of ALL Detect heads in one launch (yolo.py:258-276: per level a biased 1x1 convolution to na*no channels,
 * then view + permute to [N][na][ny][nx][no]): xs[l] bf16 head inputs (channels % 8 == 0), ws[l] packed forward weights
 * [na*no][K] (dsn_pack_weight_fwd), biases[l] fp32 [na*no] (the array or an entry may be NULL), raws[l] fp32 outputs.  The result
 * is rounded to bf16 on the way, as the head convolution's own output is.  na*no <= 64, bf16 only: DSN_EUNSUPPORTED (nothing
 * launched) otherwise -- callers then run dsn_conv2d_fwd per level + dsn_detect_decode_multi.  _supported: the same predicate for
 * host-side planning (k_min_multiple: any common divisor of the levels' input channel counts). */
int dsn_detect_head_fwd_supported(int32_t dtype, int32_t na, int32_t no, int32_t k_min_multiple);
int dsn_detect_head_fwd_multi(const dsn_tensor* xs, const void* const* ws, const float* const* biases, float* const* raws,
                              int32_t nl, int32_t na, int32_t no, void* stream);
int dsn_detect_raw_bwd_multi(const float* const* draws, const dsn_tensor* dts, int32_t nl, int32_t na, int32_t no,
                             const int32_t* zero_pad_to, float* const* bias_grads, void* workspace, int64_t workspace_bytes,
                             void* stream);

/* ---- NMS (general.py:659-750 + torchvision.ops.nms) ------------------------------------------------------------
 * pred: fp32 [bs, n, 5+nc].  For every image: candidate filter (obj > conf), conf = obj*cls, best-class or
 * multi-label expansion, optional class filter, cap at 30000 by descending conf, class-offset boxes (4096*cls unless
 * agnostic), greedy suppression (IoU > iou_thres, stable descending-score order), first max_det.
 * out: fp32 [bs, max_det, 6] rows [x1,y1,x2,y2,conf,cls]; out_count: int32 [bs].
 * classes_mask: bit c set = keep class c (0 = keep all).  workspace from dsn_nms_workspace_bytes.
 */
int64_t dsn_nms_workspace_bytes(int32_t bs, int32_t n, int32_t nc, int32_t multi_label);
int dsn_nms(const float* pred, int32_t bs, int32_t n, int32_t nc, float conf_thres, float iou_thres,
            int32_t multi_label, int32_t agnostic, uint64_t classes_mask, int32_t max_det, float* out,
            int32_t* out_count, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- training losses (core/utils/loss.py:91-243, core/utils/metrics.py:202-244); forward AND gradient w.r.t. the network
 * outputs in one call, no host synchronisation (hipGraph-capturable) -------------------------------------------------------
 * dsn_det_loss: ComputeLoss.__call__ + build_targets + CIoU.  p[i] / dp[i]: fp32 [bs,na,ny_i,nx_i,5+nc] (raw Detect outputs
 *   and d(out[0])/d(p[i])); targets: device fp32 [nt,6] (image, class, x, y, w, h normalised); anchors: HOST [nl][na][2] in
 *   grid units; balance: HOST [nl]; out: device [4] = {(lbox+lobj+lcls)*bs, lbox, lobj, lcls}.
 * dsn_seg_ce: nn.CrossEntropyLoss(ignore_index), mean; logits contiguous NCHW fp32, target int64 [N,H,W];
 *   out: device [2] = {loss, 1/valid}; dlogits (may be NULL) = d(loss)/d(logits). */
int64_t dsn_det_loss_workspace_bytes(int32_t nl, int32_t na, int32_t nt, int32_t nc, int64_t max_cells);
int dsn_det_loss(const float* const* p, float* const* dp, const int32_t* ny, const int32_t* nx, int32_t nl, int32_t bs,
                 int32_t na, int32_t nc, const float* targets, int32_t nt, const float* anchors, const float* balance,
                 float h_box, float h_obj, float h_cls, float cls_pw, float obj_pw, float anchor_t, float cp, float cn,
                 float* out, void* workspace, int64_t workspace_bytes, void* stream);
/* dsn_det_loss with the options scripts/train.py leaves off (core/utils/loss.py:91-168):
 *   fl_gamma > 0   : both BCE criteria wrapped in FocalLoss (loss.py:36-61,106-110; alpha 0.25, element-wise, then the mean);
 *   balance_dev    : [nl] DEVICE floats used as the per-level objectness weights instead of the host array `balance` (may be NULL
 *                    then); with autobalance != 0 they are updated in place after the losses are formed,
 *                    b_i <- 0.9999 b_i + 0.0001 / mean objectness loss of level i, then all divided by b_ssi (loss.py:158-164; ssi =
 *                    index of the stride-16 level, loss.py:113).  The reference reads each level's loss back with `.item()`; here
 *                    the state stays on the device and the step remains graph-capturable. */
int dsn_det_loss_opt(const float* const* p, float* const* dp, const int32_t* ny, const int32_t* nx, int32_t nl, int32_t bs,
                     int32_t na, int32_t nc, const float* targets, int32_t nt, const float* anchors, const float* balance,
                     float h_box, float h_obj, float h_cls, float cls_pw, float obj_pw, float anchor_t, float cp, float cn,
                     float fl_gamma, float* balance_dev, int32_t autobalance, int32_t ssi, float* out, void* workspace,
                     int64_t workspace_bytes, void* stream);
int64_t dsn_seg_ce_workspace_bytes(void);
int dsn_seg_ce(const float* logits, const int64_t* target, int32_t n, int32_t c, int32_t h, int32_t w,
               int32_t ignore_index, float* out, float* dlogits, void* workspace, int64_t workspace_bytes, void* stream);
/* nn.Upsample(scale_factor, 'bilinear', align_corners=True) (yolo.py:183) + CrossEntropyLoss(ignore_index) (loss.py:242-243)
 * fused, loss AND gradient, from the LOW-resolution classifier output: the N x C x H x W fp32 logits (and their gradient) are never
 * formed.  logits / dlogits: NHWC tensors of the same shape (dlogits' row padding is zeroed); target [n, H, W] int64;
 * out[0] = mean CE, out[1] = 1 / valid pixels; dlogits = gain * d out[0] / d logits.  workspace: dsn_seg_ce_up_workspace_bytes
 * bytes whose first n*h*w*c floats are ZERO on entry (restored on exit).  DSN_EUNSUPPORTED unless c == 2, w <= 160, scale >= 3:
 * use dsn_bilinear_ac + dsn_seg_ce + dsn_bilinear_ac_bwd. */
int64_t dsn_seg_ce_up_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t c, int32_t H);
int dsn_seg_ce_up(const dsn_tensor* logits, const int64_t* target, int32_t H, int32_t W, int32_t ignore_index, float gain,
                  float* out, const dsn_tensor* dlogits, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- evaluation arithmetic (scripts/val.py:101-122; core/utils/metrics.py:247-269,350-388) ----------------------------
 * dsn_box_iou: out[n][m] = IoU of xyxy boxes, the reference's operation order (bit-exact matching in process_batch).
 * dsn_seg_eval_counts: over NCHW fp32 logits and int64 targets: out (int64[2 + 3*(nclass-1)]) = {correct, labelled,
 *   intersection[b], prediction area[b], label area[b]} with predict = first arg-max over classes and the nclass-1 bins of
 *   np.histogram(range=(1, nclass)) -- what batch_pix_accuracy / batch_intersection_union count on the host. */
int dsn_box_iou(const float* boxes1, int32_t n, const float* boxes2, int32_t m, float* out, void* stream);
int dsn_seg_eval_counts(const float* logits, const int64_t* target, int32_t n, int32_t c, int32_t h, int32_t w,
                        int32_t nclass, int64_t* out, void* stream);
/* dsn_resize_bilinear_nchw: F.interpolate(pred, size, mode='bilinear', align_corners=...) on contiguous NCHW fp32 planes
 *   (seg_validation's resize of the logits to the label size, scripts/val.py:47: align_corners=False), ATen's arithmetic.
 * dsn_seg_argmax_nearest: segoutput_to_target (core/utils/plots.py:222-229): first arg-max over classes as float, resized
 *   with ATen's legacy 'nearest' index rule; out [n][ho][wo] fp32. */
int dsn_resize_bilinear_nchw(const float* x, float* y, int64_t planes, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                             int32_t align_corners, void* stream);
int dsn_seg_argmax_nearest(const float* logits, float* out, int32_t n, int32_t c, int32_t h, int32_t w, int32_t ho,
                           int32_t wo, void* stream);

/* ---- misc ---------------------------------------------------------------------------------------------------- */
/* dst[i] = (dtype) src[i]  (flat fp32 master -> bf16 copy) */
int dsn_cast(const float* src, void* dst, int32_t dtype, int64_t n, void* stream);

/* p[0 .. n_words) = value (32-bit words; p 4-byte aligned): `optimizer.zero_grad()` on the flat gradient buffer
 * (scripts/train.py:378) and the per-step accumulator clears, as a kernel (no memset graph nodes). */
int dsn_fill32(void* p, uint32_t value, int64_t n_words, void* stream);
/* Up to DSN_COPY_MAXSEG flat device-to-device copies in one launch; bytes [copy_bytes, total_bytes) of each destination are
 * zero-filled.  Stages a training batch (scripts/train.py:329 imgs, :352-354 targets / masks) into the static buffers a
 * captured step reads.  Sizes and pointers: multiples of 4 bytes (16-byte vectors are used where everything is 16-aligned). */
#define DSN_COPY_MAXSEG 4
typedef struct {
    void*       dst;
    const void* src;
    int64_t     copy_bytes;
    int64_t     total_bytes;
} dsn_copy_seg;
int dsn_copy_multi(const dsn_copy_seg* segs_host, int32_t n, void* stream);
/* p[i] += value for n int64 elements: BatchNorm's `num_batches_tracked += 1` for every layer in one launch. */
int dsn_add_i64(void* p, int64_t n, int64_t value, void* stream);

/* ---- the end-of-forward BatchNorm finalisation and the BatchNorm + activation pass from accumulators ----------------------------
 * dsn_bn_finalize_multi: for n BatchNorm modules whose sums are complete, write scale = g*rstd, shift = b - mean*scale, mean, rstd
 *   (saved for the backward pass) and update the running statistics (torch_utils.py:164-165 momentum 0.03, unbiased variance) --
 *   ONE launch per 40 modules at the end of the forward pass instead of one per layer.  entries: HOST array. */
typedef struct {
    const void*  acc;            /* [DSN_BN_NREP][2][acc_c] doubles */
    int32_t      acc_c, ch0;     /* accumulator channel count, first channel of this module */
    int32_t      n, _pad;        /* channels of this module */
    double       count;
    const float* gamma;
    const float* beta;
    float*       running_mean;   /* may be NULL (with running_var) */
    float*       running_var;
    float*       scale;          /* outputs, n floats each */
    float*       shift;
    float*       mean;
    float*       rstd;
    float        momentum, eps;
} dsn_bn_final;
int dsn_bn_finalize_multi(const dsn_bn_final* entries_host, int32_t n, void* stream);
/* z = act(x*scale + shift) per segment of lx (NULL: copy) [+ the same of `residual` under lres: a Bottleneck shortcut,
 * common.py:111, rounded to the storage type before the add exactly as a materialised shortcut would be].  No side effects on
 * BatchNorm state.  DSN_EUNSUPPORTED without 16-byte channel vectors or above 1024 channels. */
int dsn_lazy_materialize(const dsn_tensor* x, const dsn_lazy_in* lx, const dsn_tensor* residual, const dsn_lazy_in* lres,
                         const dsn_tensor* z, void* stream);

/* ---- live profiler (bench.py roofline): HIP events recorded on the launch stream around the hot kernels ------------
 * dsn_profile_enable(1) starts recording, (0) stops; dsn_profile_collect waits for the recorded events and returns, per
 * kernel id, {launches, total_ms, total_algorithmic_flops, total_algorithmic_bytes}. */
int         dsn_profile_enable(int32_t on);
int         dsn_profile_collect(double* out /* [kernel_count][4] */, int32_t kernel_count);
int32_t     dsn_profile_kernel_count(void);
const char* dsn_profile_kernel_name(int32_t kid);
/* Everything recorded since dsn_profile_enable(1), aggregated per (label, layer) as text, one line per pair:
 * "label\tlayer\tlaunches\ttotal_ms\ttotal_flops\ttotal_bytes\n".  label = "<kernel symbol family as rocprofv3 --kernel-trace
 * prints it>/<dtype>/<tile>/<fwd|dgrad>" for the convolution launches (the name of the legacy id otherwise), layer = shape key
 * of the launch ("k3s1d1 64->64 @8x80x80"; empty for the non-convolution ids).  Returns the bytes the dump needs; when they
 * exceed `cap` nothing is written and the records are kept (call again with a larger buffer).  < 0: -hipError_t. */
int64_t     dsn_profile_dump(char* out, int64_t cap);

/* ---- PyramidPooling branches in one launch each way (common.py:588-615) ------------------------------------------------------
 * Branch j: x_j [P_j pixels][C] (the adaptive-average-pooled map, contiguous NHWC) -> 1x1 convolution with the forward-packed
 * weights w_j [Co][C] -> training-mode BatchNorm over the P_j pixels (has_bn = 0 on 1x1 maps: quirk Q1, common.py:53) -> act.
 *   dsn_pp_stages_fwd: writes z_j (raw convolution output, rounded to the activation dtype), y_j = act(z_j * scale + shift),
 *     stats_j = [scale | shift | mean | rstd] (Co floats each) and updates running_mean / running_var (momentum, unbiased variance).
 *   dsn_pp_stages_bwd: from dy_j, z_j, x_j, w_j, stats_j: dx_j [P_j][C], dgamma_j / dbeta_j and dw_j [Co][C] fp32 (accumulate = 1:
 *     added to what they hold; 0: overwritten); null dgamma / dbeta / dw pointers are skipped.
 * bf16 with C % 8 == 0 and Co % 8 == 0, and every branch must fit the 160 KB of LDS of one CU (dsn_pp_stages_supported; DeSeNet-s:
 * 288 pixels x 128 -> 32 channels does): otherwise DSN_EUNSUPPORTED and the caller runs the branches as ordinary layers. */
#define DSN_PP_MAXSTAGE 4
typedef struct {
    const void* x;
    const void* w;
    void* z;
    void* y;                 /* forward */
    float* stats;            /* [4][Co]; unused when has_bn == 0 */
    const float *gamma, *beta;
    float *running_mean, *running_var;
    const void* dy;          /* backward */
    void* dx;
    float *dgamma, *dbeta, *dw;
    int64_t zld, yld, dyld;  /* row strides of z, y, dy in elements */
    int32_t P, has_bn;
} dsn_pp_stage;
typedef struct {
    dsn_pp_stage s[DSN_PP_MAXSTAGE];
    int32_t nstage, C, Co, dtype, act, accumulate;
    float momentum, eps;
} dsn_pp_args;
int dsn_pp_stages_supported(int32_t max_pixels, int32_t c, int32_t co, int32_t dtype);
int dsn_pp_stages_fwd(const dsn_pp_args* a, void* stream);
int dsn_pp_stages_bwd(const dsn_pp_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DESENET_HIP_H */
