/* unetk -- C ABI of the MI355X (gfx950) U-Net kernel library (libunetk.so).
 *
 * The reference (BoySun045/Pytorch-UNet) has no FFI: its hot path is a tree of stock torch.nn
 * modules and every FLOP runs inside ATen.  This header is the boundary the build inserts
 * BENEATH that nn.Module surface: each entry point replaces the ATen op that one
 * reference line dispatches, cited per function as file:line into the reference.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator in the shipped binding); the library allocates nothing.
 *   - activations are NHWC ("channels_last"): element (n,y,x,c) of a tensor with pixel pitch
 *     `ld` lives at ((n*H + y)*W + x)*ld + c.  `ld` >= C lets a tensor be a channel slice of a
 *     wider buffer (the decoder's concat buffer, unet_parts.py:67, is never materialised by a copy).
 *   - dtype: UNETK_BF16 / UNETK_F16 (16-bit storage, fp32 accumulate; the same kernels over the element type) or
 *     UNETK_F32 (exact fp32 MFMA).
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, and
 *     return 0 or a negative UNETK_ERR_* code; no exceptions cross the ABI.
 */
#ifndef UNETK_H
#define UNETK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UNETK_BF16 0
#define UNETK_F32 1
#define UNETK_F16 2

#define UNETK_OK 0
#define UNETK_ERR_SHAPE (-1)   /* unsupported / inconsistent shape, pitch or channel count */
#define UNETK_ERR_DTYPE (-2)
#define UNETK_ERR_ALIGN (-3)   /* pointer or pitch not aligned to 16 bytes where required */
#define UNETK_ERR_LAUNCH (-4)  /* hipGetLastError() after launch */
#define UNETK_ERR_ARG (-5)
#define UNETK_ERR_TOO_LARGE (-6) /* problem exceeds the fixed capacity of a no-allocation entry point */

/* Library / ABI version and the gfx target it was built for ("gfx950"). */
int unetk_version(void);
const char* unetk_arch(void);

/* ---- weight packing ------------------------------------------------------------------------
 * Repack a PyTorch-layout fp32 parameter into the MFMA B-fragment order the conv kernels stream
 * through LDS.  `kind`: 0 = conv3x3 fprop   (src [Co][Ci][3][3], unet_parts.py:16,19)
 *                       1 = conv3x3 dgrad   (same src; taps rotated 180 deg, Ci/Co swapped)
 *                       2 = convT2x2 fwd    (src [Ci][Co][2][2], unet_parts.py:53)
 *                       3 = convT2x2 dgrad  (same src)
 * unetk_packed_weight_bytes() returns the size of `dst` (or a negative error). */
int64_t unetk_packed_weight_bytes(int kind, int c_out, int c_in, int dtype);
int unetk_pack_weight(int kind, const float* src, void* dst, int c_out, int c_in, int dtype, void* stream);

/* All weights of a network in one launch.  unetk_pack_plan_build() translates n_items (parameter, packed buffer,
 * kind, channels, dtype) records -- the arguments of n_items unetk_pack_weight() calls -- into an opaque plan in a
 * HOST buffer of unetk_pack_plan_bytes(n_items) bytes and returns the total number of work units (> 0) or an
 * error code; the caller copies the plan to the device once and calls unetk_pack_plan_run() after every optimizer
 * step.  The plan stays valid while the pointers it names do. */
typedef struct {
  const float* src; /* fp32 parameter, PyTorch layout (device) */
  void* dst;        /* packed buffer of unetk_packed_weight_bytes() bytes (device, 16-byte aligned) */
  int kind, c_out, c_in, dtype;
} unetk_pack_item;
int64_t unetk_pack_plan_bytes(int n_items);
int64_t unetk_pack_plan_build(const unetk_pack_item* items, int n_items, void* plan_host, int64_t plan_bytes);
int unetk_pack_plan_run(const void* plan_dev, int n_items, int64_t total_units, void* stream);

/* ---- conv3x3, stride 1, pad 1, no bias: unet_parts.py:16,19 (aten::conv2d) -------------------
 * y[n,y,x,co] = sum_{kh,kw,ci} x[n,y+kh-1,x+kw-1,ci] * w[co,ci,kh,kw]      (zero padding)
 * `wpk` = unetk_pack_weight(kind 0).  c_in % 64 == 0 (bf16: % 32), c_out % 64 == 0.
 * If `stats` != NULL it must hold UNETK_STAT_REPLICAS*c_out*2 doubles, zeroed by the caller; the
 * kernel adds per-channel sum and sum-of-squares of the (rounded) outputs: the batch statistics
 * BatchNorm2d needs (unet_parts.py:17,20), fused into the conv epilogue. */
#define UNETK_STAT_REPLICAS 32
int unetk_conv3x3_fprop(const void* x, int ldx, int x_choff, const void* wpk, void* y, int ldy, int y_choff,
                        double* stats, int n, int h, int w, int c_in, int c_out, int dtype, void* stream);
/* The eval()-mode chain of unet_parts.py:16-18 / 19-21 in one kernel (evaluate.py:10, predict.py:20 put the
 * net in eval mode): y = relu(scale[c] * conv3x3(x) + shift[c]) with scale/shift from unetk_bn_eval_params
 * (running statistics).  The affine + ReLU run on the fp32 accumulators in the epilogue; no pre-BN tensor is
 * written.  `pool` != NULL ([n][h/2][w/2][c_out], pitch ldp): also the MaxPool2d(2) of y (unet_parts.py:34, floor mode),
 * taken from the registers of the same epilogue.  bf16 only (UNETK_ERR_DTYPE otherwise: use unetk_conv3x3_fprop +
 * unetk_bn_relu_apply). */
int unetk_conv3x3_fprop_bn_relu(const void* x, int ldx, int x_choff, const void* wpk, const float* scale,
                                const float* shift, void* y, int ldy, int y_choff, void* pool, int ldp, int n, int h,
                                int w, int c_in, int c_out, int dtype, void* stream);
/* dgrad of the same layer (autograd of unet_parts.py:16,19): dx = conv3x3(dy, rot180(w)^T).
 * `wpk` = unetk_pack_weight(kind 1); c_in/c_out are those of the FORWARD layer. */
int unetk_conv3x3_dgrad(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx,
                        int dx_choff, int n, int h, int w, int c_in, int c_out, int dtype, void* stream);

/* wgrad of the same layer (autograd of unet_parts.py:16,19): dw[co,ci,kh,kw] = sum x[n,y+kh-1,x+kw-1,ci]*dy[n,y,x,co],
 * written (accumulate=0) or added (1) into the fp32 PyTorch-layout gradient `dw` [Co][Ci][3][3].  Split-K partial
 * sums go through `workspace` (size from unetk_wgrad_workspace_bytes, kind 0 = conv3x3, 1 = convT2x2) and are
 * added in a fixed order: bitwise reproducible.  bf16 activations (bf16 MFMA, transposed LDS reads) or fp32
 * (exact-fp32 MFMA). */
int64_t unetk_wgrad_workspace_bytes(int kind, int n, int h, int w, int c_in, int c_out);
int unetk_conv3x3_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                        int64_t workspace_bytes, int n, int h, int w, int c_in, int c_out, int accumulate, int dtype,
                        void* stream);

/* ---- first layer: Conv2d(n_channels<=4 -> c_out, 3x3, pad 1, no bias), unet_model.py:13 -> unet_parts.py:16 ----
 * Reads the fp32 NHWC image directly (train.py:97 hands over fp32 channels_last) and the UNPACKED parameter. */
int unetk_conv3x3_first_fprop(const float* x, int ldx, const float* w, void* y, int ldy, double* stats, int n, int h,
                              int wd, int c_in, int c_out, int dtype, void* stream);
/* Training-mode forward of the layer with its BatchNorm2d + ReLU (unet_parts.py:16-18 at unet_model.py:13) without reading
 * the 64-channel output back: call unetk_conv3x3_first_fprop with y = NULL (statistics only: the sums are those of the
 * output rounded to `dtype`, nothing is stored), unetk_bn_train_finalize, then this entry point, which computes the
 * layer again and stores y (saved for backward) together with act = relu(scale[c] * y + shift[c]) formed from the
 * ROUNDED y -- the same bits as unetk_conv3x3_first_fprop + unetk_bn_relu_apply. */
int unetk_conv3x3_first_fprop_bn_relu_train(const float* x, int ldx, const float* w, const float* scale,
                                            const float* shift, void* y, int ldy, void* act, int lda, int n, int h, int wd,
                                            int c_in, int c_out, int dtype, void* stream);
/* Same layer under net.eval() (evaluate.py:10, predict.py:20): y = relu(scale[c] * conv + shift[c]), the eval-mode
 * BatchNorm2d + ReLU of unet_parts.py:17-18 folded into the store (scale/shift from unetk_bn_eval_params). */
int unetk_conv3x3_first_fprop_bn_relu(const float* x, int ldx, const float* w, const float* scale, const float* shift,
                                      void* y, int ldy, int n, int h, int wd, int c_in, int c_out, int dtype,
                                      void* stream);
int64_t unetk_conv3x3_first_wgrad_workspace_bytes(int n, int h, int w, int c_out);
int unetk_conv3x3_first_wgrad(const float* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                              int64_t workspace_bytes, int n, int h, int w, int c_in, int c_out, int dtype,
                              void* stream);

/* The same weight gradient together with the layer's BatchNorm2d+ReLU backward (autograd of unet_parts.py:16-18 at the
 * network's first layer, whose input -- the image of train.py:97 -- needs no gradient): the gradient w.r.t. the
 * convolution's output has no reader but this kernel, so it is formed on the fly and never stored,
 *   dy = coef[0][c] * (scale[c]*z + shift[c] > 0 ? dout : 0) - coef[1][c] - coef[2][c] * (z - mean[c]),
 * the expression and rounding of unetk_bn_relu_bwd_apply: dw equals unetk_bn_relu_bwd_apply followed by
 * unetk_conv3x3_first_wgrad bit for bit.  dout: gradient w.r.t. the layer's activation (pitch lddo); z: the
 * convolution's saved output (pitch ldz); coef: [3][c_out] from unetk_bn_relu_bwd_finalize.  bf16 / fp16 only
 * (UNETK_ERR_DTYPE for fp32: run the two entry points). */
int unetk_conv3x3_first_wgrad_bn(const float* x, int ldx, const void* dout, int lddo, const void* z, int ldz,
                                 const float* scale, const float* shift, const float* mean, const float* coef,
                                 float* dw, void* workspace, int64_t workspace_bytes, int n, int h, int w, int c_in,
                                 int c_out, int dtype, void* stream);

/* Gradient w.r.t. the network INPUT of that layer (autograd of unet_parts.py:16 when the image requires a gradient --
 * the reference's train loop never does, saliency / adversarial uses do): dx fp32 [n][h][wd][c_in] (pitch lddx),
 * dx[q][ci] = sum_{tap,co} dy[q - (tap - 1)][co] * w[co][ci][tap]; w is the UNPACKED parameter; c_out <= 512. */
int unetk_conv3x3_first_dgrad(const void* dy, int lddy, const float* w, float* dx, int lddx, int n, int h, int wd,
                              int c_in, int c_out, int dtype, void* stream);

/* ---- ConvTranspose2d(c_in -> c_out, k2, s2) + bias, fused with F.pad and the concat placement:
 * unet_parts.py:53 (aten::conv_transpose2d), :59-63 (constant_pad_nd), :67 (cat).
 * x [n,h,w,c_in]; y = channel slice [y_choff, y_choff+c_out) of the [n,hs,ws,ldy] concat buffer;
 * y[n, pad_t+2i+a, pad_l+2j+b, co] = bias[co] + sum_ci x[n,i,j,ci]*w[ci,co,a,b].  The pad border itself is
 * zeroed by unetk_zero_pad_border.  wpk = unetk_pack_weight(kind 2 / 3). */
int unetk_convT2x2_fwd(const void* x, int ldx, const void* wpk, const float* bias, void* y, int ldy, int y_choff,
                       int n, int h, int w, int c_in, int c_out, int hs, int ws, int pad_t, int pad_l, int dtype,
                       void* stream);
int unetk_convT2x2_dgrad(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int n, int h,
                         int w, int c_in, int c_out, int hs, int ws, int pad_t, int pad_l, int dtype, void* stream);
/* dw [c_in][c_out][2][2] fp32; the bias gradient is a plain per-channel sum over the up-sampled window: by
 * unetk_channel_sum, or without a pass of its own by unetk_convT2x2_bias_grad_from_sums (below). */
int unetk_convT2x2_wgrad(const void* x, int ldx, const void* dy, int lddy, int dy_choff, float* dw, void* workspace,
                         int64_t workspace_bytes, int n, int h, int w, int c_in, int c_out, int hs, int ws, int pad_t,
                         int pad_l, int accumulate, int dtype, void* stream);
int unetk_zero_pad_border(void* buf, int ld, int n, int hs, int ws, int c, int pad_t, int pad_l, int h_up, int w_up,
                          int dtype, void* stream);
/* The same bias gradient without a pass over the gradient tensor: the decoder's first conv3x3 dgrad WRITES that tensor
 * (the concat gradient), and unetk_conv3x3_dgrad_sums -- dgrad with the fused per-channel sums of the conv epilogue,
 * `sums` [UNETK_STAT_REPLICAS][c_in][2] doubles zeroed by the caller -- hands over the sums of all its pixels;
 * unetk_convT2x2_bias_grad_from_sums subtracts the F.pad border (a few rows/columns, `dup` = the up-sampled channel
 * slice, `scratch` c_up doubles) and writes db [c_up]. */
int unetk_conv3x3_dgrad_sums(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int dx_choff,
                             double* sums, int n, int h, int w, int c_in, int c_out, int dtype, void* stream);
/* dgrad of a DoubleConv's SECOND convolution fused with the first pass of the BatchNorm backward of its FIRST layer
 * (unet_parts.py:15-21: conv -> BN -> ReLU -> conv -> ...; autograd runs ConvolutionBackward, ThresholdBackward and the
 * reduction half of NativeBatchNormBackward as separate passes): besides dx it accumulates, from the value it stores
 * and one read of that layer's pre-BN tensor `z_prev` [N][H][W][c_in] (pitch ldz), the two per-channel sums that
 * unetk_bn_relu_bwd_reduce would form from dx and z_prev -- `sums` [UNETK_STAT_REPLICAS][c_in][2] doubles zeroed by the
 * caller; scale/shift/mean/invstd: that layer's BatchNorm state [c_in].  Shapes whose kernel has no such epilogue run
 * dgrad followed by unetk_bn_relu_bwd_reduce; the result is the same either way. */
/* 1 when the call below folds the reduction into the dgrad kernel for these channel counts, 0 when it runs the two
 * kernels one after the other (a caller that times or schedules the passes itself may then prefer to call them). */
int unetk_conv3x3_dgrad_bn_reduce_is_fused(int c_in, int c_out, int dtype);
int unetk_conv3x3_dgrad_bn_reduce(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx,
                                  int dx_choff, const void* z_prev, int ldz, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, double* sums, int n, int h, int w, int c_in,
                                  int c_out, int dtype, void* stream);
/* The same fusion for the Up block's ConvTranspose2d (unet_parts.py:53, :65 x1 = self.up(x1)): its input gradient IS the
 * gradient of the activation of the conv -> BN -> ReLU layer below (the previous block's last layer, unet_parts.py:20-21),
 * so the dgrad epilogue forms that layer's two BN-backward sums from the value it stores and one read of `z_prev`
 * [n][h][w][c_in] (pitch ldz) -- `sums` [UNETK_STAT_REPLICAS][c_in][2] doubles zeroed by the caller.  Other shapes /
 * dtypes run unetk_convT2x2_dgrad followed by unetk_bn_relu_bwd_reduce: same result. */
int unetk_convT2x2_dgrad_bn_reduce_is_fused(int c_in, int c_out, int dtype);
int unetk_convT2x2_dgrad_bn_reduce(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx,
                                   const void* z_prev, int ldz, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, double* sums, int n, int h, int w, int c_in,
                                   int c_out, int hs, int ws, int pad_t, int pad_l, int dtype, void* stream);
int unetk_convT2x2_bias_grad_from_sums(const double* sums, int c_total, int c_off, const void* dup, int ld, int n,
                                       int hs, int ws, int c_up, int pad_t, int pad_l, int h_up, int w_up,
                                       double* scratch, float* db, int dtype, void* stream);
/* out[c] = sum over images and the window [win_y,win_y+win_h) x [win_x,win_x+win_w) of x[n,y,x,c];
 * scratch: c doubles. */
int unetk_channel_sum(const void* x, int ld, int n, int hs, int ws, int c, int win_y, int win_x, int win_h, int win_w,
                      double* scratch, float* out, int dtype, void* stream);

/* ---- BatchNorm2d + ReLU: unet_parts.py:17-18,20-21 (aten::batch_norm, aten::relu_) ---------------------------
 * train: statistics come from the conv epilogue (`stats`, [UNETK_STAT_REPLICAS][c][2] doubles); finalize turns
 * them into scale/shift (+ mean/invstd for backward), updates running_mean/var with momentum and the UNBIASED
 * variance, and increments num_batches_tracked.  eval: scale/shift from the running statistics. */
int unetk_bn_train_finalize(const double* stats, int64_t count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                            float eps, float* scale, float* shift, float* mean, float* invstd, int c, void* stream);
int unetk_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                         float eps, float* scale, float* shift, int c, void* stream);
/* out = relu(y*scale + shift) */
int unetk_bn_relu_apply(const void* y, int ldy, const float* scale, const float* shift, void* out, int ldo,
                        int64_t npix, int c, int dtype, void* stream);
/* The same, fused with the MaxPool2d(2) that consumes the activation in `Down` (unet_parts.py:34; floor mode: an
 * odd last row/column is not pooled): out = relu(y*scale+shift) [n][h][w], pool = 2x2 max of out [n][h/2][w/2].
 * Returns UNETK_ERR_SHAPE when c/(16/es) does not divide 256 (use bn_relu_apply + maxpool2x2_fwd then). */
int unetk_bn_relu_apply_pool(const void* y, int ldy, const float* scale, const float* shift, void* out, int ldo,
                             void* pool, int ldp, int n, int h, int w, int c, int dtype, void* stream);
/* The same apply for the LAST block's last layer, which feeds OutConv (unet_model.py:35, unet_parts.py:74): besides
 * out = relu(y*scale+shift) it emits logits[p][k] = sum_c out[p][c] * w[k][c] + bias[k] (w the fp32 parameter
 * [n_classes][c]; the arithmetic of unetk_conv1x1_fwd on the rounded activation) -- no pass over the activation.
 * UNETK_ERR_SHAPE when n_classes > 4 or c/(16/es) is not a power of two <= 64: use bn_relu_apply + conv1x1_fwd.
 * out may be NULL: the activation is then not stored (nothing reads it when the backward pass forms OutConv's
 * gradients from y -- unetk_bn_relu_bwd_reduce_head / _apply_head -- or when there is no backward pass). */
int unetk_bn_relu_apply_head(const void* y, int ldy, const float* scale, const float* shift, void* out, int ldo,
                             const float* w, const float* bias, int n_classes, void* logits, int ldl, int64_t npix,
                             int c, int dtype, void* stream);
/* backward through ReLU and train-mode BN: reduce (sums [REPLICAS][c][2] doubles, zeroed by the caller) ->
 * finalize (dgamma, dbeta, coef [3][c]) -> apply (dy = coef0*dz - coef1 - coef2*(y-mean)). */
int unetk_bn_relu_bwd_reduce(const void* dout, int lddo, const void* y, int ldy, const float* scale, const float* shift,
                             const float* mean, const float* invstd, double* sums, int64_t npix, int c, int dtype,
                             void* stream);
int unetk_bn_relu_bwd_finalize(const double* sums, int64_t count, const float* gamma, const float* invstd, float* dgamma,
                               float* dbeta, float* coef, int c, void* stream);
int unetk_bn_relu_bwd_apply(const void* dout, int lddo, const void* y, int ldy, const float* scale, const float* shift,
                            const float* mean, const float* coef, void* dy, int lddy, int64_t npix, int c, int dtype,
                            void* stream);
/* The same two passes for an ENCODER block's last layer, whose output feeds the skip connection AND MaxPool2d(2)
 * (unet_model.py:26-30, unet_parts.py:34): dout = dskip + scatter(dpool -> first maximum of each 2x2 window) is formed
 * on the fly from dskip [n][h][w] and dpool [n][h/2][w/2] (the arg-max is recomputed from y), so no max-pool
 * backward pass runs.  UNETK_ERR_SHAPE when c/(16/es) does not divide 256: use unetk_maxpool2x2_bwd + the calls above. */
int unetk_bn_relu_bwd_reduce_pool(const void* dskip, int ldds, const void* dpool, int lddp, const void* y, int ldy,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  double* sums, int n, int h, int w, int c, int dtype, void* stream);
int unetk_bn_relu_bwd_apply_pool(const void* dskip, int ldds, const void* dpool, int lddp, const void* y, int ldy,
                                 const float* scale, const float* shift, const float* mean, const float* coef, void* dy,
                                 int lddy, int n, int h, int w, int c, int dtype, void* stream);
/* ... and for the LAST block's last layer, whose output feeds OutConv (unet_model.py:35, unet_parts.py:74): dout is the
 * 1x1 convolution's input gradient, dout[p][c] = sum_k dl[p][k] * w[k][c] (dl [npix][n_classes] with pitch lddl, w the
 * fp32 parameter [n_classes][c]), formed on the fly -- unetk_conv1x1_dgrad and its [npix][c] output are not needed.
 * UNETK_ERR_SHAPE when n_classes > 4 or c/(16/es) does not divide 256: use unetk_conv1x1_dgrad + the plain calls. */
int unetk_bn_relu_bwd_reduce_head(const void* dl, int lddl, const float* w, int n_classes, const void* y, int ldy,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  double* sums, double* head_sums, int64_t npix, int c, int dtype, void* stream);
/* head_sums != NULL (UNETK_STAT_REPLICAS * n_classes * (c+1) doubles, zeroed by the caller): the reduce pass also
 * accumulates OutConv's own weight/bias gradient -- its input is this layer's activation relu(bn(y)), which the pass
 * recomputes anyway -- and unetk_conv1x1_wgrad_finalize turns the replicas into dw [n_classes][c] and db [n_classes]
 * (autograd of unet_parts.py:74; replaces unetk_conv1x1_wgrad and its pass over the activation). */
int unetk_conv1x1_wgrad_finalize(const double* head_sums, float* dw, float* db, int c_in, int n_classes, void* stream);
int unetk_bn_relu_bwd_apply_head(const void* dl, int lddl, const float* w, int n_classes, const void* y, int ldy,
                                 const float* scale, const float* shift, const float* mean, const float* coef, void* dy,
                                 int lddy, int64_t npix, int c, int dtype, void* stream);

/* ---- MaxPool2d(2): unet_parts.py:34 (aten::max_pool2d), floor mode ---------------------------------------------
 * bwd: dx = (dskip ? dskip : 0) + dpool routed to the first maximum of each window; `dskip` is the gradient the
 * same tensor receives through the skip connection (unet_model.py:31-34), folded in here. dx may alias dskip. */
int unetk_maxpool2x2_fwd(const void* x, int ldx, void* out, int ldo, int n, int h, int w, int c, int dtype,
                         void* stream);
int unetk_maxpool2x2_bwd(const void* x, int ldx, const void* dpool, int lddp, const void* dskip, int ldds, void* dx,
                         int lddx, int n, int h, int w, int c, int dtype, void* stream);

/* ---- Upsample(scale 2, bilinear, align_corners=True) fused with F.pad + concat placement: unet_parts.py:50,59-67
 * `out`/`dcat` point at the first channel of the up-sampled slice of the [n,hs,ws,ld] concat buffer. */
int unetk_bilinear2x_fwd(const void* x, int ldx, void* out, int ldo, int n, int h, int w, int c, int hs, int ws,
                         int pad_t, int pad_l, int dtype, void* stream);
int unetk_bilinear2x_bwd(const void* dcat, int ldd, void* dx, int lddx, int n, int h, int w, int c, int hs, int ws,
                         int pad_t, int pad_l, int dtype, void* stream);
/* copy a [npix, c] channel slice between buffers of different pitch (torch.cat's skip half, unet_parts.py:67,
 * when the caller did not produce the skip tensor inside the concat buffer already) */
int unetk_copy_channels(const void* src, int ld_src, void* dst, int ld_dst, int64_t npix, int c, int dtype,
                        void* stream);

/* ---- OutConv: Conv2d(c_in -> n_classes, k1) + bias: unet_parts.py:74 --------------------------------------------
 * w [n_classes][c_in] fp32 (the parameter itself), logits NHWC [npix][ldo >= n_classes]. */
int unetk_conv1x1_fwd(const void* x, int ldx, const float* w, const float* bias, void* out, int ldo, int64_t npix,
                      int c_in, int n_classes, int dtype, void* stream);
int unetk_conv1x1_dgrad(const void* dl, int lddl, const float* w, void* dx, int lddx, int64_t npix, int c_in,
                        int n_classes, int dtype, void* stream);
int64_t unetk_conv1x1_wgrad_workspace_bytes(int c_in, int n_classes);
int unetk_conv1x1_wgrad(const void* x, int ldx, const void* dl, int lddl, float* dw, float* db, void* workspace,
                        int64_t workspace_bytes, int64_t npix, int c_in, int n_classes, int dtype, void* stream);

/* ---- loss: train.py:81,102-111 (CrossEntropyLoss | BCEWithLogitsLoss) + utils/dice_score.py:25-28 (dice_loss) --
 * fwd: sums4 (4 doubles, scratch) and out4 = {loss, dice, alpha, beta} (floats; alpha/beta feed bwd).
 * bwd: dlogits = grad_out * dloss/dlogits (grad_out: device float, NULL = 1). mask is int64 [npix]. */
int unetk_seg_loss_fwd(const void* logits, int ldz, const int64_t* mask, int64_t npix, int n_classes, float eps,
                       double* sums4, float* out4, int dtype, void* stream);
int unetk_seg_loss_bwd(const void* logits, int ldz, const int64_t* mask, int64_t npix, int n_classes,
                       const float* out4, const float* grad_out, void* dlogits, int lddz, int dtype, void* stream);
/* dice_coeff (utils/dice_score.py:5-17) over n_groups groups of group_size contiguous fp32 elements:
 * dice_out = mean_g (2*sum(x*t)+eps)/(S_g+eps), S_g = sum x + sum t, replaced by 2*sum(x*t) where it is 0.
 * sums: n_groups*3 doubles scratch; coef: n_groups*2 floats for bwd (may be NULL).
 * bwd: dinput = sign * grad_out * d dice/d input. */
int unetk_dice_fwd(const float* input, const float* target, int n_groups, int64_t group_size, float eps, double* sums,
                   float* dice_out, float* coef, void* stream);
int unetk_dice_bwd(const float* target, const float* coef, const float* grad_out, float sign, int n_groups,
                   int64_t group_size, float* dinput, void* stream);

/* ---- validation metric ------------------------------------------------------------------
 * Replaces reference evaluate.py:31-36 for n_classes > 1:
 *   mask_true = one_hot(mask_true); mask_pred = one_hot(logits.argmax(1));
 *   multiclass_dice_coeff(mask_pred[:, first_class:], mask_true[:, first_class:], reduce_batch_first=False)
 * as one pass of integer counting + a finalize: one dice per (image, class >= first_class) with dice_coeff's
 * empty-set rule (dice_score.py:14), mean over the groups -> dice_out[0].  logits: NHWC [n][hw][ld >= n_classes],
 * bf16 or fp32; mask: int64 [n][hw] with values in [0, n_classes) (values outside count for no class). */
int64_t unetk_eval_dice_workspace_bytes(int n, int n_classes);
int unetk_eval_dice(const void* logits, int ld, const int64_t* mask, int n, int64_t hw, int n_classes, int first_class,
                    float eps, void* workspace, int64_t workspace_bytes, float* dice_out, int dtype, void* stream);
/* The same metric for callers that cannot allocate (code under HIP graph capture, tight serving / validation loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference, which allocates freely through PyTorch.
 * Same arguments and result as unetk_eval_dice minus the workspace: the per-(image, class) counters live in a
 * file-scope device array of the library (UNETK_EVAL_NOALLOC_MAX_GROUPS * 24 bytes = 96 KiB per GPU, held until the
 * process ends); the call itself performs no hipMalloc/hipFree.  n * n_classes > UNETK_EVAL_NOALLOC_MAX_GROUPS
 * returns UNETK_ERR_TOO_LARGE (use unetk_eval_dice with a workspace).  All calls on one GPU share that array: calls
 * on the SAME stream are ordered and safe; calls that overlap on one GPU (different streams, or threads launching
 * concurrently) race on the counters and must be serialised by the caller.
 * Preconditions: the device that owns `logits` (and `stream`) must be the CURRENT device of the calling thread -- each
 * GPU has its own copy of the array at its own address; a mismatch (or a host pointer) returns UNETK_ERR_ARG.  The
 * first call on a device looks the array's address up (hipGetSymbolAddress: a lazy code-object load that may allocate
 * and is not allowed under stream capture): make ONE ordinary, un-captured call per device (or per process and
 * device) before capturing or entering a no-allocation region; every later call allocates nothing. */
#define UNETK_EVAL_NOALLOC_MAX_GROUPS 4096
int unetk_eval_dice_noalloc(const void* logits, int ld, const int64_t* mask, int n, int64_t hw, int n_classes,
                            int first_class, float eps, float* dice_out, int dtype, void* stream);

/* ---- prediction mask --------------------------------------------------------------------
 * Replaces reference predict.py:26-31: F.interpolate(output, (out_h, out_w), mode='bilinear') (align_corners=False)
 * followed by output.argmax(dim=1) (n_classes > 1) or torch.sigmoid(output) > threshold (n_classes == 1), fused;
 * logits NHWC [n][h][w][ld >= n_classes] bf16/fp32 -> mask int64 [n][out_h][out_w]. */
int unetk_predict_mask(const void* logits, int ld, int n, int h, int w, int n_classes, int out_h, int out_w,
                       float threshold, int64_t* mask, int dtype, void* stream);

/* ---- optimizer step --------------------------------------------------------------------
 * Replaces torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) followed by
 * torch.optim.RMSprop(lr, alpha, eps, weight_decay, momentum, centered=False).step() -- reference train.py:77-78
 * (construction) and train.py:115-116 (use) -- for parameters, gradients and optimizer state that live in ONE
 * flat fp32 buffer each (n elements, 16-byte aligned):
 *   coef = min(1, max_norm / (||grad||_2 + 1e-6))  (max_norm <= 0: no clipping);  grad *= coef  (written back)
 *   g = grad + weight_decay * param;  square_avg = alpha * square_avg + (1 - alpha) * g * g
 *   avg = sqrt(square_avg) + eps;  momentum > 0:  buf = momentum * buf + g / avg,  param -= lr * buf
 *                                  else:          param -= lr * g / avg
 * workspace: unetk_clip_rmsprop_workspace_bytes() bytes; total_norm_out (device, may be NULL) receives ||grad||_2
 * before clipping, the value clip_grad_norm_ returns. */
int64_t unetk_clip_rmsprop_workspace_bytes(void);
int unetk_clip_rmsprop_step(float* param, float* grad, float* square_avg, float* momentum_buf, int64_t n,
                            void* workspace, int64_t workspace_bytes, double* total_norm_out, float max_norm, float lr,
                            float alpha, float eps, float weight_decay, float momentum, void* stream);

/* ---- input formatting: BasicDataset.preprocess, utils/data_loading.py:65-91 ----------
 * The reference resizes on the CPU with Pillow (`pil_img.resize`, :69: BICUBIC for images, NEAREST for masks) and
 * converts (:72-91).  These entry points do the same on decoded uint8 pixels already in HBM, BIT-EXACTLY
 * (Pillow's fixed-point two-pass convolution / its NEAREST position tables, restated in csrc/preprocess.hip).
 *
 * unetk_resize_plan_build: HOST-side, once per (in size, out size, filter): the per-output-pixel windows and 22-bit
 *   coefficients (filter 3 = BICUBIC) or source positions (filter 0 = NEAREST) that Pillow precomputes per call.
 *   plan_host == NULL returns the plan size; otherwise fills `plan_host` and returns the bytes written.  The caller
 *   copies the plan to the device (`plan_dev`) and passes BOTH to the calls below (the host copy carries the sizes).
 * unetk_preprocess_image: src [n][in_h][in_w][channels] uint8 -> out [n][out_h][out_w][channels] fp32 (NHWC: what
 *   the first convolution reads; the reference's CHW is a permuted view), value/255.0 if any value of THAT image
 *   exceeds 1 (:88-89), else the raw value.  `workspace` of unetk_resize_workspace_bytes() bytes.
 * unetk_preprocess_mask: src [n][in_h][in_w][channels] uint8 -> out [n][out_h][out_w] int64 = index i of the LAST
 *   mask_values[i] ([n_values][channels] uint8, device) equal to the NEAREST-resized pixel, 0 if none (:73-78). */
int64_t unetk_resize_plan_build(int in_w, int in_h, int out_w, int out_h, int filter, void* plan_host,
                                int64_t plan_bytes);
int64_t unetk_resize_workspace_bytes(const void* plan_host, int n, int channels);
int unetk_preprocess_image(const uint8_t* src, int n, int channels, const void* plan_host, const void* plan_dev,
                           float* out, void* workspace, int64_t workspace_bytes, void* stream);
int unetk_preprocess_mask(const uint8_t* src, int n, int channels, const void* plan_host, const void* plan_dev,
                          const uint8_t* mask_values, int n_values, int64_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
