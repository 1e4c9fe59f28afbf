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
 *   - dtype: UNETK_BF16 (bf16 storage, fp32 accumulate) or UNETK_F32 (exact fp32 MFMA).
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

#define UNETK_OK 0
#define UNETK_ERR_SHAPE (-1)   /* unsupported / inconsistent shape, pitch or channel count */
#define UNETK_ERR_DTYPE (-2)
#define UNETK_ERR_ALIGN (-3)   /* pointer or pitch not aligned to 16 bytes where required */
#define UNETK_ERR_LAUNCH (-4)  /* hipGetLastError() after launch */
#define UNETK_ERR_ARG (-5)

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

/* ---- conv3x3, stride 1, pad 1, no bias: unet_parts.py:16,19 (aten::conv2d) -------------------
 * y[n,y,x,co] = sum_{kh,kw,ci} x[n,y+kh-1,x+kw-1,ci] * w[co,ci,kh,kw]      (zero padding)
 * `wpk` = unetk_pack_weight(kind 0).  c_in % 64 == 0 (bf16: % 32), c_out % 64 == 0.
 * If `stats` != NULL it must hold UNETK_STAT_REPLICAS*c_out*2 doubles, zeroed by the caller; the
 * kernel adds per-channel sum and sum-of-squares of the (rounded) outputs: the batch statistics
 * BatchNorm2d needs (unet_parts.py:17,20), fused into the conv epilogue. */
#define UNETK_STAT_REPLICAS 32
int unetk_conv3x3_fprop(const void* x, int ldx, int x_choff, const void* wpk, void* y, int ldy, int y_choff,
                        double* stats, int n, int h, int w, int c_in, int c_out, int dtype, void* stream);
/* dgrad of the same layer (autograd of unet_parts.py:16,19): dx = conv3x3(dy, rot180(w)^T).
 * `wpk` = unetk_pack_weight(kind 1); c_in/c_out are those of the FORWARD layer. */
int unetk_conv3x3_dgrad(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx,
                        int dx_choff, int n, int h, int w, int c_in, int c_out, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
