// HBM-bound kernels of the U-Net hot path: BatchNorm(+ReLU) forward/backward, MaxPool2d(2) forward/backward,
// bilinear x2 (align_corners=True) up-sampling fused with pad + concat placement, channel-slice copies.
// All of them move 16 bytes per lane per access (cdna_hip_programming.md Guideline 13) over NHWC tensors whose
// pixel pitch `ld` may exceed the channel count (channel slices of the decoder's concat buffer).
#include "common.h"

namespace {

template <typename T> struct Vec {
  static constexpr int E = ElemTraits<T>::E;
  static __device__ __forceinline__ void load(const T* p, float (&v)[E]) {
#ifndef EW_NO_NT_LOAD
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const uint4 raw = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
#else
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
#endif
    if constexpr (sizeof(T) == 2) {
      const typename H16<T>::x8 b = __builtin_bit_cast(typename H16<T>::x8, raw);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (float)b[i];
    } else {
      const f32x4 f = __builtin_bit_cast(f32x4, raw);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = f[i];
    }
  }
  static __device__ __forceinline__ void store(T* p, const float (&v)[E]) {
    if constexpr (sizeof(T) == 2) {
      typename H16<T>::x8 b;
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i] = (T)v[i];
      put(p, __builtin_bit_cast(uint4, b));
    } else {
      f32x4 f = {v[0], v[1], v[2], v[3]};
      put(p, __builtin_bit_cast(uint4, f));
    }
  }
  static __device__ __forceinline__ void put(T* p, uint4 raw) {
#ifndef EW_NO_NT_STORE
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, raw), reinterpret_cast<u32x4*>(p));
#else
    *reinterpret_cast<uint4*>(p) = raw;
#endif
  }
};

#ifndef EW_GRID_MUL
#define EW_GRID_MUL 8
#endif
constexpr int kMaxGrid = 256 * EW_GRID_MUL;   // memory-bound grids: ~8 blocks per CU, grid-stride the rest (Guideline 11)

inline int ew_grid(long work_items, int per_block = 256, int max_grid = kMaxGrid) {
  long g = (work_items + per_block - 1) / per_block;
  return (int)(g > max_grid ? max_grid : (g < 1 ? 1 : g));
}
#ifndef EW_APPLY_MUL
#define EW_APPLY_MUL 2
#endif
#ifndef EW_BWD_MUL
#define EW_BWD_MUL 4
#endif

// ------------------------------------------------------------------------------------------ BN finalize
// unet_parts.py:17,20 (aten::batch_norm, training): biased variance for normalisation, unbiased for the
// running estimate, momentum update, num_batches_tracked += 1.
__global__ void bn_train_finalize_kernel(const double* __restrict__ stats, long count, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float* running_mean, float* running_var,
                                         long long* nbt, float momentum, float eps, float* scale, float* shift,
                                         float* mean_out, float* invstd_out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double s = 0, q = 0;
  for (int r = 0; r < UNETK_STAT_REPLICAS; ++r) {
    s += stats[((size_t)r * C + c) * 2 + 0];
    q += stats[((size_t)r * C + c) * 2 + 1];
  }
  const double mean = s / (double)count;
  double var = q / (double)count - mean * mean;
  if (var < 0) var = 0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * invstd;
  shift[c] = b - (float)mean * g * invstd;
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  if (running_mean) {
    const double unbiased = count > 1 ? var * ((double)count / (double)(count - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_params_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * invstd;
  shift[c] = b - rm[c] * g * invstd;
}

// ------------------------------------------------------------------------------------------ BN + ReLU apply
// out = relu(y*scale + shift)   (unet_parts.py:17-18, 20-21 fused)
template <typename T>
__global__ void bn_relu_apply_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                     const float* __restrict__ shift, T* __restrict__ out, int ldo, long npix, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const long total = npix * ppp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx / ppp;
    const int c0 = (int)(idx - pix * ppp) * E;
    float v[E], sc[E], sh[E];
    Vec<T>::load(y + pix * ldy + c0, v);
#pragma unroll
    for (int i = 0; i < E; ++i) { sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; }
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f);
    Vec<T>::store(out + pix * ldo + c0, v);
  }
}

// The same pass when the pieces of one pixel divide the 256-thread block (every channel count the U-Net has):
// a thread keeps ONE channel piece for the whole launch, so its per-channel coefficients live in registers and
// the loop body is pure streaming (two pixels per trip for more bytes in flight).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_fixed_kernel(const T* __restrict__ y, int ldy,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ out,
                                                                  int ldo, long npix, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const int lanes = 256 / ppp;
  const int c0 = (threadIdx.x % ppp) * E;
  const int plane = threadIdx.x / ppp;
  float sc[E], sh[E];
#pragma unroll
  for (int i = 0; i < E; ++i) { sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; }
  const long step = (long)gridDim.x * lanes;
  long pix = (long)blockIdx.x * lanes + plane;
  for (; pix + step < npix; pix += 2 * step) {
    float v0[E], v1[E];
    Vec<T>::load(y + pix * ldy + c0, v0);
    Vec<T>::load(y + (pix + step) * ldy + c0, v1);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      v0[i] = fmaxf(fmaf(v0[i], sc[i], sh[i]), 0.f);
      v1[i] = fmaxf(fmaf(v1[i], sc[i], sh[i]), 0.f);
    }
    Vec<T>::store(out + pix * ldo + c0, v0);
    Vec<T>::store(out + (pix + step) * ldo + c0, v1);
  }
  if (pix < npix) {
    float v0[E];
    Vec<T>::load(y + pix * ldy + c0, v0);
#pragma unroll
    for (int i = 0; i < E; ++i) v0[i] = fmaxf(fmaf(v0[i], sc[i], sh[i]), 0.f);
    Vec<T>::store(out + pix * ldo + c0, v0);
  }
}

// BN + ReLU apply fused with the MaxPool2d(2) that follows it in the encoder (unet_parts.py:20-21 then :34): one
// thread owns a 2x2 pixel quad x one 16-byte channel piece, writes the four activations (into the concat buffer's
// skip half) and, for complete quads (floor mode drops an odd last row/column), their maximum -- the pooled tensor
// costs no read pass of its own.  Fixed-piece layout as above (coefficients in registers).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_pool_kernel(const T* __restrict__ y, int ldy,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, T* __restrict__ out,
                                                                 int ldo, T* __restrict__ pool, int ldp, int N, int H,
                                                                 int W, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const int lanes = 256 / ppp;
  const int c0 = (threadIdx.x % ppp) * E;
  const int plane = threadIdx.x / ppp;
  float sc[E], sh[E];
#pragma unroll
  for (int i = 0; i < E; ++i) { sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; }
  const int QH = (H + 1) / 2, QW = (W + 1) / 2, PH = H / 2, PW = W / 2;
  const long nquads = (long)N * QH * QW;
  for (long q = (long)blockIdx.x * lanes + plane; q < nquads; q += (long)gridDim.x * lanes) {
    const int qx = (int)(q % QW);
    const long r = q / QW;
    const int qy = (int)(r % QH);
    const long n = r / QH;
    const int y0 = 2 * qy, x0 = 2 * qx;
    const bool has_x1 = x0 + 1 < W, has_y1 = y0 + 1 < H;
    const long p00 = (n * H + y0) * W + x0;
    // clamped addresses instead of branches: an absent pixel re-reads a present one and is never stored, and all four
    // loads are in flight together
    const long p01 = p00 + (has_x1 ? 1 : 0), p10 = p00 + (has_y1 ? W : 0), p11 = p10 + (has_x1 ? 1 : 0);
    float v[4][E];
    Vec<T>::load(y + p00 * ldy + c0, v[0]);
    Vec<T>::load(y + p01 * ldy + c0, v[1]);
    Vec<T>::load(y + p10 * ldy + c0, v[2]);
    Vec<T>::load(y + p11 * ldy + c0, v[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < E; ++i) v[k][i] = fmaxf(fmaf(v[k][i], sc[i], sh[i]), 0.f);
    Vec<T>::store(out + p00 * ldo + c0, v[0]);
    if (has_x1) Vec<T>::store(out + p01 * ldo + c0, v[1]);
    if (has_y1) Vec<T>::store(out + p10 * ldo + c0, v[2]);
    if (has_x1 && has_y1) {
      Vec<T>::store(out + p11 * ldo + c0, v[3]);
      float m[E];
#pragma unroll
      for (int i = 0; i < E; ++i) {
        // the pooled value is the maximum of the STORED activations (rounded to T), as MaxPool2d sees them: rounding is
        // monotone, so that is the rounded maximum of the unrounded ones -- the store below rounds it
        m[i] = fmaxf(fmaxf(v[0][i], v[1][i]), fmaxf(v[2][i], v[3][i]));
      }
      Vec<T>::store(pool + ((n * PH + qy) * PW + qx) * ldp + c0, m);
    }
  }
}

// Second sum of the BatchNorm-backward reduction, S2 = sum dz * (y - mean) * invstd.  16-bit tensors accumulate sum dz*y
// and are corrected once per thread (one multiply-add per element; the 8-11 bit mantissa of the stored y bounds
// |mean| / std, so the correction cancels nothing the rounding of y has not lost already -- the dgrad-fused reductions
// do the same); fp32 tensors keep the centred operand and only move the multiplication by invstd to the end.
template <typename T> __device__ __forceinline__ float s2_step(float s2, float dz, float v, float mu) {
  if constexpr (sizeof(T) == 2) return fmaf(dz, v, s2);
  else return fmaf(dz, v - mu, s2);
}
template <typename T> __device__ __forceinline__ float s2_final(float s1, float s2, float mu, float is) {
  if constexpr (sizeof(T) == 2) return fmaf(-mu, s1, s2) * is;
  else return s2 * is;
}

// ------------------------------------------------------------------------------------------ BN + ReLU backward
// pass 1: per-channel  S1 = sum dz,  S2 = sum dz*xhat   with dz = dout * [bn(y) > 0], xhat = (y-mean)*invstd
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const T* __restrict__ dout, int lddo,
                                                                 const T* __restrict__ y, int ldy,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, double* sums,
                                                                 long npix, int C) {
  constexpr int E = Vec<T>::E;
  __shared__ float red[256 * E * 2];   // [pieces in group][E][2] <= 256*E*2 floats
  const int ppp = C / E;
  const int pg = ppp < 256 ? ppp : 256;           // pieces handled by this block's channel group
  const int lanes = 256 / pg;                     // pixel lanes
  const int piece_in_group = threadIdx.x % pg;
  const int plane = threadIdx.x / pg;
  const int piece = blockIdx.y * 256 + piece_in_group;
  const bool active = plane < lanes && piece < ppp;
  const int c0 = piece * E;
  for (int i = threadIdx.x; i < pg * E * 2; i += 256) red[i] = 0.f;
  __syncthreads();
  float s1[E], s2[E];
#pragma unroll
  for (int i = 0; i < E; ++i) s1[i] = s2[i] = 0.f;
  if (active) {
    float sc[E], sh[E], mu[E];
#pragma unroll
    for (int i = 0; i < E; ++i) { sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; mu[i] = mean[c0 + i]; }
    for (long pix = (long)blockIdx.x * lanes + plane; pix < npix; pix += (long)gridDim.x * lanes) {
      float g[E], v[E];
      Vec<T>::load(dout + pix * lddo + c0, g);
      Vec<T>::load(y + pix * ldy + c0, v);
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const float dz = fmaf(v[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
        s1[i] += dz;
        s2[i] = s2_step<T>(s2[i], dz, v[i], mu[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
      atomicAdd(&red[(piece_in_group * E + i) * 2 + 0], s1[i]);
      atomicAdd(&red[(piece_in_group * E + i) * 2 + 1], s2_final<T>(s1[i], s2[i], mu[i], invstd[c0 + i]));
    }
  }
  __syncthreads();
  double* dst = sums + ((size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * C + blockIdx.y * 256 * E) * 2;
  const int nvals = pg * E * 2;
  for (int i = threadIdx.x; i < nvals; i += 256)
    if (blockIdx.y * 256 * E + i / 2 < C) unsafeAtomicAdd(dst + i, (double)red[i]);
}

// finalize: dgamma = S2, dbeta = S1; coefficients of  dy = a*dz - b - c*(y - mean)
__global__ void bn_relu_bwd_finalize_kernel(const double* __restrict__ sums, long count, const float* __restrict__ gamma,
                                            const float* __restrict__ invstd, float* dgamma, float* dbeta, float* coef,
                                            int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int r = 0; r < UNETK_STAT_REPLICAS; ++r) {
    s1 += sums[((size_t)r * C + c) * 2 + 0];
    s2 += sums[((size_t)r * C + c) * 2 + 1];
  }
  const float g = gamma ? gamma[c] : 1.f, is = invstd[c];
  if (dgamma) dgamma[c] = (float)s2;
  if (dbeta) dbeta[c] = (float)s1;
  const double a = (double)g * is;
  coef[c] = (float)a;
  coef[C + c] = (float)(a * s1 / (double)count);
  coef[2 * C + c] = (float)(a * s2 / (double)count * is);
}

template <typename T>
__global__ void bn_relu_bwd_apply_kernel(const T* __restrict__ dout, int lddo, const T* __restrict__ y, int ldy,
                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                         const float* __restrict__ mean, const float* __restrict__ coef,
                                         T* __restrict__ dy, int lddy, long npix, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const long total = npix * ppp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx / ppp;
    const int c0 = (int)(idx - pix * ppp) * E;
    float g[E], v[E], o[E];
    Vec<T>::load(dout + pix * lddo + c0, g);
    Vec<T>::load(y + pix * ldy + c0, v);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int c = c0 + i;
      const float dz = fmaf(v[i], scale[c], shift[c]) > 0.f ? g[i] : 0.f;
      o[i] = coef[c] * dz - coef[C + c] - coef[2 * C + c] * (v[i] - mean[c]);
    }
    Vec<T>::store(dy + pix * lddy + c0, o);
  }
}

// fixed-piece variant (see bn_relu_apply_fixed_kernel): 5 coefficient rows in registers, two pixels per trip
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_fixed_kernel(
    const T* __restrict__ dout, int lddo, const T* __restrict__ y, int ldy, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ coef,
    T* __restrict__ dy, int lddy, long npix, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const int lanes = 256 / ppp;
  const int c0 = (threadIdx.x % ppp) * E;
  const int plane = threadIdx.x / ppp;
  float sc[E], sh[E], mu[E], ka[E], kb[E], kc[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; mu[i] = mean[c0 + i];
    ka[i] = coef[c0 + i]; kb[i] = coef[C + c0 + i]; kc[i] = coef[2 * C + c0 + i];
  }
  const long step = (long)gridDim.x * lanes;
  long pix = (long)blockIdx.x * lanes + plane;
  for (; pix + step < npix; pix += 2 * step) {
    float g0[E], v0[E], g1[E], v1[E];
    Vec<T>::load(dout + pix * lddo + c0, g0);
    Vec<T>::load(y + pix * ldy + c0, v0);
    Vec<T>::load(dout + (pix + step) * lddo + c0, g1);
    Vec<T>::load(y + (pix + step) * ldy + c0, v1);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float d0 = fmaf(v0[i], sc[i], sh[i]) > 0.f ? g0[i] : 0.f;
      const float d1 = fmaf(v1[i], sc[i], sh[i]) > 0.f ? g1[i] : 0.f;
      g0[i] = ka[i] * d0 - kb[i] - kc[i] * (v0[i] - mu[i]);
      g1[i] = ka[i] * d1 - kb[i] - kc[i] * (v1[i] - mu[i]);
    }
    Vec<T>::store(dy + pix * lddy + c0, g0);
    Vec<T>::store(dy + (pix + step) * lddy + c0, g1);
  }
  if (pix < npix) {
    float g0[E], v0[E];
    Vec<T>::load(dout + pix * lddo + c0, g0);
    Vec<T>::load(y + pix * ldy + c0, v0);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float d0 = fmaf(v0[i], sc[i], sh[i]) > 0.f ? g0[i] : 0.f;
      g0[i] = ka[i] * d0 - kb[i] - kc[i] * (v0[i] - mu[i]);
    }
    Vec<T>::store(dy + pix * lddy + c0, g0);
  }
}

// ------------------------------------------------------------------------------------------ BN + ReLU backward
// of an ENCODER block's second layer, whose output feeds both the skip connection and MaxPool2d(2)
// (unet_model.py:26-30): its gradient is  dskip + scatter(dpool to the first maximum of each 2x2 window).  These two
// variants form that sum on the fly -- one thread owns a 2x2 pixel quad x one 16-byte channel piece, recomputes the
// four stored activations relu(bn(y)) (rounded to T, which is what MaxPool2d compared) and their arg-max -- so the
// max-pool backward pass (read x, dskip, dpool, write dx: 3.25 tensor passes) disappears; the BN kernels read
// dpool (a quarter tensor) instead.
// The quad's operands stay packed (16-byte registers) until the element they belong to is processed.  Measured
// (scripts/perf_bn_pool.py): letting hipcc take ~195 VGPRs (2 waves/SIMD, no scratch) runs at 4.8-5.5 TB/s; capping
// it at 128 or 170 VGPRs by launch bounds spills and drops to 3.4-4.3 TB/s.
template <typename T> __device__ __forceinline__ float raw_elem(const uint4& r, int i) {
  const unsigned w = (i >> 1) == 0 ? r.x : (i >> 1) == 1 ? r.y : (i >> 1) == 2 ? r.z : r.w;
  return (i & 1) ? H16<T>::hi(w) : H16<T>::lo(w);
}
template <> __device__ __forceinline__ float raw_elem<float>(const uint4& r, int i) {
  return __uint_as_float(i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w);
}
template <typename T> __device__ __forceinline__ uint4 raw_load(const T* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
}

// relu(bn(.)) of two values, rounded to T as the forward stored them: one packed conversion per pair
template <typename T>
__device__ __forceinline__ void act_pair(float p0, float p1, float& a0, float& a1) {
  if constexpr (sizeof(T) == 2) {
    const unsigned r = pack2<T>(fmaxf(p0, 0.f), fmaxf(p1, 0.f));
    a0 = H16<T>::lo(r);
    a1 = H16<T>::hi(r);
  } else {
    a0 = fmaxf(p0, 0.f);
    a1 = fmaxf(p1, 0.f);
  }
}

// Walks the 2x2 quads of an [N][QH][QW] grid with a fixed stride: (n, qy, qx) advance by the stride's own decomposition
// (wave-uniform) with two carries per step instead of two 64-bit divisions per quad -- the pool kernels are bound by
// their instruction count (the forward bn_relu_apply_pool_kernel is not: there the walker measured 4 % slower).
// 32-bit counters: the host entry points refuse grids of 2^31 quads or more.
struct QuadWalk {
  int qx, qy, n, q;
  int sx, sy, sn, stride, QW, QH, nquads;
  __device__ __forceinline__ QuadWalk(int first, int stride_, int nquads_, int QH_, int QW_)
      : q(first), stride(stride_), QW(QW_), QH(QH_), nquads(nquads_) {
    qx = first % QW; int r = first / QW; qy = r % QH; n = r / QH;
    sx = stride % QW; r = stride / QW; sy = r % QH; sn = r / QH;
  }
  __device__ __forceinline__ bool valid() const { return q < nquads; }
  __device__ __forceinline__ void next() {
    q += stride;
    qx += sx;
    const bool cx = qx >= QW;
    qx -= cx ? QW : 0;
    qy += sy + (cx ? 1 : 0);
    const bool cy = qy >= QH;
    qy -= cy ? QH : 0;
    n += sn + (cy ? 1 : 0);
  }
};

template <typename T>
struct QuadGrad {
  static constexpr int E = Vec<T>::E;
  uint4 v[4], g[4], gp;
  bool hx, hy;
  __device__ __forceinline__ bool has(int k) const { return (!(k & 1) || hx) && (!(k >> 1) || hy); }
  __device__ __forceinline__ void load(const T* __restrict__ dskip, int ldds, const T* __restrict__ dpool, int lddp,
                                       const T* __restrict__ y, int ldy, long n, int qy, int qx, int H, int W, int c0) {
    const int y0 = 2 * qy, x0 = 2 * qx;
    hx = x0 + 1 < W;
    hy = y0 + 1 < H;
    const long p00 = (n * H + y0) * W + x0;
    // clamped addresses instead of branches: an absent pixel re-reads pixel 0 of the quad and is masked on use.
    // One 64-bit product per tensor; the other three pixels are that address plus the (selected) pixel / row pitch.
    const T* yq = y + p00 * ldy + c0;
    const T* gq = dskip + p00 * ldds + c0;
    const long yx = hx ? ldy : 0, yr = hy ? (long)W * ldy : 0;
    const long gx = hx ? ldds : 0, gr = hy ? (long)W * ldds : 0;
    v[0] = raw_load(yq);           g[0] = raw_load(gq);
    v[1] = raw_load(yq + yx);      g[1] = raw_load(gq + gx);
    v[2] = raw_load(yq + yr);      g[2] = raw_load(gq + gr);
    v[3] = raw_load(yq + yr + yx); g[3] = raw_load(gq + gr + gx);
    const int py = hy ? qy : max(qy - 1, 0), px = hx ? qx : max(qx - 1, 0);      // any valid pooled pixel
    gp = raw_load(dpool + ((n * (H / 2) + py) * (W / 2) + px) * lddp + c0);
  }
  // full output gradient of the quad's four pixels for channel element i, and the pre-BN values
  __device__ __forceinline__ void element(int i, float sc, float sh, float (&yv)[4], float (&gv)[4]) const {
#pragma unroll
    for (int k = 0; k < 4; ++k) { yv[k] = raw_elem<T>(v[k], i); gv[k] = raw_elem<T>(g[k], i); }
    if (hx && hy) {       // complete window (floor mode drops an odd last row / column)
      int a = 0;
      float act[4];
      act_pair<T>(fmaf(yv[0], sc, sh), fmaf(yv[1], sc, sh), act[0], act[1]);
      act_pair<T>(fmaf(yv[2], sc, sh), fmaf(yv[3], sc, sh), act[2], act[3]);
      float m = act[0];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (act[k] > m) { m = act[k]; a = k; }
      const float gpi = raw_elem<T>(gp, i);
#pragma unroll
      for (int k = 0; k < 4; ++k) gv[k] += (a == k) ? gpi : 0.f;
    }
  }
};

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_pool_kernel(
    const T* __restrict__ dskip, int ldds, const T* __restrict__ dpool, int lddp, const T* __restrict__ y, int ldy,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, double* sums, int N, int H, int W, int C) {
  constexpr int E = Vec<T>::E;
  __shared__ float red[256 * E * 2];
  const int ppp = C / E;                      // <= 256 and a divisor of 256 (checked by the host)
  const int lanes = 256 / ppp;
  const int piece = threadIdx.x % ppp, plane = threadIdx.x / ppp;
  const int c0 = piece * E;
  for (int i = threadIdx.x; i < ppp * E * 2; i += 256) red[i] = 0.f;
  __syncthreads();
  float sc[E], sh[E], mu[E], s1[E], s2[E];   // (mu: only the fp32 form reads it inside the loop, see s2_step)
#pragma unroll
  for (int i = 0; i < E; ++i) {
    sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; mu[i] = mean[c0 + i];
    s1[i] = s2[i] = 0.f;
  }
  const int QH = (H + 1) / 2, QW = (W + 1) / 2;
  const long nquads = (long)N * QH * QW;
  for (QuadWalk it(blockIdx.x * lanes + plane, gridDim.x * lanes, (int)nquads, QH, QW); it.valid(); it.next()) {
    QuadGrad<T> Q;
    Q.load(dskip, ldds, dpool, lddp, y, ldy, it.n, it.qy, it.qx, H, W, c0);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      float yv[4], gv[4];
      Q.element(i, sc[i], sh[i], yv, gv);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dz = (Q.has(k) && fmaf(yv[k], sc[i], sh[i]) > 0.f) ? gv[k] : 0.f;
        s1[i] += dz;
        s2[i] = s2_step<T>(s2[i], dz, yv[k], mu[i]);
      }
      __builtin_amdgcn_sched_barrier(0);     // one channel element at a time
    }
  }
#pragma unroll
  for (int i = 0; i < E; ++i) {
    atomicAdd(&red[(piece * E + i) * 2 + 0], s1[i]);
    atomicAdd(&red[(piece * E + i) * 2 + 1], s2_final<T>(s1[i], s2[i], mean[c0 + i], invstd[c0 + i]));
  }
  __syncthreads();
  double* dst = sums + (size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * C * 2;
  for (int i = threadIdx.x; i < C * 2; i += 256) unsafeAtomicAdd(dst + i, (double)red[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_pool_kernel(
    const T* __restrict__ dskip, int ldds, const T* __restrict__ dpool, int lddp, const T* __restrict__ y, int ldy,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ coef, T* __restrict__ dy, int lddy, int N, int H, int W, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const int lanes = 256 / ppp;
  const int c0 = (threadIdx.x % ppp) * E;
  const int plane = threadIdx.x / ppp;
  float sc[E], sh[E], mu[E], ka[E], kb[E], kc[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; mu[i] = mean[c0 + i];
    ka[i] = coef[c0 + i]; kb[i] = coef[C + c0 + i]; kc[i] = coef[2 * C + c0 + i];
  }
  const int QH = (H + 1) / 2, QW = (W + 1) / 2;
  const long nquads = (long)N * QH * QW;
  for (QuadWalk it(blockIdx.x * lanes + plane, gridDim.x * lanes, (int)nquads, QH, QW); it.valid(); it.next()) {
    const int qx = it.qx, qy = it.qy;
    const long n = it.n;
    QuadGrad<T> Q;
    Q.load(dskip, ldds, dpool, lddp, y, ldy, n, qy, qx, H, W, c0);
    float o[4][E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      float yv[4], gv[4];
      Q.element(i, sc[i], sh[i], yv, gv);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dz = fmaf(yv[k], sc[i], sh[i]) > 0.f ? gv[k] : 0.f;
        o[k][i] = ka[i] * dz - kb[i] - kc[i] * (yv[k] - mu[i]);
      }
      __builtin_amdgcn_sched_barrier(0);     // one channel element at a time
    }
    const long p00 = (n * H + 2 * qy) * W + 2 * qx;
    Vec<T>::store(dy + p00 * lddy + c0, o[0]);
    if (Q.hx) Vec<T>::store(dy + (p00 + 1) * lddy + c0, o[1]);
    if (Q.hy) Vec<T>::store(dy + (p00 + W) * lddy + c0, o[2]);
    if (Q.hx && Q.hy) Vec<T>::store(dy + (p00 + W + 1) * lddy + c0, o[3]);
  }
}

// ------------------------------------------------------------------------------------------ BN + ReLU backward
// of the LAST block's second layer, whose output feeds OutConv (unet_model.py:35, unet_parts.py:74): its gradient is
// the 1x1 convolution's input gradient  dout[p][c] = sum_k dl[p][k] * w[k][c]  -- n_classes multiply-adds per element
// from a tensor 32x smaller than dout.  Formed on the fly (n_classes <= 4; the fp32 chain of unetk_conv1x1_dgrad,
// without its rounding to T), it saves that kernel's 1.26 GB write and the two 1.26 GB reads of these passes.
constexpr int kHeadMax = 4;

// A lane owns FOUR CONSECUTIVE pixels x one 16-byte channel piece per trip.  Their 4 x NCLS logit gradients are
// contiguous (pitch == NCLS) and come in one or two 16-byte loads -- per-pixel 2-byte loads cost the texture
// addresser 8x the cycles of the 16-byte load of y and made the fused kernels slower than the plain ones.
template <typename T, int NCLS>
struct HeadQuad {
  static constexpr int E = Vec<T>::E;
  static constexpr int BYTES = 4 * NCLS * (int)sizeof(T);
  static constexpr int NV = (BYTES + 15) / 16;
  static constexpr bool kVector = BYTES % 16 == 0;
  uint4 yraw[4];
  float gk[4][NCLS];
  // q = index of the pixel group; `full` groups have all four pixels inside the tensor
  __device__ __forceinline__ void load(const T* __restrict__ dl, int lddl, const T* __restrict__ y, int ldy, long q,
                                       long npix, int c0) {
    const long p0 = 4 * q;
    const bool full = p0 + 3 < npix;
    const T* yq = y + p0 * ldy + c0;                              // one 64-bit product; the pixels are pitches apart
#pragma unroll
    for (int u = 0; u < 4; ++u) yraw[u] = raw_load(yq + (p0 + u < npix ? u * ldy : 0));   // clamped; masked by the callers
    if (kVector && full && lddl == NCLS && (reinterpret_cast<uintptr_t>(dl) & 15) == 0) {
      uint4 v[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const uint4*>(dl + p0 * NCLS + i * (16 / (int)sizeof(T)));
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < NCLS; ++k) {
          constexpr int PER = 16 / (int)sizeof(T);
          const int e = u * NCLS + k;
          gk[u][k] = raw_elem<T>(v[e / PER], e % PER);
        }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long p = p0 + u < npix ? p0 + u : p0;
#pragma unroll
        for (int k = 0; k < NCLS; ++k) gk[u][k] = to_f32(dl[p * lddl + k]);
      }
    }
  }
};

// the value another lane of the row holds, through the DPP path of the VALU (no LDS round trip); every lane active
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Forward of the same pair (unet_parts.py:20-21 then :74): BatchNorm + ReLU apply of the last block's last layer that
// also emits OutConv's logits -- the activation is in registers, so the 1x1 convolution costs n_classes dot products
// per pixel (8 channels per lane, then an xor-shuffle sum over the pixel's lanes: the arithmetic of
// conv1x1_fwd_kernel on the rounded activation) and no pass over the activation.
template <typename T, int NCLS, bool STORE>
__global__ __launch_bounds__(256) void bn_relu_apply_head_kernel(const T* __restrict__ y, int ldy,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, T* __restrict__ out,
                                                                 int ldo, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, T* __restrict__ logits,
                                                                 int ldl, long npix, int C) {
  constexpr int E = Vec<T>::E;
  constexpr int PER = 16 / (int)sizeof(T);
  constexpr int BYTES = 4 * NCLS * (int)sizeof(T);
  const int ppp = C / E;                      // a power of two <= 64: a pixel's lanes sit in one wave
  const int lanes = 256 / ppp;
  const int piece = threadIdx.x % ppp, plane = threadIdx.x / ppp;
  const int c0 = piece * E;
  float sc[E], sh[E], wk[NCLS][E], bk[NCLS];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) wk[k][i] = w[k * C + c0 + i];
  }
#pragma unroll
  for (int k = 0; k < NCLS; ++k) bk[k] = bias[k];
  const bool vec_ok = BYTES % 16 == 0 && ldl == NCLS && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
  const long ngroups = (npix + 3) / 4;
  // every lane of a wave runs the same number of trips (the shuffles below need the pixel's lanes converged)
  const long trips = (ngroups + (long)gridDim.x * lanes - 1) / ((long)gridDim.x * lanes);
  for (long t = 0; t < trips; ++t) {
    const long q = (t * gridDim.x + blockIdx.x) * lanes + plane;
    const long p0 = 4 * q;
    float acc[4][NCLS];
    uint4 raws[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raws[u] = raw_load(y + (p0 + u < npix ? p0 + u : 0) * ldy + c0);   // all in flight
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool live = p0 + u < npix;
      const long p = live ? p0 + u : 0;
      const uint4 raw = raws[u];
      float a[E];
#pragma unroll
      for (int i = 0; i < E; i += 2)    // a[] = the activation as stored (rounded to T): what OutConv reads
        act_pair<T>(fmaf(raw_elem<T>(raw, i), sc[i], sh[i]), fmaf(raw_elem<T>(raw, i + 1), sc[i + 1], sh[i + 1]), a[i], a[i + 1]);
      if (STORE && live) Vec<T>::store(out + p * ldo + c0, a);
#pragma unroll
      for (int k = 0; k < NCLS; ++k) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < E; ++i) d = fmaf(a[i], wk[k][i], d);
        acc[u][k] = d;
      }
    }
    if (ppp == 8) {
      // a pixel = one aligned group of 8 lanes (64 channels in 16-bit elements: the U-Net's last layer): the sum runs
      // on DPP operands of the adds -- lane ^ 1, lane ^ 2 inside a quad, then the mirrored half row, which pairs the
      // two quads (after the first two steps the four lanes of a quad hold the same value, so this equals lane ^ 4 bit
      // for bit: the ascending xor order of the generic loop below and of conv1x1_fwd) -- instead of 3 x 4 x NCLS
      // ds_bpermute round trips through the LDS pipeline per trip
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < NCLS; ++k) {
          float v = acc[u][k];
          v += dpp_f32<0xB1>(v);        // quad_perm [1,0,3,2]
          v += dpp_f32<0x4E>(v);        // quad_perm [2,3,0,1]
          v += dpp_f32<0x141>(v);       // row_half_mirror
          acc[u][k] = v;
        }
    } else {
      for (int o = 1; o < ppp; o <<= 1)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < NCLS; ++k) acc[u][k] += __shfl_xor(acc[u][k], o, 64);
    }
    if (piece == 0 && p0 < npix) {
      if (vec_ok && p0 + 3 < npix) {
        alignas(16) T tmp[((4 * NCLS + PER - 1) / PER) * PER];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < NCLS; ++k) tmp[u * NCLS + k] = from_f32<T>(acc[u][k] + bk[k]);
#pragma unroll
        for (int i = 0; i < BYTES / 16; ++i)
          *reinterpret_cast<uint4*>(logits + p0 * NCLS + i * PER) = *reinterpret_cast<const uint4*>(tmp + i * PER);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (p0 + u < npix)
#pragma unroll
            for (int k = 0; k < NCLS; ++k) logits[(p0 + u) * ldl + k] = from_f32<T>(acc[u][k] + bk[k]);
      }
    }
  }
}

template <typename T, int NCLS>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_head_kernel(
    const T* __restrict__ dl, int lddl, const float* __restrict__ w, const T* __restrict__ y, int ldy,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, double* sums, double* head_sums, long npix, int C) {
  constexpr int E = Vec<T>::E;
  __shared__ float red[256 * E * 2];
  __shared__ float hred[NCLS * (256 * E + 1)];
  const int ppp = C / E;                      // <= 256 and a divisor of 256 (checked by the host)
  const int lanes = 256 / ppp;
  const int piece = threadIdx.x % ppp, plane = threadIdx.x / ppp;
  const int c0 = piece * E;
  for (int i = threadIdx.x; i < ppp * E * 2; i += 256) red[i] = 0.f;
  for (int i = threadIdx.x; i < NCLS * (C + 1); i += 256) hred[i] = 0.f;
  __syncthreads();
  float sc[E], sh[E], mu[E], s1[E], s2[E], wk[NCLS][E];      // (mu: see s2_step)
  // head_sums != NULL: OutConv's weight / bias gradient as a by-product -- its input IS this layer's activation
  // relu(bn(y)) (rounded to T as the forward stored it), which this pass recomputes for the mask anyway:
  //   dw[k][c] = sum_p dl[p][k] * a[p][c],  db[k] = sum_p dl[p][k]           (no pass over the activation)
  float hw[NCLS][E], hb[NCLS];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; mu[i] = mean[c0 + i];
    s1[i] = s2[i] = 0.f;
#pragma unroll
    for (int k = 0; k < NCLS; ++k) { wk[k][i] = w[k * C + c0 + i]; hw[k][i] = 0.f; }
  }
#pragma unroll
  for (int k = 0; k < NCLS; ++k) hb[k] = 0.f;
  const long ngroups = (npix + 3) / 4;
  for (long q = (long)blockIdx.x * lanes + plane; q < ngroups; q += (long)gridDim.x * lanes) {
    HeadQuad<T, NCLS> Q;
    Q.load(dl, lddl, y, ldy, q, npix, c0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool live = 4 * q + u < npix;
#pragma unroll
      for (int k = 0; k < NCLS; ++k) {
        if (!live) Q.gk[u][k] = 0.f;
        hb[k] += Q.gk[u][k];
      }
#pragma unroll
      for (int i = 0; i < E; i += 2) {
        const float v0 = raw_elem<T>(Q.yraw[u], i), v1 = raw_elem<T>(Q.yraw[u], i + 1);
        const float pre0 = fmaf(v0, sc[i], sh[i]), pre1 = fmaf(v1, sc[i + 1], sh[i + 1]);
        float a0, a1;
        act_pair<T>(pre0, pre1, a0, a1);
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int k = 0; k < NCLS; ++k) {
          g0 = fmaf(Q.gk[u][k], wk[k][i], g0);
          g1 = fmaf(Q.gk[u][k], wk[k][i + 1], g1);
          hw[k][i] = fmaf(Q.gk[u][k], a0, hw[k][i]);
          hw[k][i + 1] = fmaf(Q.gk[u][k], a1, hw[k][i + 1]);
        }
        const float dz0 = pre0 > 0.f ? g0 : 0.f, dz1 = pre1 > 0.f ? g1 : 0.f;
        s1[i] += dz0;
        s1[i + 1] += dz1;
        s2[i] = s2_step<T>(s2[i], dz0, v0, mu[i]);
        s2[i + 1] = s2_step<T>(s2[i + 1], dz1, v1, mu[i + 1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < E; ++i) {
    atomicAdd(&red[(piece * E + i) * 2 + 0], s1[i]);
    atomicAdd(&red[(piece * E + i) * 2 + 1], s2_final<T>(s1[i], s2[i], mean[c0 + i], invstd[c0 + i]));
  }
  if (head_sums != nullptr) {
#pragma unroll
    for (int k = 0; k < NCLS; ++k) {
#pragma unroll
      for (int i = 0; i < E; ++i) atomicAdd(&hred[k * (C + 1) + c0 + i], hw[k][i]);
      if (piece == 0) atomicAdd(&hred[k * (C + 1) + C], hb[k]);
    }
  }
  __syncthreads();
  double* dst = sums + (size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * C * 2;
  for (int i = threadIdx.x; i < C * 2; i += 256) unsafeAtomicAdd(dst + i, (double)red[i]);
  if (head_sums != nullptr) {
    double* hd = head_sums + (size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * NCLS * (C + 1);
    for (int i = threadIdx.x; i < NCLS * (C + 1); i += 256) unsafeAtomicAdd(hd + i, (double)hred[i]);
  }
}

// dw [n_classes][C] and db [n_classes] of OutConv from the replicas of the by-product above
__global__ void head_wgrad_finalize_kernel(const double* __restrict__ head_sums, float* __restrict__ dw,
                                           float* __restrict__ db, int C, int ncls) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncls * (C + 1)) return;
  double s = 0;
  for (int r = 0; r < UNETK_STAT_REPLICAS; ++r) s += head_sums[(size_t)r * ncls * (C + 1) + i];
  const int k = i / (C + 1), c = i - k * (C + 1);
  if (c < C) dw[k * C + c] = (float)s;
  else if (db) db[k] = (float)s;
}

template <typename T, int NCLS>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_head_kernel(
    const T* __restrict__ dl, int lddl, const float* __restrict__ w, const T* __restrict__ y, int ldy,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ coef, T* __restrict__ dy, int lddy, long npix, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const int lanes = 256 / ppp;
  const int c0 = (threadIdx.x % ppp) * E;
  const int plane = threadIdx.x / ppp;
  float sc[E], sh[E], mu[E], ka[E], kb[E], kc[E], wk[NCLS][E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    sc[i] = scale[c0 + i]; sh[i] = shift[c0 + i]; mu[i] = mean[c0 + i];
    ka[i] = coef[c0 + i]; kb[i] = coef[C + c0 + i]; kc[i] = coef[2 * C + c0 + i];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) wk[k][i] = w[k * C + c0 + i];
  }
  const long ngroups = (npix + 3) / 4;
  for (long q = (long)blockIdx.x * lanes + plane; q < ngroups; q += (long)gridDim.x * lanes) {
    HeadQuad<T, NCLS> Q;
    Q.load(dl, lddl, y, ldy, q, npix, c0);
    T* dyq = dy + 4 * q * lddy + c0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (4 * q + u >= npix) break;
      float o[E];
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const float v = raw_elem<T>(Q.yraw[u], i);
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < NCLS; ++k) g = fmaf(Q.gk[u][k], wk[k][i], g);
        const float dz = fmaf(v, sc[i], sh[i]) > 0.f ? g : 0.f;
        o[i] = ka[i] * dz - kb[i] - kc[i] * (v - mu[i]);
      }
      Vec<T>::store(dyq + u * lddy, o);
    }
  }
}

// ------------------------------------------------------------------------------------------ MaxPool2d(2)
// unet_parts.py:34: k2 s2, floor mode -- an odd last row/column is dropped.
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo, int N, int H, int W,
                                   int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E, Ho = H / 2, Wo = W / 2;
  const long total = (long)N * Ho * Wo * ppp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int c0 = (int)(r % ppp) * E; r /= ppp;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const long n = r / Ho;
    const T* src = x + ((n * H + 2 * oy) * W + 2 * ox) * ldx + c0;
    float a[E], b[E], c[E], d[E];
    Vec<T>::load(src, a);
    Vec<T>::load(src + ldx, b);
    Vec<T>::load(src + (long)W * ldx, c);
    Vec<T>::load(src + (long)W * ldx + ldx, d);
#pragma unroll
    for (int i = 0; i < E; ++i) a[i] = fmaxf(fmaxf(a[i], b[i]), fmaxf(c[i], d[i]));
    Vec<T>::store(out + ((n * Ho + oy) * Wo + ox) * ldo + c0, a);
  }
}

// dx = (dskip ? dskip : 0) + scatter(dpool to the FIRST maximum of each window, row-major, as ATen does).
// Folding the skip-connection gradient in here replaces autograd's separate accumulation pass.
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dpool, int lddp,
                                   const T* dskip, int ldds, T* dx, int lddx, int N, int H, int W, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E, Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
  const long total = (long)N * Hc * Wc * ppp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int c0 = (int)(r % ppp) * E; r /= ppp;
    const int ox = (int)(r % Wc); r /= Wc;
    const int oy = (int)(r % Hc);
    const long n = r / Hc;
    const bool pooled = oy < Ho && ox < Wo;
    float g[E], v[4][E];
    int arg[E];
    if (pooled) {
      Vec<T>::load(dpool + ((n * Ho + oy) * Wo + ox) * lddp + c0, g);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        Vec<T>::load(x + ((n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * ldx + c0, v[k]);
#pragma unroll
      for (int i = 0; i < E; ++i) {
        int a = 0;
        float m = v[0][i];
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (v[k][i] > m) { m = v[k][i]; a = k; }
        arg[i] = a;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
      if (yy >= H || xx >= W) continue;
      const long off = (n * H + yy) * W + xx;
      float o[E];
      if (dskip) Vec<T>::load(dskip + off * ldds + c0, o);
      else {
#pragma unroll
        for (int i = 0; i < E; ++i) o[i] = 0.f;
      }
      if (pooled) {
#pragma unroll
        for (int i = 0; i < E; ++i)
          if (arg[i] == k) o[i] += g[i];
      }
      Vec<T>::store(dx + off * lddx + c0, o);
    }
  }
}

// ------------------------------------------------------------------------------------------ bilinear x2, AC
// unet_parts.py:50 + 59-67: Upsample(scale 2, bilinear, align_corners=True), F.pad to the skip size, and
// placement into the concat buffer at a channel offset, in one pass: every pixel of the Hs x Ws destination
// slice is written exactly once (zero outside the up-sampled window).
__device__ __forceinline__ void ac_tap(int o, float scale, int n_in, int& i0, int& i1, float& f) {
  const float src = scale * (float)o;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  f = src - (float)i0;
}

template <typename T>
__device__ __forceinline__ void load_cached(const T* p, float (&v)[Vec<T>::E]) {
  // the four taps of neighbouring outputs overlap: ordinary (cacheable) loads, not the streaming ones of Vec<T>::load
  const uint4 raw = *reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < Vec<T>::E; ++i) v[i] = raw_elem<T>(raw, i);
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo,
                                                           int N, int h, int w, int C, int Hs, int Ws, int padT,
                                                           int padL) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const float sy = (2 * h > 1) ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
  const float sx = (2 * w > 1) ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
  // one block row of the grid per destination row (n, Y): no 64-bit divisions, the row's taps y0/y1/fy are uniform
  const int rows = N * Hs;
  const int per_row = Ws * ppp;
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {
    const int n = row / Hs, Y = row - n * Hs;
    const int oy = Y - padT;
    const bool row_in = oy >= 0 && oy < 2 * h;
    int y0 = 0, y1 = 0;
    float fy = 0.f;
    if (row_in) ac_tap(oy, sy, h, y0, y1, fy);
    const T* base = x + (size_t)n * h * w * ldx;
    T* orow = out + (size_t)row * Ws * ldo;
    for (int it = blockIdx.x * 256 + threadIdx.x; it < per_row; it += gridDim.x * 256) {
      const int X = it / ppp, c0 = (it - X * ppp) * E;
      const int ox = X - padL;
      float o[E];
      if (row_in && ox >= 0 && ox < 2 * w) {
        int x0, x1;
        float fx;
        ac_tap(ox, sx, w, x0, x1, fx);
        float a[E], b[E], c[E], d[E];
        load_cached<T>(base + (size_t)(y0 * w + x0) * ldx + c0, a);
        load_cached<T>(base + (size_t)(y0 * w + x1) * ldx + c0, b);
        load_cached<T>(base + (size_t)(y1 * w + x0) * ldx + c0, c);
        load_cached<T>(base + (size_t)(y1 * w + x1) * ldx + c0, d);
#pragma unroll
        for (int i = 0; i < E; ++i)
          o[i] = (1.f - fy) * ((1.f - fx) * a[i] + fx * b[i]) + fy * ((1.f - fx) * c[i] + fx * d[i]);
      } else {
#pragma unroll
        for (int i = 0; i < E; ++i) o[i] = 0.f;
      }
      Vec<T>::store(orow + (size_t)X * ldo + c0, o);
    }
  }
}

template <typename T>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dcat, int ldd, T* __restrict__ dx, int lddx, int N, int h,
                                    int w, int C, int Hs, int Ws, int padT, int padL) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const float sy = (2 * h > 1) ? (float)(h - 1) / (float)(2 * h - 1) : 0.f;
  const float sx = (2 * w > 1) ? (float)(w - 1) / (float)(2 * w - 1) : 0.f;
  const long total = (long)N * h * w * ppp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int c0 = (int)(r % ppp) * E; r /= ppp;
    const int ix = (int)(r % w); r /= w;
    const int iy = (int)(r % h);
    const long n = r / h;
    float acc[E];
#pragma unroll
    for (int i = 0; i < E; ++i) acc[i] = 0.f;
    for (int oy = max(0, 2 * iy - 2); oy <= min(2 * h - 1, 2 * iy + 4); ++oy) {
      int y0, y1; float fy;
      ac_tap(oy, sy, h, y0, y1, fy);
      const float wy = (y0 == iy ? 1.f - fy : 0.f) + (y1 == iy ? fy : 0.f);
      if (wy == 0.f) continue;
      for (int ox = max(0, 2 * ix - 2); ox <= min(2 * w - 1, 2 * ix + 4); ++ox) {
        int x0, x1; float fx;
        ac_tap(ox, sx, w, x0, x1, fx);
        const float wx = (x0 == ix ? 1.f - fx : 0.f) + (x1 == ix ? fx : 0.f);
        if (wx == 0.f) continue;
        float g[E];
        Vec<T>::load(dcat + ((n * Hs + oy + padT) * Ws + ox + padL) * ldd + c0, g);
#pragma unroll
        for (int i = 0; i < E; ++i) acc[i] += wy * wx * g[i];
      }
    }
    Vec<T>::store(dx + ((n * h + iy) * w + ix) * lddx + c0, acc);
  }
}

// ------------------------------------------------------------------------------------------ slices
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, long npix, int C) {
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const long total = npix * ppp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx / ppp;
    const int c0 = (int)(idx - pix * ppp) * E;
    *reinterpret_cast<uint4*>(dst + pix * ldd + c0) = *reinterpret_cast<const uint4*>(src + pix * lds_ + c0);
  }
}

// zero the F.pad border (unet_parts.py:62-63) of the up-sampled channel slice that convT does not write
template <typename T>
__global__ void zero_border_kernel(T* buf, int ld, int N, int Hs, int Ws, int C, int padT, int padL, int hu, int wu) {
  // enumerates ONLY the border pixels: padT full rows on top, (Ws - wu) pixels of each of the hu middle rows,
  // the remaining full rows at the bottom
  constexpr int E = Vec<T>::E;
  const int ppp = C / E;
  const int side = Ws - wu;                       // border pixels per middle row
  const int n_top = padT * Ws, n_mid = hu * side;
  const int per_img = Hs * Ws - hu * wu;
  const long total = (long)N * per_img * ppp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int c0 = (int)(r % ppp) * E; r /= ppp;
    int b = (int)(r % per_img);
    const long n = r / per_img;
    int X, Y;
    if (b < n_top) {
      Y = b / Ws; X = b - Y * Ws;
    } else if (b < n_top + n_mid) {
      b -= n_top;
      const int row = b / side, k = b - row * side;
      Y = padT + row; X = k < padL ? k : k + wu;
    } else {
      b -= n_top + n_mid;
      const int row = b / Ws;
      Y = padT + hu + row; X = b - row * Ws;
    }
    *reinterpret_cast<uint4*>(buf + ((n * Hs + Y) * Ws + X) * ld + c0) = uint4{0u, 0u, 0u, 0u};
  }
}

// ------------------------------------------------------------------------------------------ prediction mask
// predict.py:26-31 in one pass: F.interpolate(logits, (H, W), mode='bilinear') [align_corners=False: src =
// (dst + 0.5) * in/out - 0.5 clamped at 0, aten upsample_bilinear2d's weights and summation order], then argmax over
// the classes (first maximum) or sigmoid(x) > threshold for a single class.  One thread per output pixel.
template <typename T>
__global__ void predict_mask_kernel(const T* __restrict__ logits, int ld, int N, int h, int w, int C, int H, int W,
                                    float threshold, long long* __restrict__ mask) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long total = (long)N * H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int X = (int)(idx % W);
    const long r = idx / W;
    const int Y = (int)(r % H);
    const long n = r / H;
    const float fy = fmaxf(sy * ((float)Y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * ((float)X + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const T* img = logits + (size_t)n * h * w * ld;
    const T* p00 = img + ((size_t)y0 * w + x0) * ld;
    const T* p01 = img + ((size_t)y0 * w + x1) * ld;
    const T* p10 = img + ((size_t)y1 * w + x0) * ld;
    const T* p11 = img + ((size_t)y1 * w + x1) * ld;
    float best = 0.f;
    long long arg = 0;
    for (int c = 0; c < C; ++c) {
      const float v = ly0 * (lx0 * to_f32(p00[c]) + lx1 * to_f32(p01[c])) + ly1 * (lx0 * to_f32(p10[c]) + lx1 * to_f32(p11[c]));
      if (c == 0 || v > best) { best = v; arg = c; }
    }
    if (C == 1) arg = (1.f / (1.f + expf(-best))) > threshold ? 1 : 0;
    mask[idx] = arg;
  }
}

// ------------------------------------------------------------------------------------------ channel sums
// out[c] = sum over a window of the image of x[n,y,x,c]   (bias gradient of ConvTranspose2d, unet_parts.py:53)
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, int ld, int N, int Hs, int Ws, int C,
                                                          int wy, int wx, int wh, int ww, double* sums) {
  constexpr int E = Vec<T>::E;
  __shared__ float red[256 * E];
  const int ppp = C / E;
  const int pg = ppp < 256 ? ppp : 256;
  const int lanes = 256 / pg;
  const int piece_in_group = threadIdx.x % pg, plane = threadIdx.x / pg;
  const int piece = blockIdx.y * 256 + piece_in_group;
  const bool active = plane < lanes && piece < ppp;
  for (int i = threadIdx.x; i < pg * E; i += 256) red[i] = 0.f;
  __syncthreads();
  if (active) {
    float s[E];
#pragma unroll
    for (int i = 0; i < E; ++i) s[i] = 0.f;
    // window pixels in 32-bit arithmetic (the entry point checks N*wh*ww < 2^31), four independent loads in flight
    const int npix = N * wh * ww;
    const int step = gridDim.x * lanes;
    auto addr = [&](int q) {
      const int xx = q % ww, r = q / ww;
      const int yy = r % wh, n = r / wh;
      return x + (((long)n * Hs + wy + yy) * Ws + wx + xx) * ld + piece * E;
    };
    int q = blockIdx.x * lanes + plane;
    for (; q + 3 * step < npix; q += 4 * step) {
      float v0[E], v1[E], v2[E], v3[E];
      Vec<T>::load(addr(q), v0);
      Vec<T>::load(addr(q + step), v1);
      Vec<T>::load(addr(q + 2 * step), v2);
      Vec<T>::load(addr(q + 3 * step), v3);
#pragma unroll
      for (int i = 0; i < E; ++i) s[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
    }
    for (; q < npix; q += step) {
      float v[E];
      Vec<T>::load(addr(q), v);
#pragma unroll
      for (int i = 0; i < E; ++i) s[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < E; ++i) atomicAdd(&red[piece_in_group * E + i], s[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < pg * E; i += 256)
    if (blockIdx.y * 256 * E + i < C) unsafeAtomicAdd(sums + blockIdx.y * 256 * E + i, (double)red[i]);
}

// db[c] = (sum over ALL pixels, from the replicas a conv epilogue accumulated) - (sum over the pad border)
__global__ void bias_from_sums_kernel(const double* __restrict__ sums, int c_total, int c_off,
                                      const double* __restrict__ border, float* __restrict__ db, int c_up) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= c_up) return;
  double s = 0;
  for (int r = 0; r < UNETK_STAT_REPLICAS; ++r) s += sums[((size_t)r * c_total + c_off + c) * 2];
  db[c] = (float)(s - border[c]);
}

__global__ void double_to_float_kernel(const double* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

bool ok16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_act(const void* p, int ld, int c, int dtype) {
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  const int es = unetk_esize(dtype), e = 16 / es;
  if (!p) return UNETK_ERR_ARG;
  if (c <= 0 || c % e || ld < c) return UNETK_ERR_SHAPE;
  if (!ok16(p) || (ld * es) % 16) return UNETK_ERR_ALIGN;
  return UNETK_OK;
}

}  // namespace

#define DISPATCH_T(dtype, ...) UNETK_WITH_T(dtype, __VA_ARGS__)

extern "C" {

int unetk_bn_train_finalize(const double* stats, int64_t count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                            float eps, float* scale, float* shift, float* mean, float* invstd, int c, void* stream) {
  if (!stats || !scale || !shift || !mean || !invstd || c <= 0 || count <= 0) return UNETK_ERR_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return UNETK_ERR_ARG;
  hipLaunchKernelGGL(bn_train_finalize_kernel, dim3((c + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats,
                     (long)count, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps,
                     scale, shift, mean, invstd, c);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                         float eps, float* scale, float* shift, int c, void* stream) {
  if (!running_mean || !running_var || !scale || !shift || c <= 0) return UNETK_ERR_ARG;
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3((c + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, scale, shift, c);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_relu_apply(const void* y, int ldy, const float* scale, const float* shift, void* out, int ldo,
                        int64_t npix, int c, int dtype, void* stream) {
  if (int e = check_act(y, ldy, c, dtype)) return e;
  if (int e = check_act(out, ldo, c, dtype)) return e;
  if (!scale || !shift || npix <= 0) return UNETK_ERR_ARG;
  DISPATCH_T(dtype, {
    const int ppp = c / Vec<T>::E;
    if (ppp <= 256 && 256 % ppp == 0)
      hipLaunchKernelGGL(bn_relu_apply_fixed_kernel<T>, dim3(ew_grid(npix, 256 / ppp, 256 * EW_APPLY_MUL)), dim3(256), 0,
                         (hipStream_t)stream, (const T*)y, ldy, scale, shift, (T*)out, ldo, (long)npix, c);
    else
      hipLaunchKernelGGL(bn_relu_apply_kernel<T>, dim3(ew_grid(npix * ppp)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)y, ldy, scale, shift, (T*)out, ldo, (long)npix, c);
  });
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_relu_bwd_reduce(const void* dout, int lddo, const void* y, int ldy, const float* scale, const float* shift,
                             const float* mean, const float* invstd, double* sums, int64_t npix, int c, int dtype,
                             void* stream) {
  if (int e = check_act(dout, lddo, c, dtype)) return e;
  if (int e = check_act(y, ldy, c, dtype)) return e;
  if (!scale || !shift || !mean || !invstd || !sums || npix <= 0) return UNETK_ERR_ARG;
  const int e_ = unetk_epp(dtype);
  const int ppp = c / e_;
  const int pg = ppp < 256 ? ppp : 256;
  if (256 % pg) return UNETK_ERR_SHAPE;   // channel counts are multiples of 64: pg is a power of two
  const int lanes = 256 / pg;
  long gx = (npix + lanes - 1) / lanes;
  UNETK_KNOB(gcap, "UNETK_BNR_GRID", 1024);
  if (gx > gcap) gx = gcap;
  dim3 grid((unsigned)gx, (unsigned)((ppp + 255) / 256));
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream,
                                       (const T*)dout, lddo, (const T*)y, ldy, scale, shift, mean, invstd, sums,
                                       (long)npix, c));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_relu_bwd_finalize(const double* sums, int64_t count, const float* gamma, const float* invstd, float* dgamma,
                               float* dbeta, float* coef, int c, void* stream) {
  if (!sums || !invstd || !coef || c <= 0 || count <= 0) return UNETK_ERR_ARG;
  hipLaunchKernelGGL(bn_relu_bwd_finalize_kernel, dim3((c + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums,
                     (long)count, gamma, invstd, dgamma, dbeta, coef, c);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_relu_bwd_apply(const void* dout, int lddo, const void* y, int ldy, const float* scale, const float* shift,
                            const float* mean, const float* coef, void* dy, int lddy, int64_t npix, int c, int dtype,
                            void* stream) {
  if (int e = check_act(dout, lddo, c, dtype)) return e;
  if (int e = check_act(y, ldy, c, dtype)) return e;
  if (int e = check_act(dy, lddy, c, dtype)) return e;
  if (!scale || !shift || !mean || !coef || npix <= 0) return UNETK_ERR_ARG;
  DISPATCH_T(dtype, {
    const int ppp = c / Vec<T>::E;
    if (ppp <= 256 && 256 % ppp == 0)
      hipLaunchKernelGGL(bn_relu_bwd_apply_fixed_kernel<T>, dim3(ew_grid(npix, 256 / ppp, 256 * EW_BWD_MUL)), dim3(256), 0,
                         (hipStream_t)stream, (const T*)dout, lddo, (const T*)y, ldy, scale, shift, mean, coef, (T*)dy,
                         lddy, (long)npix, c);
    else
      hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<T>, dim3(ew_grid(npix * ppp)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)dout, lddo, (const T*)y, ldy, scale, shift, mean, coef, (T*)dy, lddy, (long)npix, c);
  });
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_relu_apply_head(const void* y, int ldy, const float* scale, const float* shift, void* out, int ldo,
                             const float* w, const float* bias, int n_classes, void* logits, int ldl, int64_t npix,
                             int c, int dtype, void* stream) {
  if (!y || !scale || !shift || !w || !bias || !logits || npix <= 0) return UNETK_ERR_ARG;
  if (int e = check_act(y, ldy, c, dtype)) return e;
  if (out != nullptr)                       // NULL: only the logits are wanted (see unetk.h)
    if (int e = check_act(out, ldo, c, dtype)) return e;
  const int E = unetk_epp(dtype);
  const int ppp = c / E;
  // a pixel's lanes must share a wave for the shuffle sum: ppp a power of two <= 64
  if (n_classes < 1 || n_classes > kHeadMax || ldl < n_classes || ppp > 64 || (ppp & (ppp - 1))) return UNETK_ERR_SHAPE;
  // 80 registers: six blocks fit a CU, and the pass is latency-bound below that (512 blocks 0.323 ms, 1024 0.261,
  // 1536 0.254, 4096 0.248 at 16 x 640 x 959 x 64: scripts/perf_head.py)
  UNETK_KNOB(hcap, "UNETK_HEAD_GRID", 4096);
  const int grid = ew_grid((npix + 3) / 4, 256 / ppp, hcap);
  // STORE: the activation itself is wanted too (out != NULL); the training forward of the whole net passes NULL
#define HEAD_FWD_S(NC, ST)                                                                                          \
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_apply_head_kernel<T, NC, ST>), dim3(grid), dim3(256), 0,             \
                                       (hipStream_t)stream, (const T*)y, ldy, scale, shift, (T*)out, ldo, w, bias,   \
                                       (T*)logits, ldl, (long)npix, c))
#define HEAD_FWD(NC)                                                                                                \
  if (out != nullptr) { HEAD_FWD_S(NC, true); } else { HEAD_FWD_S(NC, false); }
  switch (n_classes) {
    case 1: HEAD_FWD(1); break;
    case 2: HEAD_FWD(2); break;
    case 3: HEAD_FWD(3); break;
    default: HEAD_FWD(4); break;
  }
#undef HEAD_FWD
#undef HEAD_FWD_S
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

static int head_bwd_check(const void* dl, int lddl, const float* w, int ncls, const void* y, int ldy, int64_t npix,
                          int c, int dtype) {
  if (!dl || !w || !y || npix <= 0 || c <= 0) return UNETK_ERR_ARG;
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  const int E = unetk_epp(dtype);
  if (ncls < 1 || ncls > kHeadMax || lddl < ncls) return UNETK_ERR_SHAPE;   // more classes: conv1x1_dgrad + plain passes
  if (c % E || c / E > 256 || 256 % (c / E) || ldy < c) return UNETK_ERR_SHAPE;
  const int es = unetk_esize(dtype);
  if ((ldy * es) % 16 || (reinterpret_cast<uintptr_t>(y) & 15)) return UNETK_ERR_ALIGN;
  return UNETK_OK;
}

int unetk_bn_relu_bwd_reduce_head(const void* dl, int lddl, const float* w, int n_classes, const void* y, int ldy,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  double* sums, double* head_sums, int64_t npix, int c, int dtype, void* stream) {
  if (int e = head_bwd_check(dl, lddl, w, n_classes, y, ldy, npix, c, dtype)) return e;
  if (!scale || !shift || !mean || !invstd || !sums) return UNETK_ERR_ARG;
  const int E = unetk_epp(dtype);
  UNETK_KNOB(rcap, "UNETK_HEADR_GRID", 1024);
  const int grid = ew_grid((npix + 3) / 4, 256 / (c / E), rcap);
#define HEAD_REDUCE(NC)                                                                                              \
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_bwd_reduce_head_kernel<T, NC>), dim3(grid), dim3(256), 0,            \
                                       (hipStream_t)stream, (const T*)dl, lddl, w, (const T*)y, ldy, scale, shift,   \
                                       mean, invstd, sums, head_sums, (long)npix, c))
  switch (n_classes) {
    case 1: HEAD_REDUCE(1); break;
    case 2: HEAD_REDUCE(2); break;
    case 3: HEAD_REDUCE(3); break;
    default: HEAD_REDUCE(4); break;
  }
#undef HEAD_REDUCE
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_conv1x1_wgrad_finalize(const double* head_sums, float* dw, float* db, int c_in, int n_classes, void* stream) {
  if (!head_sums || !dw || c_in <= 0 || n_classes < 1 || n_classes > kHeadMax) return UNETK_ERR_ARG;
  const int total = n_classes * (c_in + 1);
  hipLaunchKernelGGL(head_wgrad_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, head_sums,
                     dw, db, c_in, n_classes);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_relu_bwd_apply_head(const void* dl, int lddl, const float* w, int n_classes, const void* y, int ldy,
                                 const float* scale, const float* shift, const float* mean, const float* coef, void* dy,
                                 int lddy, int64_t npix, int c, int dtype, void* stream) {
  if (int e = head_bwd_check(dl, lddl, w, n_classes, y, ldy, npix, c, dtype)) return e;
  if (!scale || !shift || !mean || !coef) return UNETK_ERR_ARG;
  if (int e = check_act(dy, lddy, c, dtype)) return e;
  const int E = unetk_epp(dtype);
  UNETK_KNOB(acap, "UNETK_HEADA_GRID", 1536);         // whole rounds of three blocks per CU (1024: 0.468 ms, 1536: 0.450)
  const int grid = ew_grid((npix + 3) / 4, 256 / (c / E), acap);
#define HEAD_APPLY(NC)                                                                                               \
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_bwd_apply_head_kernel<T, NC>), dim3(grid), dim3(256), 0,             \
                                       (hipStream_t)stream, (const T*)dl, lddl, w, (const T*)y, ldy, scale, shift,   \
                                       mean, coef, (T*)dy, lddy, (long)npix, c))
  switch (n_classes) {
    case 1: HEAD_APPLY(1); break;
    case 2: HEAD_APPLY(2); break;
    case 3: HEAD_APPLY(3); break;
    default: HEAD_APPLY(4); break;
  }
#undef HEAD_APPLY
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

static int pool_bwd_check(const void* dskip, int ldds, const void* dpool, int lddp, const void* y, int ldy, int n,
                          int h, int w, int c, int dtype) {
  if (!dskip || !dpool || !y || n <= 0 || h < 2 || w < 2 || c <= 0) return UNETK_ERR_ARG;
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  const int E = unetk_epp(dtype);
  if (c % E || c / E > 256 || 256 % (c / E)) return UNETK_ERR_SHAPE;   // callers fall back to max-pool backward + plain BN backward
  if (ldds < c || lddp < c || ldy < c) return UNETK_ERR_SHAPE;
  const int es = unetk_esize(dtype);
  if ((ldds * es) % 16 || (lddp * es) % 16 || (ldy * es) % 16) return UNETK_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(dskip) | reinterpret_cast<uintptr_t>(dpool) | reinterpret_cast<uintptr_t>(y)) & 15)
    return UNETK_ERR_ALIGN;
  return UNETK_OK;
}

int unetk_bn_relu_bwd_reduce_pool(const void* dskip, int ldds, const void* dpool, int lddp, const void* y, int ldy,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  double* sums, int n, int h, int w, int c, int dtype, void* stream) {
  if (int e = pool_bwd_check(dskip, ldds, dpool, lddp, y, ldy, n, h, w, c, dtype)) return e;
  if (!scale || !shift || !mean || !invstd || !sums) return UNETK_ERR_ARG;
  const int E = unetk_epp(dtype);
  const long nquads = (long)n * ((h + 1) / 2) * ((w + 1) / 2);
  if (nquads >= (1L << 30)) return UNETK_ERR_SHAPE;          // 32-bit quad counters (QuadWalk)
  // two blocks per CU (172 registers): one whole round; fewer blocks also means fewer fp64 atomics at the deep levels
  // (1024 blocks: 0.508 / 0.218 / 0.119 / 0.072 ms at the four encoder levels, 512: 0.498 / 0.213 / 0.112 / 0.060)
  UNETK_KNOB(pcap, "UNETK_POOLR_GRID", 512);
  const int grid = ew_grid(nquads, 256 / (c / E), pcap);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_bwd_reduce_pool_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)dskip, ldds, (const T*)dpool, lddp, (const T*)y, ldy, scale, shift, mean,
                                       invstd, sums, n, h, w, c));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bn_relu_bwd_apply_pool(const void* dskip, int ldds, const void* dpool, int lddp, const void* y, int ldy,
                                 const float* scale, const float* shift, const float* mean, const float* coef, void* dy,
                                 int lddy, int n, int h, int w, int c, int dtype, void* stream) {
  if (int e = pool_bwd_check(dskip, ldds, dpool, lddp, y, ldy, n, h, w, c, dtype)) return e;
  if (!scale || !shift || !mean || !coef || !dy || lddy < c) return UNETK_ERR_ARG;
  const int E = unetk_epp(dtype);
  const long nquads = (long)n * ((h + 1) / 2) * ((w + 1) / 2);
  if (nquads >= (1L << 30)) return UNETK_ERR_SHAPE;          // 32-bit quad counters (QuadWalk)
  const int grid = ew_grid(nquads, 256 / (c / E), 256 * EW_BWD_MUL);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_relu_bwd_apply_pool_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)dskip, ldds, (const T*)dpool, lddp, (const T*)y, ldy, scale, shift, mean,
                                       coef, (T*)dy, lddy, n, h, w, c));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_maxpool2x2_fwd(const void* x, int ldx, void* out, int ldo, int n, int h, int w, int c, int dtype,
                         void* stream) {
  if (int e = check_act(x, ldx, c, dtype)) return e;
  if (int e = check_act(out, ldo, c, dtype)) return e;
  if (n <= 0 || h < 2 || w < 2) return UNETK_ERR_SHAPE;
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(ew_grid((long)n * (h / 2) * (w / 2) * (c / Vec<T>::E))),
                                       dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (T*)out, ldo, n, h, w, c));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_maxpool2x2_bwd(const void* x, int ldx, const void* dpool, int lddp, const void* dskip, int ldds, void* dx,
                         int lddx, int n, int h, int w, int c, int dtype, void* stream) {
  if (int e = check_act(x, ldx, c, dtype)) return e;
  if (int e = check_act(dpool, lddp, c, dtype)) return e;
  if (int e = check_act(dx, lddx, c, dtype)) return e;
  if (dskip)
    if (int e = check_act(dskip, ldds, c, dtype)) return e;
  if (n <= 0 || h < 2 || w < 2) return UNETK_ERR_SHAPE;
  const long items = (long)n * ((h + 1) / 2) * ((w + 1) / 2) * (c / unetk_epp(dtype));
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_grid(items)), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)x, ldx, (const T*)dpool, lddp, (const T*)dskip, ldds, (T*)dx, lddx, n,
                                       h, w, c));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bilinear2x_fwd(const void* x, int ldx, void* out, int ldo, int n, int h, int w, int c, int hs, int ws,
                         int pad_t, int pad_l, int dtype, void* stream) {
  if (int e = check_act(x, ldx, c, dtype)) return e;
  if (int e = check_act(out, ldo, c, dtype)) return e;
  if (n <= 0 || h <= 0 || w <= 0 || pad_t < 0 || pad_l < 0 || pad_t + 2 * h > hs || pad_l + 2 * w > ws)
    return UNETK_ERR_SHAPE;
  const int per_row = ws * (c / unetk_epp(dtype));
  if ((long)n * hs > 0x7fffffffL || (long)h * w >= 0x7fffffffL) return UNETK_ERR_SHAPE;
  const int gx = (per_row + 255) / 256 > 8 ? 8 : (per_row + 255) / 256;
  const long rows = (long)n * hs;
  const int gy = (int)(rows > 4096 ? 4096 : rows);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_fwd_kernel<T>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)x, ldx, (T*)out, ldo, n, h, w, c, hs, ws, pad_t, pad_l));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_bilinear2x_bwd(const void* dcat, int ldd, void* dx, int lddx, int n, int h, int w, int c, int hs, int ws,
                         int pad_t, int pad_l, int dtype, void* stream) {
  if (int e = check_act(dcat, ldd, c, dtype)) return e;
  if (int e = check_act(dx, lddx, c, dtype)) return e;
  if (n <= 0 || h <= 0 || w <= 0 || pad_t < 0 || pad_l < 0 || pad_t + 2 * h > hs || pad_l + 2 * w > ws)
    return UNETK_ERR_SHAPE;
  const long items = (long)n * h * w * (c / unetk_epp(dtype));
  DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_bwd_kernel<T>, dim3(ew_grid(items)), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)dcat, ldd, (T*)dx, lddx, n, h, w, c, hs, ws, pad_t, pad_l));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_copy_channels(const void* src, int ld_src, void* dst, int ld_dst, int64_t npix, int c, int dtype,
                        void* stream) {
  if (int e = check_act(src, ld_src, c, dtype)) return e;
  if (int e = check_act(dst, ld_dst, c, dtype)) return e;
  if (npix <= 0) return UNETK_ERR_SHAPE;
  DISPATCH_T(dtype, hipLaunchKernelGGL(copy_channels_kernel<T>, dim3(ew_grid(npix * (c / Vec<T>::E))), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)src, ld_src, (T*)dst, ld_dst, (long)npix, c));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_zero_pad_border(void* buf, int ld, int n, int hs, int ws, int c, int pad_t, int pad_l, int h_up, int w_up,
                          int dtype, void* stream) {
  if (int e = check_act(buf, ld, c, dtype)) return e;
  if (n <= 0 || pad_t < 0 || pad_l < 0 || pad_t + h_up > hs || pad_l + w_up > ws) return UNETK_ERR_SHAPE;
  if (h_up == hs && w_up == ws) return UNETK_OK;   // nothing to pad
  if ((long)hs * ws >= 0x7fffffffL) return UNETK_ERR_SHAPE;
  const long items = (long)n * ((long)hs * ws - (long)h_up * w_up) * (c / unetk_epp(dtype));
  DISPATCH_T(dtype, hipLaunchKernelGGL(zero_border_kernel<T>, dim3(ew_grid(items)), dim3(256), 0, (hipStream_t)stream,
                                       (T*)buf, ld, n, hs, ws, c, pad_t, pad_l, h_up, w_up));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_channel_sum(const void* x, int ld, int n, int hs, int ws, int c, int win_y, int win_x, int win_h, int win_w,
                      double* scratch, float* out, int dtype, void* stream) {
  if (int e = check_act(x, ld, c, dtype)) return e;
  if (!scratch || !out || n <= 0 || win_y < 0 || win_x < 0 || win_h <= 0 || win_w <= 0 || win_y + win_h > hs ||
      win_x + win_w > ws || (long)n * win_h * win_w >= 0x7fffffffL - 4L * kMaxGrid * 256)
    return UNETK_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(scratch, 0, (size_t)c * sizeof(double), s) != hipSuccess) return UNETK_ERR_LAUNCH;
  const int e_ = unetk_epp(dtype);
  const int ppp = c / e_;
  const int pg = ppp < 256 ? ppp : 256;
  if (256 % pg) return UNETK_ERR_SHAPE;
  const int lanes = 256 / pg;
  long gx = ((long)n * win_h * win_w + lanes - 1) / lanes;
  if (gx > kMaxGrid) gx = kMaxGrid;
  dim3 grid((unsigned)gx, (unsigned)((ppp + 255) / 256));
  DISPATCH_T(dtype, hipLaunchKernelGGL(channel_sum_kernel<T>, grid, dim3(256), 0, s, (const T*)x, ld, n, hs, ws, c,
                                       win_y, win_x, win_h, win_w, scratch));
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(double_to_float_kernel, dim3((c + 255) / 256), dim3(256), 0, s, (const double*)scratch, out, c);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}
int unetk_convT2x2_bias_grad_from_sums(const double* sums, int c_total, int c_off, const void* dup, int ld, int n,
                                       int hs, int ws, int c_up, int pad_t, int pad_l, int h_up, int w_up,
                                       double* scratch, float* db, int dtype, void* stream) {
  if (int e = check_act(dup, ld, c_up, dtype)) return e;
  if (!sums || !scratch || !db || n <= 0 || c_off < 0 || c_off + c_up > c_total || pad_t < 0 || pad_l < 0 ||
      pad_t + h_up > hs || pad_l + w_up > ws || (long)n * hs * ws >= 0x7fffffffL - 4L * kMaxGrid * 256)
    return UNETK_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(scratch, 0, (size_t)c_up * sizeof(double), s) != hipSuccess) return UNETK_ERR_LAUNCH;
  const int e_ = unetk_epp(dtype);
  const int ppp = c_up / e_;
  const int pg = ppp < 256 ? ppp : 256;
  if (256 % pg) return UNETK_ERR_SHAPE;
  const int lanes = 256 / pg;
  // the F.pad border (unet_parts.py:62-63) as up to four strips: top rows, bottom rows, left and right columns
  const int strips[4][4] = {{0, 0, pad_t, ws},
                            {pad_t + h_up, 0, hs - pad_t - h_up, ws},
                            {pad_t, 0, h_up, pad_l},
                            {pad_t, pad_l + w_up, h_up, ws - pad_l - w_up}};
  for (int k = 0; k < 4; ++k) {
    const int wy = strips[k][0], wx = strips[k][1], wh = strips[k][2], ww = strips[k][3];
    if (wh <= 0 || ww <= 0) continue;
    long gx = ((long)n * wh * ww + lanes - 1) / lanes;
    if (gx > kMaxGrid) gx = kMaxGrid;
    dim3 grid((unsigned)gx, (unsigned)((ppp + 255) / 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(channel_sum_kernel<T>, grid, dim3(256), 0, s, (const T*)dup, ld, n, hs, ws, c_up,
                                         wy, wx, wh, ww, scratch));
    UNETK_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(bias_from_sums_kernel, dim3((c_up + 255) / 256), dim3(256), 0, s, sums, c_total, c_off,
                     (const double*)scratch, db, c_up);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_predict_mask(const void* logits, int ld, int n, int h, int w, int n_classes, int out_h, int out_w,
                       float threshold, int64_t* mask, int dtype, void* stream) {
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  if (!logits || !mask) return UNETK_ERR_ARG;
  if (n <= 0 || h <= 0 || w <= 0 || n_classes <= 0 || ld < n_classes || out_h <= 0 || out_w <= 0) return UNETK_ERR_SHAPE;
  const long total = (long)n * out_h * out_w;
  DISPATCH_T(dtype, hipLaunchKernelGGL(predict_mask_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)logits, ld, n, h, w, n_classes, out_h, out_w, threshold,
                                       (long long*)mask));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}
int unetk_bn_relu_apply_pool(const void* y, int ldy, const float* scale, const float* shift, void* out, int ldo,
                             void* pool, int ldp, int n, int h, int w, int c, int dtype, void* stream) {
  if (int e = check_act(y, ldy, c, dtype)) return e;
  if (int e = check_act(out, ldo, c, dtype)) return e;
  if (int e = check_act(pool, ldp, c, dtype)) return e;
  if (!scale || !shift) return UNETK_ERR_ARG;
  if (n <= 0 || h < 2 || w < 2) return UNETK_ERR_SHAPE;
  const int ppp = c / unetk_epp(dtype);
  if (ppp > 256 || 256 % ppp) return UNETK_ERR_SHAPE;   // callers fall back to apply + maxpool2x2_fwd
  const long nquads = (long)n * ((h + 1) / 2) * ((w + 1) / 2);
  UNETK_KNOB(fcap, "UNETK_POOLF_GRID", 256 * 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_relu_apply_pool_kernel<T>, dim3(ew_grid(nquads, 256 / ppp, fcap)), dim3(256),
                                       0, (hipStream_t)stream, (const T*)y, ldy, scale, shift, (T*)out, ldo, (T*)pool,
                                       ldp, n, h, w, c));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
