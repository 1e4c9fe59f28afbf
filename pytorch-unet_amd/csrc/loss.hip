// Segmentation loss of the reference's train step (train.py:81,102-111) and the dice helpers
// (utils/dice_score.py:5-28) as wavefront reductions.
//
//   n_classes > 1 : mean CrossEntropy(logits, mask) + 1 - dice(softmax(logits), one_hot(mask))
//   n_classes == 1: mean BCEWithLogits(z, mask)     + 1 - dice(sigmoid(z), mask)
//
// In both cases dice_loss() calls dice_coeff(..., reduce_batch_first=True) on a 3-D tensor, i.e. ONE global
// ratio (2*sum(p*t) + eps) / (sum(p) + sum(t) + eps) over all N, C, H, W  (dice_score.py:10-16).
// The forward reads each logit once and produces four fp64 sums; a 1-thread finalize turns them into the
// scalar loss and the two coefficients the backward needs; the backward re-reads the logits once and writes
// dlogits.  The eager reference takes ~10 kernels and several [N,C,H,W] fp32 temporaries for the same thing.
#include "common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
  // wave reduce
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// sums[0] = sum CE (or BCE), sums[1] = sum p*t, sums[2] = sum p, sums[3] = sum t
template <typename T>
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const T* __restrict__ z, int ldz,
                                                           const long long* __restrict__ mask, long npix, int ncls,
                                                           double* sums) {
  __shared__ double sh[4];
  double s_ce = 0, s_pt = 0, s_p = 0, s_t = 0;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    const T* zp = z + pix * ldz;
    const long long t = mask[pix];
    if (ncls == 1) {
      const float v = to_f32(zp[0]);
      const float tf = (float)t;
      const float p = 1.f / (1.f + __expf(-v));
      s_ce += fmaxf(v, 0.f) - v * tf + log1pf(__expf(-fabsf(v)));
      s_pt += p * tf;
      s_p += p;
      s_t += tf;
    } else {
      float m = -INFINITY;
      for (int k = 0; k < ncls; ++k) m = fmaxf(m, to_f32(zp[k]));
      float se = 0.f, zt = 0.f;
      for (int k = 0; k < ncls; ++k) {
        const float v = to_f32(zp[k]);
        se += __expf(v - m);
        if (k == t) zt = v;
      }
      const bool valid = t >= 0 && t < ncls;
      if (valid) {
        s_ce += __logf(se) - (zt - m);
        s_pt += __expf(zt - m) / se;
        s_t += 1.0;
      }
      s_p += 1.0;   // softmax rows sum to one
    }
  }
  s_ce = block_sum(s_ce, sh);
  s_pt = block_sum(s_pt, sh);
  s_p = block_sum(s_p, sh);
  s_t = block_sum(s_t, sh);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(sums + 0, s_ce);
    unsafeAtomicAdd(sums + 1, s_pt);
    unsafeAtomicAdd(sums + 2, s_p);
    unsafeAtomicAdd(sums + 3, s_t);
  }
}

// The two-class case of the benchmark configurations (packed logits, ldz == 2): four pixels per thread and step -- one
// 16-byte (bf16: 8 logits) resp. two 16-byte (fp32) logit loads and two 16-byte mask loads -- and a grid of at most 512
// workgroups: the generic kernel above spends most of its 0.13 ms at [16, 640, 959] in 8192 fp64 atomics on ONE cache
// line (2048 workgroups x 4 sums; MI355X_MICROARCH "Global float atomics") and in 2-byte loads.  Same arithmetic per
// pixel as the generic path.
template <typename T>
__global__ __launch_bounds__(256) void seg_loss_fwd2_kernel(const T* __restrict__ z, const long long* __restrict__ mask,
                                                            long npix, double* sums) {
  __shared__ double sh[4];
  double s_ce = 0, s_pt = 0, s_p = 0, s_t = 0;
  const long nquad = npix >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (long)gridDim.x * blockDim.x) {
    float v[8];
    if constexpr (sizeof(T) == 2) {
      const typename H16<T>::x8 r = *reinterpret_cast<const typename H16<T>::x8*>(z + q * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(z + q * 8), b = *reinterpret_cast<const f32x4*>(z + q * 8 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    typedef __attribute__((ext_vector_type(2))) long long i64x2;
    const i64x2 t01 = *reinterpret_cast<const i64x2*>(mask + q * 4), t23 = *reinterpret_cast<const i64x2*>(mask + q * 4 + 2);
    const long long t[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float z0 = v[2 * i], z1 = v[2 * i + 1];
      const float m = fmaxf(fmaxf(-INFINITY, z0), z1);
      const float se = __expf(z0 - m) + __expf(z1 - m);
      const bool valid = t[i] >= 0 && t[i] < 2;
      const float zt = t[i] == 0 ? z0 : (t[i] == 1 ? z1 : 0.f);
      if (valid) {
        s_ce += __logf(se) - (zt - m);
        s_pt += __expf(zt - m) / se;
        s_t += 1.0;
      }
      s_p += 1.0;
    }
  }
  // the last npix % 4 pixels
  if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {
    const long pix = (nquad << 2) + threadIdx.x;
    const float z0 = to_f32(z[pix * 2]), z1 = to_f32(z[pix * 2 + 1]);
    const long long t = mask[pix];
    const float m = fmaxf(fmaxf(-INFINITY, z0), z1);
    const float se = __expf(z0 - m) + __expf(z1 - m);
    const float zt = t == 0 ? z0 : (t == 1 ? z1 : 0.f);
    if (t >= 0 && t < 2) {
      s_ce += __logf(se) - (zt - m);
      s_pt += __expf(zt - m) / se;
      s_t += 1.0;
    }
    s_p += 1.0;
  }
  s_ce = block_sum(s_ce, sh);
  s_pt = block_sum(s_pt, sh);
  s_p = block_sum(s_p, sh);
  s_t = block_sum(s_t, sh);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(sums + 0, s_ce);
    unsafeAtomicAdd(sums + 1, s_pt);
    unsafeAtomicAdd(sums + 2, s_p);
    unsafeAtomicAdd(sums + 3, s_t);
  }
}

// out[0] = loss, out[1] = dice, out[2] = alpha, out[3] = beta  where d(1-dice)/dp = alpha*t + beta
__global__ void seg_loss_finalize_kernel(const double* sums, long npix, float eps, float* out) {
  const double inter = 2.0 * sums[1];
  const double sets = sums[2] + sums[3];
  double dice, alpha, beta;
  if (sets == 0.0) {          // torch.where(sets_sum == 0, inter, sets_sum): ratio is identically 1
    dice = (inter + eps) / (inter + eps);
    alpha = 0.0;
    beta = 0.0;
  } else {
    dice = (inter + eps) / (sets + eps);
    alpha = -2.0 / (sets + eps);
    beta = (inter + eps) / ((sets + eps) * (sets + eps));
  }
  out[0] = (float)(sums[0] / (double)npix + 1.0 - dice);
  out[1] = (float)dice;
  out[2] = (float)alpha;
  out[3] = (float)beta;
}

template <typename T>
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const T* __restrict__ z, int ldz,
                                                           const long long* __restrict__ mask, long npix, int ncls,
                                                           const float* __restrict__ fin, const float* gout,
                                                           T* __restrict__ dz, int lddz) {
  const float go = gout ? *gout : 1.f;
  const float alpha = fin[2], beta = fin[3];
  const float inv_m = 1.f / (float)npix;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    const T* zp = z + pix * ldz;
    T* dp = dz + pix * lddz;
    const long long t = mask[pix];
    if (ncls == 1) {
      const float v = to_f32(zp[0]);
      const float tf = (float)t;
      const float p = 1.f / (1.f + __expf(-v));
      // d/dz [ BCE/M + (1 - dice) ] ;  d(1-dice)/dp = alpha*t + beta ; dp/dz = p(1-p)
      dp[0] = from_f32<T>(go * ((p - tf) * inv_m + (alpha * tf + beta) * p * (1.f - p)));
    } else {
      float m = -INFINITY;
      for (int k = 0; k < ncls; ++k) m = fmaxf(m, to_f32(zp[k]));
      float se = 0.f, zt = 0.f;
      for (int k = 0; k < ncls; ++k) {
        const float v = to_f32(zp[k]);
        se += __expf(v - m);
        if (k == t) zt = v;
      }
      const bool valid = t >= 0 && t < ncls;
      const float inv = 1.f / se;
      const float pt = valid ? __expf(zt - m) * inv : 0.f;
      for (int k = 0; k < ncls; ++k) {
        const float p = __expf(to_f32(zp[k]) - m) * inv;
        const float oh = (valid && k == t) ? 1.f : 0.f;
        // CE: (p - onehot)/M ; dice through softmax: p_k * (g_k - sum_j p_j g_j), g = alpha*onehot + beta
        const float g = (valid ? (p - oh) * inv_m : 0.f) + alpha * p * (oh - pt);
        dp[k] = from_f32<T>(go * g);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ dice_coeff
// sums[g] = { sum x*t, sum x, sum t } over group g (a group = the dims dice_score.py:10 sums over)
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                        long group_size, double* sums) {
  __shared__ double sh[4];
  const long g = blockIdx.y;
  const float* xg = x + g * group_size;
  const float* tg = t + g * group_size;
  double a = 0, b = 0, c = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < group_size; i += (long)gridDim.x * blockDim.x) {
    const float xv = xg[i], tv = tg[i];
    a += xv * tv;
    b += xv;
    c += tv;
  }
  a = block_sum(a, sh);
  b = block_sum(b, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(sums + g * 3 + 0, a);
    unsafeAtomicAdd(sums + g * 3 + 1, b);
    unsafeAtomicAdd(sums + g * 3 + 2, c);
  }
}

// dice = mean_g (2*I_g + eps) / (S_g + eps), S_g := (S_g == 0 ? 2*I_g : S_g);  coef[g] = {a_g, b_g} with
// d dice / d x[g][i] = a_g * t + b_g
__global__ void dice_finalize_kernel(const double* sums, int ngroups, float eps, float* dice_out, float* coef) {
  double acc = 0;
  for (int g = 0; g < ngroups; ++g) {
    const double inter = 2.0 * sums[g * 3 + 0];
    const double sets = sums[g * 3 + 1] + sums[g * 3 + 2];
    if (sets == 0.0) {
      acc += (inter + eps) / (inter + eps);
      if (coef) { coef[g * 2] = 0.f; coef[g * 2 + 1] = 0.f; }
    } else {
      acc += (inter + eps) / (sets + eps);
      if (coef) {
        coef[g * 2] = (float)(2.0 / (sets + eps) / ngroups);
        coef[g * 2 + 1] = (float)(-(inter + eps) / ((sets + eps) * (sets + eps)) / ngroups);
      }
    }
  }
  *dice_out = (float)(acc / ngroups);
}

__global__ void dice_bwd_kernel(const float* __restrict__ t, const float* __restrict__ coef, const float* gout,
                                float sign, long group_size, float* __restrict__ dx) {
  const long g = blockIdx.y;
  const float go = (gout ? *gout : 1.f) * sign;
  const float a = coef[g * 2] * go, b = coef[g * 2 + 1] * go;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < group_size; i += (long)gridDim.x * blockDim.x)
    dx[g * group_size + i] = a * t[g * group_size + i] + b;
}

// ------------------------------------------------------------------------------------------ evaluation metric
// evaluate.py:31-36 in one pass: pred = argmax_c logits (first maximum, like torch.argmax), then per (image, class)
//   inter = #{pred == c and true == c},  npred = #{pred == c},  ntrue = #{true == c}
// Integer counts (wave ballots, no float atomics): exact and order independent.
template <typename T>
__global__ __launch_bounds__(256) void eval_counts_kernel(const T* __restrict__ logits, int ld, const long long* __restrict__ mask,
                                                          long hw, int C, unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned cnt[];           // [4 waves][C][3]
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * C * 3; i += 256) cnt[i] = 0u;
  __syncthreads();
  unsigned* mine = cnt + wid * C * 3;
  const T* lg = logits + (size_t)n * hw * ld;
  const long long* mk = mask + (size_t)n * hw;
  for (long p0 = (long)blockIdx.x * 256 + wid * 64; p0 < hw; p0 += (long)gridDim.x * 256) {
    const long p = p0 + lane;
    int pred = -1, tru = -1;
    if (p < hw) {
      const T* row = lg + p * ld;
      float best = to_f32(row[0]);
      pred = 0;
      for (int c = 1; c < C; ++c) {
        const float v = to_f32(row[c]);
        if (v > best) { best = v; pred = c; }
      }
      tru = (int)mk[p];
    }
    for (int c = 0; c < C; ++c) {
      const unsigned long long bp = __ballot(pred == c), bt = __ballot(tru == c);
      if (lane == 0) {
        mine[c * 3 + 0] += __popcll(bp & bt);
        mine[c * 3 + 1] += __popcll(bp);
        mine[c * 3 + 2] += __popcll(bt);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 3; i += 256) {
    const unsigned v = cnt[i] + cnt[C * 3 + i] + cnt[2 * C * 3 + i] + cnt[3 * C * 3 + i];
    if (v) atomicAdd(counts + (size_t)n * C * 3 + i, (unsigned long long)v);
  }
}

// multiclass_dice_coeff(one_hot(pred)[:, 1:], one_hot(true)[:, 1:], reduce_batch_first=False)  (evaluate.py:36,
// dice_score.py:5-22): one dice per (image, class >= first_class), sets == 0 -> dice 1, mean over the groups
__global__ void eval_dice_finalize_kernel(const unsigned long long* __restrict__ counts, int N, int C, int first_class,
                                          float eps, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double acc = 0.0;
  int groups = 0;
  for (int n = 0; n < N; ++n)
    for (int c = first_class; c < C; ++c) {
      const unsigned long long* k = counts + ((size_t)n * C + c) * 3;
      const float inter = 2.f * (float)k[0];
      float sets = (float)k[1] + (float)k[2];
      if (sets == 0.f) sets = inter;
      acc += (double)((inter + eps) / (sets + eps));
      ++groups;
    }
  out[0] = groups ? (float)(acc / groups) : 0.f;
}


// ---- multi-class variants that move the logits as whole tiles.  With contiguous logits (pitch == n_classes) a
// block copies the 256 x n_classes values of 256 consecutive pixels with 16-byte loads into LDS (and, backward, the
// gradients back out with 16-byte stores); each lane then works on its pixel's row in LDS (row stride n_classes
// elements: odd for the 21-class case, so conflict-free).  The per-pixel kernels above read n_classes scattered 2/4-byte
// values per lane, three times over: 2.1 ms backward and 0.8 ms forward for [16,512,512,21] fp32, against ~0.2 ms of
// HBM time.
constexpr int kLossTile = 256;

template <typename T>
__device__ __forceinline__ void loss_tile_load(const T* __restrict__ z, long tile, long npix, int ncls, T* lds) {
  constexpr int PER = 16 / (int)sizeof(T);
  const long e0 = tile * kLossTile * ncls;                 // first element of the tile
  const int nelem = (int)(min((long)kLossTile, npix - tile * kLossTile) * ncls);
  if (nelem == kLossTile * ncls) {
    const uint4* src = reinterpret_cast<const uint4*>(z + e0);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int i = threadIdx.x; i < nelem / PER; i += 256) dst[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < nelem; i += 256) lds[i] = z[e0 + i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void seg_loss_fwd_tiled_kernel(const T* __restrict__ z,
                                                                 const long long* __restrict__ mask, long npix,
                                                                 int ncls, double* sums) {
  extern __shared__ __attribute__((aligned(16))) char loss_lds[];
  T* tile_z = reinterpret_cast<T*>(loss_lds);
  __shared__ double sh[4];
  double s_ce = 0, s_pt = 0, s_p = 0, s_t = 0;
  const long ntiles = (npix + kLossTile - 1) / kLossTile;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    loss_tile_load<T>(z, tile, npix, ncls, tile_z);
    __syncthreads();
    const long pix = tile * kLossTile + threadIdx.x;
    if (pix < npix) {
      const T* zp = tile_z + threadIdx.x * ncls;
      const long long t = mask[pix];
      float m = -INFINITY;
      for (int k = 0; k < ncls; ++k) m = fmaxf(m, to_f32(zp[k]));
      float se = 0.f, zt = 0.f;
      for (int k = 0; k < ncls; ++k) {
        const float v = to_f32(zp[k]);
        se += __expf(v - m);
        if (k == t) zt = v;
      }
      if (t >= 0 && t < ncls) {
        s_ce += __logf(se) - (zt - m);
        s_pt += __expf(zt - m) / se;
        s_t += 1.0;
      }
      s_p += 1.0;   // softmax rows sum to one
    }
  }
  s_ce = block_sum(s_ce, sh);
  s_pt = block_sum(s_pt, sh);
  s_p = block_sum(s_p, sh);
  s_t = block_sum(s_t, sh);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(sums + 0, s_ce);
    unsafeAtomicAdd(sums + 1, s_pt);
    unsafeAtomicAdd(sums + 2, s_p);
    unsafeAtomicAdd(sums + 3, s_t);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void seg_loss_bwd_tiled_kernel(const T* __restrict__ z,
                                                                 const long long* __restrict__ mask, long npix,
                                                                 int ncls, const float* __restrict__ fin,
                                                                 const float* gout, T* __restrict__ dz) {
  extern __shared__ __attribute__((aligned(16))) char loss_lds[];
  constexpr int PER = 16 / (int)sizeof(T);
  T* tile_z = reinterpret_cast<T*>(loss_lds);
  T* tile_d = tile_z + ((kLossTile * ncls + PER - 1) / PER) * PER;
  const float go = gout ? *gout : 1.f;
  const float alpha = fin[2];
  const float inv_m = 1.f / (float)npix;
  const long ntiles = (npix + kLossTile - 1) / kLossTile;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    loss_tile_load<T>(z, tile, npix, ncls, tile_z);
    __syncthreads();
    const long pix = tile * kLossTile + threadIdx.x;
    if (pix < npix) {
      const T* zp = tile_z + threadIdx.x * ncls;
      T* dp = tile_d + threadIdx.x * ncls;
      const long long t = mask[pix];
      float m = -INFINITY;
      for (int k = 0; k < ncls; ++k) m = fmaxf(m, to_f32(zp[k]));
      float se = 0.f, zt = 0.f;
      for (int k = 0; k < ncls; ++k) {
        const float v = to_f32(zp[k]);
        se += __expf(v - m);
        if (k == t) zt = v;
      }
      const bool valid = t >= 0 && t < ncls;
      const float inv = 1.f / se;
      const float pt = valid ? __expf(zt - m) * inv : 0.f;
      for (int k = 0; k < ncls; ++k) {
        const float p = __expf(to_f32(zp[k]) - m) * inv;
        const float oh = (valid && k == t) ? 1.f : 0.f;
        const float g = (valid ? (p - oh) * inv_m : 0.f) + alpha * p * (oh - pt);
        dp[k] = from_f32<T>(go * g);
      }
    }
    __syncthreads();
    const long e0 = tile * kLossTile * ncls;
    const int nelem = (int)(min((long)kLossTile, npix - tile * kLossTile) * ncls);
    if (nelem == kLossTile * ncls) {
      uint4* dst = reinterpret_cast<uint4*>(dz + e0);
      const uint4* src = reinterpret_cast<const uint4*>(tile_d);
      for (int i = threadIdx.x; i < nelem / PER; i += 256) dst[i] = src[i];
    } else {
      for (int i = threadIdx.x; i < nelem; i += 256) dz[e0 + i] = tile_d[i];
    }
  }
}

// contiguous rows, 16-byte aligned, more than one class, tiles that fit the default 64 KiB of LDS
bool loss_tiled_ok(const void* z, int ldz, const void* dz, int lddz, int ncls, int es, bool bwd) {
  if (ncls < 2 || ldz != ncls || (bwd && lddz != ncls)) return false;
  if ((reinterpret_cast<uintptr_t>(z) & 15) || (bwd && (reinterpret_cast<uintptr_t>(dz) & 15))) return false;
  const size_t bytes = ((size_t)kLossTile * ncls * es + 15) / 16 * 16;
  return (bwd ? 2 * bytes : bytes) <= 60 * 1024;
}


int eval_dice_check(const void* logits, int ld, const int64_t* mask, int n, int64_t hw, int n_classes, int first_class,
                    const float* dice_out, int dtype) {
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  if (!logits || !mask || !dice_out) return UNETK_ERR_ARG;
  if (n <= 0 || n > 65535 || hw <= 0 || n_classes < 2 || n_classes > 1024 || ld < n_classes || first_class < 0 ||
      first_class >= n_classes)
    return UNETK_ERR_SHAPE;
  return UNETK_OK;
}

// memset of the counters + counting pass + finalize, all on `s`; `counts` holds n * n_classes * 3 u64
int eval_dice_launch(const void* logits, int ld, const int64_t* mask, int n, int64_t hw, int n_classes, int first_class,
                     float eps, unsigned long long* counts, float* dice_out, int dtype, hipStream_t s) {
  if (hipMemsetAsync(counts, 0, (size_t)n * n_classes * 3 * sizeof(unsigned long long), s) != hipSuccess)
    return UNETK_ERR_LAUNCH;
  long gx = (hw + 255) / 256;
  const long cap = n >= 2048 ? 1 : 2048 / n;
  if (gx > cap) gx = cap;
  const size_t lds = (size_t)4 * n_classes * 3 * sizeof(unsigned);
  UNETK_WITH_T(dtype, hipLaunchKernelGGL(eval_counts_kernel<T>, dim3((unsigned)gx, (unsigned)n), dim3(256), lds, s, (const T*)logits,
                       ld, (const long long*)mask, (long)hw, n_classes, counts));
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(eval_dice_finalize_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long*)counts, n,
                     n_classes, first_class, eps, dice_out);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

// ---- unetk_eval_dice_noalloc: the per-(image, class) counters live in this file-scope device array instead of a
// caller workspace (96 KiB per GPU the library runs on, held until the process ends).  In HOST code the array's name
// is only a stand-in for the symbol, not a device address: the address is asked for once per device, with that device
// current, and only that pointer is handed to hipMemsetAsync and to the kernels.
__device__ __align__(256) unsigned long long g_eval_scratch[UNETK_EVAL_NOALLOC_MAX_GROUPS * 3];
constexpr int kMaxDevices = 64;
unsigned long long* g_eval_scratch_ptr[kMaxDevices];     // per device; 0 = not asked yet

// The scratch of the device that OWNS `logits` -- which must also be the current device (the stream belongs to it and
// the kernels are launched on it): a caller whose current device differs would otherwise get another GPU's symbol
// address.  Returns nullptr with *err set: UNETK_ERR_ARG for the device mismatch, UNETK_ERR_LAUNCH for a HIP failure.
unsigned long long* eval_scratch_for(const void* logits, int* err) {
  int dev = -1;
  *err = UNETK_ERR_LAUNCH;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, logits) != hipSuccess) {
    (void)hipGetLastError();
    *err = UNETK_ERR_ARG;
    return nullptr;
  }
  if (attr.type != hipMemoryTypeDevice || attr.device != dev) {
    *err = UNETK_ERR_ARG;
    return nullptr;
  }
  unsigned long long* p = __atomic_load_n(&g_eval_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (p) return p;
  // first call on this device: the symbol lookup loads the code object (it may allocate, and is refused under stream
  // capture) -- include/unetk.h asks for one un-captured warm-up call per device
  void* addr = nullptr;
  if (hipGetSymbolAddress(&addr, HIP_SYMBOL(g_eval_scratch)) != hipSuccess || !addr) return nullptr;
  p = static_cast<unsigned long long*>(addr);
  __atomic_store_n(&g_eval_scratch_ptr[dev], p, __ATOMIC_RELEASE);   // racing first calls store the same address
  return p;
}

}  // namespace

extern "C" {

int unetk_seg_loss_fwd(const void* logits, int ldz, const int64_t* mask, int64_t npix, int n_classes, float eps,
                       double* sums4, float* out4, int dtype, void* stream) {
  if (!logits || !mask || !sums4 || !out4 || npix <= 0 || n_classes <= 0 || ldz < n_classes) return UNETK_ERR_ARG;
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums4, 0, 4 * sizeof(double), s) != hipSuccess) return UNETK_ERR_LAUNCH;
  long g = (npix + 255) / 256;
  if (g > 2048) g = 2048;
  const int es = unetk_esize(dtype);
  if (n_classes > 2 && loss_tiled_ok(logits, ldz, nullptr, 0, n_classes, es, false)) {
    const size_t lds = ((size_t)kLossTile * n_classes * es + 15) / 16 * 16;
    UNETK_WITH_T(dtype, hipLaunchKernelGGL(seg_loss_fwd_tiled_kernel<T>, dim3((unsigned)g), dim3(256), lds, s, (const T*)logits,
                         (const long long*)mask, (long)npix, n_classes, sums4));
  } else if (n_classes == 2 && ldz == 2 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(mask) & 15) == 0) {
    long g2 = (npix / 4 + 255) / 256;
    g2 = g2 < 1 ? 1 : (g2 > 512 ? 512 : g2);
    UNETK_WITH_T(dtype, hipLaunchKernelGGL(seg_loss_fwd2_kernel<T>, dim3((unsigned)g2), dim3(256), 0, s, (const T*)logits,
                         (const long long*)mask, (long)npix, sums4));
  } else {
    UNETK_WITH_T(dtype, hipLaunchKernelGGL(seg_loss_fwd_kernel<T>, dim3((unsigned)g), dim3(256), 0, s, (const T*)logits, ldz,
                                           (const long long*)mask, (long)npix, n_classes, sums4));
  }
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(1), 0, s, (const double*)sums4, (long)npix, eps, out4);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_seg_loss_bwd(const void* logits, int ldz, const int64_t* mask, int64_t npix, int n_classes,
                       const float* out4, const float* grad_out, void* dlogits, int lddz, int dtype, void* stream) {
  if (!logits || !mask || !out4 || !dlogits || npix <= 0 || n_classes <= 0 || ldz < n_classes || lddz < n_classes)
    return UNETK_ERR_ARG;
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  long g = (npix + 255) / 256;
  if (g > 2048) g = 2048;
  const int es = unetk_esize(dtype);
  if (n_classes > 2 && loss_tiled_ok(logits, ldz, dlogits, lddz, n_classes, es, true)) {
    const size_t lds = 2 * (((size_t)kLossTile * n_classes * es + 15) / 16 * 16);
    UNETK_WITH_T(dtype, hipLaunchKernelGGL(seg_loss_bwd_tiled_kernel<T>, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream,
                         (const T*)logits, (const long long*)mask, (long)npix, n_classes, out4, grad_out,
                         (T*)dlogits));
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  UNETK_WITH_T(dtype, hipLaunchKernelGGL(seg_loss_bwd_kernel<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       (const T*)logits, ldz, (const long long*)mask, (long)npix, n_classes, out4, grad_out,
                       (T*)dlogits, lddz));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_dice_fwd(const float* input, const float* target, int n_groups, int64_t group_size, float eps, double* sums,
                   float* dice_out, float* coef, void* stream) {
  if (!input || !target || !sums || !dice_out || n_groups <= 0 || n_groups > 65535 || group_size <= 0)
    return UNETK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, (size_t)n_groups * 3 * sizeof(double), s) != hipSuccess) return UNETK_ERR_LAUNCH;
  long gx = (group_size + 255) / 256;
  const long cap = n_groups >= 2048 ? 1 : 2048 / n_groups;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(dice_sums_kernel, dim3((unsigned)gx, (unsigned)n_groups), dim3(256), 0, s, input, target,
                     (long)group_size, sums);
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(1), 0, s, (const double*)sums, n_groups, eps, dice_out, coef);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_dice_bwd(const float* target, const float* coef, const float* grad_out, float sign, int n_groups,
                   int64_t group_size, float* dinput, void* stream) {
  if (!target || !coef || !dinput || n_groups <= 0 || n_groups > 65535 || group_size <= 0) return UNETK_ERR_ARG;
  long gx = (group_size + 255) / 256;
  const long cap = n_groups >= 2048 ? 1 : 2048 / n_groups;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(dice_bwd_kernel, dim3((unsigned)gx, (unsigned)n_groups), dim3(256), 0, (hipStream_t)stream, target,
                     coef, grad_out, sign, (long)group_size, dinput);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int64_t unetk_eval_dice_workspace_bytes(int n, int n_classes) {
  if (n <= 0 || n_classes <= 0) return UNETK_ERR_SHAPE;
  return (int64_t)n * n_classes * 3 * sizeof(unsigned long long);
}

int unetk_eval_dice(const void* logits, int ld, const int64_t* mask, int n, int64_t hw, int n_classes, int first_class,
                    float eps, void* workspace, int64_t workspace_bytes, float* dice_out, int dtype, void* stream) {
  if (int e = eval_dice_check(logits, ld, mask, n, hw, n_classes, first_class, dice_out, dtype)) return e;
  if (!workspace) return UNETK_ERR_ARG;
  if (workspace_bytes < unetk_eval_dice_workspace_bytes(n, n_classes)) return UNETK_ERR_ARG;
  return eval_dice_launch(logits, ld, mask, n, hw, n_classes, first_class, eps, (unsigned long long*)workspace, dice_out,
                          dtype, (hipStream_t)stream);
}

int unetk_eval_dice_noalloc(const void* logits, int ld, const int64_t* mask, int n, int64_t hw, int n_classes,
                            int first_class, float eps, float* dice_out, int dtype, void* stream) {
  if (int e = eval_dice_check(logits, ld, mask, n, hw, n_classes, first_class, dice_out, dtype)) return e;
  if ((long)n * n_classes > UNETK_EVAL_NOALLOC_MAX_GROUPS) return UNETK_ERR_TOO_LARGE;
  int err = UNETK_OK;
  unsigned long long* counts = eval_scratch_for(logits, &err);
  if (!counts) return err;
  return eval_dice_launch(logits, ld, mask, n, hw, n_classes, first_class, eps, counts, dice_out, dtype,
                          (hipStream_t)stream);
}

}  // extern "C"
