// OutConv: Conv2d(Cin -> n_classes, kernel 1) with bias (unet_parts.py:74).  n_classes is 1, 2 or 21 in the
// reference's configurations, so this is an HBM-bound GEMV-like pass over the 64-channel feature map, not an
// MFMA problem: it reads Cin*es bytes per pixel and writes n_classes*es.
//
// Lanes are laid out [pixel][16-byte piece] so that a wave reads 1 KiB contiguous per load; the pieces of one
// pixel are summed with lane shuffles.  Classes are processed CT at a time (grid.y).
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ void load_piece(const T* p, float* v) {
  const uint4 raw = *reinterpret_cast<const uint4*>(p);
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

template <typename T> __device__ __forceinline__ void store_piece(T* p, const float* v) {
  if constexpr (sizeof(T) == 2) {
    typename H16<T>::x8 b;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (T)v[i];
    __builtin_nontemporal_store(__builtin_bit_cast(f32x4, b), reinterpret_cast<f32x4*>(p));
  } else {
    f32x4 f = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(f, reinterpret_cast<f32x4*>(p));
  }
}

constexpr int CT = 4;   // classes per pass

// G = lanes per pixel (power of two <= 64, <= pieces per pixel)
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                          const float* __restrict__ bias, T* __restrict__ out, int ldo,
                                                          long npix, int Cin, int ncls, int G) {
  constexpr int E = ElemTraits<T>::E;
  extern __shared__ float wl[];   // [CT][Cin]
  const int cbase = blockIdx.y * CT;
  for (int i = threadIdx.x; i < CT * Cin; i += 256) {
    const int k = i / Cin;
    wl[i] = (cbase + k < ncls) ? w[(size_t)(cbase + k) * Cin + (i - k * Cin)] : 0.f;
  }
  __syncthreads();
  const int ppp = Cin / E;
  const int sub = threadIdx.x % G, pl = threadIdx.x / G, lanes = 256 / G;
  if (ppp == G) {
    // one piece per lane (Cin <= 64 pieces, the U-Net's 64-channel head): weights in registers, four pixels'
    // loads in flight per trip
    float wr[CT][E];
#pragma unroll
    for (int k = 0; k < CT; ++k)
#pragma unroll
      for (int i = 0; i < E; ++i) wr[k][i] = wl[k * Cin + sub * E + i];
    const long step = (long)gridDim.x * lanes;
    for (long pix0 = (long)blockIdx.x * lanes + pl; pix0 < npix; pix0 += 4 * step) {
      float v[4][E], acc[4][CT];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long pix = pix0 + u * step;
        load_piece<T>(x + (pix < npix ? pix : pix0) * ldx + sub * E, v[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < CT; ++k) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < E; ++i) a = fmaf(v[u][i], wr[k][i], a);
          acc[u][k] = a;
        }
      for (int o = 1; o < G; o <<= 1)       // ascending: the order bn_relu_apply_head's DPP sum reproduces bit for bit
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < CT; ++k) acc[u][k] += __shfl_xor(acc[u][k], o, 64);
      if (sub == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long pix = pix0 + u * step;
          if (pix < npix)
#pragma unroll
            for (int k = 0; k < CT; ++k)
              if (cbase + k < ncls) out[pix * ldo + cbase + k] = from_f32<T>(acc[u][k] + bias[cbase + k]);
        }
      }
    }
    return;
  }
  for (long pix0 = (long)blockIdx.x * lanes; pix0 < npix; pix0 += (long)gridDim.x * lanes) {
    const long pix = pix0 + pl;
    float acc[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) acc[k] = 0.f;
    if (pix < npix) {
      for (int p = sub; p < ppp; p += G) {
        float v[E];
        load_piece<T>(x + pix * ldx + p * E, v);
#pragma unroll
        for (int k = 0; k < CT; ++k)
#pragma unroll
          for (int i = 0; i < E; ++i) acc[k] = fmaf(v[i], wl[k * Cin + p * E + i], acc[k]);
      }
    }
    for (int o = 1; o < G; o <<= 1)       // ascending: the order bn_relu_apply_head's DPP sum reproduces bit for bit
#pragma unroll
      for (int k = 0; k < CT; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
    if (sub == 0 && pix < npix) {
#pragma unroll
      for (int k = 0; k < CT; ++k)
        if (cbase + k < ncls) out[pix * ldo + cbase + k] = from_f32<T>(acc[k] + bias[cbase + k]);
    }
  }
}

// dx[pix][c] = sum_k dl[pix][k] * w[k][c]
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_dgrad_kernel(const T* __restrict__ dl, int lddl,
                                                            const float* __restrict__ w, T* __restrict__ dx, int lddx,
                                                            long npix, int Cin, int ncls) {
  constexpr int E = ElemTraits<T>::E;
  extern __shared__ float wl[];   // [ncls][Cin]
  for (int i = threadIdx.x; i < ncls * Cin; i += 256) wl[i] = w[i];
  __syncthreads();
  const int ppp = Cin / E;
  const long total = npix * ppp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx / ppp;
    const int c0 = (int)(idx - pix * ppp) * E;
    float o[E];
#pragma unroll
    for (int i = 0; i < E; ++i) o[i] = 0.f;
    for (int k = 0; k < ncls; ++k) {
      const float g = to_f32(dl[pix * lddl + k]);
#pragma unroll
      for (int i = 0; i < E; ++i) o[i] = fmaf(g, wl[k * Cin + c0 + i], o[i]);
    }
    store_piece<T>(dx + pix * lddx + c0, o);
  }
}

// per-block partials: slab[block][k][c] = sum over the block's pixels dl[pix][k]*x[pix][c]; column Cin = sum dl (bias)
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dl,
                                                            int lddl, float* __restrict__ slab, long npix, int Cin,
                                                            int ncls) {
  constexpr int E = ElemTraits<T>::E;
  extern __shared__ float red[];   // [CT][Cin + 1]
  const int cbase = blockIdx.y * CT;
  const int ppp = Cin / E;                  // host guarantees ppp divides 256
  const int piece = threadIdx.x % ppp, pl = threadIdx.x / ppp, lanes = 256 / ppp;
  for (int i = threadIdx.x; i < CT * (Cin + 1); i += 256) red[i] = 0.f;
  __syncthreads();
  float acc[CT][E], accb[CT];
#pragma unroll
  for (int k = 0; k < CT; ++k) {
    accb[k] = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) acc[k][i] = 0.f;
  }
  const long step = (long)gridDim.x * lanes;
  long pix = (long)blockIdx.x * lanes + pl;
  for (; pix + step < npix; pix += 2 * step) {      // two pixels' loads in flight
    float v0[E], v1[E], g0[CT], g1[CT];
    load_piece<T>(x + pix * ldx + piece * E, v0);
    load_piece<T>(x + (pix + step) * ldx + piece * E, v1);
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      g0[k] = (cbase + k < ncls) ? to_f32(dl[pix * lddl + cbase + k]) : 0.f;
      g1[k] = (cbase + k < ncls) ? to_f32(dl[(pix + step) * lddl + cbase + k]) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      accb[k] += g0[k] + g1[k];
#pragma unroll
      for (int i = 0; i < E; ++i) acc[k][i] = fmaf(g1[k], v1[i], fmaf(g0[k], v0[i], acc[k][i]));
    }
  }
  if (pix < npix) {
    float v[E];
    load_piece<T>(x + pix * ldx + piece * E, v);
#pragma unroll
    for (int k = 0; k < CT; ++k) {
      const float g = (cbase + k < ncls) ? to_f32(dl[pix * lddl + cbase + k]) : 0.f;
      accb[k] += g;
#pragma unroll
      for (int i = 0; i < E; ++i) acc[k][i] = fmaf(g, v[i], acc[k][i]);
    }
  }
#pragma unroll
  for (int k = 0; k < CT; ++k) {
#pragma unroll
    for (int i = 0; i < E; ++i) atomicAdd(&red[k * (Cin + 1) + piece * E + i], acc[k][i]);
    if (piece == 0) atomicAdd(&red[k * (Cin + 1) + Cin], accb[k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CT * (Cin + 1); i += 256) {
    const int k = i / (Cin + 1);
    if (cbase + k < ncls) slab[((size_t)blockIdx.x * ncls + cbase + k) * (Cin + 1) + (i - k * (Cin + 1))] = red[i];
  }
}

__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const float* __restrict__ slab, int nblocks, float* dw,
                                                                   float* db, int Cin, int ncls) {
  // one workgroup per output element; fixed summation order (thread-strided partials, then a tree)
  __shared__ float part[256];
  const int idx = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += slab[(size_t)b * ncls * (Cin + 1) + idx];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int k = idx / (Cin + 1), c = idx - k * (Cin + 1);
    if (c < Cin) dw[k * Cin + c] = part[0];
    else if (db) db[k] = part[0];
  }
}

// ---- many classes (n_classes > CT, e.g. the 21-class configuration), Cin == 64: the per-class-group kernels above
// pass over the activation ceil(n_classes/CT) times and store each logit with a scalar store from one lane in 16
// (2.4 ms forward, 1.6 ms wgrad for [16,512,512] pixels x 21 classes in fp32, 12x their HBM time).  These variants make
// ONE pass: a block stages 64 pixels x 64 channels in LDS channel-major ([c][65]: a lane per pixel reads a column
// without bank conflicts), lane (pixel, wave = class group) keeps ceil(n_classes/4) accumulators, the weights come as
// broadcast 16-byte LDS reads, and the 64 x n_classes logits leave through LDS as one contiguous 16-byte-vector copy.
constexpr int kHeadPix = 64;       // pixels per trip
constexpr int kHeadCin = 64;
constexpr int kHeadMaxGroup = 8;   // classes per lane: n_classes <= 32

template <typename T>
__device__ __forceinline__ void head_stage_x(const T* __restrict__ x, int ldx, long p0, long npix, float* xs) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int PPP = kHeadCin / E;                       // 16-byte pieces per pixel
  for (int i = threadIdx.x; i < kHeadPix * PPP; i += 256) {
    const int px = i / PPP, piece = i - px * PPP;
    float v[E];
    if (p0 + px < npix) load_piece<T>(x + (p0 + px) * ldx + piece * E, v);
    else {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) xs[(piece * E + e) * (kHeadPix + 1) + px] = v[e];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv1x1_fwd_tiled_kernel(const T* __restrict__ x, int ldx,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ bias, T* __restrict__ out,
                                                                long npix, int ncls) {
  extern __shared__ __attribute__((aligned(16))) float head_lds[];
  float* xs = head_lds;                                   // [64 c][65]
  float* ws = head_lds + ((kHeadCin * (kHeadPix + 1) + 3) / 4) * 4;   // [ncls][64], 16-byte aligned
  T* ot = reinterpret_cast<T*>(ws + ncls * kHeadCin);     // [64 px][ncls]
  for (int i = threadIdx.x; i < ncls * kHeadCin; i += 256) ws[i] = w[i];
  const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long ntrips = (npix + kHeadPix - 1) / kHeadPix;
  for (long trip = blockIdx.x; trip < ntrips; trip += gridDim.x) {
    const long p0 = trip * kHeadPix;
    __syncthreads();                                      // previous trip's output copy and x reads are done
    head_stage_x<T>(x, ldx, p0, npix, xs);
    __syncthreads();
    float acc[kHeadMaxGroup];
#pragma unroll
    for (int q = 0; q < kHeadMaxGroup; ++q) acc[q] = 0.f;
    for (int c = 0; c < kHeadCin; c += 4) {
      const float x0 = xs[(c + 0) * (kHeadPix + 1) + px], x1 = xs[(c + 1) * (kHeadPix + 1) + px];
      const float x2 = xs[(c + 2) * (kHeadPix + 1) + px], x3 = xs[(c + 3) * (kHeadPix + 1) + px];
#pragma unroll
      for (int q = 0; q < kHeadMaxGroup; ++q) {
        const int k = g + 4 * q;                          // wave-uniform
        if (k < ncls) {
          const float4 wv = *reinterpret_cast<const float4*>(ws + k * kHeadCin + c);
          acc[q] = fmaf(x3, wv.w, fmaf(x2, wv.z, fmaf(x1, wv.y, fmaf(x0, wv.x, acc[q]))));
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kHeadMaxGroup; ++q) {
      const int k = g + 4 * q;
      if (k < ncls) ot[px * ncls + k] = from_f32<T>(acc[q] + bias[k]);
    }
    __syncthreads();
    const int nelem = (int)(min((long)kHeadPix, npix - p0) * ncls);
    constexpr int PER = 16 / (int)sizeof(T);
    T* dst = out + p0 * ncls;
    if (nelem == kHeadPix * ncls) {
      for (int i = threadIdx.x; i < nelem / PER; i += 256)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(ot)[i];
    } else {
      for (int i = threadIdx.x; i < nelem; i += 256) dst[i] = ot[i];
    }
  }
}

// slab[block][k][c] (c == Cin: bias column) like conv1x1_wgrad_kernel, one pass over x and dl
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_wgrad_tiled_kernel(const T* __restrict__ x, int ldx,
                                                                  const T* __restrict__ dl, float* __restrict__ slab,
                                                                  long npix, int ncls) {
  extern __shared__ __attribute__((aligned(16))) float head_lds[];
  float* xs = head_lds;                                   // [64 c][65]
  float* ds = head_lds + ((kHeadCin * (kHeadPix + 1) + 3) / 4) * 4;   // [ncls][64 px] as float
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  float acc[kHeadMaxGroup], accb[kHeadMaxGroup];
#pragma unroll
  for (int q = 0; q < kHeadMaxGroup; ++q) acc[q] = accb[q] = 0.f;
  const long ntrips = (npix + kHeadPix - 1) / kHeadPix;
  for (long trip = blockIdx.x; trip < ntrips; trip += gridDim.x) {
    const long p0 = trip * kHeadPix;
    __syncthreads();
    head_stage_x<T>(x, ldx, p0, npix, xs);
    const int nelem = (int)(min((long)kHeadPix, npix - p0) * ncls);
    // class-major in LDS ([k][64 px]): a lane then takes four pixels of one class with one broadcast 16-byte read
    for (int i = threadIdx.x; i < kHeadPix * ncls; i += 256) {
      const int p = i / ncls, k = i - p * ncls;
      ds[k * kHeadPix + p] = i < nelem ? to_f32(dl[p0 * ncls + i]) : 0.f;
    }
    __syncthreads();
    const float* xrow = xs + c * (kHeadPix + 1);
    for (int p = 0; p < kHeadPix; p += 4) {
      const float x0 = xrow[p], x1 = xrow[p + 1], x2 = xrow[p + 2], x3 = xrow[p + 3];
#pragma unroll
      for (int q = 0; q < kHeadMaxGroup; ++q) {
        const int k = g + 4 * q;
        if (k < ncls) {
          const float4 d = *reinterpret_cast<const float4*>(ds + k * kHeadPix + p);   // broadcast
          acc[q] = fmaf(d.w, x3, fmaf(d.z, x2, fmaf(d.y, x1, fmaf(d.x, x0, acc[q]))));
          accb[q] += (d.x + d.y) + (d.z + d.w);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kHeadMaxGroup; ++q) {
    const int k = g + 4 * q;
    if (k < ncls) {
      float* row = slab + ((size_t)blockIdx.x * ncls + k) * (kHeadCin + 1);
      row[c] = acc[q];
      if (c == 0) row[kHeadCin] = accb[q];
    }
  }
}

size_t head_tiled_lds(int ncls, int es_out) {
  const size_t xs = ((kHeadCin * (kHeadPix + 1) + 3) / 4) * 4 * sizeof(float);
  return xs + (size_t)ncls * kHeadCin * sizeof(float) + (((size_t)kHeadPix * ncls * es_out + 15) / 16) * 16;
}

constexpr int kWgradBlocks = 2048;

int check_head(const void* x, int ldx, int cin, int ncls, int dtype) {
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  const int es = unetk_esize(dtype), e = 16 / es;
  if (!x) return UNETK_ERR_ARG;
  if (cin <= 0 || ncls <= 0 || cin % e || ldx < cin || (long)ncls * cin > 12288) return UNETK_ERR_SHAPE;
  const int ppp = cin / e;
  if (ppp > 256 || (256 % ppp)) return UNETK_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (ldx * es) % 16) return UNETK_ERR_ALIGN;
  return UNETK_OK;
}

}  // namespace

extern "C" {

int unetk_conv1x1_fwd(const void* x, int ldx, const float* w, const float* bias, void* out, int ldo, int64_t npix,
                      int c_in, int n_classes, int dtype, void* stream) {
  if (int e = check_head(x, ldx, c_in, n_classes, dtype)) return e;
  if (!w || !bias || !out || npix <= 0 || ldo < n_classes) return UNETK_ERR_ARG;
  const int e_ = unetk_epp(dtype);
  if (n_classes > CT && n_classes <= 4 * kHeadMaxGroup && c_in == kHeadCin && ldo == n_classes &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const size_t lds = head_tiled_lds(n_classes, unetk_esize(dtype));
    long gx = (npix + kHeadPix - 1) / kHeadPix;
    if (gx > 1024) gx = 1024;
    UNETK_WITH_T(dtype, hipLaunchKernelGGL(conv1x1_fwd_tiled_kernel<T>, dim3((unsigned)gx), dim3(256), lds, (hipStream_t)stream,
                         (const T*)x, ldx, w, bias, (T*)out, (long)npix, n_classes));
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  const int ppp = c_in / e_;
  int G = 1;
  while (G * 2 <= ppp && G < 64) G *= 2;
  const int lanes = 256 / G;
  long gx = (npix + lanes - 1) / lanes;
  if (gx > 2048) gx = 2048;
  dim3 grid((unsigned)gx, (unsigned)((n_classes + CT - 1) / CT));
  const size_t lds = (size_t)CT * c_in * sizeof(float);
  UNETK_WITH_T(dtype, hipLaunchKernelGGL(conv1x1_fwd_kernel<T>, grid, dim3(256), lds, (hipStream_t)stream, (const T*)x, ldx, w, bias,
                       (T*)out, ldo, (long)npix, c_in, n_classes, G));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_conv1x1_dgrad(const void* dl, int lddl, const float* w, void* dx, int lddx, int64_t npix, int c_in,
                        int n_classes, int dtype, void* stream) {
  if (int e = check_head(dx, lddx, c_in, n_classes, dtype)) return e;
  if (!w || !dl || npix <= 0 || lddl < n_classes) return UNETK_ERR_ARG;
  const int e_ = unetk_epp(dtype);
  long g = (npix * (c_in / e_) + 255) / 256;
  if (g > 2048) g = 2048;
  const size_t lds = (size_t)n_classes * c_in * sizeof(float);
  UNETK_WITH_T(dtype, hipLaunchKernelGGL(conv1x1_dgrad_kernel<T>, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream,
                       (const T*)dl, lddl, w, (T*)dx, lddx, (long)npix, c_in, n_classes));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int64_t unetk_conv1x1_wgrad_workspace_bytes(int c_in, int n_classes) {
  if (c_in <= 0 || n_classes <= 0) return UNETK_ERR_SHAPE;
  return (int64_t)kWgradBlocks * n_classes * (c_in + 1) * sizeof(float);
}

int unetk_conv1x1_wgrad(const void* x, int ldx, const void* dl, int lddl, float* dw, float* db, void* workspace,
                        int64_t workspace_bytes, int64_t npix, int c_in, int n_classes, int dtype, void* stream) {
  if (int e = check_head(x, ldx, c_in, n_classes, dtype)) return e;
  if (!dl || !dw || !workspace || npix <= 0 || lddl < n_classes) return UNETK_ERR_ARG;
  if (workspace_bytes < unetk_conv1x1_wgrad_workspace_bytes(c_in, n_classes)) return UNETK_ERR_ARG;
  const int e_ = unetk_epp(dtype);
  if (n_classes > CT && n_classes <= 4 * kHeadMaxGroup && c_in == kHeadCin && lddl == n_classes) {
    long gt = (npix + kHeadPix - 1) / kHeadPix;
    if (gt > 1024) gt = 1024;
    const size_t lds = head_tiled_lds(n_classes, 4) - (size_t)n_classes * kHeadCin * sizeof(float);
    UNETK_WITH_T(dtype, hipLaunchKernelGGL(conv1x1_wgrad_tiled_kernel<T>, dim3((unsigned)gt), dim3(256), lds, (hipStream_t)stream,
                         (const T*)x, ldx, (const T*)dl, (float*)workspace, (long)npix, n_classes));
    UNETK_CHECK_LAUNCH();
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(n_classes * (c_in + 1)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, (int)gt, dw, db, c_in, n_classes);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  const int lanes = 256 / (c_in / e_);
  long gx = (npix + lanes - 1) / lanes;
  if (gx > kWgradBlocks) gx = kWgradBlocks;
  dim3 grid((unsigned)gx, (unsigned)((n_classes + CT - 1) / CT));
  const size_t lds = (size_t)CT * (c_in + 1) * sizeof(float);
  UNETK_WITH_T(dtype, hipLaunchKernelGGL(conv1x1_wgrad_kernel<T>, grid, dim3(256), lds, (hipStream_t)stream, (const T*)x, ldx,
                       (const T*)dl, lddl, (float*)workspace, (long)npix, c_in, n_classes));
  UNETK_CHECK_LAUNCH();
  const int total = n_classes * (c_in + 1);
  hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, (int)gx, dw, db, c_in, n_classes);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
