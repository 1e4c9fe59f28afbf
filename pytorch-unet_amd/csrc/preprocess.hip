// GPU side of the reference's input formatting, BasicDataset.preprocess (utils/data_loading.py:65-91):
//
//   image:  pil_img.resize((newW, newH), BICUBIC) -> HWC uint8 -> CHW -> (/255.0 if any value > 1)     (:69,:83-91)
//   mask:   pil_img.resize((newW, newH), NEAREST) -> index of the pixel's value in mask_values (int64)  (:69,:72-80)
//
// The arithmetic of `resize` lives in Pillow (third-party; requirements.txt pins Pillow==9.3.0, this image has
// 12.2.0 -- libImaging/Resample.c and Geometry.c are unchanged between them for these paths).  Its published
// algorithm is restated here so that the result is BIT-EXACT:
//
//   BICUBIC  two passes (horizontal, then vertical over the rows the vertical pass needs), each a 1-D convolution
//            with per-output-pixel windows [xmin, xmin+n) and coefficients a(x) = Keys cubic (a = -0.5) evaluated at
//            (x + xmin - center + 0.5) / filterscale, normalised to sum 1 in double, then rounded to 22-bit fixed
//            point; a pass computes clip8((2^21 + sum u8 * k) >> 22) -- an 8-bit intermediate between the passes.
//   NEAREST  xin = COORD(xo) with xo accumulated by repeated addition of in/out (sequential double adds).
//
// Everything that is sequential or double-precision (the windows, the coefficients, the NEAREST position tables)
// is computed ON THE HOST into a "plan" (as Pillow computes it once per resize call) and uploaded by the caller;
// the kernels do the integer convolutions / gathers.  All of it is byte/integer work bound by HBM; an image is
// 7.4 MB in, 1.8 MB out -- the point is not speed of one image but that the decoded uint8 batch never needs a CPU
// resize pass to keep 8 GPUs fed.
#include <math.h>
#include <string.h>

#include "common.h"

namespace {

struct ResizePlan {               // host-built, POD; offsets in bytes from the start of the plan
  int magic, filter;              // filter: 0 NEAREST, 3 BICUBIC (Pillow's enum values)
  int in_w, in_h, out_w, out_h;
  int need_h, need_v;             // BICUBIC: which passes run (Resample.c ImagingResampleInner)
  int ksize_h, ksize_v;
  int ybox_first, ybox_rows;      // source rows the horizontal pass produces (those the vertical pass reads)
  int off_bounds_h, off_coef_h, off_bounds_v, off_coef_v;
  int total_bytes, pad_;
};
constexpr int kMagic = 0x52535a31;   // "RSZ1"
constexpr int kPrecisionBits = 32 - 8 - 2;

#pragma clang fp contract(off)
double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

int coeff_ksize(int in_size, int out_size) {
  double filterscale = (double)((float)in_size - 0.f) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  return (int)ceil(support) * 2 + 1;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box [0, in_size)
void precompute_coeffs(int in_size, int out_size, int ksize, int* bounds, int* kk) {
  const float in0 = 0.f, in1 = (float)in_size;
  double scale, filterscale;
  filterscale = scale = (double)(in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  double* k = new double[ksize];
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int x;
    for (x = 0; x < xmax; ++x) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (; x < ksize; ++x) k[x] = 0;
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
    for (x = 0; x < ksize; ++x) {
      const double p = k[x];
      kk[xx * ksize + x] = p < 0 ? (int)(-0.5 + p * (1 << kPrecisionBits)) : (int)(0.5 + p * (1 << kPrecisionBits));
    }
  }
  delete[] k;
}

// Geometry.c ImagingScaleAffine: positions by repeated addition (the rounding of the running sum is part of the result)
void nearest_table(int in_size, int out_size, int* tab) {
  const double a = (double)((float)in_size - 0.f) / out_size;
  double xo = 0.0 + a * 0.5;
  for (int x = 0; x < out_size; ++x) {
    const int xin = xo < 0.0 ? -1 : (int)xo;
    tab[x] = (xin >= 0 && xin < in_size) ? xin : -1;     // -1: the pixel keeps the fill value 0
    xo += a;
  }
}

__device__ __forceinline__ int clip8(int v) {
  v >>= kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One convolution pass over [n][rows][len][C] uint8.  AXIS 0: along x (dst x <- window of src x), AXIS 1: along y.
template <int AXIS>
__global__ __launch_bounds__(256) void resample_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk,
                                                            int ksize, int n, int src_rows, int src_w, int row0,
                                                            int dst_rows, int dst_w, int C, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int x = (int)(r % dst_w); r /= dst_w;
    const int y = (int)(r % dst_rows);
    const int img = (int)(r / dst_rows);
    const int o = AXIS == 0 ? x : y;
    const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int* k = kk + (long)o * ksize;
    int ss = 1 << (kPrecisionBits - 1);
    const uint8_t* base = src + (size_t)img * src_rows * src_w * C;
    if (AXIS == 0) {
      const uint8_t* p = base + ((size_t)(y + row0) * src_w + lo) * C + c;
      for (int i = 0; i < cnt; ++i) ss += (int)p[(size_t)i * C] * k[i];
    } else {
      const uint8_t* p = base + ((size_t)(lo + row0) * src_w + x) * C + c;
      for (int i = 0; i < cnt; ++i) ss += (int)p[(size_t)i * src_w * C] * k[i];
    }
    dst[idx] = (uint8_t)clip8(ss);
  }
}

// data_loading.py:88-89: `if (img > 1).any(): img = img / 255.0`, per image.  Pass 1 finds the flag, pass 2 converts:
// float32(double(v) / 255.0) -- the reference divides in float64 and the loader casts with .float() (:110).
__global__ __launch_bounds__(256) void any_gt1_kernel(const uint8_t* __restrict__ img, long per_image, int* flags) {
  const int n = blockIdx.y;
  const uint8_t* p = img + (size_t)n * per_image;
  int hit = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_image; i += (long)gridDim.x * 256) hit |= p[i] > 1;
  if (__any(hit) && (threadIdx.x & 63) == 0) atomicOr(flags + n, 1);
}

__global__ __launch_bounds__(256) void u8_to_float_kernel(const uint8_t* __restrict__ img, float* __restrict__ out,
                                                          long per_image, int n, const int* __restrict__ flags) {
  const long total = per_image * n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int f = flags[i / per_image];
    const double v = (double)img[i];
    out[i] = (float)(f ? v / 255.0 : v);
  }
}

// NEAREST gather + data_loading.py:73-78: mask[img == v] = i for i, v in enumerate(mask_values) (later entries win)
__global__ __launch_bounds__(256) void mask_index_kernel(const uint8_t* __restrict__ src, long long* __restrict__ out,
                                                         const int* __restrict__ xtab, const int* __restrict__ ytab,
                                                         const uint8_t* __restrict__ values, int n_values, int n,
                                                         int in_h, int in_w, int out_h, int out_w, int C) {
  const long total = (long)n * out_h * out_w;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx % out_w);
    long r = idx / out_w;
    const int y = (int)(r % out_h);
    const int img = (int)(r / out_h);
    const int xi = xtab[x], yi = ytab[y];
    uint8_t px[4] = {0, 0, 0, 0};
    if (xi >= 0 && yi >= 0) {
      const uint8_t* p = src + (((size_t)img * in_h + yi) * in_w + xi) * C;
      for (int c = 0; c < C; ++c) px[c] = p[c];
    }
    long long cls = 0;
    for (int i = 0; i < n_values; ++i) {
      bool eq = true;
      for (int c = 0; c < C; ++c) eq = eq && (values[i * C + c] == px[c]);
      if (eq) cls = i;
    }
    out[idx] = cls;
  }
}

int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

const ResizePlan* check_plan(const void* plan_host, int filter) {
  const ResizePlan* p = static_cast<const ResizePlan*>(plan_host);
  if (!p || p->magic != kMagic || p->filter != filter) return nullptr;
  return p;
}

}  // namespace

extern "C" {

int64_t unetk_resize_plan_build(int in_w, int in_h, int out_w, int out_h, int filter, void* plan_host,
                                int64_t plan_bytes) {
  if (in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0) return UNETK_ERR_SHAPE;
  if (filter != 0 && filter != 3) return UNETK_ERR_ARG;
  ResizePlan h{};
  h.magic = kMagic; h.filter = filter;
  h.in_w = in_w; h.in_h = in_h; h.out_w = out_w; h.out_h = out_h;
  int64_t off = sizeof(ResizePlan);
  if (filter == 0) {
    h.off_bounds_h = (int)off; off += (int64_t)out_w * 4;
    h.off_bounds_v = (int)off; off += (int64_t)out_h * 4;
  } else {
    h.need_h = out_w != in_w;
    h.need_v = out_h != in_h;
    h.ksize_h = coeff_ksize(in_w, out_w);
    h.ksize_v = coeff_ksize(in_h, out_h);
    h.off_bounds_h = (int)off; off += (int64_t)out_w * 8;
    h.off_coef_h = (int)off; off += (int64_t)out_w * h.ksize_h * 4;
    h.off_bounds_v = (int)off; off += (int64_t)out_h * 8;
    h.off_coef_v = (int)off; off += (int64_t)out_h * h.ksize_v * 4;
  }
  off = (off + 15) & ~(int64_t)15;
  if (off > 0x7fffffffLL) return UNETK_ERR_SHAPE;
  h.total_bytes = (int)off;
  if (!plan_host) return off;                       // size query
  if (plan_bytes < off) return UNETK_ERR_ARG;
  char* base = static_cast<char*>(plan_host);
  memset(base, 0, (size_t)off);
  if (filter == 0) {
    nearest_table(in_w, out_w, reinterpret_cast<int*>(base + h.off_bounds_h));
    nearest_table(in_h, out_h, reinterpret_cast<int*>(base + h.off_bounds_v));
  } else {
    int* bh = reinterpret_cast<int*>(base + h.off_bounds_h);
    int* bv = reinterpret_cast<int*>(base + h.off_bounds_v);
    precompute_coeffs(in_w, out_w, h.ksize_h, bh, reinterpret_cast<int*>(base + h.off_coef_h));
    precompute_coeffs(in_h, out_h, h.ksize_v, bv, reinterpret_cast<int*>(base + h.off_coef_v));
    h.ybox_first = bv[0];
    h.ybox_rows = bv[out_h * 2 - 2] + bv[out_h * 2 - 1] - h.ybox_first;
    if (!h.need_h || !h.need_v) { h.ybox_first = 0; h.ybox_rows = in_h; }
  }
  memcpy(base, &h, sizeof h);
  return off;
}

int64_t unetk_resize_workspace_bytes(const void* plan_host, int n, int channels) {
  const ResizePlan* p = check_plan(plan_host, 3);
  if (!p || n <= 0 || channels < 1 || channels > 4) return UNETK_ERR_ARG;
  // [temp of the horizontal pass][resized uint8][per-image flags]
  const int64_t tmp = (p->need_h && p->need_v) ? (int64_t)n * p->ybox_rows * p->out_w * channels : 0;
  const int64_t res = (int64_t)n * p->out_h * p->out_w * channels;
  return ((tmp + 15) & ~15LL) + ((res + 15) & ~15LL) + (((int64_t)n * 4 + 15) & ~15LL);
}

int unetk_preprocess_image(const uint8_t* src, int n, int channels, const void* plan_host, const void* plan_dev,
                           float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  const ResizePlan* p = check_plan(plan_host, 3);
  if (!p || !src || !plan_dev || !out || !workspace) return UNETK_ERR_ARG;
  const int64_t need = unetk_resize_workspace_bytes(plan_host, n, channels);
  if (need < 0) return (int)need;
  if (workspace_bytes < need) return UNETK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int C = channels;
  const char* pd = static_cast<const char*>(plan_dev);
  const int* bh = reinterpret_cast<const int*>(pd + p->off_bounds_h);
  const int* kh = reinterpret_cast<const int*>(pd + p->off_coef_h);
  const int* bv = reinterpret_cast<const int*>(pd + p->off_bounds_v);
  const int* kv = reinterpret_cast<const int*>(pd + p->off_coef_v);
  const int64_t tmp_bytes = (p->need_h && p->need_v) ? (((int64_t)n * p->ybox_rows * p->out_w * C + 15) & ~15LL) : 0;
  const int64_t res_bytes = ((int64_t)n * p->out_h * p->out_w * C + 15) & ~15LL;
  uint8_t* tmp = static_cast<uint8_t*>(workspace);
  uint8_t* res = tmp + tmp_bytes;
  int* flags = reinterpret_cast<int*>(res + res_bytes);
  const long per_image = (long)p->out_h * p->out_w * C;
  const uint8_t* resized = res;
  if (p->need_h && p->need_v) {
    const long t1 = (long)n * p->ybox_rows * p->out_w * C;
    hipLaunchKernelGGL(resample_pass_kernel<0>, dim3(grid_for(t1)), dim3(256), 0, s, src, tmp, bh, kh, p->ksize_h, n,
                       p->in_h, p->in_w, p->ybox_first, p->ybox_rows, p->out_w, C, t1);
    // the vertical windows are relative to the first row the horizontal pass produced
    hipLaunchKernelGGL(resample_pass_kernel<1>, dim3(grid_for(per_image * n)), dim3(256), 0, s, tmp, res, bv, kv,
                       p->ksize_v, n, p->ybox_rows, p->out_w, -p->ybox_first, p->out_h, p->out_w, C, per_image * n);
  } else if (p->need_h) {
    hipLaunchKernelGGL(resample_pass_kernel<0>, dim3(grid_for(per_image * n)), dim3(256), 0, s, src, res, bh, kh,
                       p->ksize_h, n, p->in_h, p->in_w, 0, p->out_h, p->out_w, C, per_image * n);
  } else if (p->need_v) {
    hipLaunchKernelGGL(resample_pass_kernel<1>, dim3(grid_for(per_image * n)), dim3(256), 0, s, src, res, bv, kv,
                       p->ksize_v, n, p->in_h, p->in_w, 0, p->out_h, p->out_w, C, per_image * n);
  } else {
    resized = src;                                  // Image.resize returns a copy when the size does not change
  }
  if (hipMemsetAsync(flags, 0, (size_t)n * 4, s) != hipSuccess) return UNETK_ERR_LAUNCH;
  int gx = grid_for(per_image);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(any_gt1_kernel, dim3(gx, n), dim3(256), 0, s, resized, per_image, flags);
  hipLaunchKernelGGL(u8_to_float_kernel, dim3(grid_for(per_image * n)), dim3(256), 0, s, resized, out, per_image, n,
                     flags);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_preprocess_mask(const uint8_t* src, int n, int channels, const void* plan_host, const void* plan_dev,
                          const uint8_t* mask_values, int n_values, int64_t* out, void* stream) {
  const ResizePlan* p = check_plan(plan_host, 0);
  if (!p || !src || !plan_dev || !out || n <= 0 || channels < 1 || channels > 4) return UNETK_ERR_ARG;
  if (n_values < 0 || (n_values > 0 && !mask_values)) return UNETK_ERR_ARG;
  const char* pd = static_cast<const char*>(plan_dev);
  const long total = (long)n * p->out_h * p->out_w;
  hipLaunchKernelGGL(mask_index_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<long long*>(out), reinterpret_cast<const int*>(pd + p->off_bounds_h),
                     reinterpret_cast<const int*>(pd + p->off_bounds_v), mask_values, n_values, n, p->in_h, p->in_w,
                     p->out_h, p->out_w, channels);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
