// Optimizer step of the reference's training loop (train.py:77-78, 114-116) over ONE flat fp32 buffer:
//   clip_grad_norm_(parameters, max_norm)  +  RMSprop(lr, alpha, eps, weight_decay, momentum, centered=False)
// Two HBM passes instead of torch's ~10 foreach passes: (1) sum of squares of the gradient (per-workgroup fp64
// partials, summed in a fixed order by the update kernel itself), (2) one read-modify-write of parameter, gradient
// (the clipped gradient is written back, as clip_grad_norm_ does in place), square_avg and momentum buffer.
#include "common.h"

namespace {

constexpr int kSumsqBlocks = 1024;

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
  __shared__ double red[256];
  double s = 0.0;
  const long n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = g4[i];
    s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += (double)g[i] * g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

struct RmspropArgs {
  float max_norm, lr, alpha, eps, weight_decay, momentum;
};

__global__ __launch_bounds__(256) void clip_rmsprop_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ sq, float* __restrict__ buf, long n,
                                                           const double* __restrict__ partial, int npartial,
                                                           double* __restrict__ norm_out, RmspropArgs a) {
  // every workgroup re-derives the clip coefficient from the partials (8 KiB from L2) in the same fixed order
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < npartial; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float total_norm = (float)sqrt(red[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = (double)total_norm;
  float coef = 1.f;
  if (a.max_norm > 0.f) coef = fminf(a.max_norm / (total_norm + 1e-6f), 1.f);   // clip_grad_norm_'s clamp

  const long n4 = n >> 2;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* g4 = reinterpret_cast<f32x4*>(g);
  f32x4* s4 = reinterpret_cast<f32x4*>(sq);
  f32x4* b4 = reinterpret_cast<f32x4*>(buf);
  auto update = [&](float& pv, float& gv, float& sv, float& bv) {
    gv *= coef;
    const float gw = fmaf(a.weight_decay, pv, gv);
    sv = fmaf(1.f - a.alpha, gw * gw, sv * a.alpha);
    const float avg = sqrtf(sv) + a.eps;
    if (a.momentum > 0.f) {
      bv = fmaf(a.momentum, bv, gw / avg);
      pv = fmaf(-a.lr, bv, pv);
    } else {
      pv = fmaf(-a.lr, gw / avg, pv);
    }
  };
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 pi = p4[i], gi = g4[i], si = s4[i], bi = a.momentum > 0.f ? b4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    float pv[4], gv[4], sv[4], bv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pv[k] = pi[k]; gv[k] = gi[k]; sv[k] = si[k]; bv[k] = bi[k];
      update(pv[k], gv[k], sv[k], bv[k]);
    }
    p4[i] = f32x4{pv[0], pv[1], pv[2], pv[3]};
    g4[i] = f32x4{gv[0], gv[1], gv[2], gv[3]};
    s4[i] = f32x4{sv[0], sv[1], sv[2], sv[3]};
    if (a.momentum > 0.f) b4[i] = f32x4{bv[0], bv[1], bv[2], bv[3]};
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      float bv = a.momentum > 0.f ? buf[i] : 0.f;
      update(p[i], g[i], sq[i], bv);
      if (a.momentum > 0.f) buf[i] = bv;
    }
}

}  // namespace

extern "C" {

int64_t unetk_clip_rmsprop_workspace_bytes(void) { return (int64_t)kSumsqBlocks * sizeof(double); }

int unetk_clip_rmsprop_step(float* param, float* grad, float* square_avg, float* momentum_buf, int64_t n,
                            void* workspace, int64_t workspace_bytes, double* total_norm_out, float max_norm, float lr,
                            float alpha, float eps, float weight_decay, float momentum, void* stream) {
  if (!param || !grad || !square_avg || n <= 0 || !workspace || (momentum > 0.f && !momentum_buf)) return UNETK_ERR_ARG;
  if (workspace_bytes < unetk_clip_rmsprop_workspace_bytes()) return UNETK_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(square_avg) |
       reinterpret_cast<uintptr_t>(momentum_buf) | reinterpret_cast<uintptr_t>(workspace)) & 15)
    return UNETK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  long nb = (n / 4 + 255) / 256;
  const int gs = (int)(nb > kSumsqBlocks ? kSumsqBlocks : (nb < 1 ? 1 : nb));
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(gs), dim3(256), 0, s, (const float*)grad, (long)n, (double*)workspace);
  UNETK_CHECK_LAUNCH();
  const int gu = (int)(nb > 2048 ? 2048 : (nb < 1 ? 1 : nb));
  const RmspropArgs a{max_norm, lr, alpha, eps, weight_decay, momentum};
  hipLaunchKernelGGL(clip_rmsprop_kernel, dim3(gu), dim3(256), 0, s, param, grad, square_avg, momentum_buf, (long)n,
                     (const double*)workspace, gs, total_norm_out, a);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
