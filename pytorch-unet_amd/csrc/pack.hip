// Weight repacking: PyTorch-layout fp32 parameter -> the 1-KiB MFMA B-fragment stream conv_igemm.hip copies
// into LDS with global_load_lds.  Fragment (nt, iter, tap, ks, nb): lane l = (j = l&31, h = l>>5) holds E
// consecutive K-channels  kc*CH + ks*2E + h*E + e  of GEMM-N channel  nt*BN + NB*j + nb  (the NB channels of
// one lane are adjacent so the epilogue stores NB-wide vectors).
#include "common.h"
#include "conv_config.h"

namespace {

struct PackGeo { int N, K, taps, KS, outer, NB; };

bool pack_geo(int kind, int c_out, int c_in, PackGeo* g) {
  switch (kind) {
    case 0: { const Conv3Config c = conv3_config(c_out); *g = {c_out, c_in, 9, c.KS, 1, c.NB}; return true; }
    case 1: { const Conv3Config c = conv3_config(c_in); *g = {c_in, c_out, 9, c.KS, 1, c.NB}; return true; }
    case 2: *g = {4 * c_out, c_in, 1, 4, 1, pw_nb(4 * c_out)}; return true;
    case 3: *g = {c_in, c_out, 1, 4, 4, pw_nb(c_in)}; return true;
  }
  return false;
}

template <typename T>
__global__ void pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int kind, int c_out, int c_in, int N,
                            int K, int taps, int KS, int outer, int NB, long npieces) {
  constexpr int E = ElemTraits<T>::E;
  const int CH = 2 * KS * E;
  const int nchunks = K / CH;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npieces; idx += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long f = idx >> 6;
    const int nb = (int)(f % NB); f /= NB;
    const int ks = (int)(f % KS); f /= KS;
    const int tap_in = (int)(f % taps); f /= taps;
    const int kc = (int)(f % nchunks); f /= nchunks;
    const int tap_out = (int)(f % outer); f /= outer;
    const int nt = (int)f;
    const int tap = outer > 1 ? tap_out : tap_in;
    const int j = lane & 31, h = lane >> 5;
    const int o = nt * 32 * NB + NB * j + nb;
    T out[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = kc * CH + ks * 2 * E + h * E + e;
      long s;
      if (kind == 0) s = ((long)o * c_in + i) * 9 + tap;
      else if (kind == 1) s = ((long)i * c_in + o) * 9 + (8 - tap);
      else if (kind == 2) { const int ab = o / c_out, co = o - ab * c_out; s = ((long)i * c_out + co) * 4 + ab; }
      else s = ((long)o * c_out + i) * 4 + tap;
      out[e] = from_f32<T>(src[s]);
    }
    *reinterpret_cast<uint4*>(dst + idx * E) = *reinterpret_cast<const uint4*>(out);
  }
}

}  // namespace

extern "C" {

int unetk_version(void) { return 1; }
const char* unetk_arch(void) { return "gfx950"; }

int64_t unetk_packed_weight_bytes(int kind, int c_out, int c_in, int dtype) {
  PackGeo g;
  if (!pack_geo(kind, c_out, c_in, &g)) return UNETK_ERR_ARG;
  if (dtype != UNETK_BF16 && dtype != UNETK_F32) return UNETK_ERR_DTYPE;
  const int es = dtype == UNETK_BF16 ? 2 : 4;
  const int CH = 2 * g.KS * (16 / es);
  if (c_out <= 0 || c_in <= 0 || g.N % 64 || g.K % CH) return UNETK_ERR_SHAPE;
  return (int64_t)g.N * g.K * g.taps * g.outer * es;
}

int unetk_pack_weight(int kind, const float* src, void* dst, int c_out, int c_in, int dtype, void* stream) {
  const int64_t bytes = unetk_packed_weight_bytes(kind, c_out, c_in, dtype);
  if (bytes < 0) return (int)bytes;
  if (!src || !dst) return UNETK_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(dst) & 15) return UNETK_ERR_ALIGN;
  PackGeo g;
  pack_geo(kind, c_out, c_in, &g);
  const long npieces = bytes / 16;
  const int grid = (int)((npieces + 255) / 256 > 4096 ? 4096 : (npieces + 255) / 256);
  if (dtype == UNETK_BF16)
    hipLaunchKernelGGL(pack_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, kind, c_out,
                       c_in, g.N, g.K, g.taps, g.KS, g.outer, g.NB, npieces);
  else
    hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, kind,
                       c_out, c_in, g.N, g.K, g.taps, g.KS, g.outer, g.NB, npieces);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
