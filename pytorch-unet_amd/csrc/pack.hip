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

// ---- all weights of a network in ONE launch: a plan (array of PlanItem, built on the host, kept on the device by
// the caller) lists every (parameter, packed buffer, geometry); the kernel finds its item by the running piece count
struct PlanItem {
  const float* src;
  void* dst;
  long long piece_begin;     // first work item (a group of all taps' pieces) of this entry in the concatenated index space
  int kind, c_out, c_in, dtype;
  int N, K, taps, KS, outer, NB;
  int pad_[2];
};
static_assert(sizeof(PlanItem) % 8 == 0, "plain layout");

// One thread packs ALL taps of one (output channel, E consecutive K-channels) group: its reads are the contiguous
// taps of each (o, i) pair of the PyTorch layout (a 288-byte run for conv3x3 fprop), its writes one 16-byte piece
// per tap, adjacent lanes writing adjacent pieces of the same fragment.  (One thread per piece re-read every
// 32-byte sector of the parameter eight times.)
template <typename T>
__device__ __forceinline__ void pack_group(const PlanItem& it, long gidx) {
  constexpr int E = ElemTraits<T>::E;
  const int CH = 2 * it.KS * E;
  const int nchunks = it.K / CH;
  const int ntap = it.taps * it.outer;             // 9 (conv3x3) or 4 (convT: as inner taps for fwd, outer for dgrad)
  const int lane = (int)(gidx & 63);
  long f = gidx >> 6;
  const int nb = (int)(f % it.NB); f /= it.NB;
  const int ks = (int)(f % it.KS); f /= it.KS;
  const int kc = (int)(f % nchunks); f /= nchunks;
  const int nt = (int)f;
  const int j = lane & 31, h = lane >> 5;
  const int o = nt * 32 * it.NB + it.NB * j + nb;
  const int i0 = kc * CH + ks * 2 * E + h * E;
  float v[E][9];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = i0 + e;
    long base;
    if (it.kind == 0) base = ((long)o * it.c_in + i) * 9;
    else if (it.kind == 1) base = ((long)i * it.c_in + o) * 9;
    else if (it.kind == 2) { const int ab = o / it.c_out, co = o - ab * it.c_out; base = ((long)i * it.c_out + co) * 4 + ab; }
    else base = ((long)o * it.c_out + i) * 4;
    if (it.kind == 2) {
      v[e][0] = it.src[base];                      // the tap is part of the GEMM-N index: one value per (o, i)
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (t < ntap) v[e][t] = it.src[base + t];
    }
  }
  // fragment index = ((((nt*outer + tap_out)*nchunks + kc)*taps + tap_in)*KS + ks)*NB + nb
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (t >= ntap) continue;
    const int tap_out = it.outer > 1 ? t : 0, tap_in = it.outer > 1 ? 0 : t;
    const int tsrc = it.kind == 1 ? 8 - t : t;     // dgrad: rotated kernel
    const long frag = (((((long)nt * it.outer + tap_out) * nchunks + kc) * it.taps + tap_in) * it.KS + ks) * it.NB + nb;
    T out[E];
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = from_f32<T>(it.kind == 2 ? v[e][0] : v[e][tsrc]);
    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(it.dst) + (frag * 64 + lane) * E) = *reinterpret_cast<const uint4*>(out);
  }
}

__global__ __launch_bounds__(256) void pack_plan_kernel(const PlanItem* __restrict__ plan, int n_items, long total) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    int lo = 0, hi = n_items - 1;          // last item whose piece_begin <= idx
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (plan[mid].piece_begin <= idx) lo = mid; else hi = mid - 1;
    }
    const PlanItem it = plan[lo];
    if (it.dtype == UNETK_BF16) pack_group<bf16>(it, idx - it.piece_begin);
    else pack_group<float>(it, idx - it.piece_begin);
  }
}

}  // namespace

extern "C" {

int64_t unetk_pack_plan_bytes(int n_items) { return n_items > 0 ? (int64_t)n_items * sizeof(PlanItem) : UNETK_ERR_ARG; }

int64_t unetk_pack_plan_build(const unetk_pack_item* items, int n_items, void* plan_host, int64_t plan_bytes) {
  if (!items || !plan_host || n_items <= 0 || plan_bytes < unetk_pack_plan_bytes(n_items)) return UNETK_ERR_ARG;
  PlanItem* plan = static_cast<PlanItem*>(plan_host);
  long long total = 0;
  for (int i = 0; i < n_items; ++i) {
    const unetk_pack_item& u = items[i];
    const int64_t bytes = unetk_packed_weight_bytes(u.kind, u.c_out, u.c_in, u.dtype);
    if (bytes < 0) return bytes;
    if (!u.src || !u.dst) return UNETK_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(u.dst) & 15) return UNETK_ERR_ALIGN;
    PackGeo g;
    pack_geo(u.kind, u.c_out, u.c_in, &g);
    PlanItem p{};
    p.src = u.src; p.dst = u.dst; p.piece_begin = total;
    p.kind = u.kind; p.c_out = u.c_out; p.c_in = u.c_in; p.dtype = u.dtype;
    p.N = g.N; p.K = g.K; p.taps = g.taps; p.KS = g.KS; p.outer = g.outer; p.NB = g.NB;
    plan[i] = p;
    total += bytes / 16 / (g.taps * g.outer);      // one work item per (all taps of a 16-byte K group)
  }
  return total;
}

int unetk_pack_plan_run(const void* plan_dev, int n_items, int64_t total_pieces, void* stream) {
  if (!plan_dev || n_items <= 0 || total_pieces <= 0) return UNETK_ERR_ARG;
  long g = (total_pieces + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_plan_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const PlanItem*>(plan_dev), n_items, (long)total_pieces);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

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
