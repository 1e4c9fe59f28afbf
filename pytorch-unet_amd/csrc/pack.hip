// Weight repacking: PyTorch-layout fp32 parameter -> the 1-KiB MFMA B-fragment stream conv_igemm.hip copies
// into LDS with global_load_lds.  Fragment (nt, iter, tap, ks, nb): lane l = (j = l&31, h = l>>5) holds E
// consecutive K-channels  kc*CH + ks*2E + h*E + e  of GEMM-N channel  nt*BN + NB*j + nb  (the NB channels of
// one lane are adjacent so the epilogue stores NB-wide vectors).
#include "common.h"
#include "conv_config.h"

namespace {

// fmt 0: the 32x32x16 fragment stream described above.  fmt 1 ("k32", bf16 conv3x3, conv_config.h): fragment
// ((nt*K/32 + kc)*9 + tap)*NB + cb; lane (c = l&15, g = l>>4) holds GEMM-K channels kc*32 + 8g + e of GEMM-N channel
// nt*16*NB + NB*c + cb (NB = 8 or 4 column blocks of 16).
struct PackGeo { int N, K, taps, KS, outer, NB, fmt; };

bool pack_geo(int kind, int c_out, int c_in, int dtype, PackGeo* g) {
  switch (kind) {
    case 0:
      if (conv3_use_k32(unetk_is16(dtype), c_out)) { *g = {c_out, c_in, 9, 2, 1, conv3_k32_nb16(c_out), 1}; return true; }
      { const Conv3Config c = conv3_config(c_out); *g = {c_out, c_in, 9, c.KS, 1, c.NB, 0}; return true; }
    case 1:
      if (conv3_use_k32(unetk_is16(dtype), c_in)) { *g = {c_in, c_out, 9, 2, 1, conv3_k32_nb16(c_in), 1}; return true; }
      { const Conv3Config c = conv3_config(c_in); *g = {c_in, c_out, 9, c.KS, 1, c.NB, 0}; return true; }
    case 2: *g = {4 * c_out, c_in, 1, 4, 1, pw_nb(4 * c_out), 0}; return true;
    case 3: *g = {c_in, c_out, 1, 4, 4, pw_nb(c_in), 0}; return true;
  }
  return false;
}

template <typename T>
__global__ void pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int kind, int c_out, int c_in, int N,
                            int K, int taps, int KS, int outer, int NB, int fmt, long npieces) {
  constexpr int E = ElemTraits<T>::E;
  const int CH = 2 * KS * E;
  const int nchunks = K / CH;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npieces; idx += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long f = idx >> 6;
    if (fmt == 1) {                       // k32 (bf16 conv3x3): kinds 0 and 1 only
      const int cb = (int)(f % NB); f /= NB;
      const int tap = (int)(f % 9); f /= 9;
      const int kc = (int)(f % (K / 32)); f /= (K / 32);
      const int nt = (int)f;
      const int c = lane & 15, g = lane >> 4;
      const int o = nt * 16 * NB + NB * c + cb;
      T out[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int i = kc * 32 + g * 8 + e;
        const long sidx = kind == 0 ? ((long)o * c_in + i) * 9 + tap : ((long)i * c_in + o) * 9 + (8 - tap);
        out[e] = from_f32<T>(src[sidx]);
      }
      *reinterpret_cast<uint4*>(dst + idx * E) = *reinterpret_cast<const uint4*>(out);
      continue;
    }
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
  long long piece_begin;     // first UNIT (64 groups; a group = all taps' pieces of one 16-byte K run) of this entry
  int kind, c_out, c_in, dtype;
  int N, K, taps, KS, outer, NB;
  int fmt;
  int pad_[1];
};
static_assert(sizeof(PlanItem) % 8 == 0, "plain layout");

// A wave packs one UNIT = 64 groups, a group = all taps of (one GEMM-N channel o, E consecutive GEMM-K channels).
// What a unit covers is chosen per kind so that BOTH sides move whole cache lines (one thread per group with
// fragment-order lanes read 64 different lines per load instruction: the launch was bound by the texture addresser
// at 0.35 ms for 124 MB):
//   kind 0 (conv3x3 fprop, src [o][i][9])   unit = fragment set (nt, kc, ks, nb).  The 32 channels o of the set each
//           own a contiguous run of 2E*9 floats: the wave copies the 32 runs with coalesced 16-byte loads into LDS
//           (row pitch odd: conflict-free) and the lanes pick their 9 x E values from there; stores are whole 1-KiB
//           fragments.
//   kind 1 (conv3x3 dgrad, src [i][o][9])   unit = 64 consecutive o x (kc, ks, h): adjacent lanes read adjacent 36-byte
//           runs; a store instruction writes NB runs of 64/NB adjacent 16-byte slots.
//   kinds 2, 3 (ConvTranspose2d, 9 % of the parameters)   fragment-order lanes, direct loads.
constexpr int kPackWaves = 4;

template <typename T>
__device__ __forceinline__ void store_piece_of(const PlanItem& it, int nt, int kc, int ks, int nb, int lane, int t,
                                               const float (&v)[ElemTraits<T>::E]) {
  constexpr int E = ElemTraits<T>::E;
  const int nchunks = it.K / (2 * it.KS * E);
  const int tap_out = it.outer > 1 ? t : 0, tap_in = it.outer > 1 ? 0 : t;
  // fragment index = ((((nt*outer + tap_out)*nchunks + kc)*taps + tap_in)*KS + ks)*NB + nb
  const long frag = (((((long)nt * it.outer + tap_out) * nchunks + kc) * it.taps + tap_in) * it.KS + ks) * it.NB + nb;
  T out[E];
#pragma unroll
  for (int e = 0; e < E; ++e) out[e] = from_f32<T>(v[e]);
  *reinterpret_cast<uint4*>(reinterpret_cast<T*>(it.dst) + (frag * 64 + lane) * E) = *reinterpret_cast<const uint4*>(out);
}

// kind 0, phase 1: the unit's 32 parameter runs -> LDS rows
template <typename T>
__device__ __forceinline__ void stage_kind0(const PlanItem& it, long unit, int lane, float* rows) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int RUN = 2 * E * 9, NQ = RUN / 4, RP = RUN + 1;
  if (it.fmt == 1) {
    // k32: unit = (nt, kc, cb): 16 channels o, each a contiguous run of 32 input channels x 9 taps = 288 floats
    constexpr int RUN1 = 288, NQ1 = RUN1 / 4, RP1 = RUN1 + 1;
    const int nch = it.K / 32;
    long f1 = unit;
    const int cb = (int)(f1 % it.NB); f1 /= it.NB;
    const int kc = (int)(f1 % nch); f1 /= nch;
    const int nt = (int)f1;
    for (int q = lane; q < 16 * NQ1; q += 64) {
      const int r = q / NQ1, c = q - r * NQ1;
      const int o = nt * 16 * it.NB + it.NB * r + cb;
      const float4 v = *reinterpret_cast<const float4*>(it.src + ((long)o * it.c_in + kc * 32) * 9 + 4 * c);
      float* d = rows + r * RP1 + 4 * c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    return;
  }
  const int nchunks = it.K / (2 * it.KS * E);
  long f = unit;
  const int nb = (int)(f % it.NB); f /= it.NB;
  const int ks = (int)(f % it.KS); f /= it.KS;
  const int kc = (int)(f % nchunks); f /= nchunks;
  const int nt = (int)f;
  const int i_base = kc * (2 * it.KS * E) + ks * 2 * E;
  for (int q = lane; q < 32 * NQ; q += 64) {
    const int r = q / NQ, c = q - r * NQ;
    const int o = nt * 32 * it.NB + it.NB * r + nb;
    const float4 v = *reinterpret_cast<const float4*>(it.src + ((long)o * it.c_in + i_base) * 9 + 4 * c);
    float* d = rows + r * RP + 4 * c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

template <typename T>
__device__ __forceinline__ void emit_unit(const PlanItem& it, long unit, int lane, const float* rows) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int RP = 2 * E * 9 + 1;
  const int CH = 2 * it.KS * E;
  const int nchunks = it.K / CH;
  const int ntap = it.taps * it.outer;
  if (it.fmt == 1) {
    const int nch = it.K / 32;
    if (it.kind == 0) {
      long f = unit;
      const int cb = (int)(f % it.NB); f /= it.NB;
      const int kc = (int)(f % nch); f /= nch;
      const int nt = (int)f;
      const int c = lane & 15, g = lane >> 4;
      const float* row = rows + c * 289 + g * 8 * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        T out[E];
#pragma unroll
        for (int e = 0; e < E; ++e) out[e] = from_f32<T>(row[e * 9 + t]);
        const long frag = (((long)nt * nch + kc) * 9 + t) * it.NB + cb;
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(it.dst) + (frag * 64 + lane) * E) =
            *reinterpret_cast<const uint4*>(out);
      }
    } else {
      // kind 1: unit = 64 consecutive GEMM-N channels o x (kc, k-group g); adjacent lanes read adjacent 36-byte runs
      long f = unit;
      const int g = (int)(f & 3); f >>= 2;
      const int kc = (int)(f % nch); f /= nch;
      const int o = (int)f * 64 + lane;
      const int bn = 16 * it.NB;
      const int nt = o / bn, oin = o - nt * bn;
      const int c = oin / it.NB, cb = oin - c * it.NB;
      const int i0 = kc * 32 + g * 8;
      float v[E][9];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float* p = it.src + ((long)(i0 + e) * it.c_in + o) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) v[e][t] = p[t];
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        T out[E];
#pragma unroll
        for (int e = 0; e < E; ++e) out[e] = from_f32<T>(v[e][8 - t]);      // dgrad: rotated kernel
        const long frag = (((long)nt * nch + kc) * 9 + t) * it.NB + cb;
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(it.dst) + (frag * 64 + g * 16 + c) * E) =
            *reinterpret_cast<const uint4*>(out);
      }
    }
    return;
  }
  if (it.kind == 1) {
    long f = unit;
    const int h = (int)(f & 1); f >>= 1;
    const int ks = (int)(f % it.KS); f /= it.KS;
    const int kc = (int)(f % nchunks); f /= nchunks;
    const int o = (int)f * 64 + lane;                      // 64 consecutive GEMM-N channels
    const int nt = o / (32 * it.NB), oin = o - nt * 32 * it.NB;
    const int j = oin / it.NB, nb = oin - j * it.NB;
    const int i0 = kc * CH + ks * 2 * E + h * E;
    float v[E][9];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float* p = it.src + ((long)(i0 + e) * it.c_in + o) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) v[e][t] = p[t];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float w[E];
#pragma unroll
      for (int e = 0; e < E; ++e) w[e] = v[e][8 - t];      // dgrad: rotated kernel
      store_piece_of<T>(it, nt, kc, ks, nb, h * 32 + j, t, w);
    }
    return;
  }
  long f = unit;
  const int nb = (int)(f % it.NB); f /= it.NB;
  const int ks = (int)(f % it.KS); f /= it.KS;
  const int kc = (int)(f % nchunks); f /= nchunks;
  const int nt = (int)f;
  const int j = lane & 31, h = lane >> 5;
  if (it.kind == 0) {
    const float* row = rows + j * RP + h * E * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float w[E];
#pragma unroll
      for (int e = 0; e < E; ++e) w[e] = row[e * 9 + t];
      store_piece_of<T>(it, nt, kc, ks, nb, lane, t, w);
    }
    return;
  }
  const int o = nt * 32 * it.NB + it.NB * j + nb;
  const int i0 = kc * CH + ks * 2 * E + h * E;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t >= ntap) continue;
    float w[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = i0 + e;
      if (it.kind == 2) {                                  // the tap is part of the GEMM-N index
        const int ab = o / it.c_out, co = o - ab * it.c_out;
        w[e] = it.src[((long)i * it.c_out + co) * 4 + ab];
      } else {
        w[e] = it.src[((long)o * it.c_out + i) * 4 + t];
      }
    }
    store_piece_of<T>(it, nt, kc, ks, nb, lane, t, w);
  }
}

__global__ __launch_bounds__(64 * kPackWaves) void pack_plan_kernel(const PlanItem* __restrict__ plan, int n_items,
                                                                   long total_units) {
  extern __shared__ float pack_lds[];                      // [kPackWaves][32 rows][145]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* rows = pack_lds + wave * (32 * 145);
  // the same number of trips for every wave of a block (the barriers below are unconditional)
  for (long ubase = (long)blockIdx.x * kPackWaves; ubase < total_units; ubase += (long)gridDim.x * kPackWaves) {
    const long unit = ubase + wave;
    const bool valid = unit < total_units;
    PlanItem it{};
    long local = 0;
    if (valid) {
      int lo = 0, hi = n_items - 1;        // last item whose first unit <= unit
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (plan[mid].piece_begin <= unit) lo = mid; else hi = mid - 1;
      }
      it = plan[lo];
      local = unit - it.piece_begin;
      if (it.kind == 0) {
        if (it.dtype == UNETK_BF16) stage_kind0<bf16>(it, local, lane, rows);
        else if (it.dtype == UNETK_F16) stage_kind0<f16>(it, local, lane, rows);
        else stage_kind0<float>(it, local, lane, rows);
      }
    }
    __syncthreads();
    if (valid) {
      if (it.dtype == UNETK_BF16) emit_unit<bf16>(it, local, lane, rows);
      else if (it.dtype == UNETK_F16) emit_unit<f16>(it, local, lane, rows);
      else emit_unit<float>(it, local, lane, rows);
    }
    __syncthreads();
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
    if (u.kind == 0 && (reinterpret_cast<uintptr_t>(u.src) & 15)) return UNETK_ERR_ALIGN;   // 16-byte loads of the runs
    PackGeo g;
    pack_geo(u.kind, u.c_out, u.c_in, u.dtype, &g);
    PlanItem p{};
    p.src = u.src; p.dst = u.dst; p.piece_begin = total;
    p.kind = u.kind; p.c_out = u.c_out; p.c_in = u.c_in; p.dtype = u.dtype;
    p.N = g.N; p.K = g.K; p.taps = g.taps; p.KS = g.KS; p.outer = g.outer; p.NB = g.NB; p.fmt = g.fmt;
    plan[i] = p;
    total += bytes / 16 / (g.taps * g.outer) / 64;   // units: one wave each
  }
  return total;
}

int unetk_pack_plan_run(const void* plan_dev, int n_items, int64_t total_units, void* stream) {
  if (!plan_dev || n_items <= 0 || total_units <= 0) return UNETK_ERR_ARG;
  constexpr size_t lds = (size_t)kPackWaves * 32 * 145 * sizeof(float);
  UNETK_ALLOW_LDS(pack_plan_kernel, lds);
  long g = (total_units + kPackWaves - 1) / kPackWaves;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_plan_kernel, dim3((unsigned)g), dim3(64 * kPackWaves), lds, (hipStream_t)stream,
                     static_cast<const PlanItem*>(plan_dev), n_items, (long)total_units);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_version(void) { return 1; }
const char* unetk_arch(void) { return "gfx950"; }

int64_t unetk_packed_weight_bytes(int kind, int c_out, int c_in, int dtype) {
  PackGeo g;
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  if (!pack_geo(kind, c_out, c_in, dtype, &g)) return UNETK_ERR_ARG;
  const int es = unetk_esize(dtype);
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
  pack_geo(kind, c_out, c_in, dtype, &g);
  const long npieces = bytes / 16;
  const int grid = (int)((npieces + 255) / 256 > 4096 ? 4096 : (npieces + 255) / 256);
  UNETK_WITH_T(dtype, hipLaunchKernelGGL(pack_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, kind, c_out,
                       c_in, g.N, g.K, g.taps, g.KS, g.outer, g.NB, g.fmt, npieces));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
