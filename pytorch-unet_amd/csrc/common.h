// Shared device helpers for the gfx950 (MI355X / CDNA4) U-Net kernels.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/unetk.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

// UNETK_STAT_REPLICAS (unetk.h): replicas of the per-channel (sum, sumsq) accumulators; spreads the fp64
// atomics of thousands of workgroups over many cache lines (MI355X_MICROARCH "Global float atomics":
// every workgroup adding into one row is 14x slower).

template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16> { static constexpr int E = 8; };   // elements per 16-byte piece
template <> struct ElemTraits<f16> { static constexpr int E = 8; };
template <> struct ElemTraits<float> { static constexpr int E = 4; };

__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(float v) { return v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }

// The two 16-bit element types (UNETK_BF16, UNETK_F16) share every kernel: data movement, tiling and LDS layouts see
// 16-bit words; what differs is collected here -- vector types, the MFMA / dot-product instructions and how a packed
// 32-bit word unpacks to two floats.  Rounding f32 -> element is the cast (round to nearest even in both).
template <typename T> struct H16;
template <> struct H16<bf16> {
  typedef bf16x2 x2;
  typedef bf16x4 x4;
  typedef bf16x8 x8;
  static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float dot2(x2 a, x2 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
  static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
};
template <> struct H16<f16> {
  typedef f16x2 x2;
  typedef f16x4 x4;
  typedef f16x8 x8;
  static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float dot2(x2 a, x2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
  static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
  static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }
};
// two floats -> one packed word of two elements (first in the low half)
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b) {
  typename H16<T>::x2 v = {(T)a, (T)b};
  return __builtin_bit_cast(unsigned, v);
}

// Direct global -> LDS copy, 16 B per lane; LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const AS1 void*)gsrc, (AS3 void*)lds_wave_base, 16, 0, 0);
}

// The same LDS-DMA, hidden from the compiler.  hipcc treats the builtin as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of EVERY later ds_read it cannot prove disjoint -- which serialises a
// double-buffered pipeline (the DMA fills buffer B while the MFMAs read buffer A).  Through inline asm the
// instruction is invisible to that bookkeeping (cdna_hip_programming.md 5.7); completion is then OUR job: an
// explicit `s_waitcnt vmcnt(0)` + barrier before anybody reads the destination.  M0 (the LDS base of the DMA) is
// saved and restored inside the statement.  lds_addr must be wave-uniform.
__device__ __forceinline__ unsigned lds_address(const void* p) {
  return (unsigned)(size_t)(const AS3 char*)p;
}
__device__ __forceinline__ void glds16_hidden(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}

// 1 KiB of LDS-DMA through a buffer descriptor: lane -> 16 bytes at desc.base + soff + voff (ZEROS when voff is out
// of the descriptor's range: scripts/microbench/buf_lds_oob.hip), LDS destination = lds_addr + lane * 16
// (wave-uniform).  Hidden from the compiler like glds16_hidden; `s_nop 4`: the SGPR operands may be fresh from
// v_readfirstlane (cdna_hip_programming 5.7).
typedef __attribute__((ext_vector_type(4))) unsigned unetk_u32x4;
__device__ __forceinline__ void bufdma16(unsigned voff, unetk_u32x4 desc, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(lds_addr), "s"(desc), "s"(soff)
      : "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The shipped library takes no tuning input from the environment: every knob is a compile-time constant unless the
// library is built with -DUNETK_TUNING (A/B builds, scripts/build_variant.sh), where it is read once per process.
#include <stdlib.h>
#ifdef UNETK_TUNING
static inline int unetk_knob(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#define UNETK_KNOB(var, name, dflt) static const int var = unetk_knob(name, dflt)
#else
#define UNETK_KNOB(var, name, dflt) constexpr int var = (dflt)
#endif

static inline int host_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// dtype codes of the C ABI (unetk.h)
static inline bool unetk_dtype_ok(int dtype) { return dtype == UNETK_BF16 || dtype == UNETK_F32 || dtype == UNETK_F16; }
static inline bool unetk_is16(int dtype) { return dtype == UNETK_BF16 || dtype == UNETK_F16; }
static inline int unetk_esize(int dtype) { return unetk_is16(dtype) ? 2 : 4; }      // bytes per element
static inline int unetk_epp(int dtype) { return unetk_is16(dtype) ? 8 : 4; }        // elements per 16-byte piece
// run `...` with T = the element type of `dtype` (validated before)
#define UNETK_WITH_T(dtype, ...)                                          \
  do {                                                                    \
    if ((dtype) == UNETK_BF16) { using T = bf16; __VA_ARGS__; }           \
    else if ((dtype) == UNETK_F16) { using T = f16; __VA_ARGS__; }        \
    else { using T = float; __VA_ARGS__; }                                \
  } while (0)
// the same for the kernels that exist for the 16-bit types only
#define UNETK_WITH_H(dtype, ...)                                          \
  do {                                                                    \
    if ((dtype) == UNETK_BF16) { using H = bf16; __VA_ARGS__; }           \
    else { using H = f16; __VA_ARGS__; }                                  \
  } while (0)

// Kernels with more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, which is a PER-DEVICE
// property of the function: remember it per call site and per device (a process may drive several GPUs).
static inline bool unetk_allow_lds(unsigned long long* done_mask, const void* func, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
  const unsigned long long bit = 1ull << dev;
  if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return true;
  if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
  return true;
}
#define UNETK_ALLOW_LDS(func, bytes)                                                              \
  do {                                                                                            \
    static unsigned long long lds_ok_ = 0;                                                        \
    if (!unetk_allow_lds(&lds_ok_, reinterpret_cast<const void*>(func), (bytes))) return UNETK_ERR_LAUNCH; \
  } while (0)

#define UNETK_CHECK_LAUNCH()                         \
  do {                                               \
    hipError_t e_ = hipGetLastError();               \
    if (e_ != hipSuccess) return UNETK_ERR_LAUNCH;   \
  } while (0)
