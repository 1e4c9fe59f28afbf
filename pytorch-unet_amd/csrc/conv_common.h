// Pieces shared by the MFMA convolution kernels (conv_igemm.hip, conv_first.hip).
#pragma once
#include "common.h"

namespace {

// One k-step: a 16-byte A piece and a 16-byte B piece per lane.
//   bf16: one v_mfma_f32_32x32x16_bf16 (lane half h holds k = 8h..8h+7)
//   fp32: four v_mfma_f32_32x32x2_f32 (lane half h holds k = 4h+e for e = 0..3) -- exact fp32, fmaf-chain numerics
template <typename T> struct Mma;
template <> struct Mma<bf16> {
  static __device__ __forceinline__ void run(f32x16& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(f32x16& c, const uint4& a, const uint4& b) {
    f32x4 av = __builtin_bit_cast(f32x4, a), bv = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], c, 0, 0, 0);
  }
};

// NB adjacent output channels of one lane -> one vector store.  Nontemporal: a conv output is streamed once to HBM
// and next read by another kernel; keeping it out of the way leaves L2 to the weight stream and the halo re-reads.
template <typename T, int NB> __device__ __forceinline__ void store_channels(T* p, const float (&v)[NB]) {
  if constexpr (sizeof(T) == 2) {
    if constexpr (NB == 2) {
      bf16x2 o = {(bf16)v[0], (bf16)v[1]};
      __builtin_nontemporal_store(__builtin_bit_cast(unsigned, o), reinterpret_cast<unsigned*>(p));
    } else {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      __builtin_nontemporal_store(__builtin_bit_cast(u32x2, o), reinterpret_cast<u32x2*>(p));
    }
  } else {
    if constexpr (NB == 2) {
      f32x2 o = {v[0], v[1]};
      __builtin_nontemporal_store(o, reinterpret_cast<f32x2*>(p));
    } else {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(p));
    }
  }
}

}  // namespace
