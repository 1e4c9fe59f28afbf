// Pieces shared by the MFMA convolution kernels (conv_igemm.hip, conv_first.hip).
#pragma once
#include "common.h"

// internal (never crosses the C ABI): the kernel serving this shape has no fused epilogue of the kind asked for;
// the caller runs the unfused sequence instead
#define UNETK_ERR_UNFUSED (-1000)

namespace {

struct ConvParams {
  const void* x;
  const void* wpk;
  void* y;
  const float* bias;
  const float* aff_scale;   // conv3x3 AFF instantiations: y = relu(scale[c] * conv + shift[c]) (eval-mode BatchNorm2d + ReLU)
  const float* aff_shift;
  void* pool;                // AFF only: MaxPool2d(2) of the stored activation [N][H/2][W/2][Cout] with pitch ldp, or NULL
  int ldp;
  double* stats;
  // BN-backward reduction of the layer that PRODUCED this convolution's output tensor, folded into a dgrad epilogue
  // (bn_z != NULL): with z that layer's pre-BatchNorm tensor [N][H][W][Cout] (pitch ld_bnz) and dx the value stored,
  //   stats[.][c][0] += sum dx * [scale*z + shift > 0],   stats[.][c][1] += sum dx * [..] * (z - mean) * invstd
  // -- the two sums unetk_bn_relu_bwd_reduce forms from a second pass over dx and z (elementwise.hip)
  const void* bn_z;
  int ld_bnz;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  int N, H, W;       // tile grid domain (CONV3: image; PWT/CTD: the low-res grid)
  int Cin, Cout;     // GEMM K channels per tap, GEMM N channels
  int ldx, ldy, xoff, yoff;
  int Hs, Ws;        // PWT: destination image; CTD: source (dy) image
  int padT, padL;
  int Co;            // PWT: channels of the up-sampled half (Cout == 4*Co)
  int tiles_x, tiles_y, n_tiles;
  int tpb;           // spatial tiles per workgroup
  int nblk;          // workgroups that have work (first-generation kernel: the grid is padded to a multiple of 8)
  unsigned long long* stamps;   // diagnostic builds only (UNETK_K32_STAMP): per (workgroup, wave) phase cycle sums
  int dbg;           // ablation flags (UNETK_CONV_DBG): 1 no DMA, 2 no MFMA loop, 4 no stores
};

// One k-step: a 16-byte A piece and a 16-byte B piece per lane.
//   bf16 / fp16: one v_mfma_f32_32x32x16_{bf16,f16} (lane half h holds k = 8h..8h+7)
//   fp32: four v_mfma_f32_32x32x2_f32 (lane half h holds k = 4h+e for e = 0..3) -- exact fp32, fmaf-chain numerics
template <typename T> struct Mma {     // the 16-bit element types
  static __device__ __forceinline__ void run(f32x16& c, const uint4& a, const uint4& b) {
    typedef typename H16<T>::x8 h8;
    c = H16<T>::mfma32(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c);
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
      typename H16<T>::x2 o = {(T)v[0], (T)v[1]};
      __builtin_nontemporal_store(__builtin_bit_cast(unsigned, o), reinterpret_cast<unsigned*>(p));
    } else {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      typename H16<T>::x4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
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

// conv3x3_k32.hip: the 16-bit conv3x3 kernel on v_mfma_f32_16x16x32_{bf16,f16} (K = 32 input channels of one tap per
// instruction); reads the "k32" packed weight format (conv_config.h).  Same parameter block as the kernels of
// conv_igemm.hip, which dispatches to it.  dtype: UNETK_BF16 or UNETK_F16.
int unetk_internal_conv3x3_k32(const void* params, int dtype, void* stream);
