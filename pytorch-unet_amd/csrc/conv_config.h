// Tile configuration of the conv3x3 implicit GEMM, shared by the kernel dispatch (conv_igemm.hip) and the weight
// packer (pack.hip): both must agree on how many 32-channel column blocks (NB) a workgroup owns and how many
// k-steps (KS) one LDS chunk holds, because the packed weight stream is laid out in exactly that order.
//
// Knobs: UNETK_KNOB (common.h) -- compile-time constants in the shipped library.
#pragma once
#include "common.h"

struct Conv3Config { int NB, KS; };

// n_gemm = GEMM-N channel count (c_out for fprop, c_in for dgrad)
static inline Conv3Config conv3_config(int n_gemm) {
  UNETK_KNOB(force_nb, "UNETK_CONV_NB", 0);
  if (n_gemm % 128 == 0 && force_nb != 2) return {4, 1};   // 128-channel tiles: fewer LDS bytes per MFMA
  return {2, 2};
}

// bf16 conv3x3 (fprop and dgrad) runs on v_mfma_f32_16x16x32_bf16 with its own packed weight format ("k32"):
//   fragment ((nt * K/32 + kc) * 9 + tap) * NB16 + cb, 1 KiB: lane (c = l & 15, g = l >> 4) holds the 8 GEMM-K channels
//   kc*32 + 8g .. 8g+7 of GEMM-N channel nt*16*NB16 + NB16*c + cb  (the NB16 channels of one lane are adjacent: the
//   epilogue stores 16-byte / 8-byte vectors).  NB16 = 8 (128-channel n-tiles) when GEMM-N allows, else 4.
// The packer and the dispatch both ask conv3_use_k32(), so the format always matches the kernel.
static inline bool conv3_use_k32(int dtype_is_bf16, int n_gemm) {
  // Every bf16 layer (mode 2).  Round 2 ran only the 64-channel n-tiles on this shape (mode 1): hipcc's allocation of
  // the 128-channel loop spilled; with one fragment set re-read after last use and buffer-descriptor DMA it fits 256
  // registers and beats the 32x32x16 kernel (conv3x3_db_kernel, still the fp32 path) by 9-13 % on every layer --
  // same LDS bytes and matrix-pipe cycles, but the chip holds a higher clock on 16x16x32.  Mode 0: never.
  UNETK_KNOB(mode, "UNETK_CONV_K32", 2);
  if (!dtype_is_bf16 || mode == 0) return false;
  return mode == 2 || n_gemm % 128 != 0;
}
static inline int conv3_k32_nb16(int n_gemm) {
  UNETK_KNOB(force_nb, "UNETK_CONV_NB", 0);
  return (n_gemm % 128 == 0 && force_nb != 2) ? 8 : 4;
}
// the 64-channel n-tile kernel with the epilogue under the next tile (conv3x3_k32n64_kernel); 1 = A/B: the plain one
static inline bool conv3_k32_nopipe() {
  UNETK_KNOB(nopipe, "UNETK_K32_NOPIPE", 0);
  return nopipe != 0;
}

// ConvTranspose2d k2 s2 forward / dgrad (pointwise GEMMs): 128-channel n-tiles (NB = 4, the double-buffered kernel)
// whenever GEMM-N allows, else the 64-channel first-generation kernel.  n_gemm = 4*c_out (forward), c_in (dgrad).
static inline int pw_nb(int n_gemm) {
  UNETK_KNOB(force_nb, "UNETK_PW_NB", 0);   // 2: A/B against the old kernel
  return (n_gemm % 128 == 0 && force_nb != 2) ? 4 : 2;
}
