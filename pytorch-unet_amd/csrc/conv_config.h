// Tile configuration of the conv3x3 implicit GEMM, shared by the kernel dispatch (conv_igemm.hip) and the weight
// packer (pack.hip): both must agree on how many 32-channel column blocks (NB) a workgroup owns and how many
// k-steps (KS) one LDS chunk holds, because the packed weight stream is laid out in exactly that order.
#pragma once
#include <stdlib.h>

struct Conv3Config { int NB, KS; };

// n_gemm = GEMM-N channel count (c_out for fprop, c_in for dgrad)
static inline Conv3Config conv3_config(int n_gemm) {
  static const int force_nb = getenv("UNETK_CONV_NB") ? atoi(getenv("UNETK_CONV_NB")) : 0;
  if (n_gemm % 128 == 0 && force_nb != 2) return {4, 1};   // 128-channel tiles: fewer LDS bytes per MFMA
  return {2, 2};
}

// ConvTranspose2d k2 s2 forward / dgrad (pointwise GEMMs): 128-channel n-tiles (NB = 4, the double-buffered kernel)
// whenever GEMM-N allows, else the 64-channel first-generation kernel.  n_gemm = 4*c_out (forward), c_in (dgrad).
static inline int pw_nb(int n_gemm) {
  static const int force_nb = getenv("UNETK_PW_NB") ? atoi(getenv("UNETK_PW_NB")) : 0;   // 2: A/B against the old kernel
  return (n_gemm % 128 == 0 && force_nb != 2) ? 4 : 2;
}
