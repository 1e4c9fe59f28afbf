// The network's first convolution: Conv2d(n_channels -> 64, 3x3, pad 1, no bias) with n_channels <= 4
// (unet_model.py:13 -> unet_parts.py:16; images have 3 channels, train.py:196).
//
// K = 9*Cin = 27 is MFMA-hostile as an implicit GEMM over channel chunks, and the layer is HBM-bound anyway
// (it writes 64 channels per pixel and reads 3), so it gets its own kernel: the fp32 image tile (with halo) is
// converted once into LDS as [pixel][4 channels], the GEMM-K index is k = tap*4 + ci (36, zero padded), and each
// lane assembles its A fragment from two 8-byte (bf16) or one 16-byte (fp32) LDS reads.  Epilogue (rounding,
// fused BatchNorm statistics, NB-wide channel stores) is the same as conv_igemm.hip.
//
// The matching weight gradient uses the transposed problem: rows = k (tap, ci), cols = co, GEMM-K = pixels.
#include <type_traits>

#include "conv_common.h"

namespace {

struct FirstParams {
  const float* x;     // fp32 NHWC image, pitch ldx
  const float* w;     // parameter [Cout][Cin][3][3]
  void* y;
  double* stats;
  const float* aff_scale;   // != NULL: y = relu(scale[c] * conv + shift[c]) (eval-mode BatchNorm2d + ReLU), no statistics
  const float* aff_shift;
  int N, H, W, Cin, Cout, ldx, ldy;
  int tiles_x, tiles_y, n_tiles;
  void* act;                // MODE 2: relu(scale[c] * y + shift[c]) of the ROUNDED y (pitch lda)
  int lda;
};

// MODE 0: y (+ statistics), or the eval-mode fused form (aff_scale != NULL).
// MODE 1: statistics only -- nothing is stored.  MODE 2: y and the training-mode BatchNorm + ReLU of it, the arithmetic
// of bn_relu_apply on the stored (rounded) y.  Training runs 1, finalize, 2: the layer is so cheap (27 MACs per output,
// 12 bytes read per pixel) that computing it twice costs less than reading its 64-channel output back for the apply pass.
template <typename T, int MODE = 0>
__global__ __launch_bounds__(256, 4) void conv_first_fprop_kernel(FirstParams P) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int NB = 2, BN = 64;
  constexpr int STEPS = (36 + 2 * E - 1) / (2 * E);     // bf16: 3 (k < 48), fp32: 5 (k < 40)
  constexpr int PXB = 4 * sizeof(T);                     // LDS bytes per pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_x = smem;                                    // [340][4] T
  char* lds_w = smem + 340 * 16;                         // [STEPS][NB][64 lanes][16 B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  // a workgroup keeps ONE channel tile and walks spatial tiles grp, grp + ngroups, ...: the weights are turned into
  // B fragments once per workgroup, not once per 256 pixels (38 400 workgroups of one tile each spent as long in
  // this prologue's scattered loads as in the tile itself: 0.53 ms for a layer whose HBM floor is 0.23 ms)
  const int nt = blockIdx.x % P.n_tiles;
  const int grp = blockIdx.x / P.n_tiles, ngroups = gridDim.x / P.n_tiles;
  const int spatial = P.N * P.tiles_y * P.tiles_x;

  // ---- weights -> B fragments (lane (j,h), step s, element e: k = s*2E + h*E + e, channel nt*64 + NB*j + nb)
  for (int i = tid; i < STEPS * NB * 64 * E; i += 256) {
    const int e = i % E;
    int r = i / E;
    const int l = r % 64; r /= 64;
    const int nb = r % NB;
    const int s = r / NB;
    const int k = s * 2 * E + (l >> 5) * E + e;
    const int tap = k >> 2, ci = k & 3;
    const int co = nt * BN + NB * (l & 31) + nb;
    float v = 0.f;
    if (tap < 9 && ci < P.Cin) v = P.w[((size_t)co * P.Cin + ci) * 9 + tap];
    reinterpret_cast<T*>(lds_w)[i] = from_f32<T>(v);
  }

  const int ch0 = nt * BN + NB * j;
  const bool aff = MODE == 0 && P.aff_scale != nullptr;
  float asc[NB], ash[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    asc[nb] = P.aff_scale != nullptr ? P.aff_scale[ch0 + nb] : 1.f;
    ash[nb] = P.aff_scale != nullptr ? P.aff_shift[ch0 + nb] : 0.f;
  }
  // BN statistics across this workgroup's tiles: fp32 for bf16 tensors, fp64 for fp32 tensors (conv_igemm.hip)
  using StatAcc = typename std::conditional<sizeof(T) == 4, double, float>::type;
  StatAcc tsum[NB], tsq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) tsum[nb] = tsq[nb] = 0;

  // the halo pixels of tile t+1 are fetched into registers while tile t is multiplied and stored (the fetch is a
  // dependent chain of scalar loads from HBM: unhidden, its latency was most of a tile's time)
  float pre[2][4];
  auto fetch = [&](int tile) {
    int bid = tile;
    const int tx = bid % P.tiles_x; bid /= P.tiles_x;
    const int ty = bid % P.tiles_y;
    const int img = bid / P.tiles_y;
    const float* ximg = P.x + (size_t)img * P.H * P.W * P.ldx;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int p = tid + 256 * k;
      const int hr = p / 34, hc = p - hr * 34;
      const int yy = ty * 8 + hr - 1, xx = tx * 32 + hc - 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) pre[k][c] = 0.f;
      if (p < 340 && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) {
        const float* src = ximg + (size_t)(yy * P.W + xx) * P.ldx;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < P.Cin) pre[k][c] = src[c];
      }
    }
  };
  if (grp < spatial) fetch(grp);

  for (int tile = grp; tile < spatial; tile += ngroups) {
    int bid = tile;
    const int tx = bid % P.tiles_x; bid /= P.tiles_x;
    const int ty = bid % P.tiles_y;
    const int img = bid / P.tiles_y;
    const int y0 = ty * 8, x0 = tx * 32;

    __syncthreads();                                      // every wave is done with the previous tile's pixels
    // ---- image tile + halo -> LDS, converted to T
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int p = tid + 256 * k;
      if (p < 340) {
        T* dst = reinterpret_cast<T*>(lds_x + p * PXB);
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[c] = from_f32<T>(pre[k][c]);
      }
    }
    __syncthreads();
    if (tile + ngroups < spatial) fetch(tile + ngroups);

    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      uint4 a[2], b[NB];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int row = 2 * wid + mb;
        if constexpr (sizeof(T) == 2) {
          const int tapA = 4 * s + 2 * h, tapB = tapA + 1;
          uint2 lo = {0u, 0u}, hi = {0u, 0u};
          if (tapA < 9) lo = *reinterpret_cast<const uint2*>(lds_x + ((row + tapA / 3) * 34 + j + tapA % 3) * PXB);
          if (tapB < 9) hi = *reinterpret_cast<const uint2*>(lds_x + ((row + tapB / 3) * 34 + j + tapB % 3) * PXB);
          a[mb] = uint4{lo.x, lo.y, hi.x, hi.y};
        } else {
          const int tap = 2 * s + h;
          uint4 v = {0u, 0u, 0u, 0u};
          if (tap < 9) v = *reinterpret_cast<const uint4*>(lds_x + ((row + tap / 3) * 34 + j + tap % 3) * PXB);
          a[mb] = v;
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const uint4*>(lds_w + ((s * NB + nb) * 64 + lane) * 16);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) Mma<T>::run(acc[mb][nb], a[mb], b[nb]);
    }

    // ---- epilogue (same contract as conv_igemm.hip)
    float ssum[NB], ssq[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) ssum[nb] = ssq[nb] = 0.f;
    // 64-bit address arithmetic once per row (wave-uniform); an element adds a 32-bit lane offset and a compile-time
    // multiple of the pixel pitch (as in conv3x3_db_kernel)
    T* yimg = reinterpret_cast<T*>(P.y) + (size_t)img * P.H * P.W * P.ldy;
    T* aimg = reinterpret_cast<T*>(P.act) + (size_t)img * P.H * P.W * P.lda;
    const int lane_off = 4 * h * P.ldy + ch0;
    const int lane_off_a = 4 * h * P.lda + ch0;
    const int xl = x0 + 4 * h;
    bool done = false;
    if constexpr (sizeof(T) == 2) {
      // interior tiles, training path: as in conv3x3_db_kernel -- round once into packed pixel pairs, statistics by
      // v_dot2c_f32_bf16 / v_dot2_f32_f16, the two-channel store words gathered with v_perm, no per-element predicates
      if (!aff && (y0 + 8 <= P.H) && (x0 + 32 <= P.W)) {
        typedef typename H16<T>::x2 h2;
        const h2 one2 = {(T)1.0f, (T)1.0f};
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const int yy = y0 + 2 * wid + mb;
          T* rowp = yimg + (size_t)(yy * P.W + x0) * P.ldy + lane_off;
          T* rowa = aimg + (size_t)(yy * P.W + x0) * P.lda + lane_off_a;
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const int ri = (i & 3) + 8 * (i >> 2);
            unsigned pk[NB], ak[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              const h2 y2 = {(T)acc[mb][nb][i], (T)acc[mb][nb][i + 1]};
              pk[nb] = __builtin_bit_cast(unsigned, y2);
              if constexpr (MODE != 2) {
                ssum[nb] = H16<T>::dot2(y2, one2, ssum[nb]);
                ssq[nb] = H16<T>::dot2(y2, y2, ssq[nb]);
              } else {
                const h2 a2 = {(T)fmaxf(fmaf((float)y2[0], asc[nb], ash[nb]), 0.f),
                               (T)fmaxf(fmaf((float)y2[1], asc[nb], ash[nb]), 0.f)};
                ak[nb] = __builtin_bit_cast(unsigned, a2);
              }
            }
            if constexpr (MODE != 1) {
              __builtin_nontemporal_store(__builtin_amdgcn_perm(pk[1], pk[0], 0x05040100u),
                                          reinterpret_cast<unsigned*>(rowp + ri * P.ldy));
              __builtin_nontemporal_store(__builtin_amdgcn_perm(pk[1], pk[0], 0x07060302u),
                                          reinterpret_cast<unsigned*>(rowp + (ri + 1) * P.ldy));
            }
            if constexpr (MODE == 2) {
              __builtin_nontemporal_store(__builtin_amdgcn_perm(ak[1], ak[0], 0x05040100u),
                                          reinterpret_cast<unsigned*>(rowa + ri * P.lda));
              __builtin_nontemporal_store(__builtin_amdgcn_perm(ak[1], ak[0], 0x07060302u),
                                          reinterpret_cast<unsigned*>(rowa + (ri + 1) * P.lda));
            }
          }
        }
        done = true;
      }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      if (done) break;
      const int yy = y0 + 2 * wid + mb;
      T* rowp = yimg + (size_t)(yy * P.W + x0) * P.ldy;
      T* rowa = aimg + (size_t)(yy * P.W + x0) * P.lda;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ri = (i & 3) + 8 * (i >> 2);
        if (yy < P.H && xl + ri < P.W) {
          float v[NB], a[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const float f = to_f32(from_f32<T>(acc[mb][nb][i]));
            v[nb] = aff ? fmaxf(fmaf(acc[mb][nb][i], asc[nb], ash[nb]), 0.f) : f;
            a[nb] = fmaxf(fmaf(f, asc[nb], ash[nb]), 0.f);
            ssum[nb] += f;
            ssq[nb] += f * f;
          }
          if constexpr (MODE != 1) store_channels<T, NB>(rowp + ri * P.ldy + lane_off, v);
          if constexpr (MODE == 2) store_channels<T, NB>(rowa + ri * P.lda + lane_off_a, a);
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { tsum[nb] += (StatAcc)ssum[nb]; tsq[nb] += (StatAcc)ssq[nb]; }
  }

  if (MODE != 2 && P.stats != nullptr && !aff) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      tsum[nb] += __shfl_xor(tsum[nb], 32, 64);
      tsq[nb] += __shfl_xor(tsq[nb], 32, 64);
    }
    __syncthreads();
    StatAcc* red = reinterpret_cast<StatAcc*>(smem);      // [4 waves][BN][2]
    if (h == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        red[(wid * BN + NB * j + nb) * 2 + 0] = tsum[nb];
        red[(wid * BN + NB * j + nb) * 2 + 1] = tsq[nb];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const StatAcc v = red[tid] + red[2 * BN + tid] + red[4 * BN + tid] + red[6 * BN + tid];
      double* dst = P.stats + ((size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * P.Cout + nt * BN) * 2 + tid;
      unsafeAtomicAdd(dst, (double)v);
    }
  }
}

// ------------------------------------------------------------------------------------------ wgrad, 16-bit elements
__device__ uint4 g_zero_page_first[8];
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct FirstWgradParams {
  const float* x;
  const void* dy;
  float* slab;        // [nsplit][col_tiles][64 rows (k)][64 cols]
  int N, H, W, Cin, Cout, ldx, lddy;
  int tiles_x, tiles_y, ntiles, nsplit, col_tiles;
  // BNF (unetk_conv3x3_first_wgrad_bn): `dy` is the gradient w.r.t. the layer's ACTIVATION (pitch lddy) and the gradient
  // w.r.t. the convolution's output is formed on the fly from it and the layer's pre-BatchNorm tensor z (pitch ldz),
  // exactly as unetk_bn_relu_bwd_apply would have stored it
  const void* z;
  int ldz;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_coef;     // coef: [3][Cout]
};

// rows i = k = tap*4 + ci (36 used of 64), cols = 64 output channels, GEMM-K = pixels.
// Each wave takes 2 of the 8 tile rows and keeps a full 64x64 accumulator; waves are summed through LDS at the end.
// (H: bf16 or f16; the name is kept from the bf16-only generation, profiles/ refers to it)
// BNF: the first layer's input needs no gradient, so the BatchNorm+ReLU backward's apply pass would write the conv-output
// gradient (one full-resolution tensor) only for this kernel to read it back once: here a thread loads its 16-byte
// pieces of the activation gradient and of z instead, applies  dy = c0 * dz - c1 - c2 * (z - mean)  (the expression of
// bn_relu_bwd_apply_kernel, rounded to H the same way) and writes the piece into the LDS tile the LDS-DMA would have
// filled -- a thread keeps ONE channel piece, its 48 coefficients live in registers.
template <typename H, bool BNF = false>
__global__ __launch_bounds__(256, 2) void conv_first_wgrad_bf16_kernel(FirstWgradParams P) {
  typedef typename H16<H>::x8 h8;
  typedef typename H16<H>::x4 h4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_dy = smem;                 // [256 px][64 co] bf16, 128-B rows, 16-B pieces XOR-swizzled
  char* lds_x = smem + 256 * 128;      // [340][4] bf16

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.x % P.col_tiles, split = blockIdx.x / P.col_tiles;
  const int t_begin = (int)((long)P.ntiles * split / P.nsplit);
  const int t_end = (int)((long)P.ntiles * (split + 1) / P.nsplit);

  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int r = lane & 31, h = lane >> 5;

  // B (dy) transposed-read addresses: [col block jb][ks][rd]
  int addrB[2][2][2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        const int p = ks * 16 + 8 * (g >> 1) + 4 * rd + q;
        const int c = jb * 4 + 2 * (g & 1) + (pp >> 1);
        addrB[jb][ks][rd] = p * 128 + ((c ^ (((p >> 1) & 1) << 2)) << 4) + (pp & 1) * 8;
      }
  // A (image) gather: row i = ib*32 + r  ->  tap = i>>2, ci = i&3 ; element jj is pixel 8h + jj of the k-step
  int addrA[2];
  bool rowok[2];
#pragma unroll
  for (int ib = 0; ib < 2; ++ib) {
    const int i = ib * 32 + r, tap = i >> 2, ci = i & 3;
    rowok[ib] = tap < 9;
    const int tp = tap < 9 ? tap : 0;
    addrA[ib] = ((tp / 3) * 34 + (tp % 3) + 8 * h) * 8 + ci * 2;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ib][jb][i] = 0.f;

  const char* zero = reinterpret_cast<const char*>(g_zero_page_first);
  const int lp = lane >> 3, ls = lane & 7;
  float bsc[8], bsh[8], bmu[8], bk0[8], bk1[8], bk2[8];
  if constexpr (BNF) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = ct * 64 + ls * 8 + i;
      bsc[i] = P.bn_scale[c]; bsh[i] = P.bn_shift[c]; bmu[i] = P.bn_mean[c];
      bk0[i] = P.bn_coef[c]; bk1[i] = P.bn_coef[P.Cout + c]; bk2[i] = P.bn_coef[2 * P.Cout + c];
    }
  }

  for (int tile = t_begin; tile < t_end; ++tile) {
    int tt = tile;
    const int tx = tt % P.tiles_x; tt /= P.tiles_x;
    const int ty = tt % P.tiles_y;
    const int img = tt / P.tiles_y;
    const int y0 = ty * 8, x0 = tx * 32;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const H* dimg = reinterpret_cast<const H*>(P.dy) + (size_t)img * P.H * P.W * P.lddy + ct * 64;
    if constexpr (BNF) {
      // dy tile from the activation gradient and z: logical piece ls of pixel p goes to slot ls ^ swizzle(p)
      const H* zimg = reinterpret_cast<const H*>(P.z) + (size_t)img * P.H * P.W * P.ldz + ct * 64;
      typedef typename H16<H>::x8 h8;
      h8 g8[8], z8[8];
      bool in[8];
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int p = (f * 4 + wid) * 8 + lp;
        const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
        in[f] = yy < P.H && xx < P.W;
        const size_t pix = in[f] ? (size_t)(yy * P.W + xx) : 0;
        g8[f] = *reinterpret_cast<const h8*>(dimg + pix * P.lddy + ls * 8);
        z8[f] = *reinterpret_cast<const h8*>(zimg + pix * P.ldz + ls * 8);
      }
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int p = (f * 4 + wid) * 8 + lp;
        h8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = (float)z8[f][i], g = (float)g8[f][i];
          const float dz = fmaf(v, bsc[i], bsh[i]) > 0.f ? g : 0.f;
          const float r = bk0[i] * dz - bk1[i] - bk2[i] * (v - bmu[i]);
          o[i] = in[f] ? (H)r : (H)0.f;
        }
        *reinterpret_cast<h8*>(lds_dy + p * 128 + ((ls ^ (((p >> 1) & 1) << 2)) << 4)) = o;
      }
    } else {
      // dy tile by LDS-DMA
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int fi = f * 4 + wid;
        const int p = fi * 8 + lp;
        const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
        const int c = ls ^ (((p >> 1) & 1) << 2);
        const char* src = zero;
        if (yy < P.H && xx < P.W) src = reinterpret_cast<const char*>(dimg + (size_t)(yy * P.W + xx) * P.lddy + c * 8);
        glds16(src, lds_dy + fi * 1024);
      }
    }
    // image halo tile, fp32 -> bf16
    const float* ximg = P.x + (size_t)img * P.H * P.W * P.ldx;
    for (int p = tid; p < 340; p += 256) {
      const int hr = p / 34, hc = p - hr * 34;
      const int yy = y0 + hr - 1, xx = x0 + hc - 1;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) {
        const float* src = ximg + (size_t)(yy * P.W + xx) * P.ldx;
        for (int c = 0; c < P.Cin; ++c) v[c] = src[c];
      }
      h4 o = {(H)v[0], (H)v[1], (H)v[2], (H)v[3]};
      *reinterpret_cast<h4*>(lds_x + p * 8) = o;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wid + rr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        h8 bfr[2], afr[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          const char* base = lds_dy + row * 4096;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(base + addrB[jb][ks][0]));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(base + addrB[jb][ks][1]));
          bfr[jb] = __builtin_bit_cast(h8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          const char* base = lds_x + (row * 34 + ks * 16) * 8 + addrA[ib];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const H v = *reinterpret_cast<const H*>(base + jj * 8);
            afr[ib][jj] = rowok[ib] ? v : (H)0.f;
          }
        }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
          for (int jb = 0; jb < 2; ++jb)
            acc[ib][jb] = H16<H>::mfma32(afr[ib], bfr[jb], acc[ib][jb]);
      }
    }
  }

  // ---- sum the 4 waves through LDS, write the block's partial [64][64]
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float* red = reinterpret_cast<float*>(smem);   // [64][64] floats = 16 KB
  // one wave at a time, in wave order: the sum is the same bit pattern on every run
  for (int wv = 0; wv < 4; ++wv) {
    if (wid == wv) {
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = ib * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            float* dst = &red[row * 64 + jb * 32 + r];
            *dst = wv == 0 ? acc[ib][jb][i] : *dst + acc[ib][jb][i];
          }
    }
    __syncthreads();
  }
  float* out = P.slab + ((size_t)split * P.col_tiles + ct) * 4096;
  for (int i = tid; i < 4096; i += 256) out[i] = red[i];
}

// ------------------------------------------------------------------------------------------ wgrad, fp32
// Same problem on the exact-fp32 MFMA (32x32x2): rows i = k = tap*4+ci (two 32-row blocks, 36 used), cols = a
// 32-channel slice of co, GEMM-K = pixels two at a time; no transposes (one float per lane per operand).
__global__ __launch_bounds__(256, 2) void conv_first_wgrad_f32_kernel(FirstWgradParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_dy = smem;                 // [256 px][32 co] fp32, 128-B rows
  char* lds_x = smem + 256 * 128;      // [340][4] fp32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.x % P.col_tiles, split = blockIdx.x / P.col_tiles;    // col tiles of 32 here
  const int t_begin = (int)((long)P.ntiles * split / P.nsplit);
  const int t_end = (int)((long)P.ntiles * (split + 1) / P.nsplit);
  const int r = lane & 31, h = lane >> 5;
  int addrA[2];
  bool rowok[2];
#pragma unroll
  for (int ib = 0; ib < 2; ++ib) {
    const int i = ib * 32 + r, tap = i >> 2, ci = i & 3;
    rowok[ib] = tap < 9;
    const int tp = tap < 9 ? tap : 0;
    addrA[ib] = ((tp / 3) * 34 + (tp % 3) + h) * 16 + ci * 4;
  }
  f32x16 acc[2];
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ib][i] = 0.f;
  const char* zero = reinterpret_cast<const char*>(g_zero_page_first);
  const int lp = lane >> 3, ls = lane & 7;
  for (int tile = t_begin; tile < t_end; ++tile) {
    int tt = tile;
    const int tx = tt % P.tiles_x; tt /= P.tiles_x;
    const int ty = tt % P.tiles_y;
    const int img = tt / P.tiles_y;
    const int y0 = ty * 8, x0 = tx * 32;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float* dimg = reinterpret_cast<const float*>(P.dy) + (size_t)img * P.H * P.W * P.lddy + ct * 32;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int fi = f * 4 + wid;
      const int p = fi * 8 + lp;
      const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
      const char* src = zero;
      if (yy < P.H && xx < P.W) src = reinterpret_cast<const char*>(dimg + (size_t)(yy * P.W + xx) * P.lddy + ls * 4);
      glds16(src, lds_dy + fi * 1024);
    }
    const float* ximg = P.x + (size_t)img * P.H * P.W * P.ldx;
    for (int p = tid; p < 340; p += 256) {
      const int hr = p / 34, hc = p - hr * 34;
      const int yy = y0 + hr - 1, xx = x0 + hc - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) {
        const float* src = ximg + (size_t)(yy * P.W + xx) * P.ldx;
        for (int c = 0; c < P.Cin; ++c) v[c] = src[c];
      }
      *reinterpret_cast<f32x4*>(lds_x + p * 16) = v;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wid + rr;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const float bv = *reinterpret_cast<const float*>(lds_dy + (row * 32 + 2 * s + h) * 128 + r * 4);
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          float av = *reinterpret_cast<const float*>(lds_x + (row * 34 + 2 * s) * 16 + addrA[ib]);
          av = rowok[ib] ? av : 0.f;
          acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ib], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float* red = reinterpret_cast<float*>(smem);   // [64][32]
  for (int wv = 0; wv < 4; ++wv) {                 // in wave order (see the 16-bit kernel)
    if (wid == wv) {
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* dst = &red[(ib * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r];
          *dst = wv == 0 ? acc[ib][i] : *dst + acc[ib][i];
        }
    }
    __syncthreads();
  }
  // slab layout shared with the bf16 kernel: [split][64-col tile][64 rows][64 cols]
  float* out = P.slab + ((size_t)split * (P.col_tiles / 2) + (ct >> 1)) * 4096 + (ct & 1) * 32;
  for (int i = tid; i < 64 * 32; i += 256) out[(i >> 5) * 64 + (i & 31)] = red[i];
}

// dw[co][ci][tap] = sum_split slab[split][ct][k = tap*4+ci][co - ct*64]
// one block per (column tile, k row): 64 channels x 4 split groups, coalesced 256-byte rows of the slab, the groups
// summed in a fixed order through LDS (a thread per output walking all 512 splits serially took 0.16 ms)
__global__ __launch_bounds__(256) void conv_first_wgrad_reduce_kernel(const float* __restrict__ slab, float* dw,
                                                                      int nsplit, int col_tiles, int Cin, int Cout) {
  __shared__ float part[4][64];
  const int ct = blockIdx.x / 36, k = blockIdx.x % 36;
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int tap = k >> 2, ci = k & 3;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* base = slab + (size_t)ct * 4096 + k * 64 + c;
  const size_t step = (size_t)col_tiles * 4096;
  int sp = grp;
  for (; sp + 12 < nsplit; sp += 16) {
    s0 += base[(size_t)sp * step];
    s1 += base[(size_t)(sp + 4) * step];
    s2 += base[(size_t)(sp + 8) * step];
    s3 += base[(size_t)(sp + 12) * step];
  }
  for (; sp < nsplit; sp += 4) s0 += base[(size_t)sp * step];
  part[grp][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && ci < Cin) {
    const int co = ct * 64 + c;
    if (co < Cout) dw[((size_t)co * Cin + ci) * 9 + tap] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
  }
}

int first_nsplit(int ntiles, int col_tiles) {
  UNETK_KNOB(slots, "UNETK_FIRST_SPLIT", 1024);   // four 256-thread workgroups per CU (35 KiB LDS each): 0.378 -> 0.351 ms
  int ns = slots / col_tiles;
  if (ns < 1) ns = 1;
  if (ns > ntiles) ns = ntiles;
  return ns;
}

// ---------------------------------------------------------------------------------------------------------------
// Gradient w.r.t. the INPUT IMAGE of the first layer (autograd of unet_parts.py:16 for the network input; the
// reference's training loop never asks for it -- saliency maps / adversarial examples do):
//   dx[q][ci] = sum_{tap, co} dy[q - (tap - 1)][co] * w[co][ci][tap]
// c_in <= 4 outputs per pixel from 9 x c_out gradient values: a memory-bound gather (dy is read once from HBM, the
// nine taps of neighbouring pixels hit the caches), one thread per pixel, the whole filter in LDS as [tap][co][4].
template <typename T>
__global__ __launch_bounds__(256) void conv_first_dgrad_kernel(const T* __restrict__ dy, int lddy,
                                                               const float* __restrict__ w, float* __restrict__ dx,
                                                               int lddx, int N, int H, int W, int Cin, int Cout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* wl = reinterpret_cast<f32x4*>(smem);            // [9][Cout] x (ci 0..3, zero-padded)
  for (int i = threadIdx.x; i < 9 * Cout; i += blockDim.x) {
    const int tap = i / Cout, co = i - tap * Cout;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int ci = 0; ci < Cin; ++ci) v[ci] = w[((size_t)co * Cin + ci) * 9 + tap];
    wl[i] = v;
  }
  __syncthreads();
  constexpr int E = ElemTraits<T>::E;
  const long npix = (long)N * H * W;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < npix; q += (long)gridDim.x * blockDim.x) {
    const int x = (int)(q % W);
    const long r = q / W;
    const int y = (int)(r % H);
    const long img = r / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < 3; ++kh) {
      const int yy = y - (kh - 1);
      if (yy < 0 || yy >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int xx = x - (kw - 1);
        if (xx < 0 || xx >= W) continue;
        const T* p = dy + ((img * H + yy) * (long)W + xx) * lddy;
        const f32x4* wt = wl + (kh * 3 + kw) * Cout;
        for (int c0 = 0; c0 < Cout; c0 += E) {
          T v[E];
          *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(p + c0);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const float g = to_f32(v[e]);
            const f32x4 ww = wt[c0 + e];
            acc[0] = fmaf(g, ww[0], acc[0]);
            acc[1] = fmaf(g, ww[1], acc[1]);
            acc[2] = fmaf(g, ww[2], acc[2]);
            acc[3] = fmaf(g, ww[3], acc[3]);
          }
        }
      }
    }
    float* o = dx + q * lddx;
    for (int ci = 0; ci < Cin; ++ci) o[ci] = acc[ci];
  }
}

}  // namespace

extern "C" {

// mode: the kernel's MODE (1: y may be NULL; 2: act / lda give the activation's buffer)
static int first_fprop(const float* x, int ldx, const float* w, void* y, int ldy, double* stats, const float* scale,
                       const float* shift, int n, int h, int wd, int c_in, int c_out, int dtype, void* stream,
                       int mode = 0, void* act = nullptr, int lda = 0) {
  if (!x || !w || n <= 0 || h <= 0 || wd <= 0) return UNETK_ERR_ARG;
  if (mode == 1 ? !stats : !y) return UNETK_ERR_ARG;
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  if (c_in < 1 || c_in > 4 || c_out % 64 || c_out <= 0 || ldx < c_in) return UNETK_ERR_SHAPE;
  const int es = unetk_esize(dtype);
  if (mode != 1) {
    if (ldy < c_out) return UNETK_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(y) & 15) || (ldy * es) % 16) return UNETK_ERR_ALIGN;
    if ((long)h * wd * ldy >= 0x7fffffffL) return UNETK_ERR_SHAPE;
  } else {
    y = nullptr; ldy = c_out;
  }
  if (mode == 2) {
    if (!act || !scale || !shift) return UNETK_ERR_ARG;
    if (lda < c_out || (long)h * wd * lda >= 0x7fffffffL) return UNETK_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(act) & 15) || (lda * es) % 16) return UNETK_ERR_ALIGN;
  }
  FirstParams P{};
  P.x = x; P.w = w; P.y = y; P.stats = stats; P.aff_scale = scale; P.aff_shift = shift; P.act = act; P.lda = lda;
  P.N = n; P.H = h; P.W = wd; P.Cin = c_in; P.Cout = c_out; P.ldx = ldx; P.ldy = ldy;
  P.tiles_x = host_cdiv(wd, 32); P.tiles_y = host_cdiv(h, 8); P.n_tiles = c_out / 64;
  const long spatial = (long)n * P.tiles_x * P.tiles_y;
  if (spatial > 0x7fffffffL) return UNETK_ERR_SHAPE;
  // ~8 workgroups per CU, each walking its share of the spatial tiles (4 fit a CU at a time)
  const long groups = spatial < 2048 ? spatial : 2048;
  const long nblk = groups * P.n_tiles;
  const size_t lds = 340 * 16 + 5 * 2 * 1024;
  if (mode == 1) {
    UNETK_WITH_T(dtype, hipLaunchKernelGGL((conv_first_fprop_kernel<T, 1>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, P));
  } else if (mode == 2) {
    UNETK_WITH_T(dtype, hipLaunchKernelGGL((conv_first_fprop_kernel<T, 2>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, P));
  } else {
    UNETK_WITH_T(dtype, hipLaunchKernelGGL(conv_first_fprop_kernel<T>, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, P));
  }
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_conv3x3_first_fprop(const float* x, int ldx, const float* w, void* y, int ldy, double* stats, int n, int h,
                              int wd, int c_in, int c_out, int dtype, void* stream) {
  return first_fprop(x, ldx, w, y, ldy, stats, nullptr, nullptr, n, h, wd, c_in, c_out, dtype, stream, y ? 0 : 1);
}

int unetk_conv3x3_first_fprop_bn_relu_train(const float* x, int ldx, const float* w, const float* scale,
                                            const float* shift, void* y, int ldy, void* act, int lda, int n, int h, int wd,
                                            int c_in, int c_out, int dtype, void* stream) {
  return first_fprop(x, ldx, w, y, ldy, nullptr, scale, shift, n, h, wd, c_in, c_out, dtype, stream, 2, act, lda);
}

int unetk_conv3x3_first_fprop_bn_relu(const float* x, int ldx, const float* w, const float* scale, const float* shift,
                                      void* y, int ldy, int n, int h, int wd, int c_in, int c_out, int dtype,
                                      void* stream) {
  if (!scale || !shift) return UNETK_ERR_ARG;
  return first_fprop(x, ldx, w, y, ldy, nullptr, scale, shift, n, h, wd, c_in, c_out, dtype, stream);
}

int64_t unetk_conv3x3_first_wgrad_workspace_bytes(int n, int h, int w, int c_out) {
  if (n <= 0 || h <= 0 || w <= 0 || c_out <= 0 || c_out % 64) return UNETK_ERR_SHAPE;
  const int ntiles = n * host_cdiv(h, 8) * host_cdiv(w, 32);
  return (int64_t)first_nsplit(ntiles, c_out / 64) * (c_out / 64) * 4096 * sizeof(float);
}

static int first_wgrad(const float* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                       int64_t workspace_bytes, int n, int h, int w, int c_in, int c_out, int dtype, void* stream,
                       const void* z, int ldz, const float* scale, const float* shift, const float* mean,
                       const float* coef) {
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  const int64_t need = unetk_conv3x3_first_wgrad_workspace_bytes(n, h, w, c_out);
  if (need < 0) return (int)need;
  if (!x || !dy || !dw || !workspace || workspace_bytes < need) return UNETK_ERR_ARG;
  if (c_in < 1 || c_in > 4 || ldx < c_in || lddy < c_out) return UNETK_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(dy) & 15) || (lddy * unetk_esize(dtype)) % 16) return UNETK_ERR_ALIGN;
  FirstWgradParams P{};
  P.x = x; P.dy = dy; P.slab = (float*)workspace;
  P.N = n; P.H = h; P.W = w; P.Cin = c_in; P.Cout = c_out; P.ldx = ldx; P.lddy = lddy;
  P.tiles_x = host_cdiv(w, 32); P.tiles_y = host_cdiv(h, 8); P.ntiles = n * P.tiles_x * P.tiles_y;
  P.col_tiles = c_out / 64;
  P.nsplit = first_nsplit(P.ntiles, P.col_tiles);
  if (z != nullptr) {
    if (!unetk_is16(dtype)) return UNETK_ERR_DTYPE;          // callers run the apply pass + the plain kernel in fp32
    if (!scale || !shift || !mean || !coef) return UNETK_ERR_ARG;
    if (ldz < c_out) return UNETK_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(z) & 15) || (ldz * 2) % 16) return UNETK_ERR_ALIGN;
    P.z = z; P.ldz = ldz; P.bn_scale = scale; P.bn_shift = shift; P.bn_mean = mean; P.bn_coef = coef;
    const size_t lds = 256 * 128 + 340 * 8;
    UNETK_WITH_H(dtype, hipLaunchKernelGGL((conv_first_wgrad_bf16_kernel<H, true>), dim3(P.nsplit * P.col_tiles),
                                           dim3(256), lds, (hipStream_t)stream, P));
  } else if (unetk_is16(dtype)) {
    const size_t lds = 256 * 128 + 340 * 8;
    UNETK_WITH_H(dtype, hipLaunchKernelGGL(conv_first_wgrad_bf16_kernel<H>, dim3(P.nsplit * P.col_tiles), dim3(256), lds,
                                           (hipStream_t)stream, P));
  } else {
    FirstWgradParams Q = P;
    Q.col_tiles = c_out / 32;     // the fp32 kernel works on 32-channel column tiles, same slab layout
    const size_t lds = 256 * 128 + 340 * 16;
    hipLaunchKernelGGL(conv_first_wgrad_f32_kernel, dim3(Q.nsplit * Q.col_tiles), dim3(256), lds, (hipStream_t)stream,
                       Q);
  }
  UNETK_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(P.col_tiles * 36), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, dw, P.nsplit, P.col_tiles, c_in, c_out);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_conv3x3_first_wgrad(const float* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                              int64_t workspace_bytes, int n, int h, int w, int c_in, int c_out, int dtype,
                              void* stream) {
  return first_wgrad(x, ldx, dy, lddy, dw, workspace, workspace_bytes, n, h, w, c_in, c_out, dtype, stream, nullptr, 0,
                     nullptr, nullptr, nullptr, nullptr);
}

int unetk_conv3x3_first_wgrad_bn(const float* x, int ldx, const void* dout, int lddo, const void* z, int ldz,
                                 const float* scale, const float* shift, const float* mean, const float* coef,
                                 float* dw, void* workspace, int64_t workspace_bytes, int n, int h, int w, int c_in,
                                 int c_out, int dtype, void* stream) {
  if (!z) return UNETK_ERR_ARG;
  return first_wgrad(x, ldx, dout, lddo, dw, workspace, workspace_bytes, n, h, w, c_in, c_out, dtype, stream, z, ldz,
                     scale, shift, mean, coef);
}

int unetk_conv3x3_first_dgrad(const void* dy, int lddy, const float* w, float* dx, int lddx, int n, int h, int wd,
                              int c_in, int c_out, int dtype, void* stream) {
  if (!dy || !w || !dx || n <= 0 || h <= 0 || wd <= 0) return UNETK_ERR_ARG;
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  if (c_in < 1 || c_in > 4 || c_out % 64 || c_out <= 0 || c_out > 512 || lddx < c_in || lddy < c_out) return UNETK_ERR_SHAPE;
  const int es = unetk_esize(dtype);
  if ((reinterpret_cast<uintptr_t>(dy) & 15) || (lddy * es) % 16) return UNETK_ERR_ALIGN;
  const long npix = (long)n * h * wd;
  long g = (npix + 255) / 256;
  if (g > 4096) g = 4096;
  const size_t lds = (size_t)9 * c_out * 16;
  UNETK_WITH_T(dtype, hipLaunchKernelGGL(conv_first_dgrad_kernel<T>, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream,
                       (const T*)dy, lddy, w, dx, lddx, n, h, wd, c_in, c_out));
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
