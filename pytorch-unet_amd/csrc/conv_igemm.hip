// NHWC implicit-GEMM convolution on MFMA for gfx950.
//
//   GEMM view:  M = output pixels (an 8 x 32 spatial tile per workgroup = 256 rows)
//               N = output channels (BN = 32*NB per workgroup)
//               K = taps x input channels, walked in chunks of 32*KS bytes per pixel.
//
// One workgroup = 256 threads = 4 waves; wave w owns tile rows 2w, 2w+1 (two 32-pixel MFMA row blocks)
// x all BN channels.  Per chunk the input tile (with its 1-pixel halo for 3x3) is staged ONCE into LDS
// (registers -> ds_write: zero fill for the padding comes for free and the loads of chunk k+1 fly under
// the MFMAs of chunk k), the packed weights of that chunk arrive by LDS-DMA (global_load_lds, already in
// fragment order), and all taps are served from the same LDS image by shifting the read address.
//
// LDS pixel rows are 32*KS bytes; 16-byte pieces are XOR-swizzled with the pixel index so that the
// 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots (cdna_hip_programming.md T2).
//
// MODES   0 CONV3  3x3, pad 1 (fprop, and dgrad with rotated weights)        unet_parts.py:16,19
//         1 PWT    ConvTranspose2d k2 s2 forward: pointwise GEMM with N = 4*Co, pixel-shuffle store into
//                  the concat buffer at a channel offset, + bias               unet_parts.py:53,62-67
//         2 CTD    ConvTranspose2d k2 s2 dgrad: 4 stride-2 taps as outer K iterations
#include "conv_common.h"

namespace {

struct ConvParams {
  const void* x;
  const void* wpk;
  void* y;
  const float* bias;
  double* stats;
  int N, H, W;       // tile grid domain (CONV3: image; PWT/CTD: the low-res grid)
  int Cin, Cout;     // GEMM K channels per tap, GEMM N channels
  int ldx, ldy, xoff, yoff;
  int Hs, Ws;        // PWT: destination image; CTD: source (dy) image
  int padT, padL;
  int Co;            // PWT: channels of the up-sampled half (Cout == 4*Co)
  int tiles_x, tiles_y, n_tiles;
};

template <int MODE> struct Geo;
template <> struct Geo<0> { static constexpr int TAPS = 9, KS = 2, NPIX = 10 * 34, OUTER = 1; };
template <> struct Geo<1> { static constexpr int TAPS = 1, KS = 4, NPIX = 256, OUTER = 1; };
template <> struct Geo<2> { static constexpr int TAPS = 1, KS = 4, NPIX = 256, OUTER = 4; };

template <typename T, int NB, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvParams P) {
  using G = Geo<MODE>;
  constexpr int E = ElemTraits<T>::E;   // elements per 16-byte piece
  constexpr int KS = G::KS;             // k-steps per chunk (one k-step = 2 pieces = 2E channels)
  constexpr int NP = 2 * KS;            // pieces per pixel per chunk
  constexpr int CH = NP * E;            // channels per chunk
  constexpr int S = KS * 32;            // LDS bytes per pixel
  constexpr int PPR = 256 / S;          // pixels per 256-byte bank row
  constexpr int TAPS = G::TAPS;
  constexpr int NPIX = G::NPIX;
  constexpr int NPIECE = NPIX * NP;
  constexpr int R = (NPIECE + 255) / 256;
  constexpr int BN = 32 * NB;
  constexpr int IN_BYTES = NPIX * S;
  constexpr int WFRAGS = TAPS * KS * NB;        // 1-KiB fragments per chunk

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_in = smem;
  char* lds_w = smem + IN_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;

  // ---- block -> (n_tile, image, tile origin)
  int bid = blockIdx.x;
  const int nt = bid % P.n_tiles;
  bid /= P.n_tiles;
  const int tx = bid % P.tiles_x;
  bid /= P.tiles_x;
  const int ty = bid % P.tiles_y;
  const int img = bid / P.tiles_y;
  const int y0 = ty * 8, x0 = tx * 32;

  const int Hx = (MODE == 2) ? P.Hs : P.H, Wx = (MODE == 2) ? P.Ws : P.W;
  const T* ximg = reinterpret_cast<const T*>(P.x) + (size_t)img * Hx * Wx * P.ldx + P.xoff;

  // ---- per-thread staging descriptors (chunk-invariant)
  int poff[R];    // element offset of this thread's piece inside the image, -1 = zero fill / unused
  int ldst[R];    // LDS byte address of the piece
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int id = r * 256 + tid;
    const int p = id / NP, c = id % NP;
    int off = -1;
    if (id < NPIECE) {
      int yy, xx;
      if constexpr (MODE == 0) {
        const int hr = p / 34, hc = p - hr * 34;
        yy = y0 + hr - 1;
        xx = x0 + hc - 1;
        if (yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) off = (yy * P.W + xx) * P.ldx + c * E;
      } else {
        yy = y0 + (p >> 5);
        xx = x0 + (p & 31);
        if (yy < P.H && xx < P.W) {
          if constexpr (MODE == 1) off = (yy * P.W + xx) * P.ldx + c * E;
          else off = ((2 * yy + P.padT) * P.Ws + 2 * xx + P.padL) * P.ldx + c * E;
        }
      }
    }
    poff[r] = off;
    ldst[r] = p * S + ((c ^ ((p / PPR) & (NP - 1))) << 4);
  }

  // ---- per-lane A-fragment read addresses (k-step 0; k-step s is `^ (s << 5)`)
  int aaddr[2][TAPS];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      int p;
      if constexpr (MODE == 0) p = (2 * wid + mb + t / 3) * 34 + j + (t % 3);
      else p = (2 * wid + mb) * 32 + j;
      aaddr[mb][t] = p * S + ((h ^ ((p / PPR) & (NP - 1))) << 4);
    }
  }

  f32x16 acc[2][NB];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  const int nchunks = P.Cin / CH;
  const int niter = nchunks * G::OUTER;
  const char* wbase = reinterpret_cast<const char*>(P.wpk) + (size_t)nt * niter * (WFRAGS * 1024);

  auto iter_offset = [&](int it) -> int {
    if constexpr (MODE == 2) {
      const int tap = it / nchunks, kc = it - tap * nchunks;
      return ((tap >> 1) * P.Ws + (tap & 1)) * P.ldx + kc * CH;
    } else {
      return it * CH;
    }
  };

  uint4 stage[R];
  auto load_stage = [&](int it) {
    const int io = iter_offset(it);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // ALWAYS issue the load (clamped address, result discarded for padding): the counted s_waitcnt below
      // relies on every wave having exactly R prefetch loads younger than its LDS-DMA.
      // (the zero fill for padding is applied when the piece is written to LDS, so that nothing here consumes
      // a load result and forces an early s_waitcnt)
      stage[r] = *reinterpret_cast<const uint4*>(ximg + (poff[r] >= 0 ? poff[r] : 0) + io);
    }
  };

  load_stage(0);

  for (int it = 0; it < niter; ++it) {
    // [A] every wave has finished reading the previous chunk's LDS image
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r * 256 + tid < NPIECE)
        *reinterpret_cast<uint4*>(lds_in + ldst[r]) = poff[r] >= 0 ? stage[r] : uint4{0u, 0u, 0u, 0u};
    // weights of this chunk: WFRAGS x 1 KiB, fragment order, straight into LDS
    {
      const char* wsrc = wbase + (size_t)it * (WFRAGS * 1024) + lane * 16;
#pragma unroll
      for (int f = 0; f < (WFRAGS + 3) / 4; ++f) {
        const int fi = f * 4 + wid;
        if (fi < WFRAGS) glds16(wsrc + fi * 1024, lds_w + fi * 1024);
      }
    }
    const bool more = (it + 1 < niter);
    asm volatile("" ::: "memory");   // keep the prefetch loads YOUNGER than the LDS-DMA in the vmcnt queue
    if (more) {
      load_stage(it + 1);
      asm volatile("" ::: "memory");
      // the R youngest vector-memory ops are the prefetch; everything older (the LDS-DMA) must have landed
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(R) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- all taps of this chunk from the one LDS image
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 a[2], b[NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) a[mb] = *reinterpret_cast<const uint4*>(lds_in + (aaddr[mb][t] ^ (ks << 5)));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          b[nb] = *reinterpret_cast<const uint4*>(lds_w + ((t * KS + ks) * NB + nb) * 1024 + lane * 16);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) Mma<T>::run(acc[mb][nb], a[mb], b[nb]);
      }
    }
  }

  // ---- epilogue.  C layout of 32x32 MFMA: column = lane&31 (-> channels NB*j .. NB*j+NB-1 of this
  // n-tile, by the weight packing), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (-> pixel column in the tile).
  const int ch0 = nt * BN + NB * j;     // first of this lane's NB consecutive channels
  float ssum[NB], ssq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) ssum[nb] = ssq[nb] = 0.f;

  float bias[NB];
  int co0 = ch0, ab = 0;
  if constexpr (MODE == 1) {
    ab = (nt * BN) / P.Co;
    co0 = ch0 - ab * P.Co;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias[nb] = P.bias[co0 + nb];
  }

  T* yimg;
  if constexpr (MODE == 1) yimg = reinterpret_cast<T*>(P.y) + (size_t)img * P.Hs * P.Ws * P.ldy + P.yoff;
  else yimg = reinterpret_cast<T*>(P.y) + (size_t)img * P.H * P.W * P.ldy + P.yoff;

#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int yy = y0 + 2 * wid + mb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int xx = x0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (yy < P.H && xx < P.W) {
        float v[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float f = acc[mb][nb][i];
          if constexpr (MODE == 1) f += bias[nb];
          f = to_f32(from_f32<T>(f));      // statistics are taken on the value as stored
          v[nb] = f;
          ssum[nb] += f;
          ssq[nb] += f * f;
        }
        T* dst;
        if constexpr (MODE == 1)
          dst = yimg + (size_t)((2 * yy + (ab >> 1) + P.padT) * P.Ws + 2 * xx + (ab & 1) + P.padL) * P.ldy + co0;
        else
          dst = yimg + (size_t)(yy * P.W + xx) * P.ldy + ch0;
        store_channels<T, NB>(dst, v);
      }
    }
  }

  if (MODE == 0 && P.stats != nullptr) {
    // lanes l and l+32 hold the same channels (different pixel columns)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      ssum[nb] += __shfl_xor(ssum[nb], 32, 64);
      ssq[nb] += __shfl_xor(ssq[nb], 32, 64);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS image no longer needed
    float* red = reinterpret_cast<float*>(smem);                       // [4 waves][BN][2]
    if (h == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        red[(wid * BN + NB * j + nb) * 2 + 0] = ssum[nb];
        red[(wid * BN + NB * j + nb) * 2 + 1] = ssq[nb];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const float v = red[tid] + red[2 * BN + tid] + red[4 * BN + tid] + red[6 * BN + tid];
      double* dst = P.stats + ((size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * P.Cout + nt * BN) * 2 + tid;
      unsafeAtomicAdd(dst, (double)v);
    }
  }
}

template <typename T, int NB, int MODE> size_t lds_bytes() {
  using G = Geo<MODE>;
  size_t main_bytes = (size_t)G::NPIX * G::KS * 32 + (size_t)G::TAPS * G::KS * NB * 1024;
  size_t red_bytes = 4 * 32 * NB * 2 * sizeof(float);
  return main_bytes > red_bytes ? main_bytes : red_bytes;
}

template <typename T, int NB, int MODE> int launch(const ConvParams& P, hipStream_t s) {
  const size_t lds = lds_bytes<T, NB, MODE>();
  static bool attr_done = false;   // benign race: idempotent
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, NB, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNETK_ERR_LAUNCH;
    attr_done = true;
  }
  const long nblk = (long)P.N * P.tiles_y * P.tiles_x * P.n_tiles;
  if (nblk <= 0 || nblk > 0x7fffffffL) return UNETK_ERR_SHAPE;
  hipLaunchKernelGGL((conv_igemm_kernel<T, NB, MODE>), dim3((unsigned)nblk), dim3(256), lds, s, P);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

template <int MODE> int dispatch(const ConvParams& P, int dtype, hipStream_t s) {
  if (dtype == UNETK_BF16) return launch<bf16, 2, MODE>(P, s);
  if (dtype == UNETK_F32) return launch<float, 2, MODE>(P, s);
  return UNETK_ERR_DTYPE;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_common(const void* x, int ldx, int xoff, const void* y, int ldy, int yoff, int cin, int cout, int dtype,
                 int chunk_bytes) {
  if (dtype != UNETK_BF16 && dtype != UNETK_F32) return UNETK_ERR_DTYPE;
  const int es = dtype == UNETK_BF16 ? 2 : 4;
  const int ch = chunk_bytes / es;
  if (cin <= 0 || cout <= 0 || cin % ch != 0 || cout % 64 != 0) return UNETK_ERR_SHAPE;
  if (ldx < xoff + cin || ldy < yoff + cout || xoff < 0 || yoff < 0) return UNETK_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(y) || (ldx * es) % 16 || (xoff * es) % 16 || (ldy * es) % 16 || (yoff * es) % 16)
    return UNETK_ERR_ALIGN;
  return UNETK_OK;
}

int conv3x3_common(const void* x, int ldx, int xoff, const void* wpk, void* y, int ldy, int yoff, double* stats,
                   int n, int h, int w, int cin, int cout, int dtype, void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || !x || !wpk || !y) return UNETK_ERR_ARG;
  if (int e = check_common(x, ldx, xoff, y, ldy, yoff, cin, cout, dtype, 64)) return e;
  if ((long)h * w * (ldx > ldy ? ldx : ldy) >= 0x7fffffffL) return UNETK_ERR_SHAPE;   // int32 in-image offsets
  ConvParams P{};
  P.x = x; P.wpk = wpk; P.y = y; P.stats = stats;
  P.N = n; P.H = h; P.W = w; P.Cin = cin; P.Cout = cout;
  P.ldx = ldx; P.ldy = ldy; P.xoff = xoff; P.yoff = yoff;
  P.tiles_x = host_cdiv(w, 32); P.tiles_y = host_cdiv(h, 8); P.n_tiles = cout / 64;
  return dispatch<0>(P, dtype, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int unetk_conv3x3_fprop(const void* x, int ldx, int x_choff, const void* wpk, void* y, int ldy, int y_choff,
                        double* stats, int n, int h, int w, int c_in, int c_out, int dtype, void* stream) {
  return conv3x3_common(x, ldx, x_choff, wpk, y, ldy, y_choff, stats, n, h, w, c_in, c_out, dtype, stream);
}

int unetk_conv3x3_dgrad(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int dx_choff,
                        int n, int h, int w, int c_in, int c_out, int dtype, void* stream) {
  // GEMM-K = forward output channels, GEMM-N = forward input channels
  return conv3x3_common(dy, lddy, dy_choff, wpk, dx, lddx, dx_choff, nullptr, n, h, w, c_out, c_in, dtype, stream);
}

int unetk_convT2x2_fwd(const void* x, int ldx, const void* wpk, const float* bias, void* y, int ldy, int y_choff,
                       int n, int h, int w, int c_in, int c_out, int hs, int ws, int pad_t, int pad_l, int dtype,
                       void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || !x || !wpk || !y || !bias) return UNETK_ERR_ARG;
  if (int e = check_common(x, ldx, 0, y, ldy, y_choff, c_in, c_out, dtype, 128)) return e;
  if (pad_t < 0 || pad_l < 0 || pad_t + 2 * h > hs || pad_l + 2 * w > ws) return UNETK_ERR_SHAPE;
  if ((long)hs * ws * ldy >= 0x7fffffffL || (long)h * w * ldx >= 0x7fffffffL) return UNETK_ERR_SHAPE;
  ConvParams P{};
  P.x = x; P.wpk = wpk; P.y = y; P.bias = bias;
  P.N = n; P.H = h; P.W = w; P.Cin = c_in; P.Cout = 4 * c_out; P.Co = c_out;
  P.ldx = ldx; P.ldy = ldy; P.xoff = 0; P.yoff = y_choff;
  P.Hs = hs; P.Ws = ws; P.padT = pad_t; P.padL = pad_l;
  P.tiles_x = host_cdiv(w, 32); P.tiles_y = host_cdiv(h, 8); P.n_tiles = 4 * c_out / 64;
  return dispatch<1>(P, dtype, (hipStream_t)stream);
}

int unetk_convT2x2_dgrad(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int n, int h,
                         int w, int c_in, int c_out, int hs, int ws, int pad_t, int pad_l, int dtype, void* stream) {
  if (n <= 0 || h <= 0 || w <= 0 || !dy || !wpk || !dx) return UNETK_ERR_ARG;
  // GEMM-K per tap = c_out (channels of the up-sampled gradient), GEMM-N = c_in
  if (int e = check_common(dy, lddy, dy_choff, dx, lddx, 0, c_out, c_in, dtype, 128)) return e;
  if (pad_t < 0 || pad_l < 0 || pad_t + 2 * h > hs || pad_l + 2 * w > ws) return UNETK_ERR_SHAPE;
  if ((long)hs * ws * lddy >= 0x7fffffffL || (long)h * w * lddx >= 0x7fffffffL) return UNETK_ERR_SHAPE;
  ConvParams P{};
  P.x = dy; P.wpk = wpk; P.y = dx;
  P.N = n; P.H = h; P.W = w; P.Cin = c_out; P.Cout = c_in;
  P.ldx = lddy; P.ldy = lddx; P.xoff = dy_choff; P.yoff = 0;
  P.Hs = hs; P.Ws = ws; P.padT = pad_t; P.padL = pad_l;
  P.tiles_x = host_cdiv(w, 32); P.tiles_y = host_cdiv(h, 8); P.n_tiles = c_in / 64;
  return dispatch<2>(P, dtype, (hipStream_t)stream);
}

}  // extern "C"
