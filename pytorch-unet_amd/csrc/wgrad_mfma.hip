// Weight gradients of conv3x3 (unet_parts.py:16,19) and ConvTranspose2d k2 s2 (unet_parts.py:53) on MFMA.
//
//   dw[tap][r][c] = sum over pixels  A[pixel + tap][r] * B[pixel][c]
//
//   conv3x3 : A = layer input (rows r = ci, 9 taps served by SHIFTED reads of one halo tile in LDS),
//             B = dy (cols c = co)
//   convT   : A = dy of the up-sampled half, gathered with stride 2 per tap (rows r = co), B = x (cols c = ci)
//
// GEMM-K is the pixel index, which is the SLOW index of NHWC tensors, so both MFMA operands need a
// transposed fragment: tiles are staged [pixel][64 channels] by LDS-DMA (global_load_lds, zero page for the
// padding / ragged edge) and read with ds_read_b64_tr_b16 (cdna_hip_programming.md T10): 16 lanes fetch a
// 4-pixel x 16-channel block and receive it channel-major.
//
// One workgroup = 4 waves as 2 (rows) x 2 (cols) of 32x32 MFMA blocks = a 64 x 64 channel tile, all taps
// (9 x 16 accumulator registers per lane), and walks a contiguous range of 8x32-pixel tiles (split-K over
// the grid).  Partial sums go to a [split][tap][row][col] fp32 slab; a second kernel adds the splits in a
// fixed order (bitwise reproducible, no float atomics) and writes PyTorch's parameter layout.
#include "common.h"

#ifndef UNETK_WGRAD_REDUCE_V4
#define UNETK_WGRAD_REDUCE_V4 1      /* split-K reduction with 16-byte slab loads (0: A/B builds) */
#endif

namespace {

typedef unsigned short u16;   // a 16-bit element as the data movement sees it (bf16 or fp16)

__device__ uint4 g_zero_page[8];   // zero-initialised: source of padding for LDS-DMA

struct WgradParams {
  const u16* a;      // tap-shifted operand
  const u16* b;      // fixed operand
  float* slab;
  int N, H, W;        // tile-grid domain (conv3: image; convT: low-res grid)
  int Ca, Cb;         // channels of a (rows) and b (cols)
  int lda, ldb;
  int Ha, Wa;         // convT: dims of the image `a` lives in (the skip-sized concat gradient)
  int padT, padL;
  int tiles_x, tiles_y, ntiles, nsplit, row_tiles, col_tiles;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned frag5_t;   // one 16-byte MFMA operand fragment (eight 16-bit elements)

__device__ __forceinline__ frag5_t tr_read8(const char* lds0, const char* lds1) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)lds0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)lds1);
  return __builtin_bit_cast(frag5_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// swizzle of the 16-byte piece index inside a 128-byte pixel row: pixels p and p+2 (same 128-byte half of
// a bank row) land in different 64-byte groups -> the 32-lane groups of the transposed read are conflict free
__device__ __forceinline__ int swz(int p) { return ((p >> 1) & 1) << 2; }


// H = the 16-bit element type of the kernel (bf16 or f16): the kernels move 16-bit words (u16) and differ only here.
// (The `_bf16_` in the kernel names dates from the bf16-only generation and is kept: profiles/ refers to them.)
#define WG_MFMA(a, b, c) \
  H16<H>::mfma32(__builtin_bit_cast(typename H16<H>::x8, a), __builtin_bit_cast(typename H16<H>::x8, b), c)

template <typename H, int MODE>   // 0 conv3x3, 1 convT2x2
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(WgradParams P) {
  constexpr int TAPS = MODE == 0 ? 9 : 4;
  constexpr int APITCH = MODE == 0 ? 36 : 32;            // pixels per LDS row of the A tile (multiple of 4)
  constexpr int AROWS = MODE == 0 ? 10 : 8;
  constexpr int A_BYTES = 10 * 36 * 128;                 // region size (conv3 needs it all)
  constexpr int A_INSTR = AROWS * APITCH * 8 / 64;       // 1-KiB LDS-DMA pieces per A tile
  constexpr int B_INSTR = 32;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_b = smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 1, wj = wid & 1;

  const int npairs = P.row_tiles * P.col_tiles;
  int lin = blockIdx.x;     // XCD-major order: the channel-tile pairs of one split share an XCD's L2 (see wgrad3)
  if ((gridDim.x & 7) == 0) lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int pair = lin % npairs, split = lin / npairs;
  const int rt = pair / P.col_tiles, ct = pair % P.col_tiles;
  const int r0 = rt * 64, c0 = ct * 64;

  const int t_begin = (int)((long)P.ntiles * split / P.nsplit);
  const int t_end = (int)((long)P.ntiles * (split + 1) / P.nsplit);

  // ---- per-lane transposed-read addresses (tile invariant)
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int sub = (pp & 1) * 8;
  constexpr int NKW = MODE == 0 ? 3 : 1;
  int addrA[NKW][2][2], addrB[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int pb = ks * 16 + 8 * (g >> 1) + 4 * rd + q;
#pragma unroll
      for (int kw = 0; kw < NKW; ++kw) {
        const int p = pb + kw;
        const int c = wi * 4 + 2 * (g & 1) + (pp >> 1);
        addrA[kw][ks][rd] = p * 128 + ((c ^ swz(p)) << 4) + sub;
      }
      const int cb = wj * 4 + 2 * (g & 1) + (pp >> 1);
      addrB[ks][rd] = pb * 128 + ((cb ^ swz(pb)) << 4) + sub;
    }

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const int lp = lane >> 3, ls = lane & 7;    // pixel-in-piece, physical 16-byte slot

  for (int tile = t_begin; tile < t_end; ++tile) {
    int tt = tile;
    const int tx = tt % P.tiles_x; tt /= P.tiles_x;
    const int ty = tt % P.tiles_y;
    const int img = tt / P.tiles_y;
    const int y0 = ty * 8, x0 = tx * 32;

    // B tile: 256 pixels x 64 channels of the un-shifted operand
    const u16* bimg = P.b + (size_t)img * P.H * P.W * P.ldb + c0;
    auto issue_b = [&]() {
#pragma unroll
      for (int f = 0; f < B_INSTR / 4; ++f) {
        const int fi = f * 4 + wid;
        const int p = fi * 8 + lp;
        const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
        const int c = ls ^ swz(p);
        const char* src = zero;
        if (yy < P.H && xx < P.W) src = reinterpret_cast<const char*>(bimg + (size_t)(yy * P.W + xx) * P.ldb + c * 8);
        glds16(src, lds_b + fi * 1024);
      }
    };
    auto issue_a = [&](int tap) {
      const u16* aimg = P.a + (size_t)img * (MODE == 0 ? P.H * P.W : P.Ha * P.Wa) * P.lda + r0;
#pragma unroll
      for (int f = 0; f < (A_INSTR + 3) / 4; ++f) {
        const int fi = f * 4 + wid;
        if (fi < A_INSTR) {
          const int p = fi * 8 + lp;
          const int c = ls ^ swz(p);
          const char* src = zero;
          if constexpr (MODE == 0) {
            const int hr = p / APITCH, hc = p - hr * APITCH;
            const int yy = y0 + hr - 1, xx = x0 + hc - 1;
            if (hc < 34 && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W)
              src = reinterpret_cast<const char*>(aimg + (size_t)(yy * P.W + xx) * P.lda + c * 8);
          } else {
            const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
            if (yy < P.H && xx < P.W)
              src = reinterpret_cast<const char*>(
                  aimg + (size_t)((2 * yy + (tap >> 1) + P.padT) * P.Wa + 2 * xx + (tap & 1) + P.padL) * P.lda + c * 8);
          }
          glds16(src, lds_a + fi * 1024);
        }
      }
    };

    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // previous tile fully consumed
    issue_b();
    if constexpr (MODE == 0) {
      issue_a(0);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
      for (int tr = 0; tr < 8; ++tr) {
        const char* arow = lds_a + tr * (APITCH * 128);
        const char* brow = lds_b + tr * 4096;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const frag5_t bf = tr_read8(brow + addrB[ks][0], brow + addrB[ks][1]);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const frag5_t af = tr_read8(arow + kh * (APITCH * 128) + addrA[kw][ks][0],
                                         arow + kh * (APITCH * 128) + addrA[kw][ks][1]);
              acc[kh * 3 + kw] = WG_MFMA(af, bf, acc[kh * 3 + kw]);
            }
        }
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        if (tap > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue_a(tap);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
        for (int tr = 0; tr < 8; ++tr) {
          const char* arow = lds_a + tr * 4096;
          const char* brow = lds_b + tr * 4096;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const frag5_t bf = tr_read8(brow + addrB[ks][0], brow + addrB[ks][1]);
            const frag5_t af = tr_read8(arow + addrA[0][ks][0], arow + addrA[0][ks][1]);
            acc[tap] = WG_MFMA(af, bf, acc[tap]);
          }
        }
      }
    }
  }

  // ---- partial sums -> slab[split][tap][row][col]; C layout: col = lane&31, row = (i&3)+8*(i>>2)+4*(lane>>5)
  const int j = lane & 31, h = lane >> 5;
  float* out = P.slab + (size_t)split * TAPS * P.Ca * P.Cb;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = r0 + wi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      out[((size_t)t * P.Ca + row) * P.Cb + c0 + wj * 32 + j] = acc[t][i];
    }
}

// ================================================================================================================
// conv3x3 wgrad, second generation.  The first-generation loop above reads 10 fragments (1 KiB each) from LDS per
// 9 MFMAs -- 1.1 KiB per MFMA, more than the 128 B/clk LDS port can deliver at the matrix pipe's rate.  Here GEMM-K
// (16 pixels per MFMA) runs DOWN a pixel column instead of along a row: the A fragment of (column c, kernel row kh)
// is then the SAME registers for kw = 0,1,2 (it pairs with the dy fragments of columns c, c-1, c-2), so a column
// step needs 3 A + 1 B fragments for 9 MFMAs (0.44 KiB per MFMA).
//
//   tile     16 rows x 16 columns of output pixels; LDS per buffer: x halo 18 x 19(pitch) pixels + dy 16 x 17(pitch)
//            pixels, 128 B (64 channels) each = 77 KiB; the odd pitches + a 64-byte XOR by bit 1 of the row make the
//            transposed column reads (ds_read_b64_tr_b16, 4 rows x 32 B per 16 lanes) bank-conflict free
//   block    8 waves, ONE workgroup per CU, two LDS buffers filled by LDS-DMA hidden from the compiler (common.h):
//            waves 0-3 take dy columns 0-7, waves 4-7 columns 8-15; inside a group 2 x 2 waves of 32 x 32 channels,
//            all nine taps in 144 accumulator registers for the whole tile range (split-K over the grid)
//   loop     per tile 10 column steps: fragments of step c+1 are fetched while the MFMAs of step c run, one DMA piece
//            of the NEXT tile is issued per step, one barrier per tile
// Partial sums: the two groups are added through LDS, then slab[split][tap][ci][co]; the splits are summed in
// fixed order by wgrad_reduce_kernel.
struct Wgrad3Params {
  const u16* a;      // layer input x
  const u16* b;      // dy
  float* slab;
  int N, H, W, Ca, Cb, lda, ldb;
  int tiles_x, tiles_y, ntiles, nsplit, row_tiles, col_tiles;
  int dbg;            // UNETK_WGRAD_DBG ablation (perf work only): 1 or 16 = every DMA source is the zero page
};

constexpr int W3_AP = 19, W3_BP = 17;
constexpr int W3_A_PIECES = (18 * W3_AP + 7) / 8;   // 43
constexpr int W3_B_PIECES = (16 * W3_BP) / 8;       // 34
constexpr int W3_PIECES = W3_A_PIECES + W3_B_PIECES;
constexpr int W3_BUF = W3_PIECES * 1024;
constexpr int W3_STEPS = 10;

__device__ __forceinline__ int swz3(int row) { return ((row >> 1) & 1) << 2; }

template <typename H>
__global__ __launch_bounds__(512) void wgrad3_bf16_kernel(Wgrad3Params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wi = (wid >> 1) & 1, wj = wid & 1;

  // Workgroup b runs on XCD b % 8 (MI355X_MICROARCH: round-robin dispatch).  All channel-tile pairs of one split read
  // the SAME pixel tiles, so consecutive (split, pair) indices are laid out XCD-major: an XCD's 32 CUs then hold one
  // split's pairs and fetch every pixel tile into their L2 once instead of once per XCD.
  const int npairs = P.row_tiles * P.col_tiles;
  int lin = blockIdx.x;
  if ((gridDim.x & 7) == 0) lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int pair = lin % npairs, split = lin / npairs;
  const int r0 = (pair / P.col_tiles) * 64, c0 = (pair % P.col_tiles) * 64;
  const int t_begin = (int)((long)P.ntiles * split / P.nsplit);
  const int t_end = (int)((long)P.ntiles * (split + 1) / P.nsplit);

  // ---- per-lane transposed-read addresses inside a buffer (column 8*grp; later columns are immediates)
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int sub = (pp & 1) * 8;
  int addrA[3][2], addrB[2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int k = 8 * (g >> 1) + 4 * rd + q4;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int row = kh + k;
      const int piece = wi * 4 + 2 * (g & 1) + (pp >> 1);
      addrA[kh][rd] = (row * W3_AP + grp * 8) * 128 + ((piece ^ swz3(row)) << 4) + sub;
    }
    const int pieceb = wj * 4 + 2 * (g & 1) + (pp >> 1);
    addrB[rd] = W3_A_PIECES * 1024 + (k * W3_BP + grp * 8) * 128 + ((pieceb ^ swz3(k)) << 4) + sub;
  }

  // ---- DMA descriptors, tile invariant: piece fi = q*8 + wid covers 8 pixel slots x 8 sixteen-byte slots
  //      packed: bits 0-7 row+1 rel. to the tile, 8-15 col+1, 16-19 channel piece, 24 valid
  //      (slots 77-79 of the 8 x 10 issue grid repeat piece 76: same bytes to the same place, no branch)
  int desc[W3_STEPS];
  const int lp = lane >> 3, ls = lane & 7;
#pragma unroll
  for (int q = 0; q < W3_STEPS; ++q) {
    const int fi = min(q * 8 + wid, W3_PIECES - 1);
    int d = 0;
    if (fi < W3_A_PIECES) {
      const int p = fi * 8 + lp;
      const int hr = p / W3_AP, hc = p - hr * W3_AP;
      if (hc < 18 && hr < 18) d = hr | (hc << 8) | ((ls ^ swz3(hr)) << 16) | (1 << 24);
    } else {
      const int p = (fi - W3_A_PIECES) * 8 + lp;
      const int hr = p / W3_BP, hc = p - hr * W3_BP;
      if (hc < 16) d = (hr + 1) | ((hc + 1) << 8) | ((ls ^ swz3(hr)) << 16) | (1 << 24);
    }
    desc[q] = d;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const unsigned lds0 = lds_address(smem);
  const int tiles_per_img = P.tiles_x * P.tiles_y;

  // scalars of the tile being staged
  const char* sa = nullptr;
  const char* sb = nullptr;
  int sy0 = 0, sx0 = 0;
  auto stage_tile = [&](int tile) {
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    sy0 = (rem / P.tiles_x) * 16;
    sx0 = (rem % P.tiles_x) * 16;
    sa = reinterpret_cast<const char*>(P.a + (size_t)img * P.H * P.W * P.lda + r0);
    sb = reinterpret_cast<const char*>(P.b + (size_t)img * P.H * P.W * P.ldb + c0);
  };
  // Branch-free on purpose: a branch inside the MFMA loop ends the basic block, and hipcc then waits for EVERY
  // outstanding ds_read (the prefetched fragments of the next step included) before the step's first MFMA.
  auto issue_piece = [&](int q, unsigned buf, bool live) {
    const int fi = min(q * 8 + wid, W3_PIECES - 1);
    const bool is_a = fi < W3_A_PIECES;
    const int d = desc[q];
    const int yy = sy0 + (d & 255) - 1, xx = sx0 + ((d >> 8) & 255) - 1;
    const int ld = is_a ? P.lda : P.ldb;
    const char* base = is_a ? sa : sb;
    const bool ok = live & ((d >> 24) != 0) & ((unsigned)yy < (unsigned)P.H) & ((unsigned)xx < (unsigned)P.W) &
                    !(P.dbg & 16);
    const unsigned off = ((unsigned)(yy * P.W + xx) * (unsigned)ld + ((d >> 16) & 15) * 8u) * 2u;
    const char* src = ok ? base + off : zero;
    glds16_hidden(src, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
  };

  if (t_begin < t_end) {
    stage_tile(t_begin);
#pragma unroll
    for (int q = 0; q < W3_STEPS; ++q) issue_piece(q, lds0, true);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  int cur = 0;

  for (int tile = t_begin; tile < t_end; ++tile) {
    const char* buf = smem + cur * W3_BUF;
    const unsigned nbuf = lds0 + (cur ^ 1) * W3_BUF;
    const bool next = tile + 1 < t_end && !(P.dbg & 1);   // last tile: the pieces still issue, from the zero page
    stage_tile(next ? tile + 1 : tile);

    frag5_t fa[2][3], fb[4];
    auto load_step = [&](int cc) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
        fa[cc & 1][kh] = tr_read8(buf + addrA[kh][0] + cc * 128, buf + addrA[kh][1] + cc * 128);
      if (cc < 8) fb[cc & 3] = tr_read8(buf + addrB[0] + cc * 128, buf + addrB[1] + cc * 128);
    };
    load_step(0);
#pragma unroll
    for (int cc = 0; cc < W3_STEPS; ++cc) {
      if (cc + 1 < W3_STEPS) load_step(cc + 1);
      issue_piece(cc, nbuf, next);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int bl = cc - kw;
        if (bl >= 0 && bl < 8) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
            acc[kh * 3 + kw] = WG_MFMA(fa[cc & 1][kh], fb[bl & 3], acc[kh * 3 + kw]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur ^= 1;
  }
  // ---- the two column groups are summed through LDS (free now: every DMA landed, every read retired at the last
  // barrier); group 0 writes the workgroup's partial sums
  float* xch = reinterpret_cast<float*>(smem);           // [9][16][256] floats = 144 KiB
  const int t256 = tid & 255;
  if (grp == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) xch[(t * 16 + i) * 256 + t256] = acc[t][i];
  }
  __syncthreads();
  if (grp == 0) {
    const int j = lane & 31, h = lane >> 5;
    float* out = P.slab + (size_t)split * 9 * P.Ca * P.Cb;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = r0 + wi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        out[((size_t)t * P.Ca + row) * P.Cb + c0 + wj * 32 + j] = acc[t][i] + xch[(t * 16 + i) * 256 + t256];
      }
  }
}

// ================================================================================================================
// conv3x3 wgrad for c_out % 128 == 0: the same column-K pipeline on a 64 (ci) x 128 (co) channel tile.
//
// The MFMA phase of wgrad3 needs 77 KiB of LDS-DMA per 2.4 us tile = 32.6 GB/s per CU, and the L2 -> LDS fill path
// delivers ~35 GB/s per CU (DMA-only ablation): the kernel is fill-bound (feeding the dy tile from the zero page
// alone lifts it from 1146 to 1513 TFLOP/s).  A 64 x 128 channel tile over 16 x 8 pixels does the same 72 MFMAs per
// wave and tile with 55 KiB of operands (343 instead of 245 FLOP per staged byte):
//   LDS per buffer: x halo 18 x 11(pitch) pixels x 128 B + dy 16 x 8 pixels x 256 B = 57 KiB, two buffers
//   8 waves = 2 (ci halves) x 4 (co quarters), every wave walks all 10 column steps -> no cross-group sum
//   dy rows are 256 B: the 64-byte group of a wave is XORed with (row & 3), so the four pixel rows of a transposed
//   read hit four different bank groups whatever the pitch
constexpr int W4_AP = 11;
constexpr int W4_A_PIECES = (18 * W4_AP + 7) / 8;     // 25
constexpr int W4_B_PIECES = 16 * 8 * 256 / 1024;      // 32
constexpr int W4_PIECES = W4_A_PIECES + W4_B_PIECES;  // 57
constexpr int W4_BUF = W4_PIECES * 1024;
constexpr int W4_STEPS = 10;
constexpr int W4_SLOTS = (W4_PIECES + 7) / 8;         // DMA pieces per wave and tile (8)

template <typename H>
__global__ __launch_bounds__(512) void wgrad4_bf16_kernel(Wgrad3Params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 2, wj = wid & 3;

  const int npairs = P.row_tiles * P.col_tiles;
  int lin = blockIdx.x;                           // XCD-major, see wgrad3
  if ((gridDim.x & 7) == 0) lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int pair = lin % npairs, split = lin / npairs;
  const int r0 = (pair / P.col_tiles) * 64, c0 = (pair % P.col_tiles) * 128;
  const int t_begin = (int)((long)P.ntiles * split / P.nsplit);
  const int t_end = (int)((long)P.ntiles * (split + 1) / P.nsplit);

  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int sub = (pp & 1) * 8;
  int addrA[3][2], addrB[2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int k = 8 * (g >> 1) + 4 * rd + q4;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int row = kh + k;
      const int piece = wi * 4 + 2 * (g & 1) + (pp >> 1);
      addrA[kh][rd] = (row * W4_AP) * 128 + ((piece ^ swz3(row)) << 4) + sub;
    }
    const int pieceb = wj * 4 + 2 * (g & 1) + (pp >> 1);
    addrB[rd] = W4_A_PIECES * 1024 + (k * 8) * 256 + ((pieceb ^ ((k & 3) << 2)) << 4) + sub;
  }

  // DMA descriptors: bits 0-7 row+1 rel. to the tile, 8-15 col+1, 16-20 sixteen-byte piece, 24 valid
  int desc[W4_SLOTS];
#pragma unroll
  for (int q = 0; q < W4_SLOTS; ++q) {
    const int fi = min(q * 8 + wid, W4_PIECES - 1);
    int d = 0;
    if (fi < W4_A_PIECES) {
      const int p = fi * 8 + (lane >> 3), ls = lane & 7;
      const int hr = p / W4_AP, hc = p - hr * W4_AP;
      if (hc < 10 && hr < 18) d = hr | (hc << 8) | ((ls ^ swz3(hr)) << 16) | (1 << 24);
    } else {
      const int p = (fi - W4_A_PIECES) * 4 + (lane >> 4), ls = lane & 15;
      const int hr = p >> 3, hc = p & 7;
      d = (hr + 1) | ((hc + 1) << 8) | ((ls ^ ((hr & 3) << 2)) << 16) | (1 << 24);
    }
    desc[q] = d;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const unsigned lds0 = lds_address(smem);
  const int tiles_per_img = P.tiles_x * P.tiles_y;
  const char* sa = nullptr;
  const char* sb = nullptr;
  int sy0 = 0, sx0 = 0;
  auto stage_tile = [&](int tile) {
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    sy0 = (rem / P.tiles_x) * 16;
    sx0 = (rem % P.tiles_x) * 8;
    sa = reinterpret_cast<const char*>(P.a + (size_t)img * P.H * P.W * P.lda + r0);
    sb = reinterpret_cast<const char*>(P.b + (size_t)img * P.H * P.W * P.ldb + c0);
  };
  auto issue_piece = [&](int q, unsigned buf, bool live) {     // branch-free, see wgrad3
    const int fi = min(q * 8 + wid, W4_PIECES - 1);
    const bool is_a = fi < W4_A_PIECES;
    const int d = desc[q];
    const int yy = sy0 + (d & 255) - 1, xx = sx0 + ((d >> 8) & 255) - 1;
    const int ld = is_a ? P.lda : P.ldb;
    const char* base = is_a ? sa : sb;
    const bool ok = live & ((d >> 24) != 0) & ((unsigned)yy < (unsigned)P.H) & ((unsigned)xx < (unsigned)P.W) &
                    !(P.dbg & 16);
    const unsigned off = ((unsigned)(yy * P.W + xx) * (unsigned)ld + ((d >> 16) & 31) * 8u) * 2u;
    const char* src = ok ? base + off : zero;
    glds16_hidden(src, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
  };

  if (t_begin < t_end) {
    stage_tile(t_begin);
#pragma unroll
    for (int q = 0; q < W4_SLOTS; ++q) issue_piece(q, lds0, true);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  int cur = 0;

  for (int tile = t_begin; tile < t_end; ++tile) {
    const char* buf = smem + cur * W4_BUF;
    const unsigned nbuf = lds0 + (cur ^ 1) * W4_BUF;
    const bool next = tile + 1 < t_end && !(P.dbg & 1);
    stage_tile(next ? tile + 1 : tile);

    frag5_t fa[2][3], fb[4];
    auto load_step = [&](int cc) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
        fa[cc & 1][kh] = tr_read8(buf + addrA[kh][0] + cc * 128, buf + addrA[kh][1] + cc * 128);
      if (cc < 8) fb[cc & 3] = tr_read8(buf + addrB[0] + cc * 256, buf + addrB[1] + cc * 256);
    };
    load_step(0);
#pragma unroll
    for (int cc = 0; cc < W4_STEPS; ++cc) {
      if (cc + 1 < W4_STEPS) load_step(cc + 1);
      if (cc < W4_SLOTS) issue_piece(cc, nbuf, next);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int bl = cc - kw;
        if (bl >= 0 && bl < 8) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
            acc[kh * 3 + kw] = WG_MFMA(fa[cc & 1][kh], fb[bl & 3], acc[kh * 3 + kw]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur ^= 1;
  }

  const int j = lane & 31, h = lane >> 5;
  float* out = P.slab + (size_t)split * 9 * P.Ca * P.Cb;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = r0 + wi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      out[((size_t)t * P.Ca + row) * P.Cb + c0 + wj * 32 + j] = acc[t][i];
    }
}

// ================================================================================================================
// conv3x3 wgrad for c_out % 128 == 0, third generation: v_mfma_f32_16x16x32_bf16.
//
// Same FLOPs, LDS bytes and matrix-pipe cycles per tile as wgrad4, but the chip holds a higher clock on the 16x16x32
// shape (MI355X_MICROARCH.md, DVFS give-back item 7; the conv3x3 fprop/dgrad kernels gained 9-13 % from the same
// change).  K = 32 pixels per instruction runs down a 32-ROW pixel column, so the reuse of wgrad3/4 stays: the A
// fragment of (column c, kernel row kh) serves kw = 0, 1, 2 against the dy fragments of columns c, c-1, c-2.
//
//   tile     32 rows x 5 columns (NCO = 64: 4) of output pixels x 64 ci x 128 co.  Rows are VIRTUAL rows: the images of the batch
//            stacked with one all-zero row between them (H + 1 virtual rows per image).  GEMM-K of a weight gradient
//            is the set of all pixels, so a column may run across an image boundary as long as the row between the
//            images reads as zero in both operands (x: that IS the zero padding both neighbours need; dy: adds
//            nothing) -- 32-row tiles then waste N*(H+1) mod 32 rows per launch instead of H mod 32 rows per image
//            (H = 40: 2.4 % instead of 37 %; the 16-row tiles of wgrad4 wasted 17 % there).
//   LDS      per buffer 62 KiB, two buffers:
//            x halo, 34 rows x 6 columns (pitch 7 pixels), as TWO PLANES of 64 B per pixel: plane h holds ci
//            16h..16h+15 and 32+16h..32+16h+15 (the two ci halves of a wave's 32 x 32 block are then the same lane
//            address + a compile-time offset).  Transposed reads (ds_read_b64_tr_b16: 16 lanes fetch 4 pixel rows x
//            32 B) are served 32 lanes at a time = pixel rows b + {0..3, 8..11}: a row pitch of 7 x 64 B = 192 mod 256
//            spreads four consecutive rows over the four 64-byte groups of a bank row, and the 32-byte half is XORed
//            with row bit 3 -> eight distinct 32-byte bank groups (SQ_LDS_BANK_CONFLICT = 0).
//            dy, 32 x 4 pixels x 256 B: the 32-byte chunk index is XORed with (row & 3, row bit 3).
//   waves    8 = 2 (32 ci) x 4 (32 co); per wave 2 x 2 blocks of 16 x 16 channels x 9 taps = 144 accumulators.
//            Column step: 6 A fragments (kh x ci half) + 2 B fragments (co half) feed up to 36 MFMAs; ONE fragment
//            set, each A fragment re-read for the next column right after its last MFMA; dy columns in a ring of 4.
//   VALU     a 16x16x32 MFMA leaves two VALU slots in its shadow, 144 MFMAs per tile: the loop must stay well under
//            300 VALU instructions per tile.  Fragment addresses are lane registers + immediates (flipped to the other
//            buffer once per tile); DMA offsets advance by a constant from tile to tile along a tile row and are
//            rebuilt only when the row changes (a first version that derived both per use: 295 VALU per tile, 7 % slower
//            than wgrad4).
//   DMA      buffer-descriptor LDS-DMA (out-of-range lanes zero-filled: padding, ragged edges, the row between
//            images), 8 pieces per wave and tile issued in the first four column steps; the tile barrier sits at the
//            head of the last column step (12 MFMAs), whose fragment re-reads already target the next buffer.
struct Wgrad5Params {
  const u16* a;      // layer input x
  const u16* b;      // dy
  float* slab;
  int N, H, W, Ca, Cb, lda, ldb;
  int tiles_x, ntiles, nsplit, row_tiles, col_tiles;
};

constexpr int W5_AP = 7;
constexpr int W5_A_PLANE = 34 * W5_AP * 64;            // 15232 B
constexpr int W5_A_SLOTS = 2 * W5_A_PLANE / 16;        // 1904 sixteen-byte slots
constexpr int W5_A_PIECES = (W5_A_SLOTS + 63) / 64;    // 30
constexpr int W5_B_PIECES = 32 * 4 * 256 / 1024;       // 32
constexpr int W5_PIECES = W5_A_PIECES + W5_B_PIECES;   // 62
constexpr int W5_BUF = W5_PIECES * 1024;
constexpr int W5_SLOTS = (W5_PIECES + 7) / 8;          // DMA pieces per wave and tile (8)

__device__ __forceinline__ int swz5b(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }

__device__ __forceinline__ frag5_t tr_read8u(int addr0, int addr1, int imm) {
  const AS3 char* lds3 = reinterpret_cast<const AS3 char*>(0);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(lds3 + addr0 + imm));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(lds3 + addr1 + imm));
  return __builtin_bit_cast(frag5_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// NCO = 128: as described above.  NCO = 64 (c_out and c_in both not multiples of 128, i.e. the 64 -> 64 layers): the dy
// tile is 32 x 6 pixels x 128 B, stored like the x halo as two planes of 64 B per pixel (pitch 7 pixels; x halo pitch 9);
// waves = 2 (32 ci) x 2 (32 co) x 2 COLUMN GROUPS: group g takes dy columns 3g..3g+2 against halo columns 3g..3g+4 (five
// column steps, 108 MFMAs per wave and tile for 36 fragment KiB; 32 x 4 tiles with two columns per group: 72 MFMAs for
// 28 KiB -- this variant sits on the LDS port, 0.76 -> 0.72 ms at 64 -> 64) and writes its own partial-sum slab -- the
// split reduction adds 2 x nsplit slabs.
// TW: tile width in pixel columns.  The x halo's pitch of 7 pixels (odd: bank spreading) holds 5 + 2 columns as well as
// 4 + 2, so NCO = 128 runs 32 x 5 tiles: the same 30 KiB of halo for 25 % more pixels, 180 MFMAs per wave and barrier
// instead of 144, and every level of the 959 x 640 pyramid (W + 1 = 960, 480, 240, 120, 60) is a whole number of tiles.
template <typename H, int NCO, int TW>
__global__ __launch_bounds__(512) void wgrad5_bf16_kernel(Wgrad5Params P) {
  constexpr bool WIDE = NCO == 128;
  static_assert(WIDE ? (TW == 4 || TW == 5) : (TW == 4 || TW == 6), "tile widths the LDS layouts are made for");
  constexpr int CG = WIDE ? TW : TW / 2;                 // dy columns per wave (NCO = 64: per column group)
  constexpr int NSTEP = CG + 2;                          // halo columns a wave walks per tile
  constexpr int NBC = CG;                                // dy columns a wave holds
  constexpr int NRING = CG == 4 ? 4 : CG;                // ... in a ring of this many fragment pairs (2, 3, 4, 5)
  constexpr int AP = (TW + 2) | 1;                       // x halo pitch in pixels: odd (bank spreading), 7 or 9
  constexpr int A_PLANE = 34 * AP * 64;                  // bytes per ci plane
  constexpr int A_SLOTS = 2 * A_PLANE / 16;
  constexpr int A_PIECES = (A_SLOTS + 63) / 64;          // 30 / 39
  constexpr int BP = TW | 1;                             // NCO = 64: dy pitch in pixels (odd), 5 or 7
  constexpr int B_PLANE = 32 * BP * 64;                  // NCO = 64: bytes per co plane
  constexpr int B_PIECES = WIDE ? 32 * TW * 256 / 1024 : 2 * B_PLANE / 1024;   // 32 (40) / 20 (28)
  constexpr int PIECES = A_PIECES + B_PIECES;            // 62 (70) / 50 (67)
  constexpr int BUF = PIECES * 1024;
  constexpr int SLOTS = (PIECES + 7) / 8;                // 8 (9) / 7 (9)
  // (NCO = 64 with THREE buffers and two tiles of DMA look-ahead ran exactly as fast as with two -- 0.752 vs 0.755 ms at
  // 64 -> 64, 16 x 640 x 959: that variant is bound by the LDS port (28 fragment KiB per 72 MFMAs), not by latency)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wid >> 2, wj = WIDE ? (wid & 3) : ((wid >> 1) & 1), grp = WIDE ? 0 : (wid & 1);

  const int npairs = P.row_tiles * P.col_tiles;
  int lin = blockIdx.x;                           // XCD-major, see wgrad3
  if ((gridDim.x & 7) == 0) lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int pair = lin % npairs, split = lin / npairs;
  const int r0 = (pair / P.col_tiles) * 64, c0 = (pair % P.col_tiles) * NCO;
  const int t_begin = (int)((long)P.ntiles * split / P.nsplit);
  const int t_end = (int)((long)P.ntiles * (split + 1) / P.nsplit);
  const int VH = P.H + 1;                         // virtual rows per image
  const unsigned lds0 = lds_address(smem);

  // ---- per-lane transposed-read addresses (absolute LDS addresses in the CURRENT buffer, column 0; other columns and
  //      the second ci plane are immediates).  Lane (g = l >> 4, q4, pp) of a 16x16x32 operand: k = 8g + 4rd + q4.
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  const int sub = (pp & 1) * 8;
  int addrA[3][2], addrB[2][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int k = 8 * g + 4 * rd + q4;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int row = kh + k;
      addrA[kh][rd] = (int)lds0 + (row * AP + CG * grp) * 64 + (((wi ^ ((row >> 3) & 1)) * 2 + (pp >> 1)) << 4) + sub;
    }
#pragma unroll
    for (int co = 0; co < 2; ++co) {
      if constexpr (WIDE)
        addrB[co][rd] = (int)lds0 + A_PIECES * 1024 + (k * TW) * 256 + (((wj * 4 + 2 * co + (pp >> 1)) ^ swz5b(k)) << 4) + sub;
      else     // co plane = immediate (addrB[1] unused)
        addrB[co][rd] = (int)lds0 + A_PIECES * 1024 + (k * BP + CG * grp) * 64 + (((wj ^ ((k >> 3) & 1)) * 2 + (pp >> 1)) << 4) + sub;
    }
  }

  // ---- DMA: slot q of this wave is piece fi = q * 8 + wid (wave-uniform: x or dy); what a lane of it copies is fixed
  //      for the launch: virtual row hr (rel. to the tile, halo row 0 = tile row - 1), column hc (rel., halo column
  //      0 = tile column - 1) and the logical 16-byte piece of the pixel
  auto slot_geom = [&](int q, int lane, int& hr, int& hc, int& piece) {
    const int fi = min(q * 8 + wid, PIECES - 1);
    if (fi < A_PIECES) {
      const int s = fi * 64 + lane;
      const int plane = s >= A_SLOTS / 2 ? 1 : 0;
      const int sp = s - plane * (A_SLOTS / 2);
      const int px = sp >> 2, sl = sp & 3;
      hr = px / AP;
      hc = px - hr * AP;
      piece = 4 * ((sl >> 1) ^ ((hr >> 3) & 1)) + 2 * plane + (sl & 1);
      if (hc >= TW + 2 || hr >= 34) hr = 255;                // pitch padding / beyond the planes: never valid
    } else if constexpr (WIDE) {
      const int p = (fi - A_PIECES) * 4 + (lane >> 4), ls = lane & 15;
      const int r = p / TW;
      hr = r + 1;
      hc = p - r * TW + 1;
      piece = ls ^ swz5b(r);
    } else {
      const int s = (fi - A_PIECES) * 64 + lane;
      const int plane = s >= B_PLANE / 16 ? 1 : 0;
      const int sp = s - plane * (B_PLANE / 16);
      const int px = sp >> 2, sl = sp & 3;
      const int r = px / BP, c = px - r * BP;
      hr = c < TW ? r + 1 : 255;                             // pitch padding: never valid
      hc = c + 1;
      piece = 4 * ((sl >> 1) ^ ((r >> 3) & 1)) + 2 * plane + (sl & 1);
    }
  };
  unsigned hc_pack = 0;                                        // 3 bits per slot: hc (0..6)
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    int hr, hc, piece;
    slot_geom(q, lane, hr, hc, piece);
    hc_pack |= (unsigned)(hc & 7) << (3 * q);
  }

  f32x4 acc[9][2][2];                              // [tap][ci half][co half]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- the tile being STAGED: column block s_tx, virtual row s_yb of image s_img (a tile's 34 halo rows reach into
  //      at most one more image: the host sends H + 1 >= 34 here).  Per tile ROW: descriptors (base = image s_img +
  //      the workgroup's channel offset, range = what is left of the tensor from there, at most two images) and the
  //      lane offsets of column block 0; an invalid row is the offset 2^31, out of every range the host admits.
  //      The eight offsets of a lane live in LDS behind the buffers ([slot][thread]; registers are all taken): a
  //      lane reads back what it wrote itself, one slot ahead of its use.
  int s_tx, s_yb, s_img;
  const int vo_addr = (int)lds0 + 2 * BUF + tid * 4;
  const AS3 char* lds3 = reinterpret_cast<const AS3 char*>(0);
  auto vo_load = [&](int q) { return *reinterpret_cast<const AS3 unsigned*>(lds3 + vo_addr + q * 2048); };
  unetk_u32x4 dsc_a = {0u, 0u, 0u, 0x00020000u}, dsc_b = {0u, 0u, 0u, 0x00020000u};
  const unsigned img_a = (unsigned)(P.H * P.W * P.lda) * 2u, img_b = (unsigned)(P.H * P.W * P.ldb) * 2u;   // bytes
  auto setup_row = [&]() {
    const int left = min(P.N - s_img, 2);
    const unsigned long long ba = (unsigned long long)(P.a + (size_t)min(s_img, P.N - 1) * P.H * P.W * P.lda + r0);
    const unsigned long long bb = (unsigned long long)(P.b + (size_t)min(s_img, P.N - 1) * P.H * P.W * P.ldb + c0);
    dsc_a[0] = (unsigned)ba; dsc_a[1] = (unsigned)(ba >> 32) & 0xffffu;
    dsc_b[0] = (unsigned)bb; dsc_b[1] = (unsigned)(bb >> 32) & 0xffffu;
    dsc_a[2] = left > 0 ? (unsigned)left * img_a - (unsigned)r0 * 2u : 0u;
    dsc_b[2] = left > 0 ? (unsigned)left * img_b - (unsigned)c0 * 2u : 0u;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));      // opaque: the geometry is recomputed here, not kept in 24 registers
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      int hr, hc, piece;
      slot_geom(q, lane_o, hr, hc, piece);
      const bool is_a = min(q * 8 + wid, PIECES - 1) < A_PIECES;
      int yl = s_yb + hr - 1;
      const bool wrap = yl >= VH && hr != 255;
      yl -= wrap ? VH : 0;
      const bool ok = (hr != 255) & (yl >= 0) & (yl < P.H) & (s_img + (wrap ? 1 : 0) < P.N);
      const unsigned ld2 = (unsigned)(is_a ? P.lda : P.ldb) * 2u;
      const unsigned off = (unsigned)(yl * P.W + hc - 1) * ld2 + (wrap ? (is_a ? img_a : img_b) : 0u) + piece * 16u;
      *reinterpret_cast<AS3 unsigned*>(const_cast<AS3 char*>(lds3) + vo_addr + q * 2048) = ok ? off : 0x80000000u;
    }
  };
  {
    const int tyv = t_begin / P.tiles_x;
    s_tx = t_begin - tyv * P.tiles_x;
    s_img = (tyv * 32) / VH;
    s_yb = tyv * 32 - s_img * VH;
    setup_row();
  }
  auto advance_stage = [&]() {
    if (++s_tx == P.tiles_x) {
      s_tx = 0;
      s_yb += 32;
      if (s_yb >= VH) { s_yb -= VH; ++s_img; }
      setup_row();
    }
  };
  auto issue_piece = [&](int q, unsigned voff_q, unsigned buf, bool live) {     // branch-free, 5 VALU
    const int fi = min(q * 8 + wid, PIECES - 1);
    const bool is_a = fi < A_PIECES;                            // wave-uniform
    unsigned hp = hc_pack;
    asm volatile("" : "+v"(hp));     // opaque: otherwise the eight fields are extracted once, kept and spilled
    const int xx = (int)((hp >> (3 * q)) & 7u) - 1 + TW * s_tx;
    const unsigned step = (unsigned)(TW * s_tx) * ((unsigned)(is_a ? P.lda : P.ldb) * 2u);   // scalar
    const unsigned off = ((unsigned)xx < (unsigned)P.W) ? voff_q + step : 0x80000000u;
    unetk_u32x4 d = is_a ? dsc_a : dsc_b;
    d[2] = live ? d[2] : 0u;
    bufdma16(off, d, 0u, buf + fi * 1024);
  };

  if (t_begin < t_end) {
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) issue_piece(q, vo_load(q), lds0, true);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  int cur = 0;

  frag5_t fa[3][2], fb[NRING][2];
  auto read_a = [&](int kh, int ci, int cc) {                  // A fragment (kh, ci half) of halo column cc
    fa[kh][ci] = tr_read8u(addrA[kh][0], addrA[kh][1], ci * A_PLANE + cc * 64);
  };
  auto read_b = [&](int co, int c) {                           // dy fragment (co half) of column c
    if constexpr (WIDE) fb[c % NRING][co] = tr_read8u(addrB[co][0], addrB[co][1], c * 256);
    else fb[c % NRING][co] = tr_read8u(addrB[0][0], addrB[0][1], co * B_PLANE + c * 64);
  };
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) read_a(kh, ci, 0);
  read_b(0, 0);
  read_b(1, 0);

  for (int tile = t_begin; tile < t_end; ++tile) {
    const unsigned nbuf = lds0 + (cur ^ 1) * BUF;
    const int flip = cur ? -BUF : BUF;
    const bool next = tile + 1 < t_end;
    advance_stage();                                           // s_* now name tile + 1 (its pieces go out below)
    unsigned vo = vo_load(0);                                  // offset of the next piece to go out
#pragma unroll
    for (int cc = 0; cc < NSTEP; ++cc) {
      const bool last = cc == NSTEP - 1;
      if (last) {
        // next tile landed (this wave's DMAs), and this wave holds every fragment it still needs of the current buffer:
        // from here on every read targets the next buffer
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) addrA[kh][rd] += flip;
          addrB[0][rd] += flip;
          if constexpr (WIDE) addrB[1][rd] += flip;
        }
      }
      // dy column for the NEXT step goes into the ring slot that step cc - 3 used last
      if (cc + 1 < NBC) {
        read_b(0, cc + 1);
        read_b(1, cc + 1);
      } else if (last) {
        read_b(0, 0);
        read_b(1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int bl = cc - kw;
            if (bl >= 0 && bl < NBC) {
#pragma unroll
              for (int co = 0; co < 2; ++co)
                acc[kh * 3 + kw][ci][co] = H16<H>::mfma16(__builtin_bit_cast(typename H16<H>::x8, fa[kh][ci]),
                                                          __builtin_bit_cast(typename H16<H>::x8, fb[bl % NRING][co]),
                                                          acc[kh * 3 + kw][ci][co]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          // this fragment's last use is behind us: fetch it for the next column (of the next tile after the last)
          read_a(kh, ci, last ? 0 : cc + 1);
          // DMA of the next tile: behind the first two (NCO = 64: three) fragment groups of the column steps before
          // the last two
          constexpr int PER = WIDE ? 2 : 3;
          const int j = kh * 2 + ci, slot = PER * cc + j;
          if (j < PER && cc < NSTEP - 1 && slot < SLOTS) {
            issue_piece(slot, vo, nbuf, next);
            if (slot + 1 < SLOTS) vo = vo_load(slot + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    cur ^= 1;
  }

  const int col = lane & 15, rg = lane >> 4;
  float* out = P.slab + (size_t)(WIDE ? split : 2 * split + grp) * 9 * P.Ca * P.Cb;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // plane h of the x tile holds ci 16h.. and 32 + 16h..: wave half wi, plane ci -> channels 32 wi + 16 ci
          const int row = r0 + wi * 32 + ci * 16 + 4 * rg + i;
          out[((size_t)t * P.Ca + row) * P.Cb + c0 + wj * 32 + co * 16 + col] = acc[t][ci][co][i];
        }
}

int pick_nsplit3(int ntiles, int npairs) {
  int ns = 256 / npairs;
  if (ns < 1) ns = 1;
  if (ns > ntiles) ns = ntiles;
  return ns;
}

// ================================================================================================================
// ConvTranspose2d k2 s2 wgrad, second generation (c_in % 128 == 0: every level of the UNet).
//
//   dw[ci][co][tap] = sum over low-res pixels p   dy[up(p, tap)][co] * x[p][ci]        (four independent GEMMs,
//   rows = co, cols = ci, GEMM-K = pixels; both operands pixel-major -> transposed LDS reads as above)
//
// The first-generation kernel stages a tile, waits for it, computes, and stages the next (0.18 matrix-pipe busy).  Here
//   tile     2 rows x 32 columns of low-res pixels (64 = four 16-pixel k-steps) x 64 co x 128 ci x 4 taps
//   LDS      per buffer six [64 px][128 B] arrays -- x channels 0-63, x 64-127, dy tap 0..3 -- = 48 KiB, THREE buffers:
//            the DMA of tile i+2 is issued while tile i is consumed (a 64-pixel tile is ~0.5 us of MFMA work, less than
//            one HBM round trip, so one tile of look-ahead would leave every wave waiting on vmcnt)
//   waves    8 = 4 taps x 2 k-groups: wave (tap, kg) takes k-steps 2kg, 2kg+1 of its tap: 2 dy fragments (co blocks) and
//            4 x fragments (ci blocks) per k-step feed 8 MFMAs -- 0.75 KiB of LDS reads per MFMA (one block per wave
//            and tap would need 1.25); the two k-groups are added through LDS at the end
//   DMA      wave w owns pixels 8w..8w+7 of every array: six 1-KiB pieces per tile, lane offsets fixed for the launch
//            (only the tile base and the ragged-edge mask change), one piece between MFMAs
// Partial sums: slab[split][tap][co][ci], summed in fixed order by the reduce kernels below.
struct WgradTParams {
  const u16* dy;     // up-sampled gradient (channel offset applied)
  const u16* x;      // low-res input
  float* slab;
  int N, H, W;        // low-res grid
  int Co, Ci, lddy, ldx;
  int Hs, Ws, padT, padL;
  int tiles_x, tiles_y, ntiles, nsplit, row_tiles, col_tiles;
};

constexpr int WT_ARR = 64 * 128;          // one array
constexpr int WT_BUF = 6 * WT_ARR;
constexpr int WT_NBUF = 3;

template <typename H>
__global__ __launch_bounds__(512) void wgradT_bf16_kernel(WgradTParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tap = wid & 3, kg = wid >> 2;

  const int npairs = P.row_tiles * P.col_tiles;
  int lin = blockIdx.x;                           // XCD-major, see wgrad3
  if ((gridDim.x & 7) == 0) lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int pair = lin % npairs, split = lin / npairs;
  const int r0 = (pair / P.col_tiles) * 64, c0 = (pair % P.col_tiles) * 128;
  const int t_begin = (int)((long)P.ntiles * split / P.nsplit);
  const int t_end = (int)((long)P.ntiles * (split + 1) / P.nsplit);

  // transposed-read addresses inside one array: [32-channel block][ds_read of the pair]; a k-step adds 16 pixel rows
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
  int addr[2][2];
#pragma unroll
  for (int cblk = 0; cblk < 2; ++cblk)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int pb = 8 * (g >> 1) + 4 * rd + q4;
      const int piece = cblk * 4 + 2 * (g & 1) + (pp >> 1);
      addr[cblk][rd] = pb * 128 + ((piece ^ swz(pb)) << 4) + (pp & 1) * 8;
    }

  // DMA: this lane's 16 bytes of the wave's pixel rows, relative to the tile's first pixel
  const int dp = 8 * wid + (lane >> 3);                       // pixel of the tile
  const int dr = dp >> 5, dc = dp & 31;
  const int dpiece = (lane & 7) ^ swz(dp);
  const unsigned offx = ((unsigned)(dr * P.W + dc) * (unsigned)P.ldx + dpiece * 8u) * 2u;
  const unsigned offd = ((unsigned)(2 * dr * P.Ws + 2 * dc) * (unsigned)P.lddy + dpiece * 8u) * 2u;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const unsigned lds0 = lds_address(smem);
  const int tiles_per_img = P.tiles_x * P.tiles_y;
  const char* sx = nullptr;
  const char* sd = nullptr;
  bool sok = false;
  // the tile being staged, advanced incrementally (a tile is 16 MFMAs per wave: two integer divisions per tile were
  // most of the 126 scalar instructions the first version spent per tile)
  int s_img, s_ty, s_tx;
  {
    const int first = min(t_begin, P.ntiles - 1);
    s_img = first / tiles_per_img;
    const int rem = first - s_img * tiles_per_img;
    s_ty = rem / P.tiles_x;
    s_tx = rem - s_ty * P.tiles_x;
  }
  auto stage_tile = [&](bool live) {                          // stages (s_img, s_ty, s_tx), then steps to the next tile
    const int y0 = s_ty * 2, x0 = s_tx * 32;
    sx = reinterpret_cast<const char*>(P.x + ((size_t)(s_img * P.H + y0) * P.W + x0) * P.ldx + c0);
    sd = reinterpret_cast<const char*>(P.dy + ((size_t)(s_img * P.Hs + 2 * y0 + P.padT) * P.Ws + 2 * x0 + P.padL) * P.lddy + r0);
    sok = live && (y0 + dr < P.H) && (x0 + dc < P.W);
    if (live) {
      if (++s_tx == P.tiles_x) {
        s_tx = 0;
        if (++s_ty == P.tiles_y) { s_ty = 0; ++s_img; }
      }
    }
  };
  auto issue_piece = [&](int q, unsigned buf) {               // branch-free; q: array
    const char* src;
    if (q < 2) src = sx + (offx + q * 128u);
    else src = sd + (offd + (unsigned)(((q - 2) >> 1) * P.Ws + ((q - 2) & 1)) * (unsigned)P.lddy * 2u);
    src = sok ? src : zero;
    glds16_hidden(src, __builtin_amdgcn_readfirstlane(buf + q * WT_ARR + wid * 1024));
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

  // Three tiles of look-ahead in three buffers: the fragments of tile i+1 are read into a second register set while the
  // MFMAs of tile i run, so buffer i % 3 is free as soon as every wave has passed the barrier of iteration i and takes
  // the DMA of tile i+3.  Every stage issues exactly six pieces per wave (zero page past the end), so `vmcnt(6)` always
  // means "everything but the youngest stage has landed".
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    stage_tile(t_begin + k < t_end);
#pragma unroll
    for (int q = 0; q < 6; ++q) issue_piece(q, lds0 + k * WT_BUF);
  }
  frag5_t fa[2][2][2], fb[2][2][4];                            // [register set][k-step][block]
  auto read_frags = [&](int bufi, frag5_t (&a)[2][2], frag5_t (&b)[2][4]) {
    const char* buf = smem + bufi * WT_BUF + (2 * kg) * 2048;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        a[ks][rb] = tr_read8(buf + ks * 2048 + (2 + tap) * WT_ARR + addr[rb][0], buf + ks * 2048 + (2 + tap) * WT_ARR + addr[rb][1]);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        b[ks][cb] = tr_read8(buf + ks * 2048 + (cb >> 1) * WT_ARR + addr[cb & 1][0],
                             buf + ks * 2048 + (cb >> 1) * WT_ARR + addr[cb & 1][1]);
    }
  };
  asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
  read_frags(0, fa[0], fb[0]);
  int cur = 0;                                                // buffer of the tile whose MFMAs run
  auto run_tile = [&](auto set_c, int tile) {
    constexpr int SET = decltype(set_c)::value;
    // tile+1 landed (own pieces; the barrier makes that all waves'), own fragments of `tile` arrived, and every wave has
    // finished reading buffer `cur`
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int nxt = cur == 2 ? 0 : cur + 1;
    read_frags(nxt, fa[SET ^ 1], fb[SET ^ 1]);
    stage_tile(tile + 3 < t_end);
    const unsigned nbuf = lds0 + cur * WT_BUF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        if (ks * 2 + rb < 3) {                                // six pieces over the tile's 16 MFMAs
          issue_piece(2 * (ks * 2 + rb), nbuf);
          issue_piece(2 * (ks * 2 + rb) + 1, nbuf);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = WG_MFMA(fa[SET][ks][rb], fb[SET][ks][cb], acc[rb][cb]);
      }
    cur = nxt;
  };
  for (int tile = t_begin; tile < t_end; tile += 2) {
    run_tile(std::integral_constant<int, 0>{}, tile);
    if (tile + 1 < t_end) run_tile(std::integral_constant<int, 1>{}, tile + 1);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- k-group 1 hands its partial sums to k-group 0 through LDS (4 taps x 8 blocks x 16 x 64 floats = 128 KiB)
  float* xch = reinterpret_cast<float*>(smem);
  if (kg == 1) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) xch[(((tap * 2 + rb) * 4 + cb) * 16 + i) * 64 + lane] = acc[rb][cb][i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (kg == 0) {
    const int j = lane & 31, h = lane >> 5;
    float* out = P.slab + ((size_t)split * 4 + tap) * P.Co * P.Ci;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = r0 + rb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          out[(size_t)row * P.Ci + c0 + cb * 32 + j] =
              acc[rb][cb][i] + xch[(((tap * 2 + rb) * 4 + cb) * 16 + i) * 64 + lane];
        }
  }
}

// out[col][row][tap] = sum_s slab[s][tap][row][col]   (out = PyTorch's [Co][Ci][3][3] resp. [Ci][Co][2][2]: s_col =
// rows*taps, s_row = taps).  One workgroup = 32 cols x 8 rows x all taps, transposed through LDS: the slab is read in
// 128-byte runs (32 cols), the parameter gradient written in runs of 8*taps floats.  Fixed summation order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                           int nsplit, int taps, int rows, int cols, long s_row,
                                                           long s_col, int accumulate, int tap_flip) {
  __shared__ float tile[9 * 8 * 33];
  const int col_tiles = cols / 32;
  const int c0 = (blockIdx.x % col_tiles) * 32, r0 = (blockIdx.x / col_tiles) * 8;
  const long total = (long)taps * rows * cols;
#if UNETK_WGRAD_REDUCE_V4
  // 16-byte slab loads: a thread owns four adjacent columns of one row, thread groups of 64 take the taps in turn
  // (four independent load streams per workgroup instead of one 4-byte load per lane and split)
  const int c4 = (threadIdx.x & 7) * 4, rl = (threadIdx.x >> 3) & 7, tg = threadIdx.x >> 6;
  for (int t = tg; t < taps; t += 4) {
    const long idx = ((long)t * rows + r0 + rl) * cols + c0 + c4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(slab + (size_t)k * total + idx);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(t * 8 + rl) * 33 + c4 + e] = s[e];
  }
#else
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  for (int t = 0; t < taps; ++t) {
    const long idx = ((long)t * rows + r0 + rl) * cols + c0 + cl;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab[(size_t)k * total + idx];
    tile[(t * 8 + rl) * 33 + cl] = s;
  }
#endif
  __syncthreads();
  const int per_col = 8 * taps;
  for (int e = threadIdx.x; e < 32 * per_col; e += 256) {
    const int c = e / per_col, rem = e - c * per_col;
    const int r = rem / taps, t = rem - r * taps;
    float* o = out + (long)(c0 + c) * s_col + (long)(r0 + r) * s_row + (tap_flip ? taps - 1 - t : t);
    const float v = tile[(t * 8 + r) * 33 + c];
    *o = accumulate ? *o + v : v;
  }
}

// The same sum, one output element per thread (coalesced slab reads, scattered writes): many splits over a small
// weight tensor (the 64..128-channel layers), where the transposing kernel above would have too few workgroups.
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                                int nsplit, int taps, int rows, int cols, long s_row,
                                                                long s_col, int accumulate, int tap_flip) {
  // 64 outputs x 4 split groups per block, four independent partial sums per thread, groups combined in a fixed
  // order through LDS: 16x the loads in flight of a thread-per-output loop (which ran at 1 TB/s on 40-75 MB slabs)
  __shared__ float part[4][64];
  const long total = (long)taps * rows * cols;
  const int l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
    const long idx = base + l;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < total) {
      const float* p = slab + idx;
      int k = grp;
      for (; k + 12 < nsplit; k += 16) {
        s0 += p[(size_t)k * total];
        s1 += p[(size_t)(k + 4) * total];
        s2 += p[(size_t)(k + 8) * total];
        s3 += p[(size_t)(k + 12) * total];
      }
      for (; k < nsplit; k += 4) s0 += p[(size_t)k * total];
    }
    part[grp][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && idx < total) {
      const float s = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
      const int col = (int)(idx % cols);
      const long rem = idx / cols;
      const int row = (int)(rem % rows);
      const int tap = (int)(rem / rows);
      float* o = out + row * s_row + col * s_col + (tap_flip ? taps - 1 - tap : tap);
      *o = accumulate ? *o + s : s;
    }
    __syncthreads();
  }
}

// swapped: the slab holds [tap'][co][ci] with tap' = 8 - tap (the kernel ran with dy as the shifted operand, see
// unetk_conv3x3_wgrad): rows = co, cols = ci, taps mirrored
void launch_wgrad_reduce(const float* slab, float* out, int nsplit, int taps, int rows, int cols, int accumulate,
                         hipStream_t stream, bool swapped = false) {
  const long s_row = swapped ? (long)cols * taps : taps, s_col = swapped ? taps : (long)rows * taps;
  const int flip = swapped ? 1 : 0;
  if (nsplit <= 16) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((cols / 32) * (rows / 8))), dim3(256), 0, stream, slab, out,
                       nsplit, taps, rows, cols, s_row, s_col, accumulate, flip);
  } else {
    const long total = (long)taps * rows * cols;
    hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, slab, out,
                       nsplit, taps, rows, cols, s_row, s_col, accumulate, flip);
  }
}

// ------------------------------------------------------------------------------------------ fp32 variant
// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: A[i][k] one float per lane, k = lane>>5), no transpose needed: a lane
// reads channel (lane&31) of pixel 2s + (lane>>5) with ds_read_b32 -- 32 consecutive floats per half-wave, conflict
// free without a swizzle.  Tile = 32 x 32 channels (128-byte pixel rows, same LDS image sizes as bf16); the four
// waves split the tile's pixel rows (2 each), keep all taps in registers and are summed through LDS at the end.
struct WgradParamsF32 {
  const float* a;
  const float* b;
  float* slab;
  int N, H, W, Ca, Cb, lda, ldb, Ha, Wa, padT, padL;
  int tiles_x, tiles_y, ntiles, nsplit, row_tiles, col_tiles;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(WgradParamsF32 P) {
  constexpr int TAPS = MODE == 0 ? 9 : 4;
  constexpr int APITCH = MODE == 0 ? 36 : 32;
  constexpr int AROWS = MODE == 0 ? 10 : 8;
  constexpr int A_BYTES = 10 * 36 * 128;
  constexpr int A_INSTR = AROWS * APITCH * 8 / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_b = smem + A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int npairs = P.row_tiles * P.col_tiles;
  const int pair = blockIdx.x % npairs, split = blockIdx.x / npairs;
  const int r0 = (pair / P.col_tiles) * 32, c0 = (pair % P.col_tiles) * 32;
  const int t_begin = (int)((long)P.ntiles * split / P.nsplit);
  const int t_end = (int)((long)P.ntiles * (split + 1) / P.nsplit);

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const int lp = lane >> 3, ls = lane & 7;

  for (int tile = t_begin; tile < t_end; ++tile) {
    int tt = tile;
    const int tx = tt % P.tiles_x; tt /= P.tiles_x;
    const int ty = tt % P.tiles_y;
    const int img = tt / P.tiles_y;
    const int y0 = ty * 8, x0 = tx * 32;
    const float* bimg = P.b + (size_t)img * P.H * P.W * P.ldb + c0;
    const float* aimg = P.a + (size_t)img * (MODE == 0 ? P.H * P.W : P.Ha * P.Wa) * P.lda + r0;
    auto issue_b = [&]() {
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int fi = f * 4 + wid;
        const int p = fi * 8 + lp;
        const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
        const char* src = zero;
        if (yy < P.H && xx < P.W) src = reinterpret_cast<const char*>(bimg + (size_t)(yy * P.W + xx) * P.ldb + ls * 4);
        glds16(src, lds_b + fi * 1024);
      }
    };
    auto issue_a = [&](int tap) {
#pragma unroll
      for (int f = 0; f < (A_INSTR + 3) / 4; ++f) {
        const int fi = f * 4 + wid;
        if (fi < A_INSTR) {
          const int p = fi * 8 + lp;
          const char* src = zero;
          if constexpr (MODE == 0) {
            const int hr = p / APITCH, hc = p - hr * APITCH;
            const int yy = y0 + hr - 1, xx = x0 + hc - 1;
            if (hc < 34 && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W)
              src = reinterpret_cast<const char*>(aimg + (size_t)(yy * P.W + xx) * P.lda + ls * 4);
          } else {
            const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
            if (yy < P.H && xx < P.W)
              src = reinterpret_cast<const char*>(
                  aimg + (size_t)((2 * yy + (tap >> 1) + P.padT) * P.Wa + 2 * xx + (tap & 1) + P.padL) * P.lda + ls * 4);
          }
          glds16(src, lds_a + fi * 1024);
        }
      }
    };
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue_b();
    if constexpr (MODE == 0) {
      issue_a(0);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = 2 * wid + rr;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
          const int px = 2 * s + h;
          const float bv = *reinterpret_cast<const float*>(lds_b + (row * 32 + px) * 128 + r * 4);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const float av = *reinterpret_cast<const float*>(lds_a + ((row + t / 3) * APITCH + px + t % 3) * 128 + r * 4);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        if (tap > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue_a(tap);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int row = 2 * wid + rr;
#pragma unroll 4
          for (int s = 0; s < 16; ++s) {
            const int px = 2 * s + h;
            const float bv = *reinterpret_cast<const float*>(lds_b + (row * 32 + px) * 128 + r * 4);
            const float av = *reinterpret_cast<const float*>(lds_a + (row * 32 + px) * 128 + r * 4);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- sum the four waves through LDS, then one coalesced slab write
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float* red = reinterpret_cast<float*>(smem);      // [TAPS][32][32]
  for (int i = tid; i < TAPS * 1024; i += 256) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) atomicAdd(&red[t * 1024 + ((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r], acc[t][i]);
  __syncthreads();
  float* out = P.slab + (size_t)split * TAPS * P.Ca * P.Cb;
  for (int i = tid; i < TAPS * 1024; i += 256) {
    const int t = i >> 10, row = (i >> 5) & 31, col = i & 31;
    out[((size_t)t * P.Ca + r0 + row) * P.Cb + c0 + col] = red[i];
  }
}

int pick_nsplit(int ntiles, int npairs) {
  int ns = 512 / npairs;
  if (ns < 1) ns = 1;
  if (ns > ntiles) ns = ntiles;
  return ns;
}

}  // namespace

extern "C" {

int64_t unetk_wgrad_workspace_bytes(int kind, int n, int h, int w, int c_in, int c_out) {
  if (n <= 0 || h <= 0 || w <= 0 || c_in % 64 || c_out % 64 || c_in <= 0 || c_out <= 0) return UNETK_ERR_SHAPE;
  const int ntiles = n * host_cdiv(h, 8) * host_cdiv(w, 32);
  const int npairs = (c_in / 64) * (c_out / 64);
  const int taps = kind == 0 ? 9 : 4;
  // fp32 uses 32x32 channel tiles (4x the pairs, hence fewer splits than the 64x64 bf16 kernels): size for bf16
  int64_t slabs = pick_nsplit(ntiles, npairs);
  if (kind == 0) {   // bf16 conv3x3: 16x16-pixel tiles
    const int64_t s3 = pick_nsplit3(n * host_cdiv(h, 16) * host_cdiv(w, 16), npairs);
    if (s3 > slabs) slabs = s3;
    if (c_out % 128 == 0) {   // 64 x 128 channel tiles over 16 x 8 pixels
      const int64_t s4 = pick_nsplit3(n * host_cdiv(h, 16) * host_cdiv(w, 8), (c_in / 64) * (c_out / 128));
      if (s4 > slabs) slabs = s4;
      // ... or (wgrad5) over 32 x 4 pixels of the vertically stacked batch
      const int64_t s5 = pick_nsplit3(host_cdiv((long)n * (h + 1), 32) * host_cdiv(w, 4), (c_in / 64) * (c_out / 128));
      if (s5 > slabs) slabs = s5;
    } else if (c_in % 128 != 0) {   // wgrad5<64>: two column groups, each with its own slab per split
      const int64_t s5 = 2 * (int64_t)pick_nsplit3(host_cdiv((long)n * (h + 1), 32) * host_cdiv(w, 4), (c_in / 64) * (c_out / 64));
      if (s5 > slabs) slabs = s5;
    } else {                        // wgrad5 with the operands' roles swapped
      const int64_t s5 = pick_nsplit3(host_cdiv((long)n * (h + 1), 32) * host_cdiv(w, 4), (c_out / 64) * (c_in / 128));
      if (s5 > slabs) slabs = s5;
    }
  }
  if (kind == 1 && c_in % 128 == 0) {   // second-generation kernel: 2 x 32-pixel tiles, 64 x 128 channel pairs
    const int64_t st = pick_nsplit3(n * host_cdiv(h, 2) * host_cdiv(w, 32), (c_in / 128) * (c_out / 64));
    if (st > slabs) slabs = st;
  }
  return slabs * taps * c_in * c_out * 4;
}

int unetk_conv3x3_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, void* workspace,
                        int64_t workspace_bytes, int n, int h, int w, int c_in, int c_out, int accumulate, int dtype,
                        void* stream) {
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  const int es = unetk_esize(dtype);
  const int64_t need = unetk_wgrad_workspace_bytes(0, n, h, w, c_in, c_out);
  if (need < 0) return (int)need;
  if (!x || !dy || !dw || !workspace || workspace_bytes < need) return UNETK_ERR_ARG;
  if (ldx < c_in || lddy < c_out || (ldx * es) % 16 || (lddy * es) % 16) return UNETK_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) return UNETK_ERR_ALIGN;
  if (dtype == UNETK_F32) {
    WgradParamsF32 Q{};
    Q.a = (const float*)x; Q.b = (const float*)dy; Q.slab = (float*)workspace;
    Q.N = n; Q.H = h; Q.W = w; Q.Ca = c_in; Q.Cb = c_out; Q.lda = ldx; Q.ldb = lddy;
    Q.tiles_x = host_cdiv(w, 32); Q.tiles_y = host_cdiv(h, 8); Q.ntiles = n * Q.tiles_x * Q.tiles_y;
    Q.row_tiles = c_in / 32; Q.col_tiles = c_out / 32;
    Q.nsplit = pick_nsplit(Q.ntiles, Q.row_tiles * Q.col_tiles);
    const size_t ldsq = 10 * 36 * 128 + 256 * 128;
    UNETK_ALLOW_LDS((wgrad_f32_kernel<0>), ldsq);
    hipLaunchKernelGGL(wgrad_f32_kernel<0>, dim3(Q.row_tiles * Q.col_tiles * Q.nsplit), dim3(256), ldsq,
                       (hipStream_t)stream, Q);
    UNETK_CHECK_LAUNCH();
    const long totalq = 9L * c_in * c_out;
    launch_wgrad_reduce((const float*)workspace, dw, Q.nsplit, 9, c_in, c_out, accumulate, (hipStream_t)stream);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  if ((int64_t)h * w * (ldx > lddy ? ldx : lddy) * 2 >= (int64_t)1 << 32) return UNETK_ERR_SHAPE;   // 32-bit DMA offsets
  UNETK_KNOB(wide, "UNETK_WGRAD_WIDE", 1);   // 0: A/B against 64x64 tiles
  UNETK_KNOB(gen5, "UNETK_WGRAD_K32", 1);    // 0: A/B against wgrad4 (32x32x16)
  // wgrad5 (16x16x32, 32-row columns over the vertically stacked batch): a tile's 34 halo rows may reach into ONE more
  // image, and its DMA offsets span two images
  // Its wide (128-channel) operand is the UNSHIFTED one.  c_out % 128 == 0: x shifted, dy fixed.  Otherwise, if
  // c_in % 128 == 0 (the decoder's 128 -> 64 layer): the same sum re-indexed by the input pixel p' = p + tap,
  //   dw[tap][ci][co] = sum_p' dy[p' - tap][co] * x[p'][ci]      (dy is zero outside the image, as the padding of x was)
  // i.e. the kernel with dy as the shifted 64-channel operand, x as the fixed one, and mirrored taps (`swapped`).
  const bool w5_direct = c_out % 128 == 0, w5_swapped = !w5_direct && c_in % 128 == 0;
  UNETK_KNOB(gen5n, "UNETK_WGRAD_K32N", 1);  // 0: A/B of the 64 x 64 variant against wgrad3
  if (!w5_direct && !w5_swapped && wide && gen5 && gen5n && h + 1 >= 34 &&
      (int64_t)h * w * (ldx > lddy ? ldx : lddy) * 4 < (int64_t)1 << 31) {
    // neither operand is 128 wide (64 -> 64): 64 x 64 channel tiles, two column groups -> 2 x nsplit partial slabs
    Wgrad5Params P{};
    P.slab = (float*)workspace;
    P.N = n; P.H = h; P.W = w;
    P.a = (const u16*)x; P.b = (const u16*)dy; P.Ca = c_in; P.Cb = c_out; P.lda = ldx; P.ldb = lddy;
    UNETK_KNOB(tw6, "UNETK_WGRAD_TW6", 1);     // 0: A/B against 32 x 4 tiles (two columns per group)
    P.tiles_x = host_cdiv(w, tw6 ? 6 : 4);
    P.ntiles = host_cdiv((long)n * (h + 1), 32) * P.tiles_x;
    P.row_tiles = c_in / 64; P.col_tiles = c_out / 64;
    P.nsplit = pick_nsplit3(P.ntiles, P.row_tiles * P.col_tiles);
    if (tw6) {      // 32 x 6 tiles, three dy columns per group: 36 fragment KiB per 108 MFMAs instead of 28 per 72
      const size_t lds = 2 * ((39 + 28) * 1024) + 9 * 512 * 4;
      UNETK_WITH_H(dtype, {
        UNETK_ALLOW_LDS((wgrad5_bf16_kernel<H, 64, 6>), lds);
        hipLaunchKernelGGL((wgrad5_bf16_kernel<H, 64, 6>), dim3(P.row_tiles * P.col_tiles * P.nsplit), dim3(512), lds,
                         (hipStream_t)stream, P);
      });
    } else {
#ifdef UNETK_TUNING
      const size_t lds = 2 * (50 * 1024) + 7 * 512 * 4;
      UNETK_WITH_H(dtype, {
        UNETK_ALLOW_LDS((wgrad5_bf16_kernel<H, 64, 4>), lds);
        hipLaunchKernelGGL((wgrad5_bf16_kernel<H, 64, 4>), dim3(P.row_tiles * P.col_tiles * P.nsplit), dim3(512), lds,
                         (hipStream_t)stream, P);
      });
#endif
    }
    UNETK_CHECK_LAUNCH();
    launch_wgrad_reduce((const float*)workspace, dw, 2 * P.nsplit, 9, c_in, c_out, accumulate, (hipStream_t)stream);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  if ((w5_direct || w5_swapped) && wide && gen5 && h + 1 >= 34 &&
      (int64_t)h * w * (ldx > lddy ? ldx : lddy) * 4 < (int64_t)1 << 31) {
    Wgrad5Params P{};
    P.slab = (float*)workspace;
    P.N = n; P.H = h; P.W = w;
    if (w5_direct) {
      P.a = (const u16*)x; P.b = (const u16*)dy; P.Ca = c_in; P.Cb = c_out; P.lda = ldx; P.ldb = lddy;
    } else {
      P.a = (const u16*)dy; P.b = (const u16*)x; P.Ca = c_out; P.Cb = c_in; P.lda = lddy; P.ldb = ldx;
    }
    UNETK_KNOB(tw5, "UNETK_WGRAD_TW5", 1);     // 0: A/B against 32 x 4 tiles
    P.tiles_x = host_cdiv(w, tw5 ? 5 : 4);
    P.ntiles = host_cdiv((long)n * (h + 1), 32) * P.tiles_x;
    P.row_tiles = P.Ca / 64; P.col_tiles = P.Cb / 128;
    P.nsplit = pick_nsplit3(P.ntiles, P.row_tiles * P.col_tiles);
    if (tw5) {
      const size_t lds = 2 * ((W5_A_PIECES + 40) * 1024) + 9 * 512 * 4;
      UNETK_WITH_H(dtype, {
        UNETK_ALLOW_LDS((wgrad5_bf16_kernel<H, 128, 5>), lds);
        hipLaunchKernelGGL((wgrad5_bf16_kernel<H, 128, 5>), dim3(P.row_tiles * P.col_tiles * P.nsplit), dim3(512), lds,
                         (hipStream_t)stream, P);
      });
    } else {
#ifdef UNETK_TUNING      // A/B builds only: the 32 x 4 form is not compiled into the shipped library
      const size_t lds = 2 * W5_BUF + W5_SLOTS * 512 * 4;
      UNETK_WITH_H(dtype, {
        UNETK_ALLOW_LDS((wgrad5_bf16_kernel<H, 128, 4>), lds);
        hipLaunchKernelGGL((wgrad5_bf16_kernel<H, 128, 4>), dim3(P.row_tiles * P.col_tiles * P.nsplit), dim3(512), lds,
                         (hipStream_t)stream, P);
      });
#endif
    }
    UNETK_CHECK_LAUNCH();
    launch_wgrad_reduce((const float*)workspace, dw, P.nsplit, 9, P.Ca, P.Cb, accumulate, (hipStream_t)stream, w5_swapped);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  if (c_out % 128 == 0 && wide) {
    Wgrad3Params P{};
    P.a = (const u16*)x; P.b = (const u16*)dy; P.slab = (float*)workspace;
    P.N = n; P.H = h; P.W = w; P.Ca = c_in; P.Cb = c_out; P.lda = ldx; P.ldb = lddy;
    P.tiles_x = host_cdiv(w, 8); P.tiles_y = host_cdiv(h, 16); P.ntiles = n * P.tiles_x * P.tiles_y;
    P.row_tiles = c_in / 64; P.col_tiles = c_out / 128;
    P.nsplit = pick_nsplit3(P.ntiles, P.row_tiles * P.col_tiles);
    {
      UNETK_KNOB(dbg, "UNETK_WGRAD_DBG", 0);
      P.dbg = dbg;
    }
    const size_t lds = 2 * W4_BUF;
    UNETK_WITH_H(dtype, {
      UNETK_ALLOW_LDS((wgrad4_bf16_kernel<H>), lds);
      hipLaunchKernelGGL((wgrad4_bf16_kernel<H>), dim3(P.row_tiles * P.col_tiles * P.nsplit), dim3(512), lds,
                       (hipStream_t)stream, P);
    });
    UNETK_CHECK_LAUNCH();
    launch_wgrad_reduce((const float*)workspace, dw, P.nsplit, 9, c_in, c_out, accumulate, (hipStream_t)stream);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  Wgrad3Params P{};
  P.a = (const u16*)x; P.b = (const u16*)dy; P.slab = (float*)workspace;
  P.N = n; P.H = h; P.W = w; P.Ca = c_in; P.Cb = c_out; P.lda = ldx; P.ldb = lddy;
  P.tiles_x = host_cdiv(w, 16); P.tiles_y = host_cdiv(h, 16); P.ntiles = n * P.tiles_x * P.tiles_y;
  P.row_tiles = c_in / 64; P.col_tiles = c_out / 64;
  P.nsplit = pick_nsplit3(P.ntiles, P.row_tiles * P.col_tiles);
  {
    UNETK_KNOB(dbg, "UNETK_WGRAD_DBG", 0);
    P.dbg = dbg;
  }
  const size_t lds = 2 * W3_BUF;
  UNETK_WITH_H(dtype, {
    UNETK_ALLOW_LDS((wgrad3_bf16_kernel<H>), lds);
    hipLaunchKernelGGL((wgrad3_bf16_kernel<H>), dim3(P.row_tiles * P.col_tiles * P.nsplit), dim3(512), lds,
                     (hipStream_t)stream, P);
  });
  UNETK_CHECK_LAUNCH();
  const long total = 9L * c_in * c_out;
  launch_wgrad_reduce((const float*)workspace, dw, P.nsplit, 9, c_in, c_out, accumulate, (hipStream_t)stream);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

int unetk_convT2x2_wgrad(const void* x, int ldx, const void* dy, int lddy, int dy_choff, float* dw, void* workspace,
                         int64_t workspace_bytes, int n, int h, int w, int c_in, int c_out, int hs, int ws, int pad_t,
                         int pad_l, int accumulate, int dtype, void* stream) {
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  const int es = unetk_esize(dtype);
  const int64_t need = unetk_wgrad_workspace_bytes(1, n, h, w, c_in, c_out);
  if (need < 0) return (int)need;
  if (!x || !dy || !dw || !workspace || workspace_bytes < need) return UNETK_ERR_ARG;
  if (pad_t < 0 || pad_l < 0 || pad_t + 2 * h > hs || pad_l + 2 * w > ws) return UNETK_ERR_SHAPE;
  if (ldx < c_in || lddy < dy_choff + c_out || (ldx * es) % 16 || (lddy * es) % 16 || (dy_choff * es) % 16)
    return UNETK_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) return UNETK_ERR_ALIGN;
  if (dtype == UNETK_F32) {
    WgradParamsF32 Q{};
    Q.a = (const float*)dy + dy_choff; Q.b = (const float*)x; Q.slab = (float*)workspace;
    Q.N = n; Q.H = h; Q.W = w; Q.Ca = c_out; Q.Cb = c_in; Q.lda = lddy; Q.ldb = ldx;
    Q.Ha = hs; Q.Wa = ws; Q.padT = pad_t; Q.padL = pad_l;
    Q.tiles_x = host_cdiv(w, 32); Q.tiles_y = host_cdiv(h, 8); Q.ntiles = n * Q.tiles_x * Q.tiles_y;
    Q.row_tiles = c_out / 32; Q.col_tiles = c_in / 32;
    Q.nsplit = pick_nsplit(Q.ntiles, Q.row_tiles * Q.col_tiles);
    const size_t ldsq = 10 * 36 * 128 + 256 * 128;
    UNETK_ALLOW_LDS((wgrad_f32_kernel<1>), ldsq);
    hipLaunchKernelGGL(wgrad_f32_kernel<1>, dim3(Q.row_tiles * Q.col_tiles * Q.nsplit), dim3(256), ldsq,
                       (hipStream_t)stream, Q);
    UNETK_CHECK_LAUNCH();
    const long totalq = 4L * c_in * c_out;
    launch_wgrad_reduce((const float*)workspace, dw, Q.nsplit, 4, c_out, c_in, accumulate, (hipStream_t)stream);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  UNETK_KNOB(gen2, "UNETK_WGRADT_GEN2", 1);   // 0: A/B runs
  if (c_in % 128 == 0 && gen2) {
    if ((int64_t)hs * ws * lddy * 2 >= (int64_t)1 << 32 || (int64_t)h * w * ldx * 2 >= (int64_t)1 << 32)
      return UNETK_ERR_SHAPE;                                  // 32-bit DMA offsets inside an image
    WgradTParams T{};
    T.dy = (const u16*)dy + dy_choff; T.x = (const u16*)x; T.slab = (float*)workspace;
    T.N = n; T.H = h; T.W = w; T.Co = c_out; T.Ci = c_in; T.lddy = lddy; T.ldx = ldx;
    T.Hs = hs; T.Ws = ws; T.padT = pad_t; T.padL = pad_l;
    T.tiles_x = host_cdiv(w, 32); T.tiles_y = host_cdiv(h, 2); T.ntiles = n * T.tiles_x * T.tiles_y;
    T.row_tiles = c_out / 64; T.col_tiles = c_in / 128;
    T.nsplit = pick_nsplit3(T.ntiles, T.row_tiles * T.col_tiles);
    const size_t ldst = (size_t)WT_NBUF * WT_BUF;
    UNETK_WITH_H(dtype, {
      UNETK_ALLOW_LDS((wgradT_bf16_kernel<H>), ldst);
      hipLaunchKernelGGL((wgradT_bf16_kernel<H>), dim3(T.row_tiles * T.col_tiles * T.nsplit), dim3(512), ldst,
                       (hipStream_t)stream, T);
    });
    UNETK_CHECK_LAUNCH();
    launch_wgrad_reduce((const float*)workspace, dw, T.nsplit, 4, c_out, c_in, accumulate, (hipStream_t)stream);
    UNETK_CHECK_LAUNCH();
    return UNETK_OK;
  }
  WgradParams P{};
  P.a = (const u16*)dy + dy_choff; P.b = (const u16*)x; P.slab = (float*)workspace;
  P.N = n; P.H = h; P.W = w; P.Ca = c_out; P.Cb = c_in; P.lda = lddy; P.ldb = ldx;
  P.Ha = hs; P.Wa = ws; P.padT = pad_t; P.padL = pad_l;
  P.tiles_x = host_cdiv(w, 32); P.tiles_y = host_cdiv(h, 8); P.ntiles = n * P.tiles_x * P.tiles_y;
  P.row_tiles = c_out / 64; P.col_tiles = c_in / 64;
  P.nsplit = pick_nsplit(P.ntiles, P.row_tiles * P.col_tiles);
  const size_t lds = 10 * 36 * 128 + 256 * 128;
  UNETK_WITH_H(dtype, {
    UNETK_ALLOW_LDS((wgrad_bf16_kernel<H, 1>), lds);
    hipLaunchKernelGGL((wgrad_bf16_kernel<H, 1>), dim3(P.row_tiles * P.col_tiles * P.nsplit), dim3(256), lds,
                     (hipStream_t)stream, P);
  });
  UNETK_CHECK_LAUNCH();
  // rows = co, cols = ci ; parameter layout [Ci][Co][2][2]
  const long total = 4L * c_in * c_out;
  launch_wgrad_reduce((const float*)workspace, dw, P.nsplit, 4, c_out, c_in, accumulate, (hipStream_t)stream);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

}  // extern "C"
