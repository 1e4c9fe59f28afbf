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
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"
#include "conv_config.h"

#ifndef UNETK_DB_PIPE
#define UNETK_DB_PIPE 1      // conv3x3_db_kernel: single fragment set, mid-step chunk barrier (0: round-2 loop)
#endif
#ifndef UNETK_DB_PRIO
#define UNETK_DB_PRIO 0
#endif
#ifndef UNETK_DB_DMAPOS
#define UNETK_DB_DMAPOS 0    // row-block-1 MFMA of a k-step behind which its DMA piece is issued
#endif

namespace {


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

  // ---- block -> (n_tile, a run of `tpb` consecutive spatial tiles).  Walking several tiles per workgroup lets
  // the first chunk of tile i+1 be prefetched under the last MFMAs and the epilogue of tile i; with one tile per
  // workgroup every tile paid an exposed HBM round trip before its first MFMA (ruinous for the K=576 layers).
  // The n-tiles of one spatial tile read the same input pixels.  Workgroup b runs on XCD b % 8, so consecutive
  // (tile group, n-tile) indices are laid out XCD-major: the n-tiles of a tile group then share one XCD's L2
  // (with b % n_tiles every XCD streamed the whole input: 7.5x the algorithmic bytes at the L2's fabric side).
  // The grid is padded to a multiple of 8; surplus workgroups leave before the first barrier.
  const int lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (lin >= P.nblk) return;
  const int nt = lin % P.n_tiles;
  const int grp = lin / P.n_tiles;
  const int tiles_per_img = P.tiles_x * P.tiles_y;
  const int tile_begin = grp * P.tpb;
  const int tile_end = min(tile_begin + P.tpb, P.N * tiles_per_img);

  const int Hx = (MODE == 2) ? P.Hs : P.H, Wx = (MODE == 2) ? P.Ws : P.W;

  // ---- per-thread staging descriptors of the tile being STAGED (chunk-invariant)
  int poff[R];    // element offset of this thread's piece inside the image, -1 = zero fill / unused
  int ldst[R];    // LDS byte address of the piece
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int id = r * 256 + tid;
    const int p = id / NP, c = id % NP;
    ldst[r] = p * S + ((c ^ ((p / PPR) & (NP - 1))) << 4);
  }
  const T* ximg = nullptr;
  auto setup_stage = [&](int tile) {
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * 8, x0 = (rem % P.tiles_x) * 32;
    ximg = reinterpret_cast<const T*>(P.x) + (size_t)img * Hx * Wx * P.ldx + P.xoff;
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
    }
  };

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
      // ALWAYS issue the load (clamped address; the zero fill for padding is applied when the piece is written
      // to LDS): the counted s_waitcnt below relies on every wave having exactly R prefetch loads younger than
      // its LDS-DMA, and nothing here may consume a load result (that would force an early s_waitcnt).
      stage[r] = *reinterpret_cast<const uint4*>(ximg + (poff[r] >= 0 ? poff[r] : 0) + io);
    }
  };

  // epilogue constants.  C layout of 32x32 MFMA: column = lane&31 (-> channels NB*j .. NB*j+NB-1 of this n-tile,
  // by the weight packing), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (-> pixel column in the tile).
  const int ch0 = nt * BN + NB * j;     // first of this lane's NB consecutive channels
  float bias[NB];
  int co0 = ch0, ab = 0;
  if constexpr (MODE == 1) {
    ab = (nt * BN) / P.Co;
    co0 = ch0 - ab * P.Co;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias[nb] = P.bias[co0 + nb];
  }

  if (tile_begin >= tile_end) return;
  setup_stage(tile_begin);
  load_stage(0);

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    for (int it = 0; it < niter; ++it) {
      // [A] every wave has finished reading the previous chunk's LDS image (and the previous tile's `red`)
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
      asm volatile("" ::: "memory");   // keep the prefetch loads YOUNGER than the LDS-DMA in the vmcnt queue
      const bool more_chunks = (it + 1 < niter);
      if (more_chunks || tile + 1 < tile_end) {
        if (more_chunks) {
          load_stage(it + 1);
        } else {                        // first chunk of the NEXT tile
          setup_stage(tile + 1);
          load_stage(0);
        }
        asm volatile("" ::: "memory");
        // the R youngest vector-memory ops are the prefetch; everything older (the LDS-DMA) must have landed
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(R) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }

      // ---- all taps of this chunk from the one LDS image; fragments of step s+1 are fetched into a second
      // register set before the MFMAs of step s
      constexpr int NSTEP = TAPS * KS;
      uint4 fa[2][2], fb[2][NB];
      auto load_frags = [&](int step, uint4 (&a)[2], uint4 (&b)[NB]) {
        const int t = step / KS, ks = step % KS;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) a[mb] = *reinterpret_cast<const uint4*>(lds_in + (aaddr[mb][t] ^ (ks << 5)));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          b[nb] = *reinterpret_cast<const uint4*>(lds_w + (step * NB + nb) * 1024 + lane * 16);
      };
      load_frags(0, fa[0], fb[0]);
      if constexpr (sizeof(T) == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NB, 0);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) load_frags(st + 1, fa[(st + 1) & 1], fb[(st + 1) & 1]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) Mma<T>::run(acc[mb][nb], fa[st & 1][mb], fb[st & 1][nb]);
        if constexpr (sizeof(T) == 2) {
          if (st + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NB, 0);   // prefetch of step st+1 ...
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * NB, 0);                       // ... then the MFMAs of step st
        }
      }
    }

    // ---- epilogue of this tile (the next tile's first chunk is already in flight)
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * 8, x0 = (rem % P.tiles_x) * 32;
    float ssum[NB], ssq[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) ssum[nb] = ssq[nb] = 0.f;
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
        acc[mb][0][i] = 0.f;
        if constexpr (NB > 1) acc[mb][1][i] = 0.f;
        if constexpr (NB > 2) { acc[mb][2][i] = 0.f; acc[mb][3][i] = 0.f; }
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
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 2 * BN) {
        const float v = red[tid] + red[2 * BN + tid] + red[4 * BN + tid] + red[6 * BN + tid];
        double* dst = P.stats + ((size_t)((blockIdx.x + tile) & (UNETK_STAT_REPLICAS - 1)) * P.Cout + nt * BN) * 2 + tid;
        unsafeAtomicAdd(dst, (double)v);
      }
    }
  }
}

// ================================================================================================================
// conv3x3, second generation: ONE 512-thread workgroup per CU, everything double-buffered in LDS by LDS-DMA.
//
//   tile   16 x 32 output pixels x 64 channels; 8 waves, wave w owns tile rows 2w, 2w+1 (as above)
//   LDS    2 x { halo tile 18 x 34 pixels x 64 B  +  9 taps x 2 k-steps x NB weight fragments of 1 KiB } = 152 KiB
//   loop   issue the LDS-DMA of chunk k+1 (input pieces with a per-lane source address, a zero page for the padding;
//          weights linear) -> MFMAs of chunk k -> s_waitcnt vmcnt(0) + ONE barrier.  The DMA has a whole MFMA phase
//          (72 MFMAs per wave) to land, no VGPR staging, no ds_write, and the chunk stream runs on across the tiles a
//          workgroup walks (the epilogue of tile i overlaps the DMA of tile i+1's first chunk).
// The first generation exposed the weight DMA latency in every chunk (its single LDS buffer can only be refilled
// after the barrier) and relied on a second resident workgroup to cover it.
__device__ uint4 g_zero_page_conv[8];

// KI >= KS: k-steps of input per LDS input buffer.  With KI = 2 KS the input tile holds 64 B per pixel and is refilled
// every second weight chunk: every 128-byte line of the activation is then fetched from L2 twice per tile instead
// of four times (a 16-channel chunk uses 32 B of each pixel's line; TCC_REQ showed 25 M of the 37 M requests of a
// 512->512 launch were such quarter-used lines, and the fill path is what the MFMA phase waits for).
template <typename T, int NB, int KS, int NW, int MB, int KI, bool AFF>
__global__ __launch_bounds__(64 * NW, 2) void conv3x3_db_kernel(ConvParams P) {
  constexpr int E = ElemTraits<T>::E;
  static_assert(KI % KS == 0, "input buffer = whole number of weight chunks");
  constexpr int R = KI / KS;                       // weight chunks per input chunk
  constexpr int NP = 2 * KI, CHI = NP * E, CH = 2 * KS * E, S = 32 * KI, PPR = 256 / S, TAPS = 9;
  constexpr int HP = 34;                          // halo row pitch (pixels)
  constexpr int TR = MB * NW;                     // tile rows: each wave owns MB
  constexpr int NPIX = (TR + 2) * HP;
  constexpr int IN_INSTR = (NPIX * NP + 63) / 64; // 39 one-KiB DMA pieces
  constexpr int IN_BYTES = IN_INSTR * 1024;
  constexpr int WFRAGS = TAPS * KS * NB;
  constexpr int W_BYTES = WFRAGS * 1024;
  constexpr int BN = 32 * NB;
  constexpr int IN_PER_WAVE = (IN_INSTR + NW - 1) / NW;
  constexpr int IN_PER_SUB = (IN_PER_WAVE + R - 1) / R;    // input pieces a wave issues per weight chunk
  constexpr int W_PER_WAVE = (WFRAGS + NW - 1) / NW;
  // LDS: [input 0][input 1][weights 0][weights 1]

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;

  const int nt = blockIdx.x % P.n_tiles;
  const int grp = blockIdx.x / P.n_tiles;
  const int tiles_per_img = P.tiles_x * P.tiles_y;
  const int tile_begin = grp * P.tpb;
  const int tile_end = min(tile_begin + P.tpb, P.N * tiles_per_img);
  if (tile_begin >= tile_end) return;

  const int nchunks = P.Cin / CH;               // weight chunks
  const int nin = P.Cin / CHI;                  // input chunks
  const char* wbase = reinterpret_cast<const char*>(P.wpk) + (size_t)nt * nchunks * (WFRAGS * 1024);
  const char* zero = reinterpret_cast<const char*>(g_zero_page_conv);

  // per-lane A-fragment read addresses (k-step 0 of the chunk; k-step 1 is ^32)
  int aaddr[MB][TAPS];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int p = (MB * wid + mb + t / 3) * HP + j + (t % 3);
      aaddr[mb][t] = p * S + ((h ^ ((p / PPR) & (NP - 1))) << 4);
    }

  // DMA descriptors of the tile being staged: byte offset of this lane's piece (chunk 0) or -1 -> zero page
  long dma_off[IN_PER_WAVE];
  const char* ximg = nullptr;
  auto setup_stage = [&](int tile) {
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * TR, x0 = (rem % P.tiles_x) * 32;
    ximg = reinterpret_cast<const char*>(reinterpret_cast<const T*>(P.x) + (size_t)img * P.H * P.W * P.ldx + P.xoff);
#pragma unroll
    for (int f = 0; f < IN_PER_WAVE; ++f) {
      const int fi = min(f * NW + wid, IN_INSTR - 1);   // surplus slots of the issue grid repeat the last piece
      const int id = fi * 64 + lane;           // LDS slot: pixel p, physical piece s
      const int p = id / NP, sl = id % NP;
      const int c = sl ^ ((p / PPR) & (NP - 1));   // logical piece stored in that slot
      const int hr = p / HP, hc = p - hr * HP;
      const int yy = y0 + hr - 1, xx = x0 + hc - 1;
      long off = -1;
      if (p < NPIX && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W)
        off = ((long)(yy * P.W + xx) * P.ldx + c * E) * (long)sizeof(T);
      dma_off[f] = off;
    }
  };
  // one 1-KiB DMA piece: q < IN_PER_WAVE -> input piece q, else weight fragment q - IN_PER_WAVE.  Branch-free
  // (a branch inside the MFMA loop ends the basic block and costs a full lgkmcnt(0) wait): surplus slots repeat the
  // last piece (same bytes to the same place), `live` = false sends every source to the zero page.
  auto issue_input = [&](int q, int in_chunk, unsigned buf, bool live) {
    const int fi = min(q * NW + wid, IN_INSTR - 1);
    const char* src = (live && dma_off[q] >= 0) ? ximg + dma_off[q] + (long)in_chunk * CHI * sizeof(T) : zero;
    // (the nontemporal form of this DMA was 15 % slower: halo rows and n-tiles re-read these lines from L2)
    glds16_hidden(src, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
  };
  auto issue_weight = [&](int q, int chunk, unsigned buf, bool live) {
    const int fi = min(q * NW + wid, WFRAGS - 1);
    const char* src = live ? wbase + (size_t)chunk * W_BYTES + lane * 16 + fi * 1024 : zero;
    glds16_hidden(src, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
  };
  constexpr int NDMA = IN_PER_SUB + W_PER_WAVE;     // pieces per wave per weight chunk

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  const int ch0 = nt * BN + NB * j;

  setup_stage(tile_begin);
  const unsigned lds0 = lds_address(smem);
  const unsigned ldsw0 = lds0 + 2 * IN_BYTES;
#pragma unroll
  for (int q = 0; q < IN_PER_WAVE; ++q) issue_input(q, 0, lds0, true);
#pragma unroll
  for (int q = 0; q < W_PER_WAVE; ++q) issue_weight(q, 0, ldsw0, true);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  int curi = 0, curw = 0;
  // BN statistics of all tiles of this workgroup: per tile in fp32 (as the values are produced), across tiles in
  // fp32 for bf16 tensors and in fp64 for fp32 tensors (their 1e-4 parity bar needs E[x^2] - mean^2 to keep its bits)
  using StatAcc = typename std::conditional<sizeof(T) == 4, double, float>::type;
  StatAcc tsum[NB], tsq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) tsum[nb] = tsq[nb] = 0;
  // AFF: the eval-mode BatchNorm2d + ReLU of the inference path folded into the store (no statistics, no second pass)
  float asc[NB], ash[NB];
  if constexpr (AFF) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { asc[nb] = P.aff_scale[ch0 + nb]; ash[nb] = P.aff_shift[ch0 + nb]; }
  }

#if UNETK_DB_PIPE
#if UNETK_DB_PRIO
  if (wid >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
  // ONE fragment register set, each fragment re-read right after the MFMA that used it last; the k-step stream runs
  // on across weight chunks, input chunks and tiles.  The chunk barrier sits in the MIDDLE of a chunk's last k-step
  // (after its row-block-0 MFMAs): every wave then holds that step's fragments in registers, i.e. is done reading the
  // current buffers, and the first fragments of the next chunk are read under the row-block-1 MFMAs like those of any
  // other step -- no LDS latency is exposed behind a barrier.
  uint4 fa[MB], fb[NB];
  {
    const char* lds_in = smem;
    const char* lds_w = smem + 2 * IN_BYTES;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) fa[mb] = *reinterpret_cast<const uint4*>(lds_in + aaddr[mb][0]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) fb[nb] = *reinterpret_cast<const uint4*>(lds_w + nb * 1024 + lane * 16);
  }
#endif

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    for (int ic = 0; ic < nin; ++ic) {
      // the NEXT input chunk (of this tile, or the first of the next tile) is staged while this one is consumed
      int next_in = ic + 1;
      bool live_in = !(P.dbg & 1);
      if (ic + 1 == nin) {
        next_in = 0;
        if (tile + 1 < tile_end) setup_stage(tile + 1);
        else live_in = false;
      }
#if UNETK_DB_PIPE
#pragma unroll
      for (int sub = 0; sub < R; ++sub) {
        const int kc = ic * R + sub;
        const char* lds_in = smem + curi * IN_BYTES;
        const char* lds_w = smem + 2 * IN_BYTES + curw * W_BYTES;
        // buffers of the next weight chunk's first k-step
        const char* nlds_in = (sub + 1 == R) ? smem + (curi ^ 1) * IN_BYTES : lds_in;
        const char* nlds_w = smem + 2 * IN_BYTES + (curw ^ 1) * W_BYTES;
        const int nsub = (sub + 1 == R) ? 0 : sub + 1;
        int next_w = kc + 1;
        bool live_w = !(P.dbg & 1);
        if (kc + 1 == nchunks) {
          next_w = 0;
          live_w = live_w && (tile + 1 < tile_end);
        }
        const unsigned nbuf_in = lds0 + (curi ^ 1) * IN_BYTES;
        const unsigned nbuf_w = ldsw0 + (curw ^ 1) * W_BYTES;
        constexpr int NSTEP = TAPS * KS;
        constexpr int DSTEPS = NSTEP - 1;                   // the last k-step carries the barrier instead
        constexpr int PPS = (NDMA + DSTEPS - 1) / DSTEPS;
        auto issue_piece = [&](int q) {
          if (q < IN_PER_SUB) {
            const int qi = sub * IN_PER_SUB + q;
            if (qi < IN_PER_WAVE) issue_input(qi, next_in, nbuf_in, live_in);
          } else {
            issue_weight(q - IN_PER_SUB, next_w, nbuf_w, live_w);
          }
        };
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
          const bool last = st + 1 == NSTEP;
          // where the fragments of the k-step after this one live
          const int nst = last ? 0 : st + 1;
          const int nt_ = nst / KS, nks = nst % KS;
          const char* rin = last ? nlds_in : lds_in;
          const char* rw = last ? nlds_w : lds_w;
          const int rsub = last ? nsub : sub;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            Mma<T>::run(acc[0][nb], fa[0], fb[nb]);
            if (nb + 1 == -UNETK_DB_DMAPOS && !last) {
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int q = st * PPS; q < (st + 1) * PPS && q < NDMA; ++q) issue_piece(q);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (last) {
            // next chunks landed (this wave's DMAs), every wave holds its last fragments of the current buffers
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          }
          fa[0] = *reinterpret_cast<const uint4*>(rin + (aaddr[0][nt_] ^ ((rsub * KS + nks) << 5)));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mb = 1; mb < MB; ++mb) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              Mma<T>::run(acc[mb][nb], fa[mb], fb[nb]);
              __builtin_amdgcn_sched_barrier(0);
              if (mb == MB - 1) {
                fb[nb] = *reinterpret_cast<const uint4*>(rw + (nst * NB + nb) * 1024 + lane * 16);
                if (nb == UNETK_DB_DMAPOS && !last) {
#pragma unroll
                  for (int q = st * PPS; q < (st + 1) * PPS && q < NDMA; ++q) issue_piece(q);
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            fa[mb] = *reinterpret_cast<const uint4*>(rin + (aaddr[mb][nt_] ^ ((rsub * KS + nks) << 5)));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        curw ^= 1;
      }
      curi ^= 1;
    }
#else
#pragma unroll
      for (int sub = 0; sub < R; ++sub) {
        const int kc = ic * R + sub;
        const char* lds_in = smem + curi * IN_BYTES;
        const char* lds_w = smem + 2 * IN_BYTES + curw * W_BYTES;
        // DMA of the next weight chunk and of 1/R of the next input chunk, ONE piece per k-step between the MFMAs:
        // an LDS-DMA instruction holds the issuing wave for 60-185 cycles (MI355X_MICROARCH "LDS-DMA piece issue
        // cost"); in a burst all eight waves stall together and the matrix pipe drains, spread out the partner wave
        // on the SIMD keeps it fed.
        int next_w = kc + 1;
        bool live_w = !(P.dbg & 1);
        if (kc + 1 == nchunks) {
          next_w = 0;
          live_w = live_w && (tile + 1 < tile_end);
        }
        const unsigned nbuf_in = lds0 + (curi ^ 1) * IN_BYTES;
        const unsigned nbuf_w = ldsw0 + (curw ^ 1) * W_BYTES;
        constexpr int NSTEP = TAPS * KS;
        constexpr int PPS = (NDMA + NSTEP - 1) / NSTEP;     // DMA pieces per k-step (1 with 8 waves)
        auto issue_piece = [&](int q) {
          if (q < IN_PER_SUB) {
            const int qi = sub * IN_PER_SUB + q;
            if (qi < IN_PER_WAVE) issue_input(qi, next_in, nbuf_in, live_in);
          } else {
            issue_weight(q - IN_PER_SUB, next_w, nbuf_w, live_w);
          }
        };
        uint4 fa[2][MB], fb[2][NB];
        auto load_frags = [&](int step, uint4 (&a)[MB], uint4 (&b)[NB]) {
          const int t = step / KS, ks = step % KS;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            a[mb] = *reinterpret_cast<const uint4*>(lds_in + (aaddr[mb][t] ^ ((sub * KS + ks) << 5)));
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            b[nb] = *reinterpret_cast<const uint4*>(lds_w + (step * NB + nb) * 1024 + lane * 16);
        };
        if (!(P.dbg & 2)) {
          load_frags(0, fa[0], fb[0]);
#pragma unroll
          for (int st = 0; st < NSTEP; ++st) {
            if (st + 1 < NSTEP) load_frags(st + 1, fa[(st + 1) & 1], fb[(st + 1) & 1]);
#pragma unroll
            for (int q = st * PPS; q < (st + 1) * PPS && q < NDMA; ++q) issue_piece(q);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) Mma<T>::run(acc[mb][nb], fa[st & 1][mb], fb[st & 1][nb]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < NDMA; ++q) issue_piece(q);
        }
        // next chunks landed (this wave's DMAs) and every wave is done reading the current buffers
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        curw ^= 1;
      }
      curi ^= 1;
    }
#endif

    // ---- epilogue of this tile.  64-bit address arithmetic only per row block (wave-uniform); an element adds a
    // 32-bit lane offset and a compile-time multiple of the pixel pitch.
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * TR, x0 = (rem % P.tiles_x) * 32;
    T* yimg = reinterpret_cast<T*>(P.y) + (size_t)img * P.H * P.W * P.ldy + P.yoff;
    const int lane_off = 4 * h * P.ldy + ch0;
    const int xl = x0 + 4 * h;
    float ssum[NB], ssq[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) ssum[nb] = ssq[nb] = 0.f;
    if constexpr (AFF && MB == 2) {
      // the encoder's MaxPool2d(2) (unet_parts.py:34) of the activation about to be stored: a wave owns tile rows
      // 2w, 2w+1 and a lane holds horizontally adjacent pixels in accumulator elements i, i+1 -- every complete
      // 2x2 window is four registers of one lane.  max commutes with the (monotone) rounding of the store.
      if (P.pool != nullptr) {
        const int yy = y0 + 2 * wid;                       // even; the window needs row yy + 1 too
        T* pimg = reinterpret_cast<T*>(P.pool) + (size_t)img * (P.H / 2) * (P.W / 2) * P.ldp;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const int xx = xl + (i & 3) + 8 * (i >> 2);        // even
          if (yy + 1 < P.H && xx + 1 < P.W) {
            float pv[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              const float a = fmaxf(fmaf(acc[0][nb][i], asc[nb], ash[nb]), 0.f);
              const float b = fmaxf(fmaf(acc[0][nb][i + 1], asc[nb], ash[nb]), 0.f);
              const float c = fmaxf(fmaf(acc[1][nb][i], asc[nb], ash[nb]), 0.f);
              const float d = fmaxf(fmaf(acc[1][nb][i + 1], asc[nb], ash[nb]), 0.f);
              pv[nb] = fmaxf(fmaxf(a, b), fmaxf(c, d));
            }
            store_channels<T, NB>(pimg + ((size_t)(yy >> 1) * (P.W / 2) + (xx >> 1)) * P.ldp + ch0, pv);
          }
        }
      }
    }
    bool done = false;
    if constexpr (sizeof(T) == 2 && !AFF && NB == 4) {
      // Interior tiles of the bf16 training path (no pixel outside the image): no per-element predicates, and the
      // accumulators are rounded ONCE into packed pairs of horizontally adjacent pixels -- the BatchNorm statistics
      // come from the packed values with v_dot2c_f32_bf16 (sum: y.1, sum of squares: y.y; exact products accumulated
      // in fp32, two pixels per instruction) and the 8-byte channel vectors of the stores are gathered with v_perm:
      // 2 VALU instructions per output value instead of 4.5.
      if ((y0 + TR <= P.H) && (x0 + 32 <= P.W) && !(P.dbg & 4)) {
        auto fast = [&](auto stats_c) {
          constexpr bool ST = decltype(stats_c)::value;
          typedef typename H16<T>::x2 h2;
          const h2 one2 = {(T)1.0f, (T)1.0f};
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int yy = y0 + MB * wid + mb;
            T* rowp = yimg + (size_t)(yy * P.W + x0) * P.ldy + lane_off;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              const int ri = (i & 3) + 8 * (i >> 2);
              unsigned pk[NB];
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                const h2 y2 = {(T)acc[mb][nb][i], (T)acc[mb][nb][i + 1]};
                pk[nb] = __builtin_bit_cast(unsigned, y2);
                if constexpr (ST) {
                  ssum[nb] = H16<T>::dot2(y2, one2, ssum[nb]);
                  ssq[nb] = H16<T>::dot2(y2, y2, ssq[nb]);
                }
                acc[mb][nb][i] = 0.f;
                acc[mb][nb][i + 1] = 0.f;
              }
              typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
              const u32x2 p0 = {__builtin_amdgcn_perm(pk[1], pk[0], 0x05040100u), __builtin_amdgcn_perm(pk[3], pk[2], 0x05040100u)};
              const u32x2 p1 = {__builtin_amdgcn_perm(pk[1], pk[0], 0x07060302u), __builtin_amdgcn_perm(pk[3], pk[2], 0x07060302u)};
              __builtin_nontemporal_store(p0, reinterpret_cast<u32x2*>(rowp + ri * P.ldy));
              __builtin_nontemporal_store(p1, reinterpret_cast<u32x2*>(rowp + (ri + 1) * P.ldy));
            }
          }
        };
        if (P.stats != nullptr) fast(std::true_type{});
        else fast(std::false_type{});
        done = true;
      }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      if (done) break;
      const int yy = y0 + MB * wid + mb;
      T* rowp = yimg + (size_t)(yy * P.W + x0) * P.ldy;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ri = (i & 3) + 8 * (i >> 2);
        if (yy < P.H && xl + ri < P.W) {
          float v[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            if constexpr (AFF) {
              v[nb] = fmaxf(fmaf(acc[mb][nb][i], asc[nb], ash[nb]), 0.f);
            } else {
              const float f = to_f32(from_f32<T>(acc[mb][nb][i]));
              v[nb] = f;
              ssum[nb] += f;
              ssq[nb] += f * f;
            }
          }
          if (!(P.dbg & 4)) store_channels<T, NB>(rowp + ri * P.ldy + lane_off, v);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb][i] = 0.f;
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { tsum[nb] += (StatAcc)ssum[nb]; tsq[nb] += (StatAcc)ssq[nb]; }
  }

  // ---- one LDS reduction and one set of fp64 atomics per workgroup
  if (!AFF && P.stats != nullptr) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      tsum[nb] += __shfl_xor(tsum[nb], 32, 64);
      tsq[nb] += __shfl_xor(tsq[nb], 32, 64);
    }
    StatAcc* red = reinterpret_cast<StatAcc*>(smem);   // [NW waves][BN][2]; every DMA landed, every read retired
    if (h == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        red[(wid * BN + NB * j + nb) * 2 + 0] = tsum[nb];
        red[(wid * BN + NB * j + nb) * 2 + 1] = tsq[nb];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < 2 * BN) {
      StatAcc v = 0;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) v += red[w8 * 2 * BN + tid];
      double* dst = P.stats + ((size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * P.Cout + nt * BN) * 2 + tid;
      unsafeAtomicAdd(dst, (double)v);
    }
  }
}

template <typename T, int NB, int KS, int NW, int MB = 2, int KI = KS, bool AFF = false>
int launch_db(const ConvParams& P, hipStream_t s) {
  constexpr size_t lds = 2 * ((size_t)(((MB * NW + 2) * 34 * 2 * KI + 63) / 64) * 1024 + 9 * KS * NB * 1024);
  UNETK_ALLOW_LDS((conv3x3_db_kernel<T, NB, KS, NW, MB, KI, AFF>), lds);
  ConvParams Q = P;
  Q.tiles_y = host_cdiv(P.H, MB * NW);
  const long ntiles = (long)P.N * Q.tiles_y * Q.tiles_x;
  const int nchunks = P.Cin / (2 * KS * ElemTraits<T>::E);
  Q.n_tiles = P.Cout / (32 * NB);
  // One workgroup per CU is resident, so a launch runs in ROUNDS of 256 workgroups and a partly filled last round is
  // idle CUs.  Tiles per workgroup: the value that minimises rounds x (tiles x chunks per tile + ~1.5 chunks of
  // prologue/epilogue) -- e.g. 5 tiles turn the 80x119 level's 1280 workgroups (5 rounds) into 256 (one round), +4 %;
  // the 320x479 level's 2400 workgroups (9.4 rounds, the last 37 % full) into 960, +3 % (measured sweep).
  int tpb = 1;
  {
    double best = 1e30;
    UNETK_KNOB(tpb_max, "UNETK_CONV_TPB_MAX", 96);
    for (int t = 1; t <= tpb_max; ++t) {
      const long wgs = ((ntiles + t - 1) / t) * Q.n_tiles;
      const long rounds = (wgs + 255) / 256;
      const double cost = (double)rounds * ((double)t * nchunks + 1.5);
      if (cost < best * 0.999) { best = cost; tpb = t; }
    }
  }
  {
    UNETK_KNOB(tpb_force, "UNETK_CONV_TPB", 0);
    if (tpb_force > 0) tpb = tpb_force;
  }
  Q.tpb = tpb;
  {
    UNETK_KNOB(dbg, "UNETK_CONV_DBG", 0);
    Q.dbg = dbg;
  }
  const long nblk = ((ntiles + tpb - 1) / tpb) * Q.n_tiles;
  if (nblk <= 0 || nblk > 0x7fffffffL) return UNETK_ERR_SHAPE;
  hipLaunchKernelGGL((conv3x3_db_kernel<T, NB, KS, NW, MB, KI, AFF>), dim3((unsigned)nblk), dim3(64 * NW), lds, s, Q);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

// ================================================================================================================
// ConvTranspose2d k2 s2 forward (MODE 1) and dgrad (MODE 2), second generation: the same double-buffered hidden
// LDS-DMA pipeline as conv3x3_db_kernel, for the pointwise GEMMs.
//
//   tile   TP = 32 * MBS * NW CONSECUTIVE pixels of the flattened low-res grid (no 2-D tile, no ragged-edge waste) x 128
//          channels; wave w owns pixels 32*MBS*w .. (MBS 32-pixel MFMA row blocks) x 4 column blocks
//   LDS    NBUF x { TP pixels x 32*KS B (16*KS bf16 / 8*KS fp32 channels)  +  KS k-steps x 4 weight fragments }
//   loop   chunk = (tap,) 16*KS channels: KS k-steps x 4*MBS MFMAs per wave; the DMA pieces of the chunk NBUF - 1 ahead go
//          out between the k-steps; one barrier per chunk, behind `vmcnt(pieces of the NBUF - 2 newest chunks)`
//   MODE 1 epilogue: + bias, pixel-shuffle store (tap = channel / Co) into the concat buffer at a channel offset
//   MODE 2 operand:  per tap the stride-2 gather of the up-sampled gradient (per-lane DMA source addresses)
// Forms: fp32 -- 8 waves x 2 row blocks, 32-channel (128-byte) chunks, two 80-KiB buffers (round 2);
// 16-bit types -- 8 waves x 1 row block (256-pixel tiles), 64-channel chunks = whole 128-byte lines per DMA piece, three
// 48-KiB buffers with two chunks of look-ahead (round 3).  The 4-wave / 32-channel forms of round 2 (two workgroups per
// CU) are compiled only into -DUNETK_TUNING builds.  The packed weight stream is the same for all (k-steps in order).
// BNR (MODE 2, MBS == 1, 16-bit types): the written gradient is the gradient of a conv -> BN -> ReLU layer's activation
// (the Up block's low-res input).  That layer's pre-BatchNorm tensor z at the tile's pixels is fetched two chunks before
// the tile ends (hidden loads, counted in the loop's own vmcnt arithmetic so that the DMA look-ahead across the tile
// boundary survives) and the epilogue forms its BN-backward sums from the ROUNDED values it stores: the stand-alone
// reduction pass (two reads of the tensor) does not run.
template <typename T, int MODE, int NW = 8, int KS = 4, int NBUF = 2, int MBS = 2, bool BNR = false>
__global__ __launch_bounds__(64 * NW, (MBS == 1 && KS == 2 ? 4 : 2)) void pw_db_kernel(ConvParams P) {
  static_assert(!BNR || (MODE == 2 && MBS == 1 && sizeof(T) == 2 && NBUF >= 3), "BNR: dgrad, one row block, 16-bit");
  constexpr int LA = NBUF - 1;                    // chunks of DMA look-ahead
  constexpr int E = ElemTraits<T>::E;
  constexpr int NB = 4;
  constexpr int NP = 2 * KS, CH = NP * E, S = 32 * KS;
  constexpr int TP = 32 * MBS * NW;               // pixels per tile (MBS 32-pixel row blocks per wave)
  constexpr int IN_INSTR = TP * S / 1024;         // 64
  constexpr int IN_BYTES = IN_INSTR * 1024;
  constexpr int WFRAGS = KS * NB;                 // 16
  constexpr int BUF = IN_BYTES + WFRAGS * 1024;
  constexpr int BN = 32 * NB;
  constexpr int IN_PER_WAVE = IN_INSTR / NW;      // 8
  constexpr int W_PER_WAVE = WFRAGS / NW;         // 2
  constexpr int NDMA = IN_PER_WAVE + W_PER_WAVE;  // 10
  constexpr int PPS = (NDMA + KS - 1) / KS;       // DMA pieces per k-step

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;

  const int lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-major, see conv_igemm_kernel
  if (lin >= P.nblk) return;
  const int nt = lin % P.n_tiles;
  const int grp = lin / P.n_tiles;
  const long npix = (long)P.N * P.H * P.W;
  const int ntiles = (int)((npix + TP - 1) / TP);
  const int tile_begin = grp * P.tpb;
  const int tile_end = min(tile_begin + P.tpb, ntiles);
  if (tile_begin >= tile_end) return;

  const int nchunks = P.Cin / CH;
  const int niter = nchunks * (MODE == 2 ? 4 : 1);
  const char* wbase = reinterpret_cast<const char*>(P.wpk) + (size_t)nt * niter * (WFRAGS * 1024);
  const char* zero = reinterpret_cast<const char*>(g_zero_page_conv);
  const char* xbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(P.x) + P.xoff);
  const int HW = P.H * P.W;

  int aaddr[MBS];
#pragma unroll
  for (int mb = 0; mb < MBS; ++mb) {
    const int p = (MBS * wid + mb) * 32 + j;
    aaddr[mb] = p * S + ((h ^ ((p >> 1) & (NP - 1))) << 4);
  }

  // byte offset (chunk 0 / tap 0) of this lane's 16 bytes of every input piece, -1 -> zero page
  long dma_off[IN_PER_WAVE];
  auto setup_stage = [&](int tile) {
#pragma unroll
    for (int f = 0; f < IN_PER_WAVE; ++f) {
      const int fi = f * NW + wid;
      const int id = fi * 64 + lane;
      const int p = id / NP, sl = id % NP;
      const int c = sl ^ ((p >> 1) & (NP - 1));
      const long g = (long)tile * TP + p;
      long off = -1;
      if (g < npix) {
        if constexpr (MODE == 1) {
          off = (g * P.ldx + c * E) * (long)sizeof(T);
        } else {
          const int img = (int)(g / HW), r = (int)(g - (long)img * HW);
          const int yy = r / P.W, xx = r - yy * P.W;
          off = ((((long)img * P.Hs + 2 * yy + P.padT) * P.Ws + 2 * xx + P.padL) * P.ldx + c * E) * (long)sizeof(T);
        }
      }
      dma_off[f] = off;
    }
  };
  auto iter_bytes = [&](int it) -> long {
    if constexpr (MODE == 2) {
      const int tap = it / nchunks, kc = it - tap * nchunks;
      return ((long)((tap >> 1) * P.Ws + (tap & 1)) * P.ldx + kc * CH) * (long)sizeof(T);
    } else {
      return (long)it * CH * (long)sizeof(T);
    }
  };
  // branch-free (a branch inside the MFMA loop ends the basic block and costs a full lgkmcnt(0) wait)
  auto issue_piece = [&](int q, int it, long ib, unsigned buf, bool live) {
    if (q < IN_PER_WAVE) {
      const int fi = q * NW + wid;
      const char* src = (live && dma_off[q] >= 0) ? xbase + dma_off[q] + ib : zero;
      glds16_hidden(src, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
    } else {
      const int fi = (q - IN_PER_WAVE) * NW + wid;
      const char* src = live ? wbase + (size_t)it * (WFRAGS * 1024) + lane * 16 + fi * 1024 : zero;
      glds16_hidden(src, __builtin_amdgcn_readfirstlane(buf + IN_BYTES + fi * 1024));
    }
  };

  f32x16 acc[MBS][NB];
#pragma unroll
  for (int mb = 0; mb < MBS; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  const int ch0 = nt * BN + NB * j;
  float bias[NB];
  int co0 = ch0, ab = 0;
  if constexpr (MODE == 1) {
    ab = ch0 / P.Co;
    co0 = ch0 - ab * P.Co;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias[nb] = P.bias[co0 + nb];
  }

  setup_stage(tile_begin);
  const unsigned lds0 = lds_address(smem);
  // the first LA chunks of the first tile (the launcher guarantees niter >= LA)
#pragma unroll
  for (int c = 0; c < LA; ++c)
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_piece(q, c, iter_bytes(c), lds0 + c * BUF, true);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NDMA * (LA - 1)) : "memory");
  int cur = 0;

  // BNR state: z words of the lane's 16 pixels x 4 adjacent channels, scale/shift of those channels, running sums
  unsigned long long zq[BNR ? 16 : 1];
  float bsc[NB], bsh[NB], ts1[NB], ts2[NB];
  if constexpr (BNR) {
#pragma unroll
    for (int i = 0; i < 16; ++i) zq[i] = 0ull;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      bsc[nb] = P.bn_scale[ch0 + nb];
      bsh[nb] = P.bn_shift[ch0 + nb];
      ts1[nb] = ts2[nb] = 0.f;
    }
  }

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    for (int it = 0; it < niter; ++it) {
      if constexpr (BNR) {
        if (it == niter - 2) {
          // 16 hidden 8-byte loads (in front of this iteration's DMA pieces: the iteration ends on vmcnt(NDMA + 16), the
          // next one on vmcnt(NDMA), by which the words have landed).  Scalar base + 32-bit lane offset, rows past the
          // last pixel clamped onto it (masked at use): no per-load 64-bit address lives in registers.
          const long gb = (long)tile * TP + wid * 32;
          const long gbc = gb < npix ? gb : npix - 1;
          const char* zb = reinterpret_cast<const char*>(P.bn_z) + (size_t)gbc * P.ld_bnz * sizeof(T);
          const long left = npix - 1 - gbc;
          const int rows_left = left < 31 ? (int)left : 31;
          const unsigned pitch = (unsigned)P.ld_bnz * (unsigned)sizeof(T);
          const unsigned lane_off = (unsigned)ch0 * (unsigned)sizeof(T);
          int h4 = 4 * h;
          asm volatile("" : "+v"(h4));      // (else the 16 offsets are hoisted out of the tile loop and held in registers)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = min((i & 3) + 8 * (i >> 2) + h4, rows_left);
            const unsigned voff = (unsigned)row * pitch + lane_off;
            asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(zq[i]) : "v"(voff), "s"(zb) : "memory");
          }
        }
      }
      const char* lds_in = smem + cur * BUF;
      const char* lds_w = lds_in + IN_BYTES;
      // the chunk issued during this one lies LA chunks ahead, possibly in the next tile
      const bool wraps = it + LA >= niter;
      const bool live = (!wraps || tile + 1 < tile_end) && !(P.dbg & 1);
      if (it + LA == niter && live) setup_stage(tile + 1);
      const int nit = wraps ? it + LA - niter : it + LA;
      const long nib = iter_bytes(nit);
      const int nidx = cur + LA >= NBUF ? cur + LA - NBUF : cur + LA;
      const unsigned nbuf = lds0 + nidx * BUF;

      uint4 fa[2][MBS], fb[2][NB];
      auto load_frags = [&](int ks, uint4 (&a)[MBS], uint4 (&b)[NB]) {
#pragma unroll
        for (int mb = 0; mb < MBS; ++mb) a[mb] = *reinterpret_cast<const uint4*>(lds_in + (aaddr[mb] ^ (ks << 5)));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          b[nb] = *reinterpret_cast<const uint4*>(lds_w + (ks * NB + nb) * 1024 + lane * 16);
      };
      load_frags(0, fa[0], fb[0]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) load_frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
        for (int q = ks * PPS; q < (ks + 1) * PPS && q < NDMA; ++q) issue_piece(q, nit, nib, nbuf, live);
#pragma unroll
        for (int mb = 0; mb < MBS; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) Mma<T>::run(acc[mb][nb], fa[ks & 1][mb], fb[ks & 1][nb]);
      }
      // the chunk issued LA - 1 iterations ago must have landed; the newest LA - 1 chunks (and nothing older: vmcnt
      // retires in issue order, epilogue stores included) may stay in flight
      if (BNR && it == niter - 2)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NDMA * (LA - 1) + 16) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NDMA * (LA - 1)) : "memory");
      cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
    if constexpr (BNR) {
      // every later use of the z words depends on this statement: nothing that reads them moves above the wait
      asm volatile("" : "+v"(zq[0]), "+v"(zq[1]), "+v"(zq[2]), "+v"(zq[3]), "+v"(zq[4]), "+v"(zq[5]), "+v"(zq[6]),
                   "+v"(zq[7]), "+v"(zq[8]), "+v"(zq[9]), "+v"(zq[10]), "+v"(zq[11]), "+v"(zq[12]), "+v"(zq[13]),
                   "+v"(zq[14]), "+v"(zq[15]));
    }

    // ---- epilogue
#pragma unroll
    for (int mb = 0; mb < MBS; ++mb) {
      const long gb = (long)tile * TP + (MBS * wid + mb) * 32;       // wave-uniform
      // 64-bit address arithmetic once per row block (wave-uniform); per element a 32-bit element offset
      if constexpr (MODE == 1) {
        const int img0 = (int)(gb / HW), r0 = (int)(gb - (long)img0 * HW);
        const int yb = r0 / P.W, xb = r0 - yb * P.W;
        T* dst0 = reinterpret_cast<T*>(P.y) + P.yoff + co0 +
                  (((size_t)img0 * P.Hs + 2 * yb + (ab >> 1) + P.padT) * P.Ws + 2 * xb + (ab & 1) + P.padL) * P.ldy;
        if (P.W >= 32 && gb + 32 <= npix) {
          // fast path (every tile but the ragged last one, any image at least 32 pixels wide): the 32 consecutive
          // low-res pixels of this row block cross at most one row end (and then at most one image end), so the
          // destination is two selects away -- no per-element branches (the general path below cost ~10 us per tile
          // in divergent while-loops and exec-mask branches: the shallow levels ran at 3.3 TB/s)
          const int to_row_end = P.W - xb;                                   // pixels left in this low-res row
          const int wrap_rows = 2 + ((yb + 1 >= P.H) ? P.Hs - 2 * P.H : 0);    // up-sampled rows to the next low-res row
          const int wrap_off = (wrap_rows * P.Ws - 2 * P.W) * P.ldy;          // element offset added after the wrap
          if constexpr (sizeof(T) == 2) {
            // 16-byte stores: this kernel is bound by the store path, which retires wave-instructions rather than
            // bytes (8-byte stores: ~7 B/clk/CU).  Lanes j and j^1 own the channel groups 4j.. and 4j+4.. of the same
            // tap; they swap every other pixel (DPP quad_perm [1,0,3,2]) so that the even lane stores 8 adjacent
            // channels of the even-numbered rows of the block and the odd lane of the odd-numbered ones.
            const bool odd = (j & 1) != 0;
            T* dstp = dst0 - (odd ? 4 : 0);
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              unsigned d[2][2];
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                d[e][0] = pack2<T>(acc[mb][0][i + e] + bias[0], acc[mb][1][i + e] + bias[1]);
                d[e][1] = pack2<T>(acc[mb][2][i + e] + bias[2], acc[mb][3][i + e] + bias[3]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb][i + e] = 0.f;
              }
              unsigned o[4];
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const unsigned give = odd ? d[0][q] : d[1][q];
                const unsigned r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xF, 0xF, false);
                o[q] = odd ? r : d[0][q];
                o[2 + q] = odd ? d[1][q] : r;
              }
              const int row = (i & 3) + 8 * (i >> 2) + 4 * h + (odd ? 1 : 0);
              const int off = 2 * row * P.ldy + (row >= to_row_end ? wrap_off : 0);
              typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
              if (!(P.dbg & 4))
                __builtin_nontemporal_store(u32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<u32x4*>(dstp + off));
            }
            continue;
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            float v[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              v[nb] = acc[mb][nb][i] + bias[nb];
              acc[mb][nb][i] = 0.f;
            }
            const int off = 2 * row * P.ldy + (row >= to_row_end ? wrap_off : 0);
            if (!(P.dbg & 4)) store_channels<T, NB>(dst0 + off, v);
          }
          continue;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
          if (gb + row < npix) {
            int xx = xb + row, dy2 = 0;          // dy2 = 2 * (rows of the up-sampled image below dst0's)
            while (xx >= P.W) { xx -= P.W; dy2 += 2; }
            // rows past the image's last one continue in the next image: H low-res rows = Hs up-sampled rows apart
            int yy = yb + (dy2 >> 1);
            while (yy >= P.H) { yy -= P.H; dy2 += P.Hs - 2 * P.H; }
            float v[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) v[nb] = acc[mb][nb][i] + bias[nb];
            if (!(P.dbg & 4)) store_channels<T, NB>(dst0 + (dy2 * P.Ws + 2 * (xx - xb)) * P.ldy, v);
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[mb][nb][i] = 0.f;
        }
      } else {
        T* dst0 = reinterpret_cast<T*>(P.y) + P.yoff + (size_t)gb * P.ldy + ch0;
        int h4 = 4 * h;
        asm volatile("" : "+v"(h4));        // keeps the 16 row offsets out of the loop-invariant (and spilled) set
        auto emit = [&](auto full_c) {
          constexpr bool FULL = decltype(full_c)::value;       // the whole row block lies inside: no per-row predicate
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + h4;
            if (FULL || gb + row < npix) {
              float v[NB];
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) v[nb] = acc[mb][nb][i];
              store_channels<T, NB>(dst0 + row * P.ldy, v);
              if constexpr (BNR) {
                const unsigned zw[2] = {(unsigned)zq[i], (unsigned)(zq[i] >> 32)};
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                  const float z = (nb & 1) ? H16<T>::hi(zw[nb >> 1]) : H16<T>::lo(zw[nb >> 1]);
                  const float f = (float)(T)v[nb];                     // the value stored
                  const float dz = fmaf(z, bsc[nb], bsh[nb]) > 0.f ? f : 0.f;
                  ts1[nb] += dz;
                  ts2[nb] = fmaf(dz, z, ts2[nb]);
                }
              }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb][i] = 0.f;
          }
        };
        if (gb + 32 <= npix) emit(std::true_type{});
        else emit(std::false_type{});
      }
    }
  }
  if constexpr (BNR) {
    // lanes l and l ^ 32 hold the same channels; then the 8 waves through LDS (a region of its own behind the buffers:
    // trailing zero-page DMA may still be landing in them), one fp64 atomic per channel and sum into a replica
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      ts1[nb] += __shfl_xor(ts1[nb], 32, 64);
      ts2[nb] += __shfl_xor(ts2[nb], 32, 64);
    }
    float* part = reinterpret_cast<float*>(smem + NBUF * BUF);          // [NW][BN][2]
    if (h == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        part[(wid * BN + NB * j + nb) * 2 + 0] = ts1[nb];
        part[(wid * BN + NB * j + nb) * 2 + 1] = ts2[nb];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      float v = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) v += part[w8 * 2 * BN + tid];
      double vd = (double)v;
      const int c = nt * BN + (tid >> 1);
      if (tid & 1) {                                                    // sum dz z -> sum dz (z - mean) invstd
        float s1 = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) s1 += part[w8 * 2 * BN + tid - 1];
        vd = (vd - (double)P.bn_mean[c] * (double)s1) * (double)P.bn_invstd[c];
      }
      unsafeAtomicAdd(P.stats + ((size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * P.Cout + c) * 2 + (tid & 1), vd);
    }
  }
}

template <typename T, int MODE, int NW = 8, int KS = 4, int NBUF = 2, int MBS = 2, bool BNR = false>
int launch_pw(const ConvParams& P, hipStream_t s) {
  constexpr int TP = 32 * MBS * NW;
  constexpr size_t lds = NBUF * ((size_t)TP * 32 * KS + KS * 4 * 1024) + (BNR ? NW * 128 * 2 * sizeof(float) : 0);
  constexpr int WG_PER_CU = (NW == 8 && !(MBS == 1 && KS == 2)) ? 1 : (160 * 1024) / (int)lds > 2 ? 2 : (160 * 1024) / (int)lds;   // VGPRs: 8 waves per CU
  UNETK_ALLOW_LDS((pw_db_kernel<T, MODE, NW, KS, NBUF, MBS, BNR>), lds);
  ConvParams Q = P;
  Q.n_tiles = P.Cout / 128;
  const long npix = (long)P.N * P.H * P.W;
  const long ntiles = (npix + TP - 1) / TP;
  const int niter = (P.Cin / (2 * KS * ElemTraits<T>::E)) * (MODE == 2 ? 4 : 1);
  // tiles per workgroup by the round model of launch_db (whole rounds of resident workgroups)
  int tpb = 1;
  {
    double best = 1e30;
    UNETK_KNOB(tpb_max, "UNETK_CONV_TPB_MAX", 96);
    for (int t = 1; t <= tpb_max; ++t) {
      const long wgs = ((ntiles + t - 1) / t) * Q.n_tiles;
      const long rounds = (wgs + 256 * WG_PER_CU - 1) / (256 * WG_PER_CU);
      const double cost = (double)rounds * ((double)t * niter + 1.5);
      if (cost < best * 0.999) { best = cost; tpb = t; }
    }
  }
  {
    UNETK_KNOB(tpb_force, "UNETK_CONV_TPB", 0);
    if (tpb_force > 0) tpb = tpb_force;
  }
  Q.tpb = tpb;
  {
    UNETK_KNOB(dbg, "UNETK_CONV_DBG", 0);
    Q.dbg = dbg;
  }
  const long nblk = ((ntiles + tpb - 1) / tpb) * Q.n_tiles;
  if (nblk <= 0 || nblk > 0x7ffffff0L) return UNETK_ERR_SHAPE;
  Q.nblk = (int)nblk;
  hipLaunchKernelGGL((pw_db_kernel<T, MODE, NW, KS, NBUF, MBS, BNR>), dim3((unsigned)((nblk + 7) / 8 * 8)), dim3(64 * NW), lds, s, Q);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

template <typename T, int NB, int MODE> size_t lds_bytes() {
  using G = Geo<MODE>;
  size_t main_bytes = (size_t)G::NPIX * G::KS * 32 + (size_t)G::TAPS * G::KS * NB * 1024;
  size_t red_bytes = 4 * 32 * NB * 2 * sizeof(float);
  return main_bytes > red_bytes ? main_bytes : red_bytes;
}

template <typename T, int NB, int MODE> int launch(const ConvParams& P, hipStream_t s) {
  const size_t lds = lds_bytes<T, NB, MODE>();
  UNETK_ALLOW_LDS((conv_igemm_kernel<T, NB, MODE>), lds);
  // tiles per workgroup: enough K-iterations per workgroup (>= ~12 chunks) to amortise the pipeline fill, as long
  // as the grid still holds >= ~8 workgroups per CU
  ConvParams Q = P;
  const long ntiles = (long)P.N * P.tiles_y * P.tiles_x;
  using G = Geo<MODE>;
  const int niter = (P.Cin / (2 * G::KS * ElemTraits<T>::E)) * G::OUTER;
  int tpb = (12 + niter - 1) / niter;
  while (tpb > 1 && ((ntiles + tpb - 1) / tpb) * P.n_tiles < 2048) --tpb;
  {
    UNETK_KNOB(tpb_force, "UNETK_CONV_TPB", 0);
    if (tpb_force > 0) tpb = tpb_force;
  }
  Q.tpb = tpb;
  const long nblk = ((ntiles + tpb - 1) / tpb) * P.n_tiles;
  if (nblk <= 0 || nblk > 0x7ffffff0L) return UNETK_ERR_SHAPE;
  Q.nblk = (int)nblk;
  hipLaunchKernelGGL((conv_igemm_kernel<T, NB, MODE>), dim3((unsigned)((nblk + 7) / 8 * 8)), dim3(256), lds, s, Q);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

template <int MODE> int dispatch(const ConvParams& P, int dtype, hipStream_t s) {
  if constexpr (MODE == 0) {
    // conv3x3 always runs the double-buffered LDS-DMA kernel (conv3x3_db_kernel); the register-staged first generation
    // below still serves the transposed-conv modes whose GEMM-N is not a multiple of 128
    const Conv3Config cfg = conv3_config(P.Cout);
    if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
    // bf16: the 16x16x32 kernel (conv3x3_k32.hip) and its weight format; the packer asks the same function
    if (conv3_use_k32(unetk_is16(dtype), P.Cout)) return unetk_internal_conv3x3_k32(&P, dtype, s);
#ifdef UNETK_TUNING
    // A/B builds only (UNETK_CONV_K32=0/1): the 32x32x16 bf16 instantiations of conv3x3_db_kernel, superseded by the
    // 16x16x32 kernels above and not compiled into the shipped library
    if (P.aff_scale != nullptr) {
      if (dtype != UNETK_BF16) return UNETK_ERR_DTYPE;
      return cfg.NB == 4 ? launch_db<bf16, 4, 1, 8, 2, 2, true>(P, s) : launch_db<bf16, 2, 2, 8, 2, 2, true>(P, s);
    }
    if (dtype == UNETK_BF16) {
      if (P.bn_z != nullptr) return UNETK_ERR_UNFUSED;
      return cfg.NB == 4 ? launch_db<bf16, 4, 1, 8, 2, 2>(P, s) : launch_db<bf16, 2, 2, 8>(P, s);
    }
#endif
    // from here on: fp32 (exact-fp32 MFMA); the inference epilogue exists for 16-bit elements only (BASELINE.json configs[2])
    if (dtype != UNETK_F32 || P.aff_scale != nullptr) return UNETK_ERR_DTYPE;
    // 128-channel tiles: 64-byte input rows (KI = 2) refilled every second weight chunk; 64-channel tiles: 16 x 32
    // pixels, 32-channel chunks (measured slower and removed: 32-byte rows for the wide tiles -7 %, 32 x 32-pixel
    // tiles with four row blocks per wave -6 %, two 4-wave workgroups per CU -10 %)
    // the BN-backward reduction of the producer layer is folded only into the pipelined 64-channel kernel above: here
    // the epilogue is not overlapped with MFMA work, the extra read of z is exposed latency, and a fused version ran
    // exactly as long as dgrad followed by the stand-alone pass (measured at 128..1024 channels)
    if (P.bn_z != nullptr) return UNETK_ERR_UNFUSED;
    if (cfg.NB == 4) return launch_db<float, 4, 1, 8, 2, 2>(P, s);
    return launch_db<float, 2, 2, 8>(P, s);
  }
  if constexpr (MODE != 0) {
    if (pw_nb(P.Cout) == 4) {     // the packer made the same choice (conv_config.h)
      // 16-bit types: 8 waves x ONE 32-pixel row block (256-pixel tiles), 64-channel chunks, three 48-KiB buffers with two
      // chunks of DMA look-ahead.  What the ablations of rounds 2-3 say about these HBM/fill-bound GEMMs: a chunk is
      // 0.3-0.5 us of MFMA work, less than a memory round trip, so one chunk of look-ahead leaves the fill path idle
      // after every barrier (two buffers -> three: -6..-8 %); and the fill rate per CU follows the number of cache LINES
      // in flight, so a DMA piece should use whole 128-byte lines (64 channels per pixel and chunk; the 32-channel
      // chunks of the round-2 4-wave form fetched every line twice, half a line each: -1..-11 % more).  Over the four
      // decoder levels, same box: forward 1.135 -> 1.03 ms, dgrad 1.14 -> 0.975 ms.
#ifdef UNETK_TUNING
      UNETK_KNOB(small, "UNETK_PW_SMALL", 1024);
      const bool few_k = P.Cin * (MODE == 2 ? 4 : 1) <= small;
      UNETK_KNOB(var, "UNETK_PW_VAR", 2);
      if (dtype == UNETK_BF16 && var == 0) return few_k ? launch_pw<bf16, MODE, 4, 2, 3>(P, s) : launch_pw<bf16, MODE>(P, s);
      if (dtype == UNETK_BF16 && var == 1) return launch_pw<bf16, MODE, 4, 4, 3>(P, s);      // full lines, 4 waves, 1 WG/CU
      if (dtype == UNETK_BF16 && var == 3) return launch_pw<bf16, MODE, 8, 4, 2, 1>(P, s);   // two buffers
      if (dtype == UNETK_BF16 && var == 4) return launch_pw<bf16, MODE, 8, 2, 3, 1>(P, s);   // half lines, two 8-wave WGs per CU
      if (dtype == UNETK_BF16 && var == 5) return few_k ? launch_pw<bf16, MODE, 4, 2>(P, s) : launch_pw<bf16, MODE>(P, s);  // round 2
#endif
      if constexpr (MODE == 2) {
        if (P.bn_z != nullptr) {      // BN-backward reduction of the layer below in the epilogue (16-bit types)
          if (dtype == UNETK_BF16) return launch_pw<bf16, MODE, 8, 4, 3, 1, true>(P, s);
          if (dtype == UNETK_F16) return launch_pw<f16, MODE, 8, 4, 3, 1, true>(P, s);
          return UNETK_ERR_UNFUSED;
        }
      }
      if (dtype == UNETK_BF16) return launch_pw<bf16, MODE, 8, 4, 3, 1>(P, s);
      if (dtype == UNETK_F16) return launch_pw<f16, MODE, 8, 4, 3, 1>(P, s);
      if (dtype == UNETK_F32) return launch_pw<float, MODE>(P, s);
      return UNETK_ERR_DTYPE;
    }
  }
  if constexpr (MODE != 0) {       // the first-generation kernel serves only the transposed-conv modes
    if (P.bn_z != nullptr) return UNETK_ERR_UNFUSED;
    if (dtype == UNETK_BF16) return launch<bf16, 2, MODE>(P, s);
    if (dtype == UNETK_F16) return launch<f16, 2, MODE>(P, s);
    if (dtype == UNETK_F32) return launch<float, 2, MODE>(P, s);
  }
  return UNETK_ERR_DTYPE;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_common(const void* x, int ldx, int xoff, const void* y, int ldy, int yoff, int cin, int cout, int dtype,
                 int chunk_bytes) {
  if (!unetk_dtype_ok(dtype)) return UNETK_ERR_DTYPE;
  const int es = unetk_esize(dtype);
  const int ch = chunk_bytes / es;
  if (cin <= 0 || cout <= 0 || cin % ch != 0 || cout % 64 != 0) return UNETK_ERR_SHAPE;
  if (ldx < xoff + cin || ldy < yoff + cout || xoff < 0 || yoff < 0) return UNETK_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(y) || (ldx * es) % 16 || (xoff * es) % 16 || (ldy * es) % 16 || (yoff * es) % 16)
    return UNETK_ERR_ALIGN;
  return UNETK_OK;
}

struct BnReduceArgs {       // ConvParams::bn_z and friends
  const void* z;
  int ldz;
  const float *scale, *shift, *mean, *invstd;
};

int conv3x3_common(const void* x, int ldx, int xoff, const void* wpk, void* y, int ldy, int yoff, double* stats,
                   int n, int h, int w, int cin, int cout, int dtype, void* stream, const float* aff_scale = nullptr,
                   const float* aff_shift = nullptr, void* pool = nullptr, int ldp = 0,
                   const BnReduceArgs* bnr = nullptr) {
  if (n <= 0 || h <= 0 || w <= 0 || !x || !wpk || !y) return UNETK_ERR_ARG;
  if (int e = check_common(x, ldx, xoff, y, ldy, yoff, cin, cout, dtype, 64)) return e;
  if ((long)h * w * (ldx > ldy ? ldx : ldy) >= 0x7fffffffL) return UNETK_ERR_SHAPE;   // int32 in-image offsets
  ConvParams P{};
  P.x = x; P.wpk = wpk; P.y = y; P.stats = stats; P.aff_scale = aff_scale; P.aff_shift = aff_shift;
  P.pool = pool; P.ldp = ldp;
  if (bnr != nullptr) {
    const int es = unetk_esize(dtype);
    if (!bnr->z || !bnr->scale || !bnr->shift || !bnr->mean || !bnr->invstd || !stats) return UNETK_ERR_ARG;
    if (bnr->ldz < cout || (long)h * w * bnr->ldz >= 0x7fffffffL) return UNETK_ERR_SHAPE;
    if (!aligned16(bnr->z) || (bnr->ldz * es) % 16) return UNETK_ERR_ALIGN;
    P.bn_z = bnr->z; P.ld_bnz = bnr->ldz;
    P.bn_scale = bnr->scale; P.bn_shift = bnr->shift; P.bn_mean = bnr->mean; P.bn_invstd = bnr->invstd;
  }
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

int unetk_conv3x3_fprop_bn_relu(const void* x, int ldx, int x_choff, const void* wpk, const float* scale,
                                const float* shift, void* y, int ldy, int y_choff, void* pool, int ldp, int n, int h,
                                int w, int c_in, int c_out, int dtype, void* stream) {
  if (!scale || !shift) return UNETK_ERR_ARG;
  if (pool) {
    const int es = unetk_esize(dtype);
    if (h < 2 || w < 2 || ldp < c_out) return UNETK_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(pool) & 15) || (ldp * es) % 16) return UNETK_ERR_ALIGN;
  }
  return conv3x3_common(x, ldx, x_choff, wpk, y, ldy, y_choff, nullptr, n, h, w, c_in, c_out, dtype, stream, scale,
                        shift, pool, ldp);
}

int unetk_conv3x3_dgrad(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int dx_choff,
                        int n, int h, int w, int c_in, int c_out, int dtype, void* stream) {
  // GEMM-K = forward output channels, GEMM-N = forward input channels
  return conv3x3_common(dy, lddy, dy_choff, wpk, dx, lddx, dx_choff, nullptr, n, h, w, c_out, c_in, dtype, stream);
}

int unetk_conv3x3_dgrad_sums(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int dx_choff,
                             double* sums, int n, int h, int w, int c_in, int c_out, int dtype, void* stream) {
  return conv3x3_common(dy, lddy, dy_choff, wpk, dx, lddx, dx_choff, sums, n, h, w, c_out, c_in, dtype, stream);
}

int unetk_conv3x3_dgrad_bn_reduce_is_fused(int c_in, int c_out, int dtype) {
  // mirrors dispatch<0> / unetk_internal_conv3x3_k32: GEMM-N = c_in, GEMM-K = c_out
  if (!unetk_is16(dtype) || c_in <= 0 || c_out < 64 || c_out % 32 != 0) return 0;
  return conv3_use_k32(1, c_in) && conv3_k32_nb16(c_in) == 4 && c_in == 64 && !conv3_k32_nopipe();
}

int unetk_conv3x3_dgrad_bn_reduce(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx,
                                  int dx_choff, const void* z_prev, int ldz, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, double* sums, int n, int h, int w, int c_in,
                                  int c_out, int dtype, void* stream) {
  const BnReduceArgs bnr{z_prev, ldz, scale, shift, mean, invstd};
  const int e = conv3x3_common(dy, lddy, dy_choff, wpk, dx, lddx, dx_choff, sums, n, h, w, c_out, c_in, dtype, stream,
                               nullptr, nullptr, nullptr, 0, &bnr);
  if (e != UNETK_ERR_UNFUSED) return e;
  // the kernel serving this shape has no such epilogue: plain dgrad, then the stand-alone reduction pass
  if (int e2 = conv3x3_common(dy, lddy, dy_choff, wpk, dx, lddx, dx_choff, nullptr, n, h, w, c_out, c_in, dtype, stream))
    return e2;
  const int es = unetk_esize(dtype);
  return unetk_bn_relu_bwd_reduce(static_cast<const char*>(dx) + (size_t)dx_choff * es, lddx, z_prev, ldz, scale, shift,
                                  mean, invstd, sums, (int64_t)n * h * w, c_in, dtype, stream);
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

namespace {
int convT_dgrad_common(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int n, int h, int w,
                       int c_in, int c_out, int hs, int ws, int pad_t, int pad_l, int dtype, void* stream,
                       const BnReduceArgs* bnr, double* sums);
}

int unetk_convT2x2_dgrad(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int n, int h,
                         int w, int c_in, int c_out, int hs, int ws, int pad_t, int pad_l, int dtype, void* stream) {
  return convT_dgrad_common(dy, lddy, dy_choff, wpk, dx, lddx, n, h, w, c_in, c_out, hs, ws, pad_t, pad_l, dtype, stream,
                            nullptr, nullptr);
}

int unetk_convT2x2_dgrad_bn_reduce_is_fused(int c_in, int c_out, int dtype) {
  // mirrors dispatch<2>: the 16-bit pointwise kernel serves GEMM-N (= c_in) in multiples of 128
  return unetk_is16(dtype) && c_in > 0 && c_out > 0 && c_out % 64 == 0 && pw_nb(c_in) == 4;
}

int unetk_convT2x2_dgrad_bn_reduce(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx,
                                   const void* z_prev, int ldz, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, double* sums, int n, int h, int w, int c_in,
                                   int c_out, int hs, int ws, int pad_t, int pad_l, int dtype, void* stream) {
  const BnReduceArgs bnr{z_prev, ldz, scale, shift, mean, invstd};
  const int e = convT_dgrad_common(dy, lddy, dy_choff, wpk, dx, lddx, n, h, w, c_in, c_out, hs, ws, pad_t, pad_l, dtype,
                                   stream, &bnr, sums);
  if (e != UNETK_ERR_UNFUSED) return e;
  // no such epilogue for this shape / dtype: plain dgrad, then the stand-alone reduction pass
  if (int e2 = convT_dgrad_common(dy, lddy, dy_choff, wpk, dx, lddx, n, h, w, c_in, c_out, hs, ws, pad_t, pad_l, dtype,
                                  stream, nullptr, nullptr))
    return e2;
  return unetk_bn_relu_bwd_reduce(dx, lddx, z_prev, ldz, scale, shift, mean, invstd, sums, (int64_t)n * h * w, c_in,
                                  dtype, stream);
}

}  // extern "C"

namespace {
int convT_dgrad_common(const void* dy, int lddy, int dy_choff, const void* wpk, void* dx, int lddx, int n, int h, int w,
                       int c_in, int c_out, int hs, int ws, int pad_t, int pad_l, int dtype, void* stream,
                       const BnReduceArgs* bnr, double* sums) {
  if (n <= 0 || h <= 0 || w <= 0 || !dy || !wpk || !dx) return UNETK_ERR_ARG;
  // GEMM-K per tap = c_out (channels of the up-sampled gradient), GEMM-N = c_in
  if (int e = check_common(dy, lddy, dy_choff, dx, lddx, 0, c_out, c_in, dtype, 128)) return e;
  if (pad_t < 0 || pad_l < 0 || pad_t + 2 * h > hs || pad_l + 2 * w > ws) return UNETK_ERR_SHAPE;
  if ((long)hs * ws * lddy >= 0x7fffffffL || (long)h * w * lddx >= 0x7fffffffL) return UNETK_ERR_SHAPE;
  ConvParams P{};
  P.x = dy; P.wpk = wpk; P.y = dx;
  if (bnr != nullptr) {
    const int es = unetk_esize(dtype);
    if (!bnr->z || !bnr->scale || !bnr->shift || !bnr->mean || !bnr->invstd || !sums) return UNETK_ERR_ARG;
    if (bnr->ldz < c_in) return UNETK_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(bnr->z) & 15) || (bnr->ldz * es) % 16) return UNETK_ERR_ALIGN;
    P.bn_z = bnr->z; P.ld_bnz = bnr->ldz; P.stats = sums;
    P.bn_scale = bnr->scale; P.bn_shift = bnr->shift; P.bn_mean = bnr->mean; P.bn_invstd = bnr->invstd;
  }
  P.N = n; P.H = h; P.W = w; P.Cin = c_out; P.Cout = c_in;
  P.ldx = lddy; P.ldy = lddx; P.xoff = dy_choff; P.yoff = 0;
  P.Hs = hs; P.Ws = ws; P.padT = pad_t; P.padL = pad_l;
  P.tiles_x = host_cdiv(w, 32); P.tiles_y = host_cdiv(h, 8); P.n_tiles = c_in / 64;
  return dispatch<2>(P, dtype, (hipStream_t)stream);
}
}  // namespace
