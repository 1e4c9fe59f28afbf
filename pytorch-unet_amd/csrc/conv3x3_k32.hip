// conv3x3 (pad 1, stride 1; fprop, and dgrad with rotated weights) in bf16 on v_mfma_f32_16x16x32_bf16.
//   unet_parts.py:16,19 (aten::conv2d) and its input gradient.
//
// Why this shape.  One 16x16x32 instruction covers K = 32 = the 32 input channels of ONE tap that an LDS pixel row
// (64 B) holds; for the same output tile per wave it moves the same LDS bytes and takes the same matrix-pipe cycles
// as the 32x32x16 form, but writes half as many accumulator values per FLOP and the chip holds a higher clock on it
// (MI355X_MICROARCH.md, DVFS give-back item 7; measured here: +7..9 % on every layer shape).  Its C layout also puts
// the GEMM-N index on 16 lanes, so with 8 column blocks a lane owns 8 ADJACENT channels of a pixel: 16-byte stores.
//
//   GEMM view   M = output pixels, N = output channels, K = 9 taps x input channels
//   tile        one 512-thread workgroup (8 waves) per CU: 16 x 32 output pixels x BN = 16*NB16 channels
//               (NB16 = 8: 128 channels; NB16 = 4: 64 channels).  Wave w owns tile rows 2w, 2w+1 = four 16-pixel
//               A fragments (row mb, pixel set ps) x NB16 column blocks -> acc[2][2][NB16] of 4 registers.
//   LDS         2 x input halo tile (18 x 34 pixels x 64 B = 32 channels, 39 KiB)  +  2 x weight stage
//               (NB16 = 8: taps 0-4 / taps 5-8 of the 32-channel chunk, 40 KiB; NB16 = 4: all 9 taps, 36 KiB)
//   loop        per weight stage: LDS-DMA of the next stage (+ its share of the next input chunk), spread between the
//               k-steps -> MFMAs of this stage -> s_waitcnt vmcnt(0) + ONE barrier.  The chunk stream runs on across
//               the tiles a workgroup walks.  64 -> 64 layers (two stages in all): both stages stay resident in the
//               two weight buffers for the whole workgroup (WRES) -- half the LDS fill traffic.
//
// A operand.  Lane (r = l & 15, g = l >> 4) holds row r, k-group g: it reads the 16-byte piece g (8 channels) of
// pixel pi(r) + 4 ps + tap shift.  Rows map to pixels as pi = {0-3, 16-19, 24-27, 8-11}: a ds_read_b128 is served in
// the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ..., i.e. rows {0-3,12-15} of one k-group with rows {4-11}
// of the next; with pi those are pixels {P..P+3, P+8..P+11} on piece a and {P+16.., P+24..} on piece a^1 -- bank rows
// Q, Q+2, Q+4, Q+6 for every pixel phase, and the image's XOR swizzle (piece ^ (bank row & 3)) then gives 16 distinct
// 16-byte slots for every tap shift (the natural map pi(r) = r is 2-way conflicted for every swizzle of this form).
// C layout: column = l & 15 -> channel NB16*c + cb; row = 4*(l >> 4) + reg -> pixel {0,16,24,8}[l >> 4] + reg + 4 ps.
#include <type_traits>

#ifndef UNETK_K32_SCHED
#define UNETK_K32_SCHED 1
#endif
#ifndef UNETK_K32_CBS
#define UNETK_K32_CBS 4
#endif
#ifndef UNETK_K32_ABL
#define UNETK_K32_ABL 0      // perf ablations only: 1 no statistics arithmetic, 2 no stores
#endif
#ifndef UNETK_K32_DSTEPS
#define UNETK_K32_DSTEPS 0   /* k-steps of a tile's first stage that carry its DMA pieces (0: 4) */
#endif
#ifndef UNETK_K32_BNR_DEFER
#define UNETK_K32_BNR_DEFER 8
#endif
#ifndef UNETK_K32_FIRSTC
#define UNETK_K32_FIRSTC 0
#endif
#ifndef UNETK_K32_LDSSTATS
#define UNETK_K32_LDSSTATS 0
#endif

#include "conv_common.h"
#include "conv_config.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(4))) unsigned frag_t;      // one 16-byte MFMA operand fragment

__device__ uint4 g_zero_page_k32[8];

typedef unetk_u32x4 u32x4;      // bufdma16 (common.h): LDS-DMA through a buffer descriptor, out-of-range lanes zero-filled

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N-1 (the loop index feeds constexpr contexts)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <typename H, int NB16> __device__ __forceinline__ void store_adjacent(H* p, const float (&v)[NB16]) {
  typedef typename H16<H>::x8 h8;
  typedef typename H16<H>::x4 h4;
  if constexpr (NB16 == 8) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    h8 o = {(H)v[0], (H)v[1], (H)v[2], (H)v[3], (H)v[4], (H)v[5], (H)v[6], (H)v[7]};
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(p));
  } else {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    h4 o = {(H)v[0], (H)v[1], (H)v[2], (H)v[3]};
    __builtin_nontemporal_store(__builtin_bit_cast(u32x2, o), reinterpret_cast<u32x2*>(p));
  }
}

template <typename H, int NB16, bool AFF, bool WRES>
__global__ __launch_bounds__(512, 2) void conv3x3_k32_kernel(ConvParams P) {
  typedef typename H16<H>::x2 h2;
  typedef typename H16<H>::x8 h8;
  constexpr int NW = 8, MB = 2, TR = MB * NW, HP = 34, S = 64, NP = 4, TAPS = 9;
  constexpr int NPIX = (TR + 2) * HP;
  constexpr int IN_INSTR = (NPIX * NP + 63) / 64;            // 39 one-KiB DMA pieces
  constexpr int IN_BYTES = IN_INSTR * 1024;
  constexpr int BN = 16 * NB16;
  constexpr int NSTAGE = NB16 == 8 ? 2 : 1;                  // weight stages per 32-channel input chunk
  constexpr int STEPS0 = NB16 == 8 ? 5 : 9;                  // taps of stage 0 (stage 1: the rest)
  constexpr int W_BYTES = STEPS0 * NB16 * 1024;              // a weight buffer holds the larger stage
  constexpr int CBS = UNETK_K32_CBS;                         // column blocks per sub-step
  constexpr int HN = NB16 / CBS;                             // sub-steps per k-step
  constexpr int IN_PER_WAVE = (IN_INSTR + NW - 1) / NW;      // 5
  // LDS: [input 0][input 1][weights 0][weights 1]

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;

  const int nt = blockIdx.x % P.n_tiles;
  const int grp = blockIdx.x / P.n_tiles;
  const int tiles_per_img = P.tiles_x * P.tiles_y;
  const int tile_begin = grp * P.tpb;
  const int tile_end = min(tile_begin + P.tpb, P.N * tiles_per_img);
  if (tile_begin >= tile_end) return;

  const int nin = P.Cin / 32;                                // 32-channel input chunks
  const char* wbase = reinterpret_cast<const char*>(P.wpk) + (size_t)nt * nin * (TAPS * NB16 * 1024);

  // A-fragment addresses.  This lane's row reads halo pixel p = p00 + rr*HP + col (rr = mb + tap/3 in 0..3, col =
  // 4 ps + tap%3 in 0..6) at byte (p << 6) + (swz << 4), swz = ((p >> 2) ^ g) & 3.  The linear part is the lane's base
  // plus a compile-time offset (the ds_read offset field); the 2-bit swizzles of all 4 x 7 pixels are packed into two
  // registers (field rr*8 + col), so a read costs one v_bfe + one v_lshl_add instead of six VALU instructions --
  // a 16x16x32 MFMA leaves only two VALU issue slots in its shadow.
  const int pi = (c16 & 3) + ((c16 >> 2) == 1 ? 16 : (c16 >> 2) == 2 ? 24 : (c16 >> 2) == 3 ? 8 : 0);
  const int p00 = (MB * wid) * HP + pi;
  unsigned swz_lo = 0, swz_hi = 0;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int col = 0; col < 7; ++col) {
      const unsigned v = (unsigned)((((p00 + rr * HP + col) >> 2) ^ g4) & 3);
      if (rr < 2) swz_lo |= v << (2 * (rr * 8 + col));
      else swz_hi |= v << (2 * ((rr - 2) * 8 + col));
    }

  // LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): a lane whose offset lies outside the
  // descriptor's range gets ZEROS written to its LDS slot (scripts/microbench/buf_lds_oob.hip), so padding pixels and
  // ragged edges are the offset ~0u -- no zero page, no per-lane 64-bit source select, no VALU work per piece -- and a
  // whole dead stream (past the workgroup's last tile) is a descriptor of zero bytes.
  //   input:   base = the staged tile's image (+ channel offset), range = to the image's end; voffset = this lane's
  //            pixel/piece (chunk 0), soffset = 64 B per 32-channel chunk
  //   weights: base = this n-tile's fragment stream; voffset = lane * 16, soffset = fragment * 1 KiB
  unsigned dma_off[IN_PER_WAVE];
  u32x4 xdesc = {0u, 0u, 0u, 0x00020000u};
  const unsigned img_bytes = (unsigned)(((size_t)P.H * P.W * P.ldx - P.xoff) * 2);
  auto setup_stage = [&](int tile) {
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * TR, x0 = (rem % P.tiles_x) * 32;
    const unsigned long long xa =
        (unsigned long long)(reinterpret_cast<const H*>(P.x) + (size_t)img * P.H * P.W * P.ldx + P.xoff);
    xdesc[0] = __builtin_amdgcn_readfirstlane((unsigned)xa);
    xdesc[1] = __builtin_amdgcn_readfirstlane((unsigned)(xa >> 32) & 0xffffu);
    xdesc[2] = img_bytes;
#pragma unroll
    for (int f = 0; f < IN_PER_WAVE; ++f) {
      const int fi = min(f * NW + wid, IN_INSTR - 1);   // surplus slots of the issue grid repeat the last piece
      const int id = fi * 64 + lane;                     // LDS slot: pixel p, physical piece sl
      const int p = id / NP, sl = id % NP;
      const int c = sl ^ ((p >> 2) & (NP - 1));          // logical piece stored in that slot
      const int hr = p / HP, hc = p - hr * HP;
      const int yy = y0 + hr - 1, xx = x0 + hc - 1;
      unsigned off = ~0u;
      if (p < NPIX && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) off = ((unsigned)(yy * P.W + xx) * P.ldx + c * 8) * 2u;
      dma_off[f] = off;
    }
  };
  u32x4 wdesc;
  {
    const unsigned long long wa = (unsigned long long)wbase;
    wdesc[0] = __builtin_amdgcn_readfirstlane((unsigned)wa);
    wdesc[1] = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
    wdesc[2] = (unsigned)nin * (TAPS * NB16 * 1024);
    wdesc[3] = 0x00020000u;
  }
  const unsigned wlane = lane * 16;
  auto issue_input = [&](int q, int in_chunk, unsigned buf, bool live) {
    const int fi = min(q * NW + wid, IN_INSTR - 1);
    u32x4 d = xdesc;
    d[2] = live ? d[2] : 0u;
    bufdma16(dma_off[q], d, (unsigned)in_chunk * 64u, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
  };
  // weight fragment q of the stage that starts at fragment `frag_base` of the n-tile's stream and has `nfrag` fragments
  auto issue_weight = [&](int q, unsigned frag_base, bool live, int nfrag, unsigned buf) {
    const int fi = min(q * NW + wid, nfrag - 1);
    u32x4 d = wdesc;
    d[2] = live ? d[2] : 0u;
    bufdma16(wlane, d, (frag_base + fi) * 1024u, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
  };

  f32x4v acc[MB][2][NB16];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int cb = 0; cb < NB16; ++cb) acc[mb][ps][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};

  const int ch0 = nt * BN + NB16 * c16;                      // first of this lane's NB16 adjacent channels

  setup_stage(tile_begin);
  const unsigned lds0 = lds_address(smem);
  const unsigned ldsw0 = lds0 + 2 * IN_BYTES;
#pragma unroll
  for (int q = 0; q < IN_PER_WAVE; ++q) issue_input(q, 0, lds0, true);
  if constexpr (WRES) {
    // every stage of the layer fits the two weight buffers (<= 2 stages): loaded once, never refilled
    const int nst = nin * NSTAGE;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (st < nst) {
        const int chunk = st / NSTAGE, s_in = st % NSTAGE;
        const int frag0 = s_in == 0 ? 0 : STEPS0 * NB16, nfrag = (s_in == 0 ? STEPS0 : TAPS - STEPS0) * NB16;
        for (int q = 0; q * NW < nfrag; ++q)
          issue_weight(q, (unsigned)(chunk * (TAPS * NB16) + frag0), true, nfrag, ldsw0 + st * W_BYTES);
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q * NW < STEPS0 * NB16; ++q) issue_weight(q, 0u, true, STEPS0 * NB16, ldsw0);
  }
  // BN statistics of all tiles of this workgroup: per tile in registers; across tiles in registers too, or (LDSSTATS)
  // in an LDS array [BN][2] behind the staging buffers by ds_add_f32
  constexpr bool LDSSTATS = UNETK_K32_LDSSTATS != 0;
  float* red = reinterpret_cast<float*>(smem + 2 * IN_BYTES + 2 * W_BYTES);
  float tsum[LDSSTATS ? 1 : NB16], tsq[LDSSTATS ? 1 : NB16];
  if constexpr (LDSSTATS) {
    if (tid < 2 * BN) red[tid] = 0.f;      // visible to every wave after the prologue barrier below
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) tsum[cb] = tsq[cb] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  int curi = 0, curw = 0;

  float asc[NB16], ash[NB16];
  if constexpr (AFF) {
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) { asc[cb] = P.aff_scale[ch0 + cb]; ash[cb] = P.aff_shift[ch0 + cb]; }
  }

  // ONE fragment register set (4 A + NB16 B fragments), each fragment re-read right after the MFMA that used it last;
  // the k-step stream runs on across stages, chunks and tiles.  The stage barrier sits INSIDE a stage's last k-step,
  // behind the MFMAs of its first A fragment: every wave then holds that step's fragments in registers (it is done
  // reading the current buffers), and the first fragments of the next stage are read under the remaining MFMAs of the
  // step like those of any other step -- no LDS latency is exposed behind a barrier.
  const AS3 char* lds3 = reinterpret_cast<const AS3 char*>(0);
  frag_t fa[MB][2], fb[NB16];
  auto read_a = [&](int mb, int ps, int t, int a_base) {     // A fragment (mb, ps) of tap t from the input buffer at a_base
    const int rr = mb + t / 3, col = 4 * ps + (t % 3);
    // (asm volatile: as plain arithmetic the 28 swizzle terms are loop invariants -- hoisted, then spilled)
    int addr;
    asm volatile("v_bfe_u32 %0, %1, %2, 2\n\tv_lshl_add_u32 %0, %0, 4, %3"
                 : "=&v"(addr)
                 : "v"(rr < 2 ? swz_lo : swz_hi), "n"(2 * ((rr & 1) * 8 + col)), "v"(a_base));
    fa[mb][ps] = *reinterpret_cast<const AS3 frag_t*>(lds3 + addr + (rr * HP + col) * 64);
  };
  {
    const int a_base = (int)lds0 + (p00 << 6), w_base = (int)ldsw0 + lane * 16;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) read_a(mb, ps, 0, a_base);
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) fb[cb] = *reinterpret_cast<const AS3 frag_t*>(lds3 + w_base + cb * 1024);
  }

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    for (int ic = 0; ic < nin; ++ic) {
      // the NEXT input chunk (of this tile, or the first of the next tile) is staged while this one is consumed
      int next_in = ic + 1;
      bool live_in = true;
      if (ic + 1 == nin) {
        next_in = 0;
        if (tile + 1 < tile_end) setup_stage(tile + 1);
        else live_in = false;
      }
      const unsigned nbuf_in = lds0 + (curi ^ 1) * IN_BYTES;
      auto run_stage = [&](auto st_c, auto first_c) {
        constexpr int st = decltype(st_c)::value;
        constexpr bool FIRSTC = decltype(first_c)::value;    // first chunk of a tile: its first k-step starts from C = 0
        constexpr int tap0 = st == 0 ? 0 : STEPS0;
        constexpr int nsteps = st == 0 ? STEPS0 : TAPS - STEPS0;
        // next weight stage: (ic, st + 1), or stage 0 of the next chunk / of the next tile's first chunk
        constexpr int nst_in = (st + 1) % NSTAGE;
        constexpr int tap0_next = nst_in == 0 ? 0 : STEPS0;
        constexpr int nfrag_next = (nst_in == 0 ? STEPS0 : TAPS - STEPS0) * NB16;
        constexpr int frag0_next = nst_in == 0 ? 0 : STEPS0 * NB16;
        constexpr int w_pieces = WRES ? 0 : (nfrag_next + NW - 1) / NW;
        int next_wc = ic;
        bool live_w = true;
        if (st + 1 == NSTAGE) {
          next_wc = ic + 1;
          if (ic + 1 == nin) {
            next_wc = 0;
            live_w = live_w && (tile + 1 < tile_end);
          }
        }
        const int widx = WRES ? ((ic * NSTAGE + st) & 1) : curw;
        const int widx_next = WRES ? ((next_wc * NSTAGE + nst_in) & 1) : (curw ^ 1);
        const unsigned nbuf_w = ldsw0 + (curw ^ 1) * W_BYTES;
        // input pieces of this stage: IN_PER_WAVE split over the NSTAGE stages of the chunk
        constexpr int IN_PER_ST = (IN_PER_WAVE + NSTAGE - 1) / NSTAGE;
        constexpr int in_q0 = st * IN_PER_ST;
        constexpr int in_pieces = (st + 1 == NSTAGE) ? IN_PER_WAVE - in_q0 : IN_PER_ST;
        constexpr int ndma = in_pieces + w_pieces;
        constexpr int dsteps = nsteps - 1;                     // the last k-step carries the barrier instead
        constexpr int pps = (ndma + dsteps - 1) / dsteps;      // DMA pieces per k-step (<= 3: one behind each of the
        static_assert(pps <= 3, "one DMA piece behind each of the first three A fragments of a k-step");
        const unsigned wfrag = (unsigned)(next_wc * (TAPS * NB16) + frag0_next);
        auto issue_piece = [&](int q) {
          if (q < in_pieces) issue_input(in_q0 + q, next_in, nbuf_in, live_in);
          else issue_weight(q - in_pieces, wfrag, live_w, nfrag_next, nbuf_w);
        };
        // LDS byte offsets of this stage's and the next stage's buffers, opaque to the optimiser: the fragment reads
        // then keep their compile-time parts (tap / column-block offsets < 64 KiB) in the instruction's offset field
        int a_base = (int)lds0 + curi * IN_BYTES + (p00 << 6);
        int a_next = (int)lds0 + ((st + 1 == NSTAGE) ? (curi ^ 1) : curi) * IN_BYTES + (p00 << 6);
        int w_base = (int)ldsw0 + widx * W_BYTES + lane * 16;
        int w_next = (int)ldsw0 + widx_next * W_BYTES + lane * 16;
        asm volatile("" : "+v"(a_base), "+v"(a_next), "+v"(w_base), "+v"(w_next));
        static_for<0, nsteps>([&](auto s_c) {
          constexpr int s = decltype(s_c)::value;
          constexpr bool last = s + 1 == nsteps;
          constexpr int tn = last ? tap0_next : tap0 + s + 1;  // tap and buffer slot of the k-step after this one
          constexpr int un = last ? 0 : s + 1;
          const int ab = last ? a_next : a_base, wb = last ? w_next : w_base;
#pragma unroll
          for (int ag = 0; ag < 2 * MB; ++ag) {
            const int mb = ag >> 1, ps = ag & 1;
#pragma unroll
            for (int cb = 0; cb < NB16; ++cb) {
              const f32x4v c_in = (FIRSTC && st == 0 && s == 0) ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[mb][ps][cb];
              acc[mb][ps][cb] =
                  H16<H>::mfma16(__builtin_bit_cast(h8, fa[mb][ps]), __builtin_bit_cast(h8, fb[cb]), c_in);
              if (ag == 2 * MB - 1) {
                __builtin_amdgcn_sched_barrier(0);
                fb[cb] = *reinterpret_cast<const AS3 frag_t*>(lds3 + wb + (un * NB16 + cb) * 1024);
                __builtin_amdgcn_sched_barrier(0);
              } else if (cb == NB16 / 2 - 1 && !last && ag < pps && s * pps + ag < ndma) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(s * pps + ag);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ag == 0 && last) {
              // next stage landed (this wave's DMAs), every wave holds its last fragments of the current buffers
              asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            read_a(mb, ps, tn, ab);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        curw ^= 1;
      };
#if UNETK_K32_FIRSTC
      if (ic == 0) run_stage(std::integral_constant<int, 0>{}, std::true_type{});
      else
#endif
        run_stage(std::integral_constant<int, 0>{}, std::false_type{});
      if constexpr (NSTAGE == 2) run_stage(std::integral_constant<int, 1>{}, std::false_type{});
      curi ^= 1;
    }

    // ---- epilogue of this tile: a lane holds, per (row mb, pixel set ps), 4 consecutive pixels x NB16 adjacent channels
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * TR, x0 = (rem % P.tiles_x) * 32;
    H* yimg = reinterpret_cast<H*>(P.y) + (size_t)img * P.H * P.W * P.ldy + P.yoff;
    const int xg = g4 == 1 ? 16 : g4 == 2 ? 24 : g4 == 3 ? 8 : 0;     // pixel of C row 4*g4 within the 32-pixel row
    const int xl = x0 + xg;
    const int lane_off = xg * P.ldy + ch0;
    float ssum[NB16], ssq[NB16];
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) ssum[cb] = ssq[cb] = 0.f;
    if constexpr (AFF) {
      // the encoder's MaxPool2d(2) (unet_parts.py:34) of the activation about to be stored: the wave owns tile rows
      // 2w, 2w+1 and a lane holds 4 horizontally consecutive pixels (first one at a multiple of 4) -- every complete
      // 2x2 window is four registers of one lane.  max commutes with the (monotone) rounding of the store.
      if (P.pool != nullptr) {
        const int yy = y0 + MB * wid;                      // even; the window needs row yy + 1 too
        H* pimg = reinterpret_cast<H*>(P.pool) + (size_t)img * (P.H / 2) * (P.W / 2) * P.ldp;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
          for (int i = 0; i < 4; i += 2) {
            const int xx = xl + 4 * ps + i;                // even
            if (yy + 1 < P.H && xx + 1 < P.W) {
              float pv[NB16];
#pragma unroll
              for (int cb = 0; cb < NB16; ++cb) {
                const float a = fmaxf(fmaf(acc[0][ps][cb][i], asc[cb], ash[cb]), 0.f);
                const float b = fmaxf(fmaf(acc[0][ps][cb][i + 1], asc[cb], ash[cb]), 0.f);
                const float c = fmaxf(fmaf(acc[1][ps][cb][i], asc[cb], ash[cb]), 0.f);
                const float d = fmaxf(fmaf(acc[1][ps][cb][i + 1], asc[cb], ash[cb]), 0.f);
                pv[cb] = fmaxf(fmaxf(a, b), fmaxf(c, d));
              }
              store_adjacent<H, NB16>(pimg + ((size_t)(yy >> 1) * (P.W / 2) + (xx >> 1)) * P.ldp + ch0, pv);
            }
          }
      }
    }
    bool done = false;
    if constexpr (!AFF) {
      // Interior tiles (no pixel outside the image): no per-element predicates; the accumulators are rounded ONCE into
      // packed pairs of horizontally adjacent pixels, the BatchNorm statistics come from the packed values with
      // v_dot2c_f32_bf16 (sum: y.1, sum of squares: y.y -- exact bf16 products accumulated in fp32, two pixels per
      // instruction) and a pixel's NB16 adjacent channels are gathered with v_perm into one 16-byte (8-byte) store.
      if ((y0 + TR <= P.H) && (x0 + 32 <= P.W)) {
        auto fast = [&](auto stats_c) {
          constexpr bool ST = decltype(stats_c)::value;
          const h2 one2 = {(H)1.0f, (H)1.0f};
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int yy = y0 + MB * wid + mb;
            H* rowp = yimg + (size_t)(yy * P.W + x0) * P.ldy + lane_off;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps)
#pragma unroll
              for (int i = 0; i < 4; i += 2) {
                unsigned pk[NB16];
#pragma unroll
                for (int cb = 0; cb < NB16; ++cb) {
                  const h2 y2 = {(H)acc[mb][ps][cb][i], (H)acc[mb][ps][cb][i + 1]};
                  pk[cb] = __builtin_bit_cast(unsigned, y2);
                  if (!UNETK_K32_FIRSTC) acc[mb][ps][cb][i] = acc[mb][ps][cb][i + 1] = 0.f;
                  if constexpr (ST && !LDSSTATS) {    // straight into the workgroup-long sums (fp32, as the 64-channel kernel)
                    tsum[cb] = H16<H>::dot2(y2, one2, tsum[cb]);
                    tsq[cb] = H16<H>::dot2(y2, y2, tsq[cb]);
                  } else if constexpr (ST) {
                    ssum[cb] = H16<H>::dot2(y2, one2, ssum[cb]);
                    ssq[cb] = H16<H>::dot2(y2, y2, ssq[cb]);
                  }
                }
                typedef __attribute__((ext_vector_type(NB16 / 2))) unsigned uvec;
                uvec p0, p1;
#pragma unroll
                for (int k = 0; k < NB16 / 2; ++k) {
                  p0[k] = __builtin_amdgcn_perm(pk[2 * k + 1], pk[2 * k], 0x05040100u);
                  p1[k] = __builtin_amdgcn_perm(pk[2 * k + 1], pk[2 * k], 0x07060302u);
                }
                __builtin_nontemporal_store(p0, reinterpret_cast<uvec*>(rowp + (4 * ps + i) * P.ldy));
                __builtin_nontemporal_store(p1, reinterpret_cast<uvec*>(rowp + (4 * ps + i + 1) * P.ldy));
                // (fence: the scheduler otherwise rounds the whole tile up front and the allocator spills)
                __builtin_amdgcn_sched_barrier(0);
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
      H* rowp = yimg + (size_t)(yy * P.W + x0) * P.ldy;   // 64-bit arithmetic once per row (wave-uniform)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ri = 4 * ps + i;
          if (yy < P.H && xl + ri < P.W) {
            float v[NB16];
#pragma unroll
            for (int cb = 0; cb < NB16; ++cb) {
              if constexpr (AFF) {
                v[cb] = fmaxf(fmaf(acc[mb][ps][cb][i], asc[cb], ash[cb]), 0.f);
              } else {
                const float f = to_f32(from_f32<H>(acc[mb][ps][cb][i]));   // statistics of the value as stored
                v[cb] = f;
                ssum[cb] += f;
                ssq[cb] += f * f;
              }
            }
            store_adjacent<H, NB16>(rowp + ri * P.ldy + lane_off, v);
          }
          if (!UNETK_K32_FIRSTC) {
#pragma unroll
            for (int cb = 0; cb < NB16; ++cb) acc[mb][ps][cb][i] = 0.f;
          }
        }
    }
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) {
      if constexpr (LDSSTATS) {
        if (!AFF && P.stats != nullptr) {
          __hip_atomic_fetch_add(red + (NB16 * c16 + cb) * 2, ssum[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(red + (NB16 * c16 + cb) * 2 + 1, ssq[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        tsum[cb] += ssum[cb];
        tsq[cb] += ssq[cb];
      }
    }
  }

  // ---- one LDS reduction (register statistics) and one set of fp64 atomics per workgroup
  if (!AFF && P.stats != nullptr) {
    if constexpr (LDSSTATS) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 2 * BN) {
        double* dst = P.stats + ((size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * P.Cout + nt * BN) * 2 + tid;
        unsafeAtomicAdd(dst, (double)red[tid]);
      }
    } else {
#pragma unroll
      for (int cb = 0; cb < NB16; ++cb) {                   // lanes l, l^16, l^32, l^48 hold the same channels
        tsum[cb] += __shfl_xor(tsum[cb], 16, 64);
        tsq[cb] += __shfl_xor(tsq[cb], 16, 64);
        tsum[cb] += __shfl_xor(tsum[cb], 32, 64);
        tsq[cb] += __shfl_xor(tsq[cb], 32, 64);
      }
      float* part = reinterpret_cast<float*>(smem);         // [NW][BN][2]; every DMA landed, every read retired
      if (g4 == 0) {
#pragma unroll
        for (int cb = 0; cb < NB16; ++cb) {
          part[(wid * BN + NB16 * c16 + cb) * 2 + 0] = tsum[cb];
          part[(wid * BN + NB16 * c16 + cb) * 2 + 1] = tsq[cb];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 2 * BN) {
        float v = 0;
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) v += part[w8 * 2 * BN + tid];
        double* dst = P.stats + ((size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * P.Cout + nt * BN) * 2 + tid;
        unsafeAtomicAdd(dst, (double)v);
      }
    }
  }
}

// ================================================================================================================
// The 64-channel n-tile (GEMM-N = 64: inc.3, up4.0, up4.3 forward; inc.3, up4.3, down1.0 dgrad) with the epilogue
// taken OFF the critical path.  At K = 576 a tile is only 288 MFMAs per wave, and ablations of the kernel above showed
// the epilogue (round + statistics arithmetic + 16 stores per lane, all eight waves at once, no MFMA running) costing
// 40 % of the launch: 0.88 ms with it, 0.52 ms without (64 -> 64 at 16 x 640 x 959).  Here
//   * the tile's outputs are rounded ONCE into 32 registers of packed bf16 and the accumulators are free again;
//   * the BatchNorm statistics are taken from the packed values, two pixels per instruction: v_dot2c_f32_bf16(y, 1, sum)
//     and v_dot2c_f32_bf16(y, y, sumsq) -- exact bf16 products accumulated in fp32, 64 instructions per tile and lane
//     instead of 64 unpack + 128 add/fma, and they too ride in the shadow of the next tile's MFMAs (a first version
//     did them as MFMAs -- ones * Y and the diagonal of Y^T * Y -- which was bit-exact as well but cost 32 accumulator
//     registers, i.e. spills, and 11 % more matrix-pipe time at K = 576);
//   * the 16 stores of tile i are issued two per k-step under the MFMAs of tile i+1's first stage (EXEC-masked inline
//     asm: no branch splits the MFMA block), 8 bytes per lane, nontemporal.
//
// EPI: 0 plain store; 1 BatchNorm statistics of the output (forward); 2 (dgrad) the BN-backward sums of the layer that
// produced this tensor (ConvParams::bn_z): that layer's z at the tile's pixels is fetched during the tile's LAST stage
// and the masked sums are formed from it and the rounded values under the MFMAs of the NEXT tile's first stage.
template <typename H, bool WRES, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3_k32n64_kernel(ConvParams P) {
  typedef typename H16<H>::x2 h2;
  typedef typename H16<H>::x8 h8;
  constexpr bool STATS = EPI == 1, BNR = EPI == 2;
  constexpr int BNR_DEFER = UNETK_K32_BNR_DEFER;             // units (of 16) whose masked sums ride under the next tile
  constexpr int NB16 = 4, NW = 8, MB = 2, TR = MB * NW, HP = 34, NP = 4, TAPS = 9;
  constexpr int NPIX = (TR + 2) * HP;
  constexpr int IN_INSTR = (NPIX * NP + 63) / 64;            // 39 one-KiB DMA pieces
  constexpr int IN_BYTES = IN_INSTR * 1024;
  constexpr int BN = 64;
  constexpr int NFRAG = TAPS * NB16;                         // 36 weight fragments per 32-channel chunk
  constexpr int W_BYTES = NFRAG * 1024;
  constexpr int IN_PER_WAVE = (IN_INSTR + NW - 1) / NW;      // 5
  constexpr int W_PER_WAVE = WRES ? 0 : (NFRAG + NW - 1) / NW;
  constexpr int NDMA = IN_PER_WAVE + W_PER_WAVE;
  // Every DMA piece of a stage goes out in its first DSTEPS k-steps and the pending stores in the steps after them:
  // the stores are then the YOUNGEST vector-memory operations of the stage and `s_waitcnt vmcnt(16)` at its end waits
  // for the DMA only -- the stores complete under the next stage instead of on the barrier.
  // (stages without pending stores spread the pieces over all nine k-steps: an LDS-DMA instruction holds its wave for
  // 60-185 cycles, and in a burst all eight phase-locked waves stall together while the matrix pipe drains)
  constexpr int DSTEPS = UNETK_K32_DSTEPS ? UNETK_K32_DSTEPS : 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;

  const int tiles_per_img = P.tiles_x * P.tiles_y;
  const int tile_begin = blockIdx.x * P.tpb;                 // n_tiles == 1
  const int tile_end = min(tile_begin + P.tpb, P.N * tiles_per_img);
  if (tile_begin >= tile_end) return;

  const int nin = P.Cin / 32;
  const char* wbase = reinterpret_cast<const char*>(P.wpk);

  const int pi = (c16 & 3) + ((c16 >> 2) == 1 ? 16 : (c16 >> 2) == 2 ? 24 : (c16 >> 2) == 3 ? 8 : 0);
  const int p00 = (MB * wid) * HP + pi;
  unsigned swz_lo = 0, swz_hi = 0;                           // see conv3x3_k32_kernel
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int col = 0; col < 7; ++col) {
      const unsigned v = (unsigned)((((p00 + rr * HP + col) >> 2) ^ g4) & 3);
      if (rr < 2) swz_lo |= v << (2 * (rr * 8 + col));
      else swz_hi |= v << (2 * ((rr - 2) * 8 + col));
    }

  // LDS-DMA through buffer descriptors, see conv3x3_k32_kernel: out-of-range lanes (~0u) are zero-filled
  unsigned dma_off[IN_PER_WAVE];
  u32x4 xdesc = {0u, 0u, 0u, 0x00020000u};
  const unsigned img_bytes = (unsigned)(((size_t)P.H * P.W * P.ldx - P.xoff) * 2);
  auto setup_stage = [&](int tile) {
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * TR, x0 = (rem % P.tiles_x) * 32;
    const unsigned long long xa =
        (unsigned long long)(reinterpret_cast<const H*>(P.x) + (size_t)img * P.H * P.W * P.ldx + P.xoff);
    xdesc[0] = __builtin_amdgcn_readfirstlane((unsigned)xa);
    xdesc[1] = __builtin_amdgcn_readfirstlane((unsigned)(xa >> 32) & 0xffffu);
    xdesc[2] = img_bytes;
#pragma unroll
    for (int f = 0; f < IN_PER_WAVE; ++f) {
      const int fi = min(f * NW + wid, IN_INSTR - 1);
      const int id = fi * 64 + lane;
      const int p = id / NP, sl = id % NP;
      const int c = sl ^ ((p >> 2) & (NP - 1));
      const int hr = p / HP, hc = p - hr * HP;
      const int yy = y0 + hr - 1, xx = x0 + hc - 1;
      unsigned off = ~0u;
      if (p < NPIX && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) off = ((unsigned)(yy * P.W + xx) * P.ldx + c * 8) * 2u;
      dma_off[f] = off;
    }
  };
  u32x4 wdesc;
  {
    const unsigned long long wa = (unsigned long long)wbase;
    wdesc[0] = __builtin_amdgcn_readfirstlane((unsigned)wa);
    wdesc[1] = __builtin_amdgcn_readfirstlane((unsigned)(wa >> 32) & 0xffffu);
    wdesc[2] = (unsigned)nin * W_BYTES;
    wdesc[3] = 0x00020000u;
  }
  const unsigned wlane = lane * 16;
  auto issue_input = [&](int q, int in_chunk, unsigned buf, bool live) {
    const int fi = min(q * NW + wid, IN_INSTR - 1);
    u32x4 d = xdesc;
    d[2] = live ? d[2] : 0u;
    bufdma16(dma_off[q], d, (unsigned)in_chunk * 64u, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
  };
  auto issue_weight = [&](int q, unsigned frag_base, bool live, unsigned buf) {
    const int fi = min(q * NW + wid, NFRAG - 1);
    u32x4 d = wdesc;
    d[2] = live ? d[2] : 0u;
    bufdma16(wlane, d, (frag_base + fi) * 1024u, __builtin_amdgcn_readfirstlane(buf + fi * 1024));
  };

  f32x4v acc[MB][2][NB16];
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  unsigned pk[MB][2][NB16][2];                                // previous tile, rounded: [row][pixel set][channel] x 4 pixels
  float tsum[NB16], tsq[NB16];                                // per-lane statistics of all tiles of this workgroup
#pragma unroll
  for (int cb = 0; cb < NB16; ++cb) tsum[cb] = tsq[cb] = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int cb = 0; cb < NB16; ++cb) pk[mb][ps][cb][0] = pk[mb][ps][cb][1] = 0u;
  // where the pending stores go: row bases (wave-uniform), first pixel column of this lane, and the column limit per
  // row (0 = nothing pending / row outside the image)
  const H* st_row[MB] = {nullptr, nullptr};
  int st_wlim[MB] = {0, 0};
  int st_xl = 0;
  bool st_all16 = false;                                      // every pending store has at least one lane to write
  const int xg = g4 == 1 ? 16 : g4 == 2 ? 24 : g4 == 3 ? 8 : 0;
  const unsigned lane_off2 = (unsigned)(xg * P.ldy + NB16 * c16) * 2u;      // byte offset inside a 32-pixel row segment

  // store k = (mb, ps, i) of the pending tile: 4 adjacent channels of one pixel, 8 bytes, nontemporal, EXEC-masked
  auto emit_store = [&](int k) {
    const int mb = k >> 3, ps = (k >> 2) & 1, i = k & 3, ri = 4 * ps + i;
    const unsigned sel = (i & 1) ? 0x07060302u : 0x05040100u;
    u32x2 d;
    d[0] = __builtin_amdgcn_perm(pk[mb][ps][1][i >> 1], pk[mb][ps][0][i >> 1], sel);
    d[1] = __builtin_amdgcn_perm(pk[mb][ps][3][i >> 1], pk[mb][ps][2][i >> 1], sel);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(st_xl + ri < st_wlim[mb]);
    const unsigned off = lane_off2 + (unsigned)(ri * P.ldy) * 2u;
    unsigned long long keep;
    if (UNETK_K32_ABL & 2) {
      asm volatile("" ::"s"(m), "v"(off), "v"(d), "s"(st_row[mb]));
      return;
    }
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b64 exec, %1\n\t"
        "global_store_dwordx2 %2, %3, %4 nt\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(keep)
        : "s"(m), "v"(off), "v"(d), "s"(st_row[mb])
        : "memory");
  };

  // statistics of the pending tile, group g = (mb, ps): 4 channels x 4 pixels = 8 packed registers, 16 instructions.
  // (pixels outside the image were zeroed when the tile was packed; with nothing pending pk is all zero)
  auto emit_stats = [&](int g, int cb0 = 0, int cb1 = NB16) {
    const int mb = g >> 1, ps = g & 1;
    const h2 one2 = {(H)1.0f, (H)1.0f};
#pragma unroll
    for (int cb = cb0; cb < cb1; ++cb)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const h2 y2 = __builtin_bit_cast(h2, pk[mb][ps][cb][e]);
        tsum[cb] = H16<H>::dot2(y2, one2, tsum[cb]);
        tsq[cb] = H16<H>::dot2(y2, y2, tsq[cb]);
      }
  };

  setup_stage(tile_begin);
  const unsigned lds0 = lds_address(smem);
  const unsigned ldsw0 = lds0 + 2 * IN_BYTES;
#pragma unroll
  for (int q = 0; q < IN_PER_WAVE; ++q) issue_input(q, 0, lds0, true);
  {
    const int nst = WRES ? nin : 1;                           // WRES: every chunk's weights (<= 2), loaded once
#pragma unroll
    for (int st = 0; st < 2; ++st)
      if (st < nst)
        for (int q = 0; q * NW < NFRAG; ++q) issue_weight(q, (unsigned)(st * NFRAG), true, ldsw0 + st * W_BYTES);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int curi = 0, curw = 0;

  // BNR: z of the producer layer at the tile's pixels, [row][pixel set][pixel] x 4 adjacent channels (packed)
  unsigned zr[BNR ? MB : 1][2][4][2];
  auto fetch_z = [&](int tile) {
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * TR, x0 = (rem % P.tiles_x) * 32;
    const char* zimg = reinterpret_cast<const char*>(reinterpret_cast<const H*>(P.bn_z) + (size_t)img * P.H * P.W * P.ld_bnz);
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int yc = min(y0 + MB * wid + mb, P.H - 1);        // clamped into the image: branch-free, masked at use
      const char* zrow = zimg + (size_t)(yc * P.W) * P.ld_bnz * 2;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int xc = min(x0 + xg + 4 * ps + i, P.W - 1);
          const u32x2 r = __builtin_nontemporal_load(
              reinterpret_cast<const u32x2*>(zrow + (unsigned)(xc * P.ld_bnz + NB16 * c16) * 2u));
          zr[mb][ps][i][0] = r[0];
          zr[mb][ps][i][1] = r[1];
        }
    }
  };

  // BNR: the masked sums of the PENDING tile (its rounded values sit in pk until their stores went out, its z in zr
  // until the next fetch in this tile's last stage) are formed under the MFMAs of the next tile's first stage, one
  // (row, pixel set, pixel) unit of four channels per slot: done at the hand-over, with every wave at once and no MFMA
  // in flight, they cost 0.23 of the 0.95 ms of a 64 -> 64 launch at 16 x 640 x 959.
  // Registers: pk and zr are both live through a first stage (+32 over the hand-over version), so the lane keeps only
  // scale and shift of its four channels; the second sum is accumulated as sum dz * z and turned into
  // sum dz * (z - mean) at the very end (- mean * sum dz: exact bf16 products and fp32 accumulation either way).
  float bsc[NB16], bsh[NB16];
  if constexpr (BNR) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int i = 0; i < 4; ++i) zr[mb][ps][i][0] = zr[mb][ps][i][1] = 0u;
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) { bsc[cb] = P.bn_scale[NB16 * c16 + cb]; bsh[cb] = P.bn_shift[NB16 * c16 + cb]; }
  }
  auto emit_bnr = [&](int u) {
    const int mb = u >> 3, ps = (u >> 2) & 1, i = u & 3;
    const bool inside = st_xl + 4 * ps + i < st_wlim[mb];
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) {
      const unsigned zw = zr[mb][ps][i][cb >> 1], dw = pk[mb][ps][cb][i >> 1];
      const float z = (cb & 1) ? H16<H>::hi(zw) : H16<H>::lo(zw);
      const float f = (i & 1) ? H16<H>::hi(dw) : H16<H>::lo(dw);
      const float dz = (inside && fmaf(z, bsc[cb], bsh[cb]) > 0.f) ? f : 0.f;
      tsum[cb] += dz;
      tsq[cb] = fmaf(dz, z, tsq[cb]);
    }
  };

  // ONE fragment register set (4 A + 4 B fragments), each re-read right after the MFMA that used it last; the k-step
  // stream runs on across stages and tiles, and the stage barrier sits inside the stage's last k-step behind the MFMAs
  // of its first A fragment (see conv3x3_k32_kernel).  What else a k-step carries -- DMA pieces, the previous tile's
  // statistics and stores -- goes into the three slots behind the MFMA groups of A fragments 0..2.
  const AS3 char* lds3 = reinterpret_cast<const AS3 char*>(0);
  frag_t fa[MB][2], fb[NB16];
  auto read_a = [&](int mb, int ps, int t, int a_base) {
    const int rr = mb + t / 3, col = 4 * ps + (t % 3);
    int addr;      // asm volatile: as plain arithmetic the 28 swizzle terms are hoisted out of the loop and spilled
    asm volatile("v_bfe_u32 %0, %1, %2, 2\n\tv_lshl_add_u32 %0, %0, 4, %3"
                 : "=&v"(addr)
                 : "v"(rr < 2 ? swz_lo : swz_hi), "n"(2 * ((rr & 1) * 8 + col)), "v"(a_base));
    fa[mb][ps] = *reinterpret_cast<const AS3 frag_t*>(lds3 + addr + (rr * HP + col) * 64);
  };
  {
    const int a_base = (int)lds0 + (p00 << 6), w_base = (int)ldsw0 + lane * 16;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) read_a(mb, ps, 0, a_base);
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) fb[cb] = *reinterpret_cast<const AS3 frag_t*>(lds3 + w_base + cb * 1024);
  }

  auto run_stage = [&](auto first_c, auto fetch_c, int tile, int ic) {
    constexpr bool FIRST = decltype(first_c)::value;         // first chunk of a tile: C = 0, and the pending stores
    // BNR: the tile's last stage (never the first: the host sends Cin < 64 elsewhere) starts with the z fetch, which
    // lands under its MFMAs (vmcnt(0) at its barrier).  A stage of its own in the code -- not a run-time test -- so that
    // the z registers are defined on every path to the hand-over and dead before it.
    if constexpr (decltype(fetch_c)::value) fetch_z(tile);
    int next_in = ic + 1;
    bool live_in = true;
    if (ic + 1 == nin) {
      next_in = 0;
      if (tile + 1 < tile_end) setup_stage(tile + 1);
      else live_in = false;
    }
    int next_wc = ic + 1;
    bool live_w = true;
    if (ic + 1 == nin) {
      next_wc = 0;
      live_w = tile + 1 < tile_end;
    }
    const unsigned nbuf_in = lds0 + (curi ^ 1) * IN_BYTES;
    const unsigned nbuf_w = ldsw0 + (curw ^ 1) * W_BYTES;
    const unsigned wfrag = (unsigned)(next_wc * NFRAG);
    const int widx = WRES ? (ic & 1) : curw;
    const int widx_next = WRES ? (next_wc & 1) : (curw ^ 1);
    int a_base = (int)(lds0 + curi * IN_BYTES) + (p00 << 6), a_next = (int)(lds0 + (curi ^ 1) * IN_BYTES) + (p00 << 6);
    int w_base = (int)(ldsw0 + widx * W_BYTES) + lane * 16, w_next = (int)(ldsw0 + widx_next * W_BYTES) + lane * 16;
    asm volatile("" : "+v"(a_base), "+v"(a_next), "+v"(w_base), "+v"(w_next));   // keep tap / fragment offsets immediate
    const bool keep_stores = FIRST && st_all16 && !BNR;      // BNR interleaves its stores with the DMA pieces
    // slots (k-step t, A fragment ag) with t < 8, ag < 3: 24 per stage.  A first stage issues its DMA pieces in the
    // first DSTEPS k-steps and the 16 pending stores behind them (so that the stores are the youngest vector-memory
    // operations and the barrier's vmcnt(16) waits for the DMA only); other stages spread the pieces over all 24 slots.
    constexpr int DS = FIRST ? DSTEPS : 8;
    constexpr int DSLOTS = 3 * DS, SSLOTS = 24 - 3 * DSTEPS;
    static_for<0, TAPS>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      constexpr bool last = t + 1 == TAPS;
      constexpr int tn = last ? 0 : t + 1;
      const int ab = last ? a_next : a_base, wb = last ? w_next : w_base;
#pragma unroll
      for (int ag = 0; ag < 2 * MB; ++ag) {
        const int mb = ag >> 1, ps = ag & 1;
#pragma unroll
        for (int cb = 0; cb < NB16; ++cb) {
          const f32x4v c_in = (FIRST && t == 0) ? f32x4v{0.f, 0.f, 0.f, 0.f} : acc[mb][ps][cb];
          acc[mb][ps][cb] = H16<H>::mfma16(__builtin_bit_cast(h8, fa[mb][ps]), __builtin_bit_cast(h8, fb[cb]), c_in);
          if (ag == 2 * MB - 1 && !(UNETK_K32_ABL & 32)) {
            __builtin_amdgcn_sched_barrier(0);
            fb[cb] = *reinterpret_cast<const AS3 frag_t*>(lds3 + wb + (tn * NB16 + cb) * 1024);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ag == 0 && last) {
          // next stage landed (this wave's DMAs; the 16 stores behind them may stay in flight -- but only when each of
          // them really was a vector-memory operation with a lane to write), every wave holds its last fragments
          if (keep_stores) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (!(UNETK_K32_ABL & 64)) read_a(mb, ps, tn, ab);
        if (!last && ag < 3) {
          const int sl = 3 * t + ag;
          // DMA pieces: piece q goes into slot q * DSLOTS / NDMA
#pragma unroll
          for (int q = 0; q < NDMA; ++q)
            if ((q * DSLOTS) / NDMA == sl) {
              if (q < IN_PER_WAVE) issue_input(q, next_in, nbuf_in, live_in);
              else issue_weight(q - IN_PER_WAVE, wfrag, live_w, nbuf_w);
            }
          if constexpr (FIRST && STATS) {
            // statistics of the previous tile, group t: channels {0,1}, {2}, {3} behind A fragments 0, 1, 2
            if (t < 4 && !(UNETK_K32_ABL & 1)) emit_stats(t, ag == 0 ? 0 : ag + 1, ag + 2);
          }
          if constexpr (FIRST && BNR) {
            // masked BN-backward sums of the previous tile, one unit per slot, each followed by the store of the same
            // pixel: pk and zr are released unit by unit while the first k-step is still defining the accumulators
            // (both held through the whole stage do not fit 256 registers)
            if (sl < 16) {
              if (sl >= 16 - BNR_DEFER && !(UNETK_K32_ABL & 8)) emit_bnr(sl);
              emit_store(sl);
            }
          } else if constexpr (FIRST) {
            // the 16 stores of the previous tile: store k goes into slot 3 * DSTEPS + k * SSLOTS / 16
#pragma unroll
            for (int k = 0; k < 16; ++k)
              if (3 * DSTEPS + (k * SSLOTS) / 16 == sl) emit_store(k);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    curw ^= 1;
    curi ^= 1;
  };

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    run_stage(std::true_type{}, std::false_type{}, tile, 0);
    if constexpr (BNR) {
      for (int ic = 1; ic + 1 < nin; ++ic) run_stage(std::false_type{}, std::false_type{}, tile, ic);
      run_stage(std::false_type{}, std::true_type{}, tile, nin - 1);
    } else {
      for (int ic = 1; ic < nin; ++ic) run_stage(std::false_type{}, std::false_type{}, tile, ic);
    }

    // ---- hand the tile over: round, statistics, remember where it goes.  Its stores ride under the next tile.
    const int img = tile / tiles_per_img, rem = tile - img * tiles_per_img;
    const int y0 = (rem / P.tiles_x) * TR, x0 = (rem % P.tiles_x) * 32;
    const H* yimg = reinterpret_cast<const H*>(P.y) + (size_t)img * P.H * P.W * P.ldy + P.yoff;
    st_xl = x0 + xg;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int yy = y0 + MB * wid + mb;
      st_row[mb] = yimg + (size_t)(yy * P.W + x0) * P.ldy;
      st_wlim[mb] = yy < P.H ? P.W : 0;
    }
    st_all16 = (y0 + MB * wid + 1 < P.H) && (x0 + 8 <= P.W);
    const bool ragged = (y0 + TR > P.H) || (x0 + 32 > P.W);  // wave-uniform: edge tiles mask what lies outside
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        unsigned rnd[NB16][2];
#pragma unroll
        for (int cb = 0; cb < NB16; ++cb) {
          f32x4v a = acc[mb][ps][cb];
          if (STATS && ragged) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = (st_xl + 4 * ps + i < st_wlim[mb]) ? a[i] : 0.f;
          }
          h2 lo = {(H)a[0], (H)a[1]}, hi = {(H)a[2], (H)a[3]};
          rnd[cb][0] = __builtin_bit_cast(unsigned, lo);
          rnd[cb][1] = __builtin_bit_cast(unsigned, hi);
        }
#pragma unroll
        for (int cb = 0; cb < NB16; ++cb) { pk[mb][ps][cb][0] = rnd[cb][0]; pk[mb][ps][cb][1] = rnd[cb][1]; }
        if constexpr (BNR) {
          // the units that are not deferred (their z registers are free again before the next tile starts)
          if (!(UNETK_K32_ABL & 8)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (mb * 8 + ps * 4 + i < 16 - BNR_DEFER) {
                __builtin_amdgcn_sched_barrier(0);   // (unfenced, all values of the tile are unpacked up front: spills)
                emit_bnr(mb * 8 + ps * 4 + i);
              }
            }
          }
        }
      }
  }
  // the last tile's statistics and stores
  if constexpr (STATS) {
#pragma unroll
    for (int g = 0; g < 4; ++g) emit_stats(g);
  }
  if constexpr (BNR) {
    if (!(UNETK_K32_ABL & 8)) {
#pragma unroll
      for (int u = 16 - BNR_DEFER; u < 16; ++u) emit_bnr(u);
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) emit_store(k);

  if (EPI != 0 && P.stats != nullptr) {
#pragma unroll
    for (int cb = 0; cb < NB16; ++cb) {                     // lanes l, l^16, l^32, l^48 hold the same channels
      tsum[cb] += __shfl_xor(tsum[cb], 16, 64);
      tsq[cb] += __shfl_xor(tsq[cb], 16, 64);
      tsum[cb] += __shfl_xor(tsum[cb], 32, 64);
      tsq[cb] += __shfl_xor(tsq[cb], 32, 64);
    }
    float* part = reinterpret_cast<float*>(smem);            // [NW][BN][2]; every DMA landed, every read retired
    if (g4 == 0) {
#pragma unroll
      for (int cb = 0; cb < NB16; ++cb) {
        part[(wid * BN + NB16 * c16 + cb) * 2 + 0] = tsum[cb];
        part[(wid * BN + NB16 * c16 + cb) * 2 + 1] = tsq[cb];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < 2 * BN) {
      float v = 0;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) v += part[w8 * 2 * BN + tid];
      double vd = (double)v;
      if constexpr (BNR) {
        if (tid & 1) {                                          // sum dz z  ->  sum dz (z - mean)  ->  sum dz xhat
          float s1 = 0;
#pragma unroll
          for (int w8 = 0; w8 < NW; ++w8) s1 += part[w8 * 2 * BN + tid - 1];
          vd = (vd - (double)P.bn_mean[tid >> 1] * (double)s1) * (double)P.bn_invstd[tid >> 1];
        }
      }
      double* dst = P.stats + ((size_t)(blockIdx.x & (UNETK_STAT_REPLICAS - 1)) * P.Cout) * 2 + tid;
      unsafeAtomicAdd(dst, vd);
    }
  }
}

template <typename H, bool WRES, int EPI> int launch_k32n64(const ConvParams& P, hipStream_t s) {
  constexpr size_t lds = 2 * (39 * 1024) + 2 * (36 * 1024);
  UNETK_ALLOW_LDS((conv3x3_k32n64_kernel<H, WRES, EPI>), lds);
  ConvParams Q = P;
  Q.tiles_x = host_cdiv(P.W, 32);
  Q.tiles_y = host_cdiv(P.H, 16);
  Q.n_tiles = 1;
  const long ntiles = (long)P.N * Q.tiles_y * Q.tiles_x;
  const int nstages = P.Cin / 32;
  int tpb = 1;
  {
    double best = 1e30;                                       // whole rounds of 256 workgroups (see launch_k32)
    UNETK_KNOB(tpb_max, "UNETK_CONV_TPB_MAX", 96);
    for (int t = 1; t <= tpb_max; ++t) {
      const long wgs = (ntiles + t - 1) / t;
      const long rounds = (wgs + 255) / 256;
      const double cost = (double)rounds * ((double)t * nstages + 1.5);
      if (cost < best * 0.999) { best = cost; tpb = t; }
    }
  }
  {
    UNETK_KNOB(tpb_force, "UNETK_CONV_TPB", 0);
    if (tpb_force > 0) tpb = tpb_force;
  }
  Q.tpb = tpb;
  const long nblk = (ntiles + tpb - 1) / tpb;
  if (nblk <= 0 || nblk > 0x7fffffffL) return UNETK_ERR_SHAPE;
  hipLaunchKernelGGL((conv3x3_k32n64_kernel<H, WRES, EPI>), dim3((unsigned)nblk), dim3(512), lds, s, Q);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

template <typename H, int NB16, bool AFF, bool WRES> int launch_k32(const ConvParams& P, hipStream_t s) {
  constexpr size_t lds = 2 * (39 * 1024) + 2 * ((NB16 == 8 ? 5 : 9) * NB16 * 1024) + 2 * 16 * NB16 * sizeof(float);
  UNETK_ALLOW_LDS((conv3x3_k32_kernel<H, NB16, AFF, WRES>), lds);
  ConvParams Q = P;
  Q.tiles_x = host_cdiv(P.W, 32);
  Q.tiles_y = host_cdiv(P.H, 16);
  const long ntiles = (long)P.N * Q.tiles_y * Q.tiles_x;
  const int nstages = (P.Cin / 32) * (NB16 == 8 ? 2 : 1);
  Q.n_tiles = P.Cout / (16 * NB16);
  // One workgroup per CU is resident, so a launch runs in ROUNDS of 256 workgroups and a partly filled last round is
  // idle CUs.  Tiles per workgroup: the value that minimises rounds x (tiles x stages per tile + ~1.5 stages of
  // prologue/epilogue)  (measured sweep: DESIGN.md 2.1).
  int tpb = 1;
  {
    double best = 1e30;
    UNETK_KNOB(tpb_max, "UNETK_CONV_TPB_MAX", 96);
    for (int t = 1; t <= tpb_max; ++t) {
      const long wgs = ((ntiles + t - 1) / t) * Q.n_tiles;
      const long rounds = (wgs + 255) / 256;
      const double cost = (double)rounds * ((double)t * nstages + 1.5);
      if (cost < best * 0.999) { best = cost; tpb = t; }
    }
  }
  {
    UNETK_KNOB(tpb_force, "UNETK_CONV_TPB", 0);
    if (tpb_force > 0) tpb = tpb_force;
  }
  Q.tpb = tpb;
  Q.dbg = 0;
  const long nblk = ((ntiles + tpb - 1) / tpb) * Q.n_tiles;
  if (nblk <= 0 || nblk > 0x7fffffffL) return UNETK_ERR_SHAPE;
  hipLaunchKernelGGL((conv3x3_k32_kernel<H, NB16, AFF, WRES>), dim3((unsigned)nblk), dim3(512), lds, s, Q);
  UNETK_CHECK_LAUNCH();
  return UNETK_OK;
}

template <typename H> int dispatch_k32(const ConvParams& P, hipStream_t s) {
  if (P.Cin % 32 != 0 || P.Cout % 64 != 0) return UNETK_ERR_SHAPE;
  const int nb16 = conv3_k32_nb16(P.Cout);
  const bool aff = P.aff_scale != nullptr;
  const bool bnr = P.bn_z != nullptr;
  if (nb16 == 8) {
    if (bnr) return UNETK_ERR_UNFUSED;
    return aff ? launch_k32<H, 8, true, false>(P, s) : launch_k32<H, 8, false, false>(P, s);
  }
  const bool wres = P.Cin <= 64;                            // <= 2 weight stages: resident in the two buffers
  const bool nopipe = conv3_k32_nopipe();
  if (!aff && P.Cout == 64 && !nopipe) {
    if (bnr) {                                              // the z fetch needs a stage of its own behind the first
      if (P.Cin < 64) return UNETK_ERR_UNFUSED;
      return wres ? launch_k32n64<H, true, 2>(P, s) : launch_k32n64<H, false, 2>(P, s);
    }
    if (P.stats != nullptr) return wres ? launch_k32n64<H, true, 1>(P, s) : launch_k32n64<H, false, 1>(P, s);
    return wres ? launch_k32n64<H, true, 0>(P, s) : launch_k32n64<H, false, 0>(P, s);
  }
  if (bnr) return UNETK_ERR_UNFUSED;
  if (aff) return wres ? launch_k32<H, 4, true, true>(P, s) : launch_k32<H, 4, true, false>(P, s);
  return wres ? launch_k32<H, 4, false, true>(P, s) : launch_k32<H, 4, false, false>(P, s);
}

}  // namespace

int unetk_internal_conv3x3_k32(const void* params, int dtype, void* stream) {
  const ConvParams& P = *static_cast<const ConvParams*>(params);
  if (dtype == UNETK_BF16) return dispatch_k32<bf16>(P, (hipStream_t)stream);
  if (dtype == UNETK_F16) return dispatch_k32<f16>(P, (hipStream_t)stream);
  return UNETK_ERR_DTYPE;
}
