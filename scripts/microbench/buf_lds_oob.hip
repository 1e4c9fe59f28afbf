// Does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) write ZEROS for lanes whose offset is
// out of range?  (conv kernels would then need no zero page and no per-lane 64-bit source select for padding pixels.)
//   hipcc --offload-arch=gfx950 -O2 buf_lds_oob.hip -o buf_lds_oob && ./buf_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 * 2; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  // raw buffer descriptor: base, stride 0, num_records = nbytes, flags as hipcc's own raw buffer intrinsics use on gfx9
  u32x4 desc;
  const uint64_t a = (uint64_t)src;
  desc[0] = (unsigned)a;
  desc[1] = (unsigned)(a >> 32) & 0xffffu;
  desc[2] = nbytes;
  desc[3] = 0x00020000u;
  // lanes 0..31 in range, lanes 32..47 offset ~0u, lanes 48..63 just past the end
  unsigned voff = lane * 16u;
  if (lane >= 32 && lane < 48) voff = 0xffffffffu;
  if (lane >= 48) voff = nbytes + (lane - 48) * 16u;
  const unsigned ldsaddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (mode ? 1024u : 0u);
  const unsigned soff = mode ? 64u : 0u;      // mode 1: soffset = 64 bytes (does it shift the in-range lanes only?)
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&s"(keep)
      : "v"(voff), "s"(ldsaddr), "s"(desc), "s"(soff)
      : "memory");
  __syncthreads();
  for (int i = lane; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}

int main() {
  const int n = 64 * 4 + 64;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0x1000u + i;
  unsigned *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 64 * 4 * 2 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 32 * 16u + (mode ? 64u : 0u), o, mode);
    std::vector<unsigned> r(64 * 4 * 2);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    printf("mode %d (num_records = %u bytes)\n", mode, 32 * 16u + (mode ? 64u : 0u));
    const int base = mode ? 256 : 0;
    for (int l = 0; l < 64; l += 1)
      if (l < 2 || (l >= 30 && l < 34) || (l >= 46 && l < 50) || l == 63)
        printf("  lane %2d: %08x %08x %08x %08x\n", l, r[base + l * 4], r[base + l * 4 + 1], r[base + l * 4 + 2], r[base + l * 4 + 3]);
    int zeros = 0, good = 0;
    for (int l = 0; l < 64; ++l) {
      bool z = true;
      for (int e = 0; e < 4; ++e) z = z && r[base + l * 4 + e] == 0u;
      zeros += z && l >= 32;
      const unsigned exp0 = 0x1000u + l * 4 + (mode ? 16 : 0);
      good += l < 32 && r[base + l * 4] == exp0;
    }
    printf("  in-range lanes correct: %d/32, out-of-range lanes zero-filled: %d/32\n", good, zeros);
  }
  return 0;
}
