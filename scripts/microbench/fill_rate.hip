// L2 -> LDS fill rate per CU on gfx950, three ways (perf probe, not part of the library):
//   dma16 : global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction)
//   dma4  : global_load_lds_dword   (LDS-DMA, 256 B per wave instruction)
//   vgpr  : global_load_dwordx4 into registers + ds_write_b128
// One 512-thread workgroup per CU streams a small (L2-resident) region into LDS over and over.
//   hipcc --offload-arch=gfx950 -O3 -o fill_rate fill_rate.hip && ./fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define AS3 __attribute__((address_space(3)))

__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma4(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512) void fill_kernel(const char* src, size_t region, int iters, unsigned long long* cycles,
                                                   float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(AS3 char*)smem;
  const char* base = src + (size_t)blockIdx.x * region;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  constexpr int PIECES = 8;                       // 1-KiB pieces per wave per iteration: 64 KiB per workgroup
  for (int it = 0; it < iters; ++it) {
    const char* p = base + ((size_t)(it & 3) * 64 * 1024) % region;
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < PIECES; ++q)
        dma16(p + (q * 8 + wid) * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + (q * 8 + wid) * 1024));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < PIECES; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          dma4(p + (q * 8 + wid) * 1024 + s * 256 + lane * 4,
               __builtin_amdgcn_readfirstlane(lds0 + (q * 8 + wid) * 1024 + s * 256));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      uint4 r[PIECES];
#pragma unroll
      for (int q = 0; q < PIECES; ++q) r[q] = *reinterpret_cast<const uint4*>(p + (q * 8 + wid) * 1024 + lane * 16);
#pragma unroll
      for (int q = 0; q < PIECES; ++q) *reinterpret_cast<uint4*>(smem + (q * 8 + wid) * 1024 + lane * 16) = r[q];
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (sink && tid == 0) sink[blockIdx.x] = reinterpret_cast<float*>(smem)[iters & 1023];
}

int main(int argc, char** argv) {
  const int nwg = 256, iters = 2000;
  const size_t region = (argc > 1 ? atoi(argv[1]) : 256) * 1024;   // KiB per workgroup: 64 -> L2-resident, 256 -> MALL
  printf("region %zu KiB per workgroup\n", region / 1024);
  char* src;
  unsigned long long* cyc;
  float* sink;
  hipMalloc(&src, nwg * region);
  hipMemset(src, 1, nwg * region);
  hipMalloc(&cyc, nwg * sizeof(*cyc));
  hipMalloc(&sink, nwg * sizeof(float));
  const char* names[3] = {"dma16", "dma4", "vgpr"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(nwg), dim3(512), 65536, 0, src, region, iters, cyc, sink);
      if (mode == 1) hipLaunchKernelGGL(fill_kernel<1>, dim3(nwg), dim3(512), 65536, 0, src, region, iters, cyc, sink);
      if (mode == 2) hipLaunchKernelGGL(fill_kernel<2>, dim3(nwg), dim3(512), 65536, 0, src, region, iters, cyc, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) {
        const double bytes = (double)nwg * iters * 65536.0;
        printf("%-6s %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s total\n", names[mode], ms, bytes / nwg / (ms * 1e-3) / 1e9,
               bytes / (ms * 1e-3) / 1e12);
      }
    }
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
  return 0;
}
