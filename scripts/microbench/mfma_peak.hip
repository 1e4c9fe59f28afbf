// What the matrix pipes of THIS box sustain ("a measured-on-box MFMA microbenchmark peak reported beside"
// the nominal 2.5 PFLOP/s): every SIMD issues independent v_mfma_f32_16x16x32_bf16 (or 32x32x16) back to back from
// registers -- no LDS, no memory -- for about a second, on zero operands and on random ones (the chip's clock under
// load depends on the data: MI355X_MICROARCH.md, DVFS).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ src, float* sink, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[8], b = __builtin_bit_cast(bf16x8, src[(t + 1777) & 4095]);
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = __builtin_bit_cast(bf16x8, src[(t + 97 * j) & 4095]);   // eight independent chains
  float out = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a[j]), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // inline-asm MFMA results: no compiler hazard padding
#pragma unroll
    for (int j = 0; j < 8; ++j) out += (c[j][0] + c[j][1]) + (c[j][2] + c[j][3]);
  } else {
    f32x16 c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) c[j][k] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a[j]), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) out += c[j][k];
  }
  if (out == 12345.678f) sink[t] = out;      // keeps the loop alive, never true in practice
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SHAPE>
static int run(const char* what, const uint4* src, float* sink, int cus) {
  // 8 waves per SIMD's worth of blocks in flight is more than the pipes need; 2 blocks of 4 waves per CU fill them
  const int blocks = cus * 8, iters = SHAPE == 16 ? 4000 : 4000;
  const double flop_per_wave_iter = SHAPE == 16 ? 8 * 2.0 * 16 * 16 * 32 : 4 * 2.0 * 32 * 32 * 16;
  const double flop = (double)blocks * 4 * iters * flop_per_wave_iter;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  double total = 0;
  for (int rep = 0; rep < 400 && total < 1500.0; ++rep) {       // ~1.5 s of back-to-back launches: the steady clock
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, src, sink, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
    total += t;
  }
  const size_t n = ms.size();
  std::vector<float> tail(ms.begin() + n / 2, ms.end());
  std::sort(tail.begin(), tail.end());
  printf("{\"shape\": \"%s\", \"operands\": \"%s\", \"launches\": %zu, \"first_tflops\": %.1f, \"steady_median_tflops\": %.1f, "
         "\"steady_best_tflops\": %.1f}\n",
         SHAPE == 16 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16", what, n, flop / ms[0] * 1e-9,
         flop / tail[tail.size() / 2] * 1e-9, flop / tail[0] * 1e-9);
  fflush(stdout);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  uint4 *zeros, *rnd;
  float* sink;
  CK(hipMalloc(&zeros, 4096 * 16)); CK(hipMalloc(&rnd, 4096 * 16)); CK(hipMalloc(&sink, (size_t)cus * 8 * 256 * 4));
  CK(hipMemset(zeros, 0, 4096 * 16));
  std::vector<unsigned short> h(4096 * 8);
  srand(1);
  for (auto& v : h) {                        // bf16 values in (-2, 2): random sign, exponent 0x7e..0x7f, random mantissa
    v = (unsigned short)(((rand() & 1) << 15) | ((0x7e + (rand() & 1)) << 7) | (rand() & 0x7f));
  }
  CK(hipMemcpy(rnd, h.data(), 4096 * 16, hipMemcpyHostToDevice));
  if (run<16>("zeros", zeros, sink, cus)) return 1;
  if (run<16>("random", rnd, sink, cus)) return 1;
  if (run<32>("zeros", zeros, sink, cus)) return 1;
  if (run<32>("random", rnd, sink, cus)) return 1;
  return 0;
}
