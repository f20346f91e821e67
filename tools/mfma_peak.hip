// Practical MFMA ceiling of the chip (measurement tool): v_mfma_f32_32x32x16_bf16 issued back to back from registers, no
// memory traffic, on every SIMD of every CU -- what "100 % MFMA" means on this part at the clock it sustains under that load.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip && tools/mfma_peak
// Variants: waves per SIMD (1, 2), independent accumulator chains per wave (1, 2, 4); also a dependent v_fma chain to read
// the shader clock (one wave64 VALU op issues in 4 cycles... measured as ops/s) with the chip idle and under MFMA load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int CHAINS>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(64) void fma_chain(float* out, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) x = __builtin_fmaf(x, y, 1e-7f);
  }
  if (x == 12345.678f) out[0] = x;
}

// same loop with eight different pseudo-random operand pairs used in turn (operand toggling as with real activations):
// the sustained clock -- hence the ceiling -- depends on the data
__global__ __launch_bounds__(512) void mfma_loop_random(float* out, int iters) {
  bf16x8 a[8], b[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int u = 0; u < 8; ++u)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; a[u][i] = (__bf16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
      h = h * 1664525u + 1013904223u; b[u][i] = (__bf16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
    }
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + c) & 7], b[(u * 3 + c) & 7], acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
  if (s == 12345.678f) out[0] = s;
}

double run_mfma_random(int cus, float* d, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_loop_random, dim3(cus), dim3(512), 0, 0, d, iters / 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(mfma_loop_random, dim3(cus), dim3(512), 0, 0, d, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)cus * 8 * iters * 8.0 * 4 * 2.0 * 32 * 32 * 16 / (ms * 1e-3);
}

template <int CHAINS>
double run_mfma(int waves_per_simd, int cus, float* d, int iters) {
  const int threads = 256 * waves_per_simd;   // 4 SIMDs x waves_per_simd waves
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(cus), dim3(threads), 0, 0, d, iters / 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(cus), dim3(threads), 0, 0, d, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)cus * (threads / 64) * iters * 8.0 * CHAINS * 2.0 * 32 * 32 * 16;
  return flops / (ms * 1e-3);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("%s: %d CUs, clockRate %.0f MHz\n", p.gcnArchName, cus, p.clockRate / 1e3);
  float* d; CK(hipMalloc(&d, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {  // shader clock from a dependent VALU chain, one wave on one CU (idle chip): 64 fma per iteration
    const int iters = 200000;
    hipLaunchKernelGGL(fma_chain, dim3(1), dim3(64), 0, 0, d, iters / 10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(fma_chain, dim3(1), dim3(64), 0, 0, d, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dependent v_fma chain, 1 wave: %.2f ns per op (at 4..8 cycles per dependent wave64 op: %.2f..%.2f GHz)\n", ms * 1e6 / (iters * 64.0),
           4.0 / (ms * 1e6 / (iters * 64.0)), 8.0 / (ms * 1e6 / (iters * 64.0)));
  }
  const int iters = 20000;
  printf("v_mfma_f32_32x32x16_bf16 from registers, all %d CUs (nominal 2500 TFLOP/s = 32 cycles per MFMA per SIMD at 2.4 GHz):\n", cus);
  printf("  1 wave/SIMD, 1 chain : %7.1f TFLOP/s\n", run_mfma<1>(1, cus, d, iters) / 1e12);
  printf("  1 wave/SIMD, 2 chains: %7.1f TFLOP/s\n", run_mfma<2>(1, cus, d, iters) / 1e12);
  printf("  1 wave/SIMD, 4 chains: %7.1f TFLOP/s\n", run_mfma<4>(1, cus, d, iters) / 1e12);
  printf("  2 waves/SIMD, 1 chain : %7.1f TFLOP/s\n", run_mfma<1>(2, cus, d, iters) / 1e12);
  printf("  2 waves/SIMD, 4 chains: %7.1f TFLOP/s\n", run_mfma<4>(2, cus, d, iters) / 1e12);
  printf("  2 waves/SIMD, 4 chains, 32 CUs only: %7.1f TFLOP/s (x%d/32 = %.1f if it scaled)\n", run_mfma<4>(2, 32, d, iters) / 1e12, cus,
         run_mfma<4>(2, 32, d, iters) / 1e12 * cus / 32);
  printf("  2 waves/SIMD, 4 chains, long run (10x): %7.1f TFLOP/s\n", run_mfma<4>(2, cus, d, iters * 10) / 1e12);
  printf("  2 waves/SIMD, 4 chains, RANDOM operands (8 pairs in turn): %7.1f TFLOP/s; long run (10x): %7.1f TFLOP/s\n",
         run_mfma_random(cus, d, iters) / 1e12, run_mfma_random(cus, d, iters * 10) / 1e12);
  return 0;
}
