// micro-benchmark: how long one wave64 waits between DEPENDENT vector instructions on gfx950, and how many resident
// waves (or independent chains inside a wave) it takes to fill a SIMD.  One workgroup = one wave; W waves per SIMD.
//   chains = independent fma chains interleaved in one wave; clk = nominal clocks (2.4 GHz) per instruction per wave
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
  float acc[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) acc[i] = (float)threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) r += acc[i];
  if (r == 12345.678f) out[0] = r;
}
template <int CHAINS>
static void run(float* d, int waves_per_simd) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 8192, blocks = 256 * 4 * waves_per_simd;
  k<CHAINS><<<blocks, 64>>>(d, 64, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS><<<blocks, 64>>>(d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_wave = (double)iters * 16 * CHAINS;
  const double clk = ms * 1e-3 * 2.4e9;
  printf("chains %d waves/SIMD %d: %.3f ms, %.2f clk per instruction per wave, %.2f clk per instruction per SIMD\n", CHAINS,
         waves_per_simd, ms, clk / inst_per_wave, clk / (inst_per_wave * waves_per_simd));
}
int main() {
  float* d; hipMalloc(&d, 4);
  for (int w : {1, 2, 4, 8}) { run<1>(d, w); run<2>(d, w); run<4>(d, w); run<8>(d, w); }
  return 0;
}
