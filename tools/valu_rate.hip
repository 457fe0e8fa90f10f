// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (wave64), many waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float r = 0.f;
  if (PK) {
    f2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f2{(float)threadIdx.x + i, (float)i};
    f2 va = {a, a * 1.0001f}, vb = {b, b * 0.999f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(acc[i], va, vb);
    for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y;
  } else {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    for (int i = 0; i < 16; ++i) r += acc[i];
  }
  if (r == 12345.678f) out[0] = r;
}
int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 4096;
  for (int wpc : {4, 8, 16, 32}) {           // waves per CU
    int blocks = 256 * wpc / 4;
    for (int pk = 0; pk < 2; ++pk) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr_per_wave = (double)iters * (pk ? 8 : 16);
      double waves_per_simd = wpc / 4.0;
      double cyc = ms * 1e-3 * 2.4e9;
      printf("waves/CU %2d %s: %.3f ms, cycles per wave-instruction per SIMD = %.2f, lane-FMAs/clk/CU = %.1f\n", wpc,
             pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, cyc / (instr_per_wave * waves_per_simd),
             (double)blocks * 256 * instr_per_wave * (pk ? 2 : 1) / 256 / cyc);
    }
  }
  return 0;
}
