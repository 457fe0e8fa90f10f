// diagnostic: what do the amdgcn log / exp2 / rsq builtins return on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(float* o) {
  float x = 0.9f + 0.001f * threadIdx.x;
  o[4 * threadIdx.x + 0] = __builtin_amdgcn_logf(x);
  o[4 * threadIdx.x + 1] = __builtin_amdgcn_exp2f(32.0f * __builtin_amdgcn_logf(x));
  o[4 * threadIdx.x + 2] = powf(x, 32.0f);
  o[4 * threadIdx.x + 3] = __builtin_amdgcn_rsqf(x);
}
int main() {
  float* d; hipMalloc(&d, 64 * 16);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 9) {
    float x = 0.9f + 0.001f * i;
    printf("x %.4f  log %.7f (log2 %.7f)  fastpow %.8f  powf %.8f  libm %.8f  rsq %.7f (%.7f)\n", x, h[4*i], log2f(x), h[4*i+1], h[4*i+2], powf(x, 32.0f), h[4*i+3], 1.0f/sqrtf(x));
  }
  return 0;
}
