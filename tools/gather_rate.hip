// micro-benchmark: rate of 16-byte-per-lane gathers (global_load_dwordx4) on gfx950 as a function of how
// the 64 lane addresses of one instruction spread over cache lines.  No arithmetic besides the address
// update; 8 independent gathers in flight per wave, 5 waves per SIMD like the DVR kernel.
//   pattern 0: all lanes one address          pattern 1: 64 consecutive float4 (8 lines, fully coalesced)
//   pattern 2: groups of 4 lanes share a 64-byte segment, groups on random lines of a small (L1/L2) set
//   pattern 3: groups of 2 lanes share a line, random lines      pattern 4: every lane its own random line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int PATTERN>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ base, uint32_t n_lines_mask, int iters, float* out) {
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 4u + (threadIdx.x >> 6));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t s = hash(wave * 977u + 13u);
  for (int it = 0; it < iters; ++it) {
    float4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 1664525u + 1013904223u;                 // wave-uniform stream
      uint32_t idx;                                    // float4 index; one line = 8 float4
      if (PATTERN == 0) idx = (s >> 8) & ((n_lines_mask << 3) | 7u);
      else if (PATTERN == 1) idx = (((s >> 8) & n_lines_mask & ~7u) << 3) + lane;
      else if (PATTERN == 2) idx = ((hash(s + (lane >> 2)) & n_lines_mask) << 3) + (lane & 3u) + ((s >> 3) & 4u);
      else if (PATTERN == 3) idx = ((hash(s + (lane >> 1)) & n_lines_mask) << 3) + (lane & 1u) + ((s >> 3) & 6u);
      else idx = ((hash(s + lane) & n_lines_mask) << 3) + ((s >> 3) & 7u);
      q[u] = base[idx];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += q[u].x; acc.y += q[u].y; acc.z += q[u].z; acc.w += q[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main() {
  const size_t lines_big = 1u << 24;                   // 2 GiB of 128-byte lines
  float4* d; float* o;
  hipMalloc(&d, lines_big * 128); hipMalloc(&o, 4);
  hipMemset(d, 0, lines_big * 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const double clk = p.clockRate * 1e3;                // Hz
  const int cus = p.multiProcessorCount, iters = 512, blocks = cus * 5;   // 20 waves per CU
  printf("device %s, %d CUs, %.0f MHz\n", p.gcnArchName, cus, clk / 1e6);
  struct Set { const char* name; uint32_t mask; } sets[] = {{"64 KiB working set (L1)", (1u << 9) - 1}, {"2 MiB (L2)", (1u << 14) - 1},
                                                           {"128 MiB (MALL)", (1u << 20) - 1}, {"2 GiB (HBM)", (1u << 24) - 1}};
  for (auto& st : sets)
    for (int pat = 0; pat < 5; ++pat) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (pat) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, st.mask, iters, o); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, st.mask, iters, o); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, st.mask, iters, o); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, st.mask, iters, o); break;
          default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, st.mask, iters, o); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      double instr = (double)blocks * 4 * iters * 8;   // gather instructions
      double per_cu_clk = instr / cus / (ms * 1e-3 * clk);
      printf("%-26s pattern %d: %8.3f ms  %7.1f G gathers/s  %6.1f clk per gather per CU  %7.1f GB/s returned\n", st.name, pat, ms,
             instr / ms / 1e6, 1.0 / per_cu_clk, instr * 1024 / ms / 1e6);
    }
  return 0;
}
