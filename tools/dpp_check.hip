// Development check of the DPP / permlane-swap reductions of csrc/common.h against host sums.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I ss_asr_amd/csrc tools/dpp_check.hip -o tools/dpp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "common.h"
__global__ void k(const float* in, float* o1, float* o2, float* o3, float* o4) {
  const float v = in[threadIdx.x];
  o1[threadIdx.x] = half_sum(v);
  o2[threadIdx.x] = wave_sum(v);
  o3[threadIdx.x] = wave_max(v);
  o4[threadIdx.x] = half_max(v);
}
int main() {
  float h[64], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = std::sin(1.7f * i) * (1 + i % 5);
  hipMalloc(&d, 256); hipMalloc(&o, 4 * 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, o + 64, o + 128, o + 192);
  float r[256];
  hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
  double s0 = 0, s1 = 0; float m0 = -1e30f, m1 = -1e30f;
  for (int i = 0; i < 32; ++i) { s0 += h[i]; s1 += h[32 + i]; m0 = fmaxf(m0, h[i]); m1 = fmaxf(m1, h[32 + i]); }
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    const double hs = i < 32 ? s0 : s1;
    if (std::fabs(r[i] - hs) > 1e-4) ++bad;
    if (std::fabs(r[64 + i] - (s0 + s1)) > 1e-4) ++bad;
    if (r[128 + i] != fmaxf(m0, m1)) ++bad;
    if (r[192 + i] != (i < 32 ? m0 : m1)) ++bad;
    if (r[i] != r[i & 32] || r[64 + i] != r[64]) ++bad;          // every lane holds the same bits
  }
  printf("dpp reductions: %d mismatches (half sums %.5f %.5f, wave %.5f)\n", bad, r[0], r[32], r[64]);
  return bad != 0;
}
