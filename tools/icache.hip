// Does straight-line code cost more than the same work in a loop right after a launch?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Rep { static __device__ __forceinline__ void run(float& x, float y) { x = __builtin_fmaf(x, y, 1.0f); asm volatile("" : "+v"(x)); Rep<N - 1>::run(x, y); } };
template <> struct Rep<0> { static __device__ __forceinline__ void run(float&, float) {} };
template <int N> __global__ void straight(float* p, float y, unsigned long long* st) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float x = y; Rep<N>::run(x, y);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (x == 12345.f) p[0] = x;
  if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}
__global__ void rolled(float* p, float y, int n, unsigned long long* st) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float x = y;
  for (int i = 0; i < n; ++i) { x = __builtin_fmaf(x, y, 1.0f); asm volatile("" : "+v"(x)); }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (x == 12345.f) p[0] = x;
  if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}
__global__ void filler(float* p) { if (p == nullptr) p[0] = 1; }
int main() {
  float* p; unsigned long long* st; hipMalloc(&p, 4096); hipMalloc(&st, 8 * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  unsigned long long h[256];
  auto med = [&]() { hipMemcpy(h, st, 8 * 128, hipMemcpyDeviceToHost); unsigned long long s = 0; for (int i = 0; i < 128; ++i) s += h[i]; return (double)s / 128; };
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 20; ++i) { launch(); hipLaunchKernelGGL(filler, dim3(128), dim3(256), 0, 0, p); }
    hipEventRecord(e0); for (int i = 0; i < 500; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("%-28s %.3f us/launch  in-kernel %.0f cycles\n", name, ms * 2.0, med());
  };
  run("straight 256 fma", [&]() { hipLaunchKernelGGL(straight<256>, dim3(128), dim3(256), 0, 0, p, 1.0f, st); });
  run("straight 1024 fma", [&]() { hipLaunchKernelGGL(straight<1024>, dim3(128), dim3(256), 0, 0, p, 1.0f, st); });
  run("straight 4096 fma", [&]() { hipLaunchKernelGGL(straight<4096>, dim3(128), dim3(256), 0, 0, p, 1.0f, st); });
  run("rolled 4096 fma", [&]() { hipLaunchKernelGGL(rolled, dim3(128), dim3(256), 0, 0, p, 1.0f, 4096, st); });
  run("rolled 1024 fma", [&]() { hipLaunchKernelGGL(rolled, dim3(128), dim3(256), 0, 0, p, 1.0f, 1024, st); });
  run("rolled 256 fma", [&]() { hipLaunchKernelGGL(rolled, dim3(128), dim3(256), 0, 0, p, 1.0f, 256, st); });
  return 0;
}
