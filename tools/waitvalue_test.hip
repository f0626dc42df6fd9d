// Does hipStreamWaitValue32 gate a second stream on a word that a RUNNING kernel of the first stream writes?
// (diagnostic for "one BPTT launch per layer, weight-gradient launches released by in-kernel progress words")
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at line %d\n", e_, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void producer(unsigned* word, unsigned long long* stamps, int spin_us) {
  const unsigned long long t0 = wall_clock64();
  stamps[0] = t0;
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) {}
  stamps[1] = wall_clock64();
  __threadfence();
  __hip_atomic_store(word, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t1 = wall_clock64();
  while (wall_clock64() - t1 < (unsigned long long)spin_us * 100) {}
  stamps[2] = wall_clock64();
}
__global__ void consumer(unsigned long long* stamps) { stamps[3] = wall_clock64(); }
int main() {
  int dev = 0, can = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  unsigned* word; unsigned long long* stamps;
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 0) CK(hipMalloc(&word, 64));
    else CK(hipExtMallocWithFlags((void**)&word, 64, hipMallocSignalMemory));
    CK(hipMalloc(&stamps, 64));
    CK(hipMemset(word, 0, 64)); CK(hipMemset(stamps, 0, 64));
    hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    CK(hipDeviceSynchronize());
    hipError_t e = hipStreamWaitValue32(b, word, 7, hipStreamWaitValueEq, 0xffffffffu);
    if (e != hipSuccess) { printf("mode %d: hipStreamWaitValue32 -> %d (%s)\n", mode, e, hipGetErrorString(e)); continue; }
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamps);
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, word, stamps, 200);
    CK(hipDeviceSynchronize());
    unsigned long long h[4]; CK(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
    printf("mode %d (%s): producer start 0, word written at %.1f us, producer end %.1f us, consumer ran at %.1f us\n", mode,
           mode ? "signal memory" : "hipMalloc", (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, ((long long)h[3] - (long long)h[0]) / 100.0);
  }
  return 0;
}
