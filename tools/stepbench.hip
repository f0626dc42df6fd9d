// Diagnostic microbenchmark (not part of the product library): times a chain
// of encoder-style LSTM step launches and, with -DSTAMPS, records where the
// cycles of one launch go.  Build: hipcc -O3 --offload-arch=gfx950 -I ss_asr_amd/csrc tools/stepbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#ifdef STAMPS
#define SSASR_STAMP(i) do { if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_stamp[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (i)] = t_; } } while (0)
__device__ unsigned long long g_stamp[8 * 4096];
#else
#define SSASR_STAMP(i)
#endif
#include "rnn_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", e, __FILE__, __LINE__); return 1; } } while (0)

__global__ void tiny_kernel(int x, float* p) { if (x < 0) p[0] = 1.f; }

int main(int argc, char** argv) {
  const int64_t S = 400, N = 32, H = 256;
  const int64_t rows = S * N;
  float *gates, *cs, *hs, *y, *whh, *dy, *whhT, *dc;
  CK(hipMalloc(&gates, sizeof(float) * 2 * rows * 4 * H));
  CK(hipMalloc(&cs, sizeof(float) * 2 * rows * H));
  CK(hipMalloc(&hs, sizeof(float) * 2 * rows * H));
  CK(hipMalloc(&y, sizeof(float) * rows * 2 * H));
  CK(hipMalloc(&dy, sizeof(float) * rows * 2 * H));
  CK(hipMalloc(&whh, sizeof(float) * 2 * 4 * H * H));
  CK(hipMalloc(&whhT, sizeof(float) * 2 * 4 * H * H));
  CK(hipMalloc(&dc, sizeof(float) * 2 * 2 * N * H));
  CK(hipMemset(gates, 0, sizeof(float) * 2 * rows * 4 * H));
  CK(hipMemset(whh, 0, sizeof(float) * 2 * 4 * H * H));
  CK(hipMemset(whhT, 0, sizeof(float) * 2 * 4 * H * H));
  CK(hipMemset(dy, 0, sizeof(float) * rows * 2 * H));
  CK(hipMemset(cs, 0, sizeof(float) * 2 * rows * H));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int64_t ys_s = N * 2 * H, ys_n = 2 * H;

  EncFwd ef{};
  ef.whh[0] = whh; ef.whh[1] = whh + 4 * H * H; ef.gates = gates; ef.cs = cs; ef.hs = hs; ef.y = y; ef.lens = nullptr;
  ef.ys_s = (int)ys_s; ef.ys_n = (int)ys_n; ef.S = (int)S; ef.N = (int)N; ef.H = (int)H;
  EncBwd eb{};
  eb.whhT = whhT; eb.gates = gates; eb.cs = cs; eb.dy = dy; eb.dc = dc; eb.lens = nullptr;
  eb.ys_s = (int)ys_s; eb.ys_n = (int)ys_n; eb.S = (int)S; eb.N = (int)N; eb.H = (int)H;
  auto run_fwd = [&]() { for (int64_t i = 0; i < S; ++i) hipLaunchKernelGGL(lstm_enc_fwd_kernel, cell_fwd_grid(H, 2, N), dim3(256), 0, st, ef, (int)i); };
  auto run_bwd = [&]() { for (int64_t i = 0; i < S; ++i) hipLaunchKernelGGL(lstm_enc_bwd_kernel, cell_bwd_grid(H, 2, N), dim3(256), 0, st, eb, (int)i); };
  auto run_empty = [&]() { for (int64_t i = 0; i < S; ++i) hipLaunchKernelGGL(tiny_kernel, cell_fwd_grid(H, 2, N), dim3(256), 0, st, 1, y); };

  float ms;
  auto timeit = [&](const char* name, auto fn) {
    fn(); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-28s %.3f us / step\n", name, ms * 1e3 / S); return 0; };
  auto graphit = [&](const char* name, auto fn) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); fn(); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-28s %.3f us / step (graph replay)\n", name, ms * 1e3 / S); return 0; };
  for (int rep = 0; rep < 2; ++rep) {
    timeit("fwd chain", run_fwd); timeit("bwd chain", run_bwd); timeit("empty 12B args", run_empty);
    graphit("fwd chain", run_fwd); graphit("bwd chain", run_bwd); graphit("empty 12B args", run_empty);
  }
#ifdef STAMPS
  // stamp one fwd and one bwd launch in the middle of a chain
  for (int which = 0; which < 2; ++which) {
    for (int64_t i = 0; i < 50; ++i) {
      if (which == 0) hipLaunchKernelGGL(lstm_enc_fwd_kernel, cell_fwd_grid(H, 2, N), dim3(256), 0, st, ef, (int)i);
      else hipLaunchKernelGGL(lstm_enc_bwd_kernel, cell_bwd_grid(H, 2, N), dim3(256), 0, st, eb, (int)i);
    }
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> hst(8 * 4096);
    CK(hipMemcpyFromSymbol(hst.data(), HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 8 * 4096));
    const int nwg = which == 0 ? 128 : 64 * 2;
    printf("%s stamps (cycles, median over %d workgroups): ", which ? "bwd" : "fwd", nwg);
    for (int k = 1; k < 6; ++k) {
      std::vector<long long> d;
      for (int w = 0; w < nwg; ++w) d.push_back((long long)(hst[w * 8 + k] - hst[w * 8 + k - 1]));
      std::sort(d.begin(), d.end());
      printf(" seg%d=%lld", k, d[d.size() / 2]);
    }
    for (int k = 6; k < 8; ++k) {
      std::vector<long long> d;
      for (int w = 0; w < nwg; ++w) d.push_back((long long)(hst[w * 8 + k] - hst[w * 8 + 0]));
      std::sort(d.begin(), d.end());
      printf(" t%d-t0=%lld", k, d[d.size() / 2]);
    }
    unsigned long long mn = ~0ull, mx = 0;
    for (int w = 0; w < nwg; ++w) { mn = std::min(mn, hst[w * 8]); mx = std::max(mx, hst[w * 8 + 5]); }
    printf("  span(first start..last end)=%llu\n", mx - mn);
  }
#endif
  return 0;
}
