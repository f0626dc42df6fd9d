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

__global__ void empty_kernel(CellFwdPair pr) { if (pr.d[0].N < 0) pr.d[0].c_out[0] = 1.f; }
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

  auto fwd_args = [&](int64_t i) {
    CellFwdPair pr;
    for (int d = 0; d < 2; ++d) {
      const int64_t s = d ? S - 1 - i : i, sp = d ? s + 1 : s - 1;
      CellFwd& c = pr.d[d]; c = CellFwd{};
      float* gd = gates + (d * rows + s * N) * 4 * H; float* cd = cs + d * rows * H; float* hd = hs + d * rows * H;
      if (i > 0) { c.sl.nseg = 1; seg_set(c.sl, 0, hd + sp * N * H, H, whh + d * 4 * H * H, H, (int)H); c.c_prev = cd + sp * N * H; }
      c.pre = gd; c.gates = gd; c.c_out = cd + s * N * H; c.h_out = hd + s * N * H;
      c.y = y + s * ys_s + d * H; c.ys_n = ys_n; c.s = (int)s; c.N = (int)N; c.H = (int)H;
    }
    return pr;
  };
  auto bwd_args = [&](int64_t i) {
    CellBwdPair pr;
    for (int d = 0; d < 2; ++d) {
      const int64_t s = d ? i : S - 1 - i, sn = d ? s - 1 : s + 1, sp = d ? s + 1 : s - 1;
      const bool has_prev = d ? (s < S - 1) : (s > 0);
      CellBwd& c = pr.d[d]; c = CellBwd{};
      float* gd = gates + d * rows * 4 * H; const float* cd = cs + d * rows * H; float* dcb = dc + d * 2 * N * H;
      if (i > 0) { c.sl.nseg = 1; seg_set(c.sl, 0, gd + sn * N * 4 * H, 4 * H, whhT + d * 4 * H * H, 4 * H, (int)(4 * H)); c.dc_in = dcb + (i & 1) * N * H; }
      c.add1 = dy + s * ys_s + d * H; c.ld1 = ys_n;
      c.gates = gd + s * N * 4 * H; c.dgates = gd + s * N * 4 * H;
      c.c_prev = has_prev ? cd + sp * N * H : nullptr; c.c = cd + s * N * H;
      c.dc_out = dcb + ((i + 1) & 1) * N * H; c.s = (int)s; c.N = (int)N; c.H = (int)H;
    }
    return pr;
  };

  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int64_t i = 0; i < S; ++i) { auto pr = fwd_args(i); hipLaunchKernelGGL(lstm_cell_fwd_kernel, cell_fwd_grid(H, 2, N), dim3(256), 0, st, pr); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fwd chain      : %.3f us / step\n", ms * 1e3 / S);
    CK(hipEventRecord(e0, st));
    for (int64_t i = 0; i < S; ++i) { auto pr = bwd_args(i); hipLaunchKernelGGL(lstm_cell_bwd_kernel, cell_bwd_grid(H, 2, N), dim3(256), 0, st, pr); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("bwd chain      : %.3f us / step\n", ms * 1e3 / S);
    CK(hipEventRecord(e0, st));
    for (int64_t i = 0; i < S; ++i) { auto pr = fwd_args(i); hipLaunchKernelGGL(empty_kernel, cell_fwd_grid(H, 2, N), dim3(256), 0, st, pr); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty, big args: %.3f us / step\n", ms * 1e3 / S);
    CK(hipEventRecord(e0, st));
    for (int64_t i = 0; i < S; ++i) { hipLaunchKernelGGL(tiny_kernel, cell_fwd_grid(H, 2, N), dim3(256), 0, st, 1, y); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty, 12B args: %.3f us / step\n", ms * 1e3 / S);
  }
#ifdef STAMPS
  // stamp one fwd and one bwd launch in the middle of a chain
  for (int which = 0; which < 2; ++which) {
    for (int64_t i = 0; i < 50; ++i) {
      if (which == 0) { auto pr = fwd_args(i); hipLaunchKernelGGL(lstm_cell_fwd_kernel, cell_fwd_grid(H, 2, N), dim3(256), 0, st, pr); }
      else { auto pr = bwd_args(i); hipLaunchKernelGGL(lstm_cell_bwd_kernel, cell_bwd_grid(H, 2, N), dim3(256), 0, st, pr); }
    }
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> hst(8 * 4096);
    CK(hipMemcpyFromSymbol(hst.data(), HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 8 * 4096));
    const int nwg = which == 0 ? 128 : 64;
    printf("%s stamps (cycles, median over %d workgroups): ", which ? "bwd" : "fwd", nwg);
    for (int k = 1; k < 6; ++k) {
      std::vector<long long> d;
      for (int w = 0; w < nwg; ++w) d.push_back((long long)(hst[w * 8 + k] - hst[w * 8 + k - 1]));
      std::sort(d.begin(), d.end());
      printf(" seg%d=%lld", k, d[d.size() / 2]);
    }
    unsigned long long mn = ~0ull, mx = 0;
    for (int w = 0; w < nwg; ++w) { mn = std::min(mn, hst[w * 8]); mx = std::max(mx, hst[w * 8 + 5]); }
    printf("  span(first start..last end)=%llu\n", mx - mn);
  }
#endif
  return 0;
}
