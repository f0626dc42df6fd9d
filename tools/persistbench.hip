// Diagnostic (not part of the product library): runs the persistent forward
// and backward recurrences of one encoder layer standalone, reports us / step
// and where one step's time goes (s_memtime stamps of thread 0 of every
// workgroup, median over workgroups and steps).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I ss_asr_amd/csrc tools/persistbench.hip -o tools/persistbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifdef REALTIME
#define SSASR_CLK "s_memrealtime"
#else
#define SSASR_CLK "s_memtime"
#endif
constexpr int TR_LO = 200, TR_N = 64, TR_SLOTS = 12, TR_WG = 512;
__device__ unsigned long long g_trace[TR_WG * TR_N * TR_SLOTS];
#define SSASR_PTRACE(step, slot) do { if (threadIdx.x == 0 && (step) >= TR_LO && (step) < TR_LO + TR_N) { \
  unsigned long long t_; asm volatile(SSASR_CLK " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  g_trace[((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * TR_N + ((step) - TR_LO)) * TR_SLOTS + (slot)] = t_; } } while (0)
__device__ unsigned g_retry[TR_WG * 4 * 4];   // per (workgroup, wave): steps needing 0, 1, 2, 3+ re-fetch rounds
#define SSASR_PRETRY(tries) do { if ((threadIdx.x & 63) == 0) \
  atomicAdd(&g_retry[((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + (threadIdx.x >> 6)) * 4 + ((tries) > 3 ? 3 : (tries))], 1u); } while (0)
#define SSASR_PTRACE_H(step, slot) do { if ((threadIdx.x & 63) == 0 && threadIdx.x >= 256 && (step) >= TR_LO && (step) < TR_LO + TR_N) { \
  unsigned long long t_; asm volatile(SSASR_CLK " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  g_trace[((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * TR_N + ((step) - TR_LO)) * TR_SLOTS + (slot)] = t_; } } while (0)
#include "rnn_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", e_, __FILE__, __LINE__); return 1; } } while (0)

static int report(const char* name, int nwg, double us_per_tick, int group_size) {
  std::vector<unsigned long long> h(TR_WG * TR_N * TR_SLOTS);
  CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_trace), h.size() * sizeof(unsigned long long)));
  printf("%s: median over %d workgroups x %d steps, us since step start (slot 0):\n ", name, nwg, TR_N);
  for (int k = 1; k < 12; ++k) {
    std::vector<double> d;
    for (int w = 0; w < nwg; ++w)
      for (int i = 0; i < TR_N; ++i) {
        const unsigned long long* p = &h[(w * TR_N + i) * TR_SLOTS];
        if (p[k] && p[0]) d.push_back((double)(long long)(p[k] - p[0]) * us_per_tick);
      }
    if (d.empty()) { printf(" s%d=--", k); continue; }
    std::sort(d.begin(), d.end());
    printf(" s%d=%.2f", k, d[d.size() / 2]);
  }
  // step period: slot 0 of step i+1 minus slot 0 of step i
  std::vector<double> per;
  for (int w = 0; w < nwg; ++w)
    for (int i = 0; i + 1 < TR_N; ++i)
      per.push_back((double)(long long)(h[(w * TR_N + i + 1) * TR_SLOTS] - h[(w * TR_N + i) * TR_SLOTS]) * us_per_tick);
  std::sort(per.begin(), per.end());
  printf("  period=%.2f\n", per[per.size() / 2]);
#ifdef REALTIME
  {
    // groups = workgroups with the same blockIdx.y/z: wg index = x + gx * (y + gy * z); group id passed as nwg / ngroup
    const int per = group_size;
    const int ngroup = nwg / per;
    std::vector<double> spread, lat, latmin;
    for (int g = 0; g < ngroup; ++g)
      for (int i = 0; i + 1 < TR_N; ++i) {
        unsigned long long smax = 0, smin = ~0ull;
        for (int w = 0; w < per; ++w) {
          const unsigned long long t = h[((g * per + w) * TR_N + i) * TR_SLOTS + 7];
          smax = std::max(smax, t); smin = std::min(smin, t);
        }
        spread.push_back((double)(smax - smin) * 0.01);
        for (int w = 0; w < per; ++w) {
          const unsigned long long rel = h[((g * per + w) * TR_N + i + 1) * TR_SLOTS + 2];
          lat.push_back((double)(long long)(rel - smax) * 0.01);
          latmin.push_back((double)(long long)(rel - smin) * 0.01);
        }
      }
    std::sort(spread.begin(), spread.end()); std::sort(lat.begin(), lat.end()); std::sort(latmin.begin(), latmin.end());
    printf("  realtime: store-time spread within a group median %.2f us (p90 %.2f); release - last store median %.2f us (p10 %.2f p90 %.2f); release - first store median %.2f\n",
           spread[spread.size() / 2], spread[spread.size() * 9 / 10], lat[lat.size() / 2], lat[lat.size() / 10], lat[lat.size() * 9 / 10], latmin[latmin.size() / 2]);
  }
#endif
  std::vector<unsigned> rt(TR_WG * 16), zero(TR_WG * 16, 0u);
  CK(hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(g_retry), rt.size() * sizeof(unsigned)));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_retry), zero.data(), zero.size() * sizeof(unsigned)));
  unsigned long long hist[4][4] = {};
  for (int w = 0; w < nwg; ++w) for (int v = 0; v < 4; ++v) for (int k = 0; k < 4; ++k) hist[v][k] += rt[(w * 4 + v) * 4 + k];
  for (int v = 0; v < 4; ++v)
    printf("  wave %d: re-fetch rounds 0/1/2/3+ = %llu %llu %llu %llu\n", v, hist[v][0], hist[v][1], hist[v][2], hist[v][3]);
  return 0;
}

int main(int argc, char** argv) {
  const int64_t S = argc > 1 ? atoi(argv[1]) : 400, N = argc > 2 ? atoi(argv[2]) : 32, H = 256;
  const int64_t rows = S * N;
  float *gates, *cs, *hs, *y, *whh, *dy, *whhT, *hx, *gx;
  int32_t* sync;
  const int64_t NpF = (N + 7) & ~7, NpB = (N + 15) & ~15;
  CK(hipMalloc(&gates, sizeof(float) * 2 * rows * 4 * H));
  CK(hipMalloc(&cs, sizeof(float) * 2 * rows * H));
  CK(hipMalloc(&hs, sizeof(float) * 2 * rows * H));
  CK(hipMalloc(&y, sizeof(float) * rows * 2 * H));
  CK(hipMalloc(&dy, sizeof(float) * rows * 2 * H));
  CK(hipMalloc(&whh, sizeof(float) * 2 * 4 * H * H));
  CK(hipMalloc(&whhT, sizeof(float) * 2 * 4 * H * H));
  CK(hipMalloc(&hx, sizeof(float) * 2 * S * NpF * H));
  CK(hipMalloc(&gx, sizeof(float) * std::max<size_t>(2 * S * 4 * H * NpB, (size_t)2 * 2 * BWD_RS_RING * 256 * 256)));
  CK(hipMalloc(&sync, 64));
  CK(hipMemset(gates, 0, sizeof(float) * 2 * rows * 4 * H));
  CK(hipMemset(whh, 0, sizeof(float) * 2 * 4 * H * H));
  CK(hipMemset(whhT, 0, sizeof(float) * 2 * 4 * H * H));
  CK(hipMemset(dy, 0, sizeof(float) * rows * 2 * H));
  CK(hipMemset(cs, 0, sizeof(float) * 2 * rows * H));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int64_t ys_s = N * 2 * H, ys_n = 2 * H;

  EncPersist pf{};
  pf.whh[0] = whh; pf.whh[1] = whh + 4 * H * H; pf.gates = gates; pf.cs = cs; pf.hs = hs; pf.hx = hx; pf.y = y;
  pf.delay = getenv("DELAY_F") ? atoi(getenv("DELAY_F")) : 24; pf.lens = nullptr; pf.status = sync + 4;
  pf.ys_s = (int)ys_s; pf.ys_n = (int)ys_n; pf.S = (int)S; pf.N = (int)N; pf.H = (int)H; pf.drop_tile = -1;
  EncPersistBwd pb{};
  pb.whhT = whhT; pb.gates = gates; pb.cs = cs; pb.dy = dy; pb.gx = gx; pb.lens = nullptr;
  pb.status = sync + 4; pb.delay = getenv("DELAY_B") ? atoi(getenv("DELAY_B")) : 16;
  pb.ys_s = (int)ys_s; pb.ys_n = (int)ys_n; pb.S = (int)S; pb.N = (int)N; pb.H = (int)H;
  const int nbF = getenv("NB_F") ? atoi(getenv("NB_F")) : 2;
  const int chF = (int)((N + 16 * nbF - 1) / (16 * nbF)), chB = (int)((N + 15) / 16);

  float ms; int status;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemsetAsync(sync, 0, 32, st));
    CK(hipMemsetD32Async((hipDeviceptr_t)hx, (int)PERSIST_SENTINEL, (size_t)(2 * S * NpF * H), st));
    CK(hipEventRecord(e0, st));
    const dim3 gridF = dim3(H / 4, 2, chF);
    if (nbF == 1) hipLaunchKernelGGL((lstm_enc_fwd_persistent_kernel<4, 1>), gridF, dim3(FWD_THREADS), 0, st, pf);
    else hipLaunchKernelGGL((lstm_enc_fwd_persistent_kernel<4, 2>), gridF, dim3(FWD_THREADS), 0, st, pf);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&status, sync + 4, 4, hipMemcpyDeviceToHost));
    printf("fwd persistent: %.3f us / step (status %d)\n", ms * 1e3 / S, status);
  }
  // tick calibration: stamps span of the traced window vs the event-timed period is
  // not exact, so calibrate s_memtime against the wall clock with a sleep kernel instead
  double us_per_tick = 0.01;   // 100 MHz
  if (report("fwd", (int)(H / 4) * 2 * chF, us_per_tick, (int)(H / 4))) return 1;
  CK(hipMemset(gates, 0, sizeof(float) * 2 * rows * 4 * H));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemsetAsync(sync, 0, 32, st));
    CK(hipMemsetD32Async((hipDeviceptr_t)gx, (int)PERSIST_SENTINEL,
                         (size_t)2 * chB * BWD_RS_RING * 256 * 256, st));
    CK(hipEventRecord(e0, st));
    if (getenv("HALVES_OFF")) hipLaunchKernelGGL((lstm_enc_bwd_rs_kernel<4, 1>), dim3(H / 16, 2, chB), dim3(320), 0, st, pb);
    else hipLaunchKernelGGL((lstm_enc_bwd_rs_kernel<4, 2>), dim3(H / 16, 2, chB * 2), dim3(320), 0, st, pb);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&status, sync + 4, 4, hipMemcpyDeviceToHost));
    printf("bwd K-split: %.3f us / step (status %d)\n", ms * 1e3 / S, status);
  }
  if (report("bwd", (int)(H / 16) * 2 * chB * (!getenv("HALVES_OFF") ? 2 : 1), us_per_tick, (int)(H / 16))) return 1;
  return 0;
}
