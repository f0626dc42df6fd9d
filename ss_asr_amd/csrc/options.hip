// Process-wide diagnostic switches and device queries shared by the entry points.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include "../../include/ssasr.h"
#include "common.h"

namespace {

SsasrOptions g_opt;
std::once_flag g_opt_once;

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
int env_flag(const char* name) { return getenv(name) != nullptr; }

void init_options() {
  g_opt.no_persistent = env_flag("SSASR_NO_PERSISTENT");
  g_opt.no_fused_input = env_flag("SSASR_NO_FUSED_INPUT");
  g_opt.bptt_halves_off = env_flag("SSASR_BPTT_HALVES_OFF");
  g_opt.bptt_reserve_kb = env_flag("SSASR_BPTT_SHARED_CU") ? 0 : env_int("SSASR_BPTT_RESERVE_KB", 118);
  g_opt.no_persistent_decoder = env_flag("SSASR_NO_PERSISTENT_DECODER");
  g_opt.no_persistent_decoder_bwd = env_flag("SSASR_NO_PERSISTENT_DECODER_BWD");
  g_opt.delay_fwd = env_int("SSASR_PERSIST_DELAY_FWD", -1);
  g_opt.delay_bwd = env_int("SSASR_PERSIST_DELAY_BWD", -1);
  g_opt.delay_bwd_ksplit = g_opt.delay_bwd;
  g_opt.gemm_tile = env_int("SSASR_GEMM_TILE", 0);
  g_opt.gemm_x6 = env_int("SSASR_GEMM_X6", 1);
  g_opt.gemm_wide = env_int("SSASR_GEMM_WIDE", 1);
  g_opt.gemm_bf16 = env_int("SSASR_GEMM_BF16", 0);
  g_opt.gemm_trace_lo = g_opt.gemm_trace_hi = 0;
  g_opt.gemm_kcat = env_int("SSASR_GEMM_KCAT", 1);
  g_opt.wgrad_fused = env_int("SSASR_WGRAD_FUSED", 1);
  g_opt.bptt_one_launch = env_int("SSASR_BPTT_ONE_LAUNCH", 1);
  g_opt.no_windows = env_flag("SSASR_NO_WINDOWS");
  g_opt.last_seg_pct = env_int("SSASR_LAST_SEG_PCT", 60);
  g_opt.tail_inline = env_int("SSASR_TAIL_INLINE", 1);
  g_opt.no_residency_check = env_flag("SSASR_NO_RESIDENCY_CHECK");
  g_opt.test_drop_tile = env_int("SSASR_TEST_DROP_TILE", -1);
  g_opt.test_drop_attn_slice = env_int("SSASR_TEST_DROP_ATTN_SLICE", -1);
  g_opt.test_drop_dec_slice = env_int("SSASR_TEST_DROP_DEC_SLICE", -1);
  g_opt.attn_rph = env_int("SSASR_ATTN_RPH", 0);
  if (g_opt.attn_rph != 2 && g_opt.attn_rph != 3 && g_opt.attn_rph != 4 && g_opt.attn_rph != 6) g_opt.attn_rph = 0;
  g_opt.no_tsave = env_flag("SSASR_NO_TSAVE");
}

struct Named { const char* name; int SsasrOptions::*field; };
const Named kNames[] = {
    {"SSASR_NO_PERSISTENT", &SsasrOptions::no_persistent},
    {"SSASR_NO_FUSED_INPUT", &SsasrOptions::no_fused_input},
    {"SSASR_BPTT_HALVES_OFF", &SsasrOptions::bptt_halves_off},
    {"SSASR_BPTT_RESERVE_KB", &SsasrOptions::bptt_reserve_kb},
    {"SSASR_NO_PERSISTENT_DECODER", &SsasrOptions::no_persistent_decoder},
    {"SSASR_NO_PERSISTENT_DECODER_BWD", &SsasrOptions::no_persistent_decoder_bwd},
    {"SSASR_PERSIST_DELAY_FWD", &SsasrOptions::delay_fwd},
    {"SSASR_PERSIST_DELAY_BWD", &SsasrOptions::delay_bwd},
    {"SSASR_GEMM_TILE", &SsasrOptions::gemm_tile},
    {"SSASR_GEMM_TRACE_LO", &SsasrOptions::gemm_trace_lo},
    {"SSASR_GEMM_TRACE_HI", &SsasrOptions::gemm_trace_hi},
    {"SSASR_GEMM_X6", &SsasrOptions::gemm_x6},
    {"SSASR_GEMM_WIDE", &SsasrOptions::gemm_wide},
    {"SSASR_GEMM_BF16", &SsasrOptions::gemm_bf16},
    {"SSASR_GEMM_KCAT", &SsasrOptions::gemm_kcat},
    {"SSASR_WGRAD_FUSED", &SsasrOptions::wgrad_fused},
    {"SSASR_BPTT_ONE_LAUNCH", &SsasrOptions::bptt_one_launch},
    {"SSASR_NO_WINDOWS", &SsasrOptions::no_windows},
    {"SSASR_LAST_SEG_PCT", &SsasrOptions::last_seg_pct},
    {"SSASR_TAIL_INLINE", &SsasrOptions::tail_inline},
    {"SSASR_NO_RESIDENCY_CHECK", &SsasrOptions::no_residency_check},
    {"SSASR_TEST_DROP_TILE", &SsasrOptions::test_drop_tile},
    {"SSASR_TEST_DROP_ATTN_SLICE", &SsasrOptions::test_drop_attn_slice},
    {"SSASR_TEST_DROP_DEC_SLICE", &SsasrOptions::test_drop_dec_slice},
    {"SSASR_ATTN_RPH", &SsasrOptions::attn_rph},
    {"SSASR_NO_TSAVE", &SsasrOptions::no_tsave},
};

}  // namespace

const SsasrOptions& ssasr_options() {
  std::call_once(g_opt_once, init_options);
  return g_opt;
}

extern "C" int ssasr_set_option(const char* name, int value) {
  if (!name) return SSASR_EARG;
  std::call_once(g_opt_once, init_options);
  for (const Named& n : kNames)
    if (strcmp(n.name, name) == 0) {
      g_opt.*(n.field) = value;
      if (n.field == &SsasrOptions::delay_bwd) g_opt.delay_bwd_ksplit = value;
      return SSASR_OK;
    }
  return SSASR_EARG;
}

extern "C" int ssasr_get_option(const char* name, int* value) {
  if (!name || !value) return SSASR_EARG;
  std::call_once(g_opt_once, init_options);
  for (const Named& n : kNames)
    if (strcmp(n.name, name) == 0) { *value = g_opt.*(n.field); return SSASR_OK; }
  return SSASR_EARG;
}

size_t ssasr_lds_reservation_against_gemm(const void* kernel) {
  constexpr size_t CU_LDS = 160 * 1024;
  hipFuncAttributes fa{};
  if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) return 0;
  const size_t own = fa.sharedSizeBytes, gemm = ssasr_gemm_min_lds_bytes();
  if (own + gemm > CU_LDS) return 0;
  // own + reserve + gemm must exceed the CU's LDS; LDS is granted in 1 KB steps on this part: one step of margin
  return CU_LDS - own - gemm + 1024;
}

int64_t ssasr_resident_capacity(const void* kernel, int threads, size_t dyn_lds) {
  static std::mutex mu;
  static std::map<std::tuple<const void*, int, size_t, int>, int64_t> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple(kernel, threads, dyn_lds, dev);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dyn_lds) != hipSuccess) per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  // MI355X_MICROARCH.md: the API answer can be one block per CU high near SGPR edges; the persistent
  // kernels here are 320- or 256-thread, register-heavy workgroups (1-2 per CU), where it is exact.
  const int64_t cap = (int64_t)per_cu * cus;
  cache[key] = cap;
  return cap;
}
