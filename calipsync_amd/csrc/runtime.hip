// Process-wide runtime state of the engine library: the option table (environment read once), the
// per-thread "options of the call in progress", and the per-(kernel, device) launch attributes.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace {

struct OptField {
  const char* name;   // option name == environment variable without the CASYNC_ prefix, lower case
  int CasyncOptions::*field;
};
const OptField kFields[] = {
    {"lanes", &CasyncOptions::lanes},
    {"trunk_lanes", &CasyncOptions::trunk_lanes},
    {"lane_skew", &CasyncOptions::lane_skew},
    {"overlap", &CasyncOptions::overlap},
    {"gemm_streamk", &CasyncOptions::gemm_streamk},
    {"gemm_glds", &CasyncOptions::gemm_glds},
    {"gemm_cfg", &CasyncOptions::gemm_cfg},
    {"gemm_ring128", &CasyncOptions::gemm_ring128},
    {"gemm_small_m", &CasyncOptions::gemm_small_m},
    {"skip_early", &CasyncOptions::skip_early},
    {"gemm_single64", &CasyncOptions::gemm_single64},
    {"gemm_persist", &CasyncOptions::gemm_persist},
    {"lane_streamk", &CasyncOptions::lane_streamk},
    {"gemm_conc", &CasyncOptions::gemm_conc},
    {"gemm_conc_tiles", &CasyncOptions::gemm_conc_tiles},
    {"fuse_ir", &CasyncOptions::fuse_ir},
    {"fuse_up", &CasyncOptions::fuse_up},
    {"fuse_min_hw", &CasyncOptions::fuse_min_hw},
    {"fuse_q", &CasyncOptions::fuse_q},
    {"ups_commute", &CasyncOptions::ups_commute},
    {"fuse_dw", &CasyncOptions::fuse_dw},
    {"fuse_dw_min", &CasyncOptions::fuse_dw_min},
    {"fuse_dw_min40", &CasyncOptions::fuse_dw_min40},
    {"fuse_dw_deep", &CasyncOptions::fuse_dw_deep},
    {"att_bf16", &CasyncOptions::att_bf16},
    {"ir_dw_mfma", &CasyncOptions::ir_dw_mfma},
    {"bf16_plan", &CasyncOptions::bf16_plan},
    {"inc_mfma", &CasyncOptions::inc_mfma},
    {"ups_commute_bf16", &CasyncOptions::ups_commute_bf16},
    {"fuse_dw_bf16", &CasyncOptions::fuse_dw_bf16},
    {"fuse_dw_bf16_bn", &CasyncOptions::fuse_dw_bf16_bn},
    {"fuse_dw_bf16_min", &CasyncOptions::fuse_dw_bf16_min},
    {"dw_lds", &CasyncOptions::dw_lds},
    {"dw_lds_bytes", &CasyncOptions::dw_lds_bytes},
    {"att_nz", &CasyncOptions::att_nz},
    {"kv_early", &CasyncOptions::kv_early},
};

thread_local const CasyncOptions* t_current = nullptr;

}  // namespace

CasyncOptions& casync_default_options() {
  static CasyncOptions defaults;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const OptField& f : kFields) {
      char env[64] = "CASYNC_";
      size_t n = strlen(env);
      for (const char* p = f.name; *p && n + 1 < sizeof(env); ++p) env[n++] = (char)(*p >= 'a' && *p <= 'z' ? *p - 32 : *p);
      env[n] = 0;
      const char* v = getenv(env);
      if (v && *v) defaults.*(f.field) = atoi(v);
    }
  });
  return defaults;
}

const CasyncOptions& casync_opts() { return t_current ? *t_current : casync_default_options(); }

int casync_option_ref(CasyncOptions& o, const char* name, int** slot) {
  if (name)
    for (const OptField& f : kFields)
      if (strcmp(f.name, name) == 0) {
        *slot = &(o.*(f.field));
        return CASYNC_OK;
      }
  casync_set_error("unknown option '%s'", name ? name : "(null)");
  return CASYNC_ERR_ARG;
}

CasyncOptScope::CasyncOptScope(const CasyncOptions* o) : prev(t_current) { t_current = o; }
CasyncOptScope::~CasyncOptScope() { t_current = prev; }

int casync_ensure_dyn_lds(unsigned long long* once_mask, const void* fn, int bytes) {
  int dev = 0;
  CASYNC_CHECK_HIP(hipGetDevice(&dev));
  auto* mask = reinterpret_cast<std::atomic<unsigned long long>*>(once_mask);
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0;
  if (bit && (mask->load(std::memory_order_acquire) & bit)) return CASYNC_OK;
  CASYNC_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (bit) mask->fetch_or(bit, std::memory_order_release);
  return CASYNC_OK;
}
