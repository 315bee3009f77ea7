// Error reporting + version of the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void adell_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* adell_last_error(void) { return g_err; }
extern "C" int adell_abi_version(void) { return ADELL_ABI_VERSION; }

// ---- launch-plan switches -------------------------------------------------------------------
#include <stdlib.h>
#include <string.h>

static int adell_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
static int adell_env_set(const char* name) { return getenv(name) != nullptr; }

static AdellTuning adell_tuning_from_env() {
  AdellTuning t;
  t.igemm_nospec = adell_env_set("ADELL_IGEMM_NOSPEC");
  t.igemm_no8 = adell_env_set("ADELL_IGEMM_NO8");
  t.no_splitk = adell_env_set("ADELL_NO_SPLITK");
  t.wgrad_nozring = adell_env_set("ADELL_WGRAD_NOZRING");
  t.attn_nomfma = adell_env_set("ADELL_ATTN_NOMFMA");
  t.dw_nomfma = adell_env_int("ADELL_DW_NOMFMA", 0);
  t.dw_wgrad_nomfma = adell_env_int("ADELL_DW_WGRAD_NOMFMA", 0);
  t.wgrad_no16 = adell_env_int("ADELL_WGRAD_NO16", 0);
  t.igemm_no16 = adell_env_int("ADELL_IGEMM_NO16", 0);
  t.gemm_norows = adell_env_int("ADELL_GEMM_NOROWS", 0);
#ifdef ADELL_DEBUG
  t.igemm_dbg = adell_env_int("ADELL_IGEMM_DBG", 0);
  t.zr_dbg = adell_env_int("ADELL_ZR_DBG", 0);
#else
  t.igemm_dbg = 0;
  t.zr_dbg = 0;
#endif
  return t;
}
AdellTuning g_adell_tune = adell_tuning_from_env();
long g_adell_plan_epoch = 0;
extern "C" long adell_plan_epoch(void) { return g_adell_plan_epoch; }

// ten switches, each selecting another built path that the parity tests compare with the default
// (variants that lost their A/B measurements are in the history, not in the library)
static int* adell_tuning_slot(const char* name) {
  if (!name) return nullptr;
  if (!strcmp(name, "igemm_nospec")) return &g_adell_tune.igemm_nospec;
  if (!strcmp(name, "igemm_no8")) return &g_adell_tune.igemm_no8;
  if (!strcmp(name, "no_splitk")) return &g_adell_tune.no_splitk;
  if (!strcmp(name, "wgrad_nozring")) return &g_adell_tune.wgrad_nozring;
  if (!strcmp(name, "attn_nomfma")) return &g_adell_tune.attn_nomfma;
  if (!strcmp(name, "dw_nomfma")) return &g_adell_tune.dw_nomfma;
  if (!strcmp(name, "dw_wgrad_nomfma")) return &g_adell_tune.dw_wgrad_nomfma;
  if (!strcmp(name, "wgrad_no16")) return &g_adell_tune.wgrad_no16;
  if (!strcmp(name, "igemm_no16")) return &g_adell_tune.igemm_no16;
  if (!strcmp(name, "gemm_norows")) return &g_adell_tune.gemm_norows;
#ifdef ADELL_DEBUG
  if (!strcmp(name, "igemm_dbg")) return &g_adell_tune.igemm_dbg;
  if (!strcmp(name, "zr_dbg")) return &g_adell_tune.zr_dbg;
#endif
  return nullptr;
}

extern "C" int adell_set_tuning(const char* name, int value) {
  int* slot = adell_tuning_slot(name);
  if (!slot) {
    adell_set_error("adell_set_tuning: unknown switch '%s'", name ? name : "(null)");
    return ADELL_E_BADARG;
  }
  if (*slot != value) ++g_adell_plan_epoch;
  *slot = value;
  return ADELL_OK;
}

extern "C" int adell_get_tuning(const char* name) {
  int* slot = adell_tuning_slot(name);
  return slot ? *slot : -1;
}

// ---- replay counter of the dropout offsets (common.h: ADELL_RNG_STEP_DEFINE) ------------------------
int adell_rng_advance_norm_act(unsigned delta, int set, hipStream_t st);
int adell_rng_advance_tokens(unsigned delta, int set, hipStream_t st);
int adell_rng_advance_window(unsigned delta, int set, hipStream_t st);

extern "C" int adell_rng_advance(uint32_t delta, int set, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = adell_rng_advance_norm_act(delta, set, st);
  if (rc == ADELL_OK) rc = adell_rng_advance_tokens(delta, set, st);
  if (rc == ADELL_OK) rc = adell_rng_advance_window(delta, set, st);
  if (rc != ADELL_OK) adell_set_error("adell_rng_advance: launch failed");
  return rc;
}
