// Error state of the C ABI: thread-local last-error string, never throws.
#include <stdio.h>
#include <string.h>
#include "common.h"
#include "hamer_hip_internal.h"

static thread_local char g_err[512] = "";

int hm_set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}

int hm_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return HM_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return HM_ERR_HIP;
}

int hm_device_cu_count(void) {
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 64 && cus[dev] > 0) return cus[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  if (dev < 64) cus[dev] = n;
  return n;
}

// Test / tuning switches (all default 0): plain ints behind explicit setters (relaxed atomics: a setter racing a launch on another host thread
// changes which kernel that launch takes, never its result).
#include <atomic>
static std::atomic<int> g_opts[HM_OPT_COUNT];
int hm_option(int key) {
  return (key >= 0 && key < HM_OPT_COUNT) ? g_opts[key].load(std::memory_order_relaxed) : 0;
}
extern "C" int hm_get_option(int key) { return hm_option(key); }
extern "C" int hm_option_count(void) { return HM_OPT_COUNT; }
extern "C" int hm_set_option(int key, int value) {
  if (key < 0 || key >= HM_OPT_COUNT) return hm_set_error(HM_ERR_ARG, "hm_set_option: unknown key");
  if (value < 0) return hm_set_error(HM_ERR_ARG, "hm_set_option: value must be >= 0");
  if ((key == HM_OPT_PX_GRID || key == HM_OPT_FP8P_GRID) && value != 0 && (value < 8 || value % 8 != 0))
    return hm_set_error(HM_ERR_ARG, "hm_set_option: a persistent grid is 0 (default) or a multiple of the 8 XCDs");
  g_opts[key].store(value, std::memory_order_relaxed);
  return HM_OK;
}

extern "C" int hm_version(void) { return HM_VERSION; }
extern "C" const char* hm_last_error_string(void) { return g_err; }
