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

extern "C" int hm_version(void) { return HM_VERSION; }
extern "C" const char* hm_last_error_string(void) { return g_err; }
