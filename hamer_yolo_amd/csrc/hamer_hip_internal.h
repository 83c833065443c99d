// Internal helpers shared by the translation units of libhamer_hip.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hamer_hip.h"

int hm_set_error(int code, const char* msg);
int hm_check_launch(const char* what);   // hipGetLastError() -> status

// profiling hooks (prof.hip): push returns a record index or -1 when profiling is off
int hm_prof_push(int kind, int epilogue, int M, int N, int K, hipStream_t s);
void hm_prof_pop(int idx, hipStream_t s);
struct HmProfScope {
  int idx; hipStream_t s;
  HmProfScope(int kind, int epi, int M, int N, int K, hipStream_t st) : idx(hm_prof_push(kind, epi, M, N, K, st)), s(st) {}
  ~HmProfScope() { hm_prof_pop(idx, s); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device), safe under concurrent host
// threads: one static HmLdsOnce per launch site.  Returns HM_OK or sets the error string.
#include <mutex>
struct HmLdsOnce {
  std::mutex mu;
  unsigned long long done = 0;      // bit d set: device d has the attribute (devices >= 64: set on every call)
  int ensure(const void* kernel, int lds_bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hm_set_error(HM_ERR_HIP, what);
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 64 && ((done >> dev) & 1ull)) return HM_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return hm_set_error(HM_ERR_HIP, what);
    if (dev < 64) done |= 1ull << dev;
    return HM_OK;
  }
};
int hm_device_cu_count(void);
int hm_option(int key);               // status.hip: value of an HM_OPT_* switch (0 when never set)
// attention.hip: the MFMA attention kernel with a runtime token count and proportional attention (used by hm_tome_attention)
int hm_attention_tome_launch(const void* qkv, const float* size, void* out, int B, int tokens, int heads, float scale, int dtype,
                             hipStream_t s);   // multiProcessorCount of the current device (cached per device), <= 0 on error
