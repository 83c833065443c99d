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
