// Optional per-launch timing with HIP events recorded on the launch stream (used by bench.py
// for the live roofline numbers).  Off by default: when disabled the hooks are two branches.
// Explicit lifetime: hm_prof_begin() creates the event pool, hm_prof_end() destroys it.
#include <vector>
#include "common.h"
#include "hamer_hip_internal.h"

namespace {
struct Rec { int kind, epilogue, M, N, K; hipEvent_t e0, e1; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_used = 0;
bool g_on = false;
}  // namespace

int hm_prof_push(int kind, int epilogue, int M, int N, int K, hipStream_t s) {
  if (!g_on || g_used + 2 > g_pool.size()) return -1;
  Rec r{kind, epilogue, M, N, K, g_pool[g_used], g_pool[g_used + 1]};
  g_used += 2;
  if (hipEventRecord(r.e0, s) != hipSuccess) return -1;
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void hm_prof_pop(int idx, hipStream_t s) {
  if (idx >= 0 && idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].e1, s);
}

extern "C" int hm_prof_begin(int capacity) {
  if (g_on) return hm_set_error(HM_ERR_ARG, "hm_prof_begin: already profiling");
  if (capacity <= 0) return hm_set_error(HM_ERR_ARG, "hm_prof_begin: capacity must be positive");
  g_pool.resize((size_t)capacity * 2);
  for (auto& e : g_pool)
    if (hipEventCreate(&e) != hipSuccess) return hm_set_error(HM_ERR_HIP, "hm_prof_begin: hipEventCreate failed");
  g_recs.clear(); g_recs.reserve(capacity);
  g_used = 0; g_on = true;
  return HM_OK;
}

// Waits for the recorded events, writes up to `cap` records, returns the count and clears the log.
extern "C" int hm_prof_collect(hm_prof_record* out, int cap) {
  if (!g_on) return hm_set_error(HM_ERR_ARG, "hm_prof_collect: not profiling");
  int n = 0;
  for (auto& r : g_recs) {
    if (n >= cap) break;
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess)
      return hm_set_error(HM_ERR_HIP, "hm_prof_collect: event query failed");
    if (out) out[n] = hm_prof_record{r.kind, r.epilogue, r.M, r.N, r.K, ms};
    ++n;
  }
  g_recs.clear(); g_used = 0;
  return n;
}

extern "C" int hm_prof_end(void) {
  for (auto& e : g_pool) (void)hipEventDestroy(e);
  g_pool.clear(); g_recs.clear(); g_used = 0; g_on = false;
  return HM_OK;
}
