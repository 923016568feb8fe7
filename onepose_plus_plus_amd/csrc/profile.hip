// Live timing of ONE kernel symbol with HIP events recorded on the launch stream (bench.py roofline leg,
// include/opp_hip.h opp_profile_start / opp_profile_stop).  A launcher wraps its hipLaunchKernelGGL in an
// OppProfScope(symbol, stream, work): when that symbol is armed the scope records an event before and after
// the launch and adds `work` (algorithmic FLOPs, or algorithmic bytes for the bandwidth-bound kernels).
#include <mutex>
#include <vector>

#include "opp_internal.h"

namespace {
struct Profiler {
  bool on = false;
  int symbol = -1;
  std::vector<hipEvent_t> ev;   // pairs (start, stop)
  size_t used = 0;
  double work = 0.0;
  long long dropped = 0;
  std::mutex mu;                // forwards may be in flight from several host threads / streams
} g_prof;
}  // namespace

OppProfScope::OppProfScope(int symbol, hipStream_t stream, double work) : stream_(stream) {
  if (!g_prof.on || g_prof.symbol != symbol) return;   // unlocked fast path: the flags only flip between forwards
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (g_prof.on && g_prof.symbol == symbol) {
    if (g_prof.used + 2 <= g_prof.ev.size()) {
      slot_ = (long long)g_prof.used;
      g_prof.used += 2;
      g_prof.work += work;
      (void)hipEventRecord(g_prof.ev[(size_t)slot_], stream_);
    } else {
      g_prof.dropped++;
    }
  }
}

OppProfScope::~OppProfScope() {
  if (slot_ >= 0) (void)hipEventRecord(g_prof.ev[(size_t)slot_ + 1], stream_);
}

extern "C" int opp_profile_start(int tile_cfg, int kind, int capacity) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.ev.resize((size_t)capacity * 2);
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) {
      opp_set_error("profile: hipEventCreate failed");
      return OPP_ERR_LAUNCH;
    }
  g_prof.used = 0;
  g_prof.work = 0.0;
  g_prof.dropped = 0;
  g_prof.symbol = tile_cfg >= OPP_PROF_FIRST_NON_GEMM ? tile_cfg : opp_prof_gemm_symbol(tile_cfg, kind);
  g_prof.on = true;
  return OPP_OK;
}

extern "C" int opp_profile_stop(double* total_ms, double* total_work, int* launches) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.on = false;
  double ms = 0.0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) {
      opp_set_error("profile: hipEventSynchronize failed");
      return OPP_ERR_LAUNCH;
    }
    float t = 0.f;
    (void)hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]);
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (total_work) *total_work = g_prof.work;
  if (launches) *launches = (int)(g_prof.used / 2);
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.used = 0;
  return OPP_OK;
}
