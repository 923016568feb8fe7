// Identity of the sources this library was built from: sha256 over every file of csrc/ and include/opp_hip.h, passed in by
// onepose_plus_plus_amd/build.py (-DOPP_SRC_HASH).  onepose_plus_plus_amd/_lib.py recomputes it from the sources next to the
// .so and refuses a stale binary (the GPU box runs the prebuilt library that travels with the repository snapshot).
#include "opp_common.h"

#ifndef OPP_SRC_HASH
#define OPP_SRC_HASH "unknown"
#endif

extern "C" const char* opp_source_hash(void) { return OPP_SRC_HASH; }

extern "C" int opp_supports_precision(int p) {
#ifdef OPP_TUNING
  return p >= 0 && p <= 3;
#else
  return p == 0 || p == 3;      // fp32, bf16x3: the arithmetics not narrower than the reference's fp32
#endif
}

// A kernel that does nothing: opp_profile_event_overhead times it exactly like every armed symbol (an event before, an event
// after, on the launch stream), which measures what the event pair itself adds to a short launch.
namespace {
__global__ void opp_empty_kernel() {}
// one wave busy for `ticks` of the constant-rate wall clock (100 MHz on gfx950: 100 ticks = 1 us): a kernel of KNOWN length for the calibration below
__global__ void opp_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
}
}  // namespace

extern "C" int opp_profile_event_overhead(int launches, double* mean_us, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OPP_CHECK_ARG(launches > 0 && launches <= 4096 && mean_us, "profile_event_overhead: bad argument");
  // like the armed symbols of a forward: all pairs enqueued back to back on a busy stream, one synchronisation at the end
  const int n = launches + 8;          // the first launches warm the code object up and are not counted
  hipEvent_t* ev = new hipEvent_t[2 * n];
  int made = 0;
  for (; made < 2 * n; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) break;
  int rc = OPP_OK;
  if (made < 2 * n) {
    opp_set_error("profile_event_overhead: hipEventCreate failed");
    rc = OPP_ERR_LAUNCH;
  } else {
    for (int i = 0; i < n; ++i) {
      (void)hipEventRecord(ev[2 * i], stream);
      hipLaunchKernelGGL(opp_empty_kernel, dim3(1), dim3(64), 0, stream);
      (void)hipEventRecord(ev[2 * i + 1], stream);
    }
    if (hipEventSynchronize(ev[2 * n - 1]) != hipSuccess) {
      opp_set_error("profile_event_overhead: hipEventSynchronize failed");
      rc = OPP_ERR_LAUNCH;
    } else {
      double total = 0.0;
      for (int i = 8; i < n; ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        total += ms;
      }
      *mean_us = total * 1e3 / launches;
    }
  }
  for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

// What an event pair adds to the reading of a REAL launch (r05): a spin kernel of ~spin_us is timed both ways on the same stream -- every launch
// between its own event pair (*pair_us = mean reading) and `launches` of them back to back between ONE pair (*b2b_us = launch-to-launch
// interval = the kernel's duration + the inter-kernel gap, i.e. slightly MORE than a kernel trace shows for it).  pair_us - b2b_us is therefore a
// slightly conservative estimate of the part of an event reading that is not the kernel; bench.py subtracts it for `frac_event_corrected`.
// (An EMPTY kernel is a poor probe: its pair reading is dominated by launch latency that real kernels overlap -- it over-corrects by ~2 us.)
extern "C" int opp_profile_event_calibration(int launches, double spin_us, double* pair_us, double* b2b_us, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OPP_CHECK_ARG(launches > 0 && launches <= 2048 && spin_us > 0.0 && spin_us <= 1000.0 && pair_us && b2b_us, "profile_event_calibration: bad argument");
  const long long ticks = (long long)(spin_us * 100.0);
  const int n = launches + 4;
  hipEvent_t* ev = new hipEvent_t[2 * n + 2];
  int made = 0;
  for (; made < 2 * n + 2; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) break;
  int rc = OPP_OK;
  if (made < 2 * n + 2) {
    opp_set_error("profile_event_calibration: hipEventCreate failed");
    rc = OPP_ERR_LAUNCH;
  } else {
    for (int i = 0; i < n; ++i) {
      (void)hipEventRecord(ev[2 * i], stream);
      hipLaunchKernelGGL(opp_spin_kernel, dim3(1), dim3(64), 0, stream, ticks);
      (void)hipEventRecord(ev[2 * i + 1], stream);
    }
    (void)hipEventRecord(ev[2 * n], stream);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(opp_spin_kernel, dim3(1), dim3(64), 0, stream, ticks);
    (void)hipEventRecord(ev[2 * n + 1], stream);
    if (hipEventSynchronize(ev[2 * n + 1]) != hipSuccess) {
      opp_set_error("profile_event_calibration: hipEventSynchronize failed");
      rc = OPP_ERR_LAUNCH;
    } else {
      double total = 0.0;
      for (int i = 4; i < n; ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        total += ms;
      }
      *pair_us = total * 1e3 / launches;
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[2 * n], ev[2 * n + 1]);
      *b2b_us = (double)ms * 1e3 / launches;
    }
  }
  for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

// The empty kernel's own back-to-back interval (kept for reference beside the calibration above).
extern "C" int opp_profile_empty_kernel(int launches, double* mean_us, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OPP_CHECK_ARG(launches > 0 && launches <= 65536 && mean_us, "profile_empty_kernel: bad argument");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    opp_set_error("profile_empty_kernel: hipEventCreate failed");
    return OPP_ERR_LAUNCH;
  }
  for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(opp_empty_kernel, dim3(1), dim3(64), 0, stream);
  (void)hipEventRecord(e0, stream);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(opp_empty_kernel, dim3(1), dim3(64), 0, stream);
  (void)hipEventRecord(e1, stream);
  int rc = OPP_OK;
  if (hipEventSynchronize(e1) != hipSuccess) {
    opp_set_error("profile_empty_kernel: hipEventSynchronize failed");
    rc = OPP_ERR_LAUNCH;
  } else {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *mean_us = (double)ms * 1e3 / launches;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}
