// Shared host-side helpers for libsafepo_hip.so (gfx950 only; no CUDA paths).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

namespace spo {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// hipFuncSetAttribute is per device: "already done" flags are kept per device ordinal (round-2 advice)
constexpr int SPO_MAX_DEVICES = 64;
inline int current_device_slot() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d >= 0 && d < SPO_MAX_DEVICES) ? d : 0;
}

inline int hip_check(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

#define SPO_LAUNCH_CHECK(what)                                    \
  do {                                                            \
    int _rc = ::spo::hip_check(hipGetLastError(), what);          \
    if (_rc) return _rc;                                          \
  } while (0)

#define SPO_REQUIRE(cond, ...)                                    \
  do {                                                            \
    if (!(cond)) return ::spo::fail(-1, __VA_ARGS__);             \
  } while (0)

// exchange scratch of the feature-split kernels (update_ks.hip), freed together with the update kernels' by spo_update_scratch_release
int ks_scratch_release(int dev, void* stream_or_null, int all);
// ... and of the row-split kernel (update_rs.hip)
int rs_scratch_release(int dev, void* stream_or_null, int all);
}  // namespace spo
