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
// A pointer's value in a scalar-register pair of its own.  The persistent kernels' arguments arrive as wide scalar loads
// (s_load_dwordx8 / x16): register TUPLES that the allocator can only spill and restore whole -- 16 v_readlane to get at one
// 64-bit pointer (update_rs.hip, round 6: 656 of the step loop's 766).  A plain copy, or an empty asm on an "s" operand, is
// coalesced back into the tuple; a trip through a VGPR is not.  The result is cast back into the global address space (a pointer
// rebuilt from integers would otherwise be dereferenced with flat_* instructions).
#ifdef __HIPCC__
template <class T>
__device__ __forceinline__ T* own_sgprs(T* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  unsigned vlo, vhi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(vlo), "=v"(vhi) : "s"((unsigned)u), "s"((unsigned)(u >> 32)));
  const unsigned lo = __builtin_amdgcn_readfirstlane(vlo), hi = __builtin_amdgcn_readfirstlane(vhi);
  typedef T __attribute__((address_space(1))) * gp;
  return (T*)(gp)(((unsigned long long)hi << 32) | lo);
}
#endif
}  // namespace spo
