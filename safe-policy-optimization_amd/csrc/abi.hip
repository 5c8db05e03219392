#include "common.h"
#include "../../include/safepo_hip.h"
namespace spo { thread_local char g_err[512] = {0}; }
extern "C" int spo_abi_version(void) { return SPO_ABI_VERSION; }
extern "C" const char* spo_last_error(void) { return spo::g_err; }

#include "mlp_mfma.h"
namespace {
__global__ void crosslane_selftest_kernel(const float* in, float* out) {
  const int l = threadIdx.x;
  const float x = in[l];
  out[l] = spo::quad_row_sum(x);
  out[64 + l] = spo::row_sum_lane15(x);
  out[128 + l] = spo::wave_sum_lane63(x);
}
}  // namespace
// Debug self-test of the DPP / permlane helpers: in[64] -> out[192] = {quad_row_sum, row_sum_lane15, wave_sum_lane63}.
extern "C" int spo_debug_crosslane_selftest(const float* in64, float* out192, void* stream) {
  hipLaunchKernelGGL(crosslane_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in64, out192);
  return spo::hip_check(hipGetLastError(), "crosslane_selftest");
}
