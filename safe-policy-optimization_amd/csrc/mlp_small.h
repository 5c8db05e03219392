// One-launch forward / backward of wide-path networks at small row counts (csrc/mlp_small.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/safepo_hip.h"

namespace spo {

constexpr int MLP_SMALL_MAX_ROWS = 128;            // 8 waves of 16 rows
constexpr size_t MLP_SMALL_MAX_LDS = 156 * 1024;   // two row images [rows][widest layer + 4] + a weight / input chunk
constexpr int MLP_SMALL_MAX_NETS = 4;              // networks of one launch (one workgroup each)

struct MlpSmallArgs {
  const float* theta;      // the network's slice of the flat parameter vector (nn.Linear order: W0, b0, W1, b1, ...)
  const float* x;          // [rows][d[0]]
  float* ws;               // activations h_1 .. h_n ([rows][d[l + 1]] each), the layout of spo_mlp_forward
  const float* d_out;      // backward: d(loss)/d(output) [rows][d[n]]
  float* grad;             // backward: the network's flat gradient (same layout as theta)
  int n, rows, ldm, big_floats;
  int d[SPO_MLP_MAX_LAYERS + 1];
  int64_t w[SPO_MLP_MAX_LAYERS], b[SPO_MLP_MAX_LAYERS], act[SPO_MLP_MAX_LAYERS];
};
struct MlpSmallBatch {
  MlpSmallArgs a[MLP_SMALL_MAX_NETS];
  int count;
};

bool mlp_small_ok(const spo_mlp_net* net, int64_t rows);
int mlp_small_args(const float* theta, const spo_mlp_net* net, const float* x, int64_t rows, float* ws, const float* d_out, float* grad,
                   MlpSmallArgs* out);
int mlp_small_launch(bool backward, const MlpSmallBatch& batch, hipStream_t st);

}  // namespace spo
