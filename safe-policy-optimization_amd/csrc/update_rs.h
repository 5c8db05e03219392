// Internal interface of the row-split persistent update kernel (update_rs.hip), called from update.hip's entry points.
#pragma once
#include <cstddef>
#include <cstdint>
#include "../../include/safepo_hip.h"

namespace spo {
// n_nets 3: one learning iteration of the PPO-Lagrangian step (clipped surrogate); 2: the critic fit (act / logp_old / adv unused).
// prof: optional device [3][12] u64 cycle accumulators (instrumented instantiation, obs_dim in (32, 64], batch <= 64).
int rs_update_launch(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs, const float* act,
                     const float* logp_old, const float* target_r, const float* target_c, const float* adv, const int32_t* perm,
                     int64_t M, const spo_ppo_cfg* cfg_host, int n_nets, float* stale_sq_io, float* losses_out, void* sync_ws,
                     unsigned long long* prof, void* stream);
// one rank of a data-parallel job (update.hip: spo_ppo_lag_update_iter_dp with SPO_XR_FORM_ROW_SPLIT); RSX_BYTES of every rank's
// exchange region at offset rsx_off belong to this form: [parity 2][row group 2][network 3][source rank 8][word 16][lane 256] x 16 B
constexpr size_t RSX_BYTES = (size_t)2 * 2 * 3 * 8 * 16 * 4096;
int rs_update_launch_dp(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs, const float* act,
                        const float* logp_old, const float* target_r, const float* target_c, const float* adv, const int32_t* perm,
                        int64_t M, const spo_ppo_cfg* cfg_host, float* losses_out, void* sync_ws, int rank, int world,
                        void* const* regions, uint32_t step0, size_t rsx_off, void* stream);
// {minibatch steps run, steps redone after a late clip verdict} of the row-split kernel since the last reset
int rs_counters(unsigned long long* out2_host, int reset);
}  // namespace spo
