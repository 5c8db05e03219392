// Multi-agent (MAPPO-L family) masked GAE with PopArt de-normalisation, gfx950.   [SURVEY.md 8 f3, first piece]
//
// Replaces SeparatedReplayBuffer.compute_returns + compute_cost_returns
// (reference safepo/common/buffer.py:356-384): a Python loop over episode_length with ~10 tiny torch ops per
// step per agent.  Layout is the reference's own: time-major [T+1, N, 1] value/mask arrays, [T, N, 1] rewards.
// One lane per rollout thread walks t = T-1..0 in fp32 with the reference's exact operation order
//     dn(x)  = x * sqrt(var) + mean                               (PopArt.denormalize, popart.py:117-133)
//     delta  = r_t + gamma * dn(v_{t+1}) * mask_{t+1} - dn(v_t)
//     gae    = delta + gamma*lambda * mask_{t+1} * gae
//     ret_t  = gae + dn(v_t)
// so results are BIT-IDENTICAL to the reference; every row access is coalesced across rollout threads.
// HBM-bound: 20 B read (r, c, v_r, v_c, mask) + 8 B written per (thread, step); loads are issued UNROLL steps ahead.
#include "common.h"
#include "../../include/safepo_hip.h"

namespace {

struct MaArgs {
  const float* rewards; const float* costs;       // [T, N]
  const float* value_preds; const float* cost_preds; const float* masks;   // [T+1, N]
  float* returns; float* cost_returns;            // [T+1, N] (row T untouched, as in the reference)
  int64_t T; int64_t N;
  float gamma, gl;                                // gamma, fp32(gamma*lambda formed in double)
  float sd_r, mu_r, sd_c, mu_c;                   // sqrt(var), mean of the two PopArt normalisers
};

template <int UNROLL>
__global__ __launch_bounds__(256) void ma_gae_kernel(MaArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  const int64_t N = a.N;
  float gae_r = 0.f, gae_c = 0.f;
  float vn_r = __fadd_rn(__fmul_rn(a.value_preds[a.T * N + n], a.sd_r), a.mu_r);      // dn(v_T)
  float vn_c = __fadd_rn(__fmul_rn(a.cost_preds[a.T * N + n], a.sd_c), a.mu_c);
  float m_next = a.masks[a.T * N + n];
  for (int64_t t0 = a.T - 1; t0 >= 0; t0 -= UNROLL) {
    float r[UNROLL], c[UNROLL], vr[UNROLL], vc[UNROLL], mk[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t t = t0 - u;
      const bool ok = t >= 0;
      const int64_t o = (ok ? t : 0) * N + n;
      r[u] = a.rewards[o]; c[u] = a.costs[o]; vr[u] = a.value_preds[o]; vc[u] = a.cost_preds[o]; mk[u] = a.masks[o];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t t = t0 - u;
      if (t < 0) break;
      const float dr = __fadd_rn(__fmul_rn(vr[u], a.sd_r), a.mu_r);
      const float dc = __fadd_rn(__fmul_rn(vc[u], a.sd_c), a.mu_c);
      // delta = rewards[t] + gamma * dn(v[t+1]) * masks[t+1] - dn(v[t])          (buffer.py:375)
      const float del_r = __fsub_rn(__fadd_rn(r[u], __fmul_rn(__fmul_rn(a.gamma, vn_r), m_next)), dr);
      const float del_c = __fsub_rn(__fadd_rn(c[u], __fmul_rn(__fmul_rn(a.gamma, vn_c), m_next)), dc);
      // gae = delta + gamma * gae_lambda * masks[t+1] * gae                        (buffer.py:376)
      gae_r = __fadd_rn(del_r, __fmul_rn(__fmul_rn(a.gl, m_next), gae_r));
      gae_c = __fadd_rn(del_c, __fmul_rn(__fmul_rn(a.gl, m_next), gae_c));
      a.returns[t * N + n] = __fadd_rn(gae_r, dr);                                  // buffer.py:377
      a.cost_returns[t * N + n] = __fadd_rn(gae_c, dc);
      vn_r = dr; vn_c = dc; m_next = mk[u];
    }
  }
}

}  // namespace

extern "C" int spo_ma_gae(const float* rewards, const float* costs, const float* value_preds, const float* cost_preds,
                          const float* masks, float* returns, float* cost_returns, int64_t T, int64_t num_threads,
                          double gamma, double gae_lambda, float denorm_std_r, float denorm_mean_r,
                          float denorm_std_c, float denorm_mean_c, void* stream) {
  SPO_REQUIRE(rewards && costs && value_preds && cost_preds && masks && returns && cost_returns, "ma_gae: null pointer");
  SPO_REQUIRE(T >= 0 && num_threads >= 0, "ma_gae: negative size");
  if (T == 0 || num_threads == 0) return 0;
  MaArgs a{rewards, costs, value_preds, cost_preds, masks, returns, cost_returns, T, num_threads,
           (float)gamma, (float)(gamma * gae_lambda), denorm_std_r, denorm_mean_r, denorm_std_c, denorm_mean_c};
  const unsigned blocks = (unsigned)((num_threads + 255) / 256);
  hipLaunchKernelGGL((ma_gae_kernel<8>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  SPO_LAUNCH_CHECK("spo_ma_gae");
  return 0;
}

// ---------------------------------------------------------------- insert of one environment step into the stacked buffers
// Runner.insert (reference mappolag.py:449-492): per step the runner copies the environment's [N, agents, ...] outputs into
// every agent's time-major buffer and forms masks / active masks from the done flags -- a dozen tiny launches per step whose
// cost is the host's launch time.  One kernel does the six fields: dst(field)[agent][n][:] = src(field)[n][agent][:], with
//     masks[a][n]        = all_a' done[n][a'] ? 0 : 1                                     (mappolag.py:458-463)
//     active_masks[a][n] = done[n][a] && !all-done ? 0 : 1                                (mappolag.py:465-467)
// dst pointers address the (agent 0, step slot) row; agent_stride = floats between two agents' buffers of that field.
namespace {
struct InsArgs {
  const float* obs; const float* share_obs; const float* rewards; const float* costs; const unsigned char* dones;
  float* obs_d; float* share_d; float* rew_d; float* cost_d; float* mask_d; float* act_d;
  int64_t obs_s, share_s, rew_s, cost_s, mask_s, act_s;      // agent strides (floats)
  int64_t N; int A, Do, Ds;
};
__global__ __launch_bounds__(256) void ma_insert_kernel(InsArgs a) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per (n, agent) row
  const int lane = threadIdx.x & 63;
  if (row >= a.N * a.A) return;
  const int64_t n = row / a.A;
  const int ag = (int)(row - n * a.A);
  for (int c = lane; c < a.Do; c += 64) a.obs_d[ag * a.obs_s + n * a.Do + c] = a.obs[row * a.Do + c];
  for (int c = lane; c < a.Ds; c += 64) a.share_d[ag * a.share_s + n * a.Ds + c] = a.share_obs[row * a.Ds + c];
  if (lane == 0) {
    a.rew_d[ag * a.rew_s + n] = a.rewards[row];
    if (a.costs) a.cost_d[ag * a.cost_s + n] = a.costs[row];
    bool all_done = true;
    for (int k = 0; k < a.A; ++k) all_done = all_done && a.dones[n * a.A + k] != 0;
    const bool mine = a.dones[row] != 0;
    a.mask_d[ag * a.mask_s + n] = all_done ? 0.f : 1.f;
    a.act_d[ag * a.act_s + n] = (mine && !all_done) ? 0.f : 1.f;
  }
}
}  // namespace

extern "C" int spo_ma_insert_step(const float* obs, const float* share_obs, const float* rewards, const float* costs,
                                  const unsigned char* dones, float* obs_dst, int64_t obs_agent_stride, float* share_obs_dst,
                                  int64_t share_obs_agent_stride, float* rewards_dst, int64_t rewards_agent_stride, float* costs_dst,
                                  int64_t costs_agent_stride, float* masks_dst, int64_t masks_agent_stride, float* active_masks_dst,
                                  int64_t active_masks_agent_stride, int64_t num_threads, int32_t num_agents, int32_t obs_dim,
                                  int32_t share_obs_dim, void* stream) {
  SPO_REQUIRE(obs && share_obs && rewards && dones && obs_dst && share_obs_dst && rewards_dst && masks_dst && active_masks_dst,
              "ma_insert_step: null pointer");
  SPO_REQUIRE((costs == nullptr) == (costs_dst == nullptr), "ma_insert_step: costs and costs_dst go together");
  SPO_REQUIRE(num_threads > 0 && num_agents > 0 && num_agents <= 64 && obs_dim > 0 && share_obs_dim > 0, "ma_insert_step: bad sizes");
  InsArgs a{obs, share_obs, rewards, costs, dones, obs_dst, share_obs_dst, rewards_dst, costs_dst, masks_dst, active_masks_dst,
            obs_agent_stride, share_obs_agent_stride, rewards_agent_stride, costs_agent_stride, masks_agent_stride,
            active_masks_agent_stride, num_threads, num_agents, obs_dim, share_obs_dim};
  const int64_t rows = num_threads * num_agents;
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffffLL) return spo::fail(-1, "ma_insert_step: %lld rows exceed the launch grid", (long long)rows);
  hipLaunchKernelGGL(ma_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  SPO_LAUNCH_CHECK("spo_ma_insert_step");
  return 0;
}
