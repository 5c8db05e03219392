/*
 * safepo_hip.h -- C ABI of libsafepo_hip.so (gfx950 / MI355X).
 *
 * The reference (PKU-Alignment/Safe-Policy-Optimization) has NO native layer: the path
 * this library replaces is Python.  Each entry point below names the
 * reference Python function (file:line relative to /root/reference) whose arithmetic it
 * replaces; the Python host mirror (safe-policy-optimization_amd/safepo) binds them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - no allocation, no ownership transfer, no implicit synchronisation; work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: 0 ok, <0 argument error (see spo_last_error()), >0 a hipError_t;
 *   - re-entrant per stream; one host thread per GPU.
 *
 * Dense on-policy buffer layout (replaces the list of per-env dicts,
 * safepo/common/buffer.py:53-73): scalars are [num_envs, T] env-major (flat row = env*T + t,
 * the order VectorizedOnPolicyBuffer.get() concatenates in, buffer.py:149-153), obs
 * [num_envs, T, obs_dim], act [num_envs, T, act_dim]; path ends are a u8 mask seg_end[N,T]
 * with the bootstrap values of finish_path() in boot_r/boot_c[N,T].
 *
 * Flat parameter vector `theta` (fp32): policy.parameters() order of ActorVCritic
 * (safepo/common/model.py:131-135): reward_critic {W1[H,D],b1,W2[H,H],b2,W3[1,H],b3},
 * cost_critic {same}, actor {log_std[A], W1[H,D],b1,W2[H,H],b2,W3[A,H],b3}; nn.Linear
 * row-major [out,in].  H = 64 (default_cfg hidden_sizes, ppo_lag.py:45-52).
 */
#ifndef SAFEPO_HIP_H
#define SAFEPO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPO_ABI_VERSION 2        /* 2 (round 6): pow4_dev of spo_wide_clip_adam_dev(_log) is double[6], a negative learning-rate entry
                                 * means "the cfg's"; spo_update_rs_supported; the row-split form behind spo_ppo_lag_update_iter */
#define SPO_HIDDEN 64          /* hidden width the MLP kernels are specialised for          */
#define SPO_MAX_ACT 16         /* act_dim <= 16 (one MFMA output tile; LDS-resident kernels) */
#define SPO_MAX_OBS 128        /* obs_dim <= 128 (LDS-resident kernels; CPO full-batch kernels: 64) */
#define SPO_WIDE_MAX_ACT 64    /* act_dim limit of the wide-network path (any obs_dim / hidden_sizes): its actor-side loss kernels
                                * give a batch row pow2ceil(act_dim) lanes of ONE 64-lane wavefront and sum over the action
                                * dimension with segmented shuffles; a wider action vector would need a row spread over several
                                * wavefronts.  The widest action space of the reference's task sets is 17 (Humanoid). */
#define SPO_GAE_PARTIAL_STRIDE 16 /* doubles per block written by spo_gae_fused: 4 waves x 4 */

int spo_abi_version(void);
const char* spo_last_error(void);

/* ---- a-5: VectorizedOnPolicyBuffer.finish_path -> calculate_adv_and_value_targets ->
 *      discount_cumsum (safepo/common/buffer.py:97-140,191-201,167-188), for ALL paths of
 *      ALL envs and for reward and cost in one launch.
 * delta in fp32 (three separately rounded ops, gamma rounded to fp32), segmented backward
 * scan in fp64 with discount gamma*lam formed in double, results rounded to fp32.
 * Steps after the last seg_end of a row (unfinished path) get adv = target = 0.
 * boot_r == boot_c == NULL selects the FOLDED form: `reward` / `cost` are then the arrays written by
 * spo_boundary_step_fold (gamma * bootstrap already added at path ends); same results, no bootstrap loads.
 * partials: [spo_gae_num_blocks(N,T)][SPO_GAE_PARTIAL_STRIDE] doubles: one row {sum adv_r, sum adv_r^2, sum adv_c,
 * count} per wave of each workgroup (no block-level combine inside the scan; spo_adv_reduce adds the rows). */
int spo_gae_num_blocks(int64_t num_envs, int64_t T);
int spo_gae_fused(const float* reward, const float* cost, const float* value_r, const float* value_c,
                  const uint8_t* seg_end, const float* boot_r, const float* boot_c,
                  float* adv_r, float* adv_c, float* target_r, float* target_c,
                  double* partials, int64_t num_envs, int64_t T,
                  double gamma, double lam, double lam_c, void* stream);

/* Debug/bench knob: 0 = automatic, 1 = eager bootstrap loads (latency-optimised, cache-resident
 * buffers), 2 = predicated bootstrap loads (fewest bytes, HBM-streaming buffers). */
int spo_debug_gae_variant(int v);
/* Measurement aid (bench.py `roofline`): spo_gae_fused `reps` times on `stream`, every dispatch carrying its own start /
 * stop events, so durations_us_host[i] (host array of `reps` floats) is the execution time of dispatch i as the dispatch
 * packet's own timestamps record it -- the per-dispatch figure rocprofv3 --kernel-trace reports.  Synchronises `stream`. */
int spo_gae_fused_timed(const float* reward, const float* cost, const float* value_r, const float* value_c,
                        const uint8_t* seg_end, const float* boot_r, const float* boot_c,
                        float* adv_r, float* adv_c, float* target_r, float* target_c,
                        double* partials, int64_t num_envs, int64_t T,
                        double gamma, double lam, double lam_c, int reps, float* durations_us_host, void* stream);

/* ---- a-6: statistics of VectorizedOnPolicyBuffer.get() (buffer.py:154-160).
 * spo_adv_reduce: partials -> sums[4] = {sum adv_r, sum adv_r^2, sum adv_c, count} (fixed order,
 * deterministic).  Multi-GPU: all-reduce(sum) sums[] between the two calls.
 * spo_adv_apply: adv_r <- (adv_r-mean)/(std_unbiased+1e-8) if standardize_r; adv_c <- adv_c-mean_c
 * if standardize_c; a-8 (ppo_lag.py:280-281): adv_mix = (adv_r - lambda*adv_c)/(lambda+1) when
 * adv_mix != NULL.  stats_out (optional, 3 floats): mean_r, std_r, mean_c. */
int spo_adv_reduce(const double* partials, int num_blocks, double* sums, void* stream);
int spo_adv_apply(float* adv_r, float* adv_c, float* adv_mix, const double* sums, int64_t count,
                  double lagrangian_multiplier, int standardize_r, int standardize_c,
                  float* stats_out, void* stream);

/* ---- a-1/a-3: ActorVCritic.step (safepo/common/model.py:149-170) + buffer.store
 * (buffer.py:84-95) for step t of every env.  act = mean + exp(log_std)*eps (rsample with the
 * noise supplied; eps == NULL -> deterministic, act = mean).  Outputs act/logp/v_r/v_c [N,...];
 * when buf_* are non-NULL the same values (and obs) are also written to slot t of the dense
 * buffer. */
int spo_policy_step(const float* theta, const float* obs, const float* eps,
                    float* act, float* logp, float* v_r, float* v_c,
                    float* buf_obs, float* buf_act, float* buf_logp, float* buf_v_r, float* buf_v_c,
                    int64_t num_envs, int64_t T, int64_t t, int obs_dim, int act_dim, void* stream);

/* ---- a-2: observation normalisation of SafeNormalizeObservation (safepo/common/wrappers.py:42-49 ->
 * gymnasium NormalizeObservation / RunningMeanStd, third-party, restated from its documented algorithm:
 * PARITY UNPINNED).  rms_state = double[2*obs_dim+1] = mean[D], var[D], count (init 0, 1, 1e-4).
 * update != 0: merge this batch (mean, biased var over num_envs) into the running statistics first;
 * then obs <- (obs - mean) / sqrt(var + 1e-8), in place. */
int spo_obs_normalize(float* obs, double* rms_state, int64_t num_envs, int obs_dim, int update, void* stream);

/* a-1 + a-2 fused (north_star: "fused obs-normalise + MLP forward"): spo_policy_step on RAW observations with the
 * SafeNormalizeObservation wrapper's arithmetic (wrappers.py:42-49 -> gymnasium NormalizeObservation / RunningMeanStd) folded
 * in.  update != 0: the batch is first merged into rms_state (mean[D], var[D], count; fp64; parallel-variance merge).  Then
 * every row is read ONCE, normalised in registers as (float)(((double)x - mean) / sqrt(var + 1e-8)), fed to the three
 * networks, stored to the buffer slot and written back to obs_inout (the wrapper hands the normalised observation on). */
int spo_policy_step_norm(const float* theta, float* obs_inout, double* rms_state, int update, const float* eps,
                         float* act, float* logp, float* v_r, float* v_c,
                         float* buf_obs, float* buf_act, float* buf_logp, float* buf_v_r, float* buf_v_c,
                         int64_t num_envs, int64_t T, int64_t t, int obs_dim, int act_dim, void* stream);

/* critics only (bootstrap values of ppo_lag.py:201-215): v_r, v_c for `rows` observations. */
int spo_values(const float* theta, const float* obs, float* v_r, float* v_c,
               int64_t rows, int obs_dim, int act_dim, void* stream);

/* ---- a-4: path-boundary logic of the collect loop (ppo_lag.py:198-234) for step t:
 * seg_end = epoch_end | terminated | truncated; boot = 0 if terminated, else value(next obs) at
 * epoch end, overridden by value(final obs) if truncated.  Also stores reward/cost of the step,
 * accumulates ep_ret/ep_cost/ep_len (ppo_lag.py:168-170) and appends one record
 * {t*num_envs + env, ep_ret, ep_cost, ep_len} (4 doubles) per finished episode to `events`, in
 * env order (the order of the reference's Python loop); events_count is a device int advanced
 * by the single block that runs this kernel. */
int spo_boundary_step(const float* reward, const float* cost, const float* terminated, const float* truncated,
                      const float* v_next_r, const float* v_next_c, const float* v_final_r, const float* v_final_c,
                      float* buf_reward, float* buf_cost, uint8_t* seg_end, float* boot_r, float* boot_c,
                      double* ep_ret, double* ep_cost, double* ep_len, double* events, int* events_count,
                      int events_capacity, int64_t num_envs, int64_t T, int64_t t, int epoch_end, void* stream);

/* The same step, additionally writing the two arrays spo_gae_fused reads in its FOLDED form (boot_r == boot_c == NULL
 * there): fold_reward[env,t] = fl(reward + fl((float)gamma * boot_r)) at a path end and the plain reward elsewhere (same
 * for cost) -- the first two of the three fp32 operations of delta_t = r_t + gamma*v_{t+1} - v_t (buffer.py:198) applied
 * to the value finish_path would append (buffer.py:118-125).  The scan then needs no bootstrap loads: exactly the 33
 * algorithmic bytes per (env, step), no second memory round trip, results bit-identical to the unfolded form. */
int spo_boundary_step_fold(const float* reward, const float* cost, const float* terminated, const float* truncated,
                           const float* v_next_r, const float* v_next_c, const float* v_final_r, const float* v_final_c,
                           float* buf_reward, float* buf_cost, uint8_t* seg_end, float* boot_r, float* boot_c,
                           double* ep_ret, double* ep_cost, double* ep_len, double* events, int* events_count,
                           int events_capacity, int64_t num_envs, int64_t T, int64_t t, int epoch_end,
                           float* fold_reward, float* fold_cost, double gamma, void* stream);

/* The same step on num_envs / 256 workgroups instead of one (the one-block kernel is a 24 us latency chain at 4 096 envs).  The
 * running event count lives in events_prefix (int[T + 1], events_prefix[0] = 0 at the start of an epoch): the kernel of step t
 * reads events_prefix[t], appends this step's finished episodes in env order and writes events_prefix[t + 1]; after the epoch
 * events_prefix[T] records are valid.  fold_reward / fold_cost may both be NULL.  Everything else as spo_boundary_step_fold;
 * results (buffers, masks, event log) are identical. */
int spo_boundary_step_fold_mb(const float* reward, const float* cost, const float* terminated, const float* truncated,
                              const float* v_next_r, const float* v_next_c, const float* v_final_r, const float* v_final_c,
                              float* buf_reward, float* buf_cost, uint8_t* seg_end, float* boot_r, float* boot_c,
                              double* ep_ret, double* ep_cost, double* ep_len, double* events, int* events_prefix,
                              int events_capacity, int64_t num_envs, int64_t T, int64_t t, int epoch_end,
                              float* fold_reward, float* fold_cost, double gamma, void* stream);

/* spo_values(theta, final_obs) -> (v_final_r, v_final_c) and spo_boundary_step_fold_mb with those values, in ONE launch (the
 * post-step half of a collect step, ppo_lag.py:198-234): the bootstrap values of truncated envs go from the critics' accumulators
 * to the boundary logic through LDS.  v_final_r / v_final_c are still written.  Same results as the two calls (which it makes
 * itself outside the side-by-side kernel's envelope: obs_dim > 64 or more than 8 192 envs). */
int spo_values_boundary_step_fold(const float* theta, const float* final_obs, float* v_final_r, float* v_final_c,
                                  int obs_dim, int act_dim, const float* reward, const float* cost,
                                  const float* terminated, const float* truncated, const float* v_next_r,
                                  const float* v_next_c, float* buf_reward, float* buf_cost, uint8_t* seg_end,
                                  float* boot_r, float* boot_c, double* ep_ret, double* ep_cost, double* ep_len,
                                  double* events, int* events_prefix, int events_capacity, int64_t num_envs,
                                  int64_t T, int64_t t, int epoch_end, float* fold_reward, float* fold_cost,
                                  double gamma, void* stream);

/* ---- a-9/a-10: one learning iteration of the PPO-Lagrangian update (ppo_lag.py:297-336):
 * for each consecutive chunk of `batch` indices of perm[M] (last partial chunk kept): gather,
 * 3x MLP fwd, loss_r/loss_c (MSE + 0.001*L2 if use_critic_norm), clipped surrogate, backward,
 * joint clip_grad_norm_(max_grad_norm, eps 1e-6), 3x Adam (betas .9/.999, eps 1e-8).
 * Persistent kernel, 3 workgroups (one per network) for the whole iteration; weights in LDS,
 * Adam moments in registers for the launch (loaded from / stored to adam_m, adam_v).
 * adam_step_host: number of optimiser steps already taken (bias correction continues from it).
 * losses_out: [num_minibatches][3] = loss_r, loss_c, loss_pi per minibatch (ppo_lag.py:330-336).
 * sync_ws: >= 72 bytes of device scratch, zero before first use.  Bytes 0..63 are exchange slots, zeroed by every
 * call; the int at byte 64 is a STICKY error word the kernels only ever set (1 = inter-workgroup exchange timed out,
 * 2 = a peer rank never answered the in-kernel gradient exchange) -- the caller reads and clears it.                */
typedef struct {
  int obs_dim, act_dim, batch;
  int use_critic_norm;          /* config.get("use_critic_norm", True)                        */
  int use_value_coefficient;    /* total = loss_pi + 2*loss_r + loss_c                        */
  float clip;                   /* 0.2                                                         */
  float max_grad_norm;          /* 40.0                                                        */
  float lr_actor, lr_critic;    /* 3e-4 * LinearLR factor, 3e-4                                */
  float beta1, beta2, adam_eps; /* 0.9, 0.999, 1e-8                                            */
  float l2_coef;                /* 0.001                                                       */
} spo_ppo_cfg;

int spo_ppo_lag_update_iter(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host,
                            const float* obs, const float* act, const float* logp_old,
                            const float* target_r, const float* target_c, const float* adv,
                            const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host,
                            float* losses_out, void* sync_ws, void* stream);

/* The main + helper form of the persistent update keeps one ~790 KB scratch block (clip backups, norm granules) per
 * (device, stream handle) pair, allocated at that pair's first launch (never under stream capture: launch once outside the
 * capture first) and kept; at most 256 pairs per process.  A process that keeps creating and destroying streams releases a
 * stream's block before destroying it: returns the number of blocks freed (>= 0) for (current device, stream), or for every
 * stream of the current device when all != 0.  hipFree synchronises the device -- not for the hot path. */
int spo_update_scratch_release(void* stream_or_null, int all);

/* Round 6: the ROW-SPLIT form of the persistent update (csrc/update_rs.hip) -- the rows of every minibatch divided over R
 * co-XCD workgroups per network (R = 2 up to 64 rows, 4 up to 128), partial weight gradients all-reduced in one hand-off through
 * the XCD's L2, replicated Adam.  spo_ppo_lag_update_iter (n_nets 3; ppo_lag.py:297-336) and spo_critic_fit_iter (n_nets 2;
 * cpo.py:534-571) run on it where this returns 1 (obs_dim <= 64, act_dim <= 16, batch <= 64 / 128) unless SPO_UPDATE_FORM=2
 * (the main + helper form) or 0 (the four-wave form) is set in the environment.  Same arguments, same results to rounding. */
int spo_update_rs_supported(int obs_dim, int act_dim, int batch, int n_nets);

/* Measurement aid: counters of the main + helper update kernel summed over the launches of this process since the last reset
 * (out4_host, host array): {minibatch steps run, steps whose speculative update turned out clipped and was redone, steps
 * clipped under the conservative protocol, steps run under the conservative protocol}.  Synchronises the device. */
int spo_debug_update_counters(unsigned long long* out4_host, int reset);

/* Debug self-test of the cross-lane helpers (DPP row sums, gfx950 permlane swaps): in[64] -> out[192]. */
int spo_debug_crosslane_selftest(const float* in64, float* out192, void* stream);

/* Debug aid (not a reference function): pass a device buffer of 30 u64 to make the next
 * spo_ppo_lag_update_iter launches accumulate shader cycles per phase of the step; NULL disables. */
int spo_debug_set_update_profile(void* dev_u64_30);

/* Split form of the same step for data-parallel training: gradient of ONE
 * minibatch into flat_grad[P] (+ losses[3]); the caller all-reduces flat_grad and then
 * applies clip + Adam.  `grad_scale` multiplies the gradient (1/world_size for averaging). */
int spo_ppo_lag_grad(const float* theta, const float* obs, const float* act, const float* logp_old,
                     const float* target_r, const float* target_c, const float* adv,
                     const int32_t* idx, int n_idx, int64_t mean_count, const spo_ppo_cfg* cfg_host,
                     float* flat_grad, float* losses3, void* stream);
int spo_clip_adam(float* theta, float* adam_m, float* adam_v, const float* flat_grad, int64_t adam_step_host,
                  float grad_scale, const spo_ppo_cfg* cfg_host, void* stream);

/* one host call per data-parallel step: spo_clip_adam(step k) followed by spo_ppo_lag_grad(step k+1)
 * into the same flat_grad buffer (next_idx == NULL: only the optimiser step). */
int spo_clip_adam_then_grad(float* theta, float* adam_m, float* adam_v, float* flat_grad, int64_t adam_step_host,
                            float grad_scale, const float* obs, const float* act, const float* logp_old,
                            const float* target_r, const float* target_c, const float* adv,
                            const int32_t* next_idx, int next_n_idx, const spo_ppo_cfg* cfg_host,
                            float* next_losses3, void* stream);

/* ---- a-11: full-batch actor forward + KL early-stop statistic (ppo_lag.py:277,338-345).
 * spo_actor_mean: mean_out[M,act_dim] = actor.mean(obs).  spo_actor_kl: partial sums of
 * KL(N(mu_old, exp(log_std_old)) || N(mu_new, exp(log_std_new))).sum(-1) over rows into
 * kl_sum (1 double, fixed-order reduction); caller divides by the (global) row count. */
int spo_actor_mean(const float* theta, const float* obs, float* mean_out, int64_t rows,
                   int obs_dim, int act_dim, void* stream);
int spo_actor_kl(const float* theta, const float* obs, const float* mean_old, const float* log_std_old,
                 double* kl_partials, int kl_partials_capacity, double* kl_sum, int64_t rows,
                 int obs_dim, int act_dim, void* stream);

/* ---- CPO (safepo/single_agent/cpo.py).  Actor-only flat vectors use the actor's own layout
 * [log_std, W1, b1, W2, b2, W3, b3] = actor.named_parameters() order (cpo.py:70-78), length
 * Pa = spo_param_count - 2*critic.  Workspaces: partial_ws float[spo_cpo_num_partials(rows)*Pa],
 * loss_ws double[spo_cpo_num_partials(rows)].
 * a-16 spo_cpo_surrogate_grad: grad_out = d/dtheta [ sign * mean(ratio * adv) ], ratio = exp(logp - logp_old)
 *      (cpo.py:356-365 with sign=-1/adv_r, :372-381 with sign=+1/adv_c); loss_sum_out[0] = sum(ratio*adv).
 * a-14 spo_cpo_fvp: out = J^T diag(1/sigma^2) J vec / (rows*act_dim) on the mean-network entries, 0 on the
 *      log_std entries; the caller adds (2/act_dim)*vec[log_std] and the 0.1*vec damping (cpo.py:132-157).
 * a-17 spo_cpo_linesearch_eval: sums3_out = {sum ratio*adv_r, sum ratio*adv_c, sum_{rows,dims} KL(old||new)}
 *      for the parameters currently in theta (cpo.py:473-491).
 * a-18 spo_critic_fit_iter: one pass of the critic fit (cpo.py:541-571) on the persistent kernel with two
 *      networks; *stale_sq_io carries ||actor.grad||^2 (the stale cost gradient that clip_grad_norm_ over ALL
 *      policy parameters still sees, and rescales in place whenever it clips). */
int spo_cpo_num_partials(int64_t rows);
int spo_cpo_surrogate_grad(const float* theta, const float* obs, const float* act, const float* logp_old,
                           const float* adv, float sign, int64_t rows, int obs_dim, int act_dim,
                           float* partial_ws, double* loss_ws, float* grad_out, double* loss_sum_out, void* stream);
int spo_cpo_fvp(const float* theta, const float* obs, const float* vec, int64_t rows, int obs_dim, int act_dim,
                float* partial_ws, double* loss_ws, float* out, void* stream);
int spo_cpo_linesearch_eval(const float* theta, const float* obs, const float* act, const float* logp_old,
                            const float* adv_r, const float* adv_c, const float* mean_old, const float* log_std_old,
                            int64_t rows, int obs_dim, int act_dim, double* partial_ws, int partial_capacity,
                            double* sums3_out, void* stream);
int spo_critic_fit_iter(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host,
                        const float* obs, const float* target_r, const float* target_c, const int32_t* perm,
                        int64_t M, const spo_ppo_cfg* cfg_host, float* stale_sq_io, float* losses_out,
                        void* sync_ws, void* stream);

/* ---- f2: FOCOPS (safepo/single_agent/focops.py:300-350) and CUP (safepo/single_agent/cup.py:300-400) on the
 * persistent update kernel.  spo_update_iter_ex is spo_ppo_lag_update_iter with
 *  - separate Adam step counts for the critics' and the actor's optimisers (CUP's second stage steps the actor alone),
 *  - actor_only != 0: only the actor network is trained and clip_grad_norm_ spans the actor's parameters (cup.py:385),
 *  - actor_loss = SPO_ACTOR_LOSS_CLIP: the PPO clipped surrogate (old_mean/old_std unused, may be NULL), or
 *    SPO_ACTOR_LOSS_KL_PENALTY: loss = mean_i(ind_i*KL_i) - pg_coef*mean_i(ind_i)*mean_j(ratio_j*adv_j) with
 *    KL_i = KL(N(mu_i,sigma) || N(old_mean_i, old_std)).sum(-1), ind_i = [KL_i <= kl_bound]
 *    (focops.py:326-337: kl_bound = target_kl, pg_coef = 1/FOCOPS_LAM; cup.py:372-383: kl_bound = +inf,
 *    pg_coef = -lagrangian_multiplier*(1-gamma*CUP_LAMBDA)/(1-gamma)).  old_mean float[M*act_dim], old_std
 *    float[act_dim].  Needs cfg.batch <= 64.
 * losses_out float[ceil(M/batch)*3]; with actor_only only column 2 is written. */
#define SPO_ACTOR_LOSS_CLIP 0
#define SPO_ACTOR_LOSS_KL_PENALTY 1
int spo_update_iter_ex(float* theta, float* adam_m, float* adam_v, int64_t adam_step_critics_host,
                       int64_t adam_step_actor_host, const float* obs, const float* act, const float* logp_old,
                       const float* target_r, const float* target_c, const float* adv, const int32_t* perm, int64_t M,
                       const spo_ppo_cfg* cfg_host, int actor_loss, const float* old_mean, const float* old_std,
                       float kl_bound, float pg_coef, int actor_only, float* losses_out, void* sync_ws, void* stream);

/* ---- (e) multi-GPU, in-kernel form (SURVEY.md 8(e) "direct all-reduce over the xGMI mesh, fused as the epilogue of
 * the backward kernel").  Each rank owns one exchange region (uncached device memory, spo_p2p_alloc) and maps its
 * peers' regions through the 64-byte IPC handle (spo_p2p_open; exchange the handles with any host collective).
 * regions[r] = rank r's region as seen from this process (own pointer at index rank), world <= 8.
 * spo_ppo_lag_update_iter_dp is spo_ppo_lag_update_iter for one rank of a data-parallel job: every rank launches it on
 * its own shard (same M, same batch, same step counts) and the per-step mean of the ranks' minibatch gradients is formed
 * inside the step, so all replicas apply bit-identical updates.  Every float travels as an 8-byte {tag, value} word
 * written with one system-scope write-through store and polled by the lane that needs it (no flags, no barriers):
 * power-of-two worlds use recursive doubling (round k: swap the running sum with rank me ^ 2^k; log2(world) hand-offs);
 * other worlds reduce-scatter then all-gather (two hand-offs, 2 x 88 KB per network per step).  Environment:
 * SPO_P2P_ALGO=twophase forces the second form everywhere.  step0 = optimiser steps already taken through these regions (monotonic tag base, identical
 * on every rank; advance it by the number of minibatches).  A peer that never answers costs ONE bounded wait and is
 * reported through sync_ws (int at byte 64 = 2), never a hang.
 * spo_p2p_selftest runs `iters` exchange rounds of known patterns on the same grid and protocol:
 * result2_dev[0] = wrong values, result2_dev[1] = 2 after a timeout; it consumes `iters` tags. */
int64_t spo_p2p_region_bytes(void);
int spo_debug_xr_profile(unsigned long long* out8_host, int reset);

/* spo_ppo_lag_update_iter for wide observations / action vectors (round 5; csrc/update_ks.hip): obs_dim <= 512, act_dim <= 32,
 * batch <= 64, hidden [64, 64], the clipped-surrogate loss, one GPU.  The first layer is split over the input features, 64 per
 * workgroup: 3 x ceil(obs_dim / 64) persistent workgroups exchange the partial pre-activations inside the step; same
 * arguments, results and per-step loss log as spo_ppo_lag_update_iter (reference: safepo/single_agent/ppo_lag.py:297-336 with
 * an ActorVCritic(376, 17), model.py:131; single_agent/benchmark.py:5-22).  spo_ks_supported: 1 when the shape fits. */
int spo_ks_supported(int obs_dim, int act_dim, int batch);
int spo_ppo_lag_update_iter_ks(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs,
                               const float* act, const float* logp_old, const float* target_r, const float* target_c,
                               const float* adv, const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host,
                               float* losses_out, void* sync_ws, void* stream);
/* spo_update_iter_ex on the feature-split kernel: the KL-penalty actor loss of FOCOPS (focops.py:326-337) and CUP's second stage
 * (cup.py:372-383), separate optimiser clocks, actor-only mode -- same arguments and semantics as spo_update_iter_ex, shapes as
 * spo_ks_supported, one GPU. */
int spo_update_iter_ex_ks(float* theta, float* adam_m, float* adam_v, int64_t adam_step_critics_host,
                          int64_t adam_step_actor_host, const float* obs, const float* act, const float* logp_old,
                          const float* target_r, const float* target_c, const float* adv, const int32_t* perm, int64_t M,
                          const spo_ppo_cfg* cfg_host, int actor_loss, const float* old_mean, const float* old_std,
                          float kl_bound, float pg_coef, int actor_only, float* losses_out, void* sync_ws, void* stream);
/* One minibatch's gradient on the feature-split kernel (round 6: the data-parallel step at these dims): the RAW
 * data gradient (no L2 term, no value coefficient) of the three networks on rows idx[0 .. n), n <= 64, into flat_grad (theta's
 * layout, log_std included) and the three data losses into losses3 -- ppo_lag.py:306-324 up to loss.backward(), the part
 * spo_mlp_forward / spo_wide_ppo_loss / spo_mlp_backward take 20 launches for.  theta is not written.  The caller averages
 * flat_grad over the ranks (the global minibatch is the concatenation of the ranks' rows; the reference itself is one process) and
 * applies spo_wide_clip_adam. */
int spo_ppo_lag_grad_ks(const float* theta, const float* obs, const float* act, const float* logp_old, const float* target_r,
                        const float* target_c, const float* adv, const int32_t* idx, int n, const spo_ppo_cfg* cfg_host,
                        float* flat_grad, float* losses3, void* sync_ws, void* stream);
/* spo_critic_fit_iter on the same feature-split kernel (two networks): the critic fit of the second-order scripts
 * (safepo/single_agent/cpo.py:541-571) for obs_dim <= 512, hidden [64, 64], batch <= 128 (a minibatch is taken as two 64-column
 * chunks whose gradients accumulate before the optimiser step), one GPU.  Same arguments and results as spo_critic_fit_iter;
 * cfg_host->act_dim only locates the critics in the flat parameter vector. */
int spo_critic_fit_ks_supported(int obs_dim, int batch);
int spo_critic_fit_iter_ks(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs,
                           const float* target_r, const float* target_c, const int32_t* perm, int64_t M,
                           const spo_ppo_cfg* cfg_host, float* stale_sq_io, float* losses_out, void* sync_ws, void* stream);

/* Form of the in-kernel gradient exchange (which of the built all-reduce forms runs inside the persistent update
 * kernel).  spo_p2p_select_form pins one for the process (-1: back to the default policy: environment overrides, else the row-split
 * form at 2 / 4 / 8 ranks -- shapes outside the row-split kernel run the four-wave kernel -- recursive doubling at 2 / 4 ranks and
 * the two-phase form elsewhere); a form that does not exist at a world size falls back to the policy.
 * safepo.parallel.PeerExchange.autotune times every valid form at start-up on the actual topology and pins the fastest. */
#define SPO_XR_FORM_TWOPHASE 0          /* packed reduce-scatter + all-gather, four-wave kernel: two hand-offs at any world size   */
#define SPO_XR_FORM_DOUBLING 1          /* packed recursive doubling, four-wave kernel: log2(world) hand-offs (power-of-two worlds) */
#define SPO_XR_FORM_HELPER_A2A 2        /* one-shot all-to-all with flags on the helper waves (worlds 2 / 4 / 8): one hand-off      */
#define SPO_XR_FORM_HELPER_DOUBLING 3   /* packed recursive doubling on the helper waves, layer by layer (worlds 2 / 4 / 8)          */
#define SPO_XR_FORM_ROW_SPLIT 4         /* round 6: the row-split kernel (spo_update_rs_supported), tagged 16-byte words behind the row
                                         * groups' L2 hand-off.  2 ranks: the gradient goes to the peer at once (ONE cross-rank hand-off);
                                         * 4 / 8 ranks: reduce-scatter + all-gather -- every word to its owner rank, which adds the
                                         * world's contributions in rank order (all polled in one batch) and sends the finished word to
                                         * everyone: two hand-offs on world - 1 links each (SPO_P2P_ALGO=rowsplit)                     */
int spo_p2p_select_form(int form);
int spo_p2p_form_valid(int form, int world);
int spo_p2p_current_form(int world);

/* self-test kernel phase cycles (debug) */
int spo_p2p_alloc(void** region_out, void* ipc_handle64_out);
int spo_p2p_open(const void* ipc_handle64, void** region_out);
int spo_p2p_close(void* peer_region);
int spo_p2p_free(void* own_region);
int spo_p2p_selftest(int rank, int world, void* const* regions, uint32_t step0, int iters, int32_t* result2_dev,
                     void* stream);
int spo_critic_fit_iter_dp(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs,
                           const float* target_r, const float* target_c, const int32_t* perm, int64_t M,
                           const spo_ppo_cfg* cfg_host, float* stale_sq_io, float* losses_out, void* sync_ws, int rank,
                           int world, void* const* regions, uint32_t step0, void* stream);
/* One-grid split critic fit (single GPU; cpo.py:534-571 at batch_size 128): the two 64-row halves of every minibatch run on
 * two workgroup pairs of ONE launch, each with its own replica (theta0/1, adam_m0/1, adam_v0/1, stale_sq_io0/1, losses0/1,
 * sync_ws0/1) and its own rows (perm0 / perm1: M_half entries each), exchanging the gradient inside the step through
 * regions2[0..1] (spo_p2p_alloc; no IPC).  Co-resident by construction -- no second stream.  Consumes ceil(M_half / batch)
 * tags from step0.  spo_p2p_selftest_one_grid: the exchange self-test in the same one-grid shape, result4_dev =
 * {wrong values, timeout} per rank. */
int spo_critic_fit_iter_split(float* theta0, float* adam_m0, float* adam_v0, float* theta1, float* adam_m1, float* adam_v1,
                              int64_t adam_step_host, const float* obs, const float* target_r, const float* target_c,
                              const int32_t* perm0, const int32_t* perm1, int64_t M_half, const spo_ppo_cfg* cfg_host,
                              float* stale_sq_io0, float* stale_sq_io1, float* losses0, float* losses1, void* sync_ws0,
                              void* sync_ws1, void* const* regions2, uint32_t step0, void* stream);
int spo_p2p_selftest_one_grid(void* const* regions2, uint32_t step0, int iters, int32_t* result4_dev, void* stream);
int spo_ppo_lag_update_iter_dp(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs,
                               const float* act, const float* logp_old, const float* target_r, const float* target_c,
                               const float* adv, const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host,
                               float* losses_out, void* sync_ws, int rank, int world, void* const* regions,
                               uint32_t step0, void* stream);

/* ---- multi-agent masked GAE: SeparatedReplayBuffer.compute_returns + compute_cost_returns
 * (safepo/common/buffer.py:356-384) with PopArt.denormalize (safepo/common/popart.py:117-133) folded in.
 * Time-major arrays as in the reference: rewards/costs [T, N], value_preds/cost_preds/masks [T+1, N] (row T =
 * bootstrap value / last mask), returns/cost_returns [T+1, N] (rows 0..T-1 written).  fp32, the reference's
 * operation order: results are bit-identical.  denorm_std = sqrt(debiased var), denorm_mean = debiased mean
 * (pass 1, 0 when no value normaliser is used). */
int spo_ma_gae(const float* rewards, const float* costs, const float* value_preds, const float* cost_preds,
               const float* masks, float* returns, float* cost_returns, int64_t T, int64_t num_threads,
               double gamma, double gae_lambda, float denorm_std_r, float denorm_mean_r,
               float denorm_std_c, float denorm_mean_c, void* stream);

/* parameter-vector geometry helpers (host) */
int64_t spo_param_count(int obs_dim, int act_dim);
int64_t spo_param_offset(int obs_dim, int act_dim, int net /*0 r-critic,1 c-critic,2 actor*/);

/* ---- bench/test utility (NOT a reference function): device-resident synthetic env
 * (SURVEY.md 8d): obs'~N(0,1), reward~N(0,1), cost~Bernoulli(p_cost), terminated~Bernoulli(p_term),
 * truncated = episode length >= trunc_len; counter-based RNG keyed by (seed, step, env). */
int spo_synth_env_step(float* next_obs, float* final_obs, float* reward, float* cost, float* terminated,
                       float* truncated, int* t_env, int64_t num_envs, int obs_dim, uint64_t seed,
                       uint64_t step, float p_term, float p_cost, int trunc_len, void* stream);

/* The same with the step counter split into a device-resident base and a host offset: step = *step_base_dev + step_rel.  A launch
 * captured into a HIP graph (step_rel = the index inside the epoch) then draws fresh numbers at every replay once the host has
 * moved *step_base_dev to the epoch's first step (the collect loop as one graph replay, safepo/common/engine.py).  One launch;
 * affine != 0: next_obs = fl(fl(x * obs_scale) + obs_shift), the raw-observation map of SynthDeviceEnv (obs <= 1024 columns). */
int spo_synth_env_step_rel(float* next_obs, float* final_obs, float* reward, float* cost, float* terminated,
                           float* truncated, int* t_env, int64_t num_envs, int obs_dim, uint64_t seed,
                           uint64_t step_rel, const uint64_t* step_base_dev, float p_term, float p_cost, int trunc_len,
                           int affine, float obs_scale, float obs_shift, void* stream);

/* ---- f3: multi-agent MAPPO-L networks and update (csrc/ma_net.hip).
 * Networks: safepo/common/model.py:172-363 + safepo/utils/{mlp,act,distributions}.py -- LayerNorm(obs), then n_blocks x
 * [Linear -> ELU -> LayerNorm], then a Linear head (actor: action mean, with a state-independent
 * std = sigmoid(log_std / std_x_coef) * std_y_coef; critics: one value).  One flat fp32 parameter vector per network in the
 * reference's state_dict order (spo_ma_param_offset: which = 0 feature_norm.weight, 1 feature_norm.bias, 2 W_k, 3 b_k,
 * 4 ln_k.weight, 5 ln_k.bias, 6 log_std, 7 head W, 8 head b).  GEMMs run on the in-tree fp32 MFMA kernels (rocBLAS is only the comparator of spo_debug_ma_gemm).
 * spo_ma_forward keeps the activations of `rows` rows in ws (spo_ma_workspace_floats) for spo_ma_backward, which turns
 * d(loss)/d(head output) into the flat gradient (log_std's entry is owned by spo_ma_actor_loss).
 * Trainer pieces (safepo/multi_agent/mappolag.py:126-199): spo_ma_actor_loss = clipped surrogate on
 * imp = prod_a exp(logp_a - old_a) with the hybrid advantage adv - lamda*cost_adv, HAPPO factor, active masks and entropy
 * bonus; scalars5_out = {policy_loss, dist_entropy, mean(imp), mean(imp*cost_adv), sum(active)};
 * spo_ma_lamda_update = the in-loop multiplier step; spo_ma_popart_forward = PopArt.forward (popart.py:86-112) on a
 * [rows] vector, state3 = {running_mean, running_mean_sq, debiasing_term}; spo_ma_value_loss = max of the clipped /
 * unclipped Huber losses (util.huber_loss, including its zero branch for e < -delta) with separately normalised targets;
 * spo_ma_clip_adam = clip_grad_norm_ + torch.optim.Adam(lr, eps, weight_decay) on one network.
 * HAPPO / MAPPO (safepo/multi_agent/{happo,mappo}.py) run on the same entry points: no cost critic / multiplier (pass
 * lamda = 0), spo_ma_value_loss with active_or_null != NULL and denom_host = the global sum of active masks for
 * use_value_active_masks (else NULL and the global row count), and per_dim_ratio = 1 for MAPPO's per-dimension ratios.
 * Data parallel over rollout threads: every mean is taken over the GLOBAL batch -- denom_host (global row count, or the
 * global sum of active masks) and rows_global are passed in, scalars / losses come back as this rank's share of the
 * global mean, and the caller all-reduces (sum) the scalars, the PopArt sums and the flat gradients before
 * spo_ma_lamda_update / spo_ma_popart_forward / spo_ma_clip_adam.  Single rank: rows_global = rows. */
typedef struct spo_ma_net {
  int32_t in_dim, hidden, n_blocks, out_dim, is_actor;
} spo_ma_net;
typedef struct spo_ma_loss_cfg {
  float clip_param, entropy_coef, std_x_coef, std_y_coef;
  int32_t use_policy_active_masks;
  int32_t per_dim_ratio;      /* MAPPO (mappo.py:150-160): per-dimension ratios, min(surr1, surr2) summed over dimensions */
} spo_ma_loss_cfg;
int64_t spo_ma_param_count(const spo_ma_net* net);
int64_t spo_ma_param_offset(const spo_ma_net* net, int which, int block);
int64_t spo_ma_workspace_floats(const spo_ma_net* net, int64_t rows);
int64_t spo_ma_backward_scratch_floats(const spo_ma_net* net, int64_t rows);
int spo_ma_forward(const float* theta, const spo_ma_net* net, const float* x, int64_t rows, float* ws, float* out,
                   void* stream);
int spo_ma_backward(const float* theta, const spo_ma_net* net, const float* x, int64_t rows, const float* ws,
                    const float* dout, float* grad, float* scratch, void* stream);
/* Tangent (forward-mode) pass of the same network: dout[rows, out_dim] = d(out)/d(theta) . tangent at the point whose
 * activations spo_ma_forward left in ws (tangent: flat, laid out like theta; a log_std slot is ignored).  With
 * spo_ma_backward this gives MACPO's Fisher-vector product J^T M J v without a second-order autograd pass
 * (replaces the double torch.autograd.grad of safepo/multi_agent/macpo.py:187-199).
 * scratch: float[spo_ma_jvp_scratch_floats(net, rows)]. */
int64_t spo_ma_jvp_scratch_floats(const spo_ma_net* net, int64_t rows);
int spo_ma_jvp(const float* theta, const spo_ma_net* net, const float* tangent, int64_t rows, const float* ws, float* dout,
               float* scratch, void* stream);
/* Insert of one environment step into the per-agent buffers (Runner.insert, mappolag.py:449-492) as ONE launch: the
 * environment's [threads, agents, ...] observations / shared observations / rewards / costs are written to every agent's
 * time-major buffer row, masks and active masks are formed from the done flags (dones: [threads, agents] bytes, non-zero =
 * done): masks = 0 where all agents of the thread are done, active_masks = 0 for a done agent of a thread that is not.
 * Each *_dst addresses the row of agent 0 at the step slot being filled ([threads, dim] contiguous); *_agent_stride is the
 * distance in floats to the same row of the next agent.  costs / costs_dst may both be NULL (happo / mappo). */
int spo_ma_insert_step(const float* obs, const float* share_obs, const float* rewards, const float* costs,
                       const unsigned char* dones, float* obs_dst, int64_t obs_agent_stride, float* share_obs_dst,
                       int64_t share_obs_agent_stride, float* rewards_dst, int64_t rewards_agent_stride, float* costs_dst,
                       int64_t costs_agent_stride, float* masks_dst, int64_t masks_agent_stride, float* active_masks_dst,
                       int64_t active_masks_agent_stride, int64_t num_threads, int32_t num_agents, int32_t obs_dim,
                       int32_t share_obs_dim, void* stream);
/* Collect step of the multi-agent runner (mappolag.py:411-447: policy.get_actions for every agent): ALL networks of ALL
 * agents in one launch, each 64-row tile taken through its whole network on chip (feature LayerNorm, blocks, head, and for
 * actors the Gaussian sample + per-dimension log-probabilities).  Results are bit-identical to spo_ma_forward (+
 * spo_ma_sample) per network.  Geometry served: hidden 128, in_dim <= 128 and a multiple of 4, out_dim <= 16, n_blocks <= 4,
 * n_nets <= SPO_MA_COLLECT_MAX_NETS, 16-byte aligned theta / x; anything else returns SPO_MA_COLLECT_UNSUPPORTED without
 * launching (the caller then runs the per-network entry points).
 * critics: out[rows, 1] required.  actors: act / logp [rows, out_dim] required, eps [rows, out_dim] unless deterministic,
 * out (the mean) optional.  scratch: float[spo_ma_collect_scratch_floats(n_nets)] (folded first-block weights). */
#define SPO_MA_COLLECT_MAX_NETS 16
#define SPO_MA_COLLECT_UNSUPPORTED 1
typedef struct spo_ma_collect_net {
  const float* theta;
  spo_ma_net net;
  int32_t deterministic;
  const float* x;
  float* out;
  const float* eps;
  float* act;
  float* logp;
  float std_x_coef, std_y_coef;
} spo_ma_collect_net;
int64_t spo_ma_collect_scratch_floats(int32_t n_nets);
int spo_ma_collect_forward(int32_t n_nets, const spo_ma_collect_net* nets, int64_t rows, float* scratch, void* stream);
int spo_ma_sample(const float* mean, const float* log_std, const float* eps, float std_x_coef, float std_y_coef,
                  int deterministic, float* act_out, float* logp_out, int64_t rows, int act_dim, void* stream);
int spo_ma_log_probs(const float* mean, const float* log_std, const float* act, float std_x_coef, float std_y_coef,
                     float* logp_out, int64_t rows, int act_dim, void* stream);
int spo_ma_actor_loss(const float* mean, const float* log_std, const float* act, const float* old_logp, const float* adv,
                      const float* cost_adv, const float* factor, const float* active, const float* lamda_dev,
                      const spo_ma_loss_cfg* cfg, int64_t rows, int act_dim, float denom_host, int64_t rows_global,
                      float* dmean_out, float* dlogstd_out, float* scalars5_out, double* partial_ws, void* stream);
int spo_ma_lamda_update(float* lamda_dev, const float* scalars5, float aver_episode_cost, float cost_limit, float gamma,
                        float lagrangian_coef_rate, void* stream);
int spo_ma_popart_stats(const float* x, int64_t rows, double* sums2_dev, double* partial_ws, void* stream);
int spo_ma_popart_forward(const float* x, int64_t rows, float* state3, double beta, float epsilon, int train,
                          const double* sums2_dev, int64_t rows_global, float* out, void* stream);
int spo_ma_value_loss(const float* values, const float* value_preds, const float* returns_norm_clipped,
                      const float* returns_norm_original, const float* active_or_null, float denom_host, float clip_param,
                      float huber_delta, float value_loss_coef, int64_t rows, int64_t rows_global, float* dvalues_out,
                      float* loss_out, double* partial_ws, void* stream);
int spo_ma_clip_adam(float* theta, const float* grad, float* adam_m, float* adam_v, int64_t n, int64_t adam_step_host,
                     float lr, float adam_eps, float weight_decay, float max_grad_norm, int use_max_grad_norm,
                     float* grad_norm_out, double* partial_ws, void* stream);

/* Test comparator for the multi-agent networks' plain products: y[B,N] = x[B,K] w[N,K]^T (mode 0) or y[B,K] = x[B,N] w[N,K]
 * (mode 1) through the hand-written fp32 MFMA kernel (use_rocblas = 0, what spo_ma_forward / backward / jvp run) or through
 * rocBLAS (use_rocblas = 1; dlopen'ed on demand, not used by any product path). */
int spo_debug_ma_gemm(int use_rocblas, int mode, const float* x, const float* w, float* y, int64_t B, int K, int N, void* stream);

/* ---- wide single-agent networks: ActorVCritic(obs_dim, act_dim, hidden_sizes) for any hidden_sizes (reference
 * safepo/common/model.py:30-48,131; isaac_gym_specific_cfg = [1024, 1024, 512], safepo/single_agent/ppo_lag.py:54-65).
 * A network is n_layers Linear layers with tanh between them; dims[0] = input width, dims[n_layers] = output width.
 * Flat layout of one network: for each layer W [out, in] row-major, then b [out] (nn.Sequential.parameters() order).
 * spo_mlp_forward leaves every layer's activations in ws (float[spo_mlp_workspace_floats]); the output is the LAST
 * rows * dims[n_layers] floats of ws.  spo_mlp_backward takes d(loss)/d(output) and writes the network's flat gradient;
 * scratch: float[spo_mlp_backward_scratch_floats].
 * spo_wide_ppo_loss (ppo_lag.py:306-323): MSE of both critics (WITHOUT their L2 terms), the clipped surrogate, their output
 * gradients and d(loss)/d(log_std); partial_ws: double[>= 256 * (3 + SPO_WIDE_MAX_ACT) + 512].
 * spo_wide_clip_adam (ppo_lag.py:310-329): adds the critics' L2 gradient 2 * l2_coef * p (cfg->use_critic_norm) and the value
 * coefficient (cfg->use_value_coefficient) to `grad` in place, adds l2_coef * sum p^2 to losses3_inout[0..1], clips the joint
 * norm over all n_params to cfg->max_grad_norm and takes one Adam step with cfg->lr_critic for [0, actor_begin) and
 * cfg->lr_actor beyond.  scalars4_out = {clip coefficient, L2 term of the reward critic, of the cost critic, ||g||};
 * partial_ws: double[>= 3 * 1024]. */
#define SPO_MLP_MAX_LAYERS 5
typedef struct spo_mlp_net {
  int32_t n_layers;
  int32_t dims[SPO_MLP_MAX_LAYERS + 1];
} spo_mlp_net;
int64_t spo_mlp_param_count(const spo_mlp_net* net);
int64_t spo_mlp_workspace_floats(const spo_mlp_net* net, int64_t rows);
int64_t spo_mlp_backward_scratch_floats(const spo_mlp_net* net, int64_t rows);
int spo_mlp_forward(const float* theta, const spo_mlp_net* net, const float* x, int64_t rows, float* ws, void* stream);
int spo_mlp_backward(const float* theta, const spo_mlp_net* net, const float* x, int64_t rows, const float* ws,
                     const float* d_out, float* grad, float* scratch, void* stream);
/* Up to 128 rows (and layer widths whose row images fit LDS: csrc/mlp_small.hip) spo_mlp_forward / spo_mlp_backward run the
 * WHOLE network in one launch each (one workgroup, weights staged through LDS layer by layer) instead of a GEMM launch per
 * layer: the reference's default minibatch of 64 is launch-latency territory.  SPO_MLP_SMALL=0 keeps the per-layer launches.
 * The _multi forms take `count` (<= 4) networks on the same number of rows -- the networks of a minibatch step -- and put them
 * into ONE launch (one workgroup per network) when all of them fit, else call the single-network entry point per network. */
int spo_mlp_forward_multi(int count, const float* const* thetas, const spo_mlp_net* const* nets, const float* const* xs,
                          int64_t rows, float* const* wss, void* stream);
int spo_mlp_backward_multi(int count, const float* const* thetas, const spo_mlp_net* const* nets, const float* const* xs,
                           int64_t rows, const float* const* wss, const float* const* d_outs, float* const* grads,
                           float* const* scratches, void* stream);
/* Round 6: the forward / loss / backward of a whole PPO-Lagrangian minibatch (ppo_lag.py:298-324: DataLoader batch, both critics'
 * MSE, the clipped surrogate, loss.backward()) for ANY hidden_sizes at up to 256 rows in ONE launch, split over the rows
 * (csrc/mlp_rows.hip): networks x ceil(rows / 16) workgroups, each carrying 16 rows of one network through every layer both ways
 * with the activations in LDS and the weights read as MFMA operands straight from global memory; the rows are
 * idx[*cursor_dev + i] (cursor_dev NULL: idx[i]; idx NULL: the rows themselves) of the FULL arrays obs [M, obs_dim], act, logp_old,
 * targets, adv -- no gather launch.  theta = [reward critic | cost critic | log_std | actor] (the ActorVCritic layout), `critic` /
 * `actor` describe the networks; actor NULL: the two critics only (the critic fit of cpo.py:541-556).  Every workgroup writes its
 * row group's partial gradient (theta's layout, WITHOUT the L2 terms) and loss sums into parts (float[spo_wide_grad_rows_part_floats]);
 * spo_wide_reduce_parts adds the groups in fixed order into grad[n_params] and forms losses_out[0 .. n_losses) = {MSE reward critic,
 * MSE cost critic, clipped surrogate}: what spo_gather_rows + spo_mlp_forward_multi + spo_wide_ppo_loss + spo_mlp_backward_multi
 * leave in `grad` / losses3, and where the data-parallel all-reduce and spo_wide_clip_adam take over.  SPO_WIDE_ROWS=0 makes
 * spo_wide_grad_rows_supported return 0 (callers then keep the launch-per-network path). */
int spo_wide_grad_rows_supported(const spo_mlp_net* critic, const spo_mlp_net* actor, int64_t rows);
int64_t spo_wide_grad_rows_part_floats(int64_t n_params, int64_t rows);
int spo_wide_ppo_grad_rows(const float* theta, const spo_mlp_net* critic, const spo_mlp_net* actor, const float* obs,
                           const float* act, const float* logp_old, const float* target_r, const float* target_c, const float* adv,
                           const int64_t* idx, const int64_t* cursor_dev, int64_t rows, float clip, float* parts, void* stream);
int spo_wide_reduce_parts(const float* parts, int64_t rows, int64_t n_params, int n_losses, float* grad, float* losses_out,
                          void* stream);
/* spo_wide_reduce_parts + spo_wide_clip_adam_dev_log (all parameters in the norm and in the Adam range: ppo_lag.py:310-329) in TWO
 * launches instead of four: the group sum rides in the pass that adds the L2 gradient and forms the norm partials (its first
 * workgroup also advances the optimiser clocks and the cursor); every workgroup of the Adam pass forms the clip coefficient from the
 * partials itself, the first one logs the losses.  Element for element the arithmetic of the four launches.  One GPU (a
 * data-parallel step all-reduces between the sum and the clip). */
int spo_wide_rows_clip_adam_dev_log(float* parts, int64_t rows, float* theta, float* grad, float* adam_m, float* adam_v,
                                    int64_t n_params, int64_t reward_critic_end, int64_t cost_critic_end, int64_t actor_begin,
                                    const spo_ppo_cfg* cfg, double* pow4_dev, float* losses3_out, float* scalars4_out,
                                    double* partial_ws, int partial_capacity, float* loss_log_dev, int64_t* cursor_dev,
                                    int64_t cursor_step, void* stream);
/* dsts[k][i, :] = srcs[k][idx[i], :] for k < count (<= SPO_GATHER_MAX) row-major arrays of widths[k] floats per row: the
 * minibatch gather of a step (the reference's DataLoader, ppo_lag.py:298-305) in one launch. */
#define SPO_GATHER_MAX 8
int spo_gather_rows(int count, const float* const* srcs, const int* widths, float* const* dsts, const int64_t* idx, int64_t n,
                    void* stream);
/* The same on the window idx[*cursor_dev .. *cursor_dev + n) of a longer index vector (cursor_dev: device int64, may be NULL =
 * 0): a launch captured into a HIP graph then walks a permutation, one minibatch per replay, with spo_wide_clip_adam_dev_log
 * advancing the cursor. */
int spo_gather_rows_at(int count, const float* const* srcs, const int* widths, float* const* dsts, const int64_t* idx,
                       const int64_t* cursor_dev, int64_t n, void* stream);
/* rsample + log-prob of a diagonal Gaussian (model.py:149-170; eps == NULL: deterministic), and the row sum of
 * KL(N(mean_old, exp(log_std_old)) || N(mean_new, exp(log_std_new))).sum(-1) (ppo_lag.py:338-345) added to (accumulate != 0) or
 * stored into *sum_inout -- the full batch is evaluated in row chunks. */
int spo_gauss_sample(const float* mean, const float* log_std, const float* eps, float* act_out, float* logp_out, int64_t rows,
                     int act_dim, void* stream);
int spo_gauss_kl_sum(const float* mean_old, const float* log_std_old, const float* mean_new, const float* log_std_new,
                     int64_t rows, int act_dim, double* partial_ws, int partial_capacity, double* sum_inout, int accumulate,
                     void* stream);
int spo_wide_ppo_loss(const float* v_r, const float* v_c, const float* mean, const float* log_std, const float* act,
                      const float* logp_old, const float* adv, const float* tgt_r, const float* tgt_c, int64_t rows,
                      int act_dim, float clip, float* d_vr, float* d_vc, float* d_mean, float* d_log_std, float* losses3,
                      double* partial_ws, int partial_capacity, void* stream);
int spo_wide_clip_adam(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                       int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, int64_t adam_step_host,
                       float* losses3_inout, float* scalars4_out, double* partial_ws, int partial_capacity, void* stream);


/* ---- round 4: the wide path as the fallback for any (obs_dim, act_dim, hidden_sizes) and every single-agent script.  The
 * reference's ActorVCritic takes any dims (safepo/common/model.py:131) and its default sweep (safepo/single_agent/benchmark.py:5-44)
 * pairs cpo / pcpo / rcpo / trpo_lag / focops / cup / ppo_lag / cppo_pid with tasks of 72-88 observations (Car, Doggo, Racecar)
 * and with HumanoidVelocity (376 observations, 17 actions).  act_dim <= SPO_WIDE_MAX_ACT.
 * spo_wide_actor_loss -- actor loss of rows [0, rows) of a (possibly chunked) batch of rows_total rows and d(loss)/d(mean):
 *   mode 0  clipped PPO surrogate -mean(min(ratio*adv, clamp(ratio, 1-p0, 1+p0)*adv))          (ppo_lag.py:316-319)
 *   mode 1  p0 * mean(ratio*adv), p0 = sign: the CPO / TRPO surrogates                          (cpo.py:356-381)
 *   mode 2  KL-penalty loss of FOCOPS / CUP's second stage, p0 = kl_bound, p1 = pg_coef (see spo_update_iter_ex); needs
 *           old_mean[rows, act_dim], old_std[act_dim]; minibatch only (rows == rows_total, accumulate == 0).
 *   sums_inout double[2 + act_dim] = {sum of the loss terms, sum of the KL indicators, d(loss)/d(log_std)[act_dim]}, added to
 *   (accumulate != 0) or overwritten; loss_out / d_log_std_out (optional, float) receive the finished loss and d(log_std).
 *   partial_ws: double[>= 256 * (2 + SPO_WIDE_MAX_ACT)].
 * spo_wide_critic_loss -- MSE of both critics and their output gradients (critic fit, cpo.py:541-556); losses2[0..1];
 *   partial_ws: double[>= 512].
 * spo_mlp_jvp -- forward-mode tangent dout[rows, out] = d(out)/d(theta) . tangent (tangent laid out like the network's theta)
 *   at the activations spo_mlp_forward left in ws; with spo_wide_fvp_cotangent (d_mean = (J t) / sigma^2 / (rows_total * act_dim))
 *   and spo_mlp_backward this is one chunk of the Fisher-vector product J^T diag(1/sigma^2) J t / (M * A) of cpo.py:132-157
 *   (the reference differentiates the KL twice).  scratch: float[spo_mlp_jvp_scratch_floats].
 * spo_wide_linesearch_sums -- sums3 = {sum ratio*adv_a, sum ratio*adv_b, sum_{rows,dims} KL(old || new)} (cpo.py:473-491) of
 *   one row chunk, added to (accumulate != 0) or stored into sums3_inout.  partial_ws: double[3 * blocks], blocks <= 512.
 * spo_wide_clip_adam_ex -- spo_wide_clip_adam with (i) the joint norm taken over [norm_begin, n_params) (0: all parameters;
 *   actor_begin - act_dim: CUP's actor-only clip, cup.py:385), (ii) Adam applied to [adam_begin, adam_end) only, the
 *   critics' optimisers at adam_step_critics_host and the actor's at adam_step_actor_host, (iii) scale_rest != 0: gradients
 *   outside the Adam range multiplied by the clip coefficient in place -- what clip_grad_norm_ over ALL policy parameters does to
 *   the actor's stale gradient during the critic fit (cpo.py:557).  The critics' L2 term / value coefficient are applied when
 *   norm_begin == 0 (as in spo_wide_clip_adam). */
int spo_wide_actor_loss(int mode, const float* mean, const float* log_std, const float* act, const float* logp_old,
                        const float* adv, const float* old_mean, const float* old_std, int64_t rows, int64_t rows_total,
                        int act_dim, float p0, float p1, float* d_mean, double* sums_inout, int accumulate, float* loss_out,
                        float* d_log_std_out, double* partial_ws, int partial_capacity, void* stream);
int spo_wide_critic_loss(const float* v_r, const float* v_c, const float* tgt_r, const float* tgt_c, int64_t rows, float* d_vr,
                         float* d_vc, float* losses2, double* partial_ws, int partial_capacity, void* stream);
int64_t spo_mlp_jvp_scratch_floats(const spo_mlp_net* net, int64_t rows);
int spo_mlp_jvp(const float* theta, const spo_mlp_net* net, const float* tangent, const float* x, int64_t rows, const float* ws,
                float* dout, float* scratch, void* stream);
int spo_wide_fvp_cotangent(const float* jv, const float* log_std, int64_t rows, int64_t rows_total, int act_dim, float* d_mean,
                           void* stream);
int spo_wide_linesearch_sums(const float* mean_new, const float* log_std_new, const float* act, const float* logp_old,
                             const float* adv_a, const float* adv_b, const float* mean_old, const float* log_std_old, int64_t rows,
                             int act_dim, double* partial_ws, int partial_capacity, double* sums3_inout, int accumulate,
                             void* stream);
/* spo_wide_clip_adam_dev -- spo_wide_clip_adam_ex with the optimiser clocks on the DEVICE: pow4_dev = double[6] = {beta1^t, beta2^t of
 * the critics' optimisers, beta1^t, beta2^t of the actor's} before this step, then {lr_actor, lr_critic} (each: >= 0 overrides the
 * cfg's value, so a schedule does not change a captured launch's arguments; negative: the cfg's -- ABI 2); the clocks of the optimisers inside the Adam range
 * advance on the device.  No argument changes from one minibatch step to the next, so the launch sequence of a step (gathers,
 * spo_mlp_forward / backward, loss kernels, this) can be captured once as a HIP graph and replayed: the wide path at the reference's
 * default batch of 64 is launch-bound (~70 launches per step).  The caller keeps pow4_dev in step with its host-side step counts. */
int spo_wide_clip_adam_dev(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                           int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, double* pow4_dev, int64_t adam_begin,
                           int64_t adam_end, int64_t norm_begin, int scale_rest, float* losses3_inout, float* scalars4_out,
                           double* partial_ws, int partial_capacity, void* stream);
/* spo_wide_clip_adam_dev that also (cursor_dev != NULL) stores the step's three losses log_src_dev[0..2] (after the L2 terms were
 * added) at loss_log_dev[3 * (*cursor_dev / cursor_step)] and advances *cursor_dev by cursor_step: together with
 * spo_gather_rows_at a replayed minibatch step takes no host copy in (indices) or out (losses). */
int spo_wide_clip_adam_dev_log(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                               int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, double* pow4_dev, int64_t adam_begin,
                               int64_t adam_end, int64_t norm_begin, int scale_rest, float* losses3_inout, float* scalars4_out,
                               double* partial_ws, int partial_capacity, float* loss_log_dev, const float* log_src_dev,
                               int64_t* cursor_dev, int64_t cursor_step, void* stream);
int spo_wide_clip_adam_ex(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                          int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, int64_t adam_step_critics_host,
                          int64_t adam_step_actor_host, int64_t adam_begin, int64_t adam_end, int64_t norm_begin, int scale_rest,
                          float* losses3_inout, float* scalars4_out, double* partial_ws, int partial_capacity, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAFEPO_HIP_H */
