// Collect-side kernels: ActorVCritic.step + buffer.store, bootstrap values, path-boundary logic,
// full-batch actor forward and KL early-stop statistic, synthetic device env.  gfx950 only.
//
// References (relative to /root/reference):
//   safepo/common/model.py:149-170   ActorVCritic.step          -> policy_step_kernel
//   safepo/common/buffer.py:84-95    VectorizedOnPolicyBuffer.store (fused into the same kernel)
//   safepo/single_agent/ppo_lag.py:198-234  boundary / bootstrap / episode stats -> boundary_kernel
//   safepo/single_agent/ppo_lag.py:277,338-345  old/new distribution + KL      -> actor_kl_kernel
#include "common.h"
#include "mlp_mfma.h"
#include "../../include/safepo_hip.h"

namespace {
using namespace spo;

struct StepArgs {
  const float* theta; const float* obs; const float* eps;
  float* act; float* logp; float* v_r; float* v_c;
  float* buf_obs; float* buf_act; float* buf_logp; float* buf_v_r; float* buf_v_c;
  int64_t N; int64_t T; int64_t t; int D; int A;
  const double* rms;     // optional running (mean[D], var[D], count): normalise the observation row on load (a-2 fused)
  float* obs_io;         // with rms: the normalised row is also written back here (the wrapper returns normalised obs)
  double rms_eps;
};

// 256 threads = 4 waves; wave w owns rows [64*block + 16w, +16).  Networks are staged through one
// LDS image in turn (critic_r, critic_c, actor) so obs_dim up to 128 fits.
template <int KIN, bool WITH_ACTOR>
__global__ __launch_bounds__(256) void policy_step_kernel(StepArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[NetLds<KIN>::SIZE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int D = a.D, A = a.A;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 16 + j;
  const bool valid = row < a.N;
  const int64_t rrow = valid ? row : a.N - 1;
  f4 x[KIN / 16];
  if (a.rms) {
    // in-place form: obs_io aliases obs, so the row is read through the pointer it is written through (no __restrict__
    // promise), and lanes past the last row read nothing -- clamping them to row N-1 would race with that row's owner
    // storing its normalised values
    const float* row_io = a.obs_io + rrow * D;
    if ((D & 3) == 0) {
#pragma unroll
      for (int nt = 0; nt < KIN / 16; ++nt) {
        const int c = 16 * nt + 4 * q;
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        x[nt] = zero;
        if (valid && c < D) x[nt] = *reinterpret_cast<const f4*>(row_io + c);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < KIN / 16; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 16 * nt + 4 * q + e;
          x[nt][e] = 0.f;
          if (valid && c < D) x[nt][e] = row_io[c];
        }
    }
  } else {
    load_obs_tiles<KIN>(a.obs + rrow * D, D, q, x);
  }
  if (a.rms) {
    // NormalizeObservation.normalize (reference wrappers.py:42-49 -> gymnasium): (obs - mean) / sqrt(var + 1e-8) in float64
    // with the statistics spo_obs_stats_update has just merged this batch into, rounded to fp32 once -- in registers, so the
    // raw row is read once and the network, the buffer slot and the caller all get the normalised values
#pragma unroll
    for (int nt = 0; nt < KIN / 16; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 16 * nt + 4 * q + e;
        if (c < D) {
          const double v = ((double)x[nt][e] - a.rms[c]) / sqrt(a.rms[D + c] + a.rms_eps);
          x[nt][e] = (float)v;
          if (valid) a.obs_io[row * D + c] = x[nt][e];
        }
      }
  }
  float vout[2] = {0.f, 0.f};
  f4 mu = {0.f, 0.f, 0.f, 0.f};
  constexpr int NNET = WITH_ACTOR ? 3 : 2;
  for (int net = 0; net < NNET; ++net) {
    if (net > 0) __syncthreads();
    stage_net<KIN>(a.theta, net_geom(D, A, net), lds, tid, 256);
    __syncthreads();
    f4 h1[4], h2[4];
    f4 o = net_forward<KIN>(lds, x, h1, h2, j, q);
    if (net < 2) vout[net] = o[0]; else mu = o;
  }
  const int64_t slot = rrow * a.T + a.t;
  if (valid && q == 0) {
    a.v_r[row] = vout[0];
    a.v_c[row] = vout[1];
    if (a.buf_v_r) { a.buf_v_r[slot] = vout[0]; a.buf_v_c[slot] = vout[1]; }
  }
  if (a.buf_obs && valid) {
#pragma unroll
    for (int nt = 0; nt < KIN / 16; ++nt) {
      const int c = 16 * nt + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < D) a.buf_obs[slot * D + c + e] = x[nt][e];
    }
  }
  if constexpr (WITH_ACTOR) {
    const int ls_off = 2 * critic_size(D);
    float lp = 0.f;
    float av[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ai = 4 * q + r;
      if (ai < A) {
        const float sd = expf(a.theta[ls_off + ai]);            // std = exp(log_std)   model.py:80
        float ac = mu[r];
        if (a.eps) ac = mu[r] + a.eps[rrow * A + ai] * sd;       // rsample: loc + eps*scale
        const float diff = ac - mu[r];
        const float var = sd * sd;
        lp += -(diff * diff) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;   // Normal.log_prob
        av[r] = ac;
      }
    }
    lp += __shfl_xor(lp, 16);
    lp += __shfl_xor(lp, 32);                                    // .sum(axis=-1)   model.py:167
    if (valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        if (ai < A) {
          a.act[row * A + ai] = av[r];
          if (a.buf_act) a.buf_act[slot * A + ai] = av[r];
        }
      }
      if (q == 0) {
        a.logp[row] = lp;
        if (a.buf_logp) a.buf_logp[slot] = lp;
      }
    }
  }
}

// Round 4: the networks side by side.  The kernel above is a latency chain on 64 of the 256 CUs -- stage critic_r, forward,
// barrier, stage critic_c, forward, barrier, stage actor, forward: 29 us per collect step at 4 096 envs (14 us for the two
// critics alone), the longest kernels of the rollout.  Here a workgroup takes ROWS = 16 * RW rows and runs the NNET networks
// CONCURRENTLY: wave group g (RW waves) stages network g into its own LDS image and carries its forward pass, so one staging
// phase and one forward pass deep instead of three, on twice as many CUs (RW = 2: 128 workgroups of 6 waves).  The observation
// rows are loaded (and normalised: the fp64 divide / square root per element, split over the groups by column tile) ONCE into an
// LDS tile every group reads -- which also removes the read-raw / write-normalised race of an in-place row between groups.
// Per row the arithmetic is net_forward on the same LDS layout: results are bit-identical to policy_step_kernel.
// With rms_count_add != 0 workgroup 0 adds the batch size to the normaliser's count (the merge kernel in front of this one has
// read the old count by then) -- one launch less per step.
struct BoundaryArgs {
  const float* reward; const float* cost; const float* terminated; const float* truncated;
  const float* v_next_r; const float* v_next_c; const float* v_final_r; const float* v_final_c;
  float* buf_reward; float* buf_cost; uint8_t* seg_end; float* boot_r; float* boot_c;
  double* ep_ret; double* ep_cost; double* ep_len; double* events; int* events_count; int events_capacity;
  int64_t N; int64_t T; int64_t t; int epoch_end;
  float* fold_reward; float* fold_cost; float gamma32;     // optional: reward / cost with gamma * bootstrap folded in at path ends
};

// BOUNDARY (critics only): the rows are an env step's final observations, and the step's path-boundary logic (boundary_mb_kernel
// below: reward / cost store, episode accumulators, seg_end / bootstrap marks with the values just computed, ordered episode
// log) runs in the same launch on the workgroup's ROWS envs -- one launch less per collect step, and the bootstrap values go
// from the MFMA accumulators to their consumer through LDS.
struct StepParArgs {
  StepArgs s;
  double rms_count_add;
  BoundaryArgs b;
  int* events_prefix;
};
template <int KIN, bool WITH_ACTOR, int RW, bool BOUNDARY = false>
__global__ __launch_bounds__((WITH_ACTOR ? 3 : 2) * RW * 64) void policy_step_par_kernel(StepParArgs pa) {
  static_assert(!(BOUNDARY && WITH_ACTOR) && RW * 16 <= 64, "boundary form: critics only, one wave of envs");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const StepArgs& a = pa.s;
  using L = NetLds<KIN>;
  constexpr int NNET = WITH_ACTOR ? 3 : 2, GT = RW * 64, ROWS = RW * 16, LDX = KIN + 4, NT = KIN / 16;
  float* xs = lds + NNET * L::SIZE;                 // [ROWS][LDX] observation rows as the networks see them
  const int tid = threadIdx.x, g = tid / GT, gtid = tid % GT, lane = tid & 63, rw = gtid >> 6, j = lane & 15, q = lane >> 4;
  const int D = a.D, A = a.A;
  const int rl = rw * 16 + j;                       // row inside the workgroup
  const int64_t row = (int64_t)blockIdx.x * ROWS + rl;
  const bool valid = row < a.N;
  const int64_t rrow = valid ? row : a.N - 1;
  const int64_t slot = rrow * a.T + a.t;
  // BOUNDARY: finished episodes among the envs in front of this workgroup's, and (lanes 0 .. ROWS-1) this env's step record --
  // loaded here, in front of the forward pass they do not depend on
  int fin_before = 0, ev_base = 0;
  float b_rwd = 0.f, b_cs = 0.f, b_term = 0.f, b_trunc = 0.f, b_vnr = 0.f, b_vnc = 0.f;
  double b_ret = 0, b_cst = 0, b_len = 0;
  if constexpr (BOUNDARY) {
    const BoundaryArgs& b = pa.b;
    const int64_t c0 = (int64_t)blockIdx.x * ROWS;
    for (int64_t i = tid; i < c0; i += NNET * GT) fin_before += (b.terminated[i] != 0.f || b.truncated[i] != 0.f) ? 1 : 0;
    if (tid < ROWS && c0 + tid < b.N) {
      const int64_t i = c0 + tid;
      ev_base = pa.events_prefix[b.t];
      b_rwd = b.reward[i]; b_cs = b.cost[i]; b_term = b.terminated[i]; b_trunc = b.truncated[i];
      b_ret = b.ep_ret[i]; b_cst = b.ep_cost[i]; b_len = b.ep_len[i];
      if (b.epoch_end) { b_vnr = b.v_next_r[i]; b_vnc = b.v_next_c[i]; }
    }
  }
  // this group's share of the observation tile: column tiles nt = g, g + NNET, ...
  const float* src = (a.rms ? a.obs_io : a.obs) + rrow * D;
  f4 xin[(NT + NNET - 1) / NNET];
#pragma unroll
  for (int k = 0; k < (NT + NNET - 1) / NNET; ++k) {
    const int c = 16 * (g + k * NNET) + 4 * q;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (g + k * NNET < NT) {
      if ((D & 3) == 0) {
        if (c < D) v = *reinterpret_cast<const f4*>(src + c);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < D) v[e] = src[c + e];
      }
    }
    xin[k] = v;
  }
  stage_net_batched<KIN, GT>(a.theta, net_geom(D, A, g), lds + g * L::SIZE, gtid);
#pragma unroll
  for (int k = 0; k < (NT + NNET - 1) / NNET; ++k) {
    const int nt = g + k * NNET;
    if (nt < NT) {
      const int c = 16 * nt + 4 * q;
      f4 v = xin[k];
      if (a.rms) {
        // NormalizeObservation.normalize (wrappers.py:42-49 -> gymnasium): (obs - mean) / sqrt(var + 1e-8) in float64 with the
        // statistics just merged, rounded to fp32 once (the expression of policy_step_kernel)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < D) {
            v[e] = (float)(((double)v[e] - a.rms[c + e]) / sqrt(a.rms[D + c + e] + a.rms_eps));
            if (valid) a.obs_io[row * D + c + e] = v[e];
          }
      }
      *reinterpret_cast<f4*>(xs + rl * LDX + c) = v;
      if (a.buf_obs && valid) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < D) a.buf_obs[slot * D + c + e] = v[e];
      }
    }
  }
  __syncthreads();
  f4 x[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) x[nt] = *reinterpret_cast<const f4*>(xs + rl * LDX + 16 * nt + 4 * q);
  f4 h1[4], h2[4];
  const f4 o = net_forward<KIN>(lds + g * L::SIZE, x, h1, h2, j, q);
  if (g < 2) {
    if (valid && q == 0) {
      float* v = g == 0 ? a.v_r : a.v_c;
      float* bv = g == 0 ? a.buf_v_r : a.buf_v_c;
      v[row] = o[0];
      if (bv) bv[slot] = o[0];
    }
    if (g == 0 && blockIdx.x == 0 && tid == 0 && pa.rms_count_add != 0.0) const_cast<double*>(a.rms)[2 * D] += pa.rms_count_add;
    if constexpr (!BOUNDARY) return;
  }
  if constexpr (BOUNDARY) {
    const BoundaryArgs& b = pa.b;
    int* vsh_cnt = reinterpret_cast<int*>(xs + ROWS * LDX);        // [NNET * RW] per-wave counts
    float* vsh = xs + ROWS * LDX + 8;                              // [2][ROWS] values of the final observations
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) fin_before += __shfl_xor(fin_before, off);
    if (lane == 0) vsh_cnt[tid >> 6] = fin_before;
    if (q == 0) vsh[g * ROWS + rl] = o[0];
    __syncthreads();
    if (tid >= 64) return;
    int before_blocks = 0;
#pragma unroll
    for (int w = 0; w < NNET * RW; ++w) before_blocks += vsh_cnt[w];
    const int base = __shfl(ev_base, 0);
    const int64_t i = (int64_t)blockIdx.x * ROWS + tid;
    const bool in = tid < ROWS && i < b.N;
    bool fin = false;
    double ret = 0, cst = 0, len = 0;
    if (in) {
      const float rwd = b_rwd, cs = b_cs;
      const bool done = b_term != 0.f, tout = b_trunc != 0.f;
      ret = b_ret + (double)rwd;                // ep_ret += reward  (float64 accumulators, ppo_lag.py:168-170)
      cst = b_cst + (double)cs;
      len = b_len + 1.0;
      const bool boundary = b.epoch_end || done || tout;
      float br = 0.f, bc = 0.f;
      if (boundary && !done) {
        if (b.epoch_end) { br = b_vnr; bc = b_vnc; }
        if (tout) { br = vsh[tid]; bc = vsh[ROWS + tid]; }          // final_observation wins (ppo_lag.py:209-213)
      }
      const int64_t bslot = i * b.T + b.t;
      b.buf_reward[bslot] = rwd;
      b.buf_cost[bslot] = cs;
      b.seg_end[bslot] = boundary ? 1 : 0;
      b.boot_r[bslot] = br;
      b.boot_c[bslot] = bc;
      if (b.fold_reward) {
        b.fold_reward[bslot] = boundary ? __fadd_rn(rwd, __fmul_rn(b.gamma32, br)) : rwd;
        b.fold_cost[bslot] = boundary ? __fadd_rn(cs, __fmul_rn(b.gamma32, bc)) : cs;
      }
      fin = done || tout;
      b.ep_ret[i] = fin ? 0.0 : ret;
      b.ep_cost[i] = fin ? 0.0 : cst;
      b.ep_len[i] = fin ? 0.0 : len;
    }
    const unsigned long long ball = __ballot(fin);
    if (fin) {
      const int pos = base + before_blocks + __popcll(ball & ((1ull << lane) - 1ull));
      if (pos < b.events_capacity) {
        double* e = b.events + (int64_t)pos * 4;
        e[0] = (double)(b.t * b.N + i);
        e[1] = ret; e[2] = cst; e[3] = len;
      }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) pa.events_prefix[b.t + 1] = base + before_blocks + __popcll(ball);
    return;
  }
  if constexpr (WITH_ACTOR) {
    const f4 mu = o;
    const int ls_off = 2 * critic_size(D);
    float lp = 0.f;
    float av[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ai = 4 * q + r;
      if (ai < A) {
        const float sd = expf(a.theta[ls_off + ai]);            // std = exp(log_std)   model.py:80
        float ac = mu[r];
        if (a.eps) ac = mu[r] + a.eps[rrow * A + ai] * sd;       // rsample: loc + eps*scale
        const float diff = ac - mu[r];
        const float var = sd * sd;
        lp += -(diff * diff) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;   // Normal.log_prob
        av[r] = ac;
      }
    }
    lp += __shfl_xor(lp, 16);
    lp += __shfl_xor(lp, 32);                                    // .sum(axis=-1)   model.py:167
    if (valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        if (ai < A) {
          a.act[row * A + ai] = av[r];
          if (a.buf_act) a.buf_act[slot * A + ai] = av[r];
        }
      }
      if (q == 0) {
        a.logp[row] = lp;
        if (a.buf_logp) a.buf_logp[slot] = lp;
      }
    }
  }
}

struct KlArgs {
  const float* theta; const float* obs; const float* mean_old; const float* log_std_old;
  float* mean_out; double* partials; int64_t rows; int D; int A;
};

// MODE 0: write means; MODE 1: accumulate KL(old || new).sum(-1) over rows.
// Full-batch actor forward (ppo_lag.py:277,338-345): FP32-matrix bound (16.9 kFLOP per row against 240 B of observation).
// A wave walks 16-row tiles; three waves share a SIMD (net_forward_lean keeps the kernel at 112-140 registers; rounds 1-2
// used 293, i.e. ONE wave per SIMD, so a tile's load latency, its 144 MFMAs and its 32 tanh ran strictly one after the other:
// 0.29 of the FP32 matrix peak), so one wave's tanh / KL arithmetic runs beside the others' MFMAs, and the next tile's
// observation rows are requested before the current tile is computed (raw loads, masked at pick-up).  The grid is
// persistent (three workgroups per CU: 3 x 39.7 KB of LDS): the network is staged once per workgroup.
#ifndef SPO_FULL_MINB
#define SPO_FULL_MINB 1          // minimum resident workgroups per SIMD-set the register budget is shaped for (A/B knob)
#endif
#ifndef SPO_FULL_GRID
#define SPO_FULL_GRID 768        // persistent grid: three resident workgroups on each of the 256 CUs (A/B knob)
#endif
#ifndef SPO_FULL_PAD
// Weight-row padding of the forward-only full-batch kernel.  8 makes its ds_read_b128 operand reads conflict-free (NetLds in
// mlp_mfma.h) -- and measured nothing: 123.5 us against 123.4 us for the KL over 524 288 rows (profiles/r04/kl_ab_pad.txt): the
// conflicts are 43 % of the LDS-active cycles, but with three waves per SIMD the LDS is not what the waves wait for.  Kept at 4.
#define SPO_FULL_PAD 4
#endif
template <int KIN, int MODE>
__global__ __launch_bounds__(256, SPO_FULL_MINB) void actor_full_kernel(KlArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[NetLds<KIN, SPO_FULL_PAD>::SIZE];
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int D = a.D, A = a.A;
  stage_net<KIN, SPO_FULL_PAD>(a.theta, net_geom(D, A, 2), lds, tid, 256);
  __syncthreads();
  const int ls_off = 2 * critic_size(D);
  float sd_new[4], sd_old[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ai = 4 * q + r;
    sd_new[r] = ai < A ? expf(a.theta[ls_off + ai]) : 1.f;
    sd_old[r] = (MODE == 1 && ai < A) ? expf(a.log_std_old[ai]) : 1.f;
  }
  double acc = 0.0;
  const int64_t ntiles = (a.rows + 15) / 16;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  f4 xn[KIN / 16];
  {
    const int64_t r0 = tile * 16 + j;
    load_obs_tiles_raw<KIN>(a.obs + (r0 < a.rows ? r0 : a.rows - 1) * D, D, q, xn);
  }
  for (; tile < ntiles; tile += stride) {
    const int64_t row = tile * 16 + j;
    const bool valid = row < a.rows;
    const int64_t rrow = valid ? row : a.rows - 1;
    f4 x[KIN / 16];
#pragma unroll
    for (int nt = 0; nt < KIN / 16; ++nt) x[nt] = xn[nt];
    mask_obs_tiles<KIN>(D, q, x);
    float mo[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        mo[r] = a.mean_old[rrow * A + (ai < A ? ai : 0)];          // unconditional (clamped) loads, used after the forward
      }
    }
    {
      const int64_t nrow = (tile + stride) * 16 + j;                // next tile of this wave (clamped: a harmless re-read at the end)
      load_obs_tiles_raw<KIN>(a.obs + (nrow < a.rows ? nrow : a.rows - 1) * D, D, q, xn);
    }
    const f4 mu = net_forward_lean<KIN, SPO_FULL_PAD>(lds, x, j, q);
    if (MODE == 0) {
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (4 * q + r < A) a.mean_out[row * A + 4 * q + r] = mu[r];
      }
    } else {
      float kl = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        if (ai < A) {
          // torch.distributions.kl._kl_normal_normal(p=old, q=new)
          const float ratio = sd_old[r] / sd_new[r];
          const float var_ratio = ratio * ratio;
          const float dm = (mo[r] - mu[r]) / sd_new[r];
          const float t1 = dm * dm;
          kl += 0.5f * (var_ratio + t1 - 1.f - logf(var_ratio));
        }
      }
      if (valid) acc += (double)kl;
    }
  }
  if (MODE == 1) {
    acc = wave_sum_d(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) a.partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* partials, int n, double* out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}


// ONE block: finished episodes are appended in env order, the order of the reference's Python
// loop `for idx, (done, time_out) in enumerate(zip(terminated, truncated))` (ppo_lag.py:199).
__global__ __launch_bounds__(1024) void boundary_kernel(BoundaryArgs a) {
  // one block of up to 16 waves (round 3: 1024 threads -- 4 passes over 4096 envs instead of 16, 24.7 -> ~9 us per step)
  __shared__ int wave_cnt[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = *a.events_count;
  __syncthreads();
  const int nthr = blockDim.x, nwaves = nthr >> 6;
  for (int64_t c0 = 0; c0 < a.N; c0 += nthr) {
    const int64_t i = c0 + tid;
    const bool in = i < a.N;
    bool fin = false;
    double ret = 0, cst = 0, len = 0;
    if (in) {
      const float rw = a.reward[i], cs = a.cost[i];
      const bool done = a.terminated[i] != 0.f, tout = a.truncated[i] != 0.f;
      ret = a.ep_ret[i] + (double)rw;           // ep_ret += reward  (float64 accumulators, ppo_lag.py:168-170)
      cst = a.ep_cost[i] + (double)cs;
      len = a.ep_len[i] + 1.0;
      const bool boundary = a.epoch_end || done || tout;
      float br = 0.f, bc = 0.f;
      if (boundary && !done) {
        if (a.epoch_end) { br = a.v_next_r[i]; bc = a.v_next_c[i]; }
        if (tout) { br = a.v_final_r[i]; bc = a.v_final_c[i]; }     // final_observation wins (ppo_lag.py:209-213)
      }
      const int64_t slot = i * a.T + a.t;
      a.buf_reward[slot] = rw;
      a.buf_cost[slot] = cs;
      a.seg_end[slot] = boundary ? 1 : 0;
      a.boot_r[slot] = br;
      a.boot_c[slot] = bc;
      if (a.fold_reward) {
        // what spo_gae_fused reads in its folded form: fl(r + fl(gamma32 * bootstrap)) at a path end, r elsewhere --
        // the first two of the three fp32 operations of delta_t (buffer.py:198) for the appended bootstrap value
        a.fold_reward[slot] = boundary ? __fadd_rn(rw, __fmul_rn(a.gamma32, br)) : rw;
        a.fold_cost[slot] = boundary ? __fadd_rn(cs, __fmul_rn(a.gamma32, bc)) : cs;
      }
      fin = done || tout;
      a.ep_ret[i] = fin ? 0.0 : ret;
      a.ep_cost[i] = fin ? 0.0 : cst;
      a.ep_len[i] = fin ? 0.0 : len;
    }
    const unsigned long long ball = __ballot(fin);
    const int before = __popcll(ball & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(ball);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wave_cnt[w];
    if (fin) {
      const int pos = base + woff + before;
      if (pos < a.events_capacity) {
        double* e = a.events + (int64_t)pos * 4;
        e[0] = (double)(a.t * a.N + i);
        e[1] = ret; e[2] = cst; e[3] = len;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int add = 0;
      for (int w = 0; w < nwaves; ++w) add += wave_cnt[w];
      base += add;
    }
    __syncthreads();
  }
  if (tid == 0) *a.events_count = base;
}

// Multi-workgroup form (round 4).  The one-block kernel above is a latency chain: four passes over 4 096 envs, each with ten
// dependent loads, six scattered 4-byte stores (stride T * 4 bytes: one cache line per lane) and three barriers -- 24 us per
// step, the longest kernel of the collect loop.  Here a workgroup takes 256 envs in ONE pass.  The event log must stay in env
// order (the order of the reference's Python loop), so a finished episode's position is base + (finished episodes of ALL envs
// before it): a workgroup counts the finished episodes of the envs in front of its chunk itself (b * 256 flag pairs, L2-resident,
// b loads per thread) instead of waiting for the other workgroups.  The running count lives in a prefix array indexed by the
// step -- the kernel of step t reads events_prefix[t] and the last workgroup writes events_prefix[t + 1] -- so no workgroup ever
// reads a word another workgroup of the same launch writes.
__global__ __launch_bounds__(256) void boundary_mb_kernel(BoundaryArgs a, int* __restrict__ events_prefix) {
  __shared__ int wave_cnt[4];
  __shared__ int before_sh[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * 256;
  const int base = events_prefix[a.t];
  // finished episodes among the envs in front of this chunk
  int cnt = 0;
  for (int64_t i = tid; i < c0; i += 256) cnt += (a.terminated[i] != 0.f || a.truncated[i] != 0.f) ? 1 : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (lane == 0) before_sh[wave] = cnt;
  const int64_t i = c0 + tid;
  const bool in = i < a.N;
  bool fin = false;
  double ret = 0, cst = 0, len = 0;
  if (in) {
    const float rw = a.reward[i], cs = a.cost[i];
    const bool done = a.terminated[i] != 0.f, tout = a.truncated[i] != 0.f;
    ret = a.ep_ret[i] + (double)rw;
    cst = a.ep_cost[i] + (double)cs;
    len = a.ep_len[i] + 1.0;
    const bool boundary = a.epoch_end || done || tout;
    float br = 0.f, bc = 0.f;
    if (boundary && !done) {
      if (a.epoch_end) { br = a.v_next_r[i]; bc = a.v_next_c[i]; }
      if (tout) { br = a.v_final_r[i]; bc = a.v_final_c[i]; }       // final_observation wins (ppo_lag.py:209-213)
    }
    const int64_t slot = i * a.T + a.t;
    a.buf_reward[slot] = rw;
    a.buf_cost[slot] = cs;
    a.seg_end[slot] = boundary ? 1 : 0;
    a.boot_r[slot] = br;
    a.boot_c[slot] = bc;
    if (a.fold_reward) {
      a.fold_reward[slot] = boundary ? __fadd_rn(rw, __fmul_rn(a.gamma32, br)) : rw;
      a.fold_cost[slot] = boundary ? __fadd_rn(cs, __fmul_rn(a.gamma32, bc)) : cs;
    }
    fin = done || tout;
    a.ep_ret[i] = fin ? 0.0 : ret;
    a.ep_cost[i] = fin ? 0.0 : cst;
    a.ep_len[i] = fin ? 0.0 : len;
  }
  const unsigned long long ball = __ballot(fin);
  const int before = __popcll(ball & ((1ull << lane) - 1ull));
  if (lane == 0) wave_cnt[wave] = __popcll(ball);
  __syncthreads();
  const int before_blocks = (before_sh[0] + before_sh[1]) + (before_sh[2] + before_sh[3]);
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wave_cnt[w];
  if (fin) {
    const int pos = base + before_blocks + woff + before;
    if (pos < a.events_capacity) {
      double* e = a.events + (int64_t)pos * 4;
      e[0] = (double)(a.t * a.N + i);
      e[1] = ret; e[2] = cst; e[3] = len;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0)
    events_prefix[a.t + 1] = base + before_blocks + (wave_cnt[0] + wave_cnt[1]) + (wave_cnt[2] + wave_cnt[3]);
}

// ---------------------------------------------------------------- synthetic env (bench/test utility)
__device__ __forceinline__ void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ void normal4(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, float (&o)[4]) {
  uint32_t ctr[4] = {a, b, c, d};
  philox4x32(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float r0 = sqrtf(-2.f * logf(u01(ctr[0]))), r1 = sqrtf(-2.f * logf(u01(ctr[2])));
  const float t0 = 6.283185307179586f * u01(ctr[1]), t1 = 6.283185307179586f * u01(ctr[3]);
  o[0] = r0 * cosf(t0); o[1] = r0 * sinf(t0); o[2] = r1 * cosf(t1); o[3] = r1 * sinf(t1);
}

// step_base (optional, device): added to `step` -- a captured launch (step = the index inside the epoch) then draws fresh
// numbers at every replay once the host has moved the base to the epoch's first step
__global__ void synth_flags_kernel(float* reward, float* cost, float* terminated, float* truncated, int* t_env,
                                   int64_t N, uint64_t seed, uint64_t step, float p_term, float p_cost, int trunc_len,
                                   const unsigned long long* __restrict__ step_base) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (step_base) step += step_base[0];
  uint32_t ctr[4] = {(uint32_t)i, (uint32_t)step, 0x5eedf1a6u, (uint32_t)(step >> 32)};
  philox4x32(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  float nrm[4];
  normal4(seed, (uint32_t)i, (uint32_t)step, 0x72657761u, (uint32_t)(step >> 32), nrm);
  const int te = t_env[i] + 1;
  const bool term = u01(ctr[0]) < p_term;
  const bool trunc = (te >= trunc_len) && !term;
  reward[i] = nrm[0];
  cost[i] = u01(ctr[1]) < p_cost ? 1.f : 0.f;
  terminated[i] = term ? 1.f : 0.f;
  truncated[i] = trunc ? 1.f : 0.f;
  t_env[i] = (term || trunc) ? 0 : te;
}

__global__ void synth_obs_kernel(float* next_obs, float* final_obs, const float* terminated, const float* truncated,
                                 int64_t N, int D, uint64_t seed, uint64_t step, const unsigned long long* __restrict__ step_base) {
  const int chunks = (D + 3) / 4;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N * chunks) return;
  if (step_base) step += step_base[0];
  const int64_t i = g / chunks;
  const int c = (int)(g % chunks) * 4;
  float o[4], f[4];
  normal4(seed, (uint32_t)i, (uint32_t)step, 0x6f627330u + (uint32_t)c, (uint32_t)(step >> 32), o);
  const bool fin = terminated[i] != 0.f || truncated[i] != 0.f;
  if (fin) normal4(seed, (uint32_t)i, (uint32_t)step, 0x66696e30u + (uint32_t)c, (uint32_t)(step >> 32), f);
  for (int e = 0; e < 4 && c + e < D; ++e) {
    next_obs[i * D + c + e] = o[e];
    final_obs[i * D + c + e] = fin ? f[e] : 0.f;
  }
}

// The two kernels above and the env's affine map of the raw observation (x * scale + shift, two elementwise launches) in ONE
// launch (round 4: four launches of ~5 us each per collect step were the synthetic env's whole cost).  A workgroup owns whole
// envs -- lane (env, chunk) -- so every chunk lane recomputes its env's flags from the old episode clock, and the chunk-0
// lane stores them and the new clock after a barrier.  Same counters, same arithmetic (the affine map as two rounded
// operations, like the two tensor ops it replaces): bit-identical outputs.
__global__ __launch_bounds__(256) void synth_step_kernel(float* next_obs, float* final_obs, float* reward, float* cost, float* terminated,
                                                         float* truncated, int* t_env, int64_t N, int D, uint64_t seed, uint64_t step,
                                                         const unsigned long long* __restrict__ step_base, float p_term, float p_cost,
                                                         int trunc_len, int affine, float scale, float shift) {
  const int chunks = (D + 3) / 4;
  const int epb = 256 / chunks;                        // envs per workgroup
  const int el = threadIdx.x / chunks, c = (threadIdx.x % chunks) * 4;
  const int64_t i = (int64_t)blockIdx.x * epb + el;
  const bool act = el < epb && i < N;
  if (step_base) step += step_base[0];
  bool term = false, trunc = false;
  int te = 0;
  uint32_t ctr[4] = {(uint32_t)i, (uint32_t)step, 0x5eedf1a6u, (uint32_t)(step >> 32)};
  if (act) {
    philox4x32(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    te = t_env[i] + 1;
    term = u01(ctr[0]) < p_term;
    trunc = (te >= trunc_len) && !term;
  }
  __syncthreads();                                     // every lane of the env has read the old clock
  if (act) {
    if (c == 0) {
      float nrm[4];
      normal4(seed, (uint32_t)i, (uint32_t)step, 0x72657761u, (uint32_t)(step >> 32), nrm);
      reward[i] = nrm[0];
      cost[i] = u01(ctr[1]) < p_cost ? 1.f : 0.f;
      terminated[i] = term ? 1.f : 0.f;
      truncated[i] = trunc ? 1.f : 0.f;
      t_env[i] = (term || trunc) ? 0 : te;
    }
    float o[4], f[4];
    normal4(seed, (uint32_t)i, (uint32_t)step, 0x6f627330u + (uint32_t)c, (uint32_t)(step >> 32), o);
    const bool fin = term || trunc;
    if (fin) normal4(seed, (uint32_t)i, (uint32_t)step, 0x66696e30u + (uint32_t)c, (uint32_t)(step >> 32), f);
    for (int e = 0; e < 4 && c + e < D; ++e) {
      next_obs[i * D + c + e] = affine ? __fadd_rn(__fmul_rn(o[e], scale), shift) : o[e];
      final_obs[i * D + c + e] = fin ? f[e] : 0.f;
    }
  }
}

// ---------------------------------------------------------------- a-2: running observation normaliser
// gymnasium.wrappers.normalize.RunningMeanStd.update + NormalizeObservation.normalize (the wrapper
// SafeNormalizeObservation applies inside the env, reference safepo/common/wrappers.py:42-49):
// batch mean / biased variance over the N envs per feature, parallel-variance merge into the running
// (mean, var, count) in fp64, then obs <- (obs - mean) / sqrt(var + 1e-8) in the input precision (fp32 out).
// One workgroup per feature column block; fp64 throughout like the numpy original.
__global__ __launch_bounds__(256) void obs_normalize_kernel(float* obs, double* rms /*[2*D+1]: mean[D], var[D], count*/,
                                                            int64_t N, int D, int update, double eps, int normalize) {
  __shared__ double sh_s[256], sh_q[256];
  const int f = blockIdx.x;                     // feature
  const int tid = threadIdx.x;
  double* mean = rms; double* var = rms + D; double* count = rms + 2 * D;
  if (update) {
    // pass 1: batch mean ; pass 2: batch variance around it (numpy's x.mean(0), x.var(0))
    double s = 0.0;
    for (int64_t i = tid; i < N; i += 256) s += (double)obs[i * D + f];
    sh_s[tid] = s;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) { if (tid < k) sh_s[tid] += sh_s[tid + k]; __syncthreads(); }
    const double bmean = sh_s[0] / (double)N;
    __syncthreads();
    double qv = 0.0;
    for (int64_t i = tid; i < N; i += 256) { const double d = (double)obs[i * D + f] - bmean; qv += d * d; }
    sh_q[tid] = qv;
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) { if (tid < k) sh_q[tid] += sh_q[tid + k]; __syncthreads(); }
    if (tid == 0) {
      const double bvar = sh_q[0] / (double)N, bn = (double)N, c = *count;   // every block reads the OLD count
      const double delta = bmean - mean[f], tot = c + bn;
      const double m2 = var[f] * c + bvar * bn + delta * delta * c * bn / tot;
      mean[f] = mean[f] + delta * bn / tot;
      var[f] = m2 / tot;
    }
    __syncthreads();
  }
  if (!normalize) return;                        // statistics only: spo_policy_step_norm normalises on load
  const double mu = mean[f];
  const double sd = sqrt(var[f] + eps);
  for (int64_t i = tid; i < N; i += 256) obs[i * D + f] = (float)(((double)obs[i * D + f] - mu) / sd);
}
__global__ void obs_normalize_count_kernel(double* rms, int D, int64_t N) { rms[2 * D] += (double)N; }

// The merge alone (what the collect step launches every step), for batches of at most 256 * RPT rows: the feature's column is
// loaded ONCE into registers (RPT independent loads in flight) and serves both passes; the block sums go through wave
// butterflies and one LDS exchange instead of two eight-barrier trees.  ~11 -> ~7 us per step at 4 096 envs.  Same two-pass
// mean / variance and the same merge as obs_normalize_kernel; the order of the fp64 partial sums differs (last-bit level).
template <int RPT>
__global__ __launch_bounds__(256) void obs_stats_kernel(const float* __restrict__ obs, double* rms, int64_t N, int D) {
  __shared__ double sh_s[4], sh_q[4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* mean = rms; double* var = rms + D; const double* count = rms + 2 * D;
  float v[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int64_t i = tid + 256 * k;
    v[k] = obs[(i < N ? i : N - 1) * D + f];
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < RPT; ++k)
    if (tid + 256 * k < N) s += (double)v[k];
  s = wave_sum_d(s);
  if (lane == 0) sh_s[wave] = s;
  __syncthreads();
  const double bmean = ((sh_s[0] + sh_s[1]) + (sh_s[2] + sh_s[3])) / (double)N;
  double qv = 0.0;
#pragma unroll
  for (int k = 0; k < RPT; ++k)
    if (tid + 256 * k < N) { const double d = (double)v[k] - bmean; qv += d * d; }
  qv = wave_sum_d(qv);
  if (lane == 0) sh_q[wave] = qv;
  __syncthreads();
  if (tid == 0) {
    const double bvar = ((sh_q[0] + sh_q[1]) + (sh_q[2] + sh_q[3])) / (double)N, bn = (double)N, c = *count;
    const double delta = bmean - mean[f], tot = c + bn;
    const double m2 = var[f] * c + bvar * bn + delta * delta * c * bn / tot;
    mean[f] = mean[f] + delta * bn / tot;
    var[f] = m2 / tot;
  }
}
// statistics merge of one batch without the count (see obs_normalize_count_kernel / policy_step_par_kernel)
void launch_obs_stats(const float* obs, double* rms, int64_t N, int D, hipStream_t st) {
  static const bool fast = [] { const char* e = getenv("SPO_OBS_STATS_REG"); return !(e && e[0] == '0'); }();
  if (fast && N <= 256 * 4) hipLaunchKernelGGL(obs_stats_kernel<4>, dim3(D), dim3(256), 0, st, obs, rms, N, D);
  else if (fast && N <= 256 * 16) hipLaunchKernelGGL(obs_stats_kernel<16>, dim3(D), dim3(256), 0, st, obs, rms, N, D);
  else hipLaunchKernelGGL(obs_normalize_kernel, dim3(D), dim3(256), 0, st, const_cast<float*>(obs), rms, N, D, 1, 1e-8, 0);
}

int pick_kin(int D) { return D <= 16 ? 16 : D <= 32 ? 32 : D <= 64 ? 64 : 128; }

// SPO_STEP_PAR=0: the one-network-at-a-time kernel everywhere (A/B knob; results are bit-identical)
bool step_par_enabled() {
  static const bool on = [] { const char* e = getenv("SPO_STEP_PAR"); return !(e && e[0] == '0'); }();
  return on;
}
constexpr int STEP_PAR_RW = 2;
constexpr int64_t STEP_PAR_MAX_ROWS = 32768;     // beyond this the 64-row kernel's fewer weight stagings win

template <int KIN, bool WITH_ACTOR, bool BOUNDARY = false>
int launch_step_par(const StepArgs& a, double count_add, hipStream_t st, const BoundaryArgs* b = nullptr, int* events_prefix = nullptr) {
  constexpr int NNET = WITH_ACTOR ? 3 : 2, RW = STEP_PAR_RW;
  constexpr size_t sh = (size_t)(NNET * NetLds<KIN>::SIZE + RW * 16 * (KIN + 4) + 8 + 2 * RW * 16) * sizeof(float);
  static bool done_dev[spo::SPO_MAX_DEVICES] = {};        /* the attribute is per device */
  bool& done = done_dev[spo::current_device_slot()];
  if (!done) {
    if (int rc = spo::hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_step_par_kernel<KIN, WITH_ACTOR, RW, BOUNDARY>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh),
                                "hipFuncSetAttribute(policy_step_par)")) return rc;
    done = true;
  }
  StepParArgs pa{a, count_add, b ? *b : BoundaryArgs{}, events_prefix};
  hipLaunchKernelGGL((policy_step_par_kernel<KIN, WITH_ACTOR, RW, BOUNDARY>), dim3((unsigned)((a.N + RW * 16 - 1) / (RW * 16))),
                     dim3(NNET * RW * 64), sh, st, pa);
  return 0;
}

// count_add != 0: the normaliser's count still has to grow by that much (see policy_step_par_kernel); returns 1 when the launch
// did it, 0 when the caller must (the sequential kernel), < 0 on error
template <bool WITH_ACTOR>
int launch_step(const StepArgs& a, hipStream_t st, double count_add = 0.0) {
  if (step_par_enabled() && a.N <= STEP_PAR_MAX_ROWS && a.D <= 64) {
    int rc;
    switch (pick_kin(a.D)) {
      case 16: rc = launch_step_par<16, WITH_ACTOR>(a, count_add, st); break;
      case 32: rc = launch_step_par<32, WITH_ACTOR>(a, count_add, st); break;
      default: rc = launch_step_par<64, WITH_ACTOR>(a, count_add, st); break;
    }
    return rc ? rc : 1;
  }
  const unsigned blocks = (unsigned)((a.N + 63) / 64);
  switch (pick_kin(a.D)) {
    case 16: hipLaunchKernelGGL((policy_step_kernel<16, WITH_ACTOR>), dim3(blocks), dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL((policy_step_kernel<32, WITH_ACTOR>), dim3(blocks), dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL((policy_step_kernel<64, WITH_ACTOR>), dim3(blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((policy_step_kernel<128, WITH_ACTOR>), dim3(blocks), dim3(256), 0, st, a); break;
  }
  return 0;
}

constexpr int FULL_GRID = SPO_FULL_GRID;
template <int MODE>
int launch_full(const KlArgs& a, unsigned blocks, hipStream_t st) {
  switch (pick_kin(a.D)) {
    case 16: hipLaunchKernelGGL((actor_full_kernel<16, MODE>), dim3(blocks), dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL((actor_full_kernel<32, MODE>), dim3(blocks), dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL((actor_full_kernel<64, MODE>), dim3(blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((actor_full_kernel<128, MODE>), dim3(blocks), dim3(256), 0, st, a); break;
  }
  return 0;
}

int check_dims(int D, int A) {
  if (D < 1 || D > SPO_MAX_OBS) return spo::fail(-2, "obs_dim %d outside [1,%d]", D, SPO_MAX_OBS);
  if (A < 1 || A > SPO_MAX_ACT) return spo::fail(-2, "act_dim %d outside [1,%d]", A, SPO_MAX_ACT);
  return 0;
}

}  // namespace

extern "C" int64_t spo_param_count(int obs_dim, int act_dim) {
  return 2 * (int64_t)spo::critic_size(obs_dim) + spo::actor_size(obs_dim, act_dim);
}
extern "C" int64_t spo_param_offset(int obs_dim, int act_dim, int net) {
  if (net == 0) return 0;
  if (net == 1) return spo::critic_size(obs_dim);
  return 2 * (int64_t)spo::critic_size(obs_dim);
}

extern "C" int spo_policy_step(const float* theta, const float* obs, const float* eps, float* act, float* logp,
                               float* v_r, float* v_c, float* buf_obs, float* buf_act, float* buf_logp,
                               float* buf_v_r, float* buf_v_c, int64_t num_envs, int64_t T, int64_t t, int obs_dim,
                               int act_dim, void* stream) {
  if (int rc = check_dims(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && obs && act && logp && v_r && v_c, "policy_step: null pointer");
  SPO_REQUIRE(num_envs > 0, "policy_step: num_envs must be > 0");
  const bool buf = buf_obs || buf_act || buf_logp || buf_v_r || buf_v_c;
  if (buf) {
    SPO_REQUIRE(buf_obs && buf_act && buf_logp && buf_v_r && buf_v_c, "policy_step: all buffer slots or none");
    SPO_REQUIRE(t >= 0 && t < T, "Buffer overflow");          /* reference assert, buffer.py:92 */
  }
  StepArgs a{theta, obs, eps, act, logp, v_r, v_c, buf_obs, buf_act, buf_logp, buf_v_r, buf_v_c,
             num_envs, T > 0 ? T : 1, buf ? t : 0, obs_dim, act_dim, nullptr, nullptr, 0.0};
  if (int rc = launch_step<true>(a, (hipStream_t)stream); rc < 0) return rc;
  SPO_LAUNCH_CHECK("spo_policy_step");
  return 0;
}

extern "C" int spo_policy_step_norm(const float* theta, float* obs_inout, double* rms_state, int update, const float* eps,
                                    float* act, float* logp, float* v_r, float* v_c, float* buf_obs, float* buf_act,
                                    float* buf_logp, float* buf_v_r, float* buf_v_c, int64_t num_envs, int64_t T,
                                    int64_t t, int obs_dim, int act_dim, void* stream) {
  if (int rc = check_dims(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && obs_inout && rms_state && act && logp && v_r && v_c, "policy_step_norm: null pointer");
  SPO_REQUIRE(num_envs > 0, "policy_step_norm: num_envs must be > 0");
  const bool buf = buf_obs || buf_act || buf_logp || buf_v_r || buf_v_c;
  if (buf) {
    SPO_REQUIRE(buf_obs && buf_act && buf_logp && buf_v_r && buf_v_c, "policy_step_norm: all buffer slots or none");
    SPO_REQUIRE(t >= 0 && t < T, "Buffer overflow");          /* reference assert, buffer.py:92 */
  }
  hipStream_t st = (hipStream_t)stream;
  // RunningMeanStd.update(batch): the merge needs the whole batch before any row can be normalised with the NEW statistics; the
  // count grows after every feature's merge has read the old one -- in the step kernel's workgroup 0 where that kernel can
  const bool count_in_step = update && step_par_enabled() && num_envs <= STEP_PAR_MAX_ROWS && obs_dim <= 64;
  if (update) {
    launch_obs_stats(obs_inout, rms_state, num_envs, obs_dim, st);
    if (!count_in_step) hipLaunchKernelGGL(obs_normalize_count_kernel, dim3(1), dim3(1), 0, st, rms_state, obs_dim, num_envs);
  }
  StepArgs a{theta, obs_inout, eps, act, logp, v_r, v_c, buf_obs, buf_act, buf_logp, buf_v_r, buf_v_c,
             num_envs, T > 0 ? T : 1, buf ? t : 0, obs_dim, act_dim, rms_state, obs_inout, 1e-8};
  if (int rc = launch_step<true>(a, st, count_in_step ? (double)num_envs : 0.0); rc < 0) return rc;
  SPO_LAUNCH_CHECK("spo_policy_step_norm");
  return 0;
}

extern "C" int spo_values(const float* theta, const float* obs, float* v_r, float* v_c, int64_t rows, int obs_dim,
                          int act_dim, void* stream) {
  if (int rc = check_dims(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && obs && v_r && v_c && rows > 0, "values: bad args");
  StepArgs a{theta, obs, nullptr, nullptr, nullptr, v_r, v_c, nullptr, nullptr, nullptr, nullptr, nullptr,
             rows, 1, 0, obs_dim, act_dim, nullptr, nullptr, 0.0};
  if (int rc = launch_step<false>(a, (hipStream_t)stream); rc < 0) return rc;
  SPO_LAUNCH_CHECK("spo_values");
  return 0;
}

static int boundary_step_impl(const float* reward, const float* cost, const float* terminated,
                              const float* truncated, const float* v_next_r, const float* v_next_c,
                              const float* v_final_r, const float* v_final_c, float* buf_reward, float* buf_cost,
                              uint8_t* seg_end, float* boot_r, float* boot_c, double* ep_ret, double* ep_cost,
                              double* ep_len, double* events, int* events_count, int events_capacity,
                              int64_t num_envs, int64_t T, int64_t t, int epoch_end, float* fold_reward, float* fold_cost,
                              double gamma, void* stream) {
  SPO_REQUIRE(reward && cost && terminated && truncated && buf_reward && buf_cost && seg_end && boot_r && boot_c &&
                  ep_ret && ep_cost && ep_len && events && events_count, "boundary: null pointer");
  SPO_REQUIRE(v_next_r && v_next_c && v_final_r && v_final_c, "boundary: null value pointer");
  SPO_REQUIRE((fold_reward == nullptr) == (fold_cost == nullptr), "boundary: fold_reward and fold_cost go together");
  SPO_REQUIRE(t >= 0 && t < T, "Buffer overflow");
  BoundaryArgs a{reward, cost, terminated, truncated, v_next_r, v_next_c, v_final_r, v_final_c, buf_reward, buf_cost,
                 seg_end, boot_r, boot_c, ep_ret, ep_cost, ep_len, events, events_count,
                 events_capacity, num_envs, T, t, epoch_end, fold_reward, fold_cost, (float)gamma};
  const int bthreads = num_envs >= 1024 ? 1024 : num_envs > 256 ? 512 : 256;
  hipLaunchKernelGGL(boundary_kernel, dim3(1), dim3(bthreads), 0, (hipStream_t)stream, a);
  SPO_LAUNCH_CHECK("spo_boundary_step");
  return 0;
}

extern "C" int spo_boundary_step_fold_mb(const float* reward, const float* cost, const float* terminated, const float* truncated,
                                         const float* v_next_r, const float* v_next_c, const float* v_final_r, const float* v_final_c,
                                         float* buf_reward, float* buf_cost, uint8_t* seg_end, float* boot_r, float* boot_c,
                                         double* ep_ret, double* ep_cost, double* ep_len, double* events, int* events_prefix,
                                         int events_capacity, int64_t num_envs, int64_t T, int64_t t, int epoch_end,
                                         float* fold_reward, float* fold_cost, double gamma, void* stream) {
  SPO_REQUIRE(reward && cost && terminated && truncated && buf_reward && buf_cost && seg_end && boot_r && boot_c &&
                  ep_ret && ep_cost && ep_len && events && events_prefix, "boundary_mb: null pointer");
  SPO_REQUIRE(v_next_r && v_next_c && v_final_r && v_final_c, "boundary_mb: null value pointer");
  SPO_REQUIRE((fold_reward == nullptr) == (fold_cost == nullptr), "boundary_mb: fold_reward and fold_cost go together");
  SPO_REQUIRE(t >= 0 && t < T && num_envs > 0, "Buffer overflow");
  BoundaryArgs a{reward, cost, terminated, truncated, v_next_r, v_next_c, v_final_r, v_final_c, buf_reward, buf_cost,
                 seg_end, boot_r, boot_c, ep_ret, ep_cost, ep_len, events, nullptr,
                 events_capacity, num_envs, T, t, epoch_end, fold_reward, fold_cost, (float)gamma};
  const int64_t blocks = (num_envs + 255) / 256;
  SPO_REQUIRE(blocks <= 65535 * 16, "boundary_mb: too many envs");
  hipLaunchKernelGGL(boundary_mb_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, events_prefix);
  SPO_LAUNCH_CHECK("spo_boundary_step_fold_mb");
  return 0;
}

extern "C" int spo_values_boundary_step_fold(const float* theta, const float* final_obs, float* v_final_r, float* v_final_c,
                                             int obs_dim, int act_dim, const float* reward, const float* cost,
                                             const float* terminated, const float* truncated, const float* v_next_r,
                                             const float* v_next_c, float* buf_reward, float* buf_cost, uint8_t* seg_end,
                                             float* boot_r, float* boot_c, double* ep_ret, double* ep_cost, double* ep_len,
                                             double* events, int* events_prefix, int events_capacity, int64_t num_envs,
                                             int64_t T, int64_t t, int epoch_end, float* fold_reward, float* fold_cost,
                                             double gamma, void* stream) {
  if (int rc = check_dims(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && final_obs && v_final_r && v_final_c, "values_boundary: null pointer");
  SPO_REQUIRE(reward && cost && terminated && truncated && v_next_r && v_next_c && buf_reward && buf_cost && seg_end && boot_r &&
              boot_c && ep_ret && ep_cost && ep_len && events && events_prefix, "values_boundary: null pointer");
  SPO_REQUIRE((fold_reward == nullptr) == (fold_cost == nullptr), "values_boundary: both fold outputs or neither");
  SPO_REQUIRE(num_envs > 0 && t >= 0 && t < T, "values_boundary: bad step index");
  hipStream_t st = (hipStream_t)stream;
  // (every 32-env workgroup counts the finished episodes of the envs in front of it: beyond 8 192 envs the 256-env workgroups
  //  of spo_boundary_step_fold_mb do an eighth of those loads)
  if (!(step_par_enabled() && num_envs <= 8192 && obs_dim <= 64)) {
    // outside the side-by-side kernel's envelope: the two launches this entry point stands for
    if (int rc = spo_values(theta, final_obs, v_final_r, v_final_c, num_envs, obs_dim, act_dim, stream)) return rc;
    return spo_boundary_step_fold_mb(reward, cost, terminated, truncated, v_next_r, v_next_c, v_final_r, v_final_c, buf_reward,
                                     buf_cost, seg_end, boot_r, boot_c, ep_ret, ep_cost, ep_len, events, events_prefix,
                                     events_capacity, num_envs, T, t, epoch_end, fold_reward, fold_cost, gamma, stream);
  }
  StepArgs a{theta, final_obs, nullptr, nullptr, nullptr, v_final_r, v_final_c, nullptr, nullptr, nullptr, nullptr, nullptr,
             num_envs, 1, 0, obs_dim, act_dim, nullptr, nullptr, 0.0};
  BoundaryArgs b{reward, cost, terminated, truncated, v_next_r, v_next_c, v_final_r, v_final_c, buf_reward, buf_cost, seg_end,
                 boot_r, boot_c, ep_ret, ep_cost, ep_len, events, nullptr, events_capacity, num_envs, T, t, epoch_end,
                 fold_reward, fold_cost, (float)gamma};
  int rc;
  switch (pick_kin(obs_dim)) {
    case 16: rc = launch_step_par<16, false, true>(a, 0.0, st, &b, events_prefix); break;
    case 32: rc = launch_step_par<32, false, true>(a, 0.0, st, &b, events_prefix); break;
    default: rc = launch_step_par<64, false, true>(a, 0.0, st, &b, events_prefix); break;
  }
  if (rc) return rc;
  SPO_LAUNCH_CHECK("spo_values_boundary_step_fold");
  return 0;
}

extern "C" int spo_boundary_step(const float* reward, const float* cost, const float* terminated,
                                 const float* truncated, const float* v_next_r, const float* v_next_c,
                                 const float* v_final_r, const float* v_final_c, float* buf_reward, float* buf_cost,
                                 uint8_t* seg_end, float* boot_r, float* boot_c, double* ep_ret, double* ep_cost,
                                 double* ep_len, double* events, int* events_count, int events_capacity,
                                 int64_t num_envs, int64_t T, int64_t t, int epoch_end, void* stream) {
  return boundary_step_impl(reward, cost, terminated, truncated, v_next_r, v_next_c, v_final_r, v_final_c, buf_reward,
                            buf_cost, seg_end, boot_r, boot_c, ep_ret, ep_cost, ep_len, events, events_count,
                            events_capacity, num_envs, T, t, epoch_end, nullptr, nullptr, 0.0, stream);
}

extern "C" int spo_boundary_step_fold(const float* reward, const float* cost, const float* terminated,
                                      const float* truncated, const float* v_next_r, const float* v_next_c,
                                      const float* v_final_r, const float* v_final_c, float* buf_reward, float* buf_cost,
                                      uint8_t* seg_end, float* boot_r, float* boot_c, double* ep_ret, double* ep_cost,
                                      double* ep_len, double* events, int* events_count, int events_capacity,
                                      int64_t num_envs, int64_t T, int64_t t, int epoch_end, float* fold_reward,
                                      float* fold_cost, double gamma, void* stream) {
  SPO_REQUIRE(fold_reward && fold_cost, "boundary_fold: null fold pointer");
  return boundary_step_impl(reward, cost, terminated, truncated, v_next_r, v_next_c, v_final_r, v_final_c, buf_reward,
                            buf_cost, seg_end, boot_r, boot_c, ep_ret, ep_cost, ep_len, events, events_count,
                            events_capacity, num_envs, T, t, epoch_end, fold_reward, fold_cost, gamma, stream);
}

extern "C" int spo_actor_mean(const float* theta, const float* obs, float* mean_out, int64_t rows, int obs_dim,
                              int act_dim, void* stream) {
  if (int rc = check_dims(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && obs && mean_out && rows > 0, "actor_mean: bad args");
  KlArgs a{theta, obs, nullptr, nullptr, mean_out, nullptr, rows, obs_dim, act_dim};
  int64_t blocks = (rows + 63) / 64;
  if (blocks > FULL_GRID) blocks = FULL_GRID;
  launch_full<0>(a, (unsigned)blocks, (hipStream_t)stream);
  SPO_LAUNCH_CHECK("spo_actor_mean");
  return 0;
}

extern "C" int spo_actor_kl(const float* theta, const float* obs, const float* mean_old, const float* log_std_old,
                            double* kl_partials, int kl_partials_capacity, double* kl_sum, int64_t rows, int obs_dim,
                            int act_dim, void* stream) {
  if (int rc = check_dims(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && obs && mean_old && log_std_old && kl_partials && kl_sum && rows > 0, "actor_kl: bad args");
  int64_t blocks = (rows + 63) / 64;
  if (blocks > FULL_GRID) blocks = FULL_GRID;
  if (blocks > kl_partials_capacity) blocks = kl_partials_capacity;
  SPO_REQUIRE(blocks >= 1, "actor_kl: partials capacity must be >= 1");
  KlArgs a{theta, obs, mean_old, log_std_old, nullptr, kl_partials, rows, obs_dim, act_dim};
  launch_full<1>(a, (unsigned)blocks, (hipStream_t)stream);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kl_partials, (int)blocks, kl_sum);
  SPO_LAUNCH_CHECK("spo_actor_kl");
  return 0;
}

static int synth_env_step_impl(float* next_obs, float* final_obs, float* reward, float* cost, float* terminated,
                               float* truncated, int* t_env, int64_t num_envs, int obs_dim, uint64_t seed, uint64_t step,
                               const unsigned long long* step_base_dev, float p_term, float p_cost, int trunc_len, void* stream) {
  SPO_REQUIRE(next_obs && final_obs && reward && cost && terminated && truncated && t_env && num_envs > 0 && obs_dim > 0,
              "synth_env: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(synth_flags_kernel, dim3((unsigned)((num_envs + 255) / 256)), dim3(256), 0, st, reward, cost,
                     terminated, truncated, t_env, num_envs, seed, step, p_term, p_cost, trunc_len, step_base_dev);
  const int64_t work = num_envs * ((obs_dim + 3) / 4);
  hipLaunchKernelGGL(synth_obs_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, next_obs, final_obs,
                     terminated, truncated, num_envs, obs_dim, seed, step, step_base_dev);
  SPO_LAUNCH_CHECK("spo_synth_env_step");
  return 0;
}
extern "C" int spo_synth_env_step(float* next_obs, float* final_obs, float* reward, float* cost, float* terminated,
                                  float* truncated, int* t_env, int64_t num_envs, int obs_dim, uint64_t seed,
                                  uint64_t step, float p_term, float p_cost, int trunc_len, void* stream) {
  return synth_env_step_impl(next_obs, final_obs, reward, cost, terminated, truncated, t_env, num_envs, obs_dim, seed, step, nullptr,
                             p_term, p_cost, trunc_len, stream);
}
extern "C" int spo_synth_env_step_rel(float* next_obs, float* final_obs, float* reward, float* cost, float* terminated,
                                      float* truncated, int* t_env, int64_t num_envs, int obs_dim, uint64_t seed,
                                      uint64_t step_rel, const uint64_t* step_base_dev, float p_term, float p_cost, int trunc_len,
                                      int affine, float obs_scale, float obs_shift, void* stream) {
  SPO_REQUIRE(step_base_dev, "synth_env_rel: null step base");
  SPO_REQUIRE(next_obs && final_obs && reward && cost && terminated && truncated && t_env && num_envs > 0 && obs_dim > 0,
              "synth_env_rel: bad args");
  if (obs_dim > 1024)        // (a workgroup owns whole envs: 256 lanes x 4 columns)
    return spo::fail(-2, "synth_env_rel: obs_dim %d > 1024", obs_dim);
  const int epb = 256 / ((obs_dim + 3) / 4);
  hipLaunchKernelGGL(synth_step_kernel, dim3((unsigned)((num_envs + epb - 1) / epb)), dim3(256), 0, (hipStream_t)stream, next_obs,
                     final_obs, reward, cost, terminated, truncated, t_env, num_envs, obs_dim, seed, step_rel,
                     reinterpret_cast<const unsigned long long*>(step_base_dev), p_term, p_cost, trunc_len, affine, obs_scale, obs_shift);
  SPO_LAUNCH_CHECK("spo_synth_env_step_rel");
  return 0;
}

extern "C" int spo_obs_normalize(float* obs, double* rms_state, int64_t num_envs, int obs_dim, int update, void* stream) {
  SPO_REQUIRE(obs && rms_state && num_envs > 0 && obs_dim > 0, "obs_normalize: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(obs_normalize_kernel, dim3(obs_dim), dim3(256), 0, st, obs, rms_state, num_envs, obs_dim, update, 1e-8, 1);
  if (update) hipLaunchKernelGGL(obs_normalize_count_kernel, dim3(1), dim3(1), 0, st, rms_state, obs_dim, num_envs);
  SPO_LAUNCH_CHECK("spo_obs_normalize");
  return 0;
}
