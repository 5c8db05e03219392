// Persistent minibatch update for WIDE OBSERVATIONS (obs_dim <= 512, act_dim <= 32, hidden [64, 64]), gfx950 -- round 5.
//
// The persistent kernels of update.hip keep one network in one CU's LDS; HumanoidVelocity's 376-wide first layer does not fit, so
// that shape ran on the launch-per-layer wide path at 95 us per 64-row step against 10.6 us (ppo_lag.py:297-336; the reference
// takes any dims, model.py:131, and its default sweep includes 376 / 17, single_agent/benchmark.py:5-22).  Here the FIRST LAYER IS
// SPLIT OVER THE INPUT FEATURES:
//
//   grid = n_nets networks x S slices (S = ceil(obs_dim / 64) <= 8), one persistent workgroup of 4 waves each, all co-resident;
//   workgroup (n, k) keeps W1[:, 64k .. 64k+63] of network n (and W2, W3, the biases, log_std: replicated) in LDS, gathers the
//   matching 64 features of the minibatch rows and computes the PARTIAL pre-activation W1_k x_k;
//   the S partials of a network are all-reduced through device memory as reduce-scatter + all-gather of bare floats (NaN
//   sentinel instead of tags, 16-byte buffer loads / stores; a unit's owner sums in slice order and broadcasts, so all S replicas
//   continue from the same bits) -- through the XCD's L2 when a placement census finds the workgroups co-resident (plain
//   stores, sc1 polls), write-through otherwise;
//   then bias + tanh, layers 2 / 3, loss, backward and the weight gradients run replicated -- identical instructions on
//   identical data -- except dW1, of which a workgroup computes (and owns the Adam state of) its own 64 columns;
//   the joint clip_grad_norm_ (ppo_lag.py:325) sums one ||g||^2 granule per workgroup: the slice's share of W1, plus everything
//   else from slice 0 only;  Adam of W1 / b1 at the end of the step, of the rest inside the next step's hand-offs.
//
// Three instantiations of one body: the clipped-surrogate step (ppo_update_ks_kernel: spo_ppo_lag_update_iter_ks), the KL-penalty
// actor loss of FOCOPS / CUP with separate optimiser clocks and an actor-only launch (klpen_update_ks_kernel, AMODE 1:
// spo_update_iter_ex_ks), and the second-order scripts' critic fit (critic_fit_ks_kernel, CFIT: two networks, minibatches of up
// to 128 rows as two 64-column chunks, the actor's stale gradient norm in the joint clip: spo_critic_fit_iter_ks).
//
// The arithmetic per element is that of ppo_update_kernel (same MFMA chaining, loss, Adam); the first layer's dot products are
// summed slice by slice instead of in one chain (rounding-level difference; tests: 1e-5 on the first steps + the fp64 drift
// envelope up to 8 192 steps at 524 288 rows x 376).  Two output tiles for the actor (act_dim <= 32).  One GPU.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "common.h"
#include "mlp_mfma.h"
#include "adam.h"
#include "../../include/safepo_hip.h"

namespace {
using namespace spo;

constexpr int KS_MAX_SLICES = 8;                 // obs_dim <= 512
constexpr int KS_NO = 2;                         // output tiles of the actor: act_dim <= 32
constexpr int KS_OUT = 16 * KS_NO;
constexpr int LDB = 64 + 4;                      // [feature][batch] LDS row stride (floats)

struct KsLds {                                   // floats
  static constexpr int W1 = 0;                   // [64][68]: this slice's 64 columns of W1
  static constexpr int B1 = W1 + HID * LDH;
  static constexpr int W2 = B1 + HID;
  static constexpr int B2 = W2 + HID * LDH;
  static constexpr int W3 = B2 + HID;            // [32][68]
  static constexpr int B3 = W3 + KS_OUT * LDH;
  static constexpr int XT = B3 + KS_OUT;         // [feature][batch] images of the weight-gradient products
  static constexpr int H1T = XT + HID * LDB;
  static constexpr int H2T = H1T + HID * LDB;
  static constexpr int DZ2T = H2T + HID * LDB;
  static constexpr int DZ1T = DZ2T + HID * LDB;
  static constexpr int DOT = DZ1T + HID * LDB;   // [32][68]
  static constexpr int RED = DOT + KS_OUT * LDB;
  static constexpr int SIZE = RED + 256;
};
static_assert(KsLds::SIZE * 4 <= 163840, "160 KB of LDS");
// RED: [0..3] loss partials per wave, [4..7] ||g||^2 of the W1 slice, [8..11] of the rest, [12..15] / [16..19] sum p^2 likewise,
//      [20..23] indicator counts per wave (KL-penalty loss), [32 + 32 wave + a] d(log_std) partials, [160 + a] log_std mirror,
//      [192 + i] polled ||g||^2 granules, [216 + i] sum p^2 granules, [250] placement census

typedef unsigned u4 __attribute__((ext_vector_type(4)));
// partial pre-activations: 16-byte groups of bare floats, [2 parities][3 nets][dst slice]...[256 lanes]
constexpr size_t KS_ZRS_BYTES = (size_t)2 * 3 * KS_MAX_SLICES * KS_MAX_SLICES * 4 * 256 * 16;   // reduce-scatter slots [dst][src][fq]
constexpr size_t KS_ZAG_BYTES = (size_t)2 * 3 * KS_MAX_SLICES * 4 * 256 * 16;                   // all-gather slots [dst][fq]
constexpr size_t KS_ZZERO_OFF = KS_ZRS_BYTES + KS_ZAG_BYTES;      // a row of zeros (never written) and a row nobody reads
constexpr size_t KS_ZDUMP_OFF = KS_ZZERO_OFF + 4096;
constexpr size_t KS_Z_BYTES = KS_ZDUMP_OFF + 4096;

struct KsArgs {
  float* theta; float* adam_m; float* adam_v;
  const float* obs; const float* act; const float* logp_old; const float* tgt_r; const float* tgt_c; const float* adv;
  const int32_t* perm; int64_t M;
  spo_ppo_cfg cfg;
  float* losses;                       // [nsteps][3]
  float* zbuf;                         // partial pre-activations (KS_Z_BYTES, all sentinel at launch)
  unsigned long long* gran;            // [2 parities][2 kinds][3 * KS_MAX_SLICES] {tag, value} granules
  int* err;
  double pow_b1, pow_b2;
  unsigned tag_base;                   // tags of this launch: tag_base + step + 1 (never reused: no clearing between launches)
  int S;
  int n_nets, first_net;               // 3 / 0: a full step; 2 / 0: the critic fit (CFIT instantiation); 1 / 2: actor only (CUP's second stage)
  const float* old_mean; const float* old_std; float kl_bound, pg_coef;   // KL-penalty actor loss (AMODE 1: FOCOPS, CUP)
  double pow_b1_actor, pow_b2_actor;   // the actor's own optimiser clock (spo_update_iter_ex)
  float* stale_io;                     // critic fit: ||actor.grad||^2 that the joint clip still sees and rescales (cpo.py:557), in / out
  int force_safe;                      // SPO_KS_SAFE=1: write-through exchange stores whatever the placement (tests)
  float* flat_grad;                    // GRAD instantiation (data-parallel step): the minibatch's raw gradient, theta's layout
};

__device__ __forceinline__ float pin(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int pin(int v) { asm volatile("" : "+v"(v)); return v; }
// (one launch at a time per device: the exchange scratch below is per device, like the kernel's use by one engine on one stream)
constexpr unsigned KS_SPIN_LIMIT = 1u << 22;

template <int AMODE>
struct KsCol {                         // per-column inputs of one minibatch, prefetched one step ahead (raw loads; settled at pick-up)
  f4 x[4];                             // this slice's observation tiles (B operand of layer 1)
  f4 actv[KS_NO];                      // actor: act[16 t + 4 q ..]
  f4 omv[AMODE == 1 ? KS_NO : 1];      // actor, KL-penalty loss: old_mean[16 t + 4 q ..]
  float t0, t1;                        // critic: target ; actor: logp_old, adv
};

#ifdef SPO_KS_PROF
__device__ unsigned long long g_ks_prof[16];      // development builds: cycles per interval of the step (thread 0 of the LAST workgroup: the actor's last slice)
#define KS_STAMP(i) { if (tid == 0 && wg == a.n_nets * a.S - 1) { const unsigned long long _t = __builtin_readcyclecounter(); pacc[i] += _t - tprev; tprev = _t; } }
#else
#define KS_STAMP(i)
#endif
// FAST: every workgroup of the launch sits on one XCD (checked by the kernel below), stores of the exchange stay plain
// CFIT: the critic fit of the second-order scripts (cpo.py:541-571): two networks, minibatches of up to 128 rows taken as two
// 64-column chunks whose weight gradients accumulate before the one optimiser step, the actor's stale gradient in the joint norm.
// AMODE 1: the KL-penalty actor loss of FOCOPS (focops.py:326-337) and CUP's second stage (cup.py:372-383), as update.hip's AMODE.
// GRAD (round 6): ONE minibatch, forward / loss / backward / weight gradients only -- the raw gradient of the
// three networks (no L2 term, no value coefficient: spo_wide_clip_adam adds them) goes to a.flat_grad in theta's layout and the
// three data losses to a.losses; no norm exchange, no Adam, theta untouched.  The per-minibatch kernel of the data-parallel step at
// these dims: kernel -> all-reduce of the flat gradient -> spo_wide_clip_adam on every rank.
template <bool FAST, bool CFIT, int AMODE, bool GRAD = false>
__device__ __forceinline__ void ks_body(const KsArgs& a, float* const lds) {
  using Col = KsCol<AMODE>;
#ifdef SPO_KS_PROF
  unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
  using L = KsLds;
  const int wg = (int)(blockIdx.x >> 3);
  const int S = a.S, net = a.first_net + wg / S, ks = wg - (wg / S) * S;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, q = lane >> 4;
  const int D = a.cfg.obs_dim, A = a.cfg.act_dim, B = a.cfg.batch;
  const NetGeom g = net_geom(D, A, net);
  const bool is_actor = (net == 2), first = (ks == 0);
  const int OUT = g.OUT, nto = is_actor ? KS_NO : 1;
  const int ls_off = g.off - A;                  // actor only
  const int c_lo = 64 * ks, DS = (D - c_lo) < 64 ? (D - c_lo) : 64;      // this slice's columns [c_lo, c_lo + DS)
  float* const red = lds + L::RED;
  float* const st_m = a.adam_m; float* const st_v = a.adam_v;
  // red[251]: set by the first poll of this workgroup that times out: from then on nobody in the workgroup waits
  // again (one bounded wait per launch, not three per step), the step loop ends at its next top, and the launch leaves theta and
  // the optimiser state untouched -- the caller sees the error word and can fall back to the launch-per-layer path
  volatile float* const dead = red + 251;

  // ---- stage the slice of the network (pads zeroed)
  for (int i = tid; i < L::XT / 4; i += 256) reinterpret_cast<f4*>(lds)[i] = f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  {
    const int c0 = (tid & 15) * 4;
    for (int r = tid >> 4; r < HID; r += 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c0 + e < DS) lds[L::W1 + r * LDH + c0 + e] = a.theta[g.w1() + r * D + c_lo + c0 + e];
        lds[L::W2 + r * LDH + c0 + e] = a.theta[g.w2() + r * HID + c0 + e];
      }
    }
    for (int r = tid >> 4; r < OUT; r += 16)
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[L::W3 + r * LDH + c0 + e] = a.theta[g.w3() + r * HID + c0 + e];
    if (tid < HID) { lds[L::B1 + tid] = a.theta[g.b1() + tid]; lds[L::B2 + tid] = a.theta[g.b2() + tid]; }
    if (tid < OUT) lds[L::B3 + tid] = a.theta[g.b3() + tid];
    if (is_actor && tid < A) red[160 + tid] = a.theta[ls_off + tid];
  }
  __syncthreads();

  // ---- ownership (C layout of the weight-gradient tiles) and optimiser state in registers
  const int orow = 16 * wave + 4 * q;
  f4 mW1[4], vW1[4], mW2[4], vW2[4], mW3[KS_NO], vW3[KS_NO];
  float mls = 0.f, vls = 0.f;                                   // log_std[tid] (threads 0 .. act_dim-1 of the actor)
  float mb1, vb1, mb2, vb2, mb3[KS_NO], vb3[KS_NO];
  const bool own_b = (q == 0);
  const bool own_w0 = (wave == 0 && q == 0);                    // b3[16 t + j], and (j == 0) log_std
  const bool own_ls = is_actor && tid < A;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * nt + j;
      const int i1 = g.w1() + (orow + r) * D + c_lo + i, i2 = g.w2() + (orow + r) * HID + i;
      mW1[nt][r] = i < DS ? a.adam_m[i1] : 0.f; vW1[nt][r] = i < DS ? a.adam_v[i1] : 0.f;
      mW2[nt][r] = a.adam_m[i2]; vW2[nt][r] = a.adam_v[i2];
    }
#pragma unroll
  for (int t = 0; t < KS_NO; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * t + 4 * q + r;
      const int idx = g.w3() + o * HID + 16 * wave + j;
      mW3[t][r] = o < OUT ? a.adam_m[idx] : 0.f; vW3[t][r] = o < OUT ? a.adam_v[idx] : 0.f;
    }
    const int ob = 16 * t + j;
    mb3[t] = ob < OUT ? a.adam_m[g.b3() + ob] : 0.f; vb3[t] = ob < OUT ? a.adam_v[g.b3() + ob] : 0.f;
  }
  if (own_ls) { mls = a.adam_m[ls_off + tid]; vls = a.adam_v[ls_off + tid]; }
  mb1 = a.adam_m[g.b1() + 16 * wave + j]; vb1 = a.adam_v[g.b1() + 16 * wave + j];
  mb2 = a.adam_m[g.b2() + 16 * wave + j]; vb2 = a.adam_v[g.b2() + 16 * wave + j];

  const float b1c = a.cfg.beta1, b2c = a.cfg.beta2, eps = a.cfg.adam_eps;
  double pw1 = is_actor ? a.pow_b1_actor : a.pow_b1, pw2 = is_actor ? a.pow_b2_actor : a.pow_b2;
  const float lr = is_actor ? a.cfg.lr_actor : a.cfg.lr_critic;
  const float l2 = (!is_actor && a.cfg.use_critic_norm) ? a.cfg.l2_coef : 0.f;
  const float l2x2 = 2.f * l2;
  const float vcoef = (net == 0 && a.cfg.use_value_coefficient) ? 2.f : 1.f;
  const float clip_lo = 1.f - a.cfg.clip, clip_hi = 1.f + a.cfg.clip;
  const float* tgt = (net == 0) ? a.tgt_r : a.tgt_c;
  const int64_t nsteps = (a.M + B - 1) / B;
  const int mycol = 16 * wave + j;

  const int H = CFIT ? (B + 63) / 64 : 1;                       // 64-column chunks per minibatch (<= 2)
  const int64_t nchunks = nsteps * H;
  auto perm_pos = [&](int64_t e2) -> int64_t {                   // row of the permutation this lane's column takes in chunk e2
    const int64_t s2 = (CFIT && H == 2) ? (e2 >> 1) : e2;
    const int h2 = (CFIT && H == 2) ? (int)(e2 & 1) : 0;
    const int64_t base = s2 * B;
    const int64_t rem = a.M - base;
    const int nst = (int)(rem < B ? rem : B);
    int nc = nst - 64 * h2; nc = nc < 0 ? 0 : (nc > 64 ? 64 : nc);
    const int64_t pos = base + 64 * h2 + (mycol < nc ? mycol : 0);
    return pos < a.M ? pos : a.M - 1;                             // (an empty second chunk: any valid row, every column masked)
  };
  auto fetch_obs = [&](int64_t smp, Col& cd) {
    {
      // this slice's 64 features of the row (clamped addresses, NO select: see load_obs_tiles_raw); 16-byte loads only when every
      // row AND every slice start is 16-byte aligned (obs_dim a multiple of 4)
      const float* const row = a.obs + smp * D + c_lo;
      if ((D & 3) == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int c = 16 * nt + 4 * q;
          cd.x[nt] = *reinterpret_cast<const f4*>(row + (c < DS ? c : 0));
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = 16 * nt + 4 * q + e;
            cd.x[nt][e] = row[c < DS ? c : 0];
          }
      }
    }
  };
  auto fetch_rest = [&](int64_t smp, Col& cd) {
    if (!is_actor) {
      cd.t0 = tgt[smp]; cd.t1 = 0.f;
#pragma unroll
      for (int t = 0; t < KS_NO; ++t) cd.actv[t] = f4{0.f, 0.f, 0.f, 0.f};
    } else {
      cd.t0 = a.logp_old[smp]; cd.t1 = a.adv[smp];
#pragma unroll
      for (int t = 0; t < KS_NO; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 16 * t + 4 * q + r;
          // (unconditional within an instruction that has any live lane: pads selected at pick-up; rows past act_dim in every
          // lane -- 16 t + r >= act_dim -- are not loaded at all: 5 gathers instead of 8 at act_dim 17)
          cd.actv[t][r] = (16 * t + r < A) ? a.act[smp * A + (ai < A ? ai : 0)] : 0.f;
          if (AMODE == 1) cd.omv[t][r] = (16 * t + r < A) ? a.old_mean[smp * A + (ai < A ? ai : 0)] : 0.f;
        }
    }
  };
  auto settle = [&](Col& cd) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) cd.x[nt][e] = pin(cd.x[nt][e]);
    mask_obs_tiles<64>(DS, q, cd.x);
    cd.t0 = pin(cd.t0);
    if (is_actor) {
      cd.t1 = pin(cd.t1);
#pragma unroll
      for (int t = 0; t < KS_NO; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float av = pin(cd.actv[t][r]);
          cd.actv[t][r] = (16 * t + 4 * q + r) < A ? av : 0.f;
          if (AMODE == 1) {
            const float ov = pin(cd.omv[t][r]);
            cd.omv[t][r] = (16 * t + 4 * q + r) < A ? ov : 0.f;
          }
        }
    }
  };

  const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(a.zbuf, 0, (int)KS_Z_BYTES, 0x00020000);
  const unsigned zvoff = (unsigned)tid * 16u;
  auto zstore = [&](u4 w, unsigned off) {                          // (the cache policy is an immediate)
    if constexpr (FAST) __builtin_amdgcn_raw_buffer_store_b128(w, zrsrc, zvoff, off, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(w, zrsrc, zvoff, off, 16);
  };
  auto gstore = [&](unsigned long long* dst, unsigned long long w) {
    if constexpr (FAST) __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // ---- deferred half of the optimiser step.  W1 / b1 are updated at the end of step s (the next partial layer 1 needs them); the
  // rest -- W2, W3, b2, b3, log_std -- is first read by layer 2 of step s+1, BEHIND the exchange of the partials, so its Adam runs
  // inside the two hand-offs of step s+1 (the store acknowledgements and the peers' latency pass under ~2.3 k cycles of vector work
  // instead of under a spin).  The gradients wait in registers, the parameters are re-read from the LDS image.
  f4 gW2[4], gW3[KS_NO];
  float gb2 = 0.f, gb3[KS_NO], gpb3[KS_NO], gls = 0.f, pcoef = 0.f, pstep = 0.f, pinv = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) gW2[nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KS_NO; ++t) { gW3[t] = f4{0.f, 0.f, 0.f, 0.f}; gb3[t] = 0.f; gpb3[t] = 0.f; }
#define KS_ADAM(DST, P, G, M, V, COEF, SS, IB)                                         \
  {                                                                                    \
    const AdamOut _o = adam1((P), (G) * (COEF), (M), (V), b1c, b2c, eps, (SS), (IB));  \
    (M) = _o.m; (V) = _o.v; (DST) = _o.p;                                              \
  }
  auto adam_w2 = [&]() {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* const w = lds + L::W2 + (orow + r) * LDH + 16 * nt + j;
        KS_ADAM(*w, *w, gW2[nt][r], mW2[nt][r], vW2[nt][r], pcoef, pstep, pinv)
      }
  };
  auto adam_w3_rest = [&]() {
#pragma unroll
    for (int t = 0; t < KS_NO; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {                                    // pad rows stay exactly 0
        float* const w = lds + L::W3 + (16 * t + 4 * q + r) * LDH + 16 * wave + j;
        KS_ADAM(*w, *w, gW3[t][r], mW3[t][r], vW3[t][r], pcoef, pstep, pinv)
      }
      float np3;
      // b3[16 t + j] has a replica in EVERY wave (own moments each): the old value comes along in a register -- read from
      // the image here, a later wave would pick up an earlier wave's update and apply the step twice
      KS_ADAM(np3, gpb3[t], gb3[t], mb3[t], vb3[t], pcoef, pstep, pinv)
      lds[L::B3 + 16 * t + j] = np3;         // (every replica writes the same bits)
    }
    if (own_ls) {
      float nl;
      KS_ADAM(nl, red[160 + tid], gls, mls, vls, pcoef, pstep, pinv)
      red[160 + tid] = nl;
    }
    float np2;
    KS_ADAM(np2, lds[L::B2 + 16 * wave + j], gb2, mb2, vb2, pcoef, pstep, pinv)
    lds[L::B2 + 16 * wave + j] = np2;
  };
  float iso[KS_NO][4];                                            // 1 / sigma_old of my action rows (KL-penalty loss)
#pragma unroll
  for (int t = 0; t < KS_NO; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ai = 16 * t + 4 * q + r;
      iso[t][r] = (AMODE == 1 && is_actor && ai < A) ? 1.f / a.old_std[ai] : 1.f;
    }
  Col nxt;
  int smp1 = 0;
  fetch_obs((int64_t)a.perm[perm_pos(0)], nxt);
  fetch_rest((int64_t)a.perm[perm_pos(0)], nxt);
  if (nchunks > 1) smp1 = a.perm[perm_pos(1)];
  float stale_sq = (CFIT && a.stale_io) ? *a.stale_io : 0.f;
  f4 aW1[4], aW2[4], aW3[KS_NO];                                // weight-gradient accumulators (across the chunks of a minibatch)
  float db1 = 0.f, db2 = 0.f, db3[KS_NO];

  for (int64_t e = 0; e < nchunks; ++e) {
    if (*dead != 0.f) break;                                           // (uniform: written before the barrier that ended the last chunk)
    const int64_t s = (CFIT && H == 2) ? (e >> 1) : e;
    const int h = (CFIT && H == 2) ? (int)(e & 1) : 0;
    const int64_t base = s * B;
    const int64_t rem = a.M - base;
    const int ncols_step = (int)(rem < B ? rem : B);
    int ncols = ncols_step - 64 * h; ncols = ncols < 0 ? 0 : (ncols > 64 ? 64 : ncols);
    const float inv_n = 1.f / (float)ncols_step;
    const bool cv = mycol < ncols;
    const unsigned tag = a.tag_base + (unsigned)s + 1u;           // (granules: one exchange per minibatch step)
    const int par = (int)(e & 1);                                  // (partials: one exchange per chunk)
    const int gpar = (int)(s & 1);

#ifdef SPO_KS_PROF
    if (tid == 0 && wg == a.n_nets * a.S - 1) tprev = __builtin_readcyclecounter();
#endif
    Col cur = nxt;
    settle(cur);
    const int smp_next = pin(smp1);
    const int64_t pos2 = (e + 2 < nchunks) ? perm_pos(e + 2) : 0;
    auto stage_xt = [&]() {                 // x^T image of this slice (B operand of the dW1 product, read after the staging barrier)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[L::XT + (16 * nt + 4 * q + e) * LDB + mycol] = cur.x[nt][e];
    };

    // ---- layer 1: this slice's partial pre-activation, then the sum over the slices (slice order: identical bits everywhere)
    f4 z1[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) z1[mt] = f4{0.f, 0.f, 0.f, 0.f};
    layer_accum<4>(lds + L::W1, LDH, cur.x, z1, j, q);
    KS_STAMP(0)                                                        // settle, partial layer 1
    const unsigned pn = (unsigned)(par * 3 + net) * KS_MAX_SLICES;
    const u4 sentinel = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const unsigned mine = ((pn + (unsigned)ks) * KS_MAX_SLICES * 4u) * 4096u;
    const unsigned ag0 = (unsigned)KS_ZRS_BYTES + (pn * 4u) * 4096u;
    int own[4];
#pragma unroll
    for (int fq = 0; fq < 4; ++fq) own[fq] = ((4 * fq + wave) * S) >> 4;
    if (S == 1) stage_xt();
    if (S > 1) {
      // All-reduce of the 64 x 64 partial pre-activations over the S workgroups of the network, as reduce-scatter + all-gather
      // (a flat all-to-all moves S^2 x 16 KB per network: 5.3 MB a step at S = 6, measured L2-bandwidth bound at 15 k cycles).
      // Unit u = 4 fq + wave (one f4 per lane of one wave, 1 KB) belongs to slice (u S) >> 4; its owner sums the S partials in
      // slice order and broadcasts the sum, so every workgroup continues from identical bits.
      // Transport: bare floats in ORDINARY device memory, buffer loads / stores with sc1 (agent scope: coherent at the L2 the
      // co-resident workgroups share; a hand-off is an L2 round trip, ~0.5 us, where the uncached system-scope words of the
      // cross-GPU protocol cost 3 us).  No tags: a slot holds the sentinel 0xFFFFFFFF (a NaN no arithmetic produces) until its
      // producer fills it; the consumer checks every dword (torn 16-byte reads are harmless) and puts the sentinel back after
      // reading.  One private slot per (destination, source, unit) / (destination, unit), two parities.  Why a slot is always
      // reset before it is refilled -- one lane's traffic: reset at step s -> vmcnt(0) -> its reduce-scatter store of step s+1 to
      // the unit's owner -> the owner's poll -> the owner's broadcast of step s+1 -> the producer's poll of that -> the
      // producer's store of step s+2 into the slot.  The builtins are memory operations the compiler tracks (no inline-asm
      // load whose outputs could be copied while in flight).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the resets of the previous step are acknowledged by the L2
      // Straight-line memory traffic: a branch between two stores makes the compiler put s_waitcnt vmcnt(0) at the join, and every
      // one of those is a full store acknowledgement (measured: 9 k cycles for the two hand-offs).  So every wave always issues the
      // same loads and stores, and the ones that do not apply go to two spare 4 KB rows: ZERO (never written: passes every poll)
      // and DUMP (never read).
#pragma unroll
      for (int fq = 0; fq < 4; ++fq) {
        u4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(z1[fq][e]);
        zstore(w, own[fq] != ks ? (((pn + (unsigned)own[fq]) * KS_MAX_SLICES + (unsigned)ks) * 4u + fq) * 4096u : (unsigned)KS_ZDUMP_OFF);
      }
      KS_STAMP(1)                                                      // reduce-scatter stores
      stage_xt();                                                      // (under the hand-off's latency)
    }
    if (s > 0 && h == 0) adam_w2();                                    // deferred from step s-1, under the first hand-off
    if (S > 1) {
#pragma unroll
      for (int fq = 0; fq < 4; ++fq)
        if (own[fq] == ks) {                                           // (wave-uniform; at most one unit per wave from S = 4 up)
          u4 zw[KS_MAX_SLICES];
          unsigned spins = 0;
          for (;;) {
#pragma unroll
            for (int k = 0; k < KS_MAX_SLICES; ++k)
              zw[k] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, zvoff, (k < S && k != ks) ? mine + (unsigned)(k * 4 + fq) * 4096u : (unsigned)KS_ZZERO_OFF, 16);
            bool ok = true;
#pragma unroll
            for (int k = 0; k < KS_MAX_SLICES; ++k)
#pragma unroll
              for (int e = 0; e < 4; ++e) ok = ok && (zw[k][e] != 0xFFFFFFFFu);
            if (ok) break;
            if (*dead != 0.f) break;                                 // (a poll of this workgroup has timed out: stop waiting)
            if (++spins > KS_SPIN_LIMIT) { *a.err = 1; *dead = 1.f; break; }      // bounded AND sticky: never hang the GPU
            __builtin_amdgcn_s_sleep(1);
          }
#ifdef SPO_KS_PROF
          if (tid == 0 && wg == a.n_nets * a.S - 1) pacc[10] += spins + 1;
#endif
#pragma unroll
          for (int k = 0; k < KS_MAX_SLICES; ++k)
            zstore(sentinel, (k < S && k != ks) ? mine + (unsigned)(k * 4 + fq) * 4096u : (unsigned)KS_ZDUMP_OFF);
          // the S partials in a FIXED PAIRWISE order, ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)) with 0.f for slices past S (x + 0.f
          // is exact): round 6 -- the sequential order of round 5 put up to seven roundings in a row on top of each slice's
          // 64-product MFMA chain; a blocked sgemm (the reference's arithmetic) sums short partials pairwise as well
          f4 acc;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v[KS_MAX_SLICES];
#pragma unroll
            for (int k = 0; k < KS_MAX_SLICES; ++k) {
              const float x = (k == ks) ? z1[fq][e] : __uint_as_float(zw[k][e]);
              v[k] = (k < S) ? x : 0.f;
            }
            acc[e] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
          }
          z1[fq] = acc;
          u4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(acc[e]);
#pragma unroll
          for (int c = 0; c < KS_MAX_SLICES; ++c)
            zstore(w, (c < S && c != ks) ? ag0 + (unsigned)(c * 4 + fq) * 4096u : (unsigned)KS_ZDUMP_OFF);
        }
    }
    if (s > 0 && h == 0) adam_w3_rest();                               // ... and under the second
    if (S > 1) {
      {
        u4 zg[4];
        unsigned spins = 0;
        for (;;) {
#pragma unroll
          for (int fq = 0; fq < 4; ++fq)
            zg[fq] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, zvoff, own[fq] != ks ? ag0 + (unsigned)(ks * 4 + fq) * 4096u : (unsigned)KS_ZZERO_OFF, 16);
          bool ok = true;
#pragma unroll
          for (int fq = 0; fq < 4; ++fq)
#pragma unroll
            for (int e = 0; e < 4; ++e) ok = ok && (zg[fq][e] != 0xFFFFFFFFu);
          if (ok) break;
          if (*dead != 0.f) break;
          if (++spins > KS_SPIN_LIMIT) { *a.err = 1; *dead = 1.f; break; }
          __builtin_amdgcn_s_sleep(1);
        }
#ifdef SPO_KS_PROF
        if (tid == 0 && wg == a.n_nets * a.S - 1) pacc[11] += spins + 1;
#endif
#pragma unroll
        for (int fq = 0; fq < 4; ++fq) {
          zstore(sentinel, own[fq] != ks ? ag0 + (unsigned)(ks * 4 + fq) * 4096u : (unsigned)KS_ZDUMP_OFF);
#pragma unroll
          for (int e = 0; e < 4; ++e) z1[fq][e] = own[fq] != ks ? __uint_as_float(zg[fq][e]) : z1[fq][e];
        }
      }
    }
    if (s > 0 && h == 0) __syncthreads();                              // W2, W3, b2, b3, log_std of step s-1's update are in place
    // prefetch AFTER the polls: loads return in order, so a poll issued behind the gather of the next minibatch would wait for
    // its HBM round trip as well.  In two halves with the layer-2 products in between: fourteen gathers in a row fill the
    // address queue and the wave sits on the issue (2.4 k cycles measured).
    if (e + 1 < nchunks) fetch_obs((int64_t)smp_next, nxt);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
    KS_STAMP(2)                                                        // polls of the partials + sums
    f4 h1[4], h2[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) h1[mt] = fast_tanh4(z1[mt] + *reinterpret_cast<const f4*>(lds + L::B1 + 16 * mt + 4 * q));
    layer_hidden<4, true>(lds + L::W2, LDH, lds + L::B2, h1, h2, j, q);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (e + 1 < nchunks) fetch_rest((int64_t)smp_next, nxt);         // next chunk's scalars and actions, then the index after
    if (e + 2 < nchunks) smp1 = a.perm[pos2];
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
    f4 o[KS_NO];
    o[0] = layer_out(lds + L::W3, lds + L::B3, h2, j, q);
    o[1] = f4{0.f, 0.f, 0.f, 0.f};
    if (nto > 1) o[1] = layer_out(lds + L::W3 + 16 * LDH, lds + L::B3 + 16, h2, j, q);

    KS_STAMP(3)                                                        // tanh, layers 2 / 3
    // ---- loss and d(loss)/d(output), C layout (rows = output unit 16 t + 4 q + r, col = batch)
    f4 dO[KS_NO], dls[KS_NO];
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < KS_NO; ++t) { dO[t] = f4{0.f, 0.f, 0.f, 0.f}; dls[t] = f4{0.f, 0.f, 0.f, 0.f}; }
    if (!is_actor) {
      // mse_loss(critic(obs), target)  (ppo_lag.py:307-309)
      const float diff = o[0][0] - cur.t0;
      const float lm = (q == 0 && cv) ? 1.f : 0.f;
      lsum = lm * diff * diff;
      dO[0][0] = lm * (2.f * diff * inv_n);
    } else {
      float lp = 0.f, dif[KS_NO][4], ivar[KS_NO][4], amask[KS_NO][4];
#pragma unroll
      for (int t = 0; t < KS_NO; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 16 * t + 4 * q + r;
          const bool on = ai < A;
          const float lsv = on ? red[160 + ai] : 0.f;
          const float sdv = __expf(lsv);                              // std = exp(log_std)
          amask[t][r] = on ? 1.f : 0.f;
          ivar[t][r] = __builtin_amdgcn_rcpf(sdv * sdv);
          const float lsd = on ? lsv + LOG_SQRT_2PI : 0.f;
          dif[t][r] = cur.actv[t][r] - o[t][r];                       // pad rows: 0 - 0
          lp += -(dif[t][r] * dif[t][r]) * (0.5f * ivar[t][r]) - lsd;
        }
      lp = quad_row_sum(lp);                                          // .sum(dim=-1)
      const float adv = cur.t1;
      const float ratio = __expf(lp - cur.t0);                        // ppo_lag.py:317
      if (AMODE == 1) {
        // loss = mean_i(ind_i KL_i) - pg_coef mean_i(ind_i) mean_j(ratio_j adv_j): the reference subtracts a [B] tensor from a
        // [B, 1] tensor, its loss is the mean of a B x B matrix = this product of means (update.hip, AMODE)
        float klp = 0.f, dm[KS_NO][4], vrat[KS_NO][4];
#pragma unroll
        for (int t = 0; t < KS_NO; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ai = 16 * t + 4 * q + r;
            const float sdv = __expf(ai < A ? red[160 + ai] : 0.f);
            const float sr = sdv * iso[t][r];                          // kl_normal_normal: var_ratio = (p.scale / q.scale)^2
            vrat[t][r] = sr * sr;
            dm[t][r] = (o[t][r] - cur.omv[t][r]) * iso[t][r];         // (loc_p - loc_q) / scale_q ; pad rows: 0
            klp += amask[t][r] * (0.5f * (vrat[t][r] + dm[t][r] * dm[t][r] - 1.f - logf(vrat[t][r])));
          }
        const float kl = quad_row_sum(klp);                            // .sum(-1, keepdim=True)
        const float ind = (kl <= a.kl_bound) ? 1.f : 0.f;
        const float cnt = wave_sum_lane63((q == 0 && cv) ? ind : 0.f);
        if (lane == 63) red[20 + wave] = cnt;
        __syncthreads();                                               // the actor's workgroups only (block-uniform branch)
        const float frac = ((red[20] + red[21]) + (red[22] + red[23])) * inv_n;
        const float pg = a.pg_coef * frac;
        const float dlp = cv ? -(pg * adv * ratio) * inv_n : 0.f;
        const float wk = cv ? ind * inv_n : 0.f;
        lsum = ((q == 0 && cv) ? 1.f : 0.f) * (pg * ratio * adv - ind * kl);      // loss = -mean(this)
#pragma unroll
        for (int t = 0; t < KS_NO; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float zz = dif[t][r] * ivar[t][r];
            dO[t][r] = fmaf(dlp, zz, wk * dm[t][r] * iso[t][r]);
            dls[t][r] = amask[t][r] * fmaf(dlp, dif[t][r] * zz - 1.f, wk * (vrat[t][r] - 1.f));
          }
      } else {
      const float rc = fminf(fmaxf(ratio, clip_lo), clip_hi);         // torch.clamp
      const float s1 = ratio * adv, s2 = rc * adv;
      const bool inr = (ratio >= clip_lo) && (ratio <= clip_hi);
      float gr;                                                       // backward of torch.min(s1, s2): ties split the gradient
      if (s1 < s2) gr = adv;
      else if (s1 > s2) gr = inr ? adv : 0.f;
      else gr = 0.5f * adv + (inr ? 0.5f * adv : 0.f);
      const float dlp = cv ? -(gr * ratio) * inv_n : 0.f;             // loss_pi = -mean(min(...))
      lsum = ((q == 0 && cv) ? 1.f : 0.f) * fminf(s1, s2);
#pragma unroll
      for (int t = 0; t < KS_NO; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float zz = dif[t][r] * ivar[t][r];
          dO[t][r] = dlp * zz;                                        // pad rows: zz == 0
          dls[t][r] = (dlp * amask[t][r]) * (dif[t][r] * zz - 1.f);
        }
      }
    }

    // ---- backward through the MLP (transposed chaining, weights read as columns)
    f4 dz2[4], dz1[4];
    {
      f4 acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KS_NO; ++t)
        if (t < nto) {
          float w3c[4][4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) w3c[r][mt] = lds[L::W3 + (16 * t + 4 * q + r) * LDH + 16 * mt + j];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma4(w3c[r][mt], dO[t][r], acc[mt]);
        }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dz2[mt][r] = acc[mt][r] * fmaf(-h2[mt][r], h2[mt][r], 1.f);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
      float w2c[2][4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) w2c[0][r][mt] = lds[L::W2 + (4 * q + r) * LDH + 16 * mt + j];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt + 1 < 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) w2c[(nt + 1) & 1][r][mt] = lds[L::W2 + (16 * (nt + 1) + 4 * q + r) * LDH + 16 * mt + j];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma4(w2c[nt & 1][r][mt], dz2[nt][r], acc[mt]);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dz1[mt][r] = acc[mt][r] * fmaf(-h1[mt][r], h1[mt][r], 1.f);
    }

    KS_STAMP(4)                                                        // loss, backward
    // ---- stage [feature][batch] images for the weight-gradient products
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = (16 * mt + 4 * q + r) * LDB + mycol;
        lds[L::H1T + f] = h1[mt][r];
        lds[L::H2T + f] = h2[mt][r];
        lds[L::DZ1T + f] = dz1[mt][r];
        lds[L::DZ2T + f] = dz2[mt][r];
      }
#pragma unroll
    for (int t = 0; t < KS_NO; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[L::DOT + (16 * t + 4 * q + r) * LDB + mycol] = dO[t][r];
    {
      const float ls = wave_sum_lane63(lsum);
      if (lane == 63) red[wave] = (CFIT && h > 0) ? red[wave] + ls : ls;
      if (is_actor) {
#pragma unroll
        for (int t = 0; t < KS_NO; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float tt = row_sum_lane15(dls[t][r]);               // over the wave's 16 columns
            if (j == 15) red[32 + 32 * wave + 16 * t + 4 * q + r] = tt;
          }
      }
    }
    __syncthreads();
    KS_STAMP(5)                                                        // staging + barrier

    // ---- dW[o][i] = sum_b dZ[b][o] * Hprev[b][i]; wave w owns rows 16w .. 16w+15 (W1 slice, W2) / h2 units 16w .. (W3)
    {
      f4 az1[4], az2[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        az1[r4] = *reinterpret_cast<const f4*>(lds + L::DZ1T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
        az2[r4] = *reinterpret_cast<const f4*>(lds + L::DZ2T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
      }
      if (!CFIT || h == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { aW1[nt] = f4{0.f, 0.f, 0.f, 0.f}; aW2[nt] = f4{0.f, 0.f, 0.f, 0.f}; }
      }
      f4 bh[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bh[0][nt] = *reinterpret_cast<const f4*>(lds + L::H1T + (16 * nt + j) * LDB + 4 * q);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        if (r4 + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            bh[(r4 + 1) & 1][nt] = *reinterpret_cast<const f4*>(lds + L::H1T + (16 * nt + j) * LDB + 16 * (r4 + 1) + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) aW2[nt] = mfma4(az2[r4][e], bh[r4 & 1][nt][e], aW2[nt]);
      }
      f4 b3[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) b3[r4] = *reinterpret_cast<const f4*>(lds + L::H2T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
#pragma unroll
      for (int t = 0; t < KS_NO; ++t) {
        if (!CFIT || h == 0) { aW3[t] = f4{0.f, 0.f, 0.f, 0.f}; db3[t] = 0.f; }
        if (t < nto) {
          f4 az3[4];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) az3[r4] = *reinterpret_cast<const f4*>(lds + L::DOT + (16 * t + j) * LDB + 16 * r4 + 4 * q);
          f4 w3a = {0.f, 0.f, 0.f, 0.f}, w3b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r4 = 0; r4 < 4; r4 += 2)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              w3a = mfma4(az3[r4][e], b3[r4][e], w3a);
              w3b = mfma4(az3[r4 + 1][e], b3[r4 + 1][e], w3b);
            }
          if (CFIT && h > 0) aW3[t] = aW3[t] + (w3a + w3b); else aW3[t] = w3a + w3b;
          float rs3 = 0.f;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) rs3 += (az3[r4][0] + az3[r4][1]) + (az3[r4][2] + az3[r4][3]);
          if (CFIT && h > 0) db3[t] += quad_row_sum(rs3); else db3[t] = quad_row_sum(rs3);
        }
      }
      f4 bx[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bx[0][nt] = *reinterpret_cast<const f4*>(lds + L::XT + (16 * nt + j) * LDB + 4 * q);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        if (r4 + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            bx[(r4 + 1) & 1][nt] = *reinterpret_cast<const f4*>(lds + L::XT + (16 * nt + j) * LDB + 16 * (r4 + 1) + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) aW1[nt] = mfma4(az1[r4][e], bx[r4 & 1][nt][e], aW1[nt]);
      }
      float rs1 = 0.f, rs2 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        rs1 += (az1[r4][0] + az1[r4][1]) + (az1[r4][2] + az1[r4][3]);
        rs2 += (az2[r4][0] + az2[r4][1]) + (az2[r4][2] + az2[r4][3]);
      }
      if (CFIT && h > 0) { db1 += quad_row_sum(rs1); db2 += quad_row_sum(rs2); }
      else { db1 = quad_row_sum(rs1); db2 = quad_row_sum(rs2); }
    }
    KS_STAMP(6)                                                        // weight gradients
    if (CFIT && h + 1 < H) { __syncthreads(); continue; }             // (the images are free for the minibatch's second chunk)
    const float loss_data = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
    const float dl = own_ls ? (red[32 + tid] + red[64 + tid]) + (red[96 + tid] + red[128 + tid]) : 0.f;   // d(loss)/d(log_std[tid])
    if constexpr (GRAD) {
      float* const fg = a.flat_grad;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * nt + j;
          if (i < DS) fg[g.w1() + (orow + r) * D + c_lo + i] = aW1[nt][r];           // every slice its own columns of W1
          if (first) fg[g.w2() + (orow + r) * HID + i] = aW2[nt][r];                  // the rest: slice 0 (the slices hold the same bits)
        }
      if (first) {
#pragma unroll
        for (int t = 0; t < KS_NO; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * t + 4 * q + r;
            if (o < OUT) fg[g.w3() + o * HID + 16 * wave + j] = aW3[t][r];
          }
          const int ob = 16 * t + j;
          if (own_w0 && ob < OUT) fg[g.b3() + ob] = db3[t];
        }
        if (own_b) { fg[g.b1() + 16 * wave + j] = db1; fg[g.b2() + 16 * wave + j] = db2; }
        if (own_ls) fg[ls_off + tid] = dl;
        if (tid == 0) a.losses[net] = is_actor ? -loss_data : loss_data;
      }
      return;
    }

    // ---- L2 regulariser of the critics (weights AND biases, ppo_lag.py:310-314), norm shares: the W1 slice / everything else
    float gsq1 = 0.f, psq1 = 0.f, gsqr = 0.f, psqr = 0.f;
    f4 pW1[4], pW2[4], pW3[KS_NO];
    float pb3[KS_NO];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pW1[nt][r] = lds[L::W1 + (orow + r) * LDH + 16 * nt + j];
        pW2[nt][r] = lds[L::W2 + (orow + r) * LDH + 16 * nt + j];
      }
#pragma unroll
    for (int t = 0; t < KS_NO; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pW3[t][r] = lds[L::W3 + (16 * t + 4 * q + r) * LDH + 16 * wave + j];
      }
      pb3[t] = lds[L::B3 + 16 * t + j];
    }
    const float pb1 = lds[L::B1 + 16 * wave + j], pb2 = lds[L::B2 + 16 * wave + j];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p1 = pW1[nt][r], p2 = pW2[nt][r];
        const float g1 = vcoef * fmaf(l2x2, p1, aW1[nt][r]), g2 = vcoef * fmaf(l2x2, p2, aW2[nt][r]);   // pad columns: 0
        aW1[nt][r] = g1; gsq1 = fmaf(g1, g1, gsq1); psq1 = fmaf(p1, p1, psq1);
        aW2[nt][r] = g2; gsqr = fmaf(g2, g2, gsqr); psqr = fmaf(p2, p2, psqr);
      }
#pragma unroll
    for (int t = 0; t < KS_NO; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p3 = pW3[t][r];
        const float g3 = vcoef * fmaf(l2x2, p3, aW3[t][r]);             // pad rows: 0
        aW3[t][r] = g3; gsqr = fmaf(g3, g3, gsqr); psqr = fmaf(p3, p3, psqr);
      }
    {
      // biases are replicated across lanes; each replica runs the same Adam, only one of them counts towards the norms
      db1 = vcoef * fmaf(l2x2, pb1, db1); db2 = vcoef * fmaf(l2x2, pb2, db2);
      const float wb = own_b ? 1.f : 0.f, w0 = own_w0 ? 1.f : 0.f;
      gsqr = fmaf(dl, dl, gsqr);
      gsqr += wb * (db1 * db1 + db2 * db2);
      psqr += wb * (pb1 * pb1 + pb2 * pb2);
#pragma unroll
      for (int t = 0; t < KS_NO; ++t) {
        db3[t] = vcoef * fmaf(l2x2, pb3[t], db3[t]);
        gsqr += w0 * (db3[t] * db3[t]);
        psqr += w0 * (pb3[t] * pb3[t]);
      }
    }
    gsq1 = wave_sum_lane63(gsq1); psq1 = wave_sum_lane63(psq1);
    gsqr = wave_sum_lane63(gsqr); psqr = wave_sum_lane63(psqr);
    if (lane == 63) { red[4 + wave] = gsq1; red[8 + wave] = gsqr; red[12 + wave] = psq1; red[16 + wave] = psqr; }
    __syncthreads();

    KS_STAMP(7)                                                        // L2 terms, norm shares, barrier
    // ---- joint clip_grad_norm_ over all networks (ppo_lag.py:325): one granule per workgroup
    unsigned long long* const gg = a.gran + (size_t)gpar * 2 * 3 * KS_MAX_SLICES;
    unsigned long long* const gp = gg + 3 * KS_MAX_SLICES;
    if (tid == 0) {
      const float gs = ((red[4] + red[5]) + (red[6] + red[7])) + (first ? ((red[8] + red[9]) + (red[10] + red[11])) : 0.f);
      const float ps = ((red[12] + red[13]) + (red[14] + red[15])) + (first ? ((red[16] + red[17]) + (red[18] + red[19])) : 0.f);
      gstore(gg + net * KS_MAX_SLICES + ks, ((unsigned long long)tag << 32) | __float_as_uint(gs));
      gstore(gp + net * KS_MAX_SLICES + ks, ((unsigned long long)tag << 32) | __float_as_uint(ps));
    }
    pw1 *= (double)b1c; pw2 *= (double)b2c;
    float step_size, inv_bc2s;
    adam_scalars(lr, pw1, pw2, step_size, inv_bc2s);
    if (tid < 2 * 3 * KS_MAX_SLICES) {
      const int kind = tid / (3 * KS_MAX_SLICES), idx = tid - kind * 3 * KS_MAX_SLICES;
      const int n2 = idx / KS_MAX_SLICES, k2 = idx - n2 * KS_MAX_SLICES;
      float val = 0.f;
      if (k2 < S && n2 >= a.first_net && n2 < a.first_net + a.n_nets) {
        unsigned long long* const src = (kind ? gp : gg) + idx;
        unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((unsigned)(v >> 32) != tag) {
          if (*dead != 0.f) break;
          if (++spins > KS_SPIN_LIMIT) { *a.err = 1; *dead = 1.f; break; }         // bounded AND sticky: never hang the GPU
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        val = __uint_as_float((unsigned)v);
      }
      red[192 + tid] = val;                                            // [192 + 24 kind + 8 net + slice]
    }
    __syncthreads();
    KS_STAMP(8)                                                        // granule exchange
    float total_sq = CFIT ? stale_sq : 0.f;
    for (int i = 0; i < 3 * KS_MAX_SLICES; ++i) total_sq += red[192 + i];   // fixed order: identical in every workgroup
    const float norm = sqrtf(total_sq);
    float coef = a.cfg.max_grad_norm / (norm + 1e-6f);                // clip_grad_norm_ (torch): eps 1e-6
    coef = coef > 1.f ? 1.f : coef;
    if (CFIT) stale_sq *= coef * coef;                                // the stale actor gradient is rescaled in place too
    if (tid == 0 && first) {
      float pp = 0.f;
      for (int k2 = 0; k2 < KS_MAX_SLICES; ++k2) pp += red[192 + 3 * KS_MAX_SLICES + net * KS_MAX_SLICES + k2];
      a.losses[s * 3 + net] = is_actor ? -loss_data : loss_data + l2 * pp;
    }

    // ---- Adam (torch.optim.Adam, ppo_lag.py:104-117), parameters back into the LDS image: layer 1 now, the rest deferred (above)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)                                      // pad columns stay exactly 0
        KS_ADAM(lds[L::W1 + (orow + r) * LDH + 16 * nt + j], pW1[nt][r], aW1[nt][r], mW1[nt][r], vW1[nt][r], coef, step_size, inv_bc2s)
    {
      float np1;
      KS_ADAM(np1, pb1, db1, mb1, vb1, coef, step_size, inv_bc2s)
      lds[L::B1 + 16 * wave + j] = np1;
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) gW2[nt] = aW2[nt];
#pragma unroll
    for (int t = 0; t < KS_NO; ++t) { gW3[t] = aW3[t]; gb3[t] = db3[t]; gpb3[t] = pb3[t]; }
    gb2 = db2; gls = dl; pcoef = coef; pstep = step_size; pinv = inv_bc2s;
    __syncthreads();
    KS_STAMP(9)                                                        // Adam + barrier
  }
  __syncthreads();
  const bool timed_out = (*dead != 0.f);                               // (uniform over the workgroup)
  if (timed_out) return;                                               // no write-back: theta, adam_m, adam_v, *stale_io as they were
  if (nsteps > 0) { adam_w2(); adam_w3_rest(); __syncthreads(); }     // the last step's deferred half
  if (CFIT && a.stale_io && tid == 0 && wg == 0) *a.stale_io = stale_sq;
#undef KS_ADAM
#ifdef SPO_KS_PROF
  if (tid == 0 && wg == a.n_nets * a.S - 1)
    for (int i = 0; i < 16; ++i) g_ks_prof[i] = i == 15 ? (unsigned long long)FAST : pacc[i];
#endif

  // ---- write back: every workgroup its W1 columns, slice 0 everything else (flat reference order)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * nt + j;
      if (i < DS) {
        const int idx = g.w1() + (orow + r) * D + c_lo + i;
        a.theta[idx] = lds[L::W1 + (orow + r) * LDH + i];
        st_m[idx] = mW1[nt][r]; st_v[idx] = vW1[nt][r];
      }
      if (first) {
        const int idx = g.w2() + (orow + r) * HID + i;
        a.theta[idx] = lds[L::W2 + (orow + r) * LDH + i];
        st_m[idx] = mW2[nt][r]; st_v[idx] = vW2[nt][r];
      }
    }
  if (first) {
#pragma unroll
    for (int t = 0; t < KS_NO; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * t + 4 * q + r;
        if (o < OUT) {
          const int idx = g.w3() + o * HID + 16 * wave + j;
          a.theta[idx] = lds[L::W3 + o * LDH + 16 * wave + j];
          st_m[idx] = mW3[t][r]; st_v[idx] = vW3[t][r];
        }
      }
      const int ob = 16 * t + j;
      if (own_w0 && ob < OUT) {
        a.theta[g.b3() + ob] = lds[L::B3 + ob];
        st_m[g.b3() + ob] = mb3[t]; st_v[g.b3() + ob] = vb3[t];
      }
    }
    if (own_ls) { a.theta[ls_off + tid] = red[160 + tid]; st_m[ls_off + tid] = mls; st_v[ls_off + tid] = vls; }
    if (own_b) {
      const int ob = 16 * wave + j;
      a.theta[g.b1() + ob] = lds[L::B1 + ob]; st_m[g.b1() + ob] = mb1; st_v[g.b1() + ob] = vb1;
      a.theta[g.b2() + ob] = lds[L::B2 + ob]; st_m[g.b2() + ob] = mb2; st_v[g.b2() + ob] = vb2;
    }
  }
}

template <bool CFIT, int AMODE, bool GRAD = false>
__device__ __forceinline__ void ks_entry(const KsArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const red = lds + KsLds::RED;
  const int wg = (int)(blockIdx.x >> 3), tid = threadIdx.x, S = a.S;
  // ---- placement census: do all workgroups of the launch share one XCD (one L2)?  Observed: block b runs on XCD b % 8, which is
  // what the grid of 8-block strides asks for -- but nothing promises it, so it is checked, once per launch, over words every
  // placement delivers (sc1 stores and loads).  Co-resident: stores of the exchange stay plain (the line stays in the shared L2,
  // the sc1 poll is an L2 hit: a hand-off ~0.5 us); otherwise they are sc1 write-through (2.4 us a hand-off measured).
  bool fast;
  {
    unsigned long long* const xid = a.gran + 2 * 2 * 3 * KS_MAX_SLICES;
    const unsigned myx = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID
    if (tid == 0) {
      red[250] = 0.f; red[251] = 0.f;
      __hip_atomic_store(xid + wg, ((unsigned long long)a.tag_base << 32) | myx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < a.n_nets * S) {
      unsigned long long v = __hip_atomic_load(xid + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while ((unsigned)(v >> 32) != a.tag_base) {
        if (++spins > KS_SPIN_LIMIT) { *a.err = 1; break; }
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(xid + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((unsigned)v != myx) red[250] = 1.f;
    }
    __syncthreads();
    fast = (red[250] == 0.f) && !a.force_safe;
    __syncthreads();
  }
  if (fast) ks_body<true, CFIT, AMODE, GRAD>(a, lds);
  else ks_body<false, CFIT, AMODE, GRAD>(a, lds);
}

__global__ __launch_bounds__(256, 1) void ppo_update_ks_kernel(KsArgs a) {
  if (blockIdx.x & 7) return;                    // placement hint (update.hip): the working blocks land on one XCD and share its L2
  ks_entry<false, 0>(a);
}
__global__ __launch_bounds__(256, 1) void klpen_update_ks_kernel(KsArgs a) {
  if (blockIdx.x & 7) return;
  ks_entry<false, 1>(a);
}
__global__ __launch_bounds__(256, 1) void critic_fit_ks_kernel(KsArgs a) {
  if (blockIdx.x & 7) return;
  ks_entry<true, 0>(a);
}
__global__ __launch_bounds__(256, 1) void ppo_grad_ks_kernel(KsArgs a) {
  if (blockIdx.x & 7) return;
  ks_entry<false, 0, true>(a);
}

// Exchange scratch of the kernel: the partial pre-activation words and the norm granules, ordinary device memory (agent-scope
// atomics).  One block per device, allocated on first use, zeroed once (tags never repeat).
struct KsScratch { float* z; unsigned long long* gran; };
constexpr size_t KS_G_BYTES = (size_t)(2 * 2 * 3 * KS_MAX_SLICES + 3 * KS_MAX_SLICES) * 8;   // granules + the placement census
// One block per (device, stream), like the update kernels' scratch in update.hip: launches on one stream are ordered and share
// it, two engines on two streams get two blocks and may run concurrently.  Released by spo_update_scratch_release.
struct KsEntry { int dev; void* stream; char* base; unsigned tag; bool primed; };   // primed: the slots have been set to the sentinel once
constexpr int KS_SCRATCH_MAX = 16;
KsEntry g_ks[KS_SCRATCH_MAX] = {};
int g_ks_n = 0;
std::mutex g_ks_mu;

int ks_scratch(hipStream_t st, KsScratch* out, unsigned* tag_base, unsigned nsteps, bool** primed = nullptr) {
  const int dev = current_device_slot();
  std::lock_guard<std::mutex> lk(g_ks_mu);
  KsEntry* e = nullptr;
  for (int i = 0; i < g_ks_n; ++i)
    if (g_ks[i].dev == dev && g_ks[i].stream == (void*)st) e = &g_ks[i];
  if (!e) {
    if (g_ks_n == KS_SCRATCH_MAX)
      return spo::fail(-1, "feature-split update kernel: more than %d (device, stream) pairs hold exchange scratch in this process; "
                           "call spo_update_scratch_release(stream) for streams that are gone", KS_SCRATCH_MAX);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return spo::fail(-1, "feature-split update kernel: first launch on a stream under capture (the scratch block is allocated on "
                           "first use: launch once outside the capture)");
    void* p = nullptr;
    if (int rc = spo::hip_check(hipMalloc(&p, KS_Z_BYTES + KS_G_BYTES), "hipMalloc(ks scratch)")) return rc;
    if (int rc = spo::hip_check(hipMemset(p, 0, KS_Z_BYTES + KS_G_BYTES), "hipMemset(ks scratch)")) { (void)hipFree(p); return rc; }
    if (int rc = spo::hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize(ks scratch)")) { (void)hipFree(p); return rc; }
    g_ks[g_ks_n] = KsEntry{dev, (void*)st, static_cast<char*>(p), 16u, false};
    e = &g_ks[g_ks_n++];
  }
  out->z = reinterpret_cast<float*>(e->base);
  out->gran = reinterpret_cast<unsigned long long*>(e->base + KS_Z_BYTES);
  *tag_base = e->tag;
  if (primed) *primed = &e->primed;
  e->tag += nsteps + 2u;                           // (wraps after 4e9 steps: a tag then meets words 2^32 steps old)
  return 0;
}

}  // namespace

// (called by spo_update_scratch_release, update.hip)
int spo::ks_scratch_release(int dev, void* stream_or_null, int all) {
  std::lock_guard<std::mutex> lk(g_ks_mu);
  int freed = 0;
  for (int i = 0; i < g_ks_n;) {
    if (g_ks[i].dev == dev && (all || g_ks[i].stream == stream_or_null)) {
      (void)spo::hip_check(hipFree(g_ks[i].base), "hipFree(ks scratch)");
      g_ks[i] = g_ks[--g_ks_n];
      ++freed;
    } else ++i;
  }
  return freed;
}

#ifdef SPO_KS_PROF
extern "C" int spo_debug_ks_profile(unsigned long long* out16_host) {
  return spo::hip_check(hipMemcpyFromSymbol(out16_host, HIP_SYMBOL(g_ks_prof), 128), "hipMemcpyFromSymbol");
}
#endif

extern "C" int spo_ks_supported(int obs_dim, int act_dim, int batch) {
  return (obs_dim >= 1 && obs_dim <= 64 * KS_MAX_SLICES && act_dim >= 1 && act_dim <= KS_OUT && batch >= 1 && batch <= 64) ? 1 : 0;
}

// kind: 0 = clipped-surrogate step, 1 = critic fit, 2 = KL-penalty actor loss, 3 = one minibatch's gradient only; the caller sets n_nets / first_net
static int ks_launch(KsArgs& a, const spo_ppo_cfg* cfg_host, int64_t adam_step_host, int64_t adam_step_actor_host, int64_t M,
                     void* sync_ws, hipStream_t st, int kind) {
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws, 0, 64, st), "hipMemsetAsync(sync_ws)")) return rc;
  a.cfg = *cfg_host; a.M = M;
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.pow_b1_actor = pow((double)cfg_host->beta1, (double)adam_step_actor_host);
  a.pow_b2_actor = pow((double)cfg_host->beta2, (double)adam_step_actor_host);
  a.S = (cfg_host->obs_dim + 63) / 64;
  { const char* e = getenv("SPO_KS_SAFE"); a.force_safe = (e && *e && *e != '0') ? 1 : 0; }
  const int64_t nsteps = (M + cfg_host->batch - 1) / cfg_host->batch;
  SPO_REQUIRE(nsteps < (1ll << 30), "update_iter_ks: too many minibatch steps in one launch");
  KsScratch sc;
  bool* primed = nullptr;
  if (int rc = ks_scratch(st, &sc, &a.tag_base, (unsigned)nsteps, &primed)) return rc;
  a.zbuf = sc.z; a.gran = sc.gran;
  // every slot starts as the sentinel (a launch leaves them that way unless it stopped on an error: cheap enough to not care for the
  // one launch per learning iteration of the persistent forms; the per-minibatch gradient launch -- kind 3 -- relies on the consumers'
  // resets after the block's first launch: 25 MB of memset per minibatch step would cost more than the kernel)
  if (kind != 3 || !*primed) {
    if (int rc = spo::hip_check(hipMemsetAsync(sc.z, 0xFF, KS_ZZERO_OFF, st), "hipMemsetAsync(ks partials)")) return rc;
    *primed = true;
  }
  const size_t sh = KsLds::SIZE * sizeof(float);
  static bool attr_done[SPO_MAX_DEVICES][4] = {};
  const int dslot = current_device_slot();
  if (!attr_done[dslot][kind]) {
    const void* fn = kind == 1 ? reinterpret_cast<const void*>(&critic_fit_ks_kernel)
                   : kind == 3 ? reinterpret_cast<const void*>(&ppo_grad_ks_kernel)
                   : kind == 2 ? reinterpret_cast<const void*>(&klpen_update_ks_kernel)
                               : reinterpret_cast<const void*>(&ppo_update_ks_kernel);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(update_ks)");
    attr_done[dslot][kind] = true;
  }
  const dim3 grid(8 * (a.n_nets * a.S - 1) + 1);
  if (kind == 3) hipLaunchKernelGGL(ppo_grad_ks_kernel, grid, dim3(256), sh, st, a);
  else if (kind == 1) hipLaunchKernelGGL(critic_fit_ks_kernel, grid, dim3(256), sh, st, a);
  else if (kind == 2) hipLaunchKernelGGL(klpen_update_ks_kernel, grid, dim3(256), sh, st, a);
  else hipLaunchKernelGGL(ppo_update_ks_kernel, grid, dim3(256), sh, st, a);
  return 0;
}

extern "C" int spo_ppo_lag_update_iter_ks(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs,
                                          const float* act, const float* logp_old, const float* target_r, const float* target_c,
                                          const float* adv, const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host,
                                          float* losses_out, void* sync_ws, void* stream) {
  SPO_REQUIRE(cfg_host, "update_iter_ks: cfg is NULL");
  SPO_REQUIRE(spo_ks_supported(cfg_host->obs_dim, cfg_host->act_dim, cfg_host->batch),
              "update_iter_ks: obs_dim %d / act_dim %d / batch %d outside [1,%d] / [1,%d] / [1,64]", cfg_host->obs_dim,
              cfg_host->act_dim, cfg_host->batch, 64 * KS_MAX_SLICES, KS_OUT);
  SPO_REQUIRE(theta && adam_m && adam_v && obs && act && logp_old && target_r && target_c && adv && perm && losses_out && sync_ws,
              "update_iter_ks: null pointer");
  SPO_REQUIRE(M > 0 && adam_step_host >= 0, "update_iter_ks: bad sizes");
  KsArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = perm; a.losses = losses_out; a.n_nets = 3; a.first_net = 0;
  if (int rc = ks_launch(a, cfg_host, adam_step_host, adam_step_host, M, sync_ws, (hipStream_t)stream, 0)) return rc;
  SPO_LAUNCH_CHECK("spo_ppo_lag_update_iter_ks");
  return 0;
}

// Data-parallel step at the feature-split kernels' dims (round 6): the RAW gradient of one minibatch (rows
// idx[0 .. n), n <= 64) of all three networks into flat_grad (theta's layout; log_std included) and the three data losses into
// losses3 -- what spo_mlp_forward / spo_wide_ppo_loss / spo_mlp_backward produce on the launch-per-layer path, in ONE launch of
// 3 x ceil(obs_dim / 64) workgroups.  The caller all-reduces flat_grad and applies spo_wide_clip_adam (L2 terms, joint clip, Adam).
extern "C" int spo_ppo_lag_grad_ks(const float* theta, const float* obs, const float* act, const float* logp_old,
                                   const float* target_r, const float* target_c, const float* adv, const int32_t* idx, int n,
                                   const spo_ppo_cfg* cfg_host, float* flat_grad, float* losses3, void* sync_ws, void* stream) {
  SPO_REQUIRE(cfg_host, "ppo_lag_grad_ks: cfg is NULL");
  SPO_REQUIRE(spo_ks_supported(cfg_host->obs_dim, cfg_host->act_dim, cfg_host->batch) && n >= 1 && n <= 64,
              "ppo_lag_grad_ks: obs_dim %d / act_dim %d / rows %d outside [1,%d] / [1,%d] / [1,64]", cfg_host->obs_dim,
              cfg_host->act_dim, n, 64 * KS_MAX_SLICES, KS_OUT);
  SPO_REQUIRE(theta && obs && act && logp_old && target_r && target_c && adv && idx && flat_grad && losses3 && sync_ws,
              "ppo_lag_grad_ks: null pointer");
  KsArgs a{};
  a.theta = const_cast<float*>(theta); a.adam_m = const_cast<float*>(theta); a.adam_v = const_cast<float*>(theta);   // (moments: loaded, never used or stored)
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = idx; a.losses = losses3; a.n_nets = 3; a.first_net = 0; a.flat_grad = flat_grad;
  spo_ppo_cfg c = *cfg_host;
  c.batch = n;                                        // one minibatch of exactly the rows given
  if (int rc = ks_launch(a, &c, 0, 0, n, sync_ws, (hipStream_t)stream, 3)) return rc;
  SPO_LAUNCH_CHECK("spo_ppo_lag_grad_ks");
  return 0;
}

extern "C" int spo_update_iter_ex_ks(float* theta, float* adam_m, float* adam_v, int64_t adam_step_critics_host,
                                     int64_t adam_step_actor_host, const float* obs, const float* act, const float* logp_old,
                                     const float* target_r, const float* target_c, const float* adv, const int32_t* perm, int64_t M,
                                     const spo_ppo_cfg* cfg_host, int actor_loss, const float* old_mean, const float* old_std,
                                     float kl_bound, float pg_coef, int actor_only, float* losses_out, void* sync_ws, void* stream) {
  SPO_REQUIRE(cfg_host, "update_iter_ex_ks: cfg is NULL");
  SPO_REQUIRE(spo_ks_supported(cfg_host->obs_dim, cfg_host->act_dim, cfg_host->batch),
              "update_iter_ex_ks: obs_dim %d / act_dim %d / batch %d outside [1,%d] / [1,%d] / [1,64]", cfg_host->obs_dim,
              cfg_host->act_dim, cfg_host->batch, 64 * KS_MAX_SLICES, KS_OUT);
  SPO_REQUIRE(theta && adam_m && adam_v && obs && act && logp_old && adv && perm && losses_out && sync_ws,
              "update_iter_ex_ks: null pointer");
  SPO_REQUIRE(actor_loss == SPO_ACTOR_LOSS_CLIP || actor_loss == SPO_ACTOR_LOSS_KL_PENALTY,
              "update_iter_ex_ks: unknown actor_loss %d", actor_loss);
  SPO_REQUIRE(actor_loss == SPO_ACTOR_LOSS_CLIP || (old_mean && old_std), "update_iter_ex_ks: old distribution is NULL");
  SPO_REQUIRE(actor_only || (target_r && target_c), "update_iter_ex_ks: critic targets are NULL");
  SPO_REQUIRE(M > 0 && adam_step_critics_host >= 0 && adam_step_actor_host >= 0, "update_iter_ex_ks: bad sizes");
  KsArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = perm; a.losses = losses_out;
  a.old_mean = old_mean; a.old_std = old_std; a.kl_bound = kl_bound; a.pg_coef = pg_coef;
  a.n_nets = actor_only ? 1 : 3; a.first_net = actor_only ? 2 : 0;
  if (int rc = ks_launch(a, cfg_host, adam_step_critics_host, adam_step_actor_host, M, sync_ws, (hipStream_t)stream,
                         actor_loss == SPO_ACTOR_LOSS_KL_PENALTY ? 2 : 0)) return rc;
  SPO_LAUNCH_CHECK("spo_update_iter_ex_ks");
  return 0;
}

extern "C" int spo_critic_fit_ks_supported(int obs_dim, int batch) {
  return (obs_dim >= 1 && obs_dim <= 64 * KS_MAX_SLICES && batch >= 1 && batch <= 128) ? 1 : 0;
}

extern "C" int spo_critic_fit_iter_ks(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs,
                                      const float* target_r, const float* target_c, const int32_t* perm, int64_t M,
                                      const spo_ppo_cfg* cfg_host, float* stale_sq_io, float* losses_out, void* sync_ws,
                                      void* stream) {
  SPO_REQUIRE(cfg_host, "critic_fit_iter_ks: cfg is NULL");
  SPO_REQUIRE(spo_critic_fit_ks_supported(cfg_host->obs_dim, cfg_host->batch),
              "critic_fit_iter_ks: obs_dim %d / batch %d outside [1,%d] / [1,128]", cfg_host->obs_dim, cfg_host->batch,
              64 * KS_MAX_SLICES);
  SPO_REQUIRE(cfg_host->act_dim >= 1, "critic_fit_iter_ks: act_dim %d (the flat parameter layout needs the actor's size)",
              cfg_host->act_dim);
  SPO_REQUIRE(theta && adam_m && adam_v && obs && target_r && target_c && perm && losses_out && sync_ws,
              "critic_fit_iter_ks: null pointer");
  SPO_REQUIRE(M > 0 && adam_step_host >= 0, "critic_fit_iter_ks: bad sizes");
  KsArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.tgt_r = target_r; a.tgt_c = target_c;
  a.perm = perm; a.losses = losses_out; a.stale_io = stale_sq_io; a.n_nets = 2; a.first_net = 0;
  if (int rc = ks_launch(a, cfg_host, adam_step_host, adam_step_host, M, sync_ws, (hipStream_t)stream, 1)) return rc;
  SPO_LAUNCH_CHECK("spo_critic_fit_iter_ks");
  return 0;
}
