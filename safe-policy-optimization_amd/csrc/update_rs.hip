// Row-split persistent minibatch update (round 6), gfx950.
//
// The persistent kernels of update.hip keep one network on ONE CU for a whole learning iteration (ppo_lag.py:297-336, cpo.py:534-571):
// every cycle of the step -- forward, backward, weight gradients, Adam -- is issued by that CU's four SIMDs, and the two waves of a
// SIMD (matrix stream, optimiser stream) ADD (tools/probes/mfma_valu_overlap.hip).  Here the ROWS of a minibatch are divided over
// R workgroups per network (R = 2 for 64-row minibatches, R = 4 for the 128-row critic fit), all on one XCD:
//
//   grid = n_nets networks x R row groups, one persistent workgroup of 6 waves each, all co-resident;
//   workgroup (n, h) keeps a full replica of network n in LDS and takes rows [32 h, 32 h + 32) of every minibatch;
//   waves 2, 3 -- alone on their SIMDs (waves i and i + 4 of a workgroup share a SIMD) -- are the COLUMN waves: 16 batch columns
//   each through forward, loss and backward (transposed MFMA chaining, mlp_mfma.h), nothing else on their SIMD;
//   waves 0, 1, 4, 5 (two per SIMD on the other two SIMDs) are the OPTIMISER waves: the weight-gradient products over the
//   workgroup's 32 rows straight into the accumulator layout that owns the Adam moments (no staging of gradients), the exchange,
//   L2 terms, the joint clip and Adam;
//   the R partial gradients of a network are ALL-REDUCED IN ONE HAND-OFF through the XCD's L2: every optimiser lane stores its
//   16-byte groups into a private slot of each peer (plain stores when a placement census finds the workgroups on one XCD,
//   write-through otherwise; NaN sentinel instead of tags, slots reset by their consumer, two parities: update_ks.hip) and adds
//   own + peer (R = 2: commutative; R = 4: (own + h^1) + (h^2 + h^3), the same tree in every workgroup), so the R replicas continue
//   from identical bits and run the identical Adam -- no second hand-off to return parameters;
//   the joint clip_grad_norm_ (ppo_lag.py:325) sums tagged ||g||^2 granules, one per optimiser wave, of the workgroups with the
//   same row-group index; layer 2 / 3 gradients travel while the column waves are still in the backward pass, only layer 1's
//   exchange is exposed.
//
// Per step and row group (b = workgroup barrier; both roles run the same sequence):
//   column : x^T | b1 | L1, h1^T | b2 | L2, h2^T | b3 | L3, loss, dO, dZ2, images | b4 | gather s+1, dZ1, image | b5 | settle s+1
//   optimis: ...W1 | b1 | Adam W2 | b2 | Adam W3, b3, log_std | b3 | loss log | b4 | dW3, dW2 -> peers | b5 | dW1 -> peers, poll,
//            sum, L2 terms, norm share -> granule, poll norms, clip coefficient, Adam W1 ...
//
// Arithmetic per element is that of ppo_update_kernel (same MFMA chaining, loss, Adam); the batch sum of a weight gradient is
// formed as (rows 0..31) + (rows 32..63) instead of one chain (rounding-level difference: tests at 1e-5 on the first steps and the
// fp64 drift envelopes at full size).  One GPU; clipped-surrogate actor loss; obs_dim <= 64, act_dim <= 16, batch <= 32 R.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include "common.h"
#include "mlp_mfma.h"
#include "adam.h"
#include "../../include/safepo_hip.h"
#include "update_rs.h"

namespace {
using namespace spo;

#ifndef SPO_RS_XR_TWO_PHASE_MIN
#define SPO_RS_XR_TWO_PHASE_MIN 4     // data-parallel form: all-to-all of every word below this world size, reduce-scatter + all-gather from it on (A/B knob)
#endif
constexpr int RS_NS = 16;                        // exchange slots (16-byte groups per lane) per (destination, source): NT1 + 8 <= 12 used
constexpr int RS_MAX_R = 4;
constexpr unsigned RS_SPIN_LIMIT = 1u << 22;
constexpr int RS_NPHASE = 12;

template <int KIN, int NCT>
struct RsLds {                                   // floats; NCT = 16-column tiles per workgroup (2: 32 rows of a minibatch, 1: 16)
  using L = NetLds<KIN>;
  static constexpr int LDC = 16 * NCT + 4;       // [feature][batch] LDS row stride
  static constexpr int LS = L::SIZE;             // log_std mirror (16), then 1 / sigma^2 (16) and log_std + log sqrt(2 pi) (16) per action
  static constexpr int XT = LS + 48;             // two x^T images (double-buffered across steps)
  static constexpr int H1T = XT + 2 * KIN * LDC;
  static constexpr int H2T = H1T + HID * LDC;
  static constexpr int DZ2T = H2T + HID * LDC;
  static constexpr int DZ1T = DZ2T + HID * LDC;
  static constexpr int DOT = DZ1T + HID * LDC;
  static constexpr int RED = DOT + OUTP * LDC;
  static constexpr int SIZE = RED + 128;
  static_assert(SIZE * 4 <= 163840, "160 KB of LDS");
};
// RED: [0..1] loss partials of the column waves, [16 + 16 c + a] d(log_std) partials of column wave c, [88 + w] sum p^2 of
//      optimiser wave w, [96] placement census, [97] dead flag (a poll timed out: stop waiting), [98] (int) index of the step
//      whose L1 the column waves must repeat (its predecessor was clipped), [100..115] the ranks' exchange-region pointers (data-parallel form)

typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct RsArgs {
  float* theta; float* adam_m; float* adam_v;
  const float* obs; const float* act; const float* logp_old; const float* tgt_r; const float* tgt_c; const float* adv;
  const int32_t* perm; int64_t M;
  spo_ppo_cfg cfg;
  float* losses;                       // [nsteps][3]
  float* zbuf;                         // exchange slots (all sentinel at launch)
  unsigned long long* gran;            // [2 parities][RS_MAX_R row groups][3 nets][4 waves] {tag, value} + the placement census
  int* err;
  double pow_b1, pow_b2;
  unsigned tag_base;                   // tags of this launch: tag_base + step + 1 (never reused: no clearing between launches)
  int n_nets, first_net;               // 3 / 0: a PPO-Lagrangian step; 2 / 0: the critic fit
  float* stale_io;                     // critic fit: ||actor.grad||^2 that the joint clip still sees and rescales (cpo.py:557), in / out
  int force_safe;                      // SPO_RS_SAFE=1: write-through exchange stores whatever the placement (tests)
  unsigned long long* prof;            // optional [3][RS_NPHASE] cycle accumulators (instrumented instantiation)
  int prof_wg;                         // ... of this workgroup (SPO_RS_PROF_WG; default: the last one = the actor's last row group)
  // data-parallel form (XW > 0 instantiations): the ranks' exchange regions (spo_p2p_alloc / spo_p2p_open: uncached, IPC-mapped),
  // this rank, the global optimiser-step count before this launch (same on every rank: tags), the row-split section's offset
  void* xr_region[8];
  int xr_rank;
  unsigned xr_step0;
  unsigned long long rsx_off;
};
constexpr size_t rs_z_bytes(int R) { return (size_t)2 * 3 * R * R * RS_NS * 4096; }
constexpr int RS_GRAN_WORDS = 2 * RS_MAX_R * 3 * 4;
constexpr int RS_CENSUS_WORDS = 3 * RS_MAX_R;

// minibatch steps run / steps whose speculative layer-1 update turned out clipped and was redone, summed over the launches of the
// process (first optimiser lane of workgroup 0); read through spo_debug_update_counters
__device__ unsigned long long g_rs_counters[2];

__device__ __forceinline__ float pin(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int pin(int v) { asm volatile("" : "+v"(v)); return v; }

template <int NT1>
struct RsCol {                         // per-column inputs of one minibatch, prefetched one step ahead (raw loads; settled at pick-up)
  f4 x[NT1];
  f4 actv;
  float t0, t1;                        // critic: target ; actor: logp_old, adv
};

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
// System-scope 16-byte word at (uniform base) + (32-bit lane offset): the base travels in SGPRs (readfirstlane pins it there), so a
// store / load costs no 64-bit address registers -- with one VGPR pair per (peer, word) the compiler keeps W x 15 addresses alive
// across the step loop and spills the optimiser state.  s_nop 4: an SGPR written by a VALU instruction (v_readfirstlane) needs 5
// wait states before a VMEM instruction reads it, and the hazard recogniser does not look inside inline assembly; s_nop 1 behind a
// store of more than 8 bytes: its data registers are read up to two wait states after issue (update.hip).
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void st16_sys(const char* base_uniform, unsigned lane_byte_off, const u4v v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(lane_byte_off), "v"(v), "s"(base_uniform) : "memory");
#endif
}
__device__ __forceinline__ u4v ld16_sys(const char* base_uniform, unsigned lane_byte_off) {
  u4v v = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(lane_byte_off), "s"(base_uniform) : "memory");
#endif
  return v;
}
__device__ __forceinline__ void wait_vm0() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void pin_u4(u4v& v) {                 // (an inline-asm load's outputs are valid only after the wait)
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
}

// A part of a hidden layer for one column tile: output tiles M0 .. M0 + NOWN - 1 of out[] (rows 16 mt + 4 q + reg, col batch) =
// tanh?(W in + b); the other entries of out[] are left alone (they are the partner waves').  NOWN = 2: two independent accumulator
// chains (dependent distance 64 cycles >= the 40-cycle accumulator latency); NOWN = 1: one chain (each element the same fmaf
// chain as in layer_hidden).  A tiles double-buffered like layer_hidden.
template <int NT_IN, int M0, int NOWN, bool TANH>
__device__ __forceinline__ void layer_part(const float* Wl, int ld, const float* bl, const f4 (&in)[NT_IN], f4 (&out)[HID / 16],
                                           int j, int q) {
  f4 acc[NOWN], a[2][NOWN];
#pragma unroll
  for (int t = 0; t < NOWN; ++t) a[0][t] = *reinterpret_cast<const f4*>(Wl + (16 * (M0 + t) + j) * ld + 4 * q);
#pragma unroll
  for (int t = 0; t < NOWN; ++t) acc[t] = *reinterpret_cast<const f4*>(bl + 16 * (M0 + t) + 4 * q);
#pragma unroll
  for (int nt = 0; nt < NT_IN; ++nt) {
    if (nt + 1 < NT_IN) {
#pragma unroll
      for (int t = 0; t < NOWN; ++t)
        a[(nt + 1) & 1][t] = *reinterpret_cast<const f4*>(Wl + (16 * (M0 + t) + j) * ld + 16 * (nt + 1) + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < NOWN; ++t) acc[t] = mfma4(a[nt & 1][t][r], in[nt][r], acc[t]);
  }
#pragma unroll
  for (int t = 0; t < NOWN; ++t) out[M0 + t] = TANH ? fast_tanh4(acc[t]) : acc[t];
}

// FAST: every workgroup of the launch sits on one XCD (checked by the kernel), stores of the exchange stay plain
// XW: world size of the data-parallel form (0: one GPU).  After the row groups of a rank have summed their partials (the L2 hand-off),
// every workgroup holds the rank's gradient; it then pushes it as tagged 16-byte words {f, f, f, global step} -- one system-scope
// store each, untorn -- into a private slot of the same row group's workgroup on EVERY other rank (one hand-off on W - 1 links at
// once), polls the W - 1 slots of its own region and adds the W contributions in rank order (its own at its
// position): the same expression on every rank, so all replicas continue from identical bits; then the mean over the ranks.  Tags are the global optimiser-step count, slots double-buffered by its parity (update.hip).
// NCT: 16-column tiles per workgroup -- 2: 32 rows of every minibatch, the four column waves are 2 column tiles x 2 feature halves;
// 1: 16 rows, the four column waves are the four feature quarters of the one tile (the PPO-Lagrangian step at R = 4: half the
// matrix work per SIMD again).
template <int KIN, int R, bool FAST, bool PROF, int XW = 0, int NCT = 2>
__device__ __forceinline__ void rs_body(const RsArgs& a, float* const lds) {
  // The launch's pointers as INDIVIDUAL scalar-register pairs.  Read straight from the argument struct they arrive as one
  // s_load_dwordx16 -- a 16-register tuple that the allocator can only spill and restore WHOLE: 16 v_readlane at each of 41 places
  // in the step loop, 656 of the loop's ~4 000 instructions, to get at one 64-bit pointer each time (round 6, ISA census).  A plain
  // copy (or an empty asm on an "s" operand) is coalesced back into the tuple; a trip through a VGPR is not (own_sgprs).
  float* const p_theta = own_sgprs(a.theta);
  float* const p_adam_m = own_sgprs(a.adam_m);
  float* const p_adam_v = own_sgprs(a.adam_v);
  const float* const p_obs = own_sgprs(a.obs);
  const float* const p_act = own_sgprs(a.act);
  const float* const p_logp_old = own_sgprs(a.logp_old);
  const float* const p_tgt_r = own_sgprs(a.tgt_r);
  const float* const p_tgt_c = own_sgprs(a.tgt_c);
  const float* const p_adv = own_sgprs(a.adv);
  using L = NetLds<KIN>;
  using S = RsLds<KIN, NCT>;
  constexpr int LDC = S::LDC;
  constexpr int NT1 = KIN / 16;
  constexpr int NOWN = NCT, NPART = 4 / NCT;      // output tiles per column wave, waves per column tile
  unsigned long long pacc[RS_NPHASE] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define RS_STAMP(i)                                            \
  if (PROF) {                                                  \
    const unsigned long long _t = __builtin_readcyclecounter(); \
    pacc[i] += _t - tprev; tprev = _t;                         \
  }
  const int wg = (int)(blockIdx.x >> 3);
  const int netl = wg / R, hf = wg - netl * R, net = a.first_net + netl;
  const int tid = threadIdx.x, lane = tid & 63, j_ = lane & 15, q_ = lane >> 4;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_col = wave8 >= 4;                // waves 0 .. 3: optimiser (one per SIMD), 4 .. 7: column (wave i + 4 shares wave i's SIMD)
  const int D = a.cfg.obs_dim, A = a.cfg.act_dim, B = a.cfg.batch;
  const NetGeom g = net_geom(D, A, net);
  const bool is_actor = (net == 2);
  const int OUT = g.OUT;
  const int ls_off = g.off - A;                  // actor only
  float* const red = lds + S::RED;
  const int64_t nsteps = (a.M + B - 1) / B;

  stage_net<KIN>(p_theta, g, lds, tid, 512);
  if (tid < 16) {
    // the Gaussian's per-action constants of the actor's loss, kept beside log_std by its owner (the optimiser lanes that run its
    // Adam): the four column waves no longer recompute exp / rcp per lane and step
    const bool on = is_actor && tid < A;
    const float lsv = on ? p_theta[ls_off + tid] : 0.f;
    const float sdv = __expf(lsv);
    lds[S::LS + tid] = lsv;
    lds[S::LS + 16 + tid] = __builtin_amdgcn_rcpf(sdv * sdv);
    lds[S::LS + 32 + tid] = on ? lsv + LOG_SQRT_2PI : 0.f;
  }
  if (tid < 128) red[tid] = 0.f;
  __syncthreads();
  if (XW > 0 && tid < 8) reinterpret_cast<unsigned long long*>(red + 100)[tid] = reinterpret_cast<unsigned long long>(a.xr_region[tid]);
  __syncthreads();

  if (is_col) {
    // =========================================================================================================== column waves
    // column wave cw: column tile ct = cw & 1 (16 batch columns), feature half fh = cw >> 1 (output tiles 2 fh, 2 fh + 1 of every
    // hidden layer / backward product).  Two separate instantiations of the loop (fh is a constant in each: every register array
    // is indexed statically).
    const int cw = wave8 - 4, ct = cw % NCT, part_rt = cw / NCT;
    int j = j_, q = q_;
    int lcol = 16 * ct + j_;                     // column inside the workgroup's 16 NCT
    const int gcol = 16 * NCT * hf + 16 * ct + j_;     // column inside the minibatch
#define RS_REIDX { j = pin(j_); q = pin(q_); lcol = 16 * ct + j; }
    const float clip_lo = 1.f - a.cfg.clip, clip_hi = 1.f + a.cfg.clip;
    const float* tgt = (net == 0) ? p_tgt_r : p_tgt_c;
    auto perm_pos = [&](int64_t s_) -> int64_t {
      const int64_t base_ = s_ * B;
      const int64_t rem_ = a.M - base_;
      const int nc_ = (int)(rem_ < B ? rem_ : B);
      return base_ + (gcol < nc_ ? gcol : 0);
    };
    auto fetch = [&](int64_t smp, RsCol<NT1>& cd) {
      load_obs_tiles_raw<KIN>(p_obs + smp * D, D, q, cd.x);
      if (!is_actor) {
        cd.t0 = tgt[smp]; cd.t1 = 0.f; cd.actv = f4{0.f, 0.f, 0.f, 0.f};
      } else {
        cd.t0 = p_logp_old[smp]; cd.t1 = p_adv[smp];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 4 * q + r;
          cd.actv[r] = p_act[smp * A + (ai < A ? ai : 0)];
        }
      }
    };
    auto settle = [&](RsCol<NT1>& cd) {
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) cd.x[nt][e] = pin(cd.x[nt][e]);
      mask_obs_tiles<KIN>(D, q, cd.x);
      cd.t0 = pin(cd.t0);
      if (is_actor) {
        cd.t1 = pin(cd.t1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 4 * q + r;
          const float av = pin(cd.actv[r]);
          cd.actv[r] = ai < A ? av : 0.f;
        }
      }
    };
    auto col_loop = [&](auto PARTC) {
    constexpr int PART = decltype(PARTC)::value;
    constexpr int M0 = PART * NOWN;              // own output tiles M0 .. M0 + NOWN - 1; the others are the partner waves'
#define RS_OWN(MT) ((MT) >= M0 && (MT) < M0 + NOWN)
    // Pipeline of the column inputs, all of it behind this wave's half of L1 -- where it would otherwise wait for the optimiser waves
    // (W2) -- and nothing in the stretch between b5 and b1, where the SIMD belongs to the optimiser wave: in step s the rows of
    // minibatch s + 1 (requested during step s - 1: a whole step to land; barriers do not drain loads) are picked up (pad selects,
    // x^T image), the rows of minibatch s + 2 are requested, the sample index for s + 3 is loaded.
    RsCol<NT1> nxt, raw;
    int smp2 = 0;
    fetch((int64_t)a.perm[perm_pos(0)], nxt);
    if (nsteps > 1) fetch((int64_t)a.perm[perm_pos(1)], raw);
    if (nsteps > 2) smp2 = a.perm[perm_pos(2)];
    settle(nxt);
    if (PART == 0) {
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[S::XT + (16 * nt + 4 * q + e) * LDC + lcol] = nxt.x[nt][e];
    }
    if (PROF) tprev = __builtin_readcyclecounter();

    for (int64_t s = 0; s < nsteps; ++s) {
      const int64_t base = s * B;
      const int64_t rem = a.M - base;
      const int ncols = (int)(rem < B ? rem : B);
      const float inv_n = 1.f / (float)ncols;
      const bool cv = gcol < ncols;
      RsCol<NT1> cur;
      f4 h1[4], h2[4];
      RS_REIDX
      cur = nxt;
      RS_STAMP(0)                                                        // settle + x^T
      __syncthreads();                                                    // b1: W1 / b1 of the previous step's update in place
      RS_STAMP(1)
      {
        RS_REIDX
        layer_part<NT1, M0, NOWN, true>(lds + L::W1, L::LD1, lds + L::B1, cur.x, h1, j, q);
#pragma unroll
        for (int t = 0; t < NOWN; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[S::H1T + (16 * (M0 + t) + 4 * q + r) * LDC + lcol] = h1[M0 + t][r];
      }
      if (s + 1 < nsteps) {
        RS_REIDX
        settle(raw);
        nxt = raw;
        if (s + 2 < nsteps) fetch((int64_t)pin(smp2), raw);
        if (s + 3 < nsteps) smp2 = a.perm[perm_pos(s + 3)];
        if (PART == 0) {
          float* const xt = lds + S::XT + (int)((s + 1) & 1) * KIN * LDC;
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[(16 * nt + 4 * q + e) * LDC + lcol] = nxt.x[nt][e];
        }
      }
      RS_STAMP(2)                                                        // L1 (own half), next minibatch's columns
      __syncthreads();                                                    // b2: W2 / b2 in place; both halves of h1^T written
      RS_STAMP(3)
      if (reinterpret_cast<const int*>(red)[98] == (int)(s & 0x3fffffff) && s > 0) {
        // the previous step turned out clipped: W1 / b1 were restored and redone exactly while this L1 ran (complete before b2) --
        // once more on the exact weights, and one more barrier for the two halves (the optimiser waves run it too)
        RS_REIDX
        layer_part<NT1, M0, NOWN, true>(lds + L::W1, L::LD1, lds + L::B1, cur.x, h1, j, q);
#pragma unroll
        for (int t = 0; t < NOWN; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[S::H1T + (16 * (M0 + t) + 4 * q + r) * LDC + lcol] = h1[M0 + t][r];
        __syncthreads();
      }
      {
        RS_REIDX
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if (!RS_OWN(mt)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[mt][r] = lds[S::H1T + (16 * mt + 4 * q + r) * LDC + lcol];
          }
        layer_part<4, M0, NOWN, true>(lds + L::W2, LDH, lds + L::B2, h1, h2, j, q);
#pragma unroll
        for (int t = 0; t < NOWN; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[S::H2T + (16 * (M0 + t) + 4 * q + r) * LDC + lcol] = h2[M0 + t][r];
      }
      RS_STAMP(4)                                                        // partner's h1, L2 (own half)
      __syncthreads();                                                    // b3: W3 / b3 / log_std in place; both halves of h2^T written
      RS_STAMP(5)
      f4 dz2[4];
      {
        RS_REIDX
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if (!RS_OWN(mt)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[mt][r] = lds[S::H2T + (16 * mt + 4 * q + r) * LDC + lcol];
          }
        float w3c[4][NOWN];                                                 // issued early: their latency hides under the output layer and the loss
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < NOWN; ++t) w3c[r][t] = lds[L::W3 + (4 * q + r) * LDH + 16 * (M0 + t) + j];
        const f4 o = layer_out(lds + L::W3, lds + L::B3, h2, j, q);      // (both halves of a column tile: the same 8 products)
        float ivar[4], lsd[4], amask[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 4 * q + r;
          amask[r] = (is_actor && ai < A) ? 1.f : 0.f;
          ivar[r] = lds[S::LS + 16 + ai];                                 // 1 / exp(log_std)^2 (1 on pad rows and for the critics)
          lsd[r] = lds[S::LS + 32 + ai];                                  // log_std + log sqrt(2 pi) (0 on pad rows)
        }
        f4 dO = {0.f, 0.f, 0.f, 0.f}, dls = {0.f, 0.f, 0.f, 0.f};
        float lsum = 0.f;
        if (!is_actor) {
          // mse_loss(critic(obs), target)  (ppo_lag.py:307-309)
          const float diff = o[0] - cur.t0;
          const float lm = (q == 0 && cv) ? 1.f : 0.f;
          lsum = lm * diff * diff;
          dO[0] = lm * (2.f * diff * inv_n);
        } else {
          float lp = 0.f, dif[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dif[r] = cur.actv[r] - o[r];
            lp += -(dif[r] * dif[r]) * (0.5f * ivar[r]) - lsd[r];
          }
          lp = quad_row_sum(lp);
          const float adv = cur.t1;
          const float ratio = __expf(lp - cur.t0);                        // ppo_lag.py:317
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
          for (int r = 0; r < 4; ++r) {
            const float z = dif[r] * ivar[r];
            dO[r] = dlp * z;
            dls[r] = (dlp * amask[r]) * (dif[r] * z - 1.f);
          }
        }
        // dO -> dZ2 (own half)
        f4 acc[NOWN];
#pragma unroll
        for (int t = 0; t < NOWN; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < NOWN; ++t) acc[t] = mfma4(w3c[r][t], dO[r], acc[t]);
#pragma unroll
        for (int t = 0; t < NOWN; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) dz2[M0 + t][r] = acc[t][r] * fmaf(-h2[M0 + t][r], h2[M0 + t][r], 1.f);
#pragma unroll
        for (int t = 0; t < NOWN; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[S::DZ2T + (16 * (M0 + t) + 4 * q + r) * LDC + lcol] = dz2[M0 + t][r];
        if (PART == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[S::DOT + (4 * q + r) * LDC + lcol] = dO[r];
          // loss / d(log_std) partials of this column tile: they travel with the layer-2 / 3 gradients
          const float ls = wave_sum_lane63(lsum);
          if (lane == 63) red[ct] = ls;
          if (is_actor) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t = row_sum_lane15(dls[r]);
              if (j == 15) red[16 + 16 * ct + 4 * q + r] = t;
            }
          }
        }
      }
      RS_STAMP(6)                                                        // partner's h2, L3, loss, dO -> dZ2 (own half), images
      __syncthreads();                                                    // b4: h1^T, h2^T, dZ2^T, dO^T complete
      RS_STAMP(7)
      {
        RS_REIDX
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if (!RS_OWN(mt)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dz2[mt][r] = lds[S::DZ2T + (16 * mt + 4 * q + r) * LDC + lcol];
          }
        f4 acc[NOWN], dz1[NOWN];
#pragma unroll
        for (int t = 0; t < NOWN; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
        float w2c[2][4][NOWN];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < NOWN; ++t) w2c[0][r][t] = lds[L::W2 + (4 * q + r) * LDH + 16 * (M0 + t) + j];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          if (nt + 1 < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int t = 0; t < NOWN; ++t)
                w2c[(nt + 1) & 1][r][t] = lds[L::W2 + (16 * (nt + 1) + 4 * q + r) * LDH + 16 * (M0 + t) + j];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NOWN; ++t) acc[t] = mfma4(w2c[nt & 1][r][t], dz2[nt][r], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < NOWN; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) dz1[t][r] = acc[t][r] * fmaf(-h1[M0 + t][r], h1[M0 + t][r], 1.f);
#pragma unroll
        for (int t = 0; t < NOWN; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[S::DZ1T + (16 * (M0 + t) + 4 * q + r) * LDC + lcol] = dz1[t][r];
      }
      RS_STAMP(8)                                                        // partner's dZ2, dZ2 -> dZ1 (own half), image
      __syncthreads();                                                    // b5: dZ1^T complete
      RS_STAMP(9)
    }
    };
    if (part_rt == 0) col_loop(std::integral_constant<int, 0>{});
    else if (part_rt == 1) col_loop(std::integral_constant<int, 1>{});
    else if (NPART == 4 && part_rt == 2) col_loop(std::integral_constant<int, NPART == 4 ? 2 : 0>{});
    else col_loop(std::integral_constant<int, NPART == 4 ? 3 : 0>{});
#undef RS_OWN
    if (PROF && a.prof && lane == 0 && cw == 0 && wg == a.prof_wg)
      for (int i = 0; i < RS_NPHASE; ++i) a.prof[i] = pacc[i];
    __syncthreads();                                                      // after the loop: the last update is complete
#undef RS_REIDX
    return;
  }

  // ============================================================================================================= optimiser waves
  const int ow = wave8;                                                   // 0 .. 3: rows [16 ow, 16 ow + 16) of the weight-gradient tiles
  const int ol = ow * 64 + lane;                                          // 0 .. 255
  int j = j_, q = q_, orow = 16 * ow + 4 * q_;
#define RS_REIDX { j = pin(j_); q = pin(q_); orow = 16 * ow + 4 * q; }
  float* const st_m = p_adam_m; float* const st_v = p_adam_v;
  f4 mW1[NT1], vW1[NT1], mW2[4], vW2[4], mW3, vW3, mls, vls;
  float mb1, vb1, mb2, vb2, mb3 = 0.f, vb3 = 0.f;
  const bool own_b = (q_ == 0);
  const bool own_b3 = (ow == 0 && q_ == 0 && j_ < OUT);
  const bool own_ls = is_actor && ow == 0 && j_ == 0;
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * nt + j;
      const int idx = g.w1() + (orow + r) * D + i;
      mW1[nt][r] = i < D ? p_adam_m[idx] : 0.f;
      vW1[nt][r] = i < D ? p_adam_v[idx] : 0.f;
    }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = g.w2() + (orow + r) * HID + 16 * nt + j;
      mW2[nt][r] = p_adam_m[idx];
      vW2[nt][r] = p_adam_v[idx];
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * q + r;
    const int idx = g.w3() + o * HID + 16 * ow + j;
    mW3[r] = o < OUT ? p_adam_m[idx] : 0.f;
    vW3[r] = o < OUT ? p_adam_v[idx] : 0.f;
    mls[r] = (is_actor && o < A) ? p_adam_m[ls_off + o] : 0.f;
    vls[r] = (is_actor && o < A) ? p_adam_v[ls_off + o] : 0.f;
  }
  mb1 = p_adam_m[g.b1() + 16 * ow + j]; vb1 = p_adam_v[g.b1() + 16 * ow + j];
  mb2 = p_adam_m[g.b2() + 16 * ow + j]; vb2 = p_adam_v[g.b2() + 16 * ow + j];
  if (j < OUT) { mb3 = p_adam_m[g.b3() + j]; vb3 = p_adam_v[g.b3() + j]; }

  const float b1c = a.cfg.beta1, b2c = a.cfg.beta2, eps = a.cfg.adam_eps;
  double pw1 = a.pow_b1, pw2 = a.pow_b2;
  const float lr = is_actor ? a.cfg.lr_actor : a.cfg.lr_critic;
  const float l2 = (!is_actor && a.cfg.use_critic_norm) ? a.cfg.l2_coef : 0.f;
  const float l2x2 = 2.f * l2;
  const float vcoef = (net == 0 && a.cfg.use_value_coefficient) ? 2.f : 1.f;
  const bool has_l2 = (l2 != 0.f) || (vcoef != 1.f);                      // (uniform over the workgroup)
  float stale_sq = a.stale_io ? *a.stale_io : 0.f;
  volatile float* const dead = red + 97;

  // ---- exchange transport (update_ks.hip): bare floats, 16-byte buffer loads / stores, NaN sentinel
  const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(a.zbuf, 0, (int)rs_z_bytes(R), 0x00020000);
  const unsigned zvoff = (unsigned)ol * 16u;
  auto zstore = [&](u4 w, unsigned off) {                          // (the cache policy is an immediate)
    if constexpr (FAST) __builtin_amdgcn_raw_buffer_store_b128(w, zrsrc, zvoff, off, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(w, zrsrc, zvoff, off, 16);
  };
  auto gstore = [&](unsigned long long* dst, unsigned long long w) {
    if constexpr (FAST) __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // slot row of (parity, destination row group, source row group, k): 256 lanes x 16 bytes
  auto slot = [&](int par, int dst, int src, int k) -> unsigned {
    return (unsigned)(((((par * 3 + net) * R + dst) * R + src) * RS_NS + k)) * 4096u;
  };
  const u4 sentinel = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  // push NF groups to every peer (slots k0 ..), no branch between the stores
#define RS_PUSH(V, NF, K0, PAR)                                                                        \
  {                                                                                                    \
    _Pragma("unroll") for (int d_ = 1; d_ < R; ++d_)                                                   \
      _Pragma("unroll") for (int k_ = 0; k_ < (NF); ++k_) {                                            \
        u4 w_;                                                                                         \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) w_[e_] = __float_as_uint((V)[k_][e_]);        \
        zstore(w_, slot((PAR), hf ^ d_, hf, (K0) + k_));                                               \
      }                                                                                                \
  }
  // poll the NF groups of peer h ^ d for d in [D0, D1), put the sentinel back; ACC += sum of those peers (D1 - D0 <= 2)
#define RS_LOADS(ZW, NF, K0, PAR, D0, D1)                                                              \
  {                                                                                                    \
    _Pragma("unroll") for (int d_ = (D0); d_ < (D1); ++d_)                                             \
      _Pragma("unroll") for (int k_ = 0; k_ < (NF); ++k_)                                              \
        (ZW)[d_ - (D0)][k_] = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, zvoff, slot((PAR), hf, hf ^ d_, (K0) + k_), 16); \
  }
  // ZW: u4 [D1 - D0][NF]; PRE: its loads were issued by the caller (RS_LOADS) -- the first look costs no round trip.
  // A slot is complete when none of its dwords is the sentinel: unsigned max over all dwords != 0xFFFFFFFF (v_max3_u32).
#define RS_POLL(ACC, ZW, NF, K0, PAR, D0, D1, PROFSLOT, PRE)                                           \
  {                                                                                                    \
    unsigned spins_ = 0;                                                                               \
    for (bool first_ = (PRE);; first_ = false) {                                                       \
      if (!first_) RS_LOADS(ZW, NF, K0, PAR, D0, D1)                                                   \
      unsigned mx_ = 0u;                                                                               \
      _Pragma("unroll") for (int d_ = 0; d_ < (D1) - (D0); ++d_)                                       \
        _Pragma("unroll") for (int k_ = 0; k_ < (NF); ++k_) {                                          \
          const unsigned m01_ = (ZW)[d_][k_][0] > (ZW)[d_][k_][1] ? (ZW)[d_][k_][0] : (ZW)[d_][k_][1]; \
          const unsigned m23_ = (ZW)[d_][k_][2] > (ZW)[d_][k_][3] ? (ZW)[d_][k_][2] : (ZW)[d_][k_][3]; \
          const unsigned m4_ = m01_ > m23_ ? m01_ : m23_;                                              \
          mx_ = mx_ > m4_ ? mx_ : m4_;                                                                 \
        }                                                                                              \
      if (mx_ != 0xFFFFFFFFu) break;                                                                   \
      if (*dead != 0.f) break;                                                                         \
      if (++spins_ > RS_SPIN_LIMIT) { *a.err = 1; *dead = 1.f; break; }   /* bounded, and sticky: never hang the GPU */ \
      __builtin_amdgcn_s_sleep(1);                                                                     \
    }                                                                                                  \
    if (PROF && (PROFSLOT) >= 0) pacc[(PROFSLOT) < 0 ? 0 : (PROFSLOT)] += spins_;                      \
    _Pragma("unroll") for (int d_ = (D0); d_ < (D1); ++d_)                                             \
      _Pragma("unroll") for (int k_ = 0; k_ < (NF); ++k_) zstore(sentinel, slot((PAR), hf, hf ^ d_, (K0) + k_)); \
    _Pragma("unroll") for (int k_ = 0; k_ < (NF); ++k_)                                                \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                               \
        if constexpr ((D1) - (D0) == 1) (ACC)[k_][e_] = (ACC)[k_][e_] + __uint_as_float((ZW)[0][k_][e_]); \
        else (ACC)[k_][e_] = (ACC)[k_][e_] + (__uint_as_float((ZW)[0][k_][e_]) + __uint_as_float((ZW)[((D1) - (D0)) - 1][k_][e_])); \
      }                                                                                                \
  }
  // all-reduce: own + peer (R = 2), (own + h^1) + (h^2 + h^3) (R = 4: the same tree in every workgroup, every node commutative).
  // ZW1: u4 [1][NF] for peer h^1 (PRE: already loaded)
#define RS_POLL_SUM(V, ZW1, NF, K0, PAR, PROFSLOT, PRE)                                                \
  {                                                                                                    \
    RS_POLL(V, ZW1, NF, K0, PAR, 1, 2, PROFSLOT, PRE)                                                  \
    if constexpr (R == 4) {                                                                            \
      u4 zw23_[2][NF];                                                                                 \
      RS_POLL(V, zw23_, NF, K0, PAR, 2, 4, PROFSLOT, false)                                            \
    }                                                                                                  \
  }

  // ---- cross-rank stage (XW > 0): NFL floats of this lane, words W0 .. of the row-split section of the exchange regions
  constexpr bool XTP = XW >= SPO_RS_XR_TWO_PHASE_MIN;                      // two-phase form (below) from this world size on
  const unsigned gtag0 = a.xr_step0 + 1u;
  // (region pointers from an LDS copy of the kernel-argument table: indexing the table itself with a run-time rank would move the
  //  whole argument block to scratch)
  const unsigned long long* const xtab = reinterpret_cast<const unsigned long long*>(red + 100);
  auto rsx_slot = [&](int rank_, int par_, int src_, int w_) -> const char* {      // (uniform: the lane offset ol * 16 goes separately)
    return uniform_ptr(reinterpret_cast<const char*>(xtab[rank_]) + a.rsx_off +
                       ((((size_t)(par_ * 2 + hf) * 3 + net) * 8 + src_) * 16 + w_) * 4096);
  };
  const unsigned xlane = (unsigned)ol * 16u;
#define RSX_PUSH(FL, NFL, W0, GTAG)                                                                    \
  {                                                                                                    \
    constexpr int NW_ = ((NFL) + 2) / 3;                                                               \
    const int xpar_ = (int)((GTAG) & 1u);                                                              \
    _Pragma("unroll") for (int d_ = 1; d_ < XW; ++d_) {                                                \
      _Pragma("unroll") for (int w_ = 0; w_ < NW_; ++w_) {                                             \
        u4v word_;                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                               \
          word_[i_] = (3 * w_ + i_) < (NFL) ? __float_as_uint((FL)[(3 * w_ + i_) < (NFL) ? 3 * w_ + i_ : 0]) : 0u; \
        word_[3] = (GTAG);                                                                             \
        st16_sys(rsx_slot(a.xr_rank ^ d_, xpar_, a.xr_rank, (W0) + w_), xlane, word_);                  \
      }                                                                                                \
    }                                                                                                  \
  }
  // the NFL floats rank SRC sent to this workgroup (its words W0 ..), polled until every word carries this step's tag
#define RSX_POLL(OUT_, NFL, W0, GTAG, SRC)                                                             \
  {                                                                                                    \
    constexpr int NW_ = ((NFL) + 2) / 3;                                                               \
    const int xpar_ = (int)((GTAG) & 1u);                                                              \
    u4v x_[NW_];                                                                                       \
    unsigned spins_ = 0;                                                                               \
    for (;;) {                                                                                         \
      _Pragma("unroll") for (int w_ = 0; w_ < NW_; ++w_)                                               \
        x_[w_] = ld16_sys(rsx_slot(a.xr_rank, xpar_, (SRC), (W0) + w_), xlane);                        \
      wait_vm0();                                                                                      \
      bool ok_ = true;                                                                                 \
      _Pragma("unroll") for (int w_ = 0; w_ < NW_; ++w_) {                                             \
        pin_u4(x_[w_]);                                                                                \
        ok_ = ok_ && (x_[w_][3] == (GTAG));                                                            \
      }                                                                                                \
      if (ok_ || *dead != 0.f) break;                                                                  \
      if (++spins_ > RS_SPIN_LIMIT) { *a.err = 2; *dead = 1.f; break; }   /* bounded, and sticky */    \
      __builtin_amdgcn_s_sleep(1);                                                                     \
    }                                                                                                  \
    _Pragma("unroll") for (int f_ = 0; f_ < (NFL); ++f_) (OUT_)[f_] = __uint_as_float(x_[f_ / 3][f_ % 3]); \
  }
  // all-reduce (mean) of FL[NFL] over the XW ranks: the ranks' contributions added IN RANK ORDER, ((v0 + v1) + v2) + ..., this
  // rank's own at its position -- the same expression on every rank, hence identical bits; one source in flight at a time (two
  // sources, or the butterfly tree's partial sums, do not fit beside the optimiser state of a 256-register wave at 4 / 8 ranks)
#define RSX_REDUCE(FL, NFL, W0, GTAG)                                                                  \
  {                                                                                                    \
    float acc_[NFL], t_[NFL];                                                                          \
    if (a.xr_rank == 0) {                                                 /* (uniform) */              \
      _Pragma("unroll") for (int f_ = 0; f_ < (NFL); ++f_) acc_[f_] = (FL)[f_];                        \
    } else {                                                                                           \
      RSX_POLL(acc_, NFL, W0, GTAG, 0)                                                                 \
    }                                                                                                  \
    _Pragma("unroll 1") for (int r_ = 1; r_ < XW; ++r_) {                 /* (a loop, not XW copies: registers) */ \
      if (r_ == a.xr_rank) {                                                                           \
        _Pragma("unroll") for (int f_ = 0; f_ < (NFL); ++f_) acc_[f_] = acc_[f_] + (FL)[f_];           \
      } else {                                                                                         \
        RSX_POLL(t_, NFL, W0, GTAG, r_)                                                                \
        _Pragma("unroll") for (int f_ = 0; f_ < (NFL); ++f_) acc_[f_] = acc_[f_] + t_[f_];             \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int f_ = 0; f_ < (NFL); ++f_) (FL)[f_] = acc_[f_] * (1.f / (float)XW);      \
  }

  // ---- two-phase form of the cross-rank stage (XW >= SPO_RS_XR_TWO_PHASE_MIN): word gw of a lane belongs to rank (gw / G) mod XW,
  // G = 8 / XW consecutive words (a unit) sharing an owner, so that a unit's contributions are ONE poll batch of 8 loads.
  // SCATTER: every rank sends each word to its owner only; REDUCE_OWN: the owner polls the XW - 1 contributions of a word TOGETHER
  // (XW loads in flight: one round trip, where the all-to-all form above polls XW - 1 sources of ALL words one after the other: its
  // poll buffer for two sources at once does not fit beside the optimiser state), adds them in rank order with its own at its
  // position, takes the mean and sends the finished word to every other rank; GATHER: a rank picks up the words it does not own, one
  // batch, one round trip.  Two cross-rank hand-offs instead of one, but 2 exposed round trips instead of XW - 1, and 13 + 14
  // words per lane and step on the links instead of 15 (XW - 1).  A word is reduced by exactly one rank: the replicas continue
  // from identical bits by construction.  Slots: the all-to-all form's [parity][row group][network][source][word] -- at the
  // owner, [source = sender] holds a contribution; elsewhere [source = owner] holds the finished word (disjoint entries).
#define RSX2_OWNER(GW) (((GW) / XG_) % XW_)
#define RSX2_SCATTER(FL, NFL, W0, GTAG)                                                                \
  {                                                                                                    \
    constexpr int NW_ = ((NFL) + 2) / 3;                                                               \
    constexpr int XW_ = XW > 0 ? XW : 1, XG_ = XW_ >= 8 ? 1 : 8 / XW_;                                 \
    const int xpar_ = (int)((GTAG) & 1u);                                                              \
    _Pragma("unroll") for (int w_ = 0; w_ < NW_; ++w_) {                                               \
      const int o_ = RSX2_OWNER((W0) + w_);                                                            \
      if (o_ != a.xr_rank) {                                              /* (uniform) */              \
        u4v word_;                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                               \
          word_[i_] = (3 * w_ + i_) < (NFL) ? __float_as_uint((FL)[(3 * w_ + i_) < (NFL) ? 3 * w_ + i_ : 0]) : 0u; \
        word_[3] = (GTAG);                                                                             \
        st16_sys(rsx_slot(o_, xpar_, a.xr_rank, (W0) + w_), xlane, word_);                             \
      }                                                                                                \
    }                                                                                                  \
  }
  // (a unit = XG_ consecutive words of one owner: 8 / XW, so that a unit's XG_ x XW contributions are one poll batch of 8 loads)
#define RSX2_REDUCE_OWN(FL, NFL, W0, GTAG)                                                             \
  {                                                                                                    \
    constexpr int NW_ = ((NFL) + 2) / 3;                                                               \
    constexpr int XW_ = XW > 0 ? XW : 1, XG_ = XW_ >= 8 ? 1 : 8 / XW_;                                 \
    const int xpar_ = (int)((GTAG) & 1u);                                                              \
    _Pragma("unroll") for (int u_ = (W0) / XG_; u_ * XG_ < (W0) + NW_; ++u_) {                         \
      if ((u_ % XW_) == a.xr_rank) {                                      /* (uniform) this rank's unit */ \
        u4v x_[XG_][XW_];                                                                              \
        unsigned spins_ = 0;                                                                           \
        for (;;) {                                                                                     \
          _Pragma("unroll") for (int g_ = 0; g_ < XG_; ++g_)                                           \
            _Pragma("unroll") for (int s_ = 0; s_ < XW_; ++s_) {          /* (its own entry is never written: loaded, ignored) */ \
              const int gw_ = u_ * XG_ + g_;                                                           \
              if (gw_ >= (W0) && gw_ < (W0) + NW_) x_[g_][s_] = ld16_sys(rsx_slot(a.xr_rank, xpar_, s_, gw_), xlane); \
            }                                                                                          \
          wait_vm0();                                                                                  \
          bool ok_ = true;                                                                             \
          _Pragma("unroll") for (int g_ = 0; g_ < XG_; ++g_)                                           \
            _Pragma("unroll") for (int s_ = 0; s_ < XW_; ++s_) {                                       \
              const int gw_ = u_ * XG_ + g_;                                                           \
              if (gw_ >= (W0) && gw_ < (W0) + NW_) {                                                   \
                pin_u4(x_[g_][s_]);                                                                    \
                ok_ = ok_ && (s_ == a.xr_rank || x_[g_][s_][3] == (GTAG));                             \
              }                                                                                        \
            }                                                                                          \
          if (ok_ || *dead != 0.f) break;                                                              \
          if (++spins_ > RS_SPIN_LIMIT) { *a.err = 2; *dead = 1.f; break; }   /* bounded, and sticky */ \
          __builtin_amdgcn_s_sleep(1);                                                                 \
        }                                                                                              \
        _Pragma("unroll") for (int g_ = 0; g_ < XG_; ++g_) {                                           \
          const int gw_ = u_ * XG_ + g_;                                                               \
          if (gw_ >= (W0) && gw_ < (W0) + NW_) {                                                       \
            const int w_ = gw_ - (W0);                                                                 \
            u4v red_;                                                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                         \
              const float own_ = (3 * w_ + i_) < (NFL) ? (FL)[(3 * w_ + i_) < (NFL) ? 3 * w_ + i_ : 0] : 0.f; \
              float acc_ = a.xr_rank == 0 ? own_ : __uint_as_float(x_[g_][0][i_]);                     \
              _Pragma("unroll") for (int s_ = 1; s_ < XW_; ++s_)                                       \
                acc_ = acc_ + (s_ == a.xr_rank ? own_ : __uint_as_float(x_[g_][s_][i_]));              \
              acc_ = acc_ * (1.f / (float)XW_);                                                        \
              if ((3 * w_ + i_) < (NFL)) (FL)[(3 * w_ + i_) < (NFL) ? 3 * w_ + i_ : 0] = acc_;         \
              red_[i_] = __float_as_uint(acc_);                                                        \
            }                                                                                          \
            red_[3] = (GTAG);                                                                          \
            _Pragma("unroll") for (int d_ = 1; d_ < XW_; ++d_)                                         \
              st16_sys(rsx_slot(a.xr_rank ^ d_, xpar_, a.xr_rank, gw_), xlane, red_);                  \
          }                                                                                            \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
  }
#define RSX2_GATHER(FL, NFL, W0, GTAG)                                                                 \
  {                                                                                                    \
    constexpr int NW_ = ((NFL) + 2) / 3;                                                               \
    constexpr int XW_ = XW > 0 ? XW : 1, XG_ = XW_ >= 8 ? 1 : 8 / XW_;                                 \
    const int xpar_ = (int)((GTAG) & 1u);                                                              \
    u4v x_[NW_];                                                                                       \
    unsigned spins_ = 0;                                                                               \
    for (;;) {                                                                                         \
      _Pragma("unroll") for (int w_ = 0; w_ < NW_; ++w_)                  /* [source = the word's owner] */ \
        x_[w_] = ld16_sys(rsx_slot(a.xr_rank, xpar_, RSX2_OWNER((W0) + w_), (W0) + w_), xlane);        \
      wait_vm0();                                                                                      \
      bool ok_ = true;                                                                                 \
      _Pragma("unroll") for (int w_ = 0; w_ < NW_; ++w_) {                                             \
        pin_u4(x_[w_]);                                                                                \
        ok_ = ok_ && (RSX2_OWNER((W0) + w_) == a.xr_rank || x_[w_][3] == (GTAG));                      \
      }                                                                                                \
      if (ok_ || *dead != 0.f) break;                                                                  \
      if (++spins_ > RS_SPIN_LIMIT) { *a.err = 2; *dead = 1.f; break; }   /* bounded, and sticky */    \
      __builtin_amdgcn_s_sleep(1);                                                                     \
    }                                                                                                  \
    _Pragma("unroll") for (int f_ = 0; f_ < (NFL); ++f_)                                               \
      (FL)[f_] = (RSX2_OWNER((W0) + f_ / 3) == a.xr_rank) ? (FL)[f_] : __uint_as_float(x_[f_ / 3][f_ % 3]); \
  }

#define RS_ADAM(ADDR, G, M, V)                                                             \
  {                                                                                        \
    const AdamOut o_ = adam1(lds[ADDR], (G), (M), (V), b1c, b2c, eps, step_size, inv_bc2s); \
    (M) = o_.m; (V) = o_.v; lds[ADDR] = o_.p;                                              \
  }

  // step 0 has nothing to wait for: its forward runs on the weights staged at launch
  __syncthreads();                                                        // b1 of step 0
  __syncthreads();                                                        // b2
  __syncthreads();                                                        // b3
  if (PROF) tprev = __builtin_readcyclecounter();
  float last_loss = 0.f;
  int n_redo = 0;
  for (int64_t s = 0; s < nsteps; ++s) {
    const int64_t base = s * B;
    const int64_t rem = a.M - base;
    const int ncols = (int)(rem < B ? rem : B);
    const float inv_n = 1.f / (float)ncols;
    const unsigned tag = a.tag_base + (unsigned)s + 1u;
    const int par = (int)(s & 1);
    unsigned long long* const grow = a.gran + ((size_t)par * RS_MAX_R + hf) * 12;   // [parity][row group][network][wave]
    f4 gA[7];                                                             // W2 tiles, W3 tile, {db2, db3, loss partial, -}, d(log_std)
    f4 gB[NT1 + 1];                                                       // W1 tiles, {db1, -, -, -}
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the resets of the previous step are acknowledged by the L2
    RS_STAMP(0)
    __syncthreads();                                                      // b4: h1^T, h2^T, dZ2^T, dO^T complete
    RS_STAMP(1)
    float gsq = 0.f, psq = 0.f;
    float gb1, gb2, gb3 = 0.f;
    {
      // ---- dW2, dW3 over this workgroup's 32 rows; db2, db3
      RS_REIDX
      f4 az2[NCT], az3[NCT], bh[NCT][4], b3[NCT];
#pragma unroll
      for (int r4 = 0; r4 < NCT; ++r4) {
        az2[r4] = *reinterpret_cast<const f4*>(lds + S::DZ2T + (16 * ow + j) * LDC + 16 * r4 + 4 * q);
        az3[r4] = *reinterpret_cast<const f4*>(lds + S::DOT + j * LDC + 16 * r4 + 4 * q);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          bh[r4][nt] = *reinterpret_cast<const f4*>(lds + S::H1T + (16 * nt + j) * LDC + 16 * r4 + 4 * q);
        b3[r4] = *reinterpret_cast<const f4*>(lds + S::H2T + (16 * ow + j) * LDC + 16 * r4 + 4 * q);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) gA[nt] = f4{0.f, 0.f, 0.f, 0.f};
      f4 w3a = {0.f, 0.f, 0.f, 0.f}, w3b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r4 = 0; r4 < NCT; ++r4)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) gA[nt] = mfma4(az2[r4][e], bh[r4][nt][e], gA[nt]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        w3a = mfma4(az3[0][e], b3[0][e], w3a);
        if constexpr (NCT > 1) w3b = mfma4(az3[NCT - 1][e], b3[NCT - 1][e], w3b);
      }
      gA[4] = w3a + w3b;
      float rs2 = 0.f, rs3 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < NCT; ++r4) {
        rs2 += (az2[r4][0] + az2[r4][1]) + (az2[r4][2] + az2[r4][3]);
        rs3 += (az3[r4][0] + az3[r4][1]) + (az3[r4][2] + az3[r4][3]);
      }
      gA[5] = f4{quad_row_sum(rs2), quad_row_sum(rs3), red[0] + red[1], 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) gA[6][r] = red[16 + 4 * q + r] + red[32 + 4 * q + r];
      RS_PUSH(gA, 7, NT1 + 1, par)
    }
    // ---- the optimiser waits for the last image here anyway (the column waves' last backward product is longer than dW2 / dW3):
    // the x^T operands of dW1 (complete since b1), this lane's W1 / b1 parameters (L2 term, backup), the optimiser scalars of the step
    f4 bx[NCT][NT1], pW1[NT1];
    float pb1;
    {
      RS_REIDX
      const float* const xt = lds + S::XT + par * KIN * LDC;
#pragma unroll
      for (int r4 = 0; r4 < NCT; ++r4)
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
          bx[r4][nt] = *reinterpret_cast<const f4*>(xt + (16 * nt + j) * LDC + 16 * r4 + 4 * q);
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pW1[nt][r] = lds[L::W1 + (orow + r) * L::LD1 + 16 * nt + j];
      pb1 = lds[L::B1 + 16 * ow + j];
    }
    pw1 *= (double)b1c; pw2 *= (double)b2c;
    float step_size, inv_bc2s;
    adam_scalars(lr, pw1, pw2, step_size, inv_bc2s);
    RS_STAMP(2)                                                            // dW2, dW3, stores, preloads
    __syncthreads();                                                      // b5: dZ1^T complete
    RS_STAMP(3)
    // the first peer's layer-2 / 3 partials (sent at b4: long there): their round trip runs under the dW1 products
    u4 zwA[1][7];
    if constexpr (R == 2) RS_LOADS(zwA, 7, NT1 + 1, par, 1, 2)            // (R = 4: 28 more live registers across dW1 spill the wave)
    {
      // ---- dW1, db1
      RS_REIDX
      f4 az1[NCT];
#pragma unroll
      for (int r4 = 0; r4 < NCT; ++r4)
        az1[r4] = *reinterpret_cast<const f4*>(lds + S::DZ1T + (16 * ow + j) * LDC + 16 * r4 + 4 * q);
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) gB[nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r4 = 0; r4 < NCT; ++r4)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) gB[nt] = mfma4(az1[r4][e], bx[r4][nt][e], gB[nt]);
      float rs1 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < NCT; ++r4) rs1 += (az1[r4][0] + az1[r4][1]) + (az1[r4][2] + az1[r4][3]);
      gB[NT1] = f4{quad_row_sum(rs1), 0.f, 0.f, 0.f};
      RS_PUSH(gB, NT1 + 1, 0, par)
    }
    RS_STAMP(4)                                                            // dW1, stores
    // ---- while layer 1's partials travel: the peers' layer-2 / 3 partials (sent at b4: long there), their L2 terms and norm share
    // (before b5 this work would come straight out of the column waves' last backward product: the two waves of a SIMD add)
    RS_POLL_SUM(gA, zwA, 7, NT1 + 1, par, 10, (R == 2))
    const unsigned gtag = gtag0 + (unsigned)s;                             // (data-parallel form: the global step's tag)
    constexpr int NFA = 26, NFB = 4 * NT1 + 1;                             // floats that cross ranks: layers 2 / 3 (+ db2, db3, d log_std), layer 1 (+ db1)
    float fA[NFA];
    if constexpr (XW > 0) {
      // the rank's layer-2 / 3 gradient goes to the other ranks now; it is picked up behind layer 1's (below)
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) fA[4 * k + e] = gA[k][e];
      fA[20] = gA[5][0]; fA[21] = gA[5][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) fA[22 + e] = gA[6][e];
      if constexpr (XTP) RSX2_SCATTER(fA, NFA, 0, gtag)
      else RSX_PUSH(fA, NFA, 0, gtag)
    }
    auto l2_terms_a = [&]() {
    if (has_l2) {
        // L2 regulariser of the critics (weights AND biases, ppo_lag.py:310-314)
        RS_REIDX
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p_ = lds[L::W2 + (orow + r) * LDH + 16 * nt + j];
            const float g_ = vcoef * fmaf(l2x2, p_, gA[nt][r]);
            gA[nt][r] = g_; gsq = fmaf(g_, g_, gsq); psq = fmaf(p_, p_, psq);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                         // pad rows hold p == 0, g == 0
          const float p_ = lds[L::W3 + (4 * q + r) * LDH + 16 * ow + j];
          const float g_ = vcoef * fmaf(l2x2, p_, gA[4][r]);
          gA[4][r] = g_; gsq = fmaf(g_, g_, gsq); psq = fmaf(p_, p_, psq);
        }
        {
          // biases are replicated over q (all replicas run the same Adam, q == 0 counts towards the norms)
          const float pb2 = lds[L::B2 + 16 * ow + j];
          gb2 = vcoef * fmaf(l2x2, pb2, gA[5][0]);
          if (own_b) { gsq = fmaf(gb2, gb2, gsq); psq = fmaf(pb2, pb2, psq); }
          if (ow == 0) {
            const float pb3 = lds[L::B3 + j];
            gb3 = vcoef * fmaf(l2x2, pb3, gA[5][1]);
            if (q == 0) { gsq = fmaf(gb3, gb3, gsq); psq = fmaf(pb3, pb3, psq); }
          }
        }
      } else {
        // no regulariser, coefficient 1 (the actor; the cost critic without use_critic_norm): the gradient as it is, no sum p^2
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) gsq = fmaf(gA[nt][r], gA[nt][r], gsq);
#pragma unroll
        for (int r = 0; r < 4; ++r) gsq = fmaf(gA[4][r], gA[4][r], gsq);
        gb2 = gA[5][0];
        if (own_b) gsq = fmaf(gb2, gb2, gsq);
        if (ow == 0) {
          gb3 = gA[5][1];
          if (q_ == 0) gsq = fmaf(gb3, gb3, gsq);
          if (own_ls) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gsq = fmaf(gA[6][r], gA[6][r], gsq);   // 0 on pad rows
          }
        }
      }
    };
    if constexpr (XW == 0) l2_terms_a();
    const float loss_data = gA[5][2] * inv_n;
    RS_STAMP(5)                                                            // layers 2 / 3: poll, sums, L2 terms
    {
      u4 zwB[1][NT1 + 1];
      RS_POLL_SUM(gB, zwB, NT1 + 1, 0, par, 11, false)
    }
    if constexpr (XW > 0) {
      float fB[NFB];
#pragma unroll
      for (int k = 0; k < NT1; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) fB[4 * k + e] = gB[k][e];
      fB[4 * NT1] = gB[NT1][0];
      if constexpr (XTP) {
        RSX2_SCATTER(fB, NFB, 9, gtag)
        RSX2_REDUCE_OWN(fA, NFA, 0, gtag)                                  // (its contributions were sent before layer 1's local hand-off)
        RSX2_REDUCE_OWN(fB, NFB, 9, gtag)                                  // exposed round trip 1
        RSX2_GATHER(fA, NFA, 0, gtag)
      } else {
        RSX_PUSH(fB, NFB, 9, gtag)
        RSX_REDUCE(fA, NFA, 0, gtag)                                       // (sent before layer 1's local hand-off: long there)
      }
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) gA[k][e] = fA[4 * k + e];
      gA[5][0] = fA[20]; gA[5][1] = fA[21];
#pragma unroll
      for (int e = 0; e < 4; ++e) gA[6][e] = fA[22 + e];
      l2_terms_a();
      if constexpr (XTP) RSX2_GATHER(fB, NFB, 9, gtag)                     // exposed round trip 2
      else RSX_REDUCE(fB, NFB, 9, gtag)                                    // the exposed cross-rank hand-off
#pragma unroll
      for (int k = 0; k < NT1; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) gB[k][e] = fB[4 * k + e];
      gB[NT1][0] = fB[4 * NT1];
    }
    // ---- layer 1: L2 term, norm share out, then Adam at once with clip coefficient 1 -- max_grad_norm almost never binds, and the
    // next forward waits for nothing else.  Backups in registers; the joint norm is checked behind b1 (below).
    f4 bmW1[NT1], bvW1[NT1];
    float bmb1, bvb1;
    unsigned long long gv0 = 0;
    {
      RS_REIDX
      if (has_l2) {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {                                     // pad columns hold p == 0, g == 0
            const float p_ = pW1[nt][r];
            const float g_ = vcoef * fmaf(l2x2, p_, gB[nt][r]);
            gB[nt][r] = g_; gsq = fmaf(g_, g_, gsq); psq = fmaf(p_, p_, psq);
          }
        gb1 = vcoef * fmaf(l2x2, pb1, gB[NT1][0]);
        if (own_b) { gsq = fmaf(gb1, gb1, gsq); psq = fmaf(pb1, pb1, psq); }
      } else {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) gsq = fmaf(gB[nt][r], gB[nt][r], gsq);
        gb1 = gB[NT1][0];
        if (own_b) gsq = fmaf(gb1, gb1, gsq);
      }
      const float wg_sq = wave_sum_lane63(gsq), wp_sq = wave_sum_lane63(psq);
      if (lane == 63) {
        gstore(grow + 4 * netl + ow, ((unsigned long long)tag << 32) | __float_as_uint(wg_sq));
        red[88 + ow] = wp_sq;                                              // sum p^2 shares: read after the next barrier
      }
      // first look at the norm granules now: for the network that publishes last (the one the step waits for) they are all there
      // by the time Adam has run, and the poll behind b1 costs no round trip; this wave's own share does not travel at all
      if (lane < 4 * a.n_nets) gv0 = __hip_atomic_load(grow + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      {
        const unsigned own_bits = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(wg_sq), 63);
        if (lane == 4 * netl + ow) gv0 = ((unsigned long long)tag << 32) | own_bits;
      }
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        bmW1[nt] = mW1[nt]; bvW1[nt] = vW1[nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const AdamOut o_ = adam1(pW1[nt][r], gB[nt][r], mW1[nt][r], vW1[nt][r], b1c, b2c, eps, step_size, inv_bc2s);
          mW1[nt][r] = o_.m; vW1[nt][r] = o_.v; lds[L::W1 + (orow + r) * L::LD1 + 16 * nt + j] = o_.p;
        }
      }
      bmb1 = mb1; bvb1 = vb1;
      const AdamOut o_ = adam1(pb1, gb1, mb1, vb1, b1c, b2c, eps, step_size, inv_bc2s);
      mb1 = o_.m; vb1 = o_.v; lds[L::B1 + 16 * ow + j] = o_.p;
    }
    RS_STAMP(6)                                                            // L2 term, norm share out, Adam W1 (coefficient 1)
    if (s + 1 < nsteps) __syncthreads();                                  // b1 of step s + 1
    // ---- joint clip_grad_norm_ over all networks (ppo_lag.py:325): the granules of the workgroups with my row-group index
    float coef;
    {
      float mine = 0.f;
      unsigned sp_n = 0;
      const int ngr = 4 * a.n_nets;
      if (lane < ngr) {                                                     // lane k polls granule k (network k / 4, wave k % 4)
        unsigned long long v = gv0;
        unsigned sp2 = 0;
        while ((unsigned)(v >> 32) != tag) {
          if (*dead != 0.f) break;
          if (++sp2 > RS_SPIN_LIMIT) { *a.err = 1; *dead = 1.f; break; }
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(grow + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        mine = __uint_as_float((unsigned)v);
        sp_n = sp2;
      }
      if (PROF) {                                                            // (x 1000: the slot also counts layer-2 / 3 poll retries)
        unsigned mxs = sp_n;
        for (int off = 32; off >= 1; off >>= 1) { const unsigned o2 = __shfl_xor(mxs, off); mxs = mxs > o2 ? mxs : o2; }
        pacc[10] += 1000u * mxs;
      }
      // fixed order: identical in every wave and workgroup.  v_readlane with constant lane numbers (a ds_bpermute per term -- __shfl --
      // cost the lone optimiser wave ~1.4 k cycles here); lanes past 4 n_nets hold 0.f: adding them changes nothing
      float total_sq = stale_sq;
#pragma unroll
      for (int kk = 0; kk < 12; ++kk)
        total_sq += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(mine), kk));
      const float norm = sqrtf(total_sq);
      coef = a.cfg.max_grad_norm / (norm + 1e-6f);                        // clip_grad_norm_ (torch): eps 1e-6
      coef = coef > 1.f ? 1.f : coef;
      stale_sq *= coef * coef;
    }
    RS_STAMP(7)                                                            // wait b1, poll norms, coefficient
    if (coef != 1.f) {
      // ---- clipped after all (rare): layer 1 restored and redone exactly.  The column waves are inside L1 of the next step on the
      // speculative weights; the flag makes them repeat it behind b2 -- by then this redo is complete, no extra barrier.
      RS_REIDX
      ++n_redo;
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        mW1[nt] = bmW1[nt]; vW1[nt] = bvW1[nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const AdamOut o_ = adam1(pW1[nt][r], gB[nt][r] * coef, mW1[nt][r], vW1[nt][r], b1c, b2c, eps, step_size, inv_bc2s);
          mW1[nt][r] = o_.m; vW1[nt][r] = o_.v; lds[L::W1 + (orow + r) * L::LD1 + 16 * nt + j] = o_.p;
        }
      }
      mb1 = bmb1; vb1 = bvb1;
      const AdamOut o_ = adam1(pb1, gb1 * coef, mb1, vb1, b1c, b2c, eps, step_size, inv_bc2s);
      mb1 = o_.m; vb1 = o_.v; lds[L::B1 + 16 * ow + j] = o_.p;
      if (ol == 0) reinterpret_cast<volatile int*>(red)[98] = (int)((s + 1) & 0x3fffffff);
    }
    {
      RS_REIDX
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          RS_ADAM(L::W2 + (orow + r) * LDH + 16 * nt + j, gA[nt][r] * coef, mW2[nt][r], vW2[nt][r])
      RS_ADAM(L::B2 + 16 * ow + j, gb2 * coef, mb2, vb2)
    }
    RS_STAMP(8)                                                            // (redo,) Adam W2
    if (s + 1 < nsteps) __syncthreads();                                  // b2 of step s + 1
    if (coef != 1.f && s + 1 < nsteps) __syncthreads();                   // (the column waves repeat L1: their halves meet at one more barrier)
    {
      RS_REIDX
#pragma unroll
      for (int r = 0; r < 4; ++r)
        RS_ADAM(L::W3 + (4 * q + r) * LDH + 16 * ow + j, gA[4][r] * coef, mW3[r], vW3[r])
      if (ow == 0) {
        RS_ADAM(L::B3 + j, gb3 * coef, mb3, vb3)
        if (is_actor) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ai = 4 * q + r;
            RS_ADAM(S::LS + ai, gA[6][r] * coef, mls[r], vls[r])             // (replicated over j: identical values)
            if (ai < A) {                                                    // ... and the loss constants that follow from it
              const float lsv = lds[S::LS + ai];
              const float sdv = __expf(lsv);
              lds[S::LS + 16 + ai] = __builtin_amdgcn_rcpf(sdv * sdv);
              lds[S::LS + 32 + ai] = lsv + LOG_SQRT_2PI;
            }
          }
        }
      }
      if (ow == 0 && lane == 0 && hf == 0 && s + 1 < nsteps) {
        const float pp = (red[88] + red[89]) + (red[90] + red[91]);        // (written before b1)
        a.losses[s * 3 + net] = is_actor ? -loss_data : loss_data + l2 * pp;
      }
      last_loss = loss_data;                                               // the last step has no barrier before this point: below
    }
    RS_STAMP(9)                                                            // wait b2 + Adam W3 ...
    if (s + 1 < nsteps) __syncthreads();                                  // b3 of step s + 1
  }
  if (PROF && a.prof && lane == 0 && ow == 0 && wg == a.prof_wg)
    for (int i = 0; i < RS_NPHASE; ++i) a.prof[RS_NPHASE + i] = pacc[i];
  __syncthreads();                                                        // after the loop: orders the final image before the write-back
  if (ol == 0 && wg == 0) {
    atomicAdd(&g_rs_counters[0], (unsigned long long)nsteps); atomicAdd(&g_rs_counters[1], (unsigned long long)n_redo);
  }
  if (ow == 0 && lane == 0 && hf == 0 && nsteps > 0) {
    const float pp = (red[88] + red[89]) + (red[90] + red[91]);
    a.losses[(nsteps - 1) * 3 + net] = is_actor ? -last_loss : last_loss + l2 * pp;
  }
#undef RS_ADAM
#undef RS_PUSH
#undef RS_POLL_SUM
#undef RS_POLL
#undef RS_LOADS
#undef RSX2_OWNER
#undef RSX2_SCATTER
#undef RSX2_REDUCE_OWN
#undef RSX2_GATHER
#undef RSX_PUSH
#undef RSX_POLL
#undef RSX_REDUCE
  // ---- write back (row group 0 of every network; a launch that timed out leaves theta and the optimiser state untouched)
  if (hf == 0 && *dead == 0.f) {
    RS_REIDX
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * nt + j;
        if (i < D) {
          const int idx = g.w1() + (orow + r) * D + i;
          p_theta[idx] = lds[L::W1 + (orow + r) * L::LD1 + i];
          st_m[idx] = mW1[nt][r]; st_v[idx] = vW1[nt][r];
        }
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = g.w2() + (orow + r) * HID + 16 * nt + j;
        p_theta[idx] = lds[L::W2 + (orow + r) * LDH + 16 * nt + j];
        st_m[idx] = mW2[nt][r]; st_v[idx] = vW2[nt][r];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * q + r;
      if (o < OUT) {
        const int idx = g.w3() + o * HID + 16 * ow + j;
        p_theta[idx] = lds[L::W3 + o * LDH + 16 * ow + j];
        st_m[idx] = mW3[r]; st_v[idx] = vW3[r];
      }
      if (own_ls && o < A) {
        p_theta[ls_off + o] = lds[S::LS + o];
        st_m[ls_off + o] = mls[r]; st_v[ls_off + o] = vls[r];
      }
    }
    if (own_b) {
      const int o = 16 * ow + j;
      p_theta[g.b1() + o] = lds[L::B1 + o]; st_m[g.b1() + o] = mb1; st_v[g.b1() + o] = vb1;
      p_theta[g.b2() + o] = lds[L::B2 + o]; st_m[g.b2() + o] = mb2; st_v[g.b2() + o] = vb2;
    }
    if (own_b3) { p_theta[g.b3() + j] = lds[L::B3 + j]; st_m[g.b3() + j] = mb3; st_v[g.b3() + j] = vb3; }
    if (ol == 0 && wg == 0 && a.stale_io) *a.stale_io = stale_sq;
  }
#undef RS_REIDX
#undef RS_STAMP
}

template <int KIN, int R, bool PROF, int XW = 0, int NCT = 2>
__global__ __launch_bounds__(512) void ppo_update_rs_kernel(RsArgs a) {
  if (blockIdx.x & 7) return;                    // placement hint (update.hip): the working blocks land on one XCD and share its L2
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const red = lds + RsLds<KIN, NCT>::RED;
  const int wg = (int)(blockIdx.x >> 3), tid = threadIdx.x;
  // ---- placement census (update_ks.hip): do all workgroups of the launch share one XCD (one L2)?  Checked, once per launch,
  // over words every placement delivers; not co-resident (or SPO_RS_SAFE=1): write-through exchange stores, same results.
  bool fast;
  {
    unsigned long long* const xid = a.gran + RS_GRAN_WORDS;
    const unsigned myx = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID
    if (tid == 0) {
      red[96] = 0.f;
      __hip_atomic_store(xid + wg, ((unsigned long long)a.tag_base << 32) | myx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < a.n_nets * R) {
      unsigned long long v = __hip_atomic_load(xid + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while ((unsigned)(v >> 32) != a.tag_base) {
        if (++spins > RS_SPIN_LIMIT) { *a.err = 1; break; }
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(xid + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((unsigned)v != myx) red[96] = 1.f;
    }
    __syncthreads();
    fast = (red[96] == 0.f) && !a.force_safe;
    __syncthreads();
  }
  if (fast) rs_body<KIN, R, true, PROF, XW, NCT>(a, lds);
  else rs_body<KIN, R, false, PROF, XW, NCT>(a, lds);
}

// Exchange scratch: the partial-gradient slots and the norm granules, ordinary device memory.  One block per (device, stream),
// allocated on first use (update_ks.hip's scheme): launches on one stream are ordered and share it.
struct RsEntry { int dev; void* stream; char* base; unsigned tag; };
constexpr int RS_SCRATCH_MAX = 16;
RsEntry g_rs[RS_SCRATCH_MAX] = {};
int g_rs_n = 0;
std::mutex g_rs_mu;
constexpr size_t RS_Z_MAX = rs_z_bytes(RS_MAX_R);
constexpr size_t RS_G_BYTES = (size_t)(RS_GRAN_WORDS + RS_CENSUS_WORDS) * 8;

int rs_scratch(hipStream_t st, float** z, unsigned long long** gran, unsigned* tag_base, unsigned nsteps) {
  const int dev = current_device_slot();
  std::lock_guard<std::mutex> lk(g_rs_mu);
  RsEntry* e = nullptr;
  for (int i = 0; i < g_rs_n; ++i)
    if (g_rs[i].dev == dev && g_rs[i].stream == (void*)st) e = &g_rs[i];
  if (!e) {
    if (g_rs_n == RS_SCRATCH_MAX)
      return spo::fail(-1, "row-split update kernel: more than %d (device, stream) pairs hold exchange scratch in this process; "
                           "call spo_update_scratch_release(stream) for streams that are gone", RS_SCRATCH_MAX);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return spo::fail(-1, "row-split update kernel: first launch on a stream under capture (the scratch block is allocated on "
                           "first use: launch once outside the capture)");
    void* p = nullptr;
    if (int rc = spo::hip_check(hipMalloc(&p, RS_Z_MAX + RS_G_BYTES), "hipMalloc(rs scratch)")) return rc;
    if (int rc = spo::hip_check(hipMemset(p, 0, RS_Z_MAX + RS_G_BYTES), "hipMemset(rs scratch)")) { (void)hipFree(p); return rc; }
    if (int rc = spo::hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize(rs scratch)")) { (void)hipFree(p); return rc; }
    g_rs[g_rs_n] = RsEntry{dev, (void*)st, static_cast<char*>(p), 16u};
    e = &g_rs[g_rs_n++];
  }
  *z = reinterpret_cast<float*>(e->base);
  *gran = reinterpret_cast<unsigned long long*>(e->base + RS_Z_MAX);
  *tag_base = e->tag;
  e->tag += nsteps + 2u;                           // (wraps after 4e9 steps: a tag then meets words 2^32 steps old)
  return 0;
}

template <int KIN, int R, bool PROF, int XW = 0, int NCT = 2>
int rs_launch_k(const RsArgs& a, hipStream_t st) {
  const size_t sh = RsLds<KIN, NCT>::SIZE * sizeof(float);
  static bool attr_done[SPO_MAX_DEVICES] = {};
  const int dslot = current_device_slot();
  if (!attr_done[dslot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_update_rs_kernel<KIN, R, PROF, XW, NCT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(update_rs)");
    attr_done[dslot] = true;
  }
  hipLaunchKernelGGL((ppo_update_rs_kernel<KIN, R, PROF, XW, NCT>), dim3(8 * (a.n_nets * R - 1) + 1), dim3(512), sh, st, a);
  return 0;
}

}  // namespace

int spo::rs_scratch_release(int dev, void* stream_or_null, int all) {
  std::lock_guard<std::mutex> lk(g_rs_mu);
  int freed = 0;
  for (int i = 0; i < g_rs_n;) {
    if (g_rs[i].dev == dev && (all || g_rs[i].stream == stream_or_null)) {
      (void)spo::hip_check(hipFree(g_rs[i].base), "hipFree(rs scratch)");
      g_rs[i] = g_rs[--g_rs_n];
      ++freed;
    } else ++i;
  }
  return freed;
}

int spo::rs_counters(unsigned long long* out2_host, int reset) {
  if (int rc = spo::hip_check(hipMemcpyFromSymbol(out2_host, HIP_SYMBOL(g_rs_counters), 16), "hipMemcpyFromSymbol")) return rc;
  if (reset) {
    unsigned long long z[2] = {0, 0};
    return spo::hip_check(hipMemcpyToSymbol(HIP_SYMBOL(g_rs_counters), z, 16), "hipMemcpyToSymbol");
  }
  return 0;
}

// Does the row-split kernel take this shape?  n_nets 3 = the PPO-Lagrangian step (clipped surrogate), 2 = the critic fit.
extern "C" int spo_update_rs_supported(int obs_dim, int act_dim, int batch, int n_nets) {
  if (obs_dim < 1 || obs_dim > 64 || act_dim < 1 || act_dim > SPO_MAX_ACT || batch < 1) return 0;
  if (n_nets == 3) return batch <= 64 ? 1 : 0;
  if (n_nets == 2) return batch <= 128 ? 1 : 0;
  return 0;
}

// Called by spo_ppo_lag_update_iter / spo_critic_fit_iter (update.hip) when the shape is supported and SPO_UPDATE_FORM selects it.
int spo::rs_update_launch(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs, const float* act,
                          const float* logp_old, const float* target_r, const float* target_c, const float* adv,
                          const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host, int n_nets, float* stale_sq_io,
                          float* losses_out, void* sync_ws, unsigned long long* prof, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  RsArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = perm; a.M = M; a.cfg = *cfg_host; a.losses = losses_out;
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.n_nets = n_nets; a.first_net = 0; a.stale_io = stale_sq_io; a.prof = prof;
  { const char* e = getenv("SPO_RS_SAFE"); a.force_safe = (e && *e && *e != '0') ? 1 : 0; }
  const int64_t nsteps = (M + cfg_host->batch - 1) / cfg_host->batch;
  SPO_REQUIRE(nsteps < (1ll << 30), "update_rs: too many minibatch steps in one launch");
  if (int rc = rs_scratch(st, &a.zbuf, &a.gran, &a.tag_base, (unsigned)nsteps)) return rc;
  // Rows of a minibatch per workgroup: 32 (two row groups up to 64 rows, four up to 128: the critic fit's 128-row minibatches) or,
  // opt-in, 16 (SPO_RS_ROWS=16: four row groups up to 64 rows, the column waves as four feature quarters of one column tile --
  // measured SLOWER, 9.46 against 7.9 us per step: the matrix work per SIMD halves again, but three peers' partials to poll, add
  // and reset put 10.9 k cycles between b5 and b1 where the two-row-group form has 5.4 k; profiles/r06/update_ab_rs.txt)
  static const int rows_env = [] { const char* e = getenv("SPO_RS_ROWS"); return e ? atoi(e) : 32; }();
  const bool rows16 = cfg_host->batch <= 64 && rows_env == 16;
  const int R = (cfg_host->batch <= 64 && !rows16) ? 2 : 4;
  { const char* e = getenv("SPO_RS_PROF_WG"); a.prof_wg = (e && *e) ? atoi(e) : n_nets * R - 1; }
  // every slot starts as the sentinel (a launch leaves them that way unless it stopped on an error)
  if (int rc = spo::hip_check(hipMemsetAsync(a.zbuf, 0xFF, R == 2 ? rs_z_bytes(2) : rs_z_bytes(4), st), "hipMemsetAsync(rs slots)")) return rc;
  const int kin = cfg_host->obs_dim <= 16 ? 16 : cfg_host->obs_dim <= 32 ? 32 : 64;
#define RS_GO(K)                                                                                  \
  {                                                                                               \
    if (rows16) { if (prof && K == 64) return rs_launch_k<K, 4, (K == 64), 0, 1>(a, st); return rs_launch_k<K, 4, false, 0, 1>(a, st); } \
    if (R == 2) return rs_launch_k<K, 2, false>(a, st);                                           \
    return rs_launch_k<K, 4, false>(a, st);                                                       \
  }
  if (kin == 16) RS_GO(16) else if (kin == 32) RS_GO(32) else RS_GO(64)
#undef RS_GO
}

// Data-parallel form (one rank of `world` = 2, 4 or 8): the same kernel with the cross-rank stage (rs_body, XW).  Called by
// spo_ppo_lag_update_iter_dp when SPO_XR_FORM_ROW_SPLIT is the form (update.hip); rsx_off = offset of the row-split section in
// every rank's exchange region.
int spo::rs_update_launch_dp(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs, const float* act,
                             const float* logp_old, const float* target_r, const float* target_c, const float* adv,
                             const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host, float* losses_out, void* sync_ws,
                             int rank, int world, void* const* regions, uint32_t step0, size_t rsx_off, void* stream) {
  SPO_REQUIRE(world == 2 || world == 4 || world == 8, "update_rs (data-parallel): world size %d is not 2, 4 or 8", world);
  SPO_REQUIRE(cfg_host->batch <= 64, "update_rs (data-parallel): batch %d > 64", cfg_host->batch);
  hipStream_t st = (hipStream_t)stream;
  RsArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = perm; a.M = M; a.cfg = *cfg_host; a.losses = losses_out;
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.n_nets = 3; a.first_net = 0; a.stale_io = nullptr; a.prof = nullptr; a.prof_wg = -1;
  { const char* e = getenv("SPO_RS_SAFE"); a.force_safe = (e && *e && *e != '0') ? 1 : 0; }
  for (int r = 0; r < 8; ++r) a.xr_region[r] = r < world ? regions[r] : nullptr;
  a.xr_rank = rank; a.xr_step0 = step0; a.rsx_off = rsx_off;
  const int64_t nsteps = (M + cfg_host->batch - 1) / cfg_host->batch;
  SPO_REQUIRE(nsteps < (1ll << 30), "update_rs: too many minibatch steps in one launch");
  if (int rc = rs_scratch(st, &a.zbuf, &a.gran, &a.tag_base, (unsigned)nsteps)) return rc;
  if (int rc = spo::hip_check(hipMemsetAsync(a.zbuf, 0xFF, rs_z_bytes(2), st), "hipMemsetAsync(rs slots)")) return rc;
  const int kin = cfg_host->obs_dim <= 16 ? 16 : cfg_host->obs_dim <= 32 ? 32 : 64;
#define RS_GO_DP(K) (world == 2 ? rs_launch_k<K, 2, false, 2>(a, st) : world == 4 ? rs_launch_k<K, 2, false, 4>(a, st) \
                                                                                     : rs_launch_k<K, 2, false, 8>(a, st))
  return kin == 16 ? RS_GO_DP(16) : kin == 32 ? RS_GO_DP(32) : RS_GO_DP(64);
#undef RS_GO_DP
}
