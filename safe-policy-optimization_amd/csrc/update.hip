// PPO-Lagrangian / critic minibatch update kernels, gfx950.
//
// Replaces the inline update loop of safepo/single_agent/ppo_lag.py:297-336 (and the critic fit of
// cpo.py:534-571): per minibatch of 64 rows the reference runs ~150 tiny torch ops (3 MLP fwd/bwd,
// 24 L2 terms, clip_grad_norm_, 3x Adam) = 3.8 ms on CPU, 327 680 strictly sequential times per
// epoch.  Here one PERSISTENT launch runs a whole learning iteration (all minibatches):
//
//   grid = one workgroup per network (reward critic, cost critic, actor), 4 waves each;
//   the network's weights live in LDS (the fp32 master copy) for the whole launch,
//   Adam moments live in REGISTERS, laid out exactly like the MFMA accumulator tiles that produce
//   the weight gradients, so the optimiser step needs no data movement at all;
//   per step the three workgroups exchange ONE scalar each (their ||grad||^2, for the joint
//   clip_grad_norm_) through 8-byte {tag,value} granules with relaxed agent-scope atomics.
//
// Wave w owns batch columns [16w,16w+16) in forward/backward (transposed chaining, mlp_mfma.h) and
// rows [16w,16w+16) of every weight-gradient tile (reduction over the 64 batch columns staged
// through LDS as [feature][batch]).
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "common.h"
#include "mlp_mfma.h"
#include "adam.h"
#include "../../include/safepo_hip.h"
#include "update_rs.h"

namespace {
using namespace spo;

#ifndef SPO_SPECULATIVE_ADAM
#define SPO_SPECULATIVE_ADAM 1
#endif
#ifndef SPO_XR_SCOPE
#define SPO_XR_SCOPE __HIP_MEMORY_SCOPE_SYSTEM     // scope of the tagged-word exchange accesses (A/B knob: ranks on ONE GPU)
#endif
#ifndef SPO_HELPER_PRIO
#define SPO_HELPER_PRIO 2      // s_setprio of the helper waves of ppo_update_h_kernel (A/B knob)
#endif
constexpr int LDB = 64 + 4;     // [feature][batch] LDS row stride (floats)
constexpr int RED_FLOATS = 160;

template <int KIN>
struct UpdLds {
  using L = NetLds<KIN>;
  static constexpr int XT = L::SIZE;
  static constexpr int H1T = XT + KIN * LDB;
  static constexpr int H2T = H1T + HID * LDB;
  static constexpr bool SPLIT = (KIN > 64);        // obs_dim > 64: dZ1^T reuses the dZ2^T buffer (two sub-phases)
  static constexpr int DZ2T = H2T + HID * LDB;
  static constexpr int DZ1T = SPLIT ? DZ2T : DZ2T + HID * LDB;
  static constexpr int DOT = DZ1T + HID * LDB;
  static constexpr int RED = DOT + OUTP * LDB;
  static constexpr int SIZE = RED + RED_FLOATS;
};

// ---- cross-rank exchange regions (direct all-reduce over the xGMI mesh, fused into the step).
// One region per rank, allocated uncached/fine-grained so that stores arriving from peer GPUs are visible to a running
// kernel, shared by IPC handle.  A network's gradient is NV*256 float4 (NV per lane, accumulator layout).  Per step:
//   phase 1 (reduce-scatter): the vector is cut into `world` contiguous slices; every rank PUSHES slice c of its gradient
//           into slot1[parity][me][net] of rank c's region; rank c sums its slice over the sources in rank order and
//           scales by 1/world;
//   phase 2 (all-gather): rank c PUSHES the reduced slice into red2[parity][net] of every region and every rank reads
//           the full reduced vector back from its own region.
// Every element is reduced by exactly one rank, so all replicas see identical bits.
// There are no flags and no barriers: every float travels as an 8-byte {tag, value} word written by ONE system-scope
// write-through store (untorn), and a consumer lane polls exactly the words it needs until their tags equal the
// step's tag -- the latency of a hand-off is one store plus one load instead of store, drain, flag, poll, load.
// Words are laid out in 4 planes (component k of element e at plane k, index e) so a wave's store covers 512
// contiguous bytes.  Tags are the global step count (monotonic, never reset).  Buffers are double-buffered by step
// parity: a word used at step g is rewritten at step g+2, which a peer can only reach after it has consumed every
// phase-2 word of step g+1 from us, all of them written after our reads of step g.
constexpr int XR_MAX_WORLD = 8;
constexpr int XR_SLOT_F4 = (8 + 7) * 256;                               // elements per slot, sized for obs_dim <= 128
constexpr size_t XR_SLOT_WORDS = (size_t)4 * XR_SLOT_F4;                // 4 planes of u64
constexpr size_t XR_SLOT1_WORDS = (size_t)2 * XR_MAX_WORLD * 3 * XR_SLOT_WORDS;     // [parity][source rank][network]
constexpr size_t XR_RED2_WORDS = (size_t)2 * 3 * XR_SLOT_WORDS;                     // [parity][network]
constexpr size_t XR_V1_BYTES = (XR_SLOT1_WORDS + XR_RED2_WORDS) * 8;                // tagged-word protocols (above)
// All-to-all form used by the helper waves of ppo_update_h_kernel (xr_a2a_* below): plain float4 rows plus one flag per
// (source rank, network, stage, helper wave), in their own part of the region so the two protocols never see each other's words.
constexpr int A2A_ROWS = 13;                                                        // slot rows 0 .. NT1 + 8
constexpr size_t A2A_DATA_F4 = (size_t)2 * XR_MAX_WORLD * 3 * A2A_ROWS * 256;       // [parity][source][network][row][lane]
constexpr size_t A2A_FLAG_WORDS = (size_t)2 * XR_MAX_WORLD * 3 * 2 * 4;             // [parity][source][network][stage][wave]
constexpr size_t A2A_DATA_OFF = XR_V1_BYTES;
constexpr size_t A2A_FLAG_OFF = A2A_DATA_OFF + A2A_DATA_F4 * 16;
// ... and (round 6) the row-split kernel's cross-rank words (update_rs.hip, SPO_XR_FORM_ROW_SPLIT) in a section of their own
constexpr size_t RSX_OFF = (A2A_FLAG_OFF + A2A_FLAG_WORDS * 8 + 4095) & ~(size_t)4095;
constexpr size_t XR_REGION_BYTES = RSX_OFF + spo::RSX_BYTES;

struct UpdArgs {
  float* theta; float* adam_m; float* adam_v;
  const float* obs; const float* act; const float* logp_old; const float* tgt_r; const float* tgt_c;
  const float* adv; const int32_t* perm; int64_t M;
  spo_ppo_cfg cfg;
  float* losses;                 // [nsteps][3]
  unsigned long long* slots;     // [2][4] granules
  int* err;                      // slots + 8
  double pow_b1, pow_b2;         // beta^adam_step at launch
  int first_net, n_nets;         // PPO-Lag: 0,3   CPO critic fit: 0,2
  float stale_sq;                // CPO: ||stale actor grad||^2 taking part in the joint clip
  float* stale_io;               // CPO: device scalar carrying stale_sq across launches (read at start, written at end)
  // split (data-parallel) form
  float* flat_grad; int64_t mean_count;
  unsigned long long* prof;      // optional [3][NPHASE] cycle accumulators (debug builds of the launch)
  // KL-penalty actor loss (FOCOPS, CUP second stage): AMODE == 1 instantiations only
  const float* old_mean;         // [M][A] mean of the distribution the KL is taken to
  const float* old_std;          // [A]    its std (state independent)
  float kl_bound, pg_coef;
  double pow_b1_actor, pow_b2_actor;   // the actor's optimiser may be ahead of the critics' (CUP steps it alone)
  // cross-rank (data-parallel) gradient exchange inside the persistent kernel: XR instantiations only
  int xr_rank, xr_world;
  int xr_algo;                         // 1: recursive doubling at power-of-two worlds; 0: reduce-scatter + all-gather everywhere
  int xr_debug;                        // SPO_A2A_DEBUG bits (development): 1 no row stores, 2 no flag stores/waits, 4 no row loads
  unsigned xr_step0;                   // global optimiser-step count before this launch (same on every rank)
  void* xr_region[XR_MAX_WORLD];       // every rank's exchange region (own + IPC-mapped peers), indexed by rank
                                       // (one-grid split form: [2], [3] = the library's CACHED pair for the shared-L2 exchange)
  unsigned long long* xr_census;       // one-grid split form: placement census words (or NULL)
  unsigned xr_census_tag;
  float* backup;                       // main + helper form: [UPD_BACKUP_ROWS][3 workgroups][512 lanes] float4 backup rows
  int xr_helper_rd;                    // main + helper form, XR > 0: 1 = recursive doubling with tagged words on the helper
                                       // waves (default), 0 = flag-based all-to-all (SPO_P2P_A2A=1)
  int spec_mode;                       // main + helper form: 1 = the clip verdict is validated AFTER the next step's forward
                                       // while the previous step was not clipped (SPO_UPDATE_SPEC, default), 0 = never
  int grid_ranks;                      // exchange self-test only: 2 = both ranks in one grid (spo_p2p_selftest_one_grid)
};
constexpr int NPHASE = 10;
constexpr int UPD_BACKUP_ROWS = 32;      // float4 rows per lane of the 512-thread kernels' backup scratch (main+helper form: 3*NT1 + 12 + 8)

__device__ __forceinline__ unsigned long long ld_granule(unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// DST = updated parameter; M, V (vector elements or scalars) updated in place.
#ifdef SPO_ABLATE_ADAM_STATE
#define SPO_ST_MV(IDX, M, V)
#else
#define SPO_ST_MV(IDX, M, V) { st_m[IDX] = (M); st_v[IDX] = (V); }   /* st_m / st_v: the kernel's moment outputs */
#endif
#ifdef SPO_ABLATE_ADAM_STATE
#define SPO_ADAM(DST, P, G, M, V)                                                      \
  {                                                                                    \
    const AdamOut _o = adam1((P), (G), 0.f, 0.f, b1c, b2c, eps, step_size, inv_bc2s);  \
    (DST) = _o.p;                                                                      \
  }
#else
#define SPO_ADAM(DST, P, G, M, V)                                                      \
  {                                                                                    \
    const AdamOut _o = adam1((P), (G), (M), (V), b1c, b2c, eps, step_size, inv_bc2s);  \
    (M) = _o.m; (V) = _o.v; (DST) = _o.p;                                              \
  }
#endif

// Per-column inputs of one 64-column chunk, prefetched one chunk ahead.  The prefetch stores what the loads returned
// (pad lanes read a clamped, valid address); the pad selects run when the chunk is picked up (settle_prefetch), never
// behind the loads -- see load_obs_tiles_raw.
template <int NT1>
struct ColData {
  f4 x[NT1];       // observation tiles (B operand of layer 1)
  f4 actv;         // actor: act[4q..4q+3]
  f4 omv;          // actor, KL-penalty loss: old_mean[4q..4q+3]
  float t0, t1;    // critic: target ; actor: logp_old, adv
};
// Pins the point where a prefetched register is first touched: everything that consumes it (selects, sign extension)
// is data-dependent on this and cannot be scheduled back up to the load.
__device__ __forceinline__ float pinned(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int pinned(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ f4 pinned4(f4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned long long gu64;    // global address space: global_* not flat_*
// Word layout inside a slot: element e (a float4) lives in row e >> 8; a row is 4 planes of 256 words, so component k of
// element e is word (e >> 8) * 1024 + k * 256 + (e & 255).  A wave's store covers 512 contiguous bytes, and the four
// components of one element sit at -4096 / -2048 / 0 / +2048 bytes around one address (13-bit immediates: ONE address
// register pair per element instead of four).
__device__ __forceinline__ gu64* xr_elem(gu64* slot, int e) { return slot + ((size_t)(e >> 8) * 1024 + (e & 255) + 512); }
__device__ __forceinline__ void st_ll(gu64* elem, const f4 v, unsigned tag) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    __hip_atomic_store(elem + (k - 2) * 256, ((u64)tag << 32) | __float_as_uint(v[k]), __ATOMIC_RELAXED,
                       SPO_XR_SCOPE);                                       // global_store_dwordx2 sc0 sc1
}
__device__ __forceinline__ bool ld_ll(gu64* elem, unsigned tag, f4& v) {
  u64 w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    w[k] = __hip_atomic_load(elem + (k - 2) * 256, __ATOMIC_RELAXED, SPO_XR_SCOPE);
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ok = ok && ((unsigned)(w[k] >> 32) == tag);
    v[k] = __uint_as_float((unsigned)w[k]);
  }
  return ok;
}
__device__ __forceinline__ gu64* xr_words(u64 region) { return (gu64*)region; }

constexpr unsigned XR_SPIN_LIMIT = 1u << 21;
constexpr int XR_BATCH = 8;      // (element, source) pairs polled together (16 spills the 64-wide kernel)
// debug profile of the exchange (self-test kernel only): cycles in {push1, reduce polls, push2, final polls} and the
// number of poll rounds {reduce, final}, summed over iterations by thread 0 of every workgroup
__device__ unsigned long long g_xr_prof[8];

// All-reduce(mean) of NV float4 per lane across the ranks for network `net` at global step tag `gtag`.
// Every lane of the workgroup calls it (no barrier inside).  `regions`: LDS table of the ranks' region base
// addresses.  `dead_word` (LDS) goes non-zero after a timed-out poll; from then on this rank keeps publishing
// (peers must not hang on us) but no longer waits, so a broken fabric costs one bounded wait, not one per step.
template <int NV, bool XPROF = false>
__device__ __forceinline__ void xr_allreduce(const u64* regions, int me, int R, int net, int tid, unsigned gtag,
                                             f4 (&pk)[NV], volatile float* dead_word, int* err) {
  unsigned long long tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, rounds_a = 0, rounds_b = 0;
  if (XPROF) tp0 = __builtin_readcyclecounter();
  constexpr int TOTAL = NV * 256;
  const int par = (int)(gtag & 1u);
  const int chunk = (TOTAL + R - 1) / R;
  // ---- phase 1 push: element i = v*256 + tid goes to rank i / chunk
  const size_t slot1_me = ((size_t)(par * XR_MAX_WORLD + me) * 3 + net) * XR_SLOT_WORDS;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int i = v * 256 + tid;
    int dst = 0;
    for (int c = 1; c < R; ++c) dst += (i >= c * chunk) ? 1 : 0;
    st_ll(xr_elem(xr_words(regions[dst]) + slot1_me, i - dst * chunk), pk[v], gtag);
  }
  if (XPROF) tp1 = __builtin_readcyclecounter();
  // ---- reduce my slice over the sources in rank order, push the result to every rank (phase 2)
  const int my_lo = me * chunk;
  const int my_n = (TOTAL - my_lo) < chunk ? (TOTAL - my_lo) : chunk;
  const int lg = R <= 2 ? 1 : R <= 4 ? 2 : 3;                  // sources padded to 2 / 4 / 8 per element
  const int G = XR_BATCH >> lg;                                // elements per batch of XR_BATCH loads
  const int E = (my_n + 255) >> 8;
  const float inv_world = 1.f / (float)R;
  gu64* const src0 = xr_words(regions[me]) + ((size_t)(par * XR_MAX_WORLD) * 3 + net) * XR_SLOT_WORDS;
  const size_t red2_off = XR_SLOT1_WORDS + ((size_t)par * 3 + net) * XR_SLOT_WORDS;
  for (int b = 0; b * G < E; ++b) {
    f4 x[XR_BATCH];
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int jj = 0; jj < XR_BATCH; ++jj) {
        const int g = jj >> lg, r = jj & ((1 << lg) - 1);
        const int e = (b * G + g) * 256 + tid;
        x[jj] = f4{0.f, 0.f, 0.f, 0.f};
        if (r < R && e < my_n) ok = ld_ll(xr_elem(src0 + (size_t)r * 3 * XR_SLOT_WORDS, e), gtag, x[jj]) && ok;
      }
      if (XPROF) rounds_a += 1;
      if (ok || *dead_word != 0.f) break;
      if (++spins > XR_SPIN_LIMIT) { *err = 2; *dead_word = 1.f; break; }        // bounded: never hang the GPU
      __builtin_amdgcn_s_sleep(1);
    }
    if (XPROF) { tp2 += __builtin_readcyclecounter() - tp1; tp1 = __builtin_readcyclecounter(); }
    f4 run = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < XR_BATCH; ++jj) {
      const int g = jj >> lg, r = jj & ((1 << lg) - 1);
      if (r < R) {
        run = (r == 0) ? x[jj] : run + x[jj];
        const int e = (b * G + g) * 256 + tid;
        if (r == R - 1 && e < my_n) {
          const f4 val = run * inv_world;
          for (int dst = 0; dst < R; ++dst) st_ll(xr_elem(xr_words(regions[dst]) + red2_off, my_lo + e), val, gtag);
        }
      }
    }
  }
  // ---- the reduced vector
  gu64* const fin = xr_words(regions[me]) + red2_off;
  unsigned spins = 0;
  if (XPROF) tp3 = __builtin_readcyclecounter();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int v = 0; v < NV; ++v) ok = ld_ll(xr_elem(fin, v * 256 + tid), gtag, pk[v]) && ok;
    if (XPROF) rounds_b += 1;
    if (ok || *dead_word != 0.f) break;
    if (++spins > XR_SPIN_LIMIT) { *err = 2; *dead_word = 1.f; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (XPROF && tid == 0) {
    const unsigned long long tend = __builtin_readcyclecounter();
    atomicAdd(&g_xr_prof[0], tp1 - tp0 - tp2);      // pushes (phase 1 + phase 2 stores), roughly
    atomicAdd(&g_xr_prof[1], tp2);                  // reduce polls
    atomicAdd(&g_xr_prof[2], tp3 - tp0);            // everything before the final poll
    atomicAdd(&g_xr_prof[3], tend - tp3);           // final polls
    atomicAdd(&g_xr_prof[4], rounds_a);
    atomicAdd(&g_xr_prof[5], rounds_b);
    atomicAdd(&g_xr_prof[6], 1ull);
  }
}

// Recursive doubling for a power-of-two number of ranks: in round k every rank swaps its running sum with rank
// me ^ (1 << k) (a whole-vector push into slot [parity][k][net] of the partner's region) and adds what it receives;
// after log2(world) hand-offs everybody holds the total.  Partners add the same two values (a + b == b + a exactly in
// IEEE arithmetic), so by induction all replicas hold identical bits.  One hand-off at 2 ranks, two at 4, three at 8 --
// against two for the reduce-scatter form at any size, but each of these is a single push/poll with no slice
// bookkeeping, and the gradient registers plus six polled rows fit the register file (the 4-rank one-shot -- own gradient
// plus three peer copies live at once -- spilled 250-500 B per lane next to the optimiser state and lost).
// ROW0: first slot row used (two calls per step with disjoint rows may share a tag).
template <int NV, int ROW0 = 0, int VB = 6>                     // VB: gradient rows polled together
__device__ __forceinline__ void xr_allreduce_rd(const u64* regions, int me, int R, int net, int tid, unsigned gtag,
                                                f4 (&pk)[NV], volatile float* dead_word, int* err) {
  static_assert((ROW0 + NV) * 256 <= XR_SLOT_F4, "slot rows");
  const int par = (int)(gtag & 1u);
#pragma unroll 1
  for (int k = 0; (1 << k) < R; ++k) {
    const int peer = me ^ (1 << k);
    const size_t slot = ((size_t)(par * XR_MAX_WORLD + k) * 3 + net) * XR_SLOT_WORDS;
    gu64* const p = xr_words(regions[peer]) + slot;
#pragma unroll
    for (int v = 0; v < NV; ++v) st_ll(xr_elem(p, (ROW0 + v) * 256 + tid), pk[v], gtag);
    gu64* const src = xr_words(regions[me]) + slot;
#pragma unroll
    for (int v0 = 0; v0 < NV; v0 += VB) {
      f4 x[VB];
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int vv = 0; vv < VB; ++vv)
          if (v0 + vv < NV) ok = ld_ll(xr_elem(src, (ROW0 + v0 + vv) * 256 + tid), gtag, x[vv]) && ok;
        if (ok || *dead_word != 0.f) break;
        if (++spins > XR_SPIN_LIMIT) { *err = 2; *dead_word = 1.f; break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int vv = 0; vv < VB; ++vv)
        if (v0 + vv < NV) pk[v0 + vv] += x[vv];
    }
  }
  const float inv_world = 1.f / (float)R;
#pragma unroll
  for (int v = 0; v < NV; ++v) pk[v] *= inv_world;
}

// The same recursive doubling with PACKED words (round 3): three gradient floats and the tag travel in one 16-byte
// {f0, f1, f2, tag} word written by one system-scope write-through global_store_dwordx4 (a lane's aligned 16-byte store
// lands in one memory line as a unit, like the 8-byte words above), so a lane moves ceil(4 NV / 3) words of 16 bytes per
// round instead of 4 NV words of 8: 15 stores and 15 polling loads instead of 44 for the 60-wide network, 133 KB per rank and
// step instead of 199 KB.  Word w of lane t sits at slot + (w * 256 + t) * 16 bytes: a wave's store covers 1 KB contiguous.
#ifndef SPO_XR_PACK16
#define SPO_XR_PACK16 1        // 0: the 8-byte {tag, value} words of rounds 1-2 (A/B knob; the self-test uses the same form)
#endif
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_sys(char* p, const u4v v) {
#if defined(__HIP_DEVICE_COMPILE__)
  // s_nop 1: a VMEM store of more than 8 bytes reads its data registers up to two wait states after issue (gfx940+), and the
  // compiler's hazard recogniser does not look inside inline assembly -- without it the VALU moves that assemble the NEXT word
  // overwrite this word's registers under the store (seen on the GPU: lanes 12-15 of every 16 sent the next word's float)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ u4v ld16_sys(const char* p) {
  u4v v = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
#endif
  return v;
}
// Both ends of the hand-off on ONE XCD (the one-grid split critic fit, checked by a placement census): ordinary cached memory,
// PLAIN stores (the line stays in the XCD's L2 -- an sc1 store writes it through and drops it) and sc1 loads (bypass the reader's
// L1, served by that L2): a hand-off costs an L2 round trip, ~0.5 us, instead of ~2.1 us through uncached memory (round 5,
// csrc/update_ks.hip).
__device__ __forceinline__ void st16_l2(char* p, const u4v v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ u4v ld16_l2(const char* p) {
  u4v v = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
#endif
  return v;
}
#ifndef SPO_XR16_VB
#define SPO_XR16_VB 15         // packed words polled together: all 15 of a 60-wide network (A/B knob; 5 and 8 measured: profiles/r03/p2p_loopback_poll_batch_ab.txt)
#endif
template <int NV, int VB = SPO_XR16_VB, bool LOCAL = false>    // VB: packed words polled together; LOCAL: shared-L2 flavour
__device__ __forceinline__ void xr_allreduce_rd16(const u64* regions, int me, int R, int net, int tid, unsigned gtag,
                                                  f4 (&pk)[NV], volatile float* dead_word, int* err) {
  constexpr int NF = 4 * NV, NW = (NF + 2) / 3;
  static_assert((size_t)NW * 256 * 16 <= XR_SLOT_WORDS * 8, "packed rows must fit the slot");
  const int par = (int)(gtag & 1u);
#pragma unroll 1
  for (int k = 0; (1 << k) < R; ++k) {
    const int peer = me ^ (1 << k);
    const size_t slot_b = (((size_t)(par * XR_MAX_WORLD + k) * 3 + net) * XR_SLOT_WORDS) * 8 + (size_t)tid * 16;
    char* const p = reinterpret_cast<char*>(regions[peer]) + slot_b;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      u4v word;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int f = 3 * w + i;
        word[i] = f < NF ? __float_as_uint(pk[(f < NF ? f : 0) >> 2][(f < NF ? f : 0) & 3]) : 0u;
      }
      word[3] = gtag;
      if (LOCAL) st16_l2(p + (size_t)w * 4096, word); else st16_sys(p + (size_t)w * 4096, word);
    }
    const char* const src = reinterpret_cast<const char*>(regions[me]) + slot_b;
#pragma unroll
    for (int w0 = 0; w0 < NW; w0 += VB) {
      u4v x[VB];
      unsigned spins = 0;
      for (;;) {
#pragma unroll
        for (int vv = 0; vv < VB; ++vv)
          if (w0 + vv < NW) x[vv] = LOCAL ? ld16_l2(src + (size_t)(w0 + vv) * 4096) : ld16_sys(src + (size_t)(w0 + vv) * 4096);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        bool ok = true;
#pragma unroll
        for (int vv = 0; vv < VB; ++vv)
          if (w0 + vv < NW) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(x[vv]));                         // the asm outputs are valid only after the wait
#endif
            ok = ok && (x[vv][3] == gtag);
          }
        if (ok || *dead_word != 0.f) break;
        if (++spins > XR_SPIN_LIMIT) { *err = 2; *dead_word = 1.f; break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int vv = 0; vv < VB; ++vv)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int f = 3 * (w0 + vv) + i;
          if (w0 + vv < NW && f < NF) pk[f >> 2][f & 3] += __uint_as_float(x[vv][i]);
        }
    }
  }
  const float inv_world = 1.f / (float)R;
#pragma unroll
  for (int v = 0; v < NV; ++v) pk[v] *= inv_world;
}

// Reduce-scatter + all-gather with the PACKED 16-byte words (round 4).  The lane's NW words are cut into `world` slices by word
// index (slice c = words [c * cw, (c + 1) * cw), cw = ceil(NW / world)): every lane pushes slice c into slot1[parity][me][net]
// of rank c's region; rank c's lane adds the `world` copies of its cw words in rank order, scales by 1 / world and pushes the
// reduced words into red2[parity][net] of EVERY region; every lane then polls its NW reduced words from its own region.  Each
// word is reduced by exactly one rank, so all replicas hold identical bits.  Against recursive doubling at 8 ranks: TWO
// hand-offs instead of three, NW + world * cw (31) stores and world * cw + NW (31) polling loads per lane instead of 3 NW (45)
// each, 1.75 x the gradient on the wire per rank and step instead of 3 x -- at 2 ranks doubling is the cheaper one (one
// hand-off, 15 + 15).  Any world size <= 8; same tag / parity discipline as the forms above.
template <int NV>
__device__ __forceinline__ void xr_allreduce_rs16(const u64* regions, int me, int R, int net, int tid, unsigned gtag,
                                                  f4 (&pk)[NV], volatile float* dead_word, int* err) {
  constexpr int NF = 4 * NV, NW = (NF + 2) / 3;
  constexpr int XB = 16;                                          // (word, source) pairs polled together in the reduce phase
  static_assert((size_t)NW * 256 * 16 <= XR_SLOT_WORDS * 8, "packed rows must fit the slot");
  const int par = (int)(gtag & 1u);
  const int cw = (NW + R - 1) / R;
  const size_t lane_b = (size_t)tid * 16;
  // ---- phase 1: word w -> rank w / cw
  {
    const size_t slot1_b = (((size_t)(par * XR_MAX_WORLD + me) * 3 + net) * XR_SLOT_WORDS) * 8 + lane_b;
    int dst = 0, left = cw;
    char* base = reinterpret_cast<char*>(regions[0]) + slot1_b;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      u4v word;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int f = 3 * w + i;
        word[i] = f < NF ? __float_as_uint(pk[(f < NF ? f : 0) >> 2][(f < NF ? f : 0) & 3]) : 0u;
      }
      word[3] = gtag;
      st16_sys(base + (size_t)w * 4096, word);
      if (--left == 0 && w + 1 < NW) { ++dst; left = cw; base = reinterpret_cast<char*>(regions[dst]) + slot1_b; }
    }
  }
  // ---- reduce my slice over the sources in rank order; push the result to every rank
  const int w_lo = me * cw;
  const int n_my = (NW - w_lo) < cw ? (NW - w_lo > 0 ? NW - w_lo : 0) : cw;
  const int lg = R <= 2 ? 1 : R <= 4 ? 2 : 3;                     // sources padded to 2 / 4 / 8 per word
  const int G = XB >> lg;                                         // words per batch of XB loads
  const float inv_world = 1.f / (float)R;
  const char* const src0 = reinterpret_cast<const char*>(regions[me]) + (((size_t)(par * XR_MAX_WORLD) * 3 + net) * XR_SLOT_WORDS) * 8 + lane_b;
  const size_t red2_b = (XR_SLOT1_WORDS + ((size_t)par * 3 + net) * XR_SLOT_WORDS) * 8 + lane_b;
#pragma unroll 1
  for (int b = 0; b * G < n_my; ++b) {
    u4v x[XB];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
      for (int jj = 0; jj < XB; ++jj) {
        const int g = jj >> lg, r = jj & ((1 << lg) - 1);
        const int wl = b * G + g;
        x[jj] = u4v{0u, 0u, 0u, gtag};
        if (r < R && wl < n_my) x[jj] = ld16_sys(src0 + (size_t)r * 3 * XR_SLOT_WORDS * 8 + (size_t)(w_lo + wl) * 4096);
      }
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      bool ok = true;
#pragma unroll
      for (int jj = 0; jj < XB; ++jj) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(x[jj]));                             // the asm outputs are valid only after the wait
#endif
        ok = ok && (x[jj][3] == gtag);
      }
      if (ok || *dead_word != 0.f) break;
      if (++spins > XR_SPIN_LIMIT) { *err = 2; *dead_word = 1.f; break; }          // bounded: never hang the GPU
      __builtin_amdgcn_s_sleep(1);
    }
    float run[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < XB; ++jj) {
      const int g = jj >> lg, r = jj & ((1 << lg) - 1);
      const int wl = b * G + g;
      if (r < R) {
#pragma unroll
        for (int i = 0; i < 3; ++i) run[i] = (r == 0) ? __uint_as_float(x[jj][i]) : run[i] + __uint_as_float(x[jj][i]);
        if (r == R - 1 && wl < n_my) {
          const u4v word = {__float_as_uint(run[0] * inv_world), __float_as_uint(run[1] * inv_world), __float_as_uint(run[2] * inv_world), gtag};
          for (int dst = 0; dst < R; ++dst) st16_sys(reinterpret_cast<char*>(regions[dst]) + red2_b + (size_t)(w_lo + wl) * 4096, word);
        }
      }
    }
  }
  // ---- the reduced vector: all NW words from my own region
  const char* const fin = reinterpret_cast<const char*>(regions[me]) + red2_b;
  constexpr int VB = SPO_XR16_VB;
#pragma unroll
  for (int w0 = 0; w0 < NW; w0 += VB) {
    u4v x[VB];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
      for (int vv = 0; vv < VB; ++vv)
        if (w0 + vv < NW) x[vv] = ld16_sys(fin + (size_t)(w0 + vv) * 4096);
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      bool ok = true;
#pragma unroll
      for (int vv = 0; vv < VB; ++vv)
        if (w0 + vv < NW) {
#if defined(__HIP_DEVICE_COMPILE__)
          asm volatile("" : "+v"(x[vv]));
#endif
          ok = ok && (x[vv][3] == gtag);
        }
      if (ok || *dead_word != 0.f) break;
      if (++spins > XR_SPIN_LIMIT) { *err = 2; *dead_word = 1.f; break; }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int vv = 0; vv < VB; ++vv)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int f = 3 * (w0 + vv) + i;
        if (w0 + vv < NW && f < NF) pk[f >> 2][f & 3] = __uint_as_float(x[vv][i]);
      }
  }
}

// Recursive doubling with packed words over a FLAT list of NF floats (round 5, the helper waves of ppo_update_h_body): NF floats
// travel as ceil(NF / 3) words {f0, f1, f2, tag} starting at word W0 of the slot, ALL of them polled together (one dependent
// round trip per hand-off), so a stage of the step (layer 1 = 4 NT1 + 1 floats, layers 2 / 3 = 26 floats) is 6 + 9 words
// instead of the 20 + 28 eight-byte words in two poll batches each of xr_allreduce_rd.  Same slot / tag / parity discipline.
template <int NF, int W0, bool LOCAL = false>
__device__ __forceinline__ void xr_rd16_flat(const u64* regions, int me, int R, int net, int tid, unsigned gtag,
                                             float (&pf)[NF], volatile float* dead_word, int* err) {
  constexpr int NW = (NF + 2) / 3;
  static_assert((size_t)(W0 + NW) * 256 * 16 <= XR_SLOT_WORDS * 8, "packed rows must fit the slot");
  const int par = (int)(gtag & 1u);
#pragma unroll 1
  for (int k = 0; (1 << k) < R; ++k) {
    const int peer = me ^ (1 << k);
    const size_t slot_b = (((size_t)(par * XR_MAX_WORLD + k) * 3 + net) * XR_SLOT_WORDS) * 8 + (size_t)tid * 16 + (size_t)W0 * 4096;
    char* const p = reinterpret_cast<char*>(regions[peer]) + slot_b;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      u4v word;
#pragma unroll
      for (int i = 0; i < 3; ++i) word[i] = (3 * w + i) < NF ? __float_as_uint(pf[(3 * w + i) < NF ? 3 * w + i : 0]) : 0u;
      word[3] = gtag;
      if (LOCAL) st16_l2(p + (size_t)w * 4096, word); else st16_sys(p + (size_t)w * 4096, word);
    }
    const char* const src = reinterpret_cast<const char*>(regions[me]) + slot_b;
    u4v x[NW];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
      for (int w = 0; w < NW; ++w) x[w] = LOCAL ? ld16_l2(src + (size_t)w * 4096) : ld16_sys(src + (size_t)w * 4096);
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      bool ok = true;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(x[w]));                               // the asm outputs are valid only after the wait
#endif
        ok = ok && (x[w][3] == gtag);
      }
      if (ok || *dead_word != 0.f) break;
      if (++spins > XR_SPIN_LIMIT) { *err = 2; *dead_word = 1.f; break; }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (3 * w + i < NF) pf[3 * w + i] += __uint_as_float(x[w][i]);
  }
  const float inv_world = 1.f / (float)R;
#pragma unroll
  for (int f = 0; f < NF; ++f) pf[f] *= inv_world;
}

// AMODE: actor loss.  0 = PPO clipped surrogate (ppo_lag.py:316-319; clip = 1e30 gives the plain policy gradient);
//        1 = KL-penalty form shared by FOCOPS (focops.py:326-337) and CUP's second stage (cup.py:372-383):
//            loss = mean_i(ind_i * KL_i) - pg_coef * mean_i(ind_i) * mean_j(ratio_j * adv_j),
//            KL_i = KL(N(mu_i, sigma) || N(mu_old_i, sigma_old)).sum(-1),  ind_i = [KL_i <= kl_bound].
//        (The reference subtracts a [B] tensor from a [B,1] tensor, so its loss is the mean of a BxB matrix: that is
//        exactly the product of means above.  CUP has no indicator: kl_bound = +inf, pg_coef = -lambda * coef.)
// XR: data-parallel form of the persistent kernel -- every rank runs it on its own env shard and the per-step
//     gradient all-reduce happens inside the step (xr_allreduce above) instead of kernel / RCCL / kernel.
// XR = 1: reduce-scatter + all-gather (any world); 2: recursive doubling (power-of-two worlds).  Separate instantiations:
//     both exchange bodies together do not fit the register file next to the optimiser state.
// The body is a device function over a CONST reference to a kernel-argument struct so that the one-grid split form below
// can run it on either of two argument structs (one per rank) under a block-uniform branch: every pointer stays a plain
// kernel-argument load.  (Selecting per-rank pointers inside one body -- by offsets or by writing the by-value struct --
// makes the gfx950 backend of ROCm 7.2 emit an illegal flat-aperture compare in the XR instantiations.)
template <int KIN, bool PERSIST, bool PROF = false, int AMODE = 0, int XR = 0>
__device__ __forceinline__ void ppo_update_body(const UpdArgs& a, const int wg) {
  unsigned long long pacc[NPHASE] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define SPO_STAMP(i)                                           \
  if (PROF) {                                                  \
    const unsigned long long _t = __builtin_readcyclecounter(); \
    pacc[i] += _t - tprev; tprev = _t;                         \
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using U = UpdLds<KIN>;
  using L = NetLds<KIN>;
  constexpr int NT1 = KIN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  // Placement hint (speed only, never correctness): workgroup b is observed to land on XCD b % 8, so the
  // persistent form launches 8*(n-1)+1 blocks and only blocks 0, 8, 16 work -- the networks then share one
  // XCD's L2 and the per-step granule exchange is ~0.1-0.3 us faster.  Any other placement is just slower.
  float* const st_m = a.adam_m; float* const st_v = a.adam_v;
  const int net = a.first_net + wg;
  const int D = a.cfg.obs_dim, A = a.cfg.act_dim, B = a.cfg.batch;
  const NetGeom g = net_geom(D, A, net);
  const bool is_actor = (net == 2);
  const int OUT = g.OUT;
  const int ls_off = g.off - A;          // actor only
  float* const red = lds + U::RED;
  stage_net<KIN>(a.theta, g, lds, tid, 256);
  if (is_actor && tid < A) red[128 + tid] = a.theta[ls_off + tid];     // log_std mirror
  if (XR) {
    static_assert(U::RED % 2 == 0 && RED_FLOATS >= 160, "8-byte aligned pointer table in red[144..159]");
    if (tid == 0) red[112] = 0.f;                                      // cross-rank "stop waiting" word
    if (tid < XR_MAX_WORLD)
      reinterpret_cast<unsigned long long*>(red + 144)[tid] = reinterpret_cast<unsigned long long>(a.xr_region[tid]);
  }
  __syncthreads();

  // ---- ownership (C layout of the weight-gradient tiles) and optimiser state in registers
  const int orow = 16 * wave + 4 * q;        // + r : row of W1/W2 tiles
  f4 mW1[NT1], vW1[NT1], mW2[4], vW2[4], mW3, vW3, mls, vls;
  float mb1 = 0, vb1 = 0, mb2 = 0, vb2 = 0, mb3 = 0, vb3 = 0;
  const bool own_b = (q == 0);
  const bool own_b3 = (wave == 0 && q == 0 && j < OUT);
  const bool own_ls = is_actor && wave == 0 && j == 0;
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) { mW1[nt] = f4{0.f, 0.f, 0.f, 0.f}; vW1[nt] = mW1[nt]; }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { mW2[nt] = f4{0.f, 0.f, 0.f, 0.f}; vW2[nt] = mW2[nt]; }
  mW3 = vW3 = mls = vls = f4{0.f, 0.f, 0.f, 0.f};
  if (PERSIST) {
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * nt + j;
        const int idx = g.w1() + (orow + r) * D + i;
        mW1[nt][r] = i < D ? a.adam_m[idx] : 0.f;
        vW1[nt][r] = i < D ? a.adam_v[idx] : 0.f;
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = g.w2() + (orow + r) * HID + 16 * nt + j;
        mW2[nt][r] = a.adam_m[idx];
        vW2[nt][r] = a.adam_v[idx];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * q + r;
      const int idx = g.w3() + o * HID + 16 * wave + j;
      mW3[r] = o < OUT ? a.adam_m[idx] : 0.f;
      vW3[r] = o < OUT ? a.adam_v[idx] : 0.f;
      mls[r] = (is_actor && o < A) ? a.adam_m[ls_off + o] : 0.f;
      vls[r] = (is_actor && o < A) ? a.adam_v[ls_off + o] : 0.f;
    }
    mb1 = a.adam_m[g.b1() + 16 * wave + j]; vb1 = a.adam_v[g.b1() + 16 * wave + j];
    mb2 = a.adam_m[g.b2() + 16 * wave + j]; vb2 = a.adam_v[g.b2() + 16 * wave + j];
    if (j < OUT) { mb3 = a.adam_m[g.b3() + j]; vb3 = a.adam_v[g.b3() + j]; }
  }

  const float b1c = a.cfg.beta1, b2c = a.cfg.beta2, eps = a.cfg.adam_eps;
  double pw1 = is_actor ? a.pow_b1_actor : a.pow_b1, pw2 = is_actor ? a.pow_b2_actor : a.pow_b2;
  const float lr = is_actor ? a.cfg.lr_actor : a.cfg.lr_critic;
  const float l2 = (!is_actor && a.cfg.use_critic_norm) ? a.cfg.l2_coef : 0.f;
  const float vcoef = (net == 0 && a.cfg.use_value_coefficient) ? 2.f : 1.f;
  const float clip_lo = 1.f - a.cfg.clip, clip_hi = 1.f + a.cfg.clip;
  float stale_sq = a.stale_io ? *a.stale_io : a.stale_sq;
  const float* tgt = (net == 0) ? a.tgt_r : a.tgt_c;

  const int64_t nsteps = PERSIST ? (a.M + B - 1) / B : 1;
  const int nhalf = (B + 63) / 64;
  const int64_t nchunks = nsteps * nhalf;
  const int mycol = 16 * wave + j;

  // position in perm[] of my column for (step s, half h) (clamped to a valid entry when masked)
  auto perm_pos = [&](int64_t s, int h) -> int64_t {
    const int64_t base = PERSIST ? s * B : 0;
    const int64_t rem = PERSIST ? (a.M - base) : a.M;
    const int ncols = (int)(rem < B ? rem : B);
    const int col = 64 * h + mycol;
    return base + (col < ncols ? col : 0);
  };
  auto fetch = [&](int64_t smp, ColData<NT1>& cd) {
    load_obs_tiles_raw<KIN>(a.obs + smp * D, D, q, cd.x);
    if (!is_actor) {
      cd.t0 = tgt[smp]; cd.t1 = 0.f; cd.actv = f4{0.f, 0.f, 0.f, 0.f};
    } else {
      cd.t0 = a.logp_old[smp]; cd.t1 = a.adv[smp];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        cd.actv[r] = a.act[smp * A + (ai < A ? ai : 0)];          // unconditional loads; pads selected at pick-up
        if (AMODE == 1) cd.omv[r] = a.old_mean[smp * A + (ai < A ? ai : 0)];
      }
    }
  };
  auto settle_prefetch = [&](ColData<NT1>& cd) {
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) cd.x[nt][e] = pinned(cd.x[nt][e]);
    mask_obs_tiles<KIN>(D, q, cd.x);
    cd.t0 = pinned(cd.t0);
    if (is_actor) {
      cd.t1 = pinned(cd.t1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        const float av = pinned(cd.actv[r]);
        cd.actv[r] = ai < A ? av : 0.f;
        if (AMODE == 1) {
          const float ov = pinned(cd.omv[r]);
          cd.omv[r] = ai < A ? ov : 0.f;
        }
      }
    }
  };

  // software pipeline: sample index two chunks ahead, column data one chunk ahead
  ColData<NT1> nxt;
  int smp1 = 0;                // sample index of chunk c+1 as loaded (widened where it is used, one step later)
  fetch((int64_t)a.perm[perm_pos(0, 0)], nxt);
  // (s1,h1) / (s2,h2): step and half of chunk c+1 / c+2, advanced without divisions
  int64_t s_cur = 0, s1 = (nhalf > 1) ? 0 : 1, s2;
  int h_cur = 0, h1n = (nhalf > 1) ? 1 : 0, h2n;
  h2n = h1n + 1; s2 = s1;
  if (h2n == nhalf) { h2n = 0; s2 = s1 + 1; }
  if (nchunks > 1) smp1 = a.perm[perm_pos(s1, h1n)];

  f4 aW1[NT1], aW2[4], aW3, dls;
  float db1 = 0.f, db2 = 0.f, db3 = 0.f, lsum = 0.f;
  float iso[4] = {1.f, 1.f, 1.f, 1.f};      // 1 / sigma_old for my 4 action rows (KL-penalty loss)
  if (AMODE == 1 && is_actor) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ai = 4 * q + r;
      iso[r] = ai < A ? 1.f / a.old_std[ai] : 1.f;
    }
  }

  for (int64_t c = 0; c < nchunks; ++c) {
    const int64_t s = s_cur;
    const int h = h_cur;
    const bool first_half = (h == 0), last_half = (h == nhalf - 1);
    const int64_t base = PERSIST ? s * B : 0;
    const int64_t rem = PERSIST ? (a.M - base) : a.M;
    const int ncols = (int)(rem < B ? rem : B);
    const float inv_n = 1.f / (float)(PERSIST ? ncols : (int)a.mean_count);
    const bool cv = (64 * h + mycol) < ncols;

    if (PROF) tprev = __builtin_readcyclecounter();
    ColData<NT1> cur = nxt;
    settle_prefetch(cur);
    const int smp_next = pinned(smp1);
    const int64_t pos2 = perm_pos(s2, h2n);
    s_cur = s1; h_cur = h1n; s1 = s2; h1n = h2n;
    h2n += 1;
    if (h2n == nhalf) { h2n = 0; s2 += 1; }

    if (first_half) {
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) aW1[nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) aW2[nt] = f4{0.f, 0.f, 0.f, 0.f};
      aW3 = dls = f4{0.f, 0.f, 0.f, 0.f};
      db1 = db2 = db3 = lsum = 0.f;
    }

    // std = exp(log_std) from the LDS mirror of log_std (a parameter: changes every step)
    float ivar[4], lsd[4], amask[4], vrat[4], lvrat[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ai = 4 * q + r;
      const bool on = is_actor && ai < A;
      const float lsv = on ? red[128 + ai] : 0.f;
      const float sdv = __expf(lsv);                        // std = exp(log_std)
      amask[r] = on ? 1.f : 0.f;
      if (AMODE == 1) {
        const float sr = sdv * iso[r];                      // kl_normal_normal: var_ratio = (p.scale / q.scale)^2
        vrat[r] = sr * sr;
        lvrat[r] = logf(vrat[r]);
      }
      ivar[r] = __builtin_amdgcn_rcpf(sdv * sdv);
      lsd[r] = on ? lsv + LOG_SQRT_2PI : 0.f;               // log(scale) = log_std (to 1 ulp) + log(sqrt(2 pi))
    }

    // ---- stage x as [feature][batch] right away (frees the registers after layer 1)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[U::XT + (16 * nt + 4 * q + e) * LDB + mycol] = cur.x[nt][e];

    SPO_STAMP(0)
    f4 h1[4], h2[4];
    const f4 o = net_forward<KIN>(lds, cur.x, h1, h2, j, q);
    // Prefetch: column data of chunk c+1 (its sample index was loaded one chunk ago), THEN the sample
    // index of chunk c+2 -- in this order, so the in-order vmcnt wait on the old index never covers
    // a load issued in this iteration.
    if (c + 1 < nchunks) fetch((int64_t)smp_next, nxt);
    if (c + 2 < nchunks) smp1 = a.perm[pos2];
    SPO_STAMP(1)

    // ---- loss and d(loss)/d(output), C layout (rows = output unit 4q+r, col = batch)
    f4 dO = {0.f, 0.f, 0.f, 0.f};
    if (!is_actor) {
      // mse_loss(critic(obs), target)  (ppo_lag.py:307-309)
      const float diff = o[0] - cur.t0;
      const float lm = (q == 0 && cv) ? 1.f : 0.f;
      lsum = fmaf(lm * diff, diff, lsum);
      dO[0] = lm * (2.f * diff * inv_n);
    } else if (AMODE == 1) {
      float lp = 0.f, klp = 0.f, dif[4], dm[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dif[r] = cur.actv[r] - o[r];
        lp += -(dif[r] * dif[r]) * (0.5f * ivar[r]) - lsd[r];
        dm[r] = (o[r] - cur.omv[r]) * iso[r];        // (loc_p - loc_q) / scale_q ; pad rows: 0
        klp += amask[r] * (0.5f * (vrat[r] + dm[r] * dm[r] - 1.f - lvrat[r]));
      }
      lp = quad_row_sum(lp);
      const float kl = quad_row_sum(klp);            // .sum(-1, keepdim=True)
      const float ind = (kl <= a.kl_bound) ? 1.f : 0.f;
      const float cnt = wave_sum_lane63((q == 0 && cv) ? ind : 0.f);
      if (lane == 63) red[104 + wave] = cnt;
      __syncthreads();                               // actor workgroup only (block-uniform branch)
      const float frac = ((red[104] + red[105]) + (red[106] + red[107])) * inv_n;
      const float adv = cur.t1;
      const float ratio = __expf(lp - cur.t0);
      const float pg = a.pg_coef * frac;
      const float dlp = cv ? -(pg * adv * ratio) * inv_n : 0.f;
      const float wk = cv ? ind * inv_n : 0.f;
      lsum += ((q == 0 && cv) ? 1.f : 0.f) * (pg * ratio * adv - ind * kl);     // loss = -mean(this)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = dif[r] * ivar[r];
        dO[r] = fmaf(dlp, z, wk * dm[r] * iso[r]);
        dls[r] += amask[r] * fmaf(dlp, dif[r] * z - 1.f, wk * (vrat[r] - 1.f));
      }
    } else {
      float lp = 0.f, dif[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dif[r] = cur.actv[r] - o[r];                 // pad rows: 0 - 0
        lp += -(dif[r] * dif[r]) * (0.5f * ivar[r]) - lsd[r];
      }
      lp = quad_row_sum(lp);                                               // .sum(dim=-1) over the 4 k-slot rows
      const float adv = cur.t1;
      const float ratio = __expf(lp - cur.t0);                             // ppo_lag.py:317 (argument ~0: 1e-7 rel)
      const float rc = fminf(fmaxf(ratio, clip_lo), clip_hi);              // torch.clamp
      const float s1 = ratio * adv, s2 = rc * adv;
      const bool inr = (ratio >= clip_lo) && (ratio <= clip_hi);
      // backward of torch.min(s1, s2): ties split the gradient; clamp passes it inside [lo,hi]
      float gr;
      if (s1 < s2) gr = adv;
      else if (s1 > s2) gr = inr ? adv : 0.f;
      else gr = 0.5f * adv + (inr ? 0.5f * adv : 0.f);
      const float dlp = cv ? -(gr * ratio) * inv_n : 0.f;                  // loss_pi = -mean(min(...))
      lsum += ((q == 0 && cv) ? 1.f : 0.f) * fminf(s1, s2);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = dif[r] * ivar[r];
        dO[r] = dlp * z;                            // pad rows: z == 0
        dls[r] = fmaf(dlp * amask[r], dif[r] * z - 1.f, dls[r]);
      }
    }

    // ---- backward through the MLP (transposed chaining, weights read as columns).  The A operands
    //      (one LDS word per MFMA) are fetched a whole k-group ahead of the MFMAs that consume them.
    f4 dz2[4], dz1[4];
    {
      f4 acc[4];
      float w3c[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) w3c[r][mt] = lds[L::W3 + (4 * q + r) * LDH + 16 * mt + j];
      float w2c[2][4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) w2c[0][r][mt] = lds[L::W2 + (4 * q + r) * LDH + 16 * mt + j];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma4(w3c[r][mt], dO[r], acc[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dz2[mt][r] = acc[mt][r] * fmaf(-h2[mt][r], h2[mt][r], 1.f);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt + 1 < 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
              w2c[(nt + 1) & 1][r][mt] = lds[L::W2 + (16 * (nt + 1) + 4 * q + r) * LDH + 16 * mt + j];
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

    SPO_STAMP(2)
    // ---- stage [feature][batch] images for the weight-gradient GEMMs
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = (16 * mt + 4 * q + r) * LDB + mycol;
        lds[U::H1T + f] = h1[mt][r];
        lds[U::H2T + f] = h2[mt][r];
        if (!U::SPLIT) lds[U::DZ1T + f] = dz1[mt][r];
        lds[U::DZ2T + f] = dz2[mt][r];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[U::DOT + (4 * q + r) * LDB + mycol] = dO[r];
    if (last_half) {
      // per-wave partials of the scalar reductions ride on the same barrier
      const float ls = wave_sum_lane63(lsum);
      if (lane == 63) red[wave] = ls;
      if (is_actor) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = row_sum_lane15(dls[r]);           // over the wave's 16 columns
          if (j == 15) red[16 + wave * 16 + 4 * q + r] = t;
        }
      }
    }
    SPO_STAMP(3)
    __syncthreads();
    SPO_STAMP(4)

    // ---- dW[o][i] += sum_b dZ[b][o] * Hprev[b][i]; wave w owns rows 16w..16w+15.
    //      A tiles (dZ^T rows) are read first; B tiles are read one k-group ahead of their MFMAs.
    {
      f4 az2[4], az3[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        az2[r4] = *reinterpret_cast<const f4*>(lds + U::DZ2T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
        az3[r4] = *reinterpret_cast<const f4*>(lds + U::DOT + j * LDB + 16 * r4 + 4 * q);
      }
      f4 az1[4];
      if (!U::SPLIT) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          az1[r4] = *reinterpret_cast<const f4*>(lds + U::DZ1T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
      }
      // layer 2 then layer 3
      f4 bh[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        bh[0][nt] = *reinterpret_cast<const f4*>(lds + U::H1T + (16 * nt + j) * LDB + 4 * q);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        if (r4 + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            bh[(r4 + 1) & 1][nt] = *reinterpret_cast<const f4*>(lds + U::H1T + (16 * nt + j) * LDB + 16 * (r4 + 1) + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) aW2[nt] = mfma4(az2[r4][e], bh[r4 & 1][nt][e], aW2[nt]);
      }
      f4 b3[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        b3[r4] = *reinterpret_cast<const f4*>(lds + U::H2T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
      f4 w3a = {0.f, 0.f, 0.f, 0.f}, w3b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r4 = 0; r4 < 4; r4 += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          w3a = mfma4(az3[r4][e], b3[r4][e], w3a);
          w3b = mfma4(az3[r4 + 1][e], b3[r4 + 1][e], w3b);
        }
      aW3 += w3a + w3b;
      float rs2 = 0.f, rs3 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        rs2 += (az2[r4][0] + az2[r4][1]) + (az2[r4][2] + az2[r4][3]);
        rs3 += (az3[r4][0] + az3[r4][1]) + (az3[r4][2] + az3[r4][3]);
      }
      db2 += rs2; db3 += rs3;
      if (U::SPLIT) {
        // second sub-phase: dZ1^T goes into the buffer dZ2^T just vacated
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[U::DZ1T + (16 * mt + 4 * q + r) * LDB + mycol] = dz1[mt][r];
        __syncthreads();
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          az1[r4] = *reinterpret_cast<const f4*>(lds + U::DZ1T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
      }
      // layer 1
      f4 bx[2][NT1];
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
        bx[0][nt] = *reinterpret_cast<const f4*>(lds + U::XT + (16 * nt + j) * LDB + 4 * q);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        if (r4 + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt)
            bx[(r4 + 1) & 1][nt] = *reinterpret_cast<const f4*>(lds + U::XT + (16 * nt + j) * LDB + 16 * (r4 + 1) + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) aW1[nt] = mfma4(az1[r4][e], bx[r4 & 1][nt][e], aW1[nt]);
      }
      float rs1 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) rs1 += (az1[r4][0] + az1[r4][1]) + (az1[r4][2] + az1[r4][3]);
      db1 += rs1;
    }
    if (!last_half) {
      __syncthreads();         // the next half overwrites the staged images
      continue;
    }

    SPO_STAMP(5)
    // =================== end of the minibatch: gradients complete ===================
    db1 = quad_row_sum(db1);
    db2 = quad_row_sum(db2);
    db3 = quad_row_sum(db3);
    const float loss_data = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
    if (is_actor) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        dls[r] = (red[16 + ai] + red[32 + ai]) + (red[48 + ai] + red[64 + ai]);     // 0 on pad rows
      }
    }

    if (XR) {
      // ---- gradient of the global minibatch = mean over ranks of the local-minibatch gradients
      constexpr int NV = NT1 + 7;
      f4 pk[NV];
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) pk[nt] = aW1[nt];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) pk[NT1 + nt] = aW2[nt];
      pk[NT1 + 4] = aW3; pk[NT1 + 5] = f4{db1, db2, db3, 0.f}; pk[NT1 + 6] = dls;
      const u64* const xr_tab = reinterpret_cast<const u64*>(red + 144);
      const unsigned gtag = a.xr_step0 + (unsigned)s + 1u;
      if (XR == 4)      // one-grid split form, all workgroups on one XCD: the library's cached pair of regions (table slots 2, 3)
        xr_allreduce_rd16<NV, SPO_XR16_VB, true>(xr_tab + 2, a.xr_rank, a.xr_world, net, tid, gtag, pk, red + 112, a.err);
      else if (XR == 2 && SPO_XR_PACK16)
        xr_allreduce_rd16<NV>(xr_tab, a.xr_rank, a.xr_world, net, tid, gtag, pk, red + 112, a.err);
      else if (XR == 2)
        xr_allreduce_rd<NV>(xr_tab, a.xr_rank, a.xr_world, net, tid, gtag, pk, red + 112, a.err);
      else if (SPO_XR_PACK16)
        xr_allreduce_rs16<NV>(xr_tab, a.xr_rank, a.xr_world, net, tid, gtag, pk, red + 112, a.err);
      else
        xr_allreduce<NV>(xr_tab, a.xr_rank, a.xr_world, net, tid, gtag, pk, red + 112, a.err);
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) aW1[nt] = pk[nt];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) aW2[nt] = pk[NT1 + nt];
      aW3 = pk[NT1 + 4]; db1 = pk[NT1 + 5][0]; db2 = pk[NT1 + 5][1]; db3 = pk[NT1 + 5][2]; dls = pk[NT1 + 6];
    }

    // ---- L2 regulariser of the critics (weights AND biases, ppo_lag.py:310-314), grad norm
    float gsq = 0.f, psq = 0.f;
    const float l2x2 = 2.f * l2;
    f4 pW1[NT1], pW2[4], pW3;
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pW1[nt][r] = lds[L::W1 + (orow + r) * L::LD1 + 16 * nt + j];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pW2[nt][r] = lds[L::W2 + (orow + r) * LDH + 16 * nt + j];
#pragma unroll
    for (int r = 0; r < 4; ++r) pW3[r] = lds[L::W3 + (4 * q + r) * LDH + 16 * wave + j];
    const float pb1 = lds[L::B1 + 16 * wave + j], pb2 = lds[L::B2 + 16 * wave + j], pb3 = lds[L::B3 + j];
    f4 pls = {0.f, 0.f, 0.f, 0.f};
    if (is_actor) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pls[r] = red[128 + 4 * q + r];
    }
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = pW1[nt][r];
        const float gg = vcoef * fmaf(l2x2, p, aW1[nt][r]);                  // pad columns: 0
        aW1[nt][r] = gg; gsq = fmaf(gg, gg, gsq); psq = fmaf(p, p, psq);     // pad columns hold p == 0
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = pW2[nt][r];
        const float gg = vcoef * fmaf(l2x2, p, aW2[nt][r]);
        aW2[nt][r] = gg; gsq = fmaf(gg, gg, gsq); psq = fmaf(p, p, psq);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = pW3[r];
      const float gg = vcoef * fmaf(l2x2, p, aW3[r]);                        // pad rows: 0
      aW3[r] = gg; gsq = fmaf(gg, gg, gsq); psq = fmaf(p, p, psq);           // pad rows hold p == 0
    }
    {
      // biases and log_std are replicated across lanes (every q-lane of row j holds the same sums);
      // each replica runs the same Adam, only one of them counts towards the norms.
      db1 = vcoef * fmaf(l2x2, pb1, db1); db2 = vcoef * fmaf(l2x2, pb2, db2); db3 = vcoef * fmaf(l2x2, pb3, db3);
      const float wb = own_b ? 1.f : 0.f, wb3 = (wave == 0 && q == 0) ? 1.f : 0.f, wls = own_ls ? 1.f : 0.f;
      gsq += wb * (db1 * db1 + db2 * db2) + wb3 * (db3 * db3);
      psq += wb * (pb1 * pb1 + pb2 * pb2) + wb3 * (pb3 * pb3);
#pragma unroll
      for (int r = 0; r < 4; ++r) gsq = fmaf(wls * dls[r], dls[r], gsq);
    }
    gsq = wave_sum_lane63(gsq);
    psq = wave_sum_lane63(psq);
    if (lane == 63) { red[4 + wave] = gsq; red[8 + wave] = psq; }
    SPO_STAMP(6)
    __syncthreads();
    const float my_sq = (red[4] + red[5]) + (red[6] + red[7]);
    if (tid == 0) {
      const float loss = is_actor ? -loss_data : loss_data + l2 * ((red[8] + red[9]) + (red[10] + red[11]));
      a.losses[s * 3 + net] = loss;
    }

    if (!PERSIST) {
      // split form: emit the flat gradient (reference parameter order) and stop
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * nt + j < D) a.flat_grad[g.w1() + (orow + r) * D + 16 * nt + j] = aW1[nt][r];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.flat_grad[g.w2() + (orow + r) * HID + 16 * nt + j] = aW2[nt][r];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * q + r < OUT) a.flat_grad[g.w3() + (4 * q + r) * HID + 16 * wave + j] = aW3[r];
      if (own_b) { a.flat_grad[g.b1() + 16 * wave + j] = db1; a.flat_grad[g.b2() + 16 * wave + j] = db2; }
      if (own_b3) a.flat_grad[g.b3() + j] = db3;
      if (own_ls) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (4 * q + r < A) a.flat_grad[ls_off + 4 * q + r] = dls[r];
      }
      return;
    }

    // ---- joint clip_grad_norm_ over all networks: publish ||g||^2 (one granule per workgroup) ...
    unsigned long long* const grow = a.slots + (s & 1) * 4;
    const unsigned tag = (unsigned)(s + 1);
    if (tid == 0) st_granule(grow + wg, ((unsigned long long)tag << 32) | __float_as_uint(my_sq));
    // first look at the peers' granules now: the load's fabric latency (~0.6 us) hides under the speculative Adam
    unsigned long long peek = 0;
    if (tid < a.n_nets) peek = ld_granule(grow + tid);

    // ---- ... and run Adam SPECULATIVELY with clip coefficient 1 while the granules are in flight
    //      (max_grad_norm = 40 almost never clips).  The moments are backed up so that a clipped step is
    //      redone exactly from the old state; parameters are only read by other waves after the final barrier.
    pw1 *= (double)b1c; pw2 *= (double)b2c;
    float step_size, inv_bc2s;
    adam_scalars(lr, pw1, pw2, step_size, inv_bc2s);
    constexpr bool SPEC = SPO_SPECULATIVE_ADAM && (KIN <= 64);   // the 128-wide variant has no registers to spare for the backups
    f4 omW1[NT1], ovW1[NT1], omW2[4], ovW2[4];
    f4 omW3 = mW3, ovW3 = vW3, omls = mls, ovls = vls;
    float omb1 = mb1, ovb1 = vb1, omb2 = mb2, ovb2 = vb2, omb3 = mb3, ovb3 = vb3;
    if (SPEC) {
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) { omW1[nt] = mW1[nt]; ovW1[nt] = vW1[nt]; }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { omW2[nt] = mW2[nt]; ovW2[nt] = vW2[nt]; }
    }
    auto run_adam = [&](const float coef) {
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)                                         // pad columns stay exactly 0
          SPO_ADAM(lds[L::W1 + (orow + r) * L::LD1 + 16 * nt + j], pW1[nt][r], aW1[nt][r] * coef, mW1[nt][r], vW1[nt][r])
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          SPO_ADAM(lds[L::W2 + (orow + r) * LDH + 16 * nt + j], pW2[nt][r], aW2[nt][r] * coef, mW2[nt][r], vW2[nt][r])
#pragma unroll
      for (int r = 0; r < 4; ++r)                                           // pad rows stay exactly 0
        SPO_ADAM(lds[L::W3 + (4 * q + r) * LDH + 16 * wave + j], pW3[r], aW3[r] * coef, mW3[r], vW3[r])
      // replicated state (biases, log_std): all replicas compute identical values and store them to the same word
      float np1, np2, np3;
      SPO_ADAM(np1, pb1, db1 * coef, mb1, vb1)
      SPO_ADAM(np2, pb2, db2 * coef, mb2, vb2)
      SPO_ADAM(np3, pb3, db3 * coef, mb3, vb3)
      lds[L::B1 + 16 * wave + j] = np1;
      lds[L::B2 + 16 * wave + j] = np2;
      lds[L::B3 + j] = np3;
      if (is_actor) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float nl;
          SPO_ADAM(nl, pls[r], dls[r] * coef, mls[r], vls[r])
          red[128 + 4 * q + r] = nl;
        }
      }
    };
    if (SPEC) run_adam(1.f);
    SPO_STAMP(7)

    // ---- collect the other workgroups' ||g||^2 (already there in the common case)
    if (tid < a.n_nets) {
      unsigned long long v = peek;
      unsigned spins = 0;
      while ((unsigned)(v >> 32) != tag) {
        if (++spins > (1u << 24)) { *a.err = 1; break; }          // bounded: never hang the GPU
        __builtin_amdgcn_s_sleep(1);
        v = ld_granule(grow + tid);
      }
      red[96 + tid] = __uint_as_float((unsigned)v);
    }
    __syncthreads();
    float total_sq = stale_sq;
    for (int k = 0; k < a.n_nets; ++k) total_sq += red[96 + k];
    const float norm = sqrtf(total_sq);
    float coef = a.cfg.max_grad_norm / (norm + 1e-6f);                // clip_grad_norm_ (torch): eps 1e-6
    coef = coef > 1.f ? 1.f : coef;
    stale_sq *= coef * coef;                                         // stale actor grads are scaled in place too
    if (!SPEC) {
      run_adam(coef);
    } else if (coef != 1.f) {
      // clipped step (uniform across the grid: every workgroup sees the same total): redo from the old state
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) { mW1[nt] = omW1[nt]; vW1[nt] = ovW1[nt]; }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { mW2[nt] = omW2[nt]; vW2[nt] = ovW2[nt]; }
      mW3 = omW3; vW3 = ovW3; mls = omls; vls = ovls;
      mb1 = omb1; vb1 = ovb1; mb2 = omb2; vb2 = ovb2; mb3 = omb3; vb3 = ovb3;
      run_adam(coef);
    }
    SPO_STAMP(8)
    __syncthreads();
    SPO_STAMP(9)
  }  // chunks
  if (PROF && a.prof && tid == 0)
    for (int i = 0; i < NPHASE; ++i) a.prof[wg * NPHASE + i] = pacc[i];
#undef SPO_STAMP

  if (PERSIST) {
    // ---- write back parameters and optimiser state (flat reference order)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * nt + j;
        if (i < D) {
          const int idx = g.w1() + (orow + r) * D + i;
          a.theta[idx] = lds[L::W1 + (orow + r) * L::LD1 + i];
          SPO_ST_MV(idx, mW1[nt][r], vW1[nt][r])
        }
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = g.w2() + (orow + r) * HID + 16 * nt + j;
        a.theta[idx] = lds[L::W2 + (orow + r) * LDH + 16 * nt + j];
        SPO_ST_MV(idx, mW2[nt][r], vW2[nt][r])
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * q + r;
      if (o < OUT) {
        const int idx = g.w3() + o * HID + 16 * wave + j;
        a.theta[idx] = lds[L::W3 + o * LDH + 16 * wave + j];
        SPO_ST_MV(idx, mW3[r], vW3[r])
      }
      if (own_ls && o < A) {
        a.theta[ls_off + o] = red[128 + o];
        SPO_ST_MV(ls_off + o, mls[r], vls[r])
      }
    }
    if (own_b) {
      const int o = 16 * wave + j;
      a.theta[g.b1() + o] = lds[L::B1 + o]; SPO_ST_MV(g.b1() + o, mb1, vb1)
      a.theta[g.b2() + o] = lds[L::B2 + o]; SPO_ST_MV(g.b2() + o, mb2, vb2)
    }
    if (own_b3) { a.theta[g.b3() + j] = lds[L::B3 + j]; SPO_ST_MV(g.b3() + j, mb3, vb3) }
    if (tid == 0 && wg == 0 && a.stale_io) {
      // every workgroup read the old value before its first step; they all finish after the last exchange
      *a.stale_io = stale_sq;
    }
  }
}

template <int KIN, bool PERSIST, bool PROF = false, int AMODE = 0, int XR = 0>
__global__ __launch_bounds__(256, 1) void ppo_update_kernel(UpdArgs a) {
  if (PERSIST && (blockIdx.x & 7)) return;                     // placement hint, see ppo_update_body
  ppo_update_body<KIN, PERSIST, PROF, AMODE, XR>(a, PERSIST ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
}

// One-grid split form (critic fit at batch 128 on one GPU): BOTH "ranks" of a two-way split of the minibatch live in ONE
// launch -- workgroups [0, n_nets) run rank 0's arguments, [n_nets, 2 n_nets) rank 1's (own replica, rows, outputs, granules,
// error word) -- so they are co-resident by construction (one grid of four workgroups) instead of by the luck of two
// streams landing on two hardware queues.  Recursive doubling at world 2: one hand-off per step.
// Placement census of the one-grid split forms: true when every workgroup of the launch reports the same XCC id (words any
// placement delivers: agent-scope atomics) and the host supplied the cached pair of regions.
__device__ __forceinline__ bool split_census(const UpdArgs& a0, const int wg) {
  extern __shared__ __attribute__((aligned(16))) float lds_split[];       // (the bodies' image: its first word serves the census)
  int& s_other = *reinterpret_cast<int*>(lds_split);
  bool local = false;
  if (a0.xr_region[2] != nullptr && a0.xr_census != nullptr) {
    const int nwg = 2 * a0.n_nets, tid = threadIdx.x;
    const unsigned myx = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID
    if (tid == 0) {
      s_other = 0;
      __hip_atomic_store(a0.xr_census + wg, ((unsigned long long)a0.xr_census_tag << 32) | myx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < nwg) {
      unsigned long long v = __hip_atomic_load(a0.xr_census + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while ((unsigned)(v >> 32) != a0.xr_census_tag) {
        if (++spins > (1u << 22)) { s_other = 1; break; }              // (a census that never completes: the uncached form)
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(a0.xr_census + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((unsigned)v != myx) s_other = 1;
    }
    __syncthreads();
    // Second round: every workgroup publishes its OWN verdict and takes the AND of all of them, so a census that timed
    // out in one workgroup only cannot leave that workgroup on the uncached body while its peers run the L2 body (each would spin on
    // words the other never writes: a clean error instead -- and, with everybody resident, a clean agreement on the uncached form)
    if (tid == 0)
      __hip_atomic_store(a0.xr_census + 16 + wg, ((unsigned long long)a0.xr_census_tag << 32) | (s_other == 0 ? 1u : 0u),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid < nwg) {
      unsigned long long v = __hip_atomic_load(a0.xr_census + 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while ((unsigned)(v >> 32) != a0.xr_census_tag) {
        if (++spins > (1u << 22)) { s_other = 1; *a0.err = 3; break; }   // (a peer workgroup is not resident: the launch cannot work)
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(a0.xr_census + 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((unsigned)v != 1u) s_other = 1;
    }
    __syncthreads();
    local = (s_other == 0);
    __syncthreads();
  }
  return local;
}

// Round 5: when a placement census finds all of them on ONE XCD (the grid of 8-block strides asks for it, nothing promises it) and
// the host supplied a cached pair of regions (xr_region[2], [3]), the exchange runs through that XCD's L2 (XR = 4) instead of
// through the uncached regions.
template <int KIN>
__global__ __launch_bounds__(256, 1) void ppo_update_split_kernel(UpdArgs a0, UpdArgs a1) {
  if (blockIdx.x & 7) return;
  const int wg = (int)(blockIdx.x >> 3);
  const bool local = split_census(a0, wg);
  if (local) {
    if (wg < a0.n_nets) ppo_update_body<KIN, true, false, 0, 4>(a0, wg);
    else ppo_update_body<KIN, true, false, 0, 4>(a1, wg - a0.n_nets);
  } else {
    if (wg < a0.n_nets) ppo_update_body<KIN, true, false, 0, 2>(a0, wg);
    else ppo_update_body<KIN, true, false, 0, 2>(a1, wg - a0.n_nets);
  }
}

// =====================================================================================================================
// Main + helper waves: ppo_update_h_kernel (512 threads per network).
//
// In ppo_update_kernel a fifth of every step is optimiser work that uses no matrix pipe at all (L2 terms and norms, Adam,
// the wait for the other networks' norms: ~6.4 k of ~29.7 k cycles), and next to nothing can be put beside it because the
// next forward pass needs the new weights.  Here the four MAIN waves (one per SIMD) run forward / loss / backward / weight
// gradients exactly as before but hold no optimiser state; four HELPER waves (the second wave of each SIMD) own the Adam
// moments and do the L2 terms, norms, Adam and the clip exchange -- layer by layer, BESIDE the main waves' MFMAs:
//
//   main   : ... stage images | dW1 -> G1 | dW2 -> G2 | dW3 -> G3 | next step: settle, x^T, L1, L2 | L3, loss, backward ...
//   helper :                  |    idle   | Adam W1,b1| Adam W2,b2| Adam W3,b3,log_std, norm exchange, clip check   |
//                         B_stage        Xa          Xb          Xc                                 Xd
//
// Weight gradients are computed in the order W1, W2, W3 so that the layer the next forward pass needs first is updated
// first; they reach the helper lane of the same lane number through LDS (G1 / G3 in the space left over by the images, G2
// in the dZ1^T image, which is dead once dW1 is done).  Adam always runs with clip coefficient 1 first (max_grad_norm
// almost never binds); the joint norm is known only after the last layer, by which time the main waves are already in the
// next forward pass.  If the clip turns out to be active, the helpers restore moments and parameters from their backup rows
// (global scratch, written fire-and-forget), redo Adam exactly with the coefficient, and raise a flag that makes the main
// waves repeat L1 / L2 of the step they had started (one extra barrier round; results identical to the unspeculated order).
// Every barrier sits in code common to both roles, so the two roles cannot disagree on the number of barriers.
// Debug / measurement counters of the main + helper kernel (spo_debug_update_counters): minibatch steps run, steps whose
// speculative update turned out clipped and was redone ("late"), steps clipped under the conservative protocol, steps run
// under the conservative protocol -- summed over launches by the first helper lane of network 0 at the end of a launch.
__device__ unsigned long long g_upd_counters[4];

#ifndef SPO_H_W3C_EARLY
#define SPO_H_W3C_EARLY 1     // 1: the W3 column operands of dO -> dZ2 are read BEFORE the loss arithmetic (their LDS latency hides under the
                              //    exp / select chain of the loss) instead of after it: 10.68 -> 10.58 us per step, bit-identical
#endif
#ifndef SPO_H_GATHER
// 0: the main waves' own prefetch pipeline (rounds 2-3).  1 / 2: the HELPER waves gather the next minibatch (sample index,
// observation rows, targets) and write its x^T image + column inputs between P1 and P3; the main waves pick the columns up with
// 16 LDS reads instead of the settle (pad selects, index widening, 16 x^T stores: 1.3 k cycles of a lone wave).  1 issues the
// row loads after B_stage (not back by P1: the helper stalls at the head of its critical stretch, 11.48 us per step), 2 right
// behind Xd (a whole backward pass to land): 10.68 against 10.84 us for 0, results bit-identical (profiles/r04/update_ab_gather.txt)
#define SPO_H_GATHER 2
#endif
template <int KIN>
struct UpdHLds {
  using U = UpdLds<KIN>;
  static_assert(!U::SPLIT, "main + helper form: obs_dim <= 64");
  static constexpr int NT1 = KIN / 16;
  static constexpr int G1 = U::SIZE;                        // [4 waves][NT1 tiles][64 lanes] float4
  static constexpr int G2 = U::DZ1T;                        // [4][4][64] float4 inside the dZ1^T image (16 384 <= 17 408 B)
  static constexpr int G3 = G1 + 4 * NT1 * 64 * 4;          // [4][64] float4: W3 tile
  static constexpr int GB = G3 + 4 * 64 * 4;                // [3][256] floats: db1, db2, db3 as the main lanes hold them
  static constexpr int XW = GB + 3 * 256;                   // 32 floats of helper-to-helper / helper-to-main words
  static constexpr int COLS = XW + 32 + 2 * XR_MAX_WORLD;   // (+ the ranks' exchange-region pointers, 8-byte aligned: XW is even)
  // COLS: [2 + OUTP][64] per-column inputs of the step the main waves are in -- target / logp_old, advantage, act[0..15] --
  // gathered by the HELPER waves (SPO_H_GATHER) together with the x^T image
  static constexpr int SIZE = COLS + (2 + OUTP) * 64;
  static_assert(SIZE * 4 <= 163840, "main + helper form: 160 KB of LDS");
  static_assert(XW % 2 == 0, "pointer table alignment");
  static_assert(4 * 4 * 64 * 4 <= HID * LDB, "G2 must fit the dZ1^T image");
};
// words in XW (as 32-bit): [0..7] {tag, ||g||^2 partial} per helper wave, [8..15] {tag, sum p^2 partial},
//                          [16,17] {tag, clip coefficient}, [18] redo tag (step whose forward must be repeated)

// All-to-all form of the cross-rank gradient mean for the helper waves of ppo_update_h_kernel: ONE exchange round at any
// world size.  The tagged-word protocols above make the CONSUMER poll every float (8 bytes each): at 8 ranks that is
// 7 x 43 words per lane, far more than fits in registers, i.e. many dependent round trips to local memory.  Here a helper
// wave writes its float4 rows plainly (write-through, system scope) into slot [parity][source = me][network][row] of every
// peer, waits until its stores are acknowledged (s_waitcnt vmcnt(0)), and then raises ONE flag per peer (the global step
// count).  The consumer wave polls world - 1 flags, after which the rows are ordinary 16-byte loads that can be issued back
// to back: G rows x (world - 1) sources in flight at once, summed IN RANK ORDER (own contribution at position `me`), so all
// replicas add the same values in the same order and hold identical bits.  Cost of the round: one store acknowledgement
// plus one flag flight plus the streamed loads, at any world size.  Buffers are double-buffered by step parity as above.
// (the row base is wave-uniform; readfirstlane pins it to SGPRs, which the "s" operands of the accesses below require)
__device__ __forceinline__ f4* a2a_row(void* region, int par, int src, int net, int row) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(reinterpret_cast<char*>(region) + A2A_DATA_OFF) +
                               ((((size_t)par * XR_MAX_WORLD + src) * 3 + net) * A2A_ROWS + row) * 256 * sizeof(f4);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  return reinterpret_cast<f4*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned long long* a2a_flag(void* region, int par, int src, int net, int stage, int wave) {
  return reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(region) + A2A_FLAG_OFF) +
         ((((size_t)par * XR_MAX_WORLD + src) * 3 + net) * 2 + stage) * 4 + wave;
}
// (uniform row base in SGPRs + one 32-bit lane offset: no per-access 64-bit address registers)
__device__ __forceinline__ void a2a_store(const f4* row_base, unsigned lane_byte_off, const f4 v) {
#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass would reject the register constraints)
  // s_nop 4: the base may have just been written by a VALU instruction (v_readfirstlane / v_readlane of a spilled SGPR);
  // gfx9 needs 5 wait states between that and a VMEM instruction reading the SGPR, and the hazard recogniser does not
  // look inside inline assembly
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(lane_byte_off), "v"(v), "s"(row_base) : "memory");
#endif
}
// region pointer of rank `idx` from the LDS copy of the kernel-argument table (the table in SGPRs costs 16 of them for the
// whole kernel, and indexing it dynamically is not possible)
__device__ __forceinline__ void* a2a_region(const unsigned long long* tab, int idx) {
  return reinterpret_cast<void*>(tab[idx]);
}
// The world size R is a template parameter and a rank treats ITSELF like any other peer (it also writes its rows and its
// flag into its own region), so every loop below is a plain constant-trip loop: no rank-dependent control flow.
template <int R>
__device__ __forceinline__ void a2a_push_row(const UpdArgs& a, const unsigned long long* tab, int net, int hl, int par, int row,
                                             const f4 g) {
  const int me = a.xr_rank;
  if (a.xr_debug & 1) return;
#pragma unroll
  for (int r = 0; r < R; ++r) a2a_store(a2a_row(a2a_region(tab, r), par, me, net, row), (unsigned)hl * 16u, g);
}
// after the pushes of a stage: stores acknowledged, then the flag (lanes 0 .. R-1 of the wave: one destination each)
template <int R>
__device__ __forceinline__ void a2a_signal(const UpdArgs& a, const unsigned long long* tab, int net, int lane, int wave, int par,
                                           int stage, unsigned gtag) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  const int me = a.xr_rank;
  if (lane < R && !(a.xr_debug & 2))
    __hip_atomic_store(a2a_flag(a2a_region(tab, lane), par, me, net, stage, wave), (unsigned long long)gtag, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
// lane r of the wave waits for rank r's flag of this (stage, wave); bounded, wave-uniform outcome
template <int R>
__device__ __forceinline__ void a2a_wait(const UpdArgs& a, const unsigned long long* tab, int net, int lane, int wave, int par,
                                         int stage, unsigned gtag, volatile int* dead, int* err) {
  if (lane < R && *dead == 0 && !(a.xr_debug & 2)) {
    unsigned long long* const f = a2a_flag(a2a_region(tab, a.xr_rank), par, lane, net, stage, wave);
    unsigned spins = 0;
    while ((unsigned)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != gtag) {
      if (++spins > XR_SPIN_LIMIT) { *err = 2; *dead = 1; break; }             // bounded: never hang the GPU
      __builtin_amdgcn_s_sleep(1);
    }
  }
  // (the branch re-converges here: every lane continues only after the polling lanes have seen their flags)
}
// G rows at once: R x G loads in flight, one wait, sums in rank order; returns the MEAN over the ranks
template <int R, int G>
__device__ __forceinline__ void a2a_pull(const UpdArgs& a, const unsigned long long* tab, int net, int hl, int par,
                                         const int (&rows)[G], f4 (&g)[G]) {
  if (a.xr_debug & 4) return;
  void* const mine = a2a_region(tab, a.xr_rank);
  f4 x[G][R];
#pragma unroll
  for (int v = 0; v < G; ++v)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const f4* const base = a2a_row(mine, par, r, net, rows[v]);
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(x[v][r]) : "v"((unsigned)hl * 16u), "s"(base) : "memory");
#else
      x[v][r] = base[hl];
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  const float inv_world = 1.f / (float)R;
#pragma unroll
  for (int v = 0; v < G; ++v) {
#pragma unroll
    for (int r = 0; r < R; ++r) x[v][r] = pinned4(x[v][r]);              // the asm outputs are valid only after the wait
    f4 run = x[v][0];
#pragma unroll
    for (int r = 1; r < R; ++r) run += x[v][r];
    g[v] = run * inv_world;
  }
}

// N rows in groups of GS (fewer rows in flight at 8 ranks: G x R float4 must fit beside the optimiser state)
template <int R, int N, int C0 = 0>
__device__ __forceinline__ void a2a_pull_all(const UpdArgs& a, const unsigned long long* tab, int net, int hl, int par,
                                             const int (&rows)[N], f4 (&g)[N]) {
  constexpr int GS = (R >= 8) ? 2 : 3;
  if constexpr (C0 < N) {
    constexpr int G = (N - C0) < GS ? (N - C0) : GS;
    int rr[G];
    f4 gg[G];
#pragma unroll
    for (int v = 0; v < G; ++v) { rr[v] = rows[C0 + v]; gg[v] = g[C0 + v]; }
    a2a_pull<R, G>(a, tab, net, hl, par, rr, gg);
#pragma unroll
    for (int v = 0; v < G; ++v) g[C0 + v] = gg[v];
    a2a_pull_all<R, N, C0 + G>(a, tab, net, hl, par, rows, g);
  }
}

// XR: world size of the in-kernel data-parallel exchange (0 = none); separate instantiations per form: both exchange bodies
// together spill.
// XRD: form of the exchange on the helper waves -- 0 = flag-based all-to-all, 2 = recursive doubling with PACKED 16-byte words,
// one poll batch per stage (round 5: xr_rd16_flat; the 8-byte-word form of rounds 2-4 spilled 250 registers and is gone)
template <int KIN, bool PROF = false, int XR = 0, int XRD = 2>
__device__ __forceinline__ void ppo_update_h_body(const UpdArgs& a, const int wg) {
  // PROF: wave 0 of each role of the LAST workgroup accumulates shader cycles per interval (a.prof rows 0 = main, 1 = helper)
  unsigned long long pacc[NPHASE] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
  // finer intervals inside the main waves' longest phase (row 2 of a.prof)
  unsigned long long pacc2[NPHASE] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev2 = 0;
#define SPO_SUB(i)                                             \
  if (PROF) {                                                  \
    const unsigned long long _t = __builtin_readcyclecounter(); \
    if ((i) >= 0) pacc2[(i) < 0 ? 0 : (i)] += _t - tprev2;     \
    tprev2 = _t;                                               \
  }
#define SPO_STAMP(i)                                           \
  if (PROF) {                                                  \
    const unsigned long long _t = __builtin_readcyclecounter(); \
    pacc[i] += _t - tprev; tprev = _t;                         \
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using U = UpdLds<KIN>;
  using H = UpdHLds<KIN>;
  using L = NetLds<KIN>;
  constexpr int NT1 = KIN / 16;
  const int tid = threadIdx.x, lane = tid & 63, j_ = lane & 15, q_ = lane >> 4;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool helper = wave8 >= 4;
  const int wave = wave8 & 3;                          // main wave w and helper wave w handle the same columns / rows
  int j = j_, q = q_, mycol = 16 * wave + j_, orow = 16 * wave + 4 * q_;
#define SPO_REIDX { j = pinned(j_); q = pinned(q_); mycol = 16 * wave + j; orow = 16 * wave + 4 * q; }
  const int net = a.first_net + wg;
  const int D = a.cfg.obs_dim, A = a.cfg.act_dim, B = a.cfg.batch;
  const NetGeom g = net_geom(D, A, net);
  const bool is_actor = (net == 2);
  const int OUT = g.OUT;
  const int ls_off = g.off - A;
  float* const red = lds + U::RED;
  volatile int* const xw = reinterpret_cast<volatile int*>(lds + H::XW);
  // The main waves read the two redo words once per step right after a barrier: through a plain LDS pointer (ds_read_b32).
  // Through the volatile pointer the read is a flat_load ... sc0 sc1 followed by s_waitcnt vmcnt(0): a round trip through
  // the memory pipeline that also waits for every prefetch in flight, twice per step on the critical path.
  const int* const xw_ro = reinterpret_cast<const int*>(lds + H::XW);
  float* const st_m = a.adam_m; float* const st_v = a.adam_v;
  stage_net<KIN>(a.theta, g, lds, tid, 512);
  if (is_actor && tid < A) red[128 + tid] = a.theta[ls_off + tid];
  if (tid < 32) xw[tid] = 0;
  const unsigned long long* const xtab = reinterpret_cast<const unsigned long long*>(lds + H::XW + 32);
  if (XR > 0 && tid < XR_MAX_WORLD)
    reinterpret_cast<unsigned long long*>(lds + H::XW + 32)[tid] = reinterpret_cast<unsigned long long>(a.xr_region[tid]);
  __syncthreads();

  const int64_t nsteps = (a.M + B - 1) / B;
  const float clip_lo = 1.f - a.cfg.clip, clip_hi = 1.f + a.cfg.clip;
  const float* tgt = (net == 0) ? a.tgt_r : a.tgt_c;

  // ---------------- helper state: Adam moments of the elements main lane (wave, lane) produces gradients for
  f4 mW1[NT1], vW1[NT1], mW2[4], vW2[4], mW3, vW3, mls, vls;
  float mb1 = 0, vb1 = 0, mb2 = 0, vb2 = 0, mb3 = 0, vb3 = 0;
  const bool own_b = (q_ == 0);
  const bool own_b3 = (wave == 0 && q_ == 0 && j_ < OUT);
  const bool own_ls = is_actor && wave == 0 && j_ == 0;
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) { mW1[nt] = f4{0.f, 0.f, 0.f, 0.f}; vW1[nt] = mW1[nt]; }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { mW2[nt] = f4{0.f, 0.f, 0.f, 0.f}; vW2[nt] = mW2[nt]; }
  mW3 = vW3 = mls = vls = f4{0.f, 0.f, 0.f, 0.f};
  if (helper) {   // (loaded here, used only in the helper branch below)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * nt + j;
        const int idx = g.w1() + (orow + r) * D + i;
        mW1[nt][r] = i < D ? a.adam_m[idx] : 0.f;
        vW1[nt][r] = i < D ? a.adam_v[idx] : 0.f;
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = g.w2() + (orow + r) * HID + 16 * nt + j;
        mW2[nt][r] = a.adam_m[idx];
        vW2[nt][r] = a.adam_v[idx];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * q + r;
      const int idx = g.w3() + o * HID + 16 * wave + j;
      mW3[r] = o < OUT ? a.adam_m[idx] : 0.f;
      vW3[r] = o < OUT ? a.adam_v[idx] : 0.f;
      mls[r] = (is_actor && o < A) ? a.adam_m[ls_off + o] : 0.f;
      vls[r] = (is_actor && o < A) ? a.adam_v[ls_off + o] : 0.f;
    }
    mb1 = a.adam_m[g.b1() + 16 * wave + j]; vb1 = a.adam_v[g.b1() + 16 * wave + j];
    mb2 = a.adam_m[g.b2() + 16 * wave + j]; vb2 = a.adam_v[g.b2() + 16 * wave + j];
    if (j < OUT) { mb3 = a.adam_m[g.b3() + j]; vb3 = a.adam_v[g.b3() + j]; }
  }
  const float b1c = a.cfg.beta1, b2c = a.cfg.beta2, eps = a.cfg.adam_eps;
  double pw1 = is_actor ? a.pow_b1_actor : a.pow_b1, pw2 = is_actor ? a.pow_b2_actor : a.pow_b2;
  const float lr = is_actor ? a.cfg.lr_actor : a.cfg.lr_critic;
  const float l2 = (!is_actor && a.cfg.use_critic_norm) ? a.cfg.l2_coef : 0.f;
  const float l2x2 = 2.f * l2;
  const float vcoef = (net == 0 && a.cfg.use_value_coefficient) ? 2.f : 1.f;
  float stale_sq = a.stale_io ? *a.stale_io : a.stale_sq;
  float step_size = 0.f, inv_bc2s = 0.f;
  float gsq = 0.f, psq = 0.f;                          // running ||g||^2 / sum p^2 over the helper lane's elements of this step
  // Backups of the two speculatively updated layers ({m, v, p} per tile) go to rows of global scratch: fire-and-forget
  // 16-byte stores that stay in L2 and are read back only on a clipped step (100 registers per lane otherwise; keeping
  // layer 1 in registers and layer 2 in the dead LDS images measured 2 % slower: 11.47 vs 11.25 us per step).
  // Row r of helper wave w: 64 consecutive float4 (a wave's store covers 1 KB), rows 1 KB apart -- four rows sit inside the 4 KB
  // immediate-offset window of one address.  (Rounds 2-4 had the rows 24 KB apart: every row its own 64-bit address, which the
  // compiler hoisted out of the step loop -- ~50 registers of loop-invariant addresses in the helper role, part of them spilled
  // and reloaded from scratch in front of the very stores they address: 38 -> 10 spilled registers, step time unchanged.)
  f4* const bk = reinterpret_cast<f4*>(a.backup) + ((size_t)(wg * 4 + wave) * UPD_BACKUP_ROWS * 64 + lane);
  constexpr size_t BKS = 64;
  constexpr int BK_W1 = 0, BK_W2 = 3 * NT1, BK_END = BK_W2 + 12;
  static_assert(BK_END <= UPD_BACKUP_ROWS, "backup rows");

  // One layer of the helper's work: L2 term + norms, backups, speculative Adam.  G = LDS gradient tiles written by main lane
  // (wave, lane); WOFF/LD = LDS weight image; the macro body is instantiated per layer to keep every index a constant.
#define SPO_H_ELEM(ADDR, GV, M, V, BKP)                                                   \
  {                                                                                        \
    const float p_ = lds[ADDR];                                                            \
    const float gg_ = vcoef * fmaf(l2x2, p_, (GV));                                        \
    gsq = fmaf(gg_, gg_, gsq); psq = fmaf(p_, p_, psq);                                    \
    (BKP) = p_;                                                                            \
    const AdamOut o_ = adam1(p_, gg_, (M), (V), b1c, b2c, eps, step_size, inv_bc2s);       \
    (M) = o_.m; (V) = o_.v; lds[ADDR] = o_.p;                                              \
  }
  // the same element in the exact redo: gradient (with its L2 term) recomputed from the restored parameter
#define SPO_H_REDO(ADDR, GV, M, V, PV, COEF)                                               \
  {                                                                                        \
    const float gg_ = vcoef * fmaf(l2x2, (PV), (GV)) * (COEF);                             \
    const AdamOut o_ = adam1((PV), gg_, (M), (V), b1c, b2c, eps, step_size, inv_bc2s);     \
    (M) = o_.m; (V) = o_.v; lds[ADDR] = o_.p;                                              \
  }

  // The two roles run two separate loops (separate register allocation: the helpers' 90 registers of optimiser state
  // must not be carried through the main waves' code); both execute the SAME barrier sequence per step --
  // Xd [, Xd again after a redo], B_stage, Xa, Xb, Xc -- and one barrier after the loop.
  if (!helper) {
  // ---------------- main state: prefetch pipeline (as in ppo_update_kernel)
  auto perm_pos = [&](int64_t s) -> int64_t {
    const int64_t base = s * B;
    const int64_t rem = a.M - base;
    const int ncols = (int)(rem < B ? rem : B);
    return base + (mycol < ncols ? mycol : 0);
  };
  auto fetch = [&](int64_t smp, ColData<NT1>& cd) {
    load_obs_tiles_raw<KIN>(a.obs + smp * D, D, q, cd.x);
    if (!is_actor) {
      cd.t0 = tgt[smp]; cd.t1 = 0.f; cd.actv = f4{0.f, 0.f, 0.f, 0.f};
    } else {
      cd.t0 = a.logp_old[smp]; cd.t1 = a.adv[smp];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        cd.actv[r] = a.act[smp * A + (ai < A ? ai : 0)];
      }
    }
  };
  auto settle_prefetch = [&](ColData<NT1>& cd) {
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) cd.x[nt][e] = pinned(cd.x[nt][e]);
    mask_obs_tiles<KIN>(D, q, cd.x);
    cd.t0 = pinned(cd.t0);
    if (is_actor) {
      cd.t1 = pinned(cd.t1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        const float av = pinned(cd.actv[r]);
        cd.actv[r] = ai < A ? av : 0.f;
      }
    }
  };
  ColData<NT1> nxt;
  int smp1 = 0;
  fetch((int64_t)a.perm[perm_pos(0)], nxt);
  if (nsteps > 1) smp1 = a.perm[perm_pos(1)];
  float* const cols = lds + H::COLS;
  if (SPO_H_GATHER) {
    // step 0's columns by the main waves themselves (every lane reads back only what lanes of its own wave wrote: no barrier);
    // from step 1 on the helpers write the x^T image and the column inputs between P1 and P3 of the step before
    settle_prefetch(nxt);
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[U::XT + (16 * nt + 4 * q + e) * LDB + mycol] = nxt.x[nt][e];
    if (q == 0) { cols[mycol] = nxt.t0; cols[64 + mycol] = nxt.t1; }
    if (is_actor) {
#pragma unroll
      for (int r = 0; r < 4; ++r) cols[(2 + 4 * q + r) * 64 + mycol] = nxt.actv[r];
    }
  }
  if (PROF) tprev = __builtin_readcyclecounter();

  for (int64_t s = 0; s < nsteps; ++s) {
    const int64_t base = s * B;
    const int64_t rem = a.M - base;
    const int ncols = (int)(rem < B ? rem : B);
    const float inv_n = 1.f / (float)ncols;
    ColData<NT1> cur;
    f4 h1[4], h2[4];
    int smp_next = 0;
    int64_t pos2 = 0;
    if (SPO_H_GATHER) {
      // the helpers put this step's x^T image in place before the barrier the previous step ended with: 16 LDS reads instead
      // of the settle (pad selects, index widening, 16 x^T stores: 1.3 k cycles of a lone wave)
      SPO_REIDX
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) cur.x[nt][e] = lds[U::XT + (16 * nt + 4 * q + e) * LDB + mycol];
    } else {
      SPO_REIDX
      cur = nxt;
      settle_prefetch(cur);
      smp_next = pinned(smp1);
      pos2 = (s + 2 < nsteps) ? perm_pos(s + 2) : 0;
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[U::XT + (16 * nt + 4 * q + e) * LDB + mycol] = cur.x[nt][e];
    }
    SPO_STAMP(0)
    bool redone = false, late_redone = false;
    for (;;) {
      {
        SPO_REIDX
        layer_hidden<NT1, true>(lds + L::W1, L::LD1, lds + L::B1, cur.x, h1, j, q);
        // h1^T goes to its LDS image right away: the stores drain beside layer 2's MFMAs instead of queueing with the other
        // three images at the end of the backward pass (the LDS store path moves 64 B/clk for the whole CU)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[U::H1T + (16 * mt + 4 * q + r) * LDB + mycol] = h1[mt][r];
      }
      SPO_STAMP(1)
      __syncthreads();                                                    // Q2: (speculative) W2 / b2 in place
      SPO_STAMP(2)
      {
        SPO_REIDX
        layer_hidden<4, true>(lds + L::W2, LDH, lds + L::B2, h1, h2, j, q);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[U::H2T + (16 * mt + 4 * q + r) * LDB + mycol] = h2[mt][r];
      }
      SPO_STAMP(3)
      __syncthreads();                                                    // Xd: W3 / b3 / log_std in place, clip verdict known
      SPO_STAMP(4)
      const int redo_tag = xw_ro[18];
      if (redo_tag == (int)(s & 0x3fffffff) + 1 && !redone) { redone = true; continue; }   // clipped: W1, W2 were redone
      {
        // ---- rest of the step of the main waves: prefetch, L3, loss, backward, staging
        SPO_REIDX
        SPO_SUB(-1)
        if (!SPO_H_GATHER) {
          if (s + 1 < nsteps) fetch((int64_t)smp_next, nxt);
          if (s + 2 < nsteps) smp1 = a.perm[pos2];
        } else {
          cur.t0 = cols[mycol]; cur.t1 = cols[64 + mycol];
#pragma unroll
          for (int r = 0; r < 4; ++r) cur.actv[r] = is_actor ? cols[(2 + 4 * q + r) * 64 + mycol] : 0.f;
        }
        const f4 o = layer_out(lds + L::W3, lds + L::B3, h2, j, q);
        SPO_SUB(0)
        const bool cv = mycol < ncols;
        float ivar[4], lsd[4], amask[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 4 * q + r;
          const bool on = is_actor && ai < A;
          const float lsv = on ? red[128 + ai] : 0.f;
          const float sdv = __expf(lsv);
          amask[r] = on ? 1.f : 0.f;
          ivar[r] = __builtin_amdgcn_rcpf(sdv * sdv);
          lsd[r] = on ? lsv + LOG_SQRT_2PI : 0.f;
        }
        float w3c[4][4];
        if (SPO_H_W3C_EARLY) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) w3c[r][mt] = lds[L::W3 + (4 * q + r) * LDH + 16 * mt + j];
        }
        f4 dO = {0.f, 0.f, 0.f, 0.f}, dls = {0.f, 0.f, 0.f, 0.f};
        float lsum = 0.f;
        if (!is_actor) {
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
          const float ratio = __expf(lp - cur.t0);
          const float rc = fminf(fmaxf(ratio, clip_lo), clip_hi);
          const float s1 = ratio * adv, s2 = rc * adv;
          const bool inr = (ratio >= clip_lo) && (ratio <= clip_hi);
          float gr;
          if (s1 < s2) gr = adv;
          else if (s1 > s2) gr = inr ? adv : 0.f;
          else gr = 0.5f * adv + (inr ? 0.5f * adv : 0.f);
          const float dlp = cv ? -(gr * ratio) * inv_n : 0.f;
          lsum = ((q == 0 && cv) ? 1.f : 0.f) * fminf(s1, s2);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = dif[r] * ivar[r];
            dO[r] = dlp * z;
            dls[r] = (dlp * amask[r]) * (dif[r] * z - 1.f);
          }
        }
        SPO_SUB(1)
        // backward through the MLP (as in ppo_update_kernel)
        f4 dz2[4], dz1[4];
        {
          f4 acc[4];
          if (!SPO_H_W3C_EARLY) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) w3c[r][mt] = lds[L::W3 + (4 * q + r) * LDH + 16 * mt + j];
          }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma4(w3c[r][mt], dO[r], acc[mt]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz2[mt][r] = acc[mt][r] * fmaf(-h2[mt][r], h2[mt][r], 1.f);
          // dZ2^T and dO^T are staged now: their stores drain beside the 64 MFMAs of the next product
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[U::DZ2T + (16 * mt + 4 * q + r) * LDB + mycol] = dz2[mt][r];
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[U::DOT + (4 * q + r) * LDB + mycol] = dO[r];
          SPO_SUB(2)
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
                for (int mt = 0; mt < 4; ++mt)
                  w2c[(nt + 1) & 1][r][mt] = lds[L::W2 + (16 * (nt + 1) + 4 * q + r) * LDH + 16 * mt + j];
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
        SPO_SUB(3)
        SPO_REIDX
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[U::DZ1T + (16 * mt + 4 * q + r) * LDB + mycol] = dz1[mt][r];
        SPO_SUB(4)
        {
          const float ls = wave_sum_lane63(lsum);
          if (lane == 63) red[wave] = ls;
          if (is_actor) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t = row_sum_lane15(dls[r]);
              if (j == 15) red[16 + wave * 16 + 4 * q + r] = t;
            }
          }
        }
        SPO_SUB(5)
      }
      SPO_STAMP(5)
      __syncthreads();                                                    // B_stage: all [feature][batch] images complete
      SPO_STAMP(6)
      // Deferred validation: the helpers updated ALL layers with clip coefficient 1 and found, while this forward /
      // backward ran, that the previous step's joint norm exceeds the bound.  They have restored and redone the update
      // exactly before this barrier; the step is repeated from layer 1 on the exact weights (x from its LDS image).
      if (xw_ro[20] == (int)(s & 0x3fffffff) + 1 && !late_redone) {
        late_redone = true;
        SPO_REIDX
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) cur.x[nt][e] = lds[U::XT + (16 * nt + 4 * q + e) * LDB + mycol];
        continue;
      }
      break;
    }
    {
      // ---- dW1 (rows 16 wave .., all NT1 column tiles) -> G1, db1 -> GB[0]
      SPO_REIDX
      f4 az1[4], aW1[NT1];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        az1[r4] = *reinterpret_cast<const f4*>(lds + U::DZ1T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) aW1[nt] = f4{0.f, 0.f, 0.f, 0.f};
      f4 bx[2][NT1];
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
        bx[0][nt] = *reinterpret_cast<const f4*>(lds + U::XT + (16 * nt + j) * LDB + 4 * q);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        if (r4 + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt)
            bx[(r4 + 1) & 1][nt] = *reinterpret_cast<const f4*>(lds + U::XT + (16 * nt + j) * LDB + 16 * (r4 + 1) + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) aW1[nt] = mfma4(az1[r4][e], bx[r4 & 1][nt][e], aW1[nt]);
      }
      float rs1 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) rs1 += (az1[r4][0] + az1[r4][1]) + (az1[r4][2] + az1[r4][3]);
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) *reinterpret_cast<f4*>(lds + H::G1 + ((wave * NT1 + nt) * 64 + lane) * 4) = aW1[nt];
      lds[H::GB + 0 * 256 + wave * 64 + lane] = quad_row_sum(rs1);
    }
    SPO_STAMP(7)
    __syncthreads();                                                      // P1: G1 complete; x^T and dZ1^T are dead
    SPO_STAMP(8)
    {
      // ---- dW2 -> G2 (inside the dead dZ1^T image), db2 -> GB[1]
      SPO_REIDX
      f4 az2[4], aW2[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        az2[r4] = *reinterpret_cast<const f4*>(lds + U::DZ2T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) aW2[nt] = f4{0.f, 0.f, 0.f, 0.f};
      f4 bh[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        bh[0][nt] = *reinterpret_cast<const f4*>(lds + U::H1T + (16 * nt + j) * LDB + 4 * q);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        if (r4 + 1 < 4) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            bh[(r4 + 1) & 1][nt] = *reinterpret_cast<const f4*>(lds + U::H1T + (16 * nt + j) * LDB + 16 * (r4 + 1) + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) aW2[nt] = mfma4(az2[r4][e], bh[r4 & 1][nt][e], aW2[nt]);
      }
      float rs2 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) rs2 += (az2[r4][0] + az2[r4][1]) + (az2[r4][2] + az2[r4][3]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f4*>(lds + H::G2 + ((wave * 4 + nt) * 64 + lane) * 4) = aW2[nt];
      lds[H::GB + 1 * 256 + wave * 64 + lane] = quad_row_sum(rs2);
    }
    __builtin_amdgcn_sched_barrier(0);        // keep dW3's operand reads out of dW2's register budget
    {
      // ---- dW3 -> G3, db3 -> GB[2]
      SPO_REIDX
      f4 az3[4], b3[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        az3[r4] = *reinterpret_cast<const f4*>(lds + U::DOT + j * LDB + 16 * r4 + 4 * q);
        b3[r4] = *reinterpret_cast<const f4*>(lds + U::H2T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
      }
      f4 w3a = {0.f, 0.f, 0.f, 0.f}, w3b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r4 = 0; r4 < 4; r4 += 2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          w3a = mfma4(az3[r4][e], b3[r4][e], w3a);
          w3b = mfma4(az3[r4 + 1][e], b3[r4 + 1][e], w3b);
        }
      float rs3 = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) rs3 += (az3[r4][0] + az3[r4][1]) + (az3[r4][2] + az3[r4][3]);
      *reinterpret_cast<f4*>(lds + H::G3 + (wave * 64 + lane) * 4) = w3a + w3b;
      lds[H::GB + 2 * 256 + wave * 64 + lane] = quad_row_sum(rs3);
    }
    SPO_STAMP(9)
    __syncthreads();                                                      // P3: G2, G3 complete (helpers: W1 / b1 updated)
    if (PROF) tprev = __builtin_readcyclecounter();
  }
  if (PROF && a.prof && tid == 0 && wg == a.n_nets - 1)
    for (int i = 0; i < NPHASE; ++i) { a.prof[i] = pacc[i]; a.prof[2 * NPHASE + i] = pacc2[i]; }
  __syncthreads();                                                        // after the loop (helpers: last verdict done)
  } else {
  // The helpers are the younger waves of their SIMDs and would get only the issue slots the main waves leave; their
  // work must be done when the main waves arrive at the next barrier, and the main waves' MFMA streams leave most VALU
  // slots free anyway: static priority for the helpers.
  __builtin_amdgcn_s_setprio(SPO_HELPER_PRIO);
  // ---- SPO_H_GATHER: the helpers gather the NEXT minibatch's columns.  Lane (wave, l) serves the column main lane (wave, l)
  // computes on.  The sample index is loaded right after the x^T image of the previous minibatch has been handed over, the
  // row / target loads are issued in the idle window before B_stage, and the data is written into the x^T image and the COLS
  // rows between P1 (dW1 has read the old image) and P3 (the main waves start the next forward) -- none of it on the main
  // waves, whose lone-wave issue rate (5.5-7.5 cycles per instruction, tools/probes/mfma_valu_overlap.hip) made the settle the
  // most expensive non-matrix phase of the step.
  ColData<NT1> hx;
  int hsmp = 0;
  // (lane indices through an opaque move at every use, like SPO_REIDX: otherwise the compiler computes the sixteen x^T store
  //  addresses once, keeps them in registers across the whole loop next to the optimiser state, and spills 100 registers)
  auto h_perm_pos = [&](int64_t s_) -> int64_t {
    const int64_t base_ = s_ * B;
    const int64_t rem_ = a.M - base_;
    const int nc_ = (int)(rem_ < B ? rem_ : B);
    const int col_ = 16 * wave + pinned(j_);
    return base_ + (col_ < nc_ ? col_ : 0);
  };
  auto h_fetch = [&](int64_t smp) {
    const int qq = pinned(q_);
    load_obs_tiles_raw<KIN>(a.obs + smp * D, D, qq, hx.x);
    if (!is_actor) {
      hx.t0 = tgt[smp]; hx.t1 = 0.f; hx.actv = f4{0.f, 0.f, 0.f, 0.f};
    } else {
      hx.t0 = a.logp_old[smp]; hx.t1 = a.adv[smp];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * qq + r;
        hx.actv[r] = a.act[smp * A + (ai < A ? ai : 0)];
      }
    }
  };
  auto h_publish = [&]() {                    // settle (pad selects) + x^T image + column inputs of the next step
    const int qq = pinned(q_), col_ = 16 * wave + pinned(j_);
    float* const xt = lds + U::XT + 4 * qq * LDB + col_;          // ONE address register: the rest are immediate offsets
    float* const cols = lds + H::COLS + col_;
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) hx.x[nt][e] = pinned(hx.x[nt][e]);
    mask_obs_tiles<KIN>(D, qq, hx.x);
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) xt[(16 * nt + e) * LDB] = hx.x[nt][e];
    if (qq == 0) { cols[0] = pinned(hx.t0); cols[64] = pinned(hx.t1); }
    if (is_actor) {
      float* const ca = cols + (2 + 4 * qq) * 64;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float av = pinned(hx.actv[r]);
        ca[r * 64] = (4 * qq + r) < A ? av : 0.f;
      }
    }
  };
  if (SPO_H_GATHER && nsteps > 1) h_fetch((int64_t)a.perm[h_perm_pos(1)]);
  // step 0 has nothing to wait for: its forward runs on the weights staged at launch
  __syncthreads();                                                        // Q2 of step 0
  __syncthreads();                                                        // Xd of step 0
  __syncthreads();                                                        // B_stage of step 0
  if (PROF) tprev = __builtin_readcyclecounter();
  bool spec = a.spec_mode != 0;                      // deferred validation of the clip (see below); off after a clipped step
  int n_late = 0, n_redo = 0, n_cons = 0;            // g_upd_counters
  for (int64_t s = 0; s < nsteps; ++s) {
    const int64_t base = s * B;
    const int64_t rem = a.M - base;
    const int ncols = (int)(rem < B ? rem : B);
    const float inv_n = 1.f / (float)ncols;
    const unsigned tag = (unsigned)(s & 0x3fffffff) + 1u;
    bool redo_next = false;
    bool h_fetched = false;
    // register backups of layer 1 only: it is the one layer updated before the joint norm is known
    float bmb1, bvb1, bpb1;
    __syncthreads();                                                      // P1: G1 complete
    SPO_STAMP(0)
    if (SPO_H_GATHER) {
      // dW1 has read the old x^T image: hand the next minibatch over (the main waves pick it up after P3), then start the
      // index load of the one after it
      if (s + 1 < nsteps) h_publish();
      if (s + 2 < nsteps) hsmp = a.perm[h_perm_pos(s + 2)];
    }
    {
      // ---- layer 1 (W1, b1): L2 term, norm share, SPECULATIVE Adam (clip coefficient 1)
      SPO_REIDX
      if constexpr (XR > 0) {
        // data-parallel: the gradient of the GLOBAL minibatch = mean over the ranks, exchanged layer by layer; layer 1 goes
        // out while the main waves still compute dW2 / dW3.  The reduced values replace the local ones in G1 / GB[0].
        const unsigned gtag = a.xr_step0 + (unsigned)s + 1u;
        const int par = (int)(gtag & 1u), hl = wave * 64 + lane;
        f4 gx[NT1 + 1];
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) gx[nt] = *reinterpret_cast<const f4*>(lds + H::G1 + ((wave * NT1 + nt) * 64 + lane) * 4);
        gx[NT1] = f4{lds[H::GB + 0 * 256 + hl], 0.f, 0.f, 0.f};
        if constexpr (XRD == 2 || XRD == 3) {
          // packed words, one poll batch: words 0 .. NWA - 1 of the slot (XRD = 3: through the XCD's L2, cached regions [2], [3])
          float gf[4 * NT1 + 1];
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) gf[4 * nt + r] = gx[nt][r];
          gf[4 * NT1] = gx[NT1][0];
          xr_rd16_flat<4 * NT1 + 1, 0, XRD == 3>(xtab + (XRD == 3 ? 2 : 0), a.xr_rank, XR, net, hl, gtag, gf,
                                       reinterpret_cast<volatile float*>(lds + H::XW + 21), a.err);
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) gx[nt][r] = gf[4 * nt + r];
          gx[NT1][0] = gf[4 * NT1];
        } else {
#pragma unroll
        for (int v = 0; v <= NT1; ++v) a2a_push_row<XR>(a, xtab, net, hl, par, v < NT1 ? v : NT1 + 5, gx[v]);
        a2a_signal<XR>(a, xtab, net, lane, wave, par, 0, gtag);
        a2a_wait<XR>(a, xtab, net, lane, wave, par, 0, gtag, xw + 19, a.err);
        {
          int rows[NT1 + 1];
#pragma unroll
          for (int v = 0; v <= NT1; ++v) rows[v] = v < NT1 ? v : NT1 + 5;
          a2a_pull_all<XR, NT1 + 1>(a, xtab, net, hl, par, rows, gx);
        }
        }
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) *reinterpret_cast<f4*>(lds + H::G1 + ((wave * NT1 + nt) * 64 + lane) * 4) = gx[nt];
        lds[H::GB + 0 * 256 + hl] = gx[NT1][0];
      }
      pw1 *= (double)b1c; pw2 *= (double)b2c;
      adam_scalars(lr, pw1, pw2, step_size, inv_bc2s);
      gsq = psq = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        const f4 gv = *reinterpret_cast<const f4*>(lds + H::G1 + ((wave * NT1 + nt) * 64 + lane) * 4);
        bk[BKS * (BK_W1 + 3 * nt + 0)] = mW1[nt];
        bk[BKS * (BK_W1 + 3 * nt + 1)] = vW1[nt];
        f4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r)                                         // pad columns hold p == 0, g == 0: stay 0
          SPO_H_ELEM(L::W1 + (orow + r) * L::LD1 + 16 * nt + j, gv[r], mW1[nt][r], vW1[nt][r], pv[r])
        bk[BKS * (BK_W1 + 3 * nt + 2)] = pv;
      }
      {
        // bias row 16 wave + j: replicated over q (all replicas compute and store identical values; q == 0 counts)
        const float gb = lds[H::GB + 0 * 256 + wave * 64 + lane];
        const float gsq0 = gsq, psq0 = psq;
        bmb1 = mb1; bvb1 = vb1;
        SPO_H_ELEM(L::B1 + 16 * wave + j, gb, mb1, vb1, bpb1)
        if (!own_b) { gsq = gsq0; psq = psq0; }
      }
    }
    SPO_STAMP(1)
    __syncthreads();                                                      // P3: G2, G3 complete; W1 / b1 updated
    SPO_STAMP(2)
    // ---- gradients (with their L2 terms) of layers 2 and 3, norm shares; the norm goes out EARLY
    f4 gg2[4], gg3, dls = {0.f, 0.f, 0.f, 0.f};
    f4 pW2[4];
    float ggb2, ggb3 = 0.f, bmb2, bvb2, bpb2;
    unsigned long long* const grow = a.slots + (s & 1) * 16;            // [parity][network][helper wave]
    if constexpr (XR > 0) {
      // data-parallel: layers 2 and 3 (and the log_std row) in one exchange; the reduced values replace the local ones in
      // G2 / G3 / GB[1..2] and in the log_std partials (first partial = reduced sum, the other three = 0)
      const unsigned gtag = a.xr_step0 + (unsigned)s + 1u;
      const int par = (int)(gtag & 1u), hl = wave * 64 + lane;
      const bool with_ls = is_actor && wave == 0;                        // wave-uniform
      f4 gx[7];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) gx[nt] = *reinterpret_cast<const f4*>(lds + H::G2 + ((wave * 4 + nt) * 64 + lane) * 4);
      gx[4] = *reinterpret_cast<const f4*>(lds + H::G3 + (wave * 64 + lane) * 4);
      gx[5] = f4{lds[H::GB + 1 * 256 + hl], lds[H::GB + 2 * 256 + hl], 0.f, 0.f};
      gx[6] = f4{0.f, 0.f, 0.f, 0.f};
      if (with_ls) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 4 * q_ + r;
          gx[6][r] = (red[16 + ai] + red[32 + ai]) + (red[48 + ai] + red[64 + ai]);
        }
      }
      const int rows7[7] = {NT1 + 0, NT1 + 1, NT1 + 2, NT1 + 3, NT1 + 4, NT1 + 7, NT1 + 6};
      if constexpr (XRD == 2 || XRD == 3) {
        // packed words NWA .. NWA + 8 of the slot: W2 tiles, W3 tile, db2, db3, the log_std row (zeros except wave 0 of the actor)
        constexpr int NWA = (4 * NT1 + 1 + 2) / 3;
        float gf[26];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) gf[4 * nt + r] = gx[nt][r];
        gf[20] = gx[5][0]; gf[21] = gx[5][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) gf[22 + r] = gx[6][r];
        xr_rd16_flat<26, NWA, XRD == 3>(xtab + (XRD == 3 ? 2 : 0), a.xr_rank, XR, net, hl, gtag, gf,
                              reinterpret_cast<volatile float*>(lds + H::XW + 21), a.err);
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) gx[nt][r] = gf[4 * nt + r];
        gx[5][0] = gf[20]; gx[5][1] = gf[21];
#pragma unroll
        for (int r = 0; r < 4; ++r) gx[6][r] = gf[22 + r];
      } else {
#pragma unroll
      for (int v = 0; v < 6; ++v) a2a_push_row<XR>(a, xtab, net, hl, par, rows7[v], gx[v]);
      if (with_ls) a2a_push_row<XR>(a, xtab, net, hl, par, rows7[6], gx[6]);
      a2a_signal<XR>(a, xtab, net, lane, wave, par, 1, gtag);
      a2a_wait<XR>(a, xtab, net, lane, wave, par, 1, gtag, xw + 19, a.err);
      {
        int r6[6];
        f4 g6[6];
#pragma unroll
        for (int v = 0; v < 6; ++v) { r6[v] = rows7[v]; g6[v] = gx[v]; }
        a2a_pull_all<XR, 6>(a, xtab, net, hl, par, r6, g6);
#pragma unroll
        for (int v = 0; v < 6; ++v) gx[v] = g6[v];
        if (with_ls) {
          const int rc[1] = {rows7[6]};
          f4 gc[1] = {gx[6]};
          a2a_pull<XR, 1>(a, xtab, net, hl, par, rc, gc);
          gx[6] = gc[0];
        }
      }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f4*>(lds + H::G2 + ((wave * 4 + nt) * 64 + lane) * 4) = gx[nt];
      *reinterpret_cast<f4*>(lds + H::G3 + (wave * 64 + lane) * 4) = gx[4];
      lds[H::GB + 1 * 256 + hl] = gx[5][0];
      lds[H::GB + 2 * 256 + hl] = gx[5][1];
      if (with_ls) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 4 * q_ + r;
          red[16 + ai] = gx[6][r]; red[32 + ai] = 0.f; red[48 + ai] = 0.f; red[64 + ai] = 0.f;
        }
      }
    }
    // exact redo of the two hidden layers from their backups (clip coefficient known)
    auto redo_layers12 = [&](float coef) {
      SPO_REIDX
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        const f4 gv = *reinterpret_cast<const f4*>(lds + H::G1 + ((wave * NT1 + nt) * 64 + lane) * 4);
        mW1[nt] = bk[BKS * (BK_W1 + 3 * nt + 0)]; vW1[nt] = bk[BKS * (BK_W1 + 3 * nt + 1)];
        const f4 pv = bk[BKS * (BK_W1 + 3 * nt + 2)];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          SPO_H_REDO(L::W1 + (orow + r) * L::LD1 + 16 * nt + j, gv[r], mW1[nt][r], vW1[nt][r], pv[r], coef)
      }
      mb1 = bmb1; vb1 = bvb1;
      const float g1b = lds[H::GB + 0 * 256 + wave * 64 + lane];
      SPO_H_REDO(L::B1 + 16 * wave + j, g1b, mb1, vb1, bpb1, coef)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        mW2[nt] = bk[BKS * (BK_W2 + 3 * nt + 0)]; vW2[nt] = bk[BKS * (BK_W2 + 3 * nt + 1)];
        const f4 pv = bk[BKS * (BK_W2 + 3 * nt + 2)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const AdamOut o_ = adam1(pv[r], gg2[nt][r] * coef, mW2[nt][r], vW2[nt][r], b1c, b2c, eps, step_size, inv_bc2s);
          mW2[nt][r] = o_.m; vW2[nt][r] = o_.v; lds[L::W2 + (orow + r) * LDH + 16 * nt + j] = o_.p;
        }
      }
      mb2 = bmb2; vb2 = bvb2;
      const AdamOut o2 = adam1(bpb2, ggb2 * coef, mb2, vb2, b1c, b2c, eps, step_size, inv_bc2s);
      mb2 = o2.m; vb2 = o2.v; lds[L::B2 + 16 * wave + j] = o2.p;
    };
    float coef = 1.f;
    if (spec) {
      // ======== deferred validation (the previous step was not clipped): EVERY layer is updated with coefficient 1 as
      // soon as its gradient is there, so the main waves never wait for the joint norm; the norm is checked while they
      // run the next forward / backward, and a clipped step is restored, redone exactly and its successor repeated.
      {
        // ---- layer 2 (W2, b2): L2 term, norm share, backups, speculative Adam
        SPO_REIDX
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f4 gv = *reinterpret_cast<const f4*>(lds + H::G2 + ((wave * 4 + nt) * 64 + lane) * 4);
          bk[BKS * (BK_W2 + 3 * nt + 0)] = mW2[nt];
          bk[BKS * (BK_W2 + 3 * nt + 1)] = vW2[nt];
          f4 pv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int addr = L::W2 + (orow + r) * LDH + 16 * nt + j;
            const float p_ = lds[addr];
            const float g_ = vcoef * fmaf(l2x2, p_, gv[r]);
            gg2[nt][r] = g_; pv[r] = p_; gsq = fmaf(g_, g_, gsq); psq = fmaf(p_, p_, psq);
            const AdamOut o_ = adam1(p_, g_, mW2[nt][r], vW2[nt][r], b1c, b2c, eps, step_size, inv_bc2s);
            mW2[nt][r] = o_.m; vW2[nt][r] = o_.v; lds[addr] = o_.p;
          }
          bk[BKS * (BK_W2 + 3 * nt + 2)] = pv;
        }
        bpb2 = lds[L::B2 + 16 * wave + j];
        ggb2 = vcoef * fmaf(l2x2, bpb2, lds[H::GB + 1 * 256 + wave * 64 + lane]);
        if (own_b) { gsq = fmaf(ggb2, ggb2, gsq); psq = fmaf(bpb2, bpb2, psq); }
        bmb2 = mb2; bvb2 = vb2;
        const AdamOut o_ = adam1(bpb2, ggb2, mb2, vb2, b1c, b2c, eps, step_size, inv_bc2s);
        mb2 = o_.m; vb2 = o_.v; lds[L::B2 + 16 * wave + j] = o_.p;
      }
      SPO_STAMP(3)
      if (s + 1 < nsteps) __syncthreads();                                // Q2 of step s + 1: (speculative) W2 / b2 in place
      SPO_STAMP(4)
      float loss_data = 0.f;
      if (wave == 0 && lane == 0 && s + 1 < nsteps) loss_data = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
      f4 bW3p, bW3m = mW3, bW3v = vW3, blsp = {0.f, 0.f, 0.f, 0.f}, blsm = mls, blsv = vls;
      float bb3p = 0.f, bb3m = mb3, bb3v = vb3;
      {
        // ---- output layer (W3, b3, log_std): the same, backups in registers; then the norm share goes out
        SPO_REIDX
        const f4 gv = *reinterpret_cast<const f4*>(lds + H::G3 + (wave * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                       // pad rows hold p == 0, g == 0
          const int addr = L::W3 + (4 * q + r) * LDH + 16 * wave + j;
          const float p_ = lds[addr];
          const float g_ = vcoef * fmaf(l2x2, p_, gv[r]);
          gg3[r] = g_; bW3p[r] = p_; gsq = fmaf(g_, g_, gsq); psq = fmaf(p_, p_, psq);
          const AdamOut o_ = adam1(p_, g_, mW3[r], vW3[r], b1c, b2c, eps, step_size, inv_bc2s);
          mW3[r] = o_.m; vW3[r] = o_.v; lds[addr] = o_.p;
        }
        if (wave == 0) {
          bb3p = lds[L::B3 + j];
          ggb3 = vcoef * fmaf(l2x2, bb3p, lds[H::GB + 2 * 256 + lane]);
          if (q == 0) { gsq = fmaf(ggb3, ggb3, gsq); psq = fmaf(bb3p, bb3p, psq); }
          const AdamOut o_ = adam1(bb3p, ggb3, mb3, vb3, b1c, b2c, eps, step_size, inv_bc2s);
          mb3 = o_.m; vb3 = o_.v; lds[L::B3 + j] = o_.p;
          if (is_actor) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ai = 4 * q + r;
              dls[r] = (red[16 + ai] + red[32 + ai]) + (red[48 + ai] + red[64 + ai]);     // 0 on pad rows
              if (own_ls) gsq = fmaf(dls[r], dls[r], gsq);
              blsp[r] = red[128 + ai];
              const AdamOut ol = adam1(blsp[r], dls[r], mls[r], vls[r], b1c, b2c, eps, step_size, inv_bc2s);
              mls[r] = ol.m; vls[r] = ol.v; red[128 + ai] = ol.p;
            }
          }
        }
        const float wg_sq = wave_sum_lane63(gsq), wp_sq = wave_sum_lane63(psq);
        if (lane == 63) {
          st_granule(grow + 4 * wg + wave, ((unsigned long long)tag << 32) | __float_as_uint(wg_sq));
          red[88 + wave] = wp_sq;                                          // sum p^2 shares: read after the next barrier
        }
      }
      SPO_STAMP(5)
      if (s + 1 < nsteps) __syncthreads();                                // Xd of step s + 1: W3 / b3 / log_std in place
#if SPO_H_GATHER == 2
      // rows / targets of minibatch s + 2 right behind Xd: a whole backward pass + dW1 of the main waves to land (issued after
      // B_stage they were not back by P1: the helper stalled on them at the head of its critical P1 -> P3 stretch, 11.5 against
      // 10.8 us per step)
      if (s + 2 < nsteps) { h_fetch((int64_t)pinned(hsmp)); h_fetched = true; }
#endif
      SPO_STAMP(9)
      {
        if (wave == 0 && lane == 0 && s + 1 < nsteps) {
          const float pp = (red[88] + red[89]) + (red[90] + red[91]);
          a.losses[s * 3 + net] = is_actor ? -loss_data : loss_data + l2 * pp;
        }
        float mine = 0.f;
        const int ngr = 4 * a.n_nets;
        if (lane < ngr) {
          __builtin_amdgcn_s_setprio(0);
          unsigned long long v = ld_granule(grow + lane);
          unsigned sp2 = 0;
          while ((unsigned)(v >> 32) != tag) {
            if (++sp2 > (1u << 22)) { *a.err = 1; break; }
            __builtin_amdgcn_s_sleep(4);
            v = ld_granule(grow + lane);
          }
          mine = __uint_as_float((unsigned)v);
          __builtin_amdgcn_s_setprio(SPO_HELPER_PRIO);
        }
        float total_sq = stale_sq;
        for (int kk = 0; kk < ngr; ++kk) total_sq += __shfl(mine, kk);
        const float norm = sqrtf(total_sq);
        coef = a.cfg.max_grad_norm / (norm + 1e-6f);
        coef = coef > 1.f ? 1.f : coef;
        stale_sq *= coef * coef;
      }
      bool late = false;
      if (coef != 1.f) {
        // ---- clipped after all: restore every layer, redo it exactly, make the main waves repeat the step they are in
        ++n_late;
        redo_layers12(coef);
        SPO_REIDX
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int addr = L::W3 + (4 * q + r) * LDH + 16 * wave + j;
          const AdamOut o_ = adam1(bW3p[r], gg3[r] * coef, bW3m[r], bW3v[r], b1c, b2c, eps, step_size, inv_bc2s);
          mW3[r] = o_.m; vW3[r] = o_.v; lds[addr] = o_.p;
        }
        if (wave == 0) {
          const AdamOut o_ = adam1(bb3p, ggb3 * coef, bb3m, bb3v, b1c, b2c, eps, step_size, inv_bc2s);
          mb3 = o_.m; vb3 = o_.v; lds[L::B3 + j] = o_.p;
          if (is_actor) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ai = 4 * q + r;
              const AdamOut ol = adam1(blsp[r], dls[r] * coef, blsm[r], blsv[r], b1c, b2c, eps, step_size, inv_bc2s);
              mls[r] = ol.m; vls[r] = ol.v; red[128 + ai] = ol.p;
            }
          }
        }
        if (wave == 0 && lane == 0) xw[20] = (int)((s + 1) & 0x3fffffff) + 1;
        late = true;
      }
      SPO_STAMP(8)
      if (s + 1 < nsteps) {
        __syncthreads();                                                  // B_stage of step s + 1: the verdict is out
        if (late) {                                                       // the main waves repeat L1, Q2, L2, Xd, ..., B_stage
          __syncthreads();
          __syncthreads();
          __syncthreads();
        }
      }
      spec = (coef == 1.f);
    } else {
      // ======== the previous step was clipped (or SPO_UPDATE_SPEC=0): the output layer waits for the joint norm
      {
        SPO_REIDX
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f4 gv = *reinterpret_cast<const f4*>(lds + H::G2 + ((wave * 4 + nt) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p_ = lds[L::W2 + (orow + r) * LDH + 16 * nt + j];
            const float g_ = vcoef * fmaf(l2x2, p_, gv[r]);
            gg2[nt][r] = g_; pW2[nt][r] = p_; gsq = fmaf(g_, g_, gsq); psq = fmaf(p_, p_, psq);
          }
        }
        {
          const f4 gv = *reinterpret_cast<const f4*>(lds + H::G3 + (wave * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {                                       // pad rows hold p == 0, g == 0
            const float p_ = lds[L::W3 + (4 * q + r) * LDH + 16 * wave + j];
            const float g_ = vcoef * fmaf(l2x2, p_, gv[r]);
            gg3[r] = g_; gsq = fmaf(g_, g_, gsq); psq = fmaf(p_, p_, psq);
          }
        }
        {
          bpb2 = lds[L::B2 + 16 * wave + j];
          ggb2 = vcoef * fmaf(l2x2, bpb2, lds[H::GB + 1 * 256 + wave * 64 + lane]);
          if (own_b) { gsq = fmaf(ggb2, ggb2, gsq); psq = fmaf(bpb2, bpb2, psq); }
        }
        if (wave == 0) {
          const float p_ = lds[L::B3 + j];
          ggb3 = vcoef * fmaf(l2x2, p_, lds[H::GB + 2 * 256 + lane]);
          if (q == 0) { gsq = fmaf(ggb3, ggb3, gsq); psq = fmaf(p_, p_, psq); }
          if (is_actor) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ai = 4 * q + r;
              dls[r] = (red[16 + ai] + red[32 + ai]) + (red[48 + ai] + red[64 + ai]);     // 0 on pad rows
              if (own_ls) gsq = fmaf(dls[r], dls[r], gsq);
            }
          }
        }
        // every helper wave publishes ITS share of ||g||^2 as a tagged granule of its own (12 granules per step); every
        // helper wave later adds all of them in the same fixed order -- no gather and no spinning inside the workgroup
        const float wg_sq = wave_sum_lane63(gsq), wp_sq = wave_sum_lane63(psq);
        if (lane == 63) {
          st_granule(grow + 4 * wg + wave, ((unsigned long long)tag << 32) | __float_as_uint(wg_sq));
          red[88 + wave] = wp_sq;                                            // sum p^2 shares: read after the next barrier
        }
        SPO_STAMP(3)
      }
      {
        // ---- layer 2 (W2, b2): SPECULATIVE Adam like layer 1 (the other networks' norms are still in flight)
        SPO_REIDX
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          bk[BKS * (BK_W2 + 3 * nt + 0)] = mW2[nt];
          bk[BKS * (BK_W2 + 3 * nt + 1)] = vW2[nt];
          bk[BKS * (BK_W2 + 3 * nt + 2)] = pW2[nt];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const AdamOut o_ = adam1(pW2[nt][r], gg2[nt][r], mW2[nt][r], vW2[nt][r], b1c, b2c, eps, step_size, inv_bc2s);
            mW2[nt][r] = o_.m; vW2[nt][r] = o_.v; lds[L::W2 + (orow + r) * LDH + 16 * nt + j] = o_.p;
          }
        }
        bmb2 = mb2; bvb2 = vb2;
        const AdamOut o_ = adam1(bpb2, ggb2, mb2, vb2, b1c, b2c, eps, step_size, inv_bc2s);
        mb2 = o_.m; vb2 = o_.v; lds[L::B2 + 16 * wave + j] = o_.p;
      }
      SPO_STAMP(9)
      if (s + 1 < nsteps) __syncthreads();                                  // Q2 of step s + 1: (speculative) W2 / b2 in place
      SPO_STAMP(4)
      // ---- the joint norm and the clip coefficient (the granules have had a forward layer's time to arrive)
      {
        if (wave == 0 && lane == 0 && s + 1 < nsteps) {                       // the loss value of this step (logging); the
          // last step has no Q2 barrier before this point: its value is written after the barrier that follows the loop
          const float loss_data = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
          const float pp = (red[88] + red[89]) + (red[90] + red[91]);
          a.losses[s * 3 + net] = is_actor ? -loss_data : loss_data + l2 * pp;
        }
        float mine = 0.f;
        const int ngr = 4 * a.n_nets;
        if (lane < ngr) {                                                     // lane k polls granule k (network k / 4, wave k % 4)
          __builtin_amdgcn_s_setprio(0);
          unsigned long long v = ld_granule(grow + lane);
          unsigned sp2 = 0;
          while ((unsigned)(v >> 32) != tag) {
            if (++sp2 > (1u << 22)) { *a.err = 1; break; }
            __builtin_amdgcn_s_sleep(4);
            v = ld_granule(grow + lane);
          }
          mine = __uint_as_float((unsigned)v);
          __builtin_amdgcn_s_setprio(SPO_HELPER_PRIO);
        }
        float total_sq = stale_sq;
        for (int kk = 0; kk < ngr; ++kk) total_sq += __shfl(mine, kk);        // fixed order: identical in every wave and workgroup
        const float norm = sqrtf(total_sq);
        coef = a.cfg.max_grad_norm / (norm + 1e-6f);                        // clip_grad_norm_ (torch): eps 1e-6
        coef = coef > 1.f ? 1.f : coef;
        stale_sq *= coef * coef;
      }
      SPO_STAMP(8)
      ++n_cons;
      if (coef != 1.f) {
        // ---- the clip is active (rare): layers 1 and 2 were updated with coefficient 1 -- restore them, redo them exactly,
        //      and make the main waves repeat L1 / L2 of the step they have started
        ++n_redo;
        redo_layers12(coef);
        if (wave == 0 && lane == 0) xw[18] = (int)((s + 1) & 0x3fffffff) + 1;
        redo_next = true;
      }
      {
        // ---- output layer (W3, b3, log_std) with the exact coefficient
        SPO_REIDX
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int addr = L::W3 + (4 * q + r) * LDH + 16 * wave + j;
          const AdamOut o_ = adam1(lds[addr], gg3[r] * coef, mW3[r], vW3[r], b1c, b2c, eps, step_size, inv_bc2s);
          mW3[r] = o_.m; vW3[r] = o_.v; lds[addr] = o_.p;
        }
        if (wave == 0) {
          const AdamOut o_ = adam1(lds[L::B3 + j], ggb3 * coef, mb3, vb3, b1c, b2c, eps, step_size, inv_bc2s);
          mb3 = o_.m; vb3 = o_.v; lds[L::B3 + j] = o_.p;
          if (is_actor) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ai = 4 * q + r;
              const AdamOut ol = adam1(red[128 + ai], dls[r] * coef, mls[r], vls[r], b1c, b2c, eps, step_size, inv_bc2s);
              mls[r] = ol.m; vls[r] = ol.v; red[128 + ai] = ol.p;
            }
          }
        }
      }
      SPO_STAMP(5)
      if (s + 1 < nsteps) {
        __syncthreads();                                                    // Xd of step s + 1: W3 / b3 / log_std updated, verdict out
        if (redo_next) {                                                    // the main waves repeat L1, Q2, L2, Xd
          __syncthreads();
          __syncthreads();
        }
        __syncthreads();                                                    // B_stage of step s + 1
      }
      spec = a.spec_mode != 0 && coef == 1.f;
    }
    SPO_STAMP(6)
    // rows / targets of minibatch s + 2: issued AFTER B_stage(s + 1), in flight while the main waves compute dW1, consumed right
    // after P1 -- the 22 registers are live only where the helper holds nothing but its optimiser state (issued before the
    // verdict code they cost 94 spilled registers)
    if (SPO_H_GATHER && s + 2 < nsteps && !h_fetched) h_fetch((int64_t)pinned(hsmp));
  }
  if (PROF && a.prof && tid == 256 && wg == a.n_nets - 1)
    for (int i = 0; i < NPHASE; ++i) a.prof[NPHASE + i] = pacc[i];
  if (tid == 256 && wg == 0) {
    atomicAdd(&g_upd_counters[0], (unsigned long long)nsteps); atomicAdd(&g_upd_counters[1], (unsigned long long)n_late);
    atomicAdd(&g_upd_counters[2], (unsigned long long)n_redo); atomicAdd(&g_upd_counters[3], (unsigned long long)n_cons);
  }
  // the last step's verdict (and a possible redo) is complete here; the barrier orders the final weight image before
  // the write-back below
  __syncthreads();
  if (wave == 0 && lane == 0 && nsteps > 0) {
    const int64_t sl = nsteps - 1;
    const int64_t reml = a.M - sl * B;
    const float inv_nl = 1.f / (float)(int)(reml < B ? reml : B);
    const float loss_data = ((red[0] + red[1]) + (red[2] + red[3])) * inv_nl;
    const float pp = (red[88] + red[89]) + (red[90] + red[91]);
    a.losses[sl * 3 + net] = is_actor ? -loss_data : loss_data + l2 * pp;
  }
#undef SPO_H_ELEM
#undef SPO_H_REDO
  {
    SPO_REIDX
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * nt + j;
        if (i < D) {
          const int idx = g.w1() + (orow + r) * D + i;
          a.theta[idx] = lds[L::W1 + (orow + r) * L::LD1 + i];
          SPO_ST_MV(idx, mW1[nt][r], vW1[nt][r])
        }
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = g.w2() + (orow + r) * HID + 16 * nt + j;
        a.theta[idx] = lds[L::W2 + (orow + r) * LDH + 16 * nt + j];
        SPO_ST_MV(idx, mW2[nt][r], vW2[nt][r])
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * q + r;
      if (o < OUT) {
        const int idx = g.w3() + o * HID + 16 * wave + j;
        a.theta[idx] = lds[L::W3 + o * LDH + 16 * wave + j];
        SPO_ST_MV(idx, mW3[r], vW3[r])
      }
      if (own_ls && o < A) {
        a.theta[ls_off + o] = red[128 + o];
        SPO_ST_MV(ls_off + o, mls[r], vls[r])
      }
    }
    if (own_b) {
      const int o = 16 * wave + j;
      a.theta[g.b1() + o] = lds[L::B1 + o]; SPO_ST_MV(g.b1() + o, mb1, vb1)
      a.theta[g.b2() + o] = lds[L::B2 + o]; SPO_ST_MV(g.b2() + o, mb2, vb2)
    }
    if (own_b3) { a.theta[g.b3() + j] = lds[L::B3 + j]; SPO_ST_MV(g.b3() + j, mb3, vb3) }
    if (tid == 256 && wg == 0 && a.stale_io) *a.stale_io = stale_sq;
  }
  }   // helper role
#undef SPO_REIDX
#undef SPO_STAMP
#undef SPO_SUB
}

template <int KIN, bool PROF = false, int XR = 0, int XRD = 2>
__global__ __launch_bounds__(512) void ppo_update_h_kernel(UpdArgs a) {
  if (blockIdx.x & 7) return;                          // placement hint, see ppo_update_kernel
  ppo_update_h_body<KIN, PROF, XR, XRD>(a, (int)(blockIdx.x >> 3));
}

// One-grid split form on the main + helper kernel (round 5, opt-in: SPO_CPO_SPLIT_FORM=h; cf. ppo_update_split_kernel): both halves of
// a two-way split of the minibatch in ONE launch -- workgroups [0, n_nets) run rank 0's arguments, [n_nets, 2 n_nets) rank 1's --
// exchanging the gradient layer by layer on the HELPER waves (packed words, one hand-off per stage).  Measured SLOWER than the
// four-wave split kernel (15.98 against 14.05 us per 128-row step, profiles/r05/helper_exchange_ab.txt): between P1 and Q2 the
// helper waves are already what the step waits for, so two hand-offs there cost more than one after the last gradient.
template <int KIN>
__global__ __launch_bounds__(512) void ppo_update_h_split_kernel(UpdArgs a0, UpdArgs a1) {
  if (blockIdx.x & 7) return;
  const int wg = (int)(blockIdx.x >> 3);
  if (split_census(a0, wg)) {
    if (wg < a0.n_nets) ppo_update_h_body<KIN, false, 2, 3>(a0, wg);
    else ppo_update_h_body<KIN, false, 2, 3>(a1, wg - a0.n_nets);
  } else {
    if (wg < a0.n_nets) ppo_update_h_body<KIN, false, 2, 2>(a0, wg);
    else ppo_update_h_body<KIN, false, 2, 2>(a1, wg - a0.n_nets);
  }
}

// Split form, second half: joint clip + Adam over the flat vector (one workgroup; P ~ 25k).
struct AdamArgs {
  float* theta; float* m; float* v; const float* grad; int64_t P; float gscale;
  float max_norm, lr_actor, lr_critic, b1, b2, eps; double pow_b1, pow_b2; int64_t actor_begin;
};
__global__ __launch_bounds__(1024) void clip_adam_kernel(AdamArgs a) {
  __shared__ float sh[16];
  const int tid = threadIdx.x;
  float sq = 0.f;
  for (int64_t i = tid; i < a.P; i += 1024) { const float g = a.grad[i] * a.gscale; sq += g * g; }
  sq = wave_sum(sq);
  if ((tid & 63) == 0) sh[tid >> 6] = sq;
  __syncthreads();
  float tot = 0.f;
  for (int k = 0; k < 16; ++k) tot += sh[k];
  float coef = a.max_norm / (sqrtf(tot) + 1e-6f);
  coef = coef > 1.f ? 1.f : coef;
  const double pw1 = a.pow_b1 * (double)a.b1, pw2 = a.pow_b2 * (double)a.b2;
  float bc2s, ss_a, ss_c, bc2s_again;
  adam_scalars(a.lr_actor, pw1, pw2, ss_a, bc2s);           // the same scalar forms as the persistent kernels
  adam_scalars(a.lr_critic, pw1, pw2, ss_c, bc2s_again);
  for (int64_t i = tid; i < a.P; i += 1024) {
    const float g = a.grad[i] * a.gscale * coef;
    const AdamOut o = adam1(a.theta[i], g, a.m[i], a.v[i], a.b1, a.b2, a.eps, i >= a.actor_begin ? ss_a : ss_c, bc2s);
    a.theta[i] = o.p; a.m[i] = o.m; a.v[i] = o.v;
  }
}

// Self-test of the exchange protocol on the same grid shape as the real kernel: every (rank, network) workgroup
// pushes small-integer patterns and checks the reduced values; result[0] = mismatches, result[1] = timeout flag.
__global__ __launch_bounds__(256, 1) void xr_selftest_kernel(int rank, int world, unsigned step0, int iters, int* result,
                                                             UpdArgs a) {
  __shared__ float dead;
  __shared__ unsigned long long regions[XR_MAX_WORLD];
  if (blockIdx.x & 7) return;
  int net = blockIdx.x >> 3;
  const int tid = threadIdx.x;
  if (a.grid_ranks == 2) {           // both ranks in one grid (41 blocks): workgroups 0..2 rank 0, 3..5 rank 1
    rank = net / 3; net -= 3 * rank;
    result += 2 * rank;
  }
  if (tid == 0) dead = 0.f;
  if (tid < XR_MAX_WORLD) regions[tid] = reinterpret_cast<unsigned long long>(a.xr_region[tid]);
  __syncthreads();
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned gtag = step0 + (unsigned)it + 1u;
    f4 pk[11];
#pragma unroll
    for (int v = 0; v < 11; ++v) {
      const float c = (float)((gtag + 3u * v + tid + 5u * net) & 15u);
      pk[v] = f4{(float)(rank + 1) + c, (float)(rank + 1) - c, c, (float)(rank + 1) * 2.f};
    }
    if (a.xr_algo == 1 && (world & (world - 1)) == 0) {
      if (SPO_XR_PACK16) xr_allreduce_rd16<11>(regions, rank, world, net, tid, gtag, pk, &dead, result + 1);
      else xr_allreduce_rd<11>(regions, rank, world, net, tid, gtag, pk, &dead, result + 1);
    }
    else if (SPO_XR_PACK16) xr_allreduce_rs16<11>(regions, rank, world, net, tid, gtag, pk, &dead, result + 1);
    else xr_allreduce<11, true>(regions, rank, world, net, tid, gtag, pk, &dead, result + 1);
    const float tri = 0.5f * (float)(world + 1);
#pragma unroll
    for (int v = 0; v < 11; ++v) {
      const float c = (float)((gtag + 3u * v + tid + 5u * net) & 15u);
      const f4 want = {tri + c, tri - c, c, 2.f * tri};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool b_ = fabsf(pk[v][e] - want[e]) > 1e-4f;
        bad += b_ ? 1 : 0;
#ifdef SPO_XR_DEBUG_MISMATCH
        if (b_) {
          const unsigned long long slot = atomicAdd(&g_xr_prof[7], 1ull);
          if (slot < 7) g_xr_prof[slot] = ((unsigned long long)(rank * 3 + net) << 56) | ((unsigned long long)tid << 40) | ((unsigned long long)(v * 4 + e) << 32) |
                                          (unsigned long long)__float_as_uint(pk[v][e]);
        }
#endif
      }
    }
  }
  if (bad) atomicAdd(result, bad);
}

int pick_kin(int D) { return D <= 16 ? 16 : D <= 32 ? 32 : D <= 64 ? 64 : 128; }


unsigned long long* g_prof_buf = nullptr;

// Scratch of the main + helper kernel: its clip backup rows (L2-resident, read back only on a clipped step) and the
// cross-workgroup norm granules.  One block per (device, stream), allocated on first use and kept: launches on one stream are
// ordered and may share a block, launches on different streams (two engines of one process, the ranks of
// tools/p2p_loopback_bench.py) get different blocks, so concurrent launches never see each other's granules or backups.
struct HScratch { int dev; void* stream; int idx; char* base; };
constexpr size_t H_BACKUP_BYTES = sizeof(float4) * UPD_BACKUP_ROWS * 3 * 512, H_SLOT_BYTES = sizeof(unsigned long long) * 32;
constexpr int H_SCRATCH_MAX = 256;
HScratch g_hs[H_SCRATCH_MAX];
int g_hs_n = 0;
std::mutex g_hs_mu;
int h_scratch_for(hipStream_t st, float** backup, unsigned long long** slots, int idx = 0) {
  // idx: 0 for ordinary launches; the one-grid split form keeps a second block (idx 1) for its second rank
  const int dev = current_device_slot();
  std::lock_guard<std::mutex> lk(g_hs_mu);
  for (int i = 0; i < g_hs_n; ++i)
    if (g_hs[i].dev == dev && g_hs[i].stream == (void*)st && g_hs[i].idx == idx) {
      *backup = reinterpret_cast<float*>(g_hs[i].base);
      *slots = reinterpret_cast<unsigned long long*>(g_hs[i].base + H_BACKUP_BYTES);
      return 0;
    }
  if (g_hs_n == H_SCRATCH_MAX)
    return spo::fail(-1, "update kernel: more than %d (device, stream) pairs hold update scratch in this process; call "
                         "spo_update_scratch_release(stream) for streams that are gone", H_SCRATCH_MAX);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return spo::fail(-1, "update kernel: first launch on a stream under capture (the scratch block is allocated on first use: "
                         "launch once outside the capture)");
  void* p = nullptr;
  if (int rc = spo::hip_check(hipMalloc(&p, H_BACKUP_BYTES + H_SLOT_BYTES), "hipMalloc(update scratch)")) return rc;
  g_hs[g_hs_n++] = HScratch{dev, (void*)st, idx, static_cast<char*>(p)};
  *backup = reinterpret_cast<float*>(p);
  *slots = reinterpret_cast<unsigned long long*>(static_cast<char*>(p) + H_BACKUP_BYTES);
  return 0;
}

struct SplitLocal { int dev; void* stream; char* base; unsigned tag; };
constexpr int SPLIT_LOCAL_MAX = 8;
SplitLocal g_split_local[SPLIT_LOCAL_MAX] = {};
int g_split_local_n = 0;
std::mutex g_split_local_mu;
int split_local_release(int dev, void* stream_or_null, int all) {
  std::lock_guard<std::mutex> lk(g_split_local_mu);
  int freed = 0;
  for (int i = 0; i < g_split_local_n;) {
    if (g_split_local[i].dev == dev && (all || g_split_local[i].stream == stream_or_null)) {
      (void)spo::hip_check(hipFree(g_split_local[i].base), "hipFree(split exchange)");
      g_split_local[i] = g_split_local[--g_split_local_n];
      ++freed;
    } else ++i;
  }
  return freed;
}

}  // namespace
// Releases the scratch block of (current device, stream) -- or of every stream of the current device when stream_or_null is
// NULL and all != 0.  The block is keyed by the raw stream handle, so a process that keeps creating and destroying streams
// should release a stream's block before destroying it (a recycled handle would otherwise silently reuse the block, and after
// H_SCRATCH_MAX distinct pairs launches fail).  hipFree synchronises the device: not for the hot path.
extern "C" int spo_update_scratch_release(void* stream_or_null, int all) {
  const int dev = spo::current_device_slot();
  std::lock_guard<std::mutex> lk(g_hs_mu);
  int freed = 0;
  for (int i = 0; i < g_hs_n;) {
    if (g_hs[i].dev == dev && (all || g_hs[i].stream == stream_or_null)) {
      if (int rc = spo::hip_check(hipFree(g_hs[i].base), "hipFree(update scratch)")) return rc;
      g_hs[i] = g_hs[--g_hs_n];
      ++freed;
    } else ++i;
  }
  freed += split_local_release(dev, stream_or_null, all);
  freed += spo::ks_scratch_release(dev, stream_or_null, all);
  freed += spo::rs_scratch_release(dev, stream_or_null, all);
  return freed;
}
namespace {
using namespace spo;
// SPO_UPDATE_FORM: 0 = four-wave kernel everywhere, 2 = main + helper waves where that form applies (persistent PPO step,
// clipped-surrogate loss, no in-kernel cross-rank exchange, batch <= 64, obs <= 64), 3 (default, round 6) = the row-split kernel
// (update_rs.hip) for the one-GPU PPO-Lagrangian step and the critic fit where spo_update_rs_supported, form 2 elsewhere.
inline int update_form() {
  static const int v = [] { const char* e = getenv("SPO_UPDATE_FORM"); return e ? atoi(e) : 3; }();
  return v;
}

// SPO_P2P_HELPER=1 (opt-in): the data-parallel step with its exchange on the HELPER waves of the main + helper kernel (packed words)
inline int helper_xr_mode() {
  static const int v = [] { const char* e = getenv("SPO_P2P_HELPER"); return e ? atoi(e) : 0; }();
  return v;
}

template <int K, bool PROF = false, int XR = 0, int XRD = 2>
int launch_update_h(const UpdArgs& a_in, int blocks, hipStream_t st) {
  UpdArgs a = a_in;
  if (int rc = h_scratch_for(st, &a.backup, &a.slots)) return rc;
  {
    static const int spec_env = [] { const char* e = getenv("SPO_UPDATE_SPEC"); return e ? atoi(e) : 1; }();
    a.spec_mode = spec_env;
  }
  // [parity][network][helper wave] norm granules of this form (tags restart at 1 with every launch)
  if (int rc = spo::hip_check(hipMemsetAsync(a.slots, 0, H_SLOT_BYTES, st), "hipMemsetAsync(update scratch slots)")) return rc;
  const size_t sh = UpdHLds<K>::SIZE * sizeof(float);
  static bool attr_done[SPO_MAX_DEVICES] = {};
  const int dslot = current_device_slot();
  if (!attr_done[dslot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_update_h_kernel<K, PROF, XR, XRD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(update_h)");
    attr_done[dslot] = true;
  }
  hipLaunchKernelGGL((ppo_update_h_kernel<K, PROF, XR, XRD>), dim3(8 * (blocks - 1) + 1), dim3(512), sh, st, a);
  return 0;
}

template <bool PERSIST, int AMODE = 0, int XR = 0>
int launch_update(const UpdArgs& a, int blocks, hipStream_t st) {
  const int kin = pick_kin(a.cfg.obs_dim);
  if (PERSIST && AMODE == 0 && XR == 0 && kin <= 64 && a.cfg.batch <= 64 && update_form() >= 2) {
    if (a.prof && kin == 64) return launch_update_h<64, true>(a, blocks, st);
    if (kin == 16) return launch_update_h<16>(a, blocks, st);
    if (kin == 32) return launch_update_h<32>(a, blocks, st);
    return launch_update_h<64>(a, blocks, st);
  }
  if (PERSIST && AMODE == 0 && a.prof && kin == 64) {
    const size_t sh = UpdLds<64>::SIZE * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_update_kernel<64, true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(update prof)");
    hipLaunchKernelGGL((ppo_update_kernel<64, true, true>), dim3(8 * (blocks - 1) + 1), dim3(256), sh, st, a);
    return 0;
  }
#define SPO_LAUNCH(K)                                                                                   \
  {                                                                                                     \
    const size_t sh = UpdLds<K>::SIZE * sizeof(float);                                                  \
    static bool attr_done[SPO_MAX_DEVICES] = {};                                                        \
    const int dslot = current_device_slot();                                                            \
    if (!attr_done[dslot]) {                                                                            \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_update_kernel<K, PERSIST, false, AMODE, XR>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);          \
      if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(update)");                     \
      attr_done[dslot] = true;                                                                          \
    }                                                                                                   \
    hipLaunchKernelGGL((ppo_update_kernel<K, PERSIST, false, AMODE, XR>), dim3(PERSIST ? 8 * (blocks - 1) + 1 : blocks), \
                       dim3(256), sh, st, a);                                                          \
  }
  if (kin == 16) SPO_LAUNCH(16) else if (kin == 32) SPO_LAUNCH(32) else if (kin == 64) SPO_LAUNCH(64) else SPO_LAUNCH(128)
#undef SPO_LAUNCH
  return 0;
}

int check_cfg(const spo_ppo_cfg* c) {
  if (!c) return spo::fail(-1, "update: cfg is NULL");
  if (c->obs_dim < 1 || c->obs_dim > SPO_MAX_OBS)
    return spo::fail(-2, "update: obs_dim %d outside [1,%d]", c->obs_dim, SPO_MAX_OBS);
  if (c->act_dim < 1 || c->act_dim > SPO_MAX_ACT) return spo::fail(-2, "update: act_dim %d outside [1,16]", c->act_dim);
  if (c->batch < 1) return spo::fail(-2, "update: batch must be >= 1");
  return 0;
}

}  // namespace

extern "C" int spo_ppo_lag_update_iter(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host,
                                       const float* obs, const float* act, const float* logp_old,
                                       const float* target_r, const float* target_c, const float* adv,
                                       const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host, float* losses_out,
                                       void* sync_ws, void* stream) {
  if (int rc = check_cfg(cfg_host)) return rc;
  SPO_REQUIRE(theta && adam_m && adam_v && obs && act && logp_old && target_r && target_c && adv && perm &&
                  losses_out && sync_ws, "update_iter: null pointer");
  SPO_REQUIRE(M > 0 && adam_step_host >= 0, "update_iter: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws, 0, 64, st), "hipMemsetAsync(sync_ws)")) return rc;
  if (update_form() >= 3 && spo_update_rs_supported(cfg_host->obs_dim, cfg_host->act_dim, cfg_host->batch, 3)) {
    if (int rc = spo::rs_update_launch(theta, adam_m, adam_v, adam_step_host, obs, act, logp_old, target_r, target_c, adv, perm, M,
                                       cfg_host, 3, nullptr, losses_out, sync_ws, g_prof_buf, stream)) return rc;
    SPO_LAUNCH_CHECK("spo_ppo_lag_update_iter (row-split)");
    return 0;
  }
  UpdArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = perm; a.M = M; a.cfg = *cfg_host; a.losses = losses_out;
  a.slots = reinterpret_cast<unsigned long long*>(sync_ws);
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.pow_b1_actor = a.pow_b1; a.pow_b2_actor = a.pow_b2;
  a.first_net = 0; a.n_nets = 3; a.stale_sq = 0.f; a.stale_io = nullptr;
  a.flat_grad = nullptr; a.mean_count = 0; a.prof = g_prof_buf;
  if (int rc = launch_update<true>(a, 3, st)) return rc;
  SPO_LAUNCH_CHECK("spo_ppo_lag_update_iter");
  return 0;
}

extern "C" int spo_critic_fit_iter(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host,
                                   const float* obs, const float* target_r, const float* target_c,
                                   const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host, float* stale_sq_io,
                                   float* losses_out, void* sync_ws, void* stream) {
  if (int rc = check_cfg(cfg_host)) return rc;
  SPO_REQUIRE(theta && adam_m && adam_v && obs && target_r && target_c && perm && losses_out && sync_ws,
              "critic_fit_iter: null pointer");
  SPO_REQUIRE(M > 0 && adam_step_host >= 0, "critic_fit_iter: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws, 0, 64, st), "hipMemsetAsync(sync_ws)")) return rc;
  if (update_form() >= 3 && spo_update_rs_supported(cfg_host->obs_dim, cfg_host->act_dim, cfg_host->batch, 2)) {
    if (int rc = spo::rs_update_launch(theta, adam_m, adam_v, adam_step_host, obs, nullptr, nullptr, target_r, target_c, nullptr,
                                       perm, M, cfg_host, 2, stale_sq_io, losses_out, sync_ws, nullptr, stream)) return rc;
    SPO_LAUNCH_CHECK("spo_critic_fit_iter (row-split)");
    return 0;
  }
  UpdArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.tgt_r = target_r; a.tgt_c = target_c; a.perm = perm; a.M = M; a.cfg = *cfg_host;
  a.losses = losses_out;
  a.slots = reinterpret_cast<unsigned long long*>(sync_ws);
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.pow_b1_actor = a.pow_b1; a.pow_b2_actor = a.pow_b2;
  a.first_net = 0; a.n_nets = 2; a.stale_sq = 0.f; a.stale_io = stale_sq_io;
  if (int rc = launch_update<true>(a, 2, st)) return rc;
  SPO_LAUNCH_CHECK("spo_critic_fit_iter");
  return 0;
}

extern "C" int spo_update_iter_ex(float* theta, float* adam_m, float* adam_v, int64_t adam_step_critics_host,
                                  int64_t adam_step_actor_host, const float* obs, const float* act,
                                  const float* logp_old, const float* target_r, const float* target_c,
                                  const float* adv, const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host,
                                  int actor_loss, const float* old_mean, const float* old_std, float kl_bound,
                                  float pg_coef, int actor_only, float* losses_out, void* sync_ws, void* stream) {
  if (int rc = check_cfg(cfg_host)) return rc;
  SPO_REQUIRE(theta && adam_m && adam_v && obs && act && logp_old && adv && perm && losses_out && sync_ws,
              "update_iter_ex: null pointer");
  SPO_REQUIRE(actor_loss == SPO_ACTOR_LOSS_CLIP || actor_loss == SPO_ACTOR_LOSS_KL_PENALTY,
              "update_iter_ex: unknown actor_loss %d", actor_loss);
  SPO_REQUIRE(actor_loss == SPO_ACTOR_LOSS_CLIP || (old_mean && old_std), "update_iter_ex: old distribution is NULL");
  SPO_REQUIRE(actor_only || (target_r && target_c), "update_iter_ex: critic targets are NULL");
  SPO_REQUIRE(M > 0 && adam_step_critics_host >= 0 && adam_step_actor_host >= 0, "update_iter_ex: bad sizes");
  // the indicator fraction couples every sample of a minibatch: one 64-column pass must hold the whole minibatch
  SPO_REQUIRE(actor_loss == SPO_ACTOR_LOSS_CLIP || cfg_host->batch <= 64,
              "update_iter_ex: KL-penalty loss with batch_size %d > 64 is not supported", cfg_host->batch);
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws, 0, 64, st), "hipMemsetAsync(sync_ws)")) return rc;
  UpdArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = perm; a.M = M; a.cfg = *cfg_host; a.losses = losses_out;
  a.slots = reinterpret_cast<unsigned long long*>(sync_ws);
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_critics_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_critics_host);
  a.pow_b1_actor = pow((double)cfg_host->beta1, (double)adam_step_actor_host);
  a.pow_b2_actor = pow((double)cfg_host->beta2, (double)adam_step_actor_host);
  a.first_net = actor_only ? 2 : 0; a.n_nets = actor_only ? 1 : 3; a.stale_sq = 0.f; a.stale_io = nullptr;
  a.old_mean = old_mean; a.old_std = old_std; a.kl_bound = kl_bound; a.pg_coef = pg_coef;
  if (actor_loss == SPO_ACTOR_LOSS_KL_PENALTY) {
    if (int rc = launch_update<true, 1>(a, a.n_nets, st)) return rc;
  } else {
    if (int rc = launch_update<true, 0>(a, a.n_nets, st)) return rc;
  }
  SPO_LAUNCH_CHECK("spo_update_iter_ex");
  return 0;
}

// ---- form of the in-kernel exchange (SPO_XR_FORM_*, include/safepo_hip.h).  spo_p2p_select_form(form) pins it for the process --
// the start-up auto-tune of safepo.parallel.PeerExchange times every form on the actual topology and pins the fastest (VERDICT
// r04: loopback on one GPU cannot rank forms whose cost is link parallelism) --, -1 returns to the policy below: the environment
// (SPO_P2P_A2A=1, SPO_P2P_HELPER=1, SPO_P2P_ALGO=twophase | doubling), else recursive doubling at 2 and 4 ranks (one / two
// hand-offs) and the packed reduce-scatter + all-gather at 8 ranks and at worlds that are not a power of two (loopback on one GPU,
// profiles/r05/helper_exchange_ab.txt: doubling 14.3 / 16.7 / 21.4 us per step at 2 / 4 / 8 ranks, two-phase 17.8 / 18.0 / 19.1).
static int g_xr_form_override = -1;
extern "C" int spo_p2p_select_form(int form) {
  SPO_REQUIRE(form >= -1 && form <= SPO_XR_FORM_ROW_SPLIT, "p2p_select_form: unknown form %d", form);
  g_xr_form_override = form;
  return 0;
}
static bool xr_form_valid(int form, int world) {
  const bool pow2 = (world & (world - 1)) == 0;
  if (world < 2 || world > XR_MAX_WORLD) return false;
  switch (form) {
    case SPO_XR_FORM_TWOPHASE: return true;
    case SPO_XR_FORM_DOUBLING: return pow2;
    case SPO_XR_FORM_HELPER_A2A: case SPO_XR_FORM_HELPER_DOUBLING: return world == 2 || world == 4 || world == 8;
    case SPO_XR_FORM_ROW_SPLIT: return (world == 2 || world == 4 || world == 8) && update_form() >= 3;
    default: return false;
  }
}
extern "C" int spo_p2p_form_valid(int form, int world) { return xr_form_valid(form, world) ? 1 : 0; }
static int xr_form(int world) {
  if (g_xr_form_override >= 0 && xr_form_valid(g_xr_form_override, world)) return g_xr_form_override;
  static const bool a2a = [] { const char* e = getenv("SPO_P2P_A2A"); return e && e[0] == '1'; }();
  const char* algo = getenv("SPO_P2P_ALGO");
  const bool pow2 = (world & (world - 1)) == 0;
  if (algo && !strcmp(algo, "twophase")) return SPO_XR_FORM_TWOPHASE;
  if (algo && !strcmp(algo, "rowsplit") && xr_form_valid(SPO_XR_FORM_ROW_SPLIT, world)) return SPO_XR_FORM_ROW_SPLIT;
  if (a2a && xr_form_valid(SPO_XR_FORM_HELPER_A2A, world)) return SPO_XR_FORM_HELPER_A2A;
  if (helper_xr_mode() && xr_form_valid(SPO_XR_FORM_HELPER_DOUBLING, world)) return SPO_XR_FORM_HELPER_DOUBLING;
  if (algo && !strcmp(algo, "doubling") && pow2) return SPO_XR_FORM_DOUBLING;
  // round 6: where it exists (2 / 4 / 8 ranks, SPO_UPDATE_FORM >= 3) the row-split form is the default -- loopback 10.9 / 14.8 / 15.1 us
  // per step at 2 / 4 / 8 ranks against 14.1 / 16.6 / 18.3 for the policy below (profiles/r06/p2p_loopback.txt); shapes the row-split
  // kernel does not take run the four-wave kernel under that policy (xr_four_wave_form)
  if (!(algo && *algo) && xr_form_valid(SPO_XR_FORM_ROW_SPLIT, world)) return SPO_XR_FORM_ROW_SPLIT;
  return (world <= 4 && pow2) ? SPO_XR_FORM_DOUBLING : SPO_XR_FORM_TWOPHASE;
}
// the form of the four-wave / main + helper kernels when the selected form is the row-split kernel's and the shape is not its own
static int xr_four_wave_form(int world) {
  const int f = xr_form(world);
  if (f != SPO_XR_FORM_ROW_SPLIT) return f;
  return (world <= 4 && (world & (world - 1)) == 0) ? SPO_XR_FORM_DOUBLING : SPO_XR_FORM_TWOPHASE;
}
extern "C" int spo_p2p_current_form(int world) { return xr_form(world); }

// ---- data-parallel persistent form: cross-rank exchange regions and the per-iteration launch
static int fill_xr(UpdArgs& a, int rank, int world, void* const* regions, unsigned step0) {
  SPO_REQUIRE(world >= 2 && world <= XR_MAX_WORLD && rank >= 0 && rank < world, "p2p: bad rank/world %d/%d", rank, world);
  SPO_REQUIRE(regions != nullptr, "p2p: regions is NULL");
  for (int r = 0; r < world; ++r) SPO_REQUIRE(regions[r] != nullptr, "p2p: region of rank %d is NULL", r);
  a.xr_rank = rank; a.xr_world = world; a.xr_step0 = step0;
  a.xr_algo = xr_four_wave_form(world) == SPO_XR_FORM_TWOPHASE ? 0 : 1;   // (the protocol of the four-wave kernels and of the self-test)
  { const char* dbg = getenv("SPO_A2A_DEBUG"); a.xr_debug = dbg ? atoi(dbg) : 0; }
  for (int r = 0; r < XR_MAX_WORLD; ++r) a.xr_region[r] = r < world ? regions[r] : nullptr;
  return 0;
}

extern "C" int spo_debug_update_counters(unsigned long long* out4_host, int reset) {
  SPO_REQUIRE(out4_host, "update_counters: null pointer");
  if (int rc = spo::hip_check(hipMemcpyFromSymbol(out4_host, HIP_SYMBOL(g_upd_counters), 32), "hipMemcpyFromSymbol")) return rc;
  {
    unsigned long long rs2[2] = {0, 0};                                 // the row-split form's steps / late redos count with the first two
    if (int rc = spo::rs_counters(rs2, reset)) return rc;
    out4_host[0] += rs2[0]; out4_host[1] += rs2[1];
  }
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    return spo::hip_check(hipMemcpyToSymbol(HIP_SYMBOL(g_upd_counters), z, 32), "hipMemcpyToSymbol");
  }
  return 0;
}

extern "C" int spo_debug_xr_profile(unsigned long long* out8_host, int reset) {
  SPO_REQUIRE(out8_host, "xr_profile: null pointer");
  if (int rc = spo::hip_check(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_xr_prof), 64), "hipMemcpyFromSymbol")) return rc;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return spo::hip_check(hipMemcpyToSymbol(HIP_SYMBOL(g_xr_prof), z, 64), "hipMemcpyToSymbol");
  }
  return 0;
}

extern "C" int64_t spo_p2p_region_bytes(void) { return (int64_t)XR_REGION_BYTES; }

extern "C" int spo_p2p_alloc(void** region_out, void* ipc_handle64_out) {
  SPO_REQUIRE(region_out && ipc_handle64_out, "p2p_alloc: null pointer");
  void* p = nullptr;
  // SPO_P2P_MEM = uncached (default) | finegrained | coarse (single-GPU experiments only: not coherent across GPUs)
  const char* kind = getenv("SPO_P2P_MEM");
  hipError_t e = hipErrorUnknown;
  if (kind && !strcmp(kind, "coarse")) {
    e = hipMalloc(&p, XR_REGION_BYTES);
  } else {
    if (!(kind && !strcmp(kind, "finegrained"))) e = hipExtMallocWithFlags(&p, XR_REGION_BYTES, hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipExtMallocWithFlags(&p, XR_REGION_BYTES, hipDeviceMallocFinegrained);
    }
  }
  if (e != hipSuccess) return spo::hip_check(e, "hipExtMallocWithFlags(p2p region)");
  if (int rc = spo::hip_check(hipMemset(p, 0, XR_REGION_BYTES), "hipMemset(p2p region)")) { (void)hipFree(p); return rc; }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) { (void)hipFree(p); return spo::hip_check(e, "hipIpcGetMemHandle"); }
  std::memcpy(ipc_handle64_out, &h, 64);
  *region_out = p;
  return 0;
}

extern "C" int spo_p2p_open(const void* ipc_handle64, void** region_out) {
  SPO_REQUIRE(ipc_handle64 && region_out, "p2p_open: null pointer");
  hipIpcMemHandle_t h;
  std::memcpy(&h, ipc_handle64, 64);
  void* p = nullptr;
  if (int rc = spo::hip_check(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) return rc;
  *region_out = p;
  return 0;
}

extern "C" int spo_p2p_close(void* peer_region) {
  if (!peer_region) return 0;
  return spo::hip_check(hipIpcCloseMemHandle(peer_region), "hipIpcCloseMemHandle");
}

extern "C" int spo_p2p_free(void* own_region) {
  if (!own_region) return 0;
  return spo::hip_check(hipFree(own_region), "hipFree(p2p region)");
}

extern "C" int spo_p2p_selftest(int rank, int world, void* const* regions, uint32_t step0, int iters,
                                int32_t* result2_dev, void* stream) {
  SPO_REQUIRE(result2_dev && iters > 0, "p2p_selftest: bad args");
  UpdArgs a{};
  if (int rc = fill_xr(a, rank, world, regions, step0)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(result2_dev, 0, 8, st), "hipMemsetAsync(selftest)")) return rc;
  hipLaunchKernelGGL(xr_selftest_kernel, dim3(17), dim3(256), 0, st, rank, world, step0, iters, result2_dev, a);
  SPO_LAUNCH_CHECK("spo_p2p_selftest");
  return 0;
}

extern "C" int spo_ppo_lag_update_iter_dp(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host,
                                          const float* obs, const float* act, const float* logp_old,
                                          const float* target_r, const float* target_c, const float* adv,
                                          const int32_t* perm, int64_t M, const spo_ppo_cfg* cfg_host, float* losses_out,
                                          void* sync_ws, int rank, int world, void* const* regions, uint32_t step0,
                                          void* stream) {
  if (int rc = check_cfg(cfg_host)) return rc;
  SPO_REQUIRE(theta && adam_m && adam_v && obs && act && logp_old && target_r && target_c && adv && perm &&
                  losses_out && sync_ws, "update_iter_dp: null pointer");
  SPO_REQUIRE(M > 0 && adam_step_host >= 0, "update_iter_dp: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws, 0, 64, st), "hipMemsetAsync(sync_ws)")) return rc;
  UpdArgs a{};
  if (int rc = fill_xr(a, rank, world, regions, step0)) return rc;
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = perm; a.M = M; a.cfg = *cfg_host; a.losses = losses_out;
  a.slots = reinterpret_cast<unsigned long long*>(sync_ws);
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.pow_b1_actor = a.pow_b1; a.pow_b2_actor = a.pow_b2;
  a.first_net = 0; a.n_nets = 3; a.stale_sq = 0.f; a.stale_io = nullptr;
  int rc = 0;
  const int kin = pick_kin(cfg_host->obs_dim);
  // SPO_XR_FORM_ROW_SPLIT (round 6): the row-split kernel with a one-hand-off all-to-all of the rank's gradient behind the row
  // groups' L2 hand-off (csrc/update_rs.hip, XW)
  if (xr_form(world) == SPO_XR_FORM_ROW_SPLIT && spo_update_rs_supported(cfg_host->obs_dim, cfg_host->act_dim, cfg_host->batch, 3)) {
    if (int rc2 = spo::rs_update_launch_dp(theta, adam_m, adam_v, adam_step_host, obs, act, logp_old, target_r, target_c, adv, perm, M,
                                           cfg_host, losses_out, sync_ws, rank, world, regions, step0, RSX_OFF, stream)) return rc2;
    SPO_LAUNCH_CHECK("spo_ppo_lag_update_iter_dp (row-split)");
    return 0;
  }
  // SPO_P2P_A2A=1 (opt-in): main + helper kernel with the flag-based all-to-all exchange on the helper waves.  One exchange
  // round at any world size, layer by layer beside the main waves' MFMAs -- but in single-GPU loopback (all ranks sharing
  // one memory system) it measured SLOWER than recursive doubling on the four-wave kernel (18.6 / 23.9 / 33.5 against
  // 15.9 / 18.8 / 24.0 us per step at 2 / 4 / 8 ranks): a store acknowledgement plus a flag flight plus the row loads per
  // stage cost more than three tagged-word hand-offs there.  Kept for measurement on a real xGMI node.
  // SPO_P2P_HELPER=1 (opt-in): the main + helper kernel with recursive doubling ON THE HELPER WAVES -- layer 1 exchanged
  // while the main waves still compute dW2 / dW3, layers 2 / 3 while they settle the next minibatch and run layer 1; round 5:
  // packed 16-byte words, one poll batch per stage (xr_rd16_flat).  Loopback: 16.2 / 22.4 / 27.9 us per step at 2 / 4 / 8 ranks
  // against 14.3 / 16.7 / 19.1 for the default below (four-wave kernel, ONE exchange of the whole gradient per step): between P1
  // and Q2 the helper waves already gate the step, so hand-offs placed there are exposed in full and there are two per step.
  const int form = xr_four_wave_form(world);
  const bool a2a = form == SPO_XR_FORM_HELPER_A2A;
  a.xr_helper_rd = a2a ? 0 : 1;
  if ((a2a || form == SPO_XR_FORM_HELPER_DOUBLING) && kin <= 64 && cfg_host->batch <= 64 && update_form() >= 2) {
    // main + helper form, exchange on the helper waves
#define SPO_H_XR(K, D) (world == 2 ? launch_update_h<K, false, 2, D>(a, 3, st) : world == 4 ? launch_update_h<K, false, 4, D>(a, 3, st) \
                                                                                               : launch_update_h<K, false, 8, D>(a, 3, st))
    if (a2a) rc = kin == 16 ? SPO_H_XR(16, 0) : kin == 32 ? SPO_H_XR(32, 0) : SPO_H_XR(64, 0);
    else rc = kin == 16 ? SPO_H_XR(16, 2) : kin == 32 ? SPO_H_XR(32, 2) : SPO_H_XR(64, 2);
#undef SPO_H_XR
  } else if (a.xr_algo == 1 && (world & (world - 1)) == 0) rc = launch_update<true, 0, 2>(a, 3, st);
  else rc = launch_update<true, 0, 1>(a, 3, st);
  if (rc) return rc;
  SPO_LAUNCH_CHECK("spo_ppo_lag_update_iter_dp");
  return 0;
}

// CPO critic fit for one rank of a data-parallel job: spo_critic_fit_iter with the in-kernel gradient exchange
// (cpo.py:541-571 over env shards; the stale actor-gradient norm is identical on every rank).
extern "C" int spo_critic_fit_iter_dp(float* theta, float* adam_m, float* adam_v, int64_t adam_step_host, const float* obs,
                                      const float* target_r, const float* target_c, const int32_t* perm, int64_t M,
                                      const spo_ppo_cfg* cfg_host, float* stale_sq_io, float* losses_out, void* sync_ws,
                                      int rank, int world, void* const* regions, uint32_t step0, void* stream) {
  if (int rc = check_cfg(cfg_host)) return rc;
  SPO_REQUIRE(theta && adam_m && adam_v && obs && target_r && target_c && perm && losses_out && sync_ws,
              "critic_fit_iter_dp: null pointer");
  SPO_REQUIRE(M > 0 && adam_step_host >= 0, "critic_fit_iter_dp: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws, 0, 64, st), "hipMemsetAsync(sync_ws)")) return rc;
  UpdArgs a{};
  if (int rc = fill_xr(a, rank, world, regions, step0)) return rc;
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.tgt_r = target_r; a.tgt_c = target_c; a.perm = perm; a.M = M; a.cfg = *cfg_host;
  a.losses = losses_out;
  a.slots = reinterpret_cast<unsigned long long*>(sync_ws);
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.pow_b1_actor = a.pow_b1; a.pow_b2_actor = a.pow_b2;
  a.first_net = 0; a.n_nets = 2; a.stale_sq = 0.f; a.stale_io = stale_sq_io;
  int rc = 0;
  const int kin = pick_kin(cfg_host->obs_dim);
  const int form = xr_four_wave_form(world);  // (the all-to-all form has no two-network instantiation: the helpers' doubling stands in)
  a.xr_helper_rd = 1;
  if ((form == SPO_XR_FORM_HELPER_A2A || form == SPO_XR_FORM_HELPER_DOUBLING) && kin <= 64 && cfg_host->batch <= 64 && update_form() >= 2) {
    // main + helper form with recursive doubling on the helper waves (see spo_ppo_lag_update_iter_dp)
#define SPO_H_XR(K) (world == 2 ? launch_update_h<K, false, 2, 2>(a, 2, st) : world == 4 ? launch_update_h<K, false, 4, 2>(a, 2, st) \
                                                                                            : launch_update_h<K, false, 8, 2>(a, 2, st))
    if (kin == 16) rc = SPO_H_XR(16);
    else if (kin == 32) rc = SPO_H_XR(32);
    else rc = SPO_H_XR(64);
#undef SPO_H_XR
  } else if (a.xr_algo == 1 && (world & (world - 1)) == 0) rc = launch_update<true, 0, 2>(a, 2, st);
  else rc = launch_update<true, 0, 1>(a, 2, st);
  if (rc) return rc;
  SPO_LAUNCH_CHECK("spo_critic_fit_iter_dp");
  return 0;
}

// ---- one-grid split critic fit (single GPU): a 128-row minibatch as two 64-row halves on TWO workgroup pairs of ONE launch.
// The data-parallel machinery above shards a step over ranks; here both "ranks" are workgroups of the same grid (rank 0:
// workgroups 0, 1 = reward / cost critic on rows perm0; rank 1: workgroups 2, 3 on rows perm1), each with its own replica of
// the parameters and moments, exchanging the gradient through the two exchange regions with the same tagged-word protocol.
// One grid of four workgroups is co-resident by construction, so -- unlike two launches on two streams (round 1 / 2: the
// second stream could land on the first one's hardware queue and never run beside it) -- the form never depends on HIP's
// queue assignment.  cpo.py:534-571: identical arithmetic to the mean over the 128 rows up to the order of the sums.
// Cached pair of exchange regions + census words of the one-grid split form (round 5), one block per (device, stream),
// allocated on first use, released by spo_update_scratch_release.  Cleared by every launch (tags restart with every engine;
// 13 MB at HBM speed is ~5 us against a 0.5 s launch).
static int split_local_for(hipStream_t st, UpdArgs& a, UpdArgs& b) {
  static const bool off = [] { const char* e = getenv("SPO_CPO_SPLIT_L2"); return e && !strcmp(e, "0"); }();
  if (off) return 0;
  const int dev = current_device_slot();
  std::lock_guard<std::mutex> lk(g_split_local_mu);
  SplitLocal* sl = nullptr;
  for (int i = 0; i < g_split_local_n; ++i)
    if (g_split_local[i].dev == dev && g_split_local[i].stream == (void*)st) sl = &g_split_local[i];
  if (!sl) {
    if (g_split_local_n == SPLIT_LOCAL_MAX) return 0;               // (table full: the uncached regions, as before round 5)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;        // (allocation + memset would invalidate a capture)
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
      return spo::fail(-1, "split critic fit: first launch on a stream under capture (the cached exchange regions are allocated on "
                           "first use: launch once outside the capture)");
    void* p = nullptr;
    if (int rc = spo::hip_check(hipMalloc(&p, 2 * XR_REGION_BYTES + 256), "hipMalloc(split exchange, cached)")) return rc;
    if (int rc = spo::hip_check(hipMemset(p, 0, 2 * XR_REGION_BYTES + 256), "hipMemset(split exchange)")) { (void)hipFree(p); return rc; }
    g_split_local[g_split_local_n] = SplitLocal{dev, (void*)st, static_cast<char*>(p), 1u};
    sl = &g_split_local[g_split_local_n++];
  }
  if (int rc = spo::hip_check(hipMemsetAsync(sl->base, 0, 2 * XR_REGION_BYTES, st), "hipMemsetAsync(split exchange)")) return rc;
  for (UpdArgs* u : {&a, &b}) {
    u->xr_region[2] = sl->base; u->xr_region[3] = sl->base + XR_REGION_BYTES;
    u->xr_census = reinterpret_cast<unsigned long long*>(sl->base + 2 * XR_REGION_BYTES);
    u->xr_census_tag = sl->tag;
  }
  sl->tag += 1u;
  return 0;
}

extern "C" int spo_critic_fit_iter_split(float* theta0, float* adam_m0, float* adam_v0, float* theta1, float* adam_m1,
                                         float* adam_v1, int64_t adam_step_host, const float* obs, const float* target_r,
                                         const float* target_c, const int32_t* perm0, const int32_t* perm1, int64_t M_half,
                                         const spo_ppo_cfg* cfg_host, float* stale_sq_io0, float* stale_sq_io1,
                                         float* losses0, float* losses1, void* sync_ws0, void* sync_ws1,
                                         void* const* regions2, uint32_t step0, void* stream) {
  if (int rc = check_cfg(cfg_host)) return rc;
  SPO_REQUIRE(theta0 && adam_m0 && adam_v0 && theta1 && adam_m1 && adam_v1 && obs && target_r && target_c && perm0 && perm1 &&
                  losses0 && losses1 && sync_ws0 && sync_ws1 && stale_sq_io0 && stale_sq_io1,
              "critic_fit_iter_split: null pointer");
  SPO_REQUIRE(theta0 != theta1 && adam_m0 != adam_m1 && adam_v0 != adam_v1 && sync_ws0 != sync_ws1 && losses0 != losses1 &&
                  stale_sq_io0 != stale_sq_io1, "critic_fit_iter_split: the two halves need separate replicas and outputs");
  SPO_REQUIRE(M_half > 0 && adam_step_host >= 0, "critic_fit_iter_split: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws0, 0, 64, st), "hipMemsetAsync(sync_ws0)")) return rc;
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws1, 0, 64, st), "hipMemsetAsync(sync_ws1)")) return rc;
  UpdArgs a{};
  if (int rc = fill_xr(a, 0, 2, regions2, step0)) return rc;
  a.theta = theta0; a.adam_m = adam_m0; a.adam_v = adam_v0;
  a.obs = obs; a.tgt_r = target_r; a.tgt_c = target_c; a.perm = perm0; a.M = M_half; a.cfg = *cfg_host;
  a.losses = losses0;
  a.slots = reinterpret_cast<unsigned long long*>(sync_ws0);
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws0) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.pow_b1_actor = a.pow_b1; a.pow_b2_actor = a.pow_b2;
  a.first_net = 0; a.n_nets = 2; a.stale_sq = 0.f; a.stale_io = stale_sq_io0;
  UpdArgs b = a;                       // rank 1: same data and configuration, its own replica / rows / outputs
  b.xr_rank = 1;
  b.theta = theta1; b.adam_m = adam_m1; b.adam_v = adam_v1; b.perm = perm1; b.losses = losses1;
  b.slots = reinterpret_cast<unsigned long long*>(sync_ws1);
  b.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws1) + 64);
  b.stale_io = stale_sq_io1;
  const int kin = pick_kin(cfg_host->obs_dim);
  // SPO_CPO_SPLIT_FORM=h (opt-in, round 5): the main + helper kernel with the exchange on the helper waves (ppo_update_h_split_kernel:
  // 15.98 us per 128-row step); default: the four-wave kernel of rounds 3-4, one exchange of the whole gradient per step (14.05 us)
  static const bool split_h = [] { const char* e = getenv("SPO_CPO_SPLIT_FORM"); return e && !strcmp(e, "h"); }();
  if (split_h && kin <= 64 && cfg_host->batch <= 64 && update_form() >= 2) {
    if (int rc = h_scratch_for(st, &a.backup, &a.slots, 0)) return rc;
    if (int rc = h_scratch_for(st, &b.backup, &b.slots, 1)) return rc;
    static const int spec_env = [] { const char* e = getenv("SPO_UPDATE_SPEC"); return e ? atoi(e) : 1; }();
    a.spec_mode = b.spec_mode = spec_env;
    a.xr_helper_rd = b.xr_helper_rd = 1;
    if (int rc = split_local_for(st, a, b)) return rc;
    if (int rc = spo::hip_check(hipMemsetAsync(a.slots, 0, H_SLOT_BYTES, st), "hipMemsetAsync(update scratch slots)")) return rc;
    if (int rc = spo::hip_check(hipMemsetAsync(b.slots, 0, H_SLOT_BYTES, st), "hipMemsetAsync(update scratch slots)")) return rc;
#define SPO_SPLIT_H(K)                                                                                            \
  {                                                                                                               \
    const size_t sh = UpdHLds<K>::SIZE * sizeof(float);                                                           \
    static bool attr_done[SPO_MAX_DEVICES] = {};                                                                  \
    const int dslot = current_device_slot();                                                                      \
    if (!attr_done[dslot]) {                                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_update_h_split_kernel<K>),            \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);                    \
      if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(update h split)");                       \
      attr_done[dslot] = true;                                                                                    \
    }                                                                                                             \
    hipLaunchKernelGGL((ppo_update_h_split_kernel<K>), dim3(8 * (2 * a.n_nets - 1) + 1), dim3(512), sh, st, a, b); \
  }
    if (kin == 16) SPO_SPLIT_H(16) else if (kin == 32) SPO_SPLIT_H(32) else SPO_SPLIT_H(64)
#undef SPO_SPLIT_H
    SPO_LAUNCH_CHECK("spo_critic_fit_iter_split");
    return 0;
  }
  if (int rc = split_local_for(st, a, b)) return rc;   // (SPO_CPO_SPLIT_L2=0: the uncached regions whatever the placement)
#define SPO_SPLIT(K)                                                                                              \
  {                                                                                                               \
    const size_t sh = UpdLds<K>::SIZE * sizeof(float);                                                            \
    static bool attr_done[SPO_MAX_DEVICES] = {};                                                                  \
    const int dslot = current_device_slot();                                                                      \
    if (!attr_done[dslot]) {                                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_update_split_kernel<K>),              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);                    \
      if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(update split)");                         \
      attr_done[dslot] = true;                                                                                    \
    }                                                                                                             \
    hipLaunchKernelGGL((ppo_update_split_kernel<K>), dim3(8 * (2 * a.n_nets - 1) + 1), dim3(256), sh, st, a, b);  \
  }
  if (kin == 16) SPO_SPLIT(16) else if (kin == 32) SPO_SPLIT(32) else if (kin == 64) SPO_SPLIT(64) else SPO_SPLIT(128)
#undef SPO_SPLIT
  SPO_LAUNCH_CHECK("spo_critic_fit_iter_split");
  return 0;
}

// The exchange self-test with both ranks in one grid (the co-residency the split form relies on): result4_dev =
// {mismatches, timeout} of rank 0 then of rank 1.
extern "C" int spo_p2p_selftest_one_grid(void* const* regions2, uint32_t step0, int iters, int32_t* result4_dev, void* stream) {
  SPO_REQUIRE(result4_dev && iters > 0, "p2p_selftest_one_grid: bad args");
  UpdArgs a{};
  if (int rc = fill_xr(a, 0, 2, regions2, step0)) return rc;
  a.grid_ranks = 2;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(result4_dev, 0, 16, st), "hipMemsetAsync(selftest)")) return rc;
  hipLaunchKernelGGL(xr_selftest_kernel, dim3(41), dim3(256), 0, st, 0, 2, step0, iters, result4_dev, a);
  SPO_LAUNCH_CHECK("spo_p2p_selftest_one_grid");
  return 0;
}

// Debug: when set (device pointer to 3*10 u64), spo_ppo_lag_update_iter runs an instrumented build of
// the kernel that accumulates shader-clock cycles per phase of the step (wave 0 lane 0 per block).
extern "C" int spo_debug_set_update_profile(void* dev_u64_30) {
  g_prof_buf = reinterpret_cast<unsigned long long*>(dev_u64_30);
  return 0;
}

extern "C" int spo_ppo_lag_grad(const float* theta, const float* obs, const float* act, const float* logp_old,
                                const float* target_r, const float* target_c, const float* adv, const int32_t* idx,
                                int n_idx, int64_t mean_count, const spo_ppo_cfg* cfg_host, float* flat_grad,
                                float* losses3, void* stream) {
  if (int rc = check_cfg(cfg_host)) return rc;
  SPO_REQUIRE(theta && obs && act && logp_old && target_r && target_c && adv && idx && flat_grad && losses3,
              "ppo_lag_grad: null pointer");
  SPO_REQUIRE(n_idx > 0 && n_idx <= cfg_host->batch && mean_count > 0, "ppo_lag_grad: bad n_idx/mean_count");
  UpdArgs a{};
  a.theta = const_cast<float*>(theta);
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = idx; a.M = n_idx; a.cfg = *cfg_host; a.losses = losses3;
  a.first_net = 0; a.n_nets = 3; a.flat_grad = flat_grad; a.mean_count = mean_count;
  if (int rc = launch_update<false>(a, 3, (hipStream_t)stream)) return rc;
  SPO_LAUNCH_CHECK("spo_ppo_lag_grad");
  return 0;
}

// Data-parallel inner loop helper: apply the (all-reduced) gradient of step k, then immediately
// enqueue the local gradient of step k+1 -- one host call per optimiser step instead of two.
extern "C" int spo_clip_adam_then_grad(float* theta, float* adam_m, float* adam_v, float* flat_grad,
                                       int64_t adam_step_host, float grad_scale, const float* obs, const float* act,
                                       const float* logp_old, const float* target_r, const float* target_c,
                                       const float* adv, const int32_t* next_idx, int next_n_idx,
                                       const spo_ppo_cfg* cfg_host, float* next_losses3, void* stream) {
  if (int rc = spo_clip_adam(theta, adam_m, adam_v, flat_grad, adam_step_host, grad_scale, cfg_host, stream)) return rc;
  if (next_idx == nullptr || next_n_idx <= 0) return 0;
  return spo_ppo_lag_grad(theta, obs, act, logp_old, target_r, target_c, adv, next_idx, next_n_idx, next_n_idx,
                          cfg_host, flat_grad, next_losses3, stream);
}

extern "C" int spo_clip_adam(float* theta, float* adam_m, float* adam_v, const float* flat_grad,
                             int64_t adam_step_host, float grad_scale, const spo_ppo_cfg* cfg_host, void* stream) {
  if (int rc = check_cfg(cfg_host)) return rc;
  SPO_REQUIRE(theta && adam_m && adam_v && flat_grad && adam_step_host >= 0, "clip_adam: bad args");
  AdamArgs a{theta, adam_m, adam_v, flat_grad, spo_param_count(cfg_host->obs_dim, cfg_host->act_dim), grad_scale,
             cfg_host->max_grad_norm, cfg_host->lr_actor, cfg_host->lr_critic, cfg_host->beta1, cfg_host->beta2,
             cfg_host->adam_eps, pow((double)cfg_host->beta1, (double)adam_step_host),
             pow((double)cfg_host->beta2, (double)adam_step_host),
             spo_param_offset(cfg_host->obs_dim, cfg_host->act_dim, 2)};
  hipLaunchKernelGGL(clip_adam_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  SPO_LAUNCH_CHECK("spo_clip_adam");
  return 0;
}
