// Fused reward + cost GAE(lambda) over the dense [num_envs, T] on-policy buffer, gfx950.
//
// Replaces VectorizedOnPolicyBuffer.finish_path -> calculate_adv_and_value_targets ->
// discount_cumsum (reference safepo/common/buffer.py:97-140,191-201,167-188): a Python loop of
// 2*L 0-dim tensor ops per path.  Here: one launch for every path of every env, reward and cost
// together; HBM-bound (33 algorithmic bytes per (env,step): 16 B in, 1 B mask, 16 B out).
//
// Mapping: env-major rows; LPR lanes x VEC elements cover one row chunk, 64/LPR rows per wave,
// 4 waves per block, >= 2 blocks per CU at N=4096,T=128 (512 blocks).  Loads are 16 B/lane
// coalesced (VEC=4).  The recurrence c_t = delta_t + (gamma*lam)*c_{t+1}, reset at seg_end, is a
// segmented backward scan of affine maps: in-lane sequential (reference order), cross-lane
// Hillis-Steele over the row's lanes with wave shuffles in fp64, then an in-lane replay with the
// incoming carry so every element's final value is produced by the reference's own
// `delta + disc*c` step.  delta is formed in fp32 with three separately rounded ops and gamma
// rounded to fp32 (buffer.py:198); fp contraction is disabled for this file.
#include <cstdlib>
#include <vector>
#include <hip/hip_ext.h>
#include "common.h"
#include "../../include/safepo_hip.h"

namespace {

template <int LPR>
__device__ __forceinline__ double shfl_down_d(double x, int off) {
  return __shfl_down(x, off, LPR);
}

// v + (v moved by the DPP control, +0.0 where the source lane is out of range / the row is masked)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_d(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  const int shi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return v + __hiloint2double(shi, slo);
}
// Wave-wide fp64 sum without LDS traffic: row_shr 1,2,4,8 (prefix sums inside each 16-lane row), then
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3.  The total ends up in lane 63.
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v = dpp_add_d<0x111, 0xf>(v);
  v = dpp_add_d<0x112, 0xf>(v);
  v = dpp_add_d<0x114, 0xf>(v);
  v = dpp_add_d<0x118, 0xf>(v);
  v = dpp_add_d<0x142, 0xa>(v);
  v = dpp_add_d<0x143, 0xc>(v);
  return v;
}

struct GaeArgs {
  const float* reward; const float* cost; const float* value_r; const float* value_c;
  const uint8_t* seg_end; const float* boot_r; const float* boot_c;
  float* adv_r; float* adv_c; float* target_r; float* target_c;
  double* partials;
  int64_t N; int64_t T;
  float gamma32; double disc_r; double disc_c;
  int ablate;     // debug timing knob: 1 skip stats reduction, 2 skip cross-lane scan, 4 skip stores, 8 no loads/stores
  int plain_stores;   // A/B knob (env SPO_GAE_PLAIN_STORES=1): ordinary stores instead of the write-through ones
  int64_t nblocks;    // logical 4-wave blocks (= partial rows / 4)
};

template <int VEC> struct VecT;
template <> struct VecT<4> { using f = float4; using u = uchar4; };
template <> struct VecT<1> { using f = float;  using u = uint8_t; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, bool ok, float (&o)[VEC]) {
  if constexpr (VEC == 4) {
    float4 v = ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
    o[0] = ok ? *p : 0.f;
  }
}
// Output stores carry agent scope (`sc1`).  Each XCD has its own L2, so a plain store parks the line there and the
// end-of-kernel release writes all 8.4 MB back in one burst AFTER the last wave has finished; a write-through store
// streams out while other waves are still loading and scanning.  Measured on the 4096 x 128 launch: 5.50 -> 4.58 us
// (32 768 envs: 32.0 -> 30.4 us; neutral once the buffer streams from HBM).  Same values, stronger visibility.
typedef float f4st __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, bool ok, const float (&o)[VEC], bool plain) {
  if (!ok) return;
  if constexpr (VEC == 4) {
    if (plain) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); return; }
    const f4st v = {o[0], o[1], o[2], o[3]};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  } else {
    if (plain) { *p = o[0]; return; }
    __hip_atomic_store(p, o[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One affine map c_out = A*c_in + B, composed right-to-left.
// BOOT: where the bootstrap value of a path end comes from.
//   BOOT_PRED   load boot_r / boot_c only at path ends (a dependent round trip seg_end -> boot, and a whole 128-byte
//               line fetched for 4 useful bytes: 1.13 x the algorithmic bytes at one path end per 64 steps);
//   BOOT_EAGER  load them unconditionally (8 extra B/element; debug / A-B knob);
//   BOOT_FOLDED no bootstrap arrays at all: `reward` / `cost` are the arrays spo_boundary_step_fold wrote, which hold
//               fl(r + fl(gamma32 * boot)) at path ends -- exactly the first two of the three fp32 operations of
//               delta = r + gamma*v_next - v (buffer.py:198), so delta = fl(folded - v) is bit-identical, the traffic is
//               the 33 algorithmic bytes per element and nothing waits for a second round trip.
// RC: the reward scan and the cost scan of a row run in DIFFERENT lane groups (group g of a wave: row g/2, quantity g%2)
// instead of both in every lane.  Same instructions in total, but each wave's dependent chain (deltas, affine
// composition, replay) is half as long and twice as many waves are in flight -- the 4096x128 launch is latency-bound.
constexpr int BOOT_PRED = 0, BOOT_EAGER = 1, BOOT_FOLDED = 2;
template <int VEC, int LPR, int BOOT, bool RC, int F = 1>
__global__ __launch_bounds__(256 * F) void gae_kernel(GaeArgs a) {
  constexpr bool EAGER_BOOT = (BOOT == BOOT_EAGER);
  static_assert(!RC || LPR <= 32, "RC needs two lane groups per wave");
  constexpr int NK = RC ? 1 : 2;
  constexpr int ROWS_PER_WAVE = RC ? 64 / LPR / 2 : 64 / LPR;
  constexpr int CHUNK = LPR * VEC;
  const int lane = threadIdx.x & 63;
  const int wave = (threadIdx.x >> 6) & 3;
  // F logical 4-wave blocks per workgroup (fewer, fatter workgroups; the numbering of rows and partial rows is unchanged)
  const int64_t lblock = (int64_t)blockIdx.x * F + (F > 1 ? (int)(threadIdx.x >> 8) : 0);
  if (F > 1 && lblock >= a.nblocks) return;
  const int grp_id = lane / LPR;
  const int sub = RC ? grp_id / 2 : grp_id;                   // row within the wave
  const int ksel = RC ? (grp_id & 1) : 0;                     // RC: 0 = reward scan, 1 = cost scan
  const int sl = lane % LPR;
  const int64_t row = (lblock * 4 + wave) * ROWS_PER_WAVE + sub;
  const bool row_ok = row < a.N && !(a.ablate & 8);           // ablate 8: no loads, no stores (launch floor)
  const int64_t T = a.T;
  const int64_t rbase = row * T;

  double carry_c[NK];                   // c at the first element of the chunk to the right
  float carry_v[NK];                    // value at that element (v_{t+1} across the chunk edge)
#pragma unroll
  for (int k = 0; k < NK; ++k) { carry_c[k] = 0.0; carry_v[k] = 0.f; }
  bool carry_any = false;               // a seg_end exists to the right of this chunk
  double s_r = 0.0, s_r2 = 0.0, s_c = 0.0;
  double disc[NK];
  const float* in_rw[NK]; const float* in_v[NK]; const float* in_boot[NK]; float* out_adv[NK]; float* out_tgt[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const bool cost_side = RC ? (ksel == 1) : (k == 1);
    disc[k] = cost_side ? a.disc_c : a.disc_r;
    in_rw[k] = cost_side ? a.cost : a.reward;
    in_v[k] = cost_side ? a.value_c : a.value_r;
    in_boot[k] = cost_side ? a.boot_c : a.boot_r;
    out_adv[k] = cost_side ? a.adv_c : a.adv_r;
    out_tgt[k] = cost_side ? a.target_c : a.target_r;
  }

  const int nchunks = (int)((T + CHUNK - 1) / CHUNK);
  for (int ch = nchunks - 1; ch >= 0; --ch) {
    const int64_t t0 = (int64_t)ch * CHUNK + (int64_t)sl * VEC;
    // VEC==4 requires T%4==0, so a lane's elements are all valid or all invalid.
    const bool ok = row_ok && (t0 < T);
    float rw[NK][VEC], vv[NK][VEC];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      load_vec<VEC>(in_rw[k] + rbase + t0, ok, rw[k]);
      load_vec<VEC>(in_v[k] + rbase + t0, ok, vv[k]);
    }
    bool seg[VEC];
    if constexpr (VEC == 4) {
      uchar4 s4 = ok ? *reinterpret_cast<const uchar4*>(a.seg_end + rbase + t0) : make_uchar4(0, 0, 0, 0);
      seg[0] = s4.x != 0; seg[1] = s4.y != 0; seg[2] = s4.z != 0; seg[3] = s4.w != 0;
    } else {
      seg[0] = ok ? (a.seg_end[rbase + t0] != 0) : false;
    }
    bool lane_seg = false;
#pragma unroll
    for (int e = 0; e < VEC; ++e) lane_seg |= seg[e];
    float bt[NK][VEC];
    if (EAGER_BOOT) {
#pragma unroll
      for (int k = 0; k < NK; ++k) load_vec<VEC>(in_boot[k] + rbase + t0, ok, bt[k]);
    }

    // which lanes of my row hold a segment end (wave-wide ballot, then my row's slice)
    const unsigned long long ball = __ballot(lane_seg);
    unsigned long long grp = (LPR == 64) ? ball : ((ball >> (grp_id * LPR)) & ((1ull << LPR) - 1ull));
    const bool any_right_lane = (sl + 1 < LPR) ? ((grp >> (sl + 1)) != 0ull) : false;

    // v_{t+1} of my last element comes from the lane to the right (or the chunk carry)
    float vnext_edge[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      float nb = __shfl_down(vv[k][0], 1, LPR);
      vnext_edge[k] = (sl == LPR - 1) ? carry_v[k] : nb;
    }

    double delta[NK][VEC];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float* boot = in_boot[k];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float vn = (e == VEC - 1) ? vnext_edge[k] : vv[k][e + 1];
        if (EAGER_BOOT) vn = seg[e] ? bt[k][e] : vn;
        else if (BOOT == BOOT_PRED && seg[e]) vn = boot[rbase + t0 + e];            // predicated: only at path ends
        // deltas = rewards[:-1] + gamma * values[1:] - values[:-1]   (fp32, buffer.py:198)
        float rg = __fadd_rn(rw[k][e], __fmul_rn(a.gamma32, vn));
        if (BOOT == BOOT_FOLDED) rg = seg[e] ? rw[k][e] : rg;   // the boundary step already added gamma * bootstrap
        float d = __fsub_rn(rg, vv[k][e]);
        delta[k][e] = (double)d;
      }
    }

    // lane composite (A,B): c_first_of_lane = A*c_in + B
    double A[NK], B[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      A[k] = 1.0; B[k] = 0.0;
      if (ok) {
#pragma unroll
        for (int e = VEC - 1; e >= 0; --e) {
          if (seg[e]) { A[k] = 0.0; B[k] = delta[k][e]; }
          else { B[k] = __dadd_rn(delta[k][e], __dmul_rn(disc[k], B[k])); A[k] = __dmul_rn(disc[k], A[k]); }
        }
      }
    }
    // inclusive suffix scan over the row's lanes
    if (!(a.ablate & 2))
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        double An = shfl_down_d<LPR>(A[k], off);
        double Bn = shfl_down_d<LPR>(B[k], off);
        if (sl + off < LPR) {
          B[k] = __dadd_rn(B[k], __dmul_rn(A[k], Bn));
          A[k] = __dmul_rn(A[k], An);
        }
      }
    }
    // c entering my lane from the right = c_first(sl+1) evaluated with the chunk carry
    double cin[NK], cfirst[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      cfirst[k] = __dadd_rn(__dmul_rn(A[k], carry_c[k]), B[k]);
      double nb = shfl_down_d<LPR>(cfirst[k], 1);
      cin[k] = (sl == LPR - 1) ? carry_c[k] : nb;
    }

    // replay in reference order and emit
    float oadv[NK][VEC], otgt[NK][VEC];
    bool fin_right = any_right_lane || carry_any;            // a path end exists right of my lane
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      double c = cin[k];
      bool fin = fin_right;
#pragma unroll
      for (int e = VEC - 1; e >= 0; --e) {
        c = seg[e] ? delta[k][e] : __dadd_rn(delta[k][e], __dmul_rn(disc[k], c));   // buffer.py:186
        fin |= seg[e];
        float adv = fin ? (float)c : 0.f;
        float tgt = fin ? (float)__dadd_rn(c, (double)vv[k][e]) : 0.f;              // buffer.py:200
        oadv[k][e] = adv; otgt[k][e] = tgt;
      }
    }
    const bool st_ok = ok && !(a.ablate & 4);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      store_vec<VEC>(out_adv[k] + rbase + t0, st_ok, oadv[k], a.plain_stores != 0);
      store_vec<VEC>(out_tgt[k] + rbase + t0, st_ok, otgt[k], a.plain_stores != 0);
    }
    if (ok) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if (RC) {
          const double x = (double)oadv[0][e];
          if (ksel == 0) { s_r += x; s_r2 += x * x; } else { s_c += x; }
        } else {
          const double x = (double)oadv[0][e];
          s_r += x; s_r2 += x * x; s_c += (double)oadv[NK - 1][e];
        }
      }
    }
    // carries for the chunk to the left: values at the first lane of my row group
    if (ch > 0) {                              // uniform: single-chunk rows (T <= LPR*VEC) never need the carries
      const int src = grp_id * LPR;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        carry_c[k] = __shfl(cfirst[k], src);
        carry_v[k] = __shfl(vv[k][0], src);
      }
      carry_any = carry_any || (grp != 0ull);
    }
  }

  // per-WAVE partial sums in a fixed order -> partials[block][wave][4] = {sum adv_r, sum adv_r^2, sum adv_c, count}.
  // No LDS stage and no barrier: the tail of the launch is a handful of DPP adds and one 32-byte row per wave
  // (the block-level combine cost 0.6 us of the 4.5 us launch at 4096 x 128; spo_adv_reduce adds the rows in a fixed order).
  if (a.ablate & 1) return;
  double* const prow = a.partials + (lblock * 4 + wave) * 4;
  int64_t wrows = a.N - (lblock * 4 + wave) * ROWS_PER_WAVE;
  wrows = wrows < 0 ? 0 : (wrows > ROWS_PER_WAVE ? ROWS_PER_WAVE : wrows);
  if constexpr (RC) {
    static_assert(!RC || LPR == 32, "RC layout: lanes 0-31 scan the reward, lanes 32-63 the cost of the wave's row");
    // after row_shr 1,2,4,8 and row_bcast:15 lane 31 holds the sum of lanes 0-31 and lane 63 that of lanes 32-63
    double x = (ksel == 0) ? s_r : s_c, y = s_r2;
    x = dpp_add_d<0x111, 0xf>(x); y = dpp_add_d<0x111, 0xf>(y);
    x = dpp_add_d<0x112, 0xf>(x); y = dpp_add_d<0x112, 0xf>(y);
    x = dpp_add_d<0x114, 0xf>(x); y = dpp_add_d<0x114, 0xf>(y);
    x = dpp_add_d<0x118, 0xf>(x); y = dpp_add_d<0x118, 0xf>(y);
    x = dpp_add_d<0x142, 0xa>(x); y = dpp_add_d<0x142, 0xa>(y);
    if (lane == 31) { prow[0] = x; prow[1] = y; }
    if (lane == 63) { prow[2] = x; prow[3] = (double)(wrows * T); }
  } else {
    s_r = wave_sum_to_lane63(s_r);
    s_r2 = wave_sum_to_lane63(s_r2);
    s_c = wave_sum_to_lane63(s_c);
    if (lane == 63) { prow[0] = s_r; prow[1] = s_r2; prow[2] = s_c; prow[3] = (double)(wrows * T); }
  }
}

__global__ __launch_bounds__(256) void adv_reduce_kernel(const double* partials, int nb, double* sums) {
  // fixed-order tree over the nb * 4 per-wave rows: thread i sums rows i, i+256, ...; then a shared-memory tree.
  __shared__ double sh[4][256];
  double acc[4] = {0, 0, 0, 0};
  const int nrows = nb * (SPO_GAE_PARTIAL_STRIDE / 4);
  for (int b = threadIdx.x; b < nrows; b += 256)
    for (int k = 0; k < 4; ++k) acc[k] += partials[(int64_t)b * 4 + k];
  for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 4) sums[threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void adv_apply_kernel(float* adv_r, float* adv_c, float* adv_mix, const double* sums,
                                                       int64_t count, float lam32, float lam_p1_32, int std_r,
                                                       int std_c, float* stats_out) {
  // buffer.py:154-160: mean, UNBIASED std (+1e-8), adv_c only centred.  Statistics in fp64 from
  // exact sums of the fp32 values, rounded once to fp32 (torch reduces in fp32: <= 1e-6 rel apart).
  const double n = sums[3];
  const double mean_r = sums[0] / n;
  double var = (sums[1] - sums[0] * sums[0] / n) / (n - 1.0);
  if (var < 0) var = 0;
  const float mr = (float)mean_r, sd = (float)sqrt(var), mc = (float)(sums[2] / n);
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats_out) { stats_out[0] = mr; stats_out[1] = sd; stats_out[2] = mc; }
  const float den = __fadd_rn(sd, 1e-8f);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += stride) {
    float r[4], c[4], m[4];
    const int nv = (count - i) >= 4 ? 4 : (int)(count - i);
    if (nv == 4) {
      float4 a = *reinterpret_cast<float4*>(adv_r + i), b = *reinterpret_cast<float4*>(adv_c + i);
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; c[0] = b.x; c[1] = b.y; c[2] = b.z; c[3] = b.w;
    } else {
      for (int e = 0; e < 4; ++e) { r[e] = e < nv ? adv_r[i + e] : 0.f; c[e] = e < nv ? adv_c[i + e] : 0.f; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (std_r) r[e] = __fdiv_rn(__fsub_rn(r[e], mr), den);
      if (std_c) c[e] = __fsub_rn(c[e], mc);
      // ppo_lag.py:280-281: advantage = adv_r - lambda*adv_c ; advantage /= (lambda + 1)
      m[e] = __fdiv_rn(__fsub_rn(r[e], __fmul_rn(lam32, c[e])), lam_p1_32);
    }
    if (nv == 4) {
      *reinterpret_cast<float4*>(adv_r + i) = make_float4(r[0], r[1], r[2], r[3]);
      *reinterpret_cast<float4*>(adv_c + i) = make_float4(c[0], c[1], c[2], c[3]);
      if (adv_mix) *reinterpret_cast<float4*>(adv_mix + i) = make_float4(m[0], m[1], m[2], m[3]);
    } else {
      for (int e = 0; e < nv; ++e) { adv_r[i + e] = r[e]; adv_c[i + e] = c[e]; if (adv_mix) adv_mix[i + e] = m[e]; }
    }
  }
}

int g_gae_force_variant = 0;    // 0 auto, 1 eager bootstrap loads, 2 predicated (debug/bench knob)
struct GaeGeom { int vec; int lpr; int rows_per_block; bool rc; };
inline bool gae_rc_enabled() {
  static const int v = [] { const char* e = getenv("SPO_GAE_RC_SPLIT"); return e ? atoi(e) : 1; }();
  return v != 0;
}
inline GaeGeom gae_geom(int64_t T, int64_t N) {
  GaeGeom g;
  g.vec = (T % 4 == 0) ? 4 : 1;
  int64_t need = (T + g.vec - 1) / g.vec;
  int lpr = 1;
  while (lpr < need && lpr < 64) lpr <<= 1;
  g.lpr = lpr;
  // 128-step rows while the buffer is cache-resident and the launch latency-bound (measured on MI355X at 4096 x 128:
  // 5.0-5.25 us against 5.2-5.65 us per launch); buffers that stream from HBM keep both scans in every lane
  // (262 144 x 128: 235 us against 270 us).
  g.rc = gae_rc_enabled() && g.vec == 4 && lpr == 32 && N * T <= ((int64_t)4 << 20);
  g.rows_per_block = g.rc ? 4 : 4 * (64 / lpr);
  return g;
}

// How a dispatch is issued is a policy of the CALLER: the product entry point spo_gae_fused uses PlainLaunch; the measurement entry
// point spo_gae_fused_timed (bench.py's roofline figure) uses TimedLaunch at the end of this file.  The launch path itself
// carries no event arguments and no branch on them (VERDICT r04 housekeeping).
struct PlainLaunch {
  hipStream_t st;
  template <class K>
  void operator()(K kernel, dim3 grid, dim3 block, const GaeArgs& a) const { hipLaunchKernelGGL(kernel, grid, block, 0, st, a); }
};

template <int VEC, int BOOT, class Launch>
int launch_gae(const GaeGeom& g, const GaeArgs& a, int blocks, const Launch& launch) {
  if (g.rc) {
    if constexpr (VEC == 4) {
      static const int fat = [] { const char* e = getenv("SPO_GAE_FAT"); return e ? atoi(e) : 1; }();
      if (fat == 2 || fat == 4) {
        const int pb = (blocks + fat - 1) / fat;
        if (fat == 2) launch(gae_kernel<4, 32, BOOT, true, 2>, dim3(pb), dim3(512), a);
        else launch(gae_kernel<4, 32, BOOT, true, 4>, dim3(pb), dim3(1024), a);
        return 0;
      }
      launch(gae_kernel<4, 32, BOOT, true>, dim3(blocks), dim3(256), a);
      return 0;
    }
  }
  switch (g.lpr) {
#define SPO_CASE(L) case L: launch(gae_kernel<VEC, L, BOOT, false>, dim3(blocks), dim3(256), a); break;
    SPO_CASE(1) SPO_CASE(2) SPO_CASE(4) SPO_CASE(8) SPO_CASE(16) SPO_CASE(32) SPO_CASE(64)
#undef SPO_CASE
    default: return spo::fail(-1, "gae: bad lanes-per-row %d", g.lpr);
  }
  return 0;
}

}  // namespace

extern "C" int spo_gae_num_blocks(int64_t num_envs, int64_t T) {
  if (num_envs <= 0 || T <= 0) return 0;
  GaeGeom g = gae_geom(T, num_envs);
  return (int)((num_envs + g.rows_per_block - 1) / g.rows_per_block);
}

static int gae_plain_stores() {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("SPO_GAE_PLAIN_STORES"); mode = (e && e[0] == '1') ? 1 : 0; }
  return mode;
}

template <class Launch>
static int gae_fused_impl(const float* reward, const float* cost, const float* value_r, const float* value_c,
                          const uint8_t* seg_end, const float* boot_r, const float* boot_c, float* adv_r,
                          float* adv_c, float* target_r, float* target_c, double* partials, int64_t num_envs,
                          int64_t T, double gamma, double lam, double lam_c, const Launch& launch) {
  SPO_REQUIRE(num_envs >= 0 && T >= 0, "gae: negative size");
  if (num_envs == 0 || T == 0) return 0;
  SPO_REQUIRE(reward && cost && value_r && value_c && seg_end && adv_r && adv_c && target_r && target_c && partials,
              "gae: null pointer");
  SPO_REQUIRE((boot_r == nullptr) == (boot_c == nullptr), "gae: boot_r and boot_c must both be given or both be NULL (folded)");
  const bool folded = boot_r == nullptr;
  GaeGeom g = gae_geom(T, num_envs);
  GaeArgs a{reward, cost, value_r, value_c, seg_end, boot_r, boot_c, adv_r, adv_c, target_r, target_c, partials,
            num_envs, T, (float)gamma, gamma * lam, gamma * lam_c, g_gae_force_variant >> 4, gae_plain_stores(), 0};
  const int blocks = spo_gae_num_blocks(num_envs, T);
  a.nblocks = blocks;
  // predicated bootstrap loads by default when bootstrap arrays are given (fewest bytes; measured equal or faster than
  // the eager form at every size); no bootstrap loads at all in the folded form
  const int fv = g_gae_force_variant & 15;
  const bool eager = fv == 1 && !folded;
  int rc;
#define SPO_GAE_GO(V)                                                                     \
  rc = folded ? launch_gae<V, BOOT_FOLDED>(g, a, blocks, launch)                          \
              : eager ? launch_gae<V, BOOT_EAGER>(g, a, blocks, launch)                   \
                      : launch_gae<V, BOOT_PRED>(g, a, blocks, launch);
  if (g.vec == 4) { SPO_GAE_GO(4) } else { SPO_GAE_GO(1) }
#undef SPO_GAE_GO
  if (rc) return rc;
  SPO_LAUNCH_CHECK("spo_gae_fused");
  return 0;
}

extern "C" int spo_gae_fused(const float* reward, const float* cost, const float* value_r, const float* value_c,
                             const uint8_t* seg_end, const float* boot_r, const float* boot_c, float* adv_r,
                             float* adv_c, float* target_r, float* target_c, double* partials, int64_t num_envs,
                             int64_t T, double gamma, double lam, double lam_c, void* stream) {
  return gae_fused_impl(reward, cost, value_r, value_c, seg_end, boot_r, boot_c, adv_r, adv_c, target_r, target_c, partials,
                        num_envs, T, gamma, lam, lam_c, PlainLaunch{(hipStream_t)stream});
}

extern "C" int spo_debug_gae_variant(int v) { g_gae_force_variant = v; return 0; }

// ---- measurement only (bench.py's roofline figure): the scan `reps` times, every dispatch carrying its own start / stop events
// (hipExtLaunchKernelGGL: the timestamps of the dispatch packet itself -- what rocprofv3 --kernel-trace reports);
// durations_us_host[i] = execution time of dispatch i.  Synchronises the stream.
namespace {
struct TimedLaunch {
  hipStream_t st;
  hipEvent_t ev_start, ev_stop;
  template <class K>
  void operator()(K kernel, dim3 grid, dim3 block, const GaeArgs& a) const {
    hipExtLaunchKernelGGL(kernel, grid, block, 0, st, ev_start, ev_stop, 0, a);
  }
};
}  // namespace
extern "C" int spo_gae_fused_timed(const float* reward, const float* cost, const float* value_r, const float* value_c,
                                   const uint8_t* seg_end, const float* boot_r, const float* boot_c, float* adv_r,
                                   float* adv_c, float* target_r, float* target_c, double* partials, int64_t num_envs,
                                   int64_t T, double gamma, double lam, double lam_c, int reps, float* durations_us_host,
                                   void* stream) {
  SPO_REQUIRE(reps > 0 && reps <= 4096 && durations_us_host, "gae_timed: bad reps / output");
  hipStream_t st = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(2 * (size_t)reps, nullptr);
  int rc = 0;
  for (auto& e : ev)
    if (!rc) rc = spo::hip_check(hipEventCreate(&e), "hipEventCreate");
  for (int i = 0; i < reps && !rc; ++i)
    rc = gae_fused_impl(reward, cost, value_r, value_c, seg_end, boot_r, boot_c, adv_r, adv_c, target_r, target_c, partials,
                        num_envs, T, gamma, lam, lam_c, TimedLaunch{st, ev[2 * i], ev[2 * i + 1]});
  if (!rc) rc = spo::hip_check(hipStreamSynchronize(st), "hipStreamSynchronize(gae_timed)");
  for (int i = 0; i < reps && !rc; ++i) {
    float ms = 0.f;
    rc = spo::hip_check(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]), "hipEventElapsedTime");
    durations_us_host[i] = ms * 1000.f;
  }
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  return rc;
}

extern "C" int spo_adv_reduce(const double* partials, int num_blocks, double* sums, void* stream) {
  SPO_REQUIRE(partials && sums && num_blocks > 0, "adv_reduce: bad args");
  hipLaunchKernelGGL(adv_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, num_blocks, sums);
  SPO_LAUNCH_CHECK("spo_adv_reduce");
  return 0;
}

extern "C" int spo_adv_apply(float* adv_r, float* adv_c, float* adv_mix, const double* sums, int64_t count,
                             double lagrangian_multiplier, int standardize_r, int standardize_c, float* stats_out,
                             void* stream) {
  SPO_REQUIRE(adv_r && adv_c && sums && count > 0, "adv_apply: bad args");
  int64_t blocks = (count / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const float lam32 = (float)lagrangian_multiplier;
  const float lam_p1 = (float)(lagrangian_multiplier + 1.0);
  hipLaunchKernelGGL(adv_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, adv_r, adv_c,
                     adv_mix, sums, count, lam32, lam_p1, standardize_r, standardize_c, stats_out);
  SPO_LAUNCH_CHECK("spo_adv_apply");
  return 0;
}
