// MFMA building blocks for SafePO's tanh MLPs (hidden 64x64) on gfx950.
//
// The reference's actor / critics are `Linear-Tanh-Linear-Tanh-Linear` with hidden [64,64]
// (safepo/common/model.py:30-48, ppo_lag.py:45-52).  A minibatch is 64 rows, so each layer is a
// 64x64x64 fp32 GEMM: far too small for a tiled-GEMM launch, but a single CU running a chain
// of 327 680 dependent optimiser steps per epoch IS bound by its fp32 matrix rate.  We therefore
// use the exact-fp32 MFMA `v_mfma_f32_16x16x4_f32` (bitwise an fmaf chain; no TF32 on gfx950)
// with a "transposed chaining" layout that needs NO data movement between layers:
//
//   C[row = output feature][col = batch]  = sum_k W[feature][k] * X[batch][k]
//
//   A operand (weights, from LDS):   lane l supplies W[16*mt + (l&15)][k(s, l>>4)]
//   B operand (activations, regs):   lane l supplies X[batch (l&15)][k(s, l>>4)]
//   C/D (accumulator):               lane l holds rows 4*(l>>4)+reg, col (l&15)
//
// With the reduction index enumerated as k = 16*nt + 4*(l>>4) + reg, the C registers of layer
// n ARE the B operand registers of layer n+1 (tile nt, step reg), and the matching A operand is
// one 16-byte LDS read `W[row][16*nt + 4*(l>>4) .. +3]`.  fp32 sums are re-associated relative
// to a BLAS dot product (rounding-level differences only; parity tolerance is 1e-5 rel).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace spo {

using f4 = __attribute__((ext_vector_type(4))) float;

constexpr int HID = 64;        // hidden width
constexpr int LDH = HID + 4;   // LDS row stride (floats) of 64-wide weight rows (16 B aligned, bank-rotated)
constexpr int OUTP = 16;       // output layer padded to one 16-row tile

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Geometry of one network inside the flat parameter vector (nn.Linear row-major [out,in]).
struct NetGeom {
  int D;        // obs_dim
  int OUT;      // 1 (critic) or act_dim (actor)
  int off;      // offset of W1 in theta (floats); actor: log_std sits at off - OUT
  __host__ __device__ int w1() const { return off; }
  __host__ __device__ int b1() const { return off + HID * D; }
  __host__ __device__ int w2() const { return b1() + HID; }
  __host__ __device__ int b2() const { return w2() + HID * HID; }
  __host__ __device__ int w3() const { return b2() + HID; }
  __host__ __device__ int b3() const { return w3() + OUT * HID; }
  __host__ __device__ int end() const { return b3() + OUT; }
};

__host__ __device__ inline int critic_size(int D) { return HID * D + HID + HID * HID + HID + HID + 1; }
__host__ __device__ inline int actor_size(int D, int A) { return A + HID * D + HID + HID * HID + HID + A * HID + A; }
__host__ __device__ inline NetGeom net_geom(int D, int A, int net) {
  NetGeom g;
  g.D = D;
  if (net == 0) { g.OUT = 1; g.off = 0; }
  else if (net == 1) { g.OUT = 1; g.off = critic_size(D); }
  else { g.OUT = A; g.off = 2 * critic_size(D) + A; }
  return g;
}

// LDS image of one network (floats).  KIN = obs_dim padded to a multiple of 16.
// PAD = floats added to every weight row.  4 (LDH): rows 16-byte aligned, and the COLUMN-wise ds_read_b32 of the backward passes
// conflict-free.  8: the ROW-wise ds_read_b128 operand reads of the forward are conflict-free under gfx950's b128 lane groups
// ({0-3, 12-15, 20-27}, ...: with stride K + 4 the 16-byte bank quad of lane (j, q) is j + q mod 16 and lanes (12, 0), (11, 1) of
// one group collide; with K + 8 it is 2j + q mod 16: sixteen different quads per group) -- for the forward-only kernels.
template <int KIN, int PAD = 4>
struct NetLds {
  static constexpr int LD1 = KIN + PAD;
  static constexpr int LDW = HID + PAD;        // == LDH for PAD = 4
  static constexpr int W1 = 0;
  static constexpr int B1 = W1 + HID * LD1;
  static constexpr int W2 = B1 + HID;
  static constexpr int B2 = W2 + HID * LDW;
  static constexpr int W3 = B2 + HID;
  static constexpr int B3 = W3 + OUTP * LDW;
  static constexpr int SIZE = B3 + OUTP;       // multiple of 4 floats
};

// Cooperative copy of one network from the flat vector into its padded LDS image (pads zeroed).
// Round 3: no integer division per element (the old form computed i / D and i % D for each of ~8 500 floats: ~40 VALU
// instructions per element in front of every load, several microseconds per staged network in kernels that stage one per
// launch): KIN / 4 lanes cover a row, each lane four consecutive columns, rows advance by nthr / (KIN / 4).
template <int KIN, int PAD = 4>
__device__ inline void stage_net(const float* __restrict__ theta, const NetGeom g, float* lds, int tid, int nthr) {
  using L = NetLds<KIN, PAD>;
  constexpr int LDH = L::LDW;                 // (shadows the global stride inside this function)
  for (int i = tid; i < L::SIZE / 4; i += nthr) reinterpret_cast<f4*>(lds)[i] = f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int D = g.D;
  {
    constexpr int TPR = KIN / 4;                       // lanes per row of W1 (a power of two: KIN is 16, 32, 64 or 128)
    const int c0 = (tid % TPR) * 4, rstep = nthr / TPR;
    for (int r = tid / TPR; r < HID; r += rstep) {
      const float* src = theta + g.w1() + r * D + c0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e < D) lds[L::W1 + r * L::LD1 + c0 + e] = src[e];
    }
  }
  {
    constexpr int TPR = HID / 4;
    const int c0 = (tid % TPR) * 4, rstep = nthr / TPR;
    for (int r = tid / TPR; r < HID; r += rstep) {
      const float* src = theta + g.w2() + r * HID + c0;
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[L::W2 + r * LDH + c0 + e] = src[e];
    }
    for (int r = tid / TPR; r < g.OUT; r += rstep) {
      const float* src = theta + g.w3() + r * HID + c0;
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[L::W3 + r * LDH + c0 + e] = src[e];
    }
  }
  for (int i = tid; i < HID; i += nthr) { lds[L::B1 + i] = theta[g.b1() + i]; lds[L::B2 + i] = theta[g.b2() + i]; }
  for (int i = tid; i < g.OUT; i += nthr) lds[L::B3 + i] = theta[g.b3() + i];
}

// stage_net with every global load of a matrix in flight before the first LDS store (round 4).  stage_net's row loops carry
// predicated loads and stores, which the compiler turns into one dependent L2 round trip per loop iteration -- 4 + 4 + 1 of
// them with 256 threads, twice that with 128: microseconds, in kernels (policy step, bootstrap values) that do nothing else
// of that length.  Here the addresses are clamped instead of predicated, NTHR is a compile-time constant so the loops unroll,
// and W1 (32 registers at KIN = 64, NTHR = 128), then W2 + W3 + biases go out as two batches.  The image is the same except
// for the pad columns, which no operand read touches (rows 16mt + j, columns 16nt + 4q .. + 3 < K) and are left unwritten.
template <int KIN, int NTHR>
__device__ inline void stage_net_batched(const float* __restrict__ theta, const NetGeom g, float* lds, int tid) {
  using L = NetLds<KIN>;
  const int D = g.D;
  {
    constexpr int TPR = KIN / 4, RSTEP = NTHR / TPR, IT = HID / RSTEP;
    static_assert(NTHR % TPR == 0 && HID % RSTEP == 0, "stage_net_batched: thread count must tile the rows");
    const int c0 = (tid % TPR) * 4, r0 = tid / TPR;
    float w[IT][4];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const float* src = theta + g.w1() + (r0 + k * RSTEP) * D;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[k][e] = src[c0 + e < D ? c0 + e : 0];
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      f4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = c0 + e < D ? w[k][e] : 0.f;
      *reinterpret_cast<f4*>(lds + L::W1 + (r0 + k * RSTEP) * L::LD1 + c0) = v;
    }
  }
  {
    constexpr int TPR = HID / 4, RSTEP = NTHR / TPR, IT = HID / RSTEP, IT3 = (OUTP + RSTEP - 1) / RSTEP;
    static_assert(NTHR % TPR == 0 && HID % RSTEP == 0, "stage_net_batched: thread count must tile the rows");
    const int c0 = (tid % TPR) * 4, r0 = tid / TPR;
    float w[IT][4], w3[IT3][4];
#pragma unroll
    for (int k = 0; k < IT; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) w[k][e] = theta[g.w2() + (r0 + k * RSTEP) * HID + c0 + e];
#pragma unroll
    for (int k = 0; k < IT3; ++k) {
      const int r = r0 + k * RSTEP;
#pragma unroll
      for (int e = 0; e < 4; ++e) w3[k][e] = theta[g.w3() + (r < g.OUT ? r : 0) * HID + c0 + e];
    }
    const float b1 = theta[g.b1() + (tid < HID ? tid : 0)], b2 = theta[g.b2() + (tid < HID ? tid : 0)];
    const float b3 = theta[g.b3() + (tid < g.OUT ? tid : 0)];
#pragma unroll
    for (int k = 0; k < IT; ++k)
      *reinterpret_cast<f4*>(lds + L::W2 + (r0 + k * RSTEP) * L::LDW + c0) = f4{w[k][0], w[k][1], w[k][2], w[k][3]};
#pragma unroll
    for (int k = 0; k < IT3; ++k) {
      const int r = r0 + k * RSTEP;
      if (r < OUTP) {
        const bool ok = r < g.OUT;
        *reinterpret_cast<f4*>(lds + L::W3 + r * L::LDW + c0) =
            f4{ok ? w3[k][0] : 0.f, ok ? w3[k][1] : 0.f, ok ? w3[k][2] : 0.f, ok ? w3[k][3] : 0.f};
      }
    }
    if (tid < HID) { lds[L::B1 + tid] = b1; lds[L::B2 + tid] = b2; }
    if (tid < OUTP) lds[L::B3 + tid] = tid < g.OUT ? b3 : 0.f;
  }
}

// B-operand tiles of one observation row: x[nt][reg] = obs[16*nt + 4*q + reg] (0 beyond D).
// Loads are UNCONDITIONAL (clamped address, select afterwards): a predicated load compiles to a
// branch plus `s_waitcnt vmcnt(0)` before the next one, which serialises the HBM latencies.
template <int KIN>
__device__ __forceinline__ void load_obs_tiles(const float* __restrict__ row, int D, int q, f4 (&x)[KIN / 16]) {
  if ((D & 3) == 0) {
    f4 v[KIN / 16];
#pragma unroll
    for (int nt = 0; nt < KIN / 16; ++nt) {
      const int c = 16 * nt + 4 * q;
      v[nt] = *reinterpret_cast<const f4*>(row + (c < D ? c : 0));
    }
#pragma unroll
    for (int nt = 0; nt < KIN / 16; ++nt) {
      const bool ok = (16 * nt + 4 * q) < D;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[nt][e] = ok ? v[nt][e] : 0.f;
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < KIN / 16; ++nt) {
      f4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 16 * nt + 4 * q + e;
        const float t = row[c < D ? c : 0];
        v[e] = c < D ? t : 0.f;
      }
      x[nt] = v;
    }
  }
}

// Prefetch form of load_obs_tiles: the loads only (clamped addresses, NO select).  A select right behind a load makes
// the wave wait for that load on the spot -- which turns a prefetch issued a whole minibatch step ahead into an exposed
// HBM round trip.  The consumer calls mask_obs_tiles when it picks the tiles up, one step later.
template <int KIN>
__device__ __forceinline__ void load_obs_tiles_raw(const float* __restrict__ row, int D, int q, f4 (&x)[KIN / 16]) {
  if ((D & 3) == 0) {
#pragma unroll
    for (int nt = 0; nt < KIN / 16; ++nt) {
      const int c = 16 * nt + 4 * q;
      x[nt] = *reinterpret_cast<const f4*>(row + (c < D ? c : 0));
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < KIN / 16; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 16 * nt + 4 * q + e;
        x[nt][e] = row[c < D ? c : 0];
      }
  }
}
template <int KIN>
__device__ __forceinline__ void mask_obs_tiles(int D, int q, f4 (&x)[KIN / 16]) {
#pragma unroll
  for (int nt = 0; nt < KIN / 16; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) x[nt][e] = (16 * nt + 4 * q + e) < D ? x[nt][e] : 0.f;
}

// tanh with RELATIVE accuracy everywhere (<= ~4 ulp, typically 1-2), in 13 VALU ops (2 of them v_exp_f32 / v_rcp_f32):
//   |x| >= 1/8 : (1 - e) / (1 + e),  e = exp(-2|x|)   (1 - e is an exact subtraction for e in [1/2, 1]; no cancellation left)
//   |x| <  1/8 : |x| (1 - x^2/3 + 2 x^4/15)             (next term 17 x^6/315 < 2e-7 relative at 1/8)
// and the sign copied back.  Round 1 used 1 - 2/(exp(2x)+1), whose ABSOLUTE error of ~1.2e-7 is a 100 % relative error
// for |x| < 1e-7: harmless for a healthy unit, but a hidden unit whose weights the critics' L2 term has driven to ~1e-12
// then sees h = 0 instead of h = x, and Adam -- which rescales arbitrarily small gradients to O(lr) steps -- revives such
// a unit along a different path than the reference does (found by the drift-envelope test at 8 192 steps:
// tests/test_gpu_parity.py::test_full_size_update_parity_drift_envelope; profiles/r02/drift_before_tanh_fix.txt).
// ocml's tanhf costs ~40 VALU ops per value and there are 32 values per lane per minibatch step.
__device__ __forceinline__ float fast_tanh(float x) {
  const float a = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(a * -2.885390081777927f);     // exp(-2|x|) = 2^(-2|x| log2(e)); -> 0 for large |x|
  const float big = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  const float x2 = x * x;
  const float small = a * fmaf(x2, fmaf(x2, 0.13333333333f, -0.33333333333f), 1.f);
  return copysignf(a < 0.125f ? small : big, x);
}

// Two values at a time on the packed-FP32 VALU (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two IEEE operations per
// instruction at the scalar instruction's issue cost).  Element for element the same operations as fast_tanh -- the sign is
// carried by x * poly instead of copysign(|x| * poly, x), which is the same product -- so the results are bit-identical;
// 13 VALU + 4 transcendentals per PAIR instead of 22 + 4.  (On this part a wave's VALU work does not overlap anybody's MFMAs
// on the same SIMD -- tools/mfma_valu_overlap.hip -- so every instruction saved is SIMD time saved.)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fast_tanh2(const f2 x) {
  const f2 t = x * -2.885390081777927f;
  f2 e;
  e.x = __builtin_amdgcn_exp2f(-__builtin_fabsf(t.x));               // exp(-2|x|); the |.| and the sign are source modifiers
  e.y = __builtin_amdgcn_exp2f(-__builtin_fabsf(t.y));
  const f2 num = 1.f - e, den = 1.f + e;
  f2 rc;
  rc.x = __builtin_amdgcn_rcpf(den.x);
  rc.y = __builtin_amdgcn_rcpf(den.y);
  const f2 big = num * rc;
  const f2 x2 = x * x;
  const f2 c1 = {0.13333333333f, 0.13333333333f}, c2 = {-0.33333333333f, -0.33333333333f}, one = {1.f, 1.f};
  const f2 poly = __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, c1, c2), one);
  const f2 small = x * poly;
  f2 r;
  r.x = __builtin_fabsf(x.x) < 0.125f ? small.x : copysignf(big.x, x.x);
  r.y = __builtin_fabsf(x.y) < 0.125f ? small.y : copysignf(big.y, x.y);
  return r;
}
#ifndef SPO_TANH_PACKED
#define SPO_TANH_PACKED 1      // 0: four scalar fast_tanh (A/B knob; results must be bit-identical either way)
#endif
__device__ __forceinline__ f4 fast_tanh4(const f4 v) {
#if SPO_TANH_PACKED
  const f2 lo = fast_tanh2(f2{v[0], v[1]}), hi = fast_tanh2(f2{v[2], v[3]});
  return f4{lo.x, lo.y, hi.x, hi.y};
#else
  return f4{fast_tanh(v[0]), fast_tanh(v[1]), fast_tanh(v[2]), fast_tanh(v[3])};
#endif
}

// Hidden layer: out[mt] (rows 16mt+4q+reg, col batch) = tanh?(W in + b).  The four output tiles
// are independent accumulators and are issued round-robin so back-to-back MFMAs never wait on the
// 40-cycle dependent-accumulator latency of v_mfma_f32_16x16x4_f32.
// Single-buffered form for throughput kernels that run two or more waves per SIMD (the partner wave covers the LDS latency;
// 16 registers fewer than the double-buffered layer_hidden below): A tiles of one k-group at a time.
template <int NT_IN, bool TANH>
__device__ __forceinline__ void layer_hidden_lean(const float* Wl, int ld, const float* bl, const f4 (&in)[NT_IN],
                                                  f4 (&out)[HID / 16], int j, int q) {
  f4 acc[HID / 16];
#pragma unroll
  for (int mt = 0; mt < HID / 16; ++mt) acc[mt] = *reinterpret_cast<const f4*>(bl + 16 * mt + 4 * q);
#pragma unroll
  for (int nt = 0; nt < NT_IN; ++nt) {
    f4 a[HID / 16];
#pragma unroll
    for (int mt = 0; mt < HID / 16; ++mt) a[mt] = *reinterpret_cast<const f4*>(Wl + (16 * mt + j) * ld + 16 * nt + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mt = 0; mt < HID / 16; ++mt) acc[mt] = mfma4(a[mt][r], in[nt][r], acc[mt]);
    __builtin_amdgcn_sched_barrier(0);      // keep the next groups' operand reads out of this group's register budget
  }
#pragma unroll
  for (int mt = 0; mt < HID / 16; ++mt) {
    if (TANH) {
      acc[mt] = fast_tanh4(acc[mt]);
      __builtin_amdgcn_sched_barrier(0);    // four tanh chains in flight at a time, not sixteen
    }
    out[mt] = acc[mt];
  }
}

template <int NT_IN, bool TANH>
__device__ __forceinline__ void layer_hidden(const float* Wl, int ld, const float* bl, const f4 (&in)[NT_IN],
                                             f4 (&out)[HID / 16], int j, int q) {
  f4 acc[HID / 16];
  f4 a[2][HID / 16];                       // A tiles of group nt and nt+1: the LDS reads of the next group are
#pragma unroll                             // in flight while the 16 MFMAs of the current group occupy the pipe
  for (int mt = 0; mt < HID / 16; ++mt) a[0][mt] = *reinterpret_cast<const f4*>(Wl + (16 * mt + j) * ld + 4 * q);
#pragma unroll
  for (int mt = 0; mt < HID / 16; ++mt) acc[mt] = *reinterpret_cast<const f4*>(bl + 16 * mt + 4 * q);
#pragma unroll
  for (int nt = 0; nt < NT_IN; ++nt) {
    if (nt + 1 < NT_IN) {
#pragma unroll
      for (int mt = 0; mt < HID / 16; ++mt)
        a[(nt + 1) & 1][mt] = *reinterpret_cast<const f4*>(Wl + (16 * mt + j) * ld + 16 * (nt + 1) + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mt = 0; mt < HID / 16; ++mt) acc[mt] = mfma4(a[nt & 1][mt][r], in[nt][r], acc[mt]);
  }
#pragma unroll
  for (int mt = 0; mt < HID / 16; ++mt) {
    if (TANH) acc[mt] = fast_tanh4(acc[mt]);
    out[mt] = acc[mt];
  }
}

// acc[mt] += W[16mt.., k] * in  (no bias, no activation): building block for tangent products.
template <int NT_IN>
__device__ __forceinline__ void layer_accum(const float* Wl, int ld, const f4 (&in)[NT_IN], f4 (&acc)[HID / 16],
                                            int j, int q) {
#pragma unroll
  for (int nt = 0; nt < NT_IN; ++nt) {
    f4 a[HID / 16];
#pragma unroll
    for (int mt = 0; mt < HID / 16; ++mt)
      a[mt] = *reinterpret_cast<const f4*>(Wl + (16 * mt + j) * ld + 16 * nt + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mt = 0; mt < HID / 16; ++mt) acc[mt] = mfma4(a[mt][r], in[nt][r], acc[mt]);
  }
}
// acc += W3[0..15, k] * in   (output tile, no bias)
__device__ __forceinline__ f4 out_accum(const float* Wl, const f4 (&in)[HID / 16], f4 acc, int j, int q) {
#pragma unroll
  for (int nt = 0; nt < HID / 16; ++nt) {
    const f4 a = *reinterpret_cast<const f4*>(Wl + j * LDH + 16 * nt + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = mfma4(a[r], in[nt][r], acc);
  }
  return acc;
}

// Output layer (one padded 16-row tile): rows 4q+reg = output unit.  Two accumulators (k halves)
// keep the MFMA pipe busy; they are summed at the end.
template <int LDW = LDH>
__device__ __forceinline__ f4 layer_out(const float* Wl, const float* bl, const f4 (&in)[HID / 16], int j, int q) {
  constexpr int LDH = LDW;                    // (shadows the global stride inside this function)
  f4 acc0 = *reinterpret_cast<const f4*>(bl + 4 * q);
  f4 acc1 = {0.f, 0.f, 0.f, 0.f};
  f4 a[HID / 16];
#pragma unroll
  for (int nt = 0; nt < HID / 16; ++nt) a[nt] = *reinterpret_cast<const f4*>(Wl + j * LDH + 16 * nt + 4 * q);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc0 = mfma4(a[0][r], in[0][r], acc0);
    acc1 = mfma4(a[2][r], in[2][r], acc1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc0 = mfma4(a[1][r], in[1][r], acc0);
    acc1 = mfma4(a[3][r], in[3][r], acc1);
  }
  return acc0 + acc1;
}

// Full forward of one network for the 16 batch columns of a wave.
template <int KIN>
__device__ __forceinline__ f4 net_forward(const float* lds_net, const f4 (&x)[KIN / 16], f4 (&h1)[4], f4 (&h2)[4],
                                          int j, int q) {
  using L = NetLds<KIN>;
  layer_hidden<KIN / 16, true>(lds_net + L::W1, L::LD1, lds_net + L::B1, x, h1, j, q);
  layer_hidden<4, true>(lds_net + L::W2, LDH, lds_net + L::B2, h1, h2, j, q);
  return layer_out(lds_net + L::W3, lds_net + L::B3, h2, j, q);
}

// net_forward on the single-buffered layers (same arithmetic, same order: bit-identical results)
template <int KIN, int PAD = 4>
__device__ __forceinline__ f4 net_forward_lean(const float* lds_net, const f4 (&x)[KIN / 16], int j, int q) {
  using L = NetLds<KIN, PAD>;
  f4 h1[4], h2[4];
  layer_hidden_lean<KIN / 16, true>(lds_net + L::W1, L::LD1, lds_net + L::B1, x, h1, j, q);
  layer_hidden_lean<4, true>(lds_net + L::W2, L::LDW, lds_net + L::B2, h1, h2, j, q);
  return layer_out<L::LDW>(lds_net + L::W3, lds_net + L::B3, h2, j, q);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---- cross-lane sums without LDS traffic (ds_bpermute costs ~100+ cycles of exposed latency each
//      when one wave owns a SIMD).  The MFMA layouts put batch column j on lanes (l & 15) = one DPP row
//      and the k-slot q on the row index, so "sum over j" is a row reduction and "sum over q" a
//      reduction across the 4 rows.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f(float v) {
  const int s = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(s);
}
// sum over the 16 lanes of each DPP row; valid in lane 15 of the row (row_shr 1,2,4,8 prefix sums)
__device__ __forceinline__ float row_sum_lane15(float v) {
  v = dpp_add_f<0x111, 0xf>(v);
  v = dpp_add_f<0x112, 0xf>(v);
  v = dpp_add_f<0x114, 0xf>(v);
  v = dpp_add_f<0x118, 0xf>(v);
  return v;
}
// sum over the whole wave; valid in lane 63 (row_bcast:15 into rows 1,3, row_bcast:31 into rows 2,3)
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = row_sum_lane15(v);
  v = dpp_add_f<0x142, 0xa>(v);
  v = dpp_add_f<0x143, 0xc>(v);
  return v;
}
// sum over the 4 rows (lanes l, l^16, l^32, l^48), result in every lane: gfx950 v_permlane16_swap /
// v_permlane32_swap exchange odd/even rows and upper/lower halves in one VALU op each.
__device__ __forceinline__ float quad_row_sum(float v) {
  const unsigned x = __float_as_uint(v);
  const auto r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const float s1 = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
  const unsigned y = __float_as_uint(s1);
  const auto r32 = __builtin_amdgcn_permlane32_swap(y, y, false, false);
  return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;   // math.log(math.sqrt(2*math.pi))

}  // namespace spo
