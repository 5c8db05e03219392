// PPO-Lagrangian minibatch update for WIDE OBSERVATIONS (obs_dim <= 512, act_dim <= 32, hidden [64, 64]), gfx950 -- round 5.
//
// The persistent kernels of update.hip keep one network in one CU's LDS; HumanoidVelocity's 376-wide first layer (96 KB of W1
// next to the staging images) does not fit, so that shape ran on the launch-per-layer wide path at 95 us per 64-row step against
// 10.6 us (ppo_lag.py:297-336; the reference takes any dims, model.py:131, and its default sweep includes 376 / 17,
// single_agent/benchmark.py:5-22).  Here the FIRST LAYER IS SPLIT OVER THE INPUT FEATURES:
//
//   grid = 3 networks x S slices (S = ceil(obs_dim / 64) <= 8), one persistent workgroup of 4 waves each, all co-resident;
//   workgroup (n, k) keeps W1[:, 64k .. 64k+63] of network n (and W2, W3, the biases, log_std: replicated) in LDS, gathers the
//   matching 64 features of the minibatch rows and computes the PARTIAL pre-activation W1_k x_k;
//   the S partials of a network are exchanged through uncached device memory (16 floats per lane as 6 packed {f, f, f, tag}
//   words, one store each; every workgroup polls all S x 6 words and adds them in slice order, so all S replicas hold the same
//   bits), then bias + tanh, layers 2 / 3, loss, backward and the weight gradients run replicated -- identical instructions
//   on identical data -- except dW1, of which a workgroup computes (and owns the Adam state of) its own 64 columns;
//   the joint clip_grad_norm_ (ppo_lag.py:325) sums one ||g||^2 granule per workgroup: the slice's share of W1, plus
//   everything else from slice 0 only.
//
// The arithmetic per element is that of ppo_update_kernel (same MFMA chaining, loss, Adam); the first layer's dot products
// are summed slice by slice instead of in one chain (rounding-level difference, tests: 1e-5 on the first steps + the fp64
// drift envelope).  Two output tiles for the actor (act_dim <= 32).  Clipped-surrogate loss, batch <= 64, one GPU.
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
constexpr int KS_ZW = 6;                         // packed words per lane of a partial pre-activation (16 floats)

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
//      [32 + 32 wave + a] d(log_std) partials, [160 + a] log_std mirror, [192 + i] polled ||g||^2 granules, [224 + i] sum p^2 granules

struct KsArgs {
  float* theta; float* adam_m; float* adam_v;
  const float* obs; const float* act; const float* logp_old; const float* tgt_r; const float* tgt_c; const float* adv;
  const int32_t* perm; int64_t M;
  spo_ppo_cfg cfg;
  float* losses;                       // [nsteps][3]
  char* zbuf;                          // partial pre-activations: [2 parities][3 nets][KS_MAX_SLICES][256 lanes][KS_ZW] x 16 B
  unsigned long long* gran;            // [2 parities][2 kinds][3 * KS_MAX_SLICES] {tag, value} granules
  int* err;
  double pow_b1, pow_b2;
  unsigned tag_base;                   // tags of this launch: tag_base + step + 1 (never reused: no clearing between launches)
  int S;
};

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(char* p, const u4v v) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (s_nop 1: a VMEM store of more than 8 bytes reads its data registers up to two wait states after issue, update.hip st16_sys)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
// The six words of up to three source slices in ONE asm statement that ends with the wait: the compiler does not know that the
// statement's outputs are loads in flight, so between a bare load statement and a later s_waitcnt statement it is free to copy or
// spill the destination registers -- saving whatever they held BEFORE the data landed (seen here with 36 words in flight: polls
// that could never succeed).  Inside one statement nothing can come between.  A lane's six words are 96 contiguous bytes.
__device__ __forceinline__ void ld6x3(const char* p0, const char* p1, const char* p2, u4v (&a)[KS_ZW], u4v (&b)[KS_ZW], u4v (&c)[KS_ZW]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "global_load_dwordx4 %0, %18, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %18, off offset:16 sc0 sc1\n\t"
      "global_load_dwordx4 %2, %18, off offset:32 sc0 sc1\n\t"
      "global_load_dwordx4 %3, %18, off offset:48 sc0 sc1\n\t"
      "global_load_dwordx4 %4, %18, off offset:64 sc0 sc1\n\t"
      "global_load_dwordx4 %5, %18, off offset:80 sc0 sc1\n\t"
      "global_load_dwordx4 %6, %19, off sc0 sc1\n\t"
      "global_load_dwordx4 %7, %19, off offset:16 sc0 sc1\n\t"
      "global_load_dwordx4 %8, %19, off offset:32 sc0 sc1\n\t"
      "global_load_dwordx4 %9, %19, off offset:48 sc0 sc1\n\t"
      "global_load_dwordx4 %10, %19, off offset:64 sc0 sc1\n\t"
      "global_load_dwordx4 %11, %19, off offset:80 sc0 sc1\n\t"
      "global_load_dwordx4 %12, %20, off sc0 sc1\n\t"
      "global_load_dwordx4 %13, %20, off offset:16 sc0 sc1\n\t"
      "global_load_dwordx4 %14, %20, off offset:32 sc0 sc1\n\t"
      "global_load_dwordx4 %15, %20, off offset:48 sc0 sc1\n\t"
      "global_load_dwordx4 %16, %20, off offset:64 sc0 sc1\n\t"
      "global_load_dwordx4 %17, %20, off offset:80 sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]),
        "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5])
      : "v"(p0), "v"(p1), "v"(p2)
      : "memory");
#endif
}
__device__ __forceinline__ float pin(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int pin(int v) { asm volatile("" : "+v"(v)); return v; }
// (one launch at a time per device: the exchange scratch below is per device, like the kernel's use by one engine on one stream)
constexpr unsigned KS_SPIN_LIMIT = 1u << 22;

struct KsCol {                         // per-column inputs of one minibatch, prefetched one step ahead (raw loads; settled at pick-up)
  f4 x[4];                             // this slice's observation tiles (B operand of layer 1)
  f4 actv[KS_NO];                      // actor: act[16 t + 4 q ..]
  float t0, t1;                        // critic: target ; actor: logp_old, adv
};

__global__ __launch_bounds__(256, 1) void ppo_update_ks_kernel(KsArgs a) {
  if (blockIdx.x & 7) return;                    // placement hint (update.hip): the working blocks land on one XCD and share its L2
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using L = KsLds;
  const int wg = (int)(blockIdx.x >> 3);
  const int S = a.S, net = wg / S, ks = wg - net * S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int D = a.cfg.obs_dim, A = a.cfg.act_dim, B = a.cfg.batch;
  const NetGeom g = net_geom(D, A, net);
  const bool is_actor = (net == 2), first = (ks == 0);
  const int OUT = g.OUT, nto = is_actor ? KS_NO : 1;
  const int ls_off = g.off - A;                  // actor only
  const int c_lo = 64 * ks, DS = (D - c_lo) < 64 ? (D - c_lo) : 64;      // this slice's columns [c_lo, c_lo + DS)
  float* const red = lds + L::RED;
  float* const st_m = a.adam_m; float* const st_v = a.adam_v;

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
  f4 mW1[4], vW1[4], mW2[4], vW2[4], mW3[KS_NO], vW3[KS_NO], mls[KS_NO], vls[KS_NO];
  float mb1, vb1, mb2, vb2, mb3[KS_NO], vb3[KS_NO];
  const bool own_b = (q == 0);
  const bool own_w0 = (wave == 0 && q == 0);                    // b3[16 t + j], and (j == 0) log_std
  const bool own_ls = is_actor && wave == 0 && j == 0;
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
      mls[t][r] = (is_actor && o < A) ? a.adam_m[ls_off + o] : 0.f;
      vls[t][r] = (is_actor && o < A) ? a.adam_v[ls_off + o] : 0.f;
    }
    const int ob = 16 * t + j;
    mb3[t] = ob < OUT ? a.adam_m[g.b3() + ob] : 0.f; vb3[t] = ob < OUT ? a.adam_v[g.b3() + ob] : 0.f;
  }
  mb1 = a.adam_m[g.b1() + 16 * wave + j]; vb1 = a.adam_v[g.b1() + 16 * wave + j];
  mb2 = a.adam_m[g.b2() + 16 * wave + j]; vb2 = a.adam_v[g.b2() + 16 * wave + j];

  const float b1c = a.cfg.beta1, b2c = a.cfg.beta2, eps = a.cfg.adam_eps;
  double pw1 = a.pow_b1, pw2 = a.pow_b2;
  const float lr = is_actor ? a.cfg.lr_actor : a.cfg.lr_critic;
  const float l2 = (!is_actor && a.cfg.use_critic_norm) ? a.cfg.l2_coef : 0.f;
  const float l2x2 = 2.f * l2;
  const float vcoef = (net == 0 && a.cfg.use_value_coefficient) ? 2.f : 1.f;
  const float clip_lo = 1.f - a.cfg.clip, clip_hi = 1.f + a.cfg.clip;
  const float* tgt = (net == 0) ? a.tgt_r : a.tgt_c;
  const int64_t nsteps = (a.M + B - 1) / B;
  const int mycol = 16 * wave + j;

  auto perm_pos = [&](int64_t s) -> int64_t {
    const int64_t base = s * B;
    const int64_t rem = a.M - base;
    const int ncols = (int)(rem < B ? rem : B);
    return base + (mycol < ncols ? mycol : 0);
  };
  auto fetch = [&](int64_t smp, KsCol& cd) {
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
          cd.actv[t][r] = a.act[smp * A + (ai < A ? ai : 0)];         // unconditional loads; pads selected at pick-up
        }
    }
  };
  auto settle = [&](KsCol& cd) {
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
        }
    }
  };

  KsCol nxt;
  int smp1 = 0;
  fetch((int64_t)a.perm[perm_pos(0)], nxt);
  if (nsteps > 1) smp1 = a.perm[perm_pos(1)];

  for (int64_t s = 0; s < nsteps; ++s) {
    const int64_t base = s * B;
    const int64_t rem = a.M - base;
    const int ncols = (int)(rem < B ? rem : B);
    const float inv_n = 1.f / (float)ncols;
    const bool cv = mycol < ncols;
    const unsigned tag = a.tag_base + (unsigned)s + 1u;
    const int par = (int)(s & 1);

    KsCol cur = nxt;
    settle(cur);
    const int smp_next = pin(smp1);
    const int64_t pos2 = (s + 2 < nsteps) ? perm_pos(s + 2) : 0;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[L::XT + (16 * nt + 4 * q + e) * LDB + mycol] = cur.x[nt][e];

    // ---- layer 1: this slice's partial pre-activation, then the sum over the slices (slice order: identical bits everywhere)
    f4 z1[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) z1[mt] = f4{0.f, 0.f, 0.f, 0.f};
    layer_accum<4>(lds + L::W1, LDH, cur.x, z1, j, q);
    if (s + 1 < nsteps) fetch((int64_t)smp_next, nxt);               // prefetch: next step's columns, then the index after
    if (s + 2 < nsteps) smp1 = a.perm[pos2];
    if (S > 1) {
      // [parity][network][slice][lane][6 words]: a lane's words are 96 contiguous bytes
      char* const zb = a.zbuf + ((size_t)(par * 3 + net) * KS_MAX_SLICES * 256 + tid) * (KS_ZW * 16);
      {
        char* const mine = zb + (size_t)ks * 256 * (KS_ZW * 16);
#pragma unroll
        for (int w = 0; w < KS_ZW; ++w) {
          u4v word;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int f = 3 * w + i;
            word[i] = f < 16 ? __float_as_uint(z1[(f < 16 ? f : 0) >> 2][(f < 16 ? f : 0) & 3]) : 0u;
          }
          word[3] = tag;
          st16(mine + w * 16, word);
        }
      }
      // three source slices per round trip, in slice order (the sum below is the same chain in every workgroup of the network)
#pragma unroll 1
      for (int k0 = 0; k0 < S; k0 += 3) {
        const int k1 = k0 + 1 < S ? k0 + 1 : k0, k2 = k0 + 2 < S ? k0 + 2 : k1;
        const char* const p0 = zb + (size_t)k0 * 256 * (KS_ZW * 16);
        const char* const p1 = zb + (size_t)k1 * 256 * (KS_ZW * 16);
        const char* const p2 = zb + (size_t)k2 * 256 * (KS_ZW * 16);
        u4v za[KS_ZW], zb2[KS_ZW], zc[KS_ZW];
        unsigned spins = 0;
        for (;;) {
          ld6x3(p0, p1, p2, za, zb2, zc);
          bool ok = true;
#pragma unroll
          for (int w = 0; w < KS_ZW; ++w) ok = ok && (za[w][3] == tag) && (zb2[w][3] == tag) && (zc[w][3] == tag);
          if (ok) break;
          if (++spins > KS_SPIN_LIMIT) { *a.err = 1; break; }        // bounded: never hang the GPU
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) {
          float acc = k0 == 0 ? __uint_as_float(za[f / 3][f % 3]) : z1[f >> 2][f & 3] + __uint_as_float(za[f / 3][f % 3]);
          if (k0 + 1 < S) acc += __uint_as_float(zb2[f / 3][f % 3]);
          if (k0 + 2 < S) acc += __uint_as_float(zc[f / 3][f % 3]);
          z1[f >> 2][f & 3] = acc;
        }
      }
    }
    f4 h1[4], h2[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) h1[mt] = fast_tanh4(z1[mt] + *reinterpret_cast<const f4*>(lds + L::B1 + 16 * mt + 4 * q));
    layer_hidden<4, true>(lds + L::W2, LDH, lds + L::B2, h1, h2, j, q);
    f4 o[KS_NO];
    o[0] = layer_out(lds + L::W3, lds + L::B3, h2, j, q);
    o[1] = f4{0.f, 0.f, 0.f, 0.f};
    if (nto > 1) o[1] = layer_out(lds + L::W3 + 16 * LDH, lds + L::B3 + 16, h2, j, q);

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
      if (lane == 63) red[wave] = ls;
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

    // ---- dW[o][i] = sum_b dZ[b][o] * Hprev[b][i]; wave w owns rows 16w .. 16w+15 (W1 slice, W2) / h2 units 16w .. (W3)
    f4 aW1[4], aW2[4], aW3[KS_NO];
    float db1, db2, db3[KS_NO];
    {
      f4 az1[4], az2[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        az1[r4] = *reinterpret_cast<const f4*>(lds + L::DZ1T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
        az2[r4] = *reinterpret_cast<const f4*>(lds + L::DZ2T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { aW1[nt] = f4{0.f, 0.f, 0.f, 0.f}; aW2[nt] = f4{0.f, 0.f, 0.f, 0.f}; }
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
        aW3[t] = f4{0.f, 0.f, 0.f, 0.f};
        db3[t] = 0.f;
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
          aW3[t] = w3a + w3b;
          float rs3 = 0.f;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) rs3 += (az3[r4][0] + az3[r4][1]) + (az3[r4][2] + az3[r4][3]);
          db3[t] = quad_row_sum(rs3);
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
      db1 = quad_row_sum(rs1); db2 = quad_row_sum(rs2);
    }
    const float loss_data = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
    if (is_actor) {
#pragma unroll
      for (int t = 0; t < KS_NO; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ai = 16 * t + 4 * q + r;
          dls[t][r] = (red[32 + ai] + red[64 + ai]) + (red[96 + ai] + red[128 + ai]);       // 0 on pad rows
        }
    }

    // ---- L2 regulariser of the critics (weights AND biases, ppo_lag.py:310-314), norm shares: the W1 slice / everything else
    float gsq1 = 0.f, psq1 = 0.f, gsqr = 0.f, psqr = 0.f;
    f4 pW1[4], pW2[4], pW3[KS_NO], pls[KS_NO];
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
        pls[t][r] = is_actor ? red[160 + 16 * t + 4 * q + r] : 0.f;
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
      // biases and log_std are replicated across lanes; each replica runs the same Adam, only one of them counts towards the norms
      db1 = vcoef * fmaf(l2x2, pb1, db1); db2 = vcoef * fmaf(l2x2, pb2, db2);
      const float wb = own_b ? 1.f : 0.f, w0 = own_w0 ? 1.f : 0.f, wls = own_ls ? 1.f : 0.f;
      gsqr += wb * (db1 * db1 + db2 * db2);
      psqr += wb * (pb1 * pb1 + pb2 * pb2);
#pragma unroll
      for (int t = 0; t < KS_NO; ++t) {
        db3[t] = vcoef * fmaf(l2x2, pb3[t], db3[t]);
        gsqr += w0 * (db3[t] * db3[t]);
        psqr += w0 * (pb3[t] * pb3[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) gsqr = fmaf(wls * dls[t][r], dls[t][r], gsqr);
      }
    }
    gsq1 = wave_sum_lane63(gsq1); psq1 = wave_sum_lane63(psq1);
    gsqr = wave_sum_lane63(gsqr); psqr = wave_sum_lane63(psqr);
    if (lane == 63) { red[4 + wave] = gsq1; red[8 + wave] = gsqr; red[12 + wave] = psq1; red[16 + wave] = psqr; }
    __syncthreads();

    // ---- joint clip_grad_norm_ over all networks (ppo_lag.py:325): one granule per workgroup
    unsigned long long* const gg = a.gran + (size_t)par * 2 * 3 * KS_MAX_SLICES;
    unsigned long long* const gp = gg + 3 * KS_MAX_SLICES;
    const int nwg = 3 * S;
    if (tid == 0) {
      const float gs = ((red[4] + red[5]) + (red[6] + red[7])) + (first ? ((red[8] + red[9]) + (red[10] + red[11])) : 0.f);
      const float ps = ((red[12] + red[13]) + (red[14] + red[15])) + (first ? ((red[16] + red[17]) + (red[18] + red[19])) : 0.f);
      __hip_atomic_store(gg + net * KS_MAX_SLICES + ks, ((unsigned long long)tag << 32) | __float_as_uint(gs), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gp + net * KS_MAX_SLICES + ks, ((unsigned long long)tag << 32) | __float_as_uint(ps), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    pw1 *= (double)b1c; pw2 *= (double)b2c;
    float step_size, inv_bc2s;
    adam_scalars(lr, pw1, pw2, step_size, inv_bc2s);
    if (tid < 2 * 3 * KS_MAX_SLICES) {
      const int kind = tid / (3 * KS_MAX_SLICES), idx = tid - kind * 3 * KS_MAX_SLICES;
      const int n2 = idx / KS_MAX_SLICES, k2 = idx - n2 * KS_MAX_SLICES;
      float val = 0.f;
      if (k2 < S) {
        unsigned long long* const src = (kind ? gp : gg) + idx;
        unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((unsigned)(v >> 32) != tag) {
          if (++spins > KS_SPIN_LIMIT) { *a.err = 1; break; }         // bounded: never hang the GPU
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        val = __uint_as_float((unsigned)v);
      }
      red[192 + tid] = val;                                            // [192 + 24 kind + 8 net + slice]
    }
    __syncthreads();
    float total_sq = 0.f;
    for (int i = 0; i < 3 * KS_MAX_SLICES; ++i) total_sq += red[192 + i];   // fixed order: identical in every workgroup
    const float norm = sqrtf(total_sq);
    float coef = a.cfg.max_grad_norm / (norm + 1e-6f);                // clip_grad_norm_ (torch): eps 1e-6
    coef = coef > 1.f ? 1.f : coef;
    if (tid == 0 && first) {
      float pp = 0.f;
      for (int k2 = 0; k2 < KS_MAX_SLICES; ++k2) pp += red[192 + 3 * KS_MAX_SLICES + net * KS_MAX_SLICES + k2];
      a.losses[s * 3 + net] = is_actor ? -loss_data : loss_data + l2 * pp;
    }
    (void)nwg;

    // ---- Adam (torch.optim.Adam, ppo_lag.py:104-117), parameters back into the LDS image
#define KS_ADAM(DST, P, G, M, V)                                                       \
  {                                                                                    \
    const AdamOut _o = adam1((P), (G) * coef, (M), (V), b1c, b2c, eps, step_size, inv_bc2s);  \
    (M) = _o.m; (V) = _o.v; (DST) = _o.p;                                              \
  }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {                                    // pad columns stay exactly 0
        KS_ADAM(lds[L::W1 + (orow + r) * LDH + 16 * nt + j], pW1[nt][r], aW1[nt][r], mW1[nt][r], vW1[nt][r])
        KS_ADAM(lds[L::W2 + (orow + r) * LDH + 16 * nt + j], pW2[nt][r], aW2[nt][r], mW2[nt][r], vW2[nt][r])
      }
#pragma unroll
    for (int t = 0; t < KS_NO; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r)                                      // pad rows stay exactly 0
        KS_ADAM(lds[L::W3 + (16 * t + 4 * q + r) * LDH + 16 * wave + j], pW3[t][r], aW3[t][r], mW3[t][r], vW3[t][r])
      float np3;
      KS_ADAM(np3, pb3[t], db3[t], mb3[t], vb3[t])
      lds[L::B3 + 16 * t + j] = np3;
      if (is_actor) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float nl;
          KS_ADAM(nl, pls[t][r], dls[t][r], mls[t][r], vls[t][r])
          red[160 + 16 * t + 4 * q + r] = nl;
        }
      }
    }
    {
      float np1, np2;
      KS_ADAM(np1, pb1, db1, mb1, vb1)
      KS_ADAM(np2, pb2, db2, mb2, vb2)
      lds[L::B1 + 16 * wave + j] = np1;
      lds[L::B2 + 16 * wave + j] = np2;
    }
#undef KS_ADAM
    __syncthreads();
  }

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
        if (own_ls && o < A) {
          a.theta[ls_off + o] = red[160 + o];
          st_m[ls_off + o] = mls[t][r]; st_v[ls_off + o] = vls[t][r];
        }
      }
      const int ob = 16 * t + j;
      if (own_w0 && ob < OUT) {
        a.theta[g.b3() + ob] = lds[L::B3 + ob];
        st_m[g.b3() + ob] = mb3[t]; st_v[g.b3() + ob] = vb3[t];
      }
    }
    if (own_b) {
      const int ob = 16 * wave + j;
      a.theta[g.b1() + ob] = lds[L::B1 + ob]; st_m[g.b1() + ob] = mb1; st_v[g.b1() + ob] = vb1;
      a.theta[g.b2() + ob] = lds[L::B2 + ob]; st_m[g.b2() + ob] = mb2; st_v[g.b2() + ob] = vb2;
    }
  }
}

// Exchange scratch of the kernel: the partial pre-activation words (uncached: visible to the other workgroups' polling loads
// without a cache flush) and the norm granules.  One block per device, allocated on first use, zeroed once (tags never repeat).
struct KsScratch { char* z; unsigned long long* gran; };
constexpr size_t KS_Z_BYTES = (size_t)2 * 3 * KS_MAX_SLICES * 256 * KS_ZW * 16;
constexpr size_t KS_G_BYTES = (size_t)2 * 2 * 3 * KS_MAX_SLICES * 8;
KsScratch g_ks[SPO_MAX_DEVICES] = {};
unsigned g_ks_tag[SPO_MAX_DEVICES] = {};
std::mutex g_ks_mu;

int ks_scratch(KsScratch* out, unsigned* tag_base, unsigned nsteps) {
  const int dev = current_device_slot();
  std::lock_guard<std::mutex> lk(g_ks_mu);
  if (!g_ks[dev].z) {
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, KS_Z_BYTES + KS_G_BYTES, hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipExtMallocWithFlags(&p, KS_Z_BYTES + KS_G_BYTES, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) return spo::hip_check(e, "hipExtMallocWithFlags(ks scratch)");
    if (int rc = spo::hip_check(hipMemset(p, 0, KS_Z_BYTES + KS_G_BYTES), "hipMemset(ks scratch)")) { (void)hipFree(p); return rc; }
    g_ks[dev].z = static_cast<char*>(p);
    g_ks[dev].gran = reinterpret_cast<unsigned long long*>(static_cast<char*>(p) + KS_Z_BYTES);
    g_ks_tag[dev] = 16u;
  }
  *out = g_ks[dev];
  *tag_base = g_ks_tag[dev];
  g_ks_tag[dev] += nsteps + 2u;                    // (wraps after 4e9 steps: a tag then meets words 2^32 steps old)
  return 0;
}

}  // namespace

extern "C" int spo_ks_supported(int obs_dim, int act_dim, int batch) {
  return (obs_dim >= 1 && obs_dim <= 64 * KS_MAX_SLICES && act_dim >= 1 && act_dim <= KS_OUT && batch >= 1 && batch <= 64) ? 1 : 0;
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
  hipStream_t st = (hipStream_t)stream;
  if (int rc = spo::hip_check(hipMemsetAsync(sync_ws, 0, 64, st), "hipMemsetAsync(sync_ws)")) return rc;
  KsArgs a{};
  a.theta = theta; a.adam_m = adam_m; a.adam_v = adam_v;
  a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.perm = perm; a.M = M; a.cfg = *cfg_host; a.losses = losses_out;
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(sync_ws) + 64);
  a.pow_b1 = pow((double)cfg_host->beta1, (double)adam_step_host);
  a.pow_b2 = pow((double)cfg_host->beta2, (double)adam_step_host);
  a.S = (cfg_host->obs_dim + 63) / 64;
  const int64_t nsteps = (M + cfg_host->batch - 1) / cfg_host->batch;
  SPO_REQUIRE(nsteps < (1ll << 31), "update_iter_ks: too many minibatch steps in one launch");
  KsScratch sc;
  if (int rc = ks_scratch(&sc, &a.tag_base, (unsigned)nsteps)) return rc;
  a.zbuf = sc.z; a.gran = sc.gran;
  const size_t sh = KsLds::SIZE * sizeof(float);
  static bool attr_done[SPO_MAX_DEVICES] = {};
  const int dslot = current_device_slot();
  if (!attr_done[dslot]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_update_ks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(update_ks)");
    attr_done[dslot] = true;
  }
  hipLaunchKernelGGL(ppo_update_ks_kernel, dim3(8 * (3 * a.S - 1) + 1), dim3(256), sh, st, a);
  SPO_LAUNCH_CHECK("spo_ppo_lag_update_iter_ks");
  return 0;
}
