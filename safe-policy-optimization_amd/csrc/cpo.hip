// CPO full-batch actor kernels, gfx950.
//
// Reference: safepo/single_agent/cpo.py
//   :356-365, :372-381  policy-gradient pair  g = -grad(-mean(ratio*adv_r)),  b = grad(mean(ratio*adv_c))
//   :132-157            fvp(): Hessian of mean(KL(old || cur)) over rows AND action dims at cur == old, times v,
//                       + 0.1*v (double backward over the full batch in the reference, 33 times per epoch)
//   :465-519            line search: loss_reward, loss_cost, KL for a candidate parameter vector
//   :534-571            critic fit (minibatch loop; runs on the persistent kernel of update.hip)
//
// One kernel, two modes, many workgroups: each workgroup walks 64-row chunks of the batch
// (chunk = blockIdx, += gridDim), accumulating weight-gradient tiles in MFMA accumulators across
// all its chunks, then writes ONE partial flat vector; a second kernel sums the partials in a
// fixed order (deterministic, no atomics).
//   MODE_SURR: forward, ratio = exp(logp - logp_old), dL/dlogp = sign*adv*ratio/M, backward.
//   MODE_FVP : forward, forward-mode tangent J v through the MLP (second LDS image holds v laid out
//              like the weights), cotangent u = (J v) / sigma^2 / (M*A), backward  ->  J^T u.
//              For a diagonal Gaussian with state-independent log_std the Hessian of the mean KL at
//              cur == old is block diagonal: J^T diag(1/sigma^2) J / (M*A) on the mean network and
//              2/A on log_std (added on the host together with the 0.1 damping).
#include "common.h"
#include "mlp_mfma.h"
#include "../../include/safepo_hip.h"

namespace {
using namespace spo;

constexpr int LDB = 64 + 4;
enum { MODE_SURR = 0, MODE_FVP = 1 };

template <int KIN, int MODE>
struct CpoLds {
  using L = NetLds<KIN>;
  static constexpr int W = 0;
  static constexpr int V = L::SIZE;                               // FVP only
  static constexpr int XT = (MODE == MODE_FVP ? 2 : 1) * L::SIZE;
  static constexpr int H1T = XT + KIN * LDB;
  static constexpr int H2T = H1T + HID * LDB;
  static constexpr int DZT = H2T + HID * LDB;                     // dZ2^T, then dZ1^T (two sub-phases)
  static constexpr int DOT = DZT + HID * LDB;
  static constexpr int RED = DOT + OUTP * LDB;
  static constexpr int SIZE = RED + 64;
};

struct CpoArgs {
  const float* theta;        // full flat parameter vector (actor segment is used)
  const float* vec;          // FVP: direction, flat ACTOR layout [log_std, W1, b1, W2, b2, W3, b3]
  const float* obs; const float* act; const float* logp_old; const float* adv;
  int64_t M; int D; int A; float sign;
  float* partial;            // [gridDim][Pa]
  double* partial_loss;      // [gridDim]
};

template <int KIN, int MODE>
__global__ __launch_bounds__(256, 1) void cpo_actor_kernel(CpoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using U = CpoLds<KIN, MODE>;
  using L = NetLds<KIN>;
  constexpr int NT1 = KIN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int D = a.D, A = a.A;
  const NetGeom g = net_geom(D, A, 2);
  const int ls_off = g.off - A;
  const int Pa = actor_size(D, A);
  stage_net<KIN>(a.theta, g, lds + U::W, tid, 256);
  if (MODE == MODE_FVP) {
    // the direction vector uses the actor's own flat layout: shift it so that net_geom offsets apply
    NetGeom gv = g;
    gv.off = A;                                   // W1 of the direction sits after its log_std block
    __syncthreads();
    stage_net<KIN>(a.vec, gv, lds + U::V, tid, 256);
  }
  __syncthreads();
  const float* Wl = lds + U::W;
  const float* Vl = lds + U::V;
  float* const red = lds + U::RED;

  float ivar[4], lsd[4], amask[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ai = 4 * q + r;
    const bool on = ai < A;
    const float lsv = on ? a.theta[ls_off + ai] : 0.f;
    const float sdv = expf(lsv);
    amask[r] = on ? 1.f : 0.f;
    ivar[r] = 1.f / (sdv * sdv);
    lsd[r] = on ? lsv + LOG_SQRT_2PI : 0.f;
  }
  const int orow = 16 * wave + 4 * q;
  const int mycol = 16 * wave + j;
  f4 aW1[NT1], aW2[4], aW3 = {0.f, 0.f, 0.f, 0.f}, dls = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) aW1[nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) aW2[nt] = f4{0.f, 0.f, 0.f, 0.f};
  float db1 = 0.f, db2 = 0.f, db3 = 0.f;
  double lsum = 0.0;
  const float inv_m = (float)(1.0 / (double)a.M);
  const float inv_ma = (float)(1.0 / ((double)a.M * (double)A));

  const int64_t nchunks = (a.M + 63) / 64;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t row = c * 64 + mycol;
    const bool cv = row < a.M;
    const int64_t smp = cv ? row : a.M - 1;
    f4 x[NT1];
    load_obs_tiles<KIN>(a.obs + smp * D, D, q, x);
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[U::XT + (16 * nt + 4 * q + e) * LDB + mycol] = x[nt][e];
    f4 h1[4], h2[4];
    const f4 o = net_forward<KIN>(Wl, x, h1, h2, j, q);
    f4 dO = {0.f, 0.f, 0.f, 0.f};
    if (MODE == MODE_SURR) {
      float lp = 0.f, dif[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ai = 4 * q + r;
        const float t = a.act[smp * A + (ai < A ? ai : 0)];
        dif[r] = (ai < A ? t : 0.f) - o[r];
        lp += -(dif[r] * dif[r]) * (0.5f * ivar[r]) - lsd[r];
      }
      lp += __shfl_xor(lp, 16);
      lp += __shfl_xor(lp, 32);
      const float ratio = expf(lp - a.logp_old[smp]);                 // cpo.py:358 / :374
      const float adv = a.adv[smp];
      if (q == 0 && cv) lsum += (double)(ratio * adv);
      const float dlp = cv ? a.sign * adv * ratio * inv_m : 0.f;      // d(sign*mean(ratio*adv))/dlogp
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = dif[r] * ivar[r];
        dO[r] = dlp * z;
        dls[r] = fmaf(dlp * amask[r], dif[r] * z - 1.f, dls[r]);
      }
    } else {
      // tangent (forward-mode) pass: t1 = V1 x + vb1 ; h1' = (1-h1^2) t1 ; t2 = W2 h1' + V2 h1 + vb2 ; ...
      f4 t1[4], t2[4];
      layer_hidden<NT1, false>(Vl + L::W1, L::LD1, Vl + L::B1, x, t1, j, q);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) t1[mt][r] *= fmaf(-h1[mt][r], h1[mt][r], 1.f);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) t2[mt] = *reinterpret_cast<const f4*>(Vl + L::B2 + 16 * mt + 4 * q);
      layer_accum<4>(Wl + L::W2, LDH, t1, t2, j, q);
      layer_accum<4>(Vl + L::W2, LDH, h1, t2, j, q);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) t2[mt][r] *= fmaf(-h2[mt][r], h2[mt][r], 1.f);
      f4 mud = *reinterpret_cast<const f4*>(Vl + L::B3 + 4 * q);
      mud = out_accum(Wl + L::W3, t2, mud, j, q);
      mud = out_accum(Vl + L::W3, h2, mud, j, q);
      const float cm = cv ? inv_ma : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) dO[r] = mud[r] * ivar[r] * amask[r] * cm;
    }

    // ---- backward through the MLP
    f4 dz2[4], dz1[4];
    {
      f4 acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt] = mfma4(Wl[L::W3 + (4 * q + r) * LDH + 16 * mt + j], dO[r], acc[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dz2[mt][r] = acc[mt][r] * fmaf(-h2[mt][r], h2[mt][r], 1.f);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[mt] = mfma4(Wl[L::W2 + (16 * nt + 4 * q + r) * LDH + 16 * mt + j], dz2[nt][r], acc[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dz1[mt][r] = acc[mt][r] * fmaf(-h1[mt][r], h1[mt][r], 1.f);
    }

    // ---- sub-phase A: dW3 (dO^T, H2^T) and dW2 (dZ2^T, H1^T)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = (16 * mt + 4 * q + r) * LDB + mycol;
        lds[U::H1T + f] = h1[mt][r];
        lds[U::H2T + f] = h2[mt][r];
        lds[U::DZT + f] = dz2[mt][r];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[U::DOT + (4 * q + r) * LDB + mycol] = dO[r];
    __syncthreads();
    {
      f4 az[4];
      float rs = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        az[r4] = *reinterpret_cast<const f4*>(lds + U::DZT + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
        rs += (az[r4][0] + az[r4][1]) + (az[r4][2] + az[r4][3]);
      }
      db2 += rs;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        f4 bh[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          bh[nt] = *reinterpret_cast<const f4*>(lds + U::H1T + (16 * nt + j) * LDB + 16 * r4 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) aW2[nt] = mfma4(az[r4][e], bh[nt][e], aW2[nt]);
      }
      rs = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        az[r4] = *reinterpret_cast<const f4*>(lds + U::DOT + j * LDB + 16 * r4 + 4 * q);
        rs += (az[r4][0] + az[r4][1]) + (az[r4][2] + az[r4][3]);
        const f4 bh = *reinterpret_cast<const f4*>(lds + U::H2T + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) aW3 = mfma4(az[r4][e], bh[e], aW3);
      }
      db3 += rs;
    }
    __syncthreads();
    // ---- sub-phase B: dW1 (dZ1^T, X^T); the dZ buffer is reused
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[U::DZT + (16 * mt + 4 * q + r) * LDB + mycol] = dz1[mt][r];
    __syncthreads();
    {
      f4 az[4];
      float rs = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        az[r4] = *reinterpret_cast<const f4*>(lds + U::DZT + (16 * wave + j) * LDB + 16 * r4 + 4 * q);
        rs += (az[r4][0] + az[r4][1]) + (az[r4][2] + az[r4][3]);
      }
      db1 += rs;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        f4 bh[NT1];
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
          bh[nt] = *reinterpret_cast<const f4*>(lds + U::XT + (16 * nt + j) * LDB + 16 * r4 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) aW1[nt] = mfma4(az[r4][e], bh[nt][e], aW1[nt]);
      }
    }
    __syncthreads();
  }

  // ---- emit this workgroup's partial flat vector (actor layout) -- every element written exactly once
  db1 += __shfl_xor(db1, 16); db1 += __shfl_xor(db1, 32);
  db2 += __shfl_xor(db2, 16); db2 += __shfl_xor(db2, 32);
  db3 += __shfl_xor(db3, 16); db3 += __shfl_xor(db3, 32);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float t = dls[r];
    t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8);
    if (j == 0) red[wave * 4 * 4 + 4 * q + r] = t;      // [wave][16]
  }
  const double lw = wave_sum_d(lsum);
  __shared__ double lred[4];
  if (lane == 0) lred[wave] = lw;
  __syncthreads();
  float* out = a.partial + (int64_t)blockIdx.x * Pa;
  const int w1 = A, b1 = w1 + HID * D, w2 = b1 + HID, b2 = w2 + HID * HID, w3 = b2 + HID, b3 = w3 + A * HID;
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * nt + j < D) out[w1 + (orow + r) * D + 16 * nt + j] = aW1[nt][r];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[w2 + (orow + r) * HID + 16 * nt + j] = aW2[nt][r];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * q + r < A) out[w3 + (4 * q + r) * HID + 16 * wave + j] = aW3[r];
  if (q == 0) { out[b1 + 16 * wave + j] = db1; out[b2 + 16 * wave + j] = db2; }
  if (wave == 0 && q == 0 && j < A) out[b3 + j] = db3;
  if (wave == 0 && j == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ai = 4 * q + r;
      if (ai < A) out[ai] = (red[ai] + red[16 + ai]) + (red[32 + ai] + red[48 + ai]);
    }
  }
  if (tid == 0) a.partial_loss[blockIdx.x] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
}

// out[i] = post_scale * sum_k partial[k][i] (+ diag terms of the FVP on the host side); fixed order.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* partial, int nparts, int P, float* out,
                                                              const double* ploss, double* loss_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P) {
    // sequential in k (the order every earlier round used: bit-identical sums), but sixteen loads are in flight before the
    // first of them is added -- one exposed memory latency per sixteen partials instead of one per partial (61 -> ~8 us for
    // 256 partial vectors of 8 592 floats)
    float acc = 0.f;
    int k = 0;
    for (; k + 16 <= nparts; k += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = partial[(int64_t)(k + u) * P + i];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; k < nparts; ++k) acc += partial[(int64_t)k * P + i];
    out[i] = acc;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) {
    double l = 0.0;
    for (int k = 0; k < nparts; ++k) l += ploss[k];
    loss_out[0] = l;
  }
}

// Line-search evaluation (cpo.py:473-491): sums of ratio*adv_r, ratio*adv_c and KL(old||new) (over
// rows AND dims) for the candidate parameters in theta.
struct LsArgs {
  const float* theta; const float* obs; const float* act; const float* logp_old; const float* adv_r;
  const float* adv_c; const float* mean_old; const float* log_std_old; double* partials; int64_t M; int D; int A;
};
template <int KIN>
__global__ __launch_bounds__(256) void cpo_linesearch_kernel(LsArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[NetLds<KIN>::SIZE];
  __shared__ double red[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int D = a.D, A = a.A;
  stage_net<KIN>(a.theta, net_geom(D, A, 2), lds, tid, 256);
  __syncthreads();
  const int ls_off = 2 * critic_size(D);
  float ivar[4], lsd[4], sdn[4], sdo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ai = 4 * q + r;
    const bool on = ai < A;
    const float lsv = on ? a.theta[ls_off + ai] : 0.f;
    sdn[r] = expf(lsv);
    sdo[r] = on ? expf(a.log_std_old[ai]) : 1.f;
    ivar[r] = 1.f / (sdn[r] * sdn[r]);
    lsd[r] = on ? logf(sdn[r]) + LOG_SQRT_2PI : 0.f;
  }
  double s_r = 0, s_c = 0, s_kl = 0;
  const int64_t ntiles = (a.M + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row = tile * 16 + j;
    const bool valid = row < a.M;
    const int64_t rr = valid ? row : a.M - 1;
    f4 x[KIN / 16];
    load_obs_tiles<KIN>(a.obs + rr * D, D, q, x);
    f4 h1[4], h2[4];
    const f4 mu = net_forward<KIN>(lds, x, h1, h2, j, q);
    float lp = 0.f, kl = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ai = 4 * q + r;
      if (ai < A) {
        const float dif = a.act[rr * A + ai] - mu[r];
        lp += -(dif * dif) * (0.5f * ivar[r]) - lsd[r];
        const float ratio = sdo[r] / sdn[r];
        const float vr = ratio * ratio;
        const float dm = (a.mean_old[rr * A + ai] - mu[r]) / sdn[r];
        kl += 0.5f * (vr + dm * dm - 1.f - logf(vr));
      }
    }
    lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);
    const float ratio = expf(lp - a.logp_old[rr]);
    if (valid) {
      s_kl += (double)kl;                       // every lane holds its own dims
      if (q == 0) { s_r += (double)(ratio * a.adv_r[rr]); s_c += (double)(ratio * a.adv_c[rr]); }
    }
  }
  s_r = wave_sum_d(s_r); s_c = wave_sum_d(s_c); s_kl = wave_sum_d(s_kl);
  if (lane == 0) { red[wave][0] = s_r; red[wave][1] = s_c; red[wave][2] = s_kl; }
  __syncthreads();
  if (tid < 3) a.partials[blockIdx.x * 3 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
__global__ void sum3_kernel(const double* partials, int n, double* out) {
  if (threadIdx.x < 3) {
    double s = 0;
    for (int k = 0; k < n; ++k) s += partials[k * 3 + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

int pick_kin(int D) { return D <= 16 ? 16 : D <= 32 ? 32 : 64; }

template <int MODE>
int launch_cpo(const CpoArgs& a, int blocks, hipStream_t st) {
  const int kin = pick_kin(a.D);
#define SPO_LAUNCH(K)                                                                                  \
  {                                                                                                    \
    const size_t sh = CpoLds<K, MODE>::SIZE * sizeof(float);                                           \
    static bool done_dev[spo::SPO_MAX_DEVICES] = {};        /* the attribute is per device */           \
    bool& done = done_dev[spo::current_device_slot()];                                                 \
    if (!done) {                                                                                       \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cpo_actor_kernel<K, MODE>),    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);         \
      if (e != hipSuccess) return spo::hip_check(e, "hipFuncSetAttribute(cpo)");                       \
      done = true;                                                                                     \
    }                                                                                                  \
    hipLaunchKernelGGL((cpo_actor_kernel<K, MODE>), dim3(blocks), dim3(256), sh, st, a);               \
  }
  if (kin == 16) SPO_LAUNCH(16) else if (kin == 32) SPO_LAUNCH(32) else SPO_LAUNCH(64)
#undef SPO_LAUNCH
  return 0;
}

int check(int D, int A) {
  if (D < 1 || D > 64) return spo::fail(-2, "cpo: obs_dim %d outside [1,64]", D);
  if (A < 1 || A > SPO_MAX_ACT) return spo::fail(-2, "cpo: act_dim %d outside [1,16]", A);
  return 0;
}

}  // namespace

extern "C" int spo_cpo_num_partials(int64_t rows) {
  int64_t n = (rows + 63) / 64;
  return (int)(n < 256 ? n : 256);
}

extern "C" int spo_cpo_surrogate_grad(const float* theta, const float* obs, const float* act, const float* logp_old,
                                      const float* adv, float sign, int64_t rows, int obs_dim, int act_dim,
                                      float* partial_ws, double* loss_ws, float* grad_out, double* loss_sum_out,
                                      void* stream) {
  if (int rc = check(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && obs && act && logp_old && adv && partial_ws && loss_ws && grad_out && loss_sum_out && rows > 0,
              "cpo_surrogate_grad: bad args");
  const int blocks = spo_cpo_num_partials(rows);
  CpoArgs a{theta, nullptr, obs, act, logp_old, adv, rows, obs_dim, act_dim, sign, partial_ws, loss_ws};
  hipStream_t st = (hipStream_t)stream;
  if (int rc = launch_cpo<MODE_SURR>(a, blocks, st)) return rc;
  const int Pa = spo::actor_size(obs_dim, act_dim);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((Pa + 63) / 64), dim3(64), 0, st, partial_ws, blocks, Pa, grad_out,
                     loss_ws, loss_sum_out);
  SPO_LAUNCH_CHECK("spo_cpo_surrogate_grad");
  return 0;
}

extern "C" int spo_cpo_fvp(const float* theta, const float* obs, const float* vec, int64_t rows, int obs_dim,
                           int act_dim, float* partial_ws, double* loss_ws, float* out, void* stream) {
  if (int rc = check(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && obs && vec && partial_ws && loss_ws && out && rows > 0, "cpo_fvp: bad args");
  const int blocks = spo_cpo_num_partials(rows);
  CpoArgs a{theta, vec, obs, nullptr, nullptr, nullptr, rows, obs_dim, act_dim, 1.f, partial_ws, loss_ws};
  hipStream_t st = (hipStream_t)stream;
  if (int rc = launch_cpo<MODE_FVP>(a, blocks, st)) return rc;
  const int Pa = spo::actor_size(obs_dim, act_dim);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((Pa + 63) / 64), dim3(64), 0, st, partial_ws, blocks, Pa, out,
                     loss_ws, (double*)nullptr);
  SPO_LAUNCH_CHECK("spo_cpo_fvp");
  return 0;
}

extern "C" int spo_cpo_linesearch_eval(const float* theta, const float* obs, const float* act, const float* logp_old,
                                       const float* adv_r, const float* adv_c, const float* mean_old,
                                       const float* log_std_old, int64_t rows, int obs_dim, int act_dim,
                                       double* partial_ws, int partial_capacity, double* sums3_out, void* stream) {
  if (int rc = check(obs_dim, act_dim)) return rc;
  SPO_REQUIRE(theta && obs && act && logp_old && adv_r && adv_c && mean_old && log_std_old && partial_ws && sums3_out &&
                  rows > 0, "cpo_linesearch_eval: bad args");
  int64_t blocks = (rows + 63) / 64;
  if (blocks > 1024) blocks = 1024;
  if (blocks * 3 > partial_capacity) blocks = partial_capacity / 3;
  SPO_REQUIRE(blocks >= 1, "cpo_linesearch_eval: partial capacity too small");
  LsArgs a{theta, obs, act, logp_old, adv_r, adv_c, mean_old, log_std_old, partial_ws, rows, obs_dim, act_dim};
  hipStream_t st = (hipStream_t)stream;
  switch (pick_kin(obs_dim)) {
    case 16: hipLaunchKernelGGL((cpo_linesearch_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL((cpo_linesearch_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((cpo_linesearch_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, st, a); break;
  }
  hipLaunchKernelGGL(sum3_kernel, dim3(1), dim3(64), 0, st, partial_ws, (int)blocks, sums3_out);
  SPO_LAUNCH_CHECK("spo_cpo_linesearch_eval");
  return 0;
}
