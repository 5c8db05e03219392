// One-launch minibatch gradient of the wide-network path, split over the ROWS (round 6).  gfx950 only.
//
// The wide path (any hidden_sizes: reference safepo/common/model.py:30-48,131; the minibatch step safepo/single_agent/ppo_lag.py:306-324)
// took a step of the reference's default 64-row minibatch as gather + one-workgroup-per-network forward (csrc/mlp_small.hip) + loss +
// one-workgroup-per-network backward + three optimiser launches: 89 us at hidden [128, 128], of which 24 + 36 us are the two
// mlp_small launches -- one CU per network staging every weight matrix through LDS behind barriers.  Nothing in forward / loss /
// backward couples the ROWS of a minibatch except the sums over rows in the weight gradients and the loss values, so here a
// workgroup carries ONE 16-row group of ONE network through all of it:
//   * grid = networks x ceil(rows / 16) workgroups of 8 waves; no workgroup waits for another (no exchange, no co-residency need);
//   * the rows are read through the minibatch's index window (perm[cursor .. cursor + rows): the DataLoader batch of
//     ppo_lag.py:298-305) -- no gather launch;
//   * activations of every level live in LDS twice: a row image [16][width + 4] (B operand of the next layer, 1 - h^2 of the
//     backward) and a transposed image [width][16 + 4] (B operand of the weight-gradient product, whose reduction index is the
//     row); dZ likewise (row image: B operand of dH = dZ W; transposed: A operand of dW = dZ^T H);
//   * weights come straight from global memory as MFMA A fragments (16 bytes per lane): 8 waves deal the 16-unit output tiles, so
//     no weight is staged and no barrier sits inside a layer -- one barrier per layer each way; a wave's first tile of the NEXT
//     layer (eight input tiles: all of a 128-wide layer) is requested a whole stage ahead into a second register set;
//   * the loss of the network (MSE of a critic, the clipped surrogate and d(log_std) of the actor: the arithmetic of
//     wide_ppo_loss_kernel, csrc/ma_net.hip) is evaluated between forward and backward on the 16 rows, with the GLOBAL 1 / rows;
//   * every workgroup writes its row group's PARTIAL gradient (theta's layout) and loss sums to parts[group]; a second, tiny launch
//     (spo_wide_reduce_parts) adds the groups in fixed order into the flat gradient -- deterministic, and exactly where the
//     data-parallel all-reduce and spo_wide_clip_adam take over.
// fp32 v_mfma_f32_16x16x4_f32 throughout (bitwise an fmaf chain); results differ from the launch-per-layer path in summation
// order only.
// Measured (hidden [128, 128], 60 / 8, 64 rows, one MI355X, profiles/r06/wide_step.txt, wide_rows_phase_cycles.txt): the kernel
// 19-19.6 us (of a 30.1 us step) of which ~4 us are MFMA issue time (1 120 useful v_mfma per workgroup, two waves per SIMD), the rest the serial chain of
// a stage -- B-operand reads, the MFMA chain, tanh, image stores, barrier -- nine times, plus cursor -> index -> observation rows
// (three dependent round trips, 2.7 us).  What did NOT help, each measured: warming the XCD's L2 with the parameters at the
// kernel's start (32.3 against 32.6 us per step), eight-tile load batches without the prefetch (slower: the loads are not the
// bound), requesting the next layer behind the current layer's MFMAs instead of in front of them (no change).  What did: the
// host-tabulated LDS map (-1.7 us: mr_level's loops were dependent scalar loads at every stage), skipping tile pairs beyond a
// layer's width (-1 us), one kernel body per load form (the VEC template parameter: with both forms in one body the compiler
// fenced every fragment's load behind the other form's registers).
#include "common.h"
#include "mlp_mfma.h"
#include "adam.h"
#include <cstdlib>
#include "../../include/safepo_hip.h"

namespace {
using namespace spo;

constexpr int MR_TS = 20;                          // row stride of a transposed image: 16 rows + 4 (16-byte aligned, bank-rotated)
constexpr int MR_MAX_GROUPS = 16;                  // row groups of one launch (256 rows)
constexpr size_t MR_MAX_LDS = 160 * 1024;          // all of a CU's LDS (the kernel has no static LDS): [256, 256] at 60 / 8 needs 157.4 KB
// behind the images: row indices (int64[16]), per-row scalars, log_std and 1 / sigma^2, the d(log_std) terms, the rows' actions
// (the act_dim-sized pieces at the launch's act_dim rounded up to 4 -- AP -- so that a narrow action vector leaves the space to the images)
constexpr int MR_RIDX = 0, MR_ROWA = 32, MR_ROWL = 48, MR_RADV = 64, MR_RLOGP = 80, MR_RTGT = 96, MR_LS = 112;
__host__ __device__ inline int mr_ap(int A) { return A < 1 ? 4 : (A + 3) & ~3; }
__host__ __device__ inline int mr_ivar(int AP) { return MR_LS + AP; }
__host__ __device__ inline int mr_dls(int AP) { return MR_LS + 2 * AP; }
__host__ __device__ inline int mr_ract(int AP) { return MR_LS + 2 * AP + 16 * AP; }
__host__ __device__ inline int mr_misc(int AP) { return MR_LS + 2 * AP + 32 * AP; }

constexpr float LOG_SQRT_2PI_F = 0.91893853320467274178f;               // (as csrc/ma_net.hip)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // four consecutive floats at any dword address

struct MrNet {
  int n;                                           // Linear layers
  int kind;                                        // 0 reward critic, 1 cost critic, 2 actor
  int d[SPO_MLP_MAX_LAYERS + 1];                   // widths: d[0] = obs_dim, d[n] = 1 or act_dim
  int64_t w[SPO_MLP_MAX_LAYERS], b[SPO_MLP_MAX_LAYERS];   // offsets of W_l [d[l+1]][d[l]] and b_l in theta
  // the LDS map (mr_level / mr_pz / mr_images_end), tabulated by the host: the kernel reads an entry with one scalar load where the
  // loops of mr_level cost it a chain of dependent ones at every stage (~1 us per stage, measured)
  int pr[SPO_MLP_MAX_LAYERS + 1], hr[SPO_MLP_MAX_LAYERS + 1], ht[SPO_MLP_MAX_LAYERS + 1];
  int pz, zbase;
};
struct MrArgs {
  const float* theta; const float* obs; const float* act; const float* logp_old; const float* tgt_r; const float* tgt_c; const float* adv;
  const int64_t* idx; const int64_t* cursor;
  float* parts; int64_t stride; int64_t P; int64_t ls_off;
  int rows, R, n_nets, A;
  float clip;
  MrNet net[3];
};

__host__ __device__ inline int mr_up16(int v) { return (v + 15) & ~15; }
__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }

// LDS map of a network: level v (0 = the observations, v = output of layer v) has a row image at hr and, below the output level, a
// transposed image at ht; behind them two dZ row images and two transposed ones (ping-pong over the layers), then mr_misc() floats.
struct MrLevel { int pr, hr, ht; };
__host__ __device__ inline MrLevel mr_level(const MrNet& nn, int v) {
  MrLevel L{0, 0, 0};
  int off = 0;
  for (int u = 0; u <= v; ++u) {
    L.pr = mr_up16(nn.d[u]);
    L.hr = off; off += 16 * (L.pr + 4);
    L.ht = off;
    if (u < nn.n) off += L.pr * MR_TS;
  }
  return L;
}
__host__ __device__ inline int mr_pz(const MrNet& nn) {
  int pz = 16;
  for (int u = 1; u <= nn.n; ++u) pz = mr_up16(nn.d[u]) > pz ? mr_up16(nn.d[u]) : pz;
  return pz;
}
__host__ __device__ inline int mr_images_end(const MrNet& nn) {
  const MrLevel L = mr_level(nn, nn.n);
  return L.hr + 16 * (L.pr + 4);
}
__host__ __device__ inline int mr_lds_floats(const MrNet& nn, int A) {
  const int pz = mr_pz(nn);
  return mr_images_end(nn) + 2 * 16 * (pz + 4) + 2 * pz * MR_TS + mr_misc(mr_ap(A));
}

// A fragment of a row-major weight row: W[row][16 nt + 4q .. + 3] (vec: K % 4 == 0, one 16-byte load at any dword address;
// otherwise four clamped dword loads).  Columns >= K hold finite garbage: the B operand is zero there.
template <bool VEC>
__device__ __forceinline__ f4 load_w_frag(const float* __restrict__ wr, int K, int nt, int q) {
  const int k0 = 16 * nt + 4 * q;
  f4 v;
  if constexpr (VEC) {
    const f4u t = *reinterpret_cast<const f4u*>(wr + imin_(k0, K - 4));
    v = f4{t[0], t[1], t[2], t[3]};
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = wr[imin_(k0 + r, K - 1)];
  }
  return v;
}

// The A fragments of eight input tiles nt0 .. nt0 + 7 of a weight row (clamped beyond KT: their B operands are zeroed) -- ONE
// memory round trip for a 128-wide layer; the caller issues them a stage ahead where it can.
// (VEC is a property of the LAUNCH -- every layer's input width a multiple of 4 -- not a run-time branch: with both forms in one
// body the compiler's wait-count bookkeeping fenced each load of one form behind the registers of the other, a serial round trip
// per fragment, measured)
template <bool VEC>
__device__ __forceinline__ void load_a8(const float* __restrict__ wr, int K, int KT, int nt0, int q, f4 (&av)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) av[i] = load_w_frag<VEC>(wr, K, imin_(nt0 + i, KT - 1), q);
}
__device__ __forceinline__ void mma8(const f4 (&av)[8], const float* hin, int KT, int nt0, f4& acc0, f4& acc1) {
  f4 bv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f4 t = *reinterpret_cast<const f4*>(hin + 16 * imin_(nt0 + i, KT - 1));
    bv[i] = nt0 + i < KT ? t : f4{0.f, 0.f, 0.f, 0.f};
  }
  // (tile pairs beyond the layer's width are skipped -- a wave-uniform branch per eight MFMAs: a 64-wide input is half a batch)
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    if (i == 0 || nt0 + i < KT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = mfma4(av[i][r], bv[i][r], acc0);
        acc1 = mfma4(av[i + 1][r], bv[i + 1][r], acc1);
      }
    }
  }
}
// the same for dH = dZ W: A[in-unit kcol][u] = W[u][kcol] for the eight unit tiles ut0 .. ut0 + 7 (32 dword loads, one round trip)
__device__ __forceinline__ void load_t8(const float* __restrict__ W, int K, int N, int NT, int ut0, int kcol, int q, float (&av)[8][4]) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) av[i][r] = W[(int64_t)imin_(16 * imin_(ut0 + i, NT - 1) + 4 * q + r, N - 1) * K + kcol];
}
__device__ __forceinline__ void mma_t8(const float (&av)[8][4], const float* zb, int NT, int ut0, f4& acc0, f4& acc1) {
  f4 bv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f4 t = *reinterpret_cast<const f4*>(zb + 16 * imin_(ut0 + i, NT - 1));
    bv[i] = ut0 + i < NT ? t : f4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    if (i == 0 || ut0 + i < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = mfma4(av[i][r], bv[i][r], acc0);
        acc1 = mfma4(av[i + 1][r], bv[i + 1][r], acc1);
      }
    }
  }
}

#ifdef SPO_MR_PROF
// development aid (tools/build_variant.py mrprof -DSPO_MR_PROF, SPO_VARIANT_SOURCES=mlp_rows.hip): wall-clock stamps (100 MHz) of
// the first lane of workgroups 0 and gridDim - 1 at the stage boundaries
__device__ unsigned long long g_mr_prof[2][32];
#define MR_STAMP(k) do { const int k_ = (k); if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && k_ < 32) \
  g_mr_prof[blockIdx.x == 0 ? 0 : 1][k_] = wall_clock64(); } while (0)
#else
#define MR_STAMP(k) do { } while (0)
#endif

template <bool VEC>
__global__ __launch_bounds__(512) void mlp_rows_grad_kernel(MrArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = blockIdx.x / a.R, rg = blockIdx.x - net * a.R;
  const MrNet& nn = a.net[net];
  const int n = nn.n, kind = nn.kind;
  const int row0 = 16 * rg, nrows = imin_(16, a.rows - row0);
  const int PZ = nn.pz, ZS = PZ + 4;
  const int zbase = nn.zbase;
  float* const misc = lds + zbase + 2 * 16 * ZS + 2 * PZ * MR_TS;
  long long* const ridx = reinterpret_cast<long long*>(misc + MR_RIDX);   // [16] (the map's pieces are multiples of 4 floats)
  float* const rowa = misc + MR_ROWA;                                   // [16] d(loss)/d(log-prob) (actor) or d(loss)/d(value)
  float* const rowl = misc + MR_ROWL;                                   // [16] the row's loss term
  const int AP = mr_ap(a.A), MR_IVAR = mr_ivar(AP);
  float* const dls = misc + mr_dls(AP);                                 // [act_dim][16] d(log_std) terms
  float* const ract = misc + mr_ract(AP);                               // [16][act_dim] the rows' actions
  float* const part = a.parts + (int64_t)rg * a.stride;
  const int OUT = nn.d[n];
  int stamp = 0;
  MR_STAMP(stamp++);

  // ---- the minibatch's rows of this group (clamped: rows beyond the minibatch repeat its last row and carry a zero dZ)
  if (tid < 16) {
    const int64_t base = a.cursor ? a.cursor[0] : 0;
    const int64_t at = base + row0 + imin_(tid, nrows - 1);
    ridx[tid] = a.idx ? a.idx[at] : at;
  }
  if (kind == 2 && tid >= 64 && tid < 64 + OUT) {
    const float ls = a.theta[a.ls_off + tid - 64], sd = __expf(ls);
    misc[MR_LS + tid - 64] = ls;
    misc[MR_IVAR + tid - 64] = 1.f / (sd * sd);
  }
  // the weights of a wave's FIRST output tile of a layer (its first eight input tiles) and that tile's biases, requested one stage
  // ahead of their use -- before the barrier that publishes their B operand
  f4 pre[8], pbias, nxt[8], nbias;
  auto prefetch_fwd = [&](int l, f4 (&dst)[8], f4& dbias) {
    const int K = nn.d[l], N = nn.d[l + 1], KT = nn.pr[l] >> 4;
    const float* __restrict__ W = a.theta + nn.w[l];
    const float* __restrict__ bias = a.theta + nn.b[l];
    const int mt = imin_(wave, (nn.pr[l + 1] >> 4) - 1);
    // (biases first: behind the weights the compiler put a full vmcnt(0) in front of them -- a round trip on the critical path)
#pragma unroll
    for (int r = 0; r < 4; ++r) dbias[r] = bias[imin_(16 * mt + 4 * q + r, N - 1)];
    load_a8<VEC>(W + (int64_t)imin_(16 * mt + j, N - 1) * K, K, KT, 0, q, dst);
  };
  prefetch_fwd(0, pre, pbias);
  __syncthreads();
  MR_STAMP(stamp++);
  {
    // observations -> both images of level 0; the loss's per-row inputs -> LDS now, so that the loss stage waits for no global load
    const MrLevel L0{nn.pr[0], nn.hr[0], nn.ht[0]};
    const int D = nn.d[0], S0 = L0.pr + 4;
    for (int r = wave; r < 16; r += 8) {
      const float* __restrict__ src = a.obs + ridx[r] * (int64_t)D;
      for (int c = lane; c < L0.pr; c += 64) {
        const float v = (r < nrows && c < D) ? src[imin_(c, D - 1)] : 0.f;
        lds[L0.hr + r * S0 + c] = v;
        lds[L0.ht + c * MR_TS + r] = v;
      }
      if (kind == 2) {
        const float* __restrict__ sa = a.act + ridx[r] * (int64_t)OUT;
        for (int c = lane; c < OUT; c += 64) ract[r * OUT + c] = sa[c];
      }
    }
    if (tid >= 448 && tid < 464) {
      const int64_t gi = ridx[tid - 448];
      misc[MR_RTGT + tid - 448] = kind == 0 ? a.tgt_r[gi] : (kind == 1 ? a.tgt_c[gi] : 0.f);
      if (kind == 2) { misc[MR_RADV + tid - 448] = a.adv[gi]; misc[MR_RLOGP + tid - 448] = a.logp_old[gi]; }
    }
  }
  __syncthreads();
  MR_STAMP(stamp++);

  // ---- forward: h_{l+1} = tanh(W_l h_l + b_l); wave w takes the output tiles w, w + 8, ...
  for (int l = 0; l < n; ++l) {
    const MrLevel Li{nn.pr[l], nn.hr[l], nn.ht[l]}, Lo{nn.pr[l + 1], nn.hr[l + 1], nn.ht[l + 1]};
    const int K = nn.d[l], N = nn.d[l + 1], KT = Li.pr >> 4, NT = Lo.pr >> 4, SO = Lo.pr + 4;
    const float* __restrict__ W = a.theta + nn.w[l];
    const float* __restrict__ bias = a.theta + nn.b[l];
    const bool act = l + 1 < n;
    const float* hin = lds + Li.hr + j * (Li.pr + 4) + 4 * q;
    // the NEXT layer's first tile: requested a whole stage ahead, into a second register set (the one in use is busy; a request
    // behind this layer's MFMAs left the loads ~0.5 us before their use and the next layer waited out the rest of the round trip)
    if (l + 1 < n) prefetch_fwd(l + 1, nxt, nbias);
    for (int mt = wave; mt < NT; mt += 8) {
      const int u0 = 16 * mt + 4 * q;
      const float* __restrict__ wr = W + (int64_t)imin_(16 * mt + j, N - 1) * K;
      f4 acc0, acc1 = f4{0.f, 0.f, 0.f, 0.f};
      if (mt == wave) {
        acc0 = pbias;
        mma8(pre, hin, KT, 0, acc0, acc1);
      } else {
        f4 av[8];
        load_a8<VEC>(wr, K, KT, 0, q, av);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc0[r] = bias[imin_(u0 + r, N - 1)];
        mma8(av, hin, KT, 0, acc0, acc1);
      }
      for (int nt0 = 8; nt0 < KT; nt0 += 8) {
        f4 av[8];
        load_a8<VEC>(wr, K, KT, nt0, q, av);
        mma8(av, hin, KT, nt0, acc0, acc1);
      }
      MR_STAMP(16 + 4 * l);
      f4 v = acc0 + acc1;
      if (act) v = fast_tanh4(v);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = u0 + r < N ? v[r] : 0.f;
      *reinterpret_cast<f4*>(lds + Lo.hr + j * SO + u0) = v;
      if (act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[Lo.ht + (u0 + r) * MR_TS + j] = v[r];
      }
    }
    MR_STAMP(17 + 4 * l);
    if (l + 1 < n) {
#pragma unroll
      for (int i = 0; i < 8; ++i) pre[i] = nxt[i];
      pbias = nbias;
    }
    MR_STAMP(18 + 4 * l);
    __syncthreads();
    MR_STAMP(stamp++);
  }

  // dH = dZ W of layer l: the weights of this wave's first input-unit tile (first eight unit tiles), a stage ahead
  float pt[8][4], npt[8][4];
  auto prefetch_bwd = [&](int l, float (&dst)[8][4]) {
    const int K = nn.d[l], N = nn.d[l + 1], NT = nn.pr[l + 1] >> 4;
    const int mt = imin_(wave, (nn.pr[l] >> 4) - 1);
    load_t8(a.theta + nn.w[l], K, N, NT, 0, imin_(16 * mt + j, K - 1), q, dst);
  };
  if (n > 1) prefetch_bwd(n - 1, pt);
  // ---- loss of this network on the group's rows -> d(loss)/d(output) in the first dZ images (zero beyond the rows / the outputs)
  float* ZR = lds + zbase;
  float* ZT = lds + zbase + 2 * 16 * ZS;
  const MrLevel Ln{nn.pr[n], nn.hr[n], nn.ht[n]};
  const int SN = Ln.pr + 4;
  const float inv_n = 1.f / (float)a.rows;
  if (tid < 16) {
    const int r = tid;
    float da = 0.f, lt = 0.f;
    if (r < nrows) {
      if (kind < 2) {                                    // MSE of a critic (ppo_lag.py:306-309)
        const float diff = lds[Ln.hr + r * SN] - misc[MR_RTGT + r];
        lt = diff * diff;
        da = 2.f * diff * inv_n;
      } else {                                           // clipped surrogate (ppo_lag.py:316-319): wide_ppo_loss_kernel's arithmetic
        const float clip_lo = 1.f - a.clip, clip_hi = 1.f + a.clip;
        float lp = 0.f;
        for (int k = 0; k < OUT; ++k) {
          const float ls = misc[MR_LS + k], ivar = misc[MR_IVAR + k];
          const float dif = ract[r * OUT + k] - lds[Ln.hr + r * SN + k];
          lp += -(dif * dif) * (0.5f * ivar) - ls - LOG_SQRT_2PI_F;
        }
        const float ad = misc[MR_RADV + r];
        const float ratio = __expf(lp - misc[MR_RLOGP + r]);
        const float rc = fminf(fmaxf(ratio, clip_lo), clip_hi);
        const float s1 = ratio * ad, s2 = rc * ad;
        const bool inr = (ratio >= clip_lo) && (ratio <= clip_hi);
        float gr;                                        // backward of torch.min / torch.clamp (ties split the gradient)
        if (s1 < s2) gr = ad;
        else if (s1 > s2) gr = inr ? ad : 0.f;
        else gr = 0.5f * ad + (inr ? 0.5f * ad : 0.f);
        da = -(gr * ratio) * inv_n;
        lt = fminf(s1, s2);
      }
    }
    rowa[r] = da; rowl[r] = lt;
  }
  __syncthreads();
  for (int e = tid; e < 16 * Ln.pr; e += 512) {
    const int r = e & 15, u = e >> 4;
    float dz = 0.f;
    if (u < OUT && r < nrows) {
      if (kind < 2) dz = rowa[r];
      else {
        const float ivar = misc[MR_IVAR + u];
        const float dif = ract[r * OUT + u] - lds[Ln.hr + r * SN + u];
        const float z = dif * ivar;
        dz = rowa[r] * z;
        dls[u * 16 + r] = rowa[r] * (dif * z - 1.f);
      }
    } else if (kind == 2 && u < OUT) dls[u * 16 + r] = 0.f;
    ZR[r * ZS + u] = dz;
    ZT[u * MR_TS + r] = dz;
  }
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int r = 0; r < 16; ++r) s += (double)rowl[r];
    part[a.P + kind] = (float)s;
  }
  if (kind == 2 && tid >= 64 && tid < 64 + OUT) {
    const int u = tid - 64;
    double s = 0.0;
    for (int r = 0; r < 16; ++r) s += (double)dls[u * 16 + r];
    part[a.ls_off + u] = (float)s;
  }

  MR_STAMP(stamp++);
  // ---- backward: db, dW (this group's rows) and dZ of the layer below; one barrier per layer
  int cur = 0;
  for (int l = n - 1; l >= 0; --l) {
    const MrLevel Li{nn.pr[l], nn.hr[l], nn.ht[l]}, Lo{nn.pr[l + 1], nn.hr[l + 1], nn.ht[l + 1]};
    const int K = nn.d[l], N = nn.d[l + 1], KT = Li.pr >> 4, NT = Lo.pr >> 4, SI = Li.pr + 4;
    const float* __restrict__ W = a.theta + nn.w[l];
    const float* ZRc = ZR + cur * 16 * ZS;
    const float* ZTc = ZT + cur * PZ * MR_TS;
    float* ZRn = ZR + (cur ^ 1) * 16 * ZS;
    float* ZTn = ZT + (cur ^ 1) * PZ * MR_TS;
    float* __restrict__ pW = part + nn.w[l];
    float* __restrict__ pb = part + nn.b[l];
    if (l > 1) prefetch_bwd(l - 1, npt);                 // (a whole stage ahead, second register set: see the forward)
    // (1) bias gradient: the column sums of dZ in row order
    for (int u = tid; u < N; u += 512) {
      float s = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f4 t = *reinterpret_cast<const f4*>(ZTc + u * MR_TS + 4 * r4);
        s += t[0]; s += t[1]; s += t[2]; s += t[3];
      }
      pb[u] = s;
    }
    if (l == 1) MR_STAMP(26);
    // (2) weight gradient dW[u][k] = sum_rows dZ[row][u] h[row][k]: tiles (unit tile, input tile) dealt to the waves; the reduction
    //     index is the row -- four MFMAs per tile
    {
      const float* za = ZTc + j * MR_TS + 4 * q;
      const float* hb = lds + Li.ht + j * MR_TS + 4 * q;
      const int ntile = NT * KT;
      // (the tile's coordinates advance without a division -- p / KT with a run-time KT is ~40 scalar instructions per tile -- and
      //  the store offsets are 32-bit on the uniform base.  Four tiles per round with their sixteen MFMAs interleaved, and the
      //  stores left out altogether, measured the same 0.3 us per tile: neither the dependent MFMAs nor the stores are the bound.)
      int mt = wave / KT, nt = wave - mt * KT;
#pragma unroll 2
      for (int p = wave; p < ntile; p += 8, nt += 8) {
        while (nt >= KT) { nt -= KT; ++mt; }
        const f4 av = *reinterpret_cast<const f4*>(za + 16 * mt * MR_TS);
        const f4 bv = *reinterpret_cast<const f4*>(hb + 16 * nt * MR_TS);
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma4(av[r], bv[r], acc);
        const int col = 16 * nt + j, u0 = 16 * mt + 4 * q;
        if (col < K) {
          const unsigned off = (unsigned)(u0 * K + col);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (u0 + r < N) pW[off + (unsigned)(r * K)] = acc[r];
        }
      }
    }
    if (l == 1) MR_STAMP(27);
    // (3) dZ of the layer below: dH = dZ W, dZ' = dH (1 - h^2); wave w takes the input-unit tiles w, w + 8, ...; the first
    //     tile's weights were requested a stage ahead (pt)
    if (l > 0) {
      const float* zb = ZRc + j * ZS + 4 * q;
      for (int mt = wave; mt < KT; mt += 8) {
        const int kcol = imin_(16 * mt + j, K - 1);
        f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        if (mt == wave) mma_t8(pt, zb, NT, 0, acc0, acc1);
        else {
          float av[8][4];
          load_t8(W, K, N, NT, 0, kcol, q, av);
          mma_t8(av, zb, NT, 0, acc0, acc1);
        }
        for (int ut0 = 8; ut0 < NT; ut0 += 8) {
          float av[8][4];
          load_t8(W, K, N, NT, ut0, kcol, q, av);
          mma_t8(av, zb, NT, ut0, acc0, acc1);
        }
        const f4 dh = acc0 + acc1;
        const int u0 = 16 * mt + 4 * q;
        const f4 h = *reinterpret_cast<const f4*>(lds + Li.hr + j * SI + u0);
        f4 dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = u0 + r < K ? dh[r] * fmaf(-h[r], h[r], 1.f) : 0.f;
        *reinterpret_cast<f4*>(ZRn + j * ZS + u0) = dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) ZTn[(u0 + r) * MR_TS + j] = dz[r];
      }
    }
    if (l == 1) MR_STAMP(28);
    if (l > 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) pt[i][r] = npt[i][r];
    }
    if (l == 1) MR_STAMP(29);
    __syncthreads();
    MR_STAMP(stamp++);
    cur ^= 1;
  }
}

// grad[i] = sum over the row groups of parts[g][i] (group order), losses3 = the three data losses of the minibatch
__global__ __launch_bounds__(256) void mlp_rows_reduce_kernel(const float* __restrict__ parts, int R, int64_t stride, int64_t P, int64_t rows,
                                                              int n_loss, float* __restrict__ grad, float* __restrict__ losses) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
    float s = parts[i];
    for (int g = 1; g < R; ++g) s += parts[(int64_t)g * stride + i];
    grad[i] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < n_loss && losses) {
    double s = 0.0;
    for (int g = 0; g < R; ++g) s += (double)parts[(int64_t)g * stride + P + threadIdx.x];
    losses[threadIdx.x] = (float)((threadIdx.x == 2 ? -s : s) / (double)rows);
  }
}

// ---- the optimiser behind the row groups in TWO launches (world size 1, device-resident clocks: the replayed step).
// The launch-per-network step ended in wide_prep_kernel -> wide_coef_kernel -> wide_adam_dev_kernel (csrc/ma_net.hip): with the
// group sum in front, four launches of ~4.5 us that each do < 1 us of work.  Here (1) the group sum, the critics' L2 gradient, the
// value coefficient and the norm partials are one pass over the parameters, whose first workgroup also advances the optimiser
// clocks and the cursor (nobody in that launch reads them), and (2) every workgroup of the Adam pass forms the clip coefficient
// itself from the partials (fixed order: the same value in every workgroup) and reads the advanced clocks; its first workgroup adds
// the L2 terms to the logged losses and stores the loss row.  (A first form kept everything wide_coef_kernel's single lane did in
// the Adam pass behind a device counter -- the last workgroup to arrive wrote the clocks: one same-address atomic per workgroup,
// +20 us a step with 290 workgroups, still +1.5 us with 19 large ones.)  Same arithmetic, element for element, as
// spo_wide_reduce_parts + spo_wide_clip_adam_dev_log.
struct MrOptArgs {
  const float* parts; int R; int64_t stride;
  float* theta; float* grad; float* m; float* v; int64_t P, r_end, c_end, actor_begin; int64_t rows;
  float l2, vcoef_r, max_norm, lr_actor, lr_critic, b1, b2, eps;
  double* partial; float* scal; float* losses3; double* pow4;
  float* loss_log; int64_t* cursor; int64_t cursor_step; int nblocks;
};
__global__ __launch_bounds__(256) void mlp_rows_prep_kernel(MrOptArgs a) {
  __shared__ double red[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double gs = 0.0, pr = 0.0, pc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < a.P; i += (int64_t)gridDim.x * 256) {
    float g = a.parts[i];
    for (int k = 1; k < a.R; ++k) g += a.parts[(int64_t)k * a.stride + i];
    if (i < a.c_end) {
      const float p = a.theta[i];
      g = fmaf(2.f * a.l2, p, g);
      if (i < a.r_end) { g *= a.vcoef_r; pr += (double)(p * p); } else pc += (double)(p * p);
    }
    a.grad[i] = g;
    gs += (double)(g * g);
  }
  gs = wave_sum_d(gs); pr = wave_sum_d(pr); pc = wave_sum_d(pc);
  if (lane == 0) { red[wave][0] = gs; red[wave][1] = pr; red[wave][2] = pc; }
  __syncthreads();
  if (tid < 3) a.partial[(int64_t)blockIdx.x * 3 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  if (blockIdx.x == 0 && tid >= 64 && tid < 67) {
    const int k = tid - 64;
    double s = 0.0;
    for (int g = 0; g < a.R; ++g) s += (double)a.parts[(int64_t)g * a.stride + a.P + k];
    a.losses3[k] = (float)((k == 2 ? -s : s) / (double)a.rows);
  }
  // the optimiser clocks and the cursor advance HERE (one lane): nobody in this launch reads them, the gradient launch that did is
  // over, and the Adam pass behind this one reads the advanced clocks as wide_adam_dev_kernel does -- no counter, no atomics
  if (blockIdx.x == 0 && tid == 128) {
    a.pow4[0] *= (double)a.b1; a.pow4[1] *= (double)a.b2; a.pow4[2] *= (double)a.b1; a.pow4[3] *= (double)a.b2;
    if (a.cursor) a.cursor[0] += a.cursor_step;
  }
}
__global__ __launch_bounds__(256) void mlp_rows_adam_kernel(MrOptArgs a) {
  __shared__ float sh[4];
  const int tid = threadIdx.x;
  if (tid < 64) {
    // wide_coef_kernel's sum: lane l adds the partials l, l + 64, ... in order, then a fixed butterfly over the lanes
    double gs = 0.0, pr = 0.0, pc = 0.0;
    for (int b = tid; b < a.nblocks; b += 64) { gs += a.partial[b * 3]; pr += a.partial[b * 3 + 1]; pc += a.partial[b * 3 + 2]; }
    gs = wave_sum_d(gs); pr = wave_sum_d(pr); pc = wave_sum_d(pc);
    if (tid == 0) {
      const float norm = sqrtf((float)gs);
      const float coef = a.max_norm / (norm + 1e-6f);              // clip_grad_norm_ (torch): eps 1e-6
      sh[0] = coef > 1.f ? 1.f : coef;
      sh[1] = a.l2 * (float)pr; sh[2] = a.l2 * (float)pc; sh[3] = norm;
    }
  }
  __syncthreads();
  const float coef = sh[0];
  const float lr_a = a.pow4[4] >= 0.0 ? (float)a.pow4[4] : a.lr_actor, lr_c = a.pow4[5] >= 0.0 ? (float)a.pow4[5] : a.lr_critic;
  float ss_a, ss_c, bc2s_a, bc2s_c;
  adam_scalars(lr_a, a.pow4[2], a.pow4[3], ss_a, bc2s_a);            // (advanced by the pass in front: beta^(t+1))
  adam_scalars(lr_c, a.pow4[0], a.pow4[1], ss_c, bc2s_c);
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < a.P; i += (int64_t)gridDim.x * 256) {
    const bool act = i >= a.actor_begin;
    const AdamOut o = adam1(a.theta[i], a.grad[i] * coef, a.m[i], a.v[i], a.b1, a.b2, a.eps, act ? ss_a : ss_c, act ? bc2s_a : bc2s_c);
    a.theta[i] = o.p; a.m[i] = o.m; a.v[i] = o.v;
  }
  // what wide_coef_kernel's single lane wrote besides the clocks: nobody else in this launch reads any of it
  if (blockIdx.x == 0 && tid == 0) {
    a.scal[0] = sh[0]; a.scal[1] = sh[1]; a.scal[2] = sh[2]; a.scal[3] = sh[3];
    const float l0 = a.losses3[0] + sh[1], l1 = a.losses3[1] + sh[2], l2v = a.losses3[2];   // logged critic losses include their L2 terms
    a.losses3[0] = l0; a.losses3[1] = l1;
    if (a.cursor && a.loss_log) {
      const int64_t at = a.cursor[0] - a.cursor_step;               // (the step's position: the cursor moved on in the pass in front)
      float* row = a.loss_log + 3 * (at / a.cursor_step);
      row[0] = l0; row[1] = l1; row[2] = l2v;
    }
  }
}

bool mr_enabled() {
  static const bool on = [] { const char* e = getenv("SPO_WIDE_ROWS"); return !(e && e[0] == '0'); }();
  return on;
}
int mr_fill_net(const spo_mlp_net* net, int64_t off, int kind, MrNet* o) {
  o->n = net->n_layers; o->kind = kind;
  for (int k = 0; k <= o->n; ++k) o->d[k] = net->dims[k];
  for (int l = 0; l < o->n; ++l) {
    o->w[l] = off; off += (int64_t)o->d[l + 1] * o->d[l];
    o->b[l] = off; off += o->d[l + 1];
  }
  for (int v = 0; v <= o->n; ++v) {
    const MrLevel L = mr_level(*o, v);
    o->pr[v] = L.pr; o->hr[v] = L.hr; o->ht[v] = L.ht;
  }
  o->pz = mr_pz(*o); o->zbase = mr_images_end(*o);
  return 0;
}
bool mr_net_ok(const spo_mlp_net* net, int A) {
  if (!net || net->n_layers < 1 || net->n_layers > SPO_MLP_MAX_LAYERS) return false;
  for (int k = 0; k <= net->n_layers; ++k)
    if (net->dims[k] < 1) return false;
  MrNet t;
  mr_fill_net(net, 0, 0, &t);
  return (size_t)mr_lds_floats(t, A) * sizeof(float) <= MR_MAX_LDS;
}
int64_t mr_stride(int64_t P) { return (P + 4 + 3) & ~(int64_t)3; }
}  // namespace

#ifdef SPO_MR_PROF
extern "C" int spo_debug_mr_profile(unsigned long long* out64_host) {
  return spo::hip_check(hipMemcpyFromSymbol(out64_host, HIP_SYMBOL(g_mr_prof), sizeof(unsigned long long) * 64), "mr_prof");
}
#endif

extern "C" int spo_wide_grad_rows_supported(const spo_mlp_net* critic, const spo_mlp_net* actor, int64_t rows) {
  if (!mr_enabled() || rows < 1 || rows > 16 * MR_MAX_GROUPS) return 0;
  if (!critic || critic->n_layers < 1 || critic->n_layers > SPO_MLP_MAX_LAYERS || critic->dims[critic->n_layers] != 1) return 0;
  if (actor && (actor->n_layers < 1 || actor->n_layers > SPO_MLP_MAX_LAYERS || actor->dims[0] != critic->dims[0] ||
                actor->dims[actor->n_layers] < 1 || actor->dims[actor->n_layers] > SPO_WIDE_MAX_ACT))
    return 0;
  const int A = actor ? actor->dims[actor->n_layers] : 0;
  if (!mr_net_ok(critic, A) || (actor && !mr_net_ok(actor, A))) return 0;
  return 1;
}

extern "C" int64_t spo_wide_grad_rows_part_floats(int64_t n_params, int64_t rows) {
  if (n_params < 1 || rows < 1) return -1;
  return ((rows + 15) / 16) * mr_stride(n_params) + 4;
}

// n_nets = 3: reward critic, cost critic, actor (theta = [critic | critic | log_std | actor], the ActorVCritic layout of
// safepo/common/wide.py); n_nets = 2: the two critics only (the critic fit of the second-order scripts, cpo.py:541-556).
extern "C" int spo_wide_ppo_grad_rows(const float* theta, const spo_mlp_net* critic, const spo_mlp_net* actor, const float* obs,
                                      const float* act, const float* logp_old, const float* target_r, const float* target_c,
                                      const float* adv, const int64_t* idx, const int64_t* cursor_dev, int64_t rows, float clip,
                                      float* parts, void* stream) {
  SPO_REQUIRE(theta && critic && obs && target_r && target_c && parts, "wide_ppo_grad_rows: null pointer");
  SPO_REQUIRE(!actor || (act && logp_old && adv), "wide_ppo_grad_rows: the actor needs act / logp_old / adv");
  SPO_REQUIRE(spo_wide_grad_rows_supported(critic, actor, rows), "wide_ppo_grad_rows: shape outside the row-group kernel (rows %lld)",
              (long long)rows);
  MrArgs a{};
  a.theta = theta; a.obs = obs; a.act = act; a.logp_old = logp_old; a.tgt_r = target_r; a.tgt_c = target_c; a.adv = adv;
  a.idx = idx; a.cursor = cursor_dev; a.rows = (int)rows; a.R = (int)((rows + 15) / 16); a.clip = clip;
  const int64_t Pc = spo_mlp_param_count(critic);
  mr_fill_net(critic, 0, 0, &a.net[0]);
  mr_fill_net(critic, Pc, 1, &a.net[1]);
  a.n_nets = 2; a.P = 2 * Pc; a.A = 0; a.ls_off = 2 * Pc;
  const int A_launch = actor ? actor->dims[actor->n_layers] : 0;
  size_t lds_floats = (size_t)mr_lds_floats(a.net[0], A_launch);
  if (actor) {
    a.A = actor->dims[actor->n_layers];
    mr_fill_net(actor, 2 * Pc + a.A, 2, &a.net[2]);
    a.n_nets = 3; a.P = 2 * Pc + a.A + spo_mlp_param_count(actor);
    const size_t la = (size_t)mr_lds_floats(a.net[2], A_launch);
    lds_floats = la > lds_floats ? la : lds_floats;
  }
  a.parts = parts; a.stride = mr_stride(a.P);
  const size_t sh = lds_floats * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  bool vec = true;                                   // every weight row a whole number of 16-byte fragments
  for (int k = 0; k < a.n_nets; ++k)
    for (int l = 0; l < a.net[k].n; ++l) vec = vec && (a.net[k].d[l] & 3) == 0;
  static bool done_dev[SPO_MAX_DEVICES] = {};
  bool& done = done_dev[spo::current_device_slot()];
  if (!done) {
    for (const void* f : {reinterpret_cast<const void*>(&mlp_rows_grad_kernel<true>), reinterpret_cast<const void*>(&mlp_rows_grad_kernel<false>)})
      if (int rc = spo::hip_check(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MR_MAX_LDS),
                                  "hipFuncSetAttribute(mlp_rows)"))
        return rc;
    done = true;
  }
  if (vec) hipLaunchKernelGGL(mlp_rows_grad_kernel<true>, dim3(a.n_nets * a.R), dim3(512), sh, st, a);
  else hipLaunchKernelGGL(mlp_rows_grad_kernel<false>, dim3(a.n_nets * a.R), dim3(512), sh, st, a);
  SPO_LAUNCH_CHECK("spo_wide_ppo_grad_rows");
  return 0;
}

// grad[0 .. n_params) = the sum of the row groups' partial gradients, losses_out[0 .. n_losses) = the data losses
// {MSE reward critic, MSE cost critic, clipped surrogate} of the `rows`-row minibatch (n_losses 2: the critics only).
extern "C" int spo_wide_reduce_parts(const float* parts, int64_t rows, int64_t n_params, int n_losses, float* grad, float* losses_out,
                                     void* stream) {
  SPO_REQUIRE(parts && grad && rows >= 1 && rows <= 16 * MR_MAX_GROUPS && n_params >= 1 && n_losses >= 0 && n_losses <= 3,
              "wide_reduce_parts: bad args");
  const int R = (int)((rows + 15) / 16);
  int64_t blocks = (n_params + 255) / 256;
  blocks = blocks > 512 ? 512 : blocks;
  hipLaunchKernelGGL(mlp_rows_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, parts, R, mr_stride(n_params),
                     n_params, rows, n_losses, grad, losses_out);
  SPO_LAUNCH_CHECK("spo_wide_reduce_parts");
  return 0;
}

// spo_wide_reduce_parts + spo_wide_clip_adam_dev_log (full ranges: the PPO-Lagrangian step, ppo_lag.py:310-329) in two launches --
// see mlp_rows_prep_kernel / mlp_rows_adam_kernel.  parts: the buffer spo_wide_ppo_grad_rows filled.  grad receives the clipped step's
// pre-clip gradient (with the L2 terms), losses3_out the step's losses, pow4_dev / cursor_dev advance as in
// spo_wide_clip_adam_dev_log (loss_log_dev row = the three losses).
extern "C" int spo_wide_rows_clip_adam_dev_log(float* parts, int64_t rows, float* theta, float* grad, float* adam_m, float* adam_v,
                                               int64_t n_params, int64_t reward_critic_end, int64_t cost_critic_end,
                                               int64_t actor_begin, const spo_ppo_cfg* cfg, double* pow4_dev, float* losses3_out,
                                               float* scalars4_out, double* partial_ws, int partial_capacity, float* loss_log_dev,
                                               int64_t* cursor_dev, int64_t cursor_step, void* stream) {
  SPO_REQUIRE(parts && theta && grad && adam_m && adam_v && cfg && pow4_dev && losses3_out && scalars4_out && partial_ws && n_params > 0 &&
                  rows >= 1 && rows <= 16 * MR_MAX_GROUPS, "wide_rows_clip_adam: bad args");
  SPO_REQUIRE(0 <= reward_critic_end && reward_critic_end <= cost_critic_end && cost_critic_end <= actor_begin && actor_begin <= n_params,
              "wide_rows_clip_adam: parameter ranges out of order");
  SPO_REQUIRE(!cursor_dev || cursor_step > 0, "wide_rows_clip_adam: cursor_step must be > 0");
  int64_t blocks = (n_params + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  SPO_REQUIRE((int64_t)partial_capacity >= blocks * 3, "wide_rows_clip_adam: partial workspace too small");
  const int R = (int)((rows + 15) / 16);
  const int64_t stride = mr_stride(n_params);
  MrOptArgs a{parts, R, stride, theta, grad, adam_m, adam_v, n_params, reward_critic_end, cost_critic_end, actor_begin, rows,
              cfg->use_critic_norm ? cfg->l2_coef : 0.f, cfg->use_value_coefficient ? 2.f : 1.f, cfg->max_grad_norm, cfg->lr_actor,
              cfg->lr_critic, cfg->beta1, cfg->beta2, cfg->adam_eps, partial_ws, scalars4_out, losses3_out, pow4_dev, loss_log_dev,
              cursor_dev, cursor_step, (int)blocks};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mlp_rows_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(mlp_rows_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  SPO_LAUNCH_CHECK("spo_wide_rows_clip_adam_dev_log");
  return 0;
}
