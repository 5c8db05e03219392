// Small-batch forward / backward of a wide-path network in ONE launch each (round 4).  gfx950 only.
//
// The wide-network path (csrc/ma_net.hip spo_mlp_forward / spo_mlp_backward; reference safepo/common/model.py:30-48 build_mlp_network,
// any hidden_sizes) runs a layer as a GEMM launch plus small kernels: right for the 8 192-row minibatches of
// isaac_gym_specific_cfg (ppo_lag.py:54-65), but at the reference's default batch of 64 a network's forward is 5 launches and
// its backward 18 -- a minibatch step of three networks ~70 launches of ~5 us, none of which has 5 us of work (0.37 ms per step,
// and replaying them from a HIP graph changes nothing: the cost is per launch on the device, not on the host).  Here one
// workgroup carries a whole network for up to 128 rows:
//   * forward: wave w owns rows 16w .. 16w+15 for ALL layers.  With v_mfma_f32_16x16x4_f32 (M = output units, N = batch rows,
//     K = inputs) the accumulator of lane (j, q) holds units 16mt + 4q .. + 3 of row j -- exactly the B operand the next
//     layer needs from that lane -- so activations never cross lanes: they are parked in an LDS row image only because their
//     number is a run-time quantity, and no barrier is needed between layers.  Weights are read from global memory (L2 / L1:
//     they change every optimiser step and are shared by the waves).
//   * backward: dZ of the current layer lives in a row-major LDS image; db = its column sums (fixed order), dW = dZ^T H over
//     ALL rows (tiles dealt to the waves; K = rows), dZ of the layer below = (dZ W) (1 - h^2) for the wave's own rows.  One
//     barrier per layer.
// Results differ from the GEMM path in summation order only (fp32, fixed order: deterministic); the callers' parity tests
// against the oracle are unchanged.
#include "common.h"
#include "mlp_mfma.h"
#include <cstdlib>
#include "mlp_small.h"

namespace {
using namespace spo;

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int up16(int v) { return (v + 15) & ~15; }

// Cooperative copy of a row-major global matrix block src[r0 + r][c0 + c] (leading dimension ld, `nr` x `nc` valid) into an LDS
// block dst[r * ldd + c] of RR x CC (CC a multiple of 16), zero outside the valid part.  A wave takes rows r = wave, wave +
// nwaves, ...; its lanes take columns lane, lane + 64, ... (256 contiguous bytes per load, any alignment); the loads of sixteen
// rows x two column blocks are in flight before the first LDS store.
__device__ __forceinline__ void stage_block(const float* __restrict__ src, int64_t ld, int nr, int nc, float* dst, int ldd, int RR,
                                            int CC, int wave, int lane, int nwaves) {
  // two column blocks (lane, lane + 64) x sixteen rows per round trip: up to 32 loads in flight per lane
  for (int c = lane; c < CC; c += 128) {
    const int c1 = c + 64;
    const bool cin0 = c < nc, cin1 = c1 < nc, has1 = c1 < CC;
    const bool blk1 = (c - lane) + 64 < CC;              // wave-uniform: the second column block exists at all
    for (int r0 = wave; r0 < RR; r0 += 16 * nwaves) {
      float v0[16], v1[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) v0[b] = src[(int64_t)imin(r0 + b * nwaves, nr - 1) * ld + imin(c, nc - 1)];
      if (blk1) {
#pragma unroll
        for (int b = 0; b < 16; ++b) v1[b] = src[(int64_t)imin(r0 + b * nwaves, nr - 1) * ld + imin(c1, nc - 1)];
      }
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const int r = r0 + b * nwaves;
        if (r < RR) {
          dst[r * ldd + c] = (cin0 && r < nr) ? v0[b] : 0.f;
          if (has1) dst[r * ldd + c1] = (cin1 && r < nr) ? v1[b] : 0.f;
        }
      }
    }
  }
}

// x[row][16 nt + 4q .. + 3] of a global row with K valid columns: the loads only (clamped addresses).  The zeroing of columns
// >= K is applied when the fragment is consumed (mask_x_frag): a select right behind a load makes the wave wait for that load on
// the spot, which turns a prefetch into an exposed round trip.
__device__ __forceinline__ f4 load_x_frag(const float* __restrict__ xrow, int K, int nt, int q, bool xvec) {
  const int k0 = 16 * nt + 4 * q;
  f4 v;
  if (xvec) {                                            // K % 4 == 0 and 16-byte aligned rows
    v = *reinterpret_cast<const f4*>(xrow + imin(k0, K - 4));
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = xrow[imin(k0 + r, K - 1)];
  }
  return v;
}
__device__ __forceinline__ f4 mask_x_frag(f4 v, int K, int nt, int q) {
  const int k0 = 16 * nt + 4 * q;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = k0 + r < K ? v[r] : 0.f;
  return v;
}

// One group of NI output tiles of a forward layer for this wave's 16 rows: acc[i] += W_tile(mt0 + i) . in over all K tiles.
// arow: the lane's A-fragment base in the staged weight chunk (row 16 mt0 + j, column 4q); brow: the lane's B-fragment base in the
// row image (FROM_IMAGE) or the lane's input row in global memory.  NI is a template parameter: predicated MFMAs in an unrolled
// loop make the compiler copy every accumulator between them.
template <int NI, bool FROM_IMAGE>
__device__ __forceinline__ void fwd_group(const float* arow, int KP, int KT, const float* brow, int K, int q, bool xvec, f4 (&acc)[NI]) {
  // the input rows come from global memory two tiles ahead (image reads: one ahead)
  f4 nb = FROM_IMAGE ? *reinterpret_cast<const f4*>(brow) : load_x_frag(brow, K, 0, q, xvec);
  f4 nb2 = FROM_IMAGE ? nb : load_x_frag(brow, K, imin(1, KT - 1), q, xvec);
#pragma unroll 2
  for (int nt = 0; nt < KT; ++nt) {
    const f4 bv = FROM_IMAGE ? nb : mask_x_frag(nb, K, nt, q);
    if (FROM_IMAGE) { if (nt + 1 < KT) nb = *reinterpret_cast<const f4*>(brow + 16 * (nt + 1)); }
    else { nb = nb2; nb2 = load_x_frag(brow, K, imin(nt + 2, KT - 1), q, xvec); }
    f4 av[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) av[i] = *reinterpret_cast<const f4*>(arow + i * 16 * KP + 16 * nt);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = mfma4(av[i][r], bv[r], acc[i]);
  }
}

// the same for the input gradient: acc[i] += W^T tile (in-unit tile mt0 + i) . dZ over all N tiles; wcol: the lane's base in the
// staged W (row 4q, column 16 mt0 + j), zrow: the lane's dZ fragments in the row image
template <int NI>
__device__ __forceinline__ void dh_group(const float* wcol, int KP, int NT, const float* zrow, f4 (&acc)[NI]) {
#pragma unroll 2
  for (int kt = 0; kt < NT; ++kt) {
    const f4 bv = *reinterpret_cast<const f4*>(zrow + 16 * kt);
    float av[NI][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NI; ++i) av[i][r] = wcol[(16 * kt + r) * KP + 16 * i];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = mfma4(av[i][r], bv[r], acc[i]);
  }
}

// weight-gradient tiles (mt0 .. mt0 + NI - 1, nt): acc[i] = sum over the RT row tiles of dZ^T tile . h tile.  zcol: the lane's base in
// the dZ image (row 4q, column 16 mt0 + j); hcol: its base in the input block (row 4q, column 16 nt + j).  NI independent
// accumulator chains share one B fragment.
template <int NI>
__device__ __forceinline__ void dw_group(const float* zcol, int LDM, const float* hcol, int KCP, int RT, f4 (&acc)[NI]) {
#pragma unroll 2
  for (int kt = 0; kt < RT; ++kt) {
    float bv[4], av[NI][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bv[r] = hcol[(16 * kt + r) * KCP];
#pragma unroll
      for (int i = 0; i < NI; ++i) av[i][r] = zcol[(16 * kt + r) * LDM + 16 * i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] = mfma4(av[i][r], bv[r], acc[i]);
  }
}

#ifdef SPO_SMALL_PROF
// development aid (tools/build_variant.py ... -DSPO_SMALL_PROF): wall-clock stamps (100 MHz) of workgroup 0's first lane
__device__ unsigned long long g_small_prof[2][64];
#define SPO_SMALL_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (k) < 64) g_small_prof[BACKWARD ? 1 : 0][k] = wall_clock64(); } while (0)
#else
#define SPO_SMALL_STAMP(k) do { } while (0)
#endif

template <bool BACKWARD>
__global__ __launch_bounds__(512) void mlp_small_kernel(MlpSmallBatch B) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MlpSmallArgs& a = B.a[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  // 8 waves whatever the row count: RT row tiles (16 rows each) x CS column sets -- the waves of a column set share the rows and
  // deal the output-tile groups among the sets, so that a SIMD holds two waves (one wave per SIMD exposes every LDS and global
  // latency: 3 x the MFMA time in the first build) and the staging copies run on 512 lanes
  const int nwaves = blockDim.x >> 6, rows = a.rows, RT = (rows + 15) >> 4, RP = 16 * RT, LDM = a.ldm;
  int RTp = 1;
  while (RTp < RT) RTp <<= 1;
  const int CS = nwaves / RTp, rw = wave % RTp, cs = wave / RTp;
  const bool rowwave = rw < RT;
  float* cur = lds;
  float* nxt = lds + RP * LDM;
  float* big = lds + 2 * RP * LDM;                    // weight chunk / input-column chunk
  const int big_floats = a.big_floats;
  const int rl = rowwave ? 16 * rw + j : 0;
  const bool rvalid = rowwave && rl < rows;
  const int rcl = rvalid ? rl : rows - 1;
  int stamp = 0;
  SPO_SMALL_STAMP(stamp++);
  if constexpr (!BACKWARD) {
    for (int l = 0; l < a.n; ++l) {
      const int K = a.d[l], N = a.d[l + 1], KT = (K + 15) >> 4, NT = (N + 15) >> 4, KP = 16 * KT + 4;
      const float* __restrict__ W = a.theta + a.w[l];
      const float* __restrict__ bias = a.theta + a.b[l];
      float* __restrict__ out = a.ws + a.act[l];
      const float* __restrict__ xin = a.x + (int64_t)rcl * K;                  // layer 0: the input rows, straight from global
      const bool xvec = (K & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
      const bool act = l + 1 < a.n;
      const int UC = imin((big_floats / KP) & ~15, 16 * NT);                   // units per weight chunk (whole tiles)
      for (int u0 = 0; u0 < 16 * NT; u0 += UC) {
        const int uc = imin(UC, 16 * NT - u0);
        // this wave's first group of the chunk: its biases are requested before the staging copy (one memory round trip, not two)
        const int tiles = uc >> 4, gs = imin(4, (tiles + CS - 1) / CS);          // tiles per group: every column set gets work
        f4 bias0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) bias0[i][r] = bias[imin(u0 + 16 * (cs * gs + i) + 4 * q + r, N - 1)];
        __syncthreads();                                                       // the previous chunk's readers are done
        stage_block(W + (int64_t)u0 * K, K, N - u0, K, big, KP, uc, 16 * KT, wave, lane, nwaves);
        __syncthreads();
        SPO_SMALL_STAMP(stamp++);
        for (int mt0 = cs * gs; rowwave && mt0 < tiles; mt0 += CS * gs) {
          const int ni = imin(gs, tiles - mt0);                                // tiles of this group (wave-uniform)
          f4 acc[4];
          if (mt0 == cs * gs) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = bias0[i];
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[i][r] = bias[imin(u0 + 16 * (mt0 + i) + 4 * q + r, N - 1)];
          }
          const float* arow = big + (16 * mt0 + j) * KP + 4 * q;
          const float* brow = l > 0 ? cur + rl * LDM + 4 * q : xin;
#define SPO_FWD_GROUP(NI_)                                                                                            \
  {                                                                                                                    \
    f4 ac[NI_];                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NI_; ++i) ac[i] = acc[i];                                                    \
    if (l > 0) fwd_group<NI_, true>(arow, KP, KT, brow, K, q, xvec, ac);                                               \
    else fwd_group<NI_, false>(arow, KP, KT, brow, K, q, xvec, ac);                                                    \
    _Pragma("unroll") for (int i = 0; i < NI_; ++i) acc[i] = ac[i];                                                    \
  }
          if (ni == 4) SPO_FWD_GROUP(4) else if (ni == 3) SPO_FWD_GROUP(3) else if (ni == 2) SPO_FWD_GROUP(2) else SPO_FWD_GROUP(1)
#undef SPO_FWD_GROUP
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < ni) {
              const int uu = u0 + 16 * (mt0 + i) + 4 * q;
              f4 v = act ? fast_tanh4(acc[i]) : acc[i];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = uu + r < N ? v[r] : 0.f;
              *reinterpret_cast<f4*>(nxt + rl * LDM + uu) = v;
              if (rvalid) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (uu + r < N) out[(int64_t)rl * N + uu + r] = v[r];
              }
            }
          }
        }
      }
      SPO_SMALL_STAMP(stamp++);
      float* t = cur; cur = nxt; nxt = t;
    }
  } else {
    // d(loss)/d(output) -> the row image, zero-padded to whole tiles and to RP rows
    {
      const int N = a.d[a.n];
      stage_block(a.d_out, N, rows, N, cur, LDM, RP, up16(N), wave, lane, nwaves);
    }
    for (int l = a.n - 1; l >= 0; --l) {
      const int K = a.d[l], N = a.d[l + 1], KT = (K + 15) >> 4, NT = (N + 15) >> 4, KP = 16 * KT + 4;
      const float* __restrict__ W = a.theta + a.w[l];
      const float* __restrict__ hin = l == 0 ? a.x : a.ws + a.act[l - 1];      // [rows][K] row-major
      float* __restrict__ gW = a.grad + a.w[l];
      float* __restrict__ gb = a.grad + a.b[l];
      // inputs of this layer: a hidden layer's h_{l-1} fits the second row image (and its weights the big buffer); the
      // network input x may be wider than an image and goes through the big buffer in column chunks
      const int KC = l > 0 ? 16 * KT : imin(((big_floats / RP) - 4) & ~15, 16 * KT);     // columns per chunk
      const int KCP = l > 0 ? LDM : KC + 4;
      float* hbuf = l > 0 ? nxt : big;
      for (int c0 = 0; c0 < 16 * KT; c0 += KC) {
        const int kc = imin(KC, 16 * KT - c0);
        __syncthreads();
        stage_block(hin + c0, K, rows, K - c0, hbuf, KCP, RP, kc, wave, lane, nwaves);
        if (l > 0) stage_block(W, K, N, K, big, KP, 16 * NT, 16 * KT, wave, lane, nwaves);
        __syncthreads();
        SPO_SMALL_STAMP(stamp++);
        // (1) bias gradient: column sums in row order
        if (c0 == 0) {
          for (int u = tid; u < N; u += blockDim.x) {
            float s = 0.f;
            for (int r = 0; r < rows; ++r) s += cur[r * LDM + u];
            gb[u] = s;
          }
        }
        // (2) weight gradient dW[u][k] = sum_rows dZ[row][u] h[row][k]: tile (mt, nt) on wave nt mod nwaves; A = dZ^T from the
        //     row image (kept in registers over the nt loop), B = h from the staged block
        {
          const int ntl = kc >> 4, ngr = (NT + 3) >> 2;                        // (group of <= 4 unit tiles, column tile) pairs, dealt to the waves
          for (int p = wave; p < ngr * ntl; p += nwaves) {
            const int g = p / ntl, nt = p - g * ntl, mt0 = 4 * g, ni = imin(4, NT - mt0);
            const float* zcol = cur + (4 * q) * LDM + 16 * mt0 + j;
            const float* hcol = hbuf + (4 * q) * KCP + 16 * nt + j;
            f4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
#define SPO_DW_GROUP(NI_)                                                                                             \
  {                                                                                                                    \
    f4 ac[NI_];                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NI_; ++i) ac[i] = acc[i];                                                    \
    dw_group<NI_>(zcol, LDM, hcol, KCP, RT, ac);                                                                       \
    _Pragma("unroll") for (int i = 0; i < NI_; ++i) acc[i] = ac[i];                                                    \
  }
            if (ni == 4) SPO_DW_GROUP(4) else if (ni == 3) SPO_DW_GROUP(3) else if (ni == 2) SPO_DW_GROUP(2) else SPO_DW_GROUP(1)
#undef SPO_DW_GROUP
            const int col = c0 + 16 * nt + j;
            if (col < K) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (i < ni) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const int u = 16 * (mt0 + i) + 4 * q + r;
                    if (u < N) gW[(int64_t)u * K + col] = acc[i][r];
                  }
                }
              }
            }
          }
        }
      }
      // (3) dZ of the layer below for this wave's rows: dH = dZ W, dZ' = dH (1 - h^2); h sits in the second image, which the
      //     result overwrites lane by lane -- after every wave is done reading it as the B operand of (2)
      SPO_SMALL_STAMP(stamp++);
      if (l > 0) {
        __syncthreads();
        const int gs = imin(4, (KT + CS - 1) / CS);
        for (int mt0 = cs * gs; rowwave && mt0 < KT; mt0 += CS * gs) {
          const int ni = imin(gs, KT - mt0);
          f4 acc[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
          const float* wcol = big + (4 * q) * KP + 16 * mt0 + j;
          const float* zrow = cur + rl * LDM + 4 * q;
#define SPO_DH_GROUP(NI_)                                                                                             \
  {                                                                                                                    \
    f4 ac[NI_];                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NI_; ++i) ac[i] = acc[i];                                                    \
    dh_group<NI_>(wcol, KP, NT, zrow, ac);                                                                             \
    _Pragma("unroll") for (int i = 0; i < NI_; ++i) acc[i] = ac[i];                                                    \
  }
          if (ni == 4) SPO_DH_GROUP(4) else if (ni == 3) SPO_DH_GROUP(3) else if (ni == 2) SPO_DH_GROUP(2) else SPO_DH_GROUP(1)
#undef SPO_DH_GROUP
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < ni) {
              const int u0 = 16 * (mt0 + i) + 4 * q;
              const f4 h = *reinterpret_cast<const f4*>(nxt + rl * LDM + u0);
              f4 dz;
#pragma unroll
              for (int r = 0; r < 4; ++r) dz[r] = (u0 + r < K && rvalid) ? acc[i][r] * fmaf(-h[r], h[r], 1.f) : 0.f;
              *reinterpret_cast<f4*>(nxt + rl * LDM + u0) = dz;
            }
          }
        }
      }
      SPO_SMALL_STAMP(stamp++);
      float* t = cur; cur = nxt; nxt = t;
    }
  }
}

int small_ldm(const spo_mlp_net* net) {
  int md = 0;
  for (int l = 1; l <= net->n_layers; ++l) md = net->dims[l] > md ? net->dims[l] : md;
  return ((md + 15) / 16) * 16 + 4;
}
// floats left for the weight / input chunk after the two row images; < 0: does not fit
int64_t small_big_floats(const spo_mlp_net* net, int64_t rows) {
  const int64_t RP = ((rows + 15) / 16) * 16;
  return (int64_t)(MLP_SMALL_MAX_LDS / sizeof(float)) - 2 * RP * small_ldm(net);
}

int small_fill(const float* theta, const spo_mlp_net* net, const float* x, int64_t rows, float* ws, const float* d_out, float* grad,
               MlpSmallArgs* o) {
  o->theta = theta; o->x = x; o->ws = ws; o->d_out = d_out; o->grad = grad;
  o->n = net->n_layers; o->rows = (int)rows;
  int64_t off = 0, aoff = 0;
  for (int k = 0; k <= o->n; ++k) o->d[k] = net->dims[k];
  for (int l = 0; l < o->n; ++l) {
    o->w[l] = off; off += (int64_t)o->d[l + 1] * o->d[l];
    o->b[l] = off; off += o->d[l + 1];
    o->act[l] = aoff; aoff += rows * o->d[l + 1];
  }
  o->ldm = small_ldm(net);
  o->big_floats = (int)small_big_floats(net, rows);
  return 0;
}

}  // namespace

namespace spo {

bool mlp_small_ok(const spo_mlp_net* net, int64_t rows) {
  static const bool on = [] { const char* e = getenv("SPO_MLP_SMALL"); return !(e && e[0] == '0'); }();
  if (!on || !net || rows < 1 || rows > MLP_SMALL_MAX_ROWS || net->n_layers < 1 || net->n_layers > SPO_MLP_MAX_LAYERS) return false;
  const int64_t big = small_big_floats(net, rows), RP = ((rows + 15) / 16) * 16;
  auto kp = [&](int l) { return (int64_t)((net->dims[l] + 15) / 16) * 16 + 4; };
  // forward: at least one 16-unit weight chunk of every layer; backward: a hidden layer's whole weight matrix (dZ W needs all
  // of it), and at least one 16-column chunk of the input rows
  for (int l = 0; l < net->n_layers; ++l) {
    if (16 * kp(l) > big) return false;
    if (l > 0 && (int64_t)((net->dims[l + 1] + 15) / 16) * 16 * kp(l) > big) return false;
  }
  return RP * (16 + 4) <= big;
}

int mlp_small_launch(bool backward, const MlpSmallBatch& batch, hipStream_t st) {
  const int rows = batch.a[0].rows;
  for (int i = 0; i < batch.count; ++i)
    if (batch.a[i].rows != rows) return fail(-2, "mlp_small: the networks of one launch take the same number of rows");
  static bool done_dev[2][SPO_MAX_DEVICES] = {};
  bool& done = done_dev[backward ? 1 : 0][current_device_slot()];
  if (!done) {
    const void* f = backward ? reinterpret_cast<const void*>(&mlp_small_kernel<true>) : reinterpret_cast<const void*>(&mlp_small_kernel<false>);
    if (int rc = hip_check(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MLP_SMALL_MAX_LDS), "hipFuncSetAttribute(mlp_small)"))
      return rc;
    done = true;
  }
  const unsigned threads = 512;
  if (backward) hipLaunchKernelGGL(mlp_small_kernel<true>, dim3(batch.count), dim3(threads), MLP_SMALL_MAX_LDS, st, batch);
  else hipLaunchKernelGGL(mlp_small_kernel<false>, dim3(batch.count), dim3(threads), MLP_SMALL_MAX_LDS, st, batch);
  return 0;
}

#ifdef SPO_SMALL_PROF
extern "C" int spo_debug_mlp_small_prof(unsigned long long* out128) {
  return hip_check(hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_small_prof), sizeof(unsigned long long) * 128), "small_prof");
}
#endif

int mlp_small_args(const float* theta, const spo_mlp_net* net, const float* x, int64_t rows, float* ws, const float* d_out, float* grad,
                   MlpSmallArgs* out) {
  return small_fill(theta, net, x, rows, ws, d_out, grad, out);
}

}  // namespace spo
