// Multi-agent (MAPPO-L) networks and update -- SURVEY.md 8 f3.
// Reference: safepo/common/model.py:172-363 (MultiAgentActor / MultiAgentCritic), safepo/utils/mlp.py (MLPBase:
// LayerNorm(obs) -> [Linear -> ELU -> LayerNorm] x (1 + layer_N)), safepo/utils/distributions.py (DiagGaussian with
// std = sigmoid(log_std / x_coef) * y_coef), safepo/multi_agent/mappolag.py:115-234 (MAPPO_L_Trainer), popart.py.
//
// Regime: unlike the single-agent path (327 680 tiny sequential steps), MAPPO-L makes `learning_iters` (5) FULL-BATCH
// steps per agent per epoch over episode_length x n_rollout_threads rows with hidden 128..512: plain large GEMMs.
// Every product runs on in-tree fp32 MFMA kernels (fused block forward / backward, the split-row weight gradient,
// gemm_mfma_kernel for the plain X*W^T and dY*W products, head_small_kernel for narrow heads); rocBLAS is only a test
// comparator behind spo_debug_ma_gemm (dlopen'ed on that call, never by the product path).  Fused around the products:
// bias + ELU + LayerNorm forward, its backward with the three column reductions, the
// Gaussian head / clipped HAPPO surrogate / entropy / lambda-delta epilogue, the clipped Huber value loss on
// PopArt-normalised targets, PopArt statistics, per-network clip_grad_norm_ + Adam.
//
// One flat fp32 parameter vector per network in the reference's state_dict order:
//   feature_norm.{weight,bias}[in], then per block k: W_k[H, in_k], b_k[H], ln_k.weight[H], ln_k.bias[H],
//   then (actor only) log_std[out], then head W[out, H], head b[out].
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include "common.h"
#include "adam.h"
#include "mlp_mfma.h"
#include "mlp_small.h"
#include "../../include/safepo_hip.h"

namespace {

using spo::fail;

// ---------------------------------------------------------------- rocBLAS through dlopen (test comparator only)
typedef void* rb_handle;
typedef int (*rb_create_t)(rb_handle*);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_sgemm_t)(rb_handle, int, int, int, int, int, const float*, const float*, int, const float*, int,
                          const float*, float*, int);
constexpr int RB_N = 111, RB_T = 112;       // rocblas_operation_none / rocblas_operation_transpose
struct RocBlas {
  void* lib = nullptr; rb_handle h = nullptr; rb_create_t create = nullptr; rb_set_stream_t set_stream = nullptr;
  rb_sgemm_t sgemm = nullptr; bool tried = false;
} g_rb;

int rb_init() {
  if (g_rb.h) return 0;
  if (g_rb.tried) return fail(-20, "rocBLAS unavailable (earlier dlopen failed)");
  g_rb.tried = true;
  const char* names[] = {"librocblas.so", "librocblas.so.5", "librocblas.so.4", "/opt/rocm/lib/librocblas.so"};
  for (const char* n : names) {
    g_rb.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rb.lib) break;
  }
  if (!g_rb.lib) return fail(-20, "dlopen(librocblas.so) failed: %s", dlerror());
  g_rb.create = (rb_create_t)dlsym(g_rb.lib, "rocblas_create_handle");
  g_rb.set_stream = (rb_set_stream_t)dlsym(g_rb.lib, "rocblas_set_stream");
  g_rb.sgemm = (rb_sgemm_t)dlsym(g_rb.lib, "rocblas_sgemm");
  if (!g_rb.create || !g_rb.set_stream || !g_rb.sgemm) return fail(-20, "rocBLAS symbols missing");
  if (int rc = g_rb.create(&g_rb.h)) return fail(-21, "rocblas_create_handle failed (%d)", rc);
  return 0;
}

#ifndef SPO_MA_FUSE_MIN_ROWS_DEFAULT
#define SPO_MA_FUSE_MIN_ROWS_DEFAULT 2048
#endif
typedef float f4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- hand-written fp32 MFMA GEMM (all plain products)
// Y[B, N] (+)= X[B, R] * Wop[R, N], where the weight operand is addressed as Wop[r][j] = W[j * w_sj + r * w_sr]:
//   Y = X W^T  (forward blocks at small batch / hidden != 128, heads, tangent passes):  w_sj = R, w_sr = 1
//   dX = dY W  (head and non-fused block input gradients):                              w_sj = 1, w_sr = N_w (= K of W)
// 64 rows x 64 columns per workgroup, 4 waves x (16 rows x 64 columns), reduction in chunks of 32 staged through LDS
// (row stride 33: the 16 x 4 operand pattern of v_mfma_f32_16x16x4_f32 touches 32 different banks twice).  These shapes
// are launch-bound (8192 x 48..128 x 128 at collect time, heads with 1-16 columns), so the kernel is kept simple; the
// large-batch training products run in the fused block kernels below.  rocBLAS stays only as a test comparator
// (spo_debug_ma_gemm).
constexpr int GM_T = 64, GM_KC = 32, GM_LD = GM_KC + 1;
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y,
                                                        int64_t B, int R, int N, int64_t w_sj, int64_t w_sr, float beta) {
  __shared__ float Xs[GM_T * GM_LD];
  __shared__ float Ws[GM_T * GM_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * GM_T;          // rows on grid.x (2^31 - 1 tiles), the few column tiles on grid.y
  const int col0 = blockIdx.y * GM_T;
  f4v acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f4v{0.f, 0.f, 0.f, 0.f};
  const int lr = tid >> 2, ls = (tid & 3) * 8;                  // staging: row / column lr, 8 reduction indices from ls
  for (int r0 = 0; r0 < R; r0 += GM_KC) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = r0 + ls + e;
      const int64_t xr = row0 + lr;
      Xs[lr * GM_LD + ls + e] = (xr < B && r < R) ? X[xr * R + r] : 0.f;
      const int wc = col0 + lr;
      Ws[lr * GM_LD + ls + e] = (wc < N && r < R) ? W[(int64_t)wc * w_sj + (int64_t)r * w_sr] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GM_KC; kk += 4) {
      const float av = Xs[(16 * wave + i) * GM_LD + kk + kq];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Ws[(16 * nt + i) * GM_LD + kk + kq], acc[nt], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t row = row0 + 16 * wave + 4 * kq + reg;
      const int col = col0 + 16 * nt + i;
      if (row < B && col < N) {
        float* const y = Y + row * N + col;
        *y = beta != 0.f ? fmaf(beta, *y, acc[nt][reg]) : acc[nt][reg];
      }
    }
}
int gemm_mfma(hipStream_t st, const float* X, const float* W, float* Y, int64_t B, int R, int N, int64_t w_sj, int64_t w_sr,
              float beta) {
  const int64_t row_tiles = (B + GM_T - 1) / GM_T, col_tiles = (N + GM_T - 1) / GM_T;
  if (row_tiles > 0x7fffffffLL || col_tiles > 65535)
    return spo::fail(-1, "ma gemm: %lld x %d exceeds the launch grid (row tiles %lld, column tiles %lld)", (long long)B, N,
                     (long long)row_tiles, (long long)col_tiles);
  const dim3 grid((unsigned)row_tiles, (unsigned)col_tiles);
  hipLaunchKernelGGL(gemm_mfma_kernel, grid, dim3(256), 0, st, X, W, Y, B, R, N, w_sj, w_sr, beta);
  return 0;
}
// Heads: N <= 16 output columns (act_dim or 1) over K <= 512 features.  A 64 x 64 MFMA tile would be mostly padding; here a
// workgroup takes 64 rows, keeps W (+ bias) in LDS, and four lanes share a row: each forms the partial dot products of its
// quarter of the features for all N outputs, two shuffles add the quarters.  Bias fused (saves the add_bias launch).
constexpr int HS_MAXN = 16, HS_MAXK = 512;
__global__ __launch_bounds__(256) void head_small_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Y, int64_t B, int K, int N) {
  __shared__ float Ws[HS_MAXN * (HS_MAXK + 4)];
  const int tid = threadIdx.x;
  const int ld = K + 4;
  for (int idx = tid; idx < N * K; idx += 256) Ws[(idx / K) * ld + (idx % K)] = W[idx];
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * 64 + (tid >> 2);
  const int part = tid & 3;
  float acc[HS_MAXN];
#pragma unroll
  for (int n = 0; n < HS_MAXN; ++n) acc[n] = 0.f;
  if (row < B) {
    const float* xr = X + row * K;
    if ((K & 15) == 0) {
      // the four lanes of a row read consecutive float4s (64 contiguous bytes per row and step)
#pragma unroll 8
      for (int k0 = 4 * part; k0 < K; k0 += 16) {
        const f4v xv = *reinterpret_cast<const f4v*>(xr + k0);
#pragma unroll
        for (int n = 0; n < HS_MAXN; ++n)
          if (n < N) {
            const float* wr = Ws + n * ld + k0;
            acc[n] = fmaf(xv[0], wr[0], fmaf(xv[1], wr[1], fmaf(xv[2], wr[2], fmaf(xv[3], wr[3], acc[n]))));
          }
      }
    } else {
      for (int k = part; k < K; k += 4) {
        const float xv = xr[k];
#pragma unroll
        for (int n = 0; n < HS_MAXN; ++n)
          if (n < N) acc[n] = fmaf(xv, Ws[n * ld + k], acc[n]);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < HS_MAXN; ++n) {
    acc[n] += __shfl_xor(acc[n], 1);
    acc[n] += __shfl_xor(acc[n], 2);
  }
  if (row < B && part == 0) {
#pragma unroll
    for (int n = 0; n < HS_MAXN; ++n)
      if (n < N) Y[row * N + n] = acc[n] + (bias ? bias[n] : 0.f);
  }
}
// rows from which the fused block kernels (128 rows per workgroup) replace GEMM + LayerNorm kernels (SPO_MA_FUSE_MIN_ROWS: A/B knob)
inline int64_t ma_fuse_min_rows() {
  static const int64_t v = [] { const char* e = getenv("SPO_MA_FUSE_MIN_ROWS"); return e ? (int64_t)atoll(e) : (int64_t)SPO_MA_FUSE_MIN_ROWS_DEFAULT; }();
  return v;
}
inline bool ma_fwd_wave_private() {       // SPO_MA_FWD_WAVE=0: A/B knob, the four-wave barrier form of the fused block forward
  static const bool v = [] { const char* e = getenv("SPO_MA_FWD_WAVE"); return !(e && e[0] == '0'); }();
  return v;
}
inline bool ma_fuse_head() {       // SPO_MA_FUSE_HEAD=0: A/B knob, the head backward as separate launches
  static const bool v = [] { const char* e = getenv("SPO_MA_FUSE_HEAD"); return !(e && e[0] == '0'); }();
  return v;
}
int g_ma_gemm_rocblas = 0;      // spo_debug_ma_gemm only: 1 routes the two helpers below through rocBLAS (comparator)

// Row-major helpers.  Y[B,N] (+)= X[B,K] * W[N,K]^T
int gemm_xwT(hipStream_t st, const float* X, const float* W, float* Y, int64_t B, int K, int N, float beta = 0.f) {
  if (!g_ma_gemm_rocblas) return gemm_mfma(st, X, W, Y, B, K, N, K, 1, beta);
  const float alpha = 1.f;
  if (int rc = rb_init()) return rc;
  if (int rc = g_rb.set_stream(g_rb.h, st)) return fail(-22, "rocblas_set_stream (%d)", rc);
  if (int rc = g_rb.sgemm(g_rb.h, RB_T, RB_N, N, (int)B, K, &alpha, W, K, X, K, &beta, Y, N)) return fail(-22, "sgemm xwT (%d)", rc);
  return 0;
}
// dX[B,K] = dY[B,N] * W[N,K]
int gemm_dyw(hipStream_t st, const float* dY, const float* W, float* dX, int64_t B, int K, int N) {
  if (!g_ma_gemm_rocblas) return gemm_mfma(st, dY, W, dX, B, N, K, 1, K, 0.f);
  const float alpha = 1.f, beta = 0.f;
  if (int rc = rb_init()) return rc;
  if (int rc = g_rb.set_stream(g_rb.h, st)) return fail(-22, "rocblas_set_stream (%d)", rc);
  if (int rc = g_rb.sgemm(g_rb.h, RB_N, RB_N, K, (int)B, N, &alpha, W, K, dY, N, &beta, dX, K)) return fail(-22, "sgemm dyw (%d)", rc);
  return 0;
}
// dW[N,K] = dY[B,N]^T * X[B,K]: a tiny output reduced over a huge row count.  rocBLAS runs this shape at 5 TFLOP/s
// (one 256x64 macro-tile marching over 524 288 rows: 3.15 ms at N = K = 128, 59 % of a MAPPO-L epoch); the shape is
// HBM-bound (both operands are read once: 537 MB -> ~0.15 ms), so it is done here: the rows are split over up to 256
// workgroups per 128x128 output tile, each accumulating its slice with fp32 MFMA from LDS-staged 32-row chunks, and a
// second kernel adds the slices in a fixed order.
#ifndef SPO_DW_R
#define SPO_DW_R 32
#endif
constexpr int DW_T = 128, DW_R = SPO_DW_R, DW_LD = DW_T + 16;      // row stride 144: the two row-groups of a half-wave land 16 banks apart

inline int64_t dw_max_splits() {            // SPO_DW_MAX_SPLITS: A/B knob (256 = the round-1/2 value)
  static const int64_t v = [] { const char* e = getenv("SPO_DW_MAX_SPLITS"); const int64_t x = e ? atoll(e) : 512; return x < 1 ? 1 : x; }();
  return v;
}
int dw_splits(int64_t B, int N, int K) {
  const int64_t tiles = (int64_t)((N + DW_T - 1) / DW_T) * ((K + DW_T - 1) / DW_T);
  int64_t s = (B + 255) / 256;                       // at least 256 rows per slice
  const int64_t cap_mem = (int64_t)(1 << 24) / ((int64_t)N * K) > 0 ? (int64_t)(1 << 24) / ((int64_t)N * K) : 1;   // <= 64 MB of slices
  const int64_t cap_grid = 2048 / tiles > 0 ? 2048 / tiles : 1;
  // two (three) workgroups per CU: 256 slices were ONE workgroup per CU -- one wave per SIMD, every barrier and load exposed
  if (s > dw_max_splits()) s = dw_max_splits();
  if (s > cap_mem) s = cap_mem;
  if (s > cap_grid) s = cap_grid;
  return (int)(s < 1 ? 1 : s);
}

template <bool VEC>
__global__ __launch_bounds__(256, 2) void dw_partial_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                            float* __restrict__ partial, int64_t B, int N, int K, int S) {
  __shared__ __attribute__((aligned(16))) float Ys[DW_R * DW_LD];
  __shared__ __attribute__((aligned(16))) float Xs[DW_R * DW_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
  const int tiles_k = (K + DW_T - 1) / DW_T;
  const int tile = blockIdx.x / S, slice = blockIdx.x % S;
  const int n0 = (tile / tiles_k) * DW_T, k0 = (tile % tiles_k) * DW_T;
  const int64_t rows_per = ((B + S - 1) / S + DW_R - 1) / DW_R * DW_R;
  const int64_t r_begin = (int64_t)slice * rows_per;
  const int64_t r_end = r_begin + rows_per < B ? r_begin + rows_per : B;
  f4v acc[2][8];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[m][t] = f4v{0.f, 0.f, 0.f, 0.f};
  // The next 32-row chunk of both operands is fetched into registers while the current one is multiplied: with one or
  // two workgroups per CU nothing else hides the ~2 us load latency (318 -> ~190 us per call at 524 288 x 128 x 128).
  f4v py[DW_R / 8], px[DW_R / 8];
  auto fetch_chunk = [&](int64_t r0) {
#pragma unroll
    for (int q = 0; q < DW_R / 8; ++q) {
      const int idx = q * 256 + tid, row = idx >> 5, c4 = (idx & 31) * 4;
      const int64_t r = r0 + row;
      f4v y = {0.f, 0.f, 0.f, 0.f}, x = {0.f, 0.f, 0.f, 0.f};
      if (r < r_end) {
        if (VEC) {
          if (n0 + c4 < N) y = *reinterpret_cast<const f4v*>(dY + r * N + n0 + c4);
          if (k0 + c4 < K) x = *reinterpret_cast<const f4v*>(X + r * K + k0 + c4);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n0 + c4 + e < N) y[e] = dY[r * N + n0 + c4 + e];
            if (k0 + c4 + e < K) x[e] = X[r * K + k0 + c4 + e];
          }
        }
      }
      py[q] = y; px[q] = x;
    }
  };
  if (r_begin < r_end) fetch_chunk(r_begin);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += DW_R) {
    // stage DW_R rows x 128 columns of both operands (zero beyond the matrix edges)
#pragma unroll
    for (int q = 0; q < DW_R / 8; ++q) {
      const int idx = q * 256 + tid, row = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<f4v*>(Ys + row * DW_LD + c4) = py[q];
      *reinterpret_cast<f4v*>(Xs + row * DW_LD + c4) = px[q];
    }
    __syncthreads();
    if (r0 + DW_R < r_end) fetch_chunk(r0 + DW_R);            // in flight during the MFMA loop
#pragma unroll
    for (int st = 0; st < DW_R / 4; ++st) {
      const float* yr = Ys + (4 * st + kk) * DW_LD + 32 * wave + i;
      const float* xr = Xs + (4 * st + kk) * DW_LD + i;
      const float a0 = yr[0], a1 = yr[16];
      float b[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) b[t] = xr[16 * t];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t], acc[1][t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // C layout: lane holds C[row = 4*(lane>>4) + e][col = lane & 15] of each 16x16 tile
  float* out = partial + (int64_t)slice * N * K;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + 32 * wave + 16 * m + 4 * kk + e, k = k0 + 16 * t + i;
        if (n < N && k < K) out[(int64_t)n * K + k] = acc[m][t][e];
      }
}
// 64 outputs x 16 strided groups of slices per workgroup (1024 threads), combined in a fixed order
__global__ __launch_bounds__(1024) void dw_reduce_kernel(const float* __restrict__ partial, int S, int64_t NK, float* __restrict__ out) {
  __shared__ float sh[16][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + c;
  float s = 0.f;
  if (j < NK)
    for (int b = q; b < S; b += 16) s += partial[(int64_t)b * NK + j];
  sh[q][c] = s;
  __syncthreads();
  if (q == 0 && j < NK) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][c];
    out[j] = t;
  }
}
// `slices`: float[dw_splits(B, N, K) * N * K]
int gemm_dyTx(hipStream_t st, const float* dY, const float* X, float* dW, int64_t B, int K, int N, float* slices) {
  const int S = dw_splits(B, N, K);
  const int tiles = ((N + DW_T - 1) / DW_T) * ((K + DW_T - 1) / DW_T);
  const bool vec = (N % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) % 16 == 0);
  if (vec) hipLaunchKernelGGL(dw_partial_kernel<true>, dim3(tiles * S), dim3(256), 0, st, dY, X, slices, B, N, K, S);
  else hipLaunchKernelGGL(dw_partial_kernel<false>, dim3(tiles * S), dim3(256), 0, st, dY, X, slices, B, N, K, S);
  const int64_t NK = (int64_t)N * K;
  hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)((NK + 63) / 64)), dim3(1024), 0, st, slices, S, NK, dW);
  return 0;
}

// ---------------------------------------------------------------- layout
struct Lay {
  int D, H, NB, O, actor;
  int64_t fn_g() const { return 0; }
  int64_t fn_b() const { return D; }
  int in_k(int k) const { return k == 0 ? D : H; }
  int64_t blk(int k) const {          // start of block k
    int64_t o = 2 * (int64_t)D;
    for (int i = 0; i < k; ++i) o += (int64_t)H * in_k(i) + 3 * (int64_t)H;
    return o;
  }
  int64_t W(int k) const { return blk(k); }
  int64_t b(int k) const { return blk(k) + (int64_t)H * in_k(k); }
  int64_t g(int k) const { return b(k) + H; }
  int64_t be(int k) const { return g(k) + H; }
  int64_t logstd() const { return blk(NB); }
  int64_t hW() const { return blk(NB) + (actor ? O : 0); }
  int64_t hb() const { return hW() + (int64_t)O * H; }
  int64_t count() const { return hb() + O; }
  // workspace (floats) for B rows: xhat[B,D], st0[B,2], per block: a[B,H], st[B,2], y[B,H]; every region starts on a
  // 16-byte boundary (row statistics padded to a multiple of 4 floats) so the 128-wide kernels can use float4 accesses
  static int64_t al4(int64_t n) { return (n + 3) & ~(int64_t)3; }
  int64_t ws_xhat() const { return 0; }
  int64_t ws_st0(int64_t B) const { return al4(B * D); }
  int64_t ws_blk(int64_t B, int k) const { return al4(B * D) + al4(2 * B) + (int64_t)k * (2 * al4(B * H) + al4(2 * B)); }
  int64_t ws_a(int64_t B, int k) const { return ws_blk(B, k); }
  int64_t ws_st(int64_t B, int k) const { return ws_blk(B, k) + al4(B * H); }
  int64_t ws_y(int64_t B, int k) const { return ws_blk(B, k) + al4(B * H) + al4(2 * B); }
  // block 0's weights with the input LayerNorm's affine folded in: W'[n][c] = W_0[n][c] * gamma[c], b'[n] = b_0[n] + W_0[n] . beta
  int64_t ws_w0f(int64_t B) const { return ws_blk(B, NB); }
  int64_t ws_b0f(int64_t B) const { return ws_w0f(B) + al4((int64_t)H * D); }
  int64_t ws_count(int64_t B) const { return ws_b0f(B) + al4(H); }
};

int lay_of(const spo_ma_net* n, Lay* L) {
  if (!n) return fail(-1, "ma: net is NULL");
  if (n->in_dim < 1 || n->in_dim > 512 || n->hidden < 1 || n->hidden > 512 || n->n_blocks < 1 || n->n_blocks > 8 ||
      n->out_dim < 1 || n->out_dim > 64)
    return fail(-2, "ma: net dims out of range (in %d, hidden %d, blocks %d, out %d; limits 512/512/8/64)", n->in_dim,
                n->hidden, n->n_blocks, n->out_dim);
  *L = Lay{n->in_dim, n->hidden, n->n_blocks, n->out_dim, n->is_actor ? 1 : 0};
  return 0;
}

constexpr float LN_EPS = 1e-5f;
constexpr int MAXE = 8;       // elements per lane in a row (dims <= 512)

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One wave per row.  MODE 0: y = LN(x)*g + b                      (feature_norm)
//                    MODE 1: a = ELU(z + bias); y = LN(a)*g + b    (block epilogue; a and the row statistics are kept)
template <int MODE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                      const float* __restrict__ g, const float* __restrict__ b,
                                                      float* __restrict__ a_out, float* __restrict__ y,
                                                      float* __restrict__ stats, int64_t B, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < B; row += (int64_t)gridDim.x * 4) {
    float v[MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int j = lane + 64 * e;
      float t = 0.f;
      if (j < D) {
        t = x[row * D + j];
        if (MODE == 1) {
          t += bias[j];
          t = t > 0.f ? t : expm1f(t);          // nn.ELU(alpha=1)
          a_out[row * D + j] = t;
        }
      }
      v[e] = t; s += t;
    }
    const float mean = wave_sum_all(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int j = lane + 64 * e;
      const float d = j < D ? v[e] - mean : 0.f;
      q += d * d;
    }
    const float var = wave_sum_all(q) / (float)D;          // biased, as nn.LayerNorm
    const float rstd = 1.f / sqrtf(var + LN_EPS);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int j = lane + 64 * e;
      if (j < D) y[row * D + j] = (MODE == 0) ? (v[e] - mean) * rstd : (v[e] - mean) * rstd * g[j] + b[j];   // MODE 0: xn, affine folded into W_0
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// Backward of y = LN(a)*g + b with a = ELU(z): dz (in place over dy is allowed) and per-block partial column sums
// of d(ln weight) = sum dy*xhat, d(ln bias) = sum dy, d(linear bias) = sum dz.   MODE 0: feature_norm (no dz: the input
// needs no gradient), `a` is the raw input x.
template <int MODE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                      const float* __restrict__ stats, const float* __restrict__ g,
                                                      float* __restrict__ dz, float* __restrict__ partial, int64_t B,
                                                      int D) {
  __shared__ float sh[3][4][64 * MAXE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float cg[MAXE], cb[MAXE], cz[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) cg[e] = cb[e] = cz[e] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < B; row += (int64_t)gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[MAXE], dxh[MAXE], av[MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int j = lane + 64 * e;
      float d = 0.f, t = 0.f, gg = 0.f;
      if (j < D) { d = dy[row * D + j]; t = a[row * D + j]; gg = g[j]; }
      av[e] = t;
      xh[e] = j < D ? (t - mean) * rstd : 0.f;
      dxh[e] = d * gg;
      cg[e] += d * xh[e]; cb[e] += d;
      s1 += dxh[e]; s2 += dxh[e] * xh[e];
    }
    if (MODE == 1) {
      const float m1 = wave_sum_all(s1) / (float)D, m2 = wave_sum_all(s2) / (float)D;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) {
        const int j = lane + 64 * e;
        if (j < D) {
          const float da = rstd * (dxh[e] - m1 - xh[e] * m2);
          const float dzv = da * (av[e] > 0.f ? 1.f : av[e] + 1.f);      // ELU' = exp(z) = a + 1 for z <= 0
          dz[row * D + j] = dzv;
          cz[e] += dzv;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < MAXE; ++e) { sh[0][wave][lane + 64 * e] = cg[e]; sh[1][wave][lane + 64 * e] = cb[e]; sh[2][wave][lane + 64 * e] = cz[e]; }
  __syncthreads();
  for (int j = threadIdx.x; j < D; j += 256)
#pragma unroll
    for (int w = 0; w < 3; ++w)
      partial[((int64_t)blockIdx.x * 3 + w) * D + j] = (sh[w][0][j] + sh[w][1][j]) + (sh[w][2][j] + sh[w][3][j]);
}

// out[w][j] = sum over blocks (fixed order) of partial[block][w][j]
// ---- 128-wide rows (hidden_size 128: the MuJoCo configs): float4 per lane, TWO rows per wave (lanes 0-31 / 32-63), so a
// wave moves 1 KB per instruction instead of 256 B and reduces over 32 lanes.  Same arithmetic as the generic kernels.
typedef float f4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float row16_allsum_fwd(float v) {      // sum over the 16 lanes of a DPP row, in every lane
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
template <int MODE>
__global__ __launch_bounds__(256) void ln_fwd128_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                         const float* __restrict__ g, const float* __restrict__ b,
                                                         float* __restrict__ a_out, float* __restrict__ y,
                                                         float* __restrict__ stats, int64_t B) {
  constexpr int D = 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = (lane & 31) * 4, sub = lane >> 5;
  const f4w gg = *reinterpret_cast<const f4w*>(g + c), bb = *reinterpret_cast<const f4w*>(b + c);
  f4w bi = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 1) bi = *reinterpret_cast<const f4w*>(bias + c);
  for (int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 2 + sub; row < B + sub; row += (int64_t)gridDim.x * 8) {
    const bool ok = row < B;                                   // the two halves of a wave stay in the loop together
    f4w v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f4w*>(x + row * D + c);
    if (MODE == 1) {
      v += bi;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);
      if (ok) *reinterpret_cast<f4w*>(a_out + row * D + c) = v;
    }
    const float mean = half_wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / D);
    const f4w d = v - mean;
    const float var = half_wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.f / D);
    const float rstd = 1.f / sqrtf(var + LN_EPS);
    if (ok) {
      *reinterpret_cast<f4w*>(y + row * D + c) = d * rstd * gg + bb;
      if ((lane & 31) == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    }
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void ln_bwd128_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                         const float* __restrict__ stats, const float* __restrict__ g,
                                                         float* __restrict__ dz, float* __restrict__ partial, int64_t B) {
  constexpr int D = 128;
  __shared__ float sh[3][8][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = (lane & 31) * 4, sub = lane >> 5;
  const f4w gg = *reinterpret_cast<const f4w*>(g + c);
  f4w cg = {0.f, 0.f, 0.f, 0.f}, cb = cg, cz = cg;
  for (int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 2 + sub; row < B + sub; row += (int64_t)gridDim.x * 8) {
    const bool ok = row < B;
    f4w d = {0.f, 0.f, 0.f, 0.f}, av = d;
    float mean = 0.f, rstd = 0.f;
    if (ok) {
      d = *reinterpret_cast<const f4w*>(dy + row * D + c);
      av = *reinterpret_cast<const f4w*>(a + row * D + c);
      mean = stats[2 * row]; rstd = stats[2 * row + 1];
    }
    const f4w xh = (av - mean) * rstd;
    const f4w dxh = d * gg;
    cg += d * xh; cb += d;
    if (MODE == 1) {
      const float m1 = half_wave_sum((dxh[0] + dxh[1]) + (dxh[2] + dxh[3])) * (1.f / D);
      const f4w t = dxh * xh;
      const float m2 = half_wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.f / D);
      f4w dzv = (dxh - m1 - xh * m2) * rstd;
#pragma unroll
      for (int e = 0; e < 4; ++e) dzv[e] *= av[e] > 0.f ? 1.f : av[e] + 1.f;
      if (ok) *reinterpret_cast<f4w*>(dz + row * D + c) = dzv;
      cz += ok ? dzv : f4w{0.f, 0.f, 0.f, 0.f};
    }
  }
  *reinterpret_cast<f4w*>(&sh[0][2 * wave + sub][c]) = cg;
  *reinterpret_cast<f4w*>(&sh[1][2 * wave + sub][c]) = cb;
  *reinterpret_cast<f4w*>(&sh[2][2 * wave + sub][c]) = cz;
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * D; idx += 256) {
    const int w = idx / D, jcol = idx % D;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += sh[w][k][jcol];
    partial[((int64_t)blockIdx.x * 3 + w) * D + jcol] = s;
  }
}

// ---- head backward fused with the top block's LayerNorm/ELU backward (hidden 128, out_dim <= 16).  The per-layer sequence
// -- dW_head = dout^T y (a 128 x 128-tile MFMA kernel with 1..16 useful columns), colsum(dout), dY = dout W_head (GEMM, 268 MB
// written at 524 288 rows) and ln_bwd128_kernel (268 MB read back) -- read y / dY / a three times; here half a wave takes a row
// once: dY = sum_o dout[o] W_head[o] in registers (W_head lives in registers: O float4 per lane), y recomputed from a and the
// row statistics exactly as the forward formed it, dW_head / db_head accumulated per lane, then the LayerNorm/ELU backward of
// ln_bwd128_kernel<1> on the dY it just formed.  HBM: a in, dz out (536 MB instead of ~1.6 GB at 524 288 rows).
// Partials: `partial` as ln_bwd128_kernel (3 x 128 per workgroup); hpw [grid][O][128], hpb [grid][O] for the finish kernel.
template <int OMAX>
__global__ __launch_bounds__(256) void head_bwd_lnbwd128_kernel(const float* __restrict__ dout, const float* __restrict__ hW,
                                                                const float* __restrict__ a, const float* __restrict__ stats,
                                                                const float* __restrict__ g, const float* __restrict__ be,
                                                                float* __restrict__ dz, float* __restrict__ partial,
                                                                float* __restrict__ hpw, float* __restrict__ hpb, int64_t B, int O) {
  constexpr int D = 128;
  __shared__ float sh[3][8][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = (lane & 31) * 4, sub = lane >> 5;
  const f4w gg = *reinterpret_cast<const f4w*>(g + c), bb = *reinterpret_cast<const f4w*>(be + c);
  f4w hw[OMAX], acc[OMAX];
  float accb[OMAX];
#pragma unroll
  for (int o = 0; o < OMAX; ++o) {
    hw[o] = f4w{0.f, 0.f, 0.f, 0.f};
    if (o < O) { hw[o][0] = hW[o * D + c]; hw[o][1] = hW[o * D + c + 1]; hw[o][2] = hW[o * D + c + 2]; hw[o][3] = hW[o * D + c + 3]; }
    acc[o] = f4w{0.f, 0.f, 0.f, 0.f};
    accb[o] = 0.f;
  }
  f4w cg = {0.f, 0.f, 0.f, 0.f}, cb = cg, cz = cg;
  // the next row's loads are issued one iteration ahead: with 200+ registers only two waves share a SIMD and nothing else
  // hides the load latency (196 -> ~100 us at out_dim 6, 524 288 rows)
  f4w av_n = {0.f, 0.f, 0.f, 0.f};
  float mean_n = 0.f, rstd_n = 0.f, dov_n[OMAX];
  auto fetch_row = [&](int64_t row) {
    av_n = f4w{0.f, 0.f, 0.f, 0.f}; mean_n = 0.f; rstd_n = 0.f;
#pragma unroll
    for (int o = 0; o < OMAX; ++o) dov_n[o] = 0.f;
    if (row < B) {
      av_n = *reinterpret_cast<const f4w*>(a + row * D + c);
      mean_n = stats[2 * row]; rstd_n = stats[2 * row + 1];
#pragma unroll
      for (int o = 0; o < OMAX; ++o)
        if (o < O) dov_n[o] = dout[row * O + o];
    }
  };
  const int64_t rstep = (int64_t)gridDim.x * 8;
  int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 2 + sub;
  fetch_row(row);
  for (; row < B + sub; row += rstep) {
    const bool ok = row < B;
    const f4w av = av_n;
    const float mean = mean_n, rstd = rstd_n;
    float dov[OMAX];
#pragma unroll
    for (int o = 0; o < OMAX; ++o) dov[o] = dov_n[o];
    fetch_row(row + rstep);
    const f4w xh = (av - mean) * rstd;
    const f4w y = xh * gg + bb;                                   // the forward's y = (a - mean) * rstd * g + b
    f4w d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < OMAX; ++o) {
      d += hw[o] * dov[o];
      acc[o] += y * dov[o];
      accb[o] += dov[o];
    }
    const f4w dxh = d * gg;
    cg += d * xh; cb += d;
    const float m1 = half_wave_sum((dxh[0] + dxh[1]) + (dxh[2] + dxh[3])) * (1.f / D);
    const f4w t = dxh * xh;
    const float m2 = half_wave_sum((t[0] + t[1]) + (t[2] + t[3])) * (1.f / D);
    f4w dzv = (dxh - m1 - xh * m2) * rstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) dzv[e] *= av[e] > 0.f ? 1.f : av[e] + 1.f;
    if (ok) *reinterpret_cast<f4w*>(dz + row * D + c) = dzv;
    cz += ok ? dzv : f4w{0.f, 0.f, 0.f, 0.f};
  }
  *reinterpret_cast<f4w*>(&sh[0][2 * wave + sub][c]) = cg;
  *reinterpret_cast<f4w*>(&sh[1][2 * wave + sub][c]) = cb;
  *reinterpret_cast<f4w*>(&sh[2][2 * wave + sub][c]) = cz;
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * D; idx += 256) {
    const int w = idx / D, jcol = idx % D;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += sh[w][k][jcol];
    partial[((int64_t)blockIdx.x * 3 + w) * D + jcol] = s;
  }
  // head-weight partials: the eight half-waves of the workgroup added in a fixed order, one output row at a time
#pragma unroll
  for (int o = 0; o < OMAX; ++o) {
    if (o >= O) break;                                            // (uniform)
    __syncthreads();
    *reinterpret_cast<f4w*>(&sh[0][2 * wave + sub][c]) = acc[o];
    if ((lane & 31) == 0) sh[1][2 * wave + sub][0] = accb[o];
    __syncthreads();
    if (threadIdx.x < D) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += sh[0][k][threadIdx.x];
      hpw[((int64_t)blockIdx.x * O + o) * D + threadIdx.x] = s;
    } else if (threadIdx.x == D) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += sh[1][k][0];
      hpb[(int64_t)blockIdx.x * O + o] = s;
    }
  }
}
// sums the workgroup partials of head_bwd_lnbwd128_kernel in a fixed order: grid (O, 4 column chunks of 32), 512 threads =
// 16 groups x 32 columns; group q adds partials q, q + 16, ... (8 loads in flight), the 16 group sums are added in order
__global__ __launch_bounds__(512) void head_dw_finish_kernel(const float* __restrict__ hpw, const float* __restrict__ hpb, int nblocks,
                                                             int O, float* __restrict__ dW, float* __restrict__ db) {
  constexpr int D = 128;
  __shared__ float sh[16][33];
  const int o = blockIdx.x, cl = threadIdx.x & 31, c = blockIdx.y * 32 + cl, q = threadIdx.x >> 5;
  const bool with_b = blockIdx.y == 0 && cl == 0;
  float s = 0.f, sb = 0.f;
  int b = q;
  for (; b + 7 * 16 < nblocks; b += 8 * 16) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = hpw[((int64_t)(b + 16 * u) * O + o) * D + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
    if (with_b) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sb += hpb[(int64_t)(b + 16 * u) * O + o];
    }
  }
  for (; b < nblocks; b += 16) {
    s += hpw[((int64_t)b * O + o) * D + c];
    if (with_b) sb += hpb[(int64_t)b * O + o];
  }
  sh[q][cl] = s;
  if (with_b) sh[q][32] = sb;
  __syncthreads();
  if (q == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][cl];
    dW[o * D + c] = t;
    if (with_b) {
      float tb = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) tb += sh[k][32];
      db[o] = tb;
    }
  }
}

// ---- input LayerNorm (feature_norm) for narrow observations (dim <= 128, multiple of 4): float4 per lane, LPR lanes per
// row (16 for dim <= 64, 32 for dim <= 128), 64/LPR rows per wave; the generic kernel spends one wave on a 48-wide row.
template <int LPR>
__device__ __forceinline__ float group_allsum(float v) {
  if (LPR == 32) v += __shfl_xor(v, 16, 64);
  return row16_allsum_fwd(v);
}
template <int LPR>
__global__ __launch_bounds__(256) void ln_fwd_narrow_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ stats, int64_t B, int D) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPR, c = (lane % LPR) * 4;
  const bool col_ok = c < D;
  f4w gg = {0.f, 0.f, 0.f, 0.f}, bb = gg;
  if (col_ok) { gg = *reinterpret_cast<const f4w*>(g + c); bb = *reinterpret_cast<const f4w*>(b + c); }
  const float inv_d = 1.f / (float)D;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; r0 < B; r0 += (int64_t)gridDim.x * 4 * RPW) {
    const int64_t row = r0 + sub;
    const bool ok = row < B && col_ok;
    f4w v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f4w*>(x + row * D + c);
    const float mean = group_allsum<LPR>((v[0] + v[1]) + (v[2] + v[3])) * inv_d;
    f4w d = v - mean;
    if (!col_ok) d = f4w{0.f, 0.f, 0.f, 0.f};
    const float var = group_allsum<LPR>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * inv_d;
    const float rstd = 1.f / sqrtf(var + LN_EPS);
    if (ok) *reinterpret_cast<f4w*>(y + row * D + c) = d * rstd;          // pre-affine: gamma / beta are folded into W_0
    if (row < B && (lane % LPR) == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// ---- fused block forward for hidden 128:  y = LayerNorm(ELU(X W^T + b))  in ONE kernel (GEMM on fp32 MFMA, epilogue in
// registers) instead of rocBLAS sgemm + the bias/ELU/LayerNorm kernel: the GEMM output never goes to HBM and back.
// Workgroup = 128 rows x all 128 output columns (a LayerNorm row stays inside one workgroup), W (128 x K) staged once per
// workgroup and reused over its grid-stride row tiles.  MFMA 16x16x4 with the k index permuted inside each 16-wide k block
// (lane group kk supplies k = 16*kb + 4*kk + reg) so that ONE ds_read_b128 per operand row feeds four MFMAs.
constexpr int FB_N = 128, FB_ROWS = 128, FB_SLD = FB_N + 4;
__device__ __forceinline__ float row16_allsum(float v) {          // sum over the 16 lanes of a DPP row, in every lane
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
// MT = 16-row MFMA tiles per wave: 2 (128-row workgroup tiles) for training batches, 1 (64-row tiles, twice the
// workgroups) for collect-size batches that would otherwise leave most CUs idle.
template <int MT>
__global__ __launch_bounds__(256, 1) void fused_block_fwd128_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                    const float* __restrict__ bias, const float* __restrict__ g,
                                                                    const float* __restrict__ be, float* __restrict__ a_out,
                                                                    float* __restrict__ y, float* __restrict__ stats,
                                                                    int64_t B, int K) {
  extern __shared__ __attribute__((aligned(16))) float fb_lds[];
  const int KP = (K + 15) & ~15;                 // k padded to whole 16-blocks (zero filled)
  const int LD = KP + 4;                         // row stride: 16-byte aligned, bank-rotated
  float* Ws = fb_lds;                            // [128][LD]
  float* Xs = fb_lds + FB_N * LD;                // [ROWS][LD]
  float* Stg = Xs;                               // [128][FB_SLD]  ELU outputs, row-major for coalesced stores: ALIASES the X tile
  float* Sst = Xs + 64 * MT * FB_SLD;            // [128][2]       row mean / rstd
  constexpr int ROWS = 64 * MT, XP = 8 * MT;     // rows per workgroup tile, staging passes of 8 rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
  // staging map: thread -> (row tid/32 + 8*pass, columns 4*(tid%32)..+3): no divisions, 16 independent loads in flight
  const int lc4 = (tid & 31) * 4, lrow = tid >> 5;
  const bool lcol_ok = lc4 < K, lcol_in = lc4 < KP;      // K % 4 == 0; columns K..KP-1 are zero padding
  {
    f4w wv[16];
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) {
      wv[ps] = f4w{0.f, 0.f, 0.f, 0.f};
      if (lcol_ok) wv[ps] = *reinterpret_cast<const f4w*>(W + (int64_t)(lrow + 8 * ps) * K + lc4);
    }
#pragma unroll
    for (int ps = 0; ps < 16; ++ps)
      if (lcol_in) *reinterpret_cast<f4w*>(Ws + (lrow + 8 * ps) * LD + lc4) = wv[ps];
  }
  float bcol[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) bcol[t] = bias[16 * t + i];
  const int64_t ntiles = (B + ROWS - 1) / ROWS;
  // The kernel is HBM-bound (read X once, write a and y once: 192 KB per 128-row tile), so the next tile's rows are
  // fetched into registers while this tile is multiplied and normalised.
  f4w xv[XP];
  auto fetch_tile = [&](int64_t t) {
    const int64_t rb = t * ROWS;
#pragma unroll
    for (int ps = 0; ps < XP; ++ps) {
      const int rr = lrow + 8 * ps;
      xv[ps] = f4w{0.f, 0.f, 0.f, 0.f};
      if (lcol_ok && t < ntiles && rb + rr < B) xv[ps] = *reinterpret_cast<const f4w*>(X + (rb + rr) * K + lc4);
    }
  };
  fetch_tile(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * ROWS;
    __syncthreads();                             // previous tile's output image has been read out (and Ws is staged)
#pragma unroll
    for (int ps = 0; ps < XP; ++ps)
      if (lcol_in) *reinterpret_cast<f4w*>(Xs + (lrow + 8 * ps) * LD + lc4) = xv[ps];
    __syncthreads();
    fetch_tile(tile + gridDim.x);                // in flight during the MFMA loop and the epilogue
    f4w acc[MT][8];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[m][t] = f4w{0.f, 0.f, 0.f, 0.f};
    const float* xa = Xs + (16 * MT * wave + i) * LD + 4 * kk;
    const float* wb = Ws + i * LD + 4 * kk;
    for (int kb = 0; kb < KP / 16; ++kb) {
      f4w am[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) am[m] = *reinterpret_cast<const f4w*>(xa + 16 * m * LD + 16 * kb);
      f4w bt[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) bt[t] = *reinterpret_cast<const f4w*>(wb + 16 * t * LD + 16 * kb);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[m][r], bt[t][r], acc[m][t], 0, 0, 0);
        }
    }
    __syncthreads();                             // every wave is done with the X tile: its LDS becomes the output image
    // epilogue: lane holds rows 4*kk + e (e = 0..3) of each 16-row tile m, columns 16*t + i.  The ELU outputs go through
    // an LDS image of this wave's 32 rows so that HBM sees whole 512-byte rows (a and y), not 64-byte column slivers.
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rl = 16 * MT * wave + 16 * m + 4 * kk + e;
        float v[8], sum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float z = acc[m][t][e] + bcol[t];
          z = z > 0.f ? z : __expf(z) - 1.f;            // nn.ELU: exp(x) - 1 (abs error < 1e-7)
          v[t] = z; sum += z;
        }
        const float mean = row16_allsum(sum) * (1.f / FB_N);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) { const float d = v[t] - mean; q += d * d; }
        const float rstd = 1.f / sqrtf(row16_allsum(q) * (1.f / FB_N) + LN_EPS);
#pragma unroll
        for (int t = 0; t < 8; ++t) Stg[rl * FB_SLD + 16 * t + i] = v[t];
        if (i == 0) { Sst[2 * rl] = mean; Sst[2 * rl + 1] = rstd; }
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes have landed (rows are wave-private)
    {
      const int c4 = (lane & 31) * 4;
      const f4w g4 = *reinterpret_cast<const f4w*>(g + c4), be4 = *reinterpret_cast<const f4w*>(be + c4);
#pragma unroll 4
      for (int jj = 0; jj < 8 * MT; ++jj) {
        const int rl = 16 * MT * wave + 2 * jj + (lane >> 5);
        const int64_t row = r0 + rl;
        const f4w a4 = *reinterpret_cast<const f4w*>(Stg + rl * FB_SLD + c4);
        const float mean = Sst[2 * rl], rstd = Sst[2 * rl + 1];
        if (row < B) {
          *reinterpret_cast<f4w*>(a_out + row * FB_N + c4) = a4;
          *reinterpret_cast<f4w*>(y + row * FB_N + c4) = (a4 - mean) * rstd * g4 + be4;
          if ((lane & 31) == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
        }
      }
    }
  }
}

// ---- the same block with WAVE-PRIVATE row tiles (round 3; training-size batches).  The kernel above synchronises its four
// waves twice per tile (stage X together, multiply, epilogue) with one wave per SIMD: every phase of every wave waits for the
// slowest and nothing hides the global loads / stores.  Here a workgroup is 8 waves around ONE shared weight image; each wave
// owns 16-row tiles end to end -- its rows prefetched into registers one tile ahead, written to its own 8 KB of LDS, multiplied,
// ELU + LayerNorm in registers, rows stored through the same LDS region -- and never meets a barrier after the weights are
// staged, so two waves per SIMD overlap one wave's memory phases with the other's MFMAs.  Per-element arithmetic, operand
// order and cross-lane sums are those of fused_block_fwd128_kernel (results bit-identical).
constexpr int FBW_WAVES = 8;
__global__ __launch_bounds__(64 * FBW_WAVES, 1) void fused_block_fwd128w_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                              const float* __restrict__ bias, const float* __restrict__ g,
                                                                              const float* __restrict__ be, float* __restrict__ a_out,
                                                                              float* __restrict__ y, float* __restrict__ stats,
                                                                              int64_t B, int K) {
  extern __shared__ __attribute__((aligned(16))) float fb_lds[];
  const int KP = (K + 15) & ~15;
  const int LD = KP + 4;
  float* Ws = fb_lds;                                          // [128][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
  float* Xw = fb_lds + FB_N * FB_SLD + wave * (16 * FB_SLD + 32);      // this wave's [16][132] tile (X rows, then the ELU outputs)
  float* Sw = Xw + 16 * FB_SLD;                                // [16][2] row mean / rstd
  {
    // weights: 512 threads, 16 rows x 128 columns per pass, 8 passes
    const int lc4 = (tid & 31) * 4, lrow = tid >> 5;
    const bool lcol_ok = lc4 < K, lcol_in = lc4 < KP;
    f4w wv[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      wv[ps] = f4w{0.f, 0.f, 0.f, 0.f};
      if (lcol_ok) wv[ps] = *reinterpret_cast<const f4w*>(W + (int64_t)(lrow + 16 * ps) * K + lc4);
    }
#pragma unroll
    for (int ps = 0; ps < 8; ++ps)
      if (lcol_in) *reinterpret_cast<f4w*>(Ws + (lrow + 16 * ps) * LD + lc4) = wv[ps];
  }
  float bcol[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) bcol[t] = bias[16 * t + i];
  const int c4 = (lane & 31) * 4, rsub = lane >> 5;            // row-major access: half a wave per row, float4 per lane
  const bool xcol_ok = c4 < K, xcol_in = c4 < KP;
  const f4w g4 = *reinterpret_cast<const f4w*>(g + c4), be4 = *reinterpret_cast<const f4w*>(be + c4);
  const int64_t ntiles = (B + 15) / 16;
  const int64_t tstride = (int64_t)gridDim.x * FBW_WAVES;
  f4w xv[8];
  auto fetch_tile = [&](int64_t t) {
    const int64_t rb = t * 16;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int rr = 2 * ps + rsub;
      xv[ps] = f4w{0.f, 0.f, 0.f, 0.f};
      if (xcol_ok && t < ntiles && rb + rr < B) xv[ps] = *reinterpret_cast<const f4w*>(X + (rb + rr) * K + c4);
    }
  };
  int64_t tile = (int64_t)blockIdx.x * FBW_WAVES + wave;
  fetch_tile(tile);
  __syncthreads();                                             // the weight image is complete (the only barrier)
  for (; tile < ntiles; tile += tstride) {
    const int64_t r0 = tile * 16;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps)
      if (xcol_in) *reinterpret_cast<f4w*>(Xw + (2 * ps + rsub) * LD + c4) = xv[ps];
    __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): the tile is wave-private
    fetch_tile(tile + tstride);                                // in flight during the MFMA loop and the epilogue
    f4w acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f4w{0.f, 0.f, 0.f, 0.f};
    const float* xa = Xw + i * LD + 4 * kk;
    const float* wb = Ws + i * LD + 4 * kk;
    for (int kb = 0; kb < KP / 16; ++kb) {
      const f4w am = *reinterpret_cast<const f4w*>(xa + 16 * kb);
      f4w bt[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) bt[t] = *reinterpret_cast<const f4w*>(wb + 16 * t * LD + 16 * kb);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[r], bt[t][r], acc[t], 0, 0, 0);
    }
    // epilogue: lane holds rows 4*kk + e of the tile, columns 16*t + i
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = 4 * kk + e;
      float v[8], sum = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float z = acc[t][e] + bcol[t];
        z = z > 0.f ? z : __expf(z) - 1.f;
        v[t] = z; sum += z;
      }
      const float mean = row16_allsum(sum) * (1.f / FB_N);
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) { const float d = v[t] - mean; q += d * d; }
      const float rstd = 1.f / sqrtf(row16_allsum(q) * (1.f / FB_N) + LN_EPS);
#pragma unroll
      for (int t = 0; t < 8; ++t) Xw[rl * FB_SLD + 16 * t + i] = v[t];
      if (i == 0) { Sw[2 * rl] = mean; Sw[2 * rl + 1] = rstd; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll 4
    for (int jj = 0; jj < 8; ++jj) {
      const int rl = 2 * jj + rsub;
      const int64_t row = r0 + rl;
      const f4w a4 = *reinterpret_cast<const f4w*>(Xw + rl * FB_SLD + c4);
      const float mean = Sw[2 * rl], rstd = Sw[2 * rl + 1];
      if (row < B) {
        *reinterpret_cast<f4w*>(a_out + row * FB_N + c4) = a4;
        *reinterpret_cast<f4w*>(y + row * FB_N + c4) = (a4 - mean) * rstd * g4 + be4;
        if ((lane & 31) == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                        // the LDS reads above are done before the next tile overwrites the region
  }
}

// ---- fused block backward for hidden 128: from dz_k (gradient at block k's pre-activation) straight to dz_{k-1}:
//   dY = dz_k * W_k                              (the dX GEMM, fp32 MFMA)
//   dz_{k-1} = LayerNorm'(dY; a_{k-1}, stats_{k-1}, gamma_{k-1}) * ELU'(a_{k-1})          (epilogue in registers)
// plus the per-workgroup column partials of block k-1's {ln weight, ln bias, linear bias} gradients, in the layout
// colsum_finish_kernel reads.  Replaces rocBLAS sgemm (285 us) + ln_bwd128_kernel (142 us): dY never goes to HBM.
// 64-row tiles; W_k is staged TRANSPOSED once per workgroup so the B operand reads 4 consecutive k per lane (b128).
constexpr int FBW_ROWS = 64;
__global__ __launch_bounds__(256, 1) void fused_dx_lnbwd128_kernel(const float* __restrict__ dz, const float* __restrict__ W,
                                                                   const float* __restrict__ a_prev, const float* __restrict__ stats_prev,
                                                                   const float* __restrict__ g_prev, float* __restrict__ dz_out,
                                                                   float* __restrict__ partial, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) float fb_lds[];
  constexpr int LD = FB_N + 4;
  float* WsT = fb_lds;                           // [128 c][LD]  W^T: WsT[c][n] = W[n][c]
  float* Ds = WsT + FB_N * LD;                   // [64][LD]     dz tile
  float* As = Ds + FBW_ROWS * LD;                // [64][LD]     a_{k-1} tile, overwritten in place by dz_{k-1}
  float* Sst = As + FBW_ROWS * LD;               // [64][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
  const int lc4 = (tid & 31) * 4, lrow = tid >> 5;
  {
    f4w wv[16];
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) wv[ps] = *reinterpret_cast<const f4w*>(W + (int64_t)(lrow + 8 * ps) * FB_N + lc4);
#pragma unroll
    for (int ps = 0; ps < 16; ++ps)
#pragma unroll
      for (int e = 0; e < 4; ++e) WsT[(lc4 + e) * LD + lrow + 8 * ps] = wv[ps][e];
  }
  float gcol[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) gcol[t] = g_prev[16 * t + i];
  float cg[8], cb[8], cz[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) cg[t] = cb[t] = cz[t] = 0.f;
  const int64_t ntiles = (B + FBW_ROWS - 1) / FBW_ROWS;
  f4w pd[8], pa[8];
  auto fetch_tile = [&](int64_t t) {
    const int64_t rb = t * FBW_ROWS;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int64_t r = rb + lrow + 8 * ps;
      pd[ps] = pa[ps] = f4w{0.f, 0.f, 0.f, 0.f};
      if (t < ntiles && r < B) {
        pd[ps] = *reinterpret_cast<const f4w*>(dz + r * FB_N + lc4);
        pa[ps] = *reinterpret_cast<const f4w*>(a_prev + r * FB_N + lc4);
      }
    }
  };
  fetch_tile(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * FBW_ROWS;
    __syncthreads();                             // the previous tile's output image has been read out (and WsT is staged)
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      *reinterpret_cast<f4w*>(Ds + (lrow + 8 * ps) * LD + lc4) = pd[ps];
      *reinterpret_cast<f4w*>(As + (lrow + 8 * ps) * LD + lc4) = pa[ps];
    }
    if (tid < FBW_ROWS) {
      const int64_t r = r0 + tid;
      Sst[2 * tid] = r < B ? stats_prev[2 * r] : 0.f;
      Sst[2 * tid + 1] = r < B ? stats_prev[2 * r + 1] : 0.f;
    }
    __syncthreads();
    fetch_tile(tile + gridDim.x);                // in flight during the MFMA loop and the epilogue
    f4w acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f4w{0.f, 0.f, 0.f, 0.f};
    const float* da = Ds + (16 * wave + i) * LD + 4 * kk;
    const float* wb = WsT + i * LD + 4 * kk;
#pragma unroll 2
    for (int kb = 0; kb < FB_N / 16; ++kb) {
      const f4w a0 = *reinterpret_cast<const f4w*>(da + 16 * kb);
      f4w bt[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) bt[t] = *reinterpret_cast<const f4w*>(wb + 16 * t * LD + 16 * kb);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], bt[t][r], acc[t], 0, 0, 0);
    }
    // epilogue: lane holds dY rows 16*wave + 4*kk + e, columns 16*t + i; rows are wave-private in As
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = 16 * wave + 4 * kk + e;
      const bool rok = r0 + rl < B;
      const float mean = Sst[2 * rl], rstd = Sst[2 * rl + 1];
      float av[8], xh[8], dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        av[t] = As[rl * LD + 16 * t + i];
        xh[t] = (av[t] - mean) * rstd;
        const float d = rok ? acc[t][e] : 0.f;
        dxh[t] = d * gcol[t];
        cg[t] += d * xh[t]; cb[t] += d;
        s1 += dxh[t]; s2 += dxh[t] * xh[t];
      }
      const float m1 = row16_allsum(s1) * (1.f / FB_N), m2 = row16_allsum(s2) * (1.f / FB_N);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float v = (dxh[t] - m1 - xh[t] * m2) * rstd;
        v *= av[t] > 0.f ? 1.f : av[t] + 1.f;
        v = rok ? v : 0.f;
        cz[t] += v;
        As[rl * LD + 16 * t + i] = v;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's rows of the output image are in LDS
    {
      const int c4 = (lane & 31) * 4;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int rl = 16 * wave + 2 * jj + (lane >> 5);
        const int64_t row = r0 + rl;
        const f4w v4 = *reinterpret_cast<const f4w*>(As + rl * LD + c4);
        if (row < B) *reinterpret_cast<f4w*>(dz_out + row * FB_N + c4) = v4;
      }
    }
  }
  // column partials: 16 (wave, kk) slots per column, summed in a fixed order
  __syncthreads();
  float* sh = Ds;                                 // [3][16][128] floats = 24.6 KB, the dz tile is dead
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int slot = 4 * wave + kk, col = 16 * t + i;
    sh[(0 * 16 + slot) * FB_N + col] = cg[t];
    sh[(1 * 16 + slot) * FB_N + col] = cb[t];
    sh[(2 * 16 + slot) * FB_N + col] = cz[t];
  }
  __syncthreads();
  for (int idx = tid; idx < 3 * FB_N; idx += 256) {
    const int w = idx / FB_N, col = idx % FB_N;
    float sacc = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) sacc += sh[(w * 16 + sl) * FB_N + col];
    partial[((int64_t)blockIdx.x * 3 + w) * FB_N + col] = sacc;
  }
}

// ---- the same backward step with WAVE-PRIVATE 16-row tiles (round 3): 8 waves around one shared W^T image, no barrier after
// it is staged.  A wave prefetches its next tile's dz and a rows into registers, runs dz -> LDS -> MFMA, then puts the a rows
// into the SAME 8 KB region (the dz tile is dead once the products are issued), applies the LayerNorm/ELU backward in place
// and streams the rows out.  Column partials: 32 (wave, kk) slots per column added in a fixed order.
__global__ __launch_bounds__(64 * FBW_WAVES, 1) void fused_dx_lnbwd128w_kernel(const float* __restrict__ dz, const float* __restrict__ W,
                                                                             const float* __restrict__ a_prev,
                                                                             const float* __restrict__ stats_prev,
                                                                             const float* __restrict__ g_prev, float* __restrict__ dz_out,
                                                                             float* __restrict__ partial, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) float fb_lds[];
  constexpr int LD = FB_N + 4;
  float* WsT = fb_lds;                                         // [128 c][LD]  W^T: WsT[c][n] = W[n][c]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
  float* Tw = fb_lds + FB_N * LD + wave * (16 * LD + 32);      // this wave's [16][LD] tile: dz rows, then a rows -> dz_{k-1} rows
  float* Sw = Tw + 16 * LD;                                    // [16][2] row mean / rstd
  {
    const int lc4 = (tid & 31) * 4, lrow = tid >> 5;           // 512 threads: 16 rows x 128 columns per pass, 8 passes
    f4w wv[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) wv[ps] = *reinterpret_cast<const f4w*>(W + (int64_t)(lrow + 16 * ps) * FB_N + lc4);
#pragma unroll
    for (int ps = 0; ps < 8; ++ps)
#pragma unroll
      for (int e = 0; e < 4; ++e) WsT[(lc4 + e) * LD + lrow + 16 * ps] = wv[ps][e];
  }
  float gcol[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) gcol[t] = g_prev[16 * t + i];
  float cg[8], cb[8], cz[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) cg[t] = cb[t] = cz[t] = 0.f;
  const int c4 = (lane & 31) * 4, rsub = lane >> 5;
  const int64_t ntiles = (B + 15) / 16;
  const int64_t tstride = (int64_t)gridDim.x * FBW_WAVES;
  // dz rows: prefetched ONE TILE ahead (issued before the products of the current tile); a rows and row statistics of the
  // current tile: issued at the same point, they arrive behind its 256 MFMAs
  f4w pd[8], pa[8];
  float pst = 0.f;                                             // lane l < 32: stats word l of the tile (16 rows x {mean, rstd})
  auto fetch_dz = [&](int64_t t) {
    const int64_t rb = t * 16;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int64_t r = rb + 2 * ps + rsub;
      pd[ps] = f4w{0.f, 0.f, 0.f, 0.f};
      if (t < ntiles && r < B) pd[ps] = *reinterpret_cast<const f4w*>(dz + r * FB_N + c4);
    }
  };
  auto fetch_a = [&](int64_t t) {
    const int64_t rb = t * 16;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int64_t r = rb + 2 * ps + rsub;
      pa[ps] = f4w{0.f, 0.f, 0.f, 0.f};
      if (t < ntiles && r < B) pa[ps] = *reinterpret_cast<const f4w*>(a_prev + r * FB_N + c4);
    }
    pst = 0.f;
    if (lane < 32 && t < ntiles && rb + (lane >> 1) < B) pst = stats_prev[2 * rb + lane];
  };
  int64_t tile = (int64_t)blockIdx.x * FBW_WAVES + wave;
  fetch_dz(tile);
  __syncthreads();                                             // W^T is staged (the only barrier before the final reduction)
  for (; tile < ntiles; tile += tstride) {
    const int64_t r0 = tile * 16;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) *reinterpret_cast<f4w*>(Tw + (2 * ps + rsub) * LD + c4) = pd[ps];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    fetch_a(tile);
    fetch_dz(tile + tstride);
    f4w acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f4w{0.f, 0.f, 0.f, 0.f};
    const float* da = Tw + i * LD + 4 * kk;
    const float* wb = WsT + i * LD + 4 * kk;
#pragma unroll 2
    for (int kb = 0; kb < FB_N / 16; ++kb) {
      const f4w a0 = *reinterpret_cast<const f4w*>(da + 16 * kb);
      f4w bt[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) bt[t] = *reinterpret_cast<const f4w*>(wb + 16 * t * LD + 16 * kb);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], bt[t][r], acc[t], 0, 0, 0);
    }
    // the dz tile has been read: the a rows take its place (LDS executes a wave's accesses in order)
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) *reinterpret_cast<f4w*>(Tw + (2 * ps + rsub) * LD + c4) = pa[ps];
    if (lane < 32) Sw[lane] = pst;
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = 4 * kk + e;
      const bool rok = r0 + rl < B;
      const float mean = Sw[2 * rl], rstd = Sw[2 * rl + 1];
      float av[8], xh[8], dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        av[t] = Tw[rl * LD + 16 * t + i];
        xh[t] = (av[t] - mean) * rstd;
        const float d = rok ? acc[t][e] : 0.f;
        dxh[t] = d * gcol[t];
        cg[t] += d * xh[t]; cb[t] += d;
        s1 += dxh[t]; s2 += dxh[t] * xh[t];
      }
      const float m1 = row16_allsum(s1) * (1.f / FB_N), m2 = row16_allsum(s2) * (1.f / FB_N);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float v = (dxh[t] - m1 - xh[t] * m2) * rstd;
        v *= av[t] > 0.f ? 1.f : av[t] + 1.f;
        v = rok ? v : 0.f;
        cz[t] += v;
        Tw[rl * LD + 16 * t + i] = v;
      }
      __builtin_amdgcn_sched_barrier(0);                       // one row group at a time: interleaving all four costs 50 spilled registers
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int rl = 2 * jj + rsub;
      const int64_t row = r0 + rl;
      const f4w v4 = *reinterpret_cast<const f4w*>(Tw + rl * LD + c4);
      if (row < B) *reinterpret_cast<f4w*>(dz_out + row * FB_N + c4) = v4;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
  // column partials: 32 (wave, kk) slots per column, summed in a fixed order (the tiles' LDS is dead; W^T's is reused)
  __syncthreads();
  float* sh = fb_lds;                                          // [3][32][128] floats = 48 KB <= the W^T image
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int slot = 4 * wave + kk, col = 16 * t + i;
    sh[(0 * 32 + slot) * FB_N + col] = cg[t];
    sh[(1 * 32 + slot) * FB_N + col] = cb[t];
    sh[(2 * 32 + slot) * FB_N + col] = cz[t];
  }
  __syncthreads();
  for (int idx = tid; idx < 3 * FB_N; idx += 64 * FBW_WAVES) {
    const int w = idx / FB_N, col = idx % FB_N;
    float sacc = 0.f;
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) sacc += sh[(w * 32 + sl) * FB_N + col];
    partial[((int64_t)blockIdx.x * 3 + w) * FB_N + col] = sacc;
  }
}

// The input LayerNorm's affine is folded into block 0:  (gamma o xn + beta) W_0^T + b_0 = xn (W_0 diag(gamma))^T + (b_0 + W_0 beta),
// so the network runs on xn (pre-affine) with W' = W_0 diag(gamma), b' = b_0 + W_0 beta (one tiny kernel per forward), and the
// backward never forms d(xn) = dz_0 W' over all rows (a B x H x D GEMM plus a B x D pass, whose only use was two D-vectors):
//   with dW' = dz_0^T xn (the weight-gradient GEMM that runs anyway) and db_0 = column sums of dz_0,
//   d(gamma)[c] = sum_n W_0[n][c] dW'[n][c],   d(beta)[c] = sum_n db_0[n] W_0[n][c],
//   dW_0[n][c]  = gamma[c] dW'[n][c] + beta[c] db_0[n].
__global__ void fn_fold_kernel(const float* __restrict__ W0, const float* __restrict__ b0, const float* __restrict__ gam,
                               const float* __restrict__ bet, int H, int D, float* __restrict__ Wf, float* __restrict__ bf) {
  const int n = blockIdx.x;                      // one workgroup (64 lanes) per output row
  float dot = 0.f;
  for (int c = threadIdx.x; c < D; c += 64) {
    const float w = W0[(int64_t)n * D + c];
    Wf[(int64_t)n * D + c] = w * gam[c];
    dot = fmaf(w, bet[c], dot);
  }
  dot = wave_sum_all(dot);
  if (threadIdx.x == 0) bf[n] = b0[n] + dot;
}
__global__ void fn_unfold_grad_kernel(const float* __restrict__ W0, const float* __restrict__ gam, const float* __restrict__ bet,
                                      float* __restrict__ dW, const float* __restrict__ db0, int H, int D, float* __restrict__ dg,
                                      float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  const float gc = gam[c], bc = bet[c];
  float sg = 0.f, sb = 0.f;
  // 16 rows of loads in flight per step (one thread walks all H rows of its column: serial latency otherwise, 50 us at
  // H = 128); the sums keep their row order
  for (int n0 = 0; n0 < H; n0 += 16) {
    float w[16], dwp[16], dbn[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int n = n0 + u < H ? n0 + u : H - 1;
      w[u] = W0[(int64_t)n * D + c]; dwp[u] = dW[(int64_t)n * D + c]; dbn[u] = db0[n];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (n0 + u < H) {
        sg = fmaf(w[u], dwp[u], sg);
        sb = fmaf(dbn[u], w[u], sb);
        dW[(int64_t)(n0 + u) * D + c] = fmaf(gc, dwp[u], bc * dbn[u]);
      }
  }
  dg[c] = sg; dbeta[c] = sb;
}

// grid (ceil(D/64), 3), 1024 threads = 64 columns x 16 strided slices of the block list, combined in a fixed order
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ partial, int nblocks, int D, float* o0,
                                                             float* o1, float* o2) {
  __shared__ float sh[16][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6, w = blockIdx.y;
  const int j = blockIdx.x * 64 + c;
  float s = 0.f;
  if (j < D)
    for (int b = q; b < nblocks; b += 16) s += partial[((int64_t)b * 3 + w) * D + j];
  sh[q][c] = s;
  __syncthreads();
  float* o = w == 0 ? o0 : w == 1 ? o1 : o2;
  if (q == 0 && j < D && o) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][c];
    o[j] = t;
  }
}

// head: out[B,O] += bias ; db[O] = column sums of dout (second form)
__global__ void add_bias_kernel(float* __restrict__ out, const float* __restrict__ b, int64_t n, int O) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += b[i % O];
}
// column sums of d[B, O] (O <= 64): grid (O, nslices) -> partial[slice][O], then a fixed-order finish
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ d, int64_t B, int O, int nslices,
                                                           float* __restrict__ partial) {
  __shared__ float sh[256];
  const int o = blockIdx.x, sl = blockIdx.y;
  const int64_t per = (B + nslices - 1) / nslices;
  const int64_t lo = sl * per, hi = lo + per < B ? lo + per : B;
  float s = 0.f;
  for (int64_t r = lo + threadIdx.x; r < hi; r += 256) s += d[r * O + o];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[sl * O + o] = sh[0];
}
__global__ void colsum_small_finish_kernel(const float* __restrict__ partial, int O, int nslices, float* __restrict__ out) {
  const int o = threadIdx.x;
  if (o >= O) return;
  float s = 0.f;
  for (int b = 0; b < nslices; ++b) s += partial[b * O + o];
  out[o] = s;
}

int grid_rows(int64_t B) {
  int64_t g = (B + 3) / 4;
  return (int)(g < 1 ? 1 : g > 1024 ? 1024 : g);
}

constexpr float LOG_SQRT_2PI_F = 0.91893853320467274178f;

// ---------------------------------------------------------------- Gaussian head: sample / evaluate
// std = sigmoid(log_std / xc) * yc (distributions.py:41); log_probs are PER DIMENSION (FixedNormal.log_probs)
__global__ void ma_sample_kernel(const float* __restrict__ mean, const float* __restrict__ log_std, const float* __restrict__ eps,
                                 float xc, float yc, int deterministic, float* __restrict__ act, float* __restrict__ logp,
                                 int64_t n, int A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = (int)(i % A);
  const float sd = yc / (1.f + expf(-log_std[a] / xc));
  const float mu = mean[i];
  const float x = deterministic ? mu : mu + sd * eps[i];
  act[i] = x;
  const float d = x - mu;
  logp[i] = -(d * d) / (2.f * sd * sd) - logf(sd) - LOG_SQRT_2PI_F;
}
__global__ void ma_logp_kernel(const float* __restrict__ mean, const float* __restrict__ log_std, const float* __restrict__ act,
                               float xc, float yc, float* __restrict__ logp, int64_t n, int A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = (int)(i % A);
  const float sd = yc / (1.f + expf(-log_std[a] / xc));
  const float d = act[i] - mean[i];
  logp[i] = -(d * d) / (2.f * sd * sd) - logf(sd) - LOG_SQRT_2PI_F;
}

// ---------------------------------------------------------------- actor loss epilogue (mappolag.py:150-176)
// per row: imp = prod_a exp(logp_a - old_a); hybrid advantage adv - lambda*cost_adv; clipped surrogate times the HAPPO
// factor; entropy bonus; the lambda-delta term mean(imp * cost_adv).  Writes d(loss)/d(mean) and per-block partials
// {sum factor*min*w, sum imp, sum imp*cost_adv, sum active, d(log_std)[A]}.
constexpr int AL_NS = 4;       // scalar partials
__global__ __launch_bounds__(256) void ma_actor_loss_kernel(
    const float* __restrict__ mean, const float* __restrict__ log_std, const float* __restrict__ act,
    const float* __restrict__ old_logp, const float* __restrict__ adv, const float* __restrict__ cost_adv,
    const float* __restrict__ factor, const float* __restrict__ active, const float* __restrict__ lamda_dev,
    spo_ma_loss_cfg c, float inv_denom, float ent_w, float* __restrict__ dmean, double* __restrict__ partial,
    int64_t B, int A) {
  __shared__ double sh[AL_NS + 64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sd[SPO_MAX_ACT], dsd_dls[SPO_MAX_ACT];
  for (int a = 0; a < A; ++a) {
    const float s = 1.f / (1.f + expf(-log_std[a] / c.std_x_coef));
    sd[a] = s * c.std_y_coef;
    dsd_dls[a] = c.std_y_coef * s * (1.f - s) / c.std_x_coef;
  }
  const float lamda = *lamda_dev;
  double acc[AL_NS] = {0, 0, 0, 0};
  double dls[SPO_MAX_ACT];
  for (int a = 0; a < A; ++a) dls[a] = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < B; r += (int64_t)gridDim.x * 256) {
    float imp = 1.f;
    float dif[SPO_MAX_ACT], impd[SPO_MAX_ACT];
    for (int a = 0; a < A; ++a) {
      dif[a] = act[r * A + a] - mean[r * A + a];
      const float lp = -(dif[a] * dif[a]) / (2.f * sd[a] * sd[a]) - logf(sd[a]) - LOG_SQRT_2PI_F;
      impd[a] = expf(lp - old_logp[r * A + a]);
      imp *= impd[a];                                        // torch.exp per dim, then torch.prod
    }
    const float advh = adv[r] - lamda * cost_adv[r];
    const float lo = 1.f - c.clip_param, hi = 1.f + c.clip_param;
    if (c.per_dim_ratio) {
      // MAPPO (mappo.py:150-160): the ratio stays per action dimension; min(surr1, surr2) is summed over the dimensions
      const float w = (c.use_policy_active_masks ? active[r] : 1.f) * inv_denom;
      const float ew = c.use_policy_active_masks ? active[r] * inv_denom : ent_w;
      float msum = 0.f;
      for (int a = 0; a < A; ++a) {
        const float ia = impd[a];
        const float rca = fminf(fmaxf(ia, lo), hi);
        const float s1 = ia * advh, s2 = rca * advh;
        const bool inr = ia >= lo && ia <= hi;
        float gr;
        if (s1 < s2) gr = advh;
        else if (s1 > s2) gr = inr ? advh : 0.f;
        else gr = 0.5f * advh + (inr ? 0.5f * advh : 0.f);
        msum += fminf(s1, s2);
        acc[1] += (double)ia;
        const float dlp = -w * gr * ia;
        const float iv = 1.f / (sd[a] * sd[a]);
        dmean[r * A + a] = dlp * dif[a] * iv;
        const float dsig = dlp * (dif[a] * dif[a] * iv / sd[a] - 1.f / sd[a]) - c.entropy_coef * ew / sd[a];
        dls[a] += (double)(dsig * dsd_dls[a]);
      }
      acc[0] += (double)(msum * w);
      acc[3] += (double)active[r];
      continue;
    }
    const float rc = fminf(fmaxf(imp, lo), hi);
    const float s1 = imp * advh, s2 = rc * advh;
    const bool inr = imp >= lo && imp <= hi;
    float gr;                                                 // d min(s1, s2) / d imp
    if (s1 < s2) gr = advh;
    else if (s1 > s2) gr = inr ? advh : 0.f;
    else gr = 0.5f * advh + (inr ? 0.5f * advh : 0.f);
    const float w = (c.use_policy_active_masks ? active[r] : 1.f) * inv_denom;
    const float f = factor[r];
    acc[0] += (double)(f * fminf(s1, s2) * w);
    acc[1] += (double)imp;
    acc[2] += (double)(imp * cost_adv[r]);
    acc[3] += (double)active[r];
    const float dimp = -f * w * gr;                           // d(policy_action_loss) / d imp
    const float dlp = dimp * imp;                             // every dimension's log-prob enters imp the same way
    const float ew = c.use_policy_active_masks ? active[r] * inv_denom : ent_w;      // entropy weight of this row
    for (int a = 0; a < A; ++a) {
      const float iv = 1.f / (sd[a] * sd[a]);
      dmean[r * A + a] = dlp * dif[a] * iv;
      // d logp / d sigma = dif^2 / sigma^3 - 1 / sigma ; entropy_a = 0.5 + log(sqrt(2 pi)) + log sigma
      const float dsig = dlp * (dif[a] * dif[a] * iv / sd[a] - 1.f / sd[a]) - c.entropy_coef * ew / sd[a];
      dls[a] += (double)(dsig * dsd_dls[a]);
    }
  }
  for (int k = 0; k < AL_NS + A; ++k) {
    double v = k < AL_NS ? acc[k] : dls[k - AL_NS];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) sh[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < AL_NS + A)
    partial[(int64_t)blockIdx.x * (AL_NS + SPO_MAX_ACT) + threadIdx.x] =
        (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// Fixed-order sum of partial[b * stride + offset], b = 0..n-1, by one 256-thread workgroup (strided per-thread sums, then a
// shared-memory tree): the single-thread loops this replaces cost 85-300 us per call at 1024 partials.  Result in thread 0.
__device__ __forceinline__ double block_sum_partials(const double* __restrict__ partial, int n, int stride, int offset) {
  __shared__ double shp[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < n; b += 256) s += partial[(int64_t)b * stride + offset];
  shp[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) shp[threadIdx.x] += shp[threadIdx.x + k];
    __syncthreads();
  }
  const double r = shp[0];
  __syncthreads();
  return r;
}

// scalars_out: {policy_loss, dist_entropy, mean(imp), mean(imp*cost_adv), sum(active)}; dlogstd_out[A]
__global__ void ma_actor_loss_finish_kernel(const double* __restrict__ partial, int nblocks, const float* __restrict__ log_std,
                                            spo_ma_loss_cfg c, int64_t B /* rows of the GLOBAL batch */, int A,
                                            float* __restrict__ scalars_out, float* __restrict__ dlogstd_out) {
  const int k = blockIdx.x;                       // one workgroup per output
  const double s = block_sum_partials(partial, nblocks, AL_NS + SPO_MAX_ACT, k);
  if (threadIdx.x != 0) return;
  if (k == 0) {
    scalars_out[0] = (float)(-s);
    // entropy of a state-independent sigma is the same in every row
    double ent = 0.0;
    for (int a = 0; a < A; ++a) {
      const float sg = c.std_y_coef / (1.f + expf(-log_std[a] / c.std_x_coef));
      ent += 0.5 + (double)LOG_SQRT_2PI_F + (double)logf(sg);
    }
    scalars_out[1] = (float)(c.use_policy_active_masks ? ent : ent / (double)A);   // (ent*mask).sum()/mask.sum() vs .mean()
  } else if (k == 1) scalars_out[2] = (float)(s / ((double)B * (c.per_dim_ratio ? (double)A : 1.0)));
  else if (k == 2) scalars_out[3] = (float)(s / (double)B);
  else if (k == 3) scalars_out[4] = (float)s;
  else dlogstd_out[k - AL_NS] = (float)s;
}

// lamda <- relu(lamda - delta*rate), delta = -((aver_cost - limit)*(1-gamma) + mean(imp*cost_adv))  (mappolag.py:178-182)
__global__ void ma_lamda_update_kernel(float* lamda, const float* scalars, float aver_episode_cost, float cost_limit,
                                       float gamma, float rate) {
  const float delta = -((aver_episode_cost - cost_limit) * (1.f - gamma) + scalars[3]);
  const float nl = *lamda - delta * rate;
  *lamda = nl > 0.f ? nl : 0.f;
}

// ---------------------------------------------------------------- PopArt (popart.py:86-112) on a [B] vector
// state = {running_mean, running_mean_sq, debiasing_term}
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t B, double* __restrict__ partial) {
  __shared__ double sh[2][4];
  double s = 0, q = 0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < B; r += (int64_t)gridDim.x * 256) {
    const double v = x[r];
    s += v; q += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    partial[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}
__global__ void sumsq_finish_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ sums2) {
  const double s = block_sum_partials(partial, nblocks, 2, blockIdx.x);          // grid 2: sum x, sum x^2
  if (threadIdx.x == 0) sums2[blockIdx.x] = s;
}
__global__ void popart_update_kernel(const double* __restrict__ sums2, int64_t B, float beta, float omb, float* state) {
  const double s = sums2[0], q = sums2[1];
  const float bm = (float)(s / (double)B), bq = (float)(q / (double)B);
  // omb = (1.0 - beta) formed in double on the host, as the reference's python float arithmetic does (popart.py:104-106):
  // 1.f - 0.99999f would be 1.00136e-5
  state[0] = state[0] * beta + bm * omb;
  state[1] = state[1] * beta + bq * omb;
  state[2] = state[2] * beta + 1.f * omb;
}
__global__ void popart_normalize_kernel(const float* __restrict__ x, const float* __restrict__ state, float eps,
                                        float* __restrict__ out, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const float deb = fmaxf(state[2], eps);
  const float m = state[0] / deb, msq = state[1] / deb;
  const float var = fmaxf(msq - m * m, 1e-2f);
  out[i] = (x[i] - m) / sqrtf(var);
}

// ---------------------------------------------------------------- value loss (mappolag.py:126-138, util.huber_loss)
__device__ __forceinline__ float huber(float e, float d) {
  const float a = fabsf(e) <= d ? 1.f : 0.f, b = e > d ? 1.f : 0.f;            // (sic) nothing for e < -d
  return a * e * e / 2.f + b * d * (fabsf(e) - d / 2.f);
}
__device__ __forceinline__ float huber_grad(float e, float d) {
  return fabsf(e) <= d ? e : (e > d ? d : 0.f);
}
__global__ __launch_bounds__(256) void ma_value_loss_kernel(const float* __restrict__ values, const float* __restrict__ value_preds,
                                                            const float* __restrict__ ret_n1, const float* __restrict__ ret_n2,
                                                            const float* __restrict__ active, float clip, float delta, float coef_over_B,
                                                            float* __restrict__ dvalues, double* __restrict__ partial, int64_t B) {
  __shared__ double sh[4];
  double s = 0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < B; r += (int64_t)gridDim.x * 256) {
    const float v = values[r], vp = value_preds[r];
    const float dv = v - vp;
    const float vc = vp + fminf(fmaxf(dv, -clip), clip);
    const float ec = ret_n1[r] - vc, eo = ret_n2[r] - v;
    const float hc = huber(ec, delta), ho = huber(eo, delta);
    const float aw = active ? active[r] : 1.f;                // happo.py:117-120: (loss * active).sum() / active.sum()
    s += (double)(fmaxf(ho, hc) * aw);
    const float go = -huber_grad(eo, delta);
    const float gc = (dv >= -clip && dv <= clip) ? -huber_grad(ec, delta) : 0.f;
    float g;
    if (ho > hc) g = go;
    else if (ho < hc) g = gc;
    else g = 0.5f * (go + gc);                               // torch.max splits ties
    dvalues[r] = g * coef_over_B * aw;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void sum_finish_kernel(const double* partial, int n, double scale, float* out) {
  const double s = block_sum_partials(partial, n, 1, 0);
  if (threadIdx.x == 0) *out = (float)(s * scale);
}

// ---------------------------------------------------------------- clip_grad_norm_ + Adam on one flat vector
__global__ __launch_bounds__(256) void sq_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
  __shared__ double sh[4];
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += (double)g[i] * g[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void norm_finish_kernel(const double* partial, int n, float* norm_out) {
  const double s = block_sum_partials(partial, n, 1, 0);
  if (threadIdx.x == 0) *norm_out = (float)sqrt(s);
}
__global__ void ma_adam_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                               float* __restrict__ v, int64_t n, const float* __restrict__ norm, float max_norm, int use_clip,
                               float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float coef = 1.f;
  if (use_clip) {
    coef = max_norm / (*norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
  }
  float g = grad[i] * coef;
  const float p = theta[i];
  if (wd != 0.f) g += wd * p;
  const float mm = m[i] + (1.f - b1) * (g - m[i]);                       // exp_avg.lerp_
  const float vv = v[i] * b2 + (1.f - b2) * g * g;
  m[i] = mm; v[i] = vv;
  const float denom = sqrtf(vv) / bc2_sqrt + eps;
  theta[i] = p - (lr / bc1) * (mm / denom);
}

}  // namespace

// ================================================================= C ABI
extern "C" int64_t spo_ma_param_count(const spo_ma_net* net) {
  Lay L;
  if (lay_of(net, &L)) return -1;
  return L.count();
}
extern "C" int64_t spo_ma_workspace_floats(const spo_ma_net* net, int64_t rows) {
  Lay L;
  if (lay_of(net, &L) || rows < 1) return -1;
  return L.ws_count(rows);
}
extern "C" int64_t spo_ma_param_offset(const spo_ma_net* net, int which, int block) {
  Lay L;
  if (lay_of(net, &L)) return -1;
  switch (which) {
    case 0: return L.fn_g();
    case 1: return L.fn_b();
    case 2: return block >= 0 && block < L.NB ? L.W(block) : -1;
    case 3: return block >= 0 && block < L.NB ? L.b(block) : -1;
    case 4: return block >= 0 && block < L.NB ? L.g(block) : -1;
    case 5: return block >= 0 && block < L.NB ? L.be(block) : -1;
    case 6: return L.actor ? L.logstd() : -1;
    case 7: return L.hW();
    case 8: return L.hb();
    default: return -1;
  }
}

extern "C" int spo_ma_forward(const float* theta, const spo_ma_net* net, const float* x, int64_t rows, float* ws,
                              float* out, void* stream) {
  Lay L;
  if (int rc = lay_of(net, &L)) return rc;
  SPO_REQUIRE(theta && x && ws && out && rows > 0, "ma_forward: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = rows;
  const int gr = grid_rows(B);
  const bool x_vec = (L.D % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
  if (x_vec && L.D <= 64)
    hipLaunchKernelGGL(ln_fwd_narrow_kernel<16>, dim3(gr), dim3(256), 0, st, x, theta + L.fn_g(), theta + L.fn_b(),
                       ws + L.ws_xhat(), ws + L.ws_st0(B), B, L.D);
  else if (x_vec && L.D <= 128)
    hipLaunchKernelGGL(ln_fwd_narrow_kernel<32>, dim3(gr), dim3(256), 0, st, x, theta + L.fn_g(), theta + L.fn_b(),
                       ws + L.ws_xhat(), ws + L.ws_st0(B), B, L.D);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<0>, dim3(gr), dim3(256), 0, st, x, nullptr, theta + L.fn_g(), theta + L.fn_b(), nullptr,
                       ws + L.ws_xhat(), ws + L.ws_st0(B), B, L.D);
  hipLaunchKernelGGL(fn_fold_kernel, dim3(L.H), dim3(64), 0, st, theta + L.W(0), theta + L.b(0), theta + L.fn_g(), theta + L.fn_b(),
                     L.H, L.D, ws + L.ws_w0f(B), ws + L.ws_b0f(B));
  const float* in = ws + L.ws_xhat();
  for (int k = 0; k < L.NB; ++k) {
    float* a = ws + L.ws_a(B, k);
    const float* Wk = k == 0 ? ws + L.ws_w0f(B) : theta + L.W(k);
    const float* bk = k == 0 ? ws + L.ws_b0f(B) : theta + L.b(k);
    if (L.H == 128 && L.in_k(k) % 4 == 0 && L.in_k(k) <= 128 && B >= ma_fuse_min_rows()) {
      // one fused MFMA kernel (collect-size batches upwards; below that the plain MFMA GEMM + LayerNorm kernel)
      const int K = L.in_k(k), KP = (K + 15) & ~15;
      const size_t sh = ((size_t)128 * (KP + 4) + 128 * 132 + 256) * sizeof(float);       // W image + max(X tile, output image)
      static bool attr_done_dev[spo::SPO_MAX_DEVICES] = {};
      bool& attr_done = attr_done_dev[spo::current_device_slot()];
      if (!attr_done) {
        if (int rc = spo::hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_block_fwd128_kernel<2>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 128 * 132 + 256) * 4),
                                    "hipFuncSetAttribute(fused_block_fwd128)")) return rc;
        if (int rc = spo::hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_block_fwd128_kernel<1>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 128 * 132 + 256) * 4),
                                    "hipFuncSetAttribute(fused_block_fwd128)")) return rc;
        attr_done = true;
      }
      if (B >= 256 * 128 && ma_fwd_wave_private()) {
        constexpr size_t shw = ((size_t)FB_N * FB_SLD + FBW_WAVES * (16 * FB_SLD + 32)) * sizeof(float);
        static bool attr_w_dev[spo::SPO_MAX_DEVICES] = {};
        bool& attr_w = attr_w_dev[spo::current_device_slot()];
        if (!attr_w) {
          if (int rc = spo::hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_block_fwd128w_kernel),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shw),
                                      "hipFuncSetAttribute(fused_block_fwd128w)")) return rc;
          attr_w = true;
        }
        hipLaunchKernelGGL(fused_block_fwd128w_kernel, dim3(256), dim3(64 * FBW_WAVES), shw, st, in, Wk, bk, theta + L.g(k),
                           theta + L.be(k), a, ws + L.ws_y(B, k), ws + L.ws_st(B, k), B, K);
      } else if (B >= 256 * 128) {
        const int64_t nt = (B + 127) / 128;
        hipLaunchKernelGGL(fused_block_fwd128_kernel<2>, dim3((unsigned)(nt < 256 ? nt : 256)), dim3(256), sh, st, in, Wk,
                           bk, theta + L.g(k), theta + L.be(k), a, ws + L.ws_y(B, k), ws + L.ws_st(B, k), B, K);
      } else {
        const int64_t nt = (B + 63) / 64;
        hipLaunchKernelGGL(fused_block_fwd128_kernel<1>, dim3((unsigned)(nt < 256 ? nt : 256)), dim3(256), sh, st, in, Wk,
                           bk, theta + L.g(k), theta + L.be(k), a, ws + L.ws_y(B, k), ws + L.ws_st(B, k), B, K);
      }
      in = ws + L.ws_y(B, k);
      continue;
    }
    if (int rc = gemm_xwT(st, in, Wk, a, B, L.in_k(k), L.H)) return rc;
    if (L.H == 128)
      hipLaunchKernelGGL(ln_fwd128_kernel<1>, dim3(gr), dim3(256), 0, st, a, bk, theta + L.g(k), theta + L.be(k), a,
                         ws + L.ws_y(B, k), ws + L.ws_st(B, k), B);
    else
      hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3(gr), dim3(256), 0, st, a, bk, theta + L.g(k), theta + L.be(k), a,
                         ws + L.ws_y(B, k), ws + L.ws_st(B, k), B, L.H);
    in = ws + L.ws_y(B, k);
  }
  if (L.O <= HS_MAXN && L.H <= HS_MAXK && !g_ma_gemm_rocblas) {
    hipLaunchKernelGGL(head_small_kernel, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, st, in, theta + L.hW(), theta + L.hb(), out,
                       B, L.H, L.O);
  } else {
    if (int rc = gemm_xwT(st, in, theta + L.hW(), out, B, L.H, L.O)) return rc;
    const int64_t n = B * L.O;
    hipLaunchKernelGGL(add_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, theta + L.hb(), n, L.O);
  }
  SPO_LAUNCH_CHECK("spo_ma_forward");
  return 0;
}

// grad: flat, same layout as theta (every entry written, log_std left untouched: the loss kernel owns it).
// scratch: float[2 * rows * max(hidden, in_dim)] + partial column sums float[1024 * 3 * max(hidden, in_dim)]
extern "C" int64_t spo_ma_backward_scratch_floats(const spo_ma_net* net, int64_t rows) {
  Lay L;
  if (lay_of(net, &L) || rows < 1) return -1;
  const int64_t W = L.H > L.D ? L.H : L.D;
  int64_t slices = (int64_t)dw_splits(rows, L.H, L.H) * L.H * L.H;
  const int64_t s0 = (int64_t)dw_splits(rows, L.H, L.D) * L.H * L.D, s1 = (int64_t)dw_splits(rows, L.O, L.H) * L.O * L.H;
  if (s0 > slices) slices = s0;
  if (s1 > slices) slices = s1;
  return 2 * rows * W + 1024 * 3 * W + slices;
}
extern "C" int spo_ma_backward(const float* theta, const spo_ma_net* net, const float* x, int64_t rows, const float* ws,
                               const float* dout, float* grad, float* scratch, void* stream) {
  Lay L;
  if (int rc = lay_of(net, &L)) return rc;
  SPO_REQUIRE(theta && x && ws && dout && grad && scratch && rows > 0, "ma_backward: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = rows;
  const int64_t Wd = L.H > L.D ? L.H : L.D;
  float* d0 = scratch;                 // [B, Wd] gradient wrt the current block's output
  float* d1 = scratch + B * Wd;        // [B, Wd] dz of the current block
  float* partial = scratch + 2 * B * Wd;
  float* slices = partial + 1024 * 3 * Wd;
  const int gr = grid_rows(B);
  // dz of the top block from the head's dX; below it, for hidden 128 at large batch, one fused kernel per block turns
  // dz_k into dz_{k-1} (dX GEMM + LayerNorm/ELU backward + column partials), otherwise the MFMA GEMM + the LayerNorm kernel.
  const bool fuse_bwd = (L.H == 128) && (B >= ma_fuse_min_rows());
  float* dzc = d1;                     // dz of the current block
  float* other = d0;                   // free buffer (holds dY of the current block until its LayerNorm backward ran)
  int nparts = gr;
  const float* y_last = ws + L.ws_y(B, L.NB - 1);
  const bool fuse_head = fuse_bwd && L.O <= 16 && !g_ma_gemm_rocblas && ma_fuse_head();
  if (fuse_head) {
    // head weight / bias gradients, the head's dX and the top block's LayerNorm/ELU backward in one pass over the rows
    const int k = L.NB - 1;
    const int64_t cap = (int64_t)dw_splits(B, L.H, L.H) * L.H * L.H / ((int64_t)L.O * (L.H + 1));      // `slices` holds the partials
    const int nb = (int)(cap < gr ? (cap < 1 ? 1 : cap) : gr);
    float* hpw = slices;
    float* hpb = slices + (int64_t)nb * L.O * L.H;
#define SPO_HEAD_BWD(OM)                                                                                                  \
    hipLaunchKernelGGL(head_bwd_lnbwd128_kernel<OM>, dim3(nb), dim3(256), 0, st, dout, theta + L.hW(), ws + L.ws_a(B, k),  \
                       ws + L.ws_st(B, k), theta + L.g(k), theta + L.be(k), d1, partial, hpw, hpb, B, L.O)
    if (L.O == 1) SPO_HEAD_BWD(1);
    else if (L.O <= 4) SPO_HEAD_BWD(4);
    else if (L.O <= 6) SPO_HEAD_BWD(6);
    else if (L.O <= 8) SPO_HEAD_BWD(8);
    else SPO_HEAD_BWD(16);
#undef SPO_HEAD_BWD
    hipLaunchKernelGGL(head_dw_finish_kernel, dim3(L.O, 4), dim3(512), 0, st, hpw, hpb, nb, L.O, grad + L.hW(), grad + L.hb());
    nparts = nb;
  } else {
  // head
  if (int rc = gemm_dyTx(st, dout, y_last, grad + L.hW(), B, L.H, L.O, slices)) return rc;
  {
    const int ns = (int)(B / 4096 < 1 ? 1 : B / 4096 > 256 ? 256 : B / 4096);
    hipLaunchKernelGGL(colsum_small_kernel, dim3(L.O, ns), dim3(256), 0, st, dout, B, L.O, ns, partial);
    hipLaunchKernelGGL(colsum_small_finish_kernel, dim3(1), dim3(64), 0, st, partial, L.O, ns, grad + L.hb());
  }
  if (int rc = gemm_dyw(st, dout, theta + L.hW(), d0, B, L.H, L.O)) return rc;
  {
    const int k = L.NB - 1;
    if (L.H == 128)
      hipLaunchKernelGGL(ln_bwd128_kernel<1>, dim3(gr), dim3(256), 0, st, d0, ws + L.ws_a(B, k), ws + L.ws_st(B, k), theta + L.g(k),
                         d1, partial, B);
    else
      hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(gr), dim3(256), 0, st, d0, ws + L.ws_a(B, k), ws + L.ws_st(B, k), theta + L.g(k),
                         d1, partial, B, L.H);
  }
  }
  for (int k = L.NB - 1; k >= 0; --k) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((L.H + 63) / 64, 3), dim3(1024), 0, st, partial, nparts, L.H, grad + L.g(k),
                       grad + L.be(k), grad + L.b(k));
    const float* in = k == 0 ? ws + L.ws_xhat() : ws + L.ws_y(B, k - 1);
    if (int rc = gemm_dyTx(st, dzc, in, grad + L.W(k), B, L.in_k(k), L.H, slices)) return rc;
    if (k >= 1 && fuse_bwd) {
      const size_t sh = ((size_t)128 * 132 + 2 * 64 * 132 + 128) * sizeof(float);
      static bool attr_done_dev[spo::SPO_MAX_DEVICES] = {};
      bool& attr_done = attr_done_dev[spo::current_device_slot()];
      if (!attr_done) {
        if (int rc = spo::hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dx_lnbwd128_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh),
                                    "hipFuncSetAttribute(fused_dx_lnbwd128)")) return rc;
        attr_done = true;
      }
      if (B >= 256 * 128 && ma_fwd_wave_private()) {
        constexpr size_t shw = ((size_t)FB_N * FB_SLD + FBW_WAVES * (16 * FB_SLD + 32)) * sizeof(float);
        static bool attr_w_dev[spo::SPO_MAX_DEVICES] = {};
        bool& attr_w = attr_w_dev[spo::current_device_slot()];
        if (!attr_w) {
          if (int rc = spo::hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dx_lnbwd128w_kernel),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shw),
                                      "hipFuncSetAttribute(fused_dx_lnbwd128w)")) return rc;
          attr_w = true;
        }
        nparts = 256;
        hipLaunchKernelGGL(fused_dx_lnbwd128w_kernel, dim3(nparts), dim3(64 * FBW_WAVES), shw, st, dzc, theta + L.W(k),
                           ws + L.ws_a(B, k - 1), ws + L.ws_st(B, k - 1), theta + L.g(k - 1), other, partial, B);
      } else {
      const int64_t nt = (B + FBW_ROWS - 1) / FBW_ROWS;
      nparts = (int)(nt < 256 ? nt : 256);
      hipLaunchKernelGGL(fused_dx_lnbwd128_kernel, dim3(nparts), dim3(256), sh, st, dzc, theta + L.W(k), ws + L.ws_a(B, k - 1),
                         ws + L.ws_st(B, k - 1), theta + L.g(k - 1), other, partial, B);
      }
      float* t = dzc; dzc = other; other = t;
    } else if (k >= 1) {
      if (int rc = gemm_dyw(st, dzc, theta + L.W(k), other, B, L.in_k(k), L.H)) return rc;      // dY of block k-1
      {
        nparts = gr;
        if (L.H == 128)
          hipLaunchKernelGGL(ln_bwd128_kernel<1>, dim3(gr), dim3(256), 0, st, other, ws + L.ws_a(B, k - 1), ws + L.ws_st(B, k - 1),
                             theta + L.g(k - 1), dzc, partial, B);
        else
          hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(gr), dim3(256), 0, st, other, ws + L.ws_a(B, k - 1), ws + L.ws_st(B, k - 1),
                             theta + L.g(k - 1), dzc, partial, B, L.H);
      }
    }
  }
  // grad + W(0) holds dW' = dz_0^T xn: turn it into dW_0 and read the input LayerNorm's two gradients off it
  hipLaunchKernelGGL(fn_unfold_grad_kernel, dim3((L.D + 63) / 64), dim3(64), 0, st, theta + L.W(0), theta + L.fn_g(), theta + L.fn_b(),
                     grad + L.W(0), grad + L.b(0), L.H, L.D, grad + L.fn_g(), grad + L.fn_b());
  SPO_LAUNCH_CHECK("spo_ma_backward");
  return 0;
}

// ---------------------------------------------------------------- tangent (forward-mode) pass
// d(out)/d(theta) . t at the point kept in ws by spo_ma_forward: the J v of MACPO's Fisher-vector product
// (macpo.py:187-199 differentiates the KL twice; at theta = theta_old that Hessian is J^T M J, see safepo/multi_agent/macpo.py).
// Per block: dz = y_{k-1} tW^T + dy_{k-1} W^T + tb;  da = ELU'(a) dz;  xhat = (a - mean) rstd;
//            dxhat = rstd (da - mean(da) - xhat mean(xhat da));  dy = tg xhat + g dxhat + tbe.
__global__ void fn_fold_tangent_kernel(const float* __restrict__ W0, const float* __restrict__ gam, const float* __restrict__ bet,
                                       const float* __restrict__ tW0, const float* __restrict__ tb0, const float* __restrict__ tgam,
                                       const float* __restrict__ tbet, int H, int D, float* __restrict__ tWf, float* __restrict__ tbf) {
  const int n = blockIdx.x;                      // W' = W_0 diag(gamma), b' = b_0 + W_0 beta  ->  their tangents
  float dot = 0.f;
  for (int c = threadIdx.x; c < D; c += 64) {
    const float w = W0[(int64_t)n * D + c], tw = tW0[(int64_t)n * D + c];
    tWf[(int64_t)n * D + c] = tw * gam[c] + w * tgam[c];
    dot += tw * bet[c] + w * tbet[c];
  }
  dot = wave_sum_all(dot);
  if (threadIdx.x == 0) tbf[n] = tb0[n] + dot;
}
// one wave per row; dz is overwritten with dy
__global__ __launch_bounds__(256) void ln_jvp_kernel(float* __restrict__ dz, const float* __restrict__ tb, const float* __restrict__ a,
                                                      const float* __restrict__ stats, const float* __restrict__ g,
                                                      const float* __restrict__ tg, const float* __restrict__ tbe, int64_t B, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < B; row += (int64_t)gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float da[MAXE], xh[MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int j = lane + 64 * e;
      da[e] = xh[e] = 0.f;
      if (j < D) {
        const float av = a[row * D + j];
        da[e] = (dz[row * D + j] + tb[j]) * (av > 0.f ? 1.f : av + 1.f);
        xh[e] = (av - mean) * rstd;
      }
      s1 += da[e]; s2 += da[e] * xh[e];
    }
    const float m1 = wave_sum_all(s1) / (float)D, m2 = wave_sum_all(s2) / (float)D;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int j = lane + 64 * e;
      if (j < D) dz[row * D + j] = tg[j] * xh[e] + g[j] * (rstd * (da[e] - m1 - xh[e] * m2)) + tbe[j];
    }
  }
}

extern "C" int64_t spo_ma_jvp_scratch_floats(const spo_ma_net* net, int64_t rows) {
  Lay L;
  if (lay_of(net, &L) || rows < 1) return -1;
  return 2 * Lay::al4(rows * L.H) + Lay::al4((int64_t)L.H * L.D) + Lay::al4(L.H);
}
extern "C" int spo_ma_jvp(const float* theta, const spo_ma_net* net, const float* tangent, int64_t rows, const float* ws,
                          float* dout, float* scratch, void* stream) {
  Lay L;
  if (int rc = lay_of(net, &L)) return rc;
  SPO_REQUIRE(theta && tangent && ws && dout && scratch && rows > 0, "ma_jvp: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t B = rows;
  const int gr = grid_rows(B);
  float* bufs[2] = {scratch, scratch + Lay::al4(B * L.H)};
  float* tWf = scratch + 2 * Lay::al4(B * L.H);
  float* tbf = tWf + Lay::al4((int64_t)L.H * L.D);
  hipLaunchKernelGGL(fn_fold_tangent_kernel, dim3(L.H), dim3(64), 0, st, theta + L.W(0), theta + L.fn_g(), theta + L.fn_b(),
                     tangent + L.W(0), tangent + L.b(0), tangent + L.fn_g(), tangent + L.fn_b(), L.H, L.D, tWf, tbf);
  const float* dy_prev = nullptr;
  for (int k = 0; k < L.NB; ++k) {
    float* dz = bufs[k & 1];
    const float* y_prev = k == 0 ? ws + L.ws_xhat() : ws + L.ws_y(B, k - 1);
    if (int rc = gemm_xwT(st, y_prev, k == 0 ? tWf : tangent + L.W(k), dz, B, L.in_k(k), L.H)) return rc;
    if (k > 0)
      if (int rc = gemm_xwT(st, dy_prev, theta + L.W(k), dz, B, L.H, L.H, 1.f)) return rc;
    hipLaunchKernelGGL(ln_jvp_kernel, dim3(gr), dim3(256), 0, st, dz, k == 0 ? tbf : tangent + L.b(k), ws + L.ws_a(B, k),
                       ws + L.ws_st(B, k), theta + L.g(k), tangent + L.g(k), tangent + L.be(k), B, L.H);
    dy_prev = dz;
  }
  if (int rc = gemm_xwT(st, ws + L.ws_y(B, L.NB - 1), tangent + L.hW(), dout, B, L.H, L.O)) return rc;
  if (int rc = gemm_xwT(st, dy_prev, theta + L.hW(), dout, B, L.H, L.O, 1.f)) return rc;
  const int64_t n = B * L.O;
  hipLaunchKernelGGL(add_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, tangent + L.hb(), n, L.O);
  SPO_LAUNCH_CHECK("spo_ma_jvp");
  return 0;
}

extern "C" int spo_ma_sample(const float* mean, const float* log_std, const float* eps, float std_x_coef, float std_y_coef,
                             int deterministic, float* act_out, float* logp_out, int64_t rows, int act_dim, void* stream) {
  SPO_REQUIRE(mean && log_std && act_out && logp_out && rows > 0 && act_dim > 0 && act_dim <= SPO_MAX_ACT && (deterministic || eps),
              "ma_sample: bad args");
  const int64_t n = rows * act_dim;
  hipLaunchKernelGGL(ma_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, log_std, eps,
                     std_x_coef, std_y_coef, deterministic, act_out, logp_out, n, act_dim);
  SPO_LAUNCH_CHECK("spo_ma_sample");
  return 0;
}
extern "C" int spo_ma_log_probs(const float* mean, const float* log_std, const float* act, float std_x_coef, float std_y_coef,
                                float* logp_out, int64_t rows, int act_dim, void* stream) {
  SPO_REQUIRE(mean && log_std && act && logp_out && rows > 0 && act_dim > 0 && act_dim <= SPO_MAX_ACT, "ma_log_probs: bad args");
  const int64_t n = rows * act_dim;
  hipLaunchKernelGGL(ma_logp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, log_std, act,
                     std_x_coef, std_y_coef, logp_out, n, act_dim);
  SPO_LAUNCH_CHECK("spo_ma_log_probs");
  return 0;
}

// partial_ws: double[1024 * (4 + SPO_MAX_ACT)]
extern "C" int spo_ma_actor_loss(const float* mean, const float* log_std, const float* act, const float* old_logp,
                                 const float* adv, const float* cost_adv, const float* factor, const float* active,
                                 const float* lamda_dev, const spo_ma_loss_cfg* cfg, int64_t rows, int act_dim,
                                 float denom_host, int64_t rows_global, float* dmean_out, float* dlogstd_out,
                                 float* scalars5_out, double* partial_ws, void* stream) {
  SPO_REQUIRE(mean && log_std && act && old_logp && adv && cost_adv && factor && active && lamda_dev && cfg && dmean_out &&
                  dlogstd_out && scalars5_out && partial_ws, "ma_actor_loss: null pointer");
  SPO_REQUIRE(rows > 0 && act_dim > 0 && act_dim <= SPO_MAX_ACT && rows_global >= rows && denom_host > 0.f, "ma_actor_loss: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  int64_t g = (rows + 255) / 256;
  const int gr = (int)(g > 1024 ? 1024 : g);
  const float inv_denom = 1.f / denom_host;
  const float ent_w = 1.f / ((float)rows_global * (float)act_dim);
  hipLaunchKernelGGL(ma_actor_loss_kernel, dim3(gr), dim3(256), 0, st, mean, log_std, act, old_logp, adv, cost_adv, factor,
                     active, lamda_dev, *cfg, inv_denom, ent_w, dmean_out, partial_ws, rows, act_dim);
  hipLaunchKernelGGL(ma_actor_loss_finish_kernel, dim3(AL_NS + act_dim), dim3(256), 0, st, partial_ws, gr, log_std, *cfg, rows_global, act_dim,
                     scalars5_out, dlogstd_out);
  SPO_LAUNCH_CHECK("spo_ma_actor_loss");
  return 0;
}

extern "C" int spo_ma_lamda_update(float* lamda_dev, const float* scalars5, float aver_episode_cost, float cost_limit,
                                   float gamma, float lagrangian_coef_rate, void* stream) {
  SPO_REQUIRE(lamda_dev && scalars5, "ma_lamda_update: null pointer");
  hipLaunchKernelGGL(ma_lamda_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, lamda_dev, scalars5, aver_episode_cost,
                     cost_limit, gamma, lagrangian_coef_rate);
  SPO_LAUNCH_CHECK("spo_ma_lamda_update");
  return 0;
}

// PopArt.forward(x, train) in two calls so that a data-parallel job can all-reduce the batch sums in between:
// spo_ma_popart_stats -> sums2_dev = {sum x, sum x^2} of the local rows; spo_ma_popart_forward(train) folds the (global)
// sums over rows_global rows into the statistics, then normalises the local rows.  partial_ws: double[2 * 1024]
extern "C" int spo_ma_popart_stats(const float* x, int64_t rows, double* sums2_dev, double* partial_ws, void* stream) {
  SPO_REQUIRE(x && sums2_dev && partial_ws && rows > 0, "ma_popart_stats: bad args");
  hipStream_t st = (hipStream_t)stream;
  int64_t g = (rows + 255) / 256;
  const int gr = (int)(g > 1024 ? 1024 : g);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(gr), dim3(256), 0, st, x, rows, partial_ws);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(2), dim3(256), 0, st, partial_ws, gr, sums2_dev);
  SPO_LAUNCH_CHECK("spo_ma_popart_stats");
  return 0;
}
extern "C" int spo_ma_popart_forward(const float* x, int64_t rows, float* state3, double beta_d, float epsilon, int train,
                                     const double* sums2_dev, int64_t rows_global, float* out, void* stream) {
  SPO_REQUIRE(x && state3 && out && rows > 0 && (!train || (sums2_dev && rows_global >= rows)), "ma_popart_forward: bad args");
  const float beta = (float)beta_d;
  hipStream_t st = (hipStream_t)stream;
  if (train)
    hipLaunchKernelGGL(popart_update_kernel, dim3(1), dim3(1), 0, st, sums2_dev, rows_global, beta, (float)(1.0 - beta_d), state3);
  hipLaunchKernelGGL(popart_normalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, x, state3, epsilon, out, rows);
  SPO_LAUNCH_CHECK("spo_ma_popart_forward");
  return 0;
}

extern "C" int spo_ma_value_loss(const float* values, const float* value_preds, const float* returns_norm_clipped,
                                 const float* returns_norm_original, const float* active_or_null, float denom_host,
                                 float clip_param, float huber_delta, float value_loss_coef, int64_t rows, int64_t rows_global,
                                 float* dvalues_out, float* loss_out, double* partial_ws, void* stream) {
  SPO_REQUIRE(values && value_preds && returns_norm_clipped && returns_norm_original && dvalues_out && loss_out && partial_ws &&
                  rows > 0 && rows_global >= rows && denom_host > 0.f, "ma_value_loss: bad args");
  hipStream_t st = (hipStream_t)stream;
  int64_t g = (rows + 255) / 256;
  const int gr = (int)(g > 1024 ? 1024 : g);
  hipLaunchKernelGGL(ma_value_loss_kernel, dim3(gr), dim3(256), 0, st, values, value_preds, returns_norm_clipped,
                     returns_norm_original, active_or_null, clip_param, huber_delta, value_loss_coef / denom_host, dvalues_out,
                     partial_ws, rows);
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, partial_ws, gr, 1.0 / (double)denom_host, loss_out);
  SPO_LAUNCH_CHECK("spo_ma_value_loss");
  return 0;
}

// clip_grad_norm_(max_norm) over one network's flat gradient, then torch.optim.Adam(lr, eps, weight_decay) step number
// adam_step_host + 1.  grad_norm_out (device float) receives the pre-clip norm.  partial_ws: double[1024]
extern "C" int spo_ma_clip_adam(float* theta, const float* grad, float* adam_m, float* adam_v, int64_t n, int64_t adam_step_host,
                                float lr, float adam_eps, float weight_decay, float max_grad_norm, int use_max_grad_norm,
                                float* grad_norm_out, double* partial_ws, void* stream) {
  SPO_REQUIRE(theta && grad && adam_m && adam_v && grad_norm_out && partial_ws && n > 0 && adam_step_host >= 0, "ma_clip_adam: bad args");
  hipStream_t st = (hipStream_t)stream;
  int64_t g = (n + 255) / 256;
  const int gr = (int)(g > 1024 ? 1024 : g);
  hipLaunchKernelGGL(sq_partial_kernel, dim3(gr), dim3(256), 0, st, grad, n, partial_ws);
  hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(256), 0, st, partial_ws, gr, grad_norm_out);
  const double b1 = 0.9, b2 = 0.999;
  const double t = (double)(adam_step_host + 1);
  const float bc1 = (float)(1.0 - pow(b1, t)), bc2s = (float)sqrt(1.0 - pow(b2, t));
  hipLaunchKernelGGL(ma_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, theta, grad, adam_m, adam_v, n,
                     grad_norm_out, max_grad_norm, use_max_grad_norm, lr, (float)b1, (float)b2, adam_eps, weight_decay, bc1, bc2s);
  SPO_LAUNCH_CHECK("spo_ma_clip_adam");
  return 0;
}

// Test comparator: Y[B,N] = X[B,K] W[N,K]^T (mode 0) or dX[B,K] = dY[B,N] W[N,K] (mode 1) through the hand-written MFMA
// kernel (use_rocblas = 0) or through rocBLAS (use_rocblas = 1: dlopen'ed, never used by the product path).
extern "C" int spo_debug_ma_gemm(int use_rocblas, int mode, const float* x, const float* w, float* y, int64_t B, int K, int N,
                                 void* stream) {
  SPO_REQUIRE(x && w && y && B > 0 && K > 0 && N > 0 && (mode == 0 || mode == 1), "debug_ma_gemm: bad args");
  g_ma_gemm_rocblas = use_rocblas ? 1 : 0;
  const int rc = mode == 0 ? gemm_xwT((hipStream_t)stream, x, w, y, B, K, N) : gemm_dyw((hipStream_t)stream, x, w, y, B, K, N);
  g_ma_gemm_rocblas = 0;
  if (rc) return rc;
  SPO_LAUNCH_CHECK("spo_debug_ma_gemm");
  return 0;
}

// ====================================================================================================================
// Collect step of the multi-agent runner: every network of every agent in ONE launch (mappolag.py:411-447 calls
// policy.get_actions per agent: actor + critic + cost critic, 12 networks of 4 agents at BASELINE config 5).  A collect-size
// batch (8 192 rollout threads) is 128 workgroups per layer kernel: launched network by network and layer by layer
// (feature LayerNorm, fold, 3 fused blocks, head, sampling: ~60 launches of 5-13 us per step) the chip idles.  Here a
// workgroup takes 64 rows of ONE network (blockIdx.y) through the WHOLE network: feature LayerNorm in registers -> LDS tile,
// per block the weights staged to LDS (next block's weights prefetched into registers behind the MFMA loop), fp32 MFMA,
// bias + ELU + LayerNorm written back IN PLACE as the next block's input tile, head on four lanes per row, and for actors the
// Gaussian sample + per-dimension log-probabilities.  No activation ever goes to HBM.  Every element is computed with the
// arithmetic, operand order and cross-lane sums of the per-layer kernels above (ln_fwd_narrow_kernel, fn_fold_kernel,
// fused_block_fwd128_kernel<1>, head_small_kernel, ma_sample_kernel), so the results are BIT-IDENTICAL to spo_ma_forward +
// spo_ma_sample (tests: test_ma_collect_forward_is_bit_identical_to_the_per_network_path).
namespace {
constexpr int MC_MAX_NETS = SPO_MA_COLLECT_MAX_NETS, MC_MAX_BLOCKS = 4, MC_ROWS = 64;
struct McNet {
  const float* theta; const float* x; float* w0f; float* b0f; float* out;
  const float* eps; float* act; float* logp;
  int D, O, NB, is_actor, deterministic;
  int oW[MC_MAX_BLOCKS], ob[MC_MAX_BLOCKS], og[MC_MAX_BLOCKS], obe[MC_MAX_BLOCKS];
  int ohW, ohb, ols;
  float xc, yc;
};
struct McArgs { McNet n[MC_MAX_NETS]; int64_t rows; };

// fn_fold_kernel for all networks of the launch: grid (128 output rows, networks), 64 lanes
__global__ void fn_fold_multi_kernel(McArgs A) {
  const McNet& nt = A.n[blockIdx.y];
  const float* W0 = nt.theta + nt.oW[0]; const float* b0 = nt.theta + nt.ob[0];
  const float* gam = nt.theta; const float* bet = nt.theta + nt.D;
  const int D = nt.D, n = blockIdx.x;
  float dot = 0.f;
  for (int c = threadIdx.x; c < D; c += 64) {
    const float w = W0[(int64_t)n * D + c];
    nt.w0f[(int64_t)n * D + c] = w * gam[c];
    dot = fmaf(w, bet[c], dot);
  }
  dot = wave_sum_all(dot);
  if (threadIdx.x == 0) nt.b0f[n] = b0[n] + dot;
}

// feature LayerNorm of the 64-row tile into the LDS tile (pre-affine, like ln_fwd_narrow_kernel<LPR>)
template <int LPR>
__device__ __forceinline__ void mc_input_ln(const float* __restrict__ x, int64_t r0, int64_t B, int D, int LD0, int KP0, float* Xs,
                                            int tid) {
  constexpr int RPW = 64 / LPR;
  const int lane = tid & 63, wave = tid >> 6, sub = lane / LPR, c = (lane % LPR) * 4;
  const bool col_ok = c < D;
  const float inv_d = 1.f / (float)D;
#pragma unroll 2
  for (int ps = 0; ps < MC_ROWS / (4 * RPW); ++ps) {
    const int rl = (ps * 4 + wave) * RPW + sub;
    const int64_t row = r0 + rl;
    const bool ok = row < B && col_ok;
    f4w v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f4w*>(x + row * D + c);
    const float mean = group_allsum<LPR>((v[0] + v[1]) + (v[2] + v[3])) * inv_d;
    f4w d = v - mean;
    if (!col_ok) d = f4w{0.f, 0.f, 0.f, 0.f};
    const float var = group_allsum<LPR>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * inv_d;
    const float rstd = 1.f / sqrtf(var + LN_EPS);
    f4w o = d * rstd;
    if (!ok) o = f4w{0.f, 0.f, 0.f, 0.f};
    if (c < KP0) *reinterpret_cast<f4w*>(Xs + rl * LD0 + c) = o;
  }
}

// K chunk of the weight image: a block's [128][K] weights go through LDS in 64-column chunks (34 KB instead of 67 KB), which
// brings the workgroup to 69 KB of LDS and two workgroups to a CU -- one fills the other's staging / epilogue bubbles
constexpr int MC_KC = 64, MC_WLD = MC_KC + 4;
__global__ __launch_bounds__(256, 2) void ma_collect_kernel(McArgs A) {
  extern __shared__ __attribute__((aligned(16))) float mc_lds[];
  float* Ws = mc_lds;                              // [128][68] one K chunk of the current block's weights; the head's [O][132] at the end
  float* Xs = mc_lds + FB_N * MC_WLD;              // [64][LD] input tile; = Stg [64][132] ELU outputs / next input, in place
  float* Sst = Xs + MC_ROWS * FB_SLD;              // [64][2] row mean / rstd
  const McNet& nt = A.n[blockIdx.y];
  const float* const theta = nt.theta;
  const int64_t B = A.rows;
  const int D = nt.D, O = nt.O, NB = nt.NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
  const int lc4 = (tid & 15) * 4, lrow = tid >> 4;                // chunk staging: 16 rows x 64 columns per pass, 8 passes
  const int64_t r0 = (int64_t)blockIdx.x * MC_ROWS;
  f4w wv[8];
  auto fetch_w = [&](const float* W, int K, int ch) {            // chunk ch of W [128][K] into registers
    const int gc = MC_KC * ch + lc4;
    const bool lcol_ok = gc < K;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      wv[ps] = f4w{0.f, 0.f, 0.f, 0.f};
      if (lcol_ok) wv[ps] = *reinterpret_cast<const f4w*>(W + (int64_t)(lrow + 16 * ps) * K + gc);
    }
  };
  // block 0's (folded) weights are on their way while the feature LayerNorm runs
  fetch_w(nt.w0f, D, 0);
  {
    const int KP0 = (D + 15) & ~15, LD0 = KP0 + 4;
    if (D <= 64) mc_input_ln<16>(nt.x, r0, B, D, LD0, KP0, Xs, tid);
    else mc_input_ln<32>(nt.x, r0, B, D, LD0, KP0, Xs, tid);
  }
  float hwv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < NB; ++k) {
    const int K = k == 0 ? D : FB_N, KP = (K + 15) & ~15, LD = KP + 4;
    const int nch = (KP + MC_KC - 1) / MC_KC;
    const float* Wk = k == 0 ? nt.w0f : theta + nt.oW[k];
    const float* bias = k == 0 ? nt.b0f : theta + nt.ob[k];
    float bcol[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bcol[t] = bias[16 * t + i];
    f4w acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f4w{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nch; ++ch) {
      const int cw = KP - MC_KC * ch < MC_KC ? KP - MC_KC * ch : MC_KC;       // columns of this chunk (multiple of 16)
      if (lc4 < cw) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) *reinterpret_cast<f4w*>(Ws + (lrow + 16 * ps) * MC_WLD + lc4) = wv[ps];
      }
      __syncthreads();                             // the chunk (and, for ch == 0, the input tile) is in LDS
      if (ch + 1 < nch) fetch_w(Wk, K, ch + 1);
      else if (k + 1 < NB) fetch_w(theta + nt.oW[k + 1], FB_N, 0);
      else {
        // head weights [O][128] (behind an actor's log_std they are not 16-byte aligned: scalar loads, coalesced)
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const int f = tid + 256 * h;
          if (f < O * FB_N) hwv[h] = theta[nt.ohW + f];
        }
      }
      const float* xa = Xs + (16 * wave + i) * LD + 4 * kk + MC_KC * ch;
      const float* wb = Ws + i * MC_WLD + 4 * kk;
      for (int kb = 0; kb < cw / 16; ++kb) {
        const f4w am = *reinterpret_cast<const f4w*>(xa + 16 * kb);
        f4w bt[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) bt[t] = *reinterpret_cast<const f4w*>(wb + 16 * t * MC_WLD + 16 * kb);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[r], bt[t][r], acc[t], 0, 0, 0);
      }
      __syncthreads();                             // every wave is done with this chunk (after the last: with the input tile)
    }
    const float* g = theta + nt.og[k]; const float* be = theta + nt.obe[k];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = 16 * wave + 4 * kk + e;
      float v[8], sum = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float z = acc[t][e] + bcol[t];
        z = z > 0.f ? z : __expf(z) - 1.f;
        v[t] = z; sum += z;
      }
      const float mean = row16_allsum(sum) * (1.f / FB_N);
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) { const float d = v[t] - mean; q += d * d; }
      const float rstd = 1.f / sqrtf(row16_allsum(q) * (1.f / FB_N) + LN_EPS);
#pragma unroll
      for (int t = 0; t < 8; ++t) Xs[rl * FB_SLD + 16 * t + i] = v[t];
      if (i == 0) { Sst[2 * rl] = mean; Sst[2 * rl + 1] = rstd; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the rows are wave-private
    {
      const int c4 = (lane & 31) * 4;
      const f4w g4 = *reinterpret_cast<const f4w*>(g + c4), be4 = *reinterpret_cast<const f4w*>(be + c4);
#pragma unroll 4
      for (int jj = 0; jj < 8; ++jj) {
        const int rl = 16 * wave + 2 * jj + (lane >> 5);
        const f4w a4 = *reinterpret_cast<const f4w*>(Xs + rl * FB_SLD + c4);
        const float mean = Sst[2 * rl], rstd = Sst[2 * rl + 1];
        *reinterpret_cast<f4w*>(Xs + rl * FB_SLD + c4) = (a4 - mean) * rstd * g4 + be4;      // next block's input, in place
      }
    }
  }
  // ---- head (head_small_kernel's arithmetic: four lanes per row, quarter k0 = 4 part + 16 step), bias, Gaussian sample
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const int f = tid + 256 * h;
    if (f < O * FB_N) Ws[(f >> 7) * FB_SLD + (f & 127)] = hwv[h];
  }
  __syncthreads();
  const int rl = tid >> 2, part = tid & 3;
  const int64_t row = r0 + rl;
  float hacc[HS_MAXN];
#pragma unroll
  for (int n = 0; n < HS_MAXN; ++n) hacc[n] = 0.f;
  {
    const float* xr = Xs + rl * FB_SLD;
    for (int k0 = 4 * part; k0 < FB_N; k0 += 16) {
      const f4w xv = *reinterpret_cast<const f4w*>(xr + k0);
#pragma unroll
      for (int n = 0; n < HS_MAXN; ++n)
        if (n < O) {
          const float* wr = Ws + n * FB_SLD + k0;
          hacc[n] = fmaf(xv[0], wr[0], fmaf(xv[1], wr[1], fmaf(xv[2], wr[2], fmaf(xv[3], wr[3], hacc[n]))));
        }
    }
  }
#pragma unroll
  for (int n = 0; n < HS_MAXN; ++n) {
    hacc[n] += __shfl_xor(hacc[n], 1);
    hacc[n] += __shfl_xor(hacc[n], 2);
  }
  if (row < B && part == 0) {
    const float* hb = theta + nt.ohb;
#pragma unroll
    for (int n = 0; n < HS_MAXN; ++n)
      if (n < O) {
        const float mu = hacc[n] + hb[n];
        if (nt.out) nt.out[row * O + n] = mu;
        if (nt.is_actor) {
          const int64_t e = row * O + n;
          const float sd = nt.yc / (1.f + expf(-theta[nt.ols + n] / nt.xc));
          const float xs = nt.deterministic ? mu : mu + sd * nt.eps[e];
          nt.act[e] = xs;
          const float d = xs - mu;
          nt.logp[e] = -(d * d) / (2.f * sd * sd) - logf(sd) - LOG_SQRT_2PI_F;
        }
      }
  }
}
}  // namespace

extern "C" int64_t spo_ma_collect_scratch_floats(int32_t n_nets) {
  if (n_nets < 1 || n_nets > MC_MAX_NETS) return -1;
  return (int64_t)n_nets * (FB_N * FB_N + FB_N);
}
// 0 = launched; SPO_MA_COLLECT_UNSUPPORTED (positive, nothing launched) = a geometry outside the fused kernel: the caller runs
// spo_ma_forward + spo_ma_sample per network instead.
extern "C" int spo_ma_collect_forward(int32_t n_nets, const spo_ma_collect_net* nets, int64_t rows, float* scratch, void* stream) {
  SPO_REQUIRE(nets && scratch && rows > 0 && n_nets >= 1, "ma_collect_forward: bad args");
  if (n_nets > MC_MAX_NETS) return SPO_MA_COLLECT_UNSUPPORTED;
  McArgs args;
  memset(&args, 0, sizeof(args));
  args.rows = rows;
  for (int n = 0; n < n_nets; ++n) {
    const spo_ma_collect_net& c = nets[n];
    Lay L;
    if (int rc = lay_of(&c.net, &L)) return rc;
    SPO_REQUIRE(c.theta && c.x, "ma_collect_forward: net %d: theta / x is NULL", n);
    if (L.H != FB_N || L.D > 128 || L.D % 4 != 0 || L.O > HS_MAXN || L.NB > MC_MAX_BLOCKS ||
        reinterpret_cast<uintptr_t>(c.x) % 16 != 0 || reinterpret_cast<uintptr_t>(c.theta) % 16 != 0)
      return SPO_MA_COLLECT_UNSUPPORTED;
    McNet& m = args.n[n];
    m.theta = c.theta; m.x = c.x; m.out = c.out;
    m.w0f = scratch + (int64_t)n * (FB_N * FB_N + FB_N); m.b0f = m.w0f + FB_N * FB_N;
    m.D = L.D; m.O = L.O; m.NB = L.NB; m.is_actor = L.actor;
    for (int k = 0; k < L.NB; ++k) {
      m.oW[k] = (int)L.W(k); m.ob[k] = (int)L.b(k); m.og[k] = (int)L.g(k); m.obe[k] = (int)L.be(k);
      // float4 reads of the block weights and LayerNorm vectors: offsets must be multiples of 4 floats
      if ((k > 0 && L.W(k) % 4 != 0) || L.g(k) % 4 != 0 || L.be(k) % 4 != 0) return SPO_MA_COLLECT_UNSUPPORTED;
    }
    m.ohW = (int)L.hW(); m.ohb = (int)L.hb(); m.ols = L.actor ? (int)L.logstd() : 0;
    if (L.actor) {
      SPO_REQUIRE(c.act && c.logp && (c.eps || c.deterministic), "ma_collect_forward: net %d (actor): act / logp / eps is NULL", n);
      m.eps = c.eps; m.act = c.act; m.logp = c.logp; m.deterministic = c.deterministic ? 1 : 0;
      m.xc = c.std_x_coef; m.yc = c.std_y_coef;
    } else {
      SPO_REQUIRE(c.out, "ma_collect_forward: net %d (critic): out is NULL", n);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  constexpr size_t sh = ((size_t)FB_N * MC_WLD + MC_ROWS * FB_SLD + 2 * MC_ROWS) * sizeof(float);
  static bool attr_done_dev[spo::SPO_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[spo::current_device_slot()];
  if (!attr_done) {
    if (int rc = spo::hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(&ma_collect_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh),
                                "hipFuncSetAttribute(ma_collect_kernel)")) return rc;
    attr_done = true;
  }
  const int64_t tiles = (rows + MC_ROWS - 1) / MC_ROWS;
  if (tiles > 0x7fffffffLL) return fail(-1, "ma_collect_forward: %lld rows exceed the launch grid", (long long)rows);
  hipLaunchKernelGGL(fn_fold_multi_kernel, dim3(FB_N, n_nets), dim3(64), 0, st, args);
  hipLaunchKernelGGL(ma_collect_kernel, dim3((unsigned)tiles, n_nets), dim3(256), sh, st, args);
  SPO_LAUNCH_CHECK("spo_ma_collect_forward");
  return 0;
}

// ====================================================================================================================
// Wide single-agent networks: ActorVCritic(obs_dim, act_dim, hidden_sizes) for ANY hidden_sizes (reference
// safepo/common/model.py:30-48,131; isaac_gym_specific_cfg uses [1024, 1024, 512] with minibatches of 8 192 rows,
// safepo/single_agent/ppo_lag.py:54-65).  The persistent kernels of update.hip keep a 64-wide network in one CU's LDS for
// thousands of dependent 64-row steps; a wide network at a wide batch is the opposite regime -- a handful of large steps
// per epoch, each a chain of real GEMMs -- so it runs as a sequence of launches on the in-tree fp32 MFMA GEMM kernels
// above (X W^T, dY W, dY^T X) with small elementwise kernels between them:
//   forward   h_l = tanh(h_{l-1} W_l^T + b_l), last layer linear                      spo_mlp_forward
//   backward  dZ_l = dH_l (1 - h_l^2); db_l = colsum dZ_l; dW_l = dZ_l^T h_{l-1}; dH_{l-1} = dZ_l W_l     spo_mlp_backward
//   losses    MSE critics, clipped PPO surrogate + d(log_std) (ppo_lag.py:306-323)     spo_wide_ppo_loss
//   step      critic L2 terms, joint clip_grad_norm_, Adam (ppo_lag.py:310-329)        spo_wide_clip_adam
// Flat parameter layout of one network: for every Linear in order, W [out, in] row-major then b [out] -- the order of
// nn.Sequential.parameters(), so the module's parameters stay views of one vector (log_std precedes the actor's layers).
namespace {
using namespace spo;

// ---- 128 x 128-tile fp32 MFMA GEMM for the wide networks (round 3).  gemm_mfma_kernel above (64 x 64 tiles, scalar staging)
// serves launch-bound shapes; a [1024, 1024, 512] network at 8 192 rows is a chain of real GEMMs (17 GFLOP each), so:
//   TRANSB = false:  Y[B, N] = act(X[B, K] W[N, K]^T + bias)            (forward: both operands K-contiguous)
//   TRANSB = true :  Y[B, N] = (X[B, K] W[K, N]) * (1 - Hm[B, N]^2)      (input gradient dY W with the tanh' factor fused)
// One workgroup = 128 x 128 outputs, 2 x 2 waves of 64 x 64 (16 accumulator tiles per wave); the reduction runs in chunks of
// 16 staged through LDS with row stride 20 floats -- a lane's ds_read_b128 at [row][4q] is the operand of four consecutive
// MFMA steps (k = 4q + r, the enumeration of mlp_mfma.h), and 20 i mod 64 walks all sixteen 4-bank groups, so the sixteen
// lanes of a read phase never collide.  The next chunk is fetched into registers while the current one is multiplied.
// Requirements (else the caller falls back to gemm_mfma_kernel): K % 4 == 0 (TRANSB: N % 4 == 0), 16-byte aligned operands.
constexpr int G2_T = 128, G2_K = 16, G2_LD = 20;
template <bool TRANSB>
__global__ __launch_bounds__(256) void gemm128_kernel(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y,
                                                      int64_t B, int K, int N, const float* __restrict__ bias, int tanh_on,
                                                      const float* __restrict__ Hm, int w_vec) {
  __shared__ __attribute__((aligned(16))) float As[G2_T * G2_LD];
  __shared__ __attribute__((aligned(16))) float Bs[G2_T * G2_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t row0 = (int64_t)blockIdx.x * G2_T;
  const int col0 = blockIdx.y * G2_T;
  f4v acc[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f4v{0.f, 0.f, 0.f, 0.f};
  // staging assignment.  A (and B when it is K-contiguous): thread -> row tid / 2, eight k from (tid & 1) * 8.
  // TRANSB B: thread -> reduction row tid / 16 of the chunk, eight output columns from (tid & 15) * 8 (stored transposed).
  const int sa_r = tid >> 1, sa_k = (tid & 1) * 8;
  const int sb_n = tid >> 4, sb_c = (tid & 15) * 8;
  f4v pa[2], pb[2];
  auto fetch = [&](int k0) {
    const int64_t r = row0 + sa_r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + sa_k + 4 * h;
      pa[h] = (r < B && k < K) ? *reinterpret_cast<const f4v*>(X + r * K + k) : f4v{0.f, 0.f, 0.f, 0.f};
    }
    if (!TRANSB) {
      const int c = col0 + sa_r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = k0 + sa_k + 4 * h;
        pb[h] = f4v{0.f, 0.f, 0.f, 0.f};
        if (c < N && k < K) {
          const float* src = W + (int64_t)c * K + k;            // a network's slice of the flat vector need not be 16-byte aligned
          if (w_vec) pb[h] = *reinterpret_cast<const f4v*>(src);
          else pb[h] = f4v{src[0], src[1], src[2], src[3]};
        }
      }
    } else {
      const int k = k0 + sb_n;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = col0 + sb_c + 4 * h;
        pb[h] = f4v{0.f, 0.f, 0.f, 0.f};
        if (k < K && c < N) {
          const float* src = W + (int64_t)k * N + c;
          if (w_vec) pb[h] = *reinterpret_cast<const f4v*>(src);
          else pb[h] = f4v{src[0], src[1], src[2], src[3]};
        }
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += G2_K) {
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<f4v*>(As + sa_r * G2_LD + sa_k + 4 * h) = pa[h];
    if (!TRANSB) {
#pragma unroll
      for (int h = 0; h < 2; ++h) *reinterpret_cast<f4v*>(Bs + sa_r * G2_LD + sa_k + 4 * h) = pb[h];
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[(sb_c + 4 * h + e) * G2_LD + sb_n] = pb[h][e];
    }
    __syncthreads();
    if (k0 + G2_K < K) fetch(k0 + G2_K);                       // in flight during the 64 MFMAs below
    f4v a[4], b[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) a[rt] = *reinterpret_cast<const f4v*>(As + (64 * wr + 16 * rt + i) * G2_LD + 4 * q);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) b[ct] = *reinterpret_cast<const f4v*>(Bs + (64 * wc + 16 * ct + i) * G2_LD + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][r], b[ct][r], acc[rt][ct], 0, 0, 0);
    __syncthreads();
  }
  // C layout: lane holds rows 4q + e, column i of every 16 x 16 tile
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int col = col0 + 64 * wc + 16 * ct + i;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t row = row0 + 64 * wr + 16 * rt + 4 * q + e;
        if (row >= B) continue;
        float v = acc[rt][ct][e] + bv;
        if (tanh_on) v = fast_tanh(v);
        if (Hm) { const float hv = Hm[row * N + col]; v = v * fmaf(-hv, hv, 1.f); }
        Y[row * N + col] = v;
      }
    }
}
// true when the 128-tile kernel applies (big enough to fill tiles, aligned, vectorisable)
inline bool gemm128_ok(const float* X, int64_t B, int K, int N, bool transb) {
  if (B < 256 || N < 64 || K < 16) return false;
  if (((B + G2_T - 1) / G2_T) * ((N + G2_T - 1) / G2_T) < 192) return false;      // too few 128 x 128 tiles to fill 256 CUs
  if (reinterpret_cast<uintptr_t>(X) & 15) return false;
  return transb ? (K % 4 == 0 && N % 4 == 0) : (K % 4 == 0);
}
inline int w_is_vec(const float* W) { return (reinterpret_cast<uintptr_t>(W) & 15) == 0 ? 1 : 0; }

struct MlpLay {
  int n;                 // Linear layers
  int d[SPO_MLP_MAX_LAYERS + 1];
  int64_t w(int l) const { int64_t o = 0; for (int k = 0; k < l; ++k) o += (int64_t)d[k + 1] * d[k] + d[k + 1]; return o; }
  int64_t b(int l) const { return w(l) + (int64_t)d[l + 1] * d[l]; }
  int64_t count() const { return w(n); }
  int64_t act_off(int l, int64_t rows) const { int64_t o = 0; for (int k = 0; k < l; ++k) o += rows * d[k + 1]; return o; }   // h_{l+1} inside ws
  int maxdim() const { int m = 0; for (int k = 0; k <= n; ++k) m = d[k] > m ? d[k] : m; return m; }
};
int mlp_lay(const spo_mlp_net* net, MlpLay* L) {
  if (!net) return fail(-1, "mlp: net is NULL");
  if (net->n_layers < 1 || net->n_layers > SPO_MLP_MAX_LAYERS) return fail(-2, "mlp: n_layers %d outside [1,%d]", net->n_layers, SPO_MLP_MAX_LAYERS);
  L->n = net->n_layers;
  for (int k = 0; k <= L->n; ++k) {
    if (net->dims[k] < 1 || net->dims[k] > 16384) return fail(-2, "mlp: dims[%d] = %d outside [1,16384]", k, net->dims[k]);
    L->d[k] = net->dims[k];
  }
  return 0;
}

__global__ __launch_bounds__(256) void mlp_bias_act_kernel(float* __restrict__ y, const float* __restrict__ b, int64_t n, int N, int tanh_on) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = y[i] + b[i % N];
    y[i] = tanh_on ? fast_tanh(v) : v;
  }
}
__global__ __launch_bounds__(256) void mlp_dtanh_kernel(float* __restrict__ dh, const float* __restrict__ h, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float hv = h[i];
    dh[i] = dh[i] * fmaf(-hv, hv, 1.f);
  }
}
// column sums in two fixed-order stages: partial[s][c] over row slice s, then over the slices
constexpr int CS_SLICES = 64;
__global__ __launch_bounds__(256) void mlp_colsum_partial_kernel(const float* __restrict__ d, int64_t B, int N, float* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (c >= N) return;
  const int64_t per = (B + CS_SLICES - 1) / CS_SLICES, r0 = s * per, r1 = r0 + per < B ? r0 + per : B;
  float acc = 0.f;
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {                       // eight loads in flight, added in row order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = d[(r + u) * N + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; r < r1; ++r) acc += d[r * N + c];
  partial[(int64_t)s * N + c] = acc;
}
__global__ __launch_bounds__(256) void mlp_colsum_finish_kernel(const float* __restrict__ partial, int N, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float acc = 0.f;
  for (int s = 0; s < CS_SLICES; ++s) acc += partial[(int64_t)s * N + c];
  out[c] = acc;
}
int ew_grid(int64_t n) { const int64_t g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

// PPO-Lagrangian losses and output gradients of one minibatch (ppo_lag.py:306-323): one thread per row.
constexpr int WL_NS = 3 + SPO_MAX_ACT;       // loss_r sum, loss_c sum, surrogate sum, d(log_std)[A]
__global__ __launch_bounds__(256) void wide_ppo_loss_kernel(const float* __restrict__ v_r, const float* __restrict__ v_c, const float* __restrict__ mean,
                                                            const float* __restrict__ log_std, const float* __restrict__ act,
                                                            const float* __restrict__ logp_old, const float* __restrict__ adv,
                                                            const float* __restrict__ tgt_r, const float* __restrict__ tgt_c, int64_t B, int A,
                                                            float clip, float* __restrict__ d_vr, float* __restrict__ d_vc,
                                                            float* __restrict__ d_mean, double* __restrict__ partial,
                                                            float* __restrict__ losses3, float* __restrict__ d_log_std) {
  // losses3 / d_log_std: given for a one-workgroup launch (minibatch-sized row counts), which finishes the sums itself -- the
  // same values wide_ppo_loss_finish_kernel forms from one partial row, one launch less
  __shared__ double red[4][WL_NS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_n = 1.f / (float)B, clip_lo = 1.f - clip, clip_hi = 1.f + clip;
  double acc[WL_NS];
#pragma unroll
  for (int k = 0; k < WL_NS; ++k) acc[k] = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + tid; r < B; r += (int64_t)gridDim.x * 256) {
    const float dr = v_r[r] - tgt_r[r], dc = v_c[r] - tgt_c[r];
    acc[0] += (double)(dr * dr); acc[1] += (double)(dc * dc);
    d_vr[r] = 2.f * dr * inv_n; d_vc[r] = 2.f * dc * inv_n;
    float lp = 0.f, dif[SPO_MAX_ACT], ivar[SPO_MAX_ACT];
    for (int k = 0; k < A; ++k) {
      const float ls = log_std[k], sd = __expf(ls);
      ivar[k] = 1.f / (sd * sd);
      dif[k] = act[r * A + k] - mean[r * A + k];
      lp += -(dif[k] * dif[k]) * (0.5f * ivar[k]) - ls - LOG_SQRT_2PI_F;
    }
    const float ad = adv[r];
    const float ratio = __expf(lp - logp_old[r]);
    const float rc = fminf(fmaxf(ratio, clip_lo), clip_hi);
    const float s1 = ratio * ad, s2 = rc * ad;
    const bool inr = (ratio >= clip_lo) && (ratio <= clip_hi);
    float gr;                                          // backward of torch.min / torch.clamp (ties split the gradient)
    if (s1 < s2) gr = ad;
    else if (s1 > s2) gr = inr ? ad : 0.f;
    else gr = 0.5f * ad + (inr ? 0.5f * ad : 0.f);
    const float dlp = -(gr * ratio) * inv_n;
    acc[2] += (double)fminf(s1, s2);
    for (int k = 0; k < A; ++k) {
      const float z = dif[k] * ivar[k];
      d_mean[r * A + k] = dlp * z;
      acc[3 + k] += (double)(dlp * (dif[k] * z - 1.f));
    }
  }
  for (int k = 0; k < 3 + A; ++k) {
    const double v = wave_sum_d(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (tid < 3 + A) {
    const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    partial[(int64_t)blockIdx.x * WL_NS + tid] = v;
    if (losses3) {
      const double s = 0.0 + v;
      if (tid < 2) losses3[tid] = (float)(s / (double)B);
      else if (tid == 2) losses3[2] = (float)(-s / (double)B);
      else d_log_std[tid - 3] = (float)s;
    }
  }
}
__global__ void wide_ppo_loss_finish_kernel(const double* __restrict__ partial, int nblocks, int A, int64_t B, float* __restrict__ losses3,
                                            float* __restrict__ d_log_std) {
  const int k = threadIdx.x;
  if (k >= 3 + A) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * WL_NS + k];
  if (k < 2) losses3[k] = (float)(s / (double)B);
  else if (k == 2) losses3[2] = (float)(-s / (double)B);
  else d_log_std[k - 3] = (float)s;
}

// Joint clip + Adam over the flat vector of all three networks: stage 1 adds the critics' L2 gradient (2 l2 p) and the value
// coefficient in place and reduces ||g||^2 and the critics' sum p^2; stage 2 forms the clip coefficient; stage 3 is Adam.
struct WideAdamArgs {
  float* theta; float* grad; float* m; float* v; int64_t P, r_end, c_end, actor_begin;
  float l2, vcoef_r, max_norm, lr_actor, lr_critic, b1, b2, eps;
  double pow_b1, pow_b2;
  double* partial; float* scal;      // scal: {coef, l2 * sum p^2 (reward critic), l2 * sum p^2 (cost critic), ||g||}
  float* losses3;
};
__global__ __launch_bounds__(256) void wide_prep_kernel(WideAdamArgs a) {
  __shared__ double red[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double gs = 0.0, pr = 0.0, pc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < a.P; i += (int64_t)gridDim.x * 256) {
    float g = a.grad[i];
    if (i < a.c_end) {
      const float p = a.theta[i];
      g = fmaf(2.f * a.l2, p, g);
      if (i < a.r_end) { g *= a.vcoef_r; pr += (double)(p * p); } else pc += (double)(p * p);
      a.grad[i] = g;
    }
    gs += (double)(g * g);
  }
  gs = wave_sum_d(gs); pr = wave_sum_d(pr); pc = wave_sum_d(pc);
  if (lane == 0) { red[wave][0] = gs; red[wave][1] = pr; red[wave][2] = pc; }
  __syncthreads();
  if (tid < 3) a.partial[(int64_t)blockIdx.x * 3 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
// pow4 (optional): the device-resident optimiser clocks {beta1^t, beta2^t of the critics, of the actor}; the clocks named by
// adv_critics / adv_actor advance here, between the norm and the Adam pass (spo_wide_clip_adam_dev)
// loss_log / log_src / cursor (optional, device): the step's three losses log_src[0..2] are stored at loss_log[3 * (*cursor / cursor_step)] and
// *cursor advances by cursor_step -- the window spo_gather_rows_at reads; a replayed step then needs no host copy in or out
__global__ __launch_bounds__(64) void wide_coef_kernel(WideAdamArgs a, int nblocks, double* pow4 = nullptr, int adv_critics = 0,
                                                       int adv_actor = 0, float* loss_log = nullptr, const float* log_src = nullptr,
                                                       int64_t* cursor = nullptr, int64_t cursor_step = 0) {
  // one wave: lane l adds the partials l, l + 64, ... in order, then a fixed butterfly over the lanes
  double gs = 0.0, pr = 0.0, pc = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) { gs += a.partial[b * 3]; pr += a.partial[b * 3 + 1]; pc += a.partial[b * 3 + 2]; }
  gs = wave_sum_d(gs); pr = wave_sum_d(pr); pc = wave_sum_d(pc);
  if (threadIdx.x != 0) return;
  if (pow4) {
    if (adv_critics) { pow4[0] *= (double)a.b1; pow4[1] *= (double)a.b2; }
    if (adv_actor) { pow4[2] *= (double)a.b1; pow4[3] *= (double)a.b2; }
  }
  const float norm = sqrtf((float)gs);
  float coef = a.max_norm / (norm + 1e-6f);                   // clip_grad_norm_ (torch): eps 1e-6
  a.scal[0] = coef > 1.f ? 1.f : coef;
  a.scal[1] = a.l2 * (float)pr; a.scal[2] = a.l2 * (float)pc; a.scal[3] = norm;
  if (a.losses3) { a.losses3[0] += a.scal[1]; a.losses3[1] += a.scal[2]; }      // logged critic losses include their L2 terms
  if (cursor) {
    const int64_t at = cursor[0];
    if (loss_log && log_src) {
      float* row = loss_log + 3 * (at / cursor_step);
      row[0] = log_src[0]; row[1] = log_src[1]; row[2] = log_src[2];
    }
    cursor[0] = at + cursor_step;
  }
}
__global__ __launch_bounds__(256) void wide_adam_kernel(WideAdamArgs a) {
  const float coef = a.scal[0];
  const double pw1 = a.pow_b1 * (double)a.b1, pw2 = a.pow_b2 * (double)a.b2;
  float ss_a, ss_c, bc2s, bc2s_again;
  adam_scalars(a.lr_actor, pw1, pw2, ss_a, bc2s);
  adam_scalars(a.lr_critic, pw1, pw2, ss_c, bc2s_again);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.P; i += (int64_t)gridDim.x * 256) {
    const AdamOut o = adam1(a.theta[i], a.grad[i] * coef, a.m[i], a.v[i], a.b1, a.b2, a.eps, i >= a.actor_begin ? ss_a : ss_c, bc2s);
    a.theta[i] = o.p; a.m[i] = o.m; a.v[i] = o.v;
  }
}
}  // namespace

namespace {
// a = mean + exp(log_std) * eps (rsample), logp = Normal(mean, exp(log_std)).log_prob(a).sum(-1)   (model.py:149-170)
__global__ __launch_bounds__(256) void gauss_sample_kernel(const float* __restrict__ mean, const float* __restrict__ log_std,
                                                           const float* __restrict__ eps, float* __restrict__ act, float* __restrict__ logp,
                                                           int64_t B, int A) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= B) return;
  float lp = 0.f;
  for (int k = 0; k < A; ++k) {
    const float sd = expf(log_std[k]), mu = mean[r * A + k];
    const float ac = eps ? mu + eps[r * A + k] * sd : mu;
    const float diff = ac - mu, var = sd * sd;
    lp += -(diff * diff) / (2.f * var) - logf(sd) - LOG_SQRT_2PI_F;
    act[r * A + k] = ac;
  }
  logp[r] = lp;
}
// sum over rows of KL(N(mean_old, std_old) || N(mean_new, std_new)).sum(-1)   (torch kl_normal_normal, ppo_lag.py:338-345)
__global__ __launch_bounds__(256) void gauss_kl_kernel(const float* __restrict__ mean_old, const float* __restrict__ log_std_old,
                                                       const float* __restrict__ mean_new, const float* __restrict__ log_std_new, int64_t B,
                                                       int A, double* __restrict__ partial) {
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double acc = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + tid; r < B; r += (int64_t)gridDim.x * 256) {
    float kl = 0.f;
    for (int k = 0; k < A; ++k) {
      const float so = expf(log_std_old[k]), sn = expf(log_std_new[k]);
      const float ratio = so / sn, var_ratio = ratio * ratio;
      const float dm = (mean_old[r * A + k] - mean_new[r * A + k]) / sn;
      kl += 0.5f * (var_ratio + dm * dm - 1.f - logf(var_ratio));
    }
    acc += (double)kl;
  }
  acc = wave_sum_d(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void gauss_kl_finish_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ sum_inout, int accumulate) {
  if (threadIdx.x != 0) return;
  double s = accumulate ? sum_inout[0] : 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[b];
  sum_inout[0] = s;
}
}  // namespace

extern "C" int spo_gauss_sample(const float* mean, const float* log_std, const float* eps, float* act_out, float* logp_out,
                                int64_t rows, int act_dim, void* stream) {
  SPO_REQUIRE(mean && log_std && act_out && logp_out && rows > 0 && act_dim >= 1 && act_dim <= SPO_WIDE_MAX_ACT, "gauss_sample: bad args");
  hipLaunchKernelGGL(gauss_sample_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, log_std, eps,
                     act_out, logp_out, rows, act_dim);
  SPO_LAUNCH_CHECK("spo_gauss_sample");
  return 0;
}
extern "C" int spo_gauss_kl_sum(const float* mean_old, const float* log_std_old, const float* mean_new, const float* log_std_new,
                                int64_t rows, int act_dim, double* partial_ws, int partial_capacity, double* sum_inout, int accumulate,
                                void* stream) {
  SPO_REQUIRE(mean_old && log_std_old && mean_new && log_std_new && partial_ws && sum_inout && rows > 0 && act_dim >= 1 &&
                  act_dim <= SPO_WIDE_MAX_ACT && partial_capacity >= 1, "gauss_kl_sum: bad args");
  int64_t blocks = (rows + 255) / 256;
  if (blocks > 512) blocks = 512;
  if (blocks > partial_capacity) blocks = partial_capacity;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gauss_kl_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mean_old, log_std_old, mean_new, log_std_new, rows, act_dim,
                     partial_ws);
  hipLaunchKernelGGL(gauss_kl_finish_kernel, dim3(1), dim3(64), 0, st, partial_ws, (int)blocks, sum_inout, accumulate);
  SPO_LAUNCH_CHECK("spo_gauss_kl_sum");
  return 0;
}

extern "C" int64_t spo_mlp_param_count(const spo_mlp_net* net) {
  MlpLay L;
  if (mlp_lay(net, &L)) return -1;
  return L.count();
}
extern "C" int64_t spo_mlp_workspace_floats(const spo_mlp_net* net, int64_t rows) {
  MlpLay L;
  if (mlp_lay(net, &L) || rows < 1) return -1;
  return L.act_off(L.n, rows);
}
extern "C" int64_t spo_mlp_backward_scratch_floats(const spo_mlp_net* net, int64_t rows) {
  MlpLay L;
  if (mlp_lay(net, &L) || rows < 1) return -1;
  int64_t slices = 0;
  for (int l = 0; l < L.n; ++l) {
    const int64_t s = (int64_t)dw_splits(rows, L.d[l + 1], L.d[l]) * L.d[l + 1] * L.d[l];
    slices = s > slices ? s : slices;
  }
  return 2 * rows * (int64_t)L.maxdim() + slices + (int64_t)CS_SLICES * L.maxdim();
}

extern "C" int spo_mlp_forward(const float* theta, const spo_mlp_net* net, const float* x, int64_t rows, float* ws, void* stream) {
  MlpLay L;
  if (int rc = mlp_lay(net, &L)) return rc;
  SPO_REQUIRE(theta && x && ws && rows > 0, "mlp_forward: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (spo::mlp_small_ok(net, rows)) {
    // up to 128 rows: the whole network in one launch (csrc/mlp_small.hip)
    spo::MlpSmallBatch batch;
    batch.count = 1;
    spo::mlp_small_args(theta, net, x, rows, ws, nullptr, nullptr, &batch.a[0]);
    if (int rc = spo::mlp_small_launch(false, batch, st)) return rc;
    SPO_LAUNCH_CHECK("spo_mlp_forward");
    return 0;
  }
  const float* in = x;
  for (int l = 0; l < L.n; ++l) {
    float* out = ws + L.act_off(l, rows);
    const int K = L.d[l], N = L.d[l + 1];
    if (gemm128_ok(in, rows, K, N, false)) {
      // bias + tanh fused into the GEMM epilogue
      hipLaunchKernelGGL(gemm128_kernel<false>, dim3((unsigned)((rows + G2_T - 1) / G2_T), (unsigned)((N + G2_T - 1) / G2_T)), dim3(256), 0, st,
                         in, theta + L.w(l), out, rows, K, N, theta + L.b(l), l + 1 < L.n ? 1 : 0, (const float*)nullptr, w_is_vec(theta + L.w(l)));
    } else {
      if (int rc = gemm_xwT(st, in, theta + L.w(l), out, rows, K, N)) return rc;
      const int64_t n = rows * (int64_t)N;
      hipLaunchKernelGGL(mlp_bias_act_kernel, dim3(ew_grid(n)), dim3(256), 0, st, out, theta + L.b(l), n, N, l + 1 < L.n ? 1 : 0);
    }
    in = out;
  }
  SPO_LAUNCH_CHECK("spo_mlp_forward");
  return 0;
}

extern "C" int spo_mlp_backward(const float* theta, const spo_mlp_net* net, const float* x, int64_t rows, const float* ws,
                                const float* d_out, float* grad, float* scratch, void* stream) {
  MlpLay L;
  if (int rc = mlp_lay(net, &L)) return rc;
  SPO_REQUIRE(theta && x && ws && d_out && grad && scratch && rows > 0, "mlp_backward: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (spo::mlp_small_ok(net, rows)) {
    spo::MlpSmallBatch batch;
    batch.count = 1;
    spo::mlp_small_args(theta, net, x, rows, const_cast<float*>(ws), d_out, grad, &batch.a[0]);
    if (int rc = spo::mlp_small_launch(true, batch, st)) return rc;
    SPO_LAUNCH_CHECK("spo_mlp_backward");
    return 0;
  }
  const int md = L.maxdim();
  float* dA = scratch;                                   // ping-pong dZ buffers
  float* dB = scratch + rows * (int64_t)md;
  float* slices = scratch + 2 * rows * (int64_t)md;
  int64_t smax = 0;
  for (int l = 0; l < L.n; ++l) {
    const int64_t s = (int64_t)dw_splits(rows, L.d[l + 1], L.d[l]) * L.d[l + 1] * L.d[l];
    smax = s > smax ? s : smax;
  }
  float* cs = slices + smax;
  const float* dz = d_out;                               // dZ of the last (linear) layer is d_out itself
  for (int l = L.n - 1; l >= 0; --l) {
    const int N = L.d[l + 1], K = L.d[l];
    const float* hin = l == 0 ? x : ws + L.act_off(l - 1, rows);
    hipLaunchKernelGGL(mlp_colsum_partial_kernel, dim3((N + 255) / 256, CS_SLICES), dim3(256), 0, st, dz, rows, N, cs);
    hipLaunchKernelGGL(mlp_colsum_finish_kernel, dim3((N + 255) / 256), dim3(256), 0, st, cs, N, grad + L.b(l));
    if (int rc = gemm_dyTx(st, dz, hin, grad + L.w(l), rows, K, N, slices)) return rc;
    if (l > 0) {
      float* dh = (dz == dA) ? dB : dA;
      // dH_{l-1} = dZ_l W_l (reduction over this layer's N outputs, K inputs wide) and dZ_{l-1} = dH_{l-1} (1 - h_{l-1}^2)
      if (gemm128_ok(dz, rows, N, K, true)) {
        hipLaunchKernelGGL(gemm128_kernel<true>, dim3((unsigned)((rows + G2_T - 1) / G2_T), (unsigned)((K + G2_T - 1) / G2_T)), dim3(256), 0, st,
                           dz, theta + L.w(l), dh, rows, N, K, (const float*)nullptr, 0, hin, w_is_vec(theta + L.w(l)));
      } else {
        if (int rc = gemm_dyw(st, dz, theta + L.w(l), dh, rows, K, N)) return rc;
        const int64_t n = rows * (int64_t)K;
        hipLaunchKernelGGL(mlp_dtanh_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dh, hin, n);
      }
      dz = dh;
    }
  }
  SPO_LAUNCH_CHECK("spo_mlp_backward");
  return 0;
}

// The same for `count` networks on the same number of rows -- the three networks of a minibatch step -- in ONE launch when every
// network fits the small-row kernel (one workgroup per network), else one call per network.
extern "C" int spo_mlp_forward_multi(int count, const float* const* thetas, const spo_mlp_net* const* nets, const float* const* xs,
                                     int64_t rows, float* const* wss, void* stream) {
  SPO_REQUIRE(count >= 1 && count <= spo::MLP_SMALL_MAX_NETS && thetas && nets && xs && wss && rows > 0, "mlp_forward_multi: bad args");
  bool small = true;
  for (int i = 0; i < count; ++i) {
    SPO_REQUIRE(thetas[i] && nets[i] && xs[i] && wss[i], "mlp_forward_multi: null pointer");
    small = small && spo::mlp_small_ok(nets[i], rows);
  }
  if (!small) {
    for (int i = 0; i < count; ++i)
      if (int rc = spo_mlp_forward(thetas[i], nets[i], xs[i], rows, wss[i], stream)) return rc;
    return 0;
  }
  spo::MlpSmallBatch batch;
  batch.count = count;
  for (int i = 0; i < count; ++i) {
    MlpLay L;
    if (int rc = mlp_lay(nets[i], &L)) return rc;
    spo::mlp_small_args(thetas[i], nets[i], xs[i], rows, wss[i], nullptr, nullptr, &batch.a[i]);
  }
  if (int rc = spo::mlp_small_launch(false, batch, (hipStream_t)stream)) return rc;
  SPO_LAUNCH_CHECK("spo_mlp_forward_multi");
  return 0;
}

extern "C" int spo_mlp_backward_multi(int count, const float* const* thetas, const spo_mlp_net* const* nets, const float* const* xs,
                                      int64_t rows, const float* const* wss, const float* const* d_outs, float* const* grads,
                                      float* const* scratches, void* stream) {
  SPO_REQUIRE(count >= 1 && count <= spo::MLP_SMALL_MAX_NETS && thetas && nets && xs && wss && d_outs && grads && scratches && rows > 0,
              "mlp_backward_multi: bad args");
  bool small = true;
  for (int i = 0; i < count; ++i) {
    SPO_REQUIRE(thetas[i] && nets[i] && xs[i] && wss[i] && d_outs[i] && grads[i] && scratches[i], "mlp_backward_multi: null pointer");
    small = small && spo::mlp_small_ok(nets[i], rows);
  }
  if (!small) {
    for (int i = 0; i < count; ++i)
      if (int rc = spo_mlp_backward(thetas[i], nets[i], xs[i], rows, wss[i], d_outs[i], grads[i], scratches[i], stream)) return rc;
    return 0;
  }
  spo::MlpSmallBatch batch;
  batch.count = count;
  for (int i = 0; i < count; ++i) {
    MlpLay L;
    if (int rc = mlp_lay(nets[i], &L)) return rc;
    spo::mlp_small_args(thetas[i], nets[i], xs[i], rows, const_cast<float*>(wss[i]), d_outs[i], grads[i], &batch.a[i]);
  }
  if (int rc = spo::mlp_small_launch(true, batch, (hipStream_t)stream)) return rc;
  SPO_LAUNCH_CHECK("spo_mlp_backward_multi");
  return 0;
}

// Rows idx[0 .. n) of up to SPO_GATHER_MAX row-major arrays (widths[k] floats per row) in one launch: the minibatch gather of a
// step (ppo_lag.py:298-305: the DataLoader's index_select of obs, act, log_prob, targets, advantage) instead of one launch per array.
namespace {
struct GatherArgs {
  const float* src[SPO_GATHER_MAX];
  float* dst[SPO_GATHER_MAX];
  int width[SPO_GATHER_MAX];
  int count;
  const int64_t* idx;
  const int64_t* cursor;       // optional (device): the window idx[*cursor .. *cursor + n) -- a captured launch then walks a permutation
  int64_t n;
};
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a) {
  const int k = blockIdx.y;
  const int w = a.width[k];
  const float* __restrict__ src = a.src[k];
  float* __restrict__ dst = a.dst[k];
  const int64_t* __restrict__ idx = a.idx + (a.cursor ? a.cursor[0] : 0);
  const int64_t total = a.n * w;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / w;
    const int c = (int)(e - r * w);
    dst[e] = src[idx[r] * w + c];
  }
}
}  // namespace
extern "C" int spo_gather_rows_at(int count, const float* const* srcs, const int* widths, float* const* dsts, const int64_t* idx,
                                  const int64_t* cursor_dev, int64_t n, void* stream);
extern "C" int spo_gather_rows(int count, const float* const* srcs, const int* widths, float* const* dsts, const int64_t* idx, int64_t n,
                               void* stream) {
  return spo_gather_rows_at(count, srcs, widths, dsts, idx, nullptr, n, stream);
}
extern "C" int spo_gather_rows_at(int count, const float* const* srcs, const int* widths, float* const* dsts, const int64_t* idx,
                                  const int64_t* cursor_dev, int64_t n, void* stream) {
  SPO_REQUIRE(count >= 1 && count <= SPO_GATHER_MAX && srcs && widths && dsts && idx && n > 0, "gather_rows: bad args");
  GatherArgs a;
  a.count = count; a.idx = idx; a.cursor = cursor_dev; a.n = n;
  int wmax = 1;
  for (int k = 0; k < count; ++k) {
    SPO_REQUIRE(srcs[k] && dsts[k] && widths[k] >= 1, "gather_rows: bad array %d", k);
    a.src[k] = srcs[k]; a.dst[k] = dsts[k]; a.width[k] = widths[k];
    wmax = widths[k] > wmax ? widths[k] : wmax;
  }
  const int64_t blocks = (n * wmax + 255) / 256;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks), (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
  SPO_LAUNCH_CHECK("spo_gather_rows");
  return 0;
}

extern "C" int spo_wide_critic_loss(const float* v_r, const float* v_c, const float* tgt_r, const float* tgt_c, int64_t rows,
                                    float* d_vr, float* d_vc, float* losses2, double* partial_ws, int partial_capacity, void* stream);
extern "C" int spo_wide_actor_loss(int mode, const float* mean, const float* log_std, const float* act, const float* logp_old,
                                   const float* adv, const float* old_mean, const float* old_std, int64_t rows, int64_t rows_total,
                                   int act_dim, float p0, float p1, float* d_mean, double* sums_inout, int accumulate,
                                   float* loss_out, float* d_log_std_out, double* partial_ws, int partial_capacity, void* stream);
extern "C" int spo_wide_ppo_loss(const float* v_r, const float* v_c, const float* mean, const float* log_std, const float* act,
                                 const float* logp_old, const float* adv, const float* tgt_r, const float* tgt_c, int64_t rows,
                                 int act_dim, float clip, float* d_vr, float* d_vc, float* d_mean, float* d_log_std,
                                 float* losses3, double* partial_ws, int partial_capacity, void* stream) {
  SPO_REQUIRE(v_r && v_c && mean && log_std && act && logp_old && adv && tgt_r && tgt_c && d_vr && d_vc && d_mean && d_log_std &&
                  losses3 && partial_ws && rows > 0, "wide_ppo_loss: bad args");
  SPO_REQUIRE(act_dim >= 1 && act_dim <= SPO_WIDE_MAX_ACT, "wide_ppo_loss: act_dim %d outside [1,%d]", act_dim, SPO_WIDE_MAX_ACT);
  if (act_dim > SPO_MAX_ACT) {
    // wider action vectors (HumanoidVelocity: 17): the lanes-per-row kernels of the round-4 section below;
    // partial_ws = [256 actor rows | 512 critic partials | sums]
    constexpr int64_t need = 256 * (2 + SPO_WIDE_MAX_ACT) + 512 + (2 + SPO_WIDE_MAX_ACT);
    SPO_REQUIRE((int64_t)partial_capacity >= need, "wide_ppo_loss: partial workspace too small (%d < %lld)", partial_capacity, (long long)need);
    double* crit = partial_ws + 256 * (2 + SPO_WIDE_MAX_ACT);
    if (int rc = spo_wide_critic_loss(v_r, v_c, tgt_r, tgt_c, rows, d_vr, d_vc, losses3, crit, 512, stream)) return rc;
    return spo_wide_actor_loss(0, mean, log_std, act, logp_old, adv, nullptr, nullptr, rows, rows, act_dim, clip, 0.f, d_mean, crit + 512, 0,
                               losses3 + 2, d_log_std, partial_ws, 256 * (2 + SPO_WIDE_MAX_ACT), stream);
  }
  int64_t blocks = (rows + 255) / 256;
  if (blocks > 256) blocks = 256;
  SPO_REQUIRE(partial_capacity >= blocks * WL_NS, "wide_ppo_loss: partial workspace too small (%d < %lld)", partial_capacity, (long long)(blocks * WL_NS));
  hipStream_t st = (hipStream_t)stream;
  const bool one = blocks == 1;
  hipLaunchKernelGGL(wide_ppo_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, st, v_r, v_c, mean, log_std, act, logp_old, adv, tgt_r,
                     tgt_c, rows, act_dim, clip, d_vr, d_vc, d_mean, partial_ws, one ? losses3 : (float*)nullptr,
                     one ? d_log_std : (float*)nullptr);
  if (!one) hipLaunchKernelGGL(wide_ppo_loss_finish_kernel, dim3(1), dim3(64), 0, st, partial_ws, (int)blocks, act_dim, rows, losses3, d_log_std);
  SPO_LAUNCH_CHECK("spo_wide_ppo_loss");
  return 0;
}

extern "C" int spo_wide_clip_adam(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                                  int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, int64_t adam_step_host,
                                  float* losses3_inout, float* scalars4_out, double* partial_ws, int partial_capacity, void* stream) {
  SPO_REQUIRE(theta && grad && adam_m && adam_v && cfg && scalars4_out && partial_ws && n_params > 0 && adam_step_host >= 0,
              "wide_clip_adam: bad args");
  SPO_REQUIRE(0 <= reward_critic_end && reward_critic_end <= cost_critic_end && cost_critic_end <= actor_begin && actor_begin <= n_params,
              "wide_clip_adam: parameter ranges out of order");
  int64_t blocks = (n_params + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  SPO_REQUIRE(partial_capacity >= blocks * 3, "wide_clip_adam: partial workspace too small");
  WideAdamArgs a{theta, grad, adam_m, adam_v, n_params, reward_critic_end, cost_critic_end, actor_begin,
                 cfg->use_critic_norm ? cfg->l2_coef : 0.f, cfg->use_value_coefficient ? 2.f : 1.f, cfg->max_grad_norm, cfg->lr_actor,
                 cfg->lr_critic, cfg->beta1, cfg->beta2, cfg->adam_eps, pow((double)cfg->beta1, (double)adam_step_host),
                 pow((double)cfg->beta2, (double)adam_step_host), partial_ws, scalars4_out, losses3_inout};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wide_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(wide_coef_kernel, dim3(1), dim3(64), 0, st, a, (int)blocks);
  hipLaunchKernelGGL(wide_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  SPO_LAUNCH_CHECK("spo_wide_clip_adam");
  return 0;
}

// ====================================================================================================================
// Round 4 -- the wide path as the fallback for ANY (obs_dim, act_dim, hidden_sizes) and for every single-agent script
// (reference model.py:131 takes any dims; benchmark.py:5-22 pairs cpo / pcpo / rcpo / trpo_lag / focops / cup with Car,
// Doggo, Racecar (obs 72-88) and HumanoidVelocity (obs 376, act 17)).  The LDS-resident kernels keep their envelope
// (obs <= 128, act <= 16; CPO full-batch kernels obs <= 64); everything outside it runs here:
//   spo_wide_actor_loss       clipped surrogate | plain surrogate sign*mean(ratio*adv) | KL-penalty (FOCOPS, CUP stage 2)
//   spo_wide_critic_loss      MSE of both critics + output gradients (critic fit, cpo.py:541-571)
//   spo_mlp_jvp               forward-mode tangent d(out)/d(theta).t of the tanh MLP (Fisher-vector product, cpo.py:132-157)
//   spo_wide_fvp_cotangent    (J t) / sigma^2 / (rows_total * act_dim)
//   spo_wide_linesearch_sums  sum ratio*adv_a, sum ratio*adv_b, sum KL(old || new)   (cpo.py:473-491)
//   spo_wide_clip_adam_ex     joint clip over all parameters, Adam on a sub-range with per-optimiser step counts, stale
//                             gradients outside the range rescaled in place (what clip_grad_norm_ does to actor.grad in the
//                             critic fit, cpo.py:557; CUP's actor-only second stage, cup.py:385)
// Actor-side loss kernels use G = pow2ceil(act_dim) lanes per row (act_dim <= SPO_WIDE_MAX_ACT = 64): coalesced loads along the
// action dimension, segmented xor-shuffle sums, no per-thread arrays.
namespace {
using namespace spo;

constexpr int WA_NS = 2 + SPO_WIDE_MAX_ACT;      // partial row of an actor-loss block: loss sum, aux sum, d(log_std)[A]
enum { WA_CLIP = 0, WA_SURR = 1, WA_KLPEN = 2, WA_KLPEN_COUNT = 3 };

__device__ __forceinline__ float group_sum_f(float v, int G) {
  for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum of a double over the lanes of the wave that hold the same action index (lanes k, k + G, k + 2G, ...)
__device__ __forceinline__ double stride_sum_d(double v, int G) {
  for (int o = G; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
inline int pow2ceil(int a) { int g = 1; while (g < a) g <<= 1; return g; }

struct WaArgs {
  const float* mean; const float* log_std; const float* act; const float* logp_old; const float* adv;
  const float* old_mean; const float* old_std;
  int64_t B; int A, G;
  float p0, p1;                 // WA_CLIP: clip, -   WA_SURR: sign, -   WA_KLPEN: kl_bound, pg_coef
  float inv_n;                  // 1 / rows the mean is taken over
  const double* frac_sum;       // WA_KLPEN: device sum of the indicators (from the WA_KLPEN_COUNT pass), mean = frac_sum * inv_n
  float* d_mean; double* partial;
};

template <int MODE>
__global__ __launch_bounds__(256) void wide_actor_loss_kernel(WaArgs a) {
  __shared__ double red[4][WA_NS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = a.A, G = a.G, k = lane & (G - 1), sub = lane / G, rpw = 64 / G;
  const bool on = k < A;
  const float ls = on ? a.log_std[k] : 0.f;
  const float sd = __expf(ls), ivar = 1.f / (sd * sd);
  float iso = 1.f, vrat = 1.f, lvrat = 0.f;
  if (MODE == WA_KLPEN || MODE == WA_KLPEN_COUNT) {
    iso = on ? 1.f / a.old_std[k] : 1.f;
    const float sr = sd * iso;                      // kl_normal_normal: var_ratio = (p.scale / q.scale)^2
    vrat = sr * sr; lvrat = logf(vrat);
  }
  const float clip_lo = 1.f - a.p0, clip_hi = 1.f + a.p0;
  float frac = 0.f;
  if (MODE == WA_KLPEN) frac = (float)a.frac_sum[0] * a.inv_n;
  double s_loss = 0.0, s_aux = 0.0, s_dls = 0.0;
  const int64_t wave_g = (int64_t)blockIdx.x * 4 + wave, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r0 = wave_g * rpw; r0 < a.B; r0 += nwaves * rpw) {
    const int64_t r = r0 + sub;
    const bool rv = r < a.B;
    const int64_t rr = rv ? r : a.B - 1;
    const bool ld = on;
    const float mu = ld ? a.mean[rr * A + k] : 0.f;
    const float ac = ld ? a.act[rr * A + k] : 0.f;
    const float dif = ac - mu;
    const float term = on ? -(dif * dif) * (0.5f * ivar) - ls - LOG_SQRT_2PI_F : 0.f;
    const float lp = group_sum_f(term, G);
    float kl = 0.f, dm = 0.f;
    if (MODE == WA_KLPEN || MODE == WA_KLPEN_COUNT) {
      dm = on ? (mu - a.old_mean[rr * A + k]) * iso : 0.f;          // (loc_p - loc_q) / scale_q
      kl = group_sum_f(on ? 0.5f * (vrat + dm * dm - 1.f - lvrat) : 0.f, G);
    }
    if (MODE == WA_KLPEN_COUNT) {
      if (rv && k == 0) s_aux += (kl <= a.p0) ? 1.0 : 0.0;
      continue;
    }
    const float ad = a.adv[rr];
    const float ratio = __expf(lp - a.logp_old[rr]);
    float dlp, wk = 0.f;
    if (MODE == WA_CLIP) {
      const float rc = fminf(fmaxf(ratio, clip_lo), clip_hi);
      const float s1 = ratio * ad, s2 = rc * ad;
      const bool inr = (ratio >= clip_lo) && (ratio <= clip_hi);
      float gr;                                        // backward of torch.min / torch.clamp (ties split the gradient)
      if (s1 < s2) gr = ad;
      else if (s1 > s2) gr = inr ? ad : 0.f;
      else gr = 0.5f * ad + (inr ? 0.5f * ad : 0.f);
      dlp = -(gr * ratio) * a.inv_n;
      if (rv && k == 0) s_loss += (double)fminf(s1, s2);
    } else if (MODE == WA_SURR) {
      dlp = a.p0 * ad * ratio * a.inv_n;               // d(sign * mean(ratio * adv)) / d logp
      if (rv && k == 0) s_loss += (double)(ratio * ad);
    } else {
      const float ind = (kl <= a.p0) ? 1.f : 0.f;
      const float pg = a.p1 * frac;
      dlp = -(pg * ad * ratio) * a.inv_n;
      wk = ind * a.inv_n;
      if (rv && k == 0) { s_loss += (double)(pg * ratio * ad - ind * kl); s_aux += (double)ind; }      // loss = -mean(this)
    }
    if (rv && on) {
      const float z = dif * ivar;
      float dmu = dlp * z, dl = dlp * (dif * z - 1.f);
      if (MODE == WA_KLPEN) { dmu = fmaf(dlp, z, wk * dm * iso); dl = fmaf(dlp, dif * z - 1.f, wk * (vrat - 1.f)); }
      a.d_mean[r * A + k] = dmu;
      s_dls += (double)dl;
    }
  }
  s_loss = wave_sum_d(s_loss); s_aux = wave_sum_d(s_aux);
  s_dls = stride_sum_d(s_dls, G);
  if (lane == 0) { red[wave][0] = s_loss; red[wave][1] = s_aux; }
  if (lane < G && lane < A) red[wave][2 + lane] = s_dls;
  __syncthreads();
  if (tid < 2 + A) a.partial[(int64_t)blockIdx.x * WA_NS + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
// sums[0] = loss sum, sums[1] = aux sum, sums[2 + k] = d(log_std)[k]; optional float outputs for the unchunked minibatch use
__global__ void wide_actor_loss_finish_kernel(const double* __restrict__ partial, int nblocks, int A, double* __restrict__ sums,
                                              int accumulate, float loss_scale, float* __restrict__ loss_out,
                                              float* __restrict__ d_log_std_out) {
  const int k = threadIdx.x;
  if (k >= 2 + A) return;
  double s = accumulate ? sums[k] : 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * WA_NS + k];
  sums[k] = s;
  if (k == 0 && loss_out) loss_out[0] = (float)(s * (double)loss_scale);
  if (k >= 2 && d_log_std_out) d_log_std_out[k - 2] = (float)s;
}

__global__ __launch_bounds__(256) void wide_critic_loss_kernel(const float* __restrict__ v_r, const float* __restrict__ v_c,
                                                               const float* __restrict__ tgt_r, const float* __restrict__ tgt_c, int64_t B,
                                                               float inv_n, float* __restrict__ d_vr, float* __restrict__ d_vc,
                                                               double* __restrict__ partial, float* __restrict__ losses) {
  // losses: given for a one-workgroup launch, which finishes the sums itself (see wide_ppo_loss_kernel)
  __shared__ double red[4][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a0 = 0.0, a1 = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + tid; r < B; r += (int64_t)gridDim.x * 256) {
    const float dr = v_r[r] - tgt_r[r], dc = v_c[r] - tgt_c[r];
    a0 += (double)(dr * dr); a1 += (double)(dc * dc);
    d_vr[r] = 2.f * dr * inv_n; d_vc[r] = 2.f * dc * inv_n;
  }
  a0 = wave_sum_d(a0); a1 = wave_sum_d(a1);
  if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; }
  __syncthreads();
  if (tid < 2) {
    const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    partial[(int64_t)blockIdx.x * 2 + tid] = v;
    if (losses) losses[tid] = (float)((0.0 + v) / (double)B);
  }
}
__global__ void wide_critic_loss_finish_kernel(const double* __restrict__ partial, int nblocks, int64_t B, float* __restrict__ losses) {
  const int k = threadIdx.x;
  if (k >= 2) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * 2 + k];
  losses[k] = (float)(s / (double)B);
}

// tangent epilogue of one layer: y = (y + tb) * (1 - h^2)  (hidden layers) or y + tb (output layer)
__global__ __launch_bounds__(256) void mlp_jvp_ew_kernel(float* __restrict__ y, const float* __restrict__ tb, const float* __restrict__ h,
                                                         int64_t n, int N) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = y[i] + tb[i % N];
    if (h) { const float hv = h[i]; v = v * fmaf(-hv, hv, 1.f); }
    y[i] = v;
  }
}
__global__ __launch_bounds__(256) void wide_fvp_cot_kernel(const float* __restrict__ jv, const float* __restrict__ log_std, int64_t B, int A,
                                                           float inv_ma, float* __restrict__ d_mean) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B * A; i += (int64_t)gridDim.x * 256) {
    const float sd = expf(log_std[i % A]);
    d_mean[i] = jv[i] * (1.f / (sd * sd)) * inv_ma;
  }
}

struct WlsArgs {
  const float* mean; const float* log_std; const float* act; const float* logp_old; const float* adv_a; const float* adv_b;
  const float* mean_old; const float* log_std_old; int64_t B; int A, G; double* partial;
};
__global__ __launch_bounds__(256) void wide_linesearch_kernel(WlsArgs a) {
  __shared__ double red[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = a.A, G = a.G, k = lane & (G - 1), sub = lane / G, rpw = 64 / G;
  const bool on = k < A;
  const float lsv = on ? a.log_std[k] : 0.f;
  const float sdn = expf(lsv), sdo = on ? expf(a.log_std_old[k]) : 1.f;
  const float ivar = 1.f / (sdn * sdn), lsd = on ? logf(sdn) + LOG_SQRT_2PI_F : 0.f;
  const float sr = sdo / sdn, vr = sr * sr, lvr = logf(vr);
  double s_a = 0.0, s_b = 0.0, s_kl = 0.0;
  const int64_t wave_g = (int64_t)blockIdx.x * 4 + wave, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r0 = wave_g * rpw; r0 < a.B; r0 += nwaves * rpw) {
    const int64_t r = r0 + sub;
    const bool rv = r < a.B;
    const int64_t rr = rv ? r : a.B - 1;
    const float mu = on ? a.mean[rr * A + k] : 0.f;
    const float dif = (on ? a.act[rr * A + k] : 0.f) - mu;
    const float lp = group_sum_f(on ? -(dif * dif) * (0.5f * ivar) - lsd : 0.f, G);
    const float ratio = expf(lp - a.logp_old[rr]);
    if (rv && on) {
      const float dm = (a.mean_old[rr * A + k] - mu) / sdn;
      s_kl += (double)(0.5f * (vr + dm * dm - 1.f - lvr));
    }
    if (rv && k == 0) { s_a += (double)(ratio * a.adv_a[rr]); s_b += (double)(ratio * a.adv_b[rr]); }
  }
  s_a = wave_sum_d(s_a); s_b = wave_sum_d(s_b); s_kl = wave_sum_d(s_kl);
  if (lane == 0) { red[wave][0] = s_a; red[wave][1] = s_b; red[wave][2] = s_kl; }
  __syncthreads();
  if (tid < 3) a.partial[(int64_t)blockIdx.x * 3 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
__global__ void wide_sum3_kernel(const double* __restrict__ partial, int n, double* __restrict__ out, int accumulate) {
  if (threadIdx.x < 3) {
    double s = accumulate ? out[threadIdx.x] : 0.0;
    for (int b = 0; b < n; ++b) s += partial[(int64_t)b * 3 + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// clip + Adam with a sub-range: prep and coefficient as wide_prep_kernel / wide_coef_kernel (the joint norm spans all
// parameters); Adam touches [adam_begin, adam_end) only, with the critics' clock below actor_begin and the actor's above;
// outside that range the (stale) gradient is multiplied by the clip coefficient in place when scale_rest != 0.
struct WideAdamExArgs {
  WideAdamArgs b; int64_t adam_begin, adam_end; int scale_rest; double pow_b1_a, pow_b2_a;
};
__global__ __launch_bounds__(256) void wide_adam_ex_kernel(WideAdamExArgs x) {
  const WideAdamArgs& a = x.b;
  const float coef = a.scal[0];
  float ss_a, ss_c, bc2s_a, bc2s_c;
  adam_scalars(a.lr_actor, x.pow_b1_a * (double)a.b1, x.pow_b2_a * (double)a.b2, ss_a, bc2s_a);
  adam_scalars(a.lr_critic, a.pow_b1 * (double)a.b1, a.pow_b2 * (double)a.b2, ss_c, bc2s_c);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.P; i += (int64_t)gridDim.x * 256) {
    if (i >= x.adam_begin && i < x.adam_end) {
      const bool act = i >= a.actor_begin;
      const AdamOut o = adam1(a.theta[i], a.grad[i] * coef, a.m[i], a.v[i], a.b1, a.b2, a.eps, act ? ss_a : ss_c, act ? bc2s_a : bc2s_c);
      a.theta[i] = o.p; a.m[i] = o.m; a.v[i] = o.v;
    } else if (x.scale_rest) {
      a.grad[i] = a.grad[i] * coef;
    }
  }
}
// prep over a sub-range for the joint norm: CUP's second stage clips over the ACTOR's parameters only (cup.py:385)
__global__ __launch_bounds__(256) void wide_prep_range_kernel(WideAdamArgs a, int64_t n0, int64_t n1) {
  __shared__ double red[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double gs = 0.0;
  for (int64_t i = n0 + (int64_t)blockIdx.x * 256 + tid; i < n1; i += (int64_t)gridDim.x * 256) {
    const float g = a.grad[i];
    gs += (double)(g * g);
  }
  gs = wave_sum_d(gs);
  if (lane == 0) { red[wave][0] = gs; red[wave][1] = 0.0; red[wave][2] = 0.0; }
  __syncthreads();
  if (tid < 3) a.partial[(int64_t)blockIdx.x * 3 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
}  // namespace

extern "C" int spo_wide_actor_loss(int mode, const float* mean, const float* log_std, const float* act, const float* logp_old,
                                   const float* adv, const float* old_mean, const float* old_std, int64_t rows, int64_t rows_total,
                                   int act_dim, float p0, float p1, float* d_mean, double* sums_inout, int accumulate,
                                   float* loss_out, float* d_log_std_out, double* partial_ws, int partial_capacity, void* stream) {
  SPO_REQUIRE(mode >= WA_CLIP && mode <= WA_KLPEN, "wide_actor_loss: mode %d outside [0,2]", mode);
  SPO_REQUIRE(mean && log_std && act && logp_old && adv && d_mean && sums_inout && partial_ws && rows > 0 && rows_total >= rows,
              "wide_actor_loss: bad args");
  SPO_REQUIRE(act_dim >= 1 && act_dim <= SPO_WIDE_MAX_ACT, "wide_actor_loss: act_dim %d outside [1,%d]", act_dim, SPO_WIDE_MAX_ACT);
  SPO_REQUIRE(mode != WA_KLPEN || (old_mean && old_std), "wide_actor_loss: the KL-penalty loss needs old_mean / old_std");
  const int G = pow2ceil(act_dim);
  const int64_t rpb = 4 * (64 / G);
  int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > 256) blocks = 256;
  SPO_REQUIRE((int64_t)partial_capacity >= blocks * WA_NS, "wide_actor_loss: partial workspace too small (%d < %lld)", partial_capacity,
              (long long)(blocks * WA_NS));
  hipStream_t st = (hipStream_t)stream;
  WaArgs a{mean, log_std, act, logp_old, adv, old_mean, old_std, rows, act_dim, G, p0, p1, 1.f / (float)rows_total, sums_inout + 1, d_mean,
           partial_ws};
  float loss_scale = 1.f;
  if (mode == WA_CLIP) {
    hipLaunchKernelGGL(wide_actor_loss_kernel<WA_CLIP>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    loss_scale = -1.f / (float)rows_total;
  } else if (mode == WA_SURR) {
    hipLaunchKernelGGL(wide_actor_loss_kernel<WA_SURR>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    loss_scale = 1.f / (float)rows_total;
  } else {
    // pass 1: the indicator count (the loss couples every row to mean_i(ind_i), focops.py:331); pass 2 reads it from sums[1]
    SPO_REQUIRE(!accumulate && rows == rows_total, "wide_actor_loss: the KL-penalty loss is a minibatch loss (no row chunks)");
    hipLaunchKernelGGL(wide_actor_loss_kernel<WA_KLPEN_COUNT>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(wide_actor_loss_finish_kernel, dim3(1), dim3(128), 0, st, partial_ws, (int)blocks, 0, sums_inout, 0, 0.f,
                       (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(wide_actor_loss_kernel<WA_KLPEN>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    loss_scale = -1.f / (float)rows_total;
  }
  hipLaunchKernelGGL(wide_actor_loss_finish_kernel, dim3(1), dim3(128), 0, st, partial_ws, (int)blocks, act_dim, sums_inout, accumulate,
                     loss_scale, loss_out, d_log_std_out);
  SPO_LAUNCH_CHECK("spo_wide_actor_loss");
  return 0;
}

extern "C" int spo_wide_critic_loss(const float* v_r, const float* v_c, const float* tgt_r, const float* tgt_c, int64_t rows,
                                    float* d_vr, float* d_vc, float* losses2, double* partial_ws, int partial_capacity, void* stream) {
  SPO_REQUIRE(v_r && v_c && tgt_r && tgt_c && d_vr && d_vc && losses2 && partial_ws && rows > 0, "wide_critic_loss: bad args");
  int64_t blocks = (rows + 255) / 256;
  if (blocks > 256) blocks = 256;
  SPO_REQUIRE((int64_t)partial_capacity >= blocks * 2, "wide_critic_loss: partial workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wide_critic_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, st, v_r, v_c, tgt_r, tgt_c, rows, 1.f / (float)rows, d_vr,
                     d_vc, partial_ws, blocks == 1 ? losses2 : (float*)nullptr);
  if (blocks > 1) hipLaunchKernelGGL(wide_critic_loss_finish_kernel, dim3(1), dim3(64), 0, st, partial_ws, (int)blocks, rows, losses2);
  SPO_LAUNCH_CHECK("spo_wide_critic_loss");
  return 0;
}

extern "C" int64_t spo_mlp_jvp_scratch_floats(const spo_mlp_net* net, int64_t rows) {
  MlpLay L;
  if (mlp_lay(net, &L) || rows < 1) return -1;
  return 2 * rows * (int64_t)L.maxdim();
}
extern "C" int spo_mlp_jvp(const float* theta, const spo_mlp_net* net, const float* tangent, const float* x, int64_t rows,
                           const float* ws, float* dout, float* scratch, void* stream) {
  MlpLay L;
  if (int rc = mlp_lay(net, &L)) return rc;
  SPO_REQUIRE(theta && tangent && x && ws && dout && scratch && rows > 0, "mlp_jvp: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int md = L.maxdim();
  float* bufA = scratch;
  float* bufB = scratch + rows * (int64_t)md;
  const float* dh = nullptr;                      // tangent of the layer input (0 for the observations)
  for (int l = 0; l < L.n; ++l) {
    const int K = L.d[l], N = L.d[l + 1];
    const bool last = l + 1 == L.n;
    const float* hin = l == 0 ? x : ws + L.act_off(l - 1, rows);
    float* y = last ? dout : (dh == bufA ? bufB : bufA);
    // dz_l = h_{l-1} tW_l^T + dh_{l-1} W_l^T + tb_l
    if (int rc = gemm_xwT(st, hin, tangent + L.w(l), y, rows, K, N, 0.f)) return rc;
    if (dh) { if (int rc = gemm_xwT(st, dh, theta + L.w(l), y, rows, K, N, 1.f)) return rc; }
    const int64_t n = rows * (int64_t)N;
    hipLaunchKernelGGL(mlp_jvp_ew_kernel, dim3(ew_grid(n)), dim3(256), 0, st, y, tangent + L.b(l),
                       last ? (const float*)nullptr : ws + L.act_off(l, rows), n, N);
    dh = y;
  }
  SPO_LAUNCH_CHECK("spo_mlp_jvp");
  return 0;
}

extern "C" int spo_wide_fvp_cotangent(const float* jv, const float* log_std, int64_t rows, int64_t rows_total, int act_dim,
                                      float* d_mean, void* stream) {
  SPO_REQUIRE(jv && log_std && d_mean && rows > 0 && rows_total >= rows && act_dim >= 1 && act_dim <= SPO_WIDE_MAX_ACT,
              "wide_fvp_cotangent: bad args");
  const float inv_ma = (float)(1.0 / ((double)rows_total * (double)act_dim));
  hipLaunchKernelGGL(wide_fvp_cot_kernel, dim3(ew_grid(rows * act_dim)), dim3(256), 0, (hipStream_t)stream, jv, log_std, rows, act_dim, inv_ma,
                     d_mean);
  SPO_LAUNCH_CHECK("spo_wide_fvp_cotangent");
  return 0;
}

extern "C" int spo_wide_linesearch_sums(const float* mean_new, const float* log_std_new, const float* act, const float* logp_old,
                                        const float* adv_a, const float* adv_b, const float* mean_old, const float* log_std_old,
                                        int64_t rows, int act_dim, double* partial_ws, int partial_capacity, double* sums3_inout,
                                        int accumulate, void* stream) {
  SPO_REQUIRE(mean_new && log_std_new && act && logp_old && adv_a && adv_b && mean_old && log_std_old && partial_ws && sums3_inout &&
                  rows > 0 && act_dim >= 1 && act_dim <= SPO_WIDE_MAX_ACT, "wide_linesearch_sums: bad args");
  const int G = pow2ceil(act_dim);
  const int64_t rpb = 4 * (64 / G);
  int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > 512) blocks = 512;
  if (blocks * 3 > partial_capacity) blocks = partial_capacity / 3;
  SPO_REQUIRE(blocks >= 1, "wide_linesearch_sums: partial capacity too small");
  WlsArgs a{mean_new, log_std_new, act, logp_old, adv_a, adv_b, mean_old, log_std_old, rows, act_dim, G, partial_ws};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wide_linesearch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(wide_sum3_kernel, dim3(1), dim3(64), 0, st, partial_ws, (int)blocks, sums3_inout, accumulate);
  SPO_LAUNCH_CHECK("spo_wide_linesearch_sums");
  return 0;
}

// Device-resident optimiser clocks: pow4_dev = double[6] = {beta1^t, beta2^t of the critics' optimisers, beta1^t, beta2^t of the actor's}
// BEFORE this step, then {lr_actor, lr_critic} (negative: take the cfg's).  The clocks of the optimisers inside [adam_begin, adam_end) advance by one step on the device (one thread, between the
// norm and the Adam pass), so the launch sequence of a minibatch step has no host-side argument that changes from step to step and
// can be captured once as a HIP graph and replayed (the wide path at small batches is launch-bound: ~70 launches per step).
namespace {
using namespace spo;
__global__ __launch_bounds__(256) void wide_adam_dev_kernel(WideAdamExArgs x, const double* __restrict__ pow4) {
  const WideAdamArgs& a = x.b;
  const float coef = a.scal[0];
  float ss_a, ss_c, bc2s_a, bc2s_c;
  // learning rates: device-resident next to the clocks when set (> 0) -- the actor's follows a per-epoch schedule, and a value baked
  // into a captured launch would make every epoch capture a new graph
  // (a NEGATIVE entry means "take the cfg's": a scheduled or frozen learning rate of exactly 0 is honoured -- ABI 2)
  const float lr_a = pow4[4] >= 0.0 ? (float)pow4[4] : a.lr_actor, lr_c = pow4[5] >= 0.0 ? (float)pow4[5] : a.lr_critic;
  adam_scalars(lr_a, pow4[2], pow4[3], ss_a, bc2s_a);                 // pow4 already advanced: beta^(t+1)
  adam_scalars(lr_c, pow4[0], pow4[1], ss_c, bc2s_c);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.P; i += (int64_t)gridDim.x * 256) {
    if (i >= x.adam_begin && i < x.adam_end) {
      const bool act = i >= a.actor_begin;
      const AdamOut o = adam1(a.theta[i], a.grad[i] * coef, a.m[i], a.v[i], a.b1, a.b2, a.eps, act ? ss_a : ss_c, act ? bc2s_a : bc2s_c);
      a.theta[i] = o.p; a.m[i] = o.m; a.v[i] = o.v;
    } else if (x.scale_rest) {
      a.grad[i] = a.grad[i] * coef;
    }
  }
}
}  // namespace
extern "C" int spo_wide_clip_adam_dev_log(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                                          int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, double* pow4_dev,
                                          int64_t adam_begin, int64_t adam_end, int64_t norm_begin, int scale_rest, float* losses3_inout,
                                          float* scalars4_out, double* partial_ws, int partial_capacity, float* loss_log_dev,
                                          const float* log_src_dev, int64_t* cursor_dev, int64_t cursor_step, void* stream);
extern "C" int spo_wide_clip_adam_dev(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                                      int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, double* pow4_dev,
                                      int64_t adam_begin, int64_t adam_end, int64_t norm_begin, int scale_rest, float* losses3_inout,
                                      float* scalars4_out, double* partial_ws, int partial_capacity, void* stream) {
  return spo_wide_clip_adam_dev_log(theta, grad, adam_m, adam_v, n_params, reward_critic_end, cost_critic_end, actor_begin, cfg, pow4_dev,
                                    adam_begin, adam_end, norm_begin, scale_rest, losses3_inout, scalars4_out, partial_ws, partial_capacity,
                                    nullptr, nullptr, nullptr, 0, stream);
}
extern "C" int spo_wide_clip_adam_dev_log(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                                          int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, double* pow4_dev,
                                          int64_t adam_begin, int64_t adam_end, int64_t norm_begin, int scale_rest, float* losses3_inout,
                                          float* scalars4_out, double* partial_ws, int partial_capacity, float* loss_log_dev,
                                          const float* log_src_dev, int64_t* cursor_dev, int64_t cursor_step, void* stream) {
  SPO_REQUIRE(!cursor_dev || cursor_step > 0, "wide_clip_adam_dev_log: cursor_step must be > 0");
  SPO_REQUIRE(theta && grad && adam_m && adam_v && cfg && pow4_dev && scalars4_out && partial_ws && n_params > 0, "wide_clip_adam_dev: bad args");
  SPO_REQUIRE(0 <= reward_critic_end && reward_critic_end <= cost_critic_end && cost_critic_end <= actor_begin && actor_begin <= n_params,
              "wide_clip_adam_dev: parameter ranges out of order");
  SPO_REQUIRE(0 <= adam_begin && adam_begin <= adam_end && adam_end <= n_params && 0 <= norm_begin && norm_begin <= n_params,
              "wide_clip_adam_dev: optimiser range out of order");
  int64_t blocks = (n_params + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  SPO_REQUIRE((int64_t)partial_capacity >= blocks * 3, "wide_clip_adam_dev: partial workspace too small");
  WideAdamArgs a{theta, grad, adam_m, adam_v, n_params, reward_critic_end, cost_critic_end, actor_begin,
                 cfg->use_critic_norm ? cfg->l2_coef : 0.f, cfg->use_value_coefficient ? 2.f : 1.f, cfg->max_grad_norm, cfg->lr_actor,
                 cfg->lr_critic, cfg->beta1, cfg->beta2, cfg->adam_eps, 1.0, 1.0, partial_ws, scalars4_out, losses3_inout};
  WideAdamExArgs x{a, adam_begin, adam_end, scale_rest, 1.0, 1.0};
  hipStream_t st = (hipStream_t)stream;
  if (norm_begin == 0) hipLaunchKernelGGL(wide_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wide_prep_range_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, norm_begin, n_params);
  hipLaunchKernelGGL(wide_coef_kernel, dim3(1), dim3(64), 0, st, a, (int)blocks, pow4_dev, adam_begin < actor_begin ? 1 : 0,
                     adam_end > actor_begin ? 1 : 0, loss_log_dev, log_src_dev, cursor_dev, cursor_step);
  hipLaunchKernelGGL(wide_adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (const double*)pow4_dev);
  SPO_LAUNCH_CHECK("spo_wide_clip_adam_dev");
  return 0;
}

extern "C" int spo_wide_clip_adam_ex(float* theta, float* grad, float* adam_m, float* adam_v, int64_t n_params, int64_t reward_critic_end,
                                     int64_t cost_critic_end, int64_t actor_begin, const spo_ppo_cfg* cfg, int64_t adam_step_critics_host,
                                     int64_t adam_step_actor_host, int64_t adam_begin, int64_t adam_end, int64_t norm_begin,
                                     int scale_rest, float* losses3_inout, float* scalars4_out, double* partial_ws,
                                     int partial_capacity, void* stream) {
  SPO_REQUIRE(theta && grad && adam_m && adam_v && cfg && scalars4_out && partial_ws && n_params > 0 && adam_step_critics_host >= 0 &&
                  adam_step_actor_host >= 0, "wide_clip_adam_ex: bad args");
  SPO_REQUIRE(0 <= reward_critic_end && reward_critic_end <= cost_critic_end && cost_critic_end <= actor_begin && actor_begin <= n_params,
              "wide_clip_adam_ex: parameter ranges out of order");
  SPO_REQUIRE(0 <= adam_begin && adam_begin <= adam_end && adam_end <= n_params && 0 <= norm_begin && norm_begin <= n_params,
              "wide_clip_adam_ex: optimiser range out of order");
  int64_t blocks = (n_params + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  SPO_REQUIRE((int64_t)partial_capacity >= blocks * 3, "wide_clip_adam_ex: partial workspace too small");
  WideAdamArgs a{theta, grad, adam_m, adam_v, n_params, reward_critic_end, cost_critic_end, actor_begin,
                 cfg->use_critic_norm ? cfg->l2_coef : 0.f, cfg->use_value_coefficient ? 2.f : 1.f, cfg->max_grad_norm, cfg->lr_actor,
                 cfg->lr_critic, cfg->beta1, cfg->beta2, cfg->adam_eps, pow((double)cfg->beta1, (double)adam_step_critics_host),
                 pow((double)cfg->beta2, (double)adam_step_critics_host), partial_ws, scalars4_out, losses3_inout};
  WideAdamExArgs x{a, adam_begin, adam_end, scale_rest, pow((double)cfg->beta1, (double)adam_step_actor_host),
                   pow((double)cfg->beta2, (double)adam_step_actor_host)};
  hipStream_t st = (hipStream_t)stream;
  if (norm_begin == 0) hipLaunchKernelGGL(wide_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wide_prep_range_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, norm_begin, n_params);
  hipLaunchKernelGGL(wide_coef_kernel, dim3(1), dim3(64), 0, st, a, (int)blocks);
  hipLaunchKernelGGL(wide_adam_ex_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x);
  SPO_LAUNCH_CHECK("spo_wide_clip_adam_ex");
  return 0;
}
