// Microbenchmark (one workgroup, 512 threads = 2 waves per SIMD): does a wave's v_mfma_f32_16x16x4_f32 stream overlap with the
// partner wave's VALU stream on the same SIMD?  Cases: MFMA waves alone, VALU waves alone, both; the same with bf16 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
template <int KIND>   // 0: f32 mfma 16x16x4, 1: bf16 mfma 16x16x32
__device__ __forceinline__ void mfma_loop(int iters, float* out) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const float a = threadIdx.x * 1e-3f, b = 1.0f;
  s8v av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (KIND == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
      else acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, av),
                                                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bv), acc[k], 0, 0, 0);
    }
  }
  out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
__device__ __forceinline__ void valu_loop(int iters, float* out) {
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 1e-4f + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = fmaf(x[k], 1.0001f, 0.5f);
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[threadIdx.x] = s;
}
// issue cost of single instruction kinds (one wave per SIMD, 8 independent chains): plain FMA, packed FMA, exp2, rcp, sqrt
template <int OP>
__global__ __launch_bounds__(256) void probe_op(int iters, float* out, unsigned long long* cyc) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  float x[8];
  f2 y[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { x[k] = 1.0f + threadIdx.x * 1e-4f + k; y[k] = f2{x[k], x[k] + 0.5f}; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (OP == 0) x[k] = fmaf(x[k], 1.0001f, 0.5f);
      if (OP == 1) y[k] = __builtin_elementwise_fma(y[k], f2{1.0001f, 0.9999f}, f2{0.5f, 0.25f});
      if (OP == 2) x[k] = __builtin_amdgcn_exp2f(x[k] * 0.01f);
      if (OP == 3) x[k] = __builtin_amdgcn_rcpf(x[k]) + 1.0f;
      if (OP == 4) x[k] = __builtin_amdgcn_sqrtf(x[k]) + 1.0f;
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k] + y[k].x + y[k].y;
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
__global__ __launch_bounds__(512) void probe(int mode, int mi, int vi, float* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool helper = wave >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (!helper) { if (mode & 1) mfma_loop<KIND>(mi, out); if (mode & 4) valu_loop(vi, out); }
  else { if (mode & 2) valu_loop(vi, out); }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// single wave per SIMD doing both streams interleaved in one instruction stream
template <int KIND>
__global__ __launch_bounds__(256) void probe_one(int mi, float* out, unsigned long long* cyc) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 1e-4f + k;
  const float a = threadIdx.x * 1e-3f, b = 1.0f;
  s8v av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {1, 1, 1, 1, 1, 1, 1, 1};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < mi; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (KIND == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
      else acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, av),
                                                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bv), acc[k], 0, 0, 0);
      x[2 * k] = fmaf(x[2 * k], 1.0001f, 0.5f); x[2 * k + 1] = fmaf(x[2 * k + 1], 1.0001f, 0.5f);       // 2 VALU per MFMA
      x[2 * k] = fmaf(x[2 * k], 1.0001f, 0.5f); x[2 * k + 1] = fmaf(x[2 * k + 1], 1.0001f, 0.5f);       // 4 VALU per MFMA
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc((void**)&out, 4096); hipMalloc((void**)&cyc, 8);
  const int MI = 4000, VI = 4000;   // 16000 MFMAs per wave; 32000 VALU per wave
  for (int kind = 0; kind < 2; ++kind) {
    for (int mode = 1; mode <= 3; ++mode) {
      unsigned long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(512), 0, 0, mode, MI, VI, out, cyc);
        else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(512), 0, 0, mode, MI, VI, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      }
      printf("%s two waves/SIMD mode %d (1=mfma wave only, 2=valu wave only, 3=both): %llu cycles; per MFMA %.1f, per VALU %.2f\n",
             kind ? "bf16 16x16x32" : "f32 16x16x4", mode, h, (double)h / (4.0 * MI), (double)h / (8.0 * VI));
    }
    unsigned long long h = 0;
    if (kind == 0) hipLaunchKernelGGL(probe_one<0>, dim3(1), dim3(256), 0, 0, MI, out, cyc);
    else hipLaunchKernelGGL(probe_one<1>, dim3(1), dim3(256), 0, 0, MI, out, cyc);
    hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s ONE wave, 4 independent VALU after every MFMA: %llu cycles; per MFMA(+4 VALU) %.1f\n", kind ? "bf16 16x16x32" : "f32 16x16x4", h, (double)h / (4.0 * MI));
  }
  {
    const char* names[5] = {"v_fma_f32", "v_pk_fma_f32", "v_mul + v_exp_f32", "v_rcp_f32 + v_add", "v_sqrt_f32 + v_add"};
    for (int op = 0; op < 5; ++op) {
      unsigned long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (op == 0) hipLaunchKernelGGL(probe_op<0>, dim3(1), dim3(256), 0, 0, VI, out, cyc);
        if (op == 1) hipLaunchKernelGGL(probe_op<1>, dim3(1), dim3(256), 0, 0, VI, out, cyc);
        if (op == 2) hipLaunchKernelGGL(probe_op<2>, dim3(1), dim3(256), 0, 0, VI, out, cyc);
        if (op == 3) hipLaunchKernelGGL(probe_op<3>, dim3(1), dim3(256), 0, 0, VI, out, cyc);
        if (op == 4) hipLaunchKernelGGL(probe_op<4>, dim3(1), dim3(256), 0, 0, VI, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      }
      printf("one wave per SIMD, 8 independent chains of %s: %.2f cycles per loop body element\n", names[op], (double)h / (8.0 * VI));
    }
  }
  {
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe<0>, dim3(1), dim3(512), 0, 0, 6, MI, VI, out, cyc);
      hipDeviceSynchronize();
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("TWO VALU waves per SIMD (mode 6: both waves stream v_fma_f32): %llu cycles; per VALU of one wave %.2f (one VALU wave alone: mode 2 above)\n",
           h, (double)h / (8.0 * VI));
  }
  printf("note: s_memtime/readcyclecounter counts at a fixed 100 MHz-multiple clock on some parts; compare ratios\n");
  return 0;
}
