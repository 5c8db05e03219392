// probe: does a 16-byte {f0,f1,f2,tag} word written with global_store_dwordx4 sc0 sc1 become visible to a polling
// global_load_dwordx4 sc0 sc1 in another workgroup of the same grid?  (uncached and ordinary device memory)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(char* p, const u4v v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u4v ld16(const char* p) { u4v v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__global__ void probe(char* buf, int iters, int* out) {
  const int tid = threadIdx.x, role = blockIdx.x;   // block 0 <-> block 1 ping-pong
  char* mine = buf + role * 65536 + tid * 16;
  char* peer = buf + (1 - role) * 65536 + tid * 16;
  int bad = 0, timeouts = 0;
  for (int it = 1; it <= iters; ++it) {
    u4v w = {(unsigned)(it * 3 + tid), (unsigned)(it ^ tid), (unsigned)role, (unsigned)it};
    st16(peer, w);
    unsigned spins = 0;
    u4v x;
    for (;;) {
      x = ld16(mine);
      if (x[3] == (unsigned)it) break;
      if (++spins > (1u << 20)) { timeouts++; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (x[0] != (unsigned)(it * 3 + tid) || x[1] != (unsigned)(it ^ tid) || x[2] != (unsigned)(1 - role)) bad++;
  }
  atomicAdd(out + 0, bad); atomicAdd(out + 1, timeouts);
}
int main() {
  for (int kind = 0; kind < 2; ++kind) {
    char* buf; int* out;
    if (kind == 0) hipExtMallocWithFlags((void**)&buf, 131072, hipDeviceMallocUncached); else hipMalloc((void**)&buf, 131072);
    hipMemset(buf, 0, 131072);
    hipMalloc((void**)&out, 8); hipMemset(out, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(2), dim3(256), 0, 0, buf, 2000, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("kind %s: bad %d timeouts %d, %.2f us per round trip, err %s\n", kind ? "hipMalloc" : "uncached", h[0], h[1], ms * 1e3 / 2000, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
