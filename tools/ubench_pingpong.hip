// Micro-benchmark (dev tool, not product): one-way latency of a tagged 8-byte granule hand-off between two workgroups
// on gfx950, by the XCD placement of the pair (same XCD: blocks 0 and 8; different: blocks 0 and 1 -- workgroups go to
// the XCDs round-robin, verified with HW_REG_XCC_ID) and by the cache-scope bits of the store and of the polling load.
//   build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_pingpong.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr unsigned LIMIT = 1u << 22;

template <int ST> __device__ __forceinline__ void st8(u64 *p, u64 v) {
  if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LD> __device__ __forceinline__ u64 ld8(const u64 *p) {
  u64 v;
  if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// slots[0]: written by the first block, slots[16]: by the second (separate 128-byte lines)
template <int ST, int LD> __global__ void k_pp(u64 *slots, int peer, int n, int *err, unsigned *xcc) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[b] = id & 15;
  if (b != 0 && b != peer) return;
  u64 *mine = slots + (b == 0 ? 0 : 16), *theirs = slots + (b == 0 ? 16 : 0);
  for (int i = 1; i <= n; ++i) {
    if (b == 0) st8<ST>(mine, (u64)i);
    unsigned spins = 0;
    while (ld8<LD>(theirs) != (u64)i)
      if (++spins > LIMIT) { *err = 1; return; }
    if (b != 0) st8<ST>(mine, (u64)i);
  }
}
template <int ST, int LD> int run(const char *what, u64 *slots, int *err, unsigned *xcc, int peer) {
  const int n = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f; int e = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(slots, 0, 64 * 8)); CK(hipMemset(err, 0, 4));
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_pp<ST, LD>), dim3(16), dim3(64), 0, 0, slots, peer, n, err, xcc);
    hipEventRecord(b, 0);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
    CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
  }
  unsigned hx[16];
  CK(hipMemcpy(hx, xcc, sizeof hx, hipMemcpyDeviceToHost));
  printf("%-44s peer block %d (XCC %u vs %u): %.3f us one way%s\n", what, peer, hx[0], hx[peer], best * 1e3f / n / 2, e ? "  TIMED OUT (stale)" : "");
  return 0;
}
int main() {
  u64 *slots; int *err; unsigned *xcc;
  CK(hipMalloc(&slots, 64 * 8)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&xcc, 64));
  for (int peer : {8, 1}) {
    run<2, 2>("store sc1, load sc1 (the engines' granules)", slots, err, xcc, peer);
    run<3, 3>("store sc0 sc1, load sc0 sc1", slots, err, xcc, peer);
    run<2, 1>("store sc1, load sc0", slots, err, xcc, peer);
    run<1, 1>("store sc0, load sc0", slots, err, xcc, peer);
    run<0, 1>("store plain, load sc0", slots, err, xcc, peer);
    run<0, 2>("store plain, load sc1", slots, err, xcc, peer);
    run<1, 2>("store sc0, load sc1", slots, err, xcc, peer);
  }
  return 0;
}
