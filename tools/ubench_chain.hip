// Micro-benchmark (dev tool, not product): cost of one dependent stage in a hipGraph chain on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float *p) {}
__global__ void k_ld_st(const float *in, float *out) { out[blockIdx.x * blockDim.x + threadIdx.x] = in[blockIdx.x * blockDim.x + threadIdx.x] + 1.f; }
__global__ void k_dep2(const int *idx, const float *in, float *out) {
  int i = idx[threadIdx.x & 63];
  out[blockIdx.x * blockDim.x + threadIdx.x] = in[i + blockIdx.x * blockDim.x + threadIdx.x] + 1.f;
}
// GEMV-like: each wave reads a 6 KB row + 6 KB x, reduces, one lane stores
__global__ void k_gemv(const float4 *W, const float4 *x, float *y) {
  int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  float a = 0.f;
  for (int k = 0; k < 6; ++k) { float4 w = W[row * 384 + lane + 64 * k], v = x[lane + 64 * k]; a += w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w; }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) y[row] = a;
}

template <class F> float run_chain(hipStream_t s, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch(i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(b, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (10.f * n);
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *a, *b; int *idx; float4 *W;
  CK(hipMalloc(&a, 1 << 24)); CK(hipMalloc(&b, 1 << 24)); CK(hipMalloc(&idx, 4096)); CK(hipMalloc(&W, 84 * 384 * 16));
  CK(hipMemset(a, 0, 1 << 24)); CK(hipMemset(b, 0, 1 << 24)); CK(hipMemset(idx, 0, 4096)); CK(hipMemset(W, 0, 84 * 384 * 16));
  const int N = 120;
  for (int blocks : {1, 21, 256}) {
    printf("blocks=%3d  empty %.2f us", blocks, run_chain(s, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, a); }));
    printf("  ld->st %.2f us", run_chain(s, N, [&](int i) { hipLaunchKernelGGL(k_ld_st, dim3(blocks), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a); }));
    printf("  dep2 %.2f us\n", run_chain(s, N, [&](int i) { hipLaunchKernelGGL(k_dep2, dim3(blocks), dim3(256), 0, s, idx, (i & 1) ? a : b, (i & 1) ? b : a); }));
  }
  printf("gemv 21 blocks (81 rows x 1536): %.2f us\n", run_chain(s, N, [&](int i) { hipLaunchKernelGGL(k_gemv, dim3(21), dim3(256), 0, s, W, (const float4 *)((i & 1) ? a : b), (i & 1) ? b : a); }));
  // eager (no graph) launch rate for comparison
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int i = 0; i < 1200; ++i) hipLaunchKernelGGL(k_ld_st, dim3(21), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a);
  hipEventRecord(e1, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("eager ld->st 21 blocks: %.2f us per kernel\n", ms * 1e3f / 1200);
  return 0;
}
