// Micro-benchmark (dev tool): do kernels on two HIP streams of one process run side by side?  Streams are mapped onto a few hardware
// queues; two streams on one queue (or on queues the command processor does not serve together) run their kernels one after the other.
// For streams 0..N-1: a 300 us spin kernel (4 workgroups) on stream 0 and on stream j together; wall time ~300 us = side by side,
// ~600 us = one after the other.  (Round 6: the sequence headline's two modes, 6.15 / 6.55 ms per utterance.)
//   hipcc --offload-arch=gfx950 -O3 -o ubench_streams tools/ubench_streams.hip && ./ubench_streams [n_streams] [priority_of_stream_1..]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_spin(unsigned long long ticks, int *sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 1) *sink = 1;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  const int hi_first = argc > 2 ? atoi(argv[2]) : 0;  // streams 1..hi_first get the highest priority
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("stream priority range: lowest %d, highest %d\n", lo, hi);
  hipStream_t s[32];
  int *sink;
  CK(hipMalloc((void **)&sink, 4));
  for (int i = 0; i < n; ++i) {
    if (i >= 1 && i <= hi_first) CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, hi));
    else CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
  }
  const unsigned long long ticks = 30000;  // 300 us at 100 MHz
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(4), dim3(64), 0, s[i], 100ull, sink);
  CK(hipDeviceSynchronize());
  for (int a = 0; a < (n > 3 ? 3 : n); ++a)
    for (int j = a + 1; j < n; ++j) {
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin, dim3(4), dim3(64), 0, s[a], ticks, sink);
        hipLaunchKernelGGL(k_spin, dim3(4), dim3(64), 0, s[j], ticks, sink);
        CK(hipStreamSynchronize(s[a]));
        CK(hipStreamSynchronize(s[j]));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        best = us < best ? us : best;
      }
      printf("streams %d + %d: %7.1f us  %s\n", a, j, best, best < 450 ? "side by side" : "ONE AFTER THE OTHER");
    }
  return 0;
}
