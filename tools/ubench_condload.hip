// Does a polling gather whose loads sit under PER-LANE conditions ("only the granules this lane still misses") deliver every value
// to the slot it was asked for?  decoder_persistent8.hip saw values of another granule of the same round with that form (N = 16)
// and issues all loads of a round unconditionally since; the older engines poll 2..4 granules per thread with the conditional
// form.  This reproduces both forms in isolation: 256 workgroups x 256 threads; per iteration every workgroup publishes its
// 4 x NB granules {tag, value = f(slot)} with a per-lane stagger, then gathers N = 4 NB of them per thread and checks each
// value against the slot it came from.  Developer tool.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_condload tools/ubench_condload.hip && /tmp/ubench_condload [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
constexpr int PT = 256, NCU = 256, K = 1024;

#define CHECK(x)                                                       \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                         \
    }                                                                  \
  } while (0)

__device__ __forceinline__ float slot_value(int s, int k) { return (float)(((s * 131 + k * 7) & 0xffff)) + 0.25f * (float)(k & 3); }

template <int NB, bool COND>
__global__ __launch_bounds__(PT) void k_cl(u64 *gran, int iters, unsigned long long *bad, int *err) {
  __shared__ float s_h[K * NB];
  const int c = blockIdx.x, tid = threadIdx.x;
  unsigned long long wrong = 0;
  for (int s = 0; s < iters; ++s) {
    const unsigned want = (unsigned)(s + 1);
    const int p = s & 1;
    if (tid < 4 * NB) {
      const int b = tid >> 2, u = tid & 3, k = b * K + 4 * c + u;
      for (int i = 0; i < ((c * 7 + b * 3 + s) & 15); ++i) __builtin_amdgcn_s_sleep(2);  // stagger: lanes of a reader see different subsets first
      __hip_atomic_store(gran + (size_t)p * NB * K + k, ((u64)want << 32) | (u64)__float_as_uint(slot_value(s, k)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    constexpr int N = 4 * NB;
    unsigned pending = N == 32 ? 0xffffffffu : (1u << (N & 31)) - 1u, spins = 0;
    const u64 *base = gran + (size_t)p * NB * K + tid;
    while (pending) {
      u64 v[N];
      unsigned zero = 0;
      asm volatile("" : "+v"(zero));
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (COND) {
          if ((pending >> i) & 1u) v[i] = __hip_atomic_load(base + (zero + i * PT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          v[i] = __hip_atomic_load(base + (zero + i * PT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (((pending >> i) & 1u) && (unsigned)(v[i] >> 32) == want) {
          const float got = __uint_as_float((unsigned)v[i]);
          s_h[i * PT + tid] = got;
          if (got != slot_value(s, i * PT + tid)) ++wrong;  // the value of ANOTHER slot (or of another step)
          pending &= ~(1u << i);
        }
      if (pending && ++spins > (1u << 22)) {
        atomicExch(err, 1);
        break;
      }
      if (pending) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    // and what landed in LDS
    for (int i = 0; i < N; ++i)
      if (s_h[i * PT + tid] != slot_value(s, i * PT + tid)) ++wrong;
    __syncthreads();
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int NB, bool COND>
void run(int iters, u64 *gran, unsigned long long *bad, int *err) {
  CHECK(hipMemset(gran, 0, sizeof(u64) * 2 * 8 * K));
  CHECK(hipMemset(bad, 0, sizeof(unsigned long long)));
  CHECK(hipMemset(err, 0, sizeof(int)));
  void *args[] = {&gran, &iters, &bad, &err};
  CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_cl<NB, COND>), dim3(NCU), dim3(PT), args, 0, 0));
  CHECK(hipDeviceSynchronize());
  unsigned long long hb = 0;
  int he = 0;
  CHECK(hipMemcpy(&hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&he, err, sizeof(he), hipMemcpyDeviceToHost));
  printf("N = %2d granules per thread, loads %-13s: %llu wrong values in %d iterations x 256 workgroups%s\n", 4 * NB, COND ? "conditional" : "unconditional", hb, iters,
         he ? " (TIMED OUT)" : "");
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  u64 *gran;
  unsigned long long *bad;
  int *err;
  CHECK(hipMalloc(&gran, sizeof(u64) * 2 * 8 * K));
  CHECK(hipMalloc(&bad, sizeof(unsigned long long)));
  CHECK(hipMalloc(&err, sizeof(int)));
  for (int rep = 0; rep < 2; ++rep) {
    run<1, true>(iters, gran, bad, err);
    run<1, false>(iters, gran, bad, err);
    run<2, true>(iters, gran, bad, err);
    run<4, true>(iters, gran, bad, err);
    run<4, false>(iters, gran, bad, err);
    run<8, true>(iters, gran, bad, err);
    run<8, false>(iters, gran, bad, err);
  }
  return 0;
}
