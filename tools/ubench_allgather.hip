// All-gather edge of the persistent decoders, alone: 256 workgroups (one per CU) each publish 4 values per chunk per iteration
// and gather all 1024 x NB of them before they publish the next -- the h_att / h_dec edges of decoder_persistent8.hip without
// the arithmetic.  Which transport is cheapest at this volume?  Developer tool.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_allgather tools/ubench_allgather.hip && /tmp/ubench_allgather [iters]
// Variants:
//   0  8-byte {tag, value} granules, 8-byte sc1 loads, two parity buffers (what the engines do)
//   1  the same granules read two at a time with 16-byte sc1 buffer loads
//   2  write-once ring of plain 4-byte values, 0xFFFFFFFF = not yet written; 16-byte loads (4 values), first try with FIRST_AUX
//      cache bits (0 plain, 1 sc0, 16 sc1), retries sc0 sc1
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int PT = 256, NCU = 256, K = 1024;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));            \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7fffffff, 0x00020000); }

template <int AUX>
__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX);
}

template <int VAR, int NB, int FIRST_AUX>
__global__ __launch_bounds__(PT) void k_ag(u64 *gran, unsigned *ring, int iters, int nap, float *out, int *err) {
  __shared__ float s_h[K * NB];
  __shared__ int s_bad;
  const int c = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (int s = 0; s < iters; ++s) {
    const unsigned want = (unsigned)(s + 1);
    const int p = s & 1;
    const float val = (float)((s * 7 + c) & 1023);
    // publish: wave 0, lane = (chunk b = lane / 4, unit u = lane % 4)
    if (tid < 4 * NB) {
      const int b = tid >> 2, u = tid & 3;
      if (VAR <= 1 || VAR == 3) {
        __hip_atomic_store(gran + (size_t)(p * NB + b) * K + 4 * c + u, ((u64)want << 32) | (u64)__float_as_uint(val + u), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_store(ring + ((size_t)s * NB + b) * K + 4 * c + u, __float_as_uint(val + u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    for (int i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(1);
    unsigned spins = 0;
    if (VAR == 0) {
      constexpr int N = 4 * NB;
      unsigned pending = N == 32 ? 0xffffffffu : (1u << (N & 31)) - 1u;
      const u64 *base = gran + (size_t)p * NB * K + tid;
      while (pending) {
        u64 v[N];
        unsigned zero = 0;
        asm volatile("" : "+v"(zero));
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __hip_atomic_load(base + (zero + i * PT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (((pending >> i) & 1u) && (unsigned)(v[i] >> 32) == want) {
            s_h[i * PT + tid] = __uint_as_float((unsigned)v[i]);
            pending &= ~(1u << i);
          }
        if (pending && ++spins > (1u << 20)) {
          atomicExch(err, 1);
          break;
        }
        if (pending) __builtin_amdgcn_s_sleep(1);
      }
    } else if (VAR == 3) {
      // 8-byte granules, TWO rounds of polls in flight `FIRST_AUX` x 64 clocks apart: a first round that comes too early then
      // costs that gap, not a round trip
      constexpr int N = 4 * NB;
      unsigned pending = N == 32 ? 0xffffffffu : (1u << (N & 31)) - 1u;
      const u64 *base = gran + (size_t)p * NB * K + tid;
      u64 va[N], vb[N];
      unsigned zero = 0;
      asm volatile("" : "+v"(zero));
#pragma unroll
      for (int i = 0; i < N; ++i) va[i] = __hip_atomic_load(base + (zero + i * PT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = 0; i < FIRST_AUX; ++i) __builtin_amdgcn_s_sleep(1);
      asm volatile("" : "+v"(zero));
#pragma unroll
      for (int i = 0; i < N; ++i) vb[i] = __hip_atomic_load(base + (zero + i * PT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < N; ++i)
        if ((unsigned)(va[i] >> 32) == want) {
          s_h[i * PT + tid] = __uint_as_float((unsigned)va[i]);
          pending &= ~(1u << i);
        }
      if (pending) {
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (((pending >> i) & 1u) && (unsigned)(vb[i] >> 32) == want) {
            s_h[i * PT + tid] = __uint_as_float((unsigned)vb[i]);
            pending &= ~(1u << i);
          }
      }
      while (pending) {
        u64 v[N];
        asm volatile("" : "+v"(zero));
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __hip_atomic_load(base + (zero + i * PT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (((pending >> i) & 1u) && (unsigned)(v[i] >> 32) == want) {
            s_h[i * PT + tid] = __uint_as_float((unsigned)v[i]);
            pending &= ~(1u << i);
          }
        if (pending && ++spins > (1u << 20)) {
          atomicExch(err, 1);
          break;
        }
        if (pending) __builtin_amdgcn_s_sleep(1);
      }
    } else if (VAR == 1) {
      constexpr int N = 2 * NB;  // 16-byte loads: granules 2 (tid + 256 i), + 1
      unsigned pending = (1u << N) - 1u;
      const __amdgpu_buffer_rsrc_t r = rsrc_of(gran + (size_t)p * NB * K);
      while (pending) {
        u32x4 v[N];
        unsigned off = 16u * (unsigned)tid;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = load16<16>(r, off + 16u * PT * i);
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (((pending >> i) & 1u) && v[i].y == want && v[i].w == want) {
            *reinterpret_cast<float2 *>(s_h + 2 * (i * PT + tid)) = make_float2(__uint_as_float(v[i].x), __uint_as_float(v[i].z));
            pending &= ~(1u << i);
          }
        if (pending && ++spins > (1u << 20)) {
          atomicExch(err, 1);
          break;
        }
        if (pending) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
    } else {
      constexpr int N = NB;  // 16-byte loads: values 4 (tid + 256 i) .. + 3
      unsigned pending = (1u << N) - 1u;
      const __amdgpu_buffer_rsrc_t r = rsrc_of(ring + (size_t)s * NB * K);
      bool first = true;
      while (pending) {
        u32x4 v[N];
        unsigned off = 16u * (unsigned)tid;
        asm volatile("" : "+v"(off));
        if (first) {
#pragma unroll
          for (int i = 0; i < N; ++i) v[i] = load16<FIRST_AUX>(r, off + 16u * PT * i);
        } else {
#pragma unroll
          for (int i = 0; i < N; ++i) v[i] = load16<17>(r, off + 16u * PT * i);
        }
        first = false;
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (((pending >> i) & 1u) && v[i].x != 0xffffffffu && v[i].y != 0xffffffffu && v[i].z != 0xffffffffu && v[i].w != 0xffffffffu) {
            *reinterpret_cast<float4 *>(s_h + 4 * (i * PT + tid)) =
                make_float4(__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z), __uint_as_float(v[i].w));
            pending &= ~(1u << i);
          }
        if (pending && ++spins > (1u << 20)) {
          atomicExch(err, 1);
          break;
        }
        if (pending) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
    }
    __syncthreads();
    // check: every value of chunk 0 (cheap: this thread's 4)
    for (int j = 0; j < 4; ++j) {
      const int k = tid * 4 + j;  // value index within the vector: workgroup k / 4, unit k % 4
      float got;
      if (VAR == 0 || VAR == 3) got = s_h[(k / PT) * PT + (k % PT)];  // i = k / 256 (chunk 0: i < 4), tid' = k % 256
      else if (VAR == 1) got = s_h[k];                      // pairs in order
      else got = s_h[k];
      const float expect = (float)((s * 7 + (k >> 2)) & 1023) + (float)(k & 3);
      if (got != expect) s_bad = 1;
      acc += got;
    }
    __syncthreads();
  }
  if (tid == 0) out[c] = acc + (s_bad ? 1e30f : 0.f);
  if (tid == 0 && s_bad) atomicExch(err, 2);
}

template <int VAR, int NB, int AUX>
double run(int iters, int nap, u64 *gran, unsigned *ring, size_t ring_bytes, float *out, int *err, const char *name) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipMemset(gran, 0, sizeof(u64) * 2 * 8 * K));
    if (VAR == 2) CHECK(hipMemset(ring, 0xff, ring_bytes));
    CHECK(hipMemset(err, 0, sizeof(int)));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    void *args[] = {&gran, &ring, &iters, &nap, &out, &err};
    CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_ag<VAR, NB, AUX>), dim3(NCU), dim3(PT), args, 0, 0));
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int h = 0;
    CHECK(hipMemcpy(&h, err, sizeof(int), hipMemcpyDeviceToHost));
    if (h) {
      printf("%-44s NB=%d nap=%2d: FAILED (%s)\n", name, NB, nap, h == 1 ? "timed out" : "wrong data");
      return -1;
    }
    const double us = ms * 1e3 / iters;
    if (us < best) best = us;
  }
  printf("%-44s NB=%d nap=%2d: %.2f us per edge\n", name, NB, nap, best);
  return best;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  u64 *gran;
  unsigned *ring;
  float *out;
  int *err;
  const size_t ring_bytes = (size_t)iters * 8 * K * sizeof(unsigned);
  // argv[2]: allocation flavour of the exchange memory: 0 hipMalloc, 1 hipDeviceMallocUncached, 2 hipDeviceMallocFinegrained
  const int flavour = argc > 2 ? atoi(argv[2]) : 0;
  if (flavour == 0) {
    CHECK(hipMalloc(&gran, sizeof(u64) * 2 * 8 * K));
    CHECK(hipMalloc(&ring, ring_bytes));
  } else {
    const unsigned fl = flavour == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
    CHECK(hipExtMallocWithFlags((void **)&gran, sizeof(u64) * 2 * 8 * K, fl));
    CHECK(hipExtMallocWithFlags((void **)&ring, ring_bytes, fl));
  }
  printf("exchange memory: %s\n", flavour == 0 ? "hipMalloc" : flavour == 1 ? "hipDeviceMallocUncached" : "hipDeviceMallocFinegrained");
  CHECK(hipMalloc(&out, sizeof(float) * NCU));
  CHECK(hipMalloc(&err, sizeof(int)));
  if (argc > 3 && argv[3][0] == 'd') {  // staggered double polls against single ones, 1 and 2 chunks
    for (int nap : {4, 6, 8, 10, 12, 14, 16, 20}) {
      run<0, 1, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, one poll round");
      run<3, 1, 4>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, two rounds 4 x 64 clocks apart");
      run<3, 1, 8>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, two rounds 8 x 64 clocks apart");
      run<0, 2, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, one poll round");
      run<3, 2, 4>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, two rounds 4 x 64 clocks apart");
      run<3, 2, 8>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, two rounds 8 x 64 clocks apart");
    }
    return 0;
  }
  if (argc > 3) {  // short form: the engines' cases only
    for (int nap : {8, 12, 16, 20}) {
      run<0, 1, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 8-B sc1 loads");
      run<0, 2, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 8-B sc1 loads");
      run<2, 4, 16>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first sc1");
      run<2, 8, 16>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first sc1");
    }
    return 0;
  }
  for (int nap : {0, 8, 12, 16, 20, 24}) {
    run<0, 1, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 8-B sc1 loads");
    run<1, 1, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 16-B sc1 loads");
    run<2, 1, 16>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first sc1");
    run<0, 2, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 8-B sc1 loads");
    run<1, 2, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 16-B sc1 loads");
    run<2, 2, 16>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first sc1");
  }
  for (int nap : {0, 8, 16, 24}) {
    run<0, 4, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 8-B sc1 loads");
    run<1, 4, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 16-B sc1 loads");
    run<2, 4, 0>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first plain");
    run<2, 4, 1>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first sc0");
    run<2, 4, 16>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first sc1");
    run<0, 8, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 8-B sc1 loads");
    run<1, 8, 0>(iters, nap, gran, ring, ring_bytes, out, err, "8-B granules, 16-B sc1 loads");
    run<2, 8, 0>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first plain");
    run<2, 8, 1>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first sc0");
    run<2, 8, 16>(iters, nap, gran, ring, ring_bytes, out, err, "ring of values, 16-B loads, first sc1");
  }
  return 0;
}
