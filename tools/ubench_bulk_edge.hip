// Micro-benchmark (dev tool, not product): what ONE bulk all-gather edge of a weight-stationary BATCHED decoder engine would
// cost inside a launch (DESIGN.md "Next for the batched path").  In that engine each of the 256
// workgroups keeps 16 + 16 LSTM gate rows in registers as the MFMA A operand and needs, per edge, the new hidden vector
// of EVERY chunk: 1024 units x B chunks.  Producer c owns units 4c..4c+3, i.e. one [64 chunks][4] = 1 KB run of the
// [K/4][Bpad][4] operand layout -- every 128-byte line has ONE producer workgroup.
//
// Transports measured (time per step of a chain of such edges, nothing else in the step):
//   0  fresh + flag + plain loads: payload sc1 (write-through) 16-B stores into an address range that is NEW every step,
//      vmcnt(0), one sc1 flag store per producer; each consumer WAVE polls the flags of the 32 producers of its K-slice
//      (sc1) and then reads their 32 KB with plain (L1/L2-cached) 16-B loads -- no fence: the lines were never cached
//   1  the same with sc1 payload loads (always served from beyond the XCD's L2)
//   2  ping-pong addresses + flag + ONE agent acquire per wave (buffer_inv sc1) + plain loads
//   3  16-byte tagged granules {tag, 3 floats} polled directly with sc1 loads (no flag; 4/3 the bytes)
// Every value is checked; every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_bulk_edge tools/ubench_bulk_edge.hip && ./ubench_bulk_edge [steps] [chunks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NCU = 256, NT = 512, NW = NT / 64, PER_WAVE = NCU / NW;  // 32 producers per consumer wave
constexpr unsigned SPIN_LIMIT = 1u << 20;

struct Args {
  float *act;      // [ring][256 producers][vec16 x 4 floats]
  unsigned *flag;  // [256]
  int *err;
  float *sink;
  int nsteps, ring, vec16;  // vec16 = 16-byte vectors per producer per step (64 = 64 chunks x 4 units x 4 B)
};

__device__ __forceinline__ float expect(int s, int c, int i) { return (float)(((s * 131 + c * 17 + i) & 4095) + 1); }
__device__ __forceinline__ void store_sc1(f4 *p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f4 load_sc1(const f4 *p) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ bool give_up(unsigned &spins, int *err) {
  if (++spins > SPIN_LIMIT || ((spins & 127u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    atomicExch(err, 1);
    return true;
  }
  __builtin_amdgcn_s_sleep(1);
  return false;
}

template <int MODE>
__global__ __launch_bounds__(NT) void k_bulk(Args a) {
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float bad = 0.f;
  for (int s = 0; s < a.nsteps; ++s) {
    const size_t slot = (size_t)(s % a.ring) * NCU * a.vec16;
    // ---- publish this workgroup's run (wave 0: one 16-byte store per lane and vector) ----
    if (wave == 0) {
      f4 *dst = reinterpret_cast<f4 *>(a.act) + slot + (size_t)c * a.vec16;
      for (int i = lane; i < a.vec16; i += 64) {
        f4 v = (f4){expect(s, c, 4 * i), expect(s, c, 4 * i + 1), expect(s, c, 4 * i + 2), expect(s, c, 4 * i + 3)};
        if (MODE == 3) v.x = __uint_as_float((unsigned)(s + 1));  // tag in the data: 3 payload floats per vector
        store_sc1(dst + i, v);
      }
      if (MODE != 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the flag (R1)
        if (lane == 0) __hip_atomic_store(a.flag + c, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- consume: wave w takes producers [32 w, 32 w + 32) ----
    const int p0 = PER_WAVE * wave;
    if (MODE != 3) {
      unsigned spins = 0;
      for (;;) {
        const unsigned f = lane < PER_WAVE ? __hip_atomic_load(a.flag + p0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)(s + 1);
        if (__all(f >= (unsigned)(s + 1))) break;
        if (give_up(spins, a.err)) break;
      }
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const f4 *src = reinterpret_cast<const f4 *>(a.act) + slot + (size_t)p0 * a.vec16;
      // 8 loads in flight per lane
      for (int j0 = 0; j0 < PER_WAVE * a.vec16; j0 += 64 * 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + 64 * u + lane;
          if (j < PER_WAVE * a.vec16) v[u] = MODE == 1 ? load_sc1(src + j) : src[j];
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + 64 * u + lane;
          if (j < PER_WAVE * a.vec16) {
            const int pc = p0 + j / a.vec16, i = j % a.vec16;
            bad += fabsf(v[u].x - expect(s, pc, 4 * i)) + fabsf(v[u].w - expect(s, pc, 4 * i + 3));
          }
        }
      }
    } else {
      const f4 *src = reinterpret_cast<const f4 *>(a.act) + slot + (size_t)p0 * a.vec16;
      for (int j0 = 0; j0 < PER_WAVE * a.vec16; j0 += 64 * 8) {
        f4 v[8];
        bool done[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) done[u] = j0 + 64 * u + lane >= PER_WAVE * a.vec16;
        unsigned spins = 0;
        for (;;) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (!done[u]) v[u] = load_sc1(src + j0 + 64 * u + lane);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          bool all = true;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (!done[u]) {
              if (__float_as_uint(v[u].x) == (unsigned)(s + 1)) done[u] = true;
              else all = false;
            }
          if (all || give_up(spins, a.err)) break;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + 64 * u + lane;
          if (j < PER_WAVE * a.vec16) bad += fabsf(v[u].w - expect(s, p0 + j / a.vec16, 4 * (j % a.vec16) + 3));
        }
      }
    }
    __syncthreads();  // (the engine's accumulator exchange sits here)
  }
  if (bad != 0.f) atomicExch(a.err, 2);
  a.sink[c * NT + tid] = bad;
}

int main(int argc, char **argv) {
  const int nsteps = argc > 1 ? atoi(argv[1]) : 400, chunks = argc > 2 ? atoi(argv[2]) : 64;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  if (prop.multiProcessorCount < NCU) {
    printf("needs %d CUs\n", NCU);
    return 1;
  }
  Args a{};
  a.nsteps = nsteps;
  a.vec16 = chunks;  // chunks x 4 units x 4 B = chunks 16-byte vectors
  const size_t per_step = (size_t)NCU * a.vec16 * 4;
  CK(hipMalloc(&a.act, per_step * nsteps * sizeof(float)));
  CK(hipMalloc(&a.flag, NCU * 4));
  CK(hipMalloc(&a.err, 4));
  CK(hipMalloc(&a.sink, sizeof(float) * NCU * NT));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char *names[4] = {"fresh addresses + flag + plain loads", "fresh addresses + flag + sc1 loads", "ping-pong + flag + acquire + plain loads",
                          "16-byte tagged granules, sc1 polls"};
  printf("one bulk edge: 256 producers x %d B -> all 256 workgroups (%.0f KB per workgroup and step)\n", a.vec16 * 16, NCU * a.vec16 * 16 / 1024.0);
  for (int mode = 0; mode < 4; ++mode) {
    a.ring = mode == 2 ? 2 : nsteps;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(a.act, 0, per_step * nsteps * sizeof(float)));
      CK(hipMemset(a.flag, 0, NCU * 4));
      CK(hipMemset(a.err, 0, 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_bulk<0>, dim3(NCU), dim3(NT), 0, 0, a); break;
        case 1: hipLaunchKernelGGL(k_bulk<1>, dim3(NCU), dim3(NT), 0, 0, a); break;
        case 2: hipLaunchKernelGGL(k_bulk<2>, dim3(NCU), dim3(NT), 0, 0, a); break;
        default: hipLaunchKernelGGL(k_bulk<3>, dim3(NCU), dim3(NT), 0, 0, a); break;
      }
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      int herr = 0;
      CK(hipMemcpy(&herr, a.err, 4, hipMemcpyDeviceToHost));
      printf("mode %d (%s) rep %d: %.3f us per step, err=%d%s\n", mode, names[mode], rep, ms * 1e3 / nsteps, herr,
             herr == 2 ? " (STALE / WRONG VALUE SEEN)" : (herr == 1 ? " (TIMEOUT)" : ""));
    }
  }
  return 0;
}
