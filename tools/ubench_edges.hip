// Micro-benchmark (dev tool, not product): the communication skeleton of a persistent
// weight-stationary decoder step on gfx950 -- 256 workgroups (one per CU, 1024 threads), six
// all-gather edges per step carried by data-tagged 8-byte granules {tag = step + 1, value}:
//   x (256 values, 16 producers -> all)        h_att (1024, all -> all)
//   e_part (8 x 100, 8 -> the same 8)          ctx (512, 8 producers -> all)
//   h_dec (1024, all -> all)                   mel (81, 16 producers -> the same 16)
// No compute beyond a checksum: the measured time per step is the floor the edges impose.
// Every value is checked on arrival (value = f(step, index)); every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned long long u64;
constexpr int NCU = 256, NT = 1024;
constexpr int ATT0 = 0, NATT = 8, PRE0 = 8, NPRE = 16;
__device__ __forceinline__ void publish_local(u64 *slot, int step, float v) {  // stays in this XCD's L2
  *reinterpret_cast<volatile u64 *>(slot) = ((u64)(unsigned)(step + 1) << 32) | (u64)__float_as_uint(v);
}
constexpr unsigned SPIN_LIMIT = 1u << 20;

struct Gran {
  u64 *x, *hatt, *ep, *ctx, *hdec, *mel;  // each [2 parity][n]
  int *err;
  float *sink;
};

__device__ __forceinline__ float expect(int step, int kind, int idx) { return (float)((step * 31 + kind * 7 + idx) & 1023); }

__device__ __forceinline__ void publish(u64 *slot, int step, float v) {
  __hip_atomic_store(slot, ((u64)(unsigned)(step + 1) << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// Pipelined polling: DEPTH loads of the same granule in flight, issued ~STAGGER apart; the oldest is
// checked while the younger ones are still travelling, so a landed value is seen within one
// stagger interval instead of one full load round trip.
template <int DEPTH>
__device__ __forceinline__ float gather_pipe(const u64 *slot, int step, int *err) {
  u64 v[DEPTH];
  unsigned spins = 0;
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) {
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[i]) : "v"(slot) : "memory");
    __builtin_amdgcn_s_sleep(2);
  }
  for (;;) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[i]) : "n"(DEPTH - 1) : "memory");
      const u64 x = v[i];
      if ((unsigned)(x >> 32) == (unsigned)(step + 1)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return __uint_as_float((unsigned)x);
      }
      asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[i]) : "v"(slot) : "memory");
      if (++spins > SPIN_LIMIT * 4u || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicExch(err, 1);
        return 0.f;
      }
    }
  }
}

template <int SLEEP>
__device__ __forceinline__ float gather_t(const u64 *slot, int step, int *err) {
  if constexpr (SLEEP >= 100) return gather_pipe<SLEEP - 98>(slot, step, err);
  unsigned spins = 0;
  for (;;) {
    const u64 v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(v >> 32) == (unsigned)(step + 1)) return __uint_as_float((unsigned)v);
    if (++spins > SPIN_LIMIT || ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      atomicExch(err, 1);
      return 0.f;
    }
    if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
  }
}

struct alignas(16) U2 { u64 a, b; };
// two adjacent granules with one 16-byte sc1 load; each granule is still checked by its own tag
__device__ __forceinline__ float2 gather_pair(const u64 *slot, int step, int *err) {
  unsigned spins = 0;
  for (;;) {
    U2 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(slot) : "memory");
    if ((unsigned)(v.a >> 32) == (unsigned)(step + 1) && (unsigned)(v.b >> 32) == (unsigned)(step + 1))
      return make_float2(__uint_as_float((unsigned)v.a), __uint_as_float((unsigned)v.b));
    if (++spins > SPIN_LIMIT || ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      atomicExch(err, 1);
      return make_float2(0.f, 0.f);
    }
  }
}

template <int SLEEP>
__global__ __launch_bounds__(NT) void k_skeleton(Gran g, int nsteps, int sleep_compute) {
  const int c = blockIdx.x, tid = threadIdx.x;
  __shared__ float s_x[256], s_hatt[1024], s_ctx[512], s_hdec[1024], s_ep[8][128], s_mel[96];
  bool attn = c >= ATT0 && c < ATT0 + NATT, pre = c >= PRE0 && c < PRE0 + NPRE;
  int ak = c - ATT0, pj = c - PRE0;
  if (SLEEP == 8) {  // same-XCD groups under the observed block -> XCD b % 8 placement
    attn = (c & 7) == 0 && c < 64;
    ak = c >> 3;
    pre = (c & 7) == 1 && c < 128;
    pj = c >> 3;
  }
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (tid == 0) g.sink[c * NT + 1] = (float)(xcc & 15);
  float bad = 0.f;
  for (int s = 0; s < nsteps; ++s) {
    const int p = s & 1;
    // P1: x(s) -> everyone
    if (tid < 256) {
      const float v = gather_t<SLEEP>(g.x + p * 256 + tid, s, g.err);
      bad += fabsf(v - expect(s, 0, tid));
      s_x[tid] = v;
    }
    __syncthreads();
    if (tid < 4) publish(g.hatt + (SLEEP == 9 ? p * 4096 + 16 * c + tid : p * 1024 + 4 * c + tid), s, expect(s, 1, 4 * c + tid) + 0.f * s_x[tid]);
    // P2: h_att(s) -> everyone
    if (SLEEP == 7) {
      if (tid < 512) {
        const float2 v = gather_pair(g.hatt + p * 1024 + 2 * tid, s, g.err);
        bad += fabsf(v.x - expect(s, 1, 2 * tid)) + fabsf(v.y - expect(s, 1, 2 * tid + 1));
        s_hatt[2 * tid] = v.x;
        s_hatt[2 * tid + 1] = v.y;
      }
    } else {
      const float v = gather_t<SLEEP>(g.hatt + (SLEEP == 9 ? p * 4096 + 16 * (tid >> 2) + (tid & 3) : p * 1024 + tid), s, g.err);
      bad += fabsf(v - expect(s, 1, tid));
      s_hatt[tid] = v;
    }
    __syncthreads();
    if (attn) {
      const int k = ak;
      if (tid < 100) { if (SLEEP == 8) publish_local(g.ep + (p * 8 + k) * 128 + tid, s, expect(s, 2, k * 128 + tid) + 0.f * s_hatt[tid]); else publish(g.ep + (p * 8 + k) * 128 + tid, s, expect(s, 2, k * 128 + tid) + 0.f * s_hatt[tid]); }
      // P3: e_part -> the 8 attention blocks (wave 0 and 1 poll 8 granules per lane)
      if (tid < 100) {
        float e = 0.f;
        for (int kk = 0; kk < 8; ++kk) {
          const float v = gather_t<SLEEP>(g.ep + (p * 8 + kk) * 128 + tid, s, g.err);
          bad += fabsf(v - expect(s, 2, kk * 128 + tid));
          e += v;
        }
        s_ep[0][tid] = e;
      }
      __syncthreads();
      if (tid < 64) publish(g.ctx + p * 512 + 64 * k + tid, s, expect(s, 3, 64 * k + tid) + 0.f * s_ep[0][tid]);
    }
    // P4: ctx(s) -> everyone
    if (SLEEP == 7) {
      if (tid < 256) {
        const float2 v = gather_pair(g.ctx + p * 512 + 2 * tid, s, g.err);
        bad += fabsf(v.x - expect(s, 3, 2 * tid)) + fabsf(v.y - expect(s, 3, 2 * tid + 1));
        s_ctx[2 * tid] = v.x;
        s_ctx[2 * tid + 1] = v.y;
      }
    } else if (tid < 512) {
      const float v = gather_t<SLEEP>(g.ctx + p * 512 + tid, s, g.err);
      bad += fabsf(v - expect(s, 3, tid));
      s_ctx[tid] = v;
    }
    __syncthreads();
    if (tid < 4) publish(g.hdec + (SLEEP == 9 ? p * 4096 + 16 * c + tid : p * 1024 + 4 * c + tid), s, expect(s, 4, 4 * c + tid) + 0.f * s_ctx[tid]);
    // P5: h_dec(s) -> everyone
    if (SLEEP == 7) {
      if (tid < 512) {
        const float2 v = gather_pair(g.hdec + p * 1024 + 2 * tid, s, g.err);
        bad += fabsf(v.x - expect(s, 4, 2 * tid)) + fabsf(v.y - expect(s, 4, 2 * tid + 1));
        s_hdec[2 * tid] = v.x;
        s_hdec[2 * tid + 1] = v.y;
      }
    } else {
      const float v = gather_t<SLEEP>(g.hdec + (SLEEP == 9 ? p * 4096 + 16 * (tid >> 2) + (tid & 3) : p * 1024 + tid), s, g.err);
      bad += fabsf(v - expect(s, 4, tid));
      s_hdec[tid] = v;
    }
    __syncthreads();
    if (pre) {
      const int j = pj;
      if (tid < 6 && j + 16 * tid < 81) { if (SLEEP == 8) publish_local(g.mel + p * 96 + j + 16 * tid, s, expect(s, 5, j + 16 * tid) + 0.f * s_hdec[tid]); else publish(g.mel + p * 96 + j + 16 * tid, s, expect(s, 5, j + 16 * tid) + 0.f * s_hdec[tid]); }
      // P6: mel -> the 16 prenet blocks, which publish x(s+1)
      if (tid < 81) {
        const float v = gather_t<SLEEP>(g.mel + p * 96 + tid, s, g.err);
        bad += fabsf(v - expect(s, 5, tid));
        s_mel[tid] = v;
      }
      __syncthreads();
      if (tid < 16) publish(g.x + (p ^ 1) * 256 + 16 * j + tid, s + 1, expect(s + 1, 0, 16 * j + tid) + 0.f * s_mel[tid]);
    }
    for (int i = 0; i < sleep_compute; ++i) __builtin_amdgcn_s_sleep(8);
  }
  if (bad != 0.f) atomicExch(g.err, 2);
  if (tid != 1) g.sink[c * NT + tid] = bad;
}

__global__ void k_seed(Gran g) {  // x(0)
  const int t = threadIdx.x;
  if (t < 256) publish(g.x + t, 0, expect(0, 0, t));
}

int main(int argc, char **argv) {
  const int nsteps = argc > 1 ? atoi(argv[1]) : 600;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("CUs: %d\n", prop.multiProcessorCount);
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_skeleton<1>, NT, 0));
  printf("blocks per CU by the occupancy API: %d\n", per_cu);
  if (prop.multiProcessorCount < NCU || per_cu < 1) { printf("grid would not be co-resident\n"); return 1; }
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Gran g;
  const size_t words = 2 * (256 + 4096 + 8 * 128 + 512 + 4096 + 96);
  u64 *base;
  CK(hipMalloc(&base, words * 8));
  CK(hipMalloc(&g.err, 4));
  CK(hipMalloc(&g.sink, NCU * NT * 4));
  g.x = base; g.hatt = g.x + 2 * 256; g.ep = g.hatt + 2 * 4096; g.ctx = g.ep + 2 * 8 * 128; g.hdec = g.ctx + 2 * 512; g.mel = g.hdec + 2 * 4096;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemsetAsync(base, 0, words * 8, st));
    CK(hipMemsetAsync(g.err, 0, 4, st));
    hipLaunchKernelGGL(k_seed, dim3(1), dim3(256), 0, st, g);
    hipEventRecord(a, st);
    if (rep == 0) hipLaunchKernelGGL(k_skeleton<1>, dim3(NCU), dim3(NT), 0, st, g, nsteps, 0);
    if (rep == 1) hipLaunchKernelGGL(k_skeleton<0>, dim3(NCU), dim3(NT), 0, st, g, nsteps, 0);
    if (rep == 2) hipLaunchKernelGGL(k_skeleton<9>, dim3(NCU), dim3(NT), 0, st, g, nsteps, 0);
    if (rep == 3) hipLaunchKernelGGL(k_skeleton<8>, dim3(NCU), dim3(NT), 0, st, g, nsteps, 0);
    hipEventRecord(b, st);
    CK(hipStreamSynchronize(st));
    float ms;
    hipEventElapsedTime(&ms, a, b);
    int err = 0;
    CK(hipMemcpy(&err, g.err, 4, hipMemcpyDeviceToHost));
    if (rep == 3) {
      static float hs[NCU * NT];
      CK(hipMemcpy(hs, g.sink, sizeof hs, hipMemcpyDeviceToHost));
      int ok = 1;
      for (int c = 0; c < NCU; ++c) ok &= ((int)hs[c * NT + 1] == (c & 7));
      printf("XCC_ID == block %% 8 for every block: %s (block 0..9 ids:", ok ? "yes" : "NO");
      for (int c = 0; c < 10; ++c) printf(" %d", (int)hs[c * NT + 1]);
      printf(")\n");
    }
    printf("rep %d (sleep1, sleep0, h granules padded to one 128-B line per producer, same-XCD groups with plain stores): %d steps, %.3f ms, %.2f us per step (6 edges), err=%d\n", rep, nsteps, ms, ms * 1e3f / nsteps, err);
  }
  return 0;
}
