// edge_floor.hip -- measurement aid, not on the product path: the communication skeleton of one persistent-decoder step
// (decoder_persistent.hip) -- 256 workgroups x 512 threads, one per CU, FIVE all-gather edges per step, each a set of
// data-tagged 8-byte granules {tag = step + 1, value} (one relaxed agent-scope store per value, polled with sc1 loads):
//   x      256 values, 16 producers (one 128-byte store each) -> all 256
//   h_att  1024 values, 256 producers (4 each)                -> all 256
//   e_part 8 x T values, 8 producers                          -> all 256 (every workgroup computes the softmax itself)
//   h_dec  1024 values, 256 producers                         -> all 256
//   mel    81 values, 16 producers                            -> the same 16, which publish x(s + 1)
// No arithmetic beyond a checksum: the time per step is the floor these five dependent exchanges impose on the step.
// bench.py runs it on the benched device at bench time (xdtts_edge_floor_us) for roofline.latency_floor_us; every value is
// checked, every spin bounded.  tools/ubench_edges5.hip is the command-line front end of the same kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

namespace xdtts_edge_floor {
typedef unsigned long long u64;
constexpr int NCU = 256, NT = 512, NATT = 8, NPRE = 16, EP_LD = 128;
constexpr unsigned SPIN_LIMIT = 1u << 20;

struct Gran {
  u64 *x, *hatt, *ep, *hdec, *mel;  // each [2 parities][n]
  int *err;
  float *sink;
};
__device__ __forceinline__ float expect(int step, int kind, int idx) { return (float)((step * 31 + kind * 7 + idx) & 1023); }
__device__ __forceinline__ void publish(u64 *slot, int step, float v) {
  __hip_atomic_store(slot, ((u64)(unsigned)(step + 1) << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// N granules at base[idx + i * stride], all loads in flight together (as the kernel's gather<N>)
template <int N>
__device__ __forceinline__ void gather(const u64 *base, int idx, int stride, int step, float (&out)[N], int *err) {
  bool done[N];
#pragma unroll
  for (int i = 0; i < N; ++i) done[i] = false;
  unsigned spins = 0;
  for (;;) {
    u64 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (!done[i]) v[i] = __hip_atomic_load(base + idx + i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool all = true;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (!done[i]) {
        if ((unsigned)(v[i] >> 32) == (unsigned)(step + 1)) {
          out[i] = __uint_as_float((unsigned)v[i]);
          done[i] = true;
        } else {
          all = false;
        }
      }
    if (all) return;
    if (++spins > SPIN_LIMIT || ((spins & 127u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      atomicExch(err, 1);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// tuned != 0: the consumers delay their first poll as the real kernel does (x 256 clocks): the ones that need a vector at once
// by `first` units, the others by `lazy` (their polls would otherwise crowd the fabric), the energies by `clazy`
struct Delays {
  int tuned, lazy, first, clazy, efirst, pfirst, xfirst, xlazy;
};
__device__ __forceinline__ void pause(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
}
__global__ __launch_bounds__(NT) void k_skeleton5(Gran g, int nsteps, int T, Delays dl) {
  const int c = blockIdx.x, tid = threadIdx.x;
  __shared__ float s_x[256], s_hatt[1024], s_hdec[1024], s_e[EP_LD], s_mel[96];
  const bool attn = c < NATT, pre = c >= NATT && c < NATT + NPRE;
  const int rk = attn ? c : c - NATT;
  float bad = 0.f;
  for (int s = 0; s < nsteps; ++s) {
    const int p = s & 1;
    // edge 1: x(s) -> everyone (256 threads poll one granule each)
    if (dl.tuned) pause(pre ? dl.xfirst : dl.xlazy);
    if (tid < 256) {
      float v[1];
      gather<1>(g.x, p * 256 + tid, 0, s, v, g.err);
      bad += fabsf(v[0] - expect(s, 0, tid));
      s_x[tid] = v[0];
    }
    __syncthreads();
    if (tid < 4) publish(g.hatt + p * 1024 + 4 * c + tid, s, expect(s, 1, 4 * c + tid) + 0.f * s_x[tid]);
    // edge 2: h_att(s) -> everyone (two granules per thread in flight together)
    if (dl.tuned) pause(attn ? dl.first : dl.lazy);
    {
      float v[2];
      gather<2>(g.hatt, p * 1024 + tid, NT, s, v, g.err);
      bad += fabsf(v[0] - expect(s, 1, tid)) + fabsf(v[1] - expect(s, 1, tid + NT));
      s_hatt[tid] = v[0];
      s_hatt[tid + NT] = v[1];
    }
    __syncthreads();
    if (attn && tid < T) publish(g.ep + (p * NATT + rk) * EP_LD + tid, s, expect(s, 2, rk * EP_LD + tid) + 0.f * s_hatt[tid]);
    // edge 3: the 8 partial-energy rows -> everyone (thread -> time step tid / 4, rows j and j + 4)
    if (dl.tuned) pause(attn ? dl.efirst : dl.clazy);
    {
      const int t = tid >> 2, j = tid & 3;
      float v[2] = {0.f, 0.f};
      if (t < T) {
        gather<2>(g.ep, (p * NATT + j) * EP_LD + t, 4 * EP_LD, s, v, g.err);
        bad += fabsf(v[0] - expect(s, 2, j * EP_LD + t)) + fabsf(v[1] - expect(s, 2, (j + 4) * EP_LD + t));
      }
      if (j == 0 && t < EP_LD) s_e[t] = v[0] + v[1];
    }
    __syncthreads();
    if (tid < 4) publish(g.hdec + p * 1024 + 4 * c + tid, s, expect(s, 3, 4 * c + tid) + 0.f * s_e[tid]);
    // edge 4: h_dec(s) -> everyone
    if (dl.tuned) pause(pre ? dl.pfirst : dl.lazy);
    {
      float v[2];
      gather<2>(g.hdec, p * 1024 + tid, NT, s, v, g.err);
      bad += fabsf(v[0] - expect(s, 3, tid)) + fabsf(v[1] - expect(s, 3, tid + NT));
      s_hdec[tid] = v[0];
      s_hdec[tid + NT] = v[1];
    }
    __syncthreads();
    if (pre) {
      // edge 5: mel rows rk + 16 w (w = 0..5) -> the 16 projection / prenet workgroups, which publish x(s + 1) as one 128-byte store
      if (tid < 6 && rk + 16 * tid < 81) publish(g.mel + p * 96 + rk + 16 * tid, s, expect(s, 4, rk + 16 * tid) + 0.f * s_hdec[tid]);
      if (tid < 81) {
        float v[1];
        gather<1>(g.mel, p * 96 + tid, 0, s, v, g.err);
        bad += fabsf(v[0] - expect(s, 4, tid));
        s_mel[tid] = v[0];
      }
      __syncthreads();
      if (tid < 16) publish(g.x + (p ^ 1) * 256 + 16 * rk + tid, s + 1, expect(s + 1, 0, 16 * rk + tid) + 0.f * s_mel[tid]);
    }
  }
  if (bad != 0.f) atomicExch(g.err, 2);
  g.sink[c * NT + tid] = bad;
}

__global__ void k_seed(Gran g) {  // x(0)
  publish(g.x + threadIdx.x, 0, expect(0, 0, threadIdx.x));
}


// best of `reps` timed launches of `nsteps` steps; < 0: the grid cannot be co-resident on this device, or an exchange failed.
// verbose: one line per launch on stdout (the command-line tool).
inline double measure(int device, int nsteps, int T, Delays dl, int reps, bool verbose) {
#define EF_CK(x) do { if ((x) != hipSuccess) return -1.0; } while (0)
  hipDeviceProp_t prop;
  EF_CK(hipSetDevice(device));
  EF_CK(hipGetDeviceProperties(&prop, device));
  int per_cu = 0;
  EF_CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_skeleton5, NT, 0));
  if (prop.multiProcessorCount < NCU || per_cu < 1 || T < 1 || T > EP_LD || nsteps < 1) return -1.0;
  const size_t words = 2 * (256 + 1024 + NATT * EP_LD + 1024 + 96);
  u64 *buf = nullptr;
  int *err = nullptr;
  float *sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;  // (its own stream: nothing of this library touches the legacy default stream)
  double best = 1e30;
  bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipMalloc(&buf, words * 8) == hipSuccess &&
            hipMalloc(&err, 4) == hipSuccess && hipMalloc(&sink, sizeof(float) * NCU * NT) == hipSuccess &&
            hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
  Gran g{};
  if (ok) {
    g.x = buf;
    g.hatt = g.x + 2 * 256;
    g.ep = g.hatt + 2 * 1024;
    g.hdec = g.ep + 2 * NATT * EP_LD;
    g.mel = g.hdec + 2 * 1024;
    g.err = err;
    g.sink = sink;
  }
  for (int rep = 0; ok && rep < reps + 1; ++rep) {  // (the first launch is a warm-up)
    ok = hipMemsetAsync(buf, 0, words * 8, st) == hipSuccess && hipMemsetAsync(err, 0, 4, st) == hipSuccess;
    if (!ok) break;
    hipLaunchKernelGGL(k_seed, dim3(1), dim3(256), 0, st, g);
    ok = hipStreamSynchronize(st) == hipSuccess && hipEventRecord(e0, st) == hipSuccess;
    if (!ok) break;
    hipLaunchKernelGGL(k_skeleton5, dim3(NCU), dim3(NT), 0, st, g, nsteps, T, dl);
    float ms = 0;
    int herr = 0;
    ok = hipEventRecord(e1, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess &&
         hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    if (!ok) break;
    const double us = ms * 1e3 / nsteps;
    if (verbose)
      printf("rep %d: %d steps (T = %d), %.3f ms, %.3f us per step (5 edges: %.3f us per edge), err=%d\n", rep, nsteps, T, ms, us, us / 5, herr);
    if (herr) ok = false;
    if (rep > 0 && !herr && us < best) best = us;
  }
  if (buf) (void)hipFree(buf);
  if (err) (void)hipFree(err);
  if (sink) (void)hipFree(sink);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
#undef EF_CK
  return ok && best < 1e29 ? best : -1.0;
}
inline Delays kernel_delays(int tuned) { return Delays{tuned, 9, 4, 4, 3, 0, 4, 0}; }  // (decoder_persistent.hip: persist_bufs)

}  // namespace xdtts_edge_floor
