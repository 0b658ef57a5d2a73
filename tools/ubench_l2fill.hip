// Micro-benchmark (dev tool, not product): how fast can a CU pull the batched engine's activation operand out of its XCD's L2,
// and does the pull overlap the fp32 MFMAs that consume it?  (the two batched launches run at ~19 TB/s of L2 -> L1
// fill, 31 B/clk/CU, whatever was tried; VERDICT round 5 asks for global_load_lds staging.)
//
// Every block reads the SAME operand X [K/4][64 chunks][4] (K = 1536: 393 kB, L2-resident, far beyond a 32 kB L1) the way
// lstm_mfma_pass does: the block's waves split K, a k-step = 16 columns = NTA 16-byte loads per lane (one per 16-chunk tile),
// consumed by 4 NTA v_mfma_f32_16x16x4_f32 (MFMA modes) or by NTA v_add (loads-only modes).
//   mode 0  loads -> registers, ring of D k-steps ahead, consumed by adds          (what the L2 -> L1 path gives this pattern)
//   mode 1  global_load_lds -> a per-wave LDS ring of D k-steps, ds_read_b128, adds (no VGPR per load in flight)
//   mode 2  mode 0 + MFMAs        mode 3  mode 1 + MFMAs        mode 4  MFMAs only (the issue-rate ruler)
// Geometry per run: blocks per CU x waves per block.  Output: GB/s per CU, TB/s over the chip, B/clk/CU at 2.4 GHz, and for the
// MFMA modes the time against mode 4's.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_l2fill tools/ubench_l2fill.hip && ./ubench_l2fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 1536, BPAD = 64, KSTEPS = K / 16;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NW waves per block; each wave walks `steps` k-steps of its K-slice (wrapping), D ahead
template <int MODE, int D, int NTA, int NW, int PRIO = 0>
__global__ __launch_bounds__(64 * NW) void k_fill(const float4 *__restrict__ X, float *sink, int steps, unsigned long long *stamps) {
  const unsigned long long t0 = wall_clock64();
  if (PRIO && blockIdx.x < gridDim.x / 2) __builtin_amdgcn_s_setprio(PRIO);  // the first half of the grid = the product's main blocks
  constexpr int RX = D + 1, JJ = KSTEPS / NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, fg = lane >> 4;
  auto src = [&](int j) { return X + ((size_t)(4 * (wave * JJ + j % JJ) + fg)) * BPAD + fi; };
  f32x4 acc[NTA];
#pragma unroll
  for (int t = 0; t < NTA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + 1e-3f * lane;
  if (MODE == 4) {
    for (int j = 0; j < steps; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a + q, acc[t], 0, 0, 0);
    }
  } else if (MODE == 0 || MODE == 2) {
    float4 ring[RX][NTA];
#pragma unroll
    for (int p = 0; p < D; ++p) {
      const float4 *sp = src(p);
#pragma unroll
      for (int t = 0; t < NTA; ++t) ring[p][t] = sp[16 * t];
    }
    for (int base = 0; base < steps; base += RX) {
#pragma unroll
      for (int r = 0; r < RX; ++r) {
        const float4 *sp = src(base + r + D);
#pragma unroll
        for (int t = 0; t < NTA; ++t) ring[(r + D) % RX][t] = sp[16 * t];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const float4(&xv)[NTA] = ring[r];
        if (MODE == 2) {
#pragma unroll
          for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[t].x, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[t].y, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[t].z, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[t].w, acc[t], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < NTA; ++t) {
            acc[t][0] += xv[t].x;
            acc[t][1] += xv[t].y;
            acc[t][2] += xv[t].z;
            acc[t][3] += xv[t].w;
          }
        }
      }
    }
  } else {
    // per-wave LDS ring: slot r, tile t at lds[((wave * RX + r) * NTA + t) * 256 + lane * 4]
    float *mine = lds + (size_t)wave * RX * NTA * 256;
    const unsigned my_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)(mine + lane * 4);
    auto issue = [&](int j, int slot) {
      const float4 *sp = src(j);
#pragma unroll
      for (int t = 0; t < NTA; ++t)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp + 16 * t),
                                         (__attribute__((address_space(3))) void *)(mine + (slot * NTA + t) * 256), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < D; ++p) issue(p, p);
    for (int base = 0; base < steps; base += RX) {
#pragma unroll
      for (int r = 0; r < RX; ++r) {
        issue(base + r + D, (r + D) % RX);
        wait_vm<D * NTA>();  // everything but the D newest k-steps has landed
        // (the reads as inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every ds_read it can see while an LDS-DMA is in flight)
        f32x4 xv[NTA];
#pragma unroll
        for (int t = 0; t < NTA; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(xv[t]) : "v"(my_addr + (unsigned)((r * NTA + t) * 1024)) : "memory");
#pragma unroll
        for (int t = 0; t < NTA; ++t) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xv[t])::"memory");  // (tied to the data: an MFMA must not be moved above it)
        if (MODE == 3) {
#pragma unroll
          for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[t].x, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[t].y, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[t].z, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[t].w, acc[t], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < NTA; ++t) {
            acc[t][0] += xv[t].x;
            acc[t][1] += xv[t].y;
            acc[t][2] += xv[t].z;
            acc[t][3] += xv[t].w;
          }
        }
      }
    }
    wait_vm<0>();
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NTA; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678f) sink[blockIdx.x * blockDim.x + tid] = s;
  if (stamps && tid == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = wall_clock64();
  }
}

static float *d_x, *d_sink;
static double mfma_ref_us[8];  // [geometry index]

template <int MODE, int D, int NTA, int NW>
static void run(int bpc, int steps, int gi, const char *label) {
  constexpr int RX = D + 1;
  const size_t lds = (MODE == 1 || MODE == 3) ? (size_t)NW * RX * NTA * 1024 : 0;
  if (lds * bpc > 160 * 1024) {
    printf("%-28s bpc %d waves %d D %d NTA %d: ring does not fit LDS\n", label, bpc, NW, D, NTA);
    return;
  }
  auto fn = k_fill<MODE, D, NTA, NW>;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64 * NW, lds));
  const int grid = 256 * bpc, st = steps / RX * RX;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * NW), lds, 0, reinterpret_cast<const float4 *>(d_x), d_sink, st, (unsigned long long *)nullptr);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * NW), lds, 0, reinterpret_cast<const float4 *>(d_x), d_sink, st, (unsigned long long *)nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = 1e3 * best, bytes_cu = (double)bpc * NW * st * NTA * 1024.0;
  if (MODE == 4) mfma_ref_us[gi] = us;
  if (MODE == 4)
    printf("%-28s bpc %d waves %d NTA %d occ %d: %9.1f us  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", label, bpc, NW, NTA, occ, us,
           us * 2400.0 / ((double)bpc * NW / 4.0 * st * 4 * NTA));
  else
    printf("%-28s bpc %d waves %d D %d NTA %d occ %d: %9.1f us  %6.1f GB/s per CU  %5.1f TB/s chip  %5.1f B/clk/CU%s", label, bpc, NW, D, NTA, occ, us,
           bytes_cu / us * 1e-3, 256.0 * bytes_cu / us * 1e-6, bytes_cu / (us * 2400.0), (MODE == 2 || MODE == 3) ? "" : "\n");
  if (MODE == 2 || MODE == 3) printf("  x%.2f of MFMA-only\n", us / mfma_ref_us[gi]);
}

// two blocks per CU of the same loop: when does the first half of the grid (s_setprio PRIO) finish, when the second?
template <int MODE, int D, int NTA, int PRIO>
static void run_prio(int steps, const char *label) {
  constexpr int NW = 8, RX = D + 1;
  auto fn = k_fill<MODE, D, NTA, NW, PRIO>;
  unsigned long long *st;
  CK(hipMalloc((void **)&st, 1024 * 16));
  const int stp = steps / RX * RX;
  double hi = 0, lo = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(fn, dim3(512), dim3(64 * NW), 0, 0, reinterpret_cast<const float4 *>(d_x), d_sink, stp, st);
    CK(hipDeviceSynchronize());
    static unsigned long long h[1024];
    CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < 512; ++b) tmin = h[2 * b] < tmin ? h[2 * b] : tmin;
    hi = lo = 0;
    for (int b = 0; b < 512; ++b) (b < 256 ? hi : lo) += 0.01 * (double)(h[2 * b + 1] - tmin) / 256.0;
  }
  printf("%-28s D %d NTA %d prio %d: first half of the grid done at %7.1f us, second half at %7.1f us (mean block end)\n", label, D, NTA, PRIO, hi, lo);
  CK(hipFree(st));
}

int main(int argc, char **argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 2400;
  CK(hipMalloc((void **)&d_x, (size_t)(K / 4 + 16) * BPAD * 16));
  CK(hipMemset(d_x, 0, (size_t)(K / 4 + 16) * BPAD * 16));
  CK(hipMalloc((void **)&d_sink, 1 << 22));
  printf("operand %d kB per block pass, %d k-steps per wave and launch\n", K * BPAD * 4 / 1024, steps);
  // geometry 0: 2 blocks x 8 waves (the product's), 1: 1 x 8, 2: 1 x 4
  run<4, 1, 4, 8>(2, steps, 0, "mfma only");
  run<4, 1, 4, 8>(1, steps, 1, "mfma only");
  run<4, 1, 4, 4>(1, steps, 2, "mfma only");
  printf("-- loads only, registers\n");
  run<0, 1, 4, 8>(2, steps, 0, "regs");
  run<0, 2, 4, 8>(2, steps, 0, "regs");
  run<0, 3, 4, 8>(2, steps, 0, "regs");
  run<0, 4, 4, 8>(2, steps, 0, "regs");
  run<0, 2, 4, 8>(1, steps, 1, "regs");
  run<0, 4, 4, 8>(1, steps, 1, "regs");
  run<0, 8, 4, 8>(1, steps, 1, "regs");
  run<0, 4, 4, 4>(1, steps, 2, "regs");
  run<0, 8, 4, 4>(1, steps, 2, "regs");
  run<0, 12, 4, 4>(1, steps, 2, "regs");
  run<0, 2, 2, 8>(2, steps, 0, "regs (2 tiles)");
  run<0, 4, 2, 8>(2, steps, 0, "regs (2 tiles)");
  run<0, 2, 1, 8>(2, steps, 0, "regs (1 tile)");
  run<0, 6, 1, 8>(2, steps, 0, "regs (1 tile)");
  printf("-- loads only, global_load_lds\n");
  run<1, 1, 4, 8>(2, steps, 0, "glds");
  run<1, 2, 4, 8>(1, steps, 1, "glds");
  run<1, 4, 4, 8>(1, steps, 1, "glds");
  run<1, 4, 4, 4>(1, steps, 2, "glds");
  run<1, 8, 4, 4>(1, steps, 2, "glds");
  run<1, 3, 2, 8>(2, steps, 0, "glds (2 tiles)");
  printf("-- with the MFMAs that consume them\n");
  run<2, 1, 4, 8>(2, steps, 0, "regs + mfma");
  run<2, 2, 4, 8>(2, steps, 0, "regs + mfma");
  run<2, 3, 4, 8>(2, steps, 0, "regs + mfma");
  run<2, 4, 4, 8>(1, steps, 1, "regs + mfma");
  run<2, 8, 4, 8>(1, steps, 1, "regs + mfma");
  run<2, 8, 4, 4>(1, steps, 2, "regs + mfma");
  run<2, 12, 4, 4>(1, steps, 2, "regs + mfma");
  run<3, 1, 4, 8>(2, steps, 0, "glds + mfma");
  run<3, 2, 4, 8>(1, steps, 1, "glds + mfma");
  run<3, 4, 4, 8>(1, steps, 1, "glds + mfma");
  run<3, 4, 4, 4>(1, steps, 2, "glds + mfma");
  run<3, 8, 4, 4>(1, steps, 2, "glds + mfma");
  printf("-- priority between the two blocks of a CU (product geometry, 2 x 8 waves)\n");
  run_prio<4, 1, 4, 0>(steps, "mfma only");
  run_prio<4, 1, 4, 3>(steps, "mfma only");
  run_prio<2, 1, 4, 0>(steps, "regs + mfma");
  run_prio<2, 1, 4, 3>(steps, "regs + mfma");
  run_prio<2, 3, 4, 0>(steps, "regs + mfma");
  run_prio<2, 3, 4, 3>(steps, "regs + mfma");
  run_prio<2, 2, 2, 3>(steps, "regs + mfma");
  run_prio<2, 4, 2, 3>(steps, "regs + mfma");
  return 0;
}
