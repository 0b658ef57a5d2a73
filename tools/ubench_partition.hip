// Micro-benchmark (dev tool, not product): the 2-D partition VERDICT round 5 asked about for the batched engine's LSTM passes,
// WITH the weight stream that tools/ubench_l2fill.hip leaves out.  One "pass" = the decoder-LSTM pre-activations of a step at 64 chunks:
// W [4096 rows x 2560 columns] (42 MB, re-read from the Infinity Cache / HBM every pass, like every decoder step) against the shared
// activation operand X [2560/4][64 chunks][4] (655 kB, L2-resident), on v_mfma_f32_16x16x4_f32, 256 blocks of 8 waves that split K.
//   form A (the product): block = 16 rows x 4 tiles of 16 chunks: per k-step and wave 1 weight quad + 4 operand quads -> 16 MFMAs
//   form B (2-D)        : block = 32 rows x 2 tiles            : per k-step and wave 2 weight quads + 2 operand quads -> 16 MFMAs
//                         (the two blocks of a row group read the same 328 kB of weights; `pair` = how far apart they are in the grid:
//                          1 = neighbours, i.e. different XCDs under round-robin dispatch; 8 = the same XCD)
//   form C              : B's loop with the weights of A (one weight quad, 2 tiles): half the MFMAs -- what the operand side alone costs
// Also with a second block per CU running the same loop over another matrix (the product's "early partial" blocks): bpc 2.
// Output: us per pass, TB/s of weight stream, MFMA issue fraction.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_partition tools/ubench_partition.hip && ./ubench_partition
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 2560, BPAD = 64, KSTEPS = K / 16, NW = 8, JJ = KSTEPS / NW;  // 20 k-steps per wave and pass

// RG row groups of 16 rows, NT tiles of 16 chunks per block.  Weights in MFMA-fragment order: slice s = [KSTEPS][64 lanes] float4.
template <int RG, int NT, int DW, int DX>
__global__ __launch_bounds__(64 * NW, 4) void k_pass(const float4 *__restrict__ W, const float4 *__restrict__ X, float *sink, int passes, int pair, int nslices) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, fg = lane >> 4;
  // which weight slices, which tiles.  The second block of a CU (blockIdx >= 256) takes "another matrix": slices 256 on.
  const int mat = blockIdx.x / 256, b = blockIdx.x % 256;
  int s0, t0;
  if (RG == 1) {
    s0 = b;
    t0 = 0;
  } else {  // partner blocks b and b ^ pair (pair a power of two) share a row group = two slices; they take the two halves of the tiles
    const int hi = (b / pair) & 1, g = (b / (2 * pair)) * pair + b % pair;
    s0 = 2 * g;
    t0 = hi * NT;
  }
  const float4 *w0 = W + ((size_t)(mat * 256 + s0) % nslices * KSTEPS + wave * JJ) * 64 + lane;
  const float4 *x0 = X + ((size_t)(4 * wave * JJ + fg)) * BPAD + 16 * t0 + fi;
  f32x4 acc[RG][NT];
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int RW = DW + 1, RX = DX + 1;
  const float4 *w00 = w0, *x00 = x0;
  for (int p = 0; p < passes; ++p) {
    size_t oz = 0;
    asm volatile("" : "+s"(oz));  // (an opaque zero: every pass must issue its loads again, the addresses are the same)
    w0 = w00 + oz;
    x0 = x00 + oz;
    float4 wr[RW][RG], xr[RX][NT];
#pragma unroll
    for (int j = 0; j < DW; ++j)
#pragma unroll
      for (int r = 0; r < RG; ++r) wr[j][r] = w0[((size_t)r * KSTEPS + j) * 64];
#pragma unroll
    for (int j = 0; j < DX; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) xr[j][t] = x0[(size_t)4 * j * BPAD + 16 * t];
#pragma unroll
    for (int j = 0; j < JJ; ++j) {
      if (j + DW < JJ)
#pragma unroll
        for (int r = 0; r < RG; ++r) wr[(j + DW) % RW][r] = w0[((size_t)r * KSTEPS + j + DW) * 64];
      if (j + DX < JJ)
#pragma unroll
        for (int t = 0; t < NT; ++t) xr[(j + DX) % RX][t] = x0[(size_t)4 * (j + DX) * BPAD + 16 * t];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const float4(&wv)[RG] = wr[j % RW];
      const float4(&xv)[NT] = xr[j % RX];
#pragma unroll
      for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r].x, xv[t].x, acc[r][t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r].y, xv[t].y, acc[r][t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r].z, xv[t].z, acc[r][t], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r].w, xv[t].w, acc[r][t], 0, 0, 0);
    }
    __syncthreads();  // (a pass ends in the product with the waves meeting for the K reduction)
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t) s += acc[r][t][0] + acc[r][t][1] + acc[r][t][2] + acc[r][t][3];
  if (s == 12345.678f) sink[blockIdx.x * blockDim.x + tid] = s;
}

static float4 *d_w, *d_x;
static float *d_sink;
constexpr int NSLICES = 512;  // two matrices of 256 slices (84 MB)

template <int RG, int NT, int DW, int DX>
static void run(int bpc, int pair, int passes, const char *label) {
  auto fn = k_pass<RG, NT, DW, DX>;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fn, dim3(256 * bpc), dim3(64 * NW), 0, 0, d_w, d_x, d_sink, passes, pair, NSLICES);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(256 * bpc), dim3(64 * NW), 0, 0, d_w, d_x, d_sink, passes, pair, NSLICES);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = 1e3 * best / passes;
  const double wbytes = (double)bpc * 256 * RG * KSTEPS * 1024.0;  // weight bytes requested per pass (form B: every slice twice)
  const double mfma = (double)bpc * NW / 4.0 * JJ * 4 * RG * NT * 32.0 / 2400.0;  // us of matrix-pipe issue per SIMD and pass
  printf("%-44s bpc %d pair %d DW %d DX %d: %6.2f us per pass   weights %5.2f TB/s requested   matrix pipe %4.2f us = %4.2f of the pass\n", label, bpc, pair, DW, DX, us,
         wbytes / us * 1e-6, mfma, mfma / us);
}

int main(int argc, char **argv) {
  const int passes = argc > 1 ? atoi(argv[1]) : 400;
  const size_t wn = (size_t)NSLICES * KSTEPS * 64, xn = (size_t)(K / 4 + 64) * BPAD;
  CK(hipMalloc((void **)&d_w, wn * 16));
  CK(hipMemset(d_w, 0, wn * 16));
  CK(hipMalloc((void **)&d_x, xn * 16));
  CK(hipMemset(d_x, 0, xn * 16));
  CK(hipMalloc((void **)&d_sink, 1 << 22));
  printf("one pass = W[4096 x 2560] (42 MB per matrix) . X[2560 x 64 chunks]; %d passes per launch, best of 5\n", passes);
  printf("-- one block per CU\n");
  run<1, 4, 2, 1>(1, 1, passes, "A: 16 rows x 4 tiles (the product)");
  run<1, 4, 3, 2>(1, 1, passes, "A: 16 rows x 4 tiles, deeper rings");
  run<2, 2, 2, 1>(1, 1, passes, "B: 32 rows x 2 tiles, partner next door");
  run<2, 2, 2, 1>(1, 8, passes, "B: 32 rows x 2 tiles, partner on the XCD");
  run<2, 2, 3, 2>(1, 8, passes, "B: ... deeper rings");
  run<1, 2, 2, 1>(1, 1, passes, "C: 16 rows x 2 tiles (half the work)");
  run<1, 1, 2, 1>(1, 1, passes, "   16 rows x 1 tile");
  printf("-- two blocks per CU (the second over another matrix: the product's early-partial blocks)\n");
  run<1, 4, 2, 1>(2, 1, passes, "A: 16 rows x 4 tiles (the product)");
  run<2, 2, 2, 1>(2, 1, passes, "B: 32 rows x 2 tiles, partner next door");
  run<2, 2, 2, 1>(2, 8, passes, "B: 32 rows x 2 tiles, partner on the XCD");
  run<1, 2, 2, 1>(2, 1, passes, "C: 16 rows x 2 tiles (half the work)");
  return 0;
}
