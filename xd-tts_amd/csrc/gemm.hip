// gemm.hip -- C = act(A W^T + bias) (+R) in exact fp32 on the gfx950 matrix cores
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, bit-equal to an fmaf chain).
//
// Every dense contraction off the per-frame critical path goes through this one kernel:
//   * post-net conv1d x5 (src/tacotron2/mod.rs:347, graph postnet.onnx) and encoder conv1d x3
//     (mod.rs:379): with time-major activations [T][C] and weights re-laid as [co][k][ci] a conv
//     row is one contiguous window of the zero-padded input, i.e. a GEMM whose A rows overlap
//     (lda = C instead of k*C) -- implicit GEMM with no im2col buffer;
//   * BiLSTM input projections, the attention memory layer, and the Griffin-Lim mel->linear
//     product (pinv(mel_basis) . exp(mel)).
// Tile: 32x32 or 64x64 per 256-thread block (one or 2x2 16x16 MFMA tiles per wave), K-slab 32 through
// LDS (three buffers and a three-stage software pipeline for the 32x32 form, two buffers for 64x64), global loads
// four slabs ahead, block -> tile mapping chosen by XCD (below).  Within a slab the contraction
// index is permuted (k = 4*(lane>>4) + kk) so each lane fetches its four K values of a tile row
// with one ds_read_b128.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace xdtts {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Output tile per 256-thread block: 2 x 2 waves, each MT x NT MFMA tiles of 16x16.
//   32x32 (MT = NT = 1): the single-utterance shapes (M = 100..800 rows put only 1-2 blocks on a CU) -- measured
//     the fastest there (round 1: 64x64 post-net 0.49 / 0.79 ms against 0.38 ms; now 0.15 ms);
//   64x64 (MT = NT = 2): batches whose grid fills the chip several times over (the 52-chunk post-net: M = 25 171
//     rows in all): four times the MFMA work per staged slab and per barrier.
// K-slab 32 through LDS, one barrier per slab.  A slab's MFMAs take 0.1-0.4 us and an L2 round
// trip ~0.7 us, so the global loads run FOUR slabs ahead in named registers (an indexed ring was demoted to
// scratch by the compiler), unconditional and clamped into the operand so nothing depends on their data until
// the slab is staged (the zero fill happens there).  Every output element accumulates its K products in
// ascending order in both shapes, so they give identical results.
constexpr int BK = 32, LDS_LD = 36;  // 144-byte rows: 16-B aligned float4 reads
template <bool C, class T>
__device__ __forceinline__ T &pick(T &a, T &b) {  // one of two named register sets, chosen at compile time
  if constexpr (C) return a;
  else return b;
}
#define P3_OR(a, b) pick<P3>(a, b)
// Issue order of a pipelined step: one memory instruction behind each MFMA (a dependent v_mfma_f32_16x16x4_f32 issues ~44
// cycles after its predecessor, the wave is otherwise idle in between) -- the LDS stores and the global loads first, the
// fragment reads of the next slab last.  Post-net at F = 800: 157 -> 142 us over its five launches; with the stage / fetch
// group fenced off ahead of the MFMAs (sched_barrier) the two phases of a wave ran back to back.
#define GEMM_SG(m) __builtin_amdgcn_sched_group_barrier(m, 1, 0)
#define GEMM_INTERLEAVE                                                                                                   \
  GEMM_SG(0x008); GEMM_SG(0x200); GEMM_SG(0x008); GEMM_SG(0x200); GEMM_SG(0x008); GEMM_SG(0x020); GEMM_SG(0x008); GEMM_SG(0x020); \
  GEMM_SG(0x008); GEMM_SG(0x100); GEMM_SG(0x008); GEMM_SG(0x100); GEMM_SG(0x008); GEMM_SG(0x100); GEMM_SG(0x008); GEMM_SG(0x100)

template <int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmArgs g) {
  constexpr int MT = BM / 32, NT = BN / 32;  // MFMA tiles per wave; also: float4 loads per thread and slab
  constexpr bool P3 = BM == 32;              // three-stage software pipeline (below) or the two-buffer form
  constexpr int NBUF = P3 ? 3 : 2;           // LDS slabs (P3: one being read into fragments, one staged, one in between)
  __shared__ __attribute__((aligned(16))) float As[NBUF][BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][BN * LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), so with a grid whose x extent is a multiple of 8 an
  // XCD sees a fixed set of column tiles and EVERY row tile: A crosses into all eight L2s.  That is the cheap way round for the
  // single-utterance shapes (A 1.6 MB, W 5.2 MB), and the expensive one for a batch (52-chunk post-net: A 51 MB, W 5.2 MB --
  // the counters showed 412 MB fetched per launch).  g.xcd_rows turns the mapping round: of 8 consecutive dispatch slots each
  // takes another row tile and keeps it for gridDim.x consecutive visits, i.e. the column tiles of one row tile share an XCD
  // (the row tiles of all items of a batch are numbered through, so the XCDs get equal shares whatever the items' lengths).
  int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
  // split-K (plain mapping only): grid z = item x K-slice.  The slices of a tile meet in a workspace; the last one to arrive sums
  // them in slice order (the same bits whatever the arrival order) and runs the epilogue.
  const int nks = g.splitk > 1 ? g.splitk : 1;
  const int ks = nks > 1 ? (int)blockIdx.z % nks : 0;
  if (nks > 1) z = (int)blockIdx.z / nks;
  if (g.xcd_rows) {  // grid (column tiles, row tiles of ALL items rounded up to 8, 1): row tile G of the batch -> XCD G % 8
    const int nx = gridDim.x, L = bx + nx * by, j = L % (8 * nx);
    int G = (L / (8 * nx)) * 8 + (j & 7);
    bx = j >> 3;
    for (z = 0; z < g.batch; ++z) {  // (scalar: at most 64 items)
      const int t = ((g.ragged ? g.Mz[z] : g.M) + BM - 1) / BM;
      if (G < t) break;
      G -= t;
    }
    if (z == g.batch) return;  // padding of the last group of 8
    by = G;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int M = g.ragged ? g.Mz[z] : g.M;
  if (m0 >= M) return;  // ragged batch: this item has fewer rows than the longest
  const float *A = g.A + (size_t)z * g.strideA;
  float *C = g.C + (size_t)z * g.strideC + (g.ragged ? g.Cz[z] : 0);
  const float *R = g.R ? g.R + (size_t)z * g.strideR : nullptr;

  // global -> LDS assignment: thread loads MT float4 of A and NT of W per slab (rows lr + 32 r)
  const int lr = tid >> 3, lc = (tid & 7) * 4;
  bool a_ok[MT], b_ok[NT];
  const float *a_src[MT], *b_src[NT];
#pragma unroll
  for (int r = 0; r < MT; ++r) {
    a_ok[r] = m0 + lr + 32 * r < M;
    a_src[r] = A + (size_t)(a_ok[r] ? m0 + lr + 32 * r : M - 1) * g.lda + lc;
  }
#pragma unroll
  for (int r = 0; r < NT; ++r) {
    b_ok[r] = n0 + lr + 32 * r < g.N;
    b_src[r] = g.W + (size_t)(b_ok[r] ? n0 + lr + 32 * r : g.N - 1) * g.K + lc;
  }
  // this block's K range [kbeg, kbeg + KL): whole slabs per slice
  const int kper = ((g.K + BK - 1) / BK + nks - 1) / nks * BK, kbeg = ks * kper, KL = min(g.K - kbeg, kper);
#pragma unroll
  for (int r = 0; r < MT; ++r) a_src[r] += kbeg;
#pragma unroll
  for (int r = 0; r < NT; ++r) b_src[r] += kbeg;
  const int nslab = (KL + BK - 1) / BK;
  const int fi = lane & 15, fg = lane >> 4;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  struct Slab {
    float4 a[MT], b[NT];
  };
  struct Frag {  // a slab's MFMA operands of this lane: [16-column group][tile]
    float4 a[BK / 16][MT], b[BK / 16][NT];
  };

  // Software pipeline over the K-slabs, three stages deep (a wave has the CU's matrix pipe to itself at the
  // single-utterance sizes -- 1-2 blocks per CU --, so whatever it waits for is idle MFMA time: with the fragments read
  // from LDS right before their MFMAs and the next slab staged after them, a slab took 0.48 us for 0.11 us of MFMA):
  //   step s:  slab s + 2 goes from its prefetch registers into LDS buffer (s + 2) % 3 and that register slot is refilled
  //            from global memory (slab s + 6: four slabs ahead of the staging point);
  //            the fragments of slab s + 1 are read from buffer (s + 1) % 3 into the second fragment set;
  //            the MFMAs of slab s run on the set read during step s - 1;  one barrier.
  // Buffer (s + 2) % 3 held slab s - 1, whose fragments every wave read before the barrier that ended step s - 2.
#define GEMM_FETCH(SLAB, Q)                                                                 \
  do {                                                                                      \
    const int k0_ = ((SLAB) < nslab && (SLAB) * BK + lc < KL) ? (SLAB) * BK : 0;           \
    _Pragma("unroll") for (int r_ = 0; r_ < MT; ++r_) Q.a[r_] = *reinterpret_cast<const float4 *>(a_src[r_] + k0_); \
    _Pragma("unroll") for (int r_ = 0; r_ < NT; ++r_) Q.b[r_] = *reinterpret_cast<const float4 *>(b_src[r_] + k0_); \
  } while (0)
#define GEMM_STAGE(SLAB, BUF, Q)                                                            \
  do {                                                                                      \
    const bool kok_ = (SLAB) * BK + lc < KL; /* K is a multiple of 16, not always of 32 */ \
    /* no `cond ? Q : zero4` on the vector class: it selects between ADDRESSES and sends the  \
       prefetch registers to scratch */                                                      \
    _Pragma("unroll") for (int r_ = 0; r_ < MT; ++r_) {                                     \
      const float ma_ = (kok_ && a_ok[r_]) ? 1.f : 0.f;                                     \
      *reinterpret_cast<float4 *>(&As[BUF][(lr + 32 * r_) * LDS_LD + lc]) =                 \
          make_float4(ma_ != 0.f ? Q.a[r_].x : 0.f, ma_ != 0.f ? Q.a[r_].y : 0.f, ma_ != 0.f ? Q.a[r_].z : 0.f, ma_ != 0.f ? Q.a[r_].w : 0.f); \
    }                                                                                       \
    _Pragma("unroll") for (int r_ = 0; r_ < NT; ++r_) {                                     \
      const float mb_ = (kok_ && b_ok[r_]) ? 1.f : 0.f;                                     \
      *reinterpret_cast<float4 *>(&Bs[BUF][(lr + 32 * r_) * LDS_LD + lc]) =                 \
          make_float4(mb_ != 0.f ? Q.b[r_].x : 0.f, mb_ != 0.f ? Q.b[r_].y : 0.f, mb_ != 0.f ? Q.b[r_].z : 0.f, mb_ != 0.f ? Q.b[r_].w : 0.f); \
    }                                                                                       \
  } while (0)
#define GEMM_READ(BUF, F)                                                                   \
  do {                                                                                      \
    _Pragma("unroll") for (int kg = 0; kg < BK / 16; ++kg) {                                \
      _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                     \
        F.a[kg][i_] = *reinterpret_cast<const float4 *>(&As[BUF][(wm * 16 * MT + 16 * i_ + fi) * LDS_LD + kg * 16 + fg * 4]); \
      _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                     \
        F.b[kg][j_] = *reinterpret_cast<const float4 *>(&Bs[BUF][(wn * 16 * NT + 16 * j_ + fi) * LDS_LD + kg * 16 + fg * 4]); \
    }                                                                                       \
  } while (0)
#define GEMM_MMA(F)                                                                         \
  do {                                                                                      \
    _Pragma("unroll") for (int kg = 0; kg < BK / 16; ++kg) {                                \
      _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                     \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[kg][i_].x, F.b[kg][j_].x, acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                     \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[kg][i_].y, F.b[kg][j_].y, acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                     \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[kg][i_].z, F.b[kg][j_].z, acc[i_][j_], 0, 0, 0); \
      _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                     \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[kg][i_].w, F.b[kg][j_].w, acc[i_][j_], 0, 0, 0); \
    }                                                                                       \
  } while (0)
  // step J of a round of 12 (register slots, LDS buffers and fragment sets are all literals then).
  // Full rounds run WITHOUT the per-step bounds checks: a conditional step is a control-flow join, and at a join the
  // compiler's waitcnt pass no longer knows how many younger loads are in flight -- it then waits for vmcnt(1)/(0)
  // before staging, i.e. for the loads it has just issued (seen in the ISA of the round-1 kernel: the four-slab
  // prefetch was lost in most steps and a slab cost an L2 round trip).  Straight-line rounds get vmcnt(6..9).
  //   P3 (32x32 tile): the three-stage form above -- slot (J + 2) % 4, buffers (J + 1) % 3 / (J + 2) % 3, fragment sets by parity;
  //   else (64x64 tile, four independent MFMA chains per wave, 2-4 blocks per CU): two LDS buffers, the fragments read right
  //   before their MFMAs -- slot (J + 1) % 4, buffers J % 2 / (J + 1) % 2 -- which keeps 37 KB of LDS per block.
#define GEMM_STEP_U(J, QA, QB, FC, FN)                                                      \
  if constexpr (P3) {                                                                       \
    GEMM_STAGE(s0 + (J) + 2, ((J) + 2) % NBUF, QA);                                         \
    GEMM_FETCH(s0 + (J) + 6, QA); /* the slot just staged: refill */                        \
    GEMM_READ(((J) + 1) % NBUF, FN);                                                        \
    GEMM_MMA(FC);                                                                           \
    GEMM_INTERLEAVE;                                                                        \
    __syncthreads();                                                                        \
  } else {                                                                                  \
    GEMM_FETCH(s0 + (J) + 4, QB); /* slot J % 4 was staged for this slab already: refill */ \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    GEMM_READ((J) % NBUF, f0);                                                              \
    GEMM_MMA(f0);                                                                           \
    GEMM_STAGE(s0 + (J) + 1, ((J) + 1) % NBUF, QA); /* buffer last read before the previous barrier */ \
    __syncthreads();                                                                        \
  }
#define GEMM_STEP(J, QA, QB, FC, FN)                                                        \
  if (s0 + (J) < nslab) {                                                                   \
    if constexpr (P3) {                                                                     \
      if (s0 + (J) + 2 < nslab) GEMM_STAGE(s0 + (J) + 2, ((J) + 2) % NBUF, QA);             \
      GEMM_FETCH(s0 + (J) + 6, QA);                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                    \
      if (s0 + (J) + 1 < nslab) GEMM_READ(((J) + 1) % NBUF, FN);                            \
      GEMM_MMA(FC);                                                                         \
    } else {                                                                                \
      GEMM_FETCH(s0 + (J) + 4, QB);                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                    \
      GEMM_READ((J) % NBUF, f0);                                                            \
      GEMM_MMA(f0);                                                                         \
      if (s0 + (J) + 1 < nslab) GEMM_STAGE(s0 + (J) + 1, ((J) + 1) % NBUF, QA);             \
    }                                                                                       \
    __syncthreads();                                                                        \
  }
  // (QA, QB) of step J: P3 stages slot (J + 2) % 4; the two-buffer form refills slot J % 4 = QB and stages slot (J + 1) % 4 = QA
#define GEMM_ROUND(STEP)                                                                    \
  STEP(0, P3_OR(q2, q1), q0, f0, f1)                                                        \
  STEP(1, P3_OR(q3, q2), q1, f1, f0)                                                        \
  STEP(2, P3_OR(q0, q3), q2, f0, f1)                                                        \
  STEP(3, P3_OR(q1, q0), q3, f1, f0)                                                        \
  STEP(4, P3_OR(q2, q1), q0, f0, f1)                                                        \
  STEP(5, P3_OR(q3, q2), q1, f1, f0)                                                        \
  STEP(6, P3_OR(q0, q3), q2, f0, f1)                                                        \
  STEP(7, P3_OR(q1, q0), q3, f1, f0)                                                        \
  STEP(8, P3_OR(q2, q1), q0, f0, f1)                                                        \
  STEP(9, P3_OR(q3, q2), q1, f1, f0)                                                        \
  STEP(10, P3_OR(q0, q3), q2, f0, f1)                                                       \
  STEP(11, P3_OR(q1, q0), q3, f1, f0)

  Slab q0, q1, q2, q3;
  Frag f0, f1;
  GEMM_FETCH(0, q0);
  GEMM_FETCH(1, q1);
  GEMM_FETCH(2, q2);
  GEMM_FETCH(3, q3);
  GEMM_STAGE(0, 0, q0);
  if constexpr (P3) {
    if (1 < nslab) GEMM_STAGE(1, 1, q1);
    GEMM_FETCH(4, q0);
    GEMM_FETCH(5, q1);
    __syncthreads();
    GEMM_READ(0, f0);
  } else {
    __syncthreads();
  }
  int s0 = 0;
  for (; s0 + 14 <= nslab; s0 += 12) {  // every slab this round stages (up to s0 + 13) exists
    GEMM_ROUND(GEMM_STEP_U)
  }
  for (; s0 < nslab; s0 += 12) {  // the last 2..13 slabs
    GEMM_ROUND(GEMM_STEP)
  }
#undef GEMM_ROUND
#undef GEMM_STEP_U
#undef GEMM_STEP
#undef GEMM_MMA
#undef GEMM_READ
#undef GEMM_STAGE
#undef GEMM_FETCH
  if (nks > 1) {
    // the slices' partial tiles, lane-linear [tile][slice][MFMA tile (i, j)][wave][lane] float4: plain 16-byte stores, one
    // agent-scope release per block, then the ticket
    __shared__ int s_last;
    const size_t tile_id = ((size_t)z * gridDim.y + by) * gridDim.x + bx;
    // write-through (sc1) 16-byte stores, drained, then a relaxed ticket; the reducer reads with sc1 loads: no release / acquire
    // fence (a fence per block -- buffer_wbl2 -- cost more than the split gained: post-net 0.153 -> 0.19 ms at two slices)
    float *wsb = g.ws + tile_id * nks * (MT * NT * 1024);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)wsb, 0, 0x7fffffff, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        u32x4 v;
        v.x = __float_as_uint(acc[i][j][0]), v.y = __float_as_uint(acc[i][j][1]), v.z = __float_as_uint(acc[i][j][2]), v.w = __float_as_uint(acc[i][j][3]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(((ks * MT * NT + i * NT + j) * 256 + tid) * 16), 0, 16);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(g.cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = t == (unsigned)(nks - 1);
      if (s_last) __hip_atomic_store(g.cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (every slice has arrived: ready for the next launch)
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nks; ++k) {
          const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((k * MT * NT + i * NT + j) * 256 + tid) * 16), 0, 16);
          v += (f32x4){__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
        }
        acc[i][j] = v;
      }
  }
  // epilogue: D register r of lane l holds row (l>>4)*4 + r, column l&15 of its 16x16 tile
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * 16 * NT + 16 * j + fi;
    if (n >= g.N) continue;
    const float bz = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 16 * MT + 16 * i + fg * 4 + r;
        if (m >= M) continue;
        float v = (g.alpha != 0.f ? g.alpha * acc[i][j][r] : acc[i][j][r]) + bz;
        const float rv = R ? (g.beta != 0.f ? g.beta : 1.f) * R[(size_t)m * g.ldr + n] : 0.f;
        if (g.r_before_act) v += rv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        else if (g.act == 2) v = tanhf(v);
        else if (g.act == 3) v = powf(fmaxf(v, 0.f), g.p);
        if (!g.r_before_act) v += rv;
        if (g.transpose_out) C[(size_t)n * (g.ldc_rows ? (long)M : g.ldc) + m] = v;
        else C[(size_t)m * g.ldc + n] = v;
      }
    }
  }
}

// ids -> embedding rows, written into the interior of the zero-padded conv input
__global__ void k_embed(const int64_t *ids, const float *__restrict__ emb, float *xpad, int T, int pad) {
  const int t = blockIdx.x, b = blockIdx.y;
  const int id = (int)ids[(size_t)b * T + t];
  const float4 *src = reinterpret_cast<const float4 *>(emb + (size_t)id * EMB);
  float4 *dst = reinterpret_cast<float4 *>(xpad + ((size_t)b * (T + 2 * pad) + pad + t) * EMB);
  for (int i = threadIdx.x; i < EMB / 4; i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// Encoder BiLSTM recurrence.  One block per (direction, chunk); thread r owns gate row r of the
// 1024 (i,f,g,o x 256); W_hh is stored transposed so the 1024 threads read consecutive floats.
// The input projection W_ih x + b was hoisted out as one GEMM over all T.
// The recurrence is 100 dependent steps on one CU, so what matters is how much of the 1 MB W_hh
// has to come through the CU's L2 port every step: each row keeps its first BL_REG weights in
// registers and the next BL_LDS in LDS (147 KB of the CU's 160 KB) for the whole sequence; only
// the remaining 156 columns (624 KB) are re-read from L2 per step, BL_CHUNK loads in flight per
// thread (the 128-VGPR budget of a 1024-thread block bounds BL_REG + BL_CHUNK).
constexpr int BL_REG = 64, BL_LDS = 36, BL_GLB = ENC_H - BL_REG - BL_LDS, BL_CHUNK = 12;
static_assert(BL_GLB % BL_CHUNK == 0 && BL_REG % 4 == 0 && BL_LDS % 4 == 0 && BL_CHUNK % 4 == 0, "bilstm column split");

__global__ __launch_bounds__(1024) void k_bilstm(const float *__restrict__ xproj,
                                                 const float *__restrict__ whhT_f,
                                                 const float *__restrict__ whhT_b, float *memory, int B,
                                                 int T) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *h = smem;                       // [256]
  float *gates = h + ENC_H;              // [1024]
  float *wl = gates + 4 * ENC_H;         // [BL_LDS][1024]
  const int dir = blockIdx.x, b = blockIdx.y, r = threadIdx.x;
  const float *whhT = dir ? whhT_b : whhT_f;
  const float *xp = xproj + ((size_t)dir * B + b) * T * (4 * ENC_H);
  float wreg[BL_REG];
#pragma unroll
  for (int j = 0; j < BL_REG; ++j) wreg[j] = whhT[(size_t)j * (4 * ENC_H) + r];
  for (int j = 0; j < BL_LDS; ++j) wl[j * (4 * ENC_H) + r] = whhT[(size_t)(BL_REG + j) * (4 * ENC_H) + r];
  const float *wg = whhT + (size_t)(BL_REG + BL_LDS) * (4 * ENC_H) + r;
  float c = 0.f;
  if (r < ENC_H) h[r] = 0.f;
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    // first chunk of the L2-served columns goes out before the register/LDS part
    float gv[BL_CHUNK];
#pragma unroll
    for (int j = 0; j < BL_CHUNK; ++j) gv[j] = wg[(size_t)j * (4 * ENC_H)];
    float a0 = xp[(size_t)t * (4 * ENC_H) + r], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int j = 0; j < BL_REG; j += 4) {
      a0 = fmaf(wreg[j + 0], h[j + 0], a0);
      a1 = fmaf(wreg[j + 1], h[j + 1], a1);
      a2 = fmaf(wreg[j + 2], h[j + 2], a2);
      a3 = fmaf(wreg[j + 3], h[j + 3], a3);
    }
#pragma unroll
    for (int j = 0; j < BL_LDS; j += 4) {
      a0 = fmaf(wl[(j + 0) * (4 * ENC_H) + r], h[BL_REG + j + 0], a0);
      a1 = fmaf(wl[(j + 1) * (4 * ENC_H) + r], h[BL_REG + j + 1], a1);
      a2 = fmaf(wl[(j + 2) * (4 * ENC_H) + r], h[BL_REG + j + 2], a2);
      a3 = fmaf(wl[(j + 3) * (4 * ENC_H) + r], h[BL_REG + j + 3], a3);
    }
#pragma unroll 1
    for (int j0 = 0; j0 < BL_GLB; j0 += BL_CHUNK) {
      float nx[BL_CHUNK];
      if (j0 + BL_CHUNK < BL_GLB) {
#pragma unroll
        for (int j = 0; j < BL_CHUNK; ++j) nx[j] = wg[(size_t)(j0 + BL_CHUNK + j) * (4 * ENC_H)];
      }
#pragma unroll
      for (int j = 0; j < BL_CHUNK; j += 4) {
        a0 = fmaf(gv[j + 0], h[BL_REG + BL_LDS + j0 + j + 0], a0);
        a1 = fmaf(gv[j + 1], h[BL_REG + BL_LDS + j0 + j + 1], a1);
        a2 = fmaf(gv[j + 2], h[BL_REG + BL_LDS + j0 + j + 2], a2);
        a3 = fmaf(gv[j + 3], h[BL_REG + BL_LDS + j0 + j + 3], a3);
      }
      if (j0 + BL_CHUNK < BL_GLB) {
#pragma unroll
        for (int j = 0; j < BL_CHUNK; ++j) gv[j] = nx[j];
      }
    }
    gates[r] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (r < ENC_H) {
      const float ig = sigm(gates[r]), fg = sigm(gates[ENC_H + r]);
      const float gg = tanhf(gates[2 * ENC_H + r]), og = sigm(gates[3 * ENC_H + r]);
      c = fmaf(fg, c, ig * gg);
      const float hn = og * tanhf(c);
      h[r] = hn;
      memory[((size_t)b * T + t) * EMB + dir * ENC_H + r] = hn;
    }
    __syncthreads();
  }
}

__global__ void k_transpose(const float *in, float *out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int x = blockIdx.x * 32 + threadIdx.x, y0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y)
    if (x < cols && y0 + j < rows) tile[j][threadIdx.x] = in[(size_t)(y0 + j) * cols + x];
  __syncthreads();
  const int ox = blockIdx.y * 32 + threadIdx.x, oy0 = blockIdx.x * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y)
    if (ox < rows && oy0 + j < cols) out[(size_t)(oy0 + j) * rows + ox] = tile[threadIdx.x][j];
}

struct RowCounts {
  int n[GEMM_RAGGED_MAX];
};
__global__ void k_copy_rows(const float *__restrict__ src, size_t src_stride, float *dst, size_t dst_stride, RowCounts rows, int cols4) {
  const int z = blockIdx.y;
  const size_t total = (size_t)rows.n[z] * cols4;
  const float4 *s4 = reinterpret_cast<const float4 *>(src + z * src_stride);
  float4 *d4 = reinterpret_cast<float4 *>(dst + z * dst_stride);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) d4[i] = s4[i];
}

}  // namespace

void launch_copy_rows(const float *src, size_t src_stride, float *dst, size_t dst_stride, const int *rows, int n, int cols,
                      hipStream_t s) {
  if (n > GEMM_RAGGED_MAX || cols % 4 != 0 || src_stride % 4 != 0 || dst_stride % 4 != 0) fail(XDTTS_ERR_BAD_ARG, "copy_rows: n=%d cols=%d", n, cols);
  RowCounts rc{};
  int most = 0;
  for (int z = 0; z < n; ++z) {
    rc.n[z] = rows[z];
    most = std::max(most, rows[z]);
  }
  if (most == 0) return;
  const int blocks = std::min(64, (most * (cols / 4) + 255) / 256);
  hipLaunchKernelGGL(k_copy_rows, dim3(blocks, n), dim3(256), 0, s, src, src_stride, dst, dst_stride, rc, cols / 4);
  HIP_CHECK(hipGetLastError());
}

// Split-K for the single-utterance shapes: M = 100..800 rows of 32x32 tiles put 100..400 blocks of four waves on 256 CUs -- one or
// two waves per SIMD, each a dependent MFMA chain behind a barrier per slab.  With the K range cut in 2..4 slices the same launch
// holds 3-4 blocks per CU; the slices meet in the launch (k_gemm_nt).  Returns the slice count (1 = no split) and what the caller
// must provide: `ws_floats` of workspace, `tiles` zero-initialised counters (the kernel leaves them zero).
int gemm_splitk_plan(const GemmArgs &g, size_t *ws_floats, size_t *tiles, int *tile) {
  *ws_floats = *tiles = 0;
  *tile = 0;
  static const int forced = getenv("XDTTS_GEMM_SPLITK") ? atoi(getenv("XDTTS_GEMM_SPLITK")) : -1;  // developer comparison aids (0 / 1: off)
  static const int forced_t = getenv("XDTTS_GEMM_SPLIT_TILE") ? atoi(getenv("XDTTS_GEMM_SPLIT_TILE")) : 0;
  if (forced == 0 || forced == 1) return 1;
  long rows = 0, t32 = 0, t64 = 0;
  for (int z = 0; z < g.batch; ++z) {
    const int m = g.ragged ? g.Mz[z] : g.M;
    rows += m;
    t32 += (long)((m + 31) / 32) * ((g.N + 31) / 32);
    t64 += (long)((m + 63) / 64) * ((g.N + 63) / 64);
  }
  const long tiles64 = (long)((g.N + 63) / 64) * ((g.M + 63) / 64) * g.batch;
  const bool big = g.N >= 64 && tiles64 >= 512 && !(g.M <= 128 && g.N >= 4096);
  if (big || rows * g.lda > (long)g.N * g.K) return 1;  // the chip is full without it / the row-tile-per-XCD mapping: batches, not split
  const int nslab = (g.K + 31) / 32;
  // Measured (profiles/r06_gemm_splitk.txt, configs[1]'s utterance): it pays where the tiles do not even give every CU a block -- the
  // encoder's M = 100 rows (128 blocks: 0.305 -> 0.277 ms per utterance at four slices) -- and not where they already do: the
  // post-net's three 512 -> 512 layers at F = 800 (416 blocks of 32x32) stay at 0.150-0.152 ms with 2 or 4 slices, and with 64x64
  // tiles x 4 / 6 / 8 slices (XDTTS_GEMM_SPLIT_TILE=64) they take 0.199 / 0.174 / 0.188 ms: that kernel is bound by the instructions it
  // issues per slab, not by how few waves it has.  Hence: 32x32 tiles, up to four slices, only while the grid stays within ~2 blocks per CU.
  const int tb = forced_t ? forced_t : 32;
  const long blocks = tb == 64 ? t64 : t32;
  int sk = forced > 1 ? forced : (int)std::min<long>(4, 512 / std::max<long>(blocks, 1));
  sk = std::min(sk, nslab / 8);  // at least 8 slabs per slice (the pipeline's prologue is 6 deep)
  if (sk < 2) return 1;
  const long grid_tiles = (long)((g.N + tb - 1) / tb) * ((g.M + tb - 1) / tb) * g.batch;
  *tile = tb;
  *tiles = (size_t)grid_tiles;
  *ws_floats = (size_t)grid_tiles * sk * tb * tb;
  return sk;
}

void launch_gemm_nt(const GemmArgs &g, hipStream_t s) {
  if (g.ragged && g.batch > GEMM_RAGGED_MAX) fail(XDTTS_ERR_BAD_ARG, "gemm: ragged batch of %d", g.batch);
  if (g.K % 16 != 0 || g.lda % 4 != 0) fail(XDTTS_ERR_BAD_ARG, "gemm: K=%d lda=%ld not supported", g.K, g.lda);
  // 64x64 tiles once they fill the chip at least twice over (XDTTS_GEMM_TILE=32|64: developer comparison aid)
  static const int forced = getenv("XDTTS_GEMM_TILE") ? atoi(getenv("XDTTS_GEMM_TILE")) : 0;
  const long tiles64 = (long)((g.N + 63) / 64) * ((g.M + 63) / 64) * g.batch;
  // (short and very wide -- the context fold of the persistent decoder, M = T = 100 rows x 8273 columns x K = 512 per chunk: its second
  // 64-row tile is 36 % full; measured 33.4 us with 32x32 tiles, 39.5 with 64x64, 44.4 with 32x64)
  const bool big = g.tile ? g.tile == 64 : (forced ? forced == 64 : (g.N >= 64 && tiles64 >= 512 && !(g.M <= 128 && g.N >= 4096)));
  // row tiles per XCD when A (unique bytes: rows x lda) outweighs W -- the batches; XDTTS_GEMM_XCD=0|1 forces it (comparison aid)
  static const int forced_x = getenv("XDTTS_GEMM_XCD") ? atoi(getenv("XDTTS_GEMM_XCD")) : -1;
  long rows = 0;
  for (int z = 0; z < g.batch; ++z) rows += g.ragged ? g.Mz[z] : g.M;
  GemmArgs a = g;
  a.xcd_rows = forced_x >= 0 ? forced_x : (rows * g.lda > (long)g.N * g.K ? 1 : 0);
  const int tb = big ? 64 : 32;
  const int nx = (g.N + tb - 1) / tb;
  int ny = (g.M + tb - 1) / tb, nz = g.batch;
  if (a.xcd_rows) {
    long tiles = 0;
    for (int z = 0; z < g.batch; ++z) tiles += ((g.ragged ? g.Mz[z] : g.M) + tb - 1) / tb;
    ny = (int)((tiles + 7) / 8 * 8);
    nz = 1;
  }
  if (a.splitk > 1 && (a.xcd_rows || !a.ws || !a.cnt)) a.splitk = 1;  // (gemm_splitk_plan never asks for it there)
  if (a.splitk > 1) nz *= a.splitk;
  if (big)
    hipLaunchKernelGGL((k_gemm_nt<64, 64>), dim3(nx, ny, nz), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_gemm_nt<32, 32>), dim3(nx, ny, nz), dim3(256), 0, s, a);
  HIP_CHECK(hipGetLastError());
}

void launch_embed(const int64_t *ids, const float *emb, float *xpad, int B, int T, int pad, hipStream_t s) {
  hipLaunchKernelGGL(k_embed, dim3(T, B), dim3(128), 0, s, ids, emb, xpad, T, pad);
  HIP_CHECK(hipGetLastError());
}

void launch_bilstm(const float *xproj, const float *whhT_fwd, const float *whhT_bwd, float *memory, int B,
                   int T, hipStream_t s) {
  const size_t lds = sizeof(float) * (ENC_H + 4 * ENC_H + (size_t)BL_LDS * 4 * ENC_H);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bilstm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_bilstm, dim3(2, B), dim3(1024), lds, s, xproj, whhT_fwd, whhT_bwd, memory, B, T);
  HIP_CHECK(hipGetLastError());
}

void launch_transpose(const float *in, float *out, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, s, in, out, rows, cols);
  HIP_CHECK(hipGetLastError());
}

}  // namespace xdtts
