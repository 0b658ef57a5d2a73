// decoder_persistent8.hip -- the Tacotron2 decoder loop (src/tacotron2/mod.rs:302-342) as ONE persistent, weight-stationary
// launch for lock-step batches of 3..8 chunks ("the batched / parallel sentences" of src/phonemes.rs:677-680 at the size a
// server with a handful of concurrent utterances has).
//
// Between the two engines that existed (decoder_persistent.hip: <= 2 chunks per launch, its per-chunk arithmetic on the VALU,
// 10.6 us per pair step; decoder.hip: two launches per lock-step iteration of up to 64 chunks, 26-30 us whatever the batch)
// a batch of 3..16 chunks paid either two pair launches one after the other or the whole latency chain of the big engine.
// This kernel keeps decoder_persistent.hip's skeleton -- 256 workgroups (x 256 threads here), one per CU; workgroup c owns the
// 16 gate rows of attention-LSTM units 4c..4c+3 and of decoder-LSTM units 4c..4c+3 with their weights in REGISTERS for the
// whole loop; the state crosses CUs through memory that every workgroup polls; roles per chunk on top of the LSTM slices -- and
// changes the three things that do not scale with the chunk count there:
//   * the LSTM pre-activations of ALL chunks are one v_mfma_f32_16x16x4_f32 stream per wave: the wave's 16 x (K/4) weight
//     slab of both LSTMs is the A operand (272 registers per lane: four waves per workgroup, one per SIMD, each with the whole
//     512-register file -- the weights mostly in AGPRs; as eight waves of 256 registers the kernel spilled 1.3 kB per lane
//     and ran 57 us per step), the chunks' state vectors in LDS in [k/4][NB chunk slots][4] order are the B operand (one
//     ds_read_b128 feeds four MFMAs; columns beyond NB of the tile carry don't-care values that nothing reads), the 16 x 16 D
//     tile holds unit u = lane / 16, chunk n = lane % 16, gates i,f,g,o in a lane's four registers -- so the four K-slices
//     meet in LDS and wave 0 does every cell update in registers.  272 MFMAs per wave and step (3.6 us of a SIMD's matrix
//     pipe) whatever the number of chunks; two instances, NB = 4 and NB = 8 chunk slots;
//   * the attention context crosses as a SIXTH edge (512 values per chunk, from the chunk's 8 attention workgroups) instead
//     of being folded into the encoder memory: the fold tables cost 20 registers or 16 kB of LDS per chunk.  In exchange the
//     partial energies only travel among a chunk's own 8 attention workgroups, which are the only ones that need the softmax.
//   * the four vectors that EVERY workgroup gathers (x, h_att, ctx, h_dec: 11 kB per chunk and step) cross as plain 4-byte values
//     in a WRITE-ONCE ring, one slab per step, filled with 0xFFFFFFFF (a NaN no arithmetic here produces) before the launch: a
//     value is its own arrival flag, a 16-byte load brings four of them (one producer's units), and no address is ever
//     written twice in a launch.  Measured on the edge alone (tools/ubench_allgather.hip, 256 workgroups, 1024 values per
//     chunk): 1.45 / 1.73 us per all-gather at 4 / 8 chunks against 2.5-3.2 / 4.4-4.8 us for {tag, value} granules.  x carries
//     the chunk's active bit in its sign (x >= 0: it leaves a ReLU).  The two narrow edges (partial energies among a chunk's
//     8 attention workgroups, mel rows among its 16 projection workgroups) stay data-tagged 8-byte granules.
// Per step:  x -> [attention LSTM] -> h_att -> [query, energies] -> e -> [softmax, context] -> ctx -> [decoder LSTM] -> h_dec
//            -> [projection rows] -> mel -> [stop rule, prenet] -> x(s+1)
// Only the columns of the newest vector are multiplied on the critical path (x: 16 MFMAs per wave, ctx -> decoder LSTM: 32); the
// others are accumulated while the next vector's producers are busy.  Every spin is bounded and watches a global error word.
#include <cstdio>
#include <cstdlib>

#include "p8_exchange.h"

namespace xdtts {

namespace {


constexpr int PT = 256, NW = PT / 64, NBMAX = 8, GSLOTS = P8_B_MAX, P_NCU = ATT_RNN / 4, TP = PERSIST_T_MAX;  // GSLOTS: chunk slots the granule arrays are laid out for (the 16-slot kernel's too)
constexpr int ATTN_CU = 8, PRE_CU = 16, EP_LD = TP, MEL_GL = 96, WPAD = TP + 32;
constexpr unsigned P_SPIN_LIMIT = 1u << 21;
#ifndef XDTTS_P8_ATTN_SCHED
#define XDTTS_P8_ATTN_SCHED 0
#endif
constexpr int DH_A = 8;  // column groups of the decoder LSTM's h_att segment ahead of the first energies poll (schedule 0)
// 1: the attention workgroups put nothing but a nap between their energies and the poll of the other seven's, and run their
// h_att MFMAs behind the context / the decoder cell instead.  Measured equal (14.6-14.9 us at 4 chunks either way: what the
// energies edge gains, the later h_dec of those 32 workgroups loses for everybody), so the simpler order is the default.
constexpr bool ATTN_SCHED = XDTTS_P8_ATTN_SCHED != 0;
static_assert(NBMAX == 8 && ATT_RNN == DEC_RNN && P_NCU == 256 && (ATTN_CU + PRE_CU) * NBMAX <= P_NCU, "role workgroups of 8 chunks fit the grid");
static_assert(TP == 128 && PRENET == PT && EMB == 2 * PT && ATT_RNN == 4 * PT, "thread <-> granule maps below");

// chunk slots of the kernel instance that serves a batch of B
__host__ __device__ constexpr int p8_slots(int B) { return B <= 4 ? 4 : (B <= NBMAX ? NBMAX : P8_B_MAX); }


// NQ b128 loads of the wave's slice of one state segment (seg: LDS base of the segment in B order [k/4][NB][4]; q0: the wave's
// first column quad-of-quads), 4 MFMAs each, into acc.  A[q] = the lane's four weights of columns 16 (q0 + q) + 4 kk .. + 3.
#ifndef XDTTS_P8_TWO_CHAINS
#define XDTTS_P8_TWO_CHAINS 0
#endif
template <int NB, int NQ, int FROM = 0, int TO = NQ>
__device__ __forceinline__ void mfma_segment(f32x4 &acc, const float4 (&A)[NQ], const float *seg, int q0, int kk, int n) {
  // one B vector ahead of the MFMAs that consume it, and no further
  if constexpr (FROM >= TO) return;
  const float *bp = seg + ((4 * q0 + kk) * NB + n) * 4;
  float4 b = lds4(bp + FROM * 4 * NB * 4);
#if XDTTS_P8_TWO_CHAINS
  // two accumulator chains (x / z and y / w components): a dependent v_mfma_f32_16x16x4_f32 issues 40 cycles after its predecessor, an
  // independent one 32 (round 6, from the 16-slot kernel).  NOT the default: this kernel sits at 256 + 253 registers, the second chain's
  // four push it into scratch (36 spill instructions) -- 16.3 / 15.6 / 18.0 / 18.2 us at 3 / 4 / 5 / 8 chunks against 14.8 / 14.7 / 15.9 / 17.0
  f32x4 acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = FROM; q < TO; ++q) {
    const float4 bn = q + 1 < TO ? lds4(bp + (q + 1) * 4 * NB * 4) : b;
    asm volatile("" ::: "memory");
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].x, b.x, acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].y, b.y, acc1, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].z, b.z, acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].w, b.w, acc1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    b = bn;
  }
  acc += acc1;
#else
#pragma unroll
  for (int q = FROM; q < TO; ++q) {
    const float4 bn = q + 1 < TO ? lds4(bp + (q + 1) * 4 * NB * 4) : b;
    asm volatile("" ::: "memory");
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].w, b.w, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    b = bn;
  }
#endif
}

// Developer build (-DXDTTS_P8_PROFILE): thread 0 of three workgroups (one per role) accumulates the 100 MHz wall clock between
// phase markers and prints the sums at exit.
#ifdef XDTTS_P8_PROFILE
#define P8_MARK(i)                                              \
  do {                                                          \
    s_ts[wave * 32 + (i)] += (unsigned)wall_clock64(); /* every lane, no branch: wave 0's copy is read */ \
  } while (0)
#else
#define P8_MARK(i) do { } while (0)
#endif

// NB = chunk slots compiled in (4 or 8): the width of the state vectors in LDS and of every gather
template <int NB>
__global__ __launch_bounds__(PT) void k_decoder_persistent8(DecoderBufs d, P8Bufs g, P8Weights w, int nsteps) {
  // LDS (148 of 160 kB at NB = 8).  The state vectors of all chunks in MFMA B-operand order; x and ctx share a buffer and so do
  // h_att and h_dec: each is consumed (by the MFMAs that follow its gather) before the other is gathered, barriers in between.
  constexpr int ATTN_FLOATS = 2 * TP * 16 + 3 * TP + 16 + PT + 2 * WPAD + 62 * 16 + 16 + 16 * PT * 4;
  constexpr int PRE_FLOATS = N_MEL * PRENET + MEL_GL + 8;
  __shared__ __attribute__((aligned(16))) float s_xc[EMB * NB], s_h[ATT_RNN * NB];
  __shared__ __attribute__((aligned(16))) float s_acc[NW * 64 * 4];  // the four K-slices' partial D tiles
  __shared__ __attribute__((aligned(16))) float s_hrow[ATT_RNN];     // the role's own chunk, row-major: h_att (attention) / h_dec (projection)
  __shared__ __attribute__((aligned(16))) float s_crow[EMB];         // projection role: ctx of its chunk, row-major
  __shared__ __attribute__((aligned(16))) float s_role[ATTN_FLOATS > PRE_FLOATS ? ATTN_FLOATS : PRE_FLOATS];
  __shared__ int s_act[NB], s_alive[NB], s_err;
  float *const s_x = s_xc, *const s_ctx = s_xc, *const s_hatt = s_h, *const s_hdec = s_h;
  // attention role
  float *s_pm = s_role, *s_loc = s_pm + TP * 16, *s_aw = s_loc + TP * 16, *s_awc = s_aw + TP, *s_e = s_awc + TP, *s_q = s_e + TP,
        *s_part = s_q + 16, *s_wpad = s_part + PT, *s_G = s_wpad + 2 * WPAD, *s_vv = s_G + 62 * 16,
        *s_qw = s_vv + 16;  // [16][PT] float4: query rows 16 rk + wave + 4 r, 4 x 16 B per lane each
  // projection + prenet role
  float *s_W0 = s_role, *s_mel = s_W0 + N_MEL * PRENET, *s_pb = s_mel + MEL_GL;  // s_W0 [20][256][4]

  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = d.B, T = d.T;
  const PollCtl pc{g.err, g.spins > 0 ? (unsigned)g.spins : P_SPIN_LIMIT};
  if (g.fault && c == g.fault - 1) return;  // test hook: this workgroup never shows up
  const int kk = lane >> 4, n16 = lane & 15, n = n16 & (NB - 1);  // MFMA lane coordinates: k-quad / chunk column
  // index of element (k, chunk b) of a state vector kept in MFMA B-operand order [k/4][NB][4]
  auto bidx = [](int k, int b) { return ((k >> 2) * NB + b) * 4 + (k & 3); };

  // wave 0 finalises both cells: lane = (unit u = lane / 16, chunk n16); its four registers of a D tile are the gates i,f,g,o
  const int cu = lane >> 4;
  float4 bias_a = make_float4(0.f, 0.f, 0.f, 0.f), bias_d = bias_a;
  float c_att = 0.f, c_dec = 0.f, h_att_last = 0.f, h_dec_last = 0.f;
  const bool cell = wave == 0 && n16 < B;
  if (wave == 0) {
    bias_a = *reinterpret_cast<const float4 *>(w.att_b + 16 * c + 4 * cu);
    bias_d = *reinterpret_cast<const float4 *>(w.dec_b + 16 * c + 4 * cu);
    if (cell) {
      c_att = d.att_c[n16 * ATT_RNN + 4 * c + cu];
      c_dec = d.dec_c[n16 * DEC_RNN + 4 * c + cu];
      h_att_last = d.att_h[0][n16 * ATT_RNN + 4 * c + cu];
      h_dec_last = d.dec_h[0][n16 * DEC_RNN + 4 * c + cu];
    }
  }

  // ---- roles ---------------------------------------------------------------------------------------------------------------
  const bool attn = c < ATTN_CU * B, pre = !attn && c < (ATTN_CU + PRE_CU) * B;
  const int rb = attn ? c / ATTN_CU : (pre ? (c - ATTN_CU * B) / PRE_CU : 0);
  const int rk = attn ? c % ATTN_CU : (c - ATTN_CU * B) % PRE_CU;
  const int step0 = d.ctl[0];
  if (tid < NB) {
    s_act[tid] = 0;
    s_alive[tid] = tid < B && step0 < d.nframes[tid];
  }
  if (tid == 0) s_err = 0;

  // state of the sequence so far (zeros at step 0; a previous launch's write-back otherwise), chunks beyond B zero: h_att and
  // ctx first (the attention LSTM's partial), h_dec behind the barrier below
#pragma unroll 1
  for (int i = tid; i < ATT_RNN * NB; i += PT) {
    const int k = i / NB, b = i & (NB - 1);
    s_hatt[bidx(k, b)] = b < B ? d.att_h[0][b * ATT_RNN + k] : 0.f;
  }
#pragma unroll 1
  for (int i = tid; i < EMB * NB; i += PT) {
    const int k = i / NB, b = i & (NB - 1);
    s_ctx[bidx(k, b)] = b < B ? d.ctx[b * EMB + k] : 0.f;
  }
  // role registers (one array, two uses; the roles are disjoint workgroups):
  //   attention role [0, 32): memory[t = wave + 4 j][64 rk + lane]: the steps this wave sums into the context column `lane`
  //   projection + prenet role [0, 16): layer-2 columns 16 rk + wave + 4 r, inputs lane + 64 k (at [4 r + k]);
  //     [16, 64): rows rk + 16 (wave + 4 r) of [W_p ; w_gate], r < 2, 24 weights per lane each
  float rreg[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) rreg[j] = 0.f;
  if (attn) {
#pragma unroll 1
    for (int i = tid; i < TP * 16; i += PT) {
      const int t = i >> 4, dd_ = i & 15;
      s_pm[i] = t < T ? d.pmem[((size_t)rb * T + t) * ATT_DIM + 16 * rk + dd_] : 0.f;
    }
    if (tid < TP) {
      s_aw[tid] = tid < T ? d.aw[rb * T + tid] : 0.f;
      s_awc[tid] = tid < T ? d.awc[rb * T + tid] : 0.f;
    }
#pragma unroll 1
    for (int i = tid; i < 62 * 16; i += PT) s_G[i] = w.loc_fused[(size_t)(i >> 4) * ATT_DIM + 16 * rk + (i & 15)];
    if (tid < 16) s_vv[tid] = w.v_w[16 * rk + tid];
#pragma unroll
    for (int j = 0; j < TP / NW; ++j) {
      const int t = wave + NW * j;
      rreg[j] = t < T ? d.memory[((size_t)rb * T + t) * EMB + 64 * rk + lane] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4 *>(s_qw + 4 * ((4 * r + j) * PT + tid)) = w.q_w[(unsigned)((16 * rk + wave + NW * r) * (ATT_RNN / 4) + lane + 64 * j)];
  }
  if (pre) {
    // layer 1 [in / 4][out][in % 4]: a thread's 80 weights are twenty conflict-free 16-byte reads
#pragma unroll 1
    for (int i = tid; i < N_MEL * PRENET; i += PT) s_W0[(((i / PRENET) >> 2) * PRENET + i % PRENET) * 4 + ((i / PRENET) & 3)] = w.pre0T[i];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) rreg[4 * r + k] = w.pre1T[(unsigned)((lane + 64 * k) * PRENET + 16 * rk + wave + NW * r)];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int prow = rk + 16 * (wave + NW * r);
      if (prow <= N_MEL) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float4 q = w.proj_w[(unsigned)(prow * (PROJ_IN / 4) + lane + 64 * j)];
          rreg[16 + 24 * r + 4 * j + 0] = q.x;
          rreg[16 + 24 * r + 4 * j + 1] = q.y;
          rreg[16 + 24 * r + 4 * j + 2] = q.z;
          rreg[16 + 24 * r + 4 * j + 3] = q.w;
        }
      }
    }
  }
  int nf_r = pre ? d.nframes[rb] : 0;
  const int nv_r = attn ? d.n_valid[rb] : 0;
  bool ctx_valid = false;
  __syncthreads();
  if (pre && tid < 8) {  // (behind the s_W0 fill: s_pb follows it in the role area)
    const int prow = rk + 16 * tid;
    s_pb[tid] = prow <= N_MEL ? w.proj_b[prow] : 0.f;
  }
  const uint32_t item = d.item_base + (uint32_t)rb;

  // location features of the NEXT step for the attention role's 16 dims (decoder_persistent.hip: a Toeplitz product on the matrix cores)
  auto location = [&]() {
#pragma unroll 1
    for (int i = tid; i < 2 * WPAD; i += PT) {
      const int ch = i / WPAD, t = i % WPAD - (LOC_K - 1) / 2;
      s_wpad[i] = (t >= 0 && t < T) ? (ch ? s_awc[t] : s_aw[t]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < TP / (16 * NW); ++hh) {
      const unsigned l = (unsigned)lane, li = l & 15u, lg = l >> 4, t0 = 16u * (unsigned)(wave + NW * hh);
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int k2 = 0; k2 < 16; k2 += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned q = 4u * (k2 + h) + lg, qa = q < 2u * LOC_K ? q : 2u * LOC_K - 1u, ch = qa >= (unsigned)LOC_K ? 1u : 0u;
          const float av = s_wpad[ch * WPAD + t0 + li + (qa - ch * LOC_K)];
          const float bv = q < 2u * LOC_K ? s_G[qa * 16u + li] : 0.f;
          if (h == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
          else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) s_loc[(t0 + 4u * lg + j) * 16u + li] = acc0[j] + acc1[j];
    }
    __syncthreads();
  };
  if (attn) location();

  // ---- resident weights: the wave's K-slice of the workgroup's 16 + 16 gate rows, as MFMA A operands -----------------------
  // row 16c + i (packed [unit][gate] order) is unit 4c + i/4, gate i%4; A lane = (row i = lane % 16, k-quad kk = lane / 16)
  float4 ax[4], ac[8], ah[16];   // attention LSTM: x 256 | ctx 512 | h_att 1024 columns, 1/4 of each
  float4 dh[16], dc[8], dd[16];  // decoder LSTM:   h_att 1024 | ctx 512 | h_dec 1024
  {
    const float4 *ra = w.att_w + (size_t)(16 * c + n16) * (ATT_COLS / 4), *rd = w.dec_w + (size_t)(16 * c + n16) * (DEC_COLS / 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) ax[q] = ld_stream(ra + (0 + 64 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 8; ++q) ac[q] = ld_stream(ra + (PRENET + 128 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 16; ++q) ah[q] = ld_stream(ra + (ATT_IN + 256 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 16; ++q) dh[q] = ld_stream(rd + (0 + 256 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 8; ++q) dc[q] = ld_stream(rd + (ATT_RNN + 128 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 16; ++q) dd[q] = ld_stream(rd + (DEC_IN + 256 * wave) / 4 + 4 * q + kk);
  }
  // the partial pre-activations that do not depend on the newest vector
  f32x4 accA = (f32x4){0.f, 0.f, 0.f, 0.f}, accD = accA;
  mfma_segment<NB, 8>(accA, ac, s_ctx, 8 * wave, kk, n);     // attention LSTM: ctx(s-1) ...
  mfma_segment<NB, 16>(accA, ah, s_hatt, 16 * wave, kk, n);  // ... and h_att(s-1)
  __syncthreads();
#pragma unroll 1
  for (int i = tid; i < DEC_RNN * NB; i += PT) {
    const int k = i / NB, b = i & (NB - 1);
    s_hdec[bidx(k, b)] = b < B ? d.dec_h[0][b * DEC_RNN + k] : 0.f;
  }
  __syncthreads();
  mfma_segment<NB, 16>(accD, dd, s_hdec, 16 * wave, kk, n);  // decoder LSTM: h_dec(s-1)
  __syncthreads();  // (x(s) is gathered into the buffer ctx(s-1) was read from)

  // wave 0: sum of the K-slices of a D tile (the caller has put a barrier behind the s_acc stores)
  auto reduce_tile = [&]() {
    f32x4 gsum = *reinterpret_cast<const f32x4 *>(s_acc + lane * 4);
#pragma unroll
    for (int q = 1; q < NW; ++q) gsum += *reinterpret_cast<const f32x4 *>(s_acc + (q * 64 + lane) * 4);
    return gsum;
  };

#ifdef XDTTS_P8_PROFILE
  __shared__ u64 s_prof[32];
  __shared__ unsigned s_ts[NW * 32];  // time stamp of every marker of the current step, in program order P8_ORDER
  if (tid < 32) s_prof[tid] = 0;
#endif
#ifdef XDTTS_P8_PROFILE
  __syncthreads();
  if (tid < NW * 32) s_ts[tid] = 0;
  __syncthreads();
  s_ts[wave * 32 + 28] = (unsigned)wall_clock64();  // start of the loop
#endif
  int s = step0;
  const int s_stop = step0 + nsteps;
  float cown = 0.f;  // attention role, tid < 64: the chunk's context column 64 rk + tid of the last step (write-back)
  for (; s < s_stop; ++s) {
    const int p = s & 1;
    const unsigned want = (unsigned)(s + 1);
    // ---- P1: x(s) and the chunks' active bits ---------------------------------------------------------------------------------
    {
      unsigned need = 0u;
#pragma unroll
      for (int b = 0; b < NB; ++b) need |= (s_alive[b] != 0 ? 1u : 0u) << b;
      // quad tid + 256 i of the slab: chunk 4 i + wave, columns 4 lane .. + 3; the sign bit of a column = chunk not active
      unsigned needx = 0u;
#pragma unroll
      for (int i = 0; i < (NB + 3) / 4; ++i) needx |= ((need >> (4 * i + wave)) & 1u) << i;
      nap(g.delay[3]);
      gather16<(NB + 3) / 4>(g.rx + (size_t)(s - step0) * NB * PRENET, needx, pc, [&](int i) { return 16u * (unsigned)(tid + PT * i); },
                             [&](int i, u32x4 v) {
                               const int b = 4 * i + wave;
                               if (lane == 0) s_act[b] = (v.x >> 31) ? 0 : 1;
                               v &= 0x7fffffffu;
                               *reinterpret_cast<float4 *>(s_x + bidx(4 * lane, b)) = as_f4(v);
                             });
      if (tid == PT - 1) s_err = __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    P8_MARK(0);
    __syncthreads();
    P8_MARK(1);
    unsigned actm = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) actm |= (s_act[b] != 0 ? 1u : 0u) << b;
    if (!actm || s_err != 0) break;  // every chunk has stopped (or an exchange failed): the launch ends by itself
    const bool act_r = (actm >> rb) & 1u;
    const bool act_n = (actm >> n16) & 1u;  // (lanes n16 >= NB: false)
    // attention LSTM: close the rows with the x columns
    mfma_segment<NB, 4>(accA, ax, s_x, 4 * wave, kk, n);
    *reinterpret_cast<f32x4 *>(s_acc + (wave * 64 + lane) * 4) = accA;
    __syncthreads();
    if (tid < NB) s_alive[tid] = s_act[tid];  // (read again only at the next P1, behind this step's barriers)
    if (wave == 0) {
      const f32x4 gs = reduce_tile();
      if (cell && act_n) {
        const float ig = fast_sigmoid(gs[0] + bias_a.x), fg = fast_sigmoid(gs[1] + bias_a.y), gg = fast_tanh(gs[2] + bias_a.z),
                    og = fast_sigmoid(gs[3] + bias_a.w);
        c_att = fmaf(fg, c_att, ig * gg);
        h_att_last = og * fast_tanh(c_att);
        put(g.rhatt + ((size_t)(s - step0) * NB + n16) * ATT_RNN + 4 * c + cu, value_bits(h_att_last));
      }
    }
    accA = (f32x4){0.f, 0.f, 0.f, 0.f};
    P8_MARK(2);
    // ---- P2: h_att(s) of every active chunk ------------------------------------------------------------------------------------
    // attention role: what the energies need besides the query is in registers before h_att arrives; thread = (encoder steps
    // t = tid / 4 and t + 64, dims 4 (tid % 4) .. + 3 of the workgroup's 16)
    float4 lp0 = make_float4(0.f, 0.f, 0.f, 0.f), lp1 = lp0, v4 = lp0;
    if (attn && act_r) {
      const float4 l0 = lds4(s_loc + 4 * tid), p0 = lds4(s_pm + 4 * tid), l1 = lds4(s_loc + 4 * (tid + PT)), p1 = lds4(s_pm + 4 * (tid + PT));
      lp0 = make_float4(l0.x + p0.x, l0.y + p0.y, l0.z + p0.z, l0.w + p0.w);
      lp1 = make_float4(l1.x + p1.x, l1.y + p1.y, l1.z + p1.z, l1.w + p1.w);
      v4 = lds4(s_vv + 4 * (tid & 3));
    }
    // quad tid of chunk i's slab row: units 4 tid .. + 3 (workgroup tid's)
    nap(g.delay[0]);
    const unsigned fr0 = gather16<NB>(g.rhatt + (size_t)(s - step0) * NB * ATT_RNN, actm, pc, [&](int i) { return 16u * (unsigned)(tid + PT * i); },
                                      [&](int i, u32x4 v) {
                                        *reinterpret_cast<float4 *>(s_hatt + bidx(4 * tid, i)) = as_f4(v);
                                        if (attn && i == rb) *reinterpret_cast<float4 *>(s_hrow + 4 * tid) = as_f4(v);
                                      });
    P8_MARK(3);
    __syncthreads();
    P8_MARK(4);
    if (attn && act_r) {
      // query rows 16 rk + wave + 4 r, then this workgroup's share of the energies
      float qv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) qv[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 hv = lds4(s_hrow + 256 * j + 4 * lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) qv[r] = dot4(lds4(s_qw + 4 * ((4 * r + j) * PT + tid)), hv, qv[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        qv[r] = wave_sum(qv[r]);
        if (lane == 0) s_q[wave + NW * r] = qv[r];
      }
      __syncthreads();
      const int t = tid >> 2, dq = 4 * (tid & 3);
      const float4 q4 = lds4(s_q + dq);
      float e0 = v4.x * fast_tanh(q4.x + lp0.x), e1 = v4.x * fast_tanh(q4.x + lp1.x);
      e0 = fmaf(v4.y, fast_tanh(q4.y + lp0.y), e0);
      e1 = fmaf(v4.y, fast_tanh(q4.y + lp1.y), e1);
      e0 = fmaf(v4.z, fast_tanh(q4.z + lp0.z), e0);
      e1 = fmaf(v4.z, fast_tanh(q4.z + lp1.z), e1);
      e0 = fmaf(v4.w, fast_tanh(q4.w + lp0.w), e0);
      e1 = fmaf(v4.w, fast_tanh(q4.w + lp1.w), e1);
      e0 += dpp_move<0xB1, 0xf>(0.f, e0);  // quad_perm:[1,0,3,2]
      e1 += dpp_move<0xB1, 0xf>(0.f, e1);
      e0 += dpp_move<0x4E, 0xf>(0.f, e0);  // quad_perm:[2,3,0,1]
      e1 += dpp_move<0x4E, 0xf>(0.f, e1);
      if ((tid & 3) == 0) {
        u64 *row = g.ep + (unsigned)(((p * NB + rb) * ATTN_CU + rk) * EP_LD);
        if (t < T) publish(row + t, want, e0);
        if (t + 64 < T) publish(row + t + 64, want, e1);
      }
    }
    P8_MARK(5);
    // decoder LSTM of this step: the h_att(s) columns (in the time the partial energies travel)
    // (attention role: the first poll of the partial energies leaves half-way through, so that it lands as the MFMAs end)
    const int ep_t = tid >> 2, ep_j = tid & 3;  // loads i: row j + 4 (i % 2), encoder step t + 64 (i / 2)
    const u64 *ep_base = g.ep + (unsigned)(((p * NB + rb) * ATTN_CU + ep_j) * EP_LD + ep_t);
    const unsigned ep_need = (attn && act_r) ? ((ep_t < T ? 3u : 0u) | (ep_t + 64 < T ? 12u : 0u)) : 0u;
    auto ep_at = [](int i) { return (unsigned)((i & 1) * 4 * EP_LD + (i >> 1) * 64); };
    u64 ep_v[4] = {0, 0, 0, 0};
    const bool attn_on = attn && act_r;
    if (ATTN_SCHED && attn_on) {
      // the attention workgroups are the step's critical chain here: nothing but a nap between their energies and the poll of
      // the chunk's other seven; their h_att MFMAs follow the context (and the next step's follow the decoder cell)
      nap(g.delay[4]);
      gather_issue<4>(ep_v, ep_base, ep_need, ep_at);
    } else {
      mfma_segment<NB, 16, 0, DH_A>(accD, dh, s_hatt, 16 * wave, kk, n);
      if (ep_need) gather_issue<4>(ep_v, ep_base, ep_need, ep_at);
      mfma_segment<NB, 16, DH_A, 16>(accD, dh, s_hatt, 16 * wave, kk, n);
      if (ATTN_SCHED) mfma_segment<NB, 16>(accA, ah, s_hatt, 16 * wave, kk, n);  // attention LSTM of the next step: h_att(s)
    }
    P8_MARK(6);
    // ---- P3 (attention role): the 8 partial-energy rows of the chunk -> softmax -> this workgroup's 64 context columns ------------
    if (attn && act_r) {
      {
        const int t = ep_t, j = ep_j;
        float ev[4] = {0.f, 0.f, 0.f, 0.f};
        gather_from<4>(ep_v, ep_base, want, ep_need, pc, ep_at, [&](int i, float v, unsigned) { ev[i] = v; });
        float e0 = ev[0] + ev[1], e1 = ev[2] + ev[3];
        e0 += dpp_move<0xB1, 0xf>(0.f, e0);
        e1 += dpp_move<0xB1, 0xf>(0.f, e1);
        e0 += dpp_move<0x4E, 0xf>(0.f, e0);
        e1 += dpp_move<0x4E, 0xf>(0.f, e1);
        if (j == 0) {  // mask, mod.rs:219-220
          s_e[t] = (t < T && t < nv_r) ? e0 : -INFINITY;
          s_e[t + 64] = (t + 64 < T && t + 64 < nv_r) ? e1 : -INFINITY;
        }
      }
      P8_MARK(20);
      __syncthreads();
      P8_MARK(21);
      {  // every wave: the softmax in registers, lane <-> steps lane, lane + 64
        const float e0 = s_e[lane], e1 = s_e[lane + 64];
        const float m = wave_max(fmaxf(e0, e1));
        const float x0 = fast_exp(e0 - m), x1 = fast_exp(e1 - m);
        const float rs = __builtin_amdgcn_rcpf(wave_sum(x0 + x1));
        const float w0 = x0 * rs, w1 = x1 * rs;
        // context columns: this wave sums the steps t = wave + 4 j; lane = column.  The weight of step t sits in lane t % 64
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < TP / NW; ++j) {
          const int t = wave + NW * j;
          const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(j < 64 / NW ? w0 : w1), t & 63));
          acc = fmaf(wt, rreg[j], acc);
        }
        s_part[tid] = acc;
        if (wave == 0) {  // kept for the next step's location features
          s_aw[lane] = w0;
          s_awc[lane] += w0;
          s_aw[lane + 64] = w1;
          s_awc[lane + 64] += w1;
        }
      }
      P8_MARK(22);
      __syncthreads();
      if (tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < NW; ++u) v += s_part[u * 64 + tid];
        cown = v;
        ctx_valid = true;
        put(g.rctx + ((size_t)(s - step0) * NB + rb) * EMB + 64 * rk + tid, value_bits(v));
      }
    }
    P8_MARK(23);
    if (!ATTN_SCHED) mfma_segment<NB, 16>(accA, ah, s_hatt, 16 * wave, kk, n);  // attention LSTM of the next step: h_att(s) (while ctx travels)
    else if (attn_on) mfma_segment<NB, 16>(accD, dh, s_hatt, 16 * wave, kk, n);
    P8_MARK(7);
    // ---- P4: ctx(s) of every active chunk -> decoder LSTM ------------------------------------------------------------------------
    // quad tid + 256 i of the slab: chunk 2 i + tid / 128, columns 4 (tid % 128) .. + 3
    unsigned needc = 0u;
#pragma unroll
    for (int i = 0; i < NB / 2; ++i) needc |= ((actm >> (2 * i + (wave >> 1))) & 1u) << i;
    if (!(attn && act_r)) nap(g.delay[1]);  // (the attention workgroups are the producers: the others have 3 us to wait)
    const unsigned fr1 = gather16<NB / 2>(g.rctx + (size_t)(s - step0) * NB * EMB, needc, pc, [&](int i) { return 16u * (unsigned)(tid + PT * i); },
                                          [&](int i, u32x4 v) {
                                            const int b = 2 * i + (wave >> 1), k = 4 * (tid & 127);
                                            *reinterpret_cast<float4 *>(s_ctx + bidx(k, b)) = as_f4(v);
                                            if (pre && b == rb) *reinterpret_cast<float4 *>(s_crow + k) = as_f4(v);
                                          });
    P8_MARK(8);
    __syncthreads();
    P8_MARK(9);
    mfma_segment<NB, 8>(accD, dc, s_ctx, 8 * wave, kk, n);
    *reinterpret_cast<f32x4 *>(s_acc + (wave * 64 + lane) * 4) = accD;
    __syncthreads();
    if (wave == 0) {
      const f32x4 gs = reduce_tile();
      if (cell && act_n) {
        const float ig = fast_sigmoid(gs[0] + bias_d.x), fg = fast_sigmoid(gs[1] + bias_d.y), gg = fast_tanh(gs[2] + bias_d.z),
                    og = fast_sigmoid(gs[3] + bias_d.w);
        c_dec = fmaf(fg, c_dec, ig * gg);
        h_dec_last = og * fast_tanh(c_dec);
        put(g.rhdec + ((size_t)(s - step0) * NB + n16) * DEC_RNN + 4 * c + cu, value_bits(h_dec_last));
      }
    }
    accD = (f32x4){0.f, 0.f, 0.f, 0.f};
    P8_MARK(10);
    if (ATTN_SCHED && attn_on) mfma_segment<NB, 16>(accA, ah, s_hatt, 16 * wave, kk, n);
    mfma_segment<NB, 8>(accA, ac, s_ctx, 8 * wave, kk, n);  // attention LSTM of the next step: ctx(s)
    if (attn && act_r) location();                           // ... and its location features
    P8_MARK(11);
    // ---- P5: h_dec(s) -> projection rows ---------------------------------------------------------------------------------------
    // projection + prenet role: the Bernoulli(0.5) masks of step s + 1 (they do not depend on the data) are hashed in the time
    // the first poll of h_dec could not succeed anyway
    bool drop1 = false;
    unsigned drop2 = 0u;
    if (pre && act_r && d.dropout_mode) {
      drop1 = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 0, tid);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        drop2 |= (prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 1, 16 * rk + wave + NW * r) ? 1u : 0u) << r;
      drop2 |= drop1 ? 16u : 0u;
      asm volatile("" : "+v"(drop2));  // (computed HERE: left alone the compiler sinks the hashes behind the gather, onto the critical path)
      drop1 = (drop2 & 16u) != 0u;
    }
    nap(g.delay[2]);
    const unsigned fr2 = gather16<NB>(g.rhdec + (size_t)(s - step0) * NB * DEC_RNN, actm, pc, [&](int i) { return 16u * (unsigned)(tid + PT * i); },
                                      [&](int i, u32x4 v) {
                                        *reinterpret_cast<float4 *>(s_hdec + bidx(4 * tid, i)) = as_f4(v);
                                        if (pre && i == rb) *reinterpret_cast<float4 *>(s_hrow + 4 * tid) = as_f4(v);
                                      });
    P8_MARK(12);
    __syncthreads();
    P8_MARK(13);
    if (pre && act_r) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int prow = rk + 16 * (wave + NW * r);
        if (prow <= N_MEL) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            a = dot4(make_float4(rreg[16 + 24 * r + 4 * j], rreg[17 + 24 * r + 4 * j], rreg[18 + 24 * r + 4 * j], rreg[19 + 24 * r + 4 * j]),
                     lds4(s_hrow + 256 * j + 4 * lane), a);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            a = dot4(make_float4(rreg[32 + 24 * r + 4 * j], rreg[33 + 24 * r + 4 * j], rreg[34 + 24 * r + 4 * j], rreg[35 + 24 * r + 4 * j]),
                     lds4(s_crow + 256 * j + 4 * lane), a);
          a = wave_sum(a);
          if (lane == 0) publish(g.mel + (unsigned)((p * NB + rb) * MEL_GL + prow), want, a + s_pb[wave + NW * r]);
        }
      }
    }
    P8_MARK(14);
    mfma_segment<NB, 16>(accD, dd, s_hdec, 16 * wave, kk, n);  // decoder LSTM of the next step: h_dec(s)
    P8_MARK(15);
    // ---- P6 (projection + prenet role): frame s, stop rule, x(s+1) ---------------------------------------------------------------
    if (pre && act_r) {  // a chunk's last x (active bit clear) is published at the step it stops
      if (tid < N_MEL + 1) {
        s_mel[tid] = 0.f;
        gather<1>(g.mel + (unsigned)((p * NB + rb) * MEL_GL + tid), want, 1u, pc, [](int) { return 0u; }, [&](int, float v, unsigned) { s_mel[tid] = v; });
      }
      P8_MARK(24);
      __syncthreads();
      P8_MARK(25);
      const float gate = s_mel[N_MEL];
      const bool fired = d.use_gate && gate_fires(gate, d.gate_lo, d.gate_hi, d.gate_threshold);  // mod.rs:319-324
      if (rk == 0) {
        if (tid < N_MEL) d.frames[((size_t)rb * d.max_steps + s) * N_MEL + tid] = s_mel[tid];
        if (tid == 0) {
          d.gates[(size_t)rb * d.max_steps + s] = gate;
          if (fired) d.nframes[rb] = s + 1;  // the tripping frame is kept
        }
      }
      if (fired) nf_r = s + 1;
      const bool nxt = s + 1 < nf_r;
      float xo[4] = {0.f, 0.f, 0.f, 0.f};
      if (nxt) {
        float acc = 0.f;  // layer 1, output tid
#pragma unroll
        for (int k = 0; k < N_MEL; k += 4) {
          const float4 w4 = lds4(s_W0 + 4u * ((unsigned)(k >> 2) * PRENET + (unsigned)tid)), m = lds4(s_mel + k);
          acc = fmaf(w4.x, m.x, acc);
          acc = fmaf(w4.y, m.y, acc);
          acc = fmaf(w4.z, m.z, acc);
          acc = fmaf(w4.w, m.w, acc);
        }
        acc = fmaxf(acc, 0.f);
        P8_MARK(26);
        __syncthreads();  // (s_mel read by everyone before s_hrow, free since the projection, takes the layer-1 outputs)
        s_hrow[tid] = drop1 ? 0.f : (d.dropout_mode ? 2.f * acc : acc);
        __syncthreads();
        float pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = s_hrow[lane + 64 * k];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) a = fmaf(rreg[4 * r + k], pk[k], a);
          a = fmaxf(wave_sum(a), 0.f);
          xo[r] = (drop2 >> r) & 1u ? 0.f : (d.dropout_mode ? 2.f * a : a);
        }
      } else {
        P8_MARK(26);  // (profile build: every marker of the role once per step)
      }
      P8_MARK(27);
      // the workgroup's 16 columns leave as ONE 128-byte store (decoder_persistent.hip)
      if (lane < 4) s_mel[MEL_GL - 16 + wave + NW * lane] = lane == 0 ? xo[0] : (lane == 1 ? xo[1] : (lane == 2 ? xo[2] : xo[3]));  // s_mel[81..95] is unused padding
      __syncthreads();
      if (tid < 16)
        put(g.rx + ((size_t)(s + 1 - step0) * NB + rb) * PRENET + 16 * rk + tid, (value_bits(s_mel[MEL_GL - 16 + tid]) & 0x7fffffffu) | (nxt ? 0u : 0x80000000u));  // (x >= 0; a -0.0 must not read as "stopped")
    }
    P8_MARK(16);
#ifdef XDTTS_P8_PROFILE
    s_ts[wave * 32 + 29] = (unsigned)wall_clock64();  // (the last marker of the last step, not summed)
    if (tid == 0) {
      s_prof[17] += fr0;
      s_prof[18] += fr1;
      s_prof[19] += fr2;
    }
#else
    (void)fr0, (void)fr1, (void)fr2;
#endif
  }

  // ---- write the state back (a later launch, or the parity hook, may continue the sequence) ----------------------------------------
  if (cell) {
    d.att_c[n16 * ATT_RNN + 4 * c + cu] = c_att;
    d.dec_c[n16 * DEC_RNN + 4 * c + cu] = c_dec;
    d.att_h[0][n16 * ATT_RNN + 4 * c + cu] = h_att_last;
    d.dec_h[0][n16 * DEC_RNN + 4 * c + cu] = h_dec_last;
  }
  if (attn) {
    if (tid < 64 && ctx_valid) d.ctx[rb * EMB + 64 * rk + tid] = cown;  // (the chunk ran no step here: the context it was started with stands)
    if (rk == 0 && tid < T) {
      d.aw[rb * T + tid] = s_aw[tid];
      d.awc[rb * T + tid] = s_awc[tid];
    }
  }
  if (c == 0 && tid == 0) d.ctl[0] = s;
#ifdef XDTTS_P8_PROFILE
  __syncthreads();
  if (g.prof && tid < 32) g.prof[c * 32 + tid] = tid == 31 ? (u64)(s - step0) : (tid >= 17 && tid < 20 ? s_prof[tid] : (u64)s_ts[tid]);  // sums of time stamps (mod 2^32)
#endif
}

// x(0) = prenet(0) = 0 (the prenet has no bias, mod.rs:208) with the chunks' initial active bits (behind the fill of the rings)
__global__ void k_p8_seed(P8Bufs g, const int *limits, int nb) {
  const int b = blockIdx.x, i = threadIdx.x;
  put(g.rx + (size_t)b * PRENET + i, limits[b] > 0 ? 0u : 0x80000000u);
  (void)nb;
}
// parity hook: a sequence that starts at `step` with the prenet output x [B][256] already computed (d.x)
__global__ void k_p8_seed_at(P8Bufs g, const int *limits, const float *x, int step, int nb) {
  const int b = blockIdx.x, i = threadIdx.x;
  put(g.rx + (size_t)b * PRENET + i, value_bits(fabsf(x[b * PRENET + i])) | (limits[b] > step ? 0u : 0x80000000u));
  (void)nb;
}

}  // namespace

// Exchange memory of a launch of `nsteps` steps of B chunks, in 8-byte words: the two granule edges (two step parities) and the
// four write-once rings (one slab per step; x one more, for the step after the last)
static size_t ring_values(int B, int nsteps) { return (size_t)p8_slots(B) * ((size_t)(nsteps + 1) * PRENET + (size_t)nsteps * (ATT_RNN + EMB + DEC_RNN)); }
size_t p8_exchange_words(int B, int nsteps) { return (size_t)2 * GSLOTS * (ATTN_CU * EP_LD + MEL_GL) + (ring_values(B, nsteps) + 1) / 2; }

P8Bufs p8_bufs(unsigned long long *base, int *err, int B, int nsteps) {
  P8Bufs g{};
  const size_t nb = (size_t)p8_slots(B);
  g.ep = base;
  g.mel = g.ep + (size_t)2 * GSLOTS * ATTN_CU * EP_LD;
  g.rx = reinterpret_cast<unsigned *>(g.mel + (size_t)2 * GSLOTS * MEL_GL);
  g.rhatt = g.rx + nb * (size_t)(nsteps + 1) * PRENET;
  g.rctx = g.rhatt + nb * (size_t)nsteps * ATT_RNN;
  g.rhdec = g.rctx + nb * (size_t)nsteps * EMB;
  g.ring_steps = nsteps;
  g.err = err;
  g.delay[0] = g.delay[2] = 16;
  g.delay[1] = 64;
  g.delay[4] = 24;
  g.delay[5] = 16;
  if (nb > NBMAX) {  // 16-slot kernel: its role workgroups run MFMAs between a publish and the poll that answers it; twice the bytes per gather
    g.delay[0] = g.delay[2] = 20;
    g.delay[1] = 80;
    g.delay[4] = g.delay[5] = 0;
  }
  if (const char *e = getenv("XDTTS_P8_DELAY")) sscanf(e, "%d,%d,%d,%d,%d,%d", &g.delay[0], &g.delay[1], &g.delay[2], &g.delay[3], &g.delay[4], &g.delay[5]);  // developer sweep
  return g;
}

static void fill_exchange(const DecoderBufs &d, const P8Bufs &g, hipStream_t s) {
  HIP_CHECK(hipMemsetAsync(g.ep, 0, (size_t)2 * GSLOTS * (ATTN_CU * EP_LD + MEL_GL) * sizeof(unsigned long long), s));
  HIP_CHECK(hipMemsetAsync(g.rx, 0xff, ring_values(d.B, g.ring_steps) * sizeof(unsigned), s));
}

// The grid must be co-resident: one workgroup per CU on a 256-CU part, nothing else of ours running.
bool decoder_p8_supported(int device, int B, int T) {
  if (B < 1 || B > P8_B_MAX || T > TP) return false;
  if (B > NBMAX) return decoder_p16_supported(device, T);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  if (prop.multiProcessorCount < P_NCU) return false;
  int per_cu = 0;
  const void *fn = p8_slots(B) == 4 ? reinterpret_cast<const void *>(k_decoder_persistent8<4>) : reinterpret_cast<const void *>(k_decoder_persistent8<NBMAX>);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, PT, 0) != hipSuccess) return false;
  return per_cu >= 1;
}

void launch_p8_seed(const DecoderBufs &d, const P8Bufs &g, const int *limits_dev, hipStream_t s) {
  fill_exchange(d, g, s);
  hipLaunchKernelGGL(k_p8_seed, dim3(d.B), dim3(PRENET), 0, s, g, limits_dev, p8_slots(d.B));
  HIP_CHECK(hipGetLastError());
}

void launch_p8_seed_at(const DecoderBufs &d, const P8Bufs &g, const int *limits_dev, int step, hipStream_t s) {
  fill_exchange(d, g, s);
  hipLaunchKernelGGL(k_p8_seed_at, dim3(d.B), dim3(PRENET), 0, s, g, limits_dev, d.x, step, p8_slots(d.B));
  HIP_CHECK(hipGetLastError());
}

void launch_decoder_p8(const DecoderBufs &d, const DeviceWeights &w, const P8Bufs &g, int nsteps, hipStream_t s) {
  if (d.B < 1 || d.B > P8_B_MAX || d.T > TP) fail(XDTTS_ERR_BAD_ARG, "persistent MFMA decoder: %d chunks of %d encoder steps (max %d, %d)", d.B, d.T, P8_B_MAX, TP);
  if (nsteps > g.ring_steps) fail(XDTTS_ERR_BAD_ARG, "persistent MFMA decoder: %d steps on an exchange laid out for %d", nsteps, g.ring_steps);
  P8Weights pw{};
  pw.att_w = reinterpret_cast<const float4 *>(w.att_w.p);
  pw.dec_w = reinterpret_cast<const float4 *>(w.dec_w.p);
  pw.q_w = reinterpret_cast<const float4 *>(w.q_w.p);
  pw.proj_w = reinterpret_cast<const float4 *>(w.proj_w.p);
  pw.att_b = w.att_b.p;
  pw.dec_b = w.dec_b.p;
  pw.v_w = w.v_w.p;
  pw.loc_fused = w.loc_fused.p;
  pw.proj_b = w.proj_b.p;
  pw.pre0T = w.pre0T.p;
  pw.pre1T = w.pre1T.p;
  if (d.B > NBMAX) {  // 9..16 chunks: the 16-slot kernel (decoder_persistent16.hip), same exchange, same seed
    launch_decoder_p16(d, pw, g, nsteps, s);
    return;
  }
  const void *fn = p8_slots(d.B) == 4 ? reinterpret_cast<const void *>(k_decoder_persistent8<4>) : reinterpret_cast<const void *>(k_decoder_persistent8<NBMAX>);
#ifdef XDTTS_P8_PROFILE
  static unsigned long long *prof_dev = nullptr;
  if (!prof_dev) HIP_CHECK(hipMalloc((void **)&prof_dev, sizeof(unsigned long long) * P_NCU * 32));
  P8Bufs gp = g;
  gp.prof = prof_dev;
  COOP_CHECK(launch_coresident(true, fn, dim3(P_NCU), dim3(PT), 0, s, d, gp, pw, nsteps));
  HIP_CHECK(hipStreamSynchronize(s));
  static unsigned long long host[P_NCU * 32];
  HIP_CHECK(hipMemcpy(host, prof_dev, sizeof(host), hipMemcpyDeviceToHost));
  for (int c : {0, ATTN_CU * d.B, P_NCU - 1}) {
    const unsigned long long *h = host + c * 32;
    const double steps = (double)h[31];
    if (steps <= 0) continue;
    const bool attn = c < ATTN_CU * d.B, pre = !attn && c < (ATTN_CU + PRE_CU) * d.B;
    static const int order[25] = {0, 1, 2, 3, 4, 5, 6, 20, 21, 22, 23, 7, 8, 9, 10, 11, 12, 13, 14, 15, 24, 25, 26, 27, 16};
    // every marker's slot holds the SUM of its time stamps over the steps: a phase = sum - sum of the marker before it
    unsigned prev = (unsigned)h[16] - (unsigned)h[29] + (unsigned)h[28];  // "marker before" the first one: the previous step's last, the loop start for step 0
    for (int i = 0; i < 25; ++i) {
      const int m = order[i];
      if (((m >= 20 && m <= 22) && !attn) || ((m >= 24 && m <= 27) && !pre)) {
        printf("P8PROF %d %d %d %.3f\n", c, (int)steps, m, 0.0);
        continue;
      }
      printf("P8PROF %d %d %d %.3f\n", c, (int)steps, m, 0.01 * (double)((unsigned)h[m] - prev) / steps);
      prev = (unsigned)h[m];
    }
    for (int i = 17; i < 20; ++i) printf("P8PROF %d %d %d %.3f\n", c, (int)steps, i, (double)h[i] / steps);
  }
  fflush(stdout);
#else
  COOP_CHECK(launch_coresident(true, fn, dim3(P_NCU), dim3(PT), 0, s, d, g, pw, nsteps));
#endif
}

}  // namespace xdtts
