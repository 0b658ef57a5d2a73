// decoder_persistent16.hip -- the Tacotron2 decoder loop (src/tacotron2/mod.rs:302-342) as ONE persistent, weight-stationary
// launch for lock-step batches of 9..16 chunks: decoder_persistent8.hip's engine with sixteen chunk slots, i.e. a full
// 16 x 16 MFMA tile of chunks per step (the 8-slot kernel leaves half of every tile empty; nine chunks used to fall onto the
// two-launch engine of decoder.hip and its 26 us latency chain -- a 54 % cliff at exactly the size a small server batch has).
//
// What sixteen slots change against decoder_persistent8.hip (whose skeleton, exchange layout, seed kernels and fault handling
// this file keeps -- p8_exchange.h, P8Bufs):
//   * the chunks' state vectors no longer live in LDS (x 16 + ctx 32 + h_att 64 + h_dec 64 kB would be 176 of the 160 kB):
//     every wave polls the write-once rings for exactly the 16-byte quads that are ITS MFMA B operands -- lane (k-quad kk,
//     chunk n) of wave w needs columns 16 q + 4 kk .. + 3 of its K-quarter of chunk n's vector, which is one producer's four
//     units -- and multiplies straight out of the registers the loads landed in.  No LDS landing, no barrier between a gather
//     and its MFMAs, and each vector is read once for both LSTMs that consume it (h_att(s): the decoder LSTM of step s and the
//     attention LSTM of step s + 1, back to back);
//   * 8 + 8 role workgroups per chunk instead of 8 + 16 (16 x 16 = the grid): a projection / prenet workgroup owns 32 layer-2
//     columns and up to 11 rows of [W_p ; w_gate];
//   * the register file decides where the role tables live: 256 of a lane's 512 registers are LSTM weights (the 16 prenet-column
//     operands of the attention LSTM sit in LDS), so the projection rows come from L2 every step (18 kB per wave, requested ahead of
//     the poll they wait behind), layer 2 ([segment][input quad][column][4]), the encoder-memory slice ([t / 4][64][4]) and the gate
//     biases live in LDS, and loop-invariant store addresses are re-formed per step.  No scratch;
//   * ROLE FIRST: a role workgroup fetches its own chunk's vector (one quad per thread, row-major into LDS), requests the first
//     operand rounds behind its arrival, runs its role and publishes; the attention role multiplies ONE round of h_att while its
//     partial energies travel and the rest behind its context publish, the projection / prenet role multiplies h_dec behind its x
//     publish -- the step's critical chain never waits for the bulk.  (The -D knobs below are the experiments of DESIGN_NOTES.md,
//     round 6; the defaults are the measured best.)
//   * the chunks' active bits are per-lane state (the sign of x(s), as before) combined by a ballot, shared through one LDS word.
// Per step:  x -> [attention LSTM] -> h_att -> [query, energies] -> e -> [softmax, context] -> ctx -> [decoder LSTM] -> h_dec
//            -> [projection rows] -> mel -> [stop rule, prenet] -> x(s+1)
// Every spin is bounded and watches a global error word; a timed-out exchange ends the launch and the request is decoded again
// by the other engines (api.cpp), exactly as for the 8-slot kernel.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "p8_exchange.h"

namespace xdtts {

namespace {

constexpr int PT = 256, NW = PT / 64, NB = P8_B_MAX, P_NCU = ATT_RNN / 4, TP = PERSIST_T_MAX;
constexpr int ATTN_CU = 8, PRE_CU = 8, EP_LD = TP, MEL_GL = 96, WPAD = TP + 32;
constexpr int L2C = PRENET / PRE_CU;  // layer-2 columns per projection / prenet workgroup
#ifndef XDTTS_P16_HSPLIT
#define XDTTS_P16_HSPLIT 4
#endif
#ifndef XDTTS_P16_NBUF
#define XDTTS_P16_NBUF 2
#endif
constexpr int HSPLIT = XDTTS_P16_HSPLIT, HQ = 16 / HSPLIT, NBUF = XDTTS_P16_NBUF;  // NBUF: rounds of operand registers (rounds in flight)  // the sixteen operand quads of a hidden vector in this many gather + MFMA rounds
#ifndef XDTTS_P16_AUX
#define XDTTS_P16_AUX 16
#endif
constexpr int AUX1 = XDTTS_P16_AUX;  // cache policy of the FIRST poll of an operand quad (16 = sc1; retries are sc0 sc1 always)
#ifndef XDTTS_P16_ROLE_FIRST
#define XDTTS_P16_ROLE_FIRST 1
#endif
#ifndef XDTTS_P16_EP_RD
#define XDTTS_P16_EP_RD 2
#endif
constexpr bool ROLE_FIRST = XDTTS_P16_ROLE_FIRST != 0;  // role workgroups: the exchange chain first, the hidden vector's MFMAs behind the role's publish
#ifndef XDTTS_P16_RF_SPLIT
#define XDTTS_P16_RF_SPLIT 1
#endif
#ifndef XDTTS_P16_RF_SPLIT_P
#define XDTTS_P16_RF_SPLIT_P 0
#endif
constexpr int RF_SPLIT = XDTTS_P16_RF_SPLIT, RF_SPLIT_P = XDTTS_P16_RF_SPLIT_P;            // ROLE_FIRST: rounds of a hidden vector's MFMAs a role workgroup runs inside its exchange's shadow (the rest behind its publish)
#ifndef XDTTS_P16_EARLY_BEGIN
#define XDTTS_P16_EARLY_BEGIN 0
#endif
constexpr bool EARLY_BEGIN = XDTTS_P16_EARLY_BEGIN != 0;  // role workgroups: the first operand rounds in flight next to the row's first poll (1) or behind its arrival (0)
constexpr int EP_RD = XDTTS_P16_EP_RD;                  // !ROLE_FIRST: the first poll of the partial energies leaves behind this round of the h_att MFMAs
constexpr int PROWS = 3;                              // rows of [W_p ; w_gate] per wave: rk + 8 (wave + 4 r), r < 3
constexpr unsigned P_SPIN_LIMIT = 1u << 21;
static_assert(NB == 16 && ATT_RNN == DEC_RNN && P_NCU == 256 && (ATTN_CU + PRE_CU) * NB == P_NCU, "the role workgroups of 16 chunks are the grid");
static_assert(TP == 128 && PRENET == PT && EMB == 2 * PT && ATT_RNN == 4 * PT, "thread <-> quad maps below");
static_assert(PRE_CU * NW * PROWS >= N_MEL + 1, "every row of [W_p ; w_gate] has a wave");

// N quads of a write-once slab straight into the registers they are multiplied from: quad i at byte offset at(i); a quad is
// complete once none of its four words is the fill pattern.  Lanes with `on` false (a chunk slot that is empty or has
// stopped: nobody publishes for it) poll nothing and get zeros.  First round sc1, retries sc0 sc1 and only for what is missing.
template <int N>
__device__ __forceinline__ unsigned operand_gather(u32x4 (&v)[N], const unsigned *slab, unsigned base, bool on, const PollCtl &pc) {
  // quad i at byte offset base + 64 i of the slab (the next column group of the same chunk): ONE address register -- made opaque
  // here, or the loop-invariant sums base + 64 i of every gather of the step are kept in registers across the whole loop -- and
  // the instruction's immediate offset
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)slab, 0, 0x7fffffff, 0x00020000);
  asm volatile("" : "+v"(base));
  unsigned pending = on ? (N < 32 ? (1u << (N & 31)) - 1u : ~0u) : 0u, spins = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = (u32x4){0u, 0u, 0u, 0u};
  if (on) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(base + 64u * i), 0, 16);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (v[i].x != UNWRITTEN && v[i].y != UNWRITTEN && v[i].z != UNWRITTEN && v[i].w != UNWRITTEN) pending &= ~(1u << i);
  }
  while (pending) {
    if (give_up(spins, pc)) break;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if ((pending >> i) & 1u) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(base + 64u * i), 0, 17);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (((pending >> i) & 1u) && v[i].x != UNWRITTEN && v[i].y != UNWRITTEN && v[i].z != UNWRITTEN && v[i].w != UNWRITTEN) pending &= ~(1u << i);
    asm volatile("" ::: "memory");
  }
  return spins;
}
// The sixteen quads of a hidden vector as R rounds of NQ through two rounds of registers.  stream_begin puts the first-round loads
// of rounds 0 and 1 in flight (a role workgroup does so next to the load of its own chunk's row and runs its role on that row
// while they land); stream_finish checks round r (re-polled where a producer is late), hands it to consume(r, quads) and
// re-uses its registers for round r + 2.  Loads are unconditional -- an empty or stopped chunk's row of the slab exists, its
// contents are replaced by zeros: loads under a per-lane condition are control flow, and at its joins the compiler waits for the
// NEXT round's loads as well.  ONE address register (made opaque, or the loop-invariant sums base + 64 i of every gather of the
// step are kept in registers across the whole loop) + the instruction's immediate offset.
template <int NQ>
struct OpBuf {
  u32x4 b[NBUF][NQ];
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(const unsigned *slab) { return __builtin_amdgcn_make_buffer_rsrc((void *)slab, 0, 0x7fffffff, 0x00020000); }
template <int NQ>
__device__ __forceinline__ void stream_issue(u32x4 (&v)[NQ], const __amdgpu_buffer_rsrc_t r, unsigned base, int rd) {
#pragma unroll
  for (int i = 0; i < NQ; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(base + 64u * (NQ * rd + i)), 0, AUX1);
}
template <int NQ>
__device__ __forceinline__ void stream_begin(OpBuf<NQ> &ob, const unsigned *slab, unsigned base) {
  const __amdgpu_buffer_rsrc_t r = slab_rsrc(slab);
  asm volatile("" : "+v"(base));
#pragma unroll
  for (int rd = 0; rd < NBUF; ++rd) stream_issue<NQ>(ob.b[rd], r, base, rd);
}
template <int NQ, int R, int FROM = 0, int TO = R, class F>
__device__ __forceinline__ void stream_finish(OpBuf<NQ> &ob, const unsigned *slab, unsigned base, bool on, const PollCtl &pc, F consume) {
  const __amdgpu_buffer_rsrc_t r = slab_rsrc(slab);
  asm volatile("" : "+v"(base));
#pragma unroll
  for (int rd = FROM; rd < TO; ++rd) {
    u32x4(&v)[NQ] = ob.b[rd % NBUF];
    unsigned pending = 0u, spins = 0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      pending |= (v[i].x == UNWRITTEN || v[i].y == UNWRITTEN || v[i].z == UNWRITTEN || v[i].w == UNWRITTEN) ? 1u << i : 0u;
      if (!on) v[i] = (u32x4){0u, 0u, 0u, 0u};
    }
    if (!on) pending = 0u;
    while (pending) {
      if (give_up(spins, pc)) break;
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        if ((pending >> i) & 1u) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(base + 64u * (NQ * rd + i)), 0, 17);
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        if (((pending >> i) & 1u) && v[i].x != UNWRITTEN && v[i].y != UNWRITTEN && v[i].z != UNWRITTEN && v[i].w != UNWRITTEN) pending &= ~(1u << i);
      asm volatile("" ::: "memory");
    }
    consume(rd, v);
    if (rd + NBUF < R) stream_issue<NQ>(v, r, base, rd + NBUF);
  }
}
// one quad whose first-round load `v` is in flight: polled until complete
__device__ __forceinline__ void quad_wait(u32x4 &v, const unsigned *slab, unsigned off, const PollCtl &pc) {
  const __amdgpu_buffer_rsrc_t r = slab_rsrc(slab);
  unsigned spins = 0;
  while (v.x == UNWRITTEN || v.y == UNWRITTEN || v.z == UNWRITTEN || v.w == UNWRITTEN) {
    if (give_up(spins, pc)) break;
    v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 17);
    asm volatile("" ::: "memory");
  }
}
__device__ __forceinline__ float bits(unsigned u) { return __uint_as_float(u); }
// NQ column groups of a K-quarter: A[q] = the lane's four weights of columns 16 q + 4 kk .. + 3 of its gate row, b[q] = the
// same columns of its chunk
template <int NQ>
__device__ __forceinline__ void mfma_regs(f32x4 &acc, const float4 (&A)[NQ], const u32x4 (&b)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].x, bits(b[q].x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].y, bits(b[q].y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].z, bits(b[q].z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].w, bits(b[q].w), acc, 0, 0, 0);
  }
}
template <int NQ, int NA>
__device__ __forceinline__ void mfma_half(f32x4 &acc, const float4 (&A)[NA], const u32x4 (&b)[NQ], int q0) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q].x, bits(b[q].x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q].y, bits(b[q].y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q].z, bits(b[q].z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q].w, bits(b[q].w), acc, 0, 0, 0);
  }
}
// two independent accumulators fed from the same operand quads, their MFMAs alternating (each chain's next MFMA issues 64 cycles
// after its last: the 40-cycle dependent latency never shows)
template <int NQ, int NA>
__device__ __forceinline__ void mfma_pair(f32x4 &acc0, const float4 (&A0)[NA], f32x4 &acc1, const float4 (&A1)[NA], const u32x4 (&b)[NQ], int q0) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[q0 + q].x, bits(b[q].x), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[q0 + q].x, bits(b[q].x), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[q0 + q].y, bits(b[q].y), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[q0 + q].y, bits(b[q].y), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[q0 + q].z, bits(b[q].z), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[q0 + q].z, bits(b[q].z), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[q0 + q].w, bits(b[q].w), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[q0 + q].w, bits(b[q].w), acc1, 0, 0, 0);
  }
}
// one weight slab, two accumulator chains (even / odd column groups), summed by the caller
template <int NQ, int NA>
__device__ __forceinline__ void mfma_two(f32x4 &acc0, f32x4 &acc1, const float4 (&A)[NA], const u32x4 (&b)[NQ], int q0) {
  static_assert(NQ % 2 == 0, "pairs of column groups");
#pragma unroll
  for (int q = 0; q < NQ; q += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q].x, bits(b[q].x), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q + 1].x, bits(b[q + 1].x), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q].y, bits(b[q].y), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q + 1].y, bits(b[q + 1].y), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q].z, bits(b[q].z), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q + 1].z, bits(b[q + 1].z), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q].w, bits(b[q].w), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q0 + q + 1].w, bits(b[q + 1].w), acc1, 0, 0, 0);
  }
}
// the same from a row-major state array of the sequence so far (launch set-up): chunk n's vector at base + n * ld
template <int NQ>
__device__ __forceinline__ void mfma_rows(f32x4 &acc, const float4 (&A)[NQ], const float *base, int ld, int col0, int kk, int n, bool on) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float4 b = on ? *reinterpret_cast<const float4 *>(base + (size_t)n * ld + col0 + 16 * q + 4 * kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].w, b.w, acc, 0, 0, 0);
  }
}

// Developer build (-DXDTTS_P8_PROFILE): thread 0 of three workgroups (one per role) accumulates the 100 MHz wall clock between
// phase markers; launch_decoder_p16 prints the sums.
#ifdef XDTTS_P8_PROFILE
#define P16_MARK(i)                                     \
  do {                                                  \
    s_ts[wave * 32 + (i)] += (unsigned)wall_clock64();  \
  } while (0)
#else
#define P16_MARK(i) do { } while (0)
#endif

__global__ __launch_bounds__(PT) void k_decoder_persistent16(DecoderBufs d, P8Bufs g, P8Weights w, int nsteps) {
  constexpr int ATTN_FLOATS = 2 * TP * 16 + 3 * TP + 16 + PT + 2 * WPAD + 62 * 16 + 16 + 16 * PT * 4 + TP * 64;
  constexpr int PRE_FLOATS = N_MEL * PRENET + MEL_GL + 16 + PRENET + L2C * PRENET;
  // ONE LDS object (role area first): accumulator exchange, the role's own chunk row-major, flags
  __shared__ __attribute__((aligned(16))) float s_all[(ATTN_FLOATS > PRE_FLOATS ? ATTN_FLOATS : PRE_FLOATS) + NW * 64 * 4 + ATT_RNN + EMB + 8 + NW * 4 * 64 * 4 + 64 * 8];
  float *const s_role = s_all, *const s_acc = s_role + (ATTN_FLOATS > PRE_FLOATS ? ATTN_FLOATS : PRE_FLOATS), *const s_hrow = s_acc + NW * 64 * 4,
               *const s_crow = s_hrow + ATT_RNN;
  int *const s_flag = reinterpret_cast<int *>(s_crow + EMB);
  float *const s_ax = s_crow + EMB + 8;
  float *const s_bias = s_ax + NW * 4 * 64 * 4;  // [64 lanes][8]: the gate biases of wave 0's (unit, chunk) lanes, attention LSTM then decoder LSTM  // [wave][4 column groups][lane] float4: the attention LSTM's prenet columns (A operands the register file has no room for)  // [0] error word seen by this workgroup, [1] active-chunk mask of the step
  // attention role
  float *s_pm = s_role, *s_loc = s_pm + TP * 16, *s_aw = s_loc + TP * 16, *s_awc = s_aw + TP, *s_e = s_awc + TP, *s_q = s_e + TP,
        *s_part = s_q + 16, *s_wpad = s_part + PT, *s_G = s_wpad + 2 * WPAD, *s_vv = s_G + 62 * 16,
        *s_qw = s_vv + 16,  // [16][PT] float4: query rows 16 rk + wave + 4 r, 4 x 16 B per lane each
        *s_mem = s_qw + 16 * PT * 4;  // [TP / 4][64][4]: the encoder memory's columns 64 rk .. + 63, four steps per 16-byte read (the context slice this workgroup sums)
  // projection + prenet role
  float *s_W0 = s_role, *s_mel = s_W0 + N_MEL * PRENET, *s_pb = s_mel + MEL_GL, *s_p2 = s_pb + 16, *s_w1 = s_p2 + PRENET;  // s_W0 [20][256][4]; s_p2: layer-2 partial sums [8 segments][32 columns]

  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = d.B, T = d.T;
  const PollCtl pc{g.err, g.spins > 0 ? (unsigned)g.spins : P_SPIN_LIMIT};
  if (g.fault && c == g.fault - 1) return;  // test hook: this workgroup never shows up
  const int kk = lane >> 4, n16 = lane & 15;  // MFMA lane coordinates: k-quad / chunk column

  // wave 0 finalises both cells: lane = (unit u = lane / 16, chunk n16); its four registers of a D tile are the gates i,f,g,o
  const int cu = lane >> 4;
  float c_att = 0.f, c_dec = 0.f, h_att_last = 0.f, h_dec_last = 0.f;
  const bool cell = wave == 0 && n16 < B;
  if (wave == 0) {
    *reinterpret_cast<float4 *>(s_bias + 8 * lane) = *reinterpret_cast<const float4 *>(w.att_b + 16 * c + 4 * cu);  // (read back by the same lane only)
    *reinterpret_cast<float4 *>(s_bias + 8 * lane + 4) = *reinterpret_cast<const float4 *>(w.dec_b + 16 * c + 4 * cu);
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
  const int step0 = __builtin_amdgcn_readfirstlane(d.ctl[0]);
  bool alive = n16 < B && step0 < d.nframes[n16 < B ? n16 : 0];  // this lane's chunk still publishes
  if (tid == 0) s_flag[0] = s_flag[1] = 0;

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
#pragma unroll 1
    for (int i = tid; i < TP * 64; i += PT) s_mem[(((i >> 8) * 64) + (i & 63)) * 4 + ((i >> 6) & 3)] = (i >> 6) < T ? d.memory[((size_t)rb * T + (i >> 6)) * EMB + 64 * rk + (i & 63)] : 0.f;
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
    // layer 2: the workgroup's 32 columns; thread (column c = tid % 32, input segment tid / 32) reads eight 16-byte vectors of its 32 inputs
#pragma unroll 1
    for (int i = tid; i < L2C * PRENET; i += PT) {  // element (column c = i % 32, input k = i / 32) -> [k / 32][(k % 32) / 4][c][k % 4]
      const int cc = i % L2C, k = i / L2C;
      s_w1[(((k >> 5) * 8 + ((k & 31) >> 2)) * L2C + cc) * 4 + (k & 3)] = w.pre1T[(unsigned)(k * PRENET + L2C * rk + cc)];
    }
  }
  int nf_r = __builtin_amdgcn_readfirstlane(pre ? d.nframes[rb] : 0);  // (wave-uniform values in scalar registers: the vector file has none to spare)
  const int nv_r = __builtin_amdgcn_readfirstlane(attn ? d.n_valid[rb] : 0);
  bool ctx_valid = false;
  __syncthreads();
  if (pre && tid < NW * PROWS) {  // (behind the s_W0 fill: s_pb follows it in the role area) bias of row rk + 8 tid
    const int prow = rk + PRE_CU * tid;
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
  float4 ac[8], ah[16];          // attention LSTM: x 256 (in LDS: s_ax) | ctx 512 | h_att 1024 columns, 1/4 of each
  float4 dh[16], dc[8], dd[16];  // decoder LSTM:   h_att 1024 | ctx 512 | h_dec 1024
  {
    const float4 *ra = w.att_w + (size_t)(16 * c + n16) * (ATT_COLS / 4), *rd = w.dec_w + (size_t)(16 * c + n16) * (DEC_COLS / 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(s_ax + ((wave * 4 + q) * 64 + lane) * 4) = ld_stream(ra + (0 + 64 * wave) / 4 + 4 * q + kk);  // (read back by the same lane only)
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
  // the partial pre-activations that do not depend on the newest vector: the state of the sequence so far (zeros at step 0; a
  // previous launch's write-back otherwise), chunk slots beyond B zero
  f32x4 accA = (f32x4){0.f, 0.f, 0.f, 0.f}, accD = accA;
  mfma_rows<8>(accA, ac, d.ctx, EMB, 128 * wave, kk, n16, n16 < B);             // attention LSTM: ctx(s-1) ...
  mfma_rows<16>(accA, ah, d.att_h[0], ATT_RNN, 256 * wave, kk, n16, n16 < B);   // ... and h_att(s-1)
  mfma_rows<16>(accD, dd, d.dec_h[0], DEC_RNN, 256 * wave, kk, n16, n16 < B);   // decoder LSTM: h_dec(s-1)

  // wave 0: sum of the K-slices of a D tile (the caller has put a barrier behind the s_acc stores)
  auto reduce_tile = [&]() {
    f32x4 gsum = *reinterpret_cast<const f32x4 *>(s_acc + lane * 4);
#pragma unroll
    for (int q = 1; q < NW; ++q) gsum += *reinterpret_cast<const f32x4 *>(s_acc + (q * 64 + lane) * 4);
    return gsum;
  };

#ifdef XDTTS_P8_PROFILE
  __shared__ unsigned s_ts[NW * 32];  // time stamp sums of every marker, in program order
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
    const size_t slot = (size_t)(s - step0) * NB;
    // (an opaque zero per step in every lane-dependent store address: left alone, the loop-invariant 64-bit address of each of the
    // step's publishes is kept in a register pair across the whole loop -- and spilled)
    unsigned oz = 0u;
    asm volatile("" : "+v"(oz));
    // ---- P1: x(s) and the chunks' active bits ---------------------------------------------------------------------------------
    u32x4 xq[4];
    float4 ax[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ax[q] = lds4(s_ax + ((wave * 4 + q) * 64 + lane) * 4);
    nap(g.delay[3]);
    operand_gather<4>(xq, g.rx + slot * PRENET, 4u * (unsigned)(n16 * PRENET + 64 * wave + 4 * kk), alive, pc);
    {
      // the sign bit of a column = chunk not active (x >= 0: it leaves a ReLU); every column of a chunk carries it
      const bool act_l = alive && !(xq[0].x >> 31);
      const unsigned m = (unsigned)(__ballot(act_l) & 0xffffull);  // lanes 0..15: k-quad 0, chunk = lane
      if (tid == 0) s_flag[1] = (int)m;
      if (tid == PT - 1) s_flag[0] = __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int q = 0; q < 4; ++q) xq[q] &= 0x7fffffffu;
    }
    P16_MARK(0);
    __syncthreads();
    P16_MARK(1);
    const unsigned actm = (unsigned)s_flag[1];
    if (!actm || s_flag[0] != 0) break;  // every chunk has stopped (or an exchange failed): the launch ends by itself
    const bool act_r = (actm >> rb) & 1u;
    const bool act_n = (actm >> n16) & 1u;
    alive = act_n;
    // attention LSTM: close the rows with the x columns
    mfma_regs<4>(accA, ax, xq);
    *reinterpret_cast<f32x4 *>(s_acc + (wave * 64 + lane) * 4) = accA;
    __syncthreads();
    if (wave == 0) {
      const float4 bias_a = lds4(s_bias + 8 * lane);
      const f32x4 gs = reduce_tile();
      if (cell && act_n) {
        const float ig = fast_sigmoid(gs[0] + bias_a.x), fg = fast_sigmoid(gs[1] + bias_a.y), gg = fast_tanh(gs[2] + bias_a.z),
                    og = fast_sigmoid(gs[3] + bias_a.w);
        c_att = fmaf(fg, c_att, ig * gg);
        h_att_last = og * fast_tanh(c_att);
        put(g.rhatt + (slot + (n16 + oz)) * ATT_RNN + 4 * c + cu, value_bits(h_att_last));
      }
    }
    accA = (f32x4){0.f, 0.f, 0.f, 0.f};
    P16_MARK(2);
    // ---- P2: h_att(s): the attention role's own chunk first, then every active chunk's operand quads ------------------------------
    const bool attn_on = attn && act_r, pre_on = pre && act_r;
    const unsigned *slab_h = g.rhatt + slot * ATT_RNN;
    // loads i of the partial-energy gather: row j + 4 (i % 2), encoder step t + 64 (i / 2)
    const int ep_t = tid >> 2, ep_j = tid & 3;
    const u64 *ep_base = g.ep + (unsigned)(((p * NB + rb) * ATTN_CU + ep_j) * EP_LD + ep_t);
    const unsigned ep_need = attn_on ? ((ep_t < T ? 3u : 0u) | (ep_t + 64 < T ? 12u : 0u)) : 0u;
    auto ep_at = [](int i) { return (unsigned)((i & 1) * 4 * EP_LD + (i >> 1) * 64); };
    u64 ep_v[4] = {0, 0, 0, 0};
    OpBuf<HQ> hb;
    const unsigned hbase = 4u * (unsigned)(n16 * ATT_RNN + 256 * wave + 4 * kk);
    if (attn_on) {
      // what the energies need besides the query is in registers before h_att arrives; thread = (encoder steps t = tid / 4 and
      // t + 64, dims 4 (tid % 4) .. + 3 of the workgroup's 16)
      const float4 l0 = lds4(s_loc + 4 * tid), p0 = lds4(s_pm + 4 * tid), l1 = lds4(s_loc + 4 * (tid + PT)), p1 = lds4(s_pm + 4 * (tid + PT));
      const float4 lp0 = make_float4(l0.x + p0.x, l0.y + p0.y, l0.z + p0.z, l0.w + p0.w);
      const float4 lp1 = make_float4(l1.x + p1.x, l1.y + p1.y, l1.z + p1.z, l1.w + p1.w);
      const float4 v4 = lds4(s_vv + 4 * (tid & 3));
      nap(g.delay[0]);
      // the workgroup's own chunk first (one quad per thread, row-major into LDS), the first operand rounds in flight behind it
      const unsigned roff = 4u * (unsigned)(rb * ATT_RNN + 4 * tid);
      u32x4 hr = __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc(slab_h), (int)roff, 0, 16);
      if (EARLY_BEGIN) stream_begin<HQ>(hb, slab_h, hbase);
      quad_wait(hr, slab_h, roff, pc);
      if (!EARLY_BEGIN) stream_begin<HQ>(hb, slab_h, hbase);  // (every producer of the row has published: so have, for the other chunks, most of them)
      *reinterpret_cast<float4 *>(s_hrow + 4 * tid) = as_f4(hr);
      P16_MARK(3);
      __syncthreads();
      P16_MARK(4);
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
    } else {
      nap(g.delay[0]);
      stream_begin<HQ>(hb, slab_h, hbase);
    }
    P16_MARK(5);
    // both LSTMs that consume h_att(s) -- the decoder LSTM of this step, the attention LSTM of the next -- out of ONE gather.  The
    // attention workgroups are the step's critical chain: they go on to the partial energies at once and multiply behind their
    // context publish, in the time the context travels (ROLE_FIRST); everybody else multiplies now.
    auto h_att_mfmas = [&](auto from, auto to) {
      stream_finish<HQ, HSPLIT, decltype(from)::value, decltype(to)::value>(hb, slab_h, hbase, act_n, pc, [&](int rd, const u32x4(&hq)[HQ]) {
        mfma_pair<HQ>(accD, dh, accA, ah, hq, HQ * rd);
        if (!ROLE_FIRST && rd == EP_RD && ep_need) gather_issue<4>(ep_v, ep_base, ep_need, ep_at);
      });
    };
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, RF_SPLIT>;
    using IR = std::integral_constant<int, HSPLIT>;
    if (!(ROLE_FIRST && attn_on)) h_att_mfmas(I0{}, IR{});
    else {
      // the first rounds (prefetched at the row gather) in the time the partial energies travel, the loads of the rest behind them
      h_att_mfmas(I0{}, IH{});
      nap(g.delay[4]);
      gather_issue<4>(ep_v, ep_base, ep_need, ep_at);
    }
    P16_MARK(6);
    // ---- P3 (attention role): the 8 partial-energy rows of the chunk -> softmax -> this workgroup's 64 context columns ------------
    if (attn_on) {
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
      P16_MARK(20);
      __syncthreads();
      P16_MARK(21);
      {  // every wave: the softmax in registers, lane <-> steps lane, lane + 64
        const float e0 = s_e[lane], e1 = s_e[lane + 64];
        const float m = wave_max(fmaxf(e0, e1));
        const float x0 = fast_exp(e0 - m), x1 = fast_exp(e1 - m);
        const float rs = __builtin_amdgcn_rcpf(wave_sum(x0 + x1));
        const float w0 = x0 * rs, w1 = x1 * rs;
        // context columns: this wave sums the step quads tq = wave + 4 j (steps 4 tq .. + 3, one 16-byte read); lane = column.  The
        // weight of step t sits in lane t % 64 of w0 (t < 64) / w1
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TP / 4 / NW; ++j) {
          const int tq = wave + NW * j;
          const float4 m4 = lds4(s_mem + (tq * 64 + lane) * 4);
          const float ws = j < 16 / NW ? w0 : w1;
          a4.x = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ws), (4 * tq + 0) & 63)), m4.x, a4.x);
          a4.y = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ws), (4 * tq + 1) & 63)), m4.y, a4.y);
          a4.z = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ws), (4 * tq + 2) & 63)), m4.z, a4.z);
          a4.w = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ws), (4 * tq + 3) & 63)), m4.w, a4.w);
        }
        const float acc = (a4.x + a4.y) + (a4.z + a4.w);
        s_part[tid] = acc;
        if (wave == 0) {  // kept for the next step's location features
          s_aw[lane] = w0;
          s_awc[lane] += w0;
          s_aw[lane + 64] = w1;
          s_awc[lane + 64] += w1;
        }
      }
      P16_MARK(22);
      __syncthreads();
      if (tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < NW; ++u) v += s_part[u * 64 + tid];
        cown = v;
        ctx_valid = true;
        put(g.rctx + (slot + rb) * EMB + 64 * rk + (tid + oz), value_bits(v));
      }
      if (ROLE_FIRST) h_att_mfmas(IH{}, IR{});
    }
    P16_MARK(7);
    // ---- P4: ctx(s) of every active chunk -> decoder LSTM ------------------------------------------------------------------------
    u32x4 cq[8];
    if (!attn_on) nap(g.delay[1]);  // (the attention workgroups are the producers: the others have 3 us to wait)
    operand_gather<8>(cq, g.rctx + slot * EMB, 4u * (unsigned)(n16 * EMB + 128 * wave + 4 * kk), act_n, pc);
    P16_MARK(8);
    mfma_regs<8>(accD, dc, cq);
    *reinterpret_cast<f32x4 *>(s_acc + (wave * 64 + lane) * 4) = accD;
    __syncthreads();
    P16_MARK(9);
    if (wave == 0) {
      const float4 bias_d = lds4(s_bias + 8 * lane + 4);
      const f32x4 gs = reduce_tile();
      if (cell && act_n) {
        const float ig = fast_sigmoid(gs[0] + bias_d.x), fg = fast_sigmoid(gs[1] + bias_d.y), gg = fast_tanh(gs[2] + bias_d.z),
                    og = fast_sigmoid(gs[3] + bias_d.w);
        c_dec = fmaf(fg, c_dec, ig * gg);
        h_dec_last = og * fast_tanh(c_dec);
        put(g.rhdec + (slot + (n16 + oz)) * DEC_RNN + 4 * c + cu, value_bits(h_dec_last));
      }
    }
    accD = (f32x4){0.f, 0.f, 0.f, 0.f};
    P16_MARK(10);
    mfma_regs<8>(accA, ac, cq);     // attention LSTM of the next step: ctx(s)
    if (attn_on) location();        // ... and its location features
    P16_MARK(11);
    // ---- P5: h_dec(s): the projection role's own chunk first (with its context), then the operand quads -------------------------
    const unsigned *slab_d = g.rhdec + slot * DEC_RNN;
    bool drop1 = false;
    unsigned drop2 = 0u;
    OpBuf<HQ> db;
    const unsigned dbase = 4u * (unsigned)(n16 * DEC_RNN + 256 * wave + 4 * kk);
    float4 pw[PROWS][6];  // projection role: rows rk + 8 (wave + 4 r), r < 2, of [W_p ; w_gate] (12 kB per wave from L2, in flight across the hashes, the nap
                      // and the poll of h_dec: the register file has no room to keep them from step to step; the waves' third rows sit in LDS)
    if (pre_on) {
#pragma unroll
      for (int r = 0; r < PROWS; ++r)
#pragma unroll
        for (int j = 0; j < 6; ++j) pw[r][j] = w.proj_w[(unsigned)(min(rk + PRE_CU * (wave + NW * r), N_MEL) * (PROJ_IN / 4) + lane + 64 * j)];  // (clamped: unconditional loads)
      // the Bernoulli(0.5) masks of step s + 1 (they do not depend on the data) are hashed in the time the first poll of h_dec
      // could not succeed anyway
      if (d.dropout_mode) {
        drop1 = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 0, tid);
        drop2 = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 1, L2C * rk + (tid & (L2C - 1))) ? 1u : 0u;
        drop2 |= drop1 ? 256u : 0u;
        asm volatile("" : "+v"(drop2));  // (computed HERE: left alone the compiler sinks the hashes behind the gather, onto the critical path)
        drop1 = (drop2 & 256u) != 0u;
      }
      nap(g.delay[2]);
      // units 4 tid .. + 3 of h_dec(rb) first, the first operand rounds in flight behind it; threads < 128: columns 4 tid .. + 3 of
      // ctx(rb), long arrived
      const unsigned roff = 4u * (unsigned)(rb * DEC_RNN + 4 * tid);
      u32x4 hr = __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc(slab_d), (int)roff, 0, 16);
      if (EARLY_BEGIN) stream_begin<HQ>(db, slab_d, dbase);
      quad_wait(hr, slab_d, roff, pc);
      if (!EARLY_BEGIN) stream_begin<HQ>(db, slab_d, dbase);
      *reinterpret_cast<float4 *>(s_hrow + 4 * tid) = as_f4(hr);
      if (tid < EMB / 4) {
        u32x4 cr[1];
        operand_gather<1>(cr, g.rctx + slot * EMB, 4u * (unsigned)(rb * EMB + 4 * tid), true, pc);
        *reinterpret_cast<float4 *>(s_crow + 4 * tid) = as_f4(cr[0]);
      }
      P16_MARK(12);
      __syncthreads();
      P16_MARK(13);
#pragma unroll
      for (int r = 0; r < PROWS; ++r) {
        const int prow = rk + PRE_CU * (wave + NW * r);
        if (prow <= N_MEL) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) a = dot4(pw[r][j], lds4(s_hrow + 256 * j + 4 * lane), a);
#pragma unroll
          for (int j = 0; j < 2; ++j) a = dot4(pw[r][4 + j], lds4(s_crow + 256 * j + 4 * lane), a);
          a = wave_sum(a);
          if (lane == 0) publish(g.mel + (unsigned)((p * NB + rb) * MEL_GL + prow), want, a + s_pb[wave + NW * r]);
        }
      }
    } else {
      nap(g.delay[2]);
      stream_begin<HQ>(db, slab_d, dbase);
    }
    P16_MARK(14);
    // decoder LSTM of the next step: h_dec(s) (two accumulator chains: the dependent-MFMA latency is 40 cycles, the issue rate 32).
    // The projection / prenet workgroups go on to the mel exchange and multiply behind their x publish (ROLE_FIRST).
    f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto h_dec_mfmas = [&](auto from, auto to) {
      stream_finish<HQ, HSPLIT, decltype(from)::value, decltype(to)::value>(db, slab_d, dbase, act_n, pc,
                                                                            [&](int rd, const u32x4(&dq)[HQ]) { mfma_two<HQ>(accD, acc2, dd, dq, HQ * rd); });
    };
    using IP = std::integral_constant<int, RF_SPLIT_P>;
    if (!(ROLE_FIRST && pre_on)) h_dec_mfmas(I0{}, IR{});
    else {
      h_dec_mfmas(I0{}, IP{});  // (in the time the mel rows travel)
      nap(g.delay[5]);
    }
    P16_MARK(15);
    // ---- P6 (projection + prenet role): frame s, stop rule, x(s+1) ---------------------------------------------------------------
    if (pre_on) {  // a chunk's last x (active bit clear) is published at the step it stops
      if (tid < N_MEL + 1) {
        s_mel[tid] = 0.f;
        gather<1>(g.mel + (unsigned)((p * NB + rb) * MEL_GL + tid), want, 1u, pc, [](int) { return 0u; }, [&](int, float v, unsigned) { s_mel[tid] = v; });
      }
      P16_MARK(24);
      __syncthreads();
      P16_MARK(25);
      const float gate = s_mel[N_MEL];
      const bool fired = d.use_gate && gate_fires(gate, d.gate_lo, d.gate_hi, d.gate_threshold);  // mod.rs:319-324
      if (rk == 0) {
        if (tid < N_MEL) d.frames[((size_t)rb * d.max_steps + s) * N_MEL + (tid + oz)] = s_mel[tid];
        if (tid == 0) {
          d.gates[(size_t)rb * d.max_steps + s] = gate;
          if (fired) d.nframes[rb] = s + 1;  // the tripping frame is kept
        }
      }
      if (fired) nf_r = s + 1;
      const bool nxt = s + 1 < nf_r;
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
        P16_MARK(26);
        __syncthreads();  // (s_mel read by everyone before s_hrow, free since the projection, takes the layer-1 outputs)
        s_hrow[tid] = drop1 ? 0.f : (d.dropout_mode ? 2.f * acc : acc);
        __syncthreads();
        // layer 2: thread = (column c of the workgroup's 32, segment of 32 inputs); two chains per thread, eight partials per column
        {
          const int cc = tid & (L2C - 1), seg = tid >> 5;
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int i = 0; i < 8; i += 2) {
            const float4 w0 = lds4(s_w1 + ((seg * 8 + i) * L2C + cc) * 4), x0 = lds4(s_hrow + seg * 32 + 4 * i);
            const float4 w1 = lds4(s_w1 + ((seg * 8 + i + 1) * L2C + cc) * 4), x1 = lds4(s_hrow + seg * 32 + 4 * i + 4);
            a0 = dot4(w0, x0, a0);
            a1 = dot4(w1, x1, a1);
          }
          s_p2[seg * L2C + cc] = a0 + a1;
        }
      } else {
        P16_MARK(26);  // (profile build: every marker of the role once per step)
      }
      P16_MARK(27);
      __syncthreads();
      if (tid < L2C) {  // the workgroup's 32 columns leave as ONE 128-byte store
        float o = 0.f;
        if (nxt) {
#pragma unroll
          for (int k = 0; k < PT / L2C; ++k) o += s_p2[k * L2C + tid];
          o = fmaxf(o, 0.f);
          o = (drop2 & 1u) ? 0.f : (d.dropout_mode ? 2.f * o : o);
        }
        put(g.rx + (slot + NB + rb) * PRENET + L2C * rk + (tid + oz), (value_bits(o) & 0x7fffffffu) | (nxt ? 0u : 0x80000000u));  // (x >= 0; a -0.0 must not read as "stopped")
      }
      if (ROLE_FIRST) h_dec_mfmas(IP{}, IR{});
    }
    accD += acc2;
    P16_MARK(16);
#ifdef XDTTS_P8_PROFILE
    s_ts[wave * 32 + 29] = (unsigned)wall_clock64();  // (the last marker of the last step, not summed)
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
  if (g.prof && tid < 32) g.prof[c * 32 + tid] = tid == 31 ? (u64)(s - step0) : (u64)s_ts[tid];  // sums of time stamps (mod 2^32)
#endif
}

}  // namespace

// The grid must be co-resident: one workgroup per CU on a 256-CU part, nothing else of ours running.
bool decoder_p16_supported(int device, int T) {
  if (T > TP) return false;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  if (prop.multiProcessorCount < P_NCU) return false;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_decoder_persistent16), PT, 0) != hipSuccess) return false;
  return per_cu >= 1;
}

void launch_decoder_p16(const DecoderBufs &d, const P8Weights &pw, const P8Bufs &g, int nsteps, hipStream_t s) {
  const void *fn = reinterpret_cast<const void *>(k_decoder_persistent16);
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
    static const int order[25] = {0, 1, 2, 3, 4, 5, 6, 20, 21, 22, 7, 8, 9, 10, 11, 12, 13, 14, 15, 24, 25, 26, 27, 16, 16};
    // every marker's slot holds the SUM of its time stamps over the steps: a phase = sum - sum of the marker before it
    unsigned prev = (unsigned)h[16] - (unsigned)h[29] + (unsigned)h[28];  // "marker before" the first one: the previous step's last, the loop start for step 0
    for (int i = 0; i < 24; ++i) {
      const int m = order[i];
      if ((((m >= 3 && m <= 4) || (m >= 20 && m <= 22)) && !attn) || (((m >= 12 && m <= 13) || (m >= 24 && m <= 27)) && !pre)) {
        printf("P16PROF %d %d %d %.3f\n", c, (int)steps, m, 0.0);
        continue;
      }
      printf("P16PROF %d %d %d %.3f\n", c, (int)steps, m, 0.01 * (double)((unsigned)h[m] - prev) / steps);
      prev = (unsigned)h[m];
    }
  }
  fflush(stdout);
#else
  COOP_CHECK(launch_coresident(true, fn, dim3(P_NCU), dim3(PT), 0, s, d, g, pw, nsteps));
#endif
}

}  // namespace xdtts
