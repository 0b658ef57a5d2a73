// decoder_persistent8.hip -- the Tacotron2 decoder loop (src/tacotron2/mod.rs:302-342) as ONE persistent, weight-stationary
// launch for lock-step batches of 3..8 chunks ("the batched / parallel sentences" of src/phonemes.rs:677-680 at the size a
// server with a handful of concurrent utterances has).
//
// Between the two engines that existed (decoder_persistent.hip: <= 2 chunks per launch, its per-chunk arithmetic on the VALU,
// 10.6 us per pair step; decoder.hip: two launches per lock-step iteration of up to 64 chunks, 26-30 us whatever the batch)
// a batch of 3..16 chunks paid either two pair launches one after the other or the whole latency chain of the big engine.
// This kernel keeps decoder_persistent.hip's skeleton -- 256 workgroups x 512 threads, one per CU; workgroup c owns the 16
// gate rows of attention-LSTM units 4c..4c+3 and of decoder-LSTM units 4c..4c+3 with their weights in REGISTERS for the
// whole loop; the state crosses CUs as data-tagged 8-byte granules {tag = step + 1, value}; roles per chunk on top of the
// LSTM slices -- and changes the two things that do not scale with the chunk count there:
//   * the LSTM pre-activations of ALL chunks are one v_mfma_f32_16x16x4_f32 stream per wave: the wave's 16 x (K/8) weight
//     slab is the A operand (136 VGPRs per lane, the same budget as the dot-product form), the chunks' state vectors in LDS
//     in [k/4][8 chunks][4] order are the B operand (one ds_read_b128 feeds four MFMAs; columns 8..15 of the tile carry
//     don't-care values that nothing reads), the 16 x 16 D tile holds unit u = lane / 16, chunk n = lane % 16, gates i,f,g,o
//     in a lane's four registers -- so the eight K-slices meet in LDS and wave 0 does every cell update in registers.
//     136 MFMAs per wave and step (1.8 us of a SIMD's matrix pipe, two waves per SIMD) whatever the number of chunks;
//   * the attention context crosses as a SIXTH edge (512 values per chunk, from the chunk's 8 attention workgroups) instead
//     of being folded into the encoder memory: the fold tables cost 20 registers or 16 kB of LDS per chunk.  In exchange the
//     partial energies only travel among a chunk's own 8 attention workgroups, which are the only ones that need the softmax.
// Per step:  x -> [attention LSTM] -> h_att -> [query, energies] -> e -> [softmax, context] -> ctx -> [decoder LSTM] -> h_dec
//            -> [projection rows] -> mel -> [stop rule, prenet] -> x(s+1)
// Only the columns of the newest vector are multiplied on the critical path (8 / 16 MFMAs per wave); the others are
// accumulated while the next vector's producers are busy.  Every spin is bounded and watches a global error word.
#include "device_utils.h"
#include "kernels.h"

namespace xdtts {

namespace {

typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PT = 512, NW = PT / 64, NB = P8_B_MAX, P_NCU = ATT_RNN / 4, TP = PERSIST_T_MAX;
constexpr int ATTN_CU = 8, PRE_CU = 16, EP_LD = TP, MEL_GL = 96, WPAD = TP + 32;
constexpr unsigned P_SPIN_LIMIT = 1u << 21, ACT_BIT = 0x80000000u;
static_assert(NB == 8 && ATT_RNN == DEC_RNN && P_NCU == 256 && (ATTN_CU + PRE_CU) * NB <= P_NCU, "role workgroups of 8 chunks fit the grid");

__device__ __forceinline__ void publish(u64 *slot, unsigned tag, float v) {
  __hip_atomic_store(slot, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 peek(const u64 *slot) { return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct PollCtl {
  int *err;
  unsigned limit;
};
__device__ __forceinline__ bool give_up(unsigned &spins, const PollCtl &pc) {
  if (++spins > pc.limit || ((spins & 127u) == 0 && __hip_atomic_load(pc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    atomicExch(pc.err, 1);
    return true;
  }
  __builtin_amdgcn_s_sleep(1);
  return false;
}
// N granules at base[idx + i * stride] (those of the bit mask `need`), all loads in flight together; every value is handed to
// sink(i, value, tag) the moment its tag matches -- nothing is kept in registers behind the loads themselves.  A timed-out slot
// is never delivered.  EVERY round issues all N loads (a granule that is not wanted, or has been delivered, is asked for again --
// or slot 0 in its place): with the loads themselves under per-lane conditions, lanes were handed the value of ANOTHER granule of
// the same round now and then (four neighbouring lanes = one 32-byte sector at a time, caught by comparing the LDS copy with the
// granule it came from: the chunk-1 value in chunk 0's place) -- the wait counts of a round assume its loads were all issued.
template <int N, class Sink>
__device__ __forceinline__ void gather(const u64 *base, unsigned idx, unsigned stride, unsigned want, unsigned need, const PollCtl &pc,
                                       Sink sink) {
  unsigned pending = need & ((1u << N) - 1u), spins = 0;
  while (pending) {
    u64 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = peek(base + (idx + (unsigned)(((need >> i) & 1u) ? i : 0) * stride));  // (see the note above)
#pragma unroll
    for (int i = 0; i < N; ++i)
      if ((pending >> i) & 1u) {
        const unsigned t = (unsigned)(v[i] >> 32);
        if ((t & ~ACT_BIT) == want) {
          sink(i, __uint_as_float((unsigned)v[i]), t);
          pending &= ~(1u << i);
        }
      }
    if (pending && give_up(spins, pc)) return;
  }
}
__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// index of element (k, chunk n) of a state vector kept in MFMA B-operand order [k/4][NB][4]
__device__ __forceinline__ int bidx(int k, int n) { return ((k >> 2) * NB + n) * 4 + (k & 3); }

struct P8Weights {
  const float4 *att_w, *dec_w, *q_w, *proj_w;
  const float *att_b, *dec_b, *v_w, *loc_fused, *proj_b, *pre0T, *pre1T;
};

// NQ b128 loads of the wave's slice of one state segment (seg: LDS base of the segment in B order; q0: the wave's first
// column quad-of-quads), 4 MFMAs each, into acc.  A[q] = the lane's four weights of columns 16 (q0 + q) + 4 kk .. + 3.
template <int NQ>
__device__ __forceinline__ void mfma_segment(f32x4 &acc, const float4 (&A)[NQ], const float *seg, int q0, int kk, int n) {
  // one B vector ahead of the MFMAs that consume it, and no further: left alone the scheduler hoists every load of a segment
  // (and of the next) above the first MFMA -- 64 registers that this kernel does not have (its weights went to scratch)
  const float *bp = seg + ((4 * q0 + kk) * NB + n) * 4;
  float4 b = lds4(bp);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float4 bn = q + 1 < NQ ? lds4(bp + (q + 1) * 4 * NB * 4) : b;
    asm volatile("" ::: "memory");
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q].w, b.w, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    b = bn;
  }
}

__global__ __launch_bounds__(PT) void k_decoder_persistent8(DecoderBufs d, P8Bufs g, P8Weights w, int nsteps) {
  // LDS (154 of 160 kB).  The state vectors of all chunks in MFMA B-operand order; x and ctx share a buffer and so do h_att
  // and h_dec: each is consumed (by the MFMAs that follow its gather) before the other is gathered, barriers in between.
  constexpr int ATTN_FLOATS = 2 * TP * 16 + 3 * TP + 16 + NW * 64 + 2 * WPAD + 62 * 16 + 16 + 8 * PT * 4;
  constexpr int PRE_FLOATS = N_MEL * PRENET + MEL_GL + 2 * PRENET + 8;
  __shared__ __attribute__((aligned(16))) float s_xc[EMB * NB], s_h[ATT_RNN * NB];
  __shared__ __attribute__((aligned(16))) float s_acc[NW * 64 * 4];  // the eight K-slices' partial D tiles
  __shared__ __attribute__((aligned(16))) float s_hrow[ATT_RNN];     // the role's own chunk, row-major: h_att (attention) / h_dec (projection)
  __shared__ __attribute__((aligned(16))) float s_crow[EMB];         // projection role: ctx of its chunk, row-major
  __shared__ __attribute__((aligned(16))) float s_role[ATTN_FLOATS > PRE_FLOATS ? ATTN_FLOATS : PRE_FLOATS];
  __shared__ int s_act[NB], s_alive[NB], s_err;
  float *const s_x = s_xc, *const s_ctx = s_xc, *const s_hatt = s_h, *const s_hdec = s_h;
  // attention role
  float *s_pm = s_role, *s_loc = s_pm + TP * 16, *s_aw = s_loc + TP * 16, *s_awc = s_aw + TP, *s_e = s_awc + TP, *s_q = s_e + TP,
        *s_part = s_q + 16, *s_wpad = s_part + NW * 64, *s_G = s_wpad + 2 * WPAD, *s_vv = s_G + 62 * 16,
        *s_qw = s_vv + 16;  // [8][PT] float4: query rows 16 rk + wave (+8), 4 x 16 B per lane each
  // projection + prenet role
  float *s_W0 = s_role, *s_mel = s_W0 + N_MEL * PRENET, *s_l1 = s_mel + MEL_GL, *s_pb = s_l1 + 2 * PRENET;  // s_W0 [20][256][4]

  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = d.B, T = d.T;
  const PollCtl pc{g.err, g.spins > 0 ? (unsigned)g.spins : P_SPIN_LIMIT};
  if (g.fault && c == g.fault - 1) return;  // test hook: this workgroup never shows up
  const int kk = lane >> 4, n16 = lane & 15, n = n16 & (NB - 1);  // MFMA lane coordinates: k-quad / chunk column

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
  const int prow = rk + 16 * wave;
  const bool prow_ok = pre && wave < 6 && prow <= N_MEL;
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
    const int k = i >> 3, b = i & (NB - 1);
    s_hatt[bidx(k, b)] = b < B ? d.att_h[0][b * ATT_RNN + k] : 0.f;
  }
#pragma unroll 1
  for (int i = tid; i < EMB * NB; i += PT) {
    const int k = i >> 3, b = i & (NB - 1);
    s_ctx[bidx(k, b)] = b < B ? d.ctx[b * EMB + k] : 0.f;
  }
  // attention role: processed memory of its 16 dims, the memory columns of its context slice (registers), location filter
  // role registers (one array, two uses): attention role [0, 16): memory[t = q + 8 j][64 rk + col], q = tid / 64, col = tid % 64
  //   (projection + prenet role: [0, 8) its two layer-2 columns, [8, 32) the wave's row of [W_p ; w_gate].  The roles are disjoint
  //   workgroups, so one array serves both.  The query rows and the prenet's layer-1 weights live in LDS.)
  float rreg[32];
  static_assert(TP / NW <= 16, "the attention role's memory columns fit their share of the role registers");
#pragma unroll
  for (int j = 0; j < 32; ++j) rreg[j] = 0.f;
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
      const int t = (tid >> 6) + NW * j;
      rreg[j] = t < T ? d.memory[((size_t)rb * T + t) * EMB + 64 * rk + (tid & 63)] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4 *>(s_qw + 4 * ((4 * r + j) * PT + tid)) = w.q_w[(unsigned)((16 * rk + wave + NW * r) * (ATT_RNN / 4) + lane + 64 * j)];
  }
  if (pre) {
    // layer 1 [in / 4][out][in % 4]: a thread's 40 weights are ten conflict-free 16-byte reads
#pragma unroll 1
    for (int i = tid; i < N_MEL * PRENET; i += PT) s_W0[(((i / PRENET) >> 2) * PRENET + i % PRENET) * 4 + ((i / PRENET) & 3)] = w.pre0T[i];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) rreg[4 * r + k] = w.pre1T[(unsigned)((lane + 64 * k) * PRENET + 16 * rk + wave + NW * r)];
    if (prow_ok) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float4 q = w.proj_w[(unsigned)(prow * (PROJ_IN / 4) + lane + 64 * j)];
        rreg[8 + 4 * j + 0] = q.x;
        rreg[8 + 4 * j + 1] = q.y;
        rreg[8 + 4 * j + 2] = q.z;
        rreg[8 + 4 * j + 3] = q.w;
      }
    }
  }
  int nf_r = pre ? d.nframes[rb] : 0;
  const int nv_r = attn ? d.n_valid[rb] : 0;
  bool ctx_valid = false;
  if (prow_ok && lane == 0) s_pb[wave] = w.proj_b[prow];  // (behind the s_W0 fill: s_pb follows it in the role area)
  const uint32_t item = d.item_base + (uint32_t)rb;
  __syncthreads();

  // location features of the NEXT step for the attention role's 16 dims (decoder_persistent.hip: a Toeplitz product on the matrix cores)
  auto location = [&]() {
#pragma unroll 1
    for (int i = tid; i < 2 * WPAD; i += PT) {
      const int ch = i / WPAD, t = i % WPAD - (LOC_K - 1) / 2;
      s_wpad[i] = (t >= 0 && t < T) ? (ch ? s_awc[t] : s_aw[t]) : 0.f;
    }
    __syncthreads();
    {
      const unsigned l = (unsigned)lane, li = l & 15u, lg = l >> 4, t0 = 16u * (unsigned)wave;
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
  float4 ax[2], ac[4], ah[8];   // attention LSTM: x 256 | ctx 512 | h_att 1024 columns, 1/8 of each
  float4 dh[8], dc[4], dd[8];   // decoder LSTM:   h_att 1024 | ctx 512 | h_dec 1024
  {
    const float4 *ra = w.att_w + (size_t)(16 * c + n16) * (ATT_COLS / 4), *rd = w.dec_w + (size_t)(16 * c + n16) * (DEC_COLS / 4);
#pragma unroll
    for (int q = 0; q < 2; ++q) ax[q] = ld_stream(ra + (0 + 32 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 4; ++q) ac[q] = ld_stream(ra + (PRENET + 64 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 8; ++q) ah[q] = ld_stream(ra + (ATT_IN + 128 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 8; ++q) dh[q] = ld_stream(rd + (0 + 128 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 4; ++q) dc[q] = ld_stream(rd + (ATT_RNN + 64 * wave) / 4 + 4 * q + kk);
#pragma unroll
    for (int q = 0; q < 8; ++q) dd[q] = ld_stream(rd + (DEC_IN + 128 * wave) / 4 + 4 * q + kk);
  }
  // the partial pre-activations that do not depend on the newest vector
  f32x4 accA = (f32x4){0.f, 0.f, 0.f, 0.f}, accD = accA;
  mfma_segment<4>(accA, ac, s_ctx, 4 * wave, kk, n);    // attention LSTM: ctx(s-1) ...
  mfma_segment<8>(accA, ah, s_hatt, 8 * wave, kk, n);   // ... and h_att(s-1)
  __syncthreads();
#pragma unroll 1
  for (int i = tid; i < DEC_RNN * NB; i += PT) {
    const int k = i >> 3, b = i & (NB - 1);
    s_hdec[bidx(k, b)] = b < B ? d.dec_h[0][b * DEC_RNN + k] : 0.f;
  }
  __syncthreads();
  mfma_segment<8>(accD, dd, s_hdec, 8 * wave, kk, n);   // decoder LSTM: h_dec(s-1)
  __syncthreads();  // (x(s) is gathered into the buffer ctx(s-1) was read from)

  // wave 0: sum of the eight K-slices of a D tile (the caller has put a barrier behind the s_acc stores)
  auto reduce_tile = [&]() {
    f32x4 gsum = *reinterpret_cast<const f32x4 *>(s_acc + lane * 4);
#pragma unroll
    for (int q = 1; q < NW; ++q) gsum += *reinterpret_cast<const f32x4 *>(s_acc + (q * 64 + lane) * 4);
    return gsum;
  };

  int s = step0;
  const int s_stop = step0 + nsteps;
  float cown = 0.f;  // attention role, tid < 64: the chunk's context column 64 rk + tid of the last step (write-back)
  for (; s < s_stop; ++s) {
    const int p = s & 1;
    const unsigned want = (unsigned)(s + 1);
    // ---- P1: x(s) and the chunks' active bits ---------------------------------------------------------------------------------
    {
      const int i = tid & 255, b0 = tid >> 8;  // chunks b0, b0 + 2, b0 + 4, b0 + 6
      unsigned need = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) need |= (s_alive[b0 + 2 * j] != 0 ? 1u : 0u) << j;
      gather<4>(g.x, (unsigned)((p * NB + b0) * PRENET + i), 2u * PRENET, want, need, pc, [&](int j, float v, unsigned tg) {
        s_x[bidx(i, b0 + 2 * j)] = v;
        if (i == 0) s_act[b0 + 2 * j] = (tg & ACT_BIT) ? 1 : 0;
      });
      if (tid == PT - 1) s_err = __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    unsigned actm = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) actm |= (s_act[b] != 0 ? 1u : 0u) << b;
    if (!actm || s_err != 0) break;  // every chunk has stopped (or an exchange failed): the launch ends by itself
    const bool act_r = (actm >> rb) & 1u;
    const bool act_n = (actm >> n16) & 1u;  // (lanes n16 >= 8: false)
    // attention LSTM: close the rows with the x columns
    mfma_segment<2>(accA, ax, s_x, 2 * wave, kk, n);
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
        publish(g.hatt + (unsigned)((p * NB + n16) * ATT_RNN + 4 * c + cu), want, h_att_last);
      }
    }
    accA = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- P2: h_att(s) of every active chunk ------------------------------------------------------------------------------------
    // attention role: what the energies need besides the query is in registers before h_att arrives
    // (and its two query rows, 8 kB per wave from L2, are requested ahead of the gather they follow)
    float4 lp4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = lp4;
    if (attn && act_r) {
      const float4 l4 = lds4(s_loc + 4 * tid), p4 = lds4(s_pm + 4 * tid);
      lp4 = make_float4(l4.x + p4.x, l4.y + p4.y, l4.z + p4.z, l4.w + p4.w);
      v4 = lds4(s_vv + 4 * (tid & 3));
    }
    // granule tid + 512 i: i = 2 b + half of the vector
    unsigned need2 = 0u;
#pragma unroll
    for (int i = 0; i < 2 * NB; ++i) need2 |= ((actm >> (i >> 1)) & 1u) << i;
    gather<2 * NB>(g.hatt, (unsigned)(p * NB * ATT_RNN + tid), PT, want, need2, pc, [&](int i, float v, unsigned) {
      const int k = tid + PT * (i & 1), b = i >> 1;
      s_hatt[bidx(k, b)] = v;
      if (attn && b == rb) s_hrow[k] = v;
    });
    __syncthreads();
    if (attn && act_r) {
      // query rows 16 rk + wave (+8) (re-read from L2: 8 kB per wave and step), then this workgroup's share of the energies
      float q0 = 0.f, q1 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 hv = lds4(s_hrow + 256 * j + 4 * lane);
        q0 = dot4(lds4(s_qw + 4 * (j * PT + tid)), hv, q0);
        q1 = dot4(lds4(s_qw + 4 * ((4 + j) * PT + tid)), hv, q1);
      }
      q0 = wave_sum(q0);
      q1 = wave_sum(q1);
      if (lane == 0) {
        s_q[wave] = q0;
        s_q[wave + NW] = q1;
      }
      __syncthreads();
      const int t = tid >> 2, dq = 4 * (tid & 3);
      const float4 q4 = lds4(s_q + dq);
      float e = v4.x * fast_tanh(q4.x + lp4.x);
      e = fmaf(v4.y, fast_tanh(q4.y + lp4.y), e);
      e = fmaf(v4.z, fast_tanh(q4.z + lp4.z), e);
      e = fmaf(v4.w, fast_tanh(q4.w + lp4.w), e);
      e += dpp_move<0xB1, 0xf>(0.f, e);  // quad_perm:[1,0,3,2]
      e += dpp_move<0x4E, 0xf>(0.f, e);  // quad_perm:[2,3,0,1]
      if ((tid & 3) == 0 && t < T) publish(g.ep + (unsigned)(((p * NB + rb) * ATTN_CU + rk) * EP_LD + t), want, e);
    }
    // both LSTMs: the h_att(s) columns (decoder LSTM of this step, attention LSTM of the next)
    mfma_segment<8>(accD, dh, s_hatt, 8 * wave, kk, n);
    mfma_segment<8>(accA, ah, s_hatt, 8 * wave, kk, n);
    // ---- P3 (attention role): the 8 partial-energy rows of the chunk -> softmax -> this workgroup's 64 context columns ------------
    if (attn && act_r) {
      {
        const int t = tid >> 2, j = tid & 3;
        float ev[2] = {0.f, 0.f};
        gather<2>(g.ep, (unsigned)(((p * NB + rb) * ATTN_CU + j) * EP_LD + t), 4u * EP_LD, want, t < T ? 3u : 0u, pc,
                  [&](int i, float v, unsigned) { ev[i] = v; });
        float e = ev[0] + ev[1];
        e += dpp_move<0xB1, 0xf>(0.f, e);
        e += dpp_move<0x4E, 0xf>(0.f, e);
        if (j == 0) s_e[t] = (t < T && t < nv_r) ? e : -INFINITY;  // mask, mod.rs:219-220
      }
      __syncthreads();
      {  // every wave: the softmax in registers, lane <-> steps lane, lane + 64
        const float e0 = s_e[lane], e1 = s_e[lane + 64];
        const float m = wave_max(fmaxf(e0, e1));
        const float x0 = fast_exp(e0 - m), x1 = fast_exp(e1 - m);
        const float rs = __builtin_amdgcn_rcpf(wave_sum(x0 + x1));
        const float w0 = x0 * rs, w1 = x1 * rs;
        // context columns: this wave sums the steps t = wave + 8 j; lane = column.  The weight of step t sits in lane t % 64
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < TP / NW; ++j) {
          const int t = wave + NW * j;
          const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t < 64 ? w0 : w1), t & 63));
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
      __syncthreads();
      if (tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < NW; ++u) v += s_part[u * 64 + tid];
        cown = v;
        ctx_valid = true;
        publish(g.ctx + (unsigned)((p * NB + rb) * EMB + 64 * rk + tid), want, v);
      }
    }
    // ---- P4: ctx(s) of every active chunk -> decoder LSTM ------------------------------------------------------------------------
    gather<NB>(g.ctx, (unsigned)(p * NB * EMB + tid), EMB, want, actm, pc, [&](int i, float v, unsigned) {
      s_ctx[bidx(tid, i)] = v;
      if (pre && i == rb) s_crow[tid] = v;
    });
    __syncthreads();
    mfma_segment<4>(accD, dc, s_ctx, 4 * wave, kk, n);
    *reinterpret_cast<f32x4 *>(s_acc + (wave * 64 + lane) * 4) = accD;
    __syncthreads();
    if (wave == 0) {
      const f32x4 gs = reduce_tile();
      if (cell && act_n) {
        const float ig = fast_sigmoid(gs[0] + bias_d.x), fg = fast_sigmoid(gs[1] + bias_d.y), gg = fast_tanh(gs[2] + bias_d.z),
                    og = fast_sigmoid(gs[3] + bias_d.w);
        c_dec = fmaf(fg, c_dec, ig * gg);
        h_dec_last = og * fast_tanh(c_dec);
        publish(g.hdec + (unsigned)((p * NB + n16) * DEC_RNN + 4 * c + cu), want, h_dec_last);
      }
    }
    accD = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_segment<4>(accA, ac, s_ctx, 4 * wave, kk, n);  // attention LSTM of the next step: ctx(s)
    if (attn && act_r) location();                       // ... and its location features
    // ---- P5: h_dec(s) -> projection rows ---------------------------------------------------------------------------------------
    // projection + prenet role: its row of [W_p ; w_gate] is requested ahead of the gather, the Bernoulli(0.5) masks of step
    // s + 1 (they do not depend on the data) are hashed in the time the first poll of h_dec could not succeed anyway
    unsigned drop1 = 0u, drop2 = 0u;
    if (pre && act_r) {
      if (d.dropout_mode) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          drop1 |= (prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 0, lane + 64 * k) ? 1u : 0u) << k;
#pragma unroll
        for (int r = 0; r < 2; ++r)
          drop2 |= (prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 1, 16 * rk + wave + NW * r) ? 1u : 0u) << r;
      }
    }
    gather<2 * NB>(g.hdec, (unsigned)(p * NB * DEC_RNN + tid), PT, want, need2, pc, [&](int i, float v, unsigned) {
      const int k = tid + PT * (i & 1), b = i >> 1;
      s_hdec[bidx(k, b)] = v;
      if (pre && b == rb) s_hrow[k] = v;
    });
    __syncthreads();
    if (prow_ok && act_r) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        a = dot4(make_float4(rreg[8 + 4 * j], rreg[9 + 4 * j], rreg[10 + 4 * j], rreg[11 + 4 * j]), lds4(s_hrow + 256 * j + 4 * lane), a);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        a = dot4(make_float4(rreg[24 + 4 * j], rreg[25 + 4 * j], rreg[26 + 4 * j], rreg[27 + 4 * j]), lds4(s_crow + 256 * j + 4 * lane), a);
      a = wave_sum(a);
      if (lane == 0) publish(g.mel + (unsigned)((p * NB + rb) * MEL_GL + prow), want, a + s_pb[wave]);
    }
    mfma_segment<8>(accD, dd, s_hdec, 8 * wave, kk, n);  // decoder LSTM of the next step: h_dec(s)
    // ---- P6 (projection + prenet role): frame s, stop rule, x(s+1) ---------------------------------------------------------------
    if (pre && act_r) {  // a chunk's last x (active bit clear) is published at the step it stops
      if (tid < N_MEL + 1) {
        s_mel[tid] = 0.f;
        gather<1>(g.mel, (unsigned)((p * NB + rb) * MEL_GL + tid), 0, want, 1u, pc, [&](int, float v, unsigned) { s_mel[tid] = v; });
      }
      __syncthreads();
      const float gate = s_mel[N_MEL];
      const bool fired = d.use_gate && gate_sigmoid(gate) > d.gate_threshold;  // mod.rs:319-324
      if (rk == 0) {
        if (tid < N_MEL) d.frames[((size_t)rb * d.max_steps + s) * N_MEL + tid] = s_mel[tid];
        if (tid == 0) {
          d.gates[(size_t)rb * d.max_steps + s] = gate;
          if (fired) d.nframes[rb] = s + 1;  // the tripping frame is kept
        }
      }
      if (fired) nf_r = s + 1;
      const bool nxt = s + 1 < nf_r;
      float xo[2] = {0.f, 0.f};
      if (nxt) {
        const unsigned HM = (unsigned)((tid >> 8) * (N_MEL / 2));
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < N_MEL / 2; k += 4) {
          const float4 w4 = lds4(s_W0 + 4u * (((HM + k) >> 2) * PRENET + ((unsigned)tid & 255u))), m = lds4(s_mel + HM + k);
          acc = fmaf(w4.x, m.x, acc);
          acc = fmaf(w4.y, m.y, acc);
          acc = fmaf(w4.z, m.z, acc);
          acc = fmaf(w4.w, m.w, acc);
        }
        s_l1[tid] = acc;
        __syncthreads();
        float pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = fmaxf(s_l1[lane + 64 * k] + s_l1[PRENET + lane + 64 * k], 0.f);
          pk[k] = (drop1 >> k) & 1u ? 0.f : (d.dropout_mode ? 2.f * v : v);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) a = fmaf(rreg[4 * r + k], pk[k], a);
          a = fmaxf(wave_sum(a), 0.f);
          xo[r] = (drop2 >> r) & 1u ? 0.f : (d.dropout_mode ? 2.f * a : a);
        }
      }
      // the workgroup's 16 columns leave as ONE 128-byte store (decoder_persistent.hip)
      if (lane < 2) s_mel[MEL_GL - 16 + wave + NW * lane] = lane ? xo[1] : xo[0];  // s_mel[81..95] is unused padding
      __syncthreads();
      if (tid < 16)
        publish(g.x + (unsigned)(((p ^ 1) * NB + rb) * PRENET + 16 * rk + tid), (want + 1u) | (nxt ? ACT_BIT : 0u), s_mel[MEL_GL - 16 + tid]);
    }
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
}

// x(0) = prenet(0) = 0 (the prenet has no bias, mod.rs:208) with the chunks' initial active bits
__global__ void k_p8_seed(P8Bufs g, const int *limits) {
  const int b = blockIdx.x, i = threadIdx.x;
  publish(g.x + (size_t)b * PRENET + i, 1u | (limits[b] > 0 ? ACT_BIT : 0u), 0.f);
}
// parity hook: a sequence that starts at `step` with the prenet output x [B][256] already computed (d.x)
__global__ void k_p8_seed_at(P8Bufs g, const int *limits, const float *x, int step) {
  const int b = blockIdx.x, i = threadIdx.x;
  publish(g.x + ((size_t)(step & 1) * NB + b) * PRENET + i, (unsigned)(step + 1) | (limits[b] > step ? ACT_BIT : 0u), x[b * PRENET + i]);
}

}  // namespace

size_t p8_granule_words() { return (size_t)2 * NB * (PRENET + ATT_RNN + ATTN_CU * EP_LD + EMB + DEC_RNN + MEL_GL); }

P8Bufs p8_bufs(unsigned long long *base, int *err) {
  P8Bufs g{};
  g.x = base;
  g.hatt = g.x + (size_t)2 * NB * PRENET;
  g.ep = g.hatt + (size_t)2 * NB * ATT_RNN;
  g.ctx = g.ep + (size_t)2 * NB * ATTN_CU * EP_LD;
  g.hdec = g.ctx + (size_t)2 * NB * EMB;
  g.mel = g.hdec + (size_t)2 * NB * DEC_RNN;
  g.err = err;
  return g;
}

// The grid must be co-resident: one workgroup per CU on a 256-CU part, nothing else of ours running.
bool decoder_p8_supported(int device, int B, int T) {
  if (B < 1 || B > NB || T > TP) return false;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  if (prop.multiProcessorCount < P_NCU) return false;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_decoder_persistent8, PT, 0) != hipSuccess) return false;
  return per_cu >= 1;
}

void launch_p8_seed(const DecoderBufs &d, const P8Bufs &g, const int *limits_dev, hipStream_t s) {
  HIP_CHECK(hipMemsetAsync(g.x, 0, p8_granule_words() * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_p8_seed, dim3(d.B), dim3(PRENET), 0, s, g, limits_dev);
  HIP_CHECK(hipGetLastError());
}

void launch_p8_seed_at(const DecoderBufs &d, const P8Bufs &g, const int *limits_dev, int step, hipStream_t s) {
  HIP_CHECK(hipMemsetAsync(g.x, 0, p8_granule_words() * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_p8_seed_at, dim3(d.B), dim3(PRENET), 0, s, g, limits_dev, d.x, step);
  HIP_CHECK(hipGetLastError());
}

void launch_decoder_p8(const DecoderBufs &d, const DeviceWeights &w, const P8Bufs &g, int nsteps, hipStream_t s) {
  if (d.B < 1 || d.B > NB || d.T > TP) fail(XDTTS_ERR_BAD_ARG, "persistent MFMA decoder: %d chunks of %d encoder steps (max %d, %d)", d.B, d.T, NB, TP);
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
  COOP_CHECK(launch_coresident(true, reinterpret_cast<const void *>(k_decoder_persistent8), dim3(P_NCU), dim3(PT), 0, s, d, g, pw, nsteps));
}

}  // namespace xdtts
