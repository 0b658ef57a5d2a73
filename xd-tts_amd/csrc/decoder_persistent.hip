// decoder_persistent.hip -- the Tacotron2 decoder loop (src/tacotron2/mod.rs:302-342) as ONE
// persistent, weight-stationary launch for small lock-step batches (B <= 4 chunks).
//
// The launch-per-stage path (decoder.hip) streams the 71.3 MB of LSTM weights from HBM every step
// (~11.5 us at the achievable bandwidth) and pays five grid boundaries (~8 us).  Here the weights
// never move: 256 workgroups, one per CU, 512 threads each; workgroup c keeps the 16 gate rows of
// attention-LSTM units 4c..4c+3 and of decoder-LSTM units 4c..4c+3 in its register file for the
// whole utterance (8 waves x 64 lanes x 136 VGPRs = 272 KB per CU), wave w owning gate rows w and
// w + 8 of each (16 waves x 68 VGPRs leaves too few working registers under the 128-VGPR cap).
// What crosses CUs per step is only the state, FIVE all-gather edges:
//     x (256 values) -> all      h_att (1024) -> all      partial energies (8 x T) -> all
//     h_dec (1024) -> all        mel + gate (81) -> 16
// carried by data-tagged 8-byte granules {tag = step + 1, value} (one relaxed agent-scope store per
// value; readers re-read until the tag matches -- MI355X_MICROARCH.md hand-off recipe R2, the
// scheme the encoder BiLSTM already uses): no flags, no fences, placement-independent.  Two slots
// per value by step parity; a slot is rewritten two steps later, after every reader has passed an
// all-to-all dependency on its producer.  tools/ubench_edges.hip measures a six-edge skeleton (edges
// only): 10.8 us per step = 1.8 us per edge.
// Neither the attention CONTEXT nor the attention WEIGHTS cross (round 1 had a sixth edge for the context):
//   * every consumer of the context is linear in it, so each workgroup folds its own context columns into the
//     encoder memory once per launch -- P[row][t] = W[row][ctx cols] . memory[t], lane <-> steps lane, lane + 64 --
//     and takes  sum_t w_t P[row][t]  instead of  W[row][ctx cols] . (sum_t w_t memory[t]);
//   * the weights w are a 100-element softmax that EVERY workgroup computes for itself from the partial energies
//     (it lands in exactly the registers the folded products need), which is cheaper than an edge to broadcast them.
//
// Roles on top of the LSTM slices (disjoint workgroups, per chunk b):
//   attention, 8 per chunk: 16 of the 128 attention dims each -- query rows and [T][16] of processed_memory in
//     LDS; partial energies -> (edge, to everybody); afterwards, off the critical path, the NEXT step's location
//     features for its dims with the conv(2->32,k31) and dense(32->128) folded into one 62-tap filter per dim;
//     the context itself is formed once, at the end of the launch, for the state that is written back;
//   projection + prenet, 16 per chunk: 5-6 rows of [W_p ; w_gate] -> mel -> (edge among the 16)
//     -> frame store, gate, stop rule (mod.rs:319-324), prenet layer 1 (recomputed by all 16, W0 in
//     LDS), 16 layer-2 columns -> x(s+1).  The x granules carry the chunk's "still active" bit, so
//     every workgroup learns of a stop with the data it waits for anyway, and the launch ends by
//     itself when no chunk is active.
// Only the pieces that depend on the newest vector sit on the critical path (x for the attention LSTM, the
// softmax for the decoder LSTM); the other column blocks are accumulated per lane while the producers of the
// next vector are busy, and one DPP wave reduction closes each row.
//
// Every spin is bounded and watches a global error word: a lost workgroup (grid not co-resident)
// drains the whole launch in microseconds and surfaces as XDTTS_ERR_HIP on the host.
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace xdtts {

namespace {

typedef unsigned long long u64;

constexpr int PT = 512;                  // threads per workgroup (8 waves, 256 VGPRs each)
constexpr int NW = PT / 64;
constexpr int P_NCU = ATT_RNN / 4;       // 256 workgroups = LSTM slices
constexpr int TP = PERSIST_T_MAX;        // encoder-step capacity of the attention role
constexpr int ATTN_CU = 8, PRE_CU = 16;  // role workgroups per chunk
constexpr int EP_LD = TP, MEL_GL = 96;
#ifndef XDTTS_MEL_ST
#define XDTTS_MEL_ST 1
#endif
constexpr int MEL_ST = XDTTS_MEL_ST;     // granules between two mel values in the exchange (16 = one 128-byte line each: measured, no gain over 1)
constexpr int GS = PERSIST_B_MAX;        // chunk stride of the granule arrays: the same for every launch width, so a
                                         // 1-chunk launch can continue a sequence that a 2-chunk launch began
constexpr unsigned P_SPIN_LIMIT = 1u << 21;
constexpr unsigned ACT_BIT = 0x80000000u;
constexpr int PERSIST_LAZY_DEFAULT = 9;  // x 256 clocks: ~0.95 us (6 / 7 / 8 / 9 / 10 -> 8.53 / 8.45 / 8.38 / 8.26 / 8.35 us per 1-chunk step; 0: +1.3 us)

constexpr int WPAD = TP + 32;            // zero-padded attention-weight window, index t + 15
static_assert(ATT_RNN == DEC_RNN && P_NCU == 256, "one workgroup per 4 + 4 hidden units");

__device__ __forceinline__ void publish(u64 *slot, unsigned tag, float v) {
  __hip_atomic_store(slot, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 peek(const u64 *slot) {
  return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
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
// N granules at base[idx + i * stride], all loads in flight together; `base` is uniform and the
// offsets are 32-bit so the loads use the SGPR-base addressing form (no 64-bit VGPR addresses kept
// live across the step loop).  A timed-out slot reads as {tag 0, 0.0f}.
// PRE: the first poll of every granule was issued earlier (pre[i], by the previous phase of the skewed pair loop); a value
// whose tag is not this step's -- not there yet, or nothing was issued -- is polled as usual.
template <int N, bool PRE = false>
__device__ __forceinline__ unsigned gather(const u64 *base, unsigned idx, unsigned stride, unsigned want,
                                           const bool (&need)[N], float (&out)[N], unsigned (&tag)[N], const PollCtl &err,
                                           const u64 *pre = nullptr) {
  bool done[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    done[i] = !need[i];
    out[i] = 0.f;
    tag[i] = 0u;
  }
  if (PRE) {
    bool all = true;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (!done[i]) {
        const unsigned t = (unsigned)(pre[i] >> 32);
        if ((t & ~ACT_BIT) == want) {
          out[i] = __uint_as_float((unsigned)pre[i]);
          tag[i] = t;
          done[i] = true;
        } else {
          all = false;
        }
      }
    if (all) return 0u;
  }
  unsigned spins = 0;
  for (;;) {
    u64 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (!done[i]) v[i] = peek(base + (idx + (unsigned)i * stride));
    bool all = true;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (!done[i]) {
        const unsigned t = (unsigned)(v[i] >> 32);
        if ((t & ~ACT_BIT) == want) {
          out[i] = __uint_as_float((unsigned)v[i]);
          tag[i] = t;
          done[i] = true;
        } else {
          all = false;
        }
      }
    if (all || give_up(spins, err)) return spins;
  }
}

// Workgroups that consume a vector only for off-critical-path work start polling it late (n x 256
// clocks): their polls would otherwise sit in the memory queues the critical consumers wait on.
__device__ __forceinline__ void lazy_wait(int n) {  // n x 256 clocks
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
}
// test hook (PersistBufs::slow): this workgroup is a straggler at point `at` of step s when at == s mod 6
__device__ __forceinline__ void straggle(bool me, int s, int at) {
  if (me && s % 6 == at)
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);  // 2 x 8128 clocks ~ 7 us
}
__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// Hides a thread-index expression's known bits from the optimiser.  Without this `idx + CONST`
// is canonicalised to `idx | CONST` wherever the bits are disjoint, the constant no longer folds
// into the ds instruction's offset field, and every unrolled access gets its own address register,
// hoisted out of the step loop and spilled (measured: ~100 scratch reloads per step).
// Keeps the per-chunk bodies of an unrolled chunk loop in program order: without it the LDS loads
// of every chunk are hoisted ahead of the first FMA and the live set grows by ~40 VGPRs per chunk.
__device__ __forceinline__ void chunk_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

constexpr int persist_lds_floats(int pb) {
  const int common = pb * (PRENET + TP + ATT_RNN + DEC_RNN + 16) + 8 + 32 + 16 * pb;
  const int attn = 2 * TP * 16 + 2 * TP + 16 + NW * 64 + 2 * WPAD + 62 * 16 + 64 + 16 + 8 * PT * 4;
  const int pre = N_MEL * PRENET + MEL_GL + 2 * PRENET + 8 + 6 * PT * 4;
  return common + (attn > pre ? attn : pre);
}

// Developer build (-DXDTTS_PERSIST_PROFILE): thread 0 of every workgroup accumulates the 100 MHz
// wall clock between phase markers into g.prof[workgroup][24] (see tools/persist_profile.py).
#ifdef XDTTS_PERSIST_PROFILE
#define PROF_MARK(i)                                       \
  do {                                                     \
    if (tid == 0) {                                        \
      const u64 now_ = wall_clock64();                     \
      s_prof[i] += now_ - prof_last;                       \
      prof_last = now_;                                    \
    }                                                      \
  } while (0)
// failed poll rounds of a gather (the slowest lane of each wave, summed over the 8 waves), into s_prof[16 + e]
#define PROF_POLLS(e, n)                                                   \
  do {                                                                     \
    const float m_ = wave_max((float)(n));                                 \
    if ((tid & 63) == 0) atomicAdd(&s_prof[16 + (e)], (u64)m_);            \
  } while (0)
// sum over the steps of the wall clock at an event, into s_prof[20 + e]: the host turns the sums into each workgroup's
// mean lateness at that event relative to the average workgroup
#define PROF_WHEN(e)                                       \
  do {                                                     \
    if (tid == 0) s_prof[20 + (e)] += wall_clock64();      \
  } while (0)
#else
#define PROF_MARK(i) do { } while (0)
#define PROF_POLLS(e, n) do { (void)(n); } while (0)
#define PROF_WHEN(e) do { } while (0)
#endif

struct PersistWeights {
  const float4 *att_w, *dec_w, *q_w, *proj_w;
  const float *att_b, *dec_b, *v_w, *loc_fused, *proj_b, *pre0T, *pre1T;
};

// SKEW (PB = 2 only): the two chunks do not run their steps in lock-step but two phases apart, a workgroup alternating
// between them -- while one chunk's vector crosses the chip the workgroup runs the other chunk's phase instead of sleeping
// (see the loop at "skewed pair").
// GATE: the stop rule decides (mod.rs:319-324; the reference's mode) -- its verdict through gate_fires, and a 1-chunk launch returns at
// once for a chunk that has stopped.  The gate-less instantiations (fixed frame counts: parity hooks, benchmarks; d.use_gate = 0)
// keep the runtime form they were tuned with: with gate_fires compiled in, the hot loop of a gate-less decode was 0.1 us per step
// slower, and with no stop-rule code at all the skewed pair loop compiled into something that ran 49 us per step.
template <int PB, bool SKEW = false, bool GATE = false>
__global__ __launch_bounds__(PT) void k_decoder_persistent(DecoderBufs d, PersistBufs g, PersistWeights w, int nsteps) {
  // static LDS: with compile-time addresses the per-access offsets fold into the ds instructions
  // (a dynamic base made the compiler keep ~100 hoisted addresses live across the step loop)
  __shared__ __attribute__((aligned(16))) float smem[persist_lds_floats(PB)];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = d.T;
  const PollCtl pc{g.err, g.spins > 0 ? (unsigned)g.spins : P_SPIN_LIMIT};
  if (g.fault && c == g.fault - 1) return;  // developer fault injection: this workgroup never shows up
  const unsigned L4 = opaque(4u * (unsigned)lane);
  // ---- LDS: state vectors of all chunks, then the role's working set -------------------------
  float *s_x = smem;                     // [PB][256]
  float *s_e = s_x + PB * PRENET;        // [PB][TP]  masked energies of the current step (-inf from T / n_valid on)
  float *s_hatt = s_e + PB * TP;         // [PB][1024]
  float *s_hdec = s_hatt + PB * ATT_RNN; // [PB][1024]
  float *s_g = s_hdec + PB * DEC_RNN;    // [PB][16] gate pre-activations of this workgroup's rows
  int *s_act = reinterpret_cast<int *>(s_g + PB * 16);  // [0..1] active at this step, [2..3] alive: not yet seen inactive, [4] error word
  float *s_bias = s_g + PB * 16 + 8;    // [2][16] b_ih + b_hh of this workgroup's attention / decoder rows
  float *s_cell = s_bias + 32;           // [4][4 PB] att_c, dec_c, h_att, h_dec of the (chunk, unit) cell threads
  float *role = s_cell + 16 * PB;
  // attention role
  float *s_pm = role;                    // [TP][16]  processed_memory, own dims
  float *s_loc = s_pm + TP * 16;         // [TP][16]  location features, own dims
  float *s_aw = s_loc + TP * 16;         // [TP]
  float *s_awc = s_aw + TP;              // [TP]
  float *s_q = s_awc + TP;               // [16]
  float *s_part = s_q + 16;              // [NW][64]
  float *s_wpad = s_part + NW * 64;      // [2][WPAD]
  float *s_G = s_wpad + 2 * WPAD;        // [62][16]  fused location filter, own dims
  float *s_cown = s_G + 62 * 16;         // [64]      own context columns (write-back only)
  float *s_vv = s_cown + 64;             // [16]      v, own dims
  float *s_qw = s_vv + 16;               // [8][PT] float4: query rows 16 rk + wave (+8), 4 x 16 B per lane each
  // projection + prenet role
  float *s_W0 = role;                    // [20][256][4]
  float *s_mel = s_W0 + N_MEL * PRENET;  // [96]
  float *s_l1 = s_mel + MEL_GL;          // [2][256]
  float *s_pb = s_l1 + 2 * PRENET;           // [8] projection biases of this workgroup's rows (by wave)
  float *s_pw = s_pb + 8;                // [6][PT] float4: row rk + 16 wave of [W_p ; w_gate] (waves 0..5)

  // ---- resident weights ----------------------------------------------------------------------
  // packed [unit][gate] order: row 16c + r is unit 4c + r/4, gate r%4; wave w owns r = w and w + 8
  float4 wa[2][ATT_COLS / 256], wd[2][DEC_COLS / 256];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = 16 * c + wave + NW * r;
#pragma unroll
    for (int k = 0; k < ATT_COLS / 256; ++k) wa[r][k] = ld_stream(w.att_w + (size_t)row * (ATT_COLS / 4) + lane + 64 * k);
#pragma unroll
    for (int k = 0; k < DEC_COLS / 256; ++k) wd[r][k] = ld_stream(w.dec_w + (size_t)row * (DEC_COLS / 4) + lane + 64 * k);
    if (lane == 0) {
      s_bias[wave + NW * r] = w.att_b[row];
      s_bias[16 + wave + NW * r] = w.dec_b[row];
    }
  }
  const bool attn = c < ATTN_CU * PB, pre = !attn && c < (ATTN_CU + PRE_CU) * PB;
  const int rb = attn ? c / ATTN_CU : (pre ? (c - ATTN_CU * PB) / PRE_CU : 0);  // the role's chunk
  const int rk = attn ? c % ATTN_CU : (c - ATTN_CU * PB) % PRE_CU;              // slice within the role
  // The roles' own weights (query rows, projection rows, prenet layer-2 columns) are NOT kept in
  // registers: they are re-read from L2 every step, issued before the gather they follow, so the
  // 136 LSTM registers are the only long-lived ones.
  const int prow = rk + 16 * wave;
  const bool prow_ok = pre && wave < 6 && prow <= N_MEL;

  // ---- state of the sequence so far (zeros at step 0; a previous launch's write-back otherwise) ----
  const int step0 = d.ctl[0];
  // A 1-chunk launch on a chunk that has already stopped (the host enqueues the continuation of BOTH chunks of a pair behind the
  // pair's launch instead of asking which one survived): nothing to do, nothing to write back -- grid-uniform, ahead of the set-up.
  if (GATE && PB == 1 && !(step0 < d.nframes[0])) return;
  if (tid < PB) {
    s_act[tid] = 0;
    s_act[2 + tid] = step0 < d.nframes[tid];  // a stopped chunk is never polled again
  }
#pragma unroll
  for (int b = 0; b < PB; ++b) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      s_hatt[b * ATT_RNN + tid + PT * h] = d.att_h[0][b * ATT_RNN + tid + PT * h];
      s_hdec[b * DEC_RNN + tid + PT * h] = d.dec_h[0][b * DEC_RNN + tid + PT * h];
    }
  }
  // attention weights of the chunks, lane <-> steps lane and lane + 64 (zero from T on): the weights of step s-1
  // stand for ctx(s-1) at the start, every step's softmax replaces them
  float wreg[PB][2];
  int nv[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    wreg[b][0] = lane < T ? d.aw[b * T + lane] : 0.f;
    wreg[b][1] = lane + 64 < T ? d.aw[b * T + lane + 64] : 0.f;
    nv[b] = d.n_valid[b];
  }
  const int cb = tid >> 2, cu = tid & 3;  // cell-update threads: tid < 4 PB -> (chunk, unit)
  const bool cell = tid < 4 * PB;
  if (cell) {
    s_cell[tid] = d.att_c[cb * ATT_RNN + 4 * c + cu];
    s_cell[4 * PB + tid] = d.dec_c[cb * DEC_RNN + 4 * c + cu];
    s_cell[8 * PB + tid] = d.att_h[0][cb * ATT_RNN + 4 * c + cu];
    s_cell[12 * PB + tid] = d.dec_h[0][cb * DEC_RNN + 4 * c + cu];
  }
  int nf_r = 0;
  if (attn) {
    for (int i = tid; i < TP * 16; i += PT) {
      const int t = i >> 4, dd = i & 15;
      s_pm[i] = t < T ? d.pmem[((size_t)rb * T + t) * ATT_DIM + 16 * rk + dd] : 0.f;
    }
    if (tid < TP) {
      s_aw[tid] = tid < T ? d.aw[rb * T + tid] : 0.f;
      s_awc[tid] = tid < T ? d.awc[rb * T + tid] : 0.f;
    }
    for (int i = tid; i < 62 * 16; i += PT) s_G[i] = w.loc_fused[(size_t)(i >> 4) * ATT_DIM + 16 * rk + (i & 15)];
    if (tid < 16) s_vv[tid] = w.v_w[16 * rk + tid];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4 *>(s_qw + 4 * ((4 * r + j) * PT + tid)) = w.q_w[(unsigned)((16 * rk + wave + NW * r) * (ATT_RNN / 4) + lane + 64 * j)];
  }
  if (pre) {
    // [in / 4][out][in % 4]: a thread's 40 layer-1 weights are ten conflict-free 16-byte reads (two-chunk kernel; the
    // one-chunk kernel keeps them in registers)
    for (int i = tid; i < N_MEL * PRENET; i += PT) s_W0[(((i / PRENET) >> 2) * PRENET + i % PRENET) * 4 + ((i / PRENET) & 3)] = w.pre0T[i];
    if (prow_ok) {
      if (lane == 0) s_pb[wave] = w.proj_b[prow];
#pragma unroll
      for (int j = 0; j < 6; ++j)
        *reinterpret_cast<float4 *>(s_pw + 4 * (j * PT + tid)) = w.proj_w[(unsigned)(prow * (PROJ_IN / 4) + lane + 64 * j)];
    }
    nf_r = d.nframes[rb];
  }
  // One-chunk kernel: the role's per-step weights move from LDS into the 40 registers the second chunk would need
  // (attention: the two query rows of this wave, 8 float4; prenet: this thread's 40 layer-1 weights) -- the LDS
  // operand reads of these two inner products were ~0.4 us of the step's critical path.
  constexpr bool ROLE_REGS = PB == 1;
  float rw[ROLE_REGS ? 40 : 1];
  if (ROLE_REGS) {
#pragma unroll
    for (int k = 0; k < 40; ++k) rw[ROLE_REGS ? k : 0] = 0.f;
    if (attn) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 q = w.q_w[(unsigned)((16 * rk + wave + NW * r) * (ATT_RNN / 4) + lane + 64 * j)];
          rw[ROLE_REGS ? 16 * r + 4 * j + 0 : 0] = q.x;
          rw[ROLE_REGS ? 16 * r + 4 * j + 1 : 0] = q.y;
          rw[ROLE_REGS ? 16 * r + 4 * j + 2 : 0] = q.z;
          rw[ROLE_REGS ? 16 * r + 4 * j + 3 : 0] = q.w;
        }
    } else if (pre) {
      // layer 1: output tid & 255, inputs [40 hf, 40 hf + 40), hf = tid >> 8
#pragma unroll
      for (int k = 0; k < N_MEL / 2; ++k) rw[ROLE_REGS ? k : 0] = w.pre0T[(unsigned)(((tid >> 8) * (N_MEL / 2) + k) * PRENET + (tid & 255))];
    }
  }
  // prenet layer 2, columns 16 rk + wave (+8), inputs lane + 64 k: RESIDENT (8 registers).  Re-read every step they
  // were 4096 line requests per workgroup (a column of the [in][out] matrix is one float per 1 kB) = 1.4 us that
  // stood in front of the mel poll (in-order return), the longest single piece of the step's critical path.
  float w1r[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) w1r[r][k] = pre ? w.pre1T[(unsigned)((lane + 64 * k) * PRENET + 16 * rk + wave + NW * r)] : 0.f;
  const uint32_t item = d.item_base + (uint32_t)rb;
  __syncthreads();

  // ---- context columns folded into the encoder memory (once per launch) -------------------------
  // pma / pmd [b][r][h]: lane l holds  W_att / W_dec [row r of this wave][ctx cols] . memory_b[t = l + 64 h];
  // pmp [h]: the same for this wave's projection row (projection role, chunk rb).  The ctx column blocks of wa /
  // wd (k = 1, 2 and k = 4, 5) are dead after this loop: 32 weight registers make room for 8 PB + 2 of these.
  float pma[PB][2][2], pmd[PB][2][2], pmp[2] = {0.f, 0.f};
  if (d.ctx_fold) {
    // ... or read from the table one GEMM per request has made of them (api.cpp): 33 us of set-up less per chunk and launch
#pragma unroll
    for (int b = 0; b < PB; ++b)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        const float *F = d.ctx_fold + (size_t)b * CTXF_ROWS * CTXF_LD + t;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int row = 16 * c + wave + NW * r;
          pma[b][r][h] = t < T ? F[(size_t)row * CTXF_LD] : 0.f;
          pmd[b][r][h] = t < T ? F[(size_t)(4 * ATT_RNN + row) * CTXF_LD] : 0.f;
        }
        if (prow_ok && b == rb) pmp[h] = t < T ? F[(size_t)(4 * ATT_RNN + 4 * DEC_RNN + prow) * CTXF_LD] : 0.f;
      }
  } else {
    float4 pw4 = make_float4(0.f, 0.f, 0.f, 0.f), pw5 = pw4;
    if (prow_ok) {
      pw4 = lds4(s_pw + 4 * (4 * PT + tid));
      pw5 = lds4(s_pw + 4 * (5 * PT + tid));
    }
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      const float4 *m4 = reinterpret_cast<const float4 *>(d.memory + (size_t)b * T * EMB);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float ra[2] = {0.f, 0.f}, rd[2] = {0.f, 0.f}, rp = 0.f;
        for (int tt = 0; tt < 64 && tt + 64 * h < T; ++tt) {
          const int t = tt + 64 * h;
          const float4 m0 = m4[(size_t)t * (EMB / 4) + lane], m1 = m4[(size_t)t * (EMB / 4) + 64 + lane];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const float va = wave_sum(dot4(wa[r][2], m1, dot4(wa[r][1], m0, 0.f)));
            const float vd = wave_sum(dot4(wd[r][5], m1, dot4(wd[r][4], m0, 0.f)));
            if (lane == tt) {
              ra[r] = va;
              rd[r] = vd;
            }
          }
          if (pre && b == rb) {  // (workgroup-uniform)
            const float vp = wave_sum(dot4(pw5, m1, dot4(pw4, m0, 0.f)));
            if (lane == tt) rp = vp;
          }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          pma[b][r][h] = ra[r];
          pmd[b][r][h] = rd[r];
        }
        if (pre && b == rb) pmp[h] = rp;
      }
    }
  }

  // ---- deferred pieces: everything that does not depend on the newest vector -----------------
  float aacc[PB][2], dacc[PB][2];
  auto att_bulk = [&](unsigned L4, bool all) {  // attention LSTM, columns [ctx(s-1) ; h_att(s-1)]
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      if (!all && !s_act[b]) continue;  // a stopped chunk's state is frozen
      const float w0 = wreg[b][0], w1 = wreg[b][1];
      float a0 = fmaf(pma[b][0][1], w1, pma[b][0][0] * w0), a1 = fmaf(pma[b][1][1], w1, pma[b][1][0] * w0);
#pragma unroll
      for (int k = 3; k < 7; ++k) {
        const float4 v = lds4(s_hatt + b * ATT_RNN + 256 * (k - 3) + L4);
        a0 = dot4(wa[0][k], v, a0);
        a1 = dot4(wa[1][k], v, a1);
      }
      aacc[b][0] = a0;
      aacc[b][1] = a1;
      chunk_fence();
    }
  };
  auto dec_bulk_h = [&](unsigned L4, bool all) {  // decoder LSTM, columns h_dec(s-1)
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      if (!all && !s_act[b]) continue;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int k = 6; k < 10; ++k) {
        const float4 v = lds4(s_hdec + b * DEC_RNN + 256 * (k - 6) + L4);
        a0 = dot4(wd[0][k], v, a0);
        a1 = dot4(wd[1][k], v, a1);
      }
      dacc[b][0] = a0;
      dacc[b][1] = a1;
      chunk_fence();
    }
  };
  // location features of the NEXT step for this workgroup's 16 dims, from s_aw / s_awc:
  //   loc[t][a] = sum_{c,k} G[a][c][k] pad_c[t + k],  G = dense . conv folded on the host,
  // as a (TP x 64) . (64 x 16) product on the matrix cores: wave w forms rows t = 16 w .. 16 w + 15 with 16
  // v_mfma_f32_16x16x4_f32 (exact fp32), the Toeplitz operand A[t][q] = pad_{q / 31}[t + q % 31] read straight from the
  // padded window (taps 62, 63 multiply zero rows of G).  0.4 us against the 2.8 us of the 496-FMA-per-thread sliding
  // window it replaces, which had become the step's critical path (the attention workgroups are LSTM slices too: their
  // h_att is late when this runs long).
  auto location = [&](int tid) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (int i = tid; i < 2 * WPAD; i += PT) {
      const int ch = i / WPAD, t = i % WPAD - (LOC_K - 1) / 2;
      s_wpad[i] = (t >= 0 && t < T) ? (ch ? s_awc[t] : s_aw[t]) : 0.f;
    }
    __syncthreads();
    {
      const unsigned l = (unsigned)tid & 63u, li = l & 15u, lg = l >> 4, t0 = 16u * ((unsigned)tid >> 6);
      float av[16], bv[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const unsigned q = 4u * kk + lg, qa = q < 2u * LOC_K ? q : 2u * LOC_K - 1u, ch = qa >= (unsigned)LOC_K ? 1u : 0u;
        av[kk] = s_wpad[ch * WPAD + t0 + li + (qa - ch * LOC_K)];
        bv[kk] = q < 2u * LOC_K ? s_G[qa * 16u + li] : 0.f;
      }
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;  // two chains: a dependent MFMA waits ~40 cycles
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk + 1], bv[kk + 1], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) s_loc[(t0 + 4u * lg + j) * 16u + li] = acc0[j] + acc1[j];
    }
    __syncthreads();
  };
  att_bulk(L4, true);
  dec_bulk_h(L4, true);
  if (attn) location(tid);

#ifdef XDTTS_PERSIST_PROFILE
  __shared__ u64 s_prof[24];
  if (tid < 24) s_prof[tid] = 0;
  u64 prof_last = wall_clock64();
  __syncthreads();
#endif
  int s = step0;
  const int s_stop = step0 + nsteps;
  if constexpr (SKEW) {
    // ---- skewed pair --------------------------------------------------------------------------------------------------
    // A step of one chunk is five phases, each a gather (wait for a vector to cross the chip, ~1 us) followed by the
    // arithmetic that needs it and a publish:  ph1 x -> attention-LSTM tail -> h_att;  ph2 h_att -> (attention role) query,
    // energies;  ph3 energies -> softmax, decoder-LSTM tail -> h_dec;  ph4 h_dec -> (projection role) mel rows;
    // ph5 (projection role) mel -> prenet -> x.  In lock-step both chunks wait together and compute one after the other;
    // here chunk 1 runs two phases behind chunk 0 and a workgroup alternates between the chunks, so that one chunk's
    // arithmetic fills the other's wait:
    //     c0.ph1(s) c1.ph4(s-1) | c0.ph2(s) c1.ph5(s-1) | c0.ph3(s) c1.ph1(s) | c0.ph4(s) c1.ph2(s) | c0.ph5(s) c1.ph3(s)
    // The launch ends when either chunk stops (the host continues the other with the 1-chunk kernel): chunk 0 found
    // stopped at its ph1(s) -> chunk 1 completes step s-1, both have run s steps; chunk 1 found stopped at its ph1(s) ->
    // chunk 0 completes step s: it has run s+1 steps, which is what ctl[0] then says (it is the survivor).
    static_assert(PB == 2, "the skewed loop is the pair kernel");
    const int tid_k = tid;
    const bool lag = g.slow && c == g.slow - 1;
    unsigned drop1 = 0u, drop2 = 0u;  // projection role: masks of step s+1, hashed in ph4, used in ph5
    // The first poll of a phase's vector is issued by the phase BEFORE it (the other chunk's), at its very end: a phase pays
    // the poll's round trip (~0.45 us) even when its vector has long arrived, and part of that now runs under the previous
    // phase's tail.  (Issued earlier -- right behind that phase's publish -- the poll samples the slots before the slowest
    // producer's store has landed: 10.98 us per step against 10.6-10.7; in the middle of the deferred column blocks: the
    // same as behind them; lock-step: 11.27.)  The granules validate themselves (tag = step + 1), so a poll that was too
    // early, or none at all at the ends of the sequence, only means the ordinary poll loop.
    u64 pf[2] = {0ull, 0ull};
    auto issue_x = [&](auto B, int s) {
      constexpr int b = decltype(B)::value;
      const int tid = tid_k + (int)opaque(0u);
      pf[0] = tid < 256 ? peek(g.x + (unsigned)(((s & 1) * GS + b) * PRENET + (tid & 255))) : 0ull;
      pf[1] = 0ull;
    };
    auto issue_h = [&](const u64 *base, auto B, int s) {  // h_att / h_dec (both 1024 wide)
      constexpr int b = decltype(B)::value;
      const int tid = tid_k + (int)opaque(0u);
      pf[0] = peek(base + (unsigned)(((s & 1) * GS + b) * ATT_RNN + tid));
      pf[1] = peek(base + (unsigned)(((s & 1) * GS + b) * ATT_RNN + tid + PT));
    };
    auto issue_ep = [&](auto B, int s) {
      constexpr int b = decltype(B)::value;
      const int tid = tid_k + (int)opaque(0u), t = tid >> 2, j = tid & 3;
      pf[0] = t < T ? peek(g.ep + (unsigned)((((s & 1) * GS + b) * ATTN_CU + j) * EP_LD + t)) : 0ull;
      pf[1] = t < T ? peek(g.ep + (unsigned)((((s & 1) * GS + b) * ATTN_CU + j + 4) * EP_LD + t)) : 0ull;
    };
    auto issue_mel = [&](auto B, int s) {
      constexpr int b = decltype(B)::value;
      const int tid = tid_k + (int)opaque(0u);
      pf[0] = (pre && rb == b && tid < N_MEL + 1) ? peek(g.mel + (unsigned)((((s & 1) * GS + b) * MEL_GL + tid) * MEL_ST)) : 0ull;
      pf[1] = 0ull;
    };
    auto I0 = std::integral_constant<int, 0>{};
    auto I1 = std::integral_constant<int, 1>{};
    auto att_bulk1 = [&](auto B, unsigned L4) {
      constexpr int b = decltype(B)::value;
      const float w0 = wreg[b][0], w1 = wreg[b][1];
      float a0 = fmaf(pma[b][0][1], w1, pma[b][0][0] * w0), a1 = fmaf(pma[b][1][1], w1, pma[b][1][0] * w0);
#pragma unroll
      for (int k = 3; k < 7; ++k) {
        const float4 v = lds4(s_hatt + b * ATT_RNN + 256 * (k - 3) + L4);
        a0 = dot4(wa[0][k], v, a0);
        a1 = dot4(wa[1][k], v, a1);
      }
      aacc[b][0] = a0;
      aacc[b][1] = a1;
    };
    auto dec_bulk_h1 = [&](auto B, unsigned L4) {
      constexpr int b = decltype(B)::value;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int k = 6; k < 10; ++k) {
        const float4 v = lds4(s_hdec + b * DEC_RNN + 256 * (k - 6) + L4);
        a0 = dot4(wd[0][k], v, a0);
        a1 = dot4(wd[1][k], v, a1);
      }
      dacc[b][0] = a0;
      dacc[b][1] = a1;
    };
    // ph1: x(s) -> attention-LSTM tail -> publish h_att(s).  false: the chunk has stopped (or the exchange failed).
    auto ph1 = [&](auto B, int s, auto nxt) -> bool {
      constexpr int b = decltype(B)::value;
      const int tid = tid_k + (int)opaque(0u), lane = tid & 63, wave = tid >> 6;
      const unsigned L4 = 4u * (unsigned)lane;
      const int p = s & 1;
      const unsigned want = (unsigned)(s + 1);
      if (b == 0) straggle(lag, s, 0);
      {
        const int i = tid & 255;
        const bool need[1] = {tid < 256};
        float v[1];
        unsigned tg[1];
        gather<1, true>(g.x, (unsigned)((p * GS + b) * PRENET + i), 0u, want, need, v, tg, pc, pf);
        if (need[0]) {
          s_x[b * PRENET + i] = v[0];
          if (i == 0) s_act[b] = (tg[0] & ACT_BIT) ? 1 : 0;
        }
        if (tid == PT - 1) s_act[4] = __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      const bool on = s_act[b] != 0 && s_act[4] == 0;
      if (!on) return false;  // (workgroup-uniform)
      {
        const float4 v = lds4(s_x + b * PRENET + L4);
        const float a0 = wave_sum(dot4(wa[0][0], v, aacc[b][0]));
        const float a1 = wave_sum(dot4(wa[1][0], v, aacc[b][1]));
        if (lane == 0) {
          s_g[b * 16 + wave] = a0 + s_bias[wave];
          s_g[b * 16 + wave + NW] = a1 + s_bias[wave + NW];
        }
      }
      __syncthreads();
      if (tid < 4 * PB && (tid >> 2) == b) {
        const int cu = tid & 3;
        const float *gp = s_g + b * 16 + 4 * cu;
        const float ig = fast_sigmoid(gp[0]), fg = fast_sigmoid(gp[1]), gg = fast_tanh(gp[2]), og = fast_sigmoid(gp[3]);
        const float cn = fmaf(fg, s_cell[tid], ig * gg), hn = og * fast_tanh(cn);
        publish(g.hatt + (unsigned)((p * GS + b) * ATT_RNN + 4 * c + cu), want, hn);
        s_cell[tid] = cn;
        s_cell[8 * PB + tid] = hn;
      }
      nxt();
      PROF_MARK(5 * b + 0);
      if (b == 0) straggle(lag, s, 1);
      __builtin_amdgcn_sched_barrier(0);
      return true;
    };
    // ph2: h_att(s) -> (attention role of this chunk) query rows, partial energies -> publish; decoder LSTM's h_att columns
    auto ph2 = [&](auto B, int s, auto nxt) {
      constexpr int b = decltype(B)::value;
      const int tid = tid_k + (int)opaque(0u), lane = tid & 63, wave = tid >> 6;
      const unsigned L4 = 4u * (unsigned)lane, TID = (unsigned)tid;
      const int p = s & 1;
      const unsigned want = (unsigned)(s + 1);
      const bool mine = attn && rb == b;
      float4 lp4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = lp4;
      if (mine) {
        const float4 l4 = lds4(s_loc + 4 * TID), p4 = lds4(s_pm + 4 * TID);
        lp4 = make_float4(l4.x + p4.x, l4.y + p4.y, l4.z + p4.z, l4.w + p4.w);
        v4 = lds4(s_vv + 4 * (tid & 3));
      }
      {
        const bool need[2] = {true, true};
        float v[2];
        unsigned tg[2];
        gather<2, true>(g.hatt, (unsigned)((p * GS + b) * ATT_RNN + tid), PT, want, need, v, tg, pc, pf);
        s_hatt[b * ATT_RNN + TID] = v[0];
        s_hatt[b * ATT_RNN + TID + PT] = v[1];
      }
      __syncthreads();
      if (mine) {
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = lds4(s_hatt + b * ATT_RNN + 256 * j + L4);
          q0 = dot4(lds4(s_qw + 4 * (j * PT + TID)), v, q0);
          q1 = dot4(lds4(s_qw + 4 * ((4 + j) * PT + TID)), v, q1);
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
        if ((tid & 3) == 0 && t < T) publish(g.ep + (unsigned)(((p * GS + b) * ATTN_CU + rk) * EP_LD + t), want, e);
      }
      {
        float a0 = dacc[b][0], a1 = dacc[b][1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 v = lds4(s_hatt + b * ATT_RNN + 256 * k + L4);
          a0 = dot4(wd[0][k], v, a0);
          a1 = dot4(wd[1][k], v, a1);
        }
        dacc[b][0] = a0;
        dacc[b][1] = a1;
      }
      nxt();
      PROF_MARK(5 * b + 1);
      if (b == 0) straggle(lag, s, 2);
      __builtin_amdgcn_sched_barrier(0);
    };
    // ph3: partial energies -> softmax -> decoder-LSTM tail -> publish h_dec(s); attention LSTM's [ctx ; h_att] columns of step s+1
    auto ph3 = [&](auto B, int s, auto nxt) {
      constexpr int b = decltype(B)::value;
      const int tid = tid_k + (int)opaque(0u), lane = tid & 63, wave = tid >> 6;
      const unsigned L4 = 4u * (unsigned)lane;
      const int p = s & 1;
      const unsigned want = (unsigned)(s + 1);
      const bool mine = attn && rb == b;
      {
        const int t = tid >> 2, j = tid & 3;
        const bool need[2] = {t < T, t < T};
        float v[2];
        unsigned tg[2];
        gather<2, true>(g.ep, (unsigned)(((p * GS + b) * ATTN_CU + j) * EP_LD + t), 4u * EP_LD, want, need, v, tg, pc, pf);
        float e = v[0] + v[1];
        e += dpp_move<0xB1, 0xf>(0.f, e);  // quad_perm:[1,0,3,2]
        e += dpp_move<0x4E, 0xf>(0.f, e);  // quad_perm:[2,3,0,1]
        if (j == 0) s_e[b * TP + t] = (t < T && t < nv[b]) ? e : -INFINITY;  // mask, mod.rs:219-220
      }
      __syncthreads();
      {
        const float e0 = s_e[b * TP + lane], e1 = s_e[b * TP + lane + 64];
        const float m = wave_max(fmaxf(e0, e1));
        const float x0 = fast_exp(e0 - m), x1 = fast_exp(e1 - m);
        const float rs = __builtin_amdgcn_rcpf(wave_sum(x0 + x1));
        wreg[b][0] = x0 * rs;
        wreg[b][1] = x1 * rs;
      }
      if (mine && wave == 0) {
        s_aw[lane] = wreg[b][0];
        s_awc[lane] += wreg[b][0];
        s_aw[lane + 64] = wreg[b][1];
        s_awc[lane + 64] += wreg[b][1];
      }
      if (b == 0) straggle(lag, s, 3);
      {
        const float w0 = wreg[b][0], w1 = wreg[b][1];
        float a0 = fmaf(pmd[b][0][1], w1, fmaf(pmd[b][0][0], w0, dacc[b][0]));
        float a1 = fmaf(pmd[b][1][1], w1, fmaf(pmd[b][1][0], w0, dacc[b][1]));
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
          s_g[b * 16 + wave] = a0 + s_bias[16 + wave];
          s_g[b * 16 + wave + NW] = a1 + s_bias[16 + wave + NW];
        }
      }
      __syncthreads();
      if (tid < 4 * PB && (tid >> 2) == b) {
        const int cu = tid & 3;
        const float *gp = s_g + b * 16 + 4 * cu;
        const float ig = fast_sigmoid(gp[0]), fg = fast_sigmoid(gp[1]), gg = fast_tanh(gp[2]), og = fast_sigmoid(gp[3]);
        const float cn = fmaf(fg, s_cell[4 * PB + tid], ig * gg), hn = og * fast_tanh(cn);
        publish(g.hdec + (unsigned)((p * GS + b) * DEC_RNN + 4 * c + cu), want, hn);
        s_cell[4 * PB + tid] = cn;
        s_cell[12 * PB + tid] = hn;
      }
      att_bulk1(B, L4);  // for step s+1: ctx(s), h_att(s)
      nxt();
      PROF_MARK(5 * b + 2);
      if (b == 0) straggle(lag, s, 4);
      __builtin_amdgcn_sched_barrier(0);
    };
    // ph4: h_dec(s) -> (projection role of this chunk) mel rows -> publish; decoder LSTM's own-state columns of step s+1;
    // (attention role of this chunk) location features of step s+1
    auto ph4 = [&](auto B, int s, auto nxt) {
      constexpr int b = decltype(B)::value;
      const int tid = tid_k + (int)opaque(0u), lane = tid & 63, wave = tid >> 6;
      const unsigned L4 = 4u * (unsigned)lane, TID = (unsigned)tid;
      const int p = s & 1;
      const unsigned want = (unsigned)(s + 1);
      const bool mine = pre && rb == b;
      const int prow = rk + 16 * wave;
      const bool prow_ok = mine && wave < 6 && prow <= N_MEL;
      if (mine) {
        drop1 = 0u;
        drop2 = 0u;
        if (d.dropout_mode) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            drop1 |= (prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 0, lane + 64 * k) ? 1u : 0u) << k;
#pragma unroll
          for (int r = 0; r < 2; ++r)
            drop2 |= (prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 1, 16 * rk + wave + NW * r) ? 1u : 0u) << r;
        }
      }
      {
        const bool need[2] = {true, true};
        float v[2];
        unsigned tg[2];
        gather<2, true>(g.hdec, (unsigned)((p * GS + b) * DEC_RNN + tid), PT, want, need, v, tg, pc, pf);
        s_hdec[b * DEC_RNN + TID] = v[0];
        s_hdec[b * DEC_RNN + TID + PT] = v[1];
      }
      __syncthreads();
      if (prow_ok) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) a = dot4(lds4(s_pw + 4 * (j * PT + TID)), lds4(s_hdec + b * DEC_RNN + 256 * j + L4), a);
        a = fmaf(pmp[1], wreg[b][1], fmaf(pmp[0], wreg[b][0], a));  // the context columns
        a = wave_sum(a);
        if (lane == 0) publish(g.mel + (unsigned)(((p * GS + b) * MEL_GL + prow) * MEL_ST), want, a + s_pb[wave]);
      }
      dec_bulk_h1(B, L4);  // for step s+1
      nxt();
      if (attn && rb == b) location(tid);
      PROF_MARK(5 * b + 3);
      if (b == 0) straggle(lag, s, 5);
      __builtin_amdgcn_sched_barrier(0);
    };
    // ph5 (projection + prenet role of this chunk): frame s, stop rule, x(s+1)
    auto ph5 = [&](auto B, int s, auto nxt) {
      constexpr int b = decltype(B)::value;
      if (!(pre && rb == b)) {  // (workgroup-uniform)
        nxt();
        return;
      }
      PROF_MARK(10 + 3 * b);  // (role only: time since the previous phase's end)
      const int tid = tid_k + (int)opaque(0u), lane = tid & 63, wave = tid >> 6;
      const unsigned L4 = 4u * (unsigned)lane, TID = (unsigned)tid;
      const int p = s & 1;
      const unsigned want = (unsigned)(s + 1);
      if (tid < N_MEL + 1) {
        const bool need[1] = {true};
        float v[1];
        unsigned tg[1];
        gather<1, true>(g.mel, (unsigned)(((p * GS + b) * MEL_GL + tid) * MEL_ST), 0, want, need, v, tg, pc, pf);
        s_mel[tid] = v[0];
      }
      __syncthreads();
      PROF_MARK(11 + 3 * b);  // (role only: mel gathered)
      const float gate = s_mel[N_MEL];
      const bool fired = GATE ? gate_fires(gate, d.gate_lo, d.gate_hi, d.gate_threshold) : (d.use_gate && gate_sigmoid(gate) > d.gate_threshold);  // mod.rs:319-324
      if (rk == 0) {
        if (tid < N_MEL) d.frames[((size_t)b * d.max_steps + s) * N_MEL + tid] = s_mel[tid];
        if (tid == 0) {
          d.gates[(size_t)b * d.max_steps + s] = gate;
          if (fired) d.nframes[b] = s + 1;  // the tripping frame is kept
        }
      }
      if (fired) nf_r = s + 1;
      const bool more = s + 1 < nf_r;
      float xo[2] = {0.f, 0.f};
      if (more) {
        const unsigned HM = (unsigned)((tid >> 8) * (N_MEL / 2));
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < N_MEL / 2; k += 4) {
          const float4 w4 = lds4(s_W0 + 4u * (((HM + k) >> 2) * PRENET + (TID & 255u))), m = lds4(s_mel + HM + k);
          acc = fmaf(w4.x, m.x, acc);
          acc = fmaf(w4.y, m.y, acc);
          acc = fmaf(w4.z, m.z, acc);
          acc = fmaf(w4.w, m.w, acc);
        }
        s_l1[TID] = acc;
        __syncthreads();
        float pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = fmaxf(s_l1[(L4 >> 2) + 64 * k] + s_l1[PRENET + (L4 >> 2) + 64 * k], 0.f);
          pk[k] = (drop1 >> k) & 1u ? 0.f : (d.dropout_mode ? 2.f * v : v);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) a = fmaf(w1r[r][k], pk[k], a);
          a = fmaxf(wave_sum(a), 0.f);
          xo[r] = (drop2 >> r) & 1u ? 0.f : (d.dropout_mode ? 2.f * a : a);
        }
      }
      if (lane < 2) s_mel[MEL_GL - 16 + wave + NW * lane] = lane ? xo[1] : xo[0];  // s_mel[81..95] is unused padding
      __syncthreads();
      if (tid < 16)
        publish(g.x + (unsigned)(((p ^ 1) * GS + b) * PRENET + 16 * rk + tid), (want + 1u) | (more ? ACT_BIT : 0u),
                s_mel[MEL_GL - 16 + tid]);
      nxt();
      PROF_MARK(5 * b + 4);
    };
    bool tail1 = false;  // chunk 1 has run ph1..ph3 of step s-1: its ph4 / ph5 come in the first half of this round
    if (s_act[2] != 0 && s_act[3] != 0) {
      issue_x(I0, s);
      for (;; ++s) {
        bool a0 = s < s_stop;
        if (a0) a0 = ph1(I0, s, [&] { if (tail1) issue_h(g.hdec, I1, s - 1); else issue_h(g.hatt, I0, s); });
        if (tail1) ph4(I1, s - 1, [&] { if (a0) issue_h(g.hatt, I0, s); });
        if (a0) ph2(I0, s, [&] { if (tail1) issue_mel(I1, s - 1); else issue_ep(I0, s); });
        if (tail1) {
          ph5(I1, s - 1, [&] { if (a0) issue_ep(I0, s); });
          tail1 = false;
        }
        if (!a0) break;  // chunk 0 has stopped (or the limit / an exchange failure): both chunks have completed s steps
        ph3(I0, s, [&] { issue_x(I1, s); });
        const bool a1 = ph1(I1, s, [&] { issue_h(g.hdec, I0, s); });
        if (!a1) issue_h(g.hdec, I0, s);
        ph4(I0, s, [&] { if (a1) issue_h(g.hatt, I1, s); else issue_mel(I0, s); });
        if (a1) ph2(I1, s, [&] { issue_mel(I0, s); });
        ph5(I0, s, [&] { if (a1) issue_ep(I1, s); });
        if (!a1) {  // chunk 1 has stopped: chunk 0, the survivor, has completed step s
          ++s;
          break;
        }
        ph3(I1, s, [&] { issue_x(I0, s + 1); });
        tail1 = true;
      }
    }
  } else
  for (; s < s_stop; ++s) {
    // Per-iteration opaque copies of the thread indices: nothing derived from them can be hoisted
    // out of the step loop, so addresses are recomputed next to their use (one VALU op each)
    // instead of living in ~100 registers (and their spill slots) across the whole iteration.
    const int tid_it = tid + (int)opaque(0u);
    {
    const int tid = tid_it, lane = tid & 63, wave = tid >> 6;
    const unsigned L4 = 4u * (unsigned)lane, TID = (unsigned)tid;
    const int cb = tid >> 2, cu = tid & 3;
    const bool cell = tid < 4 * PB;
    const int prow = rk + 16 * wave;
    const bool prow_ok = pre && wave < 6 && prow <= N_MEL;
    const int p = s & 1;
    const unsigned want = (unsigned)(s + 1);
    const bool lag = g.slow && c == g.slow - 1;
    straggle(lag, s, 0);
    PROF_MARK(0);  // loop overhead / previous P6 tail
    // ---- P1: x(s) and the chunks' active bits ------------------------------------------------
    {
      const int b0 = tid >> 8, i = tid & 255;  // chunks b0 and b0 + 2
      const bool need[2] = {b0 < PB && s_act[2 + (b0 < PB ? b0 : 0)] != 0, false};  // (PB <= 2: chunk b0 only)
      float v[2];
      unsigned tg[2];
      lazy_wait(pre ? g.xfirst : g.xlazy);  // x(s+1) cannot arrive before the projection / prenet chain has run
      const unsigned np_ = gather<2>(g.x, (unsigned)((p * GS + b0) * PRENET + i), 2u * PRENET, want, need, v, tg, pc);
      PROF_POLLS(0, np_);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (need[j]) {
          s_x[(b0 + 2 * j) * PRENET + i] = v[j];
          if (i == 0) s_act[b0 + 2 * j] = (tg[j] & ACT_BIT) ? 1 : 0;
        }
      if (tid == PT - 1) s_act[4] = __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    bool act[PB], any = false;
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      act[b] = s_act[b] != 0;
      any = any || act[b];
    }
    if (tid < PB) s_act[2 + tid] = s_act[tid];  // read again only after the barriers of this step
    PROF_MARK(1);  // wait x
    PROF_WHEN(2);
    // Every chunk has stopped (or the exchange failed): the launch ends by itself.  A 2-chunk launch
    // also ends when one of its chunks stops: the host continues the other with the 1-chunk kernel,
    // which is ~1 us per step faster (state goes through the write-back below, x(s) stays in place).
    // (and it ends at once when any workgroup has reported a timed-out exchange)
    if (!any || s_act[4] != 0 || (PB == 2 && g.shrink && !(act[0] && act[1]))) break;
    const bool act_r = s_act[rb] != 0;
    // attention LSTM: close the rows with the x columns
#pragma unroll
    for (int b = 0; b < PB; ++b)
      if (act[b]) {
        const float4 v = lds4(s_x + b * PRENET + L4);
        const float a0 = wave_sum(dot4(wa[0][0], v, aacc[b][0]));
        const float a1 = wave_sum(dot4(wa[1][0], v, aacc[b][1]));
        if (lane == 0) {
          s_g[b * 16 + wave] = a0 + s_bias[wave];
          s_g[b * 16 + wave + NW] = a1 + s_bias[wave + NW];
        }
      }
    __syncthreads();
    if (cell && s_act[cb]) {
      const float *gp = s_g + cb * 16 + 4 * cu;
      const float ig = fast_sigmoid(gp[0]), fg = fast_sigmoid(gp[1]), gg = fast_tanh(gp[2]), og = fast_sigmoid(gp[3]);
      const float cn = fmaf(fg, s_cell[tid], ig * gg), hn = og * fast_tanh(cn);
      publish(g.hatt + (unsigned)((p * GS + cb) * ATT_RNN + 4 * c + cu), want, hn);
      s_cell[tid] = cn;
      s_cell[8 * PB + tid] = hn;
    }
    PROF_MARK(2);  // att tail + cell + publish
    PROF_WHEN(0);
    straggle(lag, s, 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P2: h_att(s) ----------------------------------------------------------------------------
    // attention role: what the energies need besides the query (location features + processed memory of this thread's
    // (t, 4 dims), v) is in registers before h_att arrives
    float4 lp4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = lp4;
    if (attn && act_r) {
      const float4 l4 = lds4(s_loc + 4 * TID), p4 = lds4(s_pm + 4 * TID);
      lp4 = make_float4(l4.x + p4.x, l4.y + p4.y, l4.z + p4.z, l4.w + p4.w);
      v4 = lds4(s_vv + 4 * (tid & 3));
    }
    {
      // both halves of every active chunk's vector in flight together: granule tid + 512 i, i = 2 b + half
      bool need[2 * PB];
#pragma unroll
      for (int i = 0; i < 2 * PB; ++i) need[i] = act[i >> 1];
      float v[2 * PB];
      unsigned tg[2 * PB];
      lazy_wait(attn ? g.first : g.lazy);  // only the attention role needs h_att at once
      const unsigned np_ = gather<2 * PB>(g.hatt, (unsigned)(p * GS * ATT_RNN + tid), PT, want, need, v, tg, pc);
      PROF_POLLS(1, np_);
#pragma unroll
      for (int i = 0; i < 2 * PB; ++i)
        if (need[i]) s_hatt[TID + PT * i] = v[i];
    }
    __syncthreads();
    PROF_MARK(3);  // wait h_att
    if (attn && act_r) {
      // query rows 16 rk + wave (+8), then this workgroup's share of the energies (mod.rs:304, D3)
      float q0 = 0.f, q1 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = lds4(s_hatt + rb * ATT_RNN + 256 * j + L4);
        if (ROLE_REGS) {
          q0 = dot4(make_float4(rw[ROLE_REGS ? 4 * j : 0], rw[ROLE_REGS ? 4 * j + 1 : 0], rw[ROLE_REGS ? 4 * j + 2 : 0], rw[ROLE_REGS ? 4 * j + 3 : 0]), v, q0);
          q1 = dot4(make_float4(rw[ROLE_REGS ? 16 + 4 * j : 0], rw[ROLE_REGS ? 17 + 4 * j : 0], rw[ROLE_REGS ? 18 + 4 * j : 0], rw[ROLE_REGS ? 19 + 4 * j : 0]), v, q1);
        } else {
          q0 = dot4(lds4(s_qw + 4 * (j * PT + TID)), v, q0);
          q1 = dot4(lds4(s_qw + 4 * ((4 + j) * PT + TID)), v, q1);
        }
      }
      q0 = wave_sum(q0);
      q1 = wave_sum(q1);
      if (lane == 0) {
        s_q[wave] = q0;
        s_q[wave + NW] = q1;
      }
      __syncthreads();
      PROF_MARK(15);  // attention role: query rows + barrier
      const int t = tid >> 2, dq = 4 * (tid & 3);
      const float4 q4 = lds4(s_q + dq);
      float e = v4.x * fast_tanh(q4.x + lp4.x);
      e = fmaf(v4.y, fast_tanh(q4.y + lp4.y), e);
      e = fmaf(v4.z, fast_tanh(q4.z + lp4.z), e);
      e = fmaf(v4.w, fast_tanh(q4.w + lp4.w), e);
      e += dpp_move<0xB1, 0xf>(0.f, e);  // quad_perm:[1,0,3,2]
      e += dpp_move<0x4E, 0xf>(0.f, e);  // quad_perm:[2,3,0,1]
      if ((tid & 3) == 0 && t < T) publish(g.ep + (unsigned)(((p * GS + rb) * ATTN_CU + rk) * EP_LD + t), want, e);
    }
    // decoder LSTM: the h_att columns (off the critical path for everyone but the above)
#pragma unroll
    for (int b = 0; b < PB; ++b)
      if (act[b]) {
        float a0 = dacc[b][0], a1 = dacc[b][1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 v = lds4(s_hatt + b * ATT_RNN + 256 * k + L4);
          a0 = dot4(wd[0][k], v, a0);
          a1 = dot4(wd[1][k], v, a1);
        }
        dacc[b][0] = a0;
        dacc[b][1] = a1;
        chunk_fence();
      }
    PROF_MARK(4);  // q + energies (attention) + dec bulk
    straggle(lag, s, 2);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P3 (every workgroup): partial energies of the 8 slices of every chunk -> softmax -> weights in registers ----
    {
      // all threads poll: thread -> time step tid / 4, slices j and j + 4 (j = tid % 4) of every active chunk; quad sum
      const int t = tid >> 2, j = tid & 3;
      bool need[2 * PB];
#pragma unroll
      for (int i = 0; i < 2 * PB; ++i) need[i] = act[i >> 1] && t < T;
      float v[2 * PB];
      unsigned tg[2 * PB];
      lazy_wait(attn ? g.efirst : g.clazy);  // the energies cannot arrive before the attention role has run
      const unsigned np_ = gather<2 * PB>(g.ep, (unsigned)((p * GS * ATTN_CU + j) * EP_LD + t), 4u * EP_LD, want, need, v, tg, pc);
      PROF_POLLS(2, np_);
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        float e = v[2 * b] + v[2 * b + 1];
        e += dpp_move<0xB1, 0xf>(0.f, e);  // quad_perm:[1,0,3,2]
        e += dpp_move<0x4E, 0xf>(0.f, e);  // quad_perm:[2,3,0,1]
        if (j == 0 && act[b]) s_e[b * TP + t] = (t < T && t < nv[b]) ? e : -INFINITY;  // mask, mod.rs:219-220
      }
    }
    __syncthreads();
    PROF_MARK(5);  // wait e_part
    PROF_WHEN(3);
#pragma unroll
    for (int b = 0; b < PB; ++b)
      if (act[b]) {  // every wave: the softmax in registers, lane <-> steps lane, lane + 64
        const float e0 = s_e[b * TP + lane], e1 = s_e[b * TP + lane + 64];
        const float m = wave_max(fmaxf(e0, e1));
        const float x0 = fast_exp(e0 - m), x1 = fast_exp(e1 - m);
        const float rs = __builtin_amdgcn_rcpf(wave_sum(x0 + x1));
        wreg[b][0] = x0 * rs;
        wreg[b][1] = x1 * rs;
      }
    if (attn && act_r && wave == 0) {  // the attention role keeps them for the next step's location features
      s_aw[lane] = wreg[rb][0];
      s_awc[lane] += wreg[rb][0];
      s_aw[lane + 64] = wreg[rb][1];
      s_awc[lane + 64] += wreg[rb][1];
    }
    PROF_MARK(6);  // softmax
    straggle(lag, s, 3);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P4: decoder LSTM (its context columns are folded into pmd) -----------------------------------
#pragma unroll
    for (int b = 0; b < PB; ++b)
      if (act[b]) {
        const float w0 = wreg[b][0], w1 = wreg[b][1];
        float a0 = fmaf(pmd[b][0][1], w1, fmaf(pmd[b][0][0], w0, dacc[b][0]));
        float a1 = fmaf(pmd[b][1][1], w1, fmaf(pmd[b][1][0], w0, dacc[b][1]));
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
          s_g[b * 16 + wave] = a0 + s_bias[16 + wave];
          s_g[b * 16 + wave + NW] = a1 + s_bias[16 + wave + NW];
        }
      }
    __syncthreads();
    if (cell && s_act[cb]) {
      const float *gp = s_g + cb * 16 + 4 * cu;
      const float ig = fast_sigmoid(gp[0]), fg = fast_sigmoid(gp[1]), gg = fast_tanh(gp[2]), og = fast_sigmoid(gp[3]);
      const float cn = fmaf(fg, s_cell[4 * PB + tid], ig * gg), hn = og * fast_tanh(cn);
      publish(g.hdec + (unsigned)((p * GS + cb) * DEC_RNN + 4 * c + cu), want, hn);
      s_cell[4 * PB + tid] = cn;
      s_cell[12 * PB + tid] = hn;
    }
    att_bulk(L4, false);  // for step s+1: ctx(s), h_att(s)
    PROF_MARK(7);  // dec tail + cell + publish + att bulk
    PROF_WHEN(1);
    straggle(lag, s, 4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P5: h_dec(s) -> projection rows ---------------------------------------------------------
    // The projection + prenet role hashes the Bernoulli(0.5) masks of step s+1 (they do not depend on the data) in the
    // time its first poll of h_dec could not succeed anyway.
    unsigned drop1 = 0u, drop2 = 0u;
    if (pre && act_r) {
      if (d.dropout_mode) {  // (mode 2: the caller's keep bytes of chunk rb -- views of a batch advance d.drop_masks)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          drop1 |= (prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 0, lane + 64 * k) ? 1u : 0u) << k;
#pragma unroll
        for (int r = 0; r < 2; ++r)
          drop2 |= (prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, rb, s + 1, 1, 16 * rk + wave + NW * r) ? 1u : 0u) << r;
      }
    }
    {
      // both halves of every active chunk's vector in flight together: granule tid + 512 i, i = 2 b + half
      bool need[2 * PB];
#pragma unroll
      for (int i = 0; i < 2 * PB; ++i) need[i] = act[i >> 1];
      float v[2 * PB];
      unsigned tg[2 * PB];
      lazy_wait(pre ? g.pfirst : g.lazy);  // only the projection role needs h_dec at once
      const unsigned np_ = gather<2 * PB>(g.hdec, (unsigned)(p * GS * DEC_RNN + tid), PT, want, need, v, tg, pc);
      PROF_POLLS(3, np_);
#pragma unroll
      for (int i = 0; i < 2 * PB; ++i)
        if (need[i]) s_hdec[TID + PT * i] = v[i];
    }
    __syncthreads();
    PROF_MARK(8);  // wait h_dec
    if (prow_ok && act_r) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) a = dot4(lds4(s_pw + 4 * (j * PT + TID)), lds4(s_hdec + rb * DEC_RNN + 256 * j + L4), a);
      a = fmaf(pmp[1], wreg[rb][1], fmaf(pmp[0], wreg[rb][0], a));  // the context columns
      a = wave_sum(a);
      if (lane == 0) publish(g.mel + (unsigned)(((p * GS + rb) * MEL_GL + prow) * MEL_ST), want, a + s_pb[wave]);
    }
    dec_bulk_h(L4, false);  // for step s+1
    if (attn && act_r) location(tid);
    PROF_MARK(9);  // projection rows + dec bulk + location
    straggle(lag, s, 5);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P6 (projection + prenet role): frame s, stop rule, x(s+1) ------------------------------
    if (pre && act_r) {  // a chunk's last x (active bit clear) is published at the step it stops
      bool nxt = false;
      if (act_r) {
        if (tid < N_MEL + 1) {
          const bool need[1] = {true};
          float v[1];
          unsigned tg[1];
          gather<1>(g.mel, (unsigned)(((p * GS + rb) * MEL_GL + tid) * MEL_ST), 0, want, need, v, tg, pc);
          s_mel[tid] = v[0];
        }
        __syncthreads();
        PROF_MARK(11);  // prenet role: mel gathered
        const float gate = s_mel[N_MEL];
        const bool fired = GATE ? gate_fires(gate, d.gate_lo, d.gate_hi, d.gate_threshold) : (d.use_gate && gate_sigmoid(gate) > d.gate_threshold);  // mod.rs:319-324
        if (rk == 0) {
          if (tid < N_MEL) d.frames[((size_t)rb * d.max_steps + s) * N_MEL + tid] = s_mel[tid];
          if (tid == 0) {
            d.gates[(size_t)rb * d.max_steps + s] = gate;
            if (fired) d.nframes[rb] = s + 1;  // the tripping frame is kept
          }
        }
        if (fired) nf_r = s + 1;
        nxt = s + 1 < nf_r;
      }
      float xo[2] = {0.f, 0.f};
      if (nxt) {
        // layer 1: output tid & 255, inputs [40 hf, 40 hf + 40), hf = tid >> 8
        const unsigned HM = (unsigned)((tid >> 8) * (N_MEL / 2));
        float acc = 0.f;
        if (ROLE_REGS) {
#pragma unroll
          for (int k = 0; k < N_MEL / 2; k += 4) {  // (16-byte broadcast reads of the frame)
            const float4 m = lds4(s_mel + HM + k);
            acc = fmaf(rw[ROLE_REGS ? k : 0], m.x, acc);
            acc = fmaf(rw[ROLE_REGS ? k + 1 : 0], m.y, acc);
            acc = fmaf(rw[ROLE_REGS ? k + 2 : 0], m.z, acc);
            acc = fmaf(rw[ROLE_REGS ? k + 3 : 0], m.w, acc);
          }
        } else {
#pragma unroll
          for (int k = 0; k < N_MEL / 2; k += 4) {
            const float4 w4 = lds4(s_W0 + 4u * (((HM + k) >> 2) * PRENET + (TID & 255u))), m = lds4(s_mel + HM + k);
            acc = fmaf(w4.x, m.x, acc);
            acc = fmaf(w4.y, m.y, acc);
            acc = fmaf(w4.z, m.z, acc);
            acc = fmaf(w4.w, m.w, acc);
          }
        }
        PROF_MARK(12);  // prenet role: gate, frame store, layer-1 products
        s_l1[TID] = acc;
        __syncthreads();
        PROF_MARK(13);  // prenet role: layer-1 barrier
        // every wave finishes layer 1 for the inputs its lanes consume (ReLU, dropout) -- no second
        // LDS round trip -- and reduces its two layer-2 columns
        float pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = fmaxf(s_l1[(L4 >> 2) + 64 * k] + s_l1[PRENET + (L4 >> 2) + 64 * k], 0.f);
          pk[k] = (drop1 >> k) & 1u ? 0.f : (d.dropout_mode ? 2.f * v : v);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) a = fmaf(w1r[r][k], pk[k], a);
          a = fmaxf(wave_sum(a), 0.f);
          xo[r] = (drop2 >> r) & 1u ? 0.f : (d.dropout_mode ? 2.f * a : a);
        }
      }
      // The workgroup's 16 columns leave as ONE 128-byte store (lanes 0..15 of wave 0).  Eight waves
      // writing two granules each into the same 128-byte line cost the x edge ~1 us per step (partial
      // write-through writes to one line serialise at the memory side); stores to different lines
      // (the mel rows, 128 B apart) do not show the effect.
      PROF_MARK(14);  // prenet role: layer 2
      if (lane < 2) s_mel[MEL_GL - 16 + wave + NW * lane] = lane ? xo[1] : xo[0];  // s_mel[81..95] is unused padding
      __syncthreads();
      if (tid < 16)
        publish(g.x + (unsigned)(((p ^ 1) * GS + rb) * PRENET + 16 * rk + tid), (want + 1u) | (nxt ? ACT_BIT : 0u),
                s_mel[MEL_GL - 16 + tid]);
    }
    PROF_MARK(10);  // prenet role: wait mel + frame store + prenet + publish x
    }
  }
#ifdef XDTTS_PERSIST_PROFILE
  __syncthreads();
  if (g.prof && tid < 24) g.prof[c * 24 + tid] = s_prof[tid];
#endif

  // ---- write the state back (a later launch may continue the sequence) -----------------------
  if (cell) {
    d.att_c[cb * ATT_RNN + 4 * c + cu] = s_cell[tid];
    d.dec_c[cb * DEC_RNN + 4 * c + cu] = s_cell[4 * PB + tid];
    d.att_h[0][cb * ATT_RNN + 4 * c + cu] = s_cell[8 * PB + tid];
    d.dec_h[0][cb * DEC_RNN + 4 * c + cu] = s_cell[12 * PB + tid];
  }
  if (attn) {
    // the context of the last step, for the state only: own 64 columns = sum_t w_t memory[t]
    {
      const int col = tid & 63, q = tid >> 6;
      float acc = 0.f;
      for (int t = q; t < T; t += NW) acc = fmaf(s_aw[t], d.memory[((size_t)rb * T + t) * EMB + 64 * rk + col], acc);
      __syncthreads();
      s_part[tid] = acc;
      __syncthreads();
      if (tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int u = 0; u < NW; ++u) v += s_part[u * 64 + tid];
        d.ctx[rb * EMB + 64 * rk + tid] = v;
      }
    }
    if (rk == 0 && tid < T) {
      d.aw[rb * T + tid] = s_aw[tid];
      d.awc[rb * T + tid] = s_awc[tid];
    }
  }
  if (c == 0 && tid == 0) d.ctl[0] = s;
}

// x(0) = prenet(0) = 0 (the prenet has no bias, mod.rs:208) with the chunks' initial active bits
// ... and clears the whole exchange first (it was a fill of its own): thread (b, i) zeroes words b 256 + i + k (B 256), the
// first of which is the slot it then seeds -- same thread, program order
__global__ void k_persist_seed(PersistBufs g, const int *limits, int B, unsigned words) {
  const int b = blockIdx.x, i = threadIdx.x;
  for (unsigned w = (unsigned)(b * PRENET + i); w < words; w += (unsigned)(B * PRENET)) g.x[w] = 0ull;
  publish(g.x + (size_t)b * PRENET + i, 1u | (limits[b] > 0 ? ACT_BIT : 0u), 0.f);
}

// parity hook: x(step) from a caller-held state, computed by the launch-per-stage prenet kernel into x [B][256]
__global__ void k_persist_seed_at(PersistBufs g, const int *limits, const float *x, int step) {
  const int b = blockIdx.x, i = threadIdx.x;
  publish(g.x + (size_t)(((step & 1) * GS + b) * PRENET + i), (unsigned)(step + 1) | (limits[b] > step ? ACT_BIT : 0u), x[b * PRENET + i]);
}

// ctx_w [CTXF_ROWS][512]: row n = the context columns of attention-LSTM row n (packed order), decoder-LSTM row n - 4096,
// projection row n - 8192; zero rows behind.  One float4 per thread.
__global__ void k_pack_ctx_rows(const float4 *__restrict__ att_w, const float4 *__restrict__ dec_w, const float4 *__restrict__ proj_w,
                                float4 *__restrict__ out) {
  const int n = blockIdx.x, q = threadIdx.x;  // q < 128: columns 4 q .. 4 q + 3
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < 4 * ATT_RNN) v = att_w[(size_t)n * (ATT_COLS / 4) + PRENET / 4 + q];
  else if (n < 4 * ATT_RNN + 4 * DEC_RNN) v = dec_w[(size_t)(n - 4 * ATT_RNN) * (DEC_COLS / 4) + ATT_RNN / 4 + q];
  else if (n <= 4 * ATT_RNN + 4 * DEC_RNN + N_MEL) v = proj_w[(size_t)(n - 4 * ATT_RNN - 4 * DEC_RNN) * (PROJ_IN / 4) + DEC_RNN / 4 + q];
  out[(size_t)n * (EMB / 4) + q] = v;
}

template <int PB, bool SKEW = false>
void launch_pb(const DecoderBufs &d, const PersistBufs &g, const PersistWeights &pw, int nsteps, hipStream_t s) {
  const void *fn = d.use_gate ? reinterpret_cast<const void *>(k_decoder_persistent<PB, SKEW, true>) : reinterpret_cast<const void *>(k_decoder_persistent<PB, SKEW, false>);
  COOP_CHECK(launch_coresident(true, fn, dim3(P_NCU), dim3(PT), 0, s, d, g, pw, nsteps));
}

}  // namespace

void launch_pack_ctx_rows(const float *att_w, const float *dec_w, const float *proj_w, float *ctx_w, hipStream_t s) {
  static_assert(CTXF_ROWS >= 4 * ATT_RNN + 4 * DEC_RNN + N_MEL + 1 && CTXF_LD >= PERSIST_T_MAX, "context-fold table shape");
  hipLaunchKernelGGL(k_pack_ctx_rows, dim3(CTXF_ROWS), dim3(EMB / 4), 0, s, reinterpret_cast<const float4 *>(att_w),
                     reinterpret_cast<const float4 *>(dec_w), reinterpret_cast<const float4 *>(proj_w), reinterpret_cast<float4 *>(ctx_w));
  HIP_CHECK(hipGetLastError());
}

size_t persist_granule_words(int B) {
  (void)B;
  return (size_t)2 * GS * (PRENET + ATT_RNN + ATTN_CU * EP_LD + DEC_RNN + MEL_GL * MEL_ST);
}

PersistBufs persist_bufs(unsigned long long *base, int *err, int B) {
  PersistBufs g{};
  g.x = base;
  (void)B;
  g.hatt = g.x + (size_t)2 * GS * PRENET;
  g.ep = g.hatt + (size_t)2 * GS * ATT_RNN;
  g.hdec = g.ep + (size_t)2 * GS * ATTN_CU * EP_LD;
  g.mel = g.hdec + (size_t)2 * GS * DEC_RNN;
  g.err = err;
  g.lazy = PERSIST_LAZY_DEFAULT;
  g.xlazy = 0;  // round 3 re-sweep (profiles/r03_lazy_sweep.txt): 0 / 1 / 2 / 3 / 4 -> 9.93 / 10.03 / 10.14 / 10.29 / 10.52 us per 1-chunk step
  g.clazy = 4;
  g.shrink = 0;
  g.spins = 0;
  g.fault = 0;
  g.slow = 0;
  g.pfirst = 0;  // x 256 clocks (behind the mask hashing; 0 / 1 / 2 / 3 -> 8.44 / 8.33 / 8.71 / 8.70 us per 1-chunk step)
  g.xfirst = 4;
  g.efirst = 3;  // (with pfirst 0 and lazy 9: 8.38 / 11.39 -> 8.19 / 11.19 us per step, 1 / 2 chunks)
  g.skew = 1;
  g.both_run = 0;
  g.first = 4;  // x 256 clocks, ~0.4 us: the first polls of a critical consumer cannot succeed earlier (measured: -0.3 us per step)
  return g;
}

PersistBufs persist_view(const PersistBufs &g, int b0) {
  PersistBufs v = g;
  v.x += (size_t)b0 * PRENET;
  v.hatt += (size_t)b0 * ATT_RNN;
  v.ep += (size_t)b0 * ATTN_CU * EP_LD;
  v.hdec += (size_t)b0 * DEC_RNN;
  v.mel += (size_t)b0 * MEL_GL * MEL_ST;
  return v;
}

// The grid must be co-resident: one workgroup per CU on a 256-CU part, nothing else of ours running.
bool decoder_persistent_supported(int device, int B, int T) {
  if (B < 1 || B > PERSIST_B_MAX || T > PERSIST_T_MAX) return false;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  if (prop.multiProcessorCount < P_NCU) return false;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_decoder_persistent<PERSIST_B_MAX, false, true>, PT, 0) != hipSuccess)
    return false;
  return per_cu >= 1;
}

void launch_persist_seed(const DecoderBufs &d, const PersistBufs &g, const int *limits_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_persist_seed, dim3(d.B), dim3(PRENET), 0, s, g, limits_dev, d.B, (unsigned)persist_granule_words(d.B));
  HIP_CHECK(hipGetLastError());
}

void launch_persist_seed_at(const DecoderBufs &d, const PersistBufs &g, const int *limits_dev, int step, hipStream_t s) {
  HIP_CHECK(hipMemsetAsync(g.x, 0, persist_granule_words(d.B) * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_persist_seed_at, dim3(d.B), dim3(PRENET), 0, s, g, limits_dev, d.x, step);
  HIP_CHECK(hipGetLastError());
}

void launch_decoder_persistent(const DecoderBufs &d, const DeviceWeights &w, const PersistBufs &g, int nsteps,
                               hipStream_t s) {
  PersistWeights pw{};
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
  switch (d.B) {
    case 1: launch_pb<1>(d, g, pw, nsteps, s); break;
    case 2:
      // a pair that ends with its first chunk -- or whose chunks both run every step of the launch -- runs the skewed loop
      // (PersistBufs::skew = 0: lock-step, the parity hooks' form)
      if (g.skew && (g.shrink || g.both_run)) launch_pb<2, true>(d, g, pw, nsteps, s);
      else launch_pb<2>(d, g, pw, nsteps, s);
      break;
    default: fail(XDTTS_ERR_BAD_ARG, "persistent decoder: %d chunks (max %d)", d.B, PERSIST_B_MAX);
  }
}

}  // namespace xdtts
