// decoder.hip -- the Tacotron2 autoregressive decoder step as a chain of HIP kernels for gfx950:
// the launch-per-stage engine.  Small lock-step batches (1-4 chunks, T <= 128) run the persistent
// weight-stationary kernel of decoder_persistent.hip instead (api.cpp: run_decoder); this file serves
// the batched MFMA form for >= 5 chunks (three launches per step: k_prenet_b, k_att_lstm_attention,
// k_lstm_mfma<DEC>, see the comments at those kernels), longer encoder memories, and is the second
// implementation the parity tests compare the persistent kernel with.
//
// Replaces the per-frame `self.decoder.run(inputs)` of the reference (src/tacotron2/mod.rs:304,
// graph decoder_iter.onnx) including the host-side stop test (mod.rs:319-324), which runs on the
// device here so the host is out of the loop.  B independent chunks advance in lock-step; a chunk
// is active at step s iff s < nframes[b].
//
// A step is a chain of dependent launches, and on this chip a dependent launch costs ~1.6 us before
// it does anything (tools/ubench_chain.hip), so the design minimises the number of grid-wide
// dependencies without serialising work onto few CUs.  The textbook chain has six stages (prenet,
// attention LSTM, query+energies, softmax/context, decoder LSTM, projection); the projection
// needs ALL of h_dec and ctx, which is what makes it a stage.  Here the producers of h_dec and ctx
// emit, in their epilogue, the partial products of their own hidden units / context columns with
// the matching columns of W_p, and the FIRST kernel of the next step sums the 264 partial vectors
// in a fixed order -- the projection stage disappears:
//
//   k_prenet      mel(s-1) = b + sum of partials -> frames[s-1], gate, stop rule; prenet -> x
//   k_lstm<ATT>   attention LSTM cell (29.4 MB weight stream)
//   k_qenergy     query rows + partial energies, spread over 32 blocks by attention dimension
//   k_softmax_ctx masked softmax, context (8 blocks), partial mel of the context columns
//   k_lstm<DEC>   decoder LSTM cell (42 MB weight stream) + partial mel of its hidden units;
//                 leading blocks compute the NEXT step's location features
// (The same trick for the query -- partial q from the attention-LSTM blocks, energies + softmax +
// context merged into one launch -- was built and measured: the merged kernel computes all 12.8k
// tanh terms redundantly in each of its 8 blocks and took 7.9 us against 2.4 + 3.3 us for the two
// distributed launches, so the query stage stays.)
//
// The LSTM GEMVs stream 71.3 MB of fp32 weights per step and are the HBM-bound part; rows are
// packed [unit][gate][cols] so each wave reads one contiguous 4-row slab with 16-byte
// lane-consecutive non-temporal loads (1 KiB per wave instruction) and owns a hidden unit
// end-to-end, which fuses the cell update into the GEMV.
//
// Latency discipline: every kernel costs ONE memory round trip: all weight and activation loads
// are issued at kernel entry.  The ping-pong parity of the recurrent state and the position `i`
// of the step inside the replayed graph are kernel arguments; the absolute step is ctl[0] + i,
// with ctl[0] advanced once per replay (k_advance).  Wavefront = 64 everywhere.
#include <algorithm>
#include <cstdlib>
#include <string>

#include "device_utils.h"
#include "kernels.h"

namespace xdtts {

namespace {

constexpr int NBLK = ATT_RNN / 4;   // 256 LSTM blocks, 4 hidden units each (both LSTMs)
constexpr int MEL_LD = 84;          // 80 mel + gate, padded to a float4 multiple
constexpr int CTX_BLOCKS = 8, CTX_COLS = EMB / CTX_BLOCKS;
constexpr int PM_ROWS = CTX_BLOCKS + NBLK;  // partial-mel rows per chunk: 8 ctx blocks, then 256 h blocks
static_assert(ATT_RNN == DEC_RNN, "both LSTMs use 256 blocks of 4 units");
typedef float f32x4 __attribute__((ext_vector_type(4)));

// DecoderState::new (mod.rs:202-233): all recurrent state zero.
__global__ void k_decoder_init(DecoderBufs d, const int *limits) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < ATT_RNN; i += blockDim.x) {
    d.att_h[0][b * ATT_RNN + i] = 0.f;
    d.att_h[1][b * ATT_RNN + i] = 0.f;
    d.att_c[b * ATT_RNN + i] = 0.f;
    d.dec_h[0][b * DEC_RNN + i] = 0.f;
    d.dec_h[1][b * DEC_RNN + i] = 0.f;
    d.dec_c[b * DEC_RNN + i] = 0.f;
  }
  for (int i = threadIdx.x; i < d.T; i += blockDim.x) {
    d.aw[b * d.T + i] = 0.f;
    d.awc[b * d.T + i] = 0.f;
  }
  for (int i = threadIdx.x; i < EMB; i += blockDim.x) d.ctx[b * EMB + i] = 0.f;
  // step 0: location features of all-zero attention weights
  for (int i = threadIdx.x; i < d.T * ATT_DIM; i += blockDim.x) d.loc[(size_t)b * d.T * ATT_DIM + i] = 0.f;
  // partial-mel rows are summed unconditionally: padding columns 81..83 and a first step's rows must read as zero (this chunk's
  // PM_ROWS x MEL_LD share; it was a fill of its own in front of this kernel)
  for (int i = threadIdx.x; i < PM_ROWS * MEL_LD; i += blockDim.x) d.pmel[(size_t)b * PM_ROWS * MEL_LD + i] = 0.f;
  if (threadIdx.x == 0) {
    d.nframes[b] = limits[b];
    if (b == 0) d.ctl[0] = 0;
  }
  if (d.xf && b == 0) {  // batched mode: the MFMA-operand copies, padding columns included
    const int n = blockDim.x, t = threadIdx.x;
    for (int i = t; i < PRENET * d.Bpad; i += n) d.xf[i] = 0.f;
    for (int i = t; i < EMB * d.Bpad; i += n) d.ctxf[i] = 0.f;
    for (int i = t; i < ATT_RNN * d.Bpad; i += n) {
      d.att_hf[0][i] = 0.f;
      d.att_hf[1][i] = 0.f;
      d.dec_hf[0][i] = 0.f;
      d.dec_hf[1][i] = 0.f;
    }
  }
}

__global__ void k_advance(DecoderBufs d, int n) { d.ctl[0] += n; }

// Location features of D3 for one (chunk, LOC_TT-step tile):
//   loc[t][a] = Dense32->128(Conv1d(2->32, k=31, pad=15)([w_prev ; w_cum]))[t][a]
// They depend only on the previous step's attention weights, so these blocks ride along in the
// previous step's k_dec_lstm launch instead of sitting on the attention critical path.
// loc_convT is the conv weight re-laid as [c][k][f] (filter index contiguous) so each thread pulls
// its 62 taps with lane-consecutive loads and keeps them in registers; the zero-padded weight
// windows are the only LDS operands of the conv.
constexpr int LOC_TT = 8;  // time steps per location block

struct LocWeights {
  float cw[2 * LOC_K];  // conv taps of filter tid & 31 (conv role)
  float wd[LOC_F];      // dense weights of attention dim tid & 127 (dense role)
};
__device__ __forceinline__ void location_weights(LocWeights &lw, const float *__restrict__ loc_convT,
                                                 const float *__restrict__ loc_denseT) {
  const int tid = threadIdx.x, f = tid & 31, a = tid & 127;
#pragma unroll
  for (int j = 0; j < 2 * LOC_K; ++j) lw.cw[j] = loc_convT[j * LOC_F + f];
#pragma unroll
  for (int g = 0; g < LOC_F; ++g) lw.wd[g] = loc_denseT[g * ATT_DIM + a];
}
// One LOC_TT-step tile of chunk b from the chunk's attention weights aw / awc (length T; global or LDS).
__device__ __forceinline__ void location_tile(const DecoderBufs &d, int b, int tile, const float *aw, const float *awc,
                                              const LocWeights &lw) {
  constexpr int TT = LOC_TT, PADK = (LOC_K - 1) / 2, WIN = TT + 2 * PADK;
  const int t0 = tile * TT, tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float s_w[2][WIN + 2], s_lc[TT][LOC_F];
  const int f = tid & 31, tl = tid >> 5;     // conv role: one (t, filter) output per thread
  const int a = tid & 127, th = tid >> 7;    // dense role: attention dim a, 4 time steps
  (void)f;
  __syncthreads();  // a previous tile's readers are done with s_w / s_lc
  if (tid < 2 * WIN) {
    const int c = tid / WIN, i = tid % WIN, t = t0 - PADK + i;
    const float *src = c ? awc : aw;  // channel 0 = previous weights, 1 = cumulative
    s_w[c][i] = (t >= 0 && t < d.T) ? src[t] : 0.f;
  }
  __syncthreads();
  {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < LOC_K; ++k) acc = fmaf(lw.cw[k], s_w[0][tl + k], acc);
#pragma unroll
    for (int k = 0; k < LOC_K; ++k) acc = fmaf(lw.cw[LOC_K + k], s_w[1][tl + k], acc);
    s_lc[tl][f] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TT / 2; ++q) {
    const int tloc = th * (TT / 2) + q, t = t0 + tloc;
    if (t >= d.T) break;
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < LOC_F; ++g) acc = fmaf(lw.wd[g], s_lc[tloc][g], acc);
    d.loc[((size_t)b * d.T + t) * ATT_DIM + a] = acc;
  }
}
__device__ __forceinline__ void location_role(const DecoderBufs &d, int b, int tile,
                                              const float *__restrict__ loc_convT,
                                              const float *__restrict__ loc_denseT) {
  LocWeights lw;
  location_weights(lw, loc_convT, loc_denseT);
  location_tile(d, b, tile, d.aw + b * d.T, d.awc + b * d.T, lw);
}

// D5 + D6 + D1.  First launch of step s.
//  * projection of the PREVIOUS step, finished here: mel(s-1)[m] = b_p[m] + sum of the 8 context
//    partials (k_attention) and the 256 hidden-state partials (k_dec_lstm), m = 80 is the gate
//    logit.  Block 0 of the chunk stores frames[s-1] / gates[s-1] and applies the stop rule of
//    mod.rs:319-324 (sigmoid(gate) > threshold; the tripping frame is kept): nframes[b] = s.
//    Every block takes the same decision from the same sums.
//  * prenet: x = relu(W1 relu(W0 mel) * m0 * 2) * m1 * 2, no bias, Bernoulli(0.5) masks from the
//    counter RNG (the exported graph keeps this dropout on at inference).  Spread over
//    PRENET_BLOCKS blocks per chunk: every block recomputes layer 1 (80 KB, L2-resident) and owns
//    256/PRENET_BLOCKS output columns of layer 2.
// `flush` (after the last step of a sequence): only the projection part runs.
constexpr int PRENET_BLOCKS = 16, PRENET_COLS = PRENET / PRENET_BLOCKS;

__global__ __launch_bounds__(256) void k_prenet(DecoderBufs d, int i, int flush, const float *__restrict__ W0T,
                                                const float *__restrict__ W1T,
                                                const float *__restrict__ proj_b) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / PRENET_BLOCKS, jblk = blockIdx.x % PRENET_BLOCKS;
  __shared__ __attribute__((aligned(16))) float s_mel[MEL_LD], s_x1[PRENET], s_part[4][PRENET], s_out[4][PRENET_COLS];
  __shared__ __attribute__((aligned(16))) float s_red[8][MEL_LD];
  // partial-mel rows: thread (m4, part) sums rows part, part+8, ... of the chunk's 264 rows for
  // the four mel bins 4*m4..4*m4+3 -- 33 independent 16-byte loads
  const int m4 = tid & 31, part = tid >> 5;
  const float4 *pm = reinterpret_cast<const float4 *>(d.pmel + (size_t)b * PM_ROWS * MEL_LD);
  float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m4 < MEL_LD / 4) {
    float4 rv[PM_ROWS / 8];
#pragma unroll
    for (int k = 0; k < PM_ROWS / 8; ++k) rv[k] = pm[(size_t)(part + 8 * k) * (MEL_LD / 4) + m4];
#pragma unroll
    for (int k = 0; k < PM_ROWS / 8; ++k) {
      racc.x += rv[k].x;
      racc.y += rv[k].y;
      racc.z += rv[k].z;
      racc.w += rv[k].w;
    }
  }
  // prenet weights, 16 B per lane.  Layer 1: wave w covers inputs [20w, 20w+20) for the four
  // output columns 4*lane..4*lane+3.  Layer-2 slice (PRENET_COLS = 16 columns of this block):
  // thread (c4, ig) covers inputs ig, ig+64, ig+128, ig+192 for columns col0 + 4*c4 .. +3.
  const float4 *W0 = reinterpret_cast<const float4 *>(W0T), *W1 = reinterpret_cast<const float4 *>(W1T);
  constexpr int L1_PER_WAVE = N_MEL / 4;
  float4 w0[L1_PER_WAVE];
  float4 w1[4];
  const int c4 = tid & 3, ig = tid >> 2, col0 = jblk * PRENET_COLS;
  if (!flush) {
#pragma unroll
    for (int k = 0; k < L1_PER_WAVE; ++k) w0[k] = W0[(wave * L1_PER_WAVE + k) * (PRENET / 4) + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) w1[k] = W1[((size_t)(ig + 64 * k) * PRENET + col0) / 4 + c4];
  }
  const float bias = tid < N_MEL + 1 ? proj_b[tid] : 0.f;
  const int step = d.ctl[0] + i;
  const int nf = d.nframes[b];
  const int chunk = d.item_perm ? d.item_perm[b] : b;
  const uint32_t item = d.item_base + (uint32_t)chunk;
  if (m4 < MEL_LD / 4) *reinterpret_cast<float4 *>(&s_red[part][4 * m4]) = racc;
  __syncthreads();
  const bool forced = d.dec_in != nullptr && !flush;  // parity hook: the caller supplies decoder_input
  const bool have_prev = !forced && step >= 1 && step - 1 < nf;  // the chunk was active at the previous step
  if (tid < MEL_LD) {
    float v = 0.f;
    if (have_prev && tid < N_MEL + 1) {
      v = bias;
#pragma unroll
      for (int k = 0; k < 8; ++k) v += s_red[k][tid];
    }
    if (forced && tid < N_MEL) v = d.dec_in[b * N_MEL + tid];
    s_mel[tid] = v;  // step 0: decoder_input = 0 (mod.rs:208)
  }
  __syncthreads();
  const float gate = s_mel[N_MEL];
  const bool fired = have_prev && d.use_gate && gate_fires(gate, d.gate_lo, d.gate_hi, d.gate_threshold);
  if (jblk == 0 && have_prev) {
    if (tid < N_MEL) d.frames[((size_t)b * d.max_steps + (step - 1)) * N_MEL + tid] = s_mel[tid];
    if (tid == 0) {
      d.gates[(size_t)b * d.max_steps + (step - 1)] = gate;
      if (fired) d.nframes[b] = step;  // frame step-1 is the last one
    }
  }
  if (flush || fired || step >= nf) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < L1_PER_WAVE; ++k) {
    const float m = s_mel[wave * L1_PER_WAVE + k];
    acc.x = fmaf(w0[k].x, m, acc.x);
    acc.y = fmaf(w0[k].y, m, acc.y);
    acc.z = fmaf(w0[k].z, m, acc.z);
    acc.w = fmaf(w0[k].w, m, acc.w);
  }
  *reinterpret_cast<float4 *>(&s_part[wave][4 * lane]) = acc;
  __syncthreads();
  {
    float v = fmaxf((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]), 0.f);
    if (d.dropout_mode) v = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, chunk, step, 0, tid) ? 0.f : 2.f * v;
    s_x1[tid] = v;
  }
  __syncthreads();
  acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float m = s_x1[ig + 64 * k];
    acc.x = fmaf(w1[k].x, m, acc.x);
    acc.y = fmaf(w1[k].y, m, acc.y);
    acc.z = fmaf(w1[k].z, m, acc.z);
    acc.w = fmaf(w1[k].w, m, acc.w);
  }
  // reduce over the 16 input groups held by lanes with equal c4 (lane bits 2..5), then over waves
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64);
    acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64);
    acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (lane < 4) *reinterpret_cast<float4 *>(&s_out[wave][4 * lane]) = acc;
  __syncthreads();
  if (tid < PRENET_COLS) {
    float o = fmaxf((s_out[0][tid] + s_out[1][tid]) + (s_out[2][tid] + s_out[3][tid]), 0.f);
    const int j = col0 + tid;
    if (d.dropout_mode) o = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, chunk, step, 1, j) ? 0.f : 2.f * o;
    d.x[b * PRENET + j] = o;
    if (d.xf) d.xf[((size_t)(j >> 2) * d.Bpad + b) * 4 + (j & 3)] = o;
  }
}

// Batched form of k_prenet (B >= BATCH_MFMA_MIN).  With 16 blocks per chunk every block re-reads the
// chunk's 264 partial-mel rows (88 KB) and all of W0 (80 KB): 150 MB of L2 reads per step at 52 chunks.
// Here a chunk is PRENET_SPLIT blocks of 1024 threads: the partial rows, W0 and half of W1 are read
// once per block (22 MB per step in total), all of them in flight at kernel entry.
constexpr int PRENET_SPLIT = 2, PRENET_BT = 1024;
// Batched mode: the location features of step s depend only on the attention weights of step s-1, so they ride along
// in the prenet launch of step s as extra 1024-thread blocks (the prenet occupies 2 B of the 256 CUs for ~6 us): a
// block takes eight 8-step tiles of one chunk, four at a time, one per 256-thread group (two blocks per 100-step
// chunk: with the prenet's two that is 4 B blocks of 1024 threads, one round on 256 CUs up to 64 chunks).  As the tail of the context kernel -- where
// they were first -- they cost every lock-step iteration 5 us (94 weight registers per thread and two to four
// barrier rounds on its critical path).
__device__ __forceinline__ int loc_blocks_per_chunk(int T) { return ((T + LOC_TT - 1) / LOC_TT + 7) / 8; }
__device__ __forceinline__ void location_blocks(const DecoderBufs &d, int i, int lb, const float *__restrict__ loc_convT,
                                                const float *__restrict__ loc_denseT) {
  constexpr int TT = LOC_TT, PADK = (LOC_K - 1) / 2;
  const int T = d.T, per = loc_blocks_per_chunk(T), b = lb / per;
  const int tid = threadIdx.x, gq = tid >> 8, ltid = tid & 255;
  const int step = d.ctl[0] + i;
  const bool act = step < d.nframes[b];
  // the weights of step s-1: aw, and the cumulative ones the previous node's context kernel wrote (ping-pong by parity);
  // the whole chunk's go to LDS first -- loads retire in issue order, and the 94 filter weights below are slower
  const float *aw = d.aw + b * T, *awc = ((i & 1) ? d.awc2 : d.awc) + b * T;
  __shared__ __attribute__((aligned(16))) float s_aw[2][T_MAX + 2 * PADK], s_lc[4][TT][LOC_F];
  for (int k = tid; k < 2 * (T + 2 * PADK); k += PRENET_BT) {
    const int c = k / (T + 2 * PADK), t = k % (T + 2 * PADK) - PADK;
    s_aw[c][t + PADK] = (t >= 0 && t < T) ? (c ? awc[t] : aw[t]) : 0.f;  // zero-padded: channel 0 = previous weights, 1 = cumulative
  }
  LocWeights lw;
  location_weights(lw, loc_convT, loc_denseT);  // (filter tid & 31, dim tid & 127: the same within every 256-thread group)
  __syncthreads();
  const int f = ltid & 31, tl = ltid >> 5;   // conv role: one (t, filter) output per thread
  const int a = ltid & 127, th = ltid >> 7;  // dense role: attention dim a, 4 time steps
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const int tile = 8 * (lb % per) + 4 * round + gq, t0 = tile * TT;
    const bool on = act && t0 < T;
    if (round) __syncthreads();  // the first round's readers are done with s_lc
    if (on) {  // the window of output step t0 + tl starts at padded index t0 + tl
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < LOC_K; ++k) acc = fmaf(lw.cw[k], s_aw[0][t0 + tl + k], acc);
#pragma unroll
      for (int k = 0; k < LOC_K; ++k) acc = fmaf(lw.cw[LOC_K + k], s_aw[1][t0 + tl + k], acc);
      s_lc[gq][tl][f] = acc;
    }
    __syncthreads();
    if (on) {
#pragma unroll
      for (int q = 0; q < TT / 2; ++q) {
        const int tloc = th * (TT / 2) + q, t = t0 + tloc;
        if (t >= T) break;
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < LOC_F; g += 4) {  // (one 16-byte broadcast read per four products: the block is LDS-issue bound)
          const float4 lc = *reinterpret_cast<const float4 *>(&s_lc[gq][tloc][g]);
          acc = fmaf(lw.wd[g], lc.x, acc);
          acc = fmaf(lw.wd[g + 1], lc.y, acc);
          acc = fmaf(lw.wd[g + 2], lc.z, acc);
          acc = fmaf(lw.wd[g + 3], lc.w, acc);
        }
        d.loc[(((size_t)b * (ATT_DIM / 4) + (a >> 2)) * T + t) * 4 + (a & 3)] = acc;  // batched layout [B][32][T][4]
      }
    }
  }
}
// The same for T <= LOC_MFMA_T (the reference's window is 100) on the matrix cores, two blocks per chunk: both layers
// are GEMMs over the chunk's time steps --  lc[T][32] = im2col(w_prev, w_cum)[T][62] . conv[62][32]  and
// loc[T][128] = lc[T][32] . dense[32][128]  -- 224 + 448 v_mfma_f32_16x16x4_f32 per chunk against 608 k FMAs that each
// need an LDS operand (measured: the FMA form keeps a 1024-thread block busy for ~5 us, longer than the prenet it
// rides with).  Operands come from LDS: lane l of an A fragment is (row l % 16, k l / 16), of a B fragment
// (k l / 16, column l % 16); a D register r of lane l is (row 4 (l / 16) + r, column l % 16).
constexpr int LOC_MFMA_T = 128;
template <int NWV = PRENET_BT / 64>  // waves of the block: 16 inside the prenet launch, 8 inside the decoder-LSTM launch (two-launch form)
__device__ __forceinline__ void location_chunk_mfma(const DecoderBufs &d, int i, int lb, const float *__restrict__ loc_convT,
                                                    const float *__restrict__ loc_denseT) {
  static_assert(NWV == 8 || NWV == 16, "conv: one (time tile, filter half) per wave 0..7; dense: 32 / NWV time tiles per wave");
  const int b = lb >> 1, half = lb & 1;  // two blocks per chunk: time tiles 0..3 and 4..7
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int PADK = (LOC_K - 1) / 2, KC = 64;  // conv contraction: 2 x 31 taps, padded to 64
  const int T = d.T, MT = (T + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  __shared__ __attribute__((aligned(16))) float s_aw[2][LOC_MFMA_T + 2 * PADK + 2], s_lc[LOC_MFMA_T][LOC_F + 1];  // (+1: fragment reads walk the rows)
  const int step = d.ctl[0] + i;
  const bool act = step < d.nframes[b];
  const float *aw = d.aw + b * T, *awc = ((i & 1) ? d.awc2 : d.awc) + b * T;  // weights of step s-1 (cumulative: ping-pong by parity)
  constexpr int SZ = LOC_MFMA_T + 2 * PADK + 2;
  for (int k = tid; k < 2 * SZ; k += 64 * NWV) {
    const int c = k / SZ, t = k % SZ - PADK;
    s_aw[c][t + PADK] = (t >= 0 && t < T) ? (c ? awc[t] : aw[t]) : 0.f;  // zero-padded: channel 0 = previous weights, 1 = cumulative
  }
  // the weights go straight from global memory into B fragments (conv [c][k][f], rows 62 and 63 zero; dense [f][a]): one round trip,
  // in flight together with the attention weights above, no LDS copy in between
  float cwf[KC / 4], bw[LOC_F / 4];
  {
    const int nt = wave & 1;
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const int kk = 4 * ks + fg;
      cwf[ks] = (wave < 8 && kk < 2 * LOC_K) ? loc_convT[kk * LOC_F + 16 * nt + fi] : 0.f;
    }
    const int dt = wave & 7;
#pragma unroll
    for (int ks = 0; ks < LOC_F / 4; ++ks) bw[ks] = loc_denseT[(4 * ks + fg) * ATT_DIM + 16 * dt + fi];
  }
  __syncthreads();
  if (!act) return;  // (block-uniform)
  // ---- conv: (time tile mt, filter tile nt) per wave ----
  for (int tp = wave; tp < 8; tp += NWV) {  // (waves 0..7)
    const int mt = 4 * half + (tp >> 1), nt = tp & 1;
    if (mt >= MT) continue;
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;  // two chains: a dependent MFMA waits ~40 cycles
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const int kk = 4 * ks + fg, c = kk >= LOC_K, k = kk - (c ? LOC_K : 0);  // (kk = 62, 63: c = 1, k = 31, 32 -- weight rows are zero)
      const float av = s_aw[c][16 * mt + fi + k];
      if (ks & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cwf[ks], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cwf[ks], acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_lc[16 * mt + 4 * fg + r][16 * nt + fi] = acc0[r] + acc1[r];
  }
  __syncthreads();
  // ---- dense: (time tile mt, dim tile nt = wave % 8) per wave; the wave's eight B fragments stay in registers ----
  {
    const int nt = wave & 7;
    // its time tiles mt = 4 half + wave / 8 + (NWV / 8) j side by side: independent accumulator chains
    constexpr int NJ = 32 / NWV;
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < LOC_F / 4; ++ks)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int mt = 4 * half + (wave >> 3) + (NWV / 8) * j;  // (rows past the chunk's tiles hold stale LDS; their results are dropped)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_lc[16 * mt + fi][4 * ks + fg], bw[ks], acc[j], 0, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = 16 * (4 * half + (wave >> 3) + (NWV / 8) * j) + 4 * fg + r;
        if (t < T) d.loc[(((size_t)b * (ATT_DIM / 4) + 4 * nt + (fi >> 2)) * T + t) * 4 + (fi & 3)] = acc[j][r];  // batched layout [B][32][T][4]
      }
  }
}

__global__ __launch_bounds__(PRENET_BT) void k_prenet_b(DecoderBufs d, int i, int flush, const float *__restrict__ W0T,
                                                        const float *__restrict__ W1T, const float *__restrict__ proj_b,
                                                        const float *__restrict__ loc_convT, const float *__restrict__ loc_denseT) {
  if ((int)blockIdx.x >= PRENET_SPLIT * d.B) {  // location role (never in a flush launch: its grid ends with the prenet blocks)
    if (d.T <= LOC_MFMA_T)
      location_chunk_mfma(d, i, (int)blockIdx.x - PRENET_SPLIT * d.B, loc_convT, loc_denseT);
    else
      location_blocks(d, i, (int)blockIdx.x - PRENET_SPLIT * d.B, loc_convT, loc_denseT);
    return;
  }
#ifdef XDTTS_LSTM_PROBE
  unsigned long long pp[8];
  pp[0] = wall_clock64();
#define PPROBE(i) pp[i] = wall_clock64()
#else
#define PPROBE(i) do { } while (0)
#endif
  constexpr int HALF = PRENET / PRENET_SPLIT;          // layer-2 columns of this block
  // The partial-mel rows are dense in memory (84 floats = 21 16-byte vectors each, no gap between rows), so the reduction reads
  // them as one run of 264 x 21 vectors: thread (part = tid / 21, m4 = tid % 21) of the first 1008 takes vectors tid + 1008 k --
  // the same column every time (1008 = 48 x 21), rows part + 48 k -- with fully used 1 KiB wave loads; 6 loads per thread, where a
  // (32 columns, 21 used) x 32 row-group mapping needed 9 with a third of the lanes idle (the block is bound by the issue of its loads).
  constexpr int C4 = MEL_LD / 4;                       // 21 vectors per row
  constexpr int NPART = PRENET_BT / C4;                // 48 row groups of the partial-mel reduction
  constexpr int ROWS = (PM_ROWS + NPART - 1) / NPART;  // 6 rows per group (the last one only for the first 24 groups)
  static_assert(NPART * C4 <= PRENET_BT && PM_ROWS * MEL_LD % 4 == 0, "dense row mapping");
  constexpr int KG1 = PRENET_BT / 64, IN1 = N_MEL / KG1;       // layer 1: 16 input groups of 5
  constexpr int KG2 = PRENET_BT / (HALF / 4), IN2 = PRENET / KG2;  // layer 2: 32 input groups of 8
  const int tid = threadIdx.x;
  const int b = blockIdx.x / PRENET_SPLIT, half = blockIdx.x % PRENET_SPLIT;
  __shared__ __attribute__((aligned(16))) float s_red[NPART][MEL_LD], s_mel[MEL_LD], s_p1[KG1][PRENET], s_x1[PRENET], s_p2[KG2][HALF];
  // ---- every global load of the kernel, issued before anything is waited for ----
  const int m4 = tid % C4, part = tid / C4;  // (part >= NPART: the 16 spare threads)
  const float4 *pm = reinterpret_cast<const float4 *>(d.pmel + (size_t)b * PM_ROWS * MEL_LD);
  // (the small loads first: vmcnt retires in issue order, so a value loaded behind the weight blocks could not be used before them)
  // and the compiler must not look at them before the last big load has been issued: it would scalarise the three block-uniform
  // ones on the spot (v_readfirstlane behind s_waitcnt vmcnt(0): one exposed L2 round trip ahead of everything else) -- they are
  // handed to it through the asm below
  float bias_raw = proj_b[tid < N_MEL + 1 ? tid : N_MEL];
  int step_v = d.ctl[0], nf_v = d.nframes[b];
  int perm_v = *(d.item_perm ? d.item_perm + b : d.nframes + b);  // (unconditional load; the value is dropped without a permutation)
  asm volatile("" ::: "memory");
  // Every load below is UNCONDITIONAL (addresses clamped, values masked where they are consumed): a load under a condition is a
  // control-flow join, after which the compiler's waitcnt pass no longer knows how many younger loads are in flight and waits
  // for vmcnt(0) -- the row sum then sat behind W0 and W1 as well, i.e. behind all 296 KB of the block (seen in the ISA; in-kernel
  // clocks: 1.7 us between the last load's issue and the summed rows).  Straight-line, the rows are waited for with vmcnt(13),
  // layer 1 with vmcnt(8), and the two weight blocks arrive while the mel is summed, stored and judged by the stop rule.
  float4 rv[ROWS];
  const bool col_ok = part < NPART;
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const int row = part + NPART * k;
    rv[k] = pm[(size_t)(col_ok && row < PM_ROWS ? row : PM_ROWS - 1) * C4 + m4];
  }
  asm volatile("" ::: "memory");
  const int o4 = tid & 63, kg1 = tid >> 6;       // layer 1: outputs 4 o4 .. +3, inputs IN1 kg1 .. +IN1
  const int c4 = tid % (HALF / 4), kg2 = tid / (HALF / 4);  // layer 2: columns HALF half + 4 c4 .. +3, inputs IN2 kg2 .. +IN2
  float4 w0[IN1], w1[IN2];
#pragma unroll
  for (int k = 0; k < IN1; ++k) w0[k] = reinterpret_cast<const float4 *>(W0T)[(size_t)(kg1 * IN1 + k) * (PRENET / 4) + o4];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = 0; k < IN2; ++k) w1[k] = reinterpret_cast<const float4 *>(W1T)[((size_t)(kg2 * IN2 + k) * PRENET + HALF * half) / 4 + c4];
  asm volatile("" : "+v"(bias_raw), "+v"(step_v), "+v"(nf_v), "+v"(perm_v) : : "memory");
  const int step = step_v + i;
  const int nf = nf_v;
  const int chunk = d.item_perm ? perm_v : b;
  const uint32_t item = d.item_base + (uint32_t)chunk;
  PPROBE(1);
  // ---- projection of the previous step: sum of the 264 partial rows in a fixed order ----
  {
    float4 r = rv[0];  // (row `part` < NPART <= PM_ROWS always exists; the spare threads' sums are dropped)
#pragma unroll
    for (int k = 1; k < ROWS; ++k) {
      const bool ok = part + NPART * k < PM_ROWS;  // (the clamped duplicate of the last row otherwise)
      r.x += ok ? rv[k].x : 0.f;
      r.y += ok ? rv[k].y : 0.f;
      r.z += ok ? rv[k].z : 0.f;
      r.w += ok ? rv[k].w : 0.f;
    }
    if (col_ok) *reinterpret_cast<float4 *>(&s_red[part][4 * m4]) = r;
  }
  __syncthreads();
  PPROBE(2);
  const bool forced = d.dec_in != nullptr && !flush;  // parity hook: the caller supplies decoder_input
  const bool have_prev = !forced && step >= 1 && step - 1 < nf;  // the chunk was active at the previous step
  if (tid < MEL_LD) {
    float v = 0.f;
    if (have_prev && tid < N_MEL + 1) {
      v = bias_raw;
#pragma unroll
      for (int k = 0; k < NPART; ++k) v += s_red[k][tid];
    }
    if (forced && tid < N_MEL) v = d.dec_in[b * N_MEL + tid];
    s_mel[tid] = v;  // step 0: decoder_input = 0 (mod.rs:208)
  }
  __syncthreads();
  const float gate = s_mel[N_MEL];
  const bool fired = have_prev && d.use_gate && gate_fires(gate, d.gate_lo, d.gate_hi, d.gate_threshold);
  if (half == 0 && have_prev) {
    if (tid < N_MEL) d.frames[((size_t)b * d.max_steps + (step - 1)) * N_MEL + tid] = s_mel[tid];
    if (tid == 0) {
      d.gates[(size_t)b * d.max_steps + (step - 1)] = gate;
      if (fired) d.nframes[b] = step;  // frame step-1 is the last one (mod.rs:319-324)
    }
  }
  if (flush || fired || step >= nf) return;
  PPROBE(3);
  // ---- prenet layer 1 (every block, all 256 outputs) ----
  {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < IN1; ++k) {
      const float m = s_mel[kg1 * IN1 + k];
      acc.x = fmaf(w0[k].x, m, acc.x);
      acc.y = fmaf(w0[k].y, m, acc.y);
      acc.z = fmaf(w0[k].z, m, acc.z);
      acc.w = fmaf(w0[k].w, m, acc.w);
    }
    *reinterpret_cast<float4 *>(&s_p1[kg1][4 * o4]) = acc;
  }
  __syncthreads();
  if (tid < PRENET) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < KG1; ++k) v += s_p1[k][tid];
    v = fmaxf(v, 0.f);
    if (d.dropout_mode) v = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, chunk, step, 0, tid) ? 0.f : 2.f * v;
    s_x1[tid] = v;
  }
  __syncthreads();
  PPROBE(4);
  // ---- prenet layer 2, this block's HALF columns ----
  {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < IN2; ++k) {
      const float m = s_x1[kg2 * IN2 + k];
      acc.x = fmaf(w1[k].x, m, acc.x);
      acc.y = fmaf(w1[k].y, m, acc.y);
      acc.z = fmaf(w1[k].z, m, acc.z);
      acc.w = fmaf(w1[k].w, m, acc.w);
    }
    *reinterpret_cast<float4 *>(&s_p2[kg2][4 * c4]) = acc;
  }
  __syncthreads();
  if (tid < HALF) {
    float o = 0.f;
#pragma unroll
    for (int k = 0; k < KG2; ++k) o += s_p2[k][tid];
    o = fmaxf(o, 0.f);
    const int j = HALF * half + tid;
    if (d.dropout_mode) o = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, chunk, step, 1, j) ? 0.f : 2.f * o;
    d.x[b * PRENET + j] = o;
    d.xf[((size_t)(j >> 2) * d.Bpad + b) * 4 + (j & 3)] = o;
  }
#ifdef XDTTS_LSTM_PROBE
  PPROBE(5);
  if (tid == 0 && (step == 100 || step == 101) && (b == 1 || b == 30) && half == 0)
    printf("probe prenet_b chunk %d step %d: loads issued %llu  rows summed+sync %llu  mel+sync %llu  layer 1 %llu  layer 2 + store %llu (x10ns)\n", b, step,
           pp[1] - pp[0], pp[2] - pp[1], pp[3] - pp[2], pp[4] - pp[3], pp[5] - pp[4]);
#endif
}

// D2 / D4: LSTM cell as a weight-streaming GEMV with the cell update fused.  One wave owns one
// hidden unit: its four gate rows (i,f,g,o) are contiguous in the packed layout.  The rows are
// pulled into registers once (NCOLS/64 floats per lane per row) and reused for every chunk of
// the batch, so HBM sees each weight once per step regardless of B.
//   KIND 0: attention_rnn, input [prenet x (256) ; ctx_prev (512)] , hidden att_h   -> 1792 cols
//   KIND 1: decoder_rnn,   input [att_h_new (1024) ; ctx (512)]    , hidden dec_h   -> 2560 cols
//           epilogue: partial mel    pmel[8+blk][m] = sum_{u in block} W_p[m][u] h_dec[u]
//           leading blocks: location-feature role for the NEXT step
template <int NCOLS, int KIND>
__global__ __launch_bounds__(256) void k_lstm(DecoderBufs d, int i, int cur, const float4 *__restrict__ Wp,
                                              const float *__restrict__ bias,
                                              const float4 *__restrict__ Wepi,  // q_w4 / proj_wh4
                                              const float *__restrict__ loc_convT,
                                              const float *__restrict__ loc_denseT) {
  constexpr int NCH = NCOLS / 256;
  constexpr int N0 = KIND == 0 ? PRENET : ATT_RNN;  // first segment length
  constexpr int N1 = EMB;
  constexpr int EPI = KIND == 0 ? 0 : MEL_LD;  // epilogue outputs per block (decoder LSTM: partial mel)
  int blk = blockIdx.x;
  if (KIND == 1) {  // the first tiles*B blocks take the location role (short; dispatched first)
    const int tiles = (d.T + LOC_TT - 1) / LOC_TT, nloc = tiles * d.B;
    if ((int)blockIdx.x < nloc) {
      location_role(d, blockIdx.x / tiles, blockIdx.x % tiles, loc_convT, loc_denseT);
      return;
    }
    blk -= nloc;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int unit = blk * 4 + wave;
  float4 w[4][NCH];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < NCH; ++k) w[g][k] = ld_stream(Wp + ((size_t)(unit * 4 + g) * NCOLS) / 4 + lane + 64 * k);
  const float4 bz = *reinterpret_cast<const float4 *>(bias + unit * 4);
  const float4 we = (KIND == 1 && tid < EPI) ? Wepi[(size_t)blk * EPI + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
  const int step = d.ctl[0] + i;
  __shared__ __attribute__((aligned(16))) float s_h[4];
  for (int b = 0; b < d.B; ++b) {
    const float *seg0, *seg1, *seg2;
    float *h_out, *c;
    if (KIND == 0) {
      seg0 = d.x + b * PRENET;
      seg1 = d.ctx + b * EMB;
      seg2 = d.att_h[cur] + b * ATT_RNN;
      h_out = d.att_h[cur ^ 1] + b * ATT_RNN;
      c = d.att_c + b * ATT_RNN;
    } else {
      seg0 = d.att_h[cur ^ 1] + b * ATT_RNN;
      seg1 = d.ctx + b * EMB;
      seg2 = d.dec_h[cur] + b * DEC_RNN;
      h_out = d.dec_h[cur ^ 1] + b * DEC_RNN;
      c = d.dec_c + b * DEC_RNN;
    }
    if (b > 0 && step >= d.nframes[b]) continue;  // chunk 0's loads go out before the counter lands
    float4 xv[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int col = 256 * k;  // wave-uniform segment choice: boundaries are multiples of 256
      const float *src = col < N0 ? seg0 + col : (col < N0 + N1 ? seg1 + (col - N0) : seg2 + (col - N0 - N1));
      xv[k] = *reinterpret_cast<const float4 *>(src + 4 * lane);
    }
    const float c_old = c[unit];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      a0 = dot4(w[0][k], xv[k], a0);
      a1 = dot4(w[1][k], xv[k], a1);
      a2 = dot4(w[2][k], xv[k], a2);
      a3 = dot4(w[3][k], xv[k], a3);
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    a3 = wave_sum(a3);
    const bool act = step < d.nframes[b];
    if (lane == 0) {
      const float ig = sigmoidf_(a0 + bz.x), fg = sigmoidf_(a1 + bz.y);
      const float gg = tanhf(a2 + bz.z), og = sigmoidf_(a3 + bz.w);
      const float cn = fmaf(fg, c_old, ig * gg);
      const float hn = og * tanhf(cn);
      s_h[wave] = hn;
      if (act) {
        c[unit] = cn;
        h_out[unit] = hn;
      }
    }
    if (KIND == 1) {
      __syncthreads();
      if (act && tid < EPI) {
        const float4 h4 = *reinterpret_cast<const float4 *>(s_h);
        const float pv = fmaf(we.w, h4.w, fmaf(we.z, h4.z, fmaf(we.y, h4.y, we.x * h4.x)));
        d.pmel[((size_t)b * PM_ROWS + CTX_BLOCKS + blk) * MEL_LD + tid] = pv;
      }
      __syncthreads();
    }
  }
}

// D2 / D4 for batches (B >= BATCH_MFMA_MIN chunks in lock-step): the LSTM pre-activations become
// a GEMM  G[16 rows of a block][B chunks] = W[16 x K] . X^T[K x B]  on the exact-fp32 matrix
// cores (v_mfma_f32_16x16x4_f32).  Still weight-streaming: a block owns the same 4 hidden units
// (16 gate rows), its MFMA_WAVES (8) waves split K, every weight is read once per step -- from the Infinity
// Cache, which holds the 71 MB between steps as long as the loads are plain ones -- and feeds 4 MFMAs per
// 16-chunk tile straight from the load (the weights are pre-laid in fragment order, weights.h); the
// activations come from L2 with 64-byte segments per chunk.  Accumulators of the K-slices meet in LDS; wave t
// then holds, per lane, the four gates of (chunk 16t + lane%16, unit lane/16) -- exactly the MFMA D layout --
// and does the cell update in place.

// Operands: the weights in MFMA A-fragment order (weights.h) stream in, one 16-byte load per
// lane = the A operands of four MFMAs, the wave's whole slab in flight at kernel entry; the
// activations come from the [K/4][Bpad][4] copies (kernels.h) with fully coalesced 16-byte loads,
// software-pipelined three k-steps deep so that L2 latency hides behind the other tiles' MFMAs.
// Tiles whose 16 chunks have all stopped are skipped (the batch is sorted by length, so the active
// tiles form a prefix): NTA = active tiles of this pass, a compile-time count per code path.
// tagged 8-byte granules {tag = step + 1, value}: the in-launch exchanges of the batched kernels (comments at k_attention_b)
typedef unsigned long long u64;
constexpr unsigned AB_SPIN_LIMIT = 1u << 20;

// One agent-scope fetch-and-add per WAVE, its result in a scalar: lane 0 alone issues it (the exec mask is narrowed and restored inside
// the statement).  Written as one opaque statement because the compiler threads `if (lane == 0) t = atomic...; t = readfirstlane(t)`
// through a surrounding loop into one loop for lane 0 and another for the other 63 lanes, whose readfirstlane then reads a lane that
// never held the result (seen: round 6, the relay's ticket loop never ended).  The wave must be whole (all 64 lanes active) here.
__device__ __forceinline__ unsigned wave_fetch_add(unsigned *p, unsigned v) {
  unsigned r;
  unsigned long long keep;
  asm volatile(
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "global_atomic_add %0, %2, %3, %4 sc0\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_mov_b64 exec, %1"
      : "=&v"(r), "=&s"(keep)
      : "v"(0u), "v"(v), "s"(p)
      : "memory");
  return (unsigned)__builtin_amdgcn_readfirstlane((int)r);
}
__device__ __forceinline__ void granule_store(u64 *slot, unsigned tag, float v) {
  __hip_atomic_store(slot, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// N granules at base[idx + k * stride], polled together until every tag equals `want`; a timed-out slot reads as 0.0f
template <int N>
__device__ __forceinline__ void granule_gather(const u64 *base, size_t idx, size_t stride, unsigned want, float (&out)[N], int *err,
                                               unsigned limit) {
  unsigned pending = (1u << N) - 1u, spins = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) out[k] = 0.f;
  while (pending) {
    u64 g[N];
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (pending >> k & 1u) g[k] = __hip_atomic_load(base + idx + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < N; ++k)
      if ((pending >> k & 1u) && (unsigned)(g[k] >> 32) == want) {
        out[k] = __uint_as_float((unsigned)g[k]);
        pending &= ~(1u << k);
      }
    if (pending) {
      if (++spins > limit || ((spins & 127u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        atomicExch(err, 1);
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

// test hook (DecoderBufs::att_slow): this block is a straggler (~7 us) between its LSTM pass and its exchange role, every other step
// (more points -- at entry, inside the pass -- cost the undisturbed path 0.8 us per iteration: every hook is a control-flow join)
__device__ __forceinline__ void batched_straggle(const DecoderBufs &d, int blk, int step, int at) {
  if (d.att_slow && blk == d.att_slow - 1 && (step & 1) == (at & 1))
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);
}
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// after_loop: runs once this wave has issued its last operand load (and the cell-state load of the tail): the
// attention-LSTM launch puts the loads of its attention phase there, behind nothing the LSTM pass still waits for
// C0 / CN: the pass multiplies columns [C0, C0 + CN) only (wsrc points at the wave's first k-step of that range) and, with
// PART, adds the early partial of the remaining columns (DecoderBufs::att_part) in the cell-update waves.
// TAIL (decoder LSTM, two-launch form): h_dec leaves as granules d.hdg for the chunk's projection / prenet blocks of the same
// launch (dec_tail_chunk) instead of the partial-mel rows the prenet launch would sum.
// C1 / CN1 (round 6): a FIRST range [C1, C1 + CN1) multiplied ahead of [C0, C0 + CN) -- the attention launch runs its h_att(s-1)
// columns (data two launches old) before the prenet columns x(s), whose first loads take ~4 us to arrive from the other XCDs' tail
// blocks; wsrc is then the BLOCK's first k-step (k-step 0 of its rows), not the wave's.
template <int NCOLS, int KIND, int NTA, class Hook = NoHook, int C0 = 0, int CN = NCOLS, bool PART = false, bool TAIL = false, int C1 = 0, int CN1 = 0>
__device__ __forceinline__ void lstm_mfma_pass(const DecoderBufs &d, int n0, int cur, int step, int blk, const float4 *__restrict__ wsrc,
                                               const float4 bz, const float (&wa)[6], float *s_acc, unsigned long long active,
                                               unsigned long long d_probe_entry = 0, Hook after_loop = Hook()) {  // active: bit j = chunk n0 + j still runs at this step
  constexpr int NW = MFMA_WAVES, KW = CN / NW, JJ0 = KW / 16, JJ1 = CN1 / NW / 16, JJ = JJ0 + JJ1;
  static_assert(CN % (16 * NW) == 0 && C0 % 16 == 0 && CN1 % (16 * NW) == 0 && C1 % 16 == 0, "whole k-steps per wave");
  constexpr int N0 = KIND == 0 ? PRENET : ATT_RNN, N1 = EMB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, fg = lane >> 4;  // (wave index as a scalar: the segment choice in src() below is then scalar code, not exec-masked branches)
  const float4 *seg0 = reinterpret_cast<const float4 *>(KIND == 0 ? d.xf : d.att_hf[cur ^ 1]);
  const float4 *seg1 = reinterpret_cast<const float4 *>(d.ctxf);
  const float4 *seg2 = reinterpret_cast<const float4 *>(KIND == 0 ? d.att_hf[cur] : d.dec_hf[cur]);
  // k-step of the weight rows / first column that loop step jj multiplies (wave-uniform)
  auto kstep = [&](int jj) { return CN1 == 0 ? C0 / 16 + wave * JJ0 + jj : (jj < JJ1 ? C1 / 16 + wave * JJ1 + jj : C0 / 16 + wave * JJ0 + (jj - JJ1)); };
  auto w_at = [&](int jj) { return CN1 == 0 ? wsrc + (size_t)jj * 64 : wsrc + (size_t)kstep(jj) * 64; };
  auto src = [&](int jj) {  // first 16-byte vector of this lane for k-step jj (wave-uniform segment choice)
    const int col = 16 * kstep(jj);
    const float4 *sb = col < N0 ? seg0 + (size_t)(col >> 2) * d.Bpad
                                : (col < N0 + N1 ? seg1 + (size_t)((col - N0) >> 2) * d.Bpad : seg2 + (size_t)((col - N0 - N1) >> 2) * d.Bpad);
    return sb + (size_t)fg * d.Bpad + n0 + fi;
  };
#ifdef XDTTS_LSTM_PROBE
  unsigned long long tp[6];
  tp[0] = wall_clock64();
#define PROBE(i) tp[i] = wall_clock64()
#else
#define PROBE(i) do { } while (0)
#endif
  f32x4 acc[NTA];
#pragma unroll
  for (int t = 0; t < NTA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // A wave's loads retire in issue order (vmcnt), so the weight stream must NOT be issued as one block
  // ahead of the activations -- the first activation vector would then wait for the wave's last weight,
  // i.e. for the whole 42 MB HBM stream, and the MFMAs could not overlap it (measured: 5-12 us from kernel
  // entry to the first MFMA).  Both streams are issued in consumption order instead: weights three
  // k-steps ahead (HBM latency), activations two (L2).
  // Depth by the number of active tiles (measured with the k-steps fenced as below, all chunks active, us per iteration at
  // 8 / 32 / 48 / 52 / 64 chunks): weights 3 + activations 2 ahead 29.6 / 32.6 / 39.0 / 44.8 / 45.7; 3 + 1: 30.5 / 33.8 / 39.4 / 44.1 / 44.8;
  // 2 + 1: 30.3 / 33.4 / 38.5 / 43.7 / 44.6; 2 + 2: 29.6 / 32.8 / 38.7 / 44.6 / 45.5; 4 + 2: 29.9 / 32.9 / 40.1 / 44.7 / 45.5 -- with three or
  // four tiles in flight per k-step every activation vector queues behind more loads, so they run one k-step shallower.
  // (-DXDTTS_LSTM_DW / -DXDTTS_LSTM_DX override both cases.)
#ifndef XDTTS_HI_ATT
#define XDTTS_HI_ATT 21  // weights / activations ahead (two digits) from two active tiles on: attention LSTM ...
#endif
#ifndef XDTTS_LO
#define XDTTS_LO 32  // ... and with one active tile (both LSTMs, and the early blocks; two tiles with one k-step ahead: 24 / 32 chunks 28.7 / 29.2 -> 27.7 / 28.0 us)
#endif
#ifndef XDTTS_HI_DEC
#define XDTTS_HI_DEC 11  // ... and decoder LSTM (two-launch engine: 11: 32.8, 21: 33.0, 22: 33.2, 32: 33.4 us per configs[2] iteration)
#endif
#ifdef XDTTS_LSTM_DW
  constexpr int DW = XDTTS_LSTM_DW;
#else
  constexpr int DW = NTA >= 2 ? (KIND == 0 ? XDTTS_HI_ATT : XDTTS_HI_DEC) / 10 : XDTTS_LO / 10;
#endif
#ifdef XDTTS_LSTM_DX
  constexpr int DX = XDTTS_LSTM_DX;
#else
  constexpr int DX = NTA >= 2 ? (KIND == 0 ? XDTTS_HI_ATT : XDTTS_HI_DEC) % 10 : XDTTS_LO % 10;
#endif
  constexpr int RX = DX + 1;
  float4 ring[RX][NTA], wring[8];
#pragma unroll
  for (int p = 0; p < (DW > DX ? DW : DX) && p < JJ; ++p) {
    if (p < DW) wring[p] = *w_at(p);  // plain loads: with the non-temporal hint of the GEMV kernels the 52-chunk iteration took 2.3 us longer
                                      // (the 71 MB of weights fit the 256 MB Infinity Cache and are read again 50 us later)
    if (p < DX) {
      const float4 *sp = src(p);
#pragma unroll
      for (int t = 0; t < NTA; ++t) ring[p][t] = sp[16 * t];
    }
    asm volatile("" ::: "memory");
  }
  // batched cell state: [256 blocks][Bpad][4 units], one 256-byte run per tile.  Loaded here, behind the first operand
  // loads, so that it is in its register long before the tail needs it (see the pin ahead of the hook below).
  float *cst = KIND == 0 ? d.att_c : d.dec_c;
  const size_t ci = ((size_t)blk * d.Bpad + n0 + 16 * (wave < NTA ? wave : 0) + fi) * 4 + fg;
  float c_old = cst[wave < NTA && n0 + 16 * wave + fi < d.B ? ci : (size_t)blk * d.Bpad * 4];  // (unconditional: clamped to the block's first state)
  f32x4 pin = (f32x4){0.f, 0.f, 0.f, 0.f};  // early partial of this lane's (chunk, unit): written by the previous launch
  if (PART) pin = reinterpret_cast<const f32x4 *>(KIND == 0 ? d.att_part : d.dec_part)[((size_t)blk * 4 + (wave < NTA ? wave : 0)) * 64 + lane];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int jj = 0; jj < JJ; ++jj) {
    // (the activations of step jj + DX before the weights of step jj + DW: loads retire in issue order, and the other
    // way round every activation vector waits for a weight vector that is needed a whole k-step later)
    if (jj + DX < JJ) {
      const float4 *sp = src(jj + DX);
#pragma unroll
      for (int t = 0; t < NTA; ++t) ring[(jj + DX) % RX][t] = sp[16 * t];  // (non-temporal: 38.5 -> 45 us per iteration, the 32 CUs of an XCD share these lines in L2)
    }
    asm volatile("" ::: "memory");
    if (jj + DW < JJ) wring[(jj + DW) % 8] = *w_at(jj + DW);
    asm volatile("" ::: "memory");
    // (the asm fences order memory operations only: the MFMAs of the NEXT k-step, whose operands are in flight already, are free
    // to be hoisted above this k-step's loads, and then every load is waited for right behind its issue -- seen in the ISA after an
    // unrelated edit: vmcnt(3)/(1)/(0) instead of (9)..(6), the decoder-LSTM launch 1.5-2.7 us slower.  Hence the scheduling fence.)
    __builtin_amdgcn_sched_barrier(0);
    const float4 wv = wring[jj % 8];
    const float4(&xv)[NTA] = ring[jj % RX];
    // interleave the tiles so consecutive MFMAs hit different accumulators (40-cycle dependent latency)
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv[t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv[t].w, acc[t], 0, 0, 0);
  }
  PROBE(1);
  float *h_out = KIND == 0 ? d.att_h[cur ^ 1] : d.dec_h[cur ^ 1];
  float *hf_out = KIND == 0 ? d.att_hf[cur ^ 1] : d.dec_hf[cur ^ 1];
  // (Cell state and gate biases are first used behind the conditional hook below, so the compiler waits for vmcnt(0) there --
  // the whole attention prefetch of the block -- before the accumulator exchange.  Pinning both in registers ahead of the hook
  // removes that wait and lets the block publish h ~2 us earlier, but measured 0.1 ms per 647-iteration batch SLOWER: the
  // attention blocks wait for the slowest publisher of 256 either way, and their own prefetch then lands later.  Not pinned.)
  const float4 bzp = bz;
  after_loop();
  // [K-slice][tile][lane][gate]
#pragma unroll
  for (int t = 0; t < NTA; ++t) *reinterpret_cast<f32x4 *>(s_acc + ((size_t)(wave * 4 + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  PROBE(2);
  if (wave < NTA) {  // wave t finalises chunk tile t: lane = (chunk fi, unit fg), regs = gates i,f,g,o
    f32x4 g = *reinterpret_cast<const f32x4 *>(s_acc + ((size_t)wave * 64 + lane) * 4);
#pragma unroll
    for (int q = 1; q < NW; ++q) g += *reinterpret_cast<const f32x4 *>(s_acc + ((size_t)(q * 4 + wave) * 64 + lane) * 4);
    if (PART) g += pin;
    const int n = n0 + 16 * wave + fi, unit = blk * 4 + fg;
    float hn = 0.f;
    if (n < d.B) {
      // hardware exp2 / rcp forms (device_utils.h), as in the persistent engine: this tail runs on NTA of
      // the waves while the others wait
      const float ig = fast_sigmoid(g[0] + bzp.x), fgt = fast_sigmoid(g[1] + bzp.y);
      const float gg = fast_tanh(g[2] + bzp.z), og = fast_sigmoid(g[3] + bzp.w);
      const float cn = fmaf(fgt, c_old, ig * gg);
      hn = og * fast_tanh(cn);
      if ((active >> (16 * wave + fi)) & 1ull) {
        cst[ci] = cn;
        if (KIND == 0) {  // (row-major copy: the energies kernel reads it; nobody reads dec_h that way)
          if (d.hg)
            __hip_atomic_store(d.hg + (size_t)n * ATT_RNN + unit, ((unsigned long long)((unsigned)step + 1u) << 32) | (unsigned long long)__float_as_uint(hn),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            h_out[(size_t)n * ATT_RNN + unit] = hn;
        }
        hf_out[((size_t)blk * d.Bpad + n) * 4 + fg] = hn;
        if (KIND == 0 && d.hring) {  // ... and into this step's slab of the write-once ring (the same launch's extra blocks poll it)
          const unsigned hb = __float_as_uint(hn);
          __hip_atomic_store(d.hring + ((size_t)step * (ATT_RNN / 4) + blk) * d.Bpad * 4 + (size_t)n * 4 + fg, hb == 0xffffffffu ? 0x7fc00000u : hb, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
        if (TAIL && !(d.tail_fault && blk == d.tail_fault - 1)) granule_store(d.hdg + (size_t)n * DEC_RNN + unit, (unsigned)step + 1u, hn);
      }
    }
    if (KIND == 1 && !TAIL) {
      // Partial mel of this block's four hidden units, pm[m][chunk] = sum_u W_p[m][4 blk + u] h[chunk][u], on the matrix cores: the lane
      // that has just produced h of (chunk fi, unit fg) holds exactly the B fragment of a 16x16x4 MFMA (k = unit, column = chunk); the A
      // fragments -- W_p[16 rt + fi][4 blk + fg], six 16-row tiles -- were fetched at kernel entry.  D register r of the lane is mel row
      // 16 rt + 4 fg + r of chunk fi: one 16-byte store.  (It was a pass of all waves over an LDS copy of h behind a barrier: ~1 us.)
      const bool on = n < d.B && ((active >> (16 * wave + fi)) & 1ull);
      float *prow = d.pmel + ((size_t)min(n, d.B - 1) * PM_ROWS + CTX_BLOCKS + blk) * MEL_LD + 4 * fg;
#pragma unroll
      for (int rt = 0; rt < 6; ++rt) {
        const f32x4 pm = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[rt], hn, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (on && 16 * rt + 4 * fg < MEL_LD) *reinterpret_cast<f32x4 *>(prow + 16 * rt) = pm;
      }
    }
  }
  PROBE(3);
  __syncthreads();
  PROBE(4);
#ifdef XDTTS_LSTM_PROBE
  if ((blk == 3 || blk == 200) && (tid == 0 || tid == 64 * 9) && (step == 100 || step == 101))
    printf("probe kind %d blk %d wave %d step %d NTA %d: entry->loop %llu  loop %llu  store+sync %llu  cell %llu  pmel+sync %llu (x10ns)\n", KIND, blk, wave, step, NTA,
           tp[0] - d_probe_entry, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3]);
#endif
}


// ---- early partial of the attention-LSTM GEMM (DecoderBufs::att_part) ----------------------------------------------------
// Of the 1792 columns the attention LSTM of step s+1 multiplies, only the 256 prenet columns x(s+1) are new when its launch
// starts: the context and its own hidden state of step s are complete when the attention launch of step s ends, so those
// 1536 columns (86 % of the pass, 25 of its 29 MB of weights) are multiplied one launch early -- inside a decode by 256 extra
// blocks of the decoder-LSTM launch of step s (two 512-thread blocks of k_lstm_mfma fit a CU, and its own blocks leave the
// matrix cores idle at entry, while they wait for their last wave and in the cell update); at the start of a sequence that
// begins from caller-held state by the stand-alone k_att_early.  Block blk = the 16 gate rows of LSTM block blk, its 8 waves
// take 12 k-steps (of 16 columns) each, operands as in lstm_mfma_pass (weights in A-fragment order, activations from the
// [K/4][Bpad][4] copies).  The 8 partial accumulators meet in LDS in a fixed order and leave as [blk][tile][lane] float4 = the
// D layout the cell-update waves of the attention launch hold, which add them to their own 256-column product.  `hcur`: half
// of att_hf that holds h_att(s).  The attention launch shrinks by 6.2 us (its pass is 2 k-steps per wave), the decoder-LSTM
// launch grows by 5.3 (configs[2], rocprofv3).  Measured against it and rejected: the same 256 blocks inside the PRENET launch
// (its 1024-thread blocks hold 128 VGPRs: ONE block per CU, so the two roles ran one after the other, 14.4 us); ONE block per
// CU that feeds the shared operand [h_att ; ctx] to both weight slabs (k-loop of 12 + 8 k-steps per wave, activation reads
// 214 -> 139 MB per iteration: 39.9 us per iteration against 36.4, with 16 waves 42.0); see DESIGN.md, Appendix A.
#ifdef XDTTS_LSTM_PROBE
__device__ __forceinline__ unsigned hw_place() {  // (xcc << 16) | HW_ID: which CU a block landed on
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  return ((xcc & 15u) << 16) | (hw & 0xffffu);
}
#endif
// KIND 1 (two-launch form, d.dec_part): the same for the DECODER LSTM's own-state columns -- W_dec[:, 1536..2559] . h_dec(s-1), 40 % of its
// pass, complete when the decoder-LSTM launch of step s-1 ends -- by 256 extra blocks of the ATTENTION launch of step s, whose own
// blocks are a latency chain; the decoder-LSTM launch of step s then multiplies [h_att(s) ; ctx(s)] only.  hcur: half of dec_hf.
constexpr int EARLY_K0 = PRENET / 16, EARLY_KS = (ATT_COLS - PRENET) / 16;  // k-steps 16 .. 111 of the 112
constexpr int EARLY_K0_D = (ATT_RNN + EMB) / 16, EARLY_KS_D = DEC_RNN / 16;  // decoder LSTM: k-steps 96 .. 159 of the 160
#ifndef XDTTS_HRING_AUX
#define XDTTS_HRING_AUX 16  // cache policy of the first poll of a ring quad (16 = sc1; retries: sc0 sc1)
#endif
template <int NTA, int KIND = 0, bool HIN = false, bool SHORT = false>
__device__ __forceinline__ void att_early_partial(const DecoderBufs &d, int hcur, int blk, const float4 *__restrict__ Wm, float *lds,
                                                  unsigned long long t_entry = 0, int step = 0, unsigned long long active = ~0ull) {
  constexpr int NWV = MFMA_WAVES, K0 = KIND ? EARLY_K0_D : EARLY_K0, KS = KIND ? EARLY_KS_D : (SHORT ? EMB / 16 : EARLY_KS), JJ = KS / NWV;  // SHORT: the 512 context columns only (DecoderBufs::att_hfirst)
  constexpr int KSTEPS = (KIND ? DEC_COLS : ATT_COLS) / 16;
  static_assert(KS % NWV == 0, "whole k-steps per wave");
#ifdef XDTTS_LSTM_PROBE
  unsigned long long ep[4];
  ep[0] = wall_clock64();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fi = lane & 15, fg = lane >> 4;
  const float4 *wsrc = Wm + ((size_t)blk * KSTEPS + K0 + wave * JJ) * 64 + lane;
  unsigned xcc = 0;
  unsigned *rcnt = nullptr;  // per-XCD relay of the h_att ring (below): this XCD's counters of the step, [0] octets of rows drawn, [1] rows in the copy
  if constexpr (HIN) {
    if (d.hstage) {  // the block draws its octet now, the answer is needed after the first K-loop
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 7u;
      rcnt = d.hcnt + ((size_t)xcc * d.hring_steps + step) * 64;  // (256 bytes per counter pair, the XCDs' counters far apart: not one line, not one channel)
      if (wave == 0) {
        const unsigned t = wave_fetch_add(rcnt, 1u);
        if (lane == 0) reinterpret_cast<volatile unsigned *>(lds)[0] = t;
      }
    }
  }
  const float4 *seg1 = reinterpret_cast<const float4 *>(d.ctxf), *seg2 = reinterpret_cast<const float4 *>(KIND ? d.dec_hf[hcur] : d.att_hf[hcur]);
  auto src = [&](int jj) {  // (wave-uniform segment choice: scalar code)
    const int col = 16 * (wave * JJ + jj);  // column behind the prenet columns (KIND 1: of h_dec)
    const float4 *sb = KIND ? seg2 + (size_t)(col >> 2) * d.Bpad : (col < EMB ? seg1 + (size_t)(col >> 2) * d.Bpad : seg2 + (size_t)((col - EMB) >> 2) * d.Bpad);
    return sb + (size_t)fg * d.Bpad + fi;
  };
#ifndef XDTTS_EARLY_HI
#define XDTTS_EARLY_HI 11  // weights / activations ahead from two active tiles on (11: 32.7, 21: 33.0, 22: 33.6, 31: 33.3 us per configs[2] iteration)
#endif
  constexpr int DW = NTA >= 2 ? XDTTS_EARLY_HI / 10 : XDTTS_LO / 10, DX = NTA >= 2 ? XDTTS_EARLY_HI % 10 : XDTTS_LO % 10, RX = DX + 1;  // prefetch depths of lstm_mfma_pass
  f32x4 acc[NTA];
#pragma unroll
  for (int t = 0; t < NTA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ring[RX][NTA], wring[4];
#pragma unroll
  for (int p = 0; p < DW; ++p) {
    wring[p] = wsrc[(size_t)p * 64];
    if (p < DX) {
      const float4 *sp = src(p);
#pragma unroll
      for (int t = 0; t < NTA; ++t) ring[p][t] = sp[16 * t];
    }
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int jj = 0; jj < JJ; ++jj) {
    if (jj + DX < JJ) {
      const float4 *sp = src(jj + DX);
#pragma unroll
      for (int t = 0; t < NTA; ++t) ring[(jj + DX) % RX][t] = sp[16 * t];
    }
    asm volatile("" ::: "memory");
    if (jj + DW < JJ) wring[(jj + DW) % 4] = wsrc[(size_t)(jj + DW) * 64];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const float4 wv = wring[jj % 4];
    const float4(&xv)[NTA] = ring[jj % RX];
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv[t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv[t].w, acc[t], 0, 0, 0);
  }
  if constexpr (HIN) {
    // KIND 1, two-launch form with d.hring: the decoder LSTM's h_att(s) columns too -- k-steps 0 .. 63 of its weights against the vector
    // the attention-LSTM blocks of THIS launch are publishing into the step's ring slab.  Every 16-byte operand quad (k-quad 4 j + fg,
    // chunk) has one producer block; it is polled (sc1, retries sc0 sc1) until none of its words is the fill pattern, one k-step
    // ahead of the MFMAs.  Lanes of chunks that do not run this step take zeros and wait for nobody.
    constexpr int JH = (ATT_RNN / 16) / NWV;  // 8 k-steps per wave
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_ring = __builtin_amdgcn_make_buffer_rsrc((void *)(d.hring + (size_t)step * ATT_RNN * d.Bpad), 0, 0x7fffffff, 0x00020000);
    const float4 *wh = Wm + ((size_t)blk * KSTEPS + wave * JH) * 64 + lane;
    const unsigned spin_limit = d.att_spins > 0 ? (unsigned)d.att_spins : AB_SPIN_LIMIT;
    bool on[NTA];
#pragma unroll
    for (int t = 0; t < NTA; ++t) on[t] = (active >> (16 * t + fi)) & 1ull;
    // Per-XCD relay (d.hstage): 256 blocks each pulling the whole slab from the other XCDs get ~24 GB/s per CU (6 TB/s over the chip: the
    // rate of every in-launch bulk edge on this part).  Instead the blocks that find themselves on XCD x (HW_REG_XCC_ID: a fact, not an
    // assumption about placement) share the pull: every block draws an octet of the slab's 256 k-quad rows of [chunks][4] from the XCD's
    // ticket counter (at entry, one atomic per block), each of its waves polls one row of it out of the ring (one quad per lane) and
    // writes it into the XCD's own copy with plain stores (they stay in that XCD's L2); everybody polls its operands out of that copy,
    // with L1-bypassing loads that hit the shared L2, under the ring's own rule (a quad with a fill word is not there yet).  The XCD has
    // two copies, by step parity: whoever fills a row of this step's copy puts the fill pattern back into the same row of the other,
    // which the previous launch is done with and the next one will poll (the host fills both at the start of a request).  Any number
    // of blocks per XCD completes the copy: a wave whose poll stays pending looks at the ticket counter and takes an undrawn octet whole.
    const bool relay = d.hstage != nullptr;
    __amdgpu_buffer_rsrc_t rs = rs_ring, rs_stage = rs_ring, rs_other = rs_ring;
    constexpr unsigned ROWS = ATT_RNN / 4, OCTETS = ROWS / NWV;
    const bool mine = lane < 16 * NTA, wanted = mine && ((active >> lane) & 1ull);
    auto move_row = [&](unsigned row) {  // ring -> this XCD's copy of the step, one quad per lane; the same row of the other copy back to "unwritten"
      const int off = (int)((row * (unsigned)d.Bpad + (unsigned)lane) * 16u);
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (wanted) {
        unsigned spins = 0;
        v = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off, 0, 16);
        while (v.x == 0xffffffffu || v.y == 0xffffffffu || v.z == 0xffffffffu || v.w == 0xffffffffu) {
          if (++spins > spin_limit || ((spins & 127u) == 0 && __hip_atomic_load(d.att_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            atomicExch(d.att_err, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          v = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off, 0, 17);
        }
      }
      if (mine) {
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_stage, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, rs_other, off, 0, 0);
      }
    };
    auto rescue = [&]() {  // octets no block has drawn (fewer than 32 of these blocks on this XCD): whoever waits long enough takes one whole
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(rcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >= OCTETS) return;
      const unsigned t2 = wave_fetch_add(rcnt, 1u);
      if (t2 < OCTETS)
        for (unsigned r = 0; r < (unsigned)NWV; ++r) move_row(t2 * NWV + r);
    };
    if (relay) {
      const size_t copy = (size_t)ATT_RNN * d.Bpad;
      rs_stage = __builtin_amdgcn_make_buffer_rsrc((void *)(d.hstage + (2 * xcc + (step & 1)) * copy), 0, 0x7fffffff, 0x00020000);
      rs_other = __builtin_amdgcn_make_buffer_rsrc((void *)(d.hstage + (2 * xcc + ((step & 1) ^ 1)) * copy), 0, 0x7fffffff, 0x00020000);
      __syncthreads();  // the block's octet (ticket drawn at entry): one row per wave
      const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<volatile unsigned *>(lds)[0]);
      if (tk < OCTETS) move_row(tk * NWV + (unsigned)wave);
      rs = rs_stage;
    }
    auto voff = [&](int jj, int t) { return (int)((((unsigned)(4 * (wave * JH + jj) + fg) * (unsigned)d.Bpad) + 16u * t + fi) * 16u); };
    // (operand quads DH k-steps ahead: a poll of another XCD's fresh data takes ~1.3 us whatever it finds, a k-step's MFMAs 0.2 us)
    constexpr int DH = NTA <= 2 ? 6 : (NTA == 3 ? 4 : 3);
    u32x4 hb[DH][NTA];
    float4 wv2[DH];
#pragma unroll
    for (int p = 0; p < DH; ++p) {
#pragma unroll
      for (int t = 0; t < NTA; ++t) hb[p][t] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff(p, t), 0, 16);
      wv2[p] = wh[(size_t)p * 64];
    }
#pragma unroll
    for (int jj = 0; jj < JH; ++jj) {
      u32x4(&v)[NTA] = hb[jj % DH];
      unsigned pending = 0u, spins = 0;
#pragma unroll
      for (int t = 0; t < NTA; ++t) {
        pending |= (on[t] && (v[t].x == 0xffffffffu || v[t].y == 0xffffffffu || v[t].z == 0xffffffffu || v[t].w == 0xffffffffu)) ? 1u << t : 0u;
        if (!on[t]) v[t] = (u32x4){0u, 0u, 0u, 0u};
      }
      while (pending) {
        if (++spins > spin_limit || ((spins & 127u) == 0 && __hip_atomic_load(d.att_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          atomicExch(d.att_err, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        if (relay && (spins & 63u) == 32u) rescue();
#pragma unroll
        for (int t = 0; t < NTA; ++t)
          if ((pending >> t) & 1u) v[t] = relay ? __builtin_amdgcn_raw_buffer_load_b128(rs, voff(jj, t), 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(rs, voff(jj, t), 0, 17);  // (the XCD's copy lives in this L2: sc1 is enough)
#pragma unroll
        for (int t = 0; t < NTA; ++t)
          if (((pending >> t) & 1u) && v[t].x != 0xffffffffu && v[t].y != 0xffffffffu && v[t].z != 0xffffffffu && v[t].w != 0xffffffffu) pending &= ~(1u << t);
        asm volatile("" ::: "memory");
      }
      const float4 wv = wv2[jj % DH];
#pragma unroll
      for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, __uint_as_float(v[t].x), acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, __uint_as_float(v[t].y), acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, __uint_as_float(v[t].z), acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NTA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, __uint_as_float(v[t].w), acc[t], 0, 0, 0);
      if (jj + DH < JH) {  // the slot just consumed: the operands DH k-steps on
#pragma unroll
        for (int t = 0; t < NTA; ++t) v[t] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff(jj + DH, t), 0, 16);
        wv2[jj % DH] = wh[(size_t)(jj + DH) * 64];
      }
    }
  }
#ifdef XDTTS_LSTM_PROBE
  ep[1] = wall_clock64();
#endif
  f32x4 *red = reinterpret_cast<f32x4 *>(lds);  // [wave][tile][lane]
#pragma unroll
  for (int t = 0; t < NTA; ++t) red[(wave * 4 + t) * 64 + lane] = acc[t];
  __syncthreads();
  if (wave < NTA) {
    f32x4 g = red[wave * 64 + lane];
#pragma unroll
    for (int q = 1; q < NWV; ++q) g += red[(q * 4 + wave) * 64 + lane];
    reinterpret_cast<f32x4 *>(KIND ? d.dec_part : d.att_part)[((size_t)blk * 4 + wave) * 64 + lane] = g;
  }
#ifdef XDTTS_LSTM_PROBE
  ep[2] = wall_clock64();
  if ((blk == 3 || blk == 100 || blk == 200 || blk == 255) && (tid == 0 || tid == 64 * 5) && (step == 100 || step == 101))
    printf("probe early blk %d wave %d step %d NTA %d place %05x: entry %llu  entry->loop %llu  loop %llu  reduce %llu  (x10ns)\n", blk, wave, step, NTA, hw_place(),
           t_entry % 100000ull, ep[0] - t_entry, ep[1] - ep[0], ep[2] - ep[1]);
#endif
}
// the role as a whole: tiles by the chunks still active at step `next`
template <int KIND = 0>
__device__ __forceinline__ void att_early_role(const DecoderBufs &d, int next, int hcur, int blk, const float4 *__restrict__ Wm, float *lds, unsigned long long t_entry,
                                               bool hin = false) {
  const int lane = threadIdx.x & 63;
  const bool a = lane < d.B && next < d.nframes[min(lane, d.B - 1)];  // (a chunk the next prenet launch stops still counts: its tile's partial is not read then)
  const unsigned long long m = __ballot(a);
  const int nta = m ? (63 - __clzll((long long)m)) / 16 + 1 : 0;
  if (KIND == 1 && hin) {  // (the attention launch's extra blocks, with the ring: + the h_att(s) columns, polled inside the launch)
    switch (nta) {
      case 1: att_early_partial<1, KIND, KIND == 1>(d, hcur, blk, Wm, lds, t_entry, next, m); break;
      case 2: att_early_partial<2, KIND, KIND == 1>(d, hcur, blk, Wm, lds, t_entry, next, m); break;
      case 3: att_early_partial<3, KIND, KIND == 1>(d, hcur, blk, Wm, lds, t_entry, next, m); break;
      case 4: att_early_partial<4, KIND, KIND == 1>(d, hcur, blk, Wm, lds, t_entry, next, m); break;
      default: break;
    }
    return;
  }
  if (KIND == 0 && d.att_hfirst) {  // (the attention launch multiplies its h_att columns itself: the context columns only here)
    switch (nta) {
      case 1: att_early_partial<1, KIND, false, KIND == 0>(d, hcur, blk, Wm, lds, t_entry, next); break;
      case 2: att_early_partial<2, KIND, false, KIND == 0>(d, hcur, blk, Wm, lds, t_entry, next); break;
      case 3: att_early_partial<3, KIND, false, KIND == 0>(d, hcur, blk, Wm, lds, t_entry, next); break;
      case 4: att_early_partial<4, KIND, false, KIND == 0>(d, hcur, blk, Wm, lds, t_entry, next); break;
      default: break;
    }
    return;
  }
  switch (nta) {
    case 1: att_early_partial<1, KIND>(d, hcur, blk, Wm, lds, t_entry, next); break;
    case 2: att_early_partial<2, KIND>(d, hcur, blk, Wm, lds, t_entry, next); break;
    case 3: att_early_partial<3, KIND>(d, hcur, blk, Wm, lds, t_entry, next); break;
    case 4: att_early_partial<4, KIND>(d, hcur, blk, Wm, lds, t_entry, next); break;
    default: break;
  }
}
// stand-alone form (parity hooks: a sequence that starts from caller-held state has no preceding decoder-LSTM launch)
__global__ __launch_bounds__(64 * MFMA_WAVES) void k_att_early(DecoderBufs d, int i, const float4 *__restrict__ att_wm, const float4 *__restrict__ dec_wm) {
  __shared__ __attribute__((aligned(16))) float s_acc[MFMA_WAVES * 4 * 64 * 4];
  if ((int)blockIdx.x < NBLK) att_early_role<0>(d, d.ctl[0] + i, i & 1, blockIdx.x, att_wm, s_acc, 0);
  else att_early_role<1>(d, d.ctl[0] + i, i & 1, (int)blockIdx.x - NBLK, dec_wm, s_acc, 0);  // (grid 512 with d.dec_part)
}



struct TailWeights {  // what the two-launch form's extra roles of the decoder-LSTM launch read
  const float4 *proj_w4;  // [81][1536] rows of [W_p ; w_gate]
  const float *proj_b, *W0T, *W1T, *loc_convT, *loc_denseT;
};
// ---- two-launch form: projection, stop rule and prenet as the tail of the decoder-LSTM launch ---------------------------------
// What made the prenet its own launch is the projection: mel(s) needs ALL of h_dec(s).  As with h_att in the attention launch,
// that vector is small enough to cross inside the launch: the 256 decoder-LSTM blocks publish their four units of every chunk as
// granules (d.hdg), and blocks 4 b .. 4 b + 3 then turn into chunk b's tail blocks -- gather h_dec(b), 21 / 20 rows of
// [W_p ; w_gate] each (row m = part + 4 r; the context columns against d.ctx of this step's attention launch), exchange the 81
// values (d.melg), frame / gate store and stop rule (mod.rs:319-324, block 0), prenet layer 1 (all 256 outputs, every block) and
// 64 columns of layer 2 -> x(s + 1) and its B-operand copy.  lds: 1024 + 512 + 96 + 8 x 256 + 256 + 32 x 64 floats.
constexpr int TAIL_PARTS = 4, TAIL_NT = 64 * MFMA_WAVES;
constexpr int TAIL_LDS_FLOATS = DEC_RNN + EMB + 96 + (TAIL_NT / 64) * PRENET + PRENET + (TAIL_NT / (PRENET / TAIL_PARTS / 4)) * (PRENET / TAIL_PARTS);
__device__ __forceinline__ void dec_tail_chunk(const DecoderBufs &d, int step, int b, int part, float *lds, const float4 *__restrict__ proj_w4,
                                               const float *__restrict__ proj_b, const float *__restrict__ W0T, const float *__restrict__ W1T) {
  constexpr int NT = TAIL_NT, NWV = NT / 64, COLS2 = PRENET / TAIL_PARTS;  // 64 layer-2 columns per block
  constexpr int KG1 = NT / 64, IN1 = N_MEL / KG1;                          // layer 1: 8 input groups of 10
  constexpr int KG2 = NT / (COLS2 / 4), IN2 = PRENET / KG2;                // layer 2: 32 input groups of 8
  static_assert(N_MEL % KG1 == 0 && PRENET % KG2 == 0 && (DEC_RNN + EMB) % 256 == 0, "tail mappings");
  float *s_h = lds, *s_c = s_h + DEC_RNN, *s_mel = s_c + EMB, *s_p1 = s_mel + 96, *s_x1 = s_p1 + KG1 * PRENET, *s_p2 = s_x1 + PRENET;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned want = (unsigned)step + 1u, spin_limit = d.att_spins > 0 ? (unsigned)d.att_spins : AB_SPIN_LIMIT;
#ifdef XDTTS_LSTM_PROBE
  unsigned long long tq[8];
  tq[0] = wall_clock64();
#define TPROBE(i) tq[i] = wall_clock64()
#else
#define TPROBE(i) do { } while (0)
#endif
  // the wave's three projection rows (18 KB) do not depend on h_dec: in flight while it is gathered (issued behind the gather,
  // row by row, the phase measured 2.4 - 4 us: three exposed round trips through an L1 the co-resident early block is filling)
  constexpr int C4 = (DEC_RNN + EMB) / 4 / 64;  // 6 vectors per lane and row
  float4 wv[3][C4];
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    const int m = min(part + TAIL_PARTS * (wave + NWV * rr), N_MEL);  // (clamped: unconditional loads)
    const float4 *wr = proj_w4 + (size_t)m * ((DEC_RNN + EMB) / 4);
#pragma unroll
    for (int k = 0; k < C4; ++k) wv[rr][k] = wr[lane + 64 * k];
  }
  asm volatile("" ::: "memory");
  // ---- h_dec(b) from the 256 LSTM blocks of this launch, ctx(b) from the attention launch ----
  {
    float g[DEC_RNN / NT];
    granule_gather<DEC_RNN / NT>(d.hdg, (size_t)b * DEC_RNN + tid, NT, want, g, d.att_err, spin_limit);
#pragma unroll
    for (int k = 0; k < DEC_RNN / NT; ++k) s_h[tid + NT * k] = g[k];
    s_c[tid] = d.ctx[b * EMB + tid];
  }
  __syncthreads();
  TPROBE(1);
  // ---- projection rows m = part + 4 r, r = wave, wave + 8, wave + 16: lane takes float4 columns lane + 64 k of the 384 ----
  {
    float4 hv[C4];
#pragma unroll
    for (int k = 0; k < C4; ++k) hv[k] = *reinterpret_cast<const float4 *>(s_h + 4 * (lane + 64 * k));  // (s_c follows s_h)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int m = part + TAIL_PARTS * (wave + NWV * rr);
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < C4; ++k) a = dot4(wv[rr][k], hv[k], a);
      a = wave_sum(a);
      if (lane == 0 && m <= N_MEL) granule_store(d.melg + (size_t)b * 96 + m, want, a + proj_b[m]);
    }
  }
  TPROBE(2);
  // the prenet weights of this thread, in flight while the mel crosses
  const int o4 = tid & 63, kg1 = tid >> 6;
  const int c4 = tid % (COLS2 / 4), kg2 = tid / (COLS2 / 4);
  float4 w0[IN1], w1[IN2];
#pragma unroll
  for (int k = 0; k < IN1; ++k) w0[k] = reinterpret_cast<const float4 *>(W0T)[(size_t)(kg1 * IN1 + k) * (PRENET / 4) + o4];
#pragma unroll
  for (int k = 0; k < IN2; ++k) w1[k] = reinterpret_cast<const float4 *>(W1T)[((size_t)(kg2 * IN2 + k) * PRENET + COLS2 * part) / 4 + c4];
  // the Bernoulli(0.5) masks of step s + 1 do not depend on the data: hashed while the mel crosses
  const int chunk = d.item_perm ? d.item_perm[b] : b;
  const uint32_t item = d.item_base + (uint32_t)chunk;
  bool drop1 = false, drop2 = false;
  if (d.dropout_mode) {
    if (tid < PRENET) drop1 = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, chunk, step + 1, 0, tid);
    if (tid < COLS2) drop2 = prenet_dropped(d.dropout_mode, d.dropout_seed, item, d.drop_masks, d.drop_steps, chunk, step + 1, 1, COLS2 * part + tid);
  }
  if (tid <= N_MEL) {
    float g[1];
    granule_gather<1>(d.melg, (size_t)b * 96 + tid, 0, want, g, d.att_err, spin_limit);
    s_mel[tid] = g[0];
  }
  __syncthreads();
  TPROBE(3);
  const float gate = s_mel[N_MEL];
  const bool fired = d.use_gate && gate_fires(gate, d.gate_lo, d.gate_hi, d.gate_threshold);
  const int nf = d.nframes[b];
  if (part == 0) {
    if (tid < N_MEL) d.frames[((size_t)b * d.max_steps + step) * N_MEL + tid] = s_mel[tid];
    if (tid == 0) {
      d.gates[(size_t)b * d.max_steps + step] = gate;
      if (fired) d.nframes[b] = step + 1;  // frame `step` is the last one (mod.rs:319-324: the tripping frame is kept)
    }
  }
  if (fired || step + 1 >= nf) return;  // (block-uniform) the chunk stops here: no x(s + 1)
  // ---- prenet layer 1 (every block, all 256 outputs) ----
  {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < IN1; ++k) {
      const float m = s_mel[kg1 * IN1 + k];
      acc.x = fmaf(w0[k].x, m, acc.x);
      acc.y = fmaf(w0[k].y, m, acc.y);
      acc.z = fmaf(w0[k].z, m, acc.z);
      acc.w = fmaf(w0[k].w, m, acc.w);
    }
    *reinterpret_cast<float4 *>(s_p1 + kg1 * PRENET + 4 * o4) = acc;
  }
  __syncthreads();
  if (tid < PRENET) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < KG1; ++k) v += s_p1[k * PRENET + tid];
    v = fmaxf(v, 0.f);
    if (d.dropout_mode) v = drop1 ? 0.f : 2.f * v;
    s_x1[tid] = v;
  }
  __syncthreads();
  TPROBE(4);
  // ---- prenet layer 2, this block's 64 columns ----
  {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < IN2; ++k) {
      const float m = s_x1[kg2 * IN2 + k];
      acc.x = fmaf(w1[k].x, m, acc.x);
      acc.y = fmaf(w1[k].y, m, acc.y);
      acc.z = fmaf(w1[k].z, m, acc.z);
      acc.w = fmaf(w1[k].w, m, acc.w);
    }
    *reinterpret_cast<float4 *>(s_p2 + kg2 * COLS2 + 4 * c4) = acc;
  }
  __syncthreads();
  if (tid < COLS2) {
    float o = 0.f;
#pragma unroll
    for (int k = 0; k < KG2; ++k) o += s_p2[k * COLS2 + tid];
    o = fmaxf(o, 0.f);
    const int j = COLS2 * part + tid;
    if (d.dropout_mode) o = drop2 ? 0.f : 2.f * o;
    d.x[b * PRENET + j] = o;
    d.xf[((size_t)(j >> 2) * d.Bpad + b) * 4 + (j & 3)] = o;
  }
#ifdef XDTTS_LSTM_PROBE
  TPROBE(5);
  if (tid == 0 && step == 100 && (b == 0 || b == 17 || b == 40) && part == 0)
    printf("probe tail chunk %d step %d: entry %llu  h gathered %llu  rows + publish %llu  mel gathered %llu  layer 1 %llu  layer 2 + store %llu (x10ns)\n", b, step,
           tq[0] % 100000ull, tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4]);
#endif
}

template <int NCOLS, int KIND>
__global__ __launch_bounds__(64 * MFMA_WAVES) void k_lstm_mfma(DecoderBufs d, int i, int cur, const float4 *__restrict__ Wm,
                                                               const float *__restrict__ bias, const float4 *__restrict__ Wepi,
                                                               const float4 *__restrict__ att_wm, TailWeights tw) {
  constexpr int NW = MFMA_WAVES, KW = NCOLS / NW, JJ = KW / 16;
#ifdef XDTTS_LSTM_PROBE
  const unsigned long long t_entry = wall_clock64();
#else
  const unsigned long long t_entry = 0;
#endif
  __shared__ __attribute__((aligned(16))) float s_acc[NW * 4 * 64 * 4];  // [K-slice][tile][lane][gate]
  // Blocks 256..511 of a decoder-LSTM launch with the early role: the NEXT step's attention-LSTM partial (att_early_partial).
  // Two 512-thread blocks of this kernel fit a CU; which role's blocks are dispatched first, and s_setprio for either role,
  // measured the same to 0.1 us per iteration.
  if (KIND == 1 && (int)blockIdx.x >= NBLK) {
    const int e = (int)blockIdx.x - NBLK;
    att_early_role(d, d.ctl[0] + i + 1, cur ^ 1, e, att_wm, s_acc, t_entry);  // (this step's attention launch wrote h_att into half cur ^ 1)
    // two-launch form: the first 2 B early blocks also compute the next step's location features (the prenet launch's location role)
    // two-launch form: the next step's location features (the prenet launch's location role, 2 units per chunk): one unit each for
    // the decoder-LSTM blocks that are no chunk's tail (below), the rest behind the partial of the LAST early blocks (the first
    // ones share their CUs with tail blocks: 0.3 us per iteration slower; three units per free block: 0.6 us slower at 52 chunks)
    if (d.hdg) {
      const int nfree = NBLK - TAIL_PARTS * d.B, ne = max(0, 2 * d.B - nfree);
      if (e >= NBLK - ne) location_chunk_mfma<MFMA_WAVES>(d, i + 1, nfree + e - (NBLK - ne), tw.loc_convT, tw.loc_denseT);
    }
#ifdef XDTTS_LSTM_PROBE
    if (threadIdx.x == 0 && d.ctl[0] + i == 100 && (e == 3 || e == 100 || e == 200))
      printf("probe early+loc blk %d: entry %llu  end %llu (x10ns)\n", e, t_entry % 100000ull, wall_clock64() % 100000ull);
#endif
    return;
  }
  const int blk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fg = lane >> 4;
  const float4 *wsrc = Wm + ((size_t)(blk * NW + wave) * JJ) * 64 + lane;
  const int n0 = 64 * blockIdx.y;  // 64 chunks (four MFMA tiles) per block row; B > 64 adds rows that stream the weights again
  const int step = d.ctl[0] + i;
  // active tiles (wave-uniform, identical in every wave): lane -> chunk n0 + lane
  const bool a = n0 + lane < d.B && step < d.nframes[min(n0 + lane, d.B - 1)];
  const unsigned long long m = __ballot(a);
  const int nta = m ? (63 - __clzll((long long)m)) / 16 + 1 : 0;
  const float4 bz = *reinterpret_cast<const float4 *>(bias + (blk * 4 + fg) * 4);  // unit fg's i,f,g,o biases
  float wa[6];  // KIND 1: A fragments of the partial-mel product (lstm_mfma_pass), rows past the 81 are zero
#pragma unroll
  for (int rt = 0; rt < 6; ++rt) {
    const int mrow = 16 * rt + (lane & 15);
    wa[rt] = (KIND == 1 && wave < 4 && mrow < MEL_LD) ? reinterpret_cast<const float *>(Wepi)[((size_t)blk * MEL_LD + mrow) * 4 + fg] : 0.f;
  }
  if (KIND == 1 && d.hdg) {  // two-launch form: h_dec as granules, then the chunk's projection / prenet tail
    if (d.dec_part && d.hring) {  // ... and so were the h_att(s) columns (polled from the ring inside that launch): the 512 context columns only
      constexpr int C0S = KIND == 1 ? ATT_RNN : 0, CNS = KIND == 1 ? EMB : NCOLS;
      const float4 *ws = Wm + ((size_t)blk * (NCOLS / 16) + C0S / 16 + wave * (CNS / NW / 16)) * 64 + lane;
      switch (nta) {
        case 1: lstm_mfma_pass<NCOLS, KIND, 1, NoHook, C0S, CNS, KIND == 1, KIND == 1>(d, n0, cur, step, blk, ws, bz, wa, s_acc, m, t_entry); break;
        case 2: lstm_mfma_pass<NCOLS, KIND, 2, NoHook, C0S, CNS, KIND == 1, KIND == 1>(d, n0, cur, step, blk, ws, bz, wa, s_acc, m, t_entry); break;
        case 3: lstm_mfma_pass<NCOLS, KIND, 3, NoHook, C0S, CNS, KIND == 1, KIND == 1>(d, n0, cur, step, blk, ws, bz, wa, s_acc, m, t_entry); break;
        case 4: lstm_mfma_pass<NCOLS, KIND, 4, NoHook, C0S, CNS, KIND == 1, KIND == 1>(d, n0, cur, step, blk, ws, bz, wa, s_acc, m, t_entry); break;
        default: return;
      }
    } else if (d.dec_part) {  // ... and the h_dec(s-1) columns were multiplied by the attention launch's extra blocks: [h_att ; ctx] only
      constexpr int CNS = KIND == 1 ? ATT_RNN + EMB : NCOLS;
      const float4 *ws = Wm + ((size_t)blk * (NCOLS / 16) + wave * (CNS / NW / 16)) * 64 + lane;
      switch (nta) {
        case 1: lstm_mfma_pass<NCOLS, KIND, 1, NoHook, 0, CNS, KIND == 1, KIND == 1>(d, n0, cur, step, blk, ws, bz, wa, s_acc, m, t_entry); break;
        case 2: lstm_mfma_pass<NCOLS, KIND, 2, NoHook, 0, CNS, KIND == 1, KIND == 1>(d, n0, cur, step, blk, ws, bz, wa, s_acc, m, t_entry); break;
        case 3: lstm_mfma_pass<NCOLS, KIND, 3, NoHook, 0, CNS, KIND == 1, KIND == 1>(d, n0, cur, step, blk, ws, bz, wa, s_acc, m, t_entry); break;
        case 4: lstm_mfma_pass<NCOLS, KIND, 4, NoHook, 0, CNS, KIND == 1, KIND == 1>(d, n0, cur, step, blk, ws, bz, wa, s_acc, m, t_entry); break;
        default: return;
      }
    } else {
      switch (nta) {
        case 1: lstm_mfma_pass<NCOLS, KIND, 1, NoHook, 0, NCOLS, false, KIND == 1>(d, n0, cur, step, blk, wsrc, bz, wa, s_acc, m, t_entry); break;
        case 2: lstm_mfma_pass<NCOLS, KIND, 2, NoHook, 0, NCOLS, false, KIND == 1>(d, n0, cur, step, blk, wsrc, bz, wa, s_acc, m, t_entry); break;
        case 3: lstm_mfma_pass<NCOLS, KIND, 3, NoHook, 0, NCOLS, false, KIND == 1>(d, n0, cur, step, blk, wsrc, bz, wa, s_acc, m, t_entry); break;
        case 4: lstm_mfma_pass<NCOLS, KIND, 4, NoHook, 0, NCOLS, false, KIND == 1>(d, n0, cur, step, blk, wsrc, bz, wa, s_acc, m, t_entry); break;
        default: return;
      }
    }
    const int b = blk >> 2;
    static_assert(TAIL_LDS_FLOATS <= NW * 4 * 64 * 4, "the tail reuses the accumulator exchange area");
#ifdef XDTTS_LSTM_PROBE
    if (threadIdx.x == 0 && step == 100 && (blk == 0 || blk == 68 || blk == 160 || blk == 250))
      printf("probe D blk %d: entry %llu  pass done %llu (x10ns)\n", blk, t_entry % 100000ull, wall_clock64() % 100000ull);
#endif
    batched_straggle(d, blk, step, 2);  // (between its publish of h_dec and its share of the chunk's mel rows)
    if (b < d.B && step < d.nframes[b]) dec_tail_chunk(d, step, b, blk & 3, s_acc, tw.proj_w4, tw.proj_b, tw.W0T, tw.W1T);
    if (blk >= TAIL_PARTS * d.B && blk - TAIL_PARTS * d.B < 2 * d.B)  // free after the pass: one location unit, done well inside the tails' time
      location_chunk_mfma<MFMA_WAVES>(d, i + 1, blk - TAIL_PARTS * d.B, tw.loc_convT, tw.loc_denseT);
    return;
  }
  switch (nta) {
    case 1: lstm_mfma_pass<NCOLS, KIND, 1>(d, n0, cur, step, blk, wsrc, bz, wa, s_acc, m, t_entry); break;
    case 2: lstm_mfma_pass<NCOLS, KIND, 2>(d, n0, cur, step, blk, wsrc, bz, wa, s_acc, m, t_entry); break;
    case 3: lstm_mfma_pass<NCOLS, KIND, 3>(d, n0, cur, step, blk, wsrc, bz, wa, s_acc, m, t_entry); break;
    case 4: lstm_mfma_pass<NCOLS, KIND, 4>(d, n0, cur, step, blk, wsrc, bz, wa, s_acc, m, t_entry); break;
    default: break;
  }
}

// D3a: processed query q = W_q att_h_new (128 x 1024, no bias) and, fused, this block's share of
// the energies.  Block `blk` owns attention dims a in [4 blk, 4 blk + 4): one wave per query row,
// then the 256 threads each take a time step and emit
//   e_part[blk][t] = sum_{a in block} v_a tanh(q_a + loc[t][a] + processed_memory[t][a]).
// This spreads the 12.8k tanh of a step over 32 CUs instead of one.
constexpr int QE_GROUP = 4;
__global__ __launch_bounds__(256) void k_qenergy(DecoderBufs d, int i, int cur, const float4 *__restrict__ Wq,
                                                 const float *__restrict__ v_w) {
#ifdef XDTTS_LSTM_PROBE
  const unsigned long long t_in = wall_clock64();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
  const int row = blk * 4 + wave;
  float4 w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = Wq[(size_t)row * (ATT_RNN / 4) + lane + 64 * k];
  const float4 v4 = *reinterpret_cast<const float4 *>(v_w + blk * 4);
  const int step = d.ctl[0] + i;
  __shared__ __attribute__((aligned(16))) float s_q[4];
  // small batches: one launch row loops over the chunks; large batches: QE_GROUP chunks per block row
  // (the query rows stay in registers across them)
  const int b_lo = gridDim.y > 1 ? blockIdx.y * QE_GROUP : 0, b_hi = gridDim.y > 1 ? min(d.B, b_lo + QE_GROUP) : d.B;
  for (int b = b_lo; b < b_hi; ++b) {
    if (b > b_lo && step >= d.nframes[b]) continue;
    const float *h = d.att_h[cur ^ 1] + b * ATT_RNN;
    float4 hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hv[k] = *reinterpret_cast<const float4 *>(h + 256 * k + 4 * lane);
    // first time step of this thread: loads do not depend on q, put them in flight now
    // batched mode: loc and processed_memory as [B][32][T][4], this block's 4 dims of consecutive steps contiguous
    const bool tl = gridDim.y > 1;
    const float *pmem = tl ? d.pmem_t : d.pmem;
    auto at = [&](int t) { return tl ? (((size_t)b * (ATT_DIM / 4) + blk) * d.T + t) * 4 : ((size_t)b * d.T + t) * ATT_DIM + blk * 4; };
    const size_t o0 = at(tid < d.T ? tid : 0);
    float4 l4 = *reinterpret_cast<const float4 *>(d.loc + o0);
    float4 p4 = *reinterpret_cast<const float4 *>(pmem + o0);
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) a = dot4(w[k], hv[k], a);
    a = wave_sum(a);
    if (lane == 0) s_q[wave] = a;
    __syncthreads();
    const float4 q4 = *reinterpret_cast<const float4 *>(s_q);
    const bool act = step < d.nframes[b];
    for (int t = tid; t < d.T; t += 256) {
      if (t != tid) {
        const size_t o = at(t);
        l4 = *reinterpret_cast<const float4 *>(d.loc + o);
        p4 = *reinterpret_cast<const float4 *>(pmem + o);
      }
      float e;
      if (gridDim.y > 1) {  // batched mode: hardware exp2 / rcp tanh, as the persistent engine uses
        e = v4.x * fast_tanh(q4.x + l4.x + p4.x);
        e = fmaf(v4.y, fast_tanh(q4.y + l4.y + p4.y), e);
        e = fmaf(v4.z, fast_tanh(q4.z + l4.z + p4.z), e);
        e = fmaf(v4.w, fast_tanh(q4.w + l4.w + p4.w), e);
      } else {
        e = v4.x * tanhf(q4.x + l4.x + p4.x);
        e = fmaf(v4.y, tanhf(q4.y + l4.y + p4.y), e);
        e = fmaf(v4.z, tanhf(q4.z + l4.z + p4.z), e);
        e = fmaf(v4.w, tanhf(q4.w + l4.w + p4.w), e);
      }
      if (act) d.e_part[((size_t)b * (ATT_DIM / 4) + blk) * d.T + t] = e;
    }
    __syncthreads();
  }
#ifdef XDTTS_LSTM_PROBE
  if (tid == 0 && step == 100 && (blockIdx.y % 17 == 0) && (blk == 0 || blk == 31))
    printf("probe qenergy blk (%d,%d) in %llu out %llu\n", blk, (int)blockIdx.y, t_in % 1000000ull, wall_clock64() % 1000000ull);
#endif
}

// D3b: e_t = sum of the 32 partial energies, -inf where t >= n_valid (mask, mod.rs:219-220);
// w = softmax_t(e); w_cum += w; ctx = sum_t w_t memory_t.  CTX_BLOCKS blocks per chunk: each
// recomputes the (tiny) softmax and owns 512/CTX_BLOCKS context columns, so the T x 2 KB read of
// the encoder memory is spread over several CUs; it is issued as 16-byte lane-consecutive loads,
// prefetched at kernel entry (addresses do not depend on the softmax).  Block 0 also stores the
// new attention weights.
__global__ __launch_bounds__(256) void k_softmax_ctx(DecoderBufs d, int i, const float *__restrict__ proj_wc) {
#ifdef XDTTS_LSTM_PROBE
  const unsigned long long t_in = wall_clock64();
  unsigned long long t_sm = 0, t_ctx = 0;
#endif
  const int b = blockIdx.x / CTX_BLOCKS, cblk = blockIdx.x % CTX_BLOCKS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = d.T;
  constexpr int C4 = CTX_COLS / 4, TG = 256 / C4;  // 16 float4 columns x 16 time groups
  constexpr int CTX_PF = 7;
  __shared__ __attribute__((aligned(16))) float s_e[T_MAX], s_part[TG][CTX_COLS], s_ctx[CTX_COLS];
  const int c4 = tid % C4, tg = tid / C4;
  const float4 *mem = reinterpret_cast<const float4 *>(d.memory + (size_t)b * T * EMB) + cblk * C4;
  // Batched mode: the cumulative weights ping-pong between two buffers by step parity because the other blocks of
  // the chunk still read the old ones while block 0 writes the new (the next step's location features are
  // computed from them by the location blocks of the next prenet launch, location_blocks above).
  const bool batched = d.xf != nullptr;
  const float *awc_in = batched && (i & 1) ? d.awc2 : d.awc;
  float *awc_out = batched ? ((i & 1) ? d.awc : d.awc2) : d.awc;
  const int nv = d.n_valid[b];
  // Issue order = retire order (vmcnt): the partial energies the softmax waits for go out FIRST; the
  // context's slice of the encoder memory and the projection / location weights follow and are consumed
  // after the softmax (batched mode: 416 blocks fetching 10.6 MB of memory made the softmax wait ~3 us).
  float ev0[ATT_DIM / 4];
  {
    const float *ep = d.e_part + (size_t)b * (ATT_DIM / 4) * T + (tid < T ? tid : 0);
#pragma unroll
    for (int k = 0; k < ATT_DIM / 4; ++k) ev0[k] = ep[(size_t)k * T];
  }
  // (the previous cumulative weights too: loaded where they are used, after the softmax's barriers, their
  // latency is exposed)
  const float awc_pre = tid < T ? awc_in[b * T + tid] : 0.f;
  const int step = d.ctl[0] + i;
  const bool act = step < d.nframes[b];
  asm volatile("" ::: "memory");
  // projection weights of this block's 64 context columns: thread (m, half) holds 32 of row m
  const int pm_m = tid >> 1, pm_half = tid & 1;
  float4 wc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    wc[k] = pm_m <= N_MEL ? reinterpret_cast<const float4 *>(proj_wc + ((size_t)cblk * (N_MEL + 1) + pm_m) * CTX_COLS + 32 * pm_half)[k]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pf[CTX_PF];
#pragma unroll
  for (int u = 0; u < CTX_PF; ++u) {
    const int t = tg + TG * u;
    pf[u] = t < T ? mem[(size_t)t * (EMB / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  asm volatile("" ::: "memory");
  if (tid < T) {
    float e = 0.f;
#pragma unroll
    for (int k = 0; k < ATT_DIM / 4; ++k) e += ev0[k];
    s_e[tid] = tid >= nv ? -INFINITY : e;
  }
  for (int t = tid + 256; t < T; t += 256) {  // encoder memories longer than 256 steps
    const float *ep = d.e_part + (size_t)b * (ATT_DIM / 4) * T + t;
    float e = 0.f;
#pragma unroll
    for (int k = 0; k < ATT_DIM / 4; ++k) e += ep[(size_t)k * T];
    s_e[t] = t >= nv ? -INFINITY : e;
  }
  __syncthreads();
  if (wave == 0) {
    float m = -INFINITY;
    for (int t = lane; t < T; t += 64) m = fmaxf(m, s_e[t]);
    m = wave_max(m);
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
      const float ex = batched ? fast_exp(s_e[t] - m) : expf(s_e[t] - m);
      s_e[t] = ex;
      sum += ex;
    }
    sum = wave_sum(sum);
    for (int t = lane; t < T; t += 64) s_e[t] = s_e[t] / sum;
  }
  __syncthreads();
  for (int t = tid; t < T; t += 256) {
    const float wv = s_e[t], cum = (t == tid ? awc_pre : awc_in[b * T + t]) + wv;
    if (act && cblk == 0) {
      d.aw[b * T + t] = wv;
      awc_out[b * T + t] = cum;
    }
  }
#ifdef XDTTS_LSTM_PROBE
  t_sm = wall_clock64();
#endif
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (int t0 = tg; t0 < T; t0 += TG * CTX_PF) {
#pragma unroll
    for (int u = 0; u < CTX_PF; ++u) {
      const int t = t0 + TG * u;
      const float wv = t < T ? s_e[t] : 0.f;
      const float4 mv = k == 0 ? pf[u] : (t < T ? mem[(size_t)t * (EMB / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f));
      acc.x = fmaf(wv, mv.x, acc.x);
      acc.y = fmaf(wv, mv.y, acc.y);
      acc.z = fmaf(wv, mv.z, acc.z);
      acc.w = fmaf(wv, mv.w, acc.w);
    }
    ++k;
  }
  *reinterpret_cast<float4 *>(&s_part[tg][4 * c4]) = acc;
  __syncthreads();
  if (tid < CTX_COLS) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < TG; ++g) v += s_part[g][tid];
    s_ctx[tid] = v;
    if (act) {
      d.ctx[b * EMB + cblk * CTX_COLS + tid] = v;
      if (d.ctxf) {
        const int j = cblk * CTX_COLS + tid;
        d.ctxf[((size_t)(j >> 2) * d.Bpad + b) * 4 + (j & 3)] = v;
      }
    }
  }
  __syncthreads();
  // partial mel of these context columns: pmel[cblk][m] = W_p[m][1024 + cols] . ctx[cols]
  {
    float pv = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) pv = dot4(wc[k], *reinterpret_cast<const float4 *>(&s_ctx[32 * pm_half + 4 * k]), pv);
    pv += dpp_move<0xB1, 0xf>(0.f, pv);  // lanes 2j, 2j+1 hold the two halves of row m
    if (act && pm_half == 0 && pm_m <= N_MEL) d.pmel[((size_t)b * PM_ROWS + cblk) * MEL_LD + pm_m] = pv;
  }
#ifdef XDTTS_LSTM_PROBE
  t_ctx = wall_clock64();
#endif
#ifdef XDTTS_LSTM_PROBE
  if (tid == 0 && step == 100 && (b % 17 == 0) && (cblk == 0 || cblk == 7))
    printf("probe softmax_ctx blk (%d,%d) in %llu softmax_done %llu ctx_pmel_done %llu out %llu\n", b, cblk, t_in % 1000000ull, t_sm % 1000000ull, t_ctx % 1000000ull, wall_clock64() % 1000000ull);
#endif
}


// D3 for batches as ONE launch: NB blocks per chunk (8 of 256 threads, or 4 of 512 inside the attention-LSTM launch
// below), block (b, part) owns 128 / NB attention dims for the energies and 512 / NB context columns.  What the two
// kernels above hand over through a grid boundary -- the partial energies, T floats per block -- crosses here the way
// the persistent engine's edges do: 8-byte {tag = step + 1, value} granules, stored and polled with relaxed
// agent-scope (sc1) accesses, no fence, no counter.  Every block publishes before it polls, the blocks of a chunk are
// neighbours in dispatch order, so the wait is short; a bounded spin sets d.att_err instead of hanging (the host
// then decodes the request again with the two-kernel form, api.cpp).  Chunks that have stopped are skipped.
// Every global load of the attention phase that does not depend on this step's attention-LSTM output, issued in the
// order the values are needed (vmcnt retires in issue order).
struct AttentionLoads {
  float4 hv[4], wq[4][4], l4[2], p4[2], v4, wc[8], pf[7];
  float awc_pre;
  int nv;
};
template <int NT, bool HG>
__device__ __forceinline__ void attention_loads(AttentionLoads &L, const DecoderBufs &d, int i, int cur, int b, int part,
                                                const float4 *__restrict__ Wq, const float *__restrict__ v_w) {
  constexpr int NWV = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = d.T;
  const int gq = NWV * part + wave;  // this wave's group of four attention dims
  if (!HG) {
    const float *h = d.att_h[cur ^ 1] + (size_t)b * ATT_RNN;
#pragma unroll
    for (int k = 0; k < 4; ++k) L.hv[k] = *reinterpret_cast<const float4 *>(h + 256 * k + 4 * lane);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) L.wq[r][k] = Wq[(size_t)(4 * gq + r) * (ATT_RNN / 4) + lane + 64 * k];
  const float *locg = d.loc + ((size_t)b * (ATT_DIM / 4) + gq) * T * 4, *pmg = d.pmem_t + ((size_t)b * (ATT_DIM / 4) + gq) * T * 4;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = lane + 64 * u < T ? lane + 64 * u : 0;
    L.l4[u] = *reinterpret_cast<const float4 *>(locg + 4 * t);
    L.p4[u] = *reinterpret_cast<const float4 *>(pmg + 4 * t);
  }
  L.v4 = *reinterpret_cast<const float4 *>(v_w + 4 * gq);
  L.nv = d.n_valid[b];
  L.awc_pre = tid < T ? ((i & 1) ? d.awc2 : d.awc)[b * T + tid] : 0.f;
  asm volatile("" ::: "memory");
}
// ... and those the stages after the softmax consume (projection rows, this block's slice of the encoder memory)
// PMEL false (two-launch form: the decoder-LSTM launch's tail multiplies the context columns of W_p itself): no projection rows
template <int NT, bool PMEL = true>
__device__ __forceinline__ void attention_loads_late(AttentionLoads &L, const DecoderBufs &d, int b, int part, const float *__restrict__ proj_wc) {
  constexpr int NWV = NT / 64, NB = ATT_DIM / (4 * NWV), COLS = EMB / NB, C4 = COLS / 4, TG = NT / C4;
  const int tid = threadIdx.x, T = d.T;
  // projection rows of the 64-column block cblk: thread (m, half) of each 256-thread group holds 32 columns of row m
  const int csub = tid >> 8, pm_m = (tid & 255) >> 1, pm_half = tid & 1, cblk = (COLS / CTX_COLS) * part + csub;
  if (PMEL) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      L.wc[k] = pm_m <= N_MEL ? reinterpret_cast<const float4 *>(proj_wc + ((size_t)cblk * (N_MEL + 1) + pm_m) * CTX_COLS + 32 * pm_half)[k]
                              : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int c4 = tid % C4, tg = tid / C4;
  const float4 *mem = reinterpret_cast<const float4 *>(d.memory + (size_t)b * T * EMB) + part * C4;
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    const int t = tg + TG * u;
    L.pf[u] = t < T ? mem[(size_t)t * (EMB / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  asm volatile("" ::: "memory");
}

// lds: T_MAX + (NT / 64) T_MAX + 17 (512 / NB) + 1024 floats.  HG: the attention-LSTM output of this step
// arrives as granules d.hg[b][1024] (published by the LSTM phase of the same launch) instead of the row-major vector.
template <int NT, bool HG, bool PMEL = true>
__device__ __forceinline__ void attention_chunk(const DecoderBufs &d, int i, int step, int b, int part, float *lds, AttentionLoads &L,
                                                const float *__restrict__ proj_wc) {
  constexpr int NWV = NT / 64, NB = ATT_DIM / (4 * NWV);       // a wave takes four attention dims
  constexpr int COLS = EMB / NB, C4 = COLS / 4, TG = NT / C4;  // context columns of this block; 16 time groups
  constexpr int CTX_PF = 7;
  static_assert(TG == 16 && COLS % CTX_COLS == 0 && NB <= ATT_EXCHANGE_BLOCKS && ATT_RNN % NT == 0, "block shape");
  float *s_e = lds, *s_eg = s_e + T_MAX, *s_part = s_eg + NWV * T_MAX, *s_ctx = s_part + TG * COLS, *s_hv = s_ctx + COLS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = d.T;
  const unsigned want = (unsigned)step + 1u, spin_limit = d.att_spins > 0 ? (unsigned)d.att_spins : AB_SPIN_LIMIT;
#ifdef XDTTS_LSTM_PROBE
  unsigned long long ap[8];
  ap[0] = wall_clock64();
#define APROBE(i) ap[i] = wall_clock64()
#else
#define APROBE(i) do { } while (0)
#endif
  const int gq = NWV * part + wave;
  const float *locg = d.loc + ((size_t)b * (ATT_DIM / 4) + gq) * T * 4, *pmg = d.pmem_t + ((size_t)b * (ATT_DIM / 4) + gq) * T * 4;
  float4 (&hv)[4] = L.hv;
  const float4 (&wq)[4][4] = L.wq;
  const float4 (&l4)[2] = L.l4, (&p4)[2] = L.p4, (&wc)[8] = L.wc, (&pf)[7] = L.pf;
  const float4 v4 = L.v4;
  const float *awc_in = (i & 1) ? d.awc2 : d.awc;
  float *awc_out = (i & 1) ? d.awc : d.awc2;
  const int nv = L.nv;
  const float awc_pre = L.awc_pre;
  const int csub = tid >> 8, pm_m = (tid & 255) >> 1, pm_half = tid & 1, cblk = (COLS / CTX_COLS) * part + csub;
  const int c4 = tid % C4, tg = tid / C4;
  const float4 *mem = reinterpret_cast<const float4 *>(d.memory + (size_t)b * T * EMB) + part * C4;
  // behind this block's own publish of h, ahead of the wait for everyone else's (issued behind the gather instead -- the polls then
  // do not queue behind these 166 KB -- the iteration measured 0.3-0.5 us slower: the other blocks' h is the later event either way)
  // (two-launch form: the memory slice is fetched behind the energies instead -- below -- so that the block fits 128 VGPRs and a
  // second block, the decoder LSTM's early partial, shares the CU)
  if (HG && PMEL) attention_loads_late<NT, PMEL>(L, d, b, part, proj_wc);
  APROBE(1);
  if (HG) {  // the 256 LSTM blocks of this launch each publish four units of every chunk
    constexpr int NG = ATT_RNN / NT;
    float g[NG];
    granule_gather<NG>(d.hg, (size_t)b * ATT_RNN + tid, NT, want, g, d.att_err, spin_limit);
#pragma unroll
    for (int k = 0; k < NG; ++k) s_hv[tid + NT * k] = g[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) hv[k] = *reinterpret_cast<const float4 *>(s_hv + 256 * k + 4 * lane);
  }
  APROBE(2);
  // ---- processed query: the four rows of this wave's dims, no LDS (wave_sum leaves the total in every lane) ----
  float4 q4;
  {
    float a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a[r] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) a[r] = dot4(wq[r][k], hv[k], a[r]);
      a[r] = wave_sum(a[r]);
    }
    q4 = make_float4(a[0], a[1], a[2], a[3]);
  }
  // ---- energies of this wave's four dims for every time step ----
  auto energy = [&](const float4 l, const float4 p) {
    float e = v4.x * fast_tanh(q4.x + l.x + p.x);
    e = fmaf(v4.y, fast_tanh(q4.y + l.y + p.y), e);
    e = fmaf(v4.z, fast_tanh(q4.z + l.z + p.z), e);
    return fmaf(v4.w, fast_tanh(q4.w + l.w + p.w), e);
  };
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (lane + 64 * u < T) s_eg[wave * T_MAX + lane + 64 * u] = energy(l4[u], p4[u]);
  for (int t = lane + 128; t < T; t += 64)
    s_eg[wave * T_MAX + t] = energy(*reinterpret_cast<const float4 *>(locg + 4 * t), *reinterpret_cast<const float4 *>(pmg + 4 * t));
  __syncthreads();
  APROBE(3);
  // ---- exchange: publish this block's partial, collect the chunk's NB ----
  u64 *slots = d.ep_g + (size_t)b * ATT_EXCHANGE_BLOCKS * T;
  for (int t = tid; t < T; t += NT) {
    float e = 0.f;
#pragma unroll
    for (int k = 0; k < NWV; k += 4) e += (s_eg[k * T_MAX + t] + s_eg[(k + 1) * T_MAX + t]) + (s_eg[(k + 2) * T_MAX + t] + s_eg[(k + 3) * T_MAX + t]);
    if (!(d.att_fault && (int)blockIdx.x == d.att_fault - 1)) granule_store(slots + (size_t)part * T + t, want, e);
  }
  if (HG && !PMEL) attention_loads_late<NT, PMEL>(L, d, b, part, proj_wc);  // (in flight while the partial energies cross)
  for (int t = tid; t < T; t += NT) {
    float pe[NB];
    granule_gather<NB>(slots, t, T, want, pe, d.att_err, spin_limit);
    float e = pe[0];
#pragma unroll
    for (int k = 1; k < NB; ++k) e += pe[k];
    s_e[t] = t >= nv ? -INFINITY : e;
  }
  __syncthreads();
  APROBE(4);
  // ---- softmax (every block of the chunk computes the same one), new weights, context slice, partial mel:
  //      as k_softmax_ctx ----
  if (wave == 0) {
    float m = -INFINITY;
    for (int t = lane; t < T; t += 64) m = fmaxf(m, s_e[t]);
    m = wave_max(m);
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
      const float ex = fast_exp(s_e[t] - m);
      s_e[t] = ex;
      sum += ex;
    }
    sum = wave_sum(sum);
    for (int t = lane; t < T; t += 64) s_e[t] = s_e[t] / sum;
  }
  __syncthreads();
  APROBE(5);
  if (part == 0)
    for (int t = tid; t < T; t += NT) {
      const float wv = s_e[t];
      d.aw[b * T + t] = wv;
      awc_out[b * T + t] = (t == tid ? awc_pre : awc_in[b * T + t]) + wv;
    }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (int t0 = tg; t0 < T; t0 += TG * CTX_PF) {
#pragma unroll
    for (int u = 0; u < CTX_PF; ++u) {
      const int t = t0 + TG * u;
      const float wv = t < T ? s_e[t] : 0.f;
      const float4 mv = k == 0 ? pf[u] : (t < T ? mem[(size_t)t * (EMB / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f));
      acc.x = fmaf(wv, mv.x, acc.x);
      acc.y = fmaf(wv, mv.y, acc.y);
      acc.z = fmaf(wv, mv.z, acc.z);
      acc.w = fmaf(wv, mv.w, acc.w);
    }
    ++k;
  }
  *reinterpret_cast<float4 *>(&s_part[tg * COLS + 4 * c4]) = acc;
  __syncthreads();
  if (tid < COLS) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < TG; ++g) v += s_part[g * COLS + tid];
    s_ctx[tid] = v;
    const int j = part * COLS + tid;
    d.ctx[b * EMB + j] = v;
    d.ctxf[((size_t)(j >> 2) * d.Bpad + b) * 4 + (j & 3)] = v;
  }
  if (PMEL) {
    __syncthreads();
    float pv = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) pv = dot4(wc[k2], *reinterpret_cast<const float4 *>(&s_ctx[CTX_COLS * csub + 32 * pm_half + 4 * k2]), pv);
    pv += dpp_move<0xB1, 0xf>(0.f, pv);  // lanes 2j, 2j+1 hold the two halves of row m
    if (pm_half == 0 && pm_m <= N_MEL) d.pmel[((size_t)b * PM_ROWS + cblk) * MEL_LD + pm_m] = pv;
  }
#ifdef XDTTS_LSTM_PROBE
  APROBE(6);
  if (tid == 0 && (step == 100 || step == 101) && (b == 0 || b == 17 || b == 40) && part == 0)
    printf("probe attention NT %d chunk %d step %d: loads issued %llu  h gathered %llu  energies %llu  e gathered %llu  softmax %llu  ctx+pmel %llu (x10ns)\n", NT, b, step,
           ap[1] - ap[0], ap[2] - ap[1], ap[3] - ap[2], ap[4] - ap[3], ap[5] - ap[4], ap[6] - ap[5]);
#endif
}

constexpr int attention_lds_floats(int nt) { return T_MAX + nt / 64 * T_MAX + 17 * (EMB / (ATT_DIM / (nt / 16))) + ATT_RNN; }

__global__ __launch_bounds__(256) void k_attention_b(DecoderBufs d, int i, int cur, const float4 *__restrict__ Wq,
                                                     const float *__restrict__ v_w, const float *__restrict__ proj_wc) {
  static_assert(CTX_BLOCKS == ATT_EXCHANGE_BLOCKS, "eight blocks per chunk");
  const int b = blockIdx.x / CTX_BLOCKS;
  const int step = d.ctl[0] + i;
  if (step >= d.nframes[b]) return;  // (the step limits change in the prenet kernel only: all blocks of a chunk agree)
  __shared__ __attribute__((aligned(16))) float lds[attention_lds_floats(256)];
  AttentionLoads L;
  attention_loads<256, false>(L, d, i, cur, b, blockIdx.x % CTX_BLOCKS, Wq, v_w);
  attention_loads_late<256>(L, d, b, blockIdx.x % CTX_BLOCKS, proj_wc);
  attention_chunk<256, false>(d, i, step, b, blockIdx.x % CTX_BLOCKS, lds, L, proj_wc);
}

// D2 + D3 for batches of up to 64 chunks in ONE launch: the 256 blocks run the attention-LSTM pass and publish their
// four hidden units of every chunk as granules (next to the B-operand copy the decoder LSTM reads after the grid
// boundary); blocks 4 b .. 4 b + 3 then turn into the attention blocks of chunk b.  All 256 blocks must be resident
// together (one per CU), as for the persistent engine; the same bounded spin covers a grid that is not.
// EARLY: the pass multiplies the 256 prenet columns only and adds d.att_part, the product of the other 1536 columns that
// 256 blocks of the preceding decoder-LSTM launch computed (att_early_partial).
// TWO: two-launch form -- no partial-mel rows of the context columns (the decoder-LSTM launch's tail reads d.ctx)
template <bool EARLY, bool TWO = false>
__global__ __launch_bounds__(64 * MFMA_WAVES, TWO ? 4 : 2) void k_att_lstm_attention(DecoderBufs d, int i, int cur, const float4 *__restrict__ Wm,
                                                                        const float *__restrict__ bias, const float4 *__restrict__ Wq,
                                                                        const float *__restrict__ v_w, const float *__restrict__ proj_wc,
                                                                        const float4 *__restrict__ dec_wm) {
  constexpr int NW = MFMA_WAVES, CN = EARLY ? PRENET : ATT_COLS, JJ = CN / NW / 16;
  static_assert(NW == 8 && attention_lds_floats(512) <= NW * 4 * 64 * 4, "the attention phase reuses the accumulator exchange area");
  const int blk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fg = lane >> 4;
  const float4 *wsrc = Wm + ((size_t)blk * (ATT_COLS / 16) + wave * JJ) * 64 + lane;
  const int step = d.ctl[0] + i;
  const bool a = lane < d.B && step < d.nframes[min(lane, d.B - 1)];
  const unsigned long long m = __ballot(a);
  const int nta = m ? (63 - __clzll((long long)m)) / 16 + 1 : 0;
  const float4 bz = *reinterpret_cast<const float4 *>(bias + (blk * 4 + fg) * 4);
  const float wa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  __shared__ __attribute__((aligned(16))) float s_acc[NW * 4 * 64 * 4];
  if (TWO && (int)blockIdx.x >= NBLK) {  // blocks 256..511 (with d.dec_part): the early partial of THIS step's decoder-LSTM pass over h_dec(s-1)
    att_early_role<1>(d, d.ctl[0] + i, cur, (int)blockIdx.x - NBLK, dec_wm, s_acc, 0, d.hring != nullptr);
    return;
  }
  // blocks 4 b .. 4 b + 3 are the attention blocks of chunk b; their loads for that phase go out as soon as the wave has
  // issued the last load of its LSTM pass, and arrive while it waits for the other waves and the other blocks
  const int b = blk >> 2, part = blk & 3;
  const bool attn = b < d.B && step < d.nframes[min(b, d.B - 1)];
  AttentionLoads L;
  auto hook = [&]() {
    if (attn) attention_loads<512, true>(L, d, i, cur, b, part, Wq, v_w);  // (the late group follows the publish of h: attention_chunk)
  };
  if (EARLY && TWO && d.att_hfirst) {  // + its own h_att(s-1) columns, ahead of the prenet columns (att_part then holds the context columns only)
    constexpr int HC = EARLY && TWO ? ATT_RNN : 0;
    const float4 *wblk = Wm + (size_t)blk * (ATT_COLS / 16) * 64 + lane;
    switch (nta) {
      case 1: lstm_mfma_pass<ATT_COLS, 0, 1, decltype(hook), 0, CN, EARLY, false, ATT_IN, HC>(d, 0, cur, step, blk, wblk, bz, wa, s_acc, m, 0, hook); break;
      case 2: lstm_mfma_pass<ATT_COLS, 0, 2, decltype(hook), 0, CN, EARLY, false, ATT_IN, HC>(d, 0, cur, step, blk, wblk, bz, wa, s_acc, m, 0, hook); break;
      case 3: lstm_mfma_pass<ATT_COLS, 0, 3, decltype(hook), 0, CN, EARLY, false, ATT_IN, HC>(d, 0, cur, step, blk, wblk, bz, wa, s_acc, m, 0, hook); break;
      case 4: lstm_mfma_pass<ATT_COLS, 0, 4, decltype(hook), 0, CN, EARLY, false, ATT_IN, HC>(d, 0, cur, step, blk, wblk, bz, wa, s_acc, m, 0, hook); break;
      default: return;
    }
  } else
  switch (nta) {
    case 1: lstm_mfma_pass<ATT_COLS, 0, 1, decltype(hook), 0, CN, EARLY>(d, 0, cur, step, blk, wsrc, bz, wa, s_acc, m, 0, hook); break;
    case 2: lstm_mfma_pass<ATT_COLS, 0, 2, decltype(hook), 0, CN, EARLY>(d, 0, cur, step, blk, wsrc, bz, wa, s_acc, m, 0, hook); break;
    case 3: lstm_mfma_pass<ATT_COLS, 0, 3, decltype(hook), 0, CN, EARLY>(d, 0, cur, step, blk, wsrc, bz, wa, s_acc, m, 0, hook); break;
    case 4: lstm_mfma_pass<ATT_COLS, 0, 4, decltype(hook), 0, CN, EARLY>(d, 0, cur, step, blk, wsrc, bz, wa, s_acc, m, 0, hook); break;
    default: return;  // (no chunk is active: nothing to attend to either)
  }
  if (!attn) return;
  batched_straggle(d, blk, step, 2);  // (between its publish of h and its share of the chunk's energies)
  attention_chunk<512, true, !TWO>(d, i, step, b, part, s_acc, L, proj_wc);
}

}  // namespace

size_t decoder_pmel_floats(int B) { return (size_t)B * PM_ROWS * MEL_LD; }

namespace {
__global__ void k_dimgroup_transpose(const float4 *__restrict__ in, float4 *out, int B, int T) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index of the [B][T][32] input
  if (i >= (size_t)B * T * (ATT_DIM / 4)) return;
  const int g = (int)(i % (ATT_DIM / 4)), t = (int)((i / (ATT_DIM / 4)) % T), b = (int)(i / ((size_t)(ATT_DIM / 4) * T));
  out[((size_t)b * (ATT_DIM / 4) + g) * T + t] = in[i];
}
}  // namespace

void launch_dimgroup_transpose(const float *in, float *out, int B, int T, hipStream_t s) {
  const size_t n = (size_t)B * T * (ATT_DIM / 4);
  hipLaunchKernelGGL(k_dimgroup_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4 *>(in),
                     reinterpret_cast<float4 *>(out), B, T);
  HIP_CHECK(hipGetLastError());
}

void launch_decoder_init(const DecoderBufs &d, const int *limits_dev, hipStream_t s) {
  if (d.xf) {  // batched mode keeps the cell states in the LSTM kernel's own order, padded to Bpad chunks
    HIP_CHECK(hipMemsetAsync(d.att_c, 0, sizeof(float) * (size_t)d.Bpad * ATT_RNN, s));
    HIP_CHECK(hipMemsetAsync(d.dec_c, 0, sizeof(float) * (size_t)d.Bpad * DEC_RNN, s));
  }
  if (d.ep_g) HIP_CHECK(hipMemsetAsync(d.ep_g, 0, sizeof(unsigned long long) * (size_t)d.B * CTX_BLOCKS * d.T, s));  // step tags restart at 1
  if (d.hg) HIP_CHECK(hipMemsetAsync(d.hg, 0, sizeof(unsigned long long) * (size_t)d.B * ATT_RNN, s));
  if (d.att_part) HIP_CHECK(hipMemsetAsync(d.att_part, 0, sizeof(float) * (size_t)NBLK * 4 * 64 * 4, s));  // step 0: context and hidden state are zero
  if (d.dec_part) HIP_CHECK(hipMemsetAsync(d.dec_part, 0, sizeof(float) * (size_t)NBLK * 4 * 64 * 4, s));
  if (d.hdg) HIP_CHECK(hipMemsetAsync(d.hdg, 0, sizeof(unsigned long long) * (size_t)d.B * (DEC_RNN + 96), s));  // (melg follows hdg)
  hipLaunchKernelGGL(k_decoder_init, dim3(d.B), dim3(256), 0, s, d, limits_dev);
  HIP_CHECK(hipGetLastError());
}

// Steps are enqueued in (even, odd) pairs: node i uses ping-pong parity i & 1, so a sequence must
// start on an even step and nsteps must be even.  The absolute step of node i is ctl[0] + i;
// k_advance moves ctl[0] on by nsteps at the end, so the same captured graph replays anywhere.
static void enqueue_steps(const DecoderBufs &d, const DeviceWeights &w, int i0, int nsteps, hipStream_t s);
void launch_decoder_steps(const DecoderBufs &d, const DeviceWeights &w, int nsteps, hipStream_t s) {
  if (nsteps % 2 != 0) fail(XDTTS_ERR_BAD_ARG, "decoder steps are enqueued in even/odd pairs");
  enqueue_steps(d, w, 0, nsteps, s);
  hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, s, d, nsteps);
  HIP_CHECK(hipGetLastError());
}
// Parity hook (xdtts_tacotron2_decoder_steps): node i alone, no advance of the step base
void launch_decoder_step_at(const DecoderBufs &d, const DeviceWeights &w, int i, hipStream_t s) {
  enqueue_steps(d, w, i, 1, s);
  HIP_CHECK(hipGetLastError());
}
// Parity hooks, batched engine with d.att_part: the early partial of node i's attention-LSTM pass from the state as it stands
// (inside a sequence the decoder-LSTM launch of node i - 1 computes it)
void launch_decoder_early(const DecoderBufs &d, const DeviceWeights &w, int i, hipStream_t s) {
  if (!(d.xf && w.att_wm.p && d.ep_g && d.hg && d.B <= 64 && d.att_part)) return;
  hipLaunchKernelGGL(k_att_early, dim3(decoder_two_launch(d) && d.dec_part ? 2 * NBLK : NBLK), dim3(64 * MFMA_WAVES), 0, s, d, i,
                     reinterpret_cast<const float4 *>(w.att_wm.p), reinterpret_cast<const float4 *>(w.dec_wm.p));
  HIP_CHECK(hipGetLastError());
}
// Two-launch form: the exchange buffers exist, the MFMA location role serves T, the fused attention launch is on
bool decoder_two_launch(const DecoderBufs &d) { return d.hdg && d.melg && d.att_part && d.xf && d.ep_g && d.hg && d.B <= 64 && d.T <= LOC_MFMA_T; }
// ... its sequence start: x and the location features of the first step (node 0) by the prenet launch; every later step's come
// from the tail / the early blocks of the previous decoder-LSTM launch.  No-op in the three-launch form.
void launch_decoder_prologue(const DecoderBufs &d, const DeviceWeights &w, hipStream_t s) {
  if (!(decoder_two_launch(d) && w.att_wm.p && w.dec_wm.p)) return;
  hipLaunchKernelGGL(k_prenet_b, dim3(PRENET_SPLIT * d.B + 2 * d.B), dim3(PRENET_BT), 0, s, d, 0, 0, w.pre0T.p, w.pre1T.p, w.proj_b.p, w.loc_conv.p, w.loc_denseT.p);
  HIP_CHECK(hipGetLastError());
}
void launch_decoder_advance(const DecoderBufs &d, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, s, d, n);
  HIP_CHECK(hipGetLastError());
}
static void enqueue_steps(const DecoderBufs &d, const DeviceWeights &w, int i0, int nsteps, hipStream_t s) {
  const int loc_tiles = (d.T + LOC_TT - 1) / LOC_TT;
  const float4 *att_w = reinterpret_cast<const float4 *>(w.att_w.p);
  const float4 *dec_w = reinterpret_cast<const float4 *>(w.dec_w.p);
  const float4 *q4 = reinterpret_cast<const float4 *>(w.q_w4.p), *wh4 = reinterpret_cast<const float4 *>(w.proj_wh4.p);
  // XDTTS_DEBUG_MIX (developer timing aid only; results are garbage when set): string over the
  // letters p,a,q,s,d selecting which kernels a step launches, e.g. "ppppp".
  const char *mix = getenv("XDTTS_DEBUG_MIX");
  const std::string order = mix ? mix : "paqsd";
  // LSTMs as MFMA GEMMs: the caller chose the batched layout (decoder_bufs: from BATCH_MFMA_MIN chunks, or the parity hook's engine 2 at any B)
  const bool batched = d.xf && w.att_wm.p && w.dec_wm.p;
  const float4 *att_wm = reinterpret_cast<const float4 *>(w.att_wm.p), *dec_wm = reinterpret_cast<const float4 *>(w.dec_wm.p);
  const bool fuse_aq = batched && d.ep_g && d.hg && d.B <= 64;
  const bool early = fuse_aq && d.att_part != nullptr;  // 1536 of the attention LSTM's 1792 columns ride in the previous decoder-LSTM launch
  const bool two = early && decoder_two_launch(d);       // ... and the prenet launch is the tail of the decoder-LSTM launch
  DecoderBufs dd = d;                                    // what the decoder-LSTM launch sees
  if (!two) dd.hdg = dd.melg = nullptr;
  if (!two) dd.dec_part = nullptr;
  if (!two) dd.hring = nullptr;
  if (!two) dd.hstage = dd.hcnt = nullptr;
  if (!two) dd.att_hfirst = 0;
  const TailWeights tw{reinterpret_cast<const float4 *>(w.proj_w.p), w.proj_b.p, w.pre0T.p, w.pre1T.p, w.loc_conv.p, w.loc_denseT.p};
  for (int i = i0; i < i0 + nsteps; ++i) {
    const int cur = i & 1;
    for (char k : order) {
      switch (k) {
        case 'p':
          if (two) break;  // (launch_decoder_prologue ran the first step's prenet; every later one is the previous decoder-LSTM launch's tail)
          if (batched)
            hipLaunchKernelGGL(k_prenet_b, dim3(PRENET_SPLIT * d.B + d.B * (d.T <= LOC_MFMA_T ? 2 : ((d.T + LOC_TT - 1) / LOC_TT + 7) / 8)), dim3(PRENET_BT), 0, s, d, i, 0, w.pre0T.p, w.pre1T.p,
                               w.proj_b.p, w.loc_conv.p, w.loc_denseT.p);
          else
            hipLaunchKernelGGL(k_prenet, dim3(PRENET_BLOCKS * d.B), dim3(256), 0, s, d, i, 0, w.pre0T.p, w.pre1T.p,
                               w.proj_b.p);
          break;
        case 'a':
          if (two)  // ... and no partial-mel rows: the decoder-LSTM launch's tail projects h_dec and the context itself
            hipLaunchKernelGGL((k_att_lstm_attention<true, true>), dim3(dd.dec_part ? 2 * NBLK : NBLK), dim3(64 * MFMA_WAVES), 0, s, dd, i, cur, att_wm, w.att_b.p,
                               reinterpret_cast<const float4 *>(w.q_w.p), w.v_w.p, w.proj_wc.p, dec_wm);  // (+ 256 blocks: the decoder LSTM's h_dec(s-1) columns)
          else if (early)  // attention LSTM (its 256 prenet columns + the early partial) + energies + softmax + context
            hipLaunchKernelGGL(k_att_lstm_attention<true>, dim3(NBLK), dim3(64 * MFMA_WAVES), 0, s, d, i, cur, att_wm, w.att_b.p,
                               reinterpret_cast<const float4 *>(w.q_w.p), w.v_w.p, w.proj_wc.p, dec_wm);
          else if (fuse_aq)  // attention LSTM + energies + softmax + context ('q' and 's' are then no-ops)
            hipLaunchKernelGGL(k_att_lstm_attention<false>, dim3(NBLK), dim3(64 * MFMA_WAVES), 0, s, d, i, cur, att_wm, w.att_b.p,
                               reinterpret_cast<const float4 *>(w.q_w.p), w.v_w.p, w.proj_wc.p, dec_wm);
          else if (batched)
            hipLaunchKernelGGL((k_lstm_mfma<ATT_COLS, 0>), dim3(NBLK, (d.B + 63) / 64), dim3(64 * MFMA_WAVES), 0, s, d, i, cur, att_wm, w.att_b.p, q4, att_wm, tw);
          else
            hipLaunchKernelGGL((k_lstm<ATT_COLS, 0>), dim3(NBLK), dim3(256), 0, s, d, i, cur, att_w, w.att_b.p, q4,
                               w.loc_conv.p, w.loc_denseT.p);
          break;
        case 'q':
          if (fuse_aq) break;
          if (batched && d.ep_g) {  // energies + softmax + context in one launch ('s' is then a no-op)
            hipLaunchKernelGGL(k_attention_b, dim3(CTX_BLOCKS * d.B), dim3(256), 0, s, d, i, cur, reinterpret_cast<const float4 *>(w.q_w.p),
                               w.v_w.p, w.proj_wc.p);
            break;
          }
          hipLaunchKernelGGL(k_qenergy, dim3(ATT_DIM / 4, batched ? std::max(2, (d.B + QE_GROUP - 1) / QE_GROUP) : 1), dim3(256), 0, s, d, i, cur,
                             reinterpret_cast<const float4 *>(w.q_w.p), w.v_w.p);
          break;
        case 's':
          if (batched && d.ep_g) break;
          hipLaunchKernelGGL(k_softmax_ctx, dim3(CTX_BLOCKS * d.B), dim3(256), 0, s, d, i, w.proj_wc.p);
          break;
        case 'd':
          if (batched) {  // (early: 256 more blocks multiply the next attention-LSTM pass's 1536 known columns)
            hipLaunchKernelGGL((k_lstm_mfma<DEC_COLS, 1>), dim3(early ? 2 * NBLK : NBLK, (d.B + 63) / 64), dim3(64 * MFMA_WAVES), 0, s, dd, i, cur, dec_wm, w.dec_b.p, wh4,
                               att_wm, tw);
          } else
            hipLaunchKernelGGL((k_lstm<DEC_COLS, 1>), dim3(loc_tiles * d.B + NBLK), dim3(256), 0, s, d, i, cur, dec_w,
                               w.dec_b.p, wh4, w.loc_conv.p, w.loc_denseT.p);
          break;
        default:
          break;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_location_all(DecoderBufs d, const float *__restrict__ loc_convT, const float *__restrict__ loc_denseT) {
  const int tiles = (d.T + LOC_TT - 1) / LOC_TT;
  location_role(d, blockIdx.x / tiles, blockIdx.x % tiles, loc_convT, loc_denseT);
}

// Parity hook, launch-per-stage engine: the location features of the CURRENT attention weights (row-major layout)
void launch_decoder_location(const DecoderBufs &d, const DeviceWeights &w, hipStream_t s) {
  const int loc_tiles = (d.T + LOC_TT - 1) / LOC_TT;
  hipLaunchKernelGGL(k_location_all, dim3(loc_tiles * d.B), dim3(256), 0, s, d, w.loc_conv.p, w.loc_denseT.p);
  HIP_CHECK(hipGetLastError());
}

// Parity hook, persistent engine: x(step) = prenet(decoder_input) by the launch-per-stage prenet kernel (node 0)
void launch_decoder_prenet(const DecoderBufs &d, const DeviceWeights &w, hipStream_t s) {
  hipLaunchKernelGGL(k_prenet, dim3(PRENET_BLOCKS * d.B), dim3(256), 0, s, d, 0, 0, w.pre0T.p, w.pre1T.p, w.proj_b.p);
  HIP_CHECK(hipGetLastError());
}

// Parity hook, batched engine: row-major [B][n] vectors <-> the MFMA-operand order [n/4][Bpad][4] the batched kernels keep
// their hidden states, cell states and context in (dir 0: import, 1: export)
namespace {
__global__ void k_frag_convert(float *rowmajor, float *frag, int B, int Bpad, int n, int dir) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  const int b = i / n, j = i % n;
  const size_t f = ((size_t)(j >> 2) * Bpad + b) * 4 + (j & 3);
  if (dir) rowmajor[i] = frag[f];
  else frag[f] = rowmajor[i];
}
}  // namespace
void launch_frag_convert(float *rowmajor, float *frag, int B, int Bpad, int n, int dir, hipStream_t s) {
  hipLaunchKernelGGL(k_frag_convert, dim3((unsigned)((B * n + 255) / 256)), dim3(256), 0, s, rowmajor, frag, B, Bpad, n, dir);
  HIP_CHECK(hipGetLastError());
}

// After the last step of a sequence: finishes the projection of the final step (frames, gate).
void launch_decoder_flush(const DecoderBufs &d, const DeviceWeights &w, hipStream_t s) {
  if (decoder_two_launch(d) && w.att_wm.p && w.dec_wm.p) return;  // (two-launch form: every step's tail stored its own frame)
  if (d.xf)
    hipLaunchKernelGGL(k_prenet_b, dim3(PRENET_SPLIT * d.B), dim3(PRENET_BT), 0, s, d, 0, 1, w.pre0T.p, w.pre1T.p, w.proj_b.p, w.loc_conv.p, w.loc_denseT.p);
  else
    hipLaunchKernelGGL(k_prenet, dim3(PRENET_BLOCKS * d.B), dim3(256), 0, s, d, 0, 1, w.pre0T.p, w.pre1T.p, w.proj_b.p);
  HIP_CHECK(hipGetLastError());
}

}  // namespace xdtts
