// kernels.h -- host-side launch interface of the HIP kernels (gfx950).
#pragma once
#include "common.h"
#include "weights.h"

namespace xdtts {

// ---- decoder loop (src/tacotron2/mod.rs:272-342) ---------------------------------------------
// Device-side state of B independent chunks decoded in lock-step.  One "step" = one
// decoder_iter.onnx call of the reference (mod.rs:304) for every still-active chunk.
struct DecoderBufs {
  int B, T;
  const float *memory;  // [B][T][512]   encoder output (mod.rs:382)
  const float *pmem;    // [B][T][128]   processed_memory (mod.rs:383)
  const int *n_valid;   // [B]           un-padded length -> mask (mod.rs:219-220)
  float *att_h[2];      // [B][1024]     ping-pong by step parity
  float *att_c;         // [B][1024]
  float *dec_h[2];      // [B][1024]
  float *dec_c;         // [B][1024]
  float *aw, *awc;      // [B][T]        attention_weights, attention_weights_cum
  float *ctx;           // [B][512]      attention_context
  float *x;             // [B][256]      prenet output
  float *loc;           // [B][T][128]   location features of the current step
  float *e_part;        // [B][32][T]    per-block partial energies
  float *pmel;          // [B][264][84]  partial mel/gate sums: 8 context blocks, 256 decoder-LSTM blocks
  float *frames;        // [B][max_steps][80]  decoder_output per step (time-major)
  float *gates;         // [B][max_steps]      gate_prediction logits
  int *nframes;         // [B] in: step limit; out: frames emitted (gate may lower it)
  int *ctl;             // [0] absolute step of the first node of the graph being replayed
  int max_steps;
  int use_gate;
  float gate_threshold;
  float gate_lo, gate_hi;  // logits below / above which sigmoid(gate) > gate_threshold is decided without the sigmoid (device_utils.h: gate_fires)
  int dropout_mode;
  uint32_t dropout_seed, item_base;
  // dropout_mode 2: the caller's keep bytes [chunk][drop_steps][2][256] on the device (chunk = index within the call:
  // item_perm[b] in a sorted batch, b otherwise; views of a batch advance the pointer)
  const unsigned char *drop_masks;
  int drop_steps;
  // Batched mode only (B >= BATCH_MFMA_MIN; null otherwise).  A second copy of the vectors the two
  // LSTM GEMMs consume, in MFMA B-operand order [K/4][Bpad][4] (Bpad = B rounded up to 16, padding
  // zero): lane (chunk, k-quad) of a 16-chunk tile loads one 16-byte vector and the 64 lanes of a wave
  // cover four fully used 256-byte runs.  Written by the producers next to the row-major vectors.
  float *xf, *ctxf, *att_hf[2], *dec_hf[2];
  int Bpad;
  float *awc2;           // [B][T] second cumulative-weights buffer (ping-pong by step parity, batched mode)
  const int *item_perm;  // [B] dropout-stream index of chunk b (the batch is sorted by length), or null = b
  // Persistent engine: the context columns of its LSTM / projection rows folded into the encoder memory,
  // ctx_fold [B][CTXF_ROWS][CTXF_LD]: row n of chunk b holds W_n[ctx cols] . memory_b[t] for t < T (rows: 4096 attention-LSTM
  // rows in packed order, 4096 decoder-LSTM rows, 81 projection rows; DeviceWeights::ctx_w) -- one GEMM per request
  // (api.cpp) instead of a fold loop in every launch.  null = the kernel folds for itself.
  const float *ctx_fold;
  const float *dec_in;   // parity hook (xdtts_tacotron2_decoder_step): decoder_input [B][80] of this step, or null
  // Batched mode: processed_memory a second time as [B][32 dim groups][T][4] -- the energies kernel reads 4 dims of
  // every time step, 16 bytes out of each 512-byte row of the [T][128] layout; in batched mode `loc` has this layout too
  const float *pmem_t;
  // Batched mode, one-launch attention (k_attention_b): the 8 blocks of a chunk exchange their partial energies as
  // {tag = step + 1, value} granules [B][8][T]; null = the energies / context kernel pair.  att_err: set by a block
  // whose bounded spin ran out.
  unsigned long long *ep_g;
  int *att_err;
  // ... and the attention LSTM in the same launch (k_att_lstm_attention, B <= 64): its output as granules [B][1024]
  // in place of the row-major att_h; null = separate launches
  unsigned long long *hg;
  // ... and, with it, the EARLY partial pre-activations: the columns of an LSTM GEMM whose operand exists one launch ahead
  // are multiplied there, on matrix cores the other launch leaves idle.  att_part [256 blocks][4 tiles][64 lanes][4 gates]
  // (the MFMA D layout of the block's cell-update waves) = W_att[:, 256..1791] . [ctx(s-1) ; h_att(s-1)], written by 256
  // extra blocks of the decoder-LSTM launch of step s-1, added by the attention-LSTM pass of step s, which then only
  // multiplies the 256 prenet columns.  null = the attention launch runs the whole K.
  float *att_part;
  // ... and the TWO-launch form (decoder.hip: dec_tail_chunk): h_dec as granules [B][1024], published by the decoder-LSTM blocks and
  // gathered by the chunk's four projection / prenet blocks of the same launch, which exchange the mel + gate values [B][96];
  // null = the prenet launch sums the partial-mel rows.  One allocation: melg = hdg + B * 1024.
  unsigned long long *hdg, *melg;
  // ... with the decoder LSTM's own-state columns early too: dec_part [256][4][64][4] = W_dec[:, 1536..2559] . h_dec(s-1), written by 256
  // extra blocks of the attention launch of step s, added by the decoder-LSTM pass of step s, which then multiplies [h_att ; ctx] only
  float *dec_part;
  int att_spins, att_fault;  // test hooks: poll limit (0 = default) and a block (index + 1) that never publishes its energies
  int tail_fault;            // test hook, two-launch form: a decoder-LSTM block (index + 1) that never publishes its h_dec granules
  int att_slow;              // test hook: a block (index + 1) of both batched launches that stalls ~7 us at a different point of every step
  // Two-launch form, round 6: h_att(s) ALSO as a write-once ring of plain values in B-operand order [step][1024 / 4][Bpad][4]
  // (0xFFFFFFFF = not yet written; a value is its own arrival flag, decoder_persistent16.hip), so that the extra blocks of the
  // attention launch multiply the decoder LSTM's h_att columns INSIDE that launch, behind its h_dec columns, while the attention
  // chain runs -- the decoder-LSTM launch's critical pass then covers the 512 context columns only.  null = that pass covers
  // [h_att ; ctx] (1536 columns) as before.  hring_steps: steps the ring is laid out for.
  unsigned *hring;
  int hring_steps;
  // ... and the per-XCD relay of that ring (decoder.hip, att_early_partial): hstage [8 XCDs][2 step parities][1024 / 4][Bpad][4], copies of
  // the step's slab per XCD; hcnt [8][steps][64] = {octets of rows drawn, ..} per XCD and step (zero at the start of a request).
  // null = every block polls the ring itself.
  unsigned *hstage, *hcnt;
  // Two-launch form, round 6: the attention launch multiplies its OWN h_att(s-1) columns (1024 of the 1792) ahead of the prenet
  // columns -- old data, in the ~4 us its first loads of x(s) take to arrive -- and att_part, written by the decoder-LSTM launch's
  // extra blocks, covers the 512 context columns only: 1024 columns x chunks of matrix work leave the launch that is bound by it.
  int att_hfirst;
};
constexpr int ATT_EXCHANGE_BLOCKS = 8;  // granule rows per chunk (CTX_BLOCKS in decoder.hip)
// [B][T][128] -> [B][32][T][4]
void launch_dimgroup_transpose(const float *in, float *out, int B, int T, hipStream_t s);

// Enqueues `nsteps` decoder steps on `s` (5 kernels each) and advances the device step base.
void launch_decoder_steps(const DecoderBufs &d, const DeviceWeights &w, int nsteps, hipStream_t s);
// Parity hook (xdtts_tacotron2_decoder_steps): node i of a sequence alone (state parity i & 1; absolute step ctl[0] + i),
// the advance of the step base, the location features of the current attention weights (launch-per-stage engine; the
// batched engine computes them inside its prenet launch), and the batched engine's state layout conversion.
void launch_decoder_step_at(const DecoderBufs &d, const DeviceWeights &w, int i, hipStream_t s);
void launch_decoder_advance(const DecoderBufs &d, int n, hipStream_t s);
void launch_decoder_early(const DecoderBufs &d, const DeviceWeights &w, int i, hipStream_t s);  // (no-op without d.att_part)
bool decoder_two_launch(const DecoderBufs &d);
void launch_decoder_prologue(const DecoderBufs &d, const DeviceWeights &w, hipStream_t s);  // (no-op in the three-launch form)
void launch_decoder_location(const DecoderBufs &d, const DeviceWeights &w, hipStream_t s);
void launch_decoder_prenet(const DecoderBufs &d, const DeviceWeights &w, hipStream_t s);
void launch_frag_convert(float *rowmajor, float *frag, int B, int Bpad, int n, int dir, hipStream_t s);
// After the last step of a sequence: completes the final frame's projection (frames, gate).
void launch_decoder_flush(const DecoderBufs &d, const DeviceWeights &w, hipStream_t s);
size_t decoder_pmel_floats(int B);
// Zeroes the recurrent state (DecoderState::new, mod.rs:202-233) and sets the step limits.
void launch_decoder_init(const DecoderBufs &d, const int *limits_dev, hipStream_t s);

// ---- persistent weight-stationary decoder (decoder_persistent.hip), small lock-step batches ------
constexpr int CTXF_ROWS = 2 * 4096 + 128, CTXF_LD = 128;  // rows of the context-fold table (81 projection rows, padded), its row stride
// ctx_w [CTXF_ROWS][512]: the context columns of att_w / dec_w (packed row order) and proj_w, zero rows behind
void launch_pack_ctx_rows(const float *att_w, const float *dec_w, const float *proj_w, float *ctx_w, hipStream_t s);
constexpr int PERSIST_B_MAX = 2;    // chunks in lock-step (register file + LDS of a CU hold the slices and two chunks' state)
constexpr int PERSIST_T_MAX = 128;  // encoder steps (the reference's window is 100, mod.rs:363)
// Granule exchange buffers: 8-byte {tag, value} records, [2 step parities][B][n] each.
struct PersistBufs {
  unsigned long long *x, *hatt, *ep, *hdec, *mel;  // (neither the context nor the attention weights cross)
  int *err;  // set by a workgroup whose bounded spin ran out
  int spins, fault;  // developer/test knobs: poll limit (0 = default) and a workgroup (index + 1) that never runs
  int slow;          // test hook: a workgroup (index + 1) that stalls ~7 us at a different point of every step (straggler:
                     // the two-slot exchange must keep every other workgroup from running more than one step ahead of it)
  int skew;    // 2-chunk launch that ends with its first chunk: the chunks run two phases apart (decoder_persistent.hip, "skewed pair")
  int both_run;  // 2-chunk launch without `shrink` whose chunks are both known to run all of its steps (gate-less decode)
  int shrink;  // 2-chunk launch: end as soon as one chunk stops (the host continues with a 1-chunk launch)
  int first;  // delay before a critical consumer's first poll, x 256 clocks (developer knob)
  int efirst;  // the attention role's first poll of the partial energies (it has just published its own slice)
  int xfirst;  // the same for the projection role's first poll of x (it has just published its own 16 columns)
  int pfirst;  // the same for the projection role's first poll of h_dec (behind its weight fetch and mask hashing)
  int xlazy, clazy;  // first-poll delay of the x / ctx consumers that are not their producers, x 256 clocks
  int lazy;  // late-poll delay of the off-critical-path consumers, x 256 clocks
  unsigned long long *prof;  // developer profile build only: [256][16] phase clocks, else null
};
size_t persist_granule_words(int B);
PersistBufs persist_bufs(unsigned long long *base, int *err, int B);
// The same exchange seen by a launch over chunks [b0, b0 + n) of the seeded batch.
PersistBufs persist_view(const PersistBufs &g, int b0);
bool decoder_persistent_supported(int device, int B, int T);
// After launch_decoder_init: clears the exchange and publishes x(0) with the chunks' active bits.
void launch_persist_seed(const DecoderBufs &d, const PersistBufs &g, const int *limits_dev, hipStream_t s);
// Parity hook: the same for a sequence that starts at `step` with the prenet output x [B][256] already computed (d.x)
void launch_persist_seed_at(const DecoderBufs &d, const PersistBufs &g, const int *limits_dev, int step, hipStream_t s);
// Runs up to `nsteps` decoder steps in one launch (ends early when every chunk has stopped);
// frames/gates/nframes are complete on return, ctl[0] = steps executed.  No flush needed.
void launch_decoder_persistent(const DecoderBufs &d, const DeviceWeights &w, const PersistBufs &g, int nsteps,
                               hipStream_t s);

// ---- persistent weight-stationary decoder for 3..8 chunks (decoder_persistent8.hip): the LSTMs of all chunks as one MFMA
// stream per wave, the context as a sixth exchange ------------------------------------------------------------------------
constexpr int P8_B_MAX = 16;  // 3..8 chunks: k_decoder_persistent8<4 / 8>; 9..16: k_decoder_persistent16 (same exchange layout, 16 chunk slots)
constexpr int P8_STEPS_MAX = 16384;  // longest request it takes: 190 s of speech, 1.5 GB of ring at 8 chunk slots
struct P8Bufs {
  unsigned *rx, *rhatt, *rctx, *rhdec;  // write-once rings of plain values [step][chunk slots][n], 0xFFFFFFFF = not yet written
  unsigned long long *ep, *mel;         // {tag, value} granules, [2 step parities][chunk slots][n] each (row strides: the serving kernel's slot count)
  int *err;          // set by a workgroup whose bounded spin ran out
  int spins, fault;  // test hooks: poll limit (0 = default) and a workgroup (index + 1) that never runs
  int delay[6];      // naps (64 clocks each) before the first poll of h_att / ctx / h_dec / x / the partial energies / the mel rows (16-slot kernel)
  int ring_steps;    // steps the rings are laid out for
  unsigned long long *prof;  // developer build (-DXDTTS_P8_PROFILE): [workgroup][32] phase clocks
};
size_t p8_exchange_words(int B, int nsteps);
P8Bufs p8_bufs(unsigned long long *base, int *err, int B, int nsteps);
bool decoder_p8_supported(int device, int B, int T);
void launch_p8_seed(const DecoderBufs &d, const P8Bufs &g, const int *limits_dev, hipStream_t s);
void launch_p8_seed_at(const DecoderBufs &d, const P8Bufs &g, const int *limits_dev, int step, hipStream_t s);
// Runs up to `nsteps` decoder steps of d.B <= 8 chunks in one launch (ends early when every chunk has stopped); frames / gates /
// nframes are complete on return, the state is written back, ctl[0] = steps executed.
void launch_decoder_p8(const DecoderBufs &d, const DeviceWeights &w, const P8Bufs &g, int nsteps, hipStream_t s);
// the 16-slot kernel behind the two functions above (decoder_persistent16.hip)
struct P8Weights {
  const float4 *att_w, *dec_w, *q_w, *proj_w;
  const float *att_b, *dec_b, *v_w, *loc_fused, *proj_b, *pre0T, *pre1T;
};
bool decoder_p16_supported(int device, int T);
void launch_decoder_p16(const DecoderBufs &d, const P8Weights &w, const P8Bufs &g, int nsteps, hipStream_t s);

// ---- NT GEMM on the f32 MFMA: C = act(A W^T + bias) (+R) ---------------------------------------
struct GemmArgs {
  const float *A;
  long lda, strideA;  // row m of item z starts at A + z*strideA + m*lda (rows may overlap: conv)
  const float *W;     // [N][K], K contiguous
  const float *bias;  // [N] or null
  float *C;
  long ldc, strideC;
  const float *R;     // residual, indexed like an un-transposed C; or null
  long ldr, strideR;
  int M, N, K, batch;
  // ragged batches (post-net over chunks of different length): per-item row count and extra element
  // offset of C, used when ragged != 0 (batch <= GEMM_RAGGED_MAX); blocks past an item's rows exit
  int ragged;
  int Mz[64];
  long Cz[64];
  int act;            // 0 none, 1 relu, 2 tanh, 3 pow(max(x,0), p)
  int transpose_out;  // store C[n*ldc + m]
  int ldc_rows;       // ragged + transpose_out: item z's row stride is its own Mz[z] -- one dense (N x Mz[z]) matrix per item at C + Cz[z]
  float p;
  // v = alpha * (A W^T) + bias; R is added scaled by beta, before the activation when r_before_act
  // (0 in alpha / beta means 1: zero-initialised args keep the plain form)
  float alpha, beta;
  int r_before_act;
  int xcd_rows;       // set by launch_gemm_nt: block -> tile mapping that keeps a row tile's column tiles on one XCD (gemm.hip)
  // split-K (gemm_splitk_plan): K slices per tile (0 / 1 = none), their meeting place [tiles][slices][1024 floats] and the tiles'
  // arrival counters (zero between launches)
  int splitk;
  float *ws;
  unsigned *cnt;
  int tile;           // 0: launch_gemm_nt's own choice; 32 / 64: the plan's
};
int gemm_splitk_plan(const GemmArgs &g, size_t *ws_floats, size_t *tiles, int *tile);
constexpr int GEMM_RAGGED_MAX = 64;  // (the argument block stays under 1 KB)
void launch_gemm_nt(const GemmArgs &g, hipStream_t s);

// ---- encoder (encoder.onnx, mod.rs:379) ---------------------------------------------------------
// ids [B][T] -> rows [pad, pad+T) of the zero-padded time-major buffer xpad [B][T+2*pad][512]
void launch_embed(const int64_t *ids, const float *emb, float *xpad, int B, int T, int pad, hipStream_t s);
// xproj [2][B][T][1024] (W_ih x + b), WhhT[dir] [256][1024] -> memory [B][T][512]
void launch_bilstm(const float *xproj, const float *whhT_fwd, const float *whhT_bwd, float *memory,
                   int B, int T, hipStream_t s);

// Same recurrence spread over 4 CUs per (direction, chunk) with W_hh resident in registers and a
// tagged-granule exchange of the hidden state (encoder.hip).  Needs all 8*B blocks co-resident.
size_t bilstm_coop_exchange_words(int B);
void launch_bilstm_coop(const float *xproj, const float *whhT_fwd, const float *whhT_bwd, float *memory,
                        unsigned long long *exchange, int *err, int B, int T, int group, hipStream_t s);

// ---- Griffin-Lim -------------------------------------------------------------------------------
struct GlBufs {
  int F, n_fft, hop, nb;   // frames, 1024, 256, 513
  float *S;                // [F][nb]        linear magnitude
  float2 *ang, *ang2;      // [F][nb]        unit-modulus phase estimate (ping-pong for the fused iteration)
  float2 *tprev;           // [F][nb]        previous rebuilt spectrum
  float *frames;           // [F][n_fft]     windowed time frames
  float *wss_inv;          // [hop*(F-1)]    window sum-of-squares divisor per output sample
  const float2 *tw;        // [n_fft]        exp(-2*pi*i*k/n_fft)
  const float *win;        // [n_fft]        periodic hann
};
// Persistent Griffin-Lim (griffinlim.hip: k_gl_persistent): all iterations + the final ISTFT in one
// launch, one workgroup per CU owning 3..TF consecutive frames, state in LDS, 768-sample overlaps
// exchanged with the two neighbours as tagged 8-byte granules.
constexpr int GLP_TF_MAX = 8;  // frames per workgroup (LDS: 10.3 KB of state + 4 KB of frame each; 8 waves = 2 per SIMD)
// One workgroup's share when a launch covers SEVERAL utterances (vocoder batch): the utterances' frames
// are concatenated in S / angles / previous spectrum, workgroups never span two utterances and exchange
// overlaps only inside their own.
struct GlSeg {
  int fbase;   // row of the utterance's first frame in the concatenated arrays
  int F;       // frames of the utterance
  int f0;      // first own frame, within the utterance
  int n_own;   // own frames (3..TF)
  int first;   // no left neighbour
  int last;    // no right neighbour
  int abase;   // offset of the utterance's samples in the audio output
  int pad;
};
struct GlPersist {
  const GlSeg *segs;        // [nblk] or null = one utterance of g.F frames split evenly over the workgroups
  unsigned long long *xch;  // [nblk][2 parities][2 sides][768] granules
  int *err;                 // set when a bounded spin ran out
  unsigned epoch;           // tag base of this call (tags = epoch + iteration + 1; never reused within the buffer's life)
  int nblk, TF;
  int per_cu;               // 2: the launch holds up to two 4-frame workgroups per CU (vocoder batch), else 0 / 1
  int spins;                // test hook: poll limit (0 = default)
  int slow;                 // test hook: a workgroup (index + 1) that stalls ~7 us at a different point of every iteration
  int poll_delay;           // first poll of the neighbours' granules: n >= 0: n x 128 clocks after the publish, before the own overlap-add; n < 0: behind it (+ (-1 - n) x 128 clocks)
  int gen_phase;            // 1: the kernel draws the seeded initial phase itself (angles = exp(2 pi i u), previous spectrum 0)
  unsigned seed;            //    instead of reading ang_in / tprev_in (saves the phase-init launch and a round trip through HBM)
  float2 *ang_out, *tprev_out;  // parity hook: final state, or null
  unsigned long long *prof;     // developer profile build only: [nblk][12] phase clocks, else null
};
bool gl_persistent_plan(int F, int n_cu, int *TF, int *nblk);
size_t gl_persistent_xch_words(int nblk);
bool gl_persistent_supported(int device, int *n_cu, int *per_cu4);
void launch_gl_persistent(const GlBufs &g, const GlPersist &p, const float2 *ang_in, const float2 *tprev_in, int n_iter,
                          float alpha, float *audio, hipStream_t s);
// mode 0: exp (natural-log mel), 1: copy (already linear), 2: 10^x
void launch_gl_exp_transpose(const float *mel_80xF, float *out_Fx80, int n_mels, int F, int mode, hipStream_t s);
// out[F][nb] = in[F][ld]^p (p == 1: copy): the exponent of mel->linear after the NNLS refinement
void launch_gl_pow_rows(const float *in, int ld, float *out, int nb, int F, float p, hipStream_t s);
// Output normalisation (mode 1: y / max|y|, 2: y * target / rms(y)) of n_utt utterances in two launches: utterance u is
// the samples [tab_dev[u].x, + tab_dev[u].y) of y (tab_dev == nullptr: one utterance, [first, first + n_max)); n_max = the
// longest of them; parts = GLN_SCRATCH floats of device scratch per utterance.  Deterministic (fixed reduction order).
constexpr int GLN_PARTS = 64;
constexpr int GLN_SCRATCH = 2 * GLN_PARTS;  // floats of scratch per utterance: sums of squares / peaks
void launch_gl_output_normalise(float *y, const int2 *tab_dev, int n_utt, int first, int n_max, int mode, float target,
                                float *parts, hipStream_t s);
void launch_gl_phase_init(const GlBufs &g, uint32_t seed, const float *phase0_dev, hipStream_t s);
// batch form: g.F = total frames of the concatenated utterances; frame_local[f] = index of row f inside its utterance
void launch_gl_phase_init_batch(const GlBufs &g, uint32_t seed, const int *frame_local, hipStream_t s);
void launch_gl_prepare(const GlBufs &g, hipStream_t s);                  // wss_inv for this F
void launch_gl_iterations(const GlBufs &g, int n_iter, float alpha, float *audio, hipStream_t s);  // + final ISTFT
const float2 *launch_gl_iterate(const GlBufs &g, int n_iter, float alpha, hipStream_t s);         // iterations only
void launch_gl_final(const GlBufs &g, const float2 *ang, float *audio, hipStream_t s);            // final ISTFT
// parity hook: iteration state in the crate's (n_bins x F x 2) layout <-> device [F][nb] float2
void launch_gl_state_import(const GlBufs &g, const float *ang_in, const float *reb_in, hipStream_t s);
void launch_gl_state_export(const GlBufs &g, const float2 *ang, const float2 *tprev, float *ang_out, float *reb_out,
                            hipStream_t s);
void launch_transpose(const float *in, float *out, int rows, int cols, hipStream_t s);
// Ragged row copy: item z (< n <= GEMM_RAGGED_MAX) has rows[z] rows of `cols` floats at src + z * src_stride; they go
// to dst + z * dst_stride (one launch instead of a memcpy per chunk: the post-net's padded layer-0 input)
void launch_copy_rows(const float *src, size_t src_stride, float *dst, size_t dst_stride, const int *rows, int n, int cols,
                      hipStream_t s);

}  // namespace xdtts
