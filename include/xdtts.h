/*
 * xdtts.h -- C ABI of libxdtts_hip.so: the MI355X (gfx950) implementation of xd-tts's
 * mel-spectrogram synthesis + vocoding hot path.
 *
 * This is the drop-in boundary.  Every entry point below is what a Rust `extern "C"` block in
 * the reference's src/tacotron2/mod.rs (and a replacement for the `griffin_lim` crate import at
 * src/tacotron2/mod.rs:67-68, src/lib.rs:5) would bind; the reference interface each one
 * replaces is cited as file:line under /root/reference.  INTEGRATION.md shows the Rust shim.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; all floats are IEEE fp32, ids are i64
 *   - every fallible call returns xdtts_status (0 = ok); xdtts_last_error() gives a thread-local
 *     message; no exception or panic crosses this boundary
 *   - handles are opaque, created by *_load / *_new and destroyed by *_free; one handle is bound
 *     to one GPU (device_id) and serialises calls on its own HIP stream
 *   - output buffers returned through `float **` are library-allocated pinned host memory: copy,
 *     then release with xdtts_free()
 *   - mel layout across the boundary is the reference's Array2<f32> (80, F), C order
 *     (src/tacotron2/mod.rs:349-355,430); audio is Vec<f32> (src/lib.rs:141)
 */
#ifndef XDTTS_H
#define XDTTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t xdtts_status;
enum {
  XDTTS_OK = 0,
  XDTTS_ERR_BAD_ARG = 1,  /* null pointer, bad shape, id out of range */
  XDTTS_ERR_IO = 2,       /* weight container missing / malformed (anyhow context, mod.rs:249,254,259) */
  XDTTS_ERR_HIP = 3,      /* HIP runtime error */
  XDTTS_ERR_OOM = 4,
  XDTTS_ERR_TOO_LONG = 5, /* a chunk longer than max_chunk: the reference's assert!, mod.rs:363 */
  XDTTS_ERR_NO_DEVICE = 6 /* no gfx950 device: the product path never falls back to the CPU */
};

typedef struct xdtts_tacotron2 xdtts_tacotron2;
typedef struct xdtts_griffinlim xdtts_griffinlim;

/* Decoder options.  Defaults (xdtts_infer_opts_default) are the reference's hard-coded
 * constants: gate_threshold 0.6 and max_steps 1000 (src/tacotron2/mod.rs:279-280), window 100
 * (src/tacotron2/mod.rs:363,369-371,399). */
typedef struct {
  float gate_threshold;
  int32_t max_steps;
  int32_t fixed_steps;   /* 0: stop on the gate (reference behaviour); >0: emit exactly this many
                            frames per chunk (deterministic work for benchmarks/parity) */
  int32_t dropout_mode;  /* 0 off; 1 seeded counter-based masks (the exported decoder graph keeps
                            the prenet's p=0.5 dropout on at inference); 2 explicit: the caller's
                            keep masks (dropout_masks below) */
  uint32_t dropout_seed;
  int32_t max_chunk;     /* encoder window; chunks are zero-padded to exactly this length */
  uint32_t item_base;    /* index of the first chunk in the dropout stream (batch sharding) */
  float fixed_frames_per_id; /* >0 (and fixed_steps == 0): chunk of n ids emits round(n * this)
                            frames -- deterministic work proportional to the chunk length */
  /* dropout_mode 2 (SURVEY.md section 8(b) "explicit(mask ptr)"): keep bytes
   * [chunk][dropout_mask_steps][2 prenet layers][256 units], non-zero = keep (the kept value is
   * doubled, p = 0.5), chunk = index of the chunk within this call, step = frame index.  This is
   * the hook for comparing with ONE recorded run of the real decoder_iter.onnx, whose in-graph
   * RandomUniform draws (src/tacotron2/mod.rs:304) cannot be seeded from outside.  Every chunk's
   * step limit must be <= dropout_mask_steps.  Host memory, read during the call only. */
  const uint8_t *dropout_masks;
  int32_t dropout_mask_steps;
} xdtts_infer_opts;

void xdtts_infer_opts_default(xdtts_infer_opts *opts);

/* ---- Tacotron2 -------------------------------------------------------------------------- */

/* Tacotron2::load(path) -- src/tacotron2/mod.rs:242-267.  `dir` is the reference's model directory:
 * encoder.onnx, decoder_iter.onnx and postnet.onnx (mod.rs:246-259) are read directly (a minimal
 * protobuf reader pulls the weights out of the graphs; nothing of the graphs is executed), or, if
 * present, the flat container `tacotron2.xdtw` written by xdtts_tacotron2_save.  A git-LFS pointer
 * file in place of a graph gives XDTTS_ERR_IO with a message naming `git lfs pull`.
 * What the reader expects of the graphs (torch.onnx.export of the NVIDIA model, csrc/onnx_load.cpp): attention_rnn,
 * decoder_rnn and the encoder BiLSTM as ONNX `LSTM` nodes (the exporter script's LSTMCell -> LSTM form; an LSTMCell
 * decomposed into Gemm/Sigmoid/Tanh is reported as "no LSTM node with input width 768"), convolutions as `Conv` with or
 * without a following `BatchNormalization` (a folded conv is loaded with identity statistics), linear layers as MatMul / Gemm
 * with a constant operand, and the graph input / output names the reference binds (xdtts_model_dir_describe). */
xdtts_status xdtts_tacotron2_load(const char *dir, int32_t device_id, xdtts_tacotron2 **out);

/* device_id of every constructor: a HIP device index, or XDTTS_DEVICE_DEFAULT = the process's default GPU -- environment variable
 * XDTTS_DEVICE (read at every handle creation), 0 without it.  The reference's constructors take no device
 * (`Tacotron2::load(path)`, `GriffinLim::new(..)`, src/lib.rs:40-58): a host that keeps their signatures passes XDTTS_DEVICE_DEFAULT
 * and is spread over the GPUs of a node by one process per GPU with XDTTS_DEVICE = its rank.
 * xdtts_default_device(): what XDTTS_DEVICE_DEFAULT resolves to now, or -1 (XDTTS_DEVICE is not a device index; xdtts_last_error). */
#define XDTTS_DEVICE_DEFAULT (-1)
int32_t xdtts_default_device(void);

/* The host half of Tacotron2::load: reads `dir` (ONNX graphs or tacotron2.xdtw, as above) into a
 * caller-held flat fp32 blob in canonical tensor order (n_floats == xdtts_tensor_total()).  Needs no
 * device; xdtts_tacotron2_load(dir) == this + xdtts_tacotron2_load_blob. */
xdtts_status xdtts_model_dir_read(const char *dir, float *blob, size_t n_floats);

/* The graph inputs / outputs of the three ONNX files of `dir`, one line per file:
 * "decoder_iter.onnx: inputs a,b,.. ; outputs x,y,..\n".  xdtts_tacotron2_load / xdtts_model_dir_read REQUIRE the names the
 * reference binds: the 11 named inputs of src/tacotron2/mod.rs:284-296, the 9 named outputs of :306-307,332-339,
 * `mel_outputs_postnet` (:349), 2 inputs / 3 outputs for the encoder (:379-385) -- XDTTS_ERR_IO otherwise.
 * buf may be NULL with cap 0 to ask for the size (*needed, terminator included). */
xdtts_status xdtts_model_dir_describe(const char *dir, char *buf, size_t cap, size_t *needed);

/* Seeded synthetic weights with the checkpoint's exact shapes. */
xdtts_status xdtts_tacotron2_load_synthetic(uint32_t seed, float rec_scale, int32_t device_id,
                                            xdtts_tacotron2 **out);

/* Weights from a caller-held flat fp32 blob in canonical tensor order (xdtts_tensor_*). */
xdtts_status xdtts_tacotron2_load_blob(const float *blob, size_t n_floats, int32_t device_id,
                                       xdtts_tacotron2 **out);

/* Writes the handle's canonical weights as `dir`/tacotron2.xdtw. */
xdtts_status xdtts_tacotron2_save(const xdtts_tacotron2 *h, const char *dir);

/* Canonical tensor table (names follow the NVIDIA checkpoint the ONNX graphs were exported
 * from, src/tacotron2/mod.rs:137-138). */
int32_t xdtts_tensor_count(void);
const char *xdtts_tensor_name(int32_t i);
int32_t xdtts_tensor_ndim(int32_t i);
int32_t xdtts_tensor_dim(int32_t i, int32_t d);
size_t xdtts_tensor_offset(int32_t i);
size_t xdtts_tensor_total(void);
/* Copies the canonical (un-packed, un-folded) tensor i out of the handle. */
xdtts_status xdtts_tacotron2_get_tensor(const xdtts_tacotron2 *h, int32_t i, float *out);

/* Tacotron2::infer(&[Unit]) -- src/tacotron2/mod.rs:398-437, after the Unit->id mapping
 * (:403-406) and find_splits (:399), which stay on the host side of the FFI (xdtts_host.h).
 * `splits` are the chunk end offsets into ids (ascending, last == n); NULL/0 means one chunk.
 * Each chunk must be <= max_chunk ids (else XDTTS_ERR_TOO_LONG, the reference's assert at :363).
 * Chunks are independent (:422-434): they are decoded together as one batch and concatenated
 * on the time axis (:430).  *mel receives 80 x (*n_frames) floats, C order. */
xdtts_status xdtts_tacotron2_infer_ids(xdtts_tacotron2 *h, const int64_t *ids, size_t n,
                                       const size_t *splits, size_t n_splits,
                                       const xdtts_infer_opts *opts, float **mel,
                                       size_t *n_frames);

/* Batched form of infer_chunk (src/tacotron2/mod.rs:361-393): B independent chunks, ids is
 * B x t_stride with lens[b] valid ids each.  fixed_steps_per_item may be NULL.  mels[b] is
 * 80 x n_frames[b]. */
xdtts_status xdtts_tacotron2_infer_batch(xdtts_tacotron2 *h, const int64_t *ids,
                                         const int32_t *lens, int32_t B, int32_t t_stride,
                                         const xdtts_infer_opts *opts,
                                         const int32_t *fixed_steps_per_item, float **mels,
                                         size_t *n_frames);

/* Parity hooks: the three graphs of the reference one at a time.
 * encoder.onnx (mod.rs:379): ids (T) -> memory (T x 512), processed_memory (T x 128). */
xdtts_status xdtts_tacotron2_encoder(xdtts_tacotron2 *h, const int64_t *ids, int32_t T,
                                     float *memory, float *processed_memory);
/* run_decoder frame loop (mod.rs:272-342) on caller-supplied encoder outputs: frames are
 * n_frames x 80 (pre-postnet, time-major), gates n_frames. Buffers sized for max_steps. */
xdtts_status xdtts_tacotron2_decoder(xdtts_tacotron2 *h, const float *memory,
                                     const float *processed_memory, int32_t T, int32_t n_valid,
                                     const xdtts_infer_opts *opts, float *frames, float *gates,
                                     size_t *n_frames);
/* ONE call of decoder_iter.onnx (mod.rs:304), teacher-forced: the per-step parity hook of SURVEY.md
 * section 8c(i).  Inputs as the reference feeds them (mod.rs:284-296): memory (T x 512), processed_memory
 * (T x 128), the mask as n_valid (mod.rs:219-220), decoder_input (80) and the seven state tensors; the
 * nine outputs (mod.rs:306-307,332-339): decoder_output (80), gate_prediction (1, the logit) and the
 * seven out_* state tensors, written over the state arguments.  `step` is the frame index (it selects the
 * prenet-dropout masks of the seeded stream; opts->item_base the chunk). */
xdtts_status xdtts_tacotron2_decoder_step(xdtts_tacotron2 *h, const float *memory, const float *processed_memory,
                                          int32_t T, int32_t n_valid, const xdtts_infer_opts *opts, uint32_t step,
                                          const float *decoder_input, float *attention_hidden,
                                          float *attention_cell, float *decoder_hidden, float *decoder_cell,
                                          float *attention_weights, float *attention_weights_cum,
                                          float *attention_context, float *decoder_output,
                                          float *gate_prediction);

/* The same hook through the frame-loop engine the caller names, for B chunks and n_steps >= 1 consecutive calls of
 * decoder_iter.onnx, each fed the previous one's outputs as the reference's loop does (mod.rs:328-341):
 *   engine 0  launch-per-stage GEMV kernels (what xdtts_tacotron2_decoder_step runs)
 *   engine 1  the persistent weight-stationary kernel that serves 1..4-chunk requests -- the BASELINE configs[1]
 *             engine (B <= 2 per launch, T <= 128).  It works from the attention WEIGHTS, so the incoming
 *             attention_context must equal attention_weights . memory, as every state the graph itself produced does
 *             (out_attention_context, mod.rs:332-339); anything else is XDTTS_ERR_BAD_ARG
 *   engine 2  the batched MFMA kernels that serve lock-step batches of >= 5 chunks (configs[2] / [3]); B <= 64
 * All arrays are [B][...] row-major: memory [B][T][512], processed_memory [B][T][128], n_valid [B], decoder_input
 * [B][80], the seven state tensors [B][1024] x4, [B][T] x2, [B][512] (updated in place to the state after the last
 * step); decoder_output [B][n_steps][80] and gate_prediction [B][n_steps] (logits; no stop rule is applied).
 * step0 is the frame index of the first call (it selects the prenet-dropout draws), opts->item_base the first chunk's
 * dropout-stream index.  n_steps > 1 from the zero state gives the engine's written-back state after a free run. */
xdtts_status xdtts_tacotron2_decoder_steps(xdtts_tacotron2 *h, int32_t engine, int32_t B, const float *memory,
                                           const float *processed_memory, int32_t T, const int32_t *n_valid,
                                           const xdtts_infer_opts *opts, uint32_t step0, int32_t n_steps,
                                           const float *decoder_input, float *attention_hidden, float *attention_cell,
                                           float *decoder_hidden, float *decoder_cell, float *attention_weights,
                                           float *attention_weights_cum, float *attention_context, float *decoder_output,
                                           float *gate_prediction);

/* Engine state of a handle: 1 = the persistent decoder / cooperative encoder is in use, 0 = the handle was
 * demoted to the launch-per-stage / single-workgroup engine after a timed-out exchange (it probes the fast
 * engine again by itself every 64 calls), -1 = not probed yet.  batched_attention (lock-step batches of 5 or more
 * chunks): 2 = attention LSTM, energies, softmax and context in one launch whose blocks exchange tagged granules,
 * 1 = the attention alone in one such launch (also: batches beyond 64 chunks), 0 = separate kernels (after a
 * timed-out exchange).  Any of the three pointers may be null.  _reset puts a demoted handle back at once. */
xdtts_status xdtts_tacotron2_engine_state(const xdtts_tacotron2 *h, int32_t *decoder_persistent,
                                          int32_t *encoder_cooperative, int32_t *batched_attention);
xdtts_status xdtts_tacotron2_engine_reset(xdtts_tacotron2 *h);
/* The engine of lock-step batches of 3..8 chunks (one persistent launch for the whole loop, the LSTMs of all chunks on the matrix
 * cores: csrc/decoder_persistent8.hip): 1 = in use, 0 = off (the device cannot host its 256-workgroup grid, an exchange timed
 * out -- such batches then run on the engines either side: pairs of the persistent decoder up to 4 chunks, the batched engine
 * from 5 -- or XDTTS_P8=0 in the environment), -1 = not probed yet.  _engine_reset puts it back to -1 unless a launch was refused. */
xdtts_status xdtts_tacotron2_small_batch_engine_state(const xdtts_tacotron2 *h, int32_t *state);

/* Identity of this build: "src_sha256=<sha256 over the library's sources in name order> arch=gfx950 built_utc=... compiler=...".
 * The .so files are not in the git history (built by `make -C xd-tts_amd`, __graft_entry__.build()); the hash lets a test tell
 * whether the library it loaded was built from the sources next to it. */
const char *xdtts_build_info(void);

/* Measurement aid, not part of the reference's surface: the latency floor of one step of the
 * persistent decoder engine on this device -- its five dependent inter-CU exchanges (x, h_att, 8 x T partial energies,
 * h_dec, mel -> x) with no arithmetic between them, best of five launches of `steps` steps.  tuned != 0: the consumers
 * delay their first polls as the engine does.  bench.py reports it as roofline.latency_floor_us. */
xdtts_status xdtts_edge_floor_us(int32_t device_id, int32_t steps, int32_t T, int32_t tuned, double *us_per_step);

/* postnet.onnx (mod.rs:345-355): frames (F x 80) -> mel_outputs_postnet (80 x F). */
xdtts_status xdtts_tacotron2_postnet(xdtts_tacotron2 *h, const float *frames, int32_t F,
                                     float *mel_out);

/* Phase timings of the last infer call on this handle, measured with HIP events on the
 * handle's stream: ms[0] encoder, ms[1] decoder loop, ms[2] postnet, ms[3] total;
 * steps = decoder steps executed. */
xdtts_status xdtts_tacotron2_last_timings(const xdtts_tacotron2 *h, float ms[4], int32_t *steps);

void xdtts_tacotron2_free(xdtts_tacotron2 *h);

/* ---- Griffin-Lim ------------------------------------------------------------------------ */

/* griffin_lim::mel::create_mel_filter_bank(sr, n_fft, n_mels, fmin, fmax: Option<f32>) --
 * src/tacotron2/mod.rs:453.  fmax = NaN means None (sr/2).  out is n_mels x (n_fft/2+1). */
xdtts_status xdtts_mel_filter_bank(float sample_rate, size_t n_fft, size_t n_mels, float fmin,
                                   float fmax_or_nan, float *out);

/* GriffinLim::new(mel_basis, noverlap, power, iter, momentum) -- src/tacotron2/mod.rs:456.
 * n_fft = 2*(n_bins-1); hop = n_fft - noverlap. */
xdtts_status xdtts_griffinlim_new(const float *mel_basis, size_t n_mels, size_t n_bins,
                                  size_t noverlap, float power, size_t iters, float momentum,
                                  int32_t device_id, xdtts_griffinlim **out);

/* The conventions of GriffinLim::infer's first step that the absent `griffin-lim` crate
 * (Cargo.lock:666-668) leaves open, as switches.  Defaults = the librosa-0.9 reading documented in
 * DESIGN.md section 2 (what every parity test and bench number uses):
 *   nnls_iters      0: S = clip(pinv(basis) . m, 0), the least-squares start of the crate's bounded NNLS
 *                   (lbfgsb 0.1.0, Cargo.lock:888-895); K > 0: K projected-gradient steps of
 *                   min 1/2 |basis x - m|^2, x >= 0 on the device after that start (fixed step
 *                   1/lambda_max(basis basis^T); the iteration's limit is the NNLS solution)
 *   power_mode      0: S = x^(1/power) (librosa mel_to_stft)   1: S = x^power   2: S = x
 *   mel_decompress  0: m = exp(mel) (Tacotron2's ln compression)   1: m = mel   2: m = 10^mel
 *   output_normalise  (G6: the last step of GriffinLim::infer before src/lib.rs:155 scales by i16::MAX)
 *                   0: audio as is   1: audio / max|audio|   2: audio * rms_target / rms(audio)
 *                   3: like 2, the scale limited to 1 / max|audio| so that no sample leaves [-1, 1]   [default 3]
 *                   (2 and 3 differ only for a crest factor above 1 / rms_target = 20 dB -- an utterance that is mostly
 *                   pause -- where 2 would hand src/lib.rs:155's saturating cast samples beyond +-1 to clip)
 *   rms_target      0.1 = -20 dBFS.  Why rms / 0.1: the only outputs of this path the reference holds --
 *                   slides/audio/goodbye.wav and capital_nonsense.wav, both exactly WAV_SPEC (src/lib.rs:25-30) -- sit
 *                   at RMS 0.099994 and 0.099995 of full scale with peaks 0.82 and 0.61: an RMS-0.1 signal after the
 *                   truncating `as i16` cast (tests/golden/reference_audio_facts.json, tools/reference_audio_facts.py)
 * and one switch of xdtts_griffinlim_infer_batch only:
 *   batch_shape     0: a workgroup owns up to 4 or up to 8 frames of an utterance, whichever shape needs less time
 *                   for the batch at hand (the overlap-add then sums in a different order than the single call:
 *                   same audio within the fp32 drift of DESIGN.md section 2, not bit for bit)
 *                   4: always the shape of the single-utterance call: every audio of a batch is bit-identical to
 *                   xdtts_griffinlim_infer on that utterance alone (the same time when its workgroups run two per CU, up to 1.6x
 *                   on a device that admits only one per CU) */
typedef struct {
  int32_t nnls_iters;
  int32_t power_mode;
  int32_t mel_decompress;
  int32_t output_normalise;
  int32_t batch_shape;
  float rms_target;
} xdtts_griffinlim_opts;
void xdtts_griffinlim_opts_default(xdtts_griffinlim_opts *opts);
xdtts_status xdtts_griffinlim_set_opts(xdtts_griffinlim *g, const xdtts_griffinlim_opts *opts);
xdtts_status xdtts_griffinlim_get_opts(const xdtts_griffinlim *g, xdtts_griffinlim_opts *opts);

/* The random initial phase of the crate is un-seeded; here it is a counter-based stream. */
xdtts_status xdtts_griffinlim_set_seed(xdtts_griffinlim *g, uint32_t seed);

/* GriffinLim::infer(&Array2<f32>) -> Vec<f32> -- src/lib.rs:141.  mel is n_mels x F (C order,
 * natural-log compressed as produced by Tacotron2); audio has hop*(F-1) samples. */
xdtts_status xdtts_griffinlim_infer(xdtts_griffinlim *g, const float *mel, size_t n_mels,
                                    size_t n_frames, float **audio, size_t *n_samples);

/* GriffinLim::infer for n_utt utterances in one call (the vocoder half of a batch; the reference calls
 * self.vocoder.infer once per utterance, src/lib.rs:141): mels[u] is n_mels x n_frames[u]; audios[u]
 * receives hop*(n_frames[u]-1) samples: the audio of xdtts_griffinlim_infer on that utterance alone (bit for bit
 * with opts.batch_shape = 4; within fp32 drift with the default, see xdtts_griffinlim_opts), whatever the mix
 * and the order.  Utterances share persistent launches (a workgroup never spans two). */
xdtts_status xdtts_griffinlim_infer_batch(xdtts_griffinlim *g, const float *const *mels, size_t n_mels,
                                          const size_t *n_frames, int32_t n_utt, float **audios,
                                          size_t *n_samples);

/* The loop alone (G2..G5 and the final ISTFT): no mel->linear inversion before it and no output normalisation
 * after it -- the bench / parity entry of SURVEY.md section 8(b).  S is n_bins x F linear magnitude; phase0 is
 * n_bins x F x 2 (cos, sin) or NULL for the seeded stream; iters = 0 uses the handle's count. */
xdtts_status xdtts_griffinlim_infer_linear(xdtts_griffinlim *g, const float *S,
                                           const float *phase0, size_t n_frames, size_t iters,
                                           float **audio, size_t *n_samples);

/* mel -> linear magnitude only (step 1 of GriffinLim::infer): S_out is n_bins x F. */
xdtts_status xdtts_griffinlim_mel_to_linear(xdtts_griffinlim *g, const float *mel,
                                            size_t n_mels, size_t n_frames, float *S_out);

/* Parity hook (teacher-forced): n_iter iterations of the loop inside
 * GriffinLim::infer (src/lib.rs:141) from a caller-held state, no final ISTFT.  S is n_bins x F;
 * angles (unit-modulus phase estimate) and rebuilt (the previous iteration's STFT; zeros before
 * the first) are n_bins x F x 2 (re, im) and are updated in place. */
xdtts_status xdtts_griffinlim_step(xdtts_griffinlim *g, const float *S, float *angles,
                                   float *rebuilt, size_t n_frames, size_t n_iter);

/* ms[0] mel->linear, ms[1] iterations, ms[2] total of the last call (HIP events). */
xdtts_status xdtts_griffinlim_last_timings(const xdtts_griffinlim *g, float ms[3]);

void xdtts_griffinlim_free(xdtts_griffinlim *g);

/* ---- XdTts::infer pipeline (src/lib.rs:110-159): mel-gen then vocoder with the mel kept in
 * HBM between the two; both outputs are returned. ------------------------------------------ */
xdtts_status xdtts_synthesize_ids(xdtts_tacotron2 *h, xdtts_griffinlim *g, const int64_t *ids,
                                  size_t n, const size_t *splits, size_t n_splits,
                                  const xdtts_infer_opts *opts, float **mel, size_t *n_frames,
                                  float **audio, size_t *n_samples);

/* XdTts::infer (src/lib.rs:110-159) for n_utt utterances in one call -- the "batched / parallel
 * sentences" the author notes at src/phonemes.rs:677-680, BASELINE.json configs[3].  The chunks of
 * all utterances are given as in xdtts_tacotron2_infer_batch (ids [B][t_stride], lens[B], optional
 * per-chunk fixed step counts); utt_chunks[u] = number of CONSECUTIVE chunks that form utterance u
 * (they sum to B; find_splits + the trailing split, src/tacotron2/mod.rs:399,412-414).  All chunks
 * decode in one lock-step batch, each utterance's chunk mels are concatenated on the time axis
 * (mod.rs:430) where the post-net writes them, and the vocoder batch reads that mel in HBM.
 * Results equal xdtts_tacotron2_infer_batch followed by xdtts_griffinlim_infer_batch bit for bit.
 * n_frames[u] / mels[u] (80 x n_frames[u]; mels may be NULL) and n_samples[u] / audios[u] are
 * per utterance; buffers are released with xdtts_free. */
xdtts_status xdtts_synthesize_batch(xdtts_tacotron2 *h, xdtts_griffinlim *g, const int64_t *ids,
                                    const int32_t *lens, int32_t B, int32_t t_stride,
                                    const int32_t *utt_chunks, int32_t n_utt,
                                    const xdtts_infer_opts *opts,
                                    const int32_t *fixed_steps_per_item, float **mels,
                                    size_t *n_frames, float **audios, size_t *n_samples);

/* XdTts::infer (src/lib.rs:110-159) for a SEQUENCE of n_utt utterances, each decoded alone (batch 1) exactly as
 * xdtts_synthesize_ids does it -- what a loop over src/lib.rs:122-141 does -- with the vocoder of utterance u overlapped with the
 * encoder of utterance u + 1 (the frame loop in between owns the whole GPU; it is ordered behind the previous vocoder).
 * ids[u] / n_ids[u] / splits[u] / n_splits[u] as in xdtts_synthesize_ids (splits may be NULL: one chunk per utterance unless it
 * exceeds the window); mels may be NULL.  Outputs per utterance, released with xdtts_free.  Same bits as n_utt calls of
 * xdtts_synthesize_ids; on an error nothing is returned.  xdtts_tacotron2_last_timings / xdtts_griffinlim_last_timings then
 * report the SUMS over the sequence (HIP events per utterance). */
xdtts_status xdtts_synthesize_sequence(xdtts_tacotron2 *h, xdtts_griffinlim *g, const int64_t *const *ids,
                                       const size_t *n_ids, const size_t *const *splits, const size_t *n_splits,
                                       int32_t n_utt, const xdtts_infer_opts *opts, float **mels,
                                       size_t *n_frames, float **audios, size_t *n_samples);

/* ---- host-side front of Tacotron2::infer (stays on the CPU side of the FFI) ---------------- */
/* generate_id_list -- src/tacotron2/mod.rs:90-122: 148 symbols; token text of an id. */
int32_t xdtts_symbol_count(void);
const char *xdtts_symbol_token(int32_t id);
/* Unit::from_str (src/phonemes.rs:450-487) + best_match_for_unit (src/phonemes.rs:627-660):
 * id of a unit token ("IH0", " ", ".", "a"), or -1 where the reference drops the unit
 * (src/tacotron2/mod.rs:403-406).  as_character != 0 looks the token up as Unit::Character. */
int64_t xdtts_unit_id(const char *token, int32_t as_character);
/* split_score -- src/phonemes.rs:663-671. */
int32_t xdtts_split_score(int64_t id);
/* find_splits(units, max_size) -- src/phonemes.rs:681-753, on the id sequence. */
xdtts_status xdtts_find_splits(const int64_t *ids, size_t n, size_t max_size, size_t *out,
                               size_t cap, size_t *n_out);

/* ---- output stage (host) -- src/lib.rs:25-30,125-176 -------------------------------------- */
#define XDTTS_SAMPLE_RATE 22050 /* WAV_SPEC, src/lib.rs:25-30: mono, 22050 Hz, 16-bit signed PCM */
/* `(*sample * i16::MAX as f32) as i16` -- src/lib.rs:153-155.  Rust's float -> int `as` truncates
 * toward zero, saturates at the i16 range and maps NaN to 0. */
xdtts_status xdtts_audio_to_i16(const float *audio, size_t n, int16_t *pcm);
/* write_silence -- src/lib.rs:162-176: number of zero samples of an SSML break,
 * round(sample_rate * seconds). */
size_t xdtts_silence_samples(double seconds, uint32_t sample_rate);
/* The same from the two integer fields of a Rust `Duration` (what write_silence receives): the
 * conversion Duration::as_secs_f32 and the product are evaluated in f32 exactly as src/lib.rs:166
 * does, so the sample count is bit-identical to the reference's at every .5 boundary. */
size_t xdtts_silence_samples_duration(uint64_t secs, uint32_t nanos, uint32_t sample_rate);
/* A complete RIFF/WAVE file with the reference's WAV_SPEC (what hound's WavWriter produces for the
 * i16 writer of src/lib.rs:152-157): 44-byte PCM header + little-endian samples. */
xdtts_status xdtts_wav_write(const char *path, const int16_t *pcm, size_t n, uint32_t sample_rate);
/* ndarray_npy::write_npy(&path, &spectrogram) -- src/lib.rs:128-141: .npy version 1.0, '<f4',
 * C order, shape (rows, cols). */
xdtts_status xdtts_npy_write_f32(const char *path, const float *data, size_t rows, size_t cols);
/* Real time factor as logged at src/lib.rs:145-151: compute seconds / (samples / 22050). */
double xdtts_real_time_factor(double compute_seconds, size_t n_samples);

/* ---- misc ------------------------------------------------------------------------------- */
void xdtts_free(void *p);
const char *xdtts_last_error(void);
int32_t xdtts_device_count(void);
/* Blocks until all work queued on the handle's stream has finished. */
xdtts_status xdtts_tacotron2_sync(xdtts_tacotron2 *h);

#ifdef __cplusplus
}
#endif
#endif /* XDTTS_H */
