/*
 * xdtts_oracle.h -- CPU restatement ("oracle") of the xd-tts hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (xd-tts_amd/, include/) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (/root/reference, xd009642/xd-tts) holds no golden mel or
 * audio vectors for this path and its arithmetic lives in artefacts that are absent from the
 * checkout (ONNX graphs are git-LFS pointers; onnxruntime 1.17.0 and the `griffin-lim` crate
 * 0.2.0 @e6415314 are un-vendored; no Rust toolchain).  What IS pinned against the reference:
 * the id known-answer vectors and symbol table (src/tacotron2/mod.rs:90-122,465-508), the
 * state/loop/stop structure (src/tacotron2/mod.rs:177-233,272-358), the pad/plen/mask quirk
 * (src/tacotron2/mod.rs:361-393), the chunker KAT (src/phonemes.rs:681-753,785), the vocoder
 * parameters (src/tacotron2/mod.rs:441-458).  Layer math restates the published algorithms
 * (NVIDIA Tacotron2 model.py; librosa 0.9 griffinlim/stft/istft/filters.mel/mel_to_stft) and
 * is cross-checked in tests/ against torch CPU (LSTMCell, conv1d, batch_norm, stft/istft) and
 * scipy (L-BFGS-B NNLS) as independent implementations.
 *
 * Build: `make -C oracle` -> liboracle_f32.so (REAL=float) and liboracle_f64.so (REAL=double;
 * weights stay fp32, all activations/accumulators in double: the drift reference).
 */
#ifndef XDTTS_ORACLE_H
#define XDTTS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#ifdef __cplusplus
extern "C" {
#endif

/* ---- model constants: src/tacotron2/mod.rs:205-208 + NVIDIA defaults ---- */
enum {
  ORC_N_SYMBOLS = 148, /* mod.rs:90-122 */
  ORC_EMB = 512,       /* encoder_embedding_dim, mod.rs:207 */
  ORC_ENC_CONVS = 3,
  ORC_ENC_K = 5,
  ORC_ENC_H = 256, /* per direction */
  ORC_N_MEL = 80,  /* mod.rs:208 */
  ORC_PRENET = 256,
  ORC_ATT_RNN = 1024, /* mod.rs:205 */
  ORC_DEC_RNN = 1024, /* mod.rs:206 */
  ORC_ATT_DIM = 128,
  ORC_LOC_F = 32,
  ORC_LOC_K = 31,
  ORC_POST_CONVS = 5,
  ORC_POST_CH = 512,
  ORC_POST_K = 5,
  ORC_T_MAX = 512,
  ORC_N_TENSORS = 76
};

/* ---- counter-based RNG shared (by specification, not by code) with the HIP library ---- */
uint32_t orc_rng_u32(uint32_t seed, uint32_t stream, uint32_t idx);
float orc_rng_uniform(uint32_t seed, uint32_t stream, uint32_t idx); /* [0,1), 24-bit */

/* ---- weight table ---- */
int orc_num_tensors(void);
const char *orc_tensor_name(int i);
int orc_tensor_ndim(int i);
int orc_tensor_dim(int i, int d);
size_t orc_tensor_numel(int i);
size_t orc_tensor_offset(int i); /* in floats into the flat blob */
size_t orc_total_floats(void);
int orc_tensor_index(const char *name);
/* U(-k,k), k=1/sqrt(fan_in); tensor i element j uses rng(seed, stream=i, idx=j).
 * rec_scale multiplies attention_rnn/decoder_rnn weight_hh (1.0 = PyTorch-default init). */
void orc_weights_synthetic(uint32_t seed, float rec_scale, float *blob);

/* ---- Tacotron2 ---- */
typedef struct {
  real att_h[ORC_ATT_RNN], att_c[ORC_ATT_RNN];
  real dec_h[ORC_DEC_RNN], dec_c[ORC_DEC_RNN];
  real aw[ORC_T_MAX], awc[ORC_T_MAX];
  real ctx[ORC_EMB];
  real dec_in[ORC_N_MEL];
} orc_decoder_state;

typedef struct {
  float gate_threshold; /* 0.6  mod.rs:279 */
  int32_t max_steps;    /* 1000 mod.rs:280 */
  int32_t fixed_steps;  /* 0 = use the gate; >0 = emit exactly this many frames */
  int32_t dropout_mode; /* 0 off, 1 seeded, 2 explicit keep masks (below) */
  uint32_t dropout_seed;
  uint32_t item; /* utterance/chunk index mixed into the dropout stream */
  const uint8_t *masks; /* dropout_mode 2: [mask_steps][2 layers][256] keep bytes of THIS chunk (non-zero = keep, x2) */
  int32_t mask_steps;
} orc_decoder_opts;

void orc_decoder_opts_default(orc_decoder_opts *o);
void orc_decoder_state_init(orc_decoder_state *s); /* DecoderState::new, mod.rs:202-233 */

/* encoder graph (mod.rs:379): ids (T) -> memory (T x 512), processed_memory (T x 128). */
void orc_encoder(const float *blob, const int64_t *ids, int T, real *memory, real *pmem);

/* prenet dropout keep-mask bit for (layer 0/1, unit j) at decoder step `step`. */
int orc_dropout_keep(uint32_t seed, uint32_t item, uint32_t step, int layer, int j);

/* one decoder_iter call (mod.rs:304): updates s in place, writes mel (80) and gate logit. */
void orc_decoder_step(const float *blob, const real *memory, const real *pmem, int T, int n_valid,
                      orc_decoder_state *s, const orc_decoder_opts *o, uint32_t step, real *mel,
                      real *gate);

/* run_decoder frame loop (mod.rs:302-342): frames[F][80]; returns F. gates[F] optional. */
int orc_run_decoder(const float *blob, const real *memory, const real *pmem, int T, int n_valid,
                    const orc_decoder_opts *o, real *frames, real *gates);

/* postnet (mod.rs:345-355): frames[F][80] -> out[80][F] (residual added, final layout). */
void orc_postnet(const float *blob, const real *frames, int F, real *out);

/* infer_chunk (mod.rs:361-393): pads ids to `window` with 0, plen = window, mask from n.
 * out must hold 80*max_steps reals; returns F. */
int orc_infer_chunk(const float *blob, const int64_t *ids, int n, int window,
                    const orc_decoder_opts *o, real *out_80xF);

real orc_sigmoid(real x); /* mod.rs:126-133 */

/* ---- Griffin-Lim (crate griffin-lim 0.2.0, restated from librosa 0.9) ---- */
void orc_mel_filter_bank(double sr, int n_fft, int n_mels, double fmin, double fmax,
                         float *out /* n_mels x (n_fft/2+1) */);
/* pinv of a full-row-rank (n_mels x n_bins) basis: out n_bins x n_mels. returns 0 on success */
int orc_pinv(const float *basis, int n_mels, int n_bins, float *out);
/* S[n_bins][F] = max(pinv @ exp(mel), 0)^(1/power) */
void orc_mel_to_linear(const float *pinv, int n_mels, int n_bins, const real *mel, int F,
                       real power, real *S);
/* Step 1 with the convention switches of xdtts_griffinlim_opts (see the .c file); the defaults
 * (nnls_iters 0, power_mode 0, decompress 0) give orc_mel_to_linear. */
double orc_nnls_lipschitz(const float *basis, int n_mels, int n_bins);
void orc_mel_to_linear_opts(const float *pinv, const float *basis, int n_mels, int n_bins, const real *mel, int F,
                            real power, int nnls_iters, int power_mode, int decompress, real *S);
/* phase0[bin][frame][2] = (cos, sin)(2*pi*u), u = rng(seed, 0x47, frame*n_bins+bin) */
void orc_phase_init(uint32_t seed, int n_bins, int F, real *phase0);
void orc_stft(const real *y, int n, int n_fft, int hop, real *out /* bins x F x 2 */, int F);
void orc_istft(const real *spec /* bins x F x 2 */, int F, int n_fft, int hop,
               real *y /* hop*(F-1) */);
/* audio[hop*(F-1)]; S[bins][F]; phase0 may be NULL -> orc_phase_init(seed). */
void orc_griffinlim(const real *S, const real *phase0, uint32_t seed, int F, int n_fft, int hop,
                    int iters, real momentum, real *audio);

/* G6 output normalisation: mode 0 none, 1 peak, 2 rms(target) -- see the definition for the evidence. */
void orc_output_normalise(real *y, size_t n, int mode, double target);

/* `iters` iterations of the loop inside orc_griffinlim on a caller-held state (teacher-forced
 * parity hook): ang and reb are [bins][F][2], updated in place; no final ISTFT. */
void orc_griffinlim_step(const real *S, real *ang, real *reb, int F, int n_fft, int hop, int iters,
                         real momentum);

#ifdef __cplusplus
}
#endif
#endif
