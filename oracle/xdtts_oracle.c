/*
 * xdtts_oracle.c -- CPU restatement of the xd-tts mel-synthesis + vocoding hot path.
 * TEST INFRASTRUCTURE ONLY -- see xdtts_oracle.h (parity unpinned; what is pinned and how).
 *
 * Every function cites the reference file:line (under /root/reference) it follows, or, where
 * the arithmetic lives in an absent third-party artefact, the published algorithm it restates.
 * Plain scalar C, single thread, written from the algorithm descriptions -- not derived from
 * any reference source text.
 */
#include "xdtts_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------ */
/* RNG: stateless 32-bit hash ("lowbias32" finaliser applied three times).  The HIP library    */
/* implements the same specification independently (xd-tts_amd/csrc/rng.h).                    */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
uint32_t orc_rng_u32(uint32_t seed, uint32_t stream, uint32_t idx) {
  return mix32(mix32(mix32(seed ^ 0x9E3779B9U) + stream) + idx);
}
float orc_rng_uniform(uint32_t seed, uint32_t stream, uint32_t idx) {
  return (float)(orc_rng_u32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------------------------------ */
/* Weight table.  Names/shapes: NVIDIA Tacotron2 (the checkpoint the reference's ONNX graphs   */
/* were exported from, src/tacotron2/mod.rs:137-138); parameter counts corroborated by the LFS */
/* object sizes (SURVEY.md section 8 header).                                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  char name[64];
  int ndim;
  int dims[3];
  size_t numel, offset;
  float k;    /* init bound */
  int kind;   /* 0 uniform(-k,k); 1 bn.weight; 2 bn.bias; 3 bn.mean; 4 bn.var */
  int rec;    /* 1 = recurrent matrix scaled by rec_scale */
} tensor_t;

static tensor_t g_tab[ORC_N_TENSORS];
static int g_ntab = 0;
static size_t g_total = 0;

static void tab_add(const char *name, int ndim, int d0, int d1, int d2, double fan_in, int kind,
                    int rec) {
  tensor_t *t = &g_tab[g_ntab++];
  memset(t, 0, sizeof(*t));
  strncpy(t->name, name, sizeof(t->name) - 1);
  t->ndim = ndim;
  t->dims[0] = d0;
  t->dims[1] = d1;
  t->dims[2] = d2;
  t->numel = (size_t)d0 * (ndim > 1 ? d1 : 1) * (ndim > 2 ? d2 : 1);
  t->offset = g_total;
  t->k = (float)(1.0 / sqrt(fan_in));
  t->kind = kind;
  t->rec = rec;
  g_total += t->numel;
}

static void tab_conv_bn(const char *prefix, int co, int ci, int k) {
  char nm[64];
  double fan = (double)ci * k;
  snprintf(nm, sizeof nm, "%s.conv.weight", prefix);
  tab_add(nm, 3, co, ci, k, fan, 0, 0);
  snprintf(nm, sizeof nm, "%s.conv.bias", prefix);
  tab_add(nm, 1, co, 1, 1, fan, 0, 0);
  snprintf(nm, sizeof nm, "%s.bn.weight", prefix);
  tab_add(nm, 1, co, 1, 1, 1, 1, 0);
  snprintf(nm, sizeof nm, "%s.bn.bias", prefix);
  tab_add(nm, 1, co, 1, 1, 1, 2, 0);
  snprintf(nm, sizeof nm, "%s.bn.running_mean", prefix);
  tab_add(nm, 1, co, 1, 1, 1, 3, 0);
  snprintf(nm, sizeof nm, "%s.bn.running_var", prefix);
  tab_add(nm, 1, co, 1, 1, 1, 4, 0);
}

static void tab_lstm(const char *prefix, int hidden, int in, int rec) {
  char nm[64];
  snprintf(nm, sizeof nm, "%s.weight_ih", prefix);
  tab_add(nm, 2, 4 * hidden, in, 1, hidden, 0, 0);
  snprintf(nm, sizeof nm, "%s.weight_hh", prefix);
  tab_add(nm, 2, 4 * hidden, hidden, 1, hidden, 0, rec);
  snprintf(nm, sizeof nm, "%s.bias_ih", prefix);
  tab_add(nm, 1, 4 * hidden, 1, 1, hidden, 0, 0);
  snprintf(nm, sizeof nm, "%s.bias_hh", prefix);
  tab_add(nm, 1, 4 * hidden, 1, 1, hidden, 0, 0);
}

static void tab_init(void) {
  if (g_ntab) return;
  char nm[64];
  /* embedding: NVIDIA init bound sqrt(3)*sqrt(2/(n_symbols+emb)) expressed as 1/sqrt(fan) */
  tab_add("embedding.weight", 2, ORC_N_SYMBOLS, ORC_EMB, 1,
          (double)(ORC_N_SYMBOLS + ORC_EMB) / 6.0, 0, 0);
  for (int i = 0; i < ORC_ENC_CONVS; ++i) {
    snprintf(nm, sizeof nm, "encoder.convolutions.%d", i);
    tab_conv_bn(nm, ORC_EMB, ORC_EMB, ORC_ENC_K);
  }
  tab_lstm("encoder.lstm.fwd", ORC_ENC_H, ORC_EMB, 0);
  tab_lstm("encoder.lstm.bwd", ORC_ENC_H, ORC_EMB, 0);
  tab_add("attention.memory_layer.weight", 2, ORC_ATT_DIM, ORC_EMB, 1, ORC_EMB, 0, 0);
  tab_add("prenet.0.weight", 2, ORC_PRENET, ORC_N_MEL, 1, ORC_N_MEL, 0, 0);
  tab_add("prenet.1.weight", 2, ORC_PRENET, ORC_PRENET, 1, ORC_PRENET, 0, 0);
  tab_lstm("attention_rnn", ORC_ATT_RNN, ORC_PRENET + ORC_EMB, 1);
  tab_add("attention.query_layer.weight", 2, ORC_ATT_DIM, ORC_ATT_RNN, 1, ORC_ATT_RNN, 0, 0);
  tab_add("attention.v.weight", 1, ORC_ATT_DIM, 1, 1, ORC_ATT_DIM, 0, 0);
  tab_add("attention.location_conv.weight", 3, ORC_LOC_F, 2, ORC_LOC_K, 2.0 * ORC_LOC_K, 0, 0);
  tab_add("attention.location_dense.weight", 2, ORC_ATT_DIM, ORC_LOC_F, 1, ORC_LOC_F, 0, 0);
  tab_lstm("decoder_rnn", ORC_DEC_RNN, ORC_ATT_RNN + ORC_EMB, 1);
  tab_add("linear_projection.weight", 2, ORC_N_MEL, ORC_DEC_RNN + ORC_EMB, 1,
          ORC_DEC_RNN + ORC_EMB, 0, 0);
  tab_add("linear_projection.bias", 1, ORC_N_MEL, 1, 1, ORC_DEC_RNN + ORC_EMB, 0, 0);
  tab_add("gate_layer.weight", 1, ORC_DEC_RNN + ORC_EMB, 1, 1, ORC_DEC_RNN + ORC_EMB, 0, 0);
  tab_add("gate_layer.bias", 1, 1, 1, 1, ORC_DEC_RNN + ORC_EMB, 0, 0);
  for (int i = 0; i < ORC_POST_CONVS; ++i) {
    int ci = i == 0 ? ORC_N_MEL : ORC_POST_CH;
    int co = i == ORC_POST_CONVS - 1 ? ORC_N_MEL : ORC_POST_CH;
    snprintf(nm, sizeof nm, "postnet.convolutions.%d", i);
    tab_conv_bn(nm, co, ci, ORC_POST_K);
  }
}

int orc_num_tensors(void) {
  tab_init();
  return g_ntab;
}
const char *orc_tensor_name(int i) {
  tab_init();
  return g_tab[i].name;
}
int orc_tensor_ndim(int i) {
  tab_init();
  return g_tab[i].ndim;
}
int orc_tensor_dim(int i, int d) {
  tab_init();
  return g_tab[i].dims[d];
}
size_t orc_tensor_numel(int i) {
  tab_init();
  return g_tab[i].numel;
}
size_t orc_tensor_offset(int i) {
  tab_init();
  return g_tab[i].offset;
}
size_t orc_total_floats(void) {
  tab_init();
  return g_total;
}
int orc_tensor_index(const char *name) {
  tab_init();
  for (int i = 0; i < g_ntab; ++i)
    if (!strcmp(g_tab[i].name, name)) return i;
  return -1;
}

void orc_weights_synthetic(uint32_t seed, float rec_scale, float *blob) {
  tab_init();
  for (int i = 0; i < g_ntab; ++i) {
    const tensor_t *t = &g_tab[i];
    float *w = blob + t->offset;
    for (size_t j = 0; j < t->numel; ++j) {
      float u = orc_rng_uniform(seed, (uint32_t)i, (uint32_t)j);
      float s = 2.0f * u - 1.0f;
      switch (t->kind) {
        case 0: w[j] = t->k * s; break;
        case 1: w[j] = fmaf(0.1f, s, 1.0f); break; /* gamma (fmaf: one rounding, same on every target) */
        case 2: w[j] = 0.1f * s; break;        /* beta */
        case 3: w[j] = 0.1f * s; break;        /* running_mean */
        default: w[j] = fmaf(0.2f, u, 1.0f); break; /* running_var */
      }
      if (t->rec) w[j] *= rec_scale;
    }
  }
}

static const float *W(const float *blob, const char *name) {
  int i = orc_tensor_index(name);
  if (i < 0) abort();
  return blob + g_tab[i].offset;
}

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers                                                                          */
/* ------------------------------------------------------------------------------------------ */
static real dotw(const float *w, const real *x, int n) {
  real a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int i = 0;
  for (; i + 8 <= n; i += 8)
    for (int k = 0; k < 8; ++k) a[k] += (real)w[i + k] * x[i + k];
  real s = ((a[0] + a[4]) + (a[1] + a[5])) + ((a[2] + a[6]) + (a[3] + a[7]));
  for (; i < n; ++i) s += (real)w[i] * x[i];
  return s;
}

static inline real sigm(real x) { return (real)1 / ((real)1 + (real)exp(-(double)x)); }

/* src/tacotron2/mod.rs:126-133 -- the host-side gate sigmoid, numerically-stable two-branch. */
real orc_sigmoid(real x) {
  if (x >= 0) {
    real e = (real)exp(-(double)x);
    return (real)1 / ((real)1 + e);
  } else {
    real e = (real)exp((double)x);
    return e / ((real)1 + e);
  }
}

/* LSTM cell, PyTorch gate order i,f,g,o:
 * g = W_ih x + b_ih + W_hh h + b_hh ; c' = s(f) c + s(i) tanh(g) ; h' = s(o) tanh(c').      */
static void lstm_cell(const float *wih, const float *whh, const float *bih, const float *bhh,
                      const real *x, int nin, real *h, real *c, int hid) {
  real *g = (real *)malloc(sizeof(real) * 4 * hid);
  /* The OpenMP pragmas in this file only split loops whose iterations are independent, so the
   * all-cores build (liboracle_f32_omp.so, bench.py's cpu_baseline) computes bit-identical results;
   * the checker builds are compiled without -fopenmp and ignore them. */
#pragma omp parallel for schedule(static)
  for (int r = 0; r < 4 * hid; ++r)
    g[r] = dotw(wih + (size_t)r * nin, x, nin) + (real)bih[r] +
           (dotw(whh + (size_t)r * hid, h, hid) + (real)bhh[r]);
  for (int j = 0; j < hid; ++j) {
    real ig = sigm(g[j]), fg = sigm(g[hid + j]);
    real gg = (real)tanh((double)g[2 * hid + j]), og = sigm(g[3 * hid + j]);
    real cn = fg * c[j] + ig * gg;
    c[j] = cn;
    h[j] = og * (real)tanh((double)cn);
  }
  free(g);
}

/* conv1d (+eval BatchNorm) over time-major x[T][ci] -> y[T][co]; weight PyTorch layout
 * [co][ci][k]; zero padding (k-1)/2.  act: 0 none, 1 relu, 2 tanh. bn may be NULL.           */
static void conv1d_bn(const float *w, const float *b, const float *bn_w, const float *bn_b,
                      const float *bn_m, const float *bn_v, const real *x, int T, int ci, int co,
                      int k, int act, real *y) {
  int pad = (k - 1) / 2;
  float *wt = (float *)malloc(sizeof(float) * (size_t)co * k * ci); /* [co][k][ci] */
  for (int o = 0; o < co; ++o)
    for (int c = 0; c < ci; ++c)
      for (int j = 0; j < k; ++j) wt[((size_t)o * k + j) * ci + c] = w[((size_t)o * ci + c) * k + j];
#pragma omp parallel for schedule(static)
  for (int t = 0; t < T; ++t)
    for (int o = 0; o < co; ++o) {
      real s = 0;
      for (int j = 0; j < k; ++j) {
        int tt = t + j - pad;
        if (tt < 0 || tt >= T) continue;
        s += dotw(wt + ((size_t)o * k + j) * ci, x + (size_t)tt * ci, ci);
      }
      s += (real)(b ? b[o] : 0.0f);
      if (bn_w) {
        real inv = (real)1 / (real)sqrt((double)bn_v[o] + 1e-5);
        s = (s - (real)bn_m[o]) * inv * (real)bn_w[o] + (real)bn_b[o];
      }
      if (act == 1) s = s > 0 ? s : 0;
      if (act == 2) s = (real)tanh((double)s);
      y[(size_t)t * co + o] = s;
    }
  free(wt);
}

/* ------------------------------------------------------------------------------------------ */
/* Encoder graph: src/tacotron2/mod.rs:379 (encoder.onnx).  NVIDIA Encoder.inference +        */
/* attention memory_layer: embedding -> 3x[conv k5 + BN + ReLU] -> BiLSTM -> memory;          */
/* processed_memory = memory_layer(memory).  plen == T (the PADDED length, mod.rs:375), so    */
/* every position including pad ids is a real time step.                                       */
/* ------------------------------------------------------------------------------------------ */
void orc_encoder(const float *blob, const int64_t *ids, int T, real *memory, real *pmem) {
  tab_init();
  const float *emb = W(blob, "embedding.weight");
  real *x = (real *)malloc(sizeof(real) * (size_t)T * ORC_EMB);
  real *y = (real *)malloc(sizeof(real) * (size_t)T * ORC_EMB);
  for (int t = 0; t < T; ++t)
    for (int c = 0; c < ORC_EMB; ++c) x[(size_t)t * ORC_EMB + c] = (real)emb[(size_t)ids[t] * ORC_EMB + c];
  char nm[96];
  for (int i = 0; i < ORC_ENC_CONVS; ++i) {
    const float *p[6];
    static const char *suf[6] = {"conv.weight", "conv.bias",       "bn.weight",
                                 "bn.bias",     "bn.running_mean", "bn.running_var"};
    for (int q = 0; q < 6; ++q) {
      snprintf(nm, sizeof nm, "encoder.convolutions.%d.%s", i, suf[q]);
      p[q] = W(blob, nm);
    }
    conv1d_bn(p[0], p[1], p[2], p[3], p[4], p[5], x, T, ORC_EMB, ORC_EMB, ORC_ENC_K, 1, y);
    real *tmp = x;
    x = y;
    y = tmp;
  }
  /* BiLSTM: out[t] = [h_fwd(t) ; h_bwd(t)] */
  for (int dir = 0; dir < 2; ++dir) {
    const char *pf = dir ? "encoder.lstm.bwd" : "encoder.lstm.fwd";
    snprintf(nm, sizeof nm, "%s.weight_ih", pf);
    const float *wih = W(blob, nm);
    snprintf(nm, sizeof nm, "%s.weight_hh", pf);
    const float *whh = W(blob, nm);
    snprintf(nm, sizeof nm, "%s.bias_ih", pf);
    const float *bih = W(blob, nm);
    snprintf(nm, sizeof nm, "%s.bias_hh", pf);
    const float *bhh = W(blob, nm);
    real h[ORC_ENC_H], c[ORC_ENC_H];
    memset(h, 0, sizeof h);
    memset(c, 0, sizeof c);
    for (int s = 0; s < T; ++s) {
      int t = dir ? T - 1 - s : s;
      lstm_cell(wih, whh, bih, bhh, x + (size_t)t * ORC_EMB, ORC_EMB, h, c, ORC_ENC_H);
      memcpy(memory + (size_t)t * ORC_EMB + dir * ORC_ENC_H, h, sizeof(real) * ORC_ENC_H);
    }
  }
  const float *wm = W(blob, "attention.memory_layer.weight");
  for (int t = 0; t < T; ++t)
    for (int a = 0; a < ORC_ATT_DIM; ++a)
      pmem[(size_t)t * ORC_ATT_DIM + a] =
          dotw(wm + (size_t)a * ORC_EMB, memory + (size_t)t * ORC_EMB, ORC_EMB);
  free(x);
  free(y);
}

/* ------------------------------------------------------------------------------------------ */
/* Decoder                                                                                     */
/* ------------------------------------------------------------------------------------------ */
void orc_decoder_opts_default(orc_decoder_opts *o) {
  o->gate_threshold = 0.6f; /* src/tacotron2/mod.rs:279 */
  o->max_steps = 1000;      /* src/tacotron2/mod.rs:280 */
  o->fixed_steps = 0;
  o->dropout_mode = 1; /* the exported graph keeps the prenet dropout on at inference */
  o->dropout_seed = 0;
  o->item = 0;
  o->masks = NULL;
  o->mask_steps = 0;
}

/* DecoderState::new, src/tacotron2/mod.rs:202-233: everything zero. */
void orc_decoder_state_init(orc_decoder_state *s) { memset(s, 0, sizeof(*s)); }

int orc_dropout_keep(uint32_t seed, uint32_t item, uint32_t step, int layer, int j) {
  return (orc_rng_u32(seed, 0x1000u + (uint32_t)layer + 2u * item, step * 256u + (uint32_t)j) >> 31) == 0;
}

/* dropout_mode 1: the seeded counter stream; 2: the caller's keep bytes [step][layer][unit] -- what the exported
 * graph's RandomUniform would have drawn in one recorded run of decoder_iter.onnx (hard part i) */
static int prenet_keep(const orc_decoder_opts *o, uint32_t step, int layer, int j) {
  if (o->dropout_mode == 2)
    return o->masks && (int64_t)step < o->mask_steps ? o->masks[((size_t)step * 2 + (size_t)layer) * ORC_PRENET + (size_t)j] != 0 : 1;
  return orc_dropout_keep(o->dropout_seed, o->item, step, layer, j);
}

/* One decoder_iter.onnx call, src/tacotron2/mod.rs:304 with I/O names at :285-295,:306-307,
 * :332-339.  Math: NVIDIA Decoder.decode.                              */
void orc_decoder_step(const float *blob, const real *memory, const real *pmem, int T, int n_valid,
                      orc_decoder_state *s, const orc_decoder_opts *o, uint32_t step, real *mel,
                      real *gate) {
  tab_init();
  /* D1 prenet: relu(W x) * mask * 2, twice; no bias; dropout always on (p=0.5) */
  const float *p0 = W(blob, "prenet.0.weight"), *p1 = W(blob, "prenet.1.weight");
  real x1[ORC_PRENET], cell_in[ORC_PRENET + ORC_EMB];
  for (int j = 0; j < ORC_PRENET; ++j) {
    real v = dotw(p0 + (size_t)j * ORC_N_MEL, s->dec_in, ORC_N_MEL);
    v = v > 0 ? v : 0;
    if (o->dropout_mode) v = prenet_keep(o, step, 0, j) ? v * 2 : 0;
    x1[j] = v;
  }
  for (int j = 0; j < ORC_PRENET; ++j) {
    real v = dotw(p1 + (size_t)j * ORC_PRENET, x1, ORC_PRENET);
    v = v > 0 ? v : 0;
    if (o->dropout_mode) v = prenet_keep(o, step, 1, j) ? v * 2 : 0;
    cell_in[j] = v;
  }
  /* D2 attention LSTM on [prenet ; previous context] */
  memcpy(cell_in + ORC_PRENET, s->ctx, sizeof(real) * ORC_EMB);
  lstm_cell(W(blob, "attention_rnn.weight_ih"), W(blob, "attention_rnn.weight_hh"),
            W(blob, "attention_rnn.bias_ih"), W(blob, "attention_rnn.bias_hh"), cell_in,
            ORC_PRENET + ORC_EMB, s->att_h, s->att_c, ORC_ATT_RNN);
  /* D3 location-sensitive attention */
  const float *wq = W(blob, "attention.query_layer.weight");
  const float *wv = W(blob, "attention.v.weight");
  const float *wc = W(blob, "attention.location_conv.weight");  /* [32][2][31] */
  const float *wd = W(blob, "attention.location_dense.weight"); /* [128][32] */
  real q[ORC_ATT_DIM];
  for (int a = 0; a < ORC_ATT_DIM; ++a) q[a] = dotw(wq + (size_t)a * ORC_ATT_RNN, s->att_h, ORC_ATT_RNN);
  real e[ORC_T_MAX];
  real emax = -INFINITY;
  const int pad = (ORC_LOC_K - 1) / 2;
#pragma omp parallel for schedule(static)
  for (int t = 0; t < T; ++t) {
    real lc[ORC_LOC_F];
    for (int f = 0; f < ORC_LOC_F; ++f) {
      real acc = 0;
      for (int c = 0; c < 2; ++c) {
        const real *src = c ? s->awc : s->aw; /* channel 0 = previous weights, 1 = cumulative */
        for (int k = 0; k < ORC_LOC_K; ++k) {
          int tt = t + k - pad;
          if (tt < 0 || tt >= T) continue;
          acc += (real)wc[(f * 2 + c) * ORC_LOC_K + k] * src[tt];
        }
      }
      lc[f] = acc;
    }
    real en = 0;
    for (int a = 0; a < ORC_ATT_DIM; ++a) {
      real loc = 0;
      for (int f = 0; f < ORC_LOC_F; ++f) loc += (real)wd[a * ORC_LOC_F + f] * lc[f];
      en += (real)wv[a] * (real)tanh((double)(q[a] + loc + pmem[(size_t)t * ORC_ATT_DIM + a]));
    }
    /* mask: true (=> -inf) for t >= unpadded length, src/tacotron2/mod.rs:219-220 */
    e[t] = t >= n_valid ? -INFINITY : en;
  }
  for (int t = 0; t < T; ++t)
    if (e[t] > emax) emax = e[t];
  real den = 0;
  for (int t = 0; t < T; ++t) {
    e[t] = (real)exp((double)(e[t] - emax));
    den += e[t];
  }
  for (int t = 0; t < T; ++t) {
    s->aw[t] = e[t] / den;
    s->awc[t] += s->aw[t];
  }
#pragma omp parallel for schedule(static)
  for (int c = 0; c < ORC_EMB; ++c) {
    real acc = 0;
    for (int t = 0; t < T; ++t) acc += s->aw[t] * memory[(size_t)t * ORC_EMB + c];
    s->ctx[c] = acc;
  }
  /* D4 decoder LSTM on [attention_hidden ; context] */
  real din[ORC_ATT_RNN + ORC_EMB];
  memcpy(din, s->att_h, sizeof(real) * ORC_ATT_RNN);
  memcpy(din + ORC_ATT_RNN, s->ctx, sizeof(real) * ORC_EMB);
  lstm_cell(W(blob, "decoder_rnn.weight_ih"), W(blob, "decoder_rnn.weight_hh"),
            W(blob, "decoder_rnn.bias_ih"), W(blob, "decoder_rnn.bias_hh"), din,
            ORC_ATT_RNN + ORC_EMB, s->dec_h, s->dec_c, ORC_DEC_RNN);
  /* D5 projection + gate on [decoder_hidden ; context] */
  real hc[ORC_DEC_RNN + ORC_EMB];
  memcpy(hc, s->dec_h, sizeof(real) * ORC_DEC_RNN);
  memcpy(hc + ORC_DEC_RNN, s->ctx, sizeof(real) * ORC_EMB);
  const float *wp = W(blob, "linear_projection.weight"), *bp = W(blob, "linear_projection.bias");
  for (int m = 0; m < ORC_N_MEL; ++m)
    mel[m] = dotw(wp + (size_t)m * (ORC_DEC_RNN + ORC_EMB), hc, ORC_DEC_RNN + ORC_EMB) + (real)bp[m];
  *gate = dotw(W(blob, "gate_layer.weight"), hc, ORC_DEC_RNN + ORC_EMB) + (real)W(blob, "gate_layer.bias")[0];
  /* next decoder_input <= decoder_output, src/tacotron2/mod.rs:332 */
  memcpy(s->dec_in, mel, sizeof(real) * ORC_N_MEL);
}

/* run_decoder frame loop, src/tacotron2/mod.rs:302-342: the frame that trips the gate is kept
 * (:312-324); stop when sigmoid(gate) > threshold or i+1 == max_steps (:319-320).             */
int orc_run_decoder(const float *blob, const real *memory, const real *pmem, int T, int n_valid,
                    const orc_decoder_opts *o, real *frames, real *gates) {
  orc_decoder_state *s = (orc_decoder_state *)malloc(sizeof(*s));
  orc_decoder_state_init(s);
  int limit = o->fixed_steps > 0 ? o->fixed_steps : o->max_steps;
  int F = 0;
  for (int i = 0; i < limit; ++i) {
    real gate;
    orc_decoder_step(blob, memory, pmem, T, n_valid, s, o, (uint32_t)i, frames + (size_t)i * ORC_N_MEL, &gate);
    if (gates) gates[i] = gate;
    F = i + 1;
    if (o->fixed_steps > 0) continue;
    if (orc_sigmoid(gate) > (real)o->gate_threshold || i + 1 == o->max_steps) break;
  }
  free(s);
  return F;
}

/* postnet.onnx, src/tacotron2/mod.rs:345-355: input (1 x 80 x F) = frames transposed (:345);
 * 5 x [conv k5 pad2 + BN], tanh after the first four, residual add inside the graph;
 * output "mel_outputs_postnet" (80 x F) is the final mel (:349-355).                          */
void orc_postnet(const float *blob, const real *frames, int F, real *out) {
  tab_init();
  real *x = (real *)malloc(sizeof(real) * (size_t)F * ORC_POST_CH);
  real *y = (real *)malloc(sizeof(real) * (size_t)F * ORC_POST_CH);
  memcpy(x, frames, sizeof(real) * (size_t)F * ORC_N_MEL);
  char nm[96];
  for (int i = 0; i < ORC_POST_CONVS; ++i) {
    int ci = i == 0 ? ORC_N_MEL : ORC_POST_CH;
    int co = i == ORC_POST_CONVS - 1 ? ORC_N_MEL : ORC_POST_CH;
    const float *p[6];
    static const char *suf[6] = {"conv.weight", "conv.bias",       "bn.weight",
                                 "bn.bias",     "bn.running_mean", "bn.running_var"};
    for (int q = 0; q < 6; ++q) {
      snprintf(nm, sizeof nm, "postnet.convolutions.%d.%s", i, suf[q]);
      p[q] = W(blob, nm);
    }
    conv1d_bn(p[0], p[1], p[2], p[3], p[4], p[5], x, F, ci, co, ORC_POST_K,
              i == ORC_POST_CONVS - 1 ? 0 : 2, y);
    real *tmp = x;
    x = y;
    y = tmp;
  }
  for (int t = 0; t < F; ++t)
    for (int m = 0; m < ORC_N_MEL; ++m)
      out[(size_t)m * F + t] = frames[(size_t)t * ORC_N_MEL + m] + x[(size_t)t * ORC_N_MEL + m];
  free(x);
  free(y);
}

/* infer_chunk, src/tacotron2/mod.rs:361-393: pad to the window with id 0 (:369-371), the
 * encoder is told plen = padded length (:375), the decoder mask uses the un-padded length
 * (:387 -> :219-220).                                                                         */
int orc_infer_chunk(const float *blob, const int64_t *ids, int n, int window,
                    const orc_decoder_opts *o, real *out_80xF) {
  int T = n < window ? window : n;
  int64_t *padded = (int64_t *)calloc((size_t)T, sizeof(int64_t));
  memcpy(padded, ids, sizeof(int64_t) * (size_t)n);
  real *memory = (real *)malloc(sizeof(real) * (size_t)T * ORC_EMB);
  real *pmem = (real *)malloc(sizeof(real) * (size_t)T * ORC_ATT_DIM);
  orc_encoder(blob, padded, T, memory, pmem);
  int limit = o->fixed_steps > 0 ? o->fixed_steps : o->max_steps;
  real *frames = (real *)malloc(sizeof(real) * (size_t)limit * ORC_N_MEL);
  int F = orc_run_decoder(blob, memory, pmem, T, n, o, frames, NULL);
  orc_postnet(blob, frames, F, out_80xF);
  free(frames);
  free(pmem);
  free(memory);
  free(padded);
  return F;
}

/* ------------------------------------------------------------------------------------------ */
/* Griffin-Lim.  The reference calls the un-vendored crate `griffin-lim` 0.2.0 @e6415314       */
/* (Cargo.lock:666-668) at src/tacotron2/mod.rs:453,456 and src/lib.rs:141; the crate is a     */
/* port of librosa (slides/vocoding.typ:50).  Restated here from librosa 0.9's published       */
/* filters.mel / mel_to_stft / nnls / griffinlim / stft / istft.                               */
/* ------------------------------------------------------------------------------------------ */
static double hz_to_mel(double f) { /* Slaney scale */
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0;
  const double min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0;
  const double min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

/* create_mel_filter_bank(22050.0, 1024, 80, 0.0, Some(8000.0)), src/tacotron2/mod.rs:453:
 * triangular filters on the Slaney mel scale, area-normalised ("slaney" norm), float32 out.   */
void orc_mel_filter_bank(double sr, int n_fft, int n_mels, double fmin, double fmax, float *out) {
  int nb = n_fft / 2 + 1;
  double *mel_f = (double *)malloc(sizeof(double) * (n_mels + 2));
  double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
  for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
  for (int i = 0; i < n_mels; ++i) {
    double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
    for (int b = 0; b < nb; ++b) {
      double f = (sr / 2.0) * b / (nb - 1);
      double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
      double upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
      double w = lower < upper ? lower : upper;
      out[(size_t)i * nb + b] = (float)((w > 0 ? w : 0) * enorm);
    }
  }
  free(mel_f);
}

/* Moore-Penrose inverse of a full-row-rank basis: pinv = A^T (A A^T)^-1 (Cholesky, fp64).
 * This is x0 = lstsq(A, M) of librosa.util.nnls; the L-BFGS-B refinement that follows it in
 * librosa exits at iteration 0 for this objective scaling (tests/test_oracle_griffinlim.py
 * re-derives that with scipy's L-BFGS-B), so clip(pinv @ M, 0) IS the NNLS result.            */
int orc_pinv(const float *basis, int n_mels, int n_bins, float *out) {
  int n = n_mels;
  double *G = (double *)calloc((size_t)n * n, sizeof(double));
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int b = 0; b < n_bins; ++b) s += (double)basis[(size_t)i * n_bins + b] * basis[(size_t)j * n_bins + b];
      G[i * n + j] = G[j * n + i] = s;
    }
  /* Cholesky G = L L^T in place (lower) */
  for (int j = 0; j < n; ++j) {
    double d = G[j * n + j];
    for (int k = 0; k < j; ++k) d -= G[j * n + k] * G[j * n + k];
    if (d <= 0) {
      free(G);
      return 1;
    }
    d = sqrt(d);
    G[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = G[i * n + j];
      for (int k = 0; k < j; ++k) s -= G[i * n + k] * G[j * n + k];
      G[i * n + j] = s / d;
    }
  }
  double *z = (double *)malloc(sizeof(double) * n);
  for (int b = 0; b < n_bins; ++b) { /* solve G z = A[:,b] */
    for (int i = 0; i < n; ++i) {
      double s = basis[(size_t)i * n_bins + b];
      for (int k = 0; k < i; ++k) s -= G[i * n + k] * z[k];
      z[i] = s / G[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = z[i];
      for (int k = i + 1; k < n; ++k) s -= G[k * n + i] * z[k];
      z[i] = s / G[i * n + i];
    }
    for (int i = 0; i < n; ++i) out[(size_t)b * n + i] = (float)z[i];
  }
  free(z);
  free(G);
  return 0;
}

/* GriffinLim::infer step 1: de-compress the natural-log mel (Tacotron2's
 * dynamic-range compression is ln(clamp(x,1e-5))), NNLS against the mel basis, then the
 * librosa mel_to_stft exponent 1/power (power = 1.7, src/tacotron2/mod.rs:456).               */
void orc_mel_to_linear(const float *pinv, int n_mels, int n_bins, const real *mel, int F,
                       real power, real *S) {
#pragma omp parallel for schedule(static)
  for (int t = 0; t < F; ++t) {
    real col[512];
    for (int m = 0; m < n_mels; ++m) col[m] = (real)exp((double)mel[(size_t)m * F + t]);
    for (int b = 0; b < n_bins; ++b) {
      real v = 0;
      for (int m = 0; m < n_mels; ++m) v += (real)pinv[(size_t)b * n_mels + m] * col[m];
      v = v > 0 ? v : 0;
      S[(size_t)b * F + t] = (real)pow((double)v, 1.0 / (double)power);
    }
  }
}

/* The general form of step 1 behind the options of xdtts_griffinlim_opts (the crate's conventions
 * are not in the checkout, so each is a switch; the defaults reproduce orc_mel_to_linear):
 *   decompress   0: m = exp(mel) (Tacotron2's natural-log compression)   1: m = mel   2: m = 10^mel
 *   NNLS         x0 = max(pinv m, 0); then nnls_iters projected-gradient steps of
 *                min 1/2 |A x - m|^2, x >= 0:   x <- max(x - (1/L) A^T (A x - m), 0),  L = lambda_max(A A^T)
 *                (the fixed point of the iteration is the bounded least-squares solution the crate's
 *                L-BFGS-B refinement seeks; 0 steps = the clipped least-squares start)
 *   power_mode   0: S = x^(1/power) (librosa mel_to_stft)   1: S = x^power   2: S = x               */
double orc_nnls_lipschitz(const float *basis, int n_mels, int n_bins) {
  /* largest eigenvalue of A A^T by power iteration in double */
  double *G = (double *)calloc((size_t)n_mels * n_mels, sizeof(double));
  double *v = (double *)malloc(sizeof(double) * n_mels), *w = (double *)malloc(sizeof(double) * n_mels);
  for (int i = 0; i < n_mels; ++i)
    for (int j = 0; j <= i; ++j) {
      double a = 0;
      for (int b = 0; b < n_bins; ++b) a += (double)basis[(size_t)i * n_bins + b] * basis[(size_t)j * n_bins + b];
      G[(size_t)i * n_mels + j] = G[(size_t)j * n_mels + i] = a;
    }
  for (int i = 0; i < n_mels; ++i) v[i] = 1.0 / sqrt((double)n_mels);
  double lam = 0;
  for (int it = 0; it < 1000; ++it) {
    double nrm = 0;
    for (int i = 0; i < n_mels; ++i) {
      double a = 0;
      for (int j = 0; j < n_mels; ++j) a += G[(size_t)i * n_mels + j] * v[j];
      w[i] = a;
      nrm += a * a;
    }
    nrm = sqrt(nrm);
    for (int i = 0; i < n_mels; ++i) v[i] = w[i] / nrm;
    if (fabs(nrm - lam) <= 1e-13 * nrm) {
      lam = nrm;
      break;
    }
    lam = nrm;
  }
  free(G);
  free(v);
  free(w);
  return lam;
}

void orc_mel_to_linear_opts(const float *pinv, const float *basis, int n_mels, int n_bins, const real *mel, int F,
                            real power, int nnls_iters, int power_mode, int decompress, real *S) {
  const real step = (real)(1.0 / orc_nnls_lipschitz(basis, n_mels, n_bins));
  const double ex = power_mode == 0 ? 1.0 / (double)power : (power_mode == 1 ? (double)power : 1.0);
#pragma omp parallel for schedule(static)
  for (int t = 0; t < F; ++t) {
    real m[512], r[512];
    real *x = (real *)malloc(sizeof(real) * n_bins);
    for (int k = 0; k < n_mels; ++k) {
      const real v = mel[(size_t)k * F + t];
      m[k] = decompress == 0 ? (real)exp((double)v) : (decompress == 2 ? (real)pow(10.0, (double)v) : v);
    }
    for (int b = 0; b < n_bins; ++b) {
      real v = 0;
      for (int k = 0; k < n_mels; ++k) v += (real)pinv[(size_t)b * n_mels + k] * m[k];
      x[b] = v > 0 ? v : 0;
    }
    for (int it = 0; it < nnls_iters; ++it) {
      for (int k = 0; k < n_mels; ++k) {
        real a = 0;
        for (int b = 0; b < n_bins; ++b) a += (real)basis[(size_t)k * n_bins + b] * x[b];
        r[k] = a - m[k];
      }
      for (int b = 0; b < n_bins; ++b) {
        real g = 0;
        for (int k = 0; k < n_mels; ++k) g += r[k] * (real)basis[(size_t)k * n_bins + b];
        const real v = x[b] - step * g;
        x[b] = v > 0 ? v : 0;
      }
    }
    for (int b = 0; b < n_bins; ++b) S[(size_t)b * F + t] = power_mode == 2 ? x[b] : (real)pow((double)x[b], ex);
    free(x);
  }
}

void orc_phase_init(uint32_t seed, int n_bins, int F, real *phase0) {
  for (int b = 0; b < n_bins; ++b)
    for (int t = 0; t < F; ++t) {
      double u = (double)orc_rng_uniform(seed, 0x47u, (uint32_t)(t * n_bins + b));
      phase0[((size_t)b * F + t) * 2 + 0] = (real)cos(2.0 * M_PI * u);
      phase0[((size_t)b * F + t) * 2 + 1] = (real)sin(2.0 * M_PI * u);
    }
}

/* in-place iterative radix-2 complex FFT, n power of two; sign -1 forward, +1 inverse (unscaled) */
static int cached_n = 0;
static double *tw_c = NULL, *tw_s = NULL;
static void fft_tables(int n) { /* called before any (possibly parallel) loop of fft_c calls */
  if (cached_n != n) {
    free(tw_c);
    free(tw_s);
    tw_c = (double *)malloc(sizeof(double) * n / 2);
    tw_s = (double *)malloc(sizeof(double) * n / 2);
    for (int k = 0; k < n / 2; ++k) {
      tw_c[k] = cos(2.0 * M_PI * k / n);
      tw_s[k] = sin(2.0 * M_PI * k / n);
    }
    cached_n = n;
  }
}
static void fft_c(real *re, real *im, int n, int sign) {
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      real t = re[i];
      re[i] = re[j];
      re[j] = t;
      t = im[i];
      im[i] = im[j];
      im[j] = t;
    }
  }
  for (int len = 2; len <= n; len <<= 1) {
    int step = n / len;
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < len / 2; ++k) {
        real wr = (real)tw_c[k * step], wi = (real)(sign * tw_s[k * step]);
        int a = i + k, b = i + k + len / 2;
        real xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - xr;
        im[b] = im[a] - xi;
        re[a] += xr;
        im[a] += xi;
      }
  }
}

static void hann_periodic(int n, real *w) {
  for (int i = 0; i < n; ++i) w[i] = (real)(0.5 - 0.5 * cos(2.0 * M_PI * i / n));
}

/* librosa.stft(center=True, pad_mode="reflect", window="hann", win_length=n_fft) */
void orc_stft(const real *y, int n, int n_fft, int hop, real *out, int F) {
  int nb = n_fft / 2 + 1, half = n_fft / 2;
  real *win = (real *)malloc(sizeof(real) * n_fft);
  hann_periodic(n_fft, win);
  fft_tables(n_fft);
#pragma omp parallel for schedule(static)
  for (int t = 0; t < F; ++t) {
    real *re = (real *)malloc(sizeof(real) * 2 * n_fft), *im = re + n_fft;
    for (int i = 0; i < n_fft; ++i) {
      /* numpy "reflect" padding: mirror without repeating the edge sample, period 2(n-1)
       * (a single mirror for the usual n > n_fft/2; folds repeatedly for very short signals) */
      int p = t * hop + i - half; /* index into the un-padded signal */
      const int period = 2 * (n - 1);
      if (period > 0) {
        p %= period;
        if (p < 0) p += period;
        if (p >= n) p = period - p;
      } else {
        p = 0;
      }
      re[i] = y[p] * win[i];
      im[i] = 0;
    }
    fft_c(re, im, n_fft, -1);
    for (int b = 0; b < nb; ++b) {
      out[((size_t)b * F + t) * 2 + 0] = re[b];
      out[((size_t)b * F + t) * 2 + 1] = im[b];
    }
    free(re);
  }
  free(win);
}

/* librosa.istft(center=True, length=None): irfft each column, window, overlap-add, divide by
 * the window sum-of-squares where it exceeds tiny, trim n_fft/2 each side.                    */
void orc_istft(const real *spec, int F, int n_fft, int hop, real *y) {
  int nb = n_fft / 2 + 1, half = n_fft / 2;
  int full = n_fft + hop * (F - 1);
  real *win = (real *)malloc(sizeof(real) * n_fft);
  real *rows = (real *)malloc(sizeof(real) * (size_t)F * n_fft); /* irfft of every column */
  real *acc = (real *)calloc((size_t)full, sizeof(real));
  real *wss = (real *)calloc((size_t)full, sizeof(real));
  hann_periodic(n_fft, win);
  fft_tables(n_fft);
#pragma omp parallel for schedule(static)
  for (int t = 0; t < F; ++t) {
    real *re = rows + (size_t)t * n_fft, *im = (real *)malloc(sizeof(real) * n_fft);
    for (int b = 0; b < nb; ++b) {
      re[b] = spec[((size_t)b * F + t) * 2 + 0];
      im[b] = spec[((size_t)b * F + t) * 2 + 1];
    }
    im[0] = 0;
    im[half] = 0; /* irfft ignores the imaginary part of DC and Nyquist */
    for (int b = 1; b < half; ++b) {
      re[n_fft - b] = re[b];
      im[n_fft - b] = -im[b];
    }
    fft_c(re, im, n_fft, +1);
    free(im);
  }
  for (int t = 0; t < F; ++t) { /* overlap-add in ascending frame order (serial: the sums overlap) */
    const real *re = rows + (size_t)t * n_fft;
    for (int i = 0; i < n_fft; ++i) {
      acc[(size_t)t * hop + i] += win[i] * (re[i] / (real)n_fft);
      wss[(size_t)t * hop + i] += win[i] * win[i];
    }
  }
  const real tiny = (sizeof(real) == 4) ? (real)1.17549435e-38 : (real)2.2250738585072014e-308;
  for (int i = 0; i < hop * (F - 1); ++i) {
    real v = acc[i + half], w = wss[i + half];
    y[i] = w > tiny ? v / w : v;
  }
  free(win);
  free(rows);
  free(acc);
  free(wss);
}

/* librosa.griffinlim(S, n_iter, hop, momentum=0.99, init="random"), called through
 * GriffinLim::infer (src/lib.rs:141) with the parameters of src/tacotron2/mod.rs:456:
 * noverlap 768 -> hop 256, 30 iterations, momentum 0.99.                                      */
/* One iteration of the loop above on a caller-held state: ang (unit-modulus phase estimate) and
 * reb (the previous iteration's rebuilt STFT, zeros before the first), both [bins][F][2], are
 * updated in place.  est/inv are scratch of ne*2 and hop*(F-1) reals.                          */
static void griffinlim_iteration(const real *S, real *ang, real *reb, int F, int n_fft, int hop,
                                 real alpha, real *prev, real *est, real *inv) {
  int nb = n_fft / 2 + 1;
  size_t ne = (size_t)nb * F;
  int n = hop * (F - 1);
  memcpy(prev, reb, sizeof(real) * ne * 2);
  for (size_t i = 0; i < ne; ++i) {
    est[2 * i] = S[i] * ang[2 * i];
    est[2 * i + 1] = S[i] * ang[2 * i + 1];
  }
  orc_istft(est, F, n_fft, hop, inv);
  orc_stft(inv, n, n_fft, hop, reb, F);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < ne; ++i) {
    real ar = reb[2 * i] - alpha * prev[2 * i], ai = reb[2 * i + 1] - alpha * prev[2 * i + 1];
    real mag = (real)sqrt((double)(ar * ar + ai * ai)) + (real)1e-16;
    ang[2 * i] = ar / mag;
    ang[2 * i + 1] = ai / mag;
  }
}

void orc_griffinlim_step(const real *S, real *ang, real *reb, int F, int n_fft, int hop, int iters,
                         real momentum) {
  int nb = n_fft / 2 + 1;
  size_t ne = (size_t)nb * F;
  int n = hop * (F - 1);
  real *prev = (real *)malloc(sizeof(real) * ne * 2);
  real *est = (real *)malloc(sizeof(real) * ne * 2);
  real *inv = (real *)malloc(sizeof(real) * (size_t)(n > 0 ? n : 1));
  const real alpha = momentum / ((real)1 + momentum);
  for (int it = 0; it < iters; ++it) griffinlim_iteration(S, ang, reb, F, n_fft, hop, alpha, prev, est, inv);
  free(prev);
  free(est);
  free(inv);
}

void orc_griffinlim(const real *S, const real *phase0, uint32_t seed, int F, int n_fft, int hop,
                    int iters, real momentum, real *audio) {
  int nb = n_fft / 2 + 1;
  size_t ne = (size_t)nb * F;
  real *ang = (real *)malloc(sizeof(real) * ne * 2);
  real *reb = (real *)calloc(ne * 2, sizeof(real));
  real *est = (real *)malloc(sizeof(real) * ne * 2);
  if (phase0)
    memcpy(ang, phase0, sizeof(real) * ne * 2);
  else
    orc_phase_init(seed, nb, F, ang);
  orc_griffinlim_step(S, ang, reb, F, n_fft, hop, iters, momentum);
  for (size_t i = 0; i < ne; ++i) {
    est[2 * i] = S[i] * ang[2 * i];
    est[2 * i + 1] = S[i] * ang[2 * i + 1];
  }
  orc_istft(est, F, n_fft, hop, audio);
  free(ang);
  free(reb);
  free(est);
}

/* G6 -- the last step of GriffinLim::infer (src/lib.rs:141) before src/lib.rs:155 scales by i16::MAX.
 * The crate is absent (Cargo.lock:666-668); the reference's own WAV_SPEC files (slides/audio/goodbye.wav,
 * capital_nonsense.wav: RMS 0.099994 / 0.099995 of full scale, peaks 0.82 / 0.61) say "RMS = 0.1".
 * mode 0: as is; 1: y / max|y|; 2: y * target / sqrt(mean(y^2)); 3: mode 2 with the scale limited to 1 / max|y|, so that
 * no sample leaves [-1, 1] (where src/lib.rs:155's saturating `as i16` cast would clip).  Sums in double in both builds. */
void orc_output_normalise(real *y, size_t n, int mode, double target) {
  if (mode == 1) {
    real peak = 0;
    for (size_t i = 0; i < n; ++i) {
      real a = y[i] < 0 ? -y[i] : y[i];
      if (a > peak) peak = a;
    }
    if (peak > 0)
      for (size_t i = 0; i < n; ++i) y[i] = y[i] / peak;
  } else if ((mode == 2 || mode == 3) && n > 0) {
    double ss = 0, peak = 0;
    for (size_t i = 0; i < n; ++i) {
      ss += (double)y[i] * (double)y[i];
      double a = y[i] < 0 ? -(double)y[i] : (double)y[i];
      if (a > peak) peak = a;
    }
    double r = sqrt(ss / (double)n);
    if (r > 0) {
      double scd = target / r;
      if (mode == 3 && scd * peak > 1.0) scd = 1.0 / peak;
      real sc = (real)scd;
      for (size_t i = 0; i < n; ++i) y[i] = y[i] * sc;
    }
  }
}
