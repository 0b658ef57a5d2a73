// weights.cpp -- tensor table, synthetic init, container IO, device packing.
//
// The reference loads three ONNX graphs (src/tacotron2/mod.rs:246-259); their initialisers are
// the NVIDIA Tacotron2 checkpoint tensors (mod.rs:137-138).  This file owns the equivalent
// parameter set in a flat fp32 container and re-lays it out for the HIP kernels.
#include "weights.h"
#include "kernels.h"

#include <cmath>
#include <fstream>

namespace xdtts {

namespace {

struct TableBuilder {
  std::vector<TensorInfo> t;
  size_t total = 0;
  void add(const std::string &name, int ndim, int d0, int d1, int d2, double fan_in, int kind,
           int rec) {
    TensorInfo ti{};
    std::snprintf(ti.name, sizeof ti.name, "%s", name.c_str());
    ti.ndim = ndim;
    ti.dims[0] = d0;
    ti.dims[1] = d1;
    ti.dims[2] = d2;
    ti.numel = (size_t)d0 * (ndim > 1 ? d1 : 1) * (ndim > 2 ? d2 : 1);
    ti.offset = total;
    ti.bound = (float)(1.0 / std::sqrt(fan_in));
    ti.kind = kind;
    ti.rec = rec;
    total += ti.numel;
    t.push_back(ti);
  }
  void conv_bn(const std::string &p, int co, int ci, int k) {
    double fan = (double)ci * k;
    add(p + ".conv.weight", 3, co, ci, k, fan, 0, 0);
    add(p + ".conv.bias", 1, co, 1, 1, fan, 0, 0);
    add(p + ".bn.weight", 1, co, 1, 1, 1, 1, 0);
    add(p + ".bn.bias", 1, co, 1, 1, 1, 2, 0);
    add(p + ".bn.running_mean", 1, co, 1, 1, 1, 3, 0);
    add(p + ".bn.running_var", 1, co, 1, 1, 1, 4, 0);
  }
  void lstm(const std::string &p, int hidden, int in, int rec) {
    add(p + ".weight_ih", 2, 4 * hidden, in, 1, hidden, 0, 0);
    add(p + ".weight_hh", 2, 4 * hidden, hidden, 1, hidden, 0, rec);
    add(p + ".bias_ih", 1, 4 * hidden, 1, 1, hidden, 0, 0);
    add(p + ".bias_hh", 1, 4 * hidden, 1, 1, hidden, 0, 0);
  }
};

TableBuilder build_table() {
  TableBuilder b;
  b.add("embedding.weight", 2, N_SYMBOLS, EMB, 1, (double)(N_SYMBOLS + EMB) / 6.0, 0, 0);
  for (int i = 0; i < ENC_CONVS; ++i)
    b.conv_bn("encoder.convolutions." + std::to_string(i), EMB, EMB, ENC_K);
  b.lstm("encoder.lstm.fwd", ENC_H, EMB, 0);
  b.lstm("encoder.lstm.bwd", ENC_H, EMB, 0);
  b.add("attention.memory_layer.weight", 2, ATT_DIM, EMB, 1, EMB, 0, 0);
  b.add("prenet.0.weight", 2, PRENET, N_MEL, 1, N_MEL, 0, 0);
  b.add("prenet.1.weight", 2, PRENET, PRENET, 1, PRENET, 0, 0);
  b.lstm("attention_rnn", ATT_RNN, ATT_IN, 1);
  b.add("attention.query_layer.weight", 2, ATT_DIM, ATT_RNN, 1, ATT_RNN, 0, 0);
  b.add("attention.v.weight", 1, ATT_DIM, 1, 1, ATT_DIM, 0, 0);
  b.add("attention.location_conv.weight", 3, LOC_F, 2, LOC_K, 2.0 * LOC_K, 0, 0);
  b.add("attention.location_dense.weight", 2, ATT_DIM, LOC_F, 1, LOC_F, 0, 0);
  b.lstm("decoder_rnn", DEC_RNN, DEC_IN, 1);
  b.add("linear_projection.weight", 2, N_MEL, PROJ_IN, 1, PROJ_IN, 0, 0);
  b.add("linear_projection.bias", 1, N_MEL, 1, 1, PROJ_IN, 0, 0);
  b.add("gate_layer.weight", 1, PROJ_IN, 1, 1, PROJ_IN, 0, 0);
  b.add("gate_layer.bias", 1, 1, 1, 1, PROJ_IN, 0, 0);
  for (int i = 0; i < POST_CONVS; ++i) {
    int ci = i == 0 ? N_MEL : POST_CH;
    int co = i == POST_CONVS - 1 ? N_MEL : POST_CH;
    b.conv_bn("postnet.convolutions." + std::to_string(i), co, ci, POST_K);
  }
  return b;
}

const TableBuilder &table() {
  static const TableBuilder b = build_table();
  return b;
}

const float *T(const std::vector<float> &blob, const std::string &name) {
  int i = tensor_index(name.c_str());
  if (i < 0) fail(XDTTS_ERR_BAD_ARG, "unknown tensor %s", name.c_str());
  return blob.data() + tensor_table()[i].offset;
}

// conv weight [co][ci][k] + conv bias + BN(eval) -> GEMM operand [co][k][ci], folded bias
void fold_conv(const std::vector<float> &blob, const std::string &p, int co, int ci, int k,
               ConvGemm &out, hipStream_t s) {
  const float *w = T(blob, p + ".conv.weight"), *b = T(blob, p + ".conv.bias");
  const float *g = T(blob, p + ".bn.weight"), *be = T(blob, p + ".bn.bias");
  const float *mu = T(blob, p + ".bn.running_mean"), *var = T(blob, p + ".bn.running_var");
  std::vector<float> wg((size_t)co * k * ci), bg(co);
  for (int o = 0; o < co; ++o) {
    double scale = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
    for (int c = 0; c < ci; ++c)
      for (int j = 0; j < k; ++j)
        wg[((size_t)o * k + j) * ci + c] = (float)((double)w[((size_t)o * ci + c) * k + j] * scale);
    bg[o] = (float)(((double)b[o] - (double)mu[o]) * scale + (double)be[o]);
  }
  out.co = co;
  out.ci = ci;
  out.k = k;
  out.w.upload(wg.data(), wg.size(), s);
  out.b.upload(bg.data(), bg.size(), s);
  HIP_CHECK(hipStreamSynchronize(s));  // host staging vectors die at scope exit
}

// LSTM -> [unit][gate][W_ih row | W_hh row], bias [unit][gate] = b_ih + b_hh
void pack_lstm(const std::vector<float> &blob, const std::string &p, int hidden, int nin,
               DevBuf<float> &w, DevBuf<float> &b, hipStream_t s) {
  const float *wih = T(blob, p + ".weight_ih"), *whh = T(blob, p + ".weight_hh");
  const float *bih = T(blob, p + ".bias_ih"), *bhh = T(blob, p + ".bias_hh");
  const int cols = nin + hidden;
  std::vector<float> pw((size_t)hidden * 4 * cols), pb((size_t)hidden * 4);
  for (int u = 0; u < hidden; ++u)
    for (int g = 0; g < 4; ++g) {
      float *dst = pw.data() + ((size_t)u * 4 + g) * cols;
      const int r = g * hidden + u;
      std::memcpy(dst, wih + (size_t)r * nin, sizeof(float) * nin);
      std::memcpy(dst + nin, whh + (size_t)r * hidden, sizeof(float) * hidden);
      pb[(size_t)u * 4 + g] = bih[r] + bhh[r];
    }
  w.upload(pw.data(), pw.size(), s);
  b.upload(pb.data(), pb.size(), s);
  HIP_CHECK(hipStreamSynchronize(s));
}

void upload_transposed(const float *src, int rows, int cols, DevBuf<float> &dst, hipStream_t s) {
  std::vector<float> t((size_t)rows * cols);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = src[(size_t)r * cols + c];
  dst.upload(t.data(), t.size(), s);
  HIP_CHECK(hipStreamSynchronize(s));
}

constexpr char MAGIC[8] = {'X', 'D', 'T', 'W', '0', '0', '0', '1'};
struct DiskEntry {
  char name[64];
  uint32_t ndim;
  uint32_t dims[3];
  uint64_t offset, numel;
};

}  // namespace

const std::vector<TensorInfo> &tensor_table() { return table().t; }
size_t tensor_total() { return table().total; }
int tensor_index(const char *name) {
  const auto &t = tensor_table();
  for (size_t i = 0; i < t.size(); ++i)
    if (!std::strcmp(t[i].name, name)) return (int)i;
  return -1;
}

void synthetic_blob(uint32_t seed, float rec_scale, std::vector<float> &blob) {
  const auto &tab = tensor_table();
  blob.resize(tensor_total());
  for (size_t i = 0; i < tab.size(); ++i) {
    const TensorInfo &t = tab[i];
    float *w = blob.data() + t.offset;
    for (size_t j = 0; j < t.numel; ++j) {
      const float u = rng_uniform(seed, (uint32_t)i, (uint32_t)j);
      const float s = 2.0f * u - 1.0f;
      float v;
      switch (t.kind) {
        case 0: v = t.bound * s; break;
        case 1: v = std::fmaf(0.1f, s, 1.0f); break;
        case 2: v = 0.1f * s; break;
        case 3: v = 0.1f * s; break;
        default: v = std::fmaf(0.2f, u, 1.0f); break;
      }
      if (t.rec) v *= rec_scale;
      w[j] = v;
    }
  }
}

void save_container(const std::string &dir, const std::vector<float> &blob) {
  const auto &tab = tensor_table();
  std::ofstream f(dir + "/tacotron2.xdtw", std::ios::binary);
  if (!f) fail(XDTTS_ERR_IO, "cannot open %s/tacotron2.xdtw for writing", dir.c_str());
  uint32_t n = (uint32_t)tab.size();
  f.write(MAGIC, 8);
  f.write((const char *)&n, 4);
  for (const auto &t : tab) {
    DiskEntry e{};
    std::memcpy(e.name, t.name, sizeof e.name);
    e.ndim = (uint32_t)t.ndim;
    for (int d = 0; d < 3; ++d) e.dims[d] = (uint32_t)t.dims[d];
    e.offset = t.offset;
    e.numel = t.numel;
    f.write((const char *)&e, sizeof e);
  }
  f.write((const char *)blob.data(), (std::streamsize)(blob.size() * sizeof(float)));
  if (!f) fail(XDTTS_ERR_IO, "short write to %s/tacotron2.xdtw", dir.c_str());
}

void load_model_dir(const std::string &dir, std::vector<float> &blob) {
  if (std::ifstream(dir + "/tacotron2.xdtw", std::ios::binary)) return load_container(dir, blob);
  if (onnx_model_dir(dir)) return load_onnx_dir(dir, blob);
  fail(XDTTS_ERR_IO, "loading tacotron2 weights: %s holds neither tacotron2.xdtw nor encoder.onnx / decoder_iter.onnx / postnet.onnx "
                     "(the reference's model directory, src/tacotron2/mod.rs:246-259)", dir.c_str());
}

void load_container(const std::string &dir, std::vector<float> &blob) {
  const std::string path = dir + "/tacotron2.xdtw";
  std::ifstream f(path, std::ios::binary);
  if (!f) fail(XDTTS_ERR_IO, "loading tacotron2 weights: cannot open %s", path.c_str());
  char magic[8];
  uint32_t n = 0;
  f.read(magic, 8);
  f.read((char *)&n, 4);
  if (!f || std::memcmp(magic, MAGIC, 8) != 0 || n == 0 || n > 4096)
    fail(XDTTS_ERR_IO, "loading tacotron2 weights: %s is not an XDTW0001 container", path.c_str());
  std::vector<DiskEntry> ent(n);
  f.read((char *)ent.data(), (std::streamsize)(sizeof(DiskEntry) * n));
  if (!f) fail(XDTTS_ERR_IO, "loading tacotron2 weights: truncated header in %s", path.c_str());
  const std::streamoff data0 = f.tellg();
  const auto &tab = tensor_table();
  blob.assign(tensor_total(), 0.0f);
  std::vector<char> seen(tab.size(), 0);
  for (const auto &e : ent) {
    char nm[65];
    std::memcpy(nm, e.name, 64);
    nm[64] = 0;
    int i = tensor_index(nm);
    if (i < 0) continue;  // unknown extras are ignored
    const TensorInfo &t = tab[i];
    bool ok = (int)e.ndim == t.ndim && e.numel == t.numel;
    for (int d = 0; d < t.ndim && ok; ++d) ok = (int)e.dims[d] == t.dims[d];
    if (!ok) fail(XDTTS_ERR_IO, "loading tacotron2 weights: tensor %s has the wrong shape", nm);
    f.seekg(data0 + (std::streamoff)(e.offset * sizeof(float)));
    f.read((char *)(blob.data() + t.offset), (std::streamsize)(t.numel * sizeof(float)));
    if (!f) fail(XDTTS_ERR_IO, "loading tacotron2 weights: tensor %s is truncated", nm);
    seen[i] = 1;
  }
  for (size_t i = 0; i < tab.size(); ++i)
    if (!seen[i]) fail(XDTTS_ERR_IO, "loading tacotron2 weights: tensor %s is missing", tab[i].name);
}

namespace {
void pack_lstm_mfma(const std::vector<float> &blob, const std::string &p, int hidden, int nin, DevBuf<float> &out,
                    hipStream_t s) {
  const float *wih = T(blob, p + ".weight_ih"), *whh = T(blob, p + ".weight_hh");
  const int cols = nin + hidden, KW = cols / MFMA_WAVES, JJ = KW / 16;
  std::vector<float> m((size_t)hidden * 4 * cols);
  auto W = [&](int unit, int gate, int c) {
    const int r = gate * hidden + unit;  // PyTorch gate order i,f,g,o
    return c < nin ? wih[(size_t)r * nin + c] : whh[(size_t)r * hidden + (c - nin)];
  };
  for (int blk = 0; blk < hidden / 4; ++blk)
    for (int wv = 0; wv < MFMA_WAVES; ++wv)
      for (int jj = 0; jj < JJ; ++jj)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, g4 = lane >> 4;
          float *dst = m.data() + ((((size_t)blk * MFMA_WAVES + wv) * JJ + jj) * 64 + lane) * 4;
          for (int c = 0; c < 4; ++c) dst[c] = W(blk * 4 + (i >> 2), i & 3, wv * KW + 16 * jj + 4 * g4 + c);
        }
  out.upload(m.data(), m.size(), s);
  HIP_CHECK(hipStreamSynchronize(s));
}
}  // namespace

void DeviceWeights::ensure_batched_layout(const std::vector<float> &blob, hipStream_t s) {
  if (att_wm.p && dec_wm.p) return;
  pack_lstm_mfma(blob, "attention_rnn", ATT_RNN, ATT_IN, att_wm, s);
  pack_lstm_mfma(blob, "decoder_rnn", DEC_RNN, DEC_IN, dec_wm, s);
}

void DeviceWeights::upload(const std::vector<float> &blob, hipStream_t s) {
  if (blob.size() != tensor_total())
    fail(XDTTS_ERR_BAD_ARG, "weight blob has %zu floats, expected %zu", blob.size(), tensor_total());
  emb.upload(T(blob, "embedding.weight"), (size_t)N_SYMBOLS * EMB, s);
  for (int i = 0; i < ENC_CONVS; ++i)
    fold_conv(blob, "encoder.convolutions." + std::to_string(i), EMB, EMB, ENC_K, enc_conv[i], s);
  for (int d = 0; d < 2; ++d) {
    const std::string p = d ? "encoder.lstm.bwd" : "encoder.lstm.fwd";
    enc_wih[d].upload(T(blob, p + ".weight_ih"), (size_t)4 * ENC_H * EMB, s);
    std::vector<float> b(4 * ENC_H);
    const float *bih = T(blob, p + ".bias_ih"), *bhh = T(blob, p + ".bias_hh");
    for (int r = 0; r < 4 * ENC_H; ++r) b[r] = bih[r] + bhh[r];
    enc_bias[d].upload(b.data(), b.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    upload_transposed(T(blob, p + ".weight_hh"), 4 * ENC_H, ENC_H, enc_whhT[d], s);
  }
  mem_w.upload(T(blob, "attention.memory_layer.weight"), (size_t)ATT_DIM * EMB, s);
  upload_transposed(T(blob, "prenet.0.weight"), PRENET, N_MEL, pre0T, s);
  upload_transposed(T(blob, "prenet.1.weight"), PRENET, PRENET, pre1T, s);
  pack_lstm(blob, "attention_rnn", ATT_RNN, ATT_IN, att_w, att_b, s);
  q_w.upload(T(blob, "attention.query_layer.weight"), (size_t)ATT_DIM * ATT_RNN, s);
  {
    const float *wq = T(blob, "attention.query_layer.weight");
    std::vector<float> q4((size_t)256 * ATT_DIM * 4);
    for (int blk = 0; blk < 256; ++blk)
      for (int a = 0; a < ATT_DIM; ++a)
        for (int i = 0; i < 4; ++i) q4[((size_t)blk * ATT_DIM + a) * 4 + i] = wq[(size_t)a * ATT_RNN + 4 * blk + i];
    q_w4.upload(q4.data(), q4.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
  v_w.upload(T(blob, "attention.v.weight"), ATT_DIM, s);
  upload_transposed(T(blob, "attention.location_conv.weight"), LOC_F, 2 * LOC_K, loc_conv, s);  // -> [c][k][f]
  upload_transposed(T(blob, "attention.location_dense.weight"), ATT_DIM, LOC_F, loc_denseT, s);
  {
    // loc[t][a] = sum_f dense[a][f] sum_{c,k} conv[f][c][k] w_c[t+k-15]: one 62-tap filter per attention dim
    const float *cv = T(blob, "attention.location_conv.weight"), *dn = T(blob, "attention.location_dense.weight");
    std::vector<float> gf((size_t)2 * LOC_K * ATT_DIM);
    for (int ck = 0; ck < 2 * LOC_K; ++ck)
      for (int a = 0; a < ATT_DIM; ++a) {
        double acc = 0.0;
        for (int f = 0; f < LOC_F; ++f) acc += (double)dn[(size_t)a * LOC_F + f] * (double)cv[(size_t)f * 2 * LOC_K + ck];
        gf[(size_t)ck * ATT_DIM + a] = (float)acc;
      }
    loc_fused.upload(gf.data(), gf.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
  pack_lstm(blob, "decoder_rnn", DEC_RNN, DEC_IN, dec_w, dec_b, s);
  {
    std::vector<float> pw((size_t)(N_MEL + 1) * PROJ_IN), pb(N_MEL + 1);
    std::memcpy(pw.data(), T(blob, "linear_projection.weight"), sizeof(float) * N_MEL * PROJ_IN);
    std::memcpy(pw.data() + (size_t)N_MEL * PROJ_IN, T(blob, "gate_layer.weight"), sizeof(float) * PROJ_IN);
    std::memcpy(pb.data(), T(blob, "linear_projection.bias"), sizeof(float) * N_MEL);
    pb[N_MEL] = T(blob, "gate_layer.bias")[0];
    proj_w.upload(pw.data(), pw.size(), s);
    std::vector<float> wh4((size_t)256 * 84 * 4, 0.f), wc((size_t)8 * (N_MEL + 1) * 64);
    for (int blk = 0; blk < 256; ++blk)
      for (int m = 0; m <= N_MEL; ++m)
        for (int i = 0; i < 4; ++i) wh4[((size_t)blk * 84 + m) * 4 + i] = pw[(size_t)m * PROJ_IN + 4 * blk + i];
    for (int cb = 0; cb < 8; ++cb)
      for (int m = 0; m <= N_MEL; ++m)
        for (int c = 0; c < 64; ++c) wc[((size_t)cb * (N_MEL + 1) + m) * 64 + c] = pw[(size_t)m * PROJ_IN + DEC_RNN + 64 * cb + c];
    proj_wh4.upload(wh4.data(), wh4.size(), s);
    proj_wc.upload(wc.data(), wc.size(), s);
    proj_b.upload(pb.data(), pb.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
  ctx_w.alloc((size_t)CTXF_ROWS * EMB);
  launch_pack_ctx_rows(att_w.p, dec_w.p, proj_w.p, ctx_w.p, s);
  for (int i = 0; i < POST_CONVS; ++i) {
    int ci = i == 0 ? N_MEL : POST_CH;
    int co = i == POST_CONVS - 1 ? N_MEL : POST_CH;
    fold_conv(blob, "postnet.convolutions." + std::to_string(i), co, ci, POST_K, post_conv[i], s);
  }
  HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace xdtts
