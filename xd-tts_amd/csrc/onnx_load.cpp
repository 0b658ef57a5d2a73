// onnx_load.cpp -- Tacotron2::load(dir) on the reference's own model directory
// (src/tacotron2/mod.rs:242-267 opens encoder.onnx, decoder_iter.onnx and postnet.onnx): a minimal
// protobuf wire-format reader that pulls the weights (initialisers and Constant nodes) out of the three
// graphs and maps them onto the canonical tensor table of weights.cpp.  No onnx / protobuf / onnxruntime
// dependency; nothing of the graphs is executed -- the kernels implement the layers, the files only
// supply the numbers.  SURVEY.md section 8(f) rank 1.
//
// An exporter with constant folding drops most parameter names, so tensors are found by how the graph
// uses them, not by name:
//   * LSTM nodes (the encoder BiLSTM; nn.LSTMCell exports as a 1-step LSTM) carry packed W [dirs, 4H, in],
//     R [dirs, 4H, H], B [dirs, 8H] in ONNX gate order i,o,f,c -> split, re-ordered to PyTorch's i,f,g,o;
//     the node is identified by W's input width (512 encoder, 768 attention_rnn, 1536 decoder_rnn);
//   * Conv nodes in graph order (encoder 3 x [512,512,5]; postnet 5; location conv [32,2,31]); a
//     BatchNormalization node fed by a conv supplies its statistics, otherwise the conv is taken as
//     already folded and the BN is written as the identity;
//   * linear layers by their (out, in) shape, oriented by how the constant is consumed (MatMul(x, W^T)
//     vs Gemm(..., transB)), biases from Gemm's C or the Add that follows a MatMul;
//   * the embedding is the Gather table [148, 512].
// UNVERIFIED AGAINST THE REAL FILES (the checkout holds git-LFS pointers only): the conventions are those of
// torch.onnx.export for the NVIDIA model, exercised on synthetic graphs written both ways
// (tests/test_onnx_import_cpu.py).  Every tensor must be found exactly once with its exact shape and
// finite values, or the load fails with XDTTS_ERR_IO naming what is wrong -- a wrong guess cannot yield
// a silently half-filled model.
#include <cmath>
#include <fstream>
#include <map>
#include <memory>

#include "weights.h"

namespace xdtts {
namespace {

typedef std::vector<unsigned char> Bytes;

struct Span {
  const unsigned char *p = nullptr;
  size_t n = 0;
};

struct Field {
  uint32_t num;
  int wire;
  uint64_t v;  // varint / fixed value
  Span s;      // length-delimited payload
};

[[noreturn]] void bad(const std::string &path, const char *what) { fail(XDTTS_ERR_IO, "%s: %s", path.c_str(), what); }

struct Reader {
  Span s;
  size_t i = 0;
  const std::string *path;
  Reader(Span sp, const std::string &p) : s(sp), path(&p) {}
  uint64_t varint() {
    uint64_t r = 0;
    for (int sh = 0; sh < 70; sh += 7) {
      if (i >= s.n) bad(*path, "truncated protobuf varint");
      const unsigned char c = s.p[i++];
      r |= (uint64_t)(c & 0x7F) << sh;
      if (c < 0x80) return r;
    }
    bad(*path, "malformed protobuf varint");
  }
  bool next(Field &f) {
    if (i >= s.n) return false;
    const uint64_t key = varint();
    f.num = (uint32_t)(key >> 3);
    f.wire = (int)(key & 7);
    f.v = 0;
    f.s = Span();
    switch (f.wire) {
      case 0: f.v = varint(); break;
      case 1:
        if (i + 8 > s.n) bad(*path, "truncated fixed64");
        std::memcpy(&f.v, s.p + i, 8);
        i += 8;
        break;
      case 2: {
        const uint64_t len = varint();
        if (len > s.n - i) bad(*path, "truncated length-delimited field");
        f.s.p = s.p + i;
        f.s.n = (size_t)len;
        i += (size_t)len;
        break;
      }
      case 5: {
        if (i + 4 > s.n) bad(*path, "truncated fixed32");
        uint32_t x;
        std::memcpy(&x, s.p + i, 4);
        f.v = x;
        i += 4;
        break;
      }
      default: bad(*path, "unsupported protobuf wire type");
    }
    return true;
  }
};

struct Tensor {
  std::vector<long> dims;
  std::vector<float> data;
  size_t numel() const {
    size_t n = 1;
    for (long d : dims) n *= (size_t)d;
    return n;
  }
  bool is(std::initializer_list<long> shape) const { return dims == std::vector<long>(shape); }
};
typedef std::shared_ptr<Tensor> TensorP;

// TensorProto: dims 1, data_type 2, float_data 4, name 8, raw_data 9, external data 13/14.  Only FLOAT
// tensors are weights; everything else (shapes, axes) returns null.
TensorP parse_tensor(Span s, const std::string &path, std::string *name) {
  Reader r(s, path);
  Field f;
  auto t = std::make_shared<Tensor>();
  int dtype = 0;
  Span raw;
  bool has_raw = false;
  std::vector<float> floats;
  while (r.next(f)) {
    if (f.num == 1) {
      if (f.wire == 2) {
        Reader pr(f.s, path);
        while (pr.i < pr.s.n) t->dims.push_back((long)(int64_t)pr.varint());
      } else {
        t->dims.push_back((long)(int64_t)f.v);
      }
    } else if (f.num == 2) {
      dtype = (int)f.v;
    } else if (f.num == 4) {
      if (f.wire == 2) {
        const size_t k = f.s.n / 4, o = floats.size();
        floats.resize(o + k);
        std::memcpy(floats.data() + o, f.s.p, k * 4);
      } else {
        float x;
        const uint32_t u = (uint32_t)f.v;
        std::memcpy(&x, &u, 4);
        floats.push_back(x);
      }
    } else if (f.num == 8) {
      if (name) name->assign((const char *)f.s.p, f.s.n);
    } else if (f.num == 9) {
      raw = f.s;
      has_raw = true;
    } else if (f.num == 13 || f.num == 14) {
      fail(XDTTS_ERR_IO, "%s: a tensor uses external data: export the model with the weights embedded", path.c_str());
    }
  }
  if (dtype != 1) return nullptr;
  {
    size_t prod = 1;  // the dims come from an untrusted file: bound the product before it can wrap
    for (long d : t->dims) {
      if (d < 0 || d > (1L << 28)) bad(path, "tensor with an invalid dimension");
      if (d != 0 && prod > ((size_t)1 << 31) / (size_t)d) bad(path, "tensor with more than 2^31 elements");
      prod *= (size_t)d;
    }
  }
  const size_t n = t->numel();
  if (has_raw) {
    if (raw.n != n * 4) bad(path, "tensor raw_data size does not match its shape");
    t->data.resize(n);
    std::memcpy(t->data.data(), raw.p, n * 4);
  } else {
    if (floats.size() != n) bad(path, "tensor float_data size does not match its shape");
    t->data.swap(floats);
  }
  return t;
}

struct Node {
  std::vector<std::string> in, out;
  std::string op, name;
  std::map<std::string, double> num;  // float / int attributes
  TensorP value;                      // Constant's tensor attribute
};

struct Graph {
  std::string path;
  Bytes file;
  std::vector<Node> nodes;
  std::map<std::string, TensorP> tensors;
  std::vector<std::string> inputs, outputs;  // GraphProto.input / .output names (initialisers listed as inputs by old exporters removed)

  explicit Graph(const std::string &p) : path(p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) fail(XDTTS_ERR_IO, "loading tacotron2 weights: cannot open %s", p.c_str());
    file.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    static const char lfs[] = "version https://git-lfs";
    if (file.size() >= sizeof lfs - 1 && std::memcmp(file.data(), lfs, sizeof lfs - 1) == 0)
      fail(XDTTS_ERR_IO, "%s is a git-LFS pointer file (%zu bytes), not the model: fetch the real file with `git lfs pull`", p.c_str(),
           file.size());
    Span model{file.data(), file.size()}, graph;
    Field fd;
    std::vector<std::string> init_names;
    for (Reader r(model, path); r.next(fd);)
      if (fd.num == 7 && fd.wire == 2) graph = fd.s;  // ModelProto.graph
    if (!graph.p) bad(path, "no GraphProto in the file (not an ONNX model?)");
    for (Reader r(graph, path); r.next(fd);) {
      if (fd.num == 1 && fd.wire == 2) {
        nodes.push_back(parse_node(fd.s));
      } else if (fd.num == 5 && fd.wire == 2) {  // initializer
        std::string nm;
        TensorP t = parse_tensor(fd.s, path, &nm);
        if (t) tensors[nm] = t;
        init_names.push_back(nm);
      } else if ((fd.num == 11 || fd.num == 12) && fd.wire == 2) {  // ValueInfoProto: name = field 1
        Field v;
        for (Reader vr(fd.s, path); vr.next(v);)
          if (v.num == 1 && v.wire == 2) (fd.num == 11 ? inputs : outputs).emplace_back((const char *)v.s.p, v.s.n);
      }
    }
    for (const std::string &nm : init_names)  // IR version < 4 lists every initialiser among the inputs
      for (size_t k = 0; k < inputs.size();)
        if (inputs[k] == nm) inputs.erase(inputs.begin() + (long)k);
        else ++k;
    for (const Node &n : nodes)  // Constant nodes are tensors too
      if (n.op == "Constant" && n.value && !n.out.empty()) tensors[n.out[0]] = n.value;
  }

  Node parse_node(Span s) {
    Node n;
    Field f;
    for (Reader r(s, path); r.next(f);) {
      if (f.wire != 2) continue;
      if (f.num == 1) n.in.emplace_back((const char *)f.s.p, f.s.n);
      else if (f.num == 2) n.out.emplace_back((const char *)f.s.p, f.s.n);
      else if (f.num == 3) n.name.assign((const char *)f.s.p, f.s.n);
      else if (f.num == 4) n.op.assign((const char *)f.s.p, f.s.n);
      else if (f.num == 5) {  // AttributeProto: name 1, f 2, i 3, t 5
        std::string an;
        Field g;
        for (Reader ar(f.s, path); ar.next(g);) {
          if (g.num == 1 && g.wire == 2) an.assign((const char *)g.s.p, g.s.n);
          else if (g.num == 2 && g.wire == 5) {
            float x;
            const uint32_t u = (uint32_t)g.v;
            std::memcpy(&x, &u, 4);
            n.num[an] = x;
          } else if (g.num == 3 && g.wire == 0) n.num[an] = (double)(int64_t)g.v;
          else if (g.num == 5 && g.wire == 2) n.value = parse_tensor(g.s, path, nullptr);
        }
      }
    }
    return n;
  }

  TensorP cst(const std::string &name) const {
    auto it = tensors.find(name);
    return it == tensors.end() ? nullptr : it->second;
  }

  // LSTM node whose W has input width `in_width`: W [dirs,4H,in], R [dirs,4H,H], B [dirs,8H]
  void lstm(long in_width, TensorP &W, TensorP &R, TensorP &B) const {
    for (const Node &n : nodes) {
      if (n.op != "LSTM" || n.in.size() < 4) continue;
      TensorP w = cst(n.in[1]);
      if (!w || w->dims.size() != 3 || w->dims[2] != in_width) continue;
      W = w;
      R = cst(n.in[2]);
      B = cst(n.in[3]);
      if (!R || !B) fail(XDTTS_ERR_IO, "%s: LSTM node %s without constant R / B", path.c_str(), n.name.c_str());
      return;
    }
    fail(XDTTS_ERR_IO, "%s: no LSTM node with input width %ld", path.c_str(), in_width);
  }

  // (W [out,in] row-major, bias or empty) of the MatMul / Gemm whose constant operand has that shape
  void linear(long out_f, long in_f, std::vector<float> &W, std::vector<float> &bias) const {
    W.clear();
    bias.clear();
    for (const Node &n : nodes) {
      if (n.op == "MatMul" && n.in.size() >= 2) {
        TensorP c = cst(n.in[1]);
        if (!c || !c->is({in_f, out_f})) continue;
        W.resize((size_t)out_f * in_f);  // stored [in, out]: transpose
        for (long i = 0; i < in_f; ++i)
          for (long o = 0; o < out_f; ++o) W[(size_t)o * in_f + i] = c->data[(size_t)i * out_f + o];
        for (const Node &a : nodes) {
          if (a.op != "Add" || a.in.size() != 2 || n.out.empty()) continue;
          const int k = a.in[0] == n.out[0] ? 1 : (a.in[1] == n.out[0] ? 0 : -1);
          if (k < 0) continue;
          TensorP b = cst(a.in[k]);
          if (b && (long)b->numel() == out_f) bias = b->data;
        }
        return;
      }
      if (n.op == "Gemm" && n.in.size() >= 2) {
        TensorP c = cst(n.in[1]);
        if (!c || c->dims.size() != 2) continue;
        auto tb = n.num.find("transB");
        const bool trans = tb != n.num.end() && tb->second != 0;
        if (trans ? !c->is({out_f, in_f}) : !c->is({in_f, out_f})) continue;
        W.resize((size_t)out_f * in_f);
        if (trans) {
          W = c->data;
        } else {
          for (long i = 0; i < in_f; ++i)
            for (long o = 0; o < out_f; ++o) W[(size_t)o * in_f + i] = c->data[(size_t)i * out_f + o];
        }
        if (n.in.size() > 2) {
          TensorP b = cst(n.in[2]);
          if (b && (long)b->numel() == out_f) bias = b->data;
        }
        return;
      }
    }
    fail(XDTTS_ERR_IO, "%s: no linear layer %ld -> %ld (MatMul / Gemm with a constant operand of that shape)", path.c_str(), in_f, out_f);
  }

  struct ConvLayer {
    TensorP W;
    std::vector<float> b;
    TensorP bn[4];  // weight, bias, running_mean, running_var; null = folded
  };
  // Conv nodes in graph order; `shape` (3 dims) filters by weight shape when non-null
  std::vector<ConvLayer> convs(const long *shape) const {
    std::vector<ConvLayer> out;
    for (const Node &n : nodes) {
      if (n.op != "Conv" || n.in.size() < 2) continue;
      TensorP w = cst(n.in[1]);
      if (!w || w->dims.size() != 3) continue;
      if (shape && !(w->dims[0] == shape[0] && w->dims[1] == shape[1] && w->dims[2] == shape[2])) continue;
      ConvLayer L;
      L.W = w;
      TensorP b = n.in.size() > 2 ? cst(n.in[2]) : nullptr;
      if (b && (long)b->numel() == w->dims[0]) L.b = b->data;
      else L.b.assign((size_t)w->dims[0], 0.f);
      for (const Node &m : nodes) {
        if (m.op != "BatchNormalization" || m.in.size() < 5 || n.out.empty() || m.in[0] != n.out[0]) continue;
        auto e = m.num.find("epsilon");
        const double eps = e == m.num.end() ? 1e-5 : e->second;
        if (std::fabs(eps - 1e-5) > 1e-9) fail(XDTTS_ERR_IO, "%s: BatchNormalization epsilon %g: the library folds with 1e-5", path.c_str(), eps);
        for (int k = 0; k < 4; ++k) {
          L.bn[k] = cst(m.in[1 + k]);
          if (!L.bn[k] || (long)L.bn[k]->numel() != w->dims[0]) fail(XDTTS_ERR_IO, "%s: BatchNormalization after a conv without constant statistics", path.c_str());
        }
      }
      out.push_back(L);
    }
    return out;
  }
};

struct Sink {
  std::vector<float> &blob;
  std::vector<char> seen;
  explicit Sink(std::vector<float> &b) : blob(b), seen(tensor_table().size(), 0) { blob.assign(tensor_total(), 0.f); }
  void put(const std::string &name, const float *src, size_t n) {
    const int i = tensor_index(name.c_str());
    if (i < 0) fail(XDTTS_ERR_IO, "onnx import: unknown canonical tensor %s", name.c_str());
    const TensorInfo &t = tensor_table()[(size_t)i];
    if (t.numel != n) fail(XDTTS_ERR_IO, "onnx import: tensor %s has %zu values, expected %zu", name.c_str(), n, t.numel);
    if (seen[(size_t)i]) fail(XDTTS_ERR_IO, "onnx import: tensor %s found twice", name.c_str());
    for (size_t k = 0; k < n; ++k)
      if (!std::isfinite(src[k])) fail(XDTTS_ERR_IO, "onnx import: tensor %s holds a non-finite value", name.c_str());
    std::memcpy(blob.data() + t.offset, src, n * sizeof(float));
    seen[(size_t)i] = 1;
  }
  void put(const std::string &name, const std::vector<float> &v) { put(name, v.data(), v.size()); }
  void finish() {
    for (size_t i = 0; i < seen.size(); ++i)
      if (!seen[i]) fail(XDTTS_ERR_IO, "onnx import: tensor %s was not found in the graphs", tensor_table()[i].name);
  }
};

// rows of one direction in ONNX LSTM order i,o,f,c -> PyTorch order i,f,g,o
std::vector<float> pt_gates(const float *a, long H, long cols) {
  std::vector<float> out((size_t)4 * H * cols);
  const long src_block[4] = {0, 2, 3, 1};  // destination blocks i,f,g,o come from source blocks i,f,c,o = 0,2,3,1
  for (int g = 0; g < 4; ++g) std::memcpy(out.data() + (size_t)g * H * cols, a + (size_t)src_block[g] * H * cols, sizeof(float) * (size_t)H * cols);
  return out;
}

void put_lstm(Sink &sink, const std::string &prefix, const Graph &g, const TensorP &W, const TensorP &R, const TensorP &B, long d, long H, long in_w) {
  if (!(W->dims.size() == 3 && W->dims[1] == 4 * H && W->dims[2] == in_w && R->dims.size() == 3 && R->dims[1] == 4 * H && R->dims[2] == H &&
        B->dims.size() == 2 && B->dims[1] == 8 * H && W->dims[0] > d && R->dims[0] > d && B->dims[0] > d))
    fail(XDTTS_ERR_IO, "%s: LSTM %s has unexpected W / R / B shapes", g.path.c_str(), prefix.c_str());
  sink.put(prefix + "weight_ih", pt_gates(W->data.data() + (size_t)d * 4 * H * in_w, H, in_w));
  sink.put(prefix + "weight_hh", pt_gates(R->data.data() + (size_t)d * 4 * H * H, H, H));
  sink.put(prefix + "bias_ih", pt_gates(B->data.data() + (size_t)d * 8 * H, H, 1));
  sink.put(prefix + "bias_hh", pt_gates(B->data.data() + (size_t)d * 8 * H + 4 * H, H, 1));
}

void put_conv(Sink &sink, const std::string &prefix, const Graph::ConvLayer &L) {
  const size_t co = (size_t)L.W->dims[0];
  sink.put(prefix + ".conv.weight", L.W->data);
  sink.put(prefix + ".conv.bias", L.b);
  static const char *nm[4] = {".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var"};
  if (L.bn[0]) {
    for (int k = 0; k < 4; ++k) sink.put(prefix + nm[k], L.bn[k]->data);
  } else {  // folded by the exporter: identity statistics (scale = 1 / sqrt(var + 1e-5) = 1)
    const float idv[4] = {1.0f, 0.0f, 0.0f, 1.0f - 1e-5f};
    for (int k = 0; k < 4; ++k) sink.put(prefix + nm[k], std::vector<float>(co, idv[k]));
  }
}

bool exists(const std::string &p) { return (bool)std::ifstream(p, std::ios::binary); }

std::string join(const std::vector<std::string> &v) {
  std::string o;
  for (size_t i = 0; i < v.size(); ++i) o += (i ? "," : "") + v[i];
  return o;
}
bool has(const std::vector<std::string> &v, const char *name) {
  for (const std::string &x : v)
    if (x == name) return true;
  return false;
}

// The reference binds decoder_iter.onnx's tensors BY NAME (ort `inputs!["decoder_input" => ..]`, src/tacotron2/mod.rs:284-296;
// outputs indexed by name at :306-307 and :332-339), postnet.onnx's output by name (:349) with one positional input (:347), and
// encoder.onnx positionally: two inputs, three outputs (:379-385).  A model directory the reference itself could not run is
// refused here too, naming what is missing.
const char *const DEC_INPUTS[] = {"decoder_input", "attention_hidden", "attention_cell", "decoder_hidden", "decoder_cell", "attention_weights",
                                  "attention_weights_cum", "attention_context", "memory", "processed_memory", "mask"};
const char *const DEC_OUTPUTS[] = {"decoder_output", "gate_prediction", "out_attention_hidden", "out_attention_cell", "out_decoder_hidden",
                                   "out_decoder_cell", "out_attention_weights", "out_attention_weights_cum", "out_attention_context"};
void check_io(const Graph &enc, const Graph &dec, const Graph &post) {
  if (enc.inputs.size() != 2 || enc.outputs.size() != 3)
    fail(XDTTS_ERR_IO, "%s: the reference feeds 2 inputs and reads 3 outputs (mod.rs:379-385); the graph has inputs [%s] outputs [%s]",
         enc.path.c_str(), join(enc.inputs).c_str(), join(enc.outputs).c_str());
  for (const char *n : DEC_INPUTS)
    if (!has(dec.inputs, n))
      fail(XDTTS_ERR_IO, "%s: no graph input named %s (the reference binds it by name, mod.rs:284-296); inputs are [%s]", dec.path.c_str(), n,
           join(dec.inputs).c_str());
  if (dec.inputs.size() != sizeof DEC_INPUTS / sizeof *DEC_INPUTS)
    fail(XDTTS_ERR_IO, "%s: %zu graph inputs, the reference feeds 11 (mod.rs:284-296): [%s]", dec.path.c_str(), dec.inputs.size(), join(dec.inputs).c_str());
  for (const char *n : DEC_OUTPUTS)
    if (!has(dec.outputs, n))
      fail(XDTTS_ERR_IO, "%s: no graph output named %s (mod.rs:306-307,332-339); outputs are [%s]", dec.path.c_str(), n, join(dec.outputs).c_str());
  if (post.inputs.size() != 1 || !has(post.outputs, "mel_outputs_postnet"))
    fail(XDTTS_ERR_IO, "%s: the reference feeds one input and reads `mel_outputs_postnet` (mod.rs:347-349); the graph has inputs [%s] outputs [%s]",
         post.path.c_str(), join(post.inputs).c_str(), join(post.outputs).c_str());
}

}  // namespace

bool onnx_model_dir(const std::string &dir) {
  return exists(dir + "/encoder.onnx") && exists(dir + "/decoder_iter.onnx") && exists(dir + "/postnet.onnx");
}

void load_onnx_dir(const std::string &dir, std::vector<float> &blob) {
  const Graph enc(dir + "/encoder.onnx"), dec(dir + "/decoder_iter.onnx"), post(dir + "/postnet.onnx");
  check_io(enc, dec, post);
  Sink sink(blob);
  std::vector<float> W, b;
  // encoder.onnx (mod.rs:246-249)
  {
    TensorP emb;
    for (const Node &n : enc.nodes)
      if (n.op == "Gather" && !n.in.empty()) {
        TensorP t = enc.cst(n.in[0]);
        if (t && t->is({N_SYMBOLS, EMB})) emb = t;
      }
    if (!emb) fail(XDTTS_ERR_IO, "%s: no Gather over a [148, 512] embedding table", enc.path.c_str());
    sink.put("embedding.weight", emb->data);
    const long shp[3] = {EMB, EMB, ENC_K};
    const auto ec = enc.convs(shp);
    if (ec.size() != ENC_CONVS) fail(XDTTS_ERR_IO, "%s: expected 3 conv layers [512,512,5], found %zu", enc.path.c_str(), ec.size());
    for (int i = 0; i < ENC_CONVS; ++i) put_conv(sink, "encoder.convolutions." + std::to_string(i), ec[(size_t)i]);
    TensorP Wl, R, B;
    enc.lstm(EMB, Wl, R, B);
    if (Wl->dims[0] != 2) fail(XDTTS_ERR_IO, "%s: the encoder LSTM is not bidirectional", enc.path.c_str());
    put_lstm(sink, "encoder.lstm.fwd.", enc, Wl, R, B, 0, ENC_H, EMB);
    put_lstm(sink, "encoder.lstm.bwd.", enc, Wl, R, B, 1, ENC_H, EMB);
    enc.linear(ATT_DIM, EMB, W, b);
    sink.put("attention.memory_layer.weight", W);
  }
  // decoder_iter.onnx (mod.rs:251-254)
  {
    dec.linear(PRENET, N_MEL, W, b);
    sink.put("prenet.0.weight", W);
    dec.linear(PRENET, PRENET, W, b);
    sink.put("prenet.1.weight", W);
    TensorP Wl, R, B;
    dec.lstm(ATT_IN, Wl, R, B);
    put_lstm(sink, "attention_rnn.", dec, Wl, R, B, 0, ATT_RNN, ATT_IN);
    dec.lstm(DEC_IN, Wl, R, B);
    put_lstm(sink, "decoder_rnn.", dec, Wl, R, B, 0, DEC_RNN, DEC_IN);
    dec.linear(ATT_DIM, ATT_RNN, W, b);
    sink.put("attention.query_layer.weight", W);
    dec.linear(1, ATT_DIM, W, b);
    sink.put("attention.v.weight", W);
    const long lshp[3] = {LOC_F, 2, LOC_K};
    const auto lc = dec.convs(lshp);
    if (lc.size() != 1) fail(XDTTS_ERR_IO, "%s: expected one location conv [32,2,31], found %zu", dec.path.c_str(), lc.size());
    sink.put("attention.location_conv.weight", lc[0].W->data);
    dec.linear(ATT_DIM, LOC_F, W, b);
    sink.put("attention.location_dense.weight", W);
    dec.linear(N_MEL, PROJ_IN, W, b);
    if (b.empty()) fail(XDTTS_ERR_IO, "%s: projection bias not found", dec.path.c_str());
    sink.put("linear_projection.weight", W);
    sink.put("linear_projection.bias", b);
    dec.linear(1, PROJ_IN, W, b);
    if (b.empty()) fail(XDTTS_ERR_IO, "%s: gate bias not found", dec.path.c_str());
    sink.put("gate_layer.weight", W);
    sink.put("gate_layer.bias", b);
  }
  // postnet.onnx (mod.rs:256-259)
  {
    const auto pc = post.convs(nullptr);
    if (pc.size() != POST_CONVS) fail(XDTTS_ERR_IO, "%s: expected 5 conv layers, found %zu", post.path.c_str(), pc.size());
    for (int i = 0; i < POST_CONVS; ++i) put_conv(sink, "postnet.convolutions." + std::to_string(i), pc[(size_t)i]);
  }
  sink.finish();
}

// "file: inputs a,b,... ; outputs x,y,...\n" per graph -- what a maintainer (and the tests) compare with the names the
// reference binds
std::string describe_onnx_dir(const std::string &dir) {
  std::string out;
  for (const char *f : {"encoder.onnx", "decoder_iter.onnx", "postnet.onnx"}) {
    const Graph g(dir + "/" + f);
    out += std::string(f) + ": inputs " + join(g.inputs) + " ; outputs " + join(g.outputs) + "\n";
  }
  return out;
}

}  // namespace xdtts
