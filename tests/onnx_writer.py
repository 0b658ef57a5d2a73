"""Test helper: a tiny ONNX (protobuf) writer, enough to lay the Tacotron2 weights out the way
torch.onnx.export does -- used only to exercise tools/onnx_to_xdtw.py (no `onnx` package here)."""
import struct

import numpy as np


def vint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def key(f, w):
    return vint((f << 3) | w)


def ld(f, payload):
    return key(f, 2) + vint(len(payload)) + payload


def tensor(name, a, how="raw"):
    a = np.ascontiguousarray(a, dtype="<f4")
    b = b"".join(key(1, 0) + vint(d) for d in a.shape) + key(2, 0) + vint(1) + ld(8, name.encode())
    if how == "raw":
        b += ld(9, a.tobytes())
    elif how == "packed":
        b += ld(4, a.tobytes())
    else:  # one fixed32 per element
        b += b"".join(key(4, 5) + struct.pack("<f", v) for v in a.ravel())
    return b


def attr_i(name, v):
    return ld(5, ld(1, name.encode()) + key(3, 0) + vint(v) + key(20, 0) + vint(2))


def attr_f(name, v):
    return ld(5, ld(1, name.encode()) + key(2, 5) + struct.pack("<f", v) + key(20, 0) + vint(1))


def attr_t(name, t):
    return ld(5, ld(1, name.encode()) + ld(5, t) + key(20, 0) + vint(4))


def node(op, inputs, outputs, attrs=b"", name=""):
    return ld(1, b"".join(ld(1, i.encode()) for i in inputs) + b"".join(ld(2, o.encode()) for o in outputs) +
              ld(3, name.encode()) + ld(4, op.encode()) + attrs)


def model(nodes, inits, inputs=(), outputs=(), inits_as_inputs=()):
    """inputs / outputs: graph value names (GraphProto.input = 11 / .output = 12, ValueInfoProto.name = 1);
    inits_as_inputs: initialiser names listed among the inputs too, as IR version < 4 exporters do."""
    graph = b"".join(nodes) + ld(2, b"g") + b"".join(ld(5, t) for t in inits)
    graph += b"".join(ld(11, ld(1, n.encode())) for n in list(inputs) + list(inits_as_inputs))
    graph += b"".join(ld(12, ld(1, n.encode())) for n in outputs)
    return key(1, 0) + vint(7) + ld(2, b"pytorch") + ld(7, graph)


# the names the reference binds (src/tacotron2/mod.rs:284-296,306-307,332-339,349; encoder positional, :379-385)
ENC_IO = (["sequences", "sequence_lengths"], ["memory", "processed_memory", "lens"])
DEC_IO = (["decoder_input", "attention_hidden", "attention_cell", "decoder_hidden", "decoder_cell", "attention_weights",
           "attention_weights_cum", "attention_context", "memory", "processed_memory", "mask"],
          ["decoder_output", "gate_prediction", "out_attention_hidden", "out_attention_cell", "out_decoder_hidden", "out_decoder_cell",
           "out_attention_weights", "out_attention_weights_cum", "out_attention_context"])
POST_IO = (["mel_outputs"], ["mel_outputs_postnet"])


def onnx_gates(a, H):
    """PyTorch gate rows i,f,g,o -> ONNX LSTM order i,o,f,c"""
    i, f, g, o = a[0:H], a[H:2 * H], a[2 * H:3 * H], a[3 * H:4 * H]
    return np.concatenate([i, o, f, g], axis=0)


def pack_lstm(dirs, H):
    """dirs: list of dicts with weight_ih, weight_hh, bias_ih, bias_hh (PyTorch order) -> W, R, B"""
    W = np.stack([onnx_gates(d["weight_ih"], H) for d in dirs])
    R = np.stack([onnx_gates(d["weight_hh"], H) for d in dirs])
    B = np.stack([np.concatenate([onnx_gates(d["bias_ih"], H), onnx_gates(d["bias_hh"], H)]) for d in dirs])
    return W, R, B


def write_models(path, T, style, dec_io=DEC_IO):
    """T: dict canonical name -> array (tools/onnx_to_xdtw.tensor_table names).  style 'folded': anonymous
    constants, MatMul with transposed weights, conv+BN pre-folded (BN tensors of T are ignored);
    style 'named': parameter names kept, Gemm transB=1, BatchNormalization nodes, mixed data encodings."""
    import os

    folded = style == "folded"
    cnt = [0]

    def nm(kind, pretty):
        cnt[0] += 1
        return ("onnx::%s_%d" % (kind, cnt[0])) if folded else pretty

    def lin(nodes, inits, x, y, W, bias, pretty):
        if folded:
            w = nm("MatMul", pretty)
            inits.append(tensor(w, W.T))
            if bias is None:
                nodes.append(node("MatMul", [x, w], [y]))
            else:
                b = nm("Add", pretty + ".bias")
                inits.append(tensor(b, bias))
                nodes.append(node("MatMul", [x, w], [y + "_mm"]))
                nodes.append(node("Add", [b, y + "_mm"], [y]))
        else:
            inits.append(tensor(pretty + ".weight", W, "packed"))
            ins = [x, pretty + ".weight"]
            if bias is not None:
                inits.append(tensor(pretty + ".bias", bias, "fixed"))
                ins.append(pretty + ".bias")
            nodes.append(node("Gemm", ins, [y], attr_i("transB", 1)))

    def conv(nodes, inits, x, y, prefix, pretty, with_bias=True):
        w = nm("Conv", pretty + ".conv.weight")
        inits.append(tensor(w, T[prefix + ".conv.weight"] if prefix + ".conv.weight" in T else T[prefix]))
        ins = [x, w]
        if with_bias:
            b = nm("Conv", pretty + ".conv.bias")
            inits.append(tensor(b, T[prefix + ".conv.bias"]))
            ins.append(b)
        if folded or not with_bias:
            nodes.append(node("Conv", ins, [y]))
            return
        nodes.append(node("Conv", ins, [y + "_c"]))
        names = []
        for k in ("weight", "bias", "running_mean", "running_var"):
            names.append(pretty + ".bn." + k)
            inits.append(tensor(names[-1], T[prefix + ".bn." + k]))
        nodes.append(node("BatchNormalization", [y + "_c"] + names, [y], attr_f("epsilon", 1e-5)))

    def lstm(nodes, inits, x, y, dirs, H, pretty):
        W, R, B = pack_lstm(dirs, H)
        names = [nm("LSTM", pretty + k) for k in (".W", ".R", ".B")]
        for n_, a in zip(names, (W, R, B)):
            inits.append(tensor(n_, a))
        nodes.append(node("LSTM", [x] + names, [y], attr_i("hidden_size", H)))

    # encoder.onnx
    nodes, inits = [], []
    e = nm("Gather", "embedding.weight")
    if folded:
        inits.append(tensor(e, T["embedding.weight"]))
    else:  # a Constant node instead of an initializer
        nodes.append(node("Constant", [], [e], attr_t("value", tensor("", T["embedding.weight"]))))
    nodes.append(node("Gather", [e, "sequences"], ["emb"]))
    x = "emb"
    for i in range(3):
        conv(nodes, inits, x, "c%d" % i, "encoder.convolutions.%d" % i, "encoder.convolutions.%d" % i)
        x = "c%d" % i
    dirs = [{k: T["encoder.lstm.%s.%s" % (d, k)] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")} for d in ("fwd", "bwd")]
    lstm(nodes, inits, x, "memory", dirs, 256, "encoder.lstm")
    lin(nodes, inits, "memory", "processed_memory", T["attention.memory_layer.weight"], None, "decoder.attention_layer.memory_layer")
    open(os.path.join(path, "encoder.onnx"), "wb").write(model(nodes, inits, *ENC_IO))

    # decoder_iter.onnx
    nodes, inits = [], []
    lin(nodes, inits, "decoder_input", "p0", T["prenet.0.weight"], None, "decoder.prenet.layers.0")
    lin(nodes, inits, "p0", "p1", T["prenet.1.weight"], None, "decoder.prenet.layers.1")
    for nm_, y in (("decoder_rnn", "hd"), ("attention_rnn", "ha")):  # graph order must not matter
        d = {k: T["%s.%s" % (nm_, k)] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
        lstm(nodes, inits, "x_" + nm_, y, [d], 1024, "decoder." + nm_)
    lin(nodes, inits, "ha", "q", T["attention.query_layer.weight"], None, "decoder.attention_layer.query_layer")
    conv(nodes, inits, "aw", "lc", "attention.location_conv.weight", "decoder.attention_layer.location_layer.location_conv", with_bias=False)
    lin(nodes, inits, "lc", "ld", T["attention.location_dense.weight"], None, "decoder.attention_layer.location_layer.location_dense")
    lin(nodes, inits, "t", "e", T["attention.v.weight"].reshape(1, 128), None, "decoder.attention_layer.v")
    lin(nodes, inits, "hc", "decoder_output", T["linear_projection.weight"], T["linear_projection.bias"], "decoder.linear_projection")
    lin(nodes, inits, "hc", "gate_prediction", T["gate_layer.weight"].reshape(1, 1536), T["gate_layer.bias"], "decoder.gate_layer")
    # (the 'named' style also lists one initialiser among the graph inputs, as pre-IR-4 exporters do: it is not an input)
    open(os.path.join(path, "decoder_iter.onnx"), "wb").write(model(nodes, inits, *dec_io, inits_as_inputs=[] if folded else ["decoder.gate_layer.weight"]))

    # postnet.onnx
    nodes, inits = [], []
    x = "mel"
    for i in range(5):
        conv(nodes, inits, x, "pc%d" % i, "postnet.convolutions.%d" % i, "postnet.convolutions.%d" % i)
        x = "pc%d" % i
    open(os.path.join(path, "postnet.onnx"), "wb").write(model(nodes, inits, *POST_IO))
