#!/usr/bin/env python3
"""Convert the reference's three Tacotron2 ONNX graphs (models/tacotron2/{encoder,decoder_iter,postnet}.onnx,
loaded at src/tacotron2/mod.rs:246-259) into the flat weight container `tacotron2.xdtw` that
`xdtts_tacotron2_load(dir)` reads.  SURVEY section 8(f) rank 1.

Pure Python + numpy: a minimal protobuf wire-format reader (no `onnx` package) and a small
graph-aware resolver, because an exporter with constant folding drops most parameter names:
  * LSTM nodes (the encoder BiLSTM, and nn.LSTMCell exports as a 1-step LSTM) carry packed
    W [dirs, 4H, in], R [dirs, 4H, H], B [dirs, 8H] in ONNX gate order i,o,f,c -> split and re-ordered
    to PyTorch's i,f,g,o; the node is identified by W's input width (512 encoder, 768 attention_rnn,
    1536 decoder_rnn);
  * Conv nodes are taken in graph order (encoder 3 x [512,512,5]; postnet 5; location conv [32,2,31]);
    a BatchNormalization node after a conv supplies its statistics, otherwise the conv is taken as
    already folded and the BN is written as the identity;
  * Linear layers are identified by their (out, in) shape, oriented by how the constant is consumed
    (MatMul(x, W^T) vs Gemm(..., transB)), biases from Gemm's C or the Add that follows a MatMul;
  * the embedding is the Gather table [148, 512].
UNVERIFIED AGAINST THE REAL FILES: the checkout holds git-LFS pointers only; the conventions above are
those of torch.onnx.export for the NVIDIA model and are exercised by tests/test_onnx_import_cpu.py on
synthetic graphs written both ways (named parameters / folded anonymous constants).

usage: onnx_to_xdtw.py MODEL_DIR [OUT_DIR]
"""
import os
import struct
import sys

import numpy as np

# ---- protobuf wire format ---------------------------------------------------------------------


def _varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if c < 0x80:
            return r, i
        s += 7


def fields(b):
    """Yields (field number, wire type, value) of one message; value is int (varint / fixed) or bytes."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v = b[i:i + 8]
            i += 8
        elif w == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif w == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % w)
        yield f, w, v


def _packed_varints(v):
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(x)
    return out


def _signed(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def parse_tensor(buf):
    """TensorProto -> (name, float32 array or None)."""
    dims, dtype, name, raw, floats, ints = [], 0, "", None, [], []
    for f, w, v in fields(buf):
        if f == 1:
            dims += _packed_varints(v) if w == 2 else [v]
        elif f == 2:
            dtype = v
        elif f == 4:
            floats.append(np.frombuffer(v, "<f4") if w == 2 else np.frombuffer(v, "<f4", 1))
        elif f == 7:
            ints += [_signed(x) for x in (_packed_varints(v) if w == 2 else [v])]
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 13 or f == 14:
            raise ValueError("tensor %s uses external data: export with the weights embedded" % name)
    dims = [_signed(d) for d in dims]
    if dtype == 1:
        a = np.frombuffer(raw, "<f4") if raw is not None else (np.concatenate(floats) if floats else np.zeros(0, "<f4"))
        return name, a.astype(np.float32).reshape(dims)
    if dtype == 7:
        a = np.frombuffer(raw, "<i8") if raw is not None else np.asarray(ints, np.int64)
        return name, None  # integer constants (shapes, axes) are not weights
    return name, None


class Node:
    def __init__(self, buf):
        self.inputs, self.outputs, self.name, self.op, self.attrs = [], [], "", "", {}
        for f, w, v in fields(buf):
            if f == 1:
                self.inputs.append(bytes(v).decode())
            elif f == 2:
                self.outputs.append(bytes(v).decode())
            elif f == 3:
                self.name = bytes(v).decode()
            elif f == 4:
                self.op = bytes(v).decode()
            elif f == 5:
                an, av = "", None
                for g, gw, gv in fields(v):
                    if g == 1:
                        an = bytes(gv).decode()
                    elif g == 2:
                        av = struct.unpack("<f", gv)[0]
                    elif g == 3:
                        av = _signed(gv)
                    elif g == 5:
                        av = parse_tensor(gv)[1]
                self.attrs[an] = av


class Graph:
    def __init__(self, path):
        data = open(path, "rb").read()
        if data[:40].startswith(b"version https://git-lfs"):
            raise ValueError("%s is a git-LFS pointer (%d bytes): fetch the real file first (git lfs pull)" % (path, len(data)))
        self.path, self.nodes, self.tensors = path, [], {}
        graph = None
        for f, w, v in fields(data):
            if f == 7:
                graph = v
        if graph is None:
            raise ValueError("%s: no GraphProto in the model" % path)
        for f, w, v in fields(graph):
            if f == 1:
                self.nodes.append(Node(v))
            elif f == 5:
                name, a = parse_tensor(v)
                if a is not None:
                    self.tensors[name] = a
        for n in self.nodes:  # Constant nodes are tensors too
            if n.op == "Constant" and n.attrs.get("value") is not None and n.outputs:
                self.tensors[n.outputs[0]] = n.attrs["value"]

    def const(self, name):
        return self.tensors.get(name)

    def of(self, op):
        return [n for n in self.nodes if n.op == op]

    def lstm(self, in_width):
        for n in self.of("LSTM"):
            W = self.const(n.inputs[1])
            if W is not None and W.shape[2] == in_width:
                R, B = self.const(n.inputs[2]), self.const(n.inputs[3]) if len(n.inputs) > 3 else None
                if R is None or B is None:
                    raise ValueError("%s: LSTM %s without constant R/B" % (self.path, n.name))
                return W, R, B
        raise ValueError("%s: no LSTM node with input width %d" % (self.path, in_width))

    def linear(self, out_f, in_f):
        """(W [out, in], bias or None) of the MatMul / Gemm whose constant operand has that shape."""
        for n in self.nodes:
            if n.op == "MatMul":
                Wc = self.const(n.inputs[1])
                if Wc is not None and Wc.ndim == 2 and Wc.shape == (in_f, out_f):
                    bias = None
                    for a in self.of("Add"):
                        if n.outputs[0] in a.inputs:
                            other = [x for x in a.inputs if x != n.outputs[0]]
                            c = self.const(other[0]) if other else None
                            if c is not None and c.size == out_f:
                                bias = c.reshape(out_f)
                    return np.ascontiguousarray(Wc.T), bias
            elif n.op == "Gemm":
                Bc = self.const(n.inputs[1])
                if Bc is None or Bc.ndim != 2:
                    continue
                Wm = Bc if n.attrs.get("transB", 0) else Bc.T
                if Wm.shape == (out_f, in_f):
                    c = self.const(n.inputs[2]) if len(n.inputs) > 2 else None
                    return np.ascontiguousarray(Wm), (c.reshape(out_f) if c is not None and c.size == out_f else None)
        raise ValueError("%s: no linear layer %d -> %d" % (self.path, in_f, out_f))

    def convs(self, shape=None):
        """[(weight, bias, bn or None)] of the Conv nodes in graph order (optionally of one weight shape)."""
        out = []
        for n in self.of("Conv"):
            W = self.const(n.inputs[1])
            if W is None or (shape is not None and W.shape != tuple(shape)):
                continue
            b = self.const(n.inputs[2]) if len(n.inputs) > 2 else None
            bn = None
            for m in self.of("BatchNormalization"):
                if m.inputs[0] == n.outputs[0]:
                    bn = [self.const(x) for x in m.inputs[1:5]]
                    eps = m.attrs.get("epsilon", 1e-5)
                    if abs(eps - 1e-5) > 1e-9:
                        raise ValueError("BatchNormalization epsilon %g: the library folds with 1e-5" % eps)
            out.append((W, b if b is not None else np.zeros(W.shape[0], np.float32), bn))
        return out


# ---- canonical tensor table of libxdtts_hip (csrc/weights.cpp) -----------------------------------

def tensor_table():
    t = [("embedding.weight", (148, 512))]

    def conv(prefix, co, ci):
        return [(prefix + ".conv.weight", (co, ci, 5)), (prefix + ".conv.bias", (co,)), (prefix + ".bn.weight", (co,)),
                (prefix + ".bn.bias", (co,)), (prefix + ".bn.running_mean", (co,)), (prefix + ".bn.running_var", (co,))]

    for i in range(3):
        t += conv("encoder.convolutions.%d" % i, 512, 512)
    for d in ("fwd", "bwd"):
        t += [("encoder.lstm.%s.weight_ih" % d, (1024, 512)), ("encoder.lstm.%s.weight_hh" % d, (1024, 256)),
              ("encoder.lstm.%s.bias_ih" % d, (1024,)), ("encoder.lstm.%s.bias_hh" % d, (1024,))]
    t += [("attention.memory_layer.weight", (128, 512)), ("prenet.0.weight", (256, 80)), ("prenet.1.weight", (256, 256)),
          ("attention_rnn.weight_ih", (4096, 768)), ("attention_rnn.weight_hh", (4096, 1024)),
          ("attention_rnn.bias_ih", (4096,)), ("attention_rnn.bias_hh", (4096,)),
          ("attention.query_layer.weight", (128, 1024)), ("attention.v.weight", (128,)),
          ("attention.location_conv.weight", (32, 2, 31)), ("attention.location_dense.weight", (128, 32)),
          ("decoder_rnn.weight_ih", (4096, 1536)), ("decoder_rnn.weight_hh", (4096, 1024)),
          ("decoder_rnn.bias_ih", (4096,)), ("decoder_rnn.bias_hh", (4096,)),
          ("linear_projection.weight", (80, 1536)), ("linear_projection.bias", (80,)),
          ("gate_layer.weight", (1536,)), ("gate_layer.bias", (1,))]
    for i, (co, ci) in enumerate([(512, 80), (512, 512), (512, 512), (512, 512), (80, 512)]):
        t += conv("postnet.convolutions.%d" % i, co, ci)
    return t


def _pt_gates(a, H):
    """rows in ONNX LSTM order i,o,f,c -> PyTorch order i,f,g,o"""
    i, o, f, c = a[0:H], a[H:2 * H], a[2 * H:3 * H], a[3 * H:4 * H]
    return np.concatenate([i, f, c, o], axis=0)


def _lstm_dir(W, R, B, d, H):
    return {"weight_ih": _pt_gates(W[d], H), "weight_hh": _pt_gates(R[d], H),
            "bias_ih": _pt_gates(B[d][:4 * H], H), "bias_hh": _pt_gates(B[d][4 * H:], H)}


def _put_conv(out, prefix, W, b, bn):
    co = W.shape[0]
    out[prefix + ".conv.weight"], out[prefix + ".conv.bias"] = W, b
    if bn is None:  # folded by the exporter: identity statistics (scale = 1 / sqrt(var + 1e-5) = 1)
        bn = [np.ones(co, np.float32), np.zeros(co, np.float32), np.zeros(co, np.float32), np.full(co, 1.0 - 1e-5, np.float32)]
    for k, v in zip(("weight", "bias", "running_mean", "running_var"), bn):
        out[prefix + ".bn." + k] = v


def collect(model_dir):
    enc, dec, post = (Graph(os.path.join(model_dir, f + ".onnx")) for f in ("encoder", "decoder_iter", "postnet"))
    out = {}
    emb = [enc.const(n.inputs[0]) for n in enc.of("Gather") if enc.const(n.inputs[0]) is not None and enc.const(n.inputs[0]).shape == (148, 512)]
    if not emb:
        raise ValueError("encoder.onnx: no Gather over a [148, 512] table")
    out["embedding.weight"] = emb[0]
    ec = enc.convs((512, 512, 5))
    if len(ec) != 3:
        raise ValueError("encoder.onnx: expected 3 conv layers [512,512,5], found %d" % len(ec))
    for i, (W, b, bn) in enumerate(ec):
        _put_conv(out, "encoder.convolutions.%d" % i, W, b, bn)
    W, R, B = enc.lstm(512)
    if W.shape[0] != 2:
        raise ValueError("encoder.onnx: the encoder LSTM is not bidirectional")
    for d, nm in enumerate(("fwd", "bwd")):
        for k, v in _lstm_dir(W, R, B, d, 256).items():
            out["encoder.lstm.%s.%s" % (nm, k)] = v
    out["attention.memory_layer.weight"] = enc.linear(128, 512)[0]
    out["prenet.0.weight"] = dec.linear(256, 80)[0]
    out["prenet.1.weight"] = dec.linear(256, 256)[0]
    for nm, width in (("attention_rnn", 768), ("decoder_rnn", 1536)):
        W, R, B = dec.lstm(width)
        for k, v in _lstm_dir(W, R, B, 0, 1024).items():
            out["%s.%s" % (nm, k)] = v
    out["attention.query_layer.weight"] = dec.linear(128, 1024)[0]
    out["attention.v.weight"] = dec.linear(1, 128)[0].reshape(128)
    lc = dec.convs((32, 2, 31))
    if len(lc) != 1:
        raise ValueError("decoder_iter.onnx: expected one location conv [32,2,31]")
    out["attention.location_conv.weight"] = lc[0][0]
    out["attention.location_dense.weight"] = dec.linear(128, 32)[0]
    out["linear_projection.weight"], pb = dec.linear(80, 1536)
    gw, gb = dec.linear(1, 1536)
    if pb is None or gb is None:
        raise ValueError("decoder_iter.onnx: projection / gate bias not found")
    out["linear_projection.bias"], out["gate_layer.weight"], out["gate_layer.bias"] = pb, gw.reshape(1536), gb.reshape(1)
    pc = post.convs()
    if len(pc) != 5:
        raise ValueError("postnet.onnx: expected 5 conv layers, found %d" % len(pc))
    for i, (W, b, bn) in enumerate(pc):
        _put_conv(out, "postnet.convolutions.%d" % i, W, b, bn)
    return out


def write_container(out_dir, tensors):
    tab = tensor_table()
    entries, blobs, off = [], [], 0
    for name, shape in tab:
        a = np.ascontiguousarray(tensors[name], dtype="<f4")
        if a.shape != tuple(shape):
            raise ValueError("tensor %s has shape %s, expected %s" % (name, a.shape, shape))
        dims = list(shape) + [0] * (3 - len(shape))
        entries.append(struct.pack("<64sI3IQQ", name.encode(), len(shape), *dims, off, a.size))
        blobs.append(a.tobytes())
        off += a.size
    path = os.path.join(out_dir, "tacotron2.xdtw")
    with open(path, "wb") as f:
        f.write(b"XDTW0001" + struct.pack("<I", len(tab)) + b"".join(entries) + b"".join(blobs))
    return path, off


def main(argv):
    if len(argv) < 2:
        raise SystemExit(__doc__)
    model_dir = argv[1]
    out_dir = argv[2] if len(argv) > 2 else model_dir
    path, n = write_container(out_dir, collect(model_dir))
    print("%s: %d tensors, %d floats" % (path, len(tensor_table()), n))


if __name__ == "__main__":
    main(sys.argv)
