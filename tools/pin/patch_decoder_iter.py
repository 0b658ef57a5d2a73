#!/usr/bin/env python3
"""Pin kit, step 1: take the randomness out of `decoder_iter.onnx` so that ONE run of the reference's graph can be recorded
and compared (the reference runs the graph at /root/reference/src/tacotron2/mod.rs:304; its prenet dropout is live at
inference -- SURVEY.md section 8(a) D1, "to re-verify" in 8(c)).

    python tools/pin/patch_decoder_iter.py models/tacotron2/decoder_iter.onnx decoder_iter.pinned.onnx

Pure Python (no `onnx` package): a protobuf wire-format reader / writer that rewrites the GraphProto only.
Every random node is replaced by a new graph INPUT, so the caller supplies the randomness and records it:

  Dropout(x, ratio, training_mode)        -> Mul(x, dropout_scale_k)        input: keep / (1 - ratio), i.e. 0 or 2
        (what torch's exporter of today writes for F.dropout(training=True))
  RandomUniformLike / RandomUniform       -> node removed, its output IS the input dropout_uniform_k: u in [0, 1)
        (torch.bernoulli(p) exports as Less(RandomUniformLike(p), p): keep = u < p -- NVIDIA's inference prenet)
  Bernoulli / RandomNormal / RandomNormalLike / Multinomial -> node removed, output becomes the input random_k

Prints one JSON line: {"random_nodes": [...], "new_inputs": [{"name", "kind", "shape", "keep_rule"}]} -- an empty list means
the exported graph draws nothing (dropout was folded away): then `dropout_mode = 0` is the reference's behaviour and the
record script needs no masks.  The shapes default to [1, 256] (the prenet width, batch 1: the reference's call, mod.rs:285).
"""
import json
import struct
import sys

RANDOM_OPS = {"RandomUniform", "RandomUniformLike", "RandomNormal", "RandomNormalLike", "Bernoulli", "Multinomial"}


# ---- protobuf wire format ---------------------------------------------------------------------------------------------
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
    """(field, wire type, value) of one message; value is int (varint) or bytes (length-delimited / fixed)."""
    i, n, out = 0, len(b), []
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v, i = bytes(b[i:i + 8]), i + 8
        elif w == 2:
            ln, i = _varint(b, i)
            v, i = bytes(b[i:i + ln]), i + ln
        elif w == 5:
            v, i = bytes(b[i:i + 4]), i + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % w)
        out.append((f, w, v))
    return out


def vint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def emit(f, w, v):
    if w == 0:
        return vint((f << 3) | 0) + vint(v)
    if w == 2:
        return vint((f << 3) | 2) + vint(len(v)) + v
    return vint((f << 3) | w) + v


def msg(items):
    return b"".join(emit(f, w, v) for f, w, v in items)


def s(f, text):
    return (f, 2, text.encode())


# ---- ONNX pieces ------------------------------------------------------------------------------------------------------
def parse_node(buf):
    n = {"inputs": [], "outputs": [], "name": "", "op": "", "attrs": {}}
    for f, w, v in fields(buf):
        if f == 1:
            n["inputs"].append(v.decode())
        elif f == 2:
            n["outputs"].append(v.decode())
        elif f == 3:
            n["name"] = v.decode()
        elif f == 4:
            n["op"] = v.decode()
        elif f == 5:
            an, ints, fl = "", [], None
            for af, aw, av in fields(v):
                if af == 1:
                    an = av.decode()
                elif af == 8:   # ints
                    ints += [x for x in ([av] if aw == 0 else _packed(av))]
                elif af == 2 and aw == 5:
                    fl = struct.unpack("<f", av)[0]
            n["attrs"][an] = ints if ints else fl
    return n


def _packed(v):
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(x)
    return out


def value_info(name, shape, elem_type=1):
    """ValueInfoProto{name, type{tensor_type{elem_type FLOAT, shape{dim{dim_value}}}}}"""
    dims = b"".join(emit(1, 2, emit(1, 0, d)) for d in shape)
    tensor_type = emit(1, 0, elem_type) + emit(2, 2, dims)
    return msg([s(1, name), (2, 2, emit(1, 2, tensor_type))])


def mul_node(x, y, out, name):
    return msg([s(1, x), s(1, y), s(2, out), s(3, name), s(4, "Mul")])


def ratio_of(graph_items, const_name):
    """the `ratio` input of a Dropout node, when it is a Constant node / initializer holding one float (else 0.5)"""
    for f, w, v in graph_items:
        if f == 1:
            n = parse_node(v)
            if n["op"] == "Constant" and const_name in n["outputs"]:
                for nf, nw, nv in fields(v):
                    if nf == 5:
                        for af, aw, av in fields(nv):
                            if af == 5 and aw == 2:   # AttributeProto.t
                                raw, floats = None, []
                                for tf, tw, tv in fields(av):
                                    if tf == 9:
                                        raw = tv
                                    elif tf == 4:
                                        floats.append(tv)
                                if raw and len(raw) >= 4:
                                    return struct.unpack("<f", raw[:4])[0]
                                if floats:
                                    return struct.unpack("<f", floats[0][:4])[0]
    return 0.5


def patch(model_bytes, shape=(1, 256)):
    top = fields(model_bytes)
    gi = [i for i, (f, w, v) in enumerate(top) if f == 7 and w == 2]
    if len(gi) != 1:
        raise ValueError("not an ONNX ModelProto (graph field missing)")
    graph = fields(top[gi[0]][2])
    new_graph, new_inputs, found = [], [], []
    k_scale = k_uni = k_rand = 0
    for f, w, v in graph:
        if f != 1:
            new_graph.append((f, w, v))
            continue
        n = parse_node(v)
        if n["op"] == "Dropout" and len(n["inputs"]) >= 3:   # opset >= 12 with a training_mode input
            name = "dropout_scale_%d" % k_scale
            k_scale += 1
            ratio = ratio_of(graph, n["inputs"][1])
            new_graph.append((1, 2, mul_node(n["inputs"][0], name, n["outputs"][0], (n["name"] or "Dropout") + "_pinned")))
            new_inputs.append({"name": name, "kind": "scale", "shape": list(shape), "ratio": ratio,
                               "keep_rule": "value = keep ? 1 / (1 - ratio) : 0   (ratio %.3g: 0 or %.3g)" % (ratio, 1.0 / (1.0 - ratio))})
            found.append({"op": n["op"], "name": n["name"], "outputs": n["outputs"], "mask_output_used": len(n["outputs"]) > 1 and any(
                n["outputs"][1] in parse_node(v2)["inputs"] for f2, w2, v2 in graph if f2 == 1)})
        elif n["op"] in RANDOM_OPS:
            kind = "uniform" if n["op"].startswith("RandomUniform") else "random"
            if kind == "uniform":
                name = "dropout_uniform_%d" % k_uni
                k_uni += 1
            else:
                name = "random_%d" % k_rand
                k_rand += 1
            shp = [int(x) for x in n["attrs"].get("shape", [])] or list(shape)
            # consumers keep reading the old value name: declare THAT name's producer as the new input through an Identity
            new_graph.append((1, 2, msg([s(1, name), s(2, n["outputs"][0]), s(3, (n["name"] or n["op"]) + "_pinned"), s(4, "Identity")])))
            new_inputs.append({"name": name, "kind": kind, "shape": shp,
                               "keep_rule": "u in [0, 1): the graph compares it itself (Less(u, p): keep = u < p)" if kind == "uniform" else "as the removed %s node would have drawn" % n["op"]})
            found.append({"op": n["op"], "name": n["name"], "outputs": n["outputs"]})
        else:
            new_graph.append((f, w, v))
    for ni in new_inputs:
        new_graph.append((11, 2, value_info(ni["name"], ni["shape"])))
    top[gi[0]] = (7, 2, msg(new_graph))
    return msg(top), {"random_nodes": found, "new_inputs": new_inputs}


def main(argv):
    if len(argv) < 3:
        print(__doc__)
        return 2
    data = open(argv[1], "rb").read()
    if len(data) < 1024 and data.startswith(b"version https://git-lfs"):
        print("%s is a git-LFS pointer, not the model: fetch the artefact first (git lfs pull)" % argv[1], file=sys.stderr)
        return 1
    out, report = patch(data)
    open(argv[2], "wb").write(out)
    with open(argv[2] + ".json", "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
