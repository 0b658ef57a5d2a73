"""An INDEPENDENT writer of the reference's model directory.

The reference loads `encoder.onnx`, `decoder_iter.onnx`, `postnet.onnx` (src/tacotron2/mod.rs:246-259), produced by NVIDIA's
`export_tacotron2_onnx.py` from the PyTorch Tacotron2 (mod.rs:137-138).  The checkout holds git-LFS pointers, so until now
`csrc/onnx_load.cpp` had only ever read files laid out by this repo's own `tests/onnx_writer.py`.  Here the three graphs are
written by torch's own TorchScript ONNX exporter -- the exporter family NVIDIA's script used -- from torch modules structured
like NVIDIA's model (ConvNorm + BatchNorm1d stacks, a bidirectional nn.LSTM, nn.LSTM in place of nn.LSTMCell for the two
decoder cells as the export script does, LinearNorm layers, location-sensitive attention), with the graph I/O names the
reference binds (mod.rs:284-296,306-307,332-339,349).  LSTM weight packing (W/R/B, gate order i,o,f,c), the Conv+BN fusion
of eval-mode export, MatMul-vs-Gemm orientation and constant folding are all the exporter's, not ours.

Test infrastructure only (needs torch; no `onnx` package: its only use inside the exporter is a post-step that scans the
finished proto for onnxscript functions, stubbed below)."""
import os
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvNorm(nn.Module):
    def __init__(self, ci, co, k, bias=True):
        super().__init__()
        self.conv = nn.Conv1d(ci, co, k, padding=(k - 1) // 2, bias=bias)

    def forward(self, x):
        return self.conv(x)


class LinearNorm(nn.Module):
    def __init__(self, i, o, bias=True):
        super().__init__()
        self.linear_layer = nn.Linear(i, o, bias=bias)

    def forward(self, x):
        return self.linear_layer(x)


class Encoder(nn.Module):
    """encoder.onnx: embedding -> 3 x (conv5 + BN + relu [+ eval dropout]) -> BiLSTM -> memory; the attention's memory_layer
    rides in this graph (processed_memory is an encoder output, mod.rs:382-385)."""

    def __init__(self):
        super().__init__()
        self.embedding = nn.Embedding(148, 512)
        self.convolutions = nn.ModuleList([nn.Sequential(ConvNorm(512, 512, 5), nn.BatchNorm1d(512)) for _ in range(3)])
        self.lstm = nn.LSTM(512, 256, 1, batch_first=True, bidirectional=True)
        self.memory_layer = LinearNorm(512, 128, bias=False)

    def forward(self, sequences, sequence_lengths):
        x = self.embedding(sequences).transpose(1, 2)
        for conv in self.convolutions:
            x = F.dropout(F.relu(conv(x)), 0.5, self.training)
        x = x.transpose(1, 2)
        # (the reference always passes plen = the padded window, mod.rs:375, so packing by length is the identity)
        outputs, _ = self.lstm(x)
        return outputs, self.memory_layer(outputs), sequence_lengths + 0


class DecoderIter(nn.Module):
    """decoder_iter.onnx: one Decoder.decode step, state in / state out (mod.rs:284-296,332-339)."""

    def __init__(self):
        super().__init__()
        self.prenet = nn.ModuleList([LinearNorm(80, 256, bias=False), LinearNorm(256, 256, bias=False)])
        self.attention_rnn = nn.LSTM(256 + 512, 1024, 1)
        self.query_layer = LinearNorm(1024, 128, bias=False)
        self.v = LinearNorm(128, 1, bias=False)
        self.location_conv = ConvNorm(2, 32, 31, bias=False)
        self.location_dense = LinearNorm(32, 128, bias=False)
        self.decoder_rnn = nn.LSTM(1024 + 512, 1024, 1)
        self.linear_projection = LinearNorm(1024 + 512, 80)
        self.gate_layer = LinearNorm(1024 + 512, 1)

    def forward(self, decoder_input, attention_hidden, attention_cell, decoder_hidden, decoder_cell, attention_weights,
                attention_weights_cum, attention_context, memory, processed_memory, mask):
        x = decoder_input
        for layer in self.prenet:  # always-on dropout: the exported graph keeps it (hard part i)
            x = F.dropout(F.relu(layer(x)), p=0.5, training=True)
        cell_input = torch.cat((x, attention_context), -1)
        _, (ah, ac) = self.attention_rnn(cell_input.unsqueeze(0), (attention_hidden.unsqueeze(0), attention_cell.unsqueeze(0)))
        ah, ac = ah.squeeze(0), ac.squeeze(0)
        cat = torch.cat((attention_weights.unsqueeze(1), attention_weights_cum.unsqueeze(1)), dim=1)
        pq = self.query_layer(ah.unsqueeze(1))
        loc = self.location_dense(self.location_conv(cat).transpose(1, 2))
        energies = self.v(torch.tanh(pq + loc + processed_memory)).squeeze(-1)
        energies = energies.masked_fill(mask, -float("inf"))
        aw = F.softmax(energies, dim=1)
        ctx = torch.bmm(aw.unsqueeze(1), memory).squeeze(1)
        awc = attention_weights_cum + aw
        _, (dh, dc) = self.decoder_rnn(torch.cat((ah, ctx), -1).unsqueeze(0), (decoder_hidden.unsqueeze(0), decoder_cell.unsqueeze(0)))
        dh, dc = dh.squeeze(0), dc.squeeze(0)
        hc = torch.cat((dh, ctx), dim=1)
        return self.linear_projection(hc), self.gate_layer(hc), ah, ac, dh, dc, aw, awc, ctx


class Postnet(nn.Module):
    """postnet.onnx: mel_outputs + 5 x (conv5 + BN [+ tanh]) -- the residual is inside the graph (mod.rs:349)."""

    def __init__(self):
        super().__init__()
        ch = [80, 512, 512, 512, 512, 80]
        self.convolutions = nn.ModuleList([nn.Sequential(ConvNorm(ch[i], ch[i + 1], 5), nn.BatchNorm1d(ch[i + 1])) for i in range(5)])

    def forward(self, mel_outputs):
        x = mel_outputs
        for i, conv in enumerate(self.convolutions):
            x = conv(x)
            if i < 4:
                x = torch.tanh(x)
            x = F.dropout(x, 0.5, self.training)
        return mel_outputs + x


def _set(p, a):
    with torch.no_grad():
        p.copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(p.shape))


def _conv_bn(seq, T, prefix):
    _set(seq[0].conv.weight, T[prefix + ".conv.weight"])
    _set(seq[0].conv.bias, T[prefix + ".conv.bias"])
    _set(seq[1].weight, T[prefix + ".bn.weight"])
    _set(seq[1].bias, T[prefix + ".bn.bias"])
    _set(seq[1].running_mean, T[prefix + ".bn.running_mean"])
    _set(seq[1].running_var, T[prefix + ".bn.running_var"])


def _lstm(m, T, prefix, suffix=""):
    for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        _set(getattr(m, k + "_l0" + suffix), T[prefix + k])


def build_modules(T):
    """The three modules with the canonical tensors T (dict name -> array, xdtts_tensor_* names) loaded."""
    enc, dec, post = Encoder(), DecoderIter(), Postnet()
    _set(enc.embedding.weight, T["embedding.weight"])
    for i in range(3):
        _conv_bn(enc.convolutions[i], T, "encoder.convolutions.%d" % i)
    _lstm(enc.lstm, T, "encoder.lstm.fwd.")
    _lstm(enc.lstm, T, "encoder.lstm.bwd.", "_reverse")
    _set(enc.memory_layer.linear_layer.weight, T["attention.memory_layer.weight"])
    _set(dec.prenet[0].linear_layer.weight, T["prenet.0.weight"])
    _set(dec.prenet[1].linear_layer.weight, T["prenet.1.weight"])
    _lstm(dec.attention_rnn, T, "attention_rnn.")
    _lstm(dec.decoder_rnn, T, "decoder_rnn.")
    _set(dec.query_layer.linear_layer.weight, T["attention.query_layer.weight"])
    _set(dec.v.linear_layer.weight, T["attention.v.weight"])
    _set(dec.location_conv.conv.weight, T["attention.location_conv.weight"])
    _set(dec.location_dense.linear_layer.weight, T["attention.location_dense.weight"])
    _set(dec.linear_projection.linear_layer.weight, T["linear_projection.weight"])
    _set(dec.linear_projection.linear_layer.bias, T["linear_projection.bias"])
    _set(dec.gate_layer.linear_layer.weight, T["gate_layer.weight"])
    _set(dec.gate_layer.linear_layer.bias, T["gate_layer.bias"])
    for i in range(5):
        _conv_bn(post.convolutions[i], T, "postnet.convolutions.%d" % i)
    return enc.eval(), dec.eval(), post.eval()


DEC_INPUTS = ["decoder_input", "attention_hidden", "attention_cell", "decoder_hidden", "decoder_cell", "attention_weights",
              "attention_weights_cum", "attention_context", "memory", "processed_memory", "mask"]
DEC_OUTPUTS = ["decoder_output", "gate_prediction", "out_attention_hidden", "out_attention_cell", "out_decoder_hidden",
               "out_decoder_cell", "out_attention_weights", "out_attention_weights_cum", "out_attention_context"]


def export_model_dir(path, T, opset=13, dec_outputs=None, fuse_bn=True):
    """Writes path/{encoder,decoder_iter,postnet}.onnx with torch.onnx.export(dynamo=False).
    fuse_bn=True: today's default eval-mode export, which folds every BatchNorm into its conv (the exporter's own
    `eval peephole`); False: TrainingMode.PRESERVE on the eval-mode modules -- constants still folded, BatchNormalization nodes
    kept, which is what exporters of the reference's vintage wrote: the three files then come out at 22 640 6xx / 72 767 4xx /
    17 414 4xx bytes against the LFS pointers' 22 641 034 / 72 766 349 / 17 414 016 (models/tacotron2/*.onnx)."""
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as opu

    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes  # (the only step that imports the `onnx` package)
    enc, dec, post = build_modules(T)
    Tn = 100
    kw = dict(dynamo=False, opset_version=opset, do_constant_folding=True)
    if not fuse_bn:
        kw["training"] = torch.onnx.TrainingMode.PRESERVE
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(enc, (torch.zeros(1, Tn, dtype=torch.int64), torch.tensor([Tn], dtype=torch.int64)), os.path.join(path, "encoder.onnx"),
                          input_names=["sequences", "sequence_lengths"], output_names=["memory", "processed_memory", "lens"], **kw)
        z = torch.zeros
        args = (z(1, 80), z(1, 1024), z(1, 1024), z(1, 1024), z(1, 1024), z(1, Tn), z(1, Tn), z(1, 512), z(1, Tn, 512), z(1, Tn, 128),
                z(1, Tn, dtype=torch.bool))
        torch.onnx.export(dec, args, os.path.join(path, "decoder_iter.onnx"), input_names=DEC_INPUTS, output_names=dec_outputs or DEC_OUTPUTS, **kw)
        torch.onnx.export(post, (z(1, 80, 50),), os.path.join(path, "postnet.onnx"), input_names=["mel_outputs"], output_names=["mel_outputs_postnet"],
                          dynamic_axes={"mel_outputs": {2: "frames"}, "mel_outputs_postnet": {2: "frames"}}, **kw)
    return enc, dec, post


def fused_conv_bn(T, prefix):
    """What an eval-mode exporter folds Conv + BatchNorm into, in fp32 as torch does it."""
    w = torch.from_numpy(np.ascontiguousarray(T[prefix + ".conv.weight"]))
    b = torch.from_numpy(np.ascontiguousarray(T[prefix + ".conv.bias"]))
    g, beta = torch.from_numpy(np.ascontiguousarray(T[prefix + ".bn.weight"])), torch.from_numpy(np.ascontiguousarray(T[prefix + ".bn.bias"]))
    mu, var = torch.from_numpy(np.ascontiguousarray(T[prefix + ".bn.running_mean"])), torch.from_numpy(np.ascontiguousarray(T[prefix + ".bn.running_var"]))
    scale = g / torch.sqrt(var + 1e-5)
    return (w * scale.reshape(-1, 1, 1)).numpy(), ((b - mu) * scale + beta).numpy()
