"""Independent restatement of the Tacotron2 layer math on torch CPU ops (float64).

Used ONLY to pin the C oracle (oracle/xdtts_oracle.c): torch.nn.functional conv1d / batch_norm /
softmax and torch's own LSTM cell are a third-party implementation of the same published layers
(NVIDIA Tacotron2 model.py), so agreement to ~1e-12 in fp64 shows the oracle computes what the
reference's ONNX graphs (src/tacotron2/mod.rs:304,347,379) are exported from.
"""
import numpy as np
import torch
import torch.nn.functional as F

torch.set_num_threads(4)
DT = torch.float64


def _t(orc, blob, name):
    return torch.from_numpy(np.array(orc.tensor(blob, name), dtype=np.float64))


def _lstm_cell(x, h, c, wih, whh, bih, bhh):
    # torch's fused cell: gate order i, f, g, o
    hn, cn = torch._VF.lstm_cell(x.unsqueeze(0), (h.unsqueeze(0), c.unsqueeze(0)), wih, whh, bih, bhh)
    return hn[0], cn[0]


def _conv_bn(orc, blob, prefix, x, act):
    w = _t(orc, blob, prefix + ".conv.weight")
    y = F.conv1d(x, w, _t(orc, blob, prefix + ".conv.bias"), padding=(w.shape[2] - 1) // 2)
    y = F.batch_norm(
        y,
        _t(orc, blob, prefix + ".bn.running_mean"),
        _t(orc, blob, prefix + ".bn.running_var"),
        _t(orc, blob, prefix + ".bn.weight"),
        _t(orc, blob, prefix + ".bn.bias"),
        training=False,
        eps=1e-5,
    )
    return act(y) if act else y


def encoder(orc, blob, ids):
    """embedding -> 3 x (conv5 + BN + relu) -> BiLSTM -> memory; processed_memory = memory_layer."""
    emb = _t(orc, blob, "embedding.weight")
    x = emb[torch.as_tensor(np.asarray(ids, dtype=np.int64))].T.unsqueeze(0)  # (1, 512, T)
    for i in range(3):
        x = _conv_bn(orc, blob, "encoder.convolutions.%d" % i, x, torch.relu)
    x = x[0].T  # (T, 512)
    T = x.shape[0]
    out = torch.zeros(T, 512, dtype=DT)
    for d, name in enumerate(("fwd", "bwd")):
        p = "encoder.lstm.%s." % name
        wih, whh = _t(orc, blob, p + "weight_ih"), _t(orc, blob, p + "weight_hh")
        bih, bhh = _t(orc, blob, p + "bias_ih"), _t(orc, blob, p + "bias_hh")
        h = torch.zeros(256, dtype=DT)
        c = torch.zeros(256, dtype=DT)
        order = range(T) if d == 0 else range(T - 1, -1, -1)
        for t in order:
            h, c = _lstm_cell(x[t], h, c, wih, whh, bih, bhh)
            out[t, d * 256 : (d + 1) * 256] = h
    pmem = out @ _t(orc, blob, "attention.memory_layer.weight").T
    return out.numpy(), pmem.numpy()


class DecoderState:
    def __init__(self, T):
        z = lambda n: torch.zeros(n, dtype=DT)
        self.att_h, self.att_c, self.dec_h, self.dec_c = z(1024), z(1024), z(1024), z(1024)
        self.aw, self.awc, self.ctx, self.dec_in = z(T), z(T), z(512), z(80)


def decoder_step(orc, blob, memory, pmem, n_valid, s, keep0=None, keep1=None):
    """One decoder_iter call (NVIDIA Decoder.decode); keep0/keep1 are the prenet dropout keep masks."""
    memory = torch.as_tensor(memory, dtype=DT)
    pmem = torch.as_tensor(pmem, dtype=DT)
    T = memory.shape[0]
    x = torch.relu(_t(orc, blob, "prenet.0.weight") @ s.dec_in)
    if keep0 is not None:
        x = x * torch.as_tensor(keep0, dtype=DT) * 2
    x = torch.relu(_t(orc, blob, "prenet.1.weight") @ x)
    if keep1 is not None:
        x = x * torch.as_tensor(keep1, dtype=DT) * 2
    p = "attention_rnn."
    s.att_h, s.att_c = _lstm_cell(
        torch.cat([x, s.ctx]), s.att_h, s.att_c, _t(orc, blob, p + "weight_ih"), _t(orc, blob, p + "weight_hh"), _t(orc, blob, p + "bias_ih"), _t(orc, blob, p + "bias_hh")
    )
    q = _t(orc, blob, "attention.query_layer.weight") @ s.att_h
    cat = torch.stack([s.aw, s.awc]).unsqueeze(0)  # (1, 2, T)
    loc = F.conv1d(cat, _t(orc, blob, "attention.location_conv.weight"), padding=15)[0].T  # (T, 32)
    loc = loc @ _t(orc, blob, "attention.location_dense.weight").T  # (T, 128)
    e = torch.tanh(q.unsqueeze(0) + loc + pmem) @ _t(orc, blob, "attention.v.weight")
    e[n_valid:] = -float("inf")
    s.aw = torch.softmax(e, dim=0)
    s.awc = s.awc + s.aw
    s.ctx = s.aw @ memory
    p = "decoder_rnn."
    s.dec_h, s.dec_c = _lstm_cell(
        torch.cat([s.att_h, s.ctx]), s.dec_h, s.dec_c, _t(orc, blob, p + "weight_ih"), _t(orc, blob, p + "weight_hh"), _t(orc, blob, p + "bias_ih"), _t(orc, blob, p + "bias_hh")
    )
    hc = torch.cat([s.dec_h, s.ctx])
    mel = _t(orc, blob, "linear_projection.weight") @ hc + _t(orc, blob, "linear_projection.bias")
    gate = _t(orc, blob, "gate_layer.weight") @ hc + _t(orc, blob, "gate_layer.bias")[0]
    s.dec_in = mel
    return mel.numpy(), float(gate)


def postnet(orc, blob, frames):
    x = torch.as_tensor(frames, dtype=DT).T.unsqueeze(0)  # (1, 80, F)
    y = x
    for i in range(5):
        y = _conv_bn(orc, blob, "postnet.convolutions.%d" % i, y, torch.tanh if i < 4 else None)
    return (x + y)[0].numpy()


def stft(y, n_fft=1024, hop=256):
    w = torch.hann_window(n_fft, periodic=True, dtype=DT)
    return torch.stft(torch.as_tensor(y, dtype=DT), n_fft, hop, n_fft, w, center=True, pad_mode="reflect", return_complex=True)


def istft(spec, length, n_fft=1024, hop=256):
    w = torch.hann_window(n_fft, periodic=True, dtype=DT)
    return torch.istft(spec, n_fft, hop, n_fft, w, center=True, length=length)


def griffinlim(S, phase0, iters, momentum=0.99, n_fft=1024, hop=256):
    """librosa.griffinlim restated on torch.stft / torch.istft (fp64)."""
    S = torch.as_tensor(S, dtype=DT)
    ang = torch.as_tensor(phase0[..., 0], dtype=DT) + 1j * torch.as_tensor(phase0[..., 1], dtype=DT)
    n = hop * (S.shape[1] - 1)
    rebuilt = torch.zeros_like(ang)
    for _ in range(iters):
        tprev = rebuilt
        inverse = istft(S * ang, n, n_fft, hop)
        rebuilt = stft(inverse, n_fft, hop)
        ang = rebuilt - (momentum / (1 + momentum)) * tprev
        ang = ang / (ang.abs() + 1e-16)
    return istft(S * ang, n, n_fft, hop).numpy()
