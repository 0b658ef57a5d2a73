"""torch-CPU restatement of the hot path, fp32, all host cores: bench.py's third CPU baseline
(BASELINE.md section 3, row C3).  TEST/BASELINE INFRASTRUCTURE ONLY -- never imported by the product.

The reference runs its three graphs through onnxruntime's MLAS kernels and the crate's realfft; torch's
CPU backend (MKL/oneDNN GEMV, pocketfft) is the same class of tuned library kernels, driven the way the
reference drives ORT: batch 1, one frame per step (src/tacotron2/mod.rs:302-342), chunks in sequence
(mod.rs:422-434), then GriffinLim::infer (src/lib.rs:141).  Same math as tests/torch_ref.py (which pins
the C oracle in fp64); here the weights are converted once and everything stays float32.
"""
import numpy as np
import torch
import torch.nn.functional as F


class TorchTacotron2:
    def __init__(self, orc, blob, threads=None):
        if threads:
            torch.set_num_threads(int(threads))
        self.threads = torch.get_num_threads()
        self.orc = orc
        t = lambda name: torch.from_numpy(np.array(orc.tensor(blob, name), dtype=np.float32))
        self.w = {n: t(n) for n, _s, _o, _k in orc.tensor_table()}
        w = self.w
        # fold eval BatchNorm into the conv weights once (what ORT's Level3 optimiser does)
        self.convs = {}
        for prefix, n in (("encoder.convolutions", 3), ("postnet.convolutions", 5)):
            for i in range(n):
                p = "%s.%d" % (prefix, i)
                inv = w[p + ".bn.weight"] / torch.sqrt(w[p + ".bn.running_var"] + 1e-5)
                self.convs[p] = (w[p + ".conv.weight"] * inv[:, None, None], (w[p + ".conv.bias"] - w[p + ".bn.running_mean"]) * inv + w[p + ".bn.bias"])

    def encoder(self, ids):
        w = self.w
        x = w["embedding.weight"][torch.as_tensor(np.asarray(ids, dtype=np.int64))].T.unsqueeze(0)
        for i in range(3):
            cw, cb = self.convs["encoder.convolutions.%d" % i]
            x = torch.relu(F.conv1d(x, cw, cb, padding=2))
        x = x[0].T.contiguous()
        T = x.shape[0]
        out = torch.zeros(T, 512)
        for d, name in enumerate(("fwd", "bwd")):
            p = "encoder.lstm.%s." % name
            wih, whh, bih, bhh = w[p + "weight_ih"], w[p + "weight_hh"], w[p + "bias_ih"], w[p + "bias_hh"]
            h = torch.zeros(1, 256)
            c = torch.zeros(1, 256)
            xin = F.linear(x, wih, bih)            # input projections for all T at once
            for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
                g = xin[t : t + 1] + F.linear(h, whh, bhh)
                i_, f_, g_, o_ = g.chunk(4, dim=1)
                c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
                h = torch.sigmoid(o_) * torch.tanh(c)
                out[t, d * 256 : (d + 1) * 256] = h[0]
        return out, out @ w["attention.memory_layer.weight"].T

    def decode(self, memory, pmem, n_valid, steps, seed, item):
        """The frame loop with the gate disabled (fixed work), dropout masks from the oracle's stream."""
        w, orc = self.w, self.orc
        T = memory.shape[0]
        att_h = torch.zeros(1, 1024); att_c = torch.zeros(1, 1024)
        dec_h = torch.zeros(1, 1024); dec_c = torch.zeros(1, 1024)
        aw = torch.zeros(T); awc = torch.zeros(T); ctx = torch.zeros(512); dec_in = torch.zeros(80)
        frames = torch.empty(steps, 80)
        keep = np.empty((steps, 2, 256), dtype=np.float32)
        for s in range(steps):
            for layer in range(2):
                keep[s, layer] = [2.0 * orc.lib.orc_dropout_keep(seed, item, s, layer, j) for j in range(256)]
        keep = torch.from_numpy(keep)
        lc_w, ld_w, v = w["attention.location_conv.weight"], w["attention.location_dense.weight"], w["attention.v.weight"]
        for s in range(steps):
            x = torch.relu(w["prenet.0.weight"] @ dec_in) * keep[s, 0]
            x = torch.relu(w["prenet.1.weight"] @ x) * keep[s, 1]
            att_h, att_c = torch._VF.lstm_cell(torch.cat([x, ctx]).unsqueeze(0), (att_h, att_c), w["attention_rnn.weight_ih"], w["attention_rnn.weight_hh"], w["attention_rnn.bias_ih"], w["attention_rnn.bias_hh"])
            q = w["attention.query_layer.weight"] @ att_h[0]
            loc = F.conv1d(torch.stack([aw, awc]).unsqueeze(0), lc_w, padding=15)[0].T @ ld_w.T
            e = torch.tanh(q.unsqueeze(0) + loc + pmem) @ v.reshape(-1)
            e[n_valid:] = -float("inf")
            aw = torch.softmax(e, dim=0)
            awc = awc + aw
            ctx = aw @ memory
            dec_h, dec_c = torch._VF.lstm_cell(torch.cat([att_h[0], ctx]).unsqueeze(0), (dec_h, dec_c), w["decoder_rnn.weight_ih"], w["decoder_rnn.weight_hh"], w["decoder_rnn.bias_ih"], w["decoder_rnn.bias_hh"])
            hc = torch.cat([dec_h[0], ctx])
            dec_in = w["linear_projection.weight"] @ hc + w["linear_projection.bias"]
            frames[s] = dec_in
        return frames

    def postnet(self, frames):
        x = frames.T.unsqueeze(0)
        y = x
        for i in range(5):
            cw, cb = self.convs["postnet.convolutions.%d" % i]
            y = F.conv1d(y, cw, cb, padding=2)
            if i < 4:
                y = torch.tanh(y)
        return (x + y)[0]

    def infer_chunk(self, ids, steps, seed, item, window=100):
        padded = np.zeros(window, dtype=np.int64)
        padded[: len(ids)] = ids
        with torch.no_grad():
            mem, pm = self.encoder(padded)
            return self.postnet(self.decode(mem, pm, len(ids), steps, seed, item)).numpy()


def griffinlim(S, phase0, iters, momentum=0.99, n_fft=1024, hop=256):
    """librosa.griffinlim on torch.stft / torch.istft, fp32."""
    with torch.no_grad():
        S = torch.as_tensor(np.asarray(S, dtype=np.float32))
        ang = torch.complex(torch.as_tensor(np.ascontiguousarray(phase0[..., 0], dtype=np.float32)), torch.as_tensor(np.ascontiguousarray(phase0[..., 1], dtype=np.float32)))
        win = torch.hann_window(n_fft, periodic=True)
        n = hop * (S.shape[1] - 1)
        rebuilt = torch.zeros_like(ang)
        alpha = momentum / (1 + momentum)
        for _ in range(iters):
            tprev = rebuilt
            inverse = torch.istft(S * ang, n_fft, hop, n_fft, win, center=True, length=n)
            rebuilt = torch.stft(inverse, n_fft, hop, n_fft, win, center=True, pad_mode="reflect", return_complex=True)
            ang = rebuilt - alpha * tprev
            ang = ang / (ang.abs() + 1e-16)
        return torch.istft(S * ang, n_fft, hop, n_fft, win, center=True, length=n).numpy()
