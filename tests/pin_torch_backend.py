"""Dry-run backend of tools/pin/record_run.py (TEST INFRASTRUCTURE): the torch modules of tests/nvidia_torch_export.py loaded with
the weights the exported model directory holds (read back through tools/onnx_to_xdtw.collect), the two prenet dropout layers
driven by the recorded masks.  It stands where onnxruntime + the real artefacts stand in a real pin; it is NOT the reference."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


class TorchBackend:
    version = "torch %s modules of tests/nvidia_torch_export.py (dry run of the pin kit, NOT the reference)" % torch.__version__

    def __init__(self, model_dir):
        import nvidia_torch_export as nte
        import onnx_to_xdtw

        self.enc, self.dec, self.post = nte.build_modules(onnx_to_xdtw.collect(model_dir))

    def encoder(self, ids, lens):
        with torch.no_grad():
            mem, pm, _ = self.enc(torch.from_numpy(ids), torch.from_numpy(lens))
        return mem.numpy(), pm.numpy()

    def decoder(self, feed):
        """DecoderIter.forward (tests/nvidia_torch_export.py) with the dropout of the two prenet layers replaced by the fed scales
        (dropout_scale_k: 0 or 2) -- or by Less(u, 0.5) on dropout_uniform_k"""
        d, t = self.dec, {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in feed.items()}
        with torch.no_grad():
            x = t["decoder_input"]
            for k, layer in enumerate(d.prenet):
                x = F.relu(layer(x))
                if "dropout_scale_%d" % k in t:
                    x = x * t["dropout_scale_%d" % k]
                elif "dropout_uniform_%d" % k in t:
                    x = x * (t["dropout_uniform_%d" % k] < 0.5).float() * 2.0
            cell_input = torch.cat((x, t["attention_context"]), -1)
            _, (ah, ac) = d.attention_rnn(cell_input.unsqueeze(0), (t["attention_hidden"].unsqueeze(0), t["attention_cell"].unsqueeze(0)))
            ah, ac = ah.squeeze(0), ac.squeeze(0)
            cat = torch.cat((t["attention_weights"].unsqueeze(1), t["attention_weights_cum"].unsqueeze(1)), dim=1)
            pq = d.query_layer(ah.unsqueeze(1))
            loc = d.location_dense(d.location_conv(cat).transpose(1, 2))
            energies = d.v(torch.tanh(pq + loc + t["processed_memory"])).squeeze(-1)
            energies = energies.masked_fill(t["mask"], -float("inf"))
            aw = F.softmax(energies, dim=1)
            ctx = torch.bmm(aw.unsqueeze(1), t["memory"]).squeeze(1)
            awc = t["attention_weights_cum"] + aw
            _, (dh, dc) = d.decoder_rnn(torch.cat((ah, ctx), -1).unsqueeze(0), (t["decoder_hidden"].unsqueeze(0), t["decoder_cell"].unsqueeze(0)))
            dh, dc = dh.squeeze(0), dc.squeeze(0)
            hc = torch.cat((dh, ctx), dim=1)
            out = dict(decoder_output=d.linear_projection(hc), gate_prediction=d.gate_layer(hc), out_attention_hidden=ah, out_attention_cell=ac,
                       out_decoder_hidden=dh, out_decoder_cell=dc, out_attention_weights=aw, out_attention_weights_cum=awc, out_attention_context=ctx)
        return {k: v.numpy() for k, v in out.items()}

    def postnet(self, mel):
        with torch.no_grad():
            return self.post(torch.from_numpy(np.ascontiguousarray(mel))).numpy()
