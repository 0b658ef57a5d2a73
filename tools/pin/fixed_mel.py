"""The fixed log-mel (80 x 64) both sides of the vocoder pin use: tools/pin/dump_mel_to_linear.rs reads fixed_mel.npy (written
by `python tools/pin/fixed_mel.py`), tests/test_gpu_reference_pinned.py rebuilds it.  Closed form, no RNG: a -11.5 floor
(ln 1e-5, the compression's clamp) with a voiced ridge, values -11.5 .. 1.4."""
import os

import numpy as np


def fixed_mel(F=64):
    t = np.arange(F, dtype=np.float64)[None, :]
    m = np.arange(80, dtype=np.float64)[:, None]
    f0 = 6.0 + 2.0 * np.sin(2.0 * np.pi * t / 37.0)
    ridge = sum(np.exp(-0.5 * ((m - k * f0) / 1.3) ** 2) for k in range(1, 9))
    voiced = (t >= 6) & (t < F - 6)
    lin = 1e-5 + 4.0 * voiced * np.exp(-m / 35.0) * ridge
    return np.log(np.maximum(lin, 1e-5)).astype(np.float32)


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixed_mel.npy")
    np.save(out, fixed_mel())
    print(out, fixed_mel().shape, float(fixed_mel().min()), float(fixed_mel().max()))
