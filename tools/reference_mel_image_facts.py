#!/usr/bin/env python3
"""Forensics on the only view of `Tacotron2::infer`'s OUTPUT the reference holds: slides/images/melgen_py_vs_rust.svg
(slides/melgen.typ:176-184, "But They Look Close!"), a matplotlib figure with two embedded PNGs -- "Python Output" and
"Rust ONNX Output" -- of one utterance's mel.

BUILD-CONTAINER ONLY (reads /root/reference; numpy + PIL): writes the `mel_images` section of
tests/golden/reference_audio_facts.json -- statistics, no pixel of the reference's figure is stored.

The PNGs are nearest-neighbour renderings (imshow), so the cell grid can be read off exactly: a run of identical pixel
rows / columns is one mel band / one frame.  What the script answers, per image:
  grid         bands x frames of the rendered array -> the layout of Tacotron2::infer's Array2 (80 rows = mel bands,
               src/tacotron2/mod.rs:349-355,430) and the frame count of that utterance
  orientation  which edge holds the low bands (speech energy sits there)
  floor share  share of cells on the darkest colour: the log-mel floor (ln 1e-5 = -11.5 in NVIDIA's compression) that leading
               and trailing silence and the bands above the voice sit on
  colour scale the colormap is sequential (magma): luminance is monotone in the value, so rank statistics survive
"""
import base64
import io
import json
import os
import re

import numpy as np

SVG = "/root/reference/slides/images/melgen_py_vs_rust.svg"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_audio_facts.json")


def runs(changed):
    return np.diff(np.flatnonzero(np.r_[True, changed, True]))


def analyse(png_bytes):
    from PIL import Image

    a = np.asarray(Image.open(io.BytesIO(png_bytes)).convert("RGB")).astype(np.int64)
    h, w = a.shape[:2]
    col_runs = runs(np.any(a[:, 1:] != a[:, :-1], axis=(0, 2)))
    row_runs = runs(np.any(a[1:, :] != a[:-1, :], axis=(1, 2)))
    # one sample per cell (its first pixel), luminance as the monotone proxy of the value
    r0 = np.r_[0, np.cumsum(row_runs)[:-1]]
    c0 = np.r_[0, np.cumsum(col_runs)[:-1]]
    cells = a[np.ix_(r0, c0)]
    lum = 0.2126 * cells[..., 0] + 0.7152 * cells[..., 1] + 0.0722 * cells[..., 2]
    lo, hi = float(lum.min()), float(lum.max())
    x = (lum - lo) / (hi - lo)
    nb, nf = x.shape
    top, bottom = float(x[: nb // 4].mean()), float(x[-(nb // 4):].mean())
    floor = x <= 0.02
    frame_floor = floor.mean(axis=0)
    quiet = frame_floor > 0.9   # frames with (nearly) every band on the floor
    lead = int(np.argmax(~quiet)) if (~quiet).any() else nf
    trail = int(np.argmax(~quiet[::-1])) if (~quiet).any() else nf
    return {
        "pixels": [w, h],
        "bands": int(nb),
        "frames_at_least": int(nf),                      # identical neighbouring frames would merge: a lower bound
        "pixels_per_frame": [int(col_runs.min()), int(col_runs.max())],
        "pixels_per_band": [int(row_runs.min()), int(row_runs.max())],
        "low_bands_at": "bottom" if bottom > top else "top",
        "mean_level_top_quarter": top,
        "mean_level_bottom_quarter": bottom,
        "floor_share": float(floor.mean()),
        "floor_share_upper_half_of_the_bands": float(floor[: nb // 2].mean() if bottom > top else floor[nb // 2 :].mean()),
        "leading_floor_frames": lead,
        "trailing_floor_frames": trail,
        "median_level": float(np.median(x)),
        "distinct_colours": int(len(np.unique(cells.reshape(-1, 3), axis=0))),
    }


def main():
    svg = open(SVG).read()
    titles = re.findall(r"<!-- ([A-Za-z ]+ Output) -->", svg)
    pngs = re.findall(r'xlink:href="data:image/png;base64,\s*([^"]+)"', svg)
    assert len(pngs) == 2 and len(titles) == 2, (len(pngs), titles)
    out = {"source": "slides/images/melgen_py_vs_rust.svg (slides/melgen.typ:176-184): statistics of the two embedded PNGs, no pixel stored",
           "images": {}}
    for title, b64 in zip(titles, pngs):
        out["images"][title] = analyse(base64.b64decode(re.sub(r"\s+", "", b64)))
    doc = json.load(open(OUT))
    doc["mel_images"] = out
    json.dump(doc, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
