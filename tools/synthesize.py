#!/usr/bin/env python3
"""`XdTts::infer` end to end through the C ABI (src/lib.rs:110-160): unit tokens -> ids -> chunks ->
Tacotron2 mel -> Griffin-Lim -> 16-bit WAV, with the reference's three log lines (mel-gen time,
vocoder time, real time factor).  Needs an MI355X.  The text front end (normaliser, CMU dictionary)
is out of scope: the input is already a unit sequence, ARPAbet phones or characters.

  python tools/synthesize.py --model DIR      --units "HH AH0 L OW1 , W ER1 L D ." --out hello.wav
  python tools/synthesize.py --synthetic      --chars "hello world." --out noise.wav   (random weights: noise)
DIR holds tacotron2.xdtw (make it from the reference's ONNX files with tools/onnx_to_xdtw.py)."""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--model", help="directory with tacotron2.xdtw")
    src.add_argument("--synthetic", action="store_true", help="seeded random weights of the NVIDIA shapes")
    txt = ap.add_mutually_exclusive_group(required=True)
    txt.add_argument("--units", help="space-separated unit tokens (ARPAbet phones, punctuation; use _ for a space)")
    txt.add_argument("--chars", help="a string, one Unit::Character per character")
    ap.add_argument("--out", required=True, help="output .wav (mono, 22050 Hz, 16-bit)")
    ap.add_argument("--mel", help="also dump the (80, F) spectrogram as .npy (src/lib.rs:128-141)")
    ap.add_argument("--iters", type=int, default=30, help="Griffin-Lim iterations (reference: 30)")
    ap.add_argument("--seed", type=int, default=0, help="dropout / initial-phase seed")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    pkg = importlib.import_module("xd-tts_amd")
    model = pkg.Tacotron2.synthetic(device_id=a.device) if a.synthetic else pkg.Tacotron2.load(a.model, device_id=a.device)
    vocoder = pkg.create_griffin_lim(device_id=a.device, iters=a.iters, seed=a.seed)
    if a.units is not None:
        ids = pkg.units_to_ids([" " if t == "_" else t for t in a.units.split()])
    else:
        ids = pkg.units_to_ids(list(a.chars), as_character=True)
    if len(ids) == 0:
        raise SystemExit("no unit of the input has an id (src/tacotron2/mod.rs:403-406 drops them all)")
    splits = pkg.find_splits(ids, 100)
    opts = pkg.default_opts(dropout_seed=a.seed)
    t0 = time.perf_counter()
    mel = model.infer(ids, splits=splits, opts=opts)
    t1 = time.perf_counter()
    audio = vocoder.infer(mel)
    t2 = time.perf_counter()
    print("Mel gen time: %.3f ms (%d ids -> %d frames)" % ((t1 - t0) * 1e3, len(ids), mel.shape[1]))
    print("Vocoder time: %.3f ms (%d samples)" % ((t2 - t1) * 1e3, audio.size))
    print("Real time factor: %.5f" % pkg.real_time_factor(t2 - t0, audio.size))
    pkg.write_wav(a.out, audio)
    if a.mel:
        pkg.write_mel_npy(a.mel, mel)
    print("wrote %s%s" % (a.out, (" and " + a.mel) if a.mel else ""))


if __name__ == "__main__":
    main()
