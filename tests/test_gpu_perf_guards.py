"""Performance guards: coarse upper bounds on the step times of the engines the bench runs on, so that a code-generation cliff does
not pass the suite unnoticed.  (Round 4 met one: the skewed pair kernel of decoder_persistent.hip compiled with no stop-rule code at
all ran 49 us per step instead of 10.6 -- correct results, no time-out, nothing but a clock would have caught it.  The shipped
instantiations are the tuned ones; VERDICT round 4, item 6 asked for a test that keeps it that way.)  Bounds are ~1.4x the measured
values (profiles/r05_batch_sweep.txt, r05_bench_n1.json): box-to-box spread is a few percent, a cliff is a factor."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
wl = importlib.import_module("xd-tts_amd.workloads")


def _us_per_step(pkg, model, B, n=300):
    chunks = [wl.synth_ids(60 + (7 * b) % 40, seed=10 + b) for b in range(B)]
    o = pkg.default_opts(dropout_seed=1)
    best = 1e9
    for _ in range(4):
        model.infer_batch(chunks, opts=o, fixed_steps=[n] * B)
        best = min(best, model.last_timings()["decoder_ms"] * 1e3 / n)
    return best


@pytest.mark.parametrize("B,bound", [(1, 12.0), (2, 15.5), (4, 20.0), (8, 23.0), (9, 23.0), (12, 23.0), (16, 23.0), (17, 36.0), (52, 54.0)])
def test_decoder_engines_step_time(pkg, model, B, bound):
    us = _us_per_step(pkg, model, B)
    assert us <= bound, "B=%d: %.1f us per lock-step iteration (bound %.1f)" % (B, us, bound)


def test_small_batch_throughput_does_not_fall_off_a_cliff(pkg, model):
    """Round 5's sweep lost 27 % of its mel-frames/s going from 8 to 9 chunks (the two-launch engine's 26 us latency chain); with the
    16-slot persistent kernel the loop's frames/s must grow with the chunk count through 16 (a 3 % dip at the engine boundary 8 -> 9 is
    box-to-box noise, a cliff is not)."""
    fps = {B: B / _us_per_step(pkg, model, B, n=200) for B in (8, 9, 12, 16)}
    assert fps[9] >= 0.97 * fps[8], fps
    assert fps[12] > fps[9] and fps[16] > fps[12], fps


def test_decoder_pair_with_the_stop_rule_compiled_in(pkg, model):
    """the gate-on instantiations (k_decoder_persistent<2, true, true>; the synthetic gate never fires, so both chunks run to the cap)
    are held to the same bound as the gate-less ones"""
    chunks = [wl.synth_ids(95, seed=10), wl.synth_ids(67, seed=11)]
    best = 1e9
    for _ in range(4):
        mels = model.infer_batch(chunks, opts=pkg.default_opts(dropout_seed=1, max_steps=300))
        best = min(best, model.last_timings()["decoder_ms"] * 1e3 / model.last_timings()["steps"])
    assert [x.shape[1] for x in mels] == [300, 300]
    assert best <= 15.5, "gate-on pair: %.1f us per step" % best


def test_vocoder_iteration_time(pkg):
    voc = pkg.create_griffin_lim(seed=3)
    S = wl.chirp_magnitude(1000)
    best = 1e9
    for _ in range(4):
        voc.infer_linear(S, iters=60)
        best = min(best, voc.last_timings()["iterations_ms"] * 1e3 / 61)
    assert best <= 5.8, "Griffin-Lim, F = 1000: %.2f us per iteration" % best
    voc.close()
