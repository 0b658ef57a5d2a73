"""Markdown table of the measured parity margins for DESIGN.md section 2, generated from
gpurun_out/parity_fullsize.json -- the file tests/test_gpu_fullsize.py writes on the GPU box.
usage: python tools/parity_table.py [gpurun_out/parity_fullsize.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_fullsize.json")))
f = lambda x: "%.1e" % x
print("| case (tests/test_gpu_fullsize.py) | GPU − f32 oracle | GPU − f64 oracle | f32 oracle − f64 oracle |")
print("|---|---|---|---|")
print("| configs[2] mel, worst of 10 spot-checked chunks of the 52-chunk batch (RMS) | %s | | |" % f(d["config3_mel_rms_worst_of_10_chunks"]))
c2 = d["config2_end_to_end"]
print("| configs[1] mel, 800 frames (RMS) | %s | | |" % f(c2["mel_rms"]))
print("| configs[1] mel → linear S, relative RMS | %s | | |" % f(c2["S_rel_rms"]))
print("| configs[1] audio, 60 free-running iterations (signal RMS %.2f) | %s | %s | %s |" % (c2["audio_signal_rms"], f(c2["audio_gpu_vs_f32"]), f(c2["audio_gpu_vs_f64"]), f(c2["audio_f32_vs_f64"])))
for F in (800, 1000):
    for i, what in ((0, "from the initial phase"), (1, "from the oracle's state after 10 iterations")):
        s = d["gl_step_F%d_state%d" % (F, i)]
        print("| Griffin-Lim ONE teacher-forced iteration, F = %d, %s: rebuilt spectrum, relative RMS | %s | %s | %s |" % (F, what, f(s["rebuilt_rel_rms_vs_f32"]), f(s["rebuilt_rel_rms_vs_f64"]), f(s["f32_vs_f64_rel_rms"])))
        print("| … unit-modulus angles, RMS (ill-conditioned where \\|a\\| ≈ 0) | %s | %s | %s |" % (f(s["angles_rms_vs_f32"]), f(s["angles_rms_vs_f64"]), f(s["angles_f32_vs_f64"])))
for it in (30, 60, 120):
    s = d["gl_audio_F1000_it%d" % it]
    print("| configs[4] audio, F = 1000, %d free-running iterations (signal RMS %.2f) | %s | %s | %s |" % (it, s["signal_rms"], f(s["gpu_vs_f32"]), f(s["gpu_vs_f64"]), f(s["f32_vs_f64"])))
if "config2_audio_30_iterations_gpu_vs_f32" in d:
    print("| configs[1] audio, 30 free-running iterations (the reference's setting, mod.rs:456) | %s | | |" % f(d["config2_audio_30_iterations_gpu_vs_f32"]))
for F in (800, 1000):
    k = "gl_step_trajectory_F%d_worst_rebuilt_rel_rms" % F
    if k in d:
        print("| Griffin-Lim ONE teacher-forced iteration along the oracle's trajectory (iterations 0, 1, 2, 5, 10, 20, 30, 45, 59), F = %d: worst rebuilt spectrum, relative RMS | %s | | |" % (F, f(d[k])))
for F, what in ((1000, "configs[4] input"), (800, "the chirp magnitude cut to 800 frames (recorded only)")):
    k = "gl_audio_F%d_vs_f32" % F
    if k in d:
        s = d[k]
        print("| free-running audio vs f32 oracle at 10 / 20 / 30 / 40 / 60 iterations, %s; first even iteration above 1e-4: %s | %s / %s / %s / %s / %s | | |" % (
            what, s["first_iteration_above_1e-4"], f(s["it10"]), f(s["it20"]), f(s["it30"]), f(s["it40"]), f(s["it60"])))
