import sys, importlib, hashlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
pkg = importlib.import_module("xd-tts_amd"); wl = importlib.import_module("xd-tts_amd.workloads")
import oracle
orc = oracle.Oracle("f32")
print("CUs", torch.cuda.get_device_properties(0).multi_processor_count, torch.cuda.get_device_name(0))
for F in (800, 1000):
    S = wl.chirp_magnitude(F)
    p0 = orc.phase_init(3, 513, F)
    voc = pkg.create_griffin_lim(iters=30, seed=3)
    for it in (1, 2, 4, 6, 30):
        a = voc.infer_linear(S, phase0=p0, iters=it)
        b = voc.infer_linear(S, phase0=p0, iters=it)
        ref = orc.griffinlim(S, phase0=p0, iters=it)
        print(F, it, hashlib.sha1(a.tobytes()).hexdigest()[:12], np.array_equal(a, b), "vs f32 oracle rms %.3e" % float(np.sqrt(np.mean((a.astype(np.float64) - ref) ** 2))), "oracle sha", hashlib.sha1(ref.tobytes()).hexdigest()[:12])
    os.environ["XDTTS_GL"] = "launch"
    v2 = pkg.create_griffin_lim(iters=30, seed=3)
    c = v2.infer_linear(S, phase0=p0, iters=30)
    print(F, "launch-per-iteration engine, 30 it:", hashlib.sha1(c.tobytes()).hexdigest()[:12], "vs f32 oracle rms %.3e" % float(np.sqrt(np.mean((c.astype(np.float64) - orc.griffinlim(S, phase0=p0, iters=30)) ** 2))))
    del os.environ["XDTTS_GL"]
