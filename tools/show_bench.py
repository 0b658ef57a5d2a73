"""Print the interesting parts of a bench.py JSON line."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.0f %s  ms/step %.3f  x_realtime %.0f" % (d["value"], d["unit"], d["ms_per_step"], d["x_realtime"]))
print("phases", {k: round(v, 3) for k, v in d["phase_ms_per_utterance"].items()})
r = d["roofline"]; print("roofline frac %.3f  us/step %.2f  traffic %s" % (r["frac"], r["us_per_step"], r["traffic"]))
e = d.get("extra") or {}
for k in ("headline_one_call_per_utterance", "headline_gate_on", "headline_30_iterations"):
    if k in e and "mel_frames_per_s" in e[k]:
        print("%s: %.0f frames/s, %.3f ms per utterance" % (k, e[k]["mel_frames_per_s"], e[k]["ms_per_utterance"]))
if "config3" in e:
    c = e["config3"]; print("config3: %.0f frames/s mel-gen, %.1f us/iteration, ms %s, mfma frac %.3f hbm frac %.3f" % (c["mel_frames_per_s_mel_gen"], c["us_per_lockstep_iteration"], {k: round(v, 2) for k, v in c["ms"].items()}, c["roofline"]["frac"], c["roofline"]["hbm_frac"]))
if "config4" in e:
    c = e["config4"]; print("config4: %.1f utt/s, %.0f frames/s, rtf %.5f, vocoder ms %.1f" % (c["utterances_per_s"], c["mel_frames_per_s"], c["rtf"], c["vocoder_ms_this_rank"]), c.get("device_ms_this_rank"), (c.get("two_call_form") or {}).get("seconds_this_rank"))
if "config5" in e:
    print("config5:", [(x["iterations"], round(x["us_per_iteration"], 2), round(x["roofline"]["frac"], 3)) for x in e["config5"]["runs"]])
cb = d.get("cpu_baseline")
if cb:
    print("cpu 1-core %.0f frames/s (%d host cores); omp %s; torch %s" % (cb["value"], cb["host_cores"], cb.get("all_cores_openmp"), cb.get("all_cores_torch")))
