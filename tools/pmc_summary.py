#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each `--kernel-trace --pmc X`
only, as gpurun requires) into profiles/rNN_pmc_hbm_traffic.{txt,json}.

usage: pmc_summary.py FETCH.db WRITE.db OUT_PREFIX ROUND "<command that was profiled>"

Units and correction (MI355X_MICROARCH.md, HBM section): rocprofv3 reports KB per dispatch;
on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced streaming reads at 64 B, so
corrected_fetch = 2 x FETCH_SIZE.  WRITE_SIZE is left uncorrected.  bench.py reads
`decoder_step_traffic_bytes` from the newest JSON for its roofline.traffic field.
"""
import json
import re
import sqlite3
import sys

STEP_KERNELS = ("k_prenet", "k_lstm<1792, 0>", "k_qenergy", "k_softmax_ctx", "k_lstm<2560, 1>")
ALGORITHMIC_STEP_BYTES = 73132835.0  # weights + per-step state, one chunk of 95 ids


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", name)
    return m.group(1) if m else name[:60]


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    out = {}
    for name, n, avg in db.execute(
            "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name",
            (counter,)):
        k = short(name)
        c, a = out.get(k, (0, 0.0))
        out[k] = (c + n, (a * c + avg * n) / (c + n))
    return out


def main(fdb, wdb, prefix, rnd, cmd):
    f = per_kernel(fdb, "FETCH_SIZE")
    w = per_kernel(wdb, "WRITE_SIZE")
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), " + cmd,
             "Units: rocprofv3 reports KB per dispatch. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B",
             "requests as 64 B for wide coalesced reads -> corrected_fetch = 2 x FETCH_SIZE.  WRITE_SIZE uncorrected.", "",
             "%-64s %6s %14s %18s %14s" % ("kernel", "calls", "FETCH_SIZE_KB", "corrected_fetch_MB", "WRITE_SIZE_KB")]
    for k in sorted(f, key=lambda k: -f[k][0] * f[k][1]):
        lines.append("%-64s %6d %14.1f %18.3f %14.1f" % (k, f[k][0], f[k][1], 2 * f[k][1] * 1024 / 1e6, w.get(k, (0, 0.0))[1]))
    missing = [k for k in STEP_KERNELS if k not in f]
    if missing:
        raise SystemExit("decoder-step kernels missing from the trace: %s" % missing)
    fetch_kb = sum(f[k][1] for k in STEP_KERNELS)
    write_kb = sum(w[k][1] for k in STEP_KERNELS)
    traffic = 2 * fetch_kb * 1024 + write_kb * 1024
    lines += ["", "decoder step (%d kernels: %s): corrected fetch %.2f MB + write %.2f MB = %.2f MB per step; "
              "algorithmic bytes per step %.2f MB -> traffic/algorithmic = %.3f" % (
                  len(STEP_KERNELS), ", ".join(STEP_KERNELS), 2 * fetch_kb * 1024 / 1e6, write_kb * 1024 / 1e6, traffic / 1e6,
                  ALGORITHMIC_STEP_BYTES / 1e6, traffic / ALGORITHMIC_STEP_BYTES)]
    open(prefix + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `%s`; FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)" % cmd,
               "decoder_step_kernels": list(STEP_KERNELS),
               "decoder_step_traffic_bytes": round(traffic),
               "decoder_step_fetch_kb_raw": round(fetch_kb, 1),
               "decoder_step_write_kb_raw": round(write_kb, 1),
               "round": int(rnd)}, open(prefix + ".json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:6])
