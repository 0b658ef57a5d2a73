#!/bin/bash
# Per-kernel stats of the configs[2] batch (rocprofv3 --kernel-trace) -> gpurun_out/r${R:-03}/$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r${R:-03}
rm -rf /tmp/prof_c3
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 -- python tools/config3_batch.py 0 > gpurun_out/r${R:-03}/prof_c3.log 2>&1
grep config3 -A1 gpurun_out/r${R:-03}/prof_c3.log | tail -2
python tools/rocprof_summary.py $(find /tmp/prof_c3 -name "*.db" | head -1) | cut -c1-70,100-170 | tee gpurun_out/r${R:-03}/$1 | head -${2:-14}
