#!/bin/bash
# GPU idle gaps of the LAST configs[2] batch call (rocprofv3 --kernel-trace --memory-copy-trace + tools/gap_report.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_g
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_g -o g -- python tools/config3_batch.py 3 > /tmp/prof_g.log 2>&1
grep -A1 "^config3" /tmp/prof_g.log
python tools/gap_report.py $(find /tmp/prof_g -name "*.db" | head -1)
