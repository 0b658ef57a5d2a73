#!/bin/bash
# On the GPU box: config3 iteration time for the default library and every xd-tts_amd/libxdtts_hip_v_*.so
cd $GRAFT_REPO_ROOT
echo "== default"; python tools/config3_batch.py 3 2>&1 | tail -1
for f in xd-tts_amd/libxdtts_hip_v_*.so; do
  [ -e "$f" ] || continue
  echo "== $f"; XDTTS_LIB=$PWD/$f python tools/config3_batch.py 3 2>&1 | grep -E "probe|wall" | sort | uniq | tail -${1:-1}
done
