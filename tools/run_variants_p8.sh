#!/bin/bash
# On the GPU box: us per iteration of the 3..8-chunk engine for the default library and every xd-tts_amd/libxdtts_hip_v_*.so
cd $GRAFT_REPO_ROOT
for f in xd-tts_amd/libxdtts_hip.so xd-tts_amd/libxdtts_hip_v_*.so; do
  [ -e "$f" ] || continue
  echo -n "$f : "; XDTTS_LIB=$PWD/$f timeout 200 python tools/batch_sweep.py ${@:-4 8} 2>&1 | tail -n +2 | awk '{printf "%s ", $2} END{print ""}'
done
