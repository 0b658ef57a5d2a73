#!/bin/bash
# On the GPU box: config3 iteration time (and a 9..16-chunk sweep with "sweep" as first argument) for the default library and
# every xd-tts_amd/libxdtts_hip_v_*.so
cd $GRAFT_REPO_ROOT
for f in xd-tts_amd/libxdtts_hip.so xd-tts_amd/libxdtts_hip_v_*.so; do
  [ -e "$f" ] || continue
  echo -n "== $f : "; XDTTS_LIB=$PWD/$f python tools/config3_batch.py 3 2>&1 | grep -E "wall" | sed 's/.*postnet [0-9.]* ms; //' | tr '\n' ' '
  if [ "$1" = sweep ]; then XDTTS_LIB=$PWD/$f timeout 200 python tools/batch_sweep.py 9 16 32 2>&1 | tail -n +2 | awk '{printf "%s ", $2}'; fi
  echo
done
