#!/bin/bash
# developer aid: A/B the non-temporal weight-load masks (libs prebuilt as libxdtts_nt<m>.so)
cd "$(dirname "$0")/../xd-tts_amd"
cp libxdtts_hip.so /tmp/orig.so
for m in 0 2 3 1 0 2; do cp libxdtts_nt$m.so libxdtts_hip.so; echo -n "nt mask $m: "; python ../tools/mix_timing.py paqsdj 2>&1 | grep -v amdgpu; done
cp /tmp/orig.so libxdtts_hip.so
