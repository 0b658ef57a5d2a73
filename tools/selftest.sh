#!/bin/bash
# Runs every developer aid of tools/ once with small arguments on the GPU box and says which still work against the current
# library (the "last ran" column of tools/README.md).  usage: tools/selftest.sh [prof]   (prof: also the profile-build tools)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out; OUT=gpurun_out/tools_selftest.txt; : > $OUT
run() { local t0=$(date +%s.%N); local name="$1"; shift; if timeout 300 "$@" > /tmp/selftest.log 2>&1; then s=ok; else s="FAIL($?)"; fi
        printf "%-34s %-9s %5ss  %s\n" "$name" "$s" "$(( $(date +%s) - ${t0%.*} ))" "$(grep -v amdgpu.ids /tmp/selftest.log | tail -1 | cut -c1-110)" | tee -a $OUT; }
run batch_sweep.py            python tools/batch_sweep.py 1 2 3 9
run config3_batch.py          python tools/config3_batch.py 1
run config5_griffinlim.py     python tools/config5_griffinlim.py
run gate_rig_check.py         python tools/gate_rig_check.py
run gemm_bench.py             python tools/gemm_bench.py 200 2
run gl_check.py               python tools/gl_check.py
run gl_hash.py                python tools/gl_hash.py
run gl_poll_sweep.py          python tools/gl_poll_sweep.py 1000 6
run gl_tf_sweep.py            python tools/gl_tf_sweep.py
run headline_call.py          python tools/headline_call.py 3
run p8_check.py               python tools/p8_check.py
run p8_rows.py                python tools/p8_rows.py
run p8_slope.py               python tools/p8_slope.py
run persist_check.py          python tools/persist_check.py
run persist_setup.py          python tools/persist_setup.py
run persist_steps.py          python tools/persist_steps.py
run step_variants.py          python tools/step_variants.py
run vocoder_batch.py          python tools/vocoder_batch.py
run vocoder_shapes.py         python tools/vocoder_shapes.py
run synthesize.py             python tools/synthesize.py --help
run parity_table.py           python tools/parity_table.py
run headline_timeline.sh      bash tools/headline_timeline.sh
run config3_gaps.sh           bash tools/config3_gaps.sh
run gemm_prof.sh              bash tools/gemm_prof.sh 200
for u in allgather bulk_edge chain condload edges edges5 pingpong; do
  run ubench_$u.hip bash -c "hipcc --offload-arch=gfx950 -O3 -I xd-tts_amd/csrc -o /tmp/ub_$u tools/ubench_$u.hip && timeout 120 /tmp/ub_$u $( [ $u = allgather ] && echo 200 )"
done
if [ "$1" = prof ]; then
  export XDTTS_LIB=$PWD/xd-tts_amd/libxdtts_hip_prof.so
  run gl_profile.py           python tools/gl_profile.py 1000 60
  run p8_profile.py           python tools/p8_profile.py 4
  run persist_profile.py      python tools/persist_profile.py
  run persist_profile_skew.py python tools/persist_profile_skew.py
fi
