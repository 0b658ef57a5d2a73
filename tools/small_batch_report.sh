#!/bin/bash
# On the GPU box: the evidence files of the 3..8-chunk engine (csrc/decoder_persistent8.hip) -> gpurun_out/r04/
#   batch sweep with the engine on and off, its in-kernel phase clocks (profile build), the all-gather edge alone
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04; O=gpurun_out/r04
{ echo "# tools/batch_sweep.py, default library"; timeout 300 python tools/batch_sweep.py 1 2 3 4 5 6 7 8 9 12 16 2>&1 | grep -v amdgpu.ids
  echo "# the same with XDTTS_P8=0 (3..8 chunks on the engines that served them before)"; XDTTS_P8=0 timeout 300 python tools/batch_sweep.py 3 4 5 6 7 8 2>&1 | grep -v amdgpu.ids; } > $O/small_batch_sweep.txt
XDTTS_LIB=xd-tts_amd/libxdtts_hip_prof.so timeout 300 python tools/p8_profile.py 3 4 8 2>&1 | grep -v amdgpu.ids > $O/small_batch_phase_clocks.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_allgather tools/ubench_allgather.hip 2>/dev/null && timeout 300 /tmp/ubench_allgather 2000 > $O/allgather_edge.txt
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p8prof -o p8 -- python $GRAFT_REPO_ROOT/tools/batch_sweep.py 4 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $(find /tmp/p8prof -name "*.db" | head -1) > $O/small_batch_kernel_stats.txt 2>&1
tail -n 40 $O/small_batch_sweep.txt
