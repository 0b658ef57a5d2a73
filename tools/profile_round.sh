#!/bin/bash
# Round profile on the GPU box: per-kernel stats (rocprofv3 --kernel-trace --stats) and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; each its own run with --kernel-trace only, as gpurun requires) of the SAME
# bench command, the in-kernel phase clocks, and the bench line itself -> gpurun_out/rNN/, from where the
# summaries are copied into profiles/.   usage: tools/profile_round.sh NN GIT_HEAD
R=${1:-02}; HEAD=${2:-unknown}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r$R; mkdir -p $OUT
make -C xd-tts_amd prof -j16 > $OUT/make_prof.log 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"   # (rocprofv3 dumps core at exit of this process, after its databases are written: harmless)
rm -rf /tmp/prof_k /tmp/prof_f /tmp/prof_w
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- $CMD > $OUT/prof_kernel.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_k -name "*.db" | head -1) > $OUT/kernel_stats.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- $CMD > $OUT/prof_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- $CMD > $OUT/prof_write.log 2>&1
python tools/pmc_summary.py $(find /tmp/prof_f -name "*.db" | head -1) $(find /tmp/prof_w -name "*.db" | head -1) $OUT/pmc_hbm_traffic $R "$CMD" $HEAD > /dev/null
timeout 200 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
XDTTS_LIB=xd-tts_amd/libxdtts_hip_prof.so timeout 100 python tools/gl_profile.py 1000 60 > $OUT/gl_phase_clocks.txt 2>&1
XDTTS_LIB=xd-tts_amd/libxdtts_hip_prof.so timeout 100 python tools/gl_profile.py 800 60 >> $OUT/gl_phase_clocks.txt 2>&1
XDTTS_LIB=xd-tts_amd/libxdtts_hip_prof.so timeout 100 python tools/persist_profile.py > $OUT/persistent_phase_clocks.txt 2>&1
head -30 $OUT/kernel_stats.txt | cut -c1-60,100-170; tail -8 $OUT/pmc_hbm_traffic.txt | cut -c1-400; python tools/show_bench.py $OUT/bench_n1.json
