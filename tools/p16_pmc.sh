#!/bin/bash
# HBM traffic and duration of the 16-slot persistent MFMA decoder launch (k_decoder_persistent16): rocprofv3 kernel trace + the two
# PMC passes (FETCH_SIZE, WRITE_SIZE; each its own run) of a 12-chunk batch of 200 steps -> gpurun_out/r${R:-06}/p16_pmc.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r${R:-06}; mkdir -p $OUT
CMD="python tools/batch_sweep.py 12"
rm -rf /tmp/p16_k /tmp/p16_f /tmp/p16_w
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p16_k -o k -- $CMD > $OUT/p16_prof_kernel.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p16_f -o f -- $CMD > $OUT/p16_prof_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p16_w -o w -- $CMD > $OUT/p16_prof_write.log 2>&1
{ echo "k_decoder_persistent16, 12 chunks x 200 steps (tools/batch_sweep.py 12: two launches), git ${1:-unknown}"
  python tools/rocprof_summary.py $(find /tmp/p16_k -name "*.db" | head -1) | grep -E "kernel|persistent16|k_p8_seed|fillBuffer" | cut -c1-70,100-170
  echo "FETCH_SIZE (KB per dispatch; gfx950: x 2 for wide coalesced reads, MI355X_MICROARCH.md):"
  python tools/pmc_kernels.py $(find /tmp/p16_f -name "*.db" | head -1) persistent16
  echo "WRITE_SIZE (KB per dispatch):"
  python tools/pmc_kernels.py $(find /tmp/p16_w -name "*.db" | head -1) persistent16
  echo "algorithmic bytes (SURVEY 8d): 200 steps x (72 759 876 + 12 x 256 000) = 15.2 GB per launch; the LSTM weights (71.3 MB) are read once per launch into the register files,"
  echo "what moves per step is the rings: 12 chunks x 11.3 kB written once, read by 256 workgroups out of L2 / the Infinity Cache."
} > $OUT/p16_pmc.txt 2>&1
cat $OUT/p16_pmc.txt
