run() { echo -n "$* : "; env "$@" timeout 200 python tools/batch_sweep.py 1 2 2>&1 | tail -2 | awk '{printf "%s ", $2} END{print ""}'; }
run A=0
for v in 2 3 4 5 6; do run XDTTS_PFIRST=$v; done
for v in 3 5 6; do run XDTTS_XFIRST=$v; done
for v in 3 5 6; do run XDTTS_FIRST_POLL=$v; done
for v in 2 4; do run XDTTS_XLAZY=$v; done
for v in 6 12; do run XDTTS_LAZY_POLL=$v; done
for v in 2 4 5; do run XDTTS_EFIRST=$v; done
