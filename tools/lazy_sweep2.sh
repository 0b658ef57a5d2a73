# developer sweep of the poll-delay knobs of the persistent decoder (x 256 clocks ~ 0.11 us each), one knob at a time
run() { echo -n "$1: "; env $1 timeout 120 python tools/persist_steps.py 2>&1 | grep "B=" | sed 's/ us per step.*//' | tr '\n' ' '; echo; }
run "XDTTS_LAZY_POLL=8"
for l in 6 7 9 10; do run "XDTTS_LAZY_POLL=$l"; done
for f in 3 5; do run "XDTTS_FIRST_POLL=$f"; done
for f in 1 3; do run "XDTTS_PFIRST=$f"; done
for c in 3 5; do run "XDTTS_CLAZY=$c"; done
for c in 3 5; do run "XDTTS_EFIRST=$c"; done
