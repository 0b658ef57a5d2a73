# developer sweep of the poll-delay knobs of the persistent decoder (x 512 clocks ~ 0.21 us each), one knob at a time
run() { echo -n "$1: "; env $1 timeout 120 python tools/persist_steps.py 2>&1 | grep "B=" | sed 's/ us per step.*//' | tr '\n' ' '; echo; }
run "XDTTS_LAZY_POLL=4"
for l in 0 2 6 8; do run "XDTTS_LAZY_POLL=$l"; done
for f in 0 1 3 4; do run "XDTTS_FIRST_POLL=$f"; done
for x in 0 1 3 4 6; do run "XDTTS_XLAZY=$x"; done
for c in 0 1 3 4 6; do run "XDTTS_CLAZY=$c"; done
