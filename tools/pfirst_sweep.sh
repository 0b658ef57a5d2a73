# developer sweep: first-poll delays of the role workgroups (x 512 clocks), product build
run() { echo -n "$1: "; env $1 timeout 120 python tools/persist_steps.py 2>&1 | grep "B=" | sed 's/ us per step.*//' | tr '\n' ' '; echo; }
for f in 2 0 1 3; do run "XDTTS_EFIRST=$f"; done
