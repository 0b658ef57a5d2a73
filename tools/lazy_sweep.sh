# developer sweep of the poll-delay knobs of the persistent decoder (x 512 clocks ~ 0.21 us each)
for f in 0 1 2 3; do for l in 2 4 8; do echo -n "first=$f lazy=$l: "; XDTTS_FIRST_POLL=$f XDTTS_XLAZY=$f XDTTS_CLAZY=$f XDTTS_LAZY_POLL=$l timeout 120 python tools/persist_check.py 2>&1 | grep -E "B=1 fixed 400|B=2 both" | sed 's/.*persistent \([0-9.]*\).*/\1/' | tr '\n' ' '; echo; done; done
