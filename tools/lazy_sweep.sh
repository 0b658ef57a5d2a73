# developer sweep of the poll-delay knobs of the persistent decoder (x 512 clocks ~ 0.21 us)
for x in 2 6 10 14; do for c in 2 6 10; do echo -n "xlazy=$x clazy=$c: "; XDTTS_XLAZY=$x XDTTS_CLAZY=$c timeout 120 python tools/persist_check.py 2>&1 | grep -E "B=1 fixed 400|B=2 both" | sed 's/.*persistent \([0-9.]*\).*/\1/' | tr '\n' ' '; echo; done; done
