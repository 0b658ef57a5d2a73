for v in 0 1 2 3 4; do echo "first=$v"; XDTTS_FIRST_POLL=$v timeout 120 python tools/persist_check.py 2>&1 | grep -E "B=1 fixed 400|B=2 fixed/id|B=2 both" | sed 's/.*| launch/launch/'; done
