"""Per-kernel averages of arbitrary rocprofv3 --pmc counters from a rocpd database.
usage: pmc_kernels.py DB [name-substring ...]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pats = sys.argv[2:]
rows = {}
for name, counter, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    k = m.group(1) if m else name[:40]
    if pats and not any(p in k for p in pats):
        continue
    rows.setdefault(k, {})[counter] = (n, avg)
for k, c in sorted(rows.items()):
    print(k, " calls", next(iter(c.values()))[0])
    for name, (n, avg) in sorted(c.items()):
        print("    %-28s %16.1f" % (name, avg))
