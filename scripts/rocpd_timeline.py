"""Timeline of a rocprofv3 rocpd database: the kernel dispatches in start order with the idle time in front of each (start - the
latest end so far), then per kernel name: calls, average duration, average idle time in front.
usage: python scripts/rocpd_timeline.py <results.db> [<dispatches to list from the end>]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute("select %s, start, end from kernels order by start" % name_col))
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 40


def short(n):
    n = re.sub(r"^void ", "", n).replace("flacgpu::", "")
    return re.sub(r"\(.*", "", n)[:60]


agg = defaultdict(lambda: [0, 0.0, 0.0])
last_end = None
lines = []
for n, s, e in rows:
    gap = 0.0 if last_end is None else (s - last_end) / 1e3
    a = agg[short(n)]
    a[0] += 1; a[1] += (e - s) / 1e3
    if gap < 200.0:                               # (longer: the host was doing something else -- not a launch boundary)
        a[2] += max(gap, 0.0)
    lines.append("%-60s dur %9.2f us   idle before %8.2f us" % (short(n), (e - s) / 1e3, gap))
    last_end = e if last_end is None else max(last_end, e)
print("\n".join(lines[-nlist:]))
print()
print("%-60s %6s %10s %14s" % ("kernel", "calls", "avg_us", "avg idle before"))
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %6d %10.2f %14.2f" % (k, c, d / c, g / c))
