"""Per-kernel PMC counter averages from a rocprofv3 rocpd sqlite db. usage: rocpd_pmc.py <db> [kernel-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
# columns usually: ..., kernel_name / name, counter_name, value
name_col = "kernel_name" if "kernel_name" in cols else "name"
q = "select %s, counter_name, count(*), avg(value), sum(value) from counters_collection group by %s, counter_name" % (name_col, name_col)
for n, c, k, a, s in cur.execute(q):
    if sub in n:
        print("%-50s %-28s n=%-4d avg=%.6g" % (n[:50], c, k, a))
