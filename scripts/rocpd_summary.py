"""Summarise a rocprofv3 rocpd sqlite database: per-kernel calls / total / avg / min / max duration,
plus VGPR/LDS/scratch per dispatch.  Usage: python scripts/rocpd_summary.py <results.db> [> profiles/xxx.txt]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
q = "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows) or 1
print("%-86s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for n, c, t, a, mn, mx in rows:
    print("%-86s %6d %12.1f %12.2f %12.2f %12.2f %6.2f" % (n[:86], c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
extra = [c for c in ("vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "workgroup_size", "grid_size") if c in cols]
if extra:
    print()
    print("per-kernel resources (%s):" % ", ".join(extra))
    for r in cur.execute("select distinct %s, %s from kernels" % (name_col, ", ".join(extra))):
        print("  %-80s %s" % (r[0][:80], r[1:]))
