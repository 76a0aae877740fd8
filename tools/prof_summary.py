"""Summarise a rocprofv3 rocpd database: per-kernel totals and the busy fraction of the GPU (union of kernel
intervals / span).  usage: prof_summary.py results.db [top_n] [name-filter for the window start]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(c.execute("select name, start, end from kernels order by start"))
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy, cur_s, cur_e = 0, None, None
for _, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for _, s, e in rows)
print("span %.1f ms  busy(union) %.1f ms (%.1f%%)  sum of kernel durations %.1f ms  launches %d" % (
    (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / 1e6, len(rows)))
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
print("%-72s %8s %10s %9s %6s" % ("kernel", "calls", "total ms", "avg us", "%"))
for n, (k, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-72s %8d %10.3f %9.2f %6.1f" % (n[:72], k, d / 1e6, d / k / 1e3, 100.0 * d / tot))
# optional 4th argument: a kernel-name substring -> its launches by grid size (which call of a step is the slow one)
if len(sys.argv) > 4:
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gcols = [x for x in ("grid_x", "grid_size_x", "grid") if x in cols]
    if gcols:
        g = {}
        for n, s, e, gx in c.execute("select name, start, end, %s from kernels order by start" % gcols[0]):
            if sys.argv[4] in n:
                a = g.setdefault((n[:48], gx), [0, 0]); a[0] += 1; a[1] += e - s
        for (n, gx), (k, d) in sorted(g.items(), key=lambda kv: -kv[1][1]):
            print("  %-48s grid %9s  calls %4d  avg %9.2f us" % (n, gx, k, d / k / 1e3))
    else:
        print("  (no grid column among %s)" % cols)
