"""Concurrency of a multi-slot run from a rocprofv3 rocpd database: span, union of busy intervals, sum of kernel
durations (their ratio = average number of kernels in flight), and per kernel the average duration next to the average
of an eager profile (how much a kernel stretches when it shares the chip).
usage: prof_overlap.py multi.db [eager_summary.txt] [t_from_fraction]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = list(c.execute("select name, start, end from kernels order by start"))
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
t0 = rows[0][1] + (rows[-1][1] - rows[0][1]) * frac  # skip capture / warm-up: the last part of the run only
rows = [r for r in rows if r[1] >= t0]
span = max(r[2] for r in rows) - rows[0][1]
busy, cs, ce = 0, None, None
for _, s, e in rows:
    if ce is None or s > ce:
        if ce is not None: busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = sum(e - s for _, s, e in rows)
print("window %.1f ms: busy(union) %.1f ms (%.1f %%), sum of kernel durations %.1f ms -> %.2f kernels in flight while busy" % (
    span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, tot / busy))
eager = {}
if len(sys.argv) > 2:
    for l in open(sys.argv[2]):
        f = l.split()
        if len(f) >= 5 and "r3d_" in l[:16]:
            try: eager[l[:72].strip()] = float(f[-2])
            except ValueError: pass
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n[:72].strip(), [0, 0]); a[0] += 1; a[1] += e - s
print("%-72s %7s %9s %9s %7s" % ("kernel", "calls", "total ms", "avg us", "x eager"))
for n, (k, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    x = (d / k / 1e3) / eager[n] if n in eager and eager[n] > 0 else float("nan")
    print("%-72s %7d %9.2f %9.2f %7.2f" % (n, k, d / 1e6, d / k / 1e3, x))
