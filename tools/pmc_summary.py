"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (second episode only when a marker is
absent: all dispatches are averaged).  usage: pmc_summary.py results.db COUNTER out.json"""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
want = sys.argv[2]
q = "select name, count(*), sum(counter_value), avg(duration), max(counter_value) from pmc_events where counter_name = ? group by name order by 3 desc"
out = {n: dict(calls=k, total=v, avg=v / k, avg_duration_ns=d, max=mx) for n, k, v, d, mx in c.execute(q, (want,))}
json.dump(out, open(sys.argv[3], "w"), indent=0)
print("%-72s %6s %14s %14s" % ("kernel", "calls", "avg " + want, "total"))
for n, d in out.items():
    print("%-72s %6d %14.1f %14.1f" % (n[:72], d["calls"], d["avg"], d["total"]))
