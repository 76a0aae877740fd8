"""Per-(kernel, grid, workgroup) average duration from a rocprofv3 kernel_trace.csv.
usage: shape_times.py kernel_trace.csv [name-regex]"""
import csv, re, sys
rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
agg = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if rx and not rx.search(n): continue
    key = (n.split("(")[0][:48], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("%-48s %9s %6s %5s %5s %6s %9s %10s" % ("kernel", "grid.x", "y", "z", "wg", "calls", "avg us", "total us"))
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-48s %9s %6s %5s %5s %6d %9.2f %10.1f" % (*k, c, d / c / 1e3, d / 1e3))
