"""Timeline of one steady-state bench step from a rocprofv3 kernel trace: python tools/timeline.py <kernel_trace.csv> [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
lbs = [e for e in ev if "mesh_fused_kernel" in e[2]] or [e for e in ev if "lbs_kernel" in e[2]]   # one per step
t0, t1 = lbs[3][0], lbs[-1][0]
n = len(lbs) - 4
sel = [e for e in ev if t0 <= e[0] < t1]
busy, cs, ce = 0, None, None
for s, e, _, _ in sel:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("steps %d, wall per step %.3f ms, GPU busy %.1f%%" % (n, (t1 - t0) / n / 1e6, 100 * busy / (t1 - t0)))
s0, s1 = lbs[5][0], lbs[6][0]
end = 0
for s, e, name, q in (x for x in ev if s0 <= x[0] < s1):
    nm = name.split("(")[0].replace("void ", "").replace("hps::", "")[-38:]
    if (e - s) > min_us * 1e3 or (end and s - end > 8000):
        print("%8.1f +%7.1f  q%s  gap %7.1f  %s" % ((s - s0) / 1e3, (e - s) / 1e3, q, (s - end) / 1e3 if end else 0, nm))
    end = max(end, e)
