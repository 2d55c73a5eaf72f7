"""Per-kernel durations INSIDE the pipelined bench loop from a rocprofv3 kernel trace, in launch order of one steady-state step,
with what else ran at the same time:  python tools/inloop_vs_alone.py <kernel_trace.csv>
For every kernel of the step between two mesh-kernel launches: start (us from the step's beginning), duration, stream (queue),
and the names of kernels on OTHER queues that overlap it for more than 20 % of its duration."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
short = lambda n: n.split("(")[0].replace("void ", "").replace("hps::", "")[:44]
mesh = [e for e in ev if "mesh_fused_kernel" in e[2]]
k = min(6, len(mesh) - 2)
s0, s1 = mesh[k][0], mesh[k + 1][0]
step = [e for e in ev if s0 <= e[0] < s1]
print("one steady-state step: %.3f ms, %d kernels" % ((s1 - s0) / 1e6, len(step)))
for s, e, name, q in step:
    d = e - s
    if d < 4000:
        continue
    others = {}
    for s2, e2, n2, q2 in ev:
        if q2 == q or e2 <= s or s2 >= e:
            continue
        ov = min(e, e2) - max(s, s2)
        if ov > 0.2 * d:
            others[short(n2)] = others.get(short(n2), 0) + ov
    print("%8.1f +%7.1f us  q%-3s %-44s | %s" % ((s - s0) / 1e3, d / 1e3, q, short(name), ", ".join("%s %.0f%%" % (n, 100 * o / d) for n, o in others.items())))
