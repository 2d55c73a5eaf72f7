import csv
rows=list(csv.DictReader(open('gpurun_out/prof_tmp/bench_kernel_trace.csv')))
ev=sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r["Queue_Id"], r["Stream_Id"]) for r in rows)
mesh=[e for e in ev if "mesh_fused" in e[2]]
for k in (5,6,7):
    s0=mesh[k][0]
    for e in ev:
        if s0-30e3 <= e[0] < s0+720e3 and e[3]==mesh[k][3]:
            print("%9.1f +%7.1f q%s s%s %s"%((e[0]-s0)/1e3,(e[1]-e[0])/1e3,e[3],e[4],e[2]))
    print()
