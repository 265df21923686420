#!/usr/bin/env python3
"""Kernel + copy timeline of the last substep in a rocprofv3 --kernel-trace --memory-copy-trace CSV pair. argv: <trace dir> <pid prefix of the csv files> <events to print>"""
import csv,sys
def load(d,pid):
    ks=[r for r in csv.DictReader(open(f"{d}/runc/{pid}_kernel_trace.csv.tail")) if r["Start_Timestamp"].isdigit()]
    cs=[r for r in csv.DictReader(open(f"{d}/runc/{pid}_memory_copy_trace.csv.tail")) if r["Start_Timestamp"].isdigit()]
    ev=[]
    for r in ks: ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K q"+r["Queue_Id"],r["Kernel_Name"].split("(")[0].replace("hns::","")[:70], r["Grid_Size_X"]))
    for r in cs: ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"C",r.get("Direction","")[12:], ""))
    ev.sort()
    return ev
d,pid,n=sys.argv[1],sys.argv[2],int(sys.argv[3])
ev=load(d,pid)
idx=[i for i,e in enumerate(ev) if "k_advect_vector" in e[3]]
i0=idx[-2]; i1=idx[-1]
t0=ev[i0][0]
print("substep span us", (ev[i1][0]-t0)/1e3, "events", i1-i0)
for e in ev[i0:i0+n]:
    print(f"{(e[0]-t0)/1e3:9.1f} +{(e[1]-e[0])/1e3:6.1f} {e[2]:6s} {e[4]:>8} {e[3]}")
