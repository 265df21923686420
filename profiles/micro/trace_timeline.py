#!/usr/bin/env python3
"""Timeline of one stretch of a rocprofv3 --kernel-trace CSV: per dispatch its queue, start (us from the first shown), duration
and the gap to the previous dispatch on the same queue. argv: <kernel_trace.csv> [first dispatch index] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = int(rows[first]["Start_Timestamp"])
last_end = {}
queues = {}
for r in rows[first:first + count]:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    name = r["Kernel_Name"].split("(")[0][-40:]
    print(f"q{q} {'          ' * q}{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:6.1f} us  gap {gap:6.1f}  grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?'):>8}  {name}")
