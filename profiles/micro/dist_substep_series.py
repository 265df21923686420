#!/usr/bin/env python3
"""Per-substep wall time (synchronised after every substep, and in bursts of 20 without) of one rank of config 5 alone, exchanged pressure loop, loopback.
argv: sweeps_per_exchange rank [option=value ...]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import device as D, dist as HD, fields
H.set_option("dist_mirror", "0")
for a in sys.argv[3:]:
    H.set_option(*a.split("=", 1))
k, rank = int(sys.argv[1]), int(sys.argv[2])
origins, R = fields.config_leaves("plume1024")
d = HD.DistRank(origins, 8, rank, 1.0 / R, n_scalars=1, sweeps_per_exchange=k)
d.connect_loopback()
g = fields.synthetic_fields(origins[d.owned_ids], R)
d.upload(g["vel"], [g["density"]])
st = D.current_stream()
series = []
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.core_substep(50, 1.0 / 24.0, st)
    torch.cuda.synchronize(); series.append(round(1e3 * (time.perf_counter() - t0), 3))
bursts, enq = [], []
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        d.core_substep(50, 1.0 / 24.0, st)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); bursts.append(round(1e3 * (time.perf_counter() - t0) / 20, 3)); enq.append(round(1e3 * (t1 - t0) / 20, 3))
print(json.dumps({"k": k, "rank": rank, "opts": sys.argv[3:], "synced_ms": series, "burst_of_20_ms_per_substep": bursts, "host_enqueue_ms_per_substep": enq}))
