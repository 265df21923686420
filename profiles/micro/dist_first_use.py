#!/usr/bin/env python3
"""Where does the first multi-rank step of a process spend its time? (diagnosis of a sporadic multi-minute first use)"""
import os, sys, time
import numpy as np
t00 = time.perf_counter()
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, dist as HD, fields
T = [("import", time.perf_counter() - t00)]
def lap(name, t0): T.append((name, time.perf_counter() - t0))
origins, R = fields.dense_leaves(32), 32
f = fields.synthetic_fields(origins, R)
t0 = time.perf_counter(); torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize(); lap("torch cuda init", t0)
t0 = time.perf_counter(); grid = api.create_grid_from_leaves(origins, 1.0 / R); lap("grid create", t0)
t0 = time.perf_counter(); sim = D.Sim(grid, ["density"]); sim.upload({"vel": f["vel"], "density": f["density"]}); lap("sim create+upload", t0)
t0 = time.perf_counter(); sim.core_substep(7, 1.0 / 24, 1.0 / R, D.current_stream()); torch.cuda.synchronize(); lap("single substep", t0)
t0 = time.perf_counter(); ranks = [HD.DistRank(origins, 2, r, 1.0 / R, n_scalars=1) for r in range(2)]; lap("DistRank x2", t0)
t0 = time.perf_counter(); HD.DistRank.connect_local(ranks); lap("connect_local", t0)
t0 = time.perf_counter()
for r, d in enumerate(ranks):
    d.upload(d.owned_voxels(f["vel"]), [d.owned_voxels(f["density"])])
lap("upload", t0)
st = int(torch.cuda.current_stream().cuda_stream)
t0 = time.perf_counter(); HD.DistRank.local_core_substep(ranks, 7, 1.0 / 24, st); lap("enqueue substep", t0)
t0 = time.perf_counter()
for d in ranks: d.synchronize(st)
lap("synchronize", t0)
t0 = time.perf_counter(); HD.DistRank.local_core_substep(ranks, 7, 1.0 / 24, st)
for d in ranks: d.synchronize(st)
lap("second substep", t0)
print(" | ".join(f"{n} {1e3 * t:.0f} ms" for n, t in T), flush=True)
