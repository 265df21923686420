#!/usr/bin/env python3
"""Device-resident full Compute_Sim substep (combustion, buoyancy, [vorticity], 5 advected fields) at 256^3: ms per substep.
Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, fields

cfg = sys.argv[1] if len(sys.argv) > 1 else "256"
fs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
origins, R = fields.config_leaves(cfg)
vs = 1.0 / R
f = fields.synthetic_fields(origins, R)
grid = api.create_grid_from_leaves(origins, vs)
names = ["density", "temperature", "fuel", "waste", "flame"]
sim = D.Sim(grid, names)
sim.upload({"vel": f["vel"], **{n: f[n] for n in names}})
p = api.CombustionParams(factorScale=fs)
st = D.current_stream()
for _ in range(3):
    sim.substep(50, 1.0 / 24.0, vs, p, False, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    sim.substep(50, 1.0 / 24.0, vs, p, False, st)
torch.cuda.synchronize()
print({"config": cfg, "factorScale": fs, "ms_per_full_substep": round(1e2 * (time.perf_counter() - t0), 3)})
