#!/usr/bin/env python3
"""Per-kernel stage times of the core substep (hipEvents between the kernels) for the library in HNS_LIBRARY. argv: config [iterations]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, fields
config = sys.argv[1] if len(sys.argv) > 1 else "256"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
origins, R = fields.config_leaves(config)
vs = 1.0 / R
f = fields.synthetic_fields(origins, R)
sim = D.Sim(api.create_grid_from_leaves(origins, vs), ["density"])
sim.upload({"vel": f["vel"], "density": f["density"]})
st = D.current_stream()
best = {}
for rep in range(3):
    for _ in range(2):
        sim.core_substep(iters, 1.0 / 24.0, vs, st)
    n = 10
    sim.stage_timing(n)
    for _ in range(n):
        sim.core_substep(iters, 1.0 / 24.0, vs, st)
    torch.cuda.synchronize()
    t, k = sim.stage_times()
    for s, ms in t.items():
        best[s] = min(best.get(s, 1e9), 1e3 * ms / k)
print(os.path.basename(os.environ.get("HNS_LIBRARY", "libhns.so")), config, {s: round(v, 1) for s, v in best.items()})
