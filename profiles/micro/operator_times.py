#!/usr/bin/env python3
"""Device time of the single-purpose kernels behind the AdvectIndexGrid / ProjectNonDivergent operators at 256^3."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
if len(sys.argv) > 1: H.set_option("advect", sys.argv[1])
origins, R = fields.config_leaves("256")
vs = 1.0 / R
f = fields.synthetic_fields(origins, R)
grid = api.create_grid_from_leaves(origins, vs)
u = torch.from_numpy(f["vel"]).cuda(); phi = torch.from_numpy(f["density"]).cuda(); out = torch.empty_like(phi)
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
print({"advect_scalar_us": round(t(lambda: D.advect_scalar(grid, u, phi, out, 1.0 / 24.0, float(R))), 1), "advect": H.get_option("advect")})
