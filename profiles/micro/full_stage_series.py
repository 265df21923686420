#!/usr/bin/env python3
"""Per-substep stage times (hipEvents, hns_sim_stage_timing) of the full Compute_Sim substep over the first N substeps behind an upload: do the kernels' times depend on how far the
simulation has run? argv: [config=256] [substeps=40] [fuse=1]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
cfg = sys.argv[1] if len(sys.argv) > 1 else "256"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
H.set_option("fuse", sys.argv[3] if len(sys.argv) > 3 else "1")
names = ["density", "temperature", "fuel", "waste", "flame"]
origins, R = fields.config_leaves(cfg)
vs = 1.0 / R
f = fields.synthetic_fields(origins, R)
sim = D.Sim(api.create_grid_from_leaves(origins, vs), names)
sim.upload({"vel": f["vel"], **{k: f[k] for k in names}})
prm = api.CombustionParams(vorticityScale=0.0)
st = D.current_stream()
rows = []
for i in range(n):
    sim.stage_timing(1)
    sim.substep(50, 1.0 / 24.0, vs, prm, False, st)
    torch.cuda.synchronize()
    ms, k = sim.stage_times()
    rows.append([round(1e3 * ms[s]) for s in ("advect_vector", "divergence", "pressure", "gradient", "advect_scalars")])
print(json.dumps({"config": cfg, "fuse": H.get_option("fuse"), "us per substep [advect_vector, div+comb+buoy, pressure, gradient, advect_scalars]": rows}))
