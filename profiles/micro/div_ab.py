#!/usr/bin/env python3
"""divergence / gradient kernel time with an option at two values, alternating in one process (stage hipEvents, 10 substeps per sample).
argv: option value_a value_b [config=256]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H  # noqa: E402
from hnanosolver_amd import api, device as D, fields  # noqa: E402

opt, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
config = sys.argv[4] if len(sys.argv) > 4 else "256"
origins, R = fields.config_leaves(config)
vs = 1.0 / R
f = fields.synthetic_fields(origins, R)
sim = D.Sim(api.create_grid_from_leaves(origins, vs), ["density"])
sim.upload({"vel": f["vel"], "density": f["density"]})
st = D.current_stream()
res = {va: [], vb: []}
for rep in range(3):
    for v in (va, vb):
        H.set_option(opt, v)
        for _ in range(2):
            sim.core_substep(2, 1.0 / 24.0, vs, st)
        n = 10
        sim.stage_timing(n)
        for _ in range(n):
            sim.core_substep(2, 1.0 / 24.0, vs, st)
        torch.cuda.synchronize()
        t, k = sim.stage_times()
        res[v].append({s: round(1e3 * ms / k, 1) for s, ms in t.items() if s in ("divergence", "gradient")})
print(config, opt, res)
