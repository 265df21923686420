#!/usr/bin/env python3
"""Round 6 A/B: the full Compute_Sim substep (S = 5, combustion, buoyancy) with option "fuse" = 1 (divergence + combustion + buoyancy as one
launch, {fuel, waste, temperature, flame} advected out of one 16-byte-per-voxel array) against "fuse" = 0 (the reference's decomposition: three
launches, five float arrays). Alternating, hipEvents around the five stages of 10 substeps each (hns_sim_stage_timing), and the two paths'
results after 3 substeps compared bit for bit first.

    python profiles/micro/full_fuse_ab.py [config ...]   (default 256)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields

NAMES = ["density", "temperature", "fuel", "waste", "flame"]


def run(cfg, fuse, f, grid, vs, n_timed=10, rounds=3):
    H.set_option("fuse", fuse)
    sim = D.Sim(grid, NAMES)
    sim.upload({"vel": f["vel"], **{n: f[n] for n in NAMES}})
    prm = api.CombustionParams(vorticityScale=0.0)
    st = D.current_stream()
    for _ in range(3):
        sim.substep(50, 1.0 / 24.0, vs, prm, False, st)
    torch.cuda.synchronize()
    out = {n: np.empty_like(f[n]) for n in NAMES}
    out["vel"] = np.empty_like(f["vel"])
    sim.download(out)
    best = None
    for _ in range(rounds):
        sim.stage_timing(n_timed)
        for _ in range(n_timed):
            sim.substep(50, 1.0 / 24.0, vs, prm, False, st)
        torch.cuda.synchronize()
        ms, n = sim.stage_times()
        per = {k: 1e3 * v / n for k, v in ms.items()}
        per["substep"] = sum(per.values())
        if best is None or per["substep"] < best["substep"]:
            best = per
    sim.close()
    H.set_option("fuse", None)
    return out, {k: round(v, 1) for k, v in best.items()}


for cfg in sys.argv[1:] or ["256"]:
    origins, R = fields.config_leaves(cfg)
    vs = 1.0 / R
    f = fields.synthetic_fields(origins, R)
    grid = api.create_grid_from_leaves(origins, vs)
    a, ta = run(cfg, "0", f, grid, vs)
    b, tb = run(cfg, "1", f, grid, vs)
    same = {n: bool(np.array_equal(a[n], b[n])) for n in a}
    a2, ta2 = run(cfg, "0", f, grid, vs)
    b2, tb2 = run(cfg, "1", f, grid, vs)
    print(json.dumps({"config": cfg, "leaves": len(origins), "bit_identical_after_3_substeps": same, "us_per_stage_unfused": [ta, ta2], "us_per_stage_fused": [tb, tb2],
                      "stages": "advect_vector | divergence (+ combustion + buoyancy) | pressure | gradient | advect_scalars S=5"}))
