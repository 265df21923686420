#!/usr/bin/env python3
"""Randomised soak of every SOR form against the two-launch form (rbgs=color): 40 random leaf sets (extent, fill, offset, voxel size), random omega,
iteration count 2..11, warm and zero start; bit for bit. Prints SOAK OK / FAILED."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
rng = np.random.default_rng(2026)
bad = 0
for trial in range(40):
    ext = int(rng.integers(3, 14))
    fill = float(rng.uniform(0.15, 1.0))
    lat = np.stack(np.meshgrid(*[np.arange(-ext // 2, ext - ext // 2)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = rng.random(len(lat)) < fill
    if keep.sum() == 0: continue
    shift = rng.integers(-3000, 3000, 3) * 8
    o = (lat[keep] * 8 + shift).astype(np.int32)
    o = np.ascontiguousarray(o[fields.nanovdb_order(o)])
    vs = float(rng.uniform(0.005, 0.05))
    grid = api.create_grid_from_leaves(o, vs)
    n = len(o) * 512
    g = torch.Generator(device="cpu").manual_seed(trial)
    div = (torch.randn(n, generator=g) * 10).cuda()
    p0 = (torch.rand(n, generator=g) * 2 - 1).cuda() if trial % 3 else torch.zeros(n, device="cuda")
    omega = float(rng.uniform(1.0, 1.98)); iters = int(rng.integers(2, 12))
    def solve(**opts):
        for k, v in opts.items(): H.set_option(k, str(v))
        a = p0.clone(); b = torch.full_like(p0, 3.0)
        out = D.rbgs_iterate(grid, div, a, b, vs, omega, iters).clone()
        for k in opts: H.set_option(k, None)
        return out
    want = solve(rbgs="color")
    for name, opts in (("auto", {}), ("lean", dict(rbgs="block", sor_block_lb=2, sor_block_k=2, sor_block_lean=1)), ("regs", dict(rbgs="block", sor_block_lb=2, sor_block_k=2, sor_block_lean=0)), ("lb1", dict(rbgs="block", sor_block_lb=1, sor_block_k=2))):
        got = solve(**opts)
        if not torch.equal(want, got):
            bad += 1; print("MISMATCH", trial, name, len(o), iters, float((want - got).abs().max()), flush=True)
    print("trial", trial, "leaves", len(o), "fill %.2f" % fill, "iters", iters, "ok", flush=True)
print("SOAK", "FAILED" if bad else "OK")
