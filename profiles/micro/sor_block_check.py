#!/usr/bin/env python3
"""Temporally blocked SOR (hns_sorblock.hip) against the reference's two-launch form (rbgs=color): bit identity on dense,
ragged and scattered leaf sets for several iteration counts, then sweep times per configuration and block shape.
argv: `check` and/or config names (64 128 256 plume ...) and `name=value` options applied to the timed runs."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields


def solve(grid, div, n, iters, **opts):
    for k, v in opts.items():
        H.set_option(k, str(v))
    p_a = torch.zeros(n, device="cuda"); p_b = torch.zeros(n, device="cuda")
    p_a.uniform_(-1, 1)  # a warm start: every form must read its input
    p0 = p_a.clone()
    out = D.rbgs_iterate(grid, div, p_a, p_b, 0.013, 1.93, iters).clone()
    for k in opts:
        H.set_option(k, None)
    return out, p0


def check():
    rng = np.random.default_rng(5)
    sets = {
        "dense32": fields.dense_leaves(32),
        "dense48": fields.dense_leaves(48),
        "plume": fields.plume_leaves(16, 1.5, 0.3),
        "scatter": None,
    }
    o = rng.integers(-6, 6, size=(400, 3)).astype(np.int32) * 8
    o = np.unique(o, axis=0)
    sets["scatter"] = np.ascontiguousarray(o[fields.nanovdb_order(o)])
    bad = 0
    for name, origins in sets.items():
        grid = api.create_grid_from_leaves(origins, 0.013)
        n = len(origins) * 512
        torch.manual_seed(3)
        div = torch.randn(n, device="cuda")
        for iters in (2, 3, 4, 7, 10):
            torch.manual_seed(11)
            want, _ = solve(grid, div, n, iters, rbgs="color")
            for lb, k in ((1, 2), (1, 4), (2, 2)):
                torch.manual_seed(11)
                got, _ = solve(grid, div, n, iters, rbgs="block", sor_block_lb=lb, sor_block_k=k)
                same = torch.equal(want, got)
                bad += not same
                print(f"{name:8s} leaves={len(origins):5d} iters={iters:2d} lb={lb} k={k}: {'bit-identical' if same else 'DIFFERENT max|d|=%g' % (want - got).abs().max().item()}", flush=True)
    print("CHECK", "FAILED" if bad else "OK", flush=True)
    return bad


def timing(cfgs, opts):
    for c in cfgs:
        if c.startswith("d"):  # dN: dense N^3
            R = int(c[1:])
            origins = fields.dense_leaves(R)
        else:
            origins, R = fields.config_leaves(c)
        grid = api.create_grid_from_leaves(origins, 1.0 / R)
        N = len(origins) * 512
        div = torch.randn(N, device="cuda")
        p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
        for form in ({"rbgs": "pair"}, {"rbgs": "wave"}, {"rbgs": "block", "sor_block_lb": 1, "sor_block_k": 2}, {"rbgs": "block", "sor_block_lb": 1, "sor_block_k": 4},
                     {"rbgs": "block", "sor_block_lb": 2, "sor_block_k": 2, "sor_block_stagger": 0}, {"rbgs": "block", "sor_block_lb": 2, "sor_block_k": 2}):
            if form.get("sor_block_lb") == 1 and len(origins) > 5000:
                continue
            if form["rbgs"] == "wave" and len(origins) > 20000:
                continue
            for k, v in {**form, **opts}.items():
                H.set_option(k, str(v))
            ms = min(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 48, 3) for _ in range(3))
            for k in {**form, **opts}:
                H.set_option(k, None)
            print(f"{c:10s} leaves={len(origins):6d} {str(form):60s} {1e3 * ms:8.2f} us / iteration  {12 * N / (ms * 1e-3) / 8e12:5.3f} of 8 TB/s at 12 B/voxel", flush=True)
        del grid, div, p_a, p_b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    args = sys.argv[1:]
    opts = dict(a.split("=", 1) for a in args if "=" in a)
    cfgs = [a for a in args if "=" not in a and a != "check"]
    rc = check() if "check" in args else 0
    if cfgs:
        timing(cfgs, opts)
    sys.exit(1 if rc else 0)
