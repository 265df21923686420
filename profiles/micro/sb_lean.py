#!/usr/bin/env python3
"""The blocked SOR kernel's two forms of 16^3 blocks -- rows in registers, two workgroups per CU (sor_block_lean=0) against rows in LDS,
three per CU (=1) -- and the lean form's launch-start stagger, per grid size. argv: config names (dN = dense N^3)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields

for c in sys.argv[1:]:
    origins, R = (fields.dense_leaves(int(c[1:])), int(c[1:])) if c.startswith("d") else fields.config_leaves(c)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    N = len(origins) * 512
    div = torch.randn(N, device="cuda"); p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
    H.set_option("rbgs", "block"); H.set_option("sor_block_lb", "2"); H.set_option("sor_block_k", "2")
    line = f"{c:10s} leaves={len(origins):6d}"
    for lean, stag in (("0", None), ("1", 0), ("1", 2), ("1", 3), ("1", 4), ("1", 6)):
        H.set_option("sor_block_lean", lean)
        if stag is not None:
            pass  # (option sor_block_lean_stagger was removed in round 4: it bought nothing)
        ms = min(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 48, 3) for _ in range(3))
        line += f"  {'regs' if lean == '0' else 'lean/' + str(stag)} {1e3 * ms:6.2f}"
    for k in ("rbgs", "sor_block_lb", "sor_block_k", "sor_block_lean"):
        H.set_option(k, None)
    print(line, " us / iteration", flush=True)
    del grid, div, p_a, p_b
    torch.cuda.empty_cache()
