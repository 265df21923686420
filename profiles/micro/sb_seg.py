#!/usr/bin/env python3
"""Blocked SOR kernel against the launch order of its blocks (option sor_block_seg: blocks per XCD segment; 0 = one chunk per XCD). argv: sizes"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
for c in sys.argv[1:] or ["256"]:
    origins, R = (fields.dense_leaves(int(c[1:])), int(c[1:])) if c.startswith("d") else fields.config_leaves(c)
    N = len(origins) * 512
    div = torch.randn(N, device="cuda"); p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
    for seg in (0, 8, 16, 32, 64, 128, 256, 1):
        H.set_option("sor_block_seg", seg)
        grid = api.create_grid_from_leaves(origins, 1.0 / R)
        ms = min(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 48, 3) for _ in range(3))
        print(c, "sor_block_seg", seg, f"{1e3 * ms:.2f} us / iteration", flush=True)
        del grid
