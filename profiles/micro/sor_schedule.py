#!/usr/bin/env python3
"""XCD-chunked vs linear launch order of the SOR sweep against grid size (argv[1] = auto | linear | chunk)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
H.set_option("schedule", sys.argv[1] if len(sys.argv) > 1 else "auto")
for R in (192, 224, 256, 288, 320, 384):
    origins = fields.dense_leaves(R)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    N = len(origins) * 512
    div = torch.randn(N, device="cuda"); p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
    ms = min(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 50, 3) for _ in range(2))
    print(H.get_option("schedule"), R, f"{1e3 * ms:.2f} us", flush=True)
    del grid, div, p_a, p_b
    torch.cuda.empty_cache()
