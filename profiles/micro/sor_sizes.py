#!/usr/bin/env python3
"""SOR sweep time and algorithmic bandwidth against grid size: where the Infinity Cache regime (three sweep arrays within
256 MB) ends. Dense R^3 grids, the library's own choice of kernel form, 50 sweeps timed with hipEvents."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, fields

for R in (64, 128, 192, 256, 288, 320, 384, 512):
    origins = fields.dense_leaves(R)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    N = len(origins) * 512
    div = torch.randn(N, device="cuda")
    p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
    ms = min(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 50, 3) for _ in range(2))
    print(f"R={R:4d}  voxels={N:10d}  arrays={12 * N / 1e6:8.1f} MB  sweep={1e3 * ms:8.2f} us  algorithmic={12 * N / (ms * 1e-3) / 1e12:5.2f} TB/s ({100 * 12 * N / (ms * 1e-3) / 8e12:4.1f} % of 8 TB/s)", flush=True)
    del grid, div, p_a, p_b
    torch.cuda.empty_cache()
