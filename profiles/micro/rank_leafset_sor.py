#!/usr/bin/env python3
"""The SOR sweep on the leaf set ONE rank of a decomposition owns, as a standalone single-GPU grid (no ghosts, no exchange):
what the kernel forms reach on a rank-sized ragged grid, beside the chained multi-GPU sweep of the same rank
(micro/dist_lone.py). argv: config world [rank ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, dist as HD, fields

config, world = sys.argv[1], int(sys.argv[2])
ranks = [int(a) for a in sys.argv[3:]] or list(range(world))
origins, R = fields.config_leaves(config)
for rank in ranks:
    d = HD.DistRank(origins, world, rank, 1.0 / R, n_scalars=1, sweeps_per_exchange=1)
    own = origins[d.owned_ids].copy()
    del d
    own = np.ascontiguousarray(own[fields.nanovdb_order(own)])
    grid = api.create_grid_from_leaves(own, 1.0 / R)
    N = len(own) * 512
    div = torch.randn(N, device="cuda"); p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
    line = f"{config} rank {rank} of {world}: {len(own)} owned leaves;"
    for form in ("wave", "pair", "auto"):
        H.set_option("rbgs", form)
        ms = min(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 48, 3) for _ in range(3))
        H.set_option("rbgs", None)
        line += f"  {form} {1e3 * ms:6.2f} us/iteration"
    print(line, "  (auto =", D.rbgs_plan(grid, 48)[0], ")", flush=True)
