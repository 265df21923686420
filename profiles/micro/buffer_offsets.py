#!/usr/bin/env python3
"""Does the relative placement of p_a / p_b / div matter? (Each is exactly 64 MiB at 256^3, so back-to-back allocations put
the three streams a wave touches at the same offset modulo every power of two: same L2 / MALL set, same HBM channel.)
Carves the three fields out of one allocation at chosen byte skews and times the SOR loop."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, fields

origins, R = fields.config_leaves("256")
vs = 1.0 / R
grid = api.create_grid_from_leaves(origins, vs)
N = len(origins) * 512
f = fields.synthetic_fields(origins, R)
u = torch.from_numpy(f["vel"]).cuda()
adv = torch.empty_like(u)
D.advect_vector(grid, u, adv, 1.0 / 24.0, float(R))
div0 = torch.empty(N, dtype=torch.float32, device="cuda")
D.divergence(grid, adv, div0, float(R))
omega = 1.9758
for skew in (0, 256, 4096, 4096 + 256, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256, 3 << 19):
    pool = torch.zeros(3 * N + 3 * (skew // 4) + 1024, dtype=torch.float32, device="cuda")
    k = skew // 4
    p_a = pool[0:N]
    p_b = pool[N + k:2 * N + k]
    div = pool[2 * N + 2 * k:3 * N + 2 * k]
    div.copy_(div0)
    ms = D.time_rbgs(grid, div, p_a, p_b, vs, omega, 50, 5)
    print(f"skew {skew:8d} B: {1e3 * ms:.2f} us per sweep")
