#!/usr/bin/env python3
"""Does the order of the leaves (= memory layout + launch order) change the SOR sweep? NanoVDB order (x, y, z-fastest
inside each 128^3 lower node) against (x,y)-tiled and Morton column orders inside the same lower nodes."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, fields

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = fields.dense_leaves(R)
vs = 1.0 / R

def morton2(a, b):
    r = np.zeros_like(a)
    for i in range(5):
        r |= ((a >> i) & 1) << (2 * i + 1) | ((b >> i) & 1) << (2 * i)
    return r

def order(kind):
    o = base.astype(np.int64)
    node = ((o[:, 0] >> 7) << 8) | ((o[:, 1] >> 7) << 4) | (o[:, 2] >> 7)
    lx, ly, lz = (o[:, 0] & 127) >> 3, (o[:, 1] & 127) >> 3, (o[:, 2] & 127) >> 3
    if kind == "nanovdb":
        key = (lx << 8) | (ly << 4) | lz
    elif kind == "tile4":
        key = (((ly >> 2) << 10) | (lx << 6) | ((ly & 3) << 4) | lz)
    elif kind == "tile4x4":
        key = ((((lx >> 2) << 2) | (ly >> 2)) << 8) | ((lx & 3) << 6) | ((ly & 3) << 4) | lz
    elif kind == "morton":
        key = (morton2(lx, ly) << 4) | lz
    return np.lexsort((key, node))

for kind in ("nanovdb", "tile4", "tile4x4", "morton"):
    origins = np.ascontiguousarray(base[order(kind)])
    grid = api.create_grid_from_leaves(origins, vs)
    N = len(origins) * 512
    f = fields.synthetic_fields(origins, R)
    u = torch.from_numpy(f["vel"]).cuda()
    adv = torch.empty_like(u)
    D.advect_vector(grid, u, adv, 1.0 / 24.0, float(R))
    div = torch.empty(N, dtype=torch.float32, device="cuda")
    D.divergence(grid, adv, div, float(R))
    p_a = torch.zeros(N, dtype=torch.float32, device="cuda"); p_b = torch.zeros_like(p_a)
    ms = min(D.time_rbgs(grid, div, p_a, p_b, vs, 1.9758, 20, 3) for _ in range(2))
    print(f"{kind:8s}: {1e3 * ms:.2f} us per sweep")
    del grid
