import os, sys, torch
sys.path.insert(0, "/root/repo")
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
for R in (192, 256):
    origins = fields.dense_leaves(R)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    N = len(origins) * 512
    div = torch.randn(N, device="cuda"); p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
    for st in (0, 5, 6, 7, 8, 9, 10, 8, 0):
        H.set_option("sor_block_stagger", st)
        ms = min(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 48, 3) for _ in range(4))
        print(R, "stagger", st, f"{1e3*ms:.2f} us/iter", D.rbgs_plan(grid, 50)[0][:30], flush=True)
