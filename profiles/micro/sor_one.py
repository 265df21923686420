#!/usr/bin/env python3
"""SOR sweep time at one configuration (argv: config names...), hipEvents around 50 sweeps; honours HNS_LIBRARY and
`name=value` option arguments (hns_set_option)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields

cfgs = [a for a in sys.argv[1:] if "=" not in a] or ["256"]
for a in sys.argv[1:]:
    if "=" in a:
        H.set_option(*a.split("=", 1))
for c in cfgs:
    origins, R = fields.config_leaves(c)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    N = len(origins) * 512
    div = torch.randn(N, device="cuda")
    p_a = torch.zeros(N, device="cuda"); p_b = torch.zeros(N, device="cuda")
    ms = min(D.time_rbgs(grid, div, p_a, p_b, 1.0 / R, 1.97, 50, 3) for _ in range(3))
    print(f"{os.path.basename(os.environ.get('HNS_LIBRARY', 'libhns.so')):18s} {' '.join(a for a in sys.argv[1:] if '=' in a):20s} {c:10s} sweep={1e3 * ms:8.2f} us  {12 * N / (ms * 1e-3) / 8e12:5.3f} of 8 TB/s", flush=True)
    del grid, div, p_a, p_b
    torch.cuda.empty_cache()
