#!/usr/bin/env python3
"""One rank of a decomposition alone on the device (loopback transport), N substeps: for rocprofv3 --kernel-trace.
argv: config world rank sweeps_per_exchange [--partition] [option=value ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import device as D, dist as HD, fields  # noqa: E402

import hnanosolver_amd as H  # noqa: E402

for a in sys.argv[1:]:  # name=value: library options (hns_set_option) for this run
    if "=" in a and not a.startswith("--"):
        H.set_option(*a.split("=", 1))
args = [a for a in sys.argv[1:] if not a.startswith("--") and "=" not in a]
config, world, rank, k = args[0], int(args[1]), int(args[2]), int(args[3])
partition = "--partition" in sys.argv
origins, R = fields.config_leaves(config)
vs, iters, dt, n = 1.0 / R, 50, 1.0 / 24.0, 10
glob = origins if partition else HD.slab_domain(origins, R, world)
d = HD.DistRank(glob, world, rank, vs, n_scalars=1, sweeps_per_exchange=k)
d.connect_loopback()
own = glob[d.owned_ids].copy()
if not partition:
    own[:, 0] %= R
g = fields.synthetic_fields(own, R)
d.upload(g["vel"], [g["density"]])
st = D.current_stream()
for _ in range(3):
    d.core_substep(iters, dt, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    d.core_substep(iters, dt, st)
torch.cuda.synchronize()
print(config, world, rank, "k", k, "ms per substep", round(1e3 * (time.perf_counter() - t0) / n, 3), d.info())
