#!/usr/bin/env python3
"""All ranks of a decomposition in this process on one stream (local transport), N substeps: for rocprofv3 --kernel-trace.
argv: config world sweeps_per_exchange [--partition]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import device as D, dist as HD, fields  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
config, world, k = args[0], int(args[1]), int(args[2])
partition = "--partition" in sys.argv
origins, R = fields.config_leaves(config)
vs, iters, dt, n = 1.0 / R, 50, 1.0 / 24.0, 5
glob = origins if partition else HD.slab_domain(origins, R, world)
ranks = [HD.DistRank(glob, world, r, vs, n_scalars=1, sweeps_per_exchange=k) for r in range(world)]
HD.DistRank.connect_local(ranks)
for d in ranks:
    own = glob[d.owned_ids].copy()
    if not partition:
        own[:, 0] %= R
    g = fields.synthetic_fields(own, R)
    d.upload(g["vel"], [g["density"]])
st = D.current_stream()
for _ in range(2):
    HD.DistRank.local_core_substep(ranks, iters, dt, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    HD.DistRank.local_core_substep(ranks, iters, dt, st)
torch.cuda.synchronize()
print(config, world, "k", k, "ms per substep, all ranks", round(1e3 * (time.perf_counter() - t0) / n, 3))
