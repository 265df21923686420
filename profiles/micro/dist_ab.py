#!/usr/bin/env python3
"""A/B of hns_dist create-time options in ONE process: the lone loopback rank of two 256^3 slabs, alternating settings.
argv: option value_a value_b [sweeps_per_exchange=1] [config=256]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H  # noqa: E402
from hnanosolver_amd import device as D, dist as HD, fields  # noqa: E402

opt, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 1
config = sys.argv[5] if len(sys.argv) > 5 else "256"
origins, R = fields.config_leaves(config)
vs, iters, dt, n = 1.0 / R, 50, 1.0 / 24.0, 20
st = D.current_stream()
f = fields.synthetic_fields(origins, R)
glob = HD.slab_domain(origins, R, 2)
res = {va: [], vb: []}
for rep in range(3):
    for v in (va, vb):
        H.set_option(opt, v)
        d = HD.DistRank(glob, 2, 0, vs, n_scalars=1, sweeps_per_exchange=k)
        d.connect_loopback()
        d.upload(f["vel"], [f["density"]])
        for _ in range(3):
            d.core_substep(iters, dt, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            d.core_substep(iters, dt, st)
        torch.cuda.synchronize()
        res[v].append(round(1e3 * (time.perf_counter() - t0) / n, 3))
        d.close()
print(opt, {v: r for v, r in res.items()})
