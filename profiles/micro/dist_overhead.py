#!/usr/bin/env python3
"""Cost of being one rank of a multi-GPU slab run, measured on ONE GPU: rank 0 of a 2-slab 256^3-per-rank domain with its
halo exchange looped back through device copies (no RCCL, everything else -- ghost leaves, ghost sweeps, pack kernels,
Python driver -- as in bench.py --gpus 2). Prints ms per substep next to the plain single-GPU substep, i.e. an upper
bound on the weak-scaling efficiency before any wire time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, dist as HD, fields  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "256"
iters, dt = 50, 1.0 / 24.0
origins, R = fields.config_leaves(config)
vs = 1.0 / R

f = fields.synthetic_fields(origins, R)
grid = api.create_grid_from_leaves(origins, vs)
sim = D.Sim(grid, ["density"])
sim.upload({"vel": f["vel"], "density": f["density"]})
st = D.current_stream()


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


single = timed(lambda: sim.core_substep(iters, dt, vs, st))

runner = HD.SlabBench(origins, R, 0, 2, iters, dt)
halo = runner.solver.halo
counts = {"exchanges": 0}


def loopback(fields_, mirror=True):
    sends, recvs = halo.pack_sends(fields_, mirror), halo.recv_targets(fields_, mirror)
    for q, dst in recvs.items():
        src = sends[q]
        n = min(src.numel(), dst.numel())
        dst.view(-1)[:n].copy_(src.view(-1)[:n])
    halo.finish(fields_, mirror)
    counts["exchanges"] += 1


halo.exchange = loopback
rank = timed(runner.step)
per_step = counts["exchanges"] / 13
# host-only cost of the driver: same calls, GPU work excluded by timing the enqueue phase only
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    runner.step()
enqueue = 1e3 * (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
print({"config": config, "single_gpu_ms": round(single, 3), "rank_of_2_loopback_ms": round(rank, 3), "host_enqueue_ms": round(enqueue, 3),
       "exchanges_per_substep": round(per_step, 1), "ghost_leaves": int(runner.plan.n_local - runner.plan.n_owned), "owned_leaves": int(runner.plan.n_owned),
       "efficiency_upper_bound": round(single / rank, 3)})
