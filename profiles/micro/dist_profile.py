#!/usr/bin/env python3
"""One rank of a 2-slab 256^3-per-rank run alone on the device (loopback transport), N substeps: run under
rocprofv3 --kernel-trace --stats to see where a rank's time goes next to the plain single-GPU substep (argv[1] = single)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, dist as HD, fields  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "rank"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
import hnanosolver_amd as H
if len(sys.argv) > 3: H.set_option("dist_wire_us", sys.argv[3])
origins, R = fields.config_leaves("256")
vs, iters, dt, n = 1.0 / R, 50, 1.0 / 24.0, 10
st = D.current_stream()
f = fields.synthetic_fields(origins, R)
if mode == "single":
    sim = D.Sim(api.create_grid_from_leaves(origins, vs), ["density"])
    sim.upload({"vel": f["vel"], "density": f["density"]})
    step = lambda: sim.core_substep(iters, dt, vs, st)
else:
    d = HD.DistRank(HD.slab_domain(origins, R, 2), 2, 0, vs, n_scalars=1, sweeps_per_exchange=k)
    d.connect_loopback(rccl=len(sys.argv) > 4 and sys.argv[4] == "rccl")
    d.upload(f["vel"], [f["density"]])
    step = lambda: d.core_substep(iters, dt, st)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(mode, "rccl" if len(sys.argv) > 4 else "", "k", k, "wire_us", H.get_option("dist_wire_us"), "ms per substep", round(1e3 * (time.perf_counter() - t0) / n, 3))
