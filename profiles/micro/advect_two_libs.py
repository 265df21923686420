#!/usr/bin/env python3
"""advect_vector and advect_scalars (S = 1) launch times and results of ONE build of the library (HNS_LIBRARY), for alternation by the caller: hipEvents around 20 launches, best of 3,
and a digest of the outputs so that two builds can be compared bit for bit. argv: [config ...] [--amp=96,160]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, fields

amps, cfgs = [96.0], []
for a in sys.argv[1:]:
    if a.startswith("--amp="):
        amps = [float(x) for x in a[6:].split(",")]
    else:
        cfgs.append(a)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return round(1e3 * best, 1)


lib = os.path.basename(os.environ.get("HNS_LIBRARY", "libhns.so"))
for cfg in cfgs or ["256"]:
    origins, R = fields.config_leaves(cfg)
    vs = 1.0 / R
    grid = api.create_grid_from_leaves(origins, vs)
    for amp in amps:
        f = fields.synthetic_fields(origins, R, amplitude_voxels=amp)
        u = torch.from_numpy(f["vel"]).cuda()
        phi = torch.from_numpy(f["density"]).cuda()
        out, po = torch.empty_like(u), torch.empty_like(phi)
        tv = timed(lambda: D.advect_vector(grid, u, out, 1.0 / 24.0, 1.0 / vs))
        ts = timed(lambda: D.advect_scalars(grid, u, [phi], [po], 1.0 / 24.0, 1.0 / vs))
        dig = hashlib.sha1(out.cpu().numpy().tobytes() + po.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"{lib:18s} {cfg:9s} amplitude {amp:5g}: advect_vector {tv:7.1f} us  advect_scalars S=1 {ts:7.1f} us  outputs {dig}", flush=True)
