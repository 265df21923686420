#!/usr/bin/env python3
"""Round 6 A/B (needs profiles/micro/exp/advect_vector_tile.patch applied: the product has no such kernel and no option word `gather`; result: profiles/r06_lds_tile_gather.txt): velocity (and scalar) advection with the taps out of an LDS tile of the 16^3 voxels around the leaf (option advect = auto)
against the kernels that gather every tap through the L1 (advect = gather). Alternating, hipEvents around `reps` launches, outputs compared
bit for bit; amplitude = peak back-trace in voxels (96/24 = 4: the bench's field; larger: more lanes leave the tile).

    python profiles/micro/advect_tile_ab.py [config ...] [--amp=96,160,400]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields

amps = [96.0]
cfgs = []
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
    return 1e3 * best


for cfg in cfgs or ["256"]:
    origins, R = fields.config_leaves(cfg)
    vs = 1.0 / R
    grid = api.create_grid_from_leaves(origins, vs)
    for amp in amps:
        f = fields.synthetic_fields(origins, R, amplitude_voxels=amp)
        u = torch.from_numpy(f["vel"]).cuda()
        phi = torch.from_numpy(f["density"]).cuda()
        res = {}
        for rep in range(2):
            for form in ("gather", "auto"):
                H.set_option("advect", form)
                out = torch.empty_like(u)
                tv = timed(lambda: D.advect_vector(grid, u, out, 1.0 / 24.0, 1.0 / vs))
                po = torch.empty_like(phi)
                ts = timed(lambda: D.advect_scalars(grid, u, [phi], [po], 1.0 / 24.0, 1.0 / vs))
                res.setdefault(form, []).append((round(tv, 1), round(ts, 1), out, po))
        H.set_option("advect", None)
        same_v = torch.equal(res["gather"][0][2], res["auto"][0][2])
        same_s = torch.equal(res["gather"][0][3], res["auto"][0][3])
        print(f"{cfg} amplitude {amp:g}: advect_vector gather {[r[0] for r in res['gather']]} us, tile {[r[0] for r in res['auto']]} us, bit-identical {same_v};  "
              f"advect_scalars S=1 gather {[r[1] for r in res['gather']]} us, tile {[r[1] for r in res['auto']]} us, bit-identical {same_s}", flush=True)
