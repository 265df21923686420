#!/usr/bin/env python3
"""Cost of being one rank of a multi-GPU run before any wire time, measured on ONE GPU with the local transport of
hns_dist (every rank in this process, a message = a device copy out of the peer's send buffer; plan, launch ranges, pack /
unpack kernels, communication stream and events exactly as with RCCL).

  argv: config (256 | 128 | plume1024 ...)  world  [sweeps_per_exchange]   [--partition: split ONE config domain]
        [--leaf-order: contiguous ranges of the leaf list, the partition of rounds 1-4]  [--rank=N: the rank measured alone]
        [--lone-only: nothing but that rank's loopback substeps -- for rocprofv3 --kernel-trace --stats of one rank's kernels]

Weak scaling (default): `world` slabs of the config stacked along x. All ranks share the device, so the device time of a
lockstep substep is compared with world x the plain single-GPU substep of one slab: their ratio is the per-rank overhead
(ghost sweeps, split launches, pack/unpack, copies). Host enqueue time per rank per substep is measured separately."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hnanosolver_amd import api, device as D, dist as HD, fields  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
config = args[0] if args else "256"
world = int(args[1]) if len(args) > 1 else 2
k = int(args[2]) if len(args) > 2 else 0
partition = "--partition" in sys.argv
leaf_order = "--leaf-order" in sys.argv
pick = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--rank=")]
iters, dt = 50, 1.0 / 24.0
origins, R = fields.config_leaves(config)
vs = 1.0 / R


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


lone_only = "--lone-only" in sys.argv
if lone_only:
    st = D.current_stream()
    glob = origins if partition else HD.slab_domain(origins, R, world)
    lone_rank = pick[0] if pick else (0 if world < 3 else world // 2)
    lone = HD.DistRank(glob, world, lone_rank, vs, n_scalars=1, sweeps_per_exchange=k, leaf_order=leaf_order)
    lone.connect_loopback(rccl="--rccl" in sys.argv)  # (--rccl: every message through a one-rank RCCL communicator instead of a device copy)
    own = glob[lone.owned_ids].copy()
    if not partition:
        own[:, 0] %= R
    g = fields.synthetic_fields(own, R)
    lone.upload(g["vel"], [g["density"]])
    # (the loopback transports answer every message with the rank's own payload: the fields drift into nonsense over tens of substeps and kernel times with them -- the
    # advection's back-traces grow; profiles/r06_dist_exchanged_notes.txt. Every repetition therefore starts from freshly uploaded fields: 2 warm-up + 8 timed substeps)
    def rep():
        lone.upload(g["vel"], [g["density"]])
        return round(timed(lambda: lone.core_substep(iters, dt, st), n=8, warm=2), 3)  # (substeps 3 .. 10 behind the upload: the drift starts around the tenth)

    reps = [rep() for _ in range(3 if "--three" in sys.argv else 1)]
    print(json.dumps({"config": config, "world": world, "rank": lone_rank, "one_rank_loopback_substep_ms": reps[0] if len(reps) == 1 else reps, "rccl": "--rccl" in sys.argv,
                      "info": {x: lone.info()[x] for x in ("boundary_leaves", "interior_leaves", "ghost_leaves", "peers", "halo_peers", "sweeps_per_exchange", "exchanges", "packed_exchanges", "messages_sent")}}))
    if "--no-plain" in sys.argv:
        sys.exit(0)
    # the same leaves (owned only) as a plain single-GPU grid: what the rank's work costs without being a rank
    f = fields.synthetic_fields(own, R)
    grid = api.create_grid_from_leaves(own, vs)
    sim = D.Sim(grid, ["density"])
    sim.upload({"vel": f["vel"], "density": f["density"]})
    print(json.dumps({"owned_leaves_as_a_plain_grid_substep_ms": round(timed(lambda: sim.core_substep(iters, dt, vs, st), n=20), 3), "leaves": len(own)}))
    sys.exit(0)
f = fields.synthetic_fields(origins, R)
grid = api.create_grid_from_leaves(origins, vs)
sim = D.Sim(grid, ["density"])
sim.upload({"vel": f["vel"], "density": f["density"]})
st = D.current_stream()
single = timed(lambda: sim.core_substep(iters, dt, vs, st))
sim.close()

glob = origins if partition else HD.slab_domain(origins, R, world)
ranks = [HD.DistRank(glob, world, r, vs, n_scalars=1, sweeps_per_exchange=k, leaf_order=leaf_order) for r in range(world)]
HD.DistRank.connect_local(ranks)
for d in ranks:
    own = glob[d.owned_ids].copy()
    if not partition:
        own[:, 0] %= R
    g = fields.synthetic_fields(own, R)
    d.upload(g["vel"], [g["density"]])
# one rank alone, its messages looped back to itself (wrong data, right sizes): the production structure of one GPU's work
lone_rank = pick[0] if pick else (0 if world < 3 else world // 2)  # a rank with neighbours on both sides where there is one
lone = HD.DistRank(glob, world, lone_rank, vs, n_scalars=1, sweeps_per_exchange=k, leaf_order=leaf_order)
lone.connect_loopback()
own = glob[lone.owned_ids].copy()
if not partition:
    own[:, 0] %= R
g = fields.synthetic_fields(own, R)
lone.upload(g["vel"], [g["density"]])
alone = timed(lambda: lone.core_substep(iters, dt, st))
torch.cuda.synchronize()
lone.timing(8)  # hipEvents around the pressure loop of the next substeps: us per red+black iteration of this rank
for _ in range(8):
    lone.core_substep(iters, dt, st)
lone.synchronize(st)
p_ms, p_sweeps = lone.pressure_time()
alone_us_per_iteration = 1e3 * p_ms / max(1, p_sweeps)
t0 = time.perf_counter()
for _ in range(5):
    lone.core_substep(iters, dt, st)
alone_enqueue = 1e3 * (time.perf_counter() - t0) / 5
lone.synchronize(st)
alone_info = lone.info()
lone.close()
lock = timed(lambda: HD.DistRank.local_core_substep(ranks, iters, dt, st))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    HD.DistRank.local_core_substep(ranks, iters, dt, st)
enqueue = 1e3 * (time.perf_counter() - t0) / 5 / world
torch.cuda.synchronize()
info = [d.info() for d in ranks]
work = single if partition else world * single
print(json.dumps({
    "config": config, "world": world, "partition": partition, "partition_axis": ranks[0].partition_axis, "sweeps_per_exchange": info[0]["sweeps_per_exchange"],
    "single_gpu_substep_ms": round(single, 3), "all_ranks_lockstep_ms": round(lock, 3), "same_work_on_one_grid_ms": round(work, 3),
    "lockstep_overhead": round(lock / work - 1.0, 4), "lockstep_host_enqueue_ms_per_rank_per_substep": round(enqueue, 3),
    "one_rank_loopback": {"rank": lone_rank, "peers": alone_info["peers"], "halo_peers": alone_info["halo_peers"], "boundary_leaves": alone_info["boundary_leaves"], "ghost_leaves": alone_info["ghost_leaves"], "owned_leaves": alone_info["boundary_leaves"] + alone_info["interior_leaves"], "substep_ms": round(alone, 3), "pressure_us_per_iteration": round(alone_us_per_iteration, 2),
                          "host_enqueue_ms": round(alone_enqueue, 3),
                          "overhead_vs_single_gpu_per_leaf": round((alone / (alone_info["boundary_leaves"] + alone_info["interior_leaves"])) / (single / len(origins) * (world if partition else 1)) - 1.0, 4)},
    "rank0": {x: info[0][x] for x in ("boundary_leaves", "interior_leaves", "ghost_leaves", "peers", "exchanges", "messages_sent", "bytes_sent")},
    "max_bytes_sent_per_rank_per_substep": max(sum(i["bytes_sent"].values()) for i in info),
}))
