"""Plan of a decomposition without a GPU (HNS_DIST_PLAN_ONLY): per rank, halo peers, boundary / interior / ghost leaves and voxels sent per exchange type,
for the slab partition (round 5 default) and for contiguous ranges of the caller's leaf order (rounds 1-4: HNS_DIST_LEAF_ORDER).
  python profiles/micro/dist_plan_stats.py [config=plume1024] [world=8] [k=2]"""
import json, sys
sys.path.insert(0, ".")
import numpy as np
from hnanosolver_amd import fields
from hnanosolver_amd.dist import DistRank
cfg = sys.argv[1] if len(sys.argv) > 1 else "plume1024"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
origins, R = fields.config_leaves(cfg)
for leaf_order in (True, False):
    rows = []
    for r in range(world):
        d = DistRank(origins, world, r, 1.0 / R, 1, k, plan_only=True, leaf_order=leaf_order)
        i = d.info()
        peers = d.peers()
        halo = sum(1 for p in peers if any(reg.voxels for t in (1, 2, 3) for reg in (p.send[t], p.recv[t])))
        rows.append({"rank": r, "axis": d.partition_axis, "peers": i["peers"], "halo_peers": halo, "boundary": i["boundary_leaves"], "interior": i["interior_leaves"],
                     "ghosts": i["ghost_leaves"], "p_voxels_sent": i["region_voxels_sent"]["p"], "adv_voxels_sent": i["region_voxels_sent"]["advection inputs"]})
        d.close()
    print(json.dumps({"config": cfg, "world": world, "k": k, "partition": "leaf order (rounds 1-4)" if leaf_order else "slabs (round 5)", "ranks": rows}))
