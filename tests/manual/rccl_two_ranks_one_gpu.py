#!/usr/bin/env python3
"""Manual probe (not part of the suites): the RCCL transport of hns_dist with two processes that share ONE device.
RCCL normally refuses two ranks on one GPU; if this build allows it, the real ncclSend/ncclRecv path can be checked
against the single-grid answer on a 1-GPU box. Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tests/manual/rccl_two_ranks_one_gpu.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hnanosolver_amd import api, device as D, dist as HD, fields  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")  # bootstrap only: the unique id travels over gloo, the halo over RCCL
origins, R = fields.dense_leaves(32), 32
f = fields.synthetic_fields(origins, R)
d = HD.DistRank(origins, world, rank, 1.0 / R, n_scalars=1)
try:
    d.connect_rccl()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: connect_rccl failed: {e}", flush=True)
    dist.destroy_process_group()
    sys.exit(3)
d.upload(d.owned_voxels(f["vel"]), [d.owned_voxels(f["density"])])  # (the library's own ownership rule: slabs along the cheapest axis, not ranges of the leaf list)
st = int(torch.cuda.current_stream().cuda_stream)
for _ in range(2):
    d.core_substep(7, 1.0 / 24.0, st)
d.synchronize(st)
got = d.download()
grid = api.create_grid_from_leaves(origins, 1.0 / R)
sim = D.Sim(grid, ["density"])
want = {"vel": f["vel"].copy(), "density": f["density"].copy()}
sim.upload(want)
for _ in range(2):
    sim.core_substep(7, 1.0 / 24.0, 1.0 / R, D.current_stream())
sim.download(want)
ok = np.array_equal(got["vel"], d.owned_voxels(want["vel"])) and np.array_equal(got["scalars"][0], d.owned_voxels(want["density"]))
print(f"rank {rank}: RCCL transport bit-identical to the single grid: {ok}; info {d.info()['bytes_sent']}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 4)
