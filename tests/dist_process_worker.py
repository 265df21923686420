"""One rank of a multi-PROCESS run of the native multi-GPU path, all processes on cuda:0 (tests/test_dist_gpu.py starts
`world` of these): host rendezvous over gloo on 127.0.0.1, halo transport hns_dist_connect_ipc (peer memory mapped with
hipIpc*, one-sided puts, device-side flags). Writes the rank's owned results to <outdir>/rank<r>.npz.

argv: rank world port case sweeps_per_exchange iterations substeps outdir"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def case_leaves(name):
    from hnanosolver_amd import fields

    if name == "dense32":
        return fields.dense_leaves(32), 32
    if name == "dense64":
        return fields.dense_leaves(64), 64
    if name == "plume":
        return fields.plume_leaves(8, 1.5, 0.35), 64
    if name == "plume12":
        return fields.plume_leaves(12, 1.8, 0.28), 96
    return fields.config_leaves(name)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    case, k, iters, substeps, outdir = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from hnanosolver_amd import dist as HD
    from hnanosolver_amd import fields

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hnanosolver_amd as H

    # every rank of this test sits on cuda:0: in-kernel waits of several processes can fill the device's wave slots and starve the
    # process they wait for (hns_dist_*.hip: "guarded" puts ONE waiting wave in front of every chained launch instead)
    H.set_option("dist_mirror", "guarded")
    case, *opts = case.split("@")  # "<case>@option=value@...": library options for this run
    for o in opts:
        H.set_option(*o.split("=", 1))
    try:
        origins, R = case_leaves(case)
        names = ["density", "temperature"]
        full = fields.synthetic_fields(origins, R)
        d = HD.DistRank(origins, world, rank, 1.0 / R, n_scalars=len(names), sweeps_per_exchange=k)
        d.connect_ipc()
        d.upload(d.owned_voxels(full["vel"]), [d.owned_voxels(full[n]) for n in names])
        stream = int(torch.cuda.current_stream().cuda_stream)
        dist.barrier()
        for _ in range(substeps):
            d.core_substep(iters, 1.0 / 24.0, stream)
        d.synchronize(stream)
        n_pairs, bad = d.ghost_check()  # collective: every ghost voxel of u and p against its owner's value, through the mapped peer memory
        assert not bad and n_pairs > 0, (n_pairs, bad[:3])
        got = d.download(pressure=True)
        info = d.info()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), vel=got["vel"], pressure=got["pressure"], exchanges=info["exchanges"], messages=info["messages_sent"],
                 **{n: a for n, a in zip(names, got["scalars"])})
        dist.barrier()  # nobody unmaps or frees while a peer may still be writing
        d.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
