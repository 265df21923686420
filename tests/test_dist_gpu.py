"""The native multi-GPU path (csrc/hns_dist_*.hip) on ONE device: every rank of a decomposition lives in this process and a
message is a device copy out of the peer's send buffer (hns_dist_connect_local) -- same plan, launch ranges, pack/unpack
kernels, communication stream and events as the RCCL transport. Owned results must be bit-identical to the single-grid
device run (which tests/test_fullsize_gpu.py ties to the oracle), including BASELINE.json configs[4]: the 1024^3-extent
plume split into 8 leaf ranges."""
import numpy as np
import pytest

from hnanosolver_amd import dist as HD
from hnanosolver_amd import fields

pytestmark = pytest.mark.gpu


def single_grid(origins, R, names, iters, substeps, dt=1.0 / 24.0):
    from hnanosolver_amd import api, device as D

    f = fields.synthetic_fields(origins, R)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    sim = D.Sim(grid, names)
    arrays = {"vel": f["vel"].copy(), **{n: f[n].copy() for n in names}}
    sim.upload(arrays)
    for _ in range(substeps):
        sim.core_substep(iters, dt, 1.0 / R, D.current_stream())
    sim.download(arrays)
    return f, arrays


def run_local(origins, R, world, k, names, iters, substeps, dt=1.0 / 24.0):
    import torch

    f = fields.synthetic_fields(origins, R)
    ranks = [HD.DistRank(origins, world, r, 1.0 / R, n_scalars=len(names), sweeps_per_exchange=k) for r in range(world)]
    HD.DistRank.connect_local(ranks)
    b = None  # (which leaves a rank owns is the rank's own knowledge: DistRank.owned_ids / owned_voxels)
    for r, d in enumerate(ranks):
        d.upload(d.owned_voxels(f["vel"]), [d.owned_voxels(f[n]) for n in names])
    stream = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(substeps):
        HD.DistRank.local_core_substep(ranks, iters, dt, stream)
    for d in ranks:
        d.synchronize(stream)
    return ranks, b


def check(ranks, b, want, names):
    # the ghost voxels the next kernels read hold their owners' bits (velocity: whole leaves; p: reach 1), whatever the transport wrote them with
    n_pairs, bad = HD.DistRank.ghost_check_local(ranks)
    assert not bad and (n_pairs > 0 or len(ranks) == 1), (n_pairs, bad[:3])
    for r, d in enumerate(ranks):
        got = d.download()
        assert np.array_equal(got["vel"], d.owned_voxels(want["vel"])), f"rank {r} velocity"
        for n, a in zip(names, got["scalars"]):
            assert np.array_equal(a, d.owned_voxels(want[n])), f"rank {r} {n}"


def scattered_leaves():
    rng = np.random.default_rng(5)
    lat = np.stack(np.meshgrid(*[np.arange(-6, 6)] * 3, indexing="ij"), -1).reshape(-1, 3)
    o = (lat[rng.random(len(lat)) < 0.5] * 8).astype(np.int32)
    return np.ascontiguousarray(o[fields.nanovdb_order(o)])


@pytest.mark.parametrize("name,world,k", [("dense32", 2, 4), ("plume", 3, 2), ("scattered", 5, 1), ("plume", 8, 3), ("dense32", 4, 0), ("dense32", 2, 1),
                                          ("plume", 8, 1), ("scattered", 3, -1), ("dense32", 4, 1)])
def test_local_ranks_match_single_grid(name, world, k):
    """Ranks of a few hundred leaves (one-leaf SOR blocks): the exchanged substep at every sweeps_per_exchange (1: a refresh of p behind every iteration); k = -1: the same with
    option dist_mirror = 0 spelled out. (Rounds 2-5 chained such ranks at k = 1 through a mirroring one-iteration kernel; the chained substep is that of 16^3-block ranks now:
    test_chained_ranks_two_iterations_per_launch.)"""
    import hnanosolver_amd as H

    if k == -1:
        H.set_option("dist_mirror", "0")
        try:
            return _local_ranks_match_single_grid(name, world, 1)
        finally:
            H.set_option("dist_mirror", None)
    return _local_ranks_match_single_grid(name, world, k)


def big_scattered_leaves():
    rng = np.random.default_rng(9)
    lat = np.stack(np.meshgrid(*[np.arange(-9, 9)] * 3, indexing="ij"), -1).reshape(-1, 3)
    o = (lat[rng.random(len(lat)) < 0.7] * 8).astype(np.int32)
    return np.ascontiguousarray(o[fields.nanovdb_order(o)])


def _local_ranks_match_single_grid(name, world, k, iters=7):
    origins, R = {"dense32": (fields.dense_leaves(32), 32), "plume": (fields.plume_leaves(8, 1.5, 0.35), 64), "scattered": (scattered_leaves(), 96),
                  "dense64": (fields.dense_leaves(64), 64), "plume16": (fields.plume_leaves(16, 1.5, 0.3), 128), "scattered_big": (big_scattered_leaves(), 144)}[name]
    names, substeps = ["density", "temperature"], 2
    _, want = single_grid(origins, R, names, iters, substeps)
    ranks, b = run_local(origins, R, world, k, names, iters, substeps)
    check(ranks, b, want, names)
    info = [d.info() for d in ranks]
    assert all(i["sweeps_per_exchange"] == (k or 4) for i in info)
    assert sum(i["boundary_leaves"] + i["interior_leaves"] for i in info) == len(origins)
    assert all(sum(i["bytes_sent"].values()) > 0 for i in info if i["peers"])


@pytest.mark.parametrize("name,world,k", [("plume", 5, 2), ("scattered", 5, 2), ("scattered", 3, 4), ("dense32", 2, 2)])
def test_exchanged_ranks_with_blocked_range_sweeps(name, world, k):
    """Round 4: with sweeps_per_exchange >= 2 a rank's boundary and interior launch ranges take the last two iterations in front of
    every exchange in ONE temporally blocked launch each (ghost leaves = tile sources; k = 2: no ghost leaf is ever swept). Equal to the single grid bit for bit."""
    _local_ranks_match_single_grid(name, world, k)


def test_exchanged_blocked_boundary_sweep_packs_its_own_messages():
    """Round 5: on the exchanged path (option dist_mirror = 0 over the local transport = what RCCL ranks run) with sweeps_per_exchange = 2 and a boundary
    range of more than 600 leaves, the boundary sweep of the 16^3 kernel writes the voxels its peers read straight into their messages (PackMirror) and the
    pack launch is skipped. The owned voxels equal the single grid bit for bit."""
    import hnanosolver_amd as H

    R, world, iters = 208, 2, 6
    origins = fields.dense_leaves(R)  # 26^3 leaves: a slab face is 676 boundary leaves, swept in 16^3 blocks
    names = ["density"]
    _, want = single_grid(origins, R, names, iters, 1)
    H.set_option("dist_mirror", "0")
    try:
        ranks, b = run_local(origins, R, world, 2, names, iters, 1)
        check(ranks, b, want, names)
        info = [d.info() for d in ranks]
    finally:
        H.set_option("dist_mirror", None)
    assert all(i["boundary_leaves"] > 600 for i in info), info
    # three exchanges of p per substep (6 iterations, two per exchange), every one packed by the boundary sweep
    assert all(i["packed_exchanges"] == iters // 2 for i in info), [(i["packed_exchanges"], i["exchanges"]) for i in info]


@pytest.mark.parametrize("name,world,iters", [("dense64", 2, 7), ("dense64", 5, 6), ("plume16", 3, 7), ("scattered_big", 4, 9), ("scattered_big", 2, 2)])
def test_chained_ranks_two_iterations_per_launch(name, world, iters):
    """Round 4: sweeps_per_exchange = 2 over the local / ipc transport with more than 600 leaves per rank = the chained substep whose
    pressure loop is the temporally blocked kernel, two iterations per launch, boundary blocks waiting for the peers' previous launch
    and writing the reach-4 region of p into the peers' ghost voxels themselves (k_rbgs_block_xy<., PhaseMirror>); an odd
    iteration left over is one more chained launch of the same kernel with two colour sweeps instead of four. Bit-identical to the single grid."""
    _local_ranks_match_single_grid(name, world, 2, iters)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_plume1024_in_8_ranges_matches_single_grid(k):
    """BASELINE.json configs[4]: the 1024^3-extent sparse plume (65,944 leaves) as the 8-GPU decomposition -- 8 contiguous
    leaf ranges, each with its ghost layer, boundary-first launch ranges and voxel-granular halo messages -- emulated on one
    device, 50 iterations, against the single-grid run of the same substep. k = 0: the exchanged pressure loop (refresh every
    4th sweep); k = 1: exchanged behind every iteration (one-iteration launches of the 16^3 kernel); k = 2 (round 4): the chained
    substep with the temporally blocked kernel, two iterations per chained launch, no exchanges."""
    origins, R = fields.config_leaves("plume1024")
    names, iters = ["density"], 50
    _, want = single_grid(origins, R, names, iters, 1)
    ranks, b = run_local(origins, R, 8, k, names, iters, 1)
    check(ranks, b, want, names)
    info = [d.info() for d in ranks]
    assert max(i["peers"] for i in info) <= 7 and min(i["boundary_leaves"] for i in info) > 0
    for i in info:
        assert i["chained"] == (1 if k == 2 else 0)
        if k == 0:  # payload accounting: the pressure loop dominates; with k = 4 that is 12 refreshes of depth 8 plus the final depth-1 one
            assert i["exchanges"] == 1 + 1 + 1 + 13 + 1 + 1
            assert i["bytes_sent"]["p"] == 12 * 4 * i["region_voxels_sent"]["p"]
        elif k == 1:  # 49 refreshes of depth 2 plus the final depth-1 one
            assert i["exchanges"] == 1 + 1 + 1 + 50 + 1 + 1
            assert i["bytes_sent"]["p"] == 49 * 4 * i["region_voxels_sent"]["p"]
        else:  # no exchange in the pressure loop: 25 blocked launches each mirror the reach-4 region
            assert i["exchanges"] == 1  # the advection inputs of the first substep; every kernel after that delivers its own halo
            assert i["bytes_sent"]["p"] == 25 * 4 * i["region_voxels_sent"]["p"]


def test_new_fields_between_substeps_and_many_substeps():
    """upload after a substep (the scalars already posted for the next substep are dropped), then more substeps"""
    origins, R = fields.plume_leaves(8, 1.5, 0.35), 64
    names, iters = ["density"], 5
    _, want = single_grid(origins, R, names, iters, 3)
    ranks, b = run_local(origins, R, 3, 2, names, iters, 2)  # two substeps on fields that are then replaced
    import torch

    f = fields.synthetic_fields(origins, R)
    for r, d in enumerate(ranks):
        d.upload(d.owned_voxels(f["vel"]), [d.owned_voxels(f["density"])])
    stream = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        HD.DistRank.local_core_substep(ranks, iters, 1.0 / 24.0, stream)
    for d in ranks:
        d.synchronize(stream)
    check(ranks, b, want, names)


def test_world_size_one_equals_sim():
    import torch

    origins, R = fields.dense_leaves(32), 32
    names, iters = ["density"], 5
    f, want = single_grid(origins, R, names, iters, 2)
    d = HD.DistRank(origins, 1, 0, 1.0 / R, n_scalars=1)
    d.upload(f["vel"], [f["density"]])
    stream = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        d.core_substep(iters, 1.0 / 24.0, stream)
    d.synchronize(stream)
    got = d.download(pressure=True)
    assert np.array_equal(got["vel"], want["vel"]) and np.array_equal(got["scalars"][0], want["density"])
    assert np.isfinite(got["pressure"]).all() and got["pressure"].any()
    i = d.info()
    assert i["peers"] == 0 and i["ghost_leaves"] == 0 and i["boundary_leaves"] == 0


def test_loopback_transport_two_stream_structure_terminates_and_repeats():
    """The RCCL path's structure -- boundary kernels, pack, transfer, unpack on a communication stream under the interior
    kernels, events both ways -- with every message answered out of the rank's own send buffer (TIMING transport: the values
    are meaningless, but the run must terminate, be finite and repeat bit for bit)."""
    import torch

    origins, R = fields.dense_leaves(64), 64
    f = fields.synthetic_fields(origins[: len(origins) // 2], R)
    outs = []
    for _ in range(2):
        d = HD.DistRank(origins, 2, 0, 1.0 / R, n_scalars=1, sweeps_per_exchange=2)
        d.connect_loopback()
        d.upload(f["vel"], [f["density"]])
        stream = int(torch.cuda.current_stream().cuda_stream)
        for _s in range(3):
            d.core_substep(9, 1.0 / 24.0, stream)
        d.synchronize(stream)
        got = d.download()
        assert np.isfinite(got["vel"]).all() and np.isfinite(got["scalars"][0]).all()
        outs.append(got)
        assert d.info()["exchanges"] == 1 + 1 + 5 + 1 + 1  # (the scalars of the next substep were already in flight)
        d.close()
    assert np.array_equal(outs[0]["vel"], outs[1]["vel"]) and np.array_equal(outs[0]["scalars"][0], outs[1]["scalars"][0])


def _loopback_run(origins, R, world, rank, k, rccl, substeps=2, iters=9):
    import torch

    d = HD.DistRank(origins, world, rank, 1.0 / R, n_scalars=1, sweeps_per_exchange=k)
    f = fields.synthetic_fields(origins[d.owned_ids], R)
    d.connect_loopback(rccl=rccl)
    d.upload(f["vel"], [f["density"]])
    stream = int(torch.cuda.current_stream().cuda_stream)
    for _s in range(substeps):
        d.core_substep(iters, 1.0 / 24.0, stream)
    d.synchronize(stream)
    got = d.download(pressure=True)
    info = d.info()
    d.close()
    return got, info


@pytest.mark.parametrize("case", ["dense_k4", "dense_k2", "plume_4ranks"])
def test_rccl_carries_the_loopback_messages_exactly(case):
    """What one GPU can verify of the RCCL transport: the same groups of ncclSend / ncclRecv the multi-rank path issues (one
    per field and peer, out of the field itself for unpacked whole-leaf regions, out of the message buffers otherwise), on
    the communication stream, through a one-rank communicator to itself -- must leave bit for bit what the copy-based
    loopback leaves (both answer every message with the rank's own payload)."""
    if case == "plume_4ranks":
        origins, R, world, rank, k = fields.plume_leaves(12, 1.8, 0.28), 96, 4, 1, 3
    else:
        origins, R, world, rank, k = fields.dense_leaves(64), 64, 2, 0, 4 if case == "dense_k4" else 2
    want, wi = _loopback_run(origins, R, world, rank, k, rccl=False)
    got, gi = _loopback_run(origins, R, world, rank, k, rccl=True)
    assert wi == gi and gi["messages_sent"] > 0
    for key in ("vel", "pressure"):
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(got["scalars"][0], want["scalars"][0])


def _run_processes(world, case, k, iters, substeps, tmp_path, timeout=420):
    import os
    import socket
    import subprocess
    import sys

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_process_worker.py")
    # output to a file per rank: a rank that fills a pipe nobody is reading yet would block inside a collective its peers wait in
    logs = [open(os.path.join(str(tmp_path), f"rank{r}.log"), "w+") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), case, str(k), str(iters), str(substeps), str(tmp_path)],
                              stdout=logs[r], stderr=subprocess.STDOUT, text=True) for r in range(world)]
    try:
        import time
        deadline = time.monotonic() + timeout
        for p in procs:
            p.wait(timeout=max(1.0, deadline - time.monotonic()))
    finally:
        for p in procs:  # exactly the processes started here
            if p.poll() is None:
                p.kill()
    outs = []
    for f in logs:
        f.seek(0)
        outs.append(f.read())
        f.close()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    return [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]


@pytest.mark.parametrize("case,world,k,iters", [("plume", 3, 2, 7), ("plume12", 4, 3, 9), ("dense64", 2, 0, 50), ("dense64", 2, 1, 50), ("plume12", 4, 1, 9), ("dense64", 2, 2, 50), ("dense64", 3, 2, 7),
                                                # round 6, ranks of 2,048 / 1,365 leaves on the EXCHANGED path (dist_mirror = 0: what RCCL ranks run): the pressure loop as one launch over all owned
                                                # leaves that packs its own messages, transfer and unpack behind it on the compute stream; divergence and gradient in line too; and the split form
                                                ("128@dist_mirror=0", 2, 2, 9), ("128@dist_mirror=0", 2, 0, 10), ("128@dist_mirror=0", 3, 2, 7), ("128@dist_mirror=0@dist_unsplit=0", 2, 2, 9),
                                                ("128@dist_mirror=0@dist_unsplit=0", 3, 0, 10)])
def test_one_process_per_rank_over_mapped_peer_memory(case, world, k, iters, tmp_path):
    """The multi-process path for real: `world` PROCESSES (here sharing the one GPU), each a rank with its own streams, the
    halos put into the peer's memory through hipIpc mappings and the ranks meeting through device-side flags while their
    kernels run concurrently (hns_dist_connect_ipc). Owned results equal the single-grid run bit for bit. (At most 4 processes:
    with 8 sharing one GPU a rank's waiting wave can keep the device from ever scheduling the process it waits for, and the
    20 s bounded wait ran out in about half of the runs -- reported as an error, not a hang; with one process per GPU a
    waiting wave shares the device with nobody it waits for.)"""
    from dist_process_worker import case_leaves

    origins, R = case_leaves(case.split("@")[0])
    names, substeps = ["density", "temperature"], 2
    _, want = single_grid(origins, R, names, iters, substeps)
    got = _run_processes(world, case, k, iters, substeps, tmp_path)
    for r, g in enumerate(got):
        ids = HD.owned_ids_of(origins, world, r)
        assert np.array_equal(g["vel"], HD.take_leaves(want["vel"], ids)), f"rank {r} velocity"
        for n in names:
            assert np.array_equal(g[n], HD.take_leaves(want[n], ids)), f"rank {r} {n}"
        assert int(g["messages"]) > 0


@pytest.mark.parametrize("extra", [[], ["--config", "plume", "--partition"]])
def test_bench_py_as_two_processes_sharing_the_gpu(extra, tmp_path):
    """bench.py's N > 1 path end to end, launched the way the driver launches it (torch.distributed.run, one process per
    rank) with both ranks on the one GPU: transport selection (the chained one-sided path checked against the exchanged one
    in place), the timed loop, the JSON line. RCCL itself refuses two ranks on one device, so the reference transport of the
    start-up check is the exchanged ipc path here (--share-one-gpu)."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--share-one-gpu", "--transport", "auto", "--steps", "3", "--warmup", "1", "--iterations", "10"] + (extra or ["--config", "64"])
    p = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["value"] > 0 and j["unit"] == "substeps/s"
    assert j["scaling"] == ("strong" if extra else "weak")
    if extra:  # ranks of ~2,000 leaves: the chained one-sided substep, checked against the exchanged one in place
        assert "verified bit for bit" in j["config"]["parallelism"], j["config"]["parallelism"]
    else:  # ranks of 256 leaves (one-leaf SOR blocks) run the exchanged substep over either transport: `auto` says so and stays on the reference transport
        assert "one-sided transport not used" in j["config"]["parallelism"] and "600 leaves and fewer" in j["config"]["parallelism"], j["config"]["parallelism"]
    assert "bit-identical to the single-GPU run of the whole domain" in j["config"]["verified"], j["config"]["verified"]  # what was timed was checked against one GPU first
    assert "bit-equal to their owners' values" in j["config"]["ghosts"], j["config"]["ghosts"]  # and after the timed loop its ghost voxels were compared with their owners
    # (the chained one-sided substep: k = 2 with the temporally blocked sweep; the exchanged one at the library's default, 4)
    assert j["config"]["halo"]["sweeps_per_exchange"] == (2 if extra else 4) and j["config"]["halo"]["bytes_sent"]["p"] > 0


def test_bench_py_launches_its_own_ranks():
    """`python bench.py --gpus 2 --share-one-gpu`, no launcher around it (the form the driver uses for --gpus 1): bench.py
    starts its two ranks itself as a child process and exactly one JSON line comes out."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-one-gpu", "--steps", "3", "--warmup", "1", "--iterations", "10", "--config", "64"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["value"] > 0
    assert "bit-identical to the single-GPU run of the whole domain" in j["config"]["verified"], j["config"]["verified"]
    assert "bit-equal to their owners' values" in j["config"]["ghosts"], j["config"]["ghosts"]


def test_bench_py_default_line_carries_the_strong_scaling_record():
    """`python bench.py --gpus 2` with the default workload (what the driver runs for its scaling curve): ONE JSON line with the 256^3 weak-scaling headline
    and, as `strong_scaling`, BASELINE config 5 -- the 1024^3-extent plume as one domain over the same ranks, slab partition -- both verified against the
    single-GPU run and with their ghost voxels checked. Both ranks on the one GPU here (--share-one-gpu)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-one-gpu", "--steps", "2", "--warmup", "1", "--iterations", "6"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["workload"].startswith("256^3") and j["value"] > 0
    s = j["strong_scaling"]
    assert "error" not in s, s
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and s["value"] > 0 and s["config"]["leaves"] == 65944
    assert "bit-identical to the single-GPU run of the whole domain" in s["config"]["verified"], s["config"]["verified"]
    assert "bit-equal to their owners' values" in s["config"]["ghosts"], s["config"]["ghosts"]
    assert s["config"]["halo"]["halo_peers"] == 1 and "slabs of leaves along axis y" in s["config"]["parallelism"]
    # round 6: and the same domain again over the one-sided transport where it connects and reproduces the reference transport bit for bit (ranks of ~33,000 leaves: the
    # chained substep) -- a record of its own under its own watchdog, so that a hang on a machine this path has never seen cannot cost the line
    o = j["strong_scaling_one_sided"]
    assert "error" not in o, o
    assert o["n_gpus"] == 2 and o["scaling"] == "strong" and o["value"] > 0 and o["config"]["leaves"] == 65944
    assert "verified bit for bit" in o["config"]["parallelism"], o["config"]["parallelism"]
    assert "bit-identical to the single-GPU run of the whole domain" in o["config"]["verified"], o["config"]["verified"]
    assert o["config"]["halo"]["sweeps_per_exchange"] == 2 and o["config"]["halo"]["exchanges"] == 0


def test_unconnected_ranks_refuse_to_step():
    import hnanosolver_amd as H

    origins = fields.dense_leaves(16)
    d = HD.DistRank(origins, 2, 0, 1.0 / 16)
    f = fields.synthetic_fields(origins[d.owned_ids], 16)
    d.upload(f["vel"], [f["density"]])
    with pytest.raises(H.HNSError):
        d.core_substep(3, 1.0 / 24.0)


@pytest.mark.parametrize("partition", [False, True])
def test_bench_driver_single_rank(partition):
    """bench.py's multi-GPU driver with world = 1 (no peers): runs, times its pressure loop, equals the single grid."""
    o, R = fields.plume_leaves(8, 1.0, 0.3), 64
    b = HD.SlabBench(o, R, 0, 1, 6, 1.0 / 24.0, partition=partition)
    b.step()
    b.timing_on(4)
    b.step()
    ms, sweeps = b.pressure_time()
    assert sweeps == 6 and ms > 0.0
    _, want = single_grid(o, R, ["density"], 6, 2)
    got = b.rank_obj.download()
    assert np.array_equal(got["vel"], want["vel"]) and np.array_equal(got["scalars"][0], want["density"])


def test_back_trace_beyond_the_ghost_layer_is_reported():
    """A rank holds one layer of ghost leaves. With |u| dt / dx ~ 21 voxels (SURVEY 8d's long-backtrace case, A = 400 / R) taps leave
    the 27-leaf neighbourhood of their leaf: the single grid follows them through its origin hash, a rank cannot tell a leaf on
    another rank from no leaf at all -- so it says so instead of returning something else. Uploading tame fields clears it."""
    import hnanosolver_amd as H
    import torch

    R, world, names, iters = 32, 2, ["density"], 5
    origins = fields.dense_leaves(R)
    ranks = [HD.DistRank(origins, world, r, 1.0 / R, n_scalars=1, sweeps_per_exchange=4) for r in range(world)]
    HD.DistRank.connect_local(ranks)
    stream = int(torch.cuda.current_stream().cuda_stream)

    def upload(amplitude):
        f = fields.synthetic_fields(origins, R, amplitude_voxels=amplitude)
        for r, d in enumerate(ranks):
            d.upload(d.owned_voxels(f["vel"]), [d.owned_voxels(f[n]) for n in names])

    upload(400.0)
    HD.DistRank.local_core_substep(ranks, iters, 1.0 / 24.0, stream)
    raised = 0
    for d in ranks:
        try:
            d.synchronize(stream)
        except H.HNSError as e:
            assert "ghost layer" in str(e)
            raised += 1
    assert raised == world
    with pytest.raises(H.HNSError, match="ghost layer"):
        HD.DistRank.local_core_substep(ranks, iters, 1.0 / 24.0, stream)
    with pytest.raises(H.HNSError, match="ghost layer"):
        ranks[0].download()
    upload(96.0)  # the benchmark amplitude (~5 voxels): within the layer, and the earlier failure is forgotten
    HD.DistRank.local_core_substep(ranks, iters, 1.0 / 24.0, stream)
    for d in ranks:
        d.synchronize(stream)
    _, want = single_grid(origins, R, names, iters, 1)
    check(ranks, None, want, names)


SIM_NAMES = ["density", "temperature", "fuel", "waste", "flame", "collision_sdf"]


def _sim_fields(origins, R):
    f = fields.synthetic_fields(origins, R)
    f["collision_sdf"] = fields.sphere_sdf(origins, R)
    f["waste"] = (0.05 * f["density"]).astype(np.float32)  # burning state: every combustion branch is exercised (tests/kats.py has the table)
    f["flame"] = (0.3 * f["fuel"]).astype(np.float32)
    return f


@pytest.mark.parametrize("name,world,k,coll,factor_scale", [("plume", 8, 4, False, 0.5), ("plume", 8, 2, True, 1.0), ("dense32", 2, 4, True, 0.5), ("scattered", 5, 1, False, 2.0),
                                                           ("dense32", 3, 3, False, 1.0), ("scattered", 4, 4, True, 1.0)])
def test_partitioned_compute_sim_matches_single_grid(name, world, k, coll, factor_scale):
    """The WHOLE Compute_Sim substep (collision, advect_vector, vorticity confinement, divergence, combustion, buoyancy, solve, gradient,
    collision, advect_scalars; reference HNanoSolver.cu:150-356) on a domain split into leaf ranges: owned results of three chained
    substeps equal hns_sim_substep's on the one grid bit for bit."""
    import torch
    from hnanosolver_amd import api, device as D

    origins, R = {"dense32": (fields.dense_leaves(32), 32), "plume": (fields.plume_leaves(16, 1.5, 0.3), 128), "scattered": (scattered_leaves(), 96)}[name]
    # (buoyancy and confinement kept gentle: at the SOP defaults the plume accelerates by ~10 voxels per step and step, and a back-trace that long leaves a
    # rank's ghost layer -- which test_back_trace_beyond_the_ghost_layer_is_reported covers)
    params = api.CombustionParams(factorScale=factor_scale, vorticityScale=0.01, buoyancyStrength=0.05)
    iters, dt, substeps = 9, 1.0 / 24.0, 3
    f = _sim_fields(origins, R)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    sim = D.Sim(grid, SIM_NAMES)
    want = {"vel": f["vel"].copy(), **{n: f[n].copy() for n in SIM_NAMES}}
    sim.upload(want)
    for _ in range(substeps):
        sim.substep(iters, dt, 1.0 / R, params, coll, D.current_stream())
    sim.download(want)

    ranks = [HD.DistRank(origins, world, r, 1.0 / R, n_scalars=len(SIM_NAMES), sweeps_per_exchange=k) for r in range(world)]
    HD.DistRank.connect_local(ranks)
    b = None
    for r, d in enumerate(ranks):
        d.upload(d.owned_voxels(f["vel"]), [d.owned_voxels(f[n]) for n in SIM_NAMES])
    stream = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(substeps):
        HD.DistRank.local_sim_substep(ranks, SIM_NAMES, iters, dt, params, coll, stream)
    for d in ranks:
        d.synchronize(stream)
    check(ranks, b, want, SIM_NAMES)


def test_partitioned_compute_sim_refuses_what_the_reference_refuses():
    import hnanosolver_amd as H
    from hnanosolver_amd import api

    origins = fields.dense_leaves(16)
    ranks = [HD.DistRank(origins, 2, r, 1.0 / 16, n_scalars=2, sweeps_per_exchange=4) for r in range(2)]
    HD.DistRank.connect_local(ranks)
    with pytest.raises(H.HNSError, match="Missing required input field for combustion"):
        HD.DistRank.local_sim_substep(ranks, ["density", "fuel"], 3, 1.0 / 24.0, api.CombustionParams())
