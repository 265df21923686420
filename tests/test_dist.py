"""Multi-rank path on CPU: the partition plan of libhns.so (hns_dist_create, plan-only) checked against brute force, and
walked with the oracle as compute engine over a real wire (gloo, world 2/3/4/8): owned results of the partitioned run are
BIT-IDENTICAL to the single-domain run. The same plan drives the HIP kernels in tests/test_dist_gpu.py."""
import os
import socket
import sys

import numpy as np
import pytest

from hnanosolver_amd import dist as HD
from hnanosolver_amd import fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def random_leaves(seed=0, shape=(12, 5, 5), keep=0.7):
    rng = np.random.default_rng(seed)
    lat = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).reshape(-1, 3)
    o = (lat[rng.random(len(lat)) < keep] * 8).astype(np.int32)
    return np.ascontiguousarray(o[fields.nanovdb_order(o)])


def brute_neighbors(o):
    index = {tuple(c): i for i, c in enumerate(o.tolist())}
    nbr = np.full((len(o), 27), -1, dtype=np.int64)
    for i, c in enumerate(o.tolist()):
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    nbr[i, (dx + 1) * 9 + (dy + 1) * 3 + dz + 1] = index.get((c[0] + 8 * dx, c[1] + 8 * dy, c[2] + 8 * dz), -1)
    return nbr


@pytest.mark.parametrize("leaf_order", [False, True])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("k", [1, 2, 4])
def test_plan_properties(world, k, leaf_order):
    """both partitions: slabs along the cheapest axis (round 5) and contiguous ranges of the caller's list (HNS_DIST_LEAF_ORDER)"""
    o = random_leaves()
    n = len(o)
    nbr = brute_neighbors(o)
    bounds = HD.partition_bounds(n, world)
    ranks = [HD.DistRank(o, world, r, 1.0 / 64, 2, k, plan_only=True, leaf_order=leaf_order) for r in range(world)]
    # the partition: every leaf owned once, equal shares, the same axis on every rank; slabs: sorted along that axis, the caller's order inside a plane
    part = np.concatenate([d.owned_ids for d in ranks])
    assert sorted(part.tolist()) == list(range(n)) and [d.n_owned for d in ranks] == np.diff(bounds).tolist()
    axis = ranks[0].partition_axis
    assert all(d.partition_axis == axis for d in ranks)
    if leaf_order or world == 1:
        assert axis == -1
    if axis < 0:
        assert part.tolist() == list(range(n))
    else:  # slabs: rank r's leaves lie before rank r+1's along the axis (a plane two ranks share is split by the caller's order); inside a rank: the caller's order
        for r, d in enumerate(ranks):
            assert (np.diff(d.owned_ids) > 0).all()
            if r + 1 < world and d.n_owned and ranks[r + 1].n_owned:
                a, b2 = o[d.owned_ids, axis], o[ranks[r + 1].owned_ids, axis]
                assert a.max() <= b2.min()
                if a.max() == b2.min():
                    assert d.owned_ids[a == a.max()].max() < ranks[r + 1].owned_ids[b2 == b2.min()].min()
    owner = np.empty(n, dtype=np.int64)
    ppos = np.empty(n, dtype=np.int64)  # position in partition order
    for r, d in enumerate(ranks):
        owner[d.owned_ids] = r
    ppos[part] = np.arange(n)

    def holders(g):  # the other ranks that hold a copy of leaf g: owners of its neighbours; of the caller's leaf 0 (whose element 0 every rank mirrors): everybody else
        if g == 0 and world > 1:
            return tuple(q for q in range(world) if q != owner[0])
        return tuple(sorted({int(owner[nb]) for nb in nbr[g] if nb >= 0 and owner[nb] != owner[g]}))
    plans = [(d.info(), d.local_leaves(), d.peers()) for d in ranks]
    depth = [99, 1, 2 * k - 1, 2 * k]
    vox = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3)  # x<<6|y<<3|z order
    for r, (info, loc, peers) in enumerate(plans):
        nB, nI, nG = info["boundary_leaves"], info["interior_leaves"], info["ghost_leaves"]
        owned = ranks[r].owned_ids
        assert ranks[r].first_owned == (-1 if axis >= 0 else (int(owned[0]) if len(owned) else 0))
        assert sorted(loc[: nB + nI].tolist()) == sorted(owned.tolist())
        # boundary = owned leaves another rank mirrors (+ global leaf 0 on its owner, mirrored everywhere)
        nb_owner = np.where(nbr[owned] >= 0, owner[np.maximum(nbr[owned], 0)], r)
        is_b = (nb_owner != r).any(axis=1)
        if world > 1 and r == owner[0]:
            is_b[owned == 0] = True
        # boundary leaves (and, below, ghosts) are ordered by WHO holds copies of them -- the sorted list of the other ranks, compared lexicographically --, then by
        # position in the partition: a whole-leaf region is then a run of consecutive local leaves (it travels straight out of the field); interior leaves: partition order
        assert loc[:nB].tolist() == sorted(owned[is_b].tolist(), key=lambda g: (holders(g), ppos[g])) and loc[nB:nB + nI].tolist() == owned[~is_b].tolist()
        ghosts = set(nbr[owned].reshape(-1).tolist()) - set(owned.tolist()) - {-1}
        if world > 1 and r != owner[0]:
            ghosts |= {0}
        assert set(loc[nB + nI:].tolist()) == ghosts and len(loc) == nB + nI + nG
        assert [p.rank for p in peers] == sorted(p.rank for p in peers)
        pos = nB + nI
        for p in peers:  # ghosts grouped by owner, in partition order inside a group
            mine = sorted((g for g in ghosts if owner[g] == p.rank), key=lambda g: (holders(g), ppos[g]))
            assert loc[pos:pos + len(mine)].tolist() == mine
            pos += len(mine)
            # what I send to p is exactly what p expects from me: same leaves (global ids), same masks, same order
            back = [q for q in plans[p.rank][2] if q.rank == r]
            assert len(back) == 1
            for t in range(4):
                s, e = p.send[t], back[0].recv[t]
                assert s.voxels == e.voxels
                assert loc[s.leaves].tolist() == plans[p.rank][1][e.leaves].tolist()
                assert np.array_equal(s.masks, e.masks)
            # region semantics by brute force: a ghost voxel travels iff a voxel I own lies within the stencil's reach
            for t in range(4):
                reg = p.recv[t]
                got = {int(loc[l]): np.unpackbits(m.reshape(64, 1), axis=1, bitorder="little").reshape(512).astype(bool) for l, m in zip(reg.leaves, reg.masks)}
                for g in mine:
                    want = np.zeros(512, dtype=bool)
                    for j in range(27):
                        nb = nbr[g, j]
                        if nb < 0 or owner[nb] != r:
                            continue
                        d = np.array([j // 9 - 1, (j // 3) % 3 - 1, j % 3 - 1])
                        dist = np.where(d < 0, vox + 1, np.where(d > 0, 8 - vox, 0)).sum(axis=1)  # L1 distance to the neighbour leaf's box
                        want |= dist <= depth[t]
                    if t == 0 and g == 0:
                        want[:] = True  # global leaf 0 (whole, so that the region stays unpacked): its element 0 is what advect_scalars' out-of-domain taps read
                    assert np.array_equal(got.get(g, np.zeros(512, dtype=bool)), want), (r, p.rank, t, g)
        assert info["region_voxels_sent"]["advection inputs"] == sum(p.send[0].voxels for p in peers)


def test_box_domains_keep_the_callers_leaf_order_and_the_plume_is_cut_along_its_axis():
    """The partition rule of round 5 on the two shapes bench.py uses: `world` 256^3 slabs stacked along x (the weak-scaling domain: whole 128-voxel NanoVDB nodes per rank)
    stay contiguous ranges of the caller's list -- rounds 1-4's partition, memory layout and all --, while BASELINE config 5's plume is cut into slabs along y, its own
    axis, where every rank exchanges halos with the rank before and the rank behind it only."""
    slab = fields.dense_leaves(256)
    for world in (2, 8):
        glob = HD.slab_domain(slab, 256, world)
        for r in (0, world - 1):
            d = HD.DistRank(glob, world, r, 1.0 / 256, 1, 0, plan_only=True)
            b = HD.partition_bounds(len(glob), world)
            assert d.partition_axis == -1 and d.first_owned == b[r] and d.owned_ids.tolist() == list(range(b[r], b[r + 1]))
            # an end slab has one halo peer; `peers` also counts who it shares the element-0 mirror with: rank 0 owns the caller's leaf 0 and sends it to everybody
            assert d.info()["halo_peers"] == 1 and d.info()["peers"] == (world - 1 if r == 0 else (1 if world == 2 else 2))
            d.close()
    d = HD.DistRank(HD.slab_domain(slab, 256, 8), 8, 3, 1.0 / 256, 1, 0, plan_only=True)
    assert d.info()["halo_peers"] == 2 and d.info()["peers"] == 3  # rank 2, rank 4, and rank 0 (the element-0 mirror)
    d.close()
    origins, R = fields.config_leaves("plume1024")
    info = []
    for r in range(8):
        d = HD.DistRank(origins, 8, r, 1.0 / R, 1, 2, plan_only=True)
        assert d.partition_axis == 1 and d.first_owned == -1 and d.n_owned == len(origins) * (r + 1) // 8 - len(origins) * r // 8
        info.append(d.info())
        d.close()
    assert [i["halo_peers"] for i in info] == [1, 2, 2, 2, 2, 2, 2, 1]
    assert sorted(i["peers"] for i in info) == [2, 2, 2, 2, 3, 3, 3, 7]  # one more where the owner of the caller's leaf 0 is not a slab neighbour; that owner: everybody
    assert HD.SlabBench._one_sided_k(len(origins), 8) == 2 and HD.SlabBench._one_sided_k(4800, 8) == 1  # the library's own rule (hns_dist_one_sided_sweeps)


def _case(name):
    if name == "dense":
        return fields.dense_leaves(32), 32
    if name == "plume8":  # a reduced rising plume for the 8-rank walk: ~230 leaves, ragged interfaces between the ranges
        return fields.plume_leaves(12, 1.8, 0.28), 96
    return fields.plume_leaves(8, 1.5, 0.35), 64


def _worker(rank, world, port, name, iters, k, out_dir):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from dist_reference import ReferenceRank
    from oracle_lib import oracle

    oracle().orc_set_threads(max(1, (os.cpu_count() or 8) // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        origins, R = _case(name)
        f = fields.synthetic_fields(origins, R)
        rr = ReferenceRank(origins, world, rank, 1.0 / R, 2, k, poison=7.0)  # ghosts start WRONG: the first exchange must repair them
        ids = rr.plan.owned_ids
        rr.load_owned(HD.take_leaves(f["vel"], ids), [HD.take_leaves(f["density"], ids), HD.take_leaves(f["temperature"], ids)])
        for _ in range(2):
            rr.core_substep(iters, 1.0 / 24.0)
        rr.complete()  # the scalars posted for the NEXT substep: drain them before any rank hangs up
        dist.barrier()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), u=rr.owned(rr.u), phi0=rr.owned(rr.phi[0]), phi1=rr.owned(rr.phi[1]), p=rr.owned(rr.p),
                 bytes_sent=rr.bytes_sent)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name,world,k", [("dense", 2, 4), ("plume", 3, 4), ("dense", 4, 2), ("plume8", 8, 1), ("plume8", 8, 3)])
def test_partitioned_substep_is_bit_identical_to_single_domain(tmp_path, name, world, k):
    import torch.multiprocessing as mp

    from oracle_lib import OracleGrid, oracle

    iters = 7  # not a multiple of any exchange period used here
    mp.spawn(_worker, args=(world, _free_port(), name, iters, k, str(tmp_path)), nprocs=world, join=True)
    origins, R = _case(name)
    f = fields.synthetic_fields(origins, R)
    G = OracleGrid(origins)
    vs, dt = 1.0 / R, 1.0 / 24.0
    u, phi = f["vel"].copy(), [f["density"].copy(), f["temperature"].copy()]
    omega = HD.omega_compute(vs)
    assert omega == float(oracle().orc_omega_compute(vs))
    inv_dx = float(np.float32(1.0) / np.float32(vs))
    for _ in range(2):
        adv = G.advect_vector(u, dt, inv_dx)
        div = G.divergence(adv, inv_dx)
        p = G.rbgs_iterations(div, float(np.float32(vs)), omega, iters)
        u = G.subtract_pressure_gradient(adv, p, inv_dx)
        phi = G.advect_scalars(u, phi, dt, inv_dx)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        ids = HD.owned_ids_of(origins, world, r)
        assert np.array_equal(z["u"], HD.take_leaves(u, ids)), f"rank {r} velocity"
        assert np.array_equal(z["p"], HD.take_leaves(p, ids)), f"rank {r} pressure"
        assert np.array_equal(z["phi0"], HD.take_leaves(phi[0], ids)) and np.array_equal(z["phi1"], HD.take_leaves(phi[1], ids)), f"rank {r} scalars"
        assert z["bytes_sent"] > 0


SIM_NAMES = ["density", "temperature", "fuel", "waste", "flame", "collision_sdf"]


def _sim_inputs(name):
    origins, R = _case(name)
    f = fields.synthetic_fields(origins, R)
    f["collision_sdf"] = fields.sphere_sdf(origins, R)
    f["waste"] = (0.05 * f["density"]).astype(np.float32)
    f["flame"] = (0.3 * f["fuel"]).astype(np.float32)
    return origins, R, f


def _sim_params(fs):
    from hnanosolver_amd import api

    return api.CombustionParams(factorScale=fs, vorticityScale=0.01, buoyancyStrength=0.05)  # gentle: back-traces stay inside the one-leaf ghost layer


def _sim_worker(rank, world, port, name, iters, k, coll, fs, out_dir):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from dist_reference import ReferenceRank
    from oracle_lib import oracle

    oracle().orc_set_threads(max(1, (os.cpu_count() or 8) // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        origins, R, f = _sim_inputs(name)
        rr = ReferenceRank(origins, world, rank, 1.0 / R, len(SIM_NAMES), k, poison=7.0)
        ids = rr.plan.owned_ids
        rr.load_owned(HD.take_leaves(f["vel"], ids), [HD.take_leaves(f[n], ids) for n in SIM_NAMES])
        for _ in range(2):
            rr.sim_substep(SIM_NAMES, iters, 1.0 / 24.0, _sim_params(fs), coll)
        rr.complete()
        dist.barrier()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), u=rr.owned(rr.u), **{n: rr.owned(rr.phi[i]) for i, n in enumerate(SIM_NAMES)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,k,coll,fs", [("plume", 2, 4, True, 1.0), ("dense", 3, 2, False, 0.5)])
def test_partitioned_compute_sim_walk_is_bit_identical_to_single_domain(tmp_path, name, world, k, coll, fs):
    """The WHOLE Compute_Sim substep walked through the library's plan on CPU (oracle as engine, gloo as wire; the phases of
    hns_dist_substep.hip: Step::run_full): owned results of two chained substeps equal the oracle's own Compute driver on the single domain.
    tests/test_dist_gpu.py ties the HIP path to the same answer."""
    import torch.multiprocessing as mp

    from oracle_lib import OracleGrid

    iters = 5
    mp.spawn(_sim_worker, args=(world, _free_port(), name, iters, k, coll, fs, str(tmp_path)), nprocs=world, join=True)
    origins, R, f = _sim_inputs(name)
    G = OracleGrid(origins)
    u = f["vel"].copy()
    phi = {n: f[n].copy() for n in SIM_NAMES}
    sdf0 = phi["collision_sdf"].copy()
    for _ in range(2):
        assert G.compute_sim(u, phi, iters, 1.0 / 24.0, 1.0 / R, _sim_params(fs), coll) == 0
        phi["collision_sdf"][:] = sdf0  # the reference's driver hands the SDF back zeroed (HNanoSolver.cu:364-369); the device-resident state keeps it
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        ids = HD.owned_ids_of(origins, world, r)
        assert np.array_equal(z["u"], HD.take_leaves(u, ids)), f"rank {r} velocity"
        for n in SIM_NAMES:
            assert np.array_equal(z[n], HD.take_leaves(phi[n], ids)), f"rank {r} {n}"


def test_ghost_digest_comparison_names_the_pair_that_differs():
    """DistRank.compare_ghost_digests (the host half of ghost_check): a pair is good only if owner and holder both reported and agree"""
    a = {(0, 1, -1, "owner"): "aa", (1, 0, -1, "ghost"): "bb", (0, 1, -2, "owner"): "cc"}
    b = {(0, 1, -1, "ghost"): "aa", (1, 0, -1, "owner"): "bb", (0, 1, -2, "ghost"): "cc"}
    assert HD.DistRank.compare_ghost_digests([a, b]) == (3, [])
    b[(0, 1, -2, "ghost")] = "xx"  # a stale ghost copy of p on rank 1
    assert HD.DistRank.compare_ghost_digests([a, b]) == (3, [(0, 1, -2)])
    del b[(1, 0, -1, "owner")]  # a region only one side knows of
    assert HD.DistRank.compare_ghost_digests([a, b]) == (3, [(0, 1, -2), (1, 0, -1)])
