"""Multi-rank path on CPU: leaf partition + halo exchange of hnanosolver_amd.dist under gloo, world_size 2 and 3.

The HIP engine cannot run here, so the exchange/partition logic is driven with a test-only engine built on the oracle
(this file is under tests/, where the oracle may be used). The assertion is the one that matters for the product:
owned results of the partitioned run are BIT-IDENTICAL to the single-domain run."""
import os
import socket
import sys

import numpy as np
import pytest

from hnanosolver_amd import dist as HD
from hnanosolver_amd import fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_properties():
    rng = np.random.default_rng(0)
    lat = np.stack(np.meshgrid(np.arange(12), np.arange(5), np.arange(5), indexing="ij"), -1).reshape(-1, 3)
    o = (lat[rng.random(len(lat)) < 0.7] * 8).astype(np.int32)
    o = o[fields.nanovdb_order(o)]
    nbr = HD.neighbor_ids(o)
    # neighbour table against brute force
    index = {tuple(c): i for i, c in enumerate(o.tolist())}
    for i in rng.integers(0, len(o), 50):
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    want = index.get((o[i, 0] + 8 * dx, o[i, 1] + 8 * dy, o[i, 2] + 8 * dz), -1)
                    assert nbr[i, (dx + 1) * 9 + (dy + 1) * 3 + dz + 1] == want
    for world in (1, 2, 3, 8):
        plans = [HD.make_plan(o, world, r, nbr=nbr) for r in range(world)]
        owned = np.concatenate([p.owned_global for p in plans])
        assert np.array_equal(np.sort(owned), np.arange(len(o)))
        for p in plans:
            own = set(p.owned_global.tolist())
            want = (set(nbr[p.owned_global].reshape(-1).tolist()) | {0}) - own - {-1}  # + the mirror of global leaf 0
            assert p.local_origins[p.outside_element // 512].tolist() == o[0].tolist()
            assert set(p.ghost_global.tolist()) == want
            assert np.array_equal(p.local_origins, o[np.concatenate([p.owned_global, p.ghost_global])])
            # what I receive from q is exactly what q sends me, in the same order
            for q, (r0, r1) in p.recv_ranges.items():
                ghosts_from_q = p.ghost_global[r0 - p.n_owned:r1 - p.n_owned]
                sent = plans[q].owned_global[plans[q].send_local[p.rank]]
                assert np.array_equal(ghosts_from_q, sent)
            for q in p.send_local:
                assert p.rank in plans[q].recv_ranges
            # peers that only trade the mirror of global leaf 0: symmetric, and never a spatial neighbour
            for q in p.mirror_only_peers:
                assert p.rank in plans[q].mirror_only_peers
                assert 0 in (p.rank, q)


def test_slab_plan_matches_generic_plan():
    slab, R = fields.dense_leaves(16), 16
    world = 3
    glob = np.concatenate([slab + np.array([r * R, 0, 0], dtype=np.int32) for r in range(world)])
    nbr = HD.neighbor_ids(glob)
    for r in range(world):
        a = HD.make_plan(glob, world, r, nbr=nbr)
        b = HD.make_plan_slabs(glob, len(slab), world, r, nbr)
        assert np.array_equal(a.local_origins, b.local_origins)
        assert a.recv_ranges == b.recv_ranges
        assert a.send_local.keys() == b.send_local.keys()
        for q in a.send_local:
            assert np.array_equal(a.send_local[q], b.send_local[q])


class OracleEngine:
    """TEST-ONLY engine: runs the oracle on the rank's local leaves (CPU torch tensors, Vec3f AoS velocity like the device engine)."""

    def __init__(self, local_origins, n_owned, voxel_size):
        import torch

        from oracle_lib import OracleGrid

        self.torch = torch
        self.G = OracleGrid(local_origins)
        self.no = n_owned * 512

    def set_outside_element(self, idx):
        self.G.set_outside_element(idx)

    def zeros(self, n):
        return self.torch.zeros(n, dtype=self.torch.float32)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    def ids(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))

    def pack(self, fld, ids, out, ncomp):
        out.copy_(fld.reshape(-1, 512 * ncomp)[ids].reshape(-1))
        return out

    def _aos(self, u):
        return np.ascontiguousarray(u.numpy().reshape(-1, 3))

    def _put(self, dst, src_np):  # only owned leaves are written, like the HIP kernels (n_active)
        dst[: self.no].copy_(self.torch.from_numpy(np.ascontiguousarray(src_np))[: self.no])

    def advect_vector(self, u, out, dt, inv_dx):
        self._put(out, self.G.advect_vector(self._aos(u), dt, inv_dx))

    def advect_scalars(self, u, srcs, dsts, dt, inv_dx):
        r = self.G.advect_scalars(self._aos(u), [s.numpy() for s in srcs], dt, inv_dx)
        for d, x in zip(dsts, r):
            self._put(d, x)

    def divergence(self, u, div, inv_dx):
        self._put(div, self.G.divergence(self._aos(u), inv_dx))

    def rbgs_iteration(self, div, p_in, p_out, dx, omega, include_ghosts=False):
        p = p_in.numpy().copy()
        self.G.rbgs(div.numpy(), p, dx, 0, omega)
        self.G.rbgs(div.numpy(), p, dx, 1, omega)
        if include_ghosts:  # the ghost leaves are swept too (their outer layers go stale until the next exchange)
            p_out.copy_(self.torch.from_numpy(p))
        else:
            self._put(p_out, p)

    def subtract_pressure_gradient(self, u, p, out, inv_dx):
        self._put(out, self.G.subtract_pressure_gradient(self._aos(u), p.numpy(), inv_dx))

    def synchronize(self):
        pass


def _case(name):
    if name == "dense":
        return fields.dense_leaves(32), 32
    o = fields.plume_leaves(8, 1.5, 0.35)
    return o, 64


def _worker(rank, world, port, name, iters, out_dir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        origins, R = _case(name)
        f = fields.synthetic_fields(origins, R)
        plan = HD.make_plan(origins, world, rank)
        loc = np.concatenate([plan.owned_global, plan.ghost_global])
        eng = OracleEngine(plan.local_origins, plan.n_owned, 1.0 / R)
        sol = HD.DistributedSolver(plan, eng, 1.0 / R, n_scalars=2)
        sel = (loc[:, None] * 512 + np.arange(512)[None, :]).reshape(-1)
        vel, den, tem = f["vel"][sel].copy(), f["density"][sel].copy(), f["temperature"][sel].copy()
        # start with WRONG ghost data: the first exchange must repair it
        g0 = plan.n_owned * 512
        vel[g0:] = 7.0
        den[g0:] = -3.0
        tem[g0:] = 5.0
        sol.load_local(vel, [den, tem])
        for _ in range(2):
            sol.core_substep(iters, 1.0 / 24.0)
        no = plan.n_owned * 512
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), owned=plan.owned_global, u=sol.u[:no].numpy(),
                 phi0=sol.phi[0][:no].numpy(), phi1=sol.phi[1][:no].numpy(), p=sol.p[:no].numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name,world", [("dense", 2), ("plume", 3), ("dense", 4)])
def test_partitioned_substep_is_bit_identical_to_single_domain(tmp_path, name, world):
    import torch.multiprocessing as mp

    from oracle_lib import OracleGrid, oracle

    iters = 7  # not a multiple of the 4 sweeps between pressure exchanges
    mp.spawn(_worker, args=(world, _free_port(), name, iters, str(tmp_path)), nprocs=world, join=True)
    origins, R = _case(name)
    f = fields.synthetic_fields(origins, R)
    G = OracleGrid(origins)
    vs, dt = 1.0 / R, 1.0 / 24.0
    u, phi = f["vel"].copy(), [f["density"].copy(), f["temperature"].copy()]
    omega = HD.omega_compute(vs)
    assert omega == float(oracle().orc_omega_compute(vs)) or abs(omega - oracle().orc_omega_compute(vs)) < 3e-7
    inv_dx = float(np.float32(1.0) / np.float32(vs))
    for _ in range(2):
        adv = G.advect_vector(u, dt, inv_dx)
        div = G.divergence(adv, inv_dx)
        p = G.rbgs_iterations(div, float(np.float32(vs)), omega, iters)
        u = G.subtract_pressure_gradient(adv, p, inv_dx)
        phi = G.advect_scalars(u, phi, dt, inv_dx)
    got_owned = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        sel = (z["owned"][:, None] * 512 + np.arange(512)[None, :]).reshape(-1)
        got_owned.append(z["owned"])
        assert np.array_equal(z["u"], u[sel]), f"rank {r} velocity"
        assert np.array_equal(z["p"], p[sel]), f"rank {r} pressure"
        assert np.array_equal(z["phi0"], phi[0][sel]) and np.array_equal(z["phi1"], phi[1][sel]), f"rank {r} scalars"
    assert np.array_equal(np.sort(np.concatenate(got_owned)), np.arange(len(origins)))
