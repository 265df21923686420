"""Leaf-partitioned multi-GPU substep: contiguous ranges of the NanoVDB-ordered leaf list per rank, one layer of ghost
leaves, halo exchange over torch.distributed point-to-point (RCCL on MI355X / xGMI, gloo on CPU in the tests).

The reference is single-GPU; this module is new design. Partitioning rule: the leaf list is ordered
x-major (upper 4096^3 -> lower 128^3 -> leaf 8^3), so equal-count contiguous ranges are x-slabs for box-like domains and
every rank talks to at most a few neighbours. Each rank keeps

    local leaves = [ owned leaves (global order) | ghost leaves grouped by owning rank ]

builds its own index grid over them with ``n_active = n_owned`` (kernels update owned leaves only and read ghosts), and
refreshes ghost payloads with whole-leaf messages: received data lands directly in the ghost range of the field
(ghosts of one peer are contiguous), sent data is gathered by the library's pack kernel.

Exchanges sit exactly where the single-GPU code has a global kernel boundary that a stencil crosses, so owned results
are bit-identical to the single-GPU run:

    exchange(u, phi) [u only on the first substep: the last exchange of a substep already refreshed it]
      -> advect_vector -> exchange(u*) -> divergence -> exchange(div)
      -> iterations x fused red+black sweep, exchange(p) after every 4th sweep
      -> gradient subtraction -> exchange(u) -> advect_scalars

(the fused sweep recomputes the red update of the face-adjacent ghost voxels itself, and the ghost leaves are swept
locally between exchanges: their outer voxel layers go stale two per sweep while the owned leaves only read the two
layers next to them, so four sweeps fit between exchanges -- 13 pressure exchanges per 50 iterations instead of the 100
a colour-by-colour scheme needs). There is no all-reduce: the reference uses a fixed iteration count, no residual norm.

The compute engine is injected: ``HipEngine`` (libhns.so through hnanosolver_amd.device) is the product; the CPU tests
inject an engine built on the oracle to exercise partition + exchange logic under gloo.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

LEAF_VOXELS = 512


# ---------------------------------------------------------------------------------------------------------------
# partition (pure numpy; no communication)
# ---------------------------------------------------------------------------------------------------------------


def _keys(origins: np.ndarray) -> np.ndarray:
    l = (np.asarray(origins, dtype=np.int64) >> 3) + (1 << 20)
    return (l[:, 0] << 42) | (l[:, 1] << 21) | l[:, 2]


def neighbor_ids(origins: np.ndarray) -> np.ndarray:
    """(n_leaves, 27) global leaf index of every 27-neighbour, -1 when absent; entry (dx+1)*9+(dy+1)*3+(dz+1)."""
    o = np.asarray(origins, dtype=np.int64).reshape(-1, 3)
    k = _keys(o)
    order = np.argsort(k, kind="stable")
    ks = k[order]
    out = np.full((len(o), 27), -1, dtype=np.int64)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                q = _keys(o + np.array([dx, dy, dz], dtype=np.int64) * 8)
                pos = np.searchsorted(ks, q)
                pos[pos >= len(ks)] = 0
                hit = ks[pos] == q
                col = (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)
                out[hit, col] = order[pos[hit]]
    return out


@dataclass
class RankPlan:
    rank: int
    world: int
    owned_global: np.ndarray  # global leaf ids owned by this rank (ascending)
    ghost_global: np.ndarray  # global leaf ids of the ghosts, grouped by owner rank, ascending inside a group
    local_origins: np.ndarray  # (n_local, 3) origins of owned + ghost leaves, in local order
    recv_ranges: Dict[int, Tuple[int, int]] = field(default_factory=dict)  # peer -> [start, end) LOCAL leaf range of its ghosts
    send_local: Dict[int, np.ndarray] = field(default_factory=dict)  # peer -> local (owned) leaf ids to send, in the peer's ghost order

    @property
    def n_owned(self) -> int:
        return len(self.owned_global)

    @property
    def n_local(self) -> int:
        return len(self.owned_global) + len(self.ghost_global)

    @property
    def outside_element(self) -> int:
        """Local flat index of GLOBAL element 0 (owned by rank 0, mirrored as a ghost everywhere else)."""
        if len(self.owned_global) and self.owned_global[0] == 0:
            return 0
        k = np.flatnonzero(self.ghost_global == 0)
        return int(self.n_owned + k[0]) * LEAF_VOXELS if len(k) else 0

    @property
    def peers(self) -> List[int]:
        return sorted(set(self.recv_ranges) | set(self.send_local))

    @property
    def mirror_only_peers(self) -> List[int]:
        """Peers whose whole traffic with this rank is the mirror of global leaf 0 (only advect_scalars reads it)."""
        out = []
        for q in self.peers:
            r0, r1 = self.recv_ranges.get(q, (0, 0))
            recv_only_mirror = (r1 - r0 == 0) or (r1 - r0 == 1 and self.ghost_global[r0 - self.n_owned] == 0)
            snd = self.send_local.get(q)
            send_only_mirror = snd is None or (len(snd) == 1 and len(self.owned_global) and self.owned_global[snd[0]] == 0)
            if recv_only_mirror and send_only_mirror:
                out.append(q)
        return out


def partition_bounds(n_leaves: int, world: int) -> np.ndarray:
    return np.array([(n_leaves * r) // world for r in range(world + 1)], dtype=np.int64)


def ghosts_of(owned_mask: np.ndarray, nbr: np.ndarray, layers: int) -> np.ndarray:
    have = owned_mask.copy()
    frontier = np.flatnonzero(owned_mask)
    for _ in range(layers):
        cand = np.unique(nbr[frontier].reshape(-1))
        cand = cand[cand >= 0]
        new = cand[~have[cand]]
        have[new] = True
        frontier = new
    return np.flatnonzero(have & ~owned_mask)


def make_plan(origins: np.ndarray, world: int, rank: int, ghost_layers: int = 1, nbr: Optional[np.ndarray] = None) -> RankPlan:
    """Plan of `rank` for the global leaf list `origins` (any order; equal-count contiguous ranges of that order)."""
    origins = np.asarray(origins, dtype=np.int32).reshape(-1, 3)
    n = len(origins)
    if nbr is None:
        nbr = neighbor_ids(origins)
    bounds = partition_bounds(n, world)
    owner = np.searchsorted(bounds, np.arange(n), side="right") - 1

    def ghosts(r):
        m = owner == r
        g = ghosts_of(m, nbr, ghost_layers)
        if n and owner[0] != r:
            g = np.union1d(g, [0])  # every rank mirrors global leaf 0: advect_scalars reads its element 0 for outside taps
        return g[np.lexsort((g, owner[g]))]  # grouped by owner, ascending id inside

    my_ghosts = ghosts(rank)
    owned = np.arange(bounds[rank], bounds[rank + 1], dtype=np.int64)
    plan = RankPlan(rank, world, owned, my_ghosts, np.ascontiguousarray(origins[np.concatenate([owned, my_ghosts])]))
    pos = len(owned)
    for q in np.unique(owner[my_ghosts]) if len(my_ghosts) else []:
        cnt = int((owner[my_ghosts] == q).sum())
        plan.recv_ranges[int(q)] = (pos, pos + cnt)
        pos += cnt
    for q in range(world):
        if q == rank:
            continue
        gq = ghosts(q)
        mine = gq[owner[gq] == rank]
        if len(mine):
            plan.send_local[q] = (mine - bounds[rank]).astype(np.int32)
    return plan


# ---------------------------------------------------------------------------------------------------------------
# engines
# ---------------------------------------------------------------------------------------------------------------


class HipEngine:
    """The product engine: every operation is a HIP kernel of libhns.so on device tensors (no CPU fallback)."""

    def __init__(self, local_origins: np.ndarray, n_owned: int, voxel_size: float):
        import torch

        from . import api, device

        assert torch.cuda.is_available(), "HipEngine needs a HIP device"
        self.torch, self.D = torch, device
        self.grid = api.create_grid_from_leaves(local_origins, voxel_size)
        self.grid.set_active_leaves(n_owned)
        # same leaves, ghosts updated too: used by the communication-avoiding pressure sweeps
        self.grid_all = api.create_grid_from_leaves(local_origins, voxel_size) if n_owned < len(local_origins) else self.grid
        self.device = torch.device("cuda", torch.cuda.current_device())

    def set_outside_element(self, idx: int):
        self.grid.set_outside_element(idx)

    def zeros(self, n: int):
        return self.torch.zeros(n, dtype=self.torch.float32, device=self.device)

    def from_numpy(self, a: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def ids(self, a: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def pack(self, fld, ids, out, ncomp):
        return self.D.pack_leaves(fld, ids, out, ncomp)

    def advect_vector(self, u, out, dt, inv_dx):
        self.D.advect_vector(self.grid, u, out, dt, inv_dx)

    def advect_scalars(self, u, srcs, dsts, dt, inv_dx):
        self.D.advect_scalars(self.grid, u, srcs, dsts, dt, inv_dx)

    def divergence(self, u, div, inv_dx):
        self.D.divergence(self.grid, u, div, inv_dx)

    def rbgs_iteration(self, div, p_in, p_out, dx, omega, include_ghosts=False):
        self.D.rbgs_iterate(self.grid_all if include_ghosts else self.grid, div, p_in, p_out, dx, omega, 1)

    def subtract_pressure_gradient(self, u, p, out, inv_dx):
        self.D.subtract_pressure_gradient(self.grid, u, p, out, inv_dx)

    def synchronize(self):
        self.torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------------------
# halo exchange
# ---------------------------------------------------------------------------------------------------------------


class HaloExchanger:
    """Whole-leaf ghost refresh of flat per-leaf fields (tensors of n_local*512 floats) with batched isend/irecv."""

    def __init__(self, plan: RankPlan, engine, group=None):
        import torch
        import torch.distributed as dist

        self.plan, self.engine, self.group, self.dist, self.torch = plan, engine, group, dist, torch
        self.send_ids = {q: engine.ids(ids) for q, ids in plan.send_local.items()}
        self._bufs: Dict[Tuple[int, int], Tuple[object, object]] = {}
        self._far = set(plan.mirror_only_peers)

    def _ncomp(self, f) -> int:
        return int(f.numel()) // (self.plan.n_local * LEAF_VOXELS)  # 1 = float field, 3 = Vec3f field

    def _buffers(self, q: int, units: int):
        key = (q, units)
        if key not in self._bufs:
            ns = len(self.plan.send_local.get(q, ()))
            r0, r1 = self.plan.recv_ranges.get(q, (0, 0))
            self._bufs[key] = (self.engine.zeros(max(1, units * ns * LEAF_VOXELS)), self.engine.zeros(max(1, units * (r1 - r0) * LEAF_VOXELS)))
        return self._bufs[key]

    def pack_sends(self, fields: Sequence, mirror: bool = True) -> Dict[int, object]:
        """Gather, per peer, the owned leaves it mirrors (all `fields` back to back) into that peer's send buffer.
        mirror=False skips the peers that only exchange the mirror of global leaf 0."""
        comps = [self._ncomp(f) for f in fields]
        units, out = sum(comps), {}
        for q, ids in self.send_ids.items():
            if not mirror and q in self._far:
                continue
            sb, _ = self._buffers(q, units)
            ns, pos = len(self.plan.send_local[q]), 0
            for f, c in zip(fields, comps):
                self.engine.pack(f, ids, sb[pos:pos + c * ns * LEAF_VOXELS], c)
                pos += c * ns * LEAF_VOXELS
            out[q] = sb[:pos]
        return out

    def recv_targets(self, fields: Sequence, mirror: bool = True) -> Dict[int, object]:
        """Per peer, the tensor its message lands in: the ghost range itself for one field (ghosts of a peer are
        contiguous), a staging buffer for several."""
        comps = [self._ncomp(f) for f in fields]
        units, out = sum(comps), {}
        for q, (r0, r1) in self.plan.recv_ranges.items():
            if not mirror and q in self._far:
                continue
            if len(fields) == 1:
                c = comps[0]
                out[q] = fields[0].view(-1)[r0 * c * LEAF_VOXELS:r1 * c * LEAF_VOXELS]
            else:
                out[q] = self._buffers(q, units)[1][: units * (r1 - r0) * LEAF_VOXELS]
        return out

    def finish(self, fields: Sequence, mirror: bool = True) -> None:
        if len(fields) == 1:
            return
        comps = [self._ncomp(f) for f in fields]
        units = sum(comps)
        for q, (r0, r1) in self.plan.recv_ranges.items():
            if not mirror and q in self._far:
                continue
            rb, nr, pos = self._buffers(q, units)[1], r1 - r0, 0
            for f, c in zip(fields, comps):
                f.view(-1)[r0 * c * LEAF_VOXELS:r1 * c * LEAF_VOXELS].copy_(rb[pos:pos + c * nr * LEAF_VOXELS])
                pos += c * nr * LEAF_VOXELS

    def exchange(self, fields: Sequence, mirror: bool = True) -> None:
        """Refresh the ghost leaves of every tensor in `fields` (one message per peer carrying all of them).
        mirror=False leaves the mirror of global leaf 0 alone where that is a peer's only traffic: only advect_scalars reads
        it, so the pressure / divergence / u* exchanges need not fan out from rank 0 to every rank."""
        if self.plan.world == 1 or not self.plan.peers:
            return
        dist = self.dist
        sends, recvs = self.pack_sends(fields, mirror), self.recv_targets(fields, mirror)
        ops = [dist.P2POp(dist.isend, t, q, self.group) for q, t in sends.items()]
        ops += [dist.P2POp(dist.irecv, t, q, self.group) for q, t in recvs.items()]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        self.finish(fields, mirror)


# ---------------------------------------------------------------------------------------------------------------
# distributed core substep
# ---------------------------------------------------------------------------------------------------------------


def omega_compute(voxel_size: float) -> float:
    """omega = 2/(1+sinf(3.14159f*voxelSize)), float arithmetic (reference HNanoSolver.cu:257)."""
    vs = np.float32(voxel_size)
    return float(np.float32(2.0) / (np.float32(1.0) + np.sin(np.float32(3.14159) * vs, dtype=np.float32)))


class DistributedSolver:
    """Core substep (advect_vector -> divergence -> RB-SOR -> projection -> advect_scalars) on one rank's leaves.

    State tensors are flat over LOCAL leaves (owned then ghosts); velocity is (n_local*512, 3) Vec3f AoS."""

    def __init__(self, plan: RankPlan, engine, voxel_size: float, n_scalars: int = 1, group=None):
        self.plan, self.e, self.vs = plan, engine, float(np.float32(voxel_size))
        self.inv_dx = float(np.float32(1.0) / np.float32(voxel_size))
        n = plan.n_local * LEAF_VOXELS
        self.u = engine.zeros(3 * n).view(-1, 3)
        self.adv = engine.zeros(3 * n).view(-1, 3)
        self.div, self.p_a, self.p_b = engine.zeros(n), engine.zeros(n), engine.zeros(n)
        self.phi = [engine.zeros(n) for _ in range(n_scalars)]
        self.phi_next = [engine.zeros(n) for _ in range(n_scalars)]
        self.p = self.p_a
        self.halo = HaloExchanger(plan, engine, group)
        self._u_ghosts_fresh = False  # whoever writes self.u outside core_substep must reset this
        self.omega = omega_compute(voxel_size)
        engine.set_outside_element(plan.outside_element)

    def load_local(self, vel_aos: np.ndarray, scalars: Sequence[np.ndarray]) -> None:
        """Initial data for the LOCAL leaves (owned + ghosts), velocity as (n_local*512, 3) AoS."""
        self._u_ghosts_fresh = False
        self.u.copy_(self.e.from_numpy(vel_aos).view(-1, 3))
        for k, s in enumerate(scalars):
            self.phi[k].copy_(self.e.from_numpy(s))

    # A fused (red, black) sweep moves information two voxels, and a ghost layer is one leaf = 8 voxels deep. If the
    # ghosts are swept locally as well, after k sweeps without an exchange only their outer 2k voxel layers are stale, and
    # the sweep of the OWNED leaves reads ghosts no deeper than 2 voxels: four sweeps fit between two exchanges
    # (2*3 = 6 stale layers before the fourth sweep, 2 valid ones left). Owned results stay bit-identical.
    SWEEPS_PER_EXCHANGE = 4

    def pressure_solve(self, iterations: int) -> None:
        self.p_a.zero_()  # never warm-started (reference HNanoSolver.cu:113)
        self.p_b.zero_()
        src, dst = self.p_a, self.p_b
        k = max(1, int(self.SWEEPS_PER_EXCHANGE))
        for it in range(iterations):
            last_before_exchange = (it + 1) % k == 0 or it + 1 == iterations
            # the sweep right before an exchange need not touch the ghosts: they are overwritten anyway
            self.e.rbgs_iteration(self.div, src, dst, self.vs, self.omega, include_ghosts=not last_before_exchange)
            if last_before_exchange:
                self.halo.exchange([dst], mirror=False)
            src, dst = dst, src
        self.p = src

    def start_exchange(self) -> None:
        """Ghosts of u and phi before advection (with the mirror of global leaf 0: advect_scalars reads phi's and u's element
        0). The previous substep ended by exchanging u and nothing has written it since: then only phi travels."""
        self.halo.exchange(([] if self._u_ghosts_fresh else [self.u]) + self.phi)
        self._u_ghosts_fresh = False

    def core_substep(self, iterations: int, dt: float) -> None:
        e, h = self.e, self.halo
        self.start_exchange()
        e.advect_vector(self.u, self.adv, dt, self.inv_dx)
        h.exchange([self.adv], mirror=False)
        e.divergence(self.adv, self.div, self.inv_dx)
        h.exchange([self.div], mirror=False)
        self.pressure_solve(iterations)
        e.subtract_pressure_gradient(self.adv, self.p, self.u, self.inv_dx)
        h.exchange([self.u])
        self._u_ghosts_fresh = True
        e.advect_scalars(self.u, self.phi, self.phi_next, dt, self.inv_dx)
        self.phi, self.phi_next = self.phi_next, self.phi

    def owned(self, t):
        return t[: self.plan.n_owned * LEAF_VOXELS]


# ---------------------------------------------------------------------------------------------------------------
# bench.py helper: weak scaling, one slab per rank stacked along x
# ---------------------------------------------------------------------------------------------------------------


class SlabBench:
    """Rank r owns the slab `origins + (r*R, 0, 0)` of a (world*R) x R x R domain; fields are the closed-form synthetic
    inputs evaluated on the global domain. Used by bench.py for --gpus N > 1."""

    def __init__(self, slab_origins: np.ndarray, R: int, rank: int, world: int, iterations: int, dt: float, partition: bool = False):
        import torch

        from . import fields

        self.torch = torch
        slab_origins = np.asarray(slab_origins, dtype=np.int32)
        if partition:
            # strong scaling: ONE domain (e.g. the 1024^3-extent plume of BASELINE.json configs[4]) split into `world`
            # contiguous leaf ranges of its NanoVDB order
            self.plan = make_plan(slab_origins, world, rank)
            eval_origins = self.plan.local_origins
        else:
            glob = np.concatenate([slab_origins + np.array([r * R, 0, 0], dtype=np.int32) for r in range(world)])
            n_slab = len(slab_origins)
            # only the leaves near this rank's slab matter for its plan: restrict the neighbour search to slabs r-1..r+1
            lo, hi = max(0, rank - 1), min(world, rank + 2)
            sub = glob[lo * n_slab:hi * n_slab]
            nbr_sub = neighbor_ids(sub)
            nbr = np.full((len(glob), 27), -1, dtype=np.int64)
            nbr[lo * n_slab:hi * n_slab] = np.where(nbr_sub >= 0, nbr_sub + lo * n_slab, -1)
            self.plan = make_plan_slabs(glob, n_slab, world, rank, nbr)
            eval_origins = self.plan.local_origins.copy()
            eval_origins[:, 0] %= R  # evaluate the closed-form fields periodically in x: every slab carries the same plume
        self.iterations, self.dt = iterations, dt
        self.vs = 1.0 / R  # same voxel size (and omega) as the single-GPU workload
        self.engine = HipEngine(self.plan.local_origins, self.plan.n_owned, self.vs)
        self.solver = DistributedSolver(self.plan, self.engine, self.vs, n_scalars=1)
        f = fields.synthetic_fields(eval_origins, R)
        self.solver.load_local(f["vel"], [f["density"]])
        self._ev = []
        self._timing = False
        self._launches = 0

    def step(self):
        s, torch = self.solver, self.torch
        if not self._timing:
            s.core_substep(self.iterations, self.dt)
            return
        # same as DistributedSolver.core_substep with the pressure loop bracketed by events on the launch stream
        e, h = s.e, s.halo
        s.start_exchange()
        e.advect_vector(s.u, s.adv, self.dt, s.inv_dx)
        h.exchange([s.adv], mirror=False)
        e.divergence(s.adv, s.div, s.inv_dx)
        h.exchange([s.div], mirror=False)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        s.pressure_solve(self.iterations)
        b.record()
        self._ev.append((a, b))
        self._launches += self.iterations
        e.subtract_pressure_gradient(s.adv, s.p, s.u, s.inv_dx)
        h.exchange([s.u])
        s._u_ghosts_fresh = True
        e.advect_scalars(s.u, s.phi, s.phi_next, self.dt, s.inv_dx)
        s.phi, s.phi_next = s.phi_next, s.phi

    def timing_on(self):
        self._timing, self._ev, self._launches = True, [], 0

    def pressure_time(self):
        """(ms spent in the event-bracketed pressure loops INCLUDING the per-iteration halo exchange, iterations timed)"""
        self.torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._ev), self._launches


def make_plan_slabs(glob: np.ndarray, n_slab: int, world: int, rank: int, nbr: np.ndarray) -> RankPlan:
    """make_plan for the slab layout where rank r owns exactly glob[r*n_slab:(r+1)*n_slab] and `nbr` is only filled near
    this rank (far slabs cannot be neighbours)."""
    n = len(glob)
    owner = np.arange(n) // n_slab

    def ghosts(r):
        m = owner == r
        g = ghosts_of(m, nbr, 1)
        if r != 0:
            g = np.union1d(g, [0])  # mirror of global leaf 0 (see make_plan)
        return g[np.lexsort((g, owner[g]))]

    my_ghosts = ghosts(rank)
    owned = np.arange(rank * n_slab, (rank + 1) * n_slab, dtype=np.int64)
    plan = RankPlan(rank, world, owned, my_ghosts, np.ascontiguousarray(glob[np.concatenate([owned, my_ghosts])]))
    pos = len(owned)
    for q in np.unique(owner[my_ghosts]) if len(my_ghosts) else []:
        cnt = int((owner[my_ghosts] == q).sum())
        plan.recv_ranges[int(q)] = (pos, pos + cnt)
        pos += cnt
    for q in sorted(set([rank - 1, rank + 1] + (list(range(1, world)) if rank == 0 else []))):
        if 0 <= q < world and q != rank:
            gq = ghosts(q)
            mine = gq[owner[gq] == rank]
            if len(mine):
                plan.send_local[q] = (mine - rank * n_slab).astype(np.int32)
    return plan
