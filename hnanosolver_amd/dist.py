"""Leaf-partitioned multi-GPU core substep: host-side mirror of the ``hns_dist_*`` entry points of libhns.so.

The reference is single-GPU; the decomposition is new design and lives in ``csrc/hns_dist_*.hip``: rank r
owns the r-th of `world` equal ranges of the global leaf list in slab order (``DistRank.owned_ids``; box domains: x-slabs = contiguous
ranges of the NanoVDB-ordered list, as in rounds 1-4; the plume of BASELINE config 5: slabs along its own axis, two halo peers per rank), keeps
one layer of ghost leaves in the local order ``[boundary | interior | ghosts]``, runs every kernel on its boundary leaves
first and ships exactly the ghost voxels the peers' next kernel can read (512-bit masks per leaf, derived on both sides
from the global leaf list) over RCCL point-to-point on a communication stream while the interior is being computed.
Everything here is plumbing: plan queries, the RCCL bootstrap (the 128-byte unique id travels over torch.distributed),
host-array upload/download, and ``SlabBench``, the multi-GPU driver of bench.py. There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import lib

LEAF_VOXELS = 512
REGION_TYPES = ("advection inputs", "reach 1", "div", "p")  # hns_dist_stats arrays, hns_dist_peer_region `type`


def _raise(code: int) -> None:
    _lib.check(code)


def omega_compute(voxel_size: float) -> float:
    """omega = 2/(1+sinf(3.14159f*voxelSize)), float arithmetic (reference HNanoSolver.cu:257)."""
    vs = np.float32(voxel_size)
    return float(np.float32(2.0) / (np.float32(1.0) + np.sin(np.float32(3.14159) * vs, dtype=np.float32)))


def partition_bounds(n_leaves: int, world: int) -> np.ndarray:
    """First position of every rank (and n_leaves at the end) in the PARTITION order of the leaves: the rule hns_dist_create applies. Positions
    in the caller's list only when the partition is contiguous ranges of it (DistRank.partition_axis == -1); see owned_ids_of."""
    return np.array([(n_leaves * r) // world for r in range(world + 1)], dtype=np.int64)


def owned_ids_of(global_origins: np.ndarray, world: int, rank: int, leaf_order: bool = False) -> np.ndarray:
    """The leaves (positions in `global_origins`) rank `rank` of `world` owns, in upload / download order: asked of the library (a plan-only handle)."""
    d = DistRank(global_origins, world, rank, 1.0, 0, 0, plan_only=True, leaf_order=leaf_order)
    ids = d.owned_ids.copy()
    d.close()
    return ids


def take_leaves(a: np.ndarray, leaf_ids: np.ndarray) -> np.ndarray:
    """rows of a per-voxel array (n_leaves * 512 [, c]) that belong to the listed leaves, in that order"""
    a = np.asarray(a)
    return np.ascontiguousarray(a.reshape((-1, LEAF_VOXELS) + a.shape[1:])[np.asarray(leaf_ids, dtype=np.int64)].reshape((-1,) + a.shape[1:]))


@dataclass
class Region:
    leaves: np.ndarray  # local leaf ids
    masks: np.ndarray  # (n, 64) uint8: byte x*8+y, bit z
    voxels: int

    def voxel_index(self) -> np.ndarray:
        """flat local voxel indices (leaf*512 + x<<6|y<<3|z) in message order"""
        bits = np.unpackbits(self.masks.reshape(-1, 64, 1), axis=2, bitorder="little").reshape(-1, 512).astype(bool)
        leaf, vox = np.nonzero(bits)
        return self.leaves[leaf].astype(np.int64) * LEAF_VOXELS + vox


@dataclass
class PeerPlan:
    rank: int
    send: List[Region]
    recv: List[Region]


class DistRank:
    """One rank of the decomposition (``hns_dist``). ``plan_only=True`` builds the host-side plan without a device."""

    def __init__(self, global_origins: np.ndarray, world: int, rank: int, voxel_size: float, n_scalars: int = 1, sweeps_per_exchange: int = 0,
                 plan_only: bool = False, leaf_order: bool = False):
        o = np.ascontiguousarray(global_origins, dtype=np.int32).reshape(-1, 3)
        err = C.c_int(0)
        self._ptr = lib.hns_dist_create(o.ctypes.data, o.shape[0], int(world), int(rank), float(voxel_size), int(n_scalars), int(sweeps_per_exchange),
                                        (_lib.HNS_DIST_PLAN_ONLY if plan_only else 0) | (_lib.HNS_DIST_LEAF_ORDER if leaf_order else 0), C.byref(err))
        if not self._ptr:
            _raise(err.value if err.value < 0 else _lib.HNS_ERR_RUNTIME)
        self.world, self.rank, self.n_scalars, self.voxel_size = int(world), int(rank), int(n_scalars), float(voxel_size)
        self.n_global = o.shape[0]
        self.n_owned = int(lib.hns_dist_owned_leaves(self._ptr))
        # the owned leaves (positions in `global_origins`) in the order upload / download use. Slabs along `partition_axis` (0 / 1 / 2), or --
        # partition_axis -1 -- the contiguous range [first_owned, first_owned + n_owned) of the caller's list (hns.h: hns_dist_partition_axis)
        self.owned_ids = np.zeros(self.n_owned, dtype=np.int64)
        if self.n_owned:
            _raise(lib.hns_dist_owned_leaf_ids(self._ptr, self.owned_ids.ctypes.data))
        self.partition_axis = int(lib.hns_dist_partition_axis(self._ptr))
        self.first_owned = int(lib.hns_dist_first_owned_leaf(self._ptr)) if self.partition_axis < 0 else -1

    def owned_voxels(self, a: np.ndarray) -> np.ndarray:
        """rows of a per-voxel global array (n_global * 512 [, c]) that belong to this rank's leaves, in upload / download order"""
        a = np.asarray(a)
        return np.ascontiguousarray(a.reshape((self.n_global, LEAF_VOXELS) + a.shape[1:])[self.owned_ids].reshape((-1,) + a.shape[1:]))

    # ---- plan ----
    def info(self) -> Dict:
        s = _lib.hns_dist_stats()
        _raise(lib.hns_dist_info(self._ptr, C.byref(s)))
        return {"world": s.world, "rank": s.rank, "peers": s.peers, "halo_peers": int(s.halo_peers), "sweeps_per_exchange": s.sweeps_per_exchange, "boundary_leaves": int(s.boundary_leaves),
                "interior_leaves": int(s.interior_leaves), "ghost_leaves": int(s.ghost_leaves),
                "region_voxels_sent": dict(zip(REGION_TYPES, [int(x) for x in s.region_voxels_sent])),
                "bytes_sent": dict(zip(REGION_TYPES, [int(x) for x in s.bytes_sent])), "messages_sent": int(s.messages_sent), "exchanges": int(s.exchanges), "packed_exchanges": int(s.packed_exchanges), "chained": int(s.chained)}

    def local_leaves(self) -> np.ndarray:
        i = self.info()
        out = np.zeros(i["boundary_leaves"] + i["interior_leaves"] + i["ghost_leaves"], dtype=np.int64)
        _raise(lib.hns_dist_local_leaves(self._ptr, out.ctypes.data))
        return out

    def peers(self) -> List[PeerPlan]:
        out = []
        for p in range(self.info()["peers"]):
            regs = {0: [], 1: []}
            for is_send in (1, 0):
                for t in range(4):
                    n, v = C.c_uint64(0), C.c_uint64(0)
                    _raise(lib.hns_dist_peer_region(self._ptr, p, t, is_send, None, None, C.byref(n), C.byref(v)))
                    leaves = np.zeros(n.value, dtype=np.int32)
                    masks = np.zeros((n.value, 64), dtype=np.uint8)
                    _raise(lib.hns_dist_peer_region(self._ptr, p, t, is_send, leaves.ctypes.data, masks.ctypes.data, None, None))
                    regs[is_send].append(Region(leaves, masks, int(v.value)))
            out.append(PeerPlan(int(lib.hns_dist_peer_rank(self._ptr, p)), regs[1], regs[0]))
        return out

    # ---- transports ----
    def connect_rccl(self, group=None) -> None:
        """Collective over torch.distributed (any backend): rank 0's RCCL unique id is broadcast, every rank joins."""
        import torch
        import torch.distributed as dist

        buf = (C.c_ubyte * 128)()
        if self.rank == 0:
            _raise(lib.hns_dist_unique_id(buf))
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor(list(buf), dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0, group=group)
        raw = bytes(t.cpu().tolist())
        _raise(lib.hns_dist_connect_rccl(self._ptr, C.create_string_buffer(raw, 128)))

    def connect_ipc(self, group=None) -> None:
        """Collective over torch.distributed (any backend): every rank exports the handles of its field memory, message
        buffers and flag page, all ranks gather them and map their peers' (hns_dist_connect_ipc). One process per rank.
        A failure on ANY rank raises on EVERY rank (the ranks agree after each step: none is left waiting in a collective)."""
        ok, reason = self.try_connect_ipc(group)
        if not ok:
            raise RuntimeError(f"hns_dist: the ipc transport could not be set up on every rank ({reason})")

    def try_connect_ipc(self, group=None):
        """connect_ipc for callers that can fall back to another transport: every step that can fail on one rank alone is
        followed by an agreement among all ranks, so that either every rank ends up connected or none (-> (False, reason))."""
        import torch
        import torch.distributed as dist

        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

        def agreed(ok: bool) -> bool:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        n = _lib.HNS_DIST_IPC_BLOB_BYTES
        mine = (C.c_ubyte * n)()
        reason = ""
        try:
            _raise(lib.hns_dist_ipc_export(self._ptr, mine))
            ok = True
        except Exception as e:  # noqa: BLE001
            ok, reason = False, f"export: {e}"
        if not agreed(ok):
            return False, reason or "a peer could not export its memory"
        t = torch.frombuffer(bytearray(bytes(mine)), dtype=torch.uint8).to(dev)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=group)
        blobs = b"".join(bytes(o.cpu().numpy().tobytes()) for o in out)
        try:
            _raise(lib.hns_dist_connect_ipc(self._ptr, C.create_string_buffer(blobs, n * self.world)))
            ok = True
        except Exception as e:  # noqa: BLE001
            ok, reason = False, f"connect: {e}"
        if not agreed(ok):
            return False, reason or "a peer could not map this rank's memory"
        return True, ""

    @staticmethod
    def connect_local(ranks: Sequence["DistRank"]) -> None:
        arr = (C.c_void_p * len(ranks))(*[r._ptr for r in ranks])
        _raise(lib.hns_dist_connect_local(arr, len(ranks)))

    def connect_loopback(self, rccl: bool = False) -> None:
        """TIMING ONLY: this rank alone, messages answered with its own payload (hns_dist_connect_loopback); rccl=True carries
        them through a one-rank RCCL communicator (hns_dist_connect_loopback_rccl)."""
        _raise((lib.hns_dist_connect_loopback_rccl if rccl else lib.hns_dist_connect_loopback)(self._ptr))

    # ---- data ----
    def upload(self, vel: np.ndarray, scalars: Sequence[np.ndarray], stream: int = 0) -> None:
        """Host arrays over the OWNED leaves in the order of `owned_ids`."""
        vel = np.ascontiguousarray(vel, dtype=np.float32)
        sc = [np.ascontiguousarray(s, dtype=np.float32) for s in scalars]
        assert vel.size == self.n_owned * LEAF_VOXELS * 3 and len(sc) == self.n_scalars and all(s.size == self.n_owned * LEAF_VOXELS for s in sc)
        ptrs = (C.c_void_p * max(1, len(sc)))(*[s.ctypes.data for s in sc])
        _raise(lib.hns_dist_upload(self._ptr, vel.ctypes.data, ptrs, stream))

    def download(self, pressure: bool = False, stream: int = 0) -> Dict[str, np.ndarray]:
        n = self.n_owned * LEAF_VOXELS
        vel = np.empty((n, 3), dtype=np.float32)
        sc = [np.empty(n, dtype=np.float32) for _ in range(self.n_scalars)]
        p = np.empty(n, dtype=np.float32) if pressure else None
        ptrs = (C.c_void_p * max(1, len(sc)))(*[s.ctypes.data for s in sc])
        _raise(lib.hns_dist_download(self._ptr, vel.ctypes.data, ptrs, p.ctypes.data if pressure else None, stream))
        out = {"vel": vel, "scalars": sc}
        if pressure:
            out["pressure"] = p
        return out

    def download_local(self, which: int, stream: int = 0) -> np.ndarray:
        """Field `which` (-2: the last solve's p, -1: velocity, s >= 0: scalar s) over ALL local leaves, ghosts included, in
        local order (hns_dist_download_local): diagnostics."""
        n = len(self.local_leaves()) * LEAF_VOXELS
        out = np.empty((n, 3) if which == -1 else (n,), dtype=np.float32)
        _raise(lib.hns_dist_download_local(self._ptr, int(which), out.ctypes.data, stream))
        return out

    # region type a field's ghost voxels are refreshed with by the END of a substep: the velocity travels as whole leaves
    # ("advection inputs") behind the gradient subtraction, p within reach 1 of the owned voxels behind the last sweep
    GHOST_REGION = {-1: 0, -2: 1}

    def ghost_digests(self, fields: Sequence[int] = (-1, -2), stream: int = 0) -> Dict:
        """{(owner rank, ghost-holder rank, field): SHA-1 of the values in message order} for every region this rank sends (its
        own boundary voxels) and receives (its ghost voxels). Owner and ghost holder enumerate a region in the same order, so the
        two digests of a pair are equal exactly when the ghost copy holds the owner's bits. Valid between substeps without
        collision (with collision the next substep rewrites u before it exchanges it)."""
        import hashlib

        out = {}
        peers = self.peers()
        for f in fields:
            a = self.download_local(f, stream)
            t = self.GHOST_REGION[f]
            for pp in peers:
                for is_send, reg in ((True, pp.send[t]), (False, pp.recv[t])):
                    if reg.voxels == 0:
                        continue
                    key = (self.rank, pp.rank, f) if is_send else (pp.rank, self.rank, f)
                    out[key + ("owner" if is_send else "ghost",)] = hashlib.sha1(np.ascontiguousarray(a[reg.voxel_index()]).tobytes()).hexdigest()
        return out

    @staticmethod
    def compare_ghost_digests(all_digests: Sequence[Dict]) -> Tuple[int, List]:
        """(pairs compared, [(owner, holder, field), ...] whose ghost copy differs from the owner's values or is missing)"""
        merged = {}
        for d in all_digests:
            merged.update(d)
        pairs = sorted({k[:3] for k in merged})
        bad = [k for k in pairs if merged.get(k + ("owner",)) is None or merged.get(k + ("owner",)) != merged.get(k + ("ghost",))]
        return len(pairs), bad

    def ghost_check(self, group=None, fields: Sequence[int] = (-1, -2), stream: int = 0) -> Tuple[int, List]:
        """Collective over torch.distributed: after a substep, is every ghost voxel of the velocity and of p (the regions the next
        kernels read) bit-equal to its owner's value? -> (pairs compared, mismatching pairs). A data-path check of the transport
        that was actually used, on the memory it actually wrote -- independent of any reference run."""
        import torch.distributed as dist

        mine = self.ghost_digests(fields, stream)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, mine, group=group)
        return self.compare_ghost_digests(gathered)

    @staticmethod
    def ghost_check_local(ranks: Sequence["DistRank"], fields: Sequence[int] = (-1, -2), stream: int = 0) -> Tuple[int, List]:
        """ghost_check for locally connected ranks (one process)"""
        return DistRank.compare_ghost_digests([r.ghost_digests(fields, stream) for r in ranks])

    # ---- stepping ----
    def core_substep(self, iterations: int, dt: float, stream: int = 0) -> None:
        _raise(lib.hns_dist_core_substep(self._ptr, int(iterations), float(dt), stream))

    @staticmethod
    def local_core_substep(ranks: Sequence["DistRank"], iterations: int, dt: float, stream: int = 0) -> None:
        arr = (C.c_void_p * len(ranks))(*[r._ptr for r in ranks])
        _raise(lib.hns_dist_local_core_substep(arr, len(ranks), int(iterations), float(dt), stream))

    @staticmethod
    def _field_index(names: Sequence[str]):
        """positions of fuel, waste, temperature, flame, collision_sdf (-1: absent) in a rank's scalar list"""
        return (C.c_int * 5)(*[list(names).index(n) if n in names else -1 for n in ("fuel", "waste", "temperature", "flame", "collision_sdf")])

    def sim_substep(self, names: Sequence[str], iterations: int, dt: float, params, has_collision: bool = False, stream: int = 0) -> None:
        """The whole Compute_Sim substep on this rank (hns_dist_sim_substep); `names` = the scalars in upload order."""
        p = params._c()
        _raise(lib.hns_dist_sim_substep(self._ptr, int(iterations), float(dt), C.byref(p), self._field_index(names), int(has_collision), stream))

    @staticmethod
    def local_sim_substep(ranks: Sequence["DistRank"], names: Sequence[str], iterations: int, dt: float, params, has_collision: bool = False, stream: int = 0) -> None:
        arr = (C.c_void_p * len(ranks))(*[r._ptr for r in ranks])
        p = params._c()
        _raise(lib.hns_dist_local_sim_substep(arr, len(ranks), int(iterations), float(dt), C.byref(p), DistRank._field_index(names), int(has_collision), stream))

    def timing(self, max_solves: int) -> None:
        _raise(lib.hns_dist_timing(self._ptr, int(max_solves)))

    def pressure_time(self):
        """(ms inside the event-bracketed pressure loops INCLUDING their halo exchanges, fused sweeps timed)"""
        ms, n = C.c_float(0.0), C.c_longlong(0)
        _raise(lib.hns_dist_pressure_time(self._ptr, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def synchronize(self, stream: int = 0) -> None:
        _raise(lib.hns_dist_synchronize(self._ptr, stream))

    def close(self) -> None:
        if getattr(self, "_ptr", None):
            lib.hns_dist_destroy(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------
# bench.py helper
# ---------------------------------------------------------------------------------------------------------------


def slab_domain(slab_origins: np.ndarray, R: int, world: int) -> np.ndarray:
    """`world` copies of a slab stacked along x: the weak-scaling domain, rank r owning slab r."""
    slab_origins = np.asarray(slab_origins, dtype=np.int32)
    return np.ascontiguousarray(np.concatenate([slab_origins + np.array([r * R, 0, 0], dtype=np.int32) for r in range(world)]))


class SlabBench:
    """bench.py --gpus N > 1. Weak scaling: rank r owns the slab `origins + (r*R, 0, 0)` of a (world*R) x R x R domain, fields
    = the closed-form synthetic inputs evaluated periodically in x (every slab carries the same plume). partition=True:
    ONE domain (e.g. BASELINE.json configs[4], the 1024^3-extent plume) split into `world` contiguous leaf ranges."""

    def __init__(self, slab_origins: np.ndarray, R: int, rank: int, world: int, iterations: int, dt: float, partition: bool = False,
                 sweeps_per_exchange: int = 0, connect: bool = True, transport: str = "rccl", reference_transport: str = "rccl"):
        import torch

        from . import fields

        self.torch = torch
        slab_origins = np.ascontiguousarray(slab_origins, dtype=np.int32)
        glob = slab_origins if partition else slab_domain(slab_origins, R, world)
        self.iterations, self.dt, self.vs = iterations, dt, 1.0 / R  # same voxel size (and omega) as the single-GPU workload
        # a stream of the run's own, not the legacy null stream: since round 6 small ranks issue their RCCL groups on the stream they step on (in-line exchanges), and RCCL on
        # the null stream -- with its implicit synchronisation against every blocking stream of the process -- is a combination nobody else exercises
        self._torch_stream = torch.cuda.Stream() if world > 1 else None
        self.stream = int((self._torch_stream or torch.cuda.current_stream()).cuda_stream)
        self.transport_note = "no peers" if world == 1 else transport

        def make(k):
            return DistRank(glob, world, rank, self.vs, n_scalars=1, sweeps_per_exchange=k)

        plan = DistRank(glob, world, rank, self.vs, n_scalars=1, plan_only=True)  # (which leaves are this rank's: the library's rule, not restated here)
        self._owned_ids = plan.owned_ids.copy()
        plan.close()
        own = glob[self._owned_ids].copy()
        if not partition:
            own[:, 0] %= R
        f = fields.synthetic_fields(own, R)
        self._fields = (f["vel"], [f["density"]])
        self._glob, self._world = glob, world
        if world > 1 and connect and transport == "auto":
            self.rank_obj = self._verified_one_sided(make, sweeps_per_exchange, reference_transport)
        else:
            # the one-sided transport runs the pressure loop whose sweep kernel delivers its own halo
            self.rank_obj = make(sweeps_per_exchange or (self._one_sided_k(len(glob), world) if transport == "ipc" else 0))
            if world > 1 and connect:
                self.rank_obj.connect_ipc() if transport == "ipc" else self.rank_obj.connect_rccl()
        self.rank_obj.upload(*self._fields, stream=self.stream)
        self._glob, self._R, self._partition, self._world = glob, R, partition, world
        self.verified_note = "single GPU" if world == 1 else "not checked"

    @staticmethod
    def _one_sided_k(n_leaves: int, world: int) -> int:
        """sweeps_per_exchange of the chained one-sided substep: 2 = the temporally blocked sweep, two iterations per chained launch, where the library
        sweeps every rank's owned range in 16^3 blocks; else 1. The library's own rule (hns_dist_one_sided_sweeps), not a copy of it."""
        return int(lib.hns_dist_one_sided_sweeps(int(n_leaves), int(world)))

    def verify_against_single_gpu(self, substeps: int = 2, max_voxels: int = 600_000_000) -> bool:
        """Every rank computes `substeps` substeps of the WHOLE domain on its own GPU with the single-GPU path (the same on every rank, bit
        for bit) and compares its owned leaves of the partitioned run with it: velocity and density must be identical. The
        transport that is about to be timed is what runs the partitioned side, so a scaling figure comes with a statement about its
        physics. Collective (the partitioned substeps are). Leaves the rank with freshly uploaded fields."""
        from . import api, device as D, fields

        if self._world == 1:
            return True
        n_vox = len(self._glob) * LEAF_VOXELS
        if n_vox > max_voxels:
            self.verified_note = f"not checked against the single-GPU result (whole domain of {n_vox} voxels kept off one GPU)"
            return True
        d = self.rank_obj
        why = ""
        try:  # (whatever goes wrong on one rank -- the partitioned substeps included -- every rank reaches the collective below)
            d.upload(*self._fields, stream=self.stream)
            for _ in range(substeps):
                d.core_substep(self.iterations, self.dt, self.stream)
            d.synchronize(self.stream)
            got = d.download(stream=self.stream)
            if self._partition:
                f = fields.synthetic_fields(self._glob, self._R)
                vel, den = f["vel"], f["density"]
            else:  # every slab carries the same fields (periodic in x)
                o = self._glob[: len(self._glob) // self._world].copy()
                f = fields.synthetic_fields(o, self._R)
                vel, den = np.tile(f["vel"], (self._world, 1)), np.tile(f["density"], self._world)
            grid = api.create_grid_from_leaves(self._glob, self.vs)
            sim = D.Sim(grid, ["density"])
            arrays = {"vel": np.ascontiguousarray(vel), "density": np.ascontiguousarray(den)}
            sim.upload(arrays, self.stream)
            for _ in range(substeps):
                sim.core_substep(self.iterations, self.dt, self.vs, self.stream)
            sim.download(arrays, self.stream)
            same = np.array_equal(got["vel"], d.owned_voxels(arrays["vel"])) and np.array_equal(got["scalars"][0], d.owned_voxels(arrays["density"]))
            sim.close()
        except Exception as e:  # noqa: BLE001
            same, why = False, f" [the check failed on a rank: {type(e).__name__}: {e}]"[:200]
        import torch.distributed as dist

        t = self.torch.tensor([1 if same else 0, 0 if why else 1], dtype=self.torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok, ran = bool(int(t[0].item())), bool(int(t[1].item()))
        if not ran:
            self.verified_note = "NOT checked: the partitioned or the single-GPU run of the check failed on a rank" + why
        else:
            self.verified_note = (f"owned velocity and density after {substeps} substeps bit-identical to the single-GPU run of the whole domain on every rank" if ok else
                                  "MISMATCH against the single-GPU run of the whole domain: THIS RUN'S PHYSICS IS WRONG, its throughput means nothing")
        d.upload(*self._fields, stream=self.stream)
        return ok

    def _verified_one_sided(self, make, sweeps_per_exchange, reference_transport):
        """transport = auto: the one-sided transport with the chained substep (every kernel delivers its own halo) if -- on THIS machine, now -- it connects
        and three substeps of it leave bit for bit what three substeps over the reference transport (RCCL) leave on every rank;
        RCCL otherwise. Every decision is taken by all ranks together."""
        import torch
        import torch.distributed as dist

        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

        def agreed(ok: bool) -> bool:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        def two_substeps(d):  # (three, at the full iteration count: ~150 sweeps over every boundary leaf)
            d.upload(*self._fields, stream=self.stream)
            for _ in range(3):
                d.core_substep(self.iterations, self.dt, self.stream)
            d.synchronize(self.stream)
            return d.download(pressure=True, stream=self.stream)

        ref = make(sweeps_per_exchange)
        ref.connect_rccl() if reference_transport == "rccl" else ref.connect_ipc()
        want = two_substeps(ref)
        cand = make(self._one_sided_k(len(self._glob), self._world))
        ok, why = cand.try_connect_ipc()
        if ok and not cand.info()["chained"]:  # (ranks of one-leaf SOR blocks: the one-sided transport would run the same exchanged substep as the reference)
            ok, why = False, "ranks of 600 leaves and fewer run the exchanged substep over either transport"
        if ok:
            try:
                got = two_substeps(cand)
                same = all(np.array_equal(got[k], want[k]) for k in ("vel", "pressure")) and np.array_equal(got["scalars"][0], want["scalars"][0])
                ok, why = same, "" if same else "results differ from the reference transport's"
            except Exception as e:  # noqa: BLE001
                ok, why = False, f"substep: {e}"
        ok = agreed(ok)
        dist.barrier()  # nobody unmaps or frees while a peer may still be writing
        if ok:
            ref.close()
            self.transport_note = f"every kernel stores its boundary values into the peers' mapped ghost voxels itself, no exchanges; verified bit for bit against {reference_transport} at start-up"
            return cand
        cand.close()
        self.transport_note = f"{reference_transport} (one-sided transport not used: {why or 'a peer rank failed its check'})"
        return ref

    @property
    def n_owned(self) -> int:
        return self.rank_obj.n_owned

    def step(self):
        self.rank_obj.core_substep(self.iterations, self.dt, self.stream)

    def timing_on(self, max_solves: int = 64):
        self.rank_obj.timing(max_solves)

    def pressure_time(self):
        self.rank_obj.synchronize(self.stream)
        return self.rank_obj.pressure_time()

    def info(self):
        return self.rank_obj.info()
