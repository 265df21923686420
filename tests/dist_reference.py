"""TEST INFRASTRUCTURE: a CPU walk through the native multi-GPU plan.

`ReferenceRank` executes the phases of hns_dist_*.hip's core substep (same order, same launch ranges [boundary | interior |
ghosts], same halo regions -- taken from the library's own plan, DistRank(plan_only=True)) with the ORACLE as the compute
engine and torch.distributed (gloo) as the wire. It exists to prove, without a GPU, that the plan's regions and exchange
points are sufficient: owned results must be bit-identical to the single-domain oracle run. The HIP path is tied to the
same single-domain answer by tests/test_dist_gpu.py.
"""
from __future__ import annotations

import numpy as np

from hnanosolver_amd import dist as HD

X_ADV, X_D1, X_DIV, X_P = 0, 1, 2, 3


class ReferenceRank:
    def __init__(self, global_origins, world, rank, voxel_size, n_scalars, k, group=None, poison=None):
        from oracle_lib import OracleGrid

        self.plan = HD.DistRank(global_origins, world, rank, voxel_size, n_scalars, k, plan_only=True)
        self.world, self.rank, self.k, self.group = world, rank, self.plan.info()["sweeps_per_exchange"], group
        info = self.plan.info()
        self.nB, self.nI, self.nG = info["boundary_leaves"], info["interior_leaves"], info["ghost_leaves"]
        self.local_global = self.plan.local_leaves()
        self.peers = self.plan.peers()
        lo = np.ascontiguousarray(np.asarray(global_origins, dtype=np.int32).reshape(-1, 3)[self.local_global])
        self.G = OracleGrid(lo)
        z = np.flatnonzero(self.local_global == 0)
        self.G.set_outside_element(int(z[0]) * 512 if len(z) else 0)
        n = len(self.local_global) * 512
        fill = 0.0 if poison is None else poison  # poison: ghosts start with garbage that the exchanges must repair
        self.u = np.full((n, 3), fill, dtype=np.float32)
        self.adv = np.zeros((n, 3), dtype=np.float32)
        self.div, self.p_a, self.p_b = (np.zeros(n, dtype=np.float32) for _ in range(3))
        self.phi = [np.full(n, fill, dtype=np.float32) for _ in range(n_scalars)]
        self.phi_next = [np.zeros(n, dtype=np.float32) for _ in range(n_scalars)]
        self.vs = float(np.float32(voxel_size))
        self.inv_dx = float(np.float32(1.0) / np.float32(voxel_size))
        self.omega = HD.omega_compute(voxel_size)
        self.pending = None
        self.phi_in_flight = self.u_fresh = False
        self.bytes_sent = 0
        self.ranges = {"B": (0, self.nB), "I": (self.nB, self.nB + self.nI), "A": (0, len(self.local_global))}

    def load_owned(self, vel, scalars):
        """owned leaves in upload order (DistRank.owned_ids) -> local order [B | I]"""
        where = {int(g): i for i, g in enumerate(self.plan.owned_ids.tolist())}
        for l in range(self.nB + self.nI):
            src = where[int(self.local_global[l])]
            self.u[l * 512:(l + 1) * 512] = vel[src * 512:(src + 1) * 512]
            for s, a in zip(self.phi, scalars):
                s[l * 512:(l + 1) * 512] = a[src * 512:(src + 1) * 512]

    def owned(self, a):
        """local [B | I] -> download order (DistRank.owned_ids)"""
        n = self.nB + self.nI
        local_of = {int(g): l for l, g in enumerate(self.local_global[:n].tolist())}
        order = [local_of[int(g)] for g in self.plan.owned_ids.tolist()]
        return np.concatenate([a[l * 512:(l + 1) * 512] for l in order]) if n else a[:0]

    def _put(self, dst, full, rng):
        a, b = self.ranges[rng]
        dst[a * 512:b * 512] = full[a * 512:b * 512]

    # ---- exchange ----
    def post(self, typ, fields):
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return
        assert self.pending is None
        reqs, recvs = [], []
        for p in self.peers:
            s, r = p.send[typ], p.recv[typ]
            comps = sum(f.shape[1] if f.ndim == 2 else 1 for f in fields)
            if s.voxels:
                idx = s.voxel_index()
                msg = torch.from_numpy(np.concatenate([np.ascontiguousarray(f.reshape(len(f), -1)[idx]).reshape(-1) for f in fields]))
                self.bytes_sent += msg.numel() * 4
                reqs.append(dist.isend(msg, p.rank, group=self.group))
            if r.voxels:
                buf = torch.empty(r.voxels * comps, dtype=torch.float32)
                reqs.append(dist.irecv(buf, p.rank, group=self.group))
                recvs.append((r, buf))
        self.pending = (reqs, recvs, fields)

    def complete(self):
        if self.pending is None:
            return
        reqs, recvs, fields = self.pending
        for q in reqs:
            q.wait()
        for r, buf in recvs:
            idx, pos = r.voxel_index(), 0
            b = buf.numpy()
            for f in fields:
                c = f.shape[1] if f.ndim == 2 else 1
                f.reshape(len(f), -1)[idx] = b[pos:pos + c * r.voxels].reshape(r.voxels, c)
                pos += c * r.voxels
        self.pending = None

    # ---- kernels on a launch range ----
    def sweep(self, src, dst, rng):
        p = src.copy()
        self.G.rbgs(self.div, p, self.vs, 0, self.omega)
        self.G.rbgs(self.div, p, self.vs, 1, self.omega)
        self._put(dst, p, rng)

    def core_substep(self, iterations, dt):
        G, k = self.G, self.k
        if not self.phi_in_flight:
            self.post(X_ADV, ([] if self.u_fresh else [self.u]) + self.phi)
        self.complete()
        adv = G.advect_vector(self.u, dt, self.inv_dx)
        self._put(self.adv, adv, "B")
        self.post(X_D1, [self.adv])
        self._put(self.adv, adv, "I")
        self.complete()
        div = G.divergence(self.adv, self.inv_dx)
        self._put(self.div, div, "B")
        self.post(X_DIV, [self.div])
        self._put(self.div, div, "I")
        self.complete()
        self.p_a[:] = 0.0
        src, dst, it = self.p_a, self.p_b, 0
        while it < iterations:
            n = min(k, iterations - it)
            for _ in range(n - 1):
                self.sweep(src, dst, "A")
                src, dst = dst, src
                it += 1
            last = it + 1 == iterations
            p = src.copy()
            G.rbgs(self.div, p, self.vs, 0, self.omega)
            G.rbgs(self.div, p, self.vs, 1, self.omega)
            self._put(dst, p, "B")
            self.post(X_D1 if last else X_P, [dst])
            self._put(dst, p, "I")
            src, dst = dst, src
            it += 1
            self.complete()
        self.p = src
        u = G.subtract_pressure_gradient(self.adv, self.p, self.inv_dx)
        self._put(self.u, u, "B")
        self.post(X_ADV, [self.u])
        self._put(self.u, u, "I")
        self.complete()
        self.u_fresh = True
        if self.phi:
            out = G.advect_scalars(self.u, self.phi, dt, self.inv_dx)
            for d, o in zip(self.phi_next, out):
                self._put(d, o, "B")
            self.post(X_ADV, self.phi_next)
            for d, o in zip(self.phi_next, out):
                self._put(d, o, "I")
            self.phi, self.phi_next = self.phi_next, self.phi
            self.phi_in_flight = self.world > 1

    # ---- the whole Compute_Sim substep (hns_dist_substep.hip: Step::run_full; reference HNanoSolver.cu:150-356) ----
    def sim_substep(self, names, iterations, dt, params, has_collision):
        """`names`: the scalars in load order. Same phases, exchange types and launch ranges as hns_dist_sim_substep."""
        G, k = self.G, self.k
        ci = [names.index(n) for n in ("fuel", "waste", "temperature", "flame")]
        i_sdf = names.index("collision_sdf") if "collision_sdf" in names else -1
        coll = bool(has_collision) and i_sdf >= 0
        sdf = self.phi[i_sdf] if coll else None
        vort = int(params.factorScale) != 0
        nO = self.nB + self.nI
        self.ranges["O"] = (0, nO)
        # open: phi (collision_sdf among it) unless already posted, u unless collision is about to rewrite it
        f = []
        if not coll and not self.u_fresh:
            self.complete()
            f.append(self.u)
        if not self.phi_in_flight:
            f += self.phi
        self.phi_in_flight = False
        if f:
            self.post(X_ADV, f)
        self.complete()
        if coll:  # the SDF's ghost voxels are there now (the collision normal reads them)
            self._put(self.u, G.enforce_collision_boundaries(self.u, sdf, self.vs), "O")
            self.post(X_ADV, [self.u])
            self.complete()
        adv = G.advect_vector(self.u, dt, self.inv_dx, sdf, coll)
        self._put(self.adv, adv, "B")
        self.post(X_ADV if vort else X_D1, [self.adv])
        self._put(self.adv, adv, "I")
        self.complete()
        if vort:
            tmp = G.vorticity_confinement(self.adv, dt, self.inv_dx, params.vorticityScale, params.factorScale)
            new = self.adv.copy()
            self._put(new, tmp, "B")
            self._put(new, tmp, "I")
            self.adv = new
            self.post(X_D1, [self.adv])
            self.complete()
        div = G.divergence(self.adv, self.inv_dx)
        fu, wa, te, fl, div2 = G.combustion_oxygen(self.phi[ci[0]], self.phi[ci[1]], self.phi[ci[2]], div, self.phi[ci[3]], params.temperatureRelease, params.expansionRate)
        self._put(self.div, div2, "B")
        self.post(X_DIV, [self.div])
        self._put(self.div, div2, "I")
        self.complete()
        self._put(self.adv, G.temperature_buoyancy(self.adv, te, dt, params.ambientTemp, params.buoyancyStrength), "O")
        for c, new in zip(ci, (fu, wa, te, fl)):
            self._put(self.phi_next[c], new, "O")
            self.phi[c], self.phi_next[c] = self.phi_next[c], self.phi[c]
        self.post(X_ADV, [self.phi[c] for c in ci])
        self.complete()
        self.p_a[:] = 0.0
        src, dst, it = self.p_a, self.p_b, 0
        while it < iterations:
            n = min(k, iterations - it)
            for _ in range(n - 1):
                self.sweep(src, dst, "A")
                src, dst = dst, src
                it += 1
            last = it + 1 == iterations
            p = src.copy()
            G.rbgs(self.div, p, self.vs, 0, self.omega)
            G.rbgs(self.div, p, self.vs, 1, self.omega)
            self._put(dst, p, "B")
            self.post(X_D1 if last else X_P, [dst])
            self._put(dst, p, "I")
            src, dst = dst, src
            it += 1
            self.complete()
        self.p = src
        u = G.subtract_pressure_gradient(self.adv, self.p, self.inv_dx, sdf, coll)
        if coll:
            u = G.enforce_collision_boundaries(u, sdf, self.vs)  # (pointwise in u: enforcing the full array and keeping the owned part = enforcing the owned part)
        self._put(self.u, u, "B")
        self.post(X_ADV, [self.u])
        self._put(self.u, u, "I")
        self.complete()
        self.u_fresh = not coll
        which = [i for i in range(len(self.phi)) if i != i_sdf]
        out = G.advect_scalars(self.u, [self.phi[i] for i in which], dt, self.inv_dx, sdf, coll)
        for i, o in zip(which, out):
            self._put(self.phi_next[i], o, "B")
        self.post(X_ADV, [self.phi_next[i] for i in which])
        for i, o in zip(which, out):
            self._put(self.phi_next[i], o, "I")
        for i in which:
            self.phi[i], self.phi_next[i] = self.phi_next[i], self.phi[i]
        self.phi_in_flight = self.world > 1
