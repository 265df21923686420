"""TEST HELPER: the OracleGrid call signatures (tests/oracle_lib.py) on top of the HIP kernels, through the C ABI
(hns_dev_* via hnanosolver_amd.device; the operators via hnanosolver_amd.api). Lets one test body run against the oracle and
against the product."""
import numpy as np


class HipKernels:
    def __init__(self, leaves, voxel_size=1.0 / 32.0):
        import torch

        from hnanosolver_amd import api, device

        self.t, self.D, self.api = torch, device, api
        self.origins = np.ascontiguousarray(leaves, dtype=np.int32)
        self.grid = api.create_grid_from_leaves(self.origins, voxel_size)

    def _d(self, a):
        return None if a is None else self.t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

    def _h(self, t):
        return t.cpu().numpy()

    def advect_vector(self, vel, dt, inv_dx, sdf=None, has_collision=False):
        u = self._d(vel)
        out = self.t.empty_like(u)
        self.D.advect_vector(self.grid, u, out, dt, inv_dx, self._d(sdf), has_collision)
        return self._h(out)

    def advect_scalar(self, vel, phi, dt, inv_dx, sdf=None, has_collision=False):
        p = self._d(phi)
        out = self.t.empty_like(p)
        self.D.advect_scalar(self.grid, self._d(vel), p, out, dt, inv_dx, self._d(sdf), has_collision)
        return self._h(out)

    def advect_scalars(self, vel, phis, dt, inv_dx, sdf=None, has_collision=False):
        src = [self._d(p) for p in phis]
        dst = [self.t.empty_like(p) for p in src]
        self.D.advect_scalars(self.grid, self._d(vel), src, dst, dt, inv_dx, self._d(sdf), has_collision)
        return [self._h(d) for d in dst]

    def divergence(self, vel, inv_dx):
        out = self.t.empty(len(vel), device="cuda")
        self.D.divergence(self.grid, self._d(vel), out, inv_dx)
        return self._h(out)

    def rbgs(self, div, p, dx, color, omega):
        q = self._d(p)
        self.D.rbgs_color(self.grid, self._d(div), q, dx, omega, color)
        return self._h(q)

    def rbgs_iterations(self, div, dx, omega, iterations, p0=None):
        """the production path: fused (red, black) launches"""
        n = len(div)
        a = self.t.zeros(n, device="cuda") if p0 is None else self._d(p0)
        b = self.t.zeros(n, device="cuda")
        return self._h(self.D.rbgs_iterate(self.grid, self._d(div), a, b, dx, omega, iterations))

    def subtract_pressure_gradient(self, vel, p, inv_dx, sdf=None, has_collision=False):
        u = self._d(vel)
        out = self.t.empty_like(u)
        self.D.subtract_pressure_gradient(self.grid, u, self._d(p), out, inv_dx, self._d(sdf), has_collision)
        return self._h(out)

    def combustion_oxygen(self, fuel, waste, temperature, div, flame, temp_gain, expansion):
        d = self._d(div)
        outs = [self.t.empty(len(fuel), device="cuda") for _ in range(4)]
        self.D.combustion_oxygen(self._d(fuel), self._d(waste), self._d(temperature), d, self._d(flame), *outs, temp_gain, expansion)
        return tuple(self._h(o) for o in outs) + (self._h(d),)

    def temperature_buoyancy(self, vel, temp, dt, ambient, strength):
        u = self._d(vel)
        out = self.t.empty_like(u)
        self.D.temperature_buoyancy(u, self._d(temp), out, dt, ambient, strength)
        return self._h(out)

    def vorticity_confinement(self, vel, dt, inv_dx, scale, factor_scale):
        u = self._d(vel)
        out = self.t.empty_like(u)
        self.D.vorticity_confinement(self.grid, u, out, dt, inv_dx, scale, factor_scale)
        return self._h(out)

    def enforce_collision_boundaries(self, vel, sdf, voxel_size):
        u = self._d(vel)
        self.D.enforce_collision_boundaries(self.grid, u, self._d(sdf), voxel_size)
        return self._h(u)

    # ---- host drivers through the drop-in operators (results in place, like the reference) ----
    def _data(self, vel, fields_: dict):
        from hnanosolver_amd import fields as F

        d = self.api.GridIndexedData()
        c = F.leaves_to_coords(self.origins)
        d.allocateCoords(len(c))
        d.pCoords()[:] = c
        for n, a in fields_.items():
            d.addValueBlock(n, d.FLOAT)
            d.pValues(n)[:] = a
        d.addValueBlock("vel", d.VEC3F)
        d.pValues("vel")[:] = vel
        return d

    def compute_sim(self, vel, fields_: dict, iterations, dt, voxel_size, params, has_collision):
        d = self._data(vel, fields_)
        h = self.api.IndexGridHandle()
        self.api.CreateIndexGrid(d, h, voxel_size)
        self.api.Compute_Sim(d, h, iterations, dt, voxel_size, params, has_collision)
        h.reset()
        for n in fields_:
            fields_[n][:] = d.pValues(n)
        vel[:] = d.pValues("vel")
        return 0

    def project_non_divergent(self, vel, iterations, voxel_size):
        d = self._data(vel, {})
        self.api.ProjectNonDivergent(d, iterations, voxel_size)
        vel[:] = d.pValues("vel")
        return 0
