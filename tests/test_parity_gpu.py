"""GPU parity: every HIP kernel and every drop-in operator of libhns.so against the CPU oracle on the same seeded inputs.

Integer/index work must be bit-exact. Floating point: the north star's bar is 1e-5 relative L-infinity per field
(``TOL``). Both sides keep the reference's operation order and are built with -ffp-contract=off, so in practice the
HIP results are bit-identical to the oracle; ``EXACT`` asserts that wherever no transcendental is involved.
"""
import numpy as np
import pytest

from hnanosolver_amd import fields

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: within 1e-5 relative L-inf (float32 advection/pressure)


def rel_linf(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() / denom


def assert_close(got, want, what, exact=True):
    r = rel_linf(got, want)
    assert r <= TOL, f"{what}: rel L-inf {r:.3e} > {TOL}"
    if exact:
        assert np.array_equal(np.asarray(got), np.asarray(want)), f"{what}: within tolerance (rel {r:.3e}) but not bit-identical to the oracle"


# ---------------------------------------------------------------------------------------------------------------
# cases
# ---------------------------------------------------------------------------------------------------------------


def sparse_leaves(seed=3):
    """~20 leaves straddling the origin (negative coordinates), with isolated leaves and missing neighbours."""
    rng = np.random.default_rng(seed)
    lat = np.stack(np.meshgrid(*[np.arange(-2, 2)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = rng.random(len(lat)) < 0.3
    o = (lat[keep] * 8).astype(np.int32)
    o = np.concatenate([o, np.array([[-4104, 0, 0], [4096, 8, -16]], dtype=np.int32)])  # far-away tiles
    return np.ascontiguousarray(o[fields.nanovdb_order(o)])


CASES = {
    "dense16": lambda: (fields.dense_leaves(16), 16),
    "dense32": lambda: (fields.dense_leaves(32), 32),
    "sparse": lambda: (sparse_leaves(), 32),
    "plume_small": lambda: (fields.plume_leaves(8, 1.0, 0.3), 64),
}


class Case:
    def __init__(self, name, amplitude=96.0, noise=0.0, seed=0):
        import torch

        from oracle_lib import OracleGrid

        self.origins, self.R = CASES[name]()
        self.vs = 1.0 / self.R
        self.inv_dx = float(np.float32(1.0) / np.float32(self.vs))
        self.dt = float(np.float32(1.0 / 24.0))
        self.oracle = OracleGrid(self.origins)
        self.N = self.oracle.N
        f = fields.synthetic_fields(self.origins, self.R, amplitude_voxels=amplitude)
        if noise:
            rng = np.random.default_rng(seed)
            for k in f:
                f[k] = (f[k] + noise * rng.standard_normal(f[k].shape) * max(1e-3, np.abs(f[k]).max())).astype(np.float32)
        self.f = f
        from hnanosolver_amd import api

        self.grid = api.create_grid_from_leaves(self.origins, self.vs)
        self.torch = torch

    def dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def device_velocity(self, vel_aos):  # velocity lives on the device as Vec3f AoS, the host layout
        return self.dev(vel_aos)

    @staticmethod
    def aos(t):
        return t.cpu().numpy().reshape(-1, 3)


@pytest.fixture(scope="module", params=list(CASES.keys()))
def case(request):
    return Case(request.param, noise=0.05)


# ---------------------------------------------------------------------------------------------------------------
# topology
# ---------------------------------------------------------------------------------------------------------------


def test_offsets_match_oracle(case):
    rng = np.random.default_rng(1)
    c = case.oracle.coords()
    ijk = np.concatenate([c[rng.integers(0, len(c), 4000)] + rng.integers(-9, 10, (4000, 3)), rng.integers(-5000, 5000, (1000, 3))]).astype(np.int32)
    assert np.array_equal(case.grid.offsets(ijk), case.oracle.offsets(ijk))
    assert np.array_equal(case.grid.coords(), c)


# ---------------------------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------------------------


def test_divergence(case):
    from hnanosolver_amd import device as D

    u = case.device_velocity(case.f["vel"])
    div = case.torch.zeros(case.N, device="cuda")
    D.divergence(case.grid, u, div, case.inv_dx)
    assert_close(div.cpu().numpy(), case.oracle.divergence(case.f["vel"], case.inv_dx), "divergence")


def test_rbgs_color_and_fused(case):
    from hnanosolver_amd import device as D
    from oracle_lib import oracle

    div_h = case.oracle.divergence(case.f["vel"], case.inv_dx)
    omega = float(oracle().orc_omega_compute(case.vs))
    div = case.dev(div_h)
    # two-launch form, starting from a non-zero pressure so that every tap matters
    p0 = (0.01 * np.random.default_rng(5).standard_normal(case.N)).astype(np.float32)
    p = case.dev(p0)
    want = p0.copy()
    for it in range(3):
        for color in (0, 1):
            D.rbgs_color(case.grid, div, p, case.vs, omega, color)
            case.oracle.rbgs(div_h, want, case.vs, color, omega)
            assert_close(p.cpu().numpy(), want, f"rbgs_color it{it} c{color}")
    # fused form: 1, 2, 7 iterations from the same non-zero start
    for iters in (1, 2, 7):
        pa, pb = case.dev(p0), case.torch.zeros(case.N, device="cuda")
        res = D.rbgs_iterate(case.grid, div, pa, pb, case.vs, omega, iters)
        assert_close(res.cpu().numpy(), case.oracle.rbgs_iterations(div_h, case.vs, omega, iters, p0), f"rbgs_fused x{iters}")


def test_subtract_pressure_gradient(case):
    from hnanosolver_amd import device as D

    p_h = (np.random.default_rng(6).standard_normal(case.N)).astype(np.float32)
    u = case.device_velocity(case.f["vel"])
    out = case.torch.zeros_like(u)
    D.subtract_pressure_gradient(case.grid, u, case.dev(p_h), out, case.inv_dx)
    assert_close(Case.aos(out), case.oracle.subtract_pressure_gradient(case.f["vel"], p_h, case.inv_dx), "subtract_pressure_gradient")
    # in place (PressureProjection.cu:64)
    D.subtract_pressure_gradient(case.grid, u, case.dev(p_h), u, case.inv_dx)
    assert np.array_equal(u.cpu().numpy(), out.cpu().numpy())


def test_advect_vector(case):
    from hnanosolver_amd import device as D

    u = case.device_velocity(case.f["vel"])
    out = case.torch.zeros_like(u)
    D.advect_vector(case.grid, u, out, case.dt, case.inv_dx)
    assert_close(Case.aos(out), case.oracle.advect_vector(case.f["vel"], case.dt, case.inv_dx), "advect_vector")


def test_advect_scalar(case):
    from hnanosolver_amd import device as D

    u = case.device_velocity(case.f["vel"])
    out = case.torch.zeros(case.N, device="cuda")
    D.advect_scalar(case.grid, u, case.dev(case.f["density"]), out, case.dt, case.inv_dx)
    assert_close(out.cpu().numpy(), case.oracle.advect_scalar(case.f["vel"], case.f["density"], case.dt, case.inv_dx), "advect_scalar")


@pytest.mark.parametrize("S", [1, 5, 11])
def test_advect_scalars(case, S):
    from hnanosolver_amd import device as D

    names = ["density", "temperature", "fuel", "waste", "flame"]
    rng = np.random.default_rng(7)
    phis = [case.f[names[i % 5]] if i < 5 else rng.standard_normal(case.N).astype(np.float32) for i in range(S)]
    phis = [p.copy() for p in phis]
    phis[0][0] = 100.0  # pins the "out-of-domain taps read element 0" behaviour (Kernel.cu:133,192,225)
    u = case.device_velocity(case.f["vel"])
    outs = [case.torch.zeros(case.N, device="cuda") for _ in range(S)]
    D.advect_scalars(case.grid, u, [case.dev(p) for p in phis], outs, case.dt, case.inv_dx)
    want = case.oracle.advect_scalars(case.f["vel"], phis, case.dt, case.inv_dx)
    for s in range(S):
        assert_close(outs[s].cpu().numpy(), want[s], f"advect_scalars[{s}/{S}]")


def test_long_backtrace_uses_hash():
    """|u| dt/dx ~ 21 voxels: taps leave the 27-leaf neighbourhood and go through the origin hash."""
    from hnanosolver_amd import device as D

    c = Case("dense32", amplitude=400.0)
    u = c.device_velocity(c.f["vel"])
    out = c.torch.zeros_like(u)
    D.advect_vector(c.grid, u, out, c.dt, c.inv_dx)
    assert_close(Case.aos(out), c.oracle.advect_vector(c.f["vel"], c.dt, c.inv_dx), "advect_vector long")
    o = c.torch.zeros(c.N, device="cuda")
    D.advect_scalar(c.grid, u, c.dev(c.f["density"]), o, c.dt, c.inv_dx)
    assert_close(o.cpu().numpy(), c.oracle.advect_scalar(c.f["vel"], c.f["density"], c.dt, c.inv_dx), "advect_scalar long")


def test_combustion_and_buoyancy(case):
    from hnanosolver_amd import device as D

    t = case.torch
    f = case.f
    waste = (0.5 * np.abs(np.random.default_rng(8).standard_normal(case.N))).astype(np.float32)  # some voxels end with oxygen < 0
    div_h = case.oracle.divergence(f["vel"], case.inv_dx)
    outs = [t.zeros(case.N, device="cuda") for _ in range(4)]
    div = case.dev(div_h)
    D.combustion_oxygen(case.dev(f["fuel"]), case.dev(waste), case.dev(f["temperature"]), div, case.dev(f["flame"]), *outs, 0.5, 0.1)
    want = case.oracle.combustion_oxygen(f["fuel"], waste, f["temperature"], div_h, f["flame"], 0.5, 0.1)
    for got, w, name in zip(outs + [div], want, ["fuel", "waste", "temperature", "flame", "divergence"]):
        assert_close(got.cpu().numpy(), w, f"combustion {name}")
    u = case.device_velocity(f["vel"])
    D.temperature_buoyancy(u, case.dev(f["temperature"]), u, case.dt, 23.0, 1.0)
    assert_close(Case.aos(u), case.oracle.temperature_buoyancy(f["vel"], f["temperature"], case.dt, 23.0, 1.0), "buoyancy")


@pytest.mark.parametrize("factor_scale", [0.5, 1.0, 2.0])
def test_vorticity_confinement(case, factor_scale):
    from hnanosolver_amd import device as D

    u = case.device_velocity(case.f["vel"])
    out = case.torch.zeros_like(u)
    D.vorticity_confinement(case.grid, u, out, case.dt, case.inv_dx, 1.0, factor_scale)
    assert_close(Case.aos(out), case.oracle.vorticity_confinement(case.f["vel"], case.dt, case.inv_dx, 1.0, factor_scale), f"vorticity fs={factor_scale}")


def test_collision_paths(case):
    from hnanosolver_amd import device as D

    sdf_h = fields.sphere_sdf(case.origins, case.R, center=(0.5, 0.3, 0.5), radius=0.2)
    # make sure the thin 0 <= sdf < 0.1 band is populated
    sdf_h[::7] = np.float32(0.05)
    sdf = case.dev(sdf_h)
    f = case.f
    u = case.device_velocity(f["vel"])
    D.enforce_collision_boundaries(case.grid, u, sdf, case.vs)
    assert_close(Case.aos(u), case.oracle.enforce_collision_boundaries(f["vel"], sdf_h, case.vs), "enforce_collision")
    u = case.device_velocity(f["vel"])
    out = case.torch.zeros_like(u)
    D.advect_vector(case.grid, u, out, case.dt, case.inv_dx, sdf, True)
    assert_close(Case.aos(out), case.oracle.advect_vector(f["vel"], case.dt, case.inv_dx, sdf_h, True), "advect_vector+collision")
    o = case.torch.zeros(case.N, device="cuda")
    D.advect_scalar(case.grid, u, case.dev(f["density"]), o, case.dt, case.inv_dx, sdf, True)
    assert_close(o.cpu().numpy(), case.oracle.advect_scalar(f["vel"], f["density"], case.dt, case.inv_dx, sdf_h, True), "advect_scalar+collision")
    outs = [case.torch.zeros(case.N, device="cuda") for _ in range(2)]
    D.advect_scalars(case.grid, u, [case.dev(f["density"]), case.dev(f["fuel"])], outs, case.dt, case.inv_dx, sdf, True)
    want = case.oracle.advect_scalars(f["vel"], [f["density"], f["fuel"]], case.dt, case.inv_dx, sdf_h, True)
    for s in range(2):
        assert_close(outs[s].cpu().numpy(), want[s], f"advect_scalars+collision[{s}]")
    p_h = np.random.default_rng(9).standard_normal(case.N).astype(np.float32)
    D.subtract_pressure_gradient(case.grid, u, case.dev(p_h), out, case.inv_dx, sdf, True)
    assert_close(Case.aos(out), case.oracle.subtract_pressure_gradient(f["vel"], p_h, case.inv_dx, sdf_h, True), "gradient+collision")


def test_pack_unpack_leaves(case):
    from hnanosolver_amd import device as D

    t = case.torch
    nl = len(case.origins)
    ids_h = np.random.default_rng(10).permutation(nl)[: max(1, nl // 2)].astype(np.int32)
    ids = t.from_numpy(ids_h).cuda()
    src = case.dev(case.f["density"])
    packed = t.zeros(len(ids_h) * 512, device="cuda")
    D.pack_leaves(src, ids, packed)
    want = case.f["density"].reshape(nl, 512)[ids_h].reshape(-1)
    assert np.array_equal(packed.cpu().numpy(), want)
    dst = t.zeros(case.N, device="cuda")
    D.unpack_leaves(packed, ids, dst)
    ref = np.zeros((nl, 512), np.float32)
    ref[ids_h] = case.f["density"].reshape(nl, 512)[ids_h]
    assert np.array_equal(dst.cpu().numpy(), ref.reshape(-1))
    # Vec3f payloads: 1536 floats per leaf
    v = case.dev(case.f["vel"])
    pv = t.zeros(len(ids_h) * 1536, device="cuda")
    D.pack_leaves(v, ids, pv, 3)
    assert np.array_equal(pv.cpu().numpy(), case.f["vel"].reshape(nl, 1536)[ids_h].reshape(-1))
    dv = t.zeros((case.N, 3), device="cuda")
    D.unpack_leaves(pv, ids, dv, 3)
    refv = np.zeros((nl, 1536), np.float32)
    refv[ids_h] = case.f["vel"].reshape(nl, 1536)[ids_h]
    assert np.array_equal(dv.cpu().numpy().reshape(-1), refv.reshape(-1))
