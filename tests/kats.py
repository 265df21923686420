"""Closed-form known-answer tests (KATs) for the kernel bodies of the substep hot path.

TEST INFRASTRUCTURE. Every expected value below is derived from the reference's formulas (reference src/Cuda/Kernel.cu,
line numbers at each case) by plain numpy on a dense zero-padded box -- nothing here calls, imports or restates
oracle/hns_oracle.c, so a transcription slip shared by the oracle and the HIP kernels cannot hide in it. The inputs are
small integers and dyadic fractions, chosen so that every intermediate of the reference's float32 expression is exactly
representable: the expected result is then THE mathematically exact value, independent of association, contraction and
FMA use (it pins which taps are read, their order and weights, the clamp set, the branch structure and the out-of-domain
rules -- not the rounding, which the bit-exact oracle-vs-HIP tests pin).

`K` in the cases below is any object with the OracleGrid call signatures (tests/oracle_lib.py); test_kats.py runs the
cases against the oracle (CPU suite) and against the HIP kernels through the C ABI (-m gpu).
"""
from __future__ import annotations

import numpy as np

from hnanosolver_amd import fields

F32 = np.float32


# ---------------------------------------------------------------------------------------------------------------
# dense box <-> flat leaf layout
# ---------------------------------------------------------------------------------------------------------------


class Box:
    """A dense B^3 box of voxels at `origin` (multiple of 8), as leaves in NanoVDB order."""

    def __init__(self, B: int = 24, origin=(0, 0, 0)):
        assert B % 8 == 0
        self.B = B
        self.origin = np.asarray(origin, dtype=np.int32)
        self.leaves = fields.dense_leaves(B) + self.origin
        self.leaves = np.ascontiguousarray(self.leaves[fields.nanovdb_order(self.leaves)], dtype=np.int32)
        c = fields.leaves_to_coords(self.leaves) - self.origin
        self.flat_of = (c[:, 0], c[:, 1], c[:, 2])  # dense index of every flat element
        self.N = len(c)

    def to_flat(self, dense: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(dense[self.flat_of], dtype=F32)

    def to_dense(self, flat: np.ndarray) -> np.ndarray:
        shape = (self.B,) * 3 + tuple(flat.shape[1:])
        out = np.zeros(shape, dtype=flat.dtype)
        out[self.flat_of] = flat
        return out

    def ijk(self):
        """local integer coordinates (i, j, k) as float64 dense arrays"""
        a = np.arange(self.B, dtype=np.float64)
        return np.meshgrid(a, a, a, indexing="ij")


def shift(a: np.ndarray, axis: int, d: int, fill=0.0) -> np.ndarray:
    """value at (.. + d ..) along `axis`, `fill` outside the box (IndexSampler<T,0>: 0 outside, Stencils.hpp:83,88)"""
    out = np.full_like(a, fill)
    src = [slice(None)] * a.ndim
    dst = [slice(None)] * a.ndim
    if d > 0:
        src[axis], dst[axis] = slice(d, None), slice(None, -d)
    elif d < 0:
        src[axis], dst[axis] = slice(None, d), slice(-d, None)
    else:
        return a.copy()
    out[tuple(dst)] = a[tuple(src)]
    return out


def exact32(a: np.ndarray) -> np.ndarray:
    """float64 -> float32, asserting that nothing is lost (the premise of every case here)"""
    b = a.astype(F32)
    assert np.array_equal(b.astype(np.float64), a), "KAT premise violated: expected value is not exactly representable in float32"
    return b


# ---------------------------------------------------------------------------------------------------------------
# divergence (Kernel.cu:499-519): (xp - xm + yp - ym + zp - zm) * inv_dx, xp = (c.x + u(+x).x) * 0.5f, ...
# ---------------------------------------------------------------------------------------------------------------


def kat_divergence_linear(K, box: Box):
    i, j, k = box.ijk()
    a, c, e = 3.0, -2.0, 5.0
    u = np.stack([a * i + 1.0, c * j - 4.0, e * k + 2.0 + 0.0 * i], axis=-1)
    # off-diagonal terms must not leak in: add shear that a correct divergence ignores
    u[..., 0] += 2.0 * j
    u[..., 1] += -3.0 * k
    u[..., 2] += 1.0 * i
    inv_dx = 32.0
    got = box.to_dense(K.divergence(box.to_flat(u), inv_dx))
    want = np.zeros_like(i)
    for ax in range(3):
        comp = u[..., ax]
        want = want + (comp + shift(comp, ax, +1)) * 0.5 - (comp + shift(comp, ax, -1)) * 0.5
    want = exact32(want * inv_dx)
    assert np.array_equal(got, want)
    inner = (slice(1, -1),) * 3
    assert np.all(got[inner] == F32((a + c + e) * inv_dx))  # the closed form: trace of the velocity gradient


# ---------------------------------------------------------------------------------------------------------------
# subtractPressureGradient (Kernel.cu:765-829, no collision): u - ((p(+) - p(-)) * 0.5f) * inv_dx per axis
# ---------------------------------------------------------------------------------------------------------------


def kat_gradient_linear(K, box: Box):
    i, j, k = box.ijk()
    a, b, c = 2.0, -6.0, 4.0
    p = a * i + b * j + c * k + 7.0
    u = np.stack([5.0 + 0 * i, -3.0 + 0 * i, 1.0 + j], axis=-1)
    inv_dx = 16.0
    got = box.to_dense(K.subtract_pressure_gradient(box.to_flat(u), box.to_flat(p), inv_dx))
    want = u.copy()
    for ax in range(3):
        want[..., ax] = u[..., ax] - ((shift(p, ax, +1) - shift(p, ax, -1)) * 0.5) * inv_dx
    assert np.array_equal(got, exact32(want))
    inner = (slice(1, -1),) * 3
    for ax, g in enumerate((a, b, c)):
        assert np.array_equal(got[inner][..., ax], exact32(u[inner][..., ax] - g * inv_dx))  # closed form: u - grad p


# ---------------------------------------------------------------------------------------------------------------
# redBlackGaussSeidelUpdate (Kernel.cu:591-623): pGS = ((sum of 6) - div * dx^2) * 0.166666667f; p += omega * (pGS - p)
# ---------------------------------------------------------------------------------------------------------------


def rbgs_numpy(p: np.ndarray, div: np.ndarray, dx: float, omega: float, color: int) -> np.ndarray:
    """One colour in place on a dense zero-padded box: float32 throughout, one numpy operation per float operation of
    Kernel.cu:621-622, the six-term sum in the reference's order ((((pxp + pxm) + pyp) + pym) + pzp) + pzm. Global parity of a
    voxel is its local parity: box origins are multiples of 8 (Kernel.cu:597-599: (i + j + k) & 1 != color -> skip)."""
    i, j, k = np.meshgrid(*[np.arange(n) for n in p.shape], indexing="ij")
    mask = ((i + j + k) & 1) == color
    p = p.astype(F32)
    taps = [shift(p, 0, +1), shift(p, 0, -1), shift(p, 1, +1), shift(p, 1, -1), shift(p, 2, +1), shift(p, 2, -1)]
    s = taps[0]
    for t in taps[1:]:
        s = (s + t).astype(F32)
    dx2 = F32(F32(dx) * F32(dx))
    pgs = ((s - (div.astype(F32) * dx2).astype(F32)).astype(F32) * F32(0.166666667)).astype(F32)
    new = (p + (F32(omega) * (pgs - p).astype(F32)).astype(F32)).astype(F32)
    return np.where(mask, new, p)


def kat_rbgs_harmonic_fixed_point(K, box: Box):
    """A discrete-harmonic p (linear in i,j,k) with div = 0: every interior voxel is a fixed point of both colours, because
    6p * 0.166666667f rounds back to p for every float p (6 * 0.166666667f = 1 + 2^-25)."""
    i, j, k = box.ijk()
    p = 3.0 * i - 2.0 * j + 5.0 * k + 11.0
    div = np.zeros_like(p)
    dx, omega = 1.0 / 32.0, 1.75
    flat = box.to_flat(p)
    for color in (0, 1):
        got = box.to_dense(K.rbgs(box.to_flat(div), flat.copy(), dx, color, omega))
        inner = (slice(1, -1),) * 3
        assert np.array_equal(got[inner], exact32(p)[inner])
        assert np.array_equal(got, rbgs_numpy(p, div, dx, omega, color))  # and the boundary voxels, which see p = 0 outside


def kat_rbgs_known_sweep(K, box: Box):
    """Two full iterations from random data against the numpy transcription above (written from Kernel.cu:621-622, not from
    the oracle). The first colour of the first iteration sees integers only, so its result is association-free; later
    half-sweeps depend on the reference's summation order, which rbgs_numpy spells out."""
    rng = np.random.default_rng(11)
    B = box.B
    p = rng.integers(-40, 41, size=(B, B, B)).astype(np.float64)
    div = rng.integers(-3, 4, size=(B, B, B)).astype(np.float64) * 1024.0  # div * dx^2 = integer
    dx, omega = 1.0 / 32.0, 1.5
    got = box.to_flat(p)
    want = p.astype(F32)
    for _ in range(2):
        for color in (0, 1):
            got = K.rbgs(box.to_flat(div), got, dx, color, omega)
            want = rbgs_numpy(want, div, dx, omega, color)
    assert np.array_equal(box.to_dense(got), want)


# ---------------------------------------------------------------------------------------------------------------
# BFECC advection (Kernel.cu:269-453 and :118-266) on half-integer displacement fields
# ---------------------------------------------------------------------------------------------------------------


def trilinear_exact(f: np.ndarray, x, y, z, outside) -> np.ndarray:
    """Exact (float64) trilinear sample of dense `f` (…,[C]) at positions (x, y, z); corners outside the box read `outside`
    (0 for IndexSampler<T,1>, Stencils.hpp:83,88; element 0 for advect_scalars, Kernel.cu:133,192,225)."""
    B = f.shape[0]
    i0, j0, k0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64), np.floor(z).astype(np.int64)
    fx, fy, fz = x - i0, y - j0, z - k0
    out = 0.0
    for di in (0, 1):
        for dj in (0, 1):
            for dk in (0, 1):
                ii, jj, kk = i0 + di, j0 + dj, k0 + dk
                inside = (ii >= 0) & (ii < B) & (jj >= 0) & (jj < B) & (kk >= 0) & (kk < B)
                v = f[np.clip(ii, 0, B - 1), np.clip(jj, 0, B - 1), np.clip(kk, 0, B - 1)].astype(np.float64)
                ins = inside if v.ndim == 3 else inside[..., None]
                v = np.where(ins, v, outside)
                w = (fx if di else 1 - fx) * (fy if dj else 1 - fy) * (fz if dk else 1 - fz)
                out = out + (w if v.ndim == 3 else w[..., None]) * v
    return out


def clamp_set(f: np.ndarray, forward: np.ndarray, outside) -> tuple:
    """min / max over {centre, 6 face neighbours, forward sample} (Kernel.cu:330-349,403-430,250-264)"""
    mn, mx = f.copy(), f.copy()
    for ax in range(3):
        for d in (-1, 1):
            n = shift(f, ax, d, fill=np.nan)
            n = np.where(np.isnan(n), outside, n)
            mn, mx = np.minimum(mn, n), np.maximum(mx, n)
    return np.minimum(mn, forward), np.maximum(mx, forward)


def half_integer_velocity(box: Box, seed: int) -> np.ndarray:
    """displacement field in voxels per unit scaled_dt: multiples of 0.5 in [-2.5, 2.5], piecewise constant on 4^3 blocks so
    that the back-and-forth traces really differ from the identity"""
    rng = np.random.default_rng(seed)
    n = box.B // 4
    coarse = rng.integers(-5, 6, size=(n, n, n, 3)).astype(np.float64) * 0.5
    return np.repeat(np.repeat(np.repeat(coarse, 4, 0), 4, 1), 4, 2)


def bfecc_expected(phi: np.ndarray, vel: np.ndarray, scaled_dt: float, box: Box, outside_phi, outside_vel) -> np.ndarray:
    """phiForward = phi(back), back = x - sdt * u(x); fwd = back + sdt * u(back); phiBackward = phi(fwd);
    corr = phiForward + 0.5 * (phi - phiBackward); clamp to the clamp set (Kernel.cu:300-351)."""
    i, j, k = box.ijk()
    bx, by, bz = i - scaled_dt * vel[..., 0], j - scaled_dt * vel[..., 1], k - scaled_dt * vel[..., 2]
    fwd_phi = trilinear_exact(phi, bx, by, bz, outside_phi)
    vb = trilinear_exact(vel, bx, by, bz, outside_vel)
    fx, fy, fz = bx + scaled_dt * vb[..., 0], by + scaled_dt * vb[..., 1], bz + scaled_dt * vb[..., 2]
    back_phi = trilinear_exact(phi, fx, fy, fz, outside_phi)
    corr = fwd_phi + 0.5 * (phi - back_phi)
    mn, mx = clamp_set(phi, fwd_phi, outside_phi)
    return np.maximum(mn, np.minimum(corr, mx))


def kat_advect_scalar_half_integer(K, box: Box):
    vel = half_integer_velocity(box, 5)
    rng = np.random.default_rng(6)
    phi = rng.integers(0, 64, size=(box.B,) * 3).astype(np.float64) * 8.0  # coarse enough for exact 1/8 weights and the 0.5 factor
    dt, inv_dx = 0.5, 2.0  # scaled_dt = dt * inv_dx = 1 (Kernel.cu:276)
    got = box.to_dense(K.advect_scalar(box.to_flat(vel), box.to_flat(phi), dt, inv_dx))
    want = bfecc_expected(phi, vel, 1.0, box, 0.0, 0.0)
    assert np.array_equal(got, exact32(want))
    assert (want != phi).mean() > 0.5  # the case is not the identity


def kat_advect_scalar_uniform_shift(K, box: Box):
    """Uniform velocity, integer displacement: BFECC's error term vanishes and the clamp is the identity, so the interior is
    a pure shift of the field -- the closed form phi(x - s)."""
    i, j, k = box.ijk()
    phi = 3.0 * i + 5.0 * j - 2.0 * k + 40.0
    s = np.array([2.0, -1.0, 3.0])
    vel = np.broadcast_to(s, phi.shape + (3,)).copy()
    got = box.to_dense(K.advect_scalar(box.to_flat(vel), box.to_flat(phi), 0.25, 4.0))
    m = 6
    inner = (slice(m, -m),) * 3
    want = 3.0 * (i - s[0]) + 5.0 * (j - s[1]) - 2.0 * (k - s[2]) + 40.0
    assert np.array_equal(got[inner], exact32(want)[inner])


def kat_advect_vector_half_integer(K, box: Box):
    vel = half_integer_velocity(box, 9) * 4.0  # multiples of 2: every lerp of the velocity itself stays dyadic and small
    scaled_dt = 0.25  # displacement = multiples of 0.5 voxel
    got = box.to_dense(K.advect_vector(box.to_flat(vel), 0.125, 2.0))
    i, j, k = box.ijk()
    bx, by, bz = i - scaled_dt * vel[..., 0], j - scaled_dt * vel[..., 1], k - scaled_dt * vel[..., 2]
    vf = trilinear_exact(vel, bx, by, bz, 0.0)
    fx, fy, fz = bx + scaled_dt * vf[..., 0], by + scaled_dt * vf[..., 1], bz + scaled_dt * vf[..., 2]
    vb = trilinear_exact(vel, fx, fy, fz, 0.0)
    corr = vf + 0.5 * (vel - vb)  # Kernel.cu:397-399
    want = np.empty_like(vel)
    for c in range(3):
        mn, mx = clamp_set(vel[..., c], vf[..., c], 0.0)
        want[..., c] = np.maximum(mn, np.minimum(corr[..., c], mx))
    assert np.array_equal(got, exact32(want))
    assert (want != vel).mean() > 0.3


def kat_advect_scalars_half_integer(K, box: Box):
    """advect_scalars (Kernel.cu:118-266): one backtrace for S fields, weight-product trilinear, and the element-0 quirk:
    taps outside the domain read ELEMENT 0 of the array (Kernel.cu:133,192,225) -- of the velocity too."""
    vel = half_integer_velocity(box, 21)
    rng = np.random.default_rng(22)
    phis = [rng.integers(0, 64, size=(box.B,) * 3).astype(np.float64) * 8.0 for _ in range(3)]
    flat_vel = box.to_flat(vel)
    flat_phis = [box.to_flat(p) for p in phis]
    flat_phis[0][0] = 104.0  # make element 0 visibly non-zero
    flat_vel[0] = (1.0, -0.5, 2.0)
    vel = box.to_dense(flat_vel).astype(np.float64)
    phis = [box.to_dense(p).astype(np.float64) for p in flat_phis]
    got = K.advect_scalars(flat_vel, flat_phis, 0.5, 2.0)
    for g, phi, fp in zip(got, phis, flat_phis):
        want = bfecc_expected(phi, vel, 1.0, box, float(fp[0]), flat_vel[0].astype(np.float64))
        assert np.array_equal(box.to_dense(g), exact32(want))


# ---------------------------------------------------------------------------------------------------------------
# pointwise kernels
# ---------------------------------------------------------------------------------------------------------------


def kat_combustion_branch_table(K, box: Box = None):
    """combustion_oxygen (Kernel.cu:923-966), one row per branch; dyadic inputs, every product exact.
    columns: fuel, waste, temperature, flame, div | expected fuel, waste, temperature, flame, div"""
    tg, ex = 4.0, 0.5  # temp_gain, expansion
    rows = [
        # fuel < 0.001 -> 0; oxygen = 1 - waste >= 0; burn = 0: everything passes through, flame = max(flame, 0)
        (0.0005, 0.25, 30.0, 0.125, 2.0, 0.0, 0.25, 30.0, 0.125, 2.0),
        # oxygen < 0 (fuel + waste > 1): invalid state, inputs copied, divergence untouched
        (0.75, 0.5, 31.0, 0.5, 3.0, 0.75, 0.5, 31.0, 0.5, 3.0),
        # oxygen-limited: fuel 0.75, waste 0.125 -> oxygen 0.125 = burn; flame = max(0, min(1, 1.25)) = 1
        (0.75, 0.125, 20.0, 0.0, 1.0, 0.625, 0.375, 20.5, 1.0, 1.0625),
        # fuel-limited: fuel 0.0625, waste 0 -> burn 0.0625; flame = max(0.25, 0.625) = 0.625
        (0.0625, 0.0, 10.0, 0.25, -1.0, 0.0, 0.125, 10.25, 0.625, -0.96875),
        # flame already above burn*10: keeps its value
        (0.03125, 0.0, 0.0, 0.75, 0.0, 0.0, 0.0625, 0.125, 0.75, 0.015625),
        # oxygen exactly 0 is NOT the invalid branch (oxygen < 0): burn = 0
        (0.5, 0.5, 5.0, 0.0, 0.5, 0.5, 0.5, 5.0, 0.0, 0.5),
        # fuel below threshold and waste > 1: threshold first, then the invalid branch writes fuel = 0
        (0.0005, 1.5, 7.0, 0.0, 0.25, 0.0, 1.5, 7.0, 0.0, 0.25),
    ]
    t = np.array(rows, dtype=np.float64)
    n = 512  # one leaf's worth, rows repeated
    rep = np.tile(t, (n // len(rows) + 1, 1))[:n]
    cols = [np.ascontiguousarray(rep[:, c], dtype=F32) for c in range(10)]
    fuel, waste, temp, flame, div = K.combustion_oxygen(cols[0], cols[1], cols[2], cols[4].copy(), cols[3], tg, ex)
    for name, got, want in (("fuel", fuel, cols[5]), ("waste", waste, cols[6]), ("temperature", temp, cols[7]), ("flame", flame, cols[8]),
                            ("divergence", div, cols[9])):
        assert np.array_equal(np.asarray(got), want), name


def kat_buoyancy_table(K, box: Box = None):
    """temperature_buoyancy (Kernel.cu:831-847): temp <= ambient copies; else v.y += max(0, (temp - ambient) * strength) * dt"""
    dt, ambient, strength = 0.25, 23.0, 2.0
    temp = np.array([23.0, 22.0, 24.0, 31.0, 23.5, -5.0], dtype=np.float64)
    vel = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [0.5, -1.0, 0.25], [0.0, 0.0, 0.0], [-2.0, 8.0, 1.0], [1.0, 1.0, 1.0]])
    want = vel.copy()
    want[:, 1] += np.where(temp > ambient, np.maximum(0.0, (temp - ambient) * strength) * dt, 0.0)
    n = 512
    T = np.ascontiguousarray(np.tile(temp, n // len(temp) + 1)[:n], dtype=F32)
    V = np.ascontiguousarray(np.tile(vel, (n // len(temp) + 1, 1))[:n], dtype=F32)
    W = exact32(np.tile(want, (n // len(temp) + 1, 1))[:n])
    got = K.temperature_buoyancy(V, T, dt, ambient, strength)
    assert np.array_equal(got, W)
    # negative strength: the force is clamped at 0 (fmaxf(0, ...))
    got = K.temperature_buoyancy(V, T, dt, ambient, -2.0)
    assert np.array_equal(got, V)


def kat_vorticity_rigid_rotation(K, box: Box):
    """vorticityConfinement (Kernel.cu:970-1024) on a rigid rotation u = w x r: the curl is 2w everywhere, its magnitude is
    constant, the gradient of the magnitude is 0, N = 0 / (0 + 1e-5) = 0, and the kernel returns u + dt * (scale * 0) = u --
    wherever the whole stencil (1 + factor_scale voxels) stays inside the domain."""
    i, j, k = box.ijk()
    c = (box.B - 1) / 2.0 - 0.5  # centre at a half-integer offset keeps r integer-valued after scaling by 2
    rx, ry, rz = 2.0 * (i - c), 2.0 * (j - c), 2.0 * (k - c)
    w = np.array([1.0, -2.0, 3.0])
    u = np.stack([w[1] * rz - w[2] * ry, w[2] * rx - w[0] * rz, w[0] * ry - w[1] * rx], axis=-1)
    for factor_scale in (1.0, 2.0):
        got = box.to_dense(K.vorticity_confinement(box.to_flat(u), 0.25, 8.0, 0.75, factor_scale))
        m = int(factor_scale) + 1
        inner = (slice(m, -m),) * 3
        assert np.array_equal(got[inner], exact32(u)[inner]), factor_scale
    # factor_scale < 1: (int)factor_scale == 0 collapses every magnitude tap onto the centre: an exact copy EVERYWHERE
    got = box.to_dense(K.vorticity_confinement(box.to_flat(u), 0.25, 8.0, 0.75, 0.5))
    assert np.array_equal(got, exact32(u))


def kat_collision_inside_is_zero(K, box: Box):
    """enforceCollisionBoundaries (Kernel.cu:77-116): sdf < 0 zeroes the velocity; sdf >= 0.1 leaves it alone. (The blended band in
    between involves a normalisation and is pinned by the oracle-vs-HIP tests only.)"""
    i, j, k = box.ijk()
    sdf = np.where(i < box.B // 2, -1.0, 4.0)  # a half-space; nowhere inside the margin [0, 0.1)
    u = np.stack([1.0 + i, 2.0 - j, 3.0 + 0 * k], axis=-1)
    got = box.to_dense(K.enforce_collision_boundaries(box.to_flat(u), box.to_flat(sdf), 1.0 / 32.0))
    want = np.where((sdf < 0.0)[..., None], 0.0, u)
    assert np.array_equal(got, exact32(want))


CASES = {
    "divergence_linear": kat_divergence_linear,
    "gradient_linear": kat_gradient_linear,
    "rbgs_harmonic_fixed_point": kat_rbgs_harmonic_fixed_point,
    "rbgs_known_sweep": kat_rbgs_known_sweep,
    "advect_scalar_half_integer": kat_advect_scalar_half_integer,
    "advect_scalar_uniform_shift": kat_advect_scalar_uniform_shift,
    "advect_vector_half_integer": kat_advect_vector_half_integer,
    "advect_scalars_half_integer": kat_advect_scalars_half_integer,
    "combustion_branch_table": kat_combustion_branch_table,
    "buoyancy_table": kat_buoyancy_table,
    "vorticity_rigid_rotation": kat_vorticity_rigid_rotation,
    "collision_inside_is_zero": kat_collision_inside_is_zero,
}
