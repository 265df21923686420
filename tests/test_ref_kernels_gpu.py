"""GPU parity against the REFERENCE'S OWN kernels: the HIP kernels and drop-in operators of libhns.so, through the C ABI,
held bit for bit to src/Cuda/Kernel.cu as compiled for the host into oracle/_ref/libhns_refk.so (built in the container
that has /root/reference; it travels to the GPU box prebuilt -- nothing here reads /root/reference). No oracle involved:
product against reference, on random sparse grids with random (non-smooth) fields."""
import numpy as np
import pytest

from hnanosolver_amd import api
from oracle_lib import RefKernelGrid, reference_kernels, reference_samplers
from test_ref_kernels import _random_case

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(reference_kernels() is None or reference_samplers() is None, reason="oracle/_ref/libhns_refk.so did not travel")]


def same(a, b, what):
    assert np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)), what


@pytest.mark.parametrize("seed,speed", [(31, 4.0), (32, 9.0), (33, 30.0)])
def test_hip_kernels_equal_reference_kernels(seed, speed):
    from hip_kernels import HipKernels

    rng, o = _random_case(seed)
    dt, vs = 1.0 / 24.0, 1.0 / 48.0
    K, H = RefKernelGrid(o), HipKernels(o, vs)
    N = K.N
    inv = float(np.float32(1.0) / np.float32(vs))
    vs32 = float(np.float32(vs))
    vel = (rng.standard_normal((N, 3)) * (speed * vs / dt / 2.0)).astype(np.float32)
    phi = [rng.standard_normal(N).astype(np.float32) for _ in range(11)]
    sdf = (rng.standard_normal(N) * 0.5).astype(np.float32)
    sdf[rng.random(N) < 0.2] = np.float32(0.05)
    p0 = rng.standard_normal(N).astype(np.float32)
    for coll in (False, True):
        s = sdf if coll else None
        same(H.advect_vector(vel, dt, inv, s, coll), K.advect_vector(vel, dt, inv, s, coll), f"advect_vector coll={coll}")
        same(H.advect_scalar(vel, phi[0], dt, inv, s, coll), K.advect_scalar(vel, phi[0], dt, inv, s, coll), f"advect_scalar coll={coll}")
        for S in (1, 5, 11):
            for a, b in zip(H.advect_scalars(vel, phi[:S], dt, inv, s, coll), K.advect_scalars(vel, phi[:S], dt, inv, s, coll)):
                same(a, b, f"advect_scalars S={S} coll={coll}")
        same(H.subtract_pressure_gradient(vel, p0, inv, s, coll), K.subtract_pressure_gradient(vel, p0, inv, s, coll), f"gradient coll={coll}")
    div = K.divergence(vel, inv)
    same(H.divergence(vel, inv), div, "divergence")
    pk = p0.copy()
    ph = p0.copy()
    for it in range(2):
        for color in (0, 1):
            K.rbgs(div, pk, vs32, color, 1.93)
            ph = H.rbgs(div, ph, vs32, color, 1.93)
            same(ph, pk, f"rbgs it={it} color={color}")
    for n_it in (1, 2, 3, 4, 7):  # the production forms (blocked / fused sweeps), warm start
        same(H.rbgs_iterations(div, vs32, 1.93, n_it, p0), K.rbgs_iterations(div, vs32, 1.93, n_it, p0), f"rbgs_iterate {n_it}")
    for fs in (0.5, 1.0, 2.0):
        same(H.vorticity_confinement(vel, dt, inv, 0.7, fs), K.vorticity_confinement(vel, dt, inv, 0.7, fs), f"vorticity fs={fs}")
    same(H.enforce_collision_boundaries(vel, sdf, vs32), K.enforce_collision_boundaries(vel, sdf, vs32), "enforce")
    same(H.temperature_buoyancy(vel, phi[1] * 30 + 20, dt, 23.0, 1.5), K.temperature_buoyancy(vel, phi[1] * 30 + 20, dt, 23.0, 1.5), "buoyancy")
    fuel, waste = np.abs(phi[2]) * 0.3, np.abs(phi[3]) * 0.6
    for a, b in zip(H.combustion_oxygen(fuel, waste, phi[4], div, np.abs(phi[5]), 0.5, 0.1), K.combustion_oxygen(fuel, waste, phi[4], div, np.abs(phi[5]), 0.5, 0.1)):
        same(a, b, "combustion_oxygen")


@pytest.mark.parametrize("collision", [False, True])
def test_drop_in_operators_equal_reference_launch_sequences(collision):
    """Compute_Sim (three chained cooks) and ProjectNonDivergent through the drop-in API against the reference's kernels
    launched in the reference's order. Compute_Sim evaluates omega with sinf on the host (HNanoSolver.cu:257): the C
    library's sinf on both sides here."""
    from hip_kernels import HipKernels

    from hnanosolver_amd import fields

    rng, o = _random_case(41, span=2, keep=0.6)
    R = 32
    f = fields.synthetic_fields(o, R)
    names = ["density", "temperature", "fuel", "waste", "flame"] + (["collision_sdf"] if collision else [])
    sdf = fields.sphere_sdf(o, R, center=(0.1, 0.1, 0.1), radius=0.2)
    state = []
    for E in (RefKernelGrid(o), HipKernels(o, 1.0 / R)):
        cur = {n: (sdf.copy() if n == "collision_sdf" else f[n].copy()) for n in names}
        vel = f["vel"].copy()
        for _ in range(3):
            if collision:
                cur["collision_sdf"][...] = sdf
            assert E.compute_sim(vel, cur, 7, 1.0 / 24.0, 1.0 / R, api.CombustionParams(factorScale=1.0, vorticityScale=0.4), collision) == 0
        u = vel.copy()
        assert E.project_non_divergent(u, 5, 1.0 / R) == 0
        state.append({"vel": vel, "proj": u, **{n: cur[n] for n in names}})
    for k in state[0]:
        same(state[1][k], state[0][k], k)


def test_config0_the_64_cube_smoke_plume_against_the_reference_itself():
    """BASELINE.json configs[0]: the 64^3 dense-active smoke plume, one substep of the WHOLE Compute sequence with 50 pressure
    iterations, "CPU reference path": the reference's own kernels (Kernel.cu built for the host, one thread, launched in
    HNanoSolver.cu:150-356's order) against Compute_Sim through the drop-in API on the GPU -- every field bit for bit."""
    from hip_kernels import HipKernels

    from hnanosolver_amd import fields

    R = 64
    o = fields.dense_leaves(R)
    f = fields.synthetic_fields(o, R)
    names = ["density", "temperature", "fuel", "waste", "flame"]
    state = []
    for E in (RefKernelGrid(o), HipKernels(o, 1.0 / R)):
        cur = {n: f[n].copy() for n in names}
        vel = f["vel"].copy()
        assert E.compute_sim(vel, cur, 50, 1.0 / 24.0, 1.0 / R, api.CombustionParams(), False) == 0
        state.append({"vel": vel, **cur})
    for k in state[0]:
        same(state[1][k], state[0][k], k)


@pytest.mark.parametrize("config,iterations", [("128", 50), ("plume", 50), ("256", 6), ("plume1024", 4)])
def test_configs_1_to_4_whole_cook_against_the_reference_itself(config, iterations):
    """BASELINE.json configs[1] (128^3 dense, 50 iterations) and configs[3] (the sparse rising plume, ~3.9k leaves): one whole
    Compute_Sim cook through the drop-in API on the GPU against the reference's own kernels (host build, one thread), bit for bit
    on every field. configs[2] (256^3) and configs[4] (the 1024^3-extent plume, 65,944 leaves, on one GPU) likewise at full SIZE
    with fewer pressure iterations, so that the single-threaded reference finishes in tens of seconds (their 50-iteration runs
    are held to the oracle in test_fullsize_gpu.py, and the oracle to the reference in test_ref_kernels.py)."""
    from hip_kernels import HipKernels

    from hnanosolver_amd import fields

    o, R = fields.config_leaves(config)
    f = fields.synthetic_fields(o, R)
    names = ["density", "temperature", "fuel", "waste", "flame"]
    state = []
    for E in (RefKernelGrid(o), HipKernels(o, 1.0 / R)):
        cur = {n: f[n].copy() for n in names}
        vel = f["vel"].copy()
        assert E.compute_sim(vel, cur, iterations, 1.0 / 24.0, 1.0 / R, api.CombustionParams(), False) == 0
        state.append({"vel": vel, **cur})
    for k in state[0]:
        same(state[1][k], state[0][k], k)
