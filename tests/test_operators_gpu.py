"""GPU parity of the drop-in operators (host pointers in, results in place) against the oracle's restatement of the
reference's host drivers (HNanoSolver.cu:9-372, PressureProjection.cu:9-125, Advection.cu:13-166), through the
GridIndexedData mirror -- so the tests read like the reference's own call sites (SOP_HNanoSolver.cpp:201-256)."""
import numpy as np
import pytest

from hnanosolver_amd import api, fields

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: 1e-5 relative L-inf


def rel_linf(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def build_data(origins, R, with_sdf=False, amplitude=96.0, combustion=True):
    f = fields.synthetic_fields(origins, R, amplitude_voxels=amplitude)
    coords = fields.leaves_to_coords(origins)
    d = api.GridIndexedData()
    d.allocateCoords(len(coords))
    d.pCoords()[:] = coords
    # insertion order as the HNanoSolver SOP adds them: float grids first, then velocity (order of getBlocksOfType matters)
    order = ["density", "temperature", "fuel", "waste", "flame"]
    for name in order:
        d.addValueBlock(name, d.FLOAT)
        d.pValues(name)[:] = f[name] if (combustion or name in ("density", "temperature")) else 0.0
    if with_sdf:
        d.addValueBlock("collision_sdf", d.FLOAT)
        sdf = fields.sphere_sdf(origins, R, center=(0.5, 0.3, 0.5), radius=0.15)
        sdf[::11] = np.float32(0.04)
        d.pValues("collision_sdf")[:] = sdf
    d.addValueBlock("vel", d.VEC3F)
    d.pValues("vel")[:] = f["vel"]
    return d


def snapshot(d):
    return {n: d.pValues(n).copy() for n in d.getBlocksOfType(d.FLOAT) + d.getBlocksOfType(d.VEC3F)}


GRIDS = {
    "dense32": (lambda: fields.dense_leaves(32), 32),
    "plume_small": (lambda: fields.plume_leaves(8, 1.0, 0.3), 64),
}


@pytest.mark.parametrize("gname", list(GRIDS))
@pytest.mark.parametrize("collision", [False, True])
@pytest.mark.parametrize("factor_scale", [0.5, 1.0])
def test_compute_sim(gname, collision, factor_scale):
    from oracle_lib import OracleGrid

    mk, R = GRIDS[gname]
    origins = mk()
    vs, dt, iters = 1.0 / R, 1.0 / 24.0, 20
    d = build_data(origins, R, with_sdf=collision)
    want = snapshot(d)
    params = api.CombustionParams(factorScale=factor_scale)
    G = OracleGrid(origins)
    names = d.getBlocksOfType(d.FLOAT)
    rc = G.compute_sim(want["vel"], {n: want[n] for n in names}, iters, dt, vs, params, collision)
    assert rc == 0

    h = api.IndexGridHandle()
    api.CreateIndexGrid(d, h, vs)
    api.Compute_Sim(d, h, iters, dt, vs, params, collision)
    for n in names + ["vel"]:
        r = rel_linf(d.pValues(n), want[n])
        assert r <= TOL, f"Compute_Sim {gname} collision={collision} fs={factor_scale}: field {n} rel L-inf {r:.3e}"
    if collision:  # the reference hands the SDF back zeroed (HNanoSolver.cu:364-369)
        assert not d.pValues("collision_sdf").any()


def test_compute_sim_with_an_unused_sdf_block():
    """A "collision_sdf" block is present but hasCollision is false: the reference ignores it, advects every other block
    and still hands the SDF array back zeroed (HNanoSolver.cu:66-75,327,364-369)."""
    from oracle_lib import OracleGrid

    origins, R = fields.plume_leaves(8, 1.0, 0.3), 64
    vs, dt, iters = 1.0 / R, 1.0 / 24.0, 10
    d = build_data(origins, R, with_sdf=True)
    want = snapshot(d)
    params = api.CombustionParams()
    names = d.getBlocksOfType(d.FLOAT)
    assert OracleGrid(origins).compute_sim(want["vel"], {n: want[n] for n in names}, iters, dt, vs, params, False) == 0
    h = api.IndexGridHandle()
    api.CreateIndexGrid(d, h, vs)
    for _ in range(2):  # second call: warm buffers
        d2 = build_data(origins, R, with_sdf=True)
        api.Compute_Sim(d2, h, iters, dt, vs, params, False)
        for n in names + ["vel"]:
            assert np.array_equal(d2.pValues(n), want[n]), n
        assert not d2.pValues("collision_sdf").any()


@pytest.mark.parametrize("iters", [1, 2, 50])
def test_project_non_divergent(iters):
    from oracle_lib import OracleGrid

    origins, R = fields.dense_leaves(32), 32
    d = api.GridIndexedData()
    c = fields.leaves_to_coords(origins)
    d.allocateCoords(len(c))
    d.pCoords()[:] = c
    d.addValueBlock("vel", d.VEC3F)
    d.pValues("vel")[:] = fields.synthetic_fields(origins, R)["vel"]
    want = d.pValues("vel").copy()
    assert OracleGrid(origins).project_non_divergent(want, iters, 1.0 / R) == 0
    api.ProjectNonDivergent(d, iters, 1.0 / R)
    assert rel_linf(d.pValues("vel"), want) <= TOL
    assert np.array_equal(d.pValues("vel"), want), "ProjectNonDivergent is expected to be bit-identical to the oracle"


def test_divergence_operator():
    from oracle_lib import OracleGrid

    origins, R = fields.plume_leaves(8, 1.0, 0.3), 64
    d = api.GridIndexedData()
    c = fields.leaves_to_coords(origins)
    d.allocateCoords(len(c))
    d.pCoords()[:] = c
    d.addValueBlock("vel", d.VEC3F)
    d.addValueBlock("divergence", d.FLOAT)
    d.pValues("vel")[:] = fields.synthetic_fields(origins, R)["vel"]
    want = np.zeros(len(c), np.float32)
    OracleGrid(origins).divergence_op(d.pValues("vel"), want, 1.0 / R)
    api.Divergence(d, 1.0 / R)
    assert np.array_equal(d.pValues("divergence"), want)


def test_advect_operators():
    from oracle_lib import OracleGrid

    origins, R = fields.dense_leaves(32), 32
    f = fields.synthetic_fields(origins, R)
    c = fields.leaves_to_coords(origins)
    d = api.GridIndexedData()
    d.allocateCoords(len(c))
    d.pCoords()[:] = c
    for n in ("density", "temperature"):
        d.addValueBlock(n, d.FLOAT)
        d.pValues(n)[:] = f[n]
    d.addValueBlock("vel", d.VEC3F)
    d.pValues("vel")[:] = f["vel"]
    G = OracleGrid(origins)
    want = [f["density"].copy(), f["temperature"].copy()]
    G.advect_index_grid(f["vel"], want, 1.0 / 24.0, 1.0 / R)
    api.AdvectIndexGrid(d, 1.0 / 24.0, 1.0 / R)
    assert np.array_equal(d.pValues("density"), want[0]) and np.array_equal(d.pValues("temperature"), want[1])
    assert np.array_equal(d.pValues("vel"), f["vel"]), "AdvectIndexGrid must leave the velocity block untouched"

    wv = f["vel"].copy()
    G.advect_index_grid_velocity(wv, 1.0 / 24.0, 1.0 / R)
    api.AdvectIndexGridVelocity(d, 1.0 / 24.0, 1.0 / R)
    assert np.array_equal(d.pValues("vel"), wv)


def test_error_behaviour_matches_reference():
    origins, R = fields.dense_leaves(16), 16
    d = build_data(origins, R)
    h = api.IndexGridHandle()
    api.CreateIndexGrid(d, h, 1.0 / R)
    p = api.CombustionParams()
    with pytest.raises(ValueError, match="voxelSize must be positive"):
        api.Compute_Sim(d, h, 10, 0.1, 0.0, p, False)
    with pytest.raises(ValueError, match="cannot be negative"):
        api.Compute_Sim(d, h, 10, -1.0, 1.0 / R, p, False)
    with pytest.raises(ValueError, match="iterations must be positive"):
        api.Compute_Sim(d, h, 0, 0.1, 1.0 / R, p, False)
    with pytest.raises(ValueError, match="null grid"):
        api.Compute_Sim(d, api.IndexGridHandle(), 10, 0.1, 1.0 / R, p, False)
    # missing combustion field -> runtime_error, host arrays untouched (HNanoSolver.cu:193-201)
    d2 = api.GridIndexedData()
    c = fields.leaves_to_coords(origins)
    d2.allocateCoords(len(c))
    d2.pCoords()[:] = c
    d2.addValueBlock("density", d2.FLOAT)
    d2.addValueBlock("vel", d2.VEC3F)
    d2.pValues("density")[:] = 1.0
    with pytest.raises(RuntimeError, match="Missing required input field for combustion"):
        api.Compute_Sim(d2, h, 10, 0.1, 1.0 / R, p, False)
    assert (d2.pValues("density") == 1.0).all()
    # two Vec3f blocks -> runtime_error (HNanoSolver.cu:42-45)
    d.addValueBlock("vel2", d.VEC3F)
    with pytest.raises(RuntimeError, match="exactly one Vec3f block"):
        api.Compute_Sim(d, h, 10, 0.1, 1.0 / R, p, False)
    with pytest.raises(RuntimeError, match="exactly one Vec3f block"):
        api.ProjectNonDivergent(d, 5, 1.0 / R)
    # not leaf-dense coordinates
    bad = api.GridIndexedData()
    bad.allocateCoords(3)
    bad.pCoords()[:] = [[1, 2, 3], [1, 2, 4], [8, 2, 3]]
    with pytest.raises(RuntimeError, match="leaf-dense"):
        api.CreateIndexGrid(bad, api.IndexGridHandle(), 1.0)
    # empty domain is a no-op (HNanoSolver.cu:26-28)
    e = api.GridIndexedData()
    e.allocateCoords(0)
    for n in ("fuel", "waste", "temperature", "flame"):
        e.addValueBlock(n, e.FLOAT)
    e.addValueBlock("vel", e.VEC3F)
    he = api.IndexGridHandle()
    api.CreateIndexGrid(e, he, 1.0)
    api.Compute_Sim(e, he, 5, 0.1, 1.0, p, False)


def test_device_resident_substeps_match_repeated_cooks():
    """Two substeps with fields left on the device == two Compute_Sim cooks through the host."""
    from hnanosolver_amd import device as D

    origins, R = fields.dense_leaves(32), 32
    vs, dt, iters = 1.0 / R, 1.0 / 24.0, 10
    d = build_data(origins, R)
    h = api.IndexGridHandle()
    api.CreateIndexGrid(d, h, vs)
    p = api.CombustionParams()
    names = d.getBlocksOfType(d.FLOAT)
    sim = D.Sim(h, names)
    arrays = {n: d.pValues(n).copy() for n in names}
    arrays["vel"] = d.pValues("vel").copy()
    sim.upload(arrays)
    sim.substep(iters, dt, vs, p)
    sim.substep(iters, dt, vs, p)
    sim.download(arrays)
    api.Compute_Sim(d, h, iters, dt, vs, p, False)
    api.Compute_Sim(d, h, iters, dt, vs, p, False)
    for n in names + ["vel"]:
        assert np.array_equal(arrays[n], d.pValues(n)), n


def test_cook_cache_is_invisible():
    """Operator calls keep their device buffers with the grid between cooks. Results must not depend on
    whether a call found warm buffers, on what the previous call left in them, or on the cache being disabled."""
    origins, R = fields.plume_leaves(8, 1.0, 0.3), 64
    vs, dt, iters = 1.0 / R, 1.0 / 24.0, 12
    p = api.CombustionParams(factorScale=1.0)

    def cook(handle, amplitude):
        d = build_data(origins, R, with_sdf=True, amplitude=amplitude)
        api.Compute_Sim(d, handle, iters, dt, vs, p, True)
        return snapshot(d)

    h = api.IndexGridHandle()
    api.CreateIndexGrid(build_data(origins, R), h, vs)
    cold = cook(h, 96.0)            # allocates the entry
    other = cook(h, 250.0)          # different data through the warm buffers
    warm = cook(h, 96.0)            # same inputs again, warm and dirty buffers
    for n in cold:
        assert np.array_equal(cold[n], warm[n]), n
    assert any(not np.array_equal(cold[n], other[n]) for n in cold)
    # a second field list (single-field operators) shares the grid with the solver entry
    d = build_data(origins, R)
    vel_only = api.GridIndexedData()
    vel_only.allocateCoords(d.size())
    vel_only.pCoords()[:] = d.pCoords()
    vel_only.addValueBlock("vel", vel_only.VEC3F)
    vel_only.pValues("vel")[:] = d.pValues("vel")
    api.ProjectNonDivergent(vel_only, 10, vs, handle=h)
    first = vel_only.pValues("vel").copy()
    vel_only.pValues("vel")[:] = d.pValues("vel")
    api.ProjectNonDivergent(vel_only, 10, vs, handle=h)
    assert np.array_equal(first, vel_only.pValues("vel"))
    again = cook(h, 96.0)
    h.release_cache()
    released = cook(h, 96.0)
    for n in cold:
        assert np.array_equal(cold[n], again[n]) and np.array_equal(cold[n], released[n]), n
    h.reset()


def test_cook_cache_disabled_matches():
    import hnanosolver_amd as H

    origins, R = fields.dense_leaves(32), 32
    vs = 1.0 / R
    p = api.CombustionParams()
    outs = []
    for flag in ("1", "0"):
        H.set_option("cook_cache", flag)
        try:
            h = api.IndexGridHandle()
            d = build_data(origins, R)
            api.CreateIndexGrid(d, h, vs)
            api.Compute_Sim(d, h, 8, 1.0 / 24.0, vs, p, False)
            api.Compute_Sim(d, h, 8, 1.0 / 24.0, vs, p, False)  # second cook feeds on the first one's output
            outs.append(snapshot(d))
            h.reset()
        finally:
            H.set_option("cook_cache", None)
    for n in outs[0]:
        assert np.array_equal(outs[0][n], outs[1][n]), n


def test_create_index_grid_keeps_a_matching_handle():
    origins, R = fields.plume_leaves(8, 1.0, 0.3), 64
    d = build_data(origins, R)
    h = api.IndexGridHandle()
    api.CreateIndexGrid(d, h, 1.0 / R)
    first = h.ptr
    api.CreateIndexGrid(d, h, 1.0 / R)
    assert h.ptr == first                      # topology unchanged: same grid, nothing rebuilt
    api.CreateIndexGrid(d, h, 2.0 / R)
    assert h.ptr != first                      # voxel size changed
    second = h.ptr
    d2 = build_data(origins[:-3], R)
    api.CreateIndexGrid(d2, h, 2.0 / R)
    assert h.ptr != second and h.leaf_count() == len(origins) - 3
    h.reset()


def test_concurrent_cooks_on_one_grid():
    """Houdini may cook from several threads. Two threads sharing one grid handle: one borrows the cached device
    buffers, the other works on private ones; both must get the serial answer."""
    import threading

    origins, R = fields.dense_leaves(32), 32
    vs = 1.0 / R
    p = api.CombustionParams()
    h = api.IndexGridHandle()
    api.CreateIndexGrid(build_data(origins, R), h, vs)
    amps = [60.0, 96.0, 140.0, 200.0]
    serial = []
    for a in amps:
        d = build_data(origins, R, amplitude=a)
        api.Compute_Sim(d, h, 10, 1.0 / 24.0, vs, p, False)
        serial.append(snapshot(d))
    results, errors = [None] * len(amps), []

    def work(i):
        try:
            for _ in range(3):
                d = build_data(origins, R, amplitude=amps[i])
                api.Compute_Sim(d, h, 10, 1.0 / 24.0, vs, p, False)
                results[i] = snapshot(d)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(amps))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(len(amps)):
        for n in serial[i]:
            assert np.array_equal(serial[i][n], results[i][n]), (i, n)
    h.reset()


def test_memory_pool_reuse_across_topologies():
    """A sparse simulation changes topology every frame: each cook builds a new grid, and its fields land in the memory
    the previous grid's fields occupied (process-wide arena pool). Results must not depend on what that memory held."""
    p = api.CombustionParams(factorScale=1.0)
    shapes = [fields.plume_leaves(8, 1.0, 0.3), fields.dense_leaves(32), fields.plume_leaves(8, 1.2, 0.28), fields.plume_leaves(8, 1.0, 0.3)]
    first = {}
    for round_ in range(2):
        for k, origins in enumerate(shapes):
            d = build_data(origins, 64, with_sdf=True, amplitude=120.0 + 10.0 * k)
            h = api.IndexGridHandle()
            api.CreateIndexGrid(d, h, 1.0 / 64)
            api.Compute_Sim(d, h, 6, 1.0 / 24.0, 1.0 / 64, p, True)
            api.ProjectNonDivergent(d, 5, 1.0 / 64, handle=h)
            snap = snapshot(d)
            h.reset()                      # the grid's buffers go to the pool
            if round_ == 0:
                first[k] = snap
            else:
                for n in snap:
                    assert np.array_equal(first[k][n], snap[n]), (k, n)
        if round_ == 0:
            assert _trim() == 0            # second round starts from an empty pool again


def _trim():
    from hnanosolver_amd import _lib

    return _lib.lib.hns_trim_memory()


def test_c_abi_misuse_returns_codes_not_crashes():
    """Every entry point called the wrong way straight through ctypes: null pointers, zero / negative counts, unknown
    component counts, unnamed fields, aliased buffers. Each must return a negative code with a message; none may fault."""
    import ctypes as C

    from hnanosolver_amd import _lib

    L = _lib.lib
    origins, R = fields.dense_leaves(16), 16
    d = build_data(origins, R)
    h = api.IndexGridHandle()
    api.CreateIndexGrid(d, h, 1.0 / R)
    flds, n, keep = d._fields()
    p = api.CombustionParams()._c()
    neg = []

    def bad(rc):
        assert rc < 0, "call was accepted"
        assert L.hns_last_error(), "no message"
        neg.append(rc)

    bad(L.hns_compute_sim(None, flds, n, 5, 0.1, 1.0 / R, C.byref(p), 0, None))
    bad(L.hns_compute_sim(h.ptr, None, n, 5, 0.1, 1.0 / R, C.byref(p), 0, None))
    bad(L.hns_compute_sim(h.ptr, flds, 0, 5, 0.1, 1.0 / R, C.byref(p), 0, None))
    bad(L.hns_compute_sim(h.ptr, flds, -3, 5, 0.1, 1.0 / R, C.byref(p), 0, None))
    bad(L.hns_compute_sim(h.ptr, flds, n, 5, 0.1, 1.0 / R, None, 0, None))
    bad(L.hns_compute_sim(h.ptr, flds, n, 5, float("nan"), -1.0, C.byref(p), 0, None))
    for op in (L.hns_advect_index_grid, L.hns_advect_index_grid_velocity):
        bad(op(None, flds, n, 0.1, 1.0 / R, None))
        bad(op(h.ptr, None, n, 0.1, 1.0 / R, None))
    bad(L.hns_project_non_divergent(h.ptr, flds, n, 1 << 40, 1.0 / R, None))
    bad(L.hns_project_non_divergent(h.ptr, flds, n, 5, 0.0, None))
    bad(L.hns_divergence(h.ptr, flds, n, 1.0 / R, None))  # no block named "divergence"
    # a field with an unknown component count, one without a name, one without memory
    saved = (flds[0].ncomp, flds[0].name, flds[1].host)
    flds[0].ncomp = 2
    bad(L.hns_compute_sim(h.ptr, flds, n, 5, 0.1, 1.0 / R, C.byref(p), 0, None))
    flds[0].ncomp = saved[0]
    flds[0].name = None
    bad(L.hns_compute_sim(h.ptr, flds, n, 5, 0.1, 1.0 / R, C.byref(p), 0, None))
    flds[0].name = saved[1]
    flds[1].host = None
    bad(L.hns_compute_sim(h.ptr, flds, n, 5, 0.1, 1.0 / R, C.byref(p), 0, None))
    flds[1].host = saved[2]
    # grid functions
    err = C.c_int(0)
    assert not L.hns_grid_create(None, 512, 1.0, 0, C.byref(err)) and err.value < 0
    assert not L.hns_grid_create_from_leaves(None, 3, 1.0, 0, C.byref(err)) and err.value < 0
    bad(L.hns_grid_offsets(h.ptr, None, 4, None))
    bad(L.hns_grid_neighbor_table(h.ptr, None))
    bad(L.hns_grid_set_active_leaves(h.ptr, 10 ** 9))
    bad(L.hns_grid_matches(None, None, 0, 0))
    bad(L.hns_grid_export_nanovdb(h.ptr, None, 0, None))
    bad(L.hns_grid_launch_tables(None, None))
    bad(L.hns_grid_release_cache(None))
    # kernel-level API
    import torch

    N = len(origins) * 512
    u, v = torch.zeros(N, 3, device="cuda"), torch.zeros(N, device="cuda")
    up, vp = u.data_ptr(), v.data_ptr()
    bad(L.hns_dev_advect_vector(h.ptr, up, up, None, 0, 0.1, float(R), None))      # output aliases input
    bad(L.hns_dev_advect_vector(h.ptr, None, up, None, 0, 0.1, float(R), None))
    bad(L.hns_dev_rbgs_iterate(h.ptr, vp, vp, vp, 1.0 / R, 1.5, 3, None, None))    # p_a == p_b
    bad(L.hns_dev_rbgs_iterate(h.ptr, vp, vp, None, 1.0 / R, 1.5, 3, None, None))
    bad(L.hns_dev_rbgs_iterate(h.ptr, vp, up, vp + 4, 1.0 / R, 1.5, -1, None, None))
    bad(L.hns_dev_divergence(None, up, vp, float(R), None))
    bad(L.hns_dev_pack_leaves(vp, None, 4, vp, 1, None))
    bad(L.hns_dev_pack_leaves(vp, vp, 4, vp, 2, None))
    bad(L.hns_sim_substep(None, 5, 0.1, 1.0 / R, C.byref(p), 0, None))
    assert not L.hns_sim_create(h.ptr, None, 3, C.byref(err)) and err.value < 0
    assert len(neg) >= 30
    # and the handle still works afterwards
    api.Compute_Sim(d, h, 3, 0.1, 1.0 / R, api.CombustionParams(), False)


@pytest.mark.parametrize("seed", range(8))
def test_random_sparse_domains_against_the_oracle(seed):
    """Seeded random domains: scattered leaves around the origin (negative coordinates, lone leaves, short and long
    z-runs, holes), random field amplitude, collision and vorticity on or off, the SOR form chosen by the library's own
    heuristics -- whole Compute_Sim cooks must equal the oracle bit for bit."""
    from oracle_lib import OracleGrid

    rng = np.random.default_rng(1000 + seed)
    span = int(rng.integers(2, 7))
    lat = np.stack(np.meshgrid(*[np.arange(-span, span)] * 3, indexing="ij"), -1).reshape(-1, 3)
    keep = rng.random(len(lat)) < rng.uniform(0.15, 0.9)
    keep[rng.integers(0, len(lat))] = True
    o = (lat[keep] * 8).astype(np.int32)
    origins = np.ascontiguousarray(o[fields.nanovdb_order(o)])
    R = 16 * span
    collision, fs = bool(seed & 1), (1.0 if seed & 2 else 0.5)
    c = fields.leaves_to_coords(origins)
    q = (c.astype(np.float64) + 0.5) / R
    N = len(c)
    amp = float(rng.uniform(20.0, 300.0)) / R
    vel = (amp * np.stack([np.sin(5.0 * q[:, 1] + seed) * np.cos(3.0 * q[:, 2]), np.cos(4.0 * q[:, 0]) + 0.3 * rng.standard_normal(N),
                           np.sin(6.0 * q[:, 0] * q[:, 1])], -1)).astype(np.float32)
    d = api.GridIndexedData()
    d.allocateCoords(N)
    d.pCoords()[:] = c
    vals = {"density": rng.random(N), "temperature": 20.0 + 60.0 * rng.random(N), "fuel": 0.3 * rng.random(N) * (rng.random(N) < 0.3),
            "waste": 0.2 * rng.random(N), "flame": rng.random(N) * (rng.random(N) < 0.1)}
    for n, v in vals.items():
        d.addValueBlock(n, d.FLOAT)
        d.pValues(n)[:] = v.astype(np.float32)
    if collision:
        d.addValueBlock("collision_sdf", d.FLOAT)
        d.pValues("collision_sdf")[:] = ((np.linalg.norm(q - 0.1, axis=1) - 0.35) * R).astype(np.float32)
    d.addValueBlock("vel", d.VEC3F)
    d.pValues("vel")[:] = vel
    want = snapshot(d)
    params = api.CombustionParams(factorScale=fs, vorticityScale=0.7)
    names = d.getBlocksOfType(d.FLOAT)
    iters = int(rng.integers(1, 40))
    assert OracleGrid(origins).compute_sim(want["vel"], {n: want[n] for n in names}, iters, 0.05, 1.0 / R, params, collision) == 0
    h = api.IndexGridHandle()
    api.CreateIndexGrid(d, h, 1.0 / R)
    api.Compute_Sim(d, h, iters, 0.05, 1.0 / R, params, collision)
    for n in names + ["vel"]:
        assert np.array_equal(d.pValues(n), want[n]), f"seed {seed} ({len(origins)} leaves, {iters} iterations, collision={collision}, fs={fs}): {n} differs, rel L-inf {rel_linf(d.pValues(n), want[n]):.2e}"
    h.reset()


@pytest.mark.parametrize("collision", [False, True])
def test_feedback_cooks_keep_fields_on_the_device(collision):
    """Round 4: the SOP feeds frame n's output back in as frame n+1's input (SOP_HNanoSolver.cpp:106).
    Compute_Sim(..., feedback=True) vouches for that, and fields whose host arrays still carry the signature of what the previous
    cook handed back are not uploaded again. Same bits as plain cooks; a broken promise (an array that changed) is noticed and
    uploaded; another operator on the same state in between invalidates what it touched."""
    origins, R = fields.plume_leaves(8, 1.0, 0.3), 64
    vs, dt, iters = 1.0 / R, 1.0 / 24.0, 9
    params = api.CombustionParams(factorScale=1.0, vorticityScale=0.3)
    names = ["density", "temperature", "fuel", "waste", "flame", "vel"]

    def cooks(feedback, n=4, poke=None, project_before=None, checked=False, poke_at=0):
        d = build_data(origins, R, with_sdf=collision)
        sdf = d.pValues("collision_sdf").copy() if collision else None
        h = api.IndexGridHandle()
        api.CreateIndexGrid(d, h, vs)
        skipped = []
        for c in range(n):
            if collision:
                d.pValues("collision_sdf")[:] = sdf  # the SOP reads the collider every cook; Compute hands it back zeroed
            if poke == c:
                d.pValues("density")[poke_at] += 1.0  # element 0 is one of the signature's samples, element 1 is not
            if project_before == c:
                api.ProjectNonDivergent(d, 3, vs, handle=h)
            skipped.append(api.Compute_Sim(d, h, iters, dt, vs, params, collision, feedback=feedback if c else None, checked=checked))
        out = snapshot(d)
        h.reset()
        return out, skipped

    want, _ = cooks(None)
    got, skipped = cooks(True, checked=False)
    assert skipped == [None, 6, 6, 6], skipped  # vouched: velocity + five float blocks stay on the device (the SDF goes up every cook)
    for n in names:
        assert np.array_equal(got[n], want[n]), n
    got, skipped = cooks(True, checked=True)
    assert skipped == [None, 0, 6, 6], skipped  # checked: the first asking cook finds no digest to compare with and uploads; from then on as above
    for n in names:
        assert np.array_equal(got[n], want[n]), n
    # a sparse edit that misses the 4,096 samples. CHECKED notices it (same bits as plain cooks); VOUCHED does not -- that is its contract
    want, _ = cooks(None, poke=2, poke_at=1)
    got, skipped = cooks(True, poke=2, poke_at=1, checked=True)
    assert skipped == [None, 0, 5, 6], skipped
    for n in names:
        assert np.array_equal(got[n], want[n]), n
    _, skipped = cooks(True, poke=2, poke_at=1, checked=False)
    assert skipped == [None, 6, 6, 6], skipped
    want, _ = cooks(None)
    got, skipped = cooks(["density", "vel", "fuel"], checked=False)
    assert skipped == [None, 3, 3, 3], skipped
    for n in names:
        assert np.array_equal(got[n], want[n]), n
    want, _ = cooks(None, poke=2)
    got, skipped = cooks(True, poke=2, checked=False)
    assert skipped == [None, 6, 5, 6], skipped  # the changed array is noticed (element 0 is a sample) and uploaded
    for n in names:
        assert np.array_equal(got[n], want[n]), n
    want, _ = cooks(None, project_before=2)
    got, skipped = cooks(True, project_before=2, checked=False)
    assert skipped == [None, 6, 5, 6], skipped  # ProjectNonDivergent uploaded a velocity of its own into the shared state
    for n in names:
        assert np.array_equal(got[n], want[n]), n
