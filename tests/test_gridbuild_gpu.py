"""Device-side index-grid build (hnanosolver_amd/csrc/hns_gridbuild.hip) against the host builder of
hns_topology.cpp and against an independent numpy restatement of the launch-order rules. Integer work: every table
must match exactly. The host builder itself is pinned against the oracle and the reference's NanoVDB in
tests/test_abi.py / tests/test_oracle_pins.py."""
import time

import numpy as np
import pytest

from hnanosolver_amd import _lib, api, fields

pytestmark = pytest.mark.gpu


def scattered_leaves(n, span, seed):
    rng = np.random.default_rng(seed)
    o = np.unique(rng.integers(-span, span, size=(n, 3)), axis=0).astype(np.int32) * 8
    return np.ascontiguousarray(o[fields.nanovdb_order(o)])


LEAF_SETS = {
    "dense64": lambda: fields.dense_leaves(64),
    "plume": lambda: fields.plume_leaves(32, 2.5, 0.22),
    "plume_small": lambda: fields.plume_leaves(8, 1.0, 0.3),
    "scattered_sparse": lambda: scattered_leaves(3000, 40, 1),
    "scattered_dense": lambda: scattered_leaves(6000, 9, 2),
    "single": lambda: np.array([[-8, 16, 0]], dtype=np.int32),
    "int32_edge": lambda: np.array([[2147483640, 0, 0], [-2147483648, 0, 0], [2147483640, 8, 0], [0, 2147483640, -2147483648]], dtype=np.int32),
    "unordered": lambda: fields.dense_leaves(32)[np.random.default_rng(3).permutation(64)],
}


def expected_sched(n):
    base, rem = n // 8, n % 8
    b = np.arange(n)
    x, i = b % 8, b // 8
    return (x * base + np.minimum(x, rem) + i).astype(np.int32)


@pytest.mark.parametrize("name", list(LEAF_SETS))
def test_device_tables_match_host_builder(name):
    origins = LEAF_SETS[name]()
    dev = api.create_grid_from_leaves(origins, 0.1)
    host = api.create_grid_from_leaves(origins, 0.1, _lib.HNS_GRID_HOST_ONLY)
    nbr_d, nbr_h = dev.neighbor_table(), host.neighbor_table()
    assert np.array_equal(nbr_d, nbr_h)
    # origin hash: same answers for voxels inside, next to and far from the domain
    rng = np.random.default_rng(7)
    c = fields.leaves_to_coords(origins)
    probes = np.concatenate([c[rng.integers(0, len(c), 4000)], c[rng.integers(0, len(c), 4000)] + rng.integers(-20, 21, (4000, 3)),
                             rng.integers(-2**31, 2**31 - 1, (2000, 3))]).astype(np.int64)
    probes = np.clip(probes, -2**31, 2**31 - 1).astype(np.int32)
    assert np.array_equal(dev.offsets(probes), host.offsets(probes))
    inside = dev.offsets(c[::97])
    assert np.array_equal(inside, np.arange(len(c), dtype=np.uint64)[::97] + 1)
    # launch order
    n = len(origins)
    sched = dev.launch_order()
    assert np.array_equal(np.sort(sched), np.arange(n))  # every leaf is worked on by exactly one workgroup
    assert np.array_equal(sched, expected_sched(n))
    dev.reset()
    host.reset()


def expected_sched_segments(n, seg):
    """launch order with XCD segments of `seg` leaves: whole rows of eight segments, the remainder one chunk per XCD"""
    body = (n // (8 * seg)) * 8 * seg
    b = np.arange(body)
    x, i = b % 8, b // 8
    head = ((i // seg) * 8 + x) * seg + i % seg
    return np.concatenate([head, body + expected_sched(n - body)]).astype(np.int32)


def test_segment_schedule_of_large_grids_and_the_linear_option():
    """beyond 40,000 leaves the eight XCDs walk through the leaf list together in segments of 128 leaves; option schedule = linear is plain leaf order"""
    import hnanosolver_amd as H

    origins = fields.dense_leaves(288)  # 46,656 leaves
    g = api.create_grid_from_leaves(origins, 0.1)
    n = len(origins)
    sched = g.launch_order()
    assert np.array_equal(np.sort(sched), np.arange(n)) and np.array_equal(sched, expected_sched_segments(n, 128))
    g.reset()
    H.set_option("schedule", "linear")
    try:
        g = api.create_grid_from_leaves(fields.plume_leaves(32, 2.5, 0.22), 0.1)
        assert np.array_equal(g.launch_order(), np.arange(g.leaf_count()))
        g.reset()
    finally:
        H.set_option("schedule", None)


def test_active_prefix_rebuilds_launch_tables():
    origins = fields.plume_leaves(8, 1.0, 0.3)
    g = api.create_grid_from_leaves(origins, 0.1)
    for n_active in (len(origins) // 2, 1, len(origins)):
        g.set_active_leaves(n_active)
        sched = g.launch_order()
        assert np.array_equal(sched, expected_sched(n_active))
    g.reset()


def test_device_build_rejects_duplicates_and_misaligned_origins():
    o = fields.dense_leaves(32).copy()
    o[17] = o[40]
    with pytest.raises(api.HNSError, match="appears twice"):
        api.create_grid_from_leaves(o, 0.1)
    o = fields.dense_leaves(32).copy()
    o[5, 1] += 3
    with pytest.raises(api.HNSError, match="not 8-aligned"):
        api.create_grid_from_leaves(o, 0.1)


def test_empty_grid():
    g = api.create_grid_from_leaves(np.zeros((0, 3), dtype=np.int32), 0.1)
    assert g.leaf_count() == 0
    assert g.offsets(np.array([[0, 0, 0]], dtype=np.int32))[0] == 0
    g.reset()


def test_create_index_grid_from_coordinates_256(capsys):
    """The cook-side entry point at the roofline configuration: 16.7M coordinates validated on the host, tables built
    on the device. Prints the time for DESIGN.md; asserts only correctness."""
    origins, R = fields.config_leaves("256")
    c = fields.leaves_to_coords(origins)
    d = api.GridIndexedData()
    d.allocateCoords(len(c))
    d.pCoords()[:] = c
    ts = []
    for _ in range(4):
        h = api.IndexGridHandle()
        t0 = time.perf_counter()
        api.CreateIndexGrid(d, h, 1.0 / R)
        ts.append(time.perf_counter() - t0)
        if _ < 3:
            h.reset()
    with capsys.disabled():
        print(f"\n[gridbuild] CreateIndexGrid 256^3 from {len(c)} coordinates: best {1e3 * min(ts):.2f} ms")
    assert h.leaf_count() == len(origins)
    host = api.create_grid_from_leaves(origins, 1.0 / R, _lib.HNS_GRID_HOST_ONLY)
    assert np.array_equal(h.neighbor_table(), host.neighbor_table())
    assert np.array_equal(np.sort(h.launch_order()), np.arange(len(origins)))
    bad = c.copy()
    bad[512 * 20000 + 77, 2] += 1
    d.pCoords()[:] = bad
    with pytest.raises(api.HNSError, match="breaks the leaf-dense"):
        api.CreateIndexGrid(d, api.IndexGridHandle(), 1.0 / R)
