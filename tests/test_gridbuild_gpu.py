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


def expected_waves(nbr, n_active, sched):
    """z-run parity rule: a leaf an even number of steps above the bottom of its z-run heads a wave and takes its +z
    neighbour as partner; waves are listed in schedule order of their head."""
    recs, lone = [], 0
    for l in sched:
        steps, m = 0, l
        while 0 <= nbr[m, 12] < n_active:
            m = nbr[m, 12]
            steps += 1
        if steps & 1:
            continue
        up = nbr[l, 14]
        p = up if 0 <= up < n_active else -1
        lone += p < 0
        recs.append(np.concatenate([[l], nbr[l], [p], nbr[p] if p >= 0 else np.full(27, -1)]))
    return np.array(recs, dtype=np.int32).reshape(-1, 56), lone


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
    sched, recs, lone = dev.launch_tables()
    assert np.array_equal(np.sort(sched), np.arange(n))
    assert np.array_equal(sched, expected_sched(n))
    exp_recs, exp_lone = expected_waves(nbr_h, n, sched)
    assert recs.shape == exp_recs.shape and np.array_equal(recs, exp_recs)
    assert lone == exp_lone
    members = np.concatenate([recs[:, 0], recs[recs[:, 28] >= 0, 28]])
    assert np.array_equal(np.sort(members), np.arange(n))  # every leaf is swept by exactly one wave
    dev.reset()
    host.reset()


def expected_sched_segments(n, seg):
    """launch order with XCD segments of `seg` leaves: whole rows of eight segments, the remainder one chunk per XCD"""
    body = (n // (8 * seg)) * 8 * seg
    b = np.arange(body)
    x, i = b % 8, b // 8
    head = ((i // seg) * 8 + x) * seg + i % seg
    return np.concatenate([head, body + expected_sched(n - body)]).astype(np.int32)


@pytest.mark.parametrize("seg", [1, 16, 100])
def test_segment_schedule(seg):
    import hnanosolver_amd as H

    origins = fields.plume_leaves(32, 2.5, 0.22)
    H.set_option("schedule_segment", seg)
    try:
        g = api.create_grid_from_leaves(origins, 0.1)
    finally:
        H.set_option("schedule_segment", None)
    n = len(origins)
    sched, recs, lone = g.launch_tables()
    assert np.array_equal(np.sort(sched), np.arange(n))
    assert np.array_equal(sched, expected_sched_segments(n, seg))
    exp_recs, exp_lone = expected_waves(g.neighbor_table(), n, sched)
    assert np.array_equal(recs, exp_recs) and lone == exp_lone
    g.reset()


@pytest.mark.parametrize("name", ["dense64", "plume", "scattered_dense", "single"])
def test_tile_groups(name):
    """Blocked SOR kernel: every wave record is in exactly one complete group or in the rest list; a group's members sit in
    the slots their first leaf's coordinates dictate, inside one aligned window, at one x."""
    origins = LEAF_SETS[name]()
    g = api.create_grid_from_leaves(origins, 0.1)
    _, recs, _ = g.launch_tables()
    groups, rest, (ty, tz) = g.tile_tables()
    assert np.array_equal(np.sort(np.concatenate([groups.reshape(-1), rest])), np.arange(len(recs)))
    o = origins[recs[:, 0]]
    for grp in groups:
        og = o[grp]
        assert len(set(og[:, 0].tolist())) == 1
        for s, oo in enumerate(og):
            assert (oo[1] >> 3) % ty == s // tz and (oo[2] >> 4) % tz == s % tz
        assert len({(oo[1] >> 3) // ty for oo in og}) == 1 and len({(oo[2] >> 4) // tz for oo in og}) == 1
    if name == "dense64":
        assert len(rest) == 0 and len(groups) * ty * tz == len(recs)
    g.reset()


def test_active_prefix_rebuilds_launch_tables():
    origins = fields.plume_leaves(8, 1.0, 0.3)
    g = api.create_grid_from_leaves(origins, 0.1)
    nbr = g.neighbor_table()
    for n_active in (len(origins) // 2, 1, len(origins)):
        g.set_active_leaves(n_active)
        sched, recs, lone = g.launch_tables()
        assert np.array_equal(sched, expected_sched(n_active))
        exp_recs, exp_lone = expected_waves(nbr, n_active, sched)
        assert np.array_equal(recs, exp_recs) and lone == exp_lone
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
    sched, recs, lone = h.launch_tables()
    assert lone == 0 and len(recs) == len(origins) // 2
    bad = c.copy()
    bad[512 * 20000 + 77, 2] += 1
    d.pCoords()[:] = bad
    with pytest.raises(api.HNSError, match="breaks the leaf-dense"):
        api.CreateIndexGrid(d, api.IndexGridHandle(), 1.0 / R)
