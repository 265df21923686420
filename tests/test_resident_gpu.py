"""The resident SOR kernel (whole pressure loop in one launch, waves handing their halo to each other through memory inside
the launch) against the reference's own decomposition, one launch per colour: bit-identical after any number of
iterations, from zero and from a given start, on dense, sparse and ragged grids; and under repetition."""
import numpy as np
import pytest

import hnanosolver_amd as H
from hnanosolver_amd import api, fields

pytestmark = pytest.mark.gpu


def scattered():
    rng = np.random.default_rng(3)
    lat = np.stack(np.meshgrid(*[np.arange(-6, 6)] * 3, indexing="ij"), -1).reshape(-1, 3)
    o = (lat[rng.random(len(lat)) < 0.5] * 8).astype(np.int32)
    return np.ascontiguousarray(o[fields.nanovdb_order(o)])


GRIDS = {"dense64": lambda: fields.dense_leaves(64), "dense128": lambda: fields.dense_leaves(128), "plume": lambda: fields.plume_leaves(32, 2.5, 0.22),
         "scattered": scattered, "two_leaves": lambda: np.array([[0, 0, 0], [0, 0, 8]], dtype=np.int32)}


def solve(grid, form, div, p0, iters, omega=1.93, vs=1.0 / 64):
    import torch
    from hnanosolver_amd import device as D

    H.set_option("rbgs", form)
    try:
        a = p0.clone()
        b = torch.full_like(a, 7.0)  # garbage in the second buffer: no form may read it before writing it
        return D.rbgs_iterate(grid, div, a, b, vs, omega, iters).clone()
    finally:
        H.set_option("rbgs", None)


@pytest.mark.parametrize("name", list(GRIDS))
def test_resident_matches_two_launch_form(name):
    import torch

    origins = GRIDS[name]()
    grid = api.create_grid_from_leaves(origins, 1.0 / 64)
    N = len(origins) * 512
    g = torch.Generator(device="cuda").manual_seed(5)
    div = torch.randn(N, device="cuda", generator=g)
    for iters, start in ((2, "zero"), (3, "given"), (50, "zero"), (7, "given")):
        p0 = torch.zeros(N, device="cuda") if start == "zero" else torch.randn(N, device="cuda", generator=g)
        want = solve(grid, "color", div, p0, iters)
        got = solve(grid, "resident", div, p0, iters)
        assert torch.equal(got, want), (name, iters, start)


def test_resident_repeats():
    """Compute_Sim-sized use: many solves back to back on the same buffers (flags are re-armed per launch). 64^3 = 256 wave
    records: all resident at once (128^3 needs 2,048, more than the 8 per CU this kernel's registers allow: the library then
    falls back to one launch per iteration by itself)."""
    import torch
    from hnanosolver_amd import device as D

    origins, R = fields.config_leaves("64")
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    N = len(origins) * 512
    div = torch.randn(N, device="cuda")
    a, b = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    first = None
    H.set_option("rbgs", "resident")
    try:
        for _ in range(20):
            a.zero_()
            r = D.rbgs_iterate(grid, div, a, b, 1.0 / R, 1.95, 50).clone()
            first = r if first is None else first
            assert torch.equal(r, first)
    finally:
        H.set_option("rbgs", None)
    assert torch.equal(first, solve(grid, "pair", div, torch.zeros(N, device="cuda"), 50, 1.95, 1.0 / R))
