"""Temporally blocked red-black SOR (hnanosolver_amd/csrc/hns_sorblock.hip: K iterations per launch on a block of leaves with a
2K-voxel halo) against the oracle's restatement of redBlackGaussSeidelUpdate (reference src/Cuda/Kernel.cu:591-623, two launches
per iteration) and against the library's own two-launch form: bit for bit, on dense, ragged and scattered leaf sets with negative
coordinates, for even and odd iteration counts, warm-started and from zero."""
import numpy as np
import pytest
import torch

import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
from oracle_lib import OracleGrid

pytestmark = pytest.mark.gpu

SHAPES = [1, 2]  # block edge in leaves: one-leaf blocks (k_rbgs_block<1, K>: four iterations per launch up to 300 leaves, two beyond) and 16^3 blocks (k_rbgs_block_xy), each forced on every leaf set


def leaf_sets():
    rng = np.random.default_rng(5)
    o = np.unique(rng.integers(-6, 6, size=(400, 3)).astype(np.int32) * 8, axis=0)
    scatter = np.ascontiguousarray(o[fields.nanovdb_order(o)])
    # straddles the 128^3 lower-node and 4096^3 upper-node borders at negative coordinates
    shifted = fields.dense_leaves(24).astype(np.int64) + np.array([-4104, -16, 120])
    shifted = shifted.astype(np.int32)
    shifted = np.ascontiguousarray(shifted[fields.nanovdb_order(shifted)])
    # 2 x 2 x 2 leaves in each corner region of the int32 coordinate range: a 16^3 block's neighbours lie outside it there
    lo, hi = -2**31, 2**31 - 16
    corner = np.array([[i, j, k] for i in (0, 8) for j in (0, 8) for k in (0, 8)], dtype=np.int64)
    edge = np.concatenate([corner + np.array(c) for c in ((lo, lo, lo), (hi, hi, hi), (lo, 0, hi), (hi, lo, 0))]).astype(np.int32)
    edge = np.ascontiguousarray(edge[fields.nanovdb_order(edge)])
    return {"dense32": fields.dense_leaves(32), "dense40": fields.dense_leaves(40), "plume": fields.plume_leaves(16, 1.5, 0.3), "scatter": scatter,
            "node_borders": shifted, "one_leaf": np.array([[8, -8, 0]], dtype=np.int32), "int32_edge": edge}


@pytest.fixture(autouse=True)
def restore_options():
    yield
    for k in ("rbgs", "sor_block_lb"):
        H.set_option(k, None)


def solve(grid, div, p0, iters, **opts):
    for k, v in opts.items():
        H.set_option(k, str(v))
    p_a = p0.clone()
    p_b = torch.full_like(p0, 7.0)  # stale content of the second buffer must not matter
    out = D.rbgs_iterate(grid, div, p_a, p_b, 0.013, 1.93, iters).clone()
    for k in opts:
        H.set_option(k, None)
    return out


@pytest.mark.parametrize("name", list(leaf_sets()))
def test_blocked_sor_matches_two_launch_form_and_oracle(name):
    origins = leaf_sets()[name]
    grid = api.create_grid_from_leaves(origins, 0.013)
    n = len(origins) * 512
    g = torch.Generator(device="cpu").manual_seed(3)
    div = torch.randn(n, generator=g).cuda()
    p0 = (torch.rand(n, generator=g) * 2 - 1).cuda()
    oracle = OracleGrid(origins)
    for iters in (1, 2, 3, 4, 7, 10):  # (odd counts: the one left over is one more launch of the same kernel with two colour sweeps instead of four)
        want = solve(grid, div, p0, iters, rbgs="color")
        if iters in (3, 4):
            ref = oracle.rbgs_iterations(div.cpu().numpy(), 0.013, 1.93, iters, p0.cpu().numpy())
            assert np.array_equal(want.cpu().numpy(), ref), (name, iters, "two-launch form vs oracle")
        for lb in SHAPES:
            got = solve(grid, div, p0, iters, sor_block_lb=lb)
            assert torch.equal(want, got), (name, iters, lb, float((want - got).abs().max()))


def test_default_form_by_size_is_bit_identical():
    """whatever `rbgs = auto` picks for a grid size (one-leaf blocks with four or two iterations per launch, 16^3 blocks) gives the two-launch bits"""
    for R in (16, 48, 96):
        origins = fields.dense_leaves(R)
        grid = api.create_grid_from_leaves(origins, 1.0 / R)
        n = len(origins) * 512
        g = torch.Generator(device="cpu").manual_seed(R)
        div = torch.randn(n, generator=g).cuda()
        p0 = torch.zeros(n, device="cuda")
        for iters in (5, 8):
            assert torch.equal(solve(grid, div, p0, iters, rbgs="color"), solve(grid, div, p0, iters)), (R, iters)


def test_blocked_sor_known_answer_harmonic_fixed_point():
    """a discrete-harmonic p with div = 0 is a fixed point of every sweep wherever all six neighbours exist (tests/kats.py):
    inside a dense box, away from the outside-is-zero rim, K iterations per launch must leave it untouched"""
    R = 32
    origins = fields.dense_leaves(R)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    c = fields.leaves_to_coords(origins).astype(np.float32)
    p = (c[:, 0] - 2 * c[:, 1] + 3 * c[:, 2]).astype(np.float32)  # linear: harmonic, small integers, sums exact
    p0 = torch.from_numpy(p).cuda()
    div = torch.zeros_like(p0)
    inner = torch.from_numpy(((c >= 8) & (c < R - 8)).all(1)).cuda()
    for lb in SHAPES:
        got = solve(grid, div, p0, 4, sor_block_lb=lb)
        assert torch.equal(got[inner], p0[inner]), lb


def test_every_occupancy_pattern_of_a_block():
    """a 16^3 block exists as soon as one of its 2 x 2 x 2 leaves does: all 255 occupancy patterns of one block, each inside a random half-filled
    neighbourhood of leaves (so that the 4 x 4 x 4 leaves under the tile are present and absent in every combination a mask can meet)"""
    rng = np.random.default_rng(11)
    cells = np.array([[i, j, k] for i in range(2) for j in range(2) for k in range(2)], dtype=np.int32)
    around = np.array([[i, j, k] for i in range(-2, 4) for j in range(-2, 4) for k in range(-2, 4) if not (0 <= i < 2 and 0 <= j < 2 and 0 <= k < 2)], dtype=np.int32)
    for pattern in range(1, 256):
        own = cells[[(pattern >> c) & 1 == 1 for c in range(8)]]
        o = np.concatenate([own, around[rng.random(len(around)) < 0.5]]) * 8 + np.array([-16, 32, -48], dtype=np.int32)
        o = np.ascontiguousarray(o[fields.nanovdb_order(o)].astype(np.int32))
        grid = api.create_grid_from_leaves(o, 0.02)
        n = len(o) * 512
        g = torch.Generator(device="cpu").manual_seed(pattern)
        div = torch.randn(n, generator=g).cuda()
        p0 = (torch.rand(n, generator=g) * 2 - 1).cuda()
        for iters in (4, 3) if pattern % 16 == 5 else (4,):  # (every sixteenth pattern also with an odd count: the one-iteration launch of the 16^3 kernel)
            want = solve(grid, div, p0, iters, rbgs="color")
            for lb in SHAPES:
                got = solve(grid, div, p0, iters, sor_block_lb=lb)
                assert torch.equal(want, got), (pattern, iters, lb, float((want - got).abs().max()))


@pytest.mark.parametrize("leaves", [8, 512, 4096])
def test_result_buffer_is_the_one_result_in_b_names(leaves):
    """include/hns.h, hns_dev_rbgs_iterate: which of p_a / p_b holds the result depends on the form the library picks, not on
    the parity of `iterations` -- the documented contract is *result_in_b, for every iteration count, under
    rbgs=auto. Checked through the raw C ABI: the named buffer holds the two-launch result, and it follows the plan's
    launch count."""
    import ctypes as C

    lib = H.load_library()
    R = {8: 16, 512: 64, 4096: 128}[leaves]
    origins = fields.dense_leaves(R)
    grid = api.create_grid_from_leaves(origins, 1.0 / R)
    n = len(origins) * 512
    g = torch.Generator(device="cpu").manual_seed(leaves)
    div = torch.randn(n, generator=g).cuda()
    p0 = torch.randn(n, generator=g).cuda()
    for iters in (1, 2, 3, 4, 6):
        want = solve(grid, div, p0, iters, rbgs="color")
        p_a, p_b = p0.clone(), torch.full_like(p0, 7.0)
        in_b = C.c_int(-1)
        rc = lib.hns_dev_rbgs_iterate(grid.ptr, div.data_ptr(), p_a.data_ptr(), p_b.data_ptr(), C.c_float(0.013), C.c_float(1.93), iters, C.byref(in_b), D.current_stream())
        assert rc == 0 and in_b.value in (0, 1)
        torch.cuda.synchronize()
        assert torch.equal(p_b if in_b.value else p_a, want), f"iterations = {iters}: the buffer named by result_in_b does not hold the result"
        _, launches, _ = D.rbgs_plan(grid, iters)
        assert in_b.value == (launches & 1), f"iterations = {iters}: result_in_b = {in_b.value} but the plan says {launches} launches"


@pytest.mark.parametrize("name", ["dense40", "plume", "scatter", "node_borders"])
def test_blocked_sor_over_a_launch_range(name):
    """Round 4: the blocked form over a LAUNCH RANGE [first, first + count) of the grid's leaves (what a multi-GPU rank's
    boundary / interior ranges are): two iterations in one launch, every leaf of the grid a tile source, only the leaves of
    the range stored. Owned leaves must equal two iterations of the two-launch form on the whole grid bit for bit; the other
    leaves of the output buffer must not be touched."""
    origins = leaf_sets()[name]
    n_leaves = len(origins)
    n = n_leaves * 512
    g = torch.Generator(device="cpu").manual_seed(n_leaves)
    div = torch.randn(n, generator=g).cuda()
    p0 = torch.randn(n, generator=g).cuda()
    whole = api.create_grid_from_leaves(origins, 0.013)
    want = solve(whole, div, p0, 2, rbgs="color")
    for first, count in ((0, n_leaves // 3), (n_leaves // 3, n_leaves // 2), (n_leaves - 5, 5), (7, 1), (0, n_leaves)):
        count = max(1, min(count, n_leaves - first))
        part = api.create_grid_from_leaves(origins, 0.013)
        part.set_active_range(first, count)
        for lb in (0, 1, 2):  # by size, one-leaf blocks, 16^3 blocks
            H.set_option("sor_block_lb", str(lb))
            desc, launches, per = D.rbgs_plan(part, 2)
            assert launches == 1 and "k_rbgs_block" in desc, (desc, launches)
            p_a, p_b = p0.clone(), torch.full_like(p0, 7.0)
            out = D.rbgs_iterate(part, div, p_a, p_b, 0.013, 1.93, 2)
            assert out is p_b
            sl = slice(first * 512, (first + count) * 512)
            assert torch.equal(out[sl], want[sl]), f"{name} range [{first}, +{count}) lb={lb}: owned leaves differ"
            keep = torch.ones(n, dtype=torch.bool, device="cuda")
            keep[sl] = False
            assert bool((out[keep] == 7.0).all()), f"{name} range [{first}, +{count}) lb={lb}: a leaf outside the range was written"
            assert torch.equal(p_a, p0)
        H.set_option("sor_block_lb", None)
