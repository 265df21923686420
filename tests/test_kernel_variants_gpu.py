"""Every alternative kernel form kept in the library behind hns_set_option() must produce the same bits as the default
one: a full Compute_Sim (collision off and on, vorticity on) plus the single-purpose operators on three grids, results
compared array by array. Options are read per call, so the variants run in this process, one after the other."""
import os
import subprocess
import sys

import numpy as np
import pytest

import hnanosolver_amd as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_workload():
    from hnanosolver_amd import api, fields
    from test_operators_gpu import build_data, snapshot

    out = {}
    rng = np.random.default_rng(77)
    lat = np.stack(np.meshgrid(*[np.arange(-5, 5)] * 3, indexing="ij"), -1).reshape(-1, 3)
    scat = (lat[rng.random(len(lat)) < 0.45] * 8).astype(np.int32)
    scat = np.ascontiguousarray(scat[fields.nanovdb_order(scat)])  # ragged z-runs, lone leaves, negative coordinates
    for gname, origins, R in (("dense32", fields.dense_leaves(32), 32), ("plume", fields.plume_leaves(8, 1.0, 0.3), 64), ("scattered", scat, 80)):
        vs = 1.0 / R
        for coll in (False, True):
            d = build_data(origins, R, with_sdf=coll, amplitude=160.0)
            h = api.IndexGridHandle()
            api.CreateIndexGrid(d, h, vs)
            api.Compute_Sim(d, h, 9, 1.0 / 24.0, vs, api.CombustionParams(factorScale=1.0), coll)
            for n, v in snapshot(d).items():
                out[f"{gname}/sim{int(coll)}/{n}"] = v
            h.reset()
        d = build_data(origins, R, amplitude=400.0)  # long backtraces: far taps through the hash
        api.AdvectIndexGrid(d, 1.0 / 24.0, vs)
        api.AdvectIndexGridVelocity(d, 1.0 / 24.0, vs)
        api.ProjectNonDivergent(d, 7, vs)
        for n, v in snapshot(d).items():
            out[f"{gname}/ops/{n}"] = v
    return out


VARIANTS = {
    "advect_64bit": {"advect": "generic"},
    "sor_two_launches_per_iteration": {"rbgs": "color"},  # the reference's own decomposition
    # temporally blocked (hns_sorblock.hip; rbgs = auto): the block edge forced either way on every grid -- one-leaf blocks (rows in registers; four iterations per launch up to
    # 300 leaves, two beyond) and 16^3 blocks (rows in LDS, the sweep threads fetch their own rows)
    "sor_leaf_blocks": {"sor_block_lb": "1"},
    "sor_16cube_blocks": {"sor_block_lb": "2"},
    "schedule_linear": {"schedule": "linear"},
    "divergence_block": {"stencil": "block"},
    "divergence_own_leaf_in_memory_order": {"divergence": "coalesced"},  # the default from 16,384 leaves (round 4)
    "divergence_z_pairs": {"divergence": "zpair"},  # round 6, the default from 16,384 leaves: two z-adjacent leaves per workgroup hand each other their common z face through LDS
    "divergence_row_form": {"divergence": "row"},
    "cook_unpipelined_uncached": {"cook_pipeline": "0", "cook_cache": "0"},
    # round 6: the default substep without a collision field runs divergence + combustion + buoyancy as ONE launch and advects {fuel, waste, temperature, flame} out of one
    # 16-byte-per-voxel array; this is the reference's own decomposition (three launches, five float arrays)
    "substep_unfused": {"fuse": "0"},
}
ALL_OPTIONS = sorted({k for v in VARIANTS.values() for k in v})


@pytest.fixture()
def options():
    """set options for one test, defaults restored afterwards"""

    def apply(d):
        for k, v in d.items():
            H.set_option(k, v)
            assert H.get_option(k) == v

    yield apply
    for k in ALL_OPTIONS:
        H.set_option(k, None)


@pytest.fixture(scope="module")
def default_outputs():
    for k in ALL_OPTIONS:
        H.set_option(k, None)
    return run_workload()


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_is_bit_identical(name, default_outputs, options):
    options(VARIANTS[name])
    got = run_workload()
    assert got.keys() == default_outputs.keys()
    for k in default_outputs:
        assert np.array_equal(got[k], default_outputs[k]), (name, k)


def test_unknown_option_is_refused():
    with pytest.raises(H.HNSError):
        H.set_option("rbgs", "no-such-form")
    with pytest.raises(H.HNSError):
        H.set_option("no-such-option", "1")
    assert H.get_option("no-such-option") is None


BIG_CHILD = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, sys.argv[2])
import hnanosolver_amd as H
from hnanosolver_amd import api, device as D, fields
if len(sys.argv) > 3: H.set_option("advect", sys.argv[3])
R = 576                                   # 72^3 leaves, 191 M voxels: a Vec3f field of 2.29 GB, byte offsets beyond 2^31
origins = fields.dense_leaves(R)
grid = api.create_grid_from_leaves(origins, 1.0 / R)
N = len(origins) * 512
o = torch.from_numpy(origins).cuda()
n = torch.arange(512, device="cuda", dtype=torch.int32)
loc = torch.stack([n >> 6, (n >> 3) & 7, n & 7], -1)
q = ((o[:, None, :] + loc[None, :, :]).reshape(-1, 3).to(torch.float32) + 0.5) / R
two_pi = 6.283185307179586
u = torch.stack([0.5 * torch.sin(two_pi * q[:, 1]) * torch.cos(two_pi * q[:, 2]), 0.25 * torch.sin(two_pi * q[:, 0]) + torch.exp(-((q - 0.5) ** 2).sum(1) / 0.02),
                 0.5 * torch.cos(two_pi * q[:, 0]) * torch.sin(two_pi * q[:, 1])], -1).contiguous() * (96.0 / R)
phi = [torch.sin(7.0 * q[:, 0] + 3.0 * q[:, 2]).contiguous(), (q[:, 1] * q[:, 2]).contiguous()]
del q
w = (torch.arange(N, device="cuda", dtype=torch.int64) % 65521) + 1
def digest(t):
    b = t.contiguous().view(torch.int32).to(torch.int64).reshape(N, -1)
    return [int(b.sum()), int((b * w[:, None]).sum())]
out = {}
adv = torch.empty_like(u)
D.advect_vector(grid, u, adv, 1.0 / 24.0, float(R))
out["advect_vector"] = digest(adv)
dst = [torch.empty_like(phi[0]), torch.empty_like(phi[1])]
D.advect_scalars(grid, u, phi, dst, 1.0 / 24.0, float(R))
out["advect_scalars"] = digest(dst[0]) + digest(dst[1])
D.advect_scalar(grid, u, phi[0], dst[1], 1.0 / 24.0, float(R))
out["advect_scalar"] = digest(dst[1])
tail = slice(N - 4096, N)                 # and the raw values of the last leaves, where the offsets are largest
out["tail"] = [adv[tail].cpu().numpy().tobytes().hex()[:512], dst[0][tail].cpu().numpy().tobytes().hex()[:512]]
json.dump(out, open(sys.argv[1], "w"))
"""


def test_byte_offsets_beyond_2_gib_match_the_64_bit_kernels(tmp_path):
    """The 32-bit addressed advection kernels on a field larger than 2 GiB (offsets with the top bit set) against the
    64-bit addressed ones on the same inputs: position-weighted checksums of the raw bits of every output."""
    import json

    res = []
    for name in ("auto", "generic"):  # child processes: each holds 10+ GB of device memory
        path = str(tmp_path / f"{name}.json")
        r = subprocess.run([sys.executable, "-c", BIG_CHILD, path, ROOT, name], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.load(open(path)))
    assert res[0] == res[1]
