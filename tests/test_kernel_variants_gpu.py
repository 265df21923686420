"""Every alternative kernel form kept in the library behind an environment switch must produce the same bits as the
default one. The switches are read once per process, so each variant runs in a child process: a full Compute_Sim
(collision off and on, vorticity on) plus the single-purpose operators on two grids, results compared array by array."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[2] + "/tests")
from hnanosolver_amd import api, fields
from test_operators_gpu import build_data, snapshot
out = {}
for gname, origins, R in (("dense32", fields.dense_leaves(32), 32), ("plume", fields.plume_leaves(8, 1.0, 0.3), 64)):
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
np.savez(sys.argv[1], **out)
"""

VARIANTS = {
    "advect_64bit": {"HNS_ADVECT": "generic"},
    "sor_wave_per_leaf": {"HNS_RBGS": "wave"},
    "sor_block_per_leaf": {"HNS_RBGS": "block"},
    "sor_graph_replay": {"HNS_GRAPH": "1"},
    "schedule_linear": {"HNS_SCHEDULE": "linear"},
    "divergence_block": {"HNS_STENCIL": "block"},
    "cook_unpipelined_uncached": {"HNS_COOK_PIPELINE": "0", "HNS_COOK_CACHE": "0"},
}


def run_child(path, extra_env):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HNS_")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", CHILD, path, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(np.load(path))


@pytest.fixture(scope="module")
def default_outputs(tmp_path_factory):
    return run_child(str(tmp_path_factory.mktemp("variants") / "default.npz"), {})


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_is_bit_identical(name, default_outputs, tmp_path):
    got = run_child(str(tmp_path / "v.npz"), VARIANTS[name])
    assert got.keys() == default_outputs.keys()
    for k in default_outputs:
        assert np.array_equal(got[k], default_outputs[k]), (name, k)
