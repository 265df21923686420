"""The frozen golden set G1-G4 (tests/golden_cases.py) -- outputs of the reference's own Kernel.cu built
for the host (tests/test_ref_kernels.py re-derives them): the oracle must reproduce the committed vectors
(CPU suite: any drift of oracle/hns_oracle.c shows), and the HIP kernels and drop-in operators must reproduce them
bit for bit on the GPU (-m gpu), from the fixture alone -- no oracle involved on that side."""
import json
import os

import numpy as np
import pytest

import golden_cases as gc
from hnanosolver_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
NPZ = os.path.join(HERE, "golden", "kernel_goldens_v1.npz")
META = json.load(open(os.path.join(HERE, "golden", "kernel_goldens_v1.json")))


def check(name, out):
    fix = np.load(NPZ)
    want_meta = META[name]["outputs"]
    assert set(out) == set(want_meta)
    origins, R = gc.grid_leaves(name)
    for k, v in gc.inputs(origins, R).items():  # the closed-form inputs regenerate bit-identically
        assert gc.digest(v)["sha256"] == META[name]["inputs"][k]["sha256"], f"{name}: input {k} differs on this machine"
    bad = []
    for k, v in out.items():
        flat = np.ascontiguousarray(v, dtype=np.float32).reshape(len(origins) * 512, -1)
        sample = flat if name in ("G1", "G2") else flat[:: gc.STRIDE]
        if not np.array_equal(sample, fix[f"{name}/{k}"]) or gc.digest(v)["sha256"] != want_meta[k]["sha256"]:
            bad.append(k)
    assert not bad, f"{name}: outputs differ from the golden fixture: {bad}"


@pytest.mark.parametrize("name", ["G1", "G2", "G3", "G4"])
def test_oracle_reproduces_golden_set(name):
    from oracle_lib import OracleGrid

    origins, _ = gc.grid_leaves(name)
    check(name, gc.run_all(OracleGrid(origins), name, api.CombustionParams))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["G1", "G2", "G3", "G4"])
def test_hip_reproduces_golden_set(name):
    from hip_kernels import HipKernels

    origins, R = gc.grid_leaves(name)
    check(name, gc.run_all(HipKernels(origins, 1.0 / R), name, api.CombustionParams))
