"""Every `path:line` citation of the reference in the boundary headers and documents resolves (three
directory names in include/hns.h did not exist). Runs where the reference checkout is present; a citation is a path
starting with src/, externals/ or Tests/ followed by :line[-line][,line[-line]...]. Bare file names (Kernel.cu:621) are
resolved by a unique basename in the checkout."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FILES = ["include/hns.h", "INTEGRATION.md", "DESIGN.md", "README.md", "oracle/README.md", "oracle/hns_oracle.h", "oracle/hns_oracle.c", "oracle/ref_kernels.cpp",
         "oracle/ref_samplers.cpp"] + sorted(glob.glob(os.path.join(ROOT, "integration", "*.h*"))) + sorted(glob.glob(os.path.join(ROOT, "integration", "*.cpp"))) + \
        sorted(glob.glob(os.path.join(ROOT, "hnanosolver_amd", "host", "*.hpp")))

PATHED = re.compile(r"((?:src|externals|Tests)/[A-Za-z0-9_./+-]+\.(?:cu|cuh|hpp|h|cpp|txt|ds)):(\d+(?:-\d+)?(?:,\s?\d+(?:-\d+)?)*)")
BARE = re.compile(r"(?<![A-Za-z0-9_/.])([A-Z][A-Za-z0-9_]+\.(?:cu|cuh|hpp|h|cpp)):(\d+(?:-\d+)?(?:,\s?\d+(?:-\d+)?)*)")


def _lines(path, cache={}):
    if path not in cache:
        with open(path, errors="replace") as f:
            cache[path] = sum(1 for _ in f)
    return cache[path]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")
def test_every_reference_citation_resolves():
    by_name = {}
    for dirpath, _, names in os.walk(REF):
        if "/.git" in dirpath:
            continue
        for n in names:
            by_name.setdefault(n, []).append(os.path.join(dirpath, n))
    bad, n_checked = [], 0
    for rel in FILES:
        path = rel if os.path.isabs(rel) else os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        text = open(path, errors="replace").read()
        found = [(m.group(1), m.group(2), True) for m in PATHED.finditer(text)]
        stripped = PATHED.sub("", text)
        found += [(m.group(1), m.group(2), False) for m in BARE.finditer(stripped)]
        for name, spans, pathed in found:
            if pathed:
                cands = [os.path.join(REF, name)]
                if not os.path.exists(cands[0]):
                    bad.append(f"{os.path.relpath(path, ROOT)}: {name} does not exist in the reference")
                    continue
            else:
                cands = by_name.get(name, [])
                if not cands:
                    continue  # not a reference file (our own sources are cited the same way)
            last = max(int(x) for x in re.findall(r"\d+", spans))
            n_checked += 1
            if not any(_lines(c) >= last for c in cands):
                bad.append(f"{os.path.relpath(path, ROOT)}: {name}:{spans} -- the file has {max(_lines(c) for c in cands)} lines")
    assert n_checked > 100, n_checked
    assert not bad, "\n".join(bad[:40])


# ---- launch order of the three host drivers (VERDICT r5, weak 1 iii) ----------------------------------------------------------------------------------------
# tests/oracle_lib.py's RefKernelGrid calls the reference's own kernel bodies (oracle/_ref/libhns_refk.so) in an order RESTATED BY HAND from HNanoSolver.cu:150-356,
# PressureProjection.cu:43-66,114 and Advection.cu:88-91,153 -- the drivers themselves need OpenVDB and CUB and cannot be built here. This pins the restatement to
# the drivers' text mechanically: the `kernel<<<` launch sites of each driver function, comments stripped, in source order, against the kernel methods the Python
# driver calls when it runs (recorded on a one-leaf grid, one iteration, collision on so that both enforceCollisionBoundaries launches appear).
KERNEL_OF = {"enforce_collision_boundaries": "enforceCollisionBoundaries", "advect_vector": "advect_vector", "vorticity_confinement": "vorticityConfinement",
             "divergence": "divergence", "divergence_opt": "divergence_opt", "combustion_oxygen": "combustion_oxygen", "temperature_buoyancy": "temperature_buoyancy",
             "rbgs": "redBlackGaussSeidelUpdate", "rbgs_opt": "redBlackGaussSeidelUpdate_opt", "subtract_pressure_gradient": "subtractPressureGradient",
             "subtract_pressure_gradient_opt": "subtractPressureGradient_opt", "advect_scalar": "advect_scalar", "advect_scalars": "advect_scalars"}


def _launches(path, function):
    """kernel names at the `<<<` launch sites inside the body of `function` (first definition in the file), comments removed, in source order"""
    text = open(os.path.join(REF, path), errors="replace").read()
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    m = re.search(r"\b" + re.escape(function) + r"\s*\([^;{]*\)\s*\{", text)
    assert m, f"{path}: no definition of {function}"
    depth, i = 1, m.end()
    while depth and i < len(text):
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return re.findall(r"\b([A-Za-z_]\w*)\s*<<<", text[m.end():i])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")
def test_launch_order_of_the_restated_drivers_is_the_reference_drivers():
    import sys

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    if oracle_lib.reference_kernels() is None:
        pytest.skip("oracle/_ref/libhns_refk.so not built")
    g = oracle_lib.RefKernelGrid(np.array([[0, 0, 0]], dtype=np.int32))
    calls = []

    def record(name):
        inner = getattr(g, name)

        def wrapped(*a, **k):
            calls.append(KERNEL_OF[name])
            return inner(*a, **k)

        setattr(g, name, wrapped)

    for name in KERNEL_OF:
        record(name)
    rng = np.random.default_rng(0)
    n = g.N

    def run(fn):
        calls.clear()
        fn()
        return list(calls)

    vel = rng.standard_normal((n, 3)).astype(np.float32)
    names = ["density", "temperature", "fuel", "waste", "flame", "collision_sdf"]
    fields = {k: rng.random(n).astype(np.float32) for k in names}
    prm = oracle_lib.orc_combustion_params(0.1, 10.0, 1.0, 23.0, 0.05, 0.5)
    got = run(lambda: g.compute_sim(vel.copy(), fields, 1, 1.0 / 24.0, 0.1, prm, True))
    want = _launches("src/Cuda/HNanoSolver.cu", "Compute")
    assert len(want) == 11 and got == want, (got, want)  # (the two redBlackGaussSeidelUpdate launches are the body of the iteration loop: one iteration here)
    got = run(lambda: g.project_non_divergent(vel.copy(), 1, 0.1))
    want = _launches("src/Cuda/PressureProjection.cu", "pressure_projection_idx")
    assert len(want) == 4 and got == want, (got, want)
    got = run(lambda: g.divergence_op(vel.copy(), np.zeros(n, dtype=np.float32), 0.1))
    assert got == _launches("src/Cuda/PressureProjection.cu", "divergence") == ["divergence"], got
    got = run(lambda: g.advect_index_grid(vel.copy(), [fields["density"].copy()], 1.0 / 24.0, 0.1))
    assert got == _launches("src/Cuda/Advection.cu", "advect_index_grid") == ["advect_scalar"], got
    got = run(lambda: g.advect_index_grid_velocity(vel.copy(), 1.0 / 24.0, 0.1))
    assert got == _launches("src/Cuda/Advection.cu", "advect_index_grid_v") == ["advect_vector"], got
