"""CPU-side checks of the drop-in boundary: libhns.so loads, exports exactly what include/hns.h declares, the host-side
topology agrees with the oracle, and nothing computes without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from hnanosolver_amd import _lib, api, fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "hns.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hns_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 45
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.library_path()], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (hns_[a-z0-9_]+)", out))
    assert set(names) <= exported, f"declared but not exported: {sorted(set(names) - exported)}"
    assert exported <= set(names), f"exported but not declared in include/hns.h: {sorted(exported - set(names))}"
    assert set(_lib.SIGNATURES) == set(names), "python binding and header disagree"
    lib = _lib.load_library()
    assert lib.hns_version() == 100


def test_no_cuda_or_reference_symbols_in_library():
    out = subprocess.run(["nm", "-D", _lib.library_path()], capture_output=True, text=True, check=True).stdout
    foreign = [l for l in out.splitlines() if not l.split()[-1].startswith("hns_")]  # hns_grid_export_nanovdb is ours: a writer, no NanoVDB code
    assert not any("cuda" in l.lower() or "nanovdb" in l.lower() or "orc_" in l for l in foreign)


def test_host_topology_matches_oracle():
    from oracle_lib import OracleGrid

    rng = np.random.default_rng(4)
    lat = np.stack(np.meshgrid(*[np.arange(-3, 3)] * 3, indexing="ij"), -1).reshape(-1, 3)
    o = (lat[rng.random(len(lat)) < 0.4] * 8).astype(np.int32)
    o = np.concatenate([o, [[-4104, 0, 0], [2 ** 31 - 8, 0, 0], [-(2 ** 31), 8, 8]]]).astype(np.int32)
    o = np.ascontiguousarray(o[rng.permutation(len(o))])  # any order is accepted: the caller's order defines the layout
    h = api.create_grid_from_leaves(o, 0.5, _lib.HNS_GRID_HOST_ONLY)
    G = OracleGrid(o)
    assert h.leaf_count() == len(o) and h.voxel_count() == len(o) * 512
    c = G.coords()
    assert np.array_equal(h.coords(), c)
    ijk = np.concatenate([c[rng.integers(0, len(c), 5000)] + rng.integers(-9, 10, (5000, 3)), rng.integers(-6000, 6000, (2000, 3))]).astype(np.int32)
    assert np.array_equal(h.offsets(ijk), G.offsets(ijk))
    # the invariant the reference relies on: offset(coords[i]) == i + 1 (Kernel.cu:505 vs :511)
    assert np.array_equal(h.offsets(c[:4096]), np.arange(1, 4097, dtype=np.uint64))
    nb = h.neighbor_table()
    for l in rng.integers(0, len(o), 40):
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    q = o[l].astype(np.int64) + 8 * np.array([dx, dy, dz])
                    want = -1
                    if np.all(q >= -(2 ** 31)) and np.all(q < 2 ** 31):
                        off = int(G.offsets([q.astype(np.int32)])[0])
                        want = (off - 1) // 512 if off else -1
                    assert nb[l, (dx + 1) * 9 + (dy + 1) * 3 + dz + 1] == want
    assert nb[:, 13].tolist() == list(range(len(o)))


def test_create_index_grid_validates_leaf_density():
    d = api.GridIndexedData()
    d.allocateCoords(3)
    d.pCoords()[:] = [[1, 2, 3], [1, 2, 4], [8, 2, 3]]  # the NanoVDB unit-test triple is NOT a leaf-dense domain
    with pytest.raises(RuntimeError, match="leaf-dense"):
        api.CreateIndexGrid(d, api.IndexGridHandle(), 1.0, _lib.HNS_GRID_HOST_ONLY)
    o = fields.dense_leaves(16)
    c = fields.leaves_to_coords(o)
    bad = c.copy()
    bad[700] += 1
    d.allocateCoords(len(c))
    d.pCoords()[:] = bad
    with pytest.raises(RuntimeError, match="breaks the leaf-dense"):
        api.CreateIndexGrid(d, api.IndexGridHandle(), 1.0, _lib.HNS_GRID_HOST_ONLY)
    dup = np.concatenate([c, c[:512]])
    d.allocateCoords(len(dup))
    d.pCoords()[:] = dup
    with pytest.raises(RuntimeError, match="appears twice"):
        api.CreateIndexGrid(d, api.IndexGridHandle(), 1.0, _lib.HNS_GRID_HOST_ONLY)
    d.allocateCoords(len(c))
    d.pCoords()[:] = c
    h = api.IndexGridHandle()
    api.CreateIndexGrid(d, h, 1.0 / 16, _lib.HNS_GRID_HOST_ONLY)
    assert h.leaf_count() == 8 and np.array_equal(h.coords(), c)


def test_grid_matches_detects_topology_changes():
    o = fields.plume_leaves(8, 1.0, 0.3)
    c = fields.leaves_to_coords(o)
    g = api.create_grid_from_leaves(o, 0.1, _lib.HNS_GRID_HOST_ONLY)

    def matches(coords, flags=0):
        coords = np.ascontiguousarray(coords, dtype=np.int32)
        return _lib.lib.hns_grid_matches(g.ptr, coords.ctypes.data, len(coords), flags)

    assert matches(c) == 1
    assert matches(c[:-512]) == 0                                   # one leaf fewer
    moved = c.copy()
    moved[512 * 5:512 * 6] += np.array([0, 0, 800], dtype=np.int32)  # same count, one leaf elsewhere
    assert matches(moved) == 0
    swapped = c.copy()
    swapped[:512], swapped[512:1024] = c[512:1024], c[:512]          # same leaves, different order = different layout
    assert matches(swapped) == 0
    bad = c.copy()
    bad[1000, 1] += 1
    assert matches(bad) == _lib.HNS_ERR_TOPOLOGY
    assert b"breaks the leaf-dense" in _lib.lib.hns_last_error()
    assert matches(bad, _lib.HNS_GRID_SKIP_VALIDATE) == 1           # only every 512th coordinate is read
    assert matches(c[:700]) == _lib.HNS_ERR_TOPOLOGY
    g.reset()


def test_grid_indexed_data_mirrors_reference_container():
    """reference Tests/IndexGrid.cpp:473-539 (GridIndexedData alloc/add/clear)"""
    d = api.GridIndexedData()
    assert d.size() == 0 and d.numValueBlocks() == 0
    d.allocateCoords(1024)
    assert d.size() == 1024 and d.pCoords().shape == (1024, 3)
    assert d.addValueBlock("density", d.FLOAT) and d.addValueBlock("vel", d.VEC3F) and d.addValueBlock("temperature", d.FLOAT)
    assert not d.addValueBlock("density", d.FLOAT)  # duplicate name refused (GridData.hpp:62-65)
    assert d.numValueBlocks() == 3
    assert d.getBlocksOfType(d.FLOAT) == ["density", "temperature"] and d.getBlocksOfType(d.VEC3F) == ["vel"]
    assert d.pValues("density").shape == (1024,) and d.pValues("vel").shape == (1024, 3)
    assert d.pValues("nope") is None and d.pValues("density", d.VEC3F) is None  # type mismatch -> nullptr
    d.pValues("density")[3] = 7.0
    assert d.pValues("density")[3] == 7.0
    d.clearValues()
    assert d.numValueBlocks() == 0 and d.size() == 1024
    d.clear()
    assert d.size() == 0 and d.pCoords() is None


def test_compute_fails_loudly_without_a_device():
    lib = _lib.load_library()
    if lib.hns_device_count() > 0:
        pytest.skip("a HIP device is present; this test is for the CPU-only container")
    o = fields.dense_leaves(16)
    with pytest.raises(_lib.HNSError) as e:
        api.create_grid_from_leaves(o, 1.0)
    assert e.value.code == _lib.HNS_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    h = api.create_grid_from_leaves(o, 1.0 / 16, _lib.HNS_GRID_HOST_ONLY)
    f = fields.synthetic_fields(o, 16)
    d = api.GridIndexedData()
    d.allocateCoords(len(f["density"]))
    d.pCoords()[:] = fields.leaves_to_coords(o)
    for n in ("density", "temperature", "fuel", "waste", "flame"):
        d.addValueBlock(n, d.FLOAT)
        d.pValues(n)[:] = f[n]
    d.addValueBlock("vel", d.VEC3F)
    d.pValues("vel")[:] = f["vel"]
    before = d.pValues("vel").copy()
    for call in (
        lambda: api.Compute_Sim(d, h, 5, 0.04, 1.0 / 16, api.CombustionParams(), False),
        lambda: api.ProjectNonDivergent(d, 5, 1.0 / 16, handle=h),
        lambda: api.AdvectIndexGrid(d, 0.04, 1.0 / 16, handle=h),
        lambda: api.AdvectIndexGridVelocity(d, 0.04, 1.0 / 16, handle=h),
    ):
        with pytest.raises(_lib.HNSError) as e:
            call()
        assert e.value.code == _lib.HNS_ERR_NO_DEVICE
    assert np.array_equal(d.pValues("vel"), before)
    # argument validation still comes first, as in the reference (HNanoSolver.cu:12-23)
    with pytest.raises(ValueError, match="voxelSize must be positive"):
        api.Compute_Sim(d, h, 5, 0.04, -1.0, api.CombustionParams(), False)


def test_synthetic_configurations():
    assert len(fields.dense_leaves(64)) == 512 and len(fields.dense_leaves(128)) == 4096
    n = len(fields.plume_leaves(32, 2.5, 0.22))
    assert abs(n - 4096) <= 0.05 * 4096  # BASELINE.json: ~4k leaves / ~2M voxels
    o = fields.dense_leaves(32)
    assert np.array_equal(o, o[fields.nanovdb_order(o)])
    f = fields.synthetic_fields(o[:8], 32)
    cfl = np.abs(f["vel"]).max() * 32 / 24
    assert f["vel"].shape == (4096, 3) and cfl < 6.0


def _build_host_mirror(tmp_path):
    exe = os.path.join(str(tmp_path), "host_mirror")
    libdir = os.path.dirname(_lib.library_path())
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "hnanosolver_amd", "host"),
           os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp"), "-o", exe, "-L", libdir, "-lhns", f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_cpp_host_mirror_builds_and_behaves(tmp_path):
    """hnanosolver_amd/host/HNanoSolver.hpp (the C++ twin of api.py) compiles with plain g++ against the C ABI and maps
    return codes to the reference's exception types."""
    exe = _build_host_mirror(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_header_is_plain_c99_and_links_from_c(tmp_path):
    """include/hns.h is the boundary a cgo / JNI / ctypes binding would consume: it must compile as strict C99 (no C++
    types in any signature) and a C program must link against libhns.so and call it."""
    src = tmp_path / "c_abi.c"
    src.write_text('#include "hns.h"\n#include <stdio.h>\nint main(void) {\n'
                   '  int err = 0; const int32_t o[3] = {0, 0, 0};\n'
                   '  hns_grid* g = hns_grid_create_from_leaves(o, 1, 0.5f, HNS_GRID_HOST_ONLY, &err);\n'
                   '  if (!g || err != HNS_OK || hns_grid_voxel_count(g) != 512) return 1;\n'
                   '  uint64_t off = 0; const int32_t q[3] = {1, 2, 3};\n'
                   '  if (hns_grid_offsets(g, q, 1, &off) != HNS_OK || off != 1 + 64 + 16 + 3) return 2;\n'
                   '  hns_grid_destroy(g);\n  printf("c abi ok %d\\n", hns_version());\n  return 0;\n}\n')
    exe = str(tmp_path / "c_abi")
    libdir = os.path.dirname(_lib.library_path())
    b = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-L" + libdir, "-lhns",
                        "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "c abi ok 100" in r.stdout, r.stdout + r.stderr


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    """hns_topology.cpp + hns_nanovdb.cpp + hns_leafio.cpp (all of libhns that runs on the host without a device) compiled with
    -fsanitize=address,undefined and driven through int32-edge origins, threaded validation, malformed inputs and
    exact-size export buffers (tests/cpp/host_sanitize.cpp). GPU sanitizers are not available on the target pool."""
    rocm_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(rocm_inc, "hip", "hip_runtime.h")):
        pytest.skip("HIP headers not found")
    exe = str(tmp_path / "host_sanitize")
    csrc = os.path.join(ROOT, "hnanosolver_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I" + rocm_inc, "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
           os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), os.path.join(csrc, "hns_topology.cpp"), os.path.join(csrc, "hns_nanovdb.cpp"), os.path.join(csrc, "hns_leafio.cpp"),
           "-pthread", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert r.returncode == 0 and "host_sanitize OK" in r.stdout, r.stdout + r.stderr[-3000:]


@pytest.mark.gpu
def test_cpp_host_mirror_matches_python_path_on_gpu(tmp_path):
    exe = _build_host_mirror(tmp_path)
    o = fields.dense_leaves(32)
    f = fields.synthetic_fields(o, 32)
    c = fields.leaves_to_coords(o)
    inp, outp = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    with open(inp, "wb") as fh:
        fh.write(np.int64(len(c)).tobytes())
        fh.write(c.tobytes())
        fh.write(f["vel"].tobytes())
        fh.write(f["density"].tobytes())
    r = subprocess.run([exe, "gpu", inp, outp], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(outp, dtype=np.float32)
    vel, den = raw[: 3 * len(c)].reshape(-1, 3), raw[3 * len(c):]
    d = api.GridIndexedData()
    d.allocateCoords(len(c))
    d.pCoords()[:] = c
    d.addValueBlock("density", d.FLOAT)
    d.addValueBlock("vel", d.VEC3F)
    d.pValues("density")[:] = f["density"]
    d.pValues("vel")[:] = f["vel"]
    api.ProjectNonDivergent(d, 20, 1.0 / 32)
    api.AdvectIndexGrid(d, 1.0 / 24, 1.0 / 32)
    assert np.array_equal(vel, d.pValues("vel")) and np.array_equal(den, d.pValues("density"))
