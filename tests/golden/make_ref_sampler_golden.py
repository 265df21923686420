#!/usr/bin/env python3
"""Generates tests/golden/ref_samplers_v1.npz: inputs and the outputs of the REFERENCE's own sampler code.

Needs oracle/_ref/libhns_ref.so, i.e. the reference checkout at /root/reference (oracle/Makefile builds it from
src/Utils/Stencils.hpp + the vendored NanoVDB, compiled where they lie). The fixture is data only: leaf origins,
seeded field values, query points, and what IndexOffsetSampler<0> / IndexSampler<T,0|1> returned for them.
Run from the repo root:  python tests/golden/make_ref_sampler_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hnanosolver_amd import fields  # noqa: E402
from oracle_lib import reference_samplers  # noqa: E402


def main():
    R = reference_samplers()
    assert R is not None, "needs /root/reference to build oracle/_ref"
    rng = np.random.default_rng(20250704)
    lat = np.stack(np.meshgrid(*[np.arange(-2, 3)] * 3, indexing="ij"), -1).reshape(-1, 3)
    o = (lat[rng.random(len(lat)) < 0.35] * 8).astype(np.int32)
    o = np.concatenate([o, np.array([[-4104, 0, 0], [4096, 8, -16], [-8, -4096, 8]], dtype=np.int32)])
    o = np.ascontiguousarray(o[rng.permutation(len(o))])  # deliberately NOT in NanoVDB order
    g = R.ref_grid_create(o.ctypes.data, len(o))
    nl = int(R.ref_leaf_count(g))
    ref_order = np.zeros((nl, 3), np.int32)
    R.ref_leaf_origins(g, ref_order.ctypes.data)
    N = nl * 512
    f = rng.standard_normal(N).astype(np.float32)
    v = rng.standard_normal((N, 3)).astype(np.float32)
    c = fields.leaves_to_coords(ref_order)
    ijk = np.concatenate([c[rng.integers(0, N, 3000)] + rng.integers(-9, 10, (3000, 3)), rng.integers(-5000, 5000, (500, 3))]).astype(np.int32)
    xyz = (c[rng.integers(0, N, 4000)] + rng.uniform(-6, 6, (4000, 3))).astype(np.float32)
    off = np.zeros(len(ijk), np.uint64)
    R.ref_offsets(g, ijk.ctypes.data, len(ijk), off.ctypes.data)
    nf = np.zeros(len(ijk), np.float32)
    R.ref_sample_nearest_f(g, f.ctypes.data, ijk.ctypes.data, len(ijk), nf.ctypes.data)
    tf = np.zeros(len(xyz), np.float32)
    R.ref_sample_trilinear_f(g, f.ctypes.data, xyz.ctypes.data, len(xyz), tf.ctypes.data)
    tv = np.zeros((len(xyz), 3), np.float32)
    R.ref_sample_trilinear_v(g, v.ctypes.data, xyz.ctypes.data, len(xyz), tv.ctypes.data)
    vc = int(R.ref_value_count(g))
    R.ref_grid_destroy(g)
    out = os.path.join(ROOT, "tests", "golden", "ref_samplers_v1.npz")
    np.savez_compressed(out, input_origins=o, ref_leaf_order=ref_order, value_count=np.int64(vc), f=f, v=v, ijk=ijk, xyz=xyz,
                        offsets=off, nearest_f=nf, trilinear_f=tf, trilinear_v_host=tv)
    print("wrote", out, os.path.getsize(out), "bytes; leaves", nl)


if __name__ == "__main__":
    main()
