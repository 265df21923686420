"""Host-side mirror of the OpenVDB-free steps either side of the path (``hns_gather_leaves`` ... in include/hns.h): what
the reference's ``HNS::IndexGridBuilder`` (src/Utils/GridBuilder.hpp:87-216) and the HNanoSolver SOP's domain dilation
(src/SOP/HNanoSolver/SOP_HNanoSolver.cpp:186-199) do to OpenVDB trees, over raw 8^3 leaf buffers. PARITY UNPINNED: OpenVDB is
absent from the build image; the tests check these against brute force."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import lib

FILL_ZERO, FILL_SDF = 0, 1


def _o(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 3)


def gather_leaves(domain_origins, src_origins, src_values, ncomp: int = 1, fill: int = FILL_ZERO) -> np.ndarray:
    d, s = _o(domain_origins), _o(src_origins)
    v = np.ascontiguousarray(src_values, dtype=np.float32)
    assert v.size == len(s) * 512 * ncomp
    out = np.empty((len(d) * 512, ncomp) if ncomp == 3 else (len(d) * 512,), dtype=np.float32)
    _lib.check(lib.hns_gather_leaves(d.ctypes.data, len(d), s.ctypes.data, len(s), v.ctypes.data, ncomp, fill, out.ctypes.data))
    return out


def scatter_leaves(flat, n_domain: int, ncomp: int = 1):
    """-> list of per-leaf buffers (what writeIndexGrid copies into every leaf of the output grid)"""
    f = np.ascontiguousarray(flat, dtype=np.float32)
    bufs = [np.empty(512 * ncomp, dtype=np.float32) for _ in range(n_domain)]
    ptrs = (C.c_void_p * max(1, n_domain))(*[b.ctypes.data for b in bufs])
    _lib.check(lib.hns_scatter_leaves(f.ctypes.data, n_domain, ncomp, ptrs))
    return bufs


def dilate_leaves(origins, padding_voxels: int, active_masks: Optional[np.ndarray] = None) -> np.ndarray:
    o = _o(origins)
    m = None if active_masks is None else np.ascontiguousarray(active_masks, dtype=np.uint8).reshape(len(o), 64)
    n = C.c_uint64(0)
    _lib.check(lib.hns_dilate_leaves(o.ctypes.data, len(o), m.ctypes.data if m is not None else None, int(padding_voxels), None, 0, C.byref(n)))
    out = np.zeros((n.value, 3), dtype=np.int32)
    _lib.check(lib.hns_dilate_leaves(o.ctypes.data, len(o), m.ctypes.data if m is not None else None, int(padding_voxels), out.ctypes.data, n.value, C.byref(n)))
    return out


def union_leaves(a, b) -> np.ndarray:
    a, b = _o(a), _o(b)
    n = C.c_uint64(0)
    _lib.check(lib.hns_union_leaves(a.ctypes.data, len(a), b.ctypes.data, len(b), None, 0, C.byref(n)))
    out = np.zeros((n.value, 3), dtype=np.int32)
    _lib.check(lib.hns_union_leaves(a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data, n.value, C.byref(n)))
    return out
