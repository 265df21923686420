"""Closed-form synthetic inputs and leaf sets for the benchmark / parity configurations.

Everything here is input generation: leaf-origin tables in NanoVDB order and smooth analytic fields evaluated per
voxel. Nothing in this module computes any part of the solver.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

LEAF = 8
LEAF_VOXELS = 512


def nanovdb_order(origins: np.ndarray) -> np.ndarray:
    """Permutation that sorts 8-aligned leaf origins the way NanoVDB's voxelsToGrid orders leaves: signed root-tile key
    (coord >> 12, x major), then child offset inside the 4096^3 upper node, then inside the 128^3 lower node
    (reference externals/nanovdb/tools/cuda/PointsToGrid.cuh:596-602,640-645)."""
    o = np.asarray(origins, dtype=np.int64).reshape(-1, 3)
    tile = o >> 12
    up = ((o & 4095) >> 7)
    lo = ((o & 127) >> 3)
    upper = (up[:, 0] << 10) | (up[:, 1] << 5) | up[:, 2]
    lower = (lo[:, 0] << 8) | (lo[:, 1] << 4) | lo[:, 2]
    return np.lexsort((lower, upper, tile[:, 2], tile[:, 1], tile[:, 0]))


def dense_leaves(R: int) -> np.ndarray:
    """Leaf origins of the dense R^3 cube [0,R)^3 in NanoVDB order. R must be a multiple of 8."""
    assert R % LEAF == 0
    n = R // LEAF
    l = np.arange(n, dtype=np.int32) * LEAF
    o = np.stack(np.meshgrid(l, l, l, indexing="ij"), axis=-1).reshape(-1, 3)
    return np.ascontiguousarray(o[nanovdb_order(o)], dtype=np.int32)


def plume_leaves(lattice: int = 32, a: float = 2.5, b: float = 0.22) -> np.ndarray:
    """Sparse rising-plume leaf set: on a lattice^3 leaf lattice keep leaf (lx,ly,lz) iff
    (lx-c)^2 + (lz-c)^2 <= (a + b*ly)^2 with c = lattice/2 (a widening cone along +y)."""
    c = lattice // 2
    l = np.arange(lattice, dtype=np.int64)
    lx, ly, lz = np.meshgrid(l, l, l, indexing="ij")
    keep = (lx - c) ** 2 + (lz - c) ** 2 <= (a + b * ly) ** 2
    o = (np.stack([lx[keep], ly[keep], lz[keep]], axis=-1) * LEAF).astype(np.int32)
    return np.ascontiguousarray(o[nanovdb_order(o)], dtype=np.int32)


def leaves_to_coords(origins: np.ndarray) -> np.ndarray:
    """The leaf-dense coordinate array the reference's IndexGridBuilder emits (GridBuilder.hpp:156-166):
    for each leaf, offsetToGlobalCoord(n) for n = x<<6 | y<<3 | z."""
    o = np.asarray(origins, dtype=np.int32).reshape(-1, 1, 3)
    n = np.arange(LEAF_VOXELS, dtype=np.int32)
    local = np.stack([n >> 6, (n >> 3) & 7, n & 7], axis=-1).reshape(1, LEAF_VOXELS, 3)
    return np.ascontiguousarray((o + local).reshape(-1, 3))


def config_leaves(name: str) -> Tuple[np.ndarray, int]:
    """Leaf set and extent R (voxelSize = 1/R) of a BASELINE.json configuration."""
    if name in ("64", "128", "256", "512"):
        R = int(name)
        return dense_leaves(R), R
    if name == "plume":  # ~4k leaves / ~2M voxels on a 256^3 extent
        return plume_leaves(32, 2.5, 0.22), 256
    if name == "plume1024":  # ~64k leaves on a 1024^3 extent
        return plume_leaves(128, 4.0, 0.125), 1024
    raise KeyError(name)


def synthetic_fields(origins: np.ndarray, R: int, amplitude_voxels: float = 96.0, chunk_leaves: int = 2048) -> Dict[str, np.ndarray]:
    """SURVEY.md 8d inputs on the given leaves: q = (ijk+0.5)/R,
    blob = exp(-|q-(0.5,0.2,0.5)|^2/0.02), velocity = A*(0.5 sin(2 pi qy) cos(2 pi qz), blob + 0.25 sin(2 pi qx),
    0.5 cos(2 pi qx) sin(2 pi qy)) with A = amplitude_voxels/R world units/s, density = blob,
    temperature = 23 + 50 blob, fuel = 0.2 blob, waste = flame = 0. Evaluated in float64, stored float32."""
    origins = np.asarray(origins, dtype=np.int32).reshape(-1, 3)
    nl = origins.shape[0]
    N = nl * LEAF_VOXELS
    out = {
        "vel": np.empty((N, 3), dtype=np.float32),
        "density": np.empty((N,), dtype=np.float32),
        "temperature": np.empty((N,), dtype=np.float32),
        "fuel": np.empty((N,), dtype=np.float32),
        "waste": np.zeros((N,), dtype=np.float32),
        "flame": np.zeros((N,), dtype=np.float32),
    }
    A = amplitude_voxels / R
    two_pi = 2.0 * np.pi
    for l0 in range(0, nl, chunk_leaves):
        l1 = min(nl, l0 + chunk_leaves)
        c = leaves_to_coords(origins[l0:l1]).astype(np.float64)
        q = (c + 0.5) / R
        qx, qy, qz = q[:, 0], q[:, 1], q[:, 2]
        blob = np.exp(-((qx - 0.5) ** 2 + (qy - 0.2) ** 2 + (qz - 0.5) ** 2) / 0.02)
        s = slice(l0 * LEAF_VOXELS, l1 * LEAF_VOXELS)
        out["vel"][s, 0] = A * 0.5 * np.sin(two_pi * qy) * np.cos(two_pi * qz)
        out["vel"][s, 1] = A * (1.0 * blob + 0.25 * np.sin(two_pi * qx))
        out["vel"][s, 2] = A * 0.5 * np.cos(two_pi * qx) * np.sin(two_pi * qy)
        out["density"][s] = blob
        out["temperature"][s] = 23.0 + 50.0 * blob
        out["fuel"][s] = 0.2 * blob
    return out


def sphere_sdf(origins: np.ndarray, R: int, center=(0.5, 0.45, 0.5), radius: float = 0.12) -> np.ndarray:
    """Signed distance (in VOXELS, as the reference's collision thresholds 0.1/1.5 assume) to a sphere, for collision cases."""
    c = leaves_to_coords(origins).astype(np.float64)
    d = np.sqrt(((c + 0.5) / R - np.asarray(center)) ** 2 @ np.ones(3)) - radius
    return (d * R).astype(np.float32)
