"""Host-side mirror of the reference's operator interface for the substep hot path, over the C ABI of libhns.so.

Same names, argument meaning and error behaviour as the reference's entry points:

=========================  =========================================================================
here                       reference (paths relative to the reference checkout)
=========================  =========================================================================
``GridIndexedData``        ``HNS::GridIndexedData``    src/Utils/GridData.hpp:16-166
``CombustionParams``       ``CombustionParams``         src/Cuda/Kernels.cuh:6-13
``CreateIndexGrid``        ``CreateIndexGrid``          src/Cuda/HNanoSolver.cu:375-390
``Compute_Sim``            ``Compute_Sim``              src/Cuda/HNanoSolver.cu:9-372,393-396
``AdvectIndexGrid``        ``AdvectIndexGrid``          src/Cuda/Advection.cu:13-112,169-171
``AdvectIndexGridVelocity````AdvectIndexGridVelocity``  src/Cuda/Advection.cu:114-166,173-175
``ProjectNonDivergent``    ``ProjectNonDivergent``      src/Cuda/PressureProjection.cu:9-78,132-135
``Divergence``             ``Divergence``               src/Cuda/PressureProjection.cu:81-129
=========================  =========================================================================

Where the reference throws ``std::invalid_argument`` this raises ``ValueError``; ``std::runtime_error`` becomes
``RuntimeError`` (both carry the library's message). Host arrays are overwritten in place and every call is
synchronous, as in the reference. The C++ twin of this file is ``hnanosolver_amd/host/HNanoSolver.hpp``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from . import _lib
from ._lib import HNSError, hns_combustion_params, hns_field, lib


@dataclass
class CombustionParams:
    """reference src/Cuda/Kernels.cuh:6-13; defaults are the SOP parameter defaults (SOP_HNanoSolver.cpp:21-88)."""

    expansionRate: float = 0.1
    temperatureRelease: float = 0.5
    buoyancyStrength: float = 1.0
    ambientTemp: float = 23.0
    vorticityScale: float = 1.0
    factorScale: float = 0.5

    def _c(self) -> hns_combustion_params:
        return hns_combustion_params(self.expansionRate, self.temperatureRelease, self.buoyancyStrength, self.ambientTemp, self.vorticityScale, self.factorScale)


class GridIndexedData:
    """Named typed SoA blocks + one coordinate array, insertion order preserved (reference GridData.hpp:16-166).

    Block types: ``float`` -> numpy float32 ``(N,)``; ``Vec3f`` -> float32 ``(N, 3)`` (AoS, as ``openvdb::Vec3f[N]``).
    """

    FLOAT = "float"
    VEC3F = "Vec3f"

    def __init__(self) -> None:
        self._coords: Optional[np.ndarray] = None
        self._blocks: List[tuple] = []  # (name, type, array) in insertion order
        self._index: Dict[str, int] = {}
        self._size = 0

    # -- coords -------------------------------------------------------------------------------------------------
    def allocateCoords(self, numElements: int) -> bool:
        self._coords = np.zeros((int(numElements), 3), dtype=np.int32)
        self._size = int(numElements)
        return True

    def pCoords(self) -> Optional[np.ndarray]:
        return self._coords

    def size(self) -> int:
        return self._size

    # -- value blocks -------------------------------------------------------------------------------------------
    def addValueBlock(self, name: str, type_: str, numElements: Optional[int] = None) -> bool:
        if name in self._index:  # GridData.hpp:62-65
            return False
        n = self._size if numElements is None else int(numElements)
        if type_ == self.FLOAT:
            arr = np.zeros((n,), dtype=np.float32)
        elif type_ == self.VEC3F:
            arr = np.zeros((n, 3), dtype=np.float32)
        else:
            raise TypeError(f"unsupported block type {type_!r}")
        self._blocks.append((name, type_, arr))
        self._index[name] = len(self._blocks) - 1
        return True

    def pValues(self, name: str, type_: Optional[str] = None) -> Optional[np.ndarray]:
        k = self._index.get(name)
        if k is None:
            return None
        n, t, arr = self._blocks[k]
        if type_ is not None and t != type_:  # type mismatch -> nullptr (GridData.hpp:84-86)
            return None
        return arr

    def getBlocksOfType(self, type_: str) -> List[str]:
        return [n for (n, t, _) in self._blocks if t == type_]  # insertion order (GridData.hpp:136-145)

    def numValueBlocks(self) -> int:
        return len(self._blocks)

    def clear(self) -> None:
        self.clearValues()
        self._coords = None
        self._size = 0

    def clearValues(self) -> None:
        self._blocks.clear()
        self._index.clear()

    # -- marshalling --------------------------------------------------------------------------------------------
    def _fields(self):
        """ctypes hns_field[] over the blocks, in insertion order; keeps the arrays alive through ``keep``."""
        arr_t = hns_field * max(1, len(self._blocks))
        out = arr_t()
        keep = []
        for i, (name, t, a) in enumerate(self._blocks):
            if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
                raise TypeError(f"block {name!r} must be C-contiguous float32")
            bname = name.encode()
            keep.append(bname)
            out[i].name = bname
            out[i].ncomp = 3 if t == self.VEC3F else 1
            out[i].host = a.ctypes.data_as(C.POINTER(C.c_float))
        return out, len(self._blocks), keep


class IndexGridHandle:
    """Owns an ``hns_grid`` (the role of ``nanovdb::GridHandle<DeviceBuffer>`` in the reference)."""

    def __init__(self, ptr: int = 0):
        self._ptr = ptr

    def isEmpty(self) -> bool:
        return not self._ptr

    @property
    def ptr(self) -> int:
        return self._ptr

    def reset(self) -> None:
        if self._ptr:
            lib.hns_grid_destroy(self._ptr)
            self._ptr = 0

    def leaf_count(self) -> int:
        return int(lib.hns_grid_leaf_count(self._ptr)) if self._ptr else 0

    def voxel_count(self) -> int:
        return int(lib.hns_grid_voxel_count(self._ptr)) if self._ptr else 0

    def offsets(self, ijk: np.ndarray) -> np.ndarray:
        ijk = np.ascontiguousarray(ijk, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((ijk.shape[0],), dtype=np.uint64)
        _raise(lib.hns_grid_offsets(self._ptr, ijk.ctypes.data, ijk.shape[0], out.ctypes.data))
        return out

    def neighbor_table(self) -> np.ndarray:
        out = np.zeros((self.leaf_count(), 27), dtype=np.int32)
        _raise(lib.hns_grid_neighbor_table(self._ptr, out.ctypes.data))
        return out

    def coords(self) -> np.ndarray:
        out = np.zeros((self.voxel_count(), 3), dtype=np.int32)
        _raise(lib.hns_grid_coords(self._ptr, out.ctypes.data))
        return out

    def set_active_leaves(self, n: int) -> None:
        _raise(lib.hns_grid_set_active_leaves(self._ptr, int(n)))

    def set_active_range(self, first: int, count: int) -> None:
        _raise(lib.hns_grid_set_active_range(self._ptr, int(first), int(count)))

    def export_nanovdb(self) -> np.ndarray:
        """The grid as a NanoVDB ``NanoGrid<ValueOnIndex>`` buffer (uint8 array, 32-byte aligned), the reference's own
        index-grid format (HNanoSolver.cu:375-384)."""
        size = C.c_uint64(0)
        _raise(lib.hns_grid_export_nanovdb(self._ptr, None, 0, C.byref(size)))
        raw = np.zeros(size.value + 32, dtype=np.uint8)
        shift = (-raw.ctypes.data) % 32
        buf = raw[shift:shift + size.value]
        _raise(lib.hns_grid_export_nanovdb(self._ptr, buf.ctypes.data, size.value, C.byref(size)))
        return buf

    def release_cache(self) -> None:
        """Free the device buffers operator calls keep with the grid between cooks."""
        _raise(lib.hns_grid_release_cache(self._ptr))

    def active_leaves(self) -> int:
        return int(lib.hns_grid_active_leaves(self._ptr)) if self._ptr else 0

    def launch_order(self) -> np.ndarray:
        """sched: the leaf each workgroup of a one-workgroup-per-leaf kernel works on, in launch order (a copy of the device-built table; inspection / tests)."""
        sched = np.zeros((self.active_leaves(),), dtype=np.int32)
        _raise(lib.hns_grid_launch_tables(self._ptr, sched.ctypes.data))
        return sched

    def set_outside_element(self, element_index: int) -> None:
        _raise(lib.hns_grid_set_outside_element(self._ptr, int(element_index)))

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass


def _raise(code: int) -> None:
    """HNS_ERR_INVALID_ARGUMENT -> ValueError (std::invalid_argument); anything else negative -> RuntimeError."""
    if code >= 0:
        return
    msg = _lib.load_library().hns_last_error().decode("utf-8", "replace")
    if code == _lib.HNS_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    raise HNSError(code, msg)


def _stream(stream) -> int:
    if stream is None:
        return 0
    return int(getattr(stream, "cuda_stream", stream))


def create_grid_from_leaves(leaf_origins: np.ndarray, voxel_size: float = 1.0, flags: int = _lib.HNS_GRID_DEFAULT) -> IndexGridHandle:
    o = np.ascontiguousarray(leaf_origins, dtype=np.int32).reshape(-1, 3)
    err = C.c_int(0)
    ptr = lib.hns_grid_create_from_leaves(o.ctypes.data, o.shape[0], float(voxel_size), flags, C.byref(err))
    if not ptr:
        _raise(err.value if err.value < 0 else _lib.HNS_ERR_RUNTIME)
    return IndexGridHandle(ptr)


def CreateIndexGrid(data: GridIndexedData, handle: IndexGridHandle, voxelSize: float, flags: int = _lib.HNS_GRID_DEFAULT) -> None:
    """Build the index grid for ``data.pCoords()`` into ``handle`` (reference HNanoSolver.cu:375-390).

    Persistent state across cooks: a handle that already holds a grid for exactly these leaves, in
    this order and at this voxel size, is kept -- with the device buffers the operators left with it -- instead of
    being rebuilt; the coordinates are validated either way."""
    coords = data.pCoords()
    if coords is None:
        raise RuntimeError("Host coordinate data pointer is null.")
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    if not handle.isEmpty() and float(lib.hns_grid_voxel_size(handle.ptr)) == float(np.float32(voxelSize)) \
            and not (flags & _lib.HNS_GRID_HOST_ONLY):
        same = lib.hns_grid_matches(handle.ptr, coords.ctypes.data, coords.shape[0], flags)
        if same < 0:
            _raise(same)
        if same == 1:
            return
    err = C.c_int(0)
    ptr = lib.hns_grid_create(coords.ctypes.data, coords.shape[0], float(voxelSize), flags, C.byref(err))
    if not ptr and err.value < 0:
        _raise(err.value)
    handle.reset()
    handle._ptr = ptr


def Compute_Sim(data: GridIndexedData, handle: IndexGridHandle, iteration: int, dt: float, voxelSize: float, params: CombustionParams,
                hasCollision: bool, stream=None, feedback=None, checked: bool = False) -> Optional[int]:
    """feedback (not in the reference's signature): True, or the names of the blocks whose arrays still hold what the previous
    Compute_Sim on this handle handed back (the SOP feeds its output back in as the next frame's input, SOP_HNanoSolver.cpp:106):
    those blocks are not uploaded again (hns_compute_sim_resident). checked=False (default): VOUCHED -- the caller knows what it changed; only a
    4,096-sample signature is compared, a sparse edit (an emitter added to a few leaves) is NOT detected, so a block the caller sourced
    into must not be named. checked=True: CHECKED -- a digest of every element (taken on the device when a block is handed back, on host threads when
    it comes in again); an edit is noticed and that block uploaded (a digest, not a comparison: missed with probability ~2^-64); 14.8 ms per cook at 256^3 against 19.8 for the plain warm cook and 11.9
    vouched (profiles/r05_final_cook256.json). Returns the number of uploads skipped."""
    if handle is None or handle.isEmpty():
        # argument checks come first in the reference (HNanoSolver.cu:12-23)
        if voxelSize <= 0.0:
            raise ValueError("voxelSize must be positive.")
        if dt < 0.0:
            raise ValueError("dt (time step) cannot be negative.")
        if iteration <= 0:
            raise ValueError("Number of pressure iterations must be positive.")
        raise ValueError("Invalid grid handle provided (null grid).")
    fields, n, keep = data._fields()
    p = params._c()
    if feedback:
        flags = bytes((2 if checked else 1) if (feedback is True or fields[i].name.decode() in feedback) else 0 for i in range(n))
        skipped = C.c_int(0)
        _raise(lib.hns_compute_sim_resident(handle.ptr, fields, n, flags, C.byref(skipped), int(iteration), float(dt), float(voxelSize), C.byref(p),
                                            int(bool(hasCollision)), _stream(stream)))
        return skipped.value
    _raise(lib.hns_compute_sim(handle.ptr, fields, n, int(iteration), float(dt), float(voxelSize), C.byref(p), int(bool(hasCollision)), _stream(stream)))
    return None


def _grid_for(data: GridIndexedData, voxelSize: float) -> IndexGridHandle:
    # the reference rebuilds the index grid from data.pCoords() inside these operators
    # (Advection.cu:71,142; PressureProjection.cu:38,108)
    h = IndexGridHandle()
    CreateIndexGrid(data, h, voxelSize)
    return h


def AdvectIndexGrid(data: GridIndexedData, dt: float, voxelSize: float, stream=None, handle: Optional[IndexGridHandle] = None) -> None:
    h = handle or _grid_for(data, voxelSize)
    fields, n, keep = data._fields()
    _raise(lib.hns_advect_index_grid(h.ptr, fields, n, float(dt), float(voxelSize), _stream(stream)))


def AdvectIndexGridVelocity(data: GridIndexedData, dt: float, voxelSize: float, stream=None, handle: Optional[IndexGridHandle] = None) -> None:
    h = handle or _grid_for(data, voxelSize)
    fields, n, keep = data._fields()
    _raise(lib.hns_advect_index_grid_velocity(h.ptr, fields, n, float(dt), float(voxelSize), _stream(stream)))


def ProjectNonDivergent(data: GridIndexedData, iterations: int, voxelSize: float, stream=None, handle: Optional[IndexGridHandle] = None) -> None:
    h = handle or _grid_for(data, voxelSize)
    fields, n, keep = data._fields()
    _raise(lib.hns_project_non_divergent(h.ptr, fields, n, int(iterations), float(voxelSize), _stream(stream)))


def Divergence(data: GridIndexedData, voxelSize: float, stream=None, handle: Optional[IndexGridHandle] = None) -> None:
    h = handle or _grid_for(data, voxelSize)
    fields, n, keep = data._fields()
    _raise(lib.hns_divergence(h.ptr, fields, n, float(voxelSize), _stream(stream)))
