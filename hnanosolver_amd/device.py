"""Kernel-level calls of libhns.so on device-resident torch tensors.

PyTorch is plumbing here: it owns device memory and streams; every computation is a HIP kernel of libhns.so reached
through the C ABI (``hns_dev_*`` / ``hns_sim_*`` in include/hns.h). Velocity tensors are ``(N, 3)`` float32 (Vec3f AoS, the
host layout). There is no CPU path: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import hns_combustion_params, hns_field, lib
from .api import CombustionParams, IndexGridHandle, _raise, create_grid_from_leaves  # noqa: F401


def _torch():
    import torch

    return torch


def _ptr(t) -> int:
    if t is None:
        return 0
    if not t.is_cuda:
        raise RuntimeError("hnanosolver_amd.device: tensor is not on a HIP device (there is no CPU fallback)")
    if t.dtype != _torch().float32 and t.dtype != _torch().int32:
        raise TypeError("expected float32 / int32 tensor")
    if not t.is_contiguous():
        raise TypeError("expected a contiguous tensor")
    return t.data_ptr()


def current_stream() -> int:
    return int(_torch().cuda.current_stream().cuda_stream)


def advect_vector(grid: IndexGridHandle, u, out, dt: float, inv_dx: float, sdf=None, has_collision: bool = False):
    _raise(lib.hns_dev_advect_vector(grid.ptr, _ptr(u), _ptr(out), _ptr(sdf), int(has_collision), dt, inv_dx, current_stream()))
    return out


def advect_scalar(grid: IndexGridHandle, u, src, dst, dt: float, inv_dx: float, sdf=None, has_collision: bool = False):
    _raise(lib.hns_dev_advect_scalar(grid.ptr, _ptr(u), _ptr(src), _ptr(dst), _ptr(sdf), int(has_collision), dt, inv_dx, current_stream()))
    return dst


def advect_scalars(grid: IndexGridHandle, u, srcs: Sequence, dsts: Sequence, dt: float, inv_dx: float, sdf=None, has_collision: bool = False):
    n = len(srcs)
    ins = (C.c_void_p * max(1, n))(*[_ptr(t) for t in srcs])
    outs = (C.c_void_p * max(1, n))(*[_ptr(t) for t in dsts])
    _raise(lib.hns_dev_advect_scalars(grid.ptr, _ptr(u), ins, outs, n, _ptr(sdf), int(has_collision), dt, inv_dx, current_stream()))
    return dsts


def divergence(grid: IndexGridHandle, u, div, inv_dx: float):
    _raise(lib.hns_dev_divergence(grid.ptr, _ptr(u), _ptr(div), inv_dx, current_stream()))
    return div


def rbgs_color(grid: IndexGridHandle, div, p, dx: float, omega: float, color: int):
    _raise(lib.hns_dev_rbgs_color(grid.ptr, _ptr(div), _ptr(p), dx, omega, color, current_stream()))
    return p


def rbgs_iterate(grid: IndexGridHandle, div, p_a, p_b, dx: float, omega: float, iterations: int):
    """Returns the tensor (p_a or p_b) that holds the result."""
    in_b = C.c_int(0)
    _raise(lib.hns_dev_rbgs_iterate(grid.ptr, _ptr(div), _ptr(p_a), _ptr(p_b), dx, omega, iterations, C.byref(in_b), current_stream()))
    return p_b if in_b.value else p_a


def rbgs_plan(grid: IndexGridHandle, iterations: int):
    """(description of the SOR kernel form this grid is swept with, kernel launches for `iterations`, iterations per launch)"""
    buf = C.create_string_buffer(256)
    n, k = C.c_int(0), C.c_int(0)
    _raise(lib.hns_grid_rbgs_plan(grid.ptr, iterations, buf, 256, C.byref(n), C.byref(k)))
    return buf.value.decode(), n.value, k.value


def time_rbgs(grid: IndexGridHandle, div, p_a, p_b, dx: float, omega: float, iterations: int, reps: int) -> float:
    """Mean milliseconds per fused-iteration launch, measured with hipEvents on the launch stream."""
    ms = C.c_float(0.0)
    _raise(lib.hns_dev_time_rbgs(grid.ptr, _ptr(div), _ptr(p_a), _ptr(p_b), dx, omega, iterations, reps, C.byref(ms), current_stream()))
    return float(ms.value)


def subtract_pressure_gradient(grid: IndexGridHandle, u, p, out, inv_dx: float, sdf=None, has_collision: bool = False):
    _raise(lib.hns_dev_subtract_pressure_gradient(grid.ptr, _ptr(u), _ptr(p), _ptr(out), _ptr(sdf), int(has_collision), inv_dx, current_stream()))
    return out


def combustion_oxygen(fuel, waste, temperature, div, flame, out_fuel, out_waste, out_temperature, out_flame, temp_gain: float, expansion: float):
    _raise(lib.hns_dev_combustion_oxygen(_ptr(fuel), _ptr(waste), _ptr(temperature), _ptr(div), _ptr(flame), _ptr(out_fuel), _ptr(out_waste),
                                         _ptr(out_temperature), _ptr(out_flame), temp_gain, expansion, fuel.numel(), current_stream()))


def temperature_buoyancy(u, temperature, out, dt: float, ambient: float, strength: float):
    _raise(lib.hns_dev_temperature_buoyancy(_ptr(u), _ptr(temperature), _ptr(out), dt, ambient, strength, temperature.numel(), current_stream()))
    return out


def vorticity_confinement(grid: IndexGridHandle, u, out, dt: float, inv_dx: float, scale: float, factor_scale: float):
    _raise(lib.hns_dev_vorticity_confinement(grid.ptr, _ptr(u), _ptr(out), dt, inv_dx, scale, factor_scale, current_stream()))
    return out


def enforce_collision_boundaries(grid: IndexGridHandle, u, sdf, voxel_size: float):
    _raise(lib.hns_dev_enforce_collision_boundaries(grid.ptr, _ptr(u), _ptr(sdf), voxel_size, current_stream()))
    return u


def pack_leaves(field, leaf_ids, packed, ncomp: int = 1):
    _raise(lib.hns_dev_pack_leaves(_ptr(field), _ptr(leaf_ids), leaf_ids.numel(), _ptr(packed), ncomp, current_stream()))
    return packed


def unpack_leaves(packed, leaf_ids, field, ncomp: int = 1):
    _raise(lib.hns_dev_unpack_leaves(_ptr(packed), _ptr(leaf_ids), leaf_ids.numel(), _ptr(field), ncomp, current_stream()))
    return field


class Sim:
    """Device-resident simulation state (``hns_sim``): upload once, run many substeps."""

    def __init__(self, grid: IndexGridHandle, float_names: Sequence[str]):
        self.grid = grid
        self.names = list(float_names)
        arr = (C.c_char_p * max(1, len(self.names)))(*[n.encode() for n in self.names])
        err = C.c_int(0)
        self._ptr = lib.hns_sim_create(grid.ptr, arr, len(self.names), C.byref(err))
        if not self._ptr:
            _raise(err.value if err.value < 0 else _lib.HNS_ERR_RUNTIME)

    def _fields(self, arrays: dict):
        arr = (hns_field * max(1, len(arrays)))()
        keep = []
        for i, (name, a) in enumerate(arrays.items()):
            if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
                raise TypeError(f"{name}: need C-contiguous float32")
            b = name.encode()
            keep.append(b)
            arr[i].name = b
            arr[i].ncomp = 3 if (a.ndim == 2 and a.shape[1] == 3) else 1
            arr[i].host = a.ctypes.data_as(C.POINTER(C.c_float))
        return arr, len(arrays), keep

    def upload(self, arrays: dict, stream: Optional[int] = None) -> None:
        arr, n, keep = self._fields(arrays)
        _raise(lib.hns_sim_upload(self._ptr, arr, n, stream or 0))

    def download(self, arrays: dict, stream: Optional[int] = None) -> None:
        arr, n, keep = self._fields(arrays)
        _raise(lib.hns_sim_download(self._ptr, arr, n, stream or 0))

    def substep(self, iterations: int, dt: float, voxel_size: float, params: CombustionParams, has_collision: bool = False, stream: int = 0) -> None:
        p = params._c()
        _raise(lib.hns_sim_substep(self._ptr, iterations, dt, voxel_size, C.byref(p), int(has_collision), stream))

    def core_substep(self, iterations: int, dt: float, voxel_size: float, stream: int = 0) -> None:
        _raise(lib.hns_sim_core_substep(self._ptr, iterations, dt, voxel_size, stream))

    def pressure_solve(self, iterations: int, voxel_size: float, stream: int = 0) -> None:
        _raise(lib.hns_sim_pressure_solve(self._ptr, iterations, voxel_size, stream))

    def timing(self, max_solves: int) -> None:
        _raise(lib.hns_sim_timing(self._ptr, max_solves))

    def pressure_time(self):
        """(total ms of the event-bracketed pressure loops, number of fused-iteration launches inside them)"""
        ms, n = C.c_float(0.0), C.c_longlong(0)
        _raise(lib.hns_sim_pressure_time(self._ptr, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def stage_timing(self, max_substeps: int) -> None:
        _raise(lib.hns_sim_stage_timing(self._ptr, max_substeps))

    def stage_times(self):
        """({stage: total ms} over the core substeps timed since timing(), number of substeps)"""
        ms, n = (C.c_float * 5)(), C.c_longlong(0)
        _raise(lib.hns_sim_stage_times(self._ptr, ms, C.byref(n)))
        return dict(zip(("advect_vector", "divergence", "pressure", "gradient", "advect_scalars"), [float(x) for x in ms])), int(n.value)

    def close(self) -> None:
        if self._ptr:
            lib.hns_sim_destroy(self._ptr)
            self._ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
