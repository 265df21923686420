"""ctypes access to the CPU oracle (oracle/liboracle.so) and to the reference-backed sampler library
(oracle/_ref/libhns_ref.so). TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_ORC = None
_REF = None

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float


class orc_combustion_params(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("expansionRate", "temperatureRelease", "buoyancyStrength", "ambientTemp", "vorticityScale", "factorScale")]


def cpu_budget() -> int:
    """Threads the oracle should use: the CPUs this process may actually run on -- its affinity mask, further limited by a
    cgroup CPU quota if there is one. (OpenMP's default is every hardware thread of the host; on a 256-thread host under a
    16-CPU quota that default made one parallel region cost 250 ms instead of 0.5 ms.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _bind(L):
    sig = {
        "orc_grid_create": (_vp, [_vp, _i64]),
        "orc_grid_destroy": (None, [_vp]),
        "orc_grid_set_outside_element": (None, [_vp, C.c_uint64]),
        "orc_grid_leaf_count": (_i64, [_vp]),
        "orc_grid_voxel_count": (_i64, [_vp]),
        "orc_offset": (C.c_uint64, [_vp, C.c_int32, C.c_int32, C.c_int32]),
        "orc_coords": (None, [_vp, _vp]),
        "orc_set_vec3_lerp_fma": (None, [_i]),
        "orc_set_threads": (None, [_i]),
        "orc_get_threads": (_i, []),
        "orc_sample_nearest_f": (None, [_vp, _vp, _vp, _i64, _vp]),
        "orc_sample_trilinear_f": (None, [_vp, _vp, _vp, _i64, _vp]),
        "orc_sample_trilinear_v": (None, [_vp, _vp, _vp, _i64, _vp]),
        "orc_advect_vector": (None, [_vp, _vp, _vp, _vp, _i, _f, _f]),
        "orc_advect_scalar": (None, [_vp, _vp, _vp, _vp, _vp, _i, _f, _f]),
        "orc_advect_scalars": (None, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, _i, _f, _f]),
        "orc_divergence": (None, [_vp, _vp, _vp, _f]),
        "orc_rbgs": (None, [_vp, _vp, _vp, _f, _i, _f]),
        "orc_subtract_pressure_gradient": (None, [_vp, _vp, _vp, _vp, _vp, _i, _f]),
        "orc_combustion_oxygen": (None, [_vp] * 9 + [_f, _f, _i64]),
        "orc_temperature_buoyancy": (None, [_vp, _vp, _vp, _f, _f, _f, _i64]),
        "orc_vorticity_confinement": (None, [_vp, _vp, _vp, _f, _f, _f, _f]),
        "orc_enforce_collision_boundaries": (None, [_vp, _vp, _vp, _f]),
        "orc_omega_compute": (_f, [_f]),
        "orc_omega_project": (_f, [_f]),
        "orc_compute_sim": (_i, [_vp, _vp, C.POINTER(C.c_char_p), C.POINTER(_vp), _i, _i, _f, _f, C.POINTER(orc_combustion_params), _i]),
        "orc_project_non_divergent": (_i, [_vp, _vp, _i64, _f]),
        "orc_divergence_op": (_i, [_vp, _vp, _vp, _f]),
        "orc_advect_index_grid": (_i, [_vp, _vp, C.POINTER(_vp), _i, _f, _f]),
        "orc_advect_index_grid_velocity": (_i, [_vp, _vp, _f, _f]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L.orc_set_threads(cpu_budget())
    return L


def oracle():
    global _ORC
    if _ORC is not None:
        return _ORC
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, capture_output=True)
    _ORC = _bind(C.CDLL(path))
    return _ORC


def oracle_contracted():
    """The same source built the way nvcc builds the reference by default: floating-point contraction on (every a*b+c may
    fuse). Used only to measure how far a contracted build can drift from the strict one (tests/test_oracle_pins.py)."""
    path = os.path.join(ORACLE_DIR, "liboracle_fma.so")
    subprocess.run(["make", "-C", ORACLE_DIR, "oracle_fma"], check=True, capture_output=True)
    return _bind(C.CDLL(path))


def reference_samplers():
    """oracle/_ref/libhns_ref.so (the reference's Stencils.hpp + NanoVDB compiled for the host), or None if absent and
    the reference checkout is not available to build it."""
    global _REF
    if _REF is not None:
        return _REF
    path = os.path.join(ORACLE_DIR, "_ref", "libhns_ref.so")
    if not os.path.exists(path):
        if not os.path.exists("/root/reference/src/Utils/Stencils.hpp"):
            return None
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)
    L = C.CDLL(path)
    sig = {
        "ref_grid_create": (_vp, [_vp, _i64]),
        "ref_grid_create_from_voxels": (_vp, [_vp, _i64]),
        "ref_grid_destroy": (None, [_vp]),
        "ref_leaf_count": (_i64, [_vp]),
        "ref_value_count": (C.c_uint64, [_vp]),
        "ref_active_voxel_count": (C.c_uint64, [_vp]),
        "ref_leaf_origins": (None, [_vp, _vp]),
        "ref_offsets": (None, [_vp, _vp, _i64, _vp]),
        "ref_sample_nearest_f": (None, [_vp, _vp, _vp, _i64, _vp]),
        "ref_sample_trilinear_f": (None, [_vp, _vp, _vp, _i64, _vp]),
        "ref_sample_trilinear_v": (None, [_vp, _vp, _vp, _i64, _vp]),
        "ref_nanovdb_check": (C.c_int, [_vp, _vp, C.c_int]),
        "ref_nanovdb_query": (None, [_vp, _vp, _i64, _vp, _vp]),
        "ref_nanovdb_info": (None, [_vp, _vp, _vp, _vp]),
        "ref_nanovdb_leaves": (_i64, [_vp, _vp, _vp, _vp, _vp]),
        "ref_nanovdb_host_build": (C.c_uint64, [_vp, _i64, C.c_double, _vp, C.c_uint64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _REF = L
    return L


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


class OracleGrid:
    """Thin numpy front-end of the oracle. Vec3f arrays are (N,3) float32 AoS."""

    def __init__(self, leaf_origins, lib=None):
        self.L = lib or oracle()
        self.origins = np.ascontiguousarray(leaf_origins, dtype=np.int32).reshape(-1, 3)
        self.g = self.L.orc_grid_create(self.origins.ctypes.data, self.origins.shape[0])
        if not self.g:
            raise ValueError("oracle rejected the leaf set (duplicate or misaligned origin)")
        self.N = int(self.L.orc_grid_voxel_count(self.g))

    def __del__(self):
        try:
            if self.g:
                self.L.orc_grid_destroy(self.g)
                self.g = None
        except Exception:
            pass

    def set_outside_element(self, idx):
        self.L.orc_grid_set_outside_element(self.g, int(idx))

    def coords(self):
        out = np.zeros((self.N, 3), dtype=np.int32)
        self.L.orc_coords(self.g, out.ctypes.data)
        return out

    def offsets(self, ijk):
        ijk = np.ascontiguousarray(ijk, dtype=np.int32).reshape(-1, 3)
        return np.array([self.L.orc_offset(self.g, int(a), int(b), int(c)) for a, b, c in ijk], dtype=np.uint64)

    def sample_nearest_f(self, data, ijk):
        data = _f32(data)
        ijk = np.ascontiguousarray(ijk, dtype=np.int32).reshape(-1, 3)
        out = np.zeros(ijk.shape[0], dtype=np.float32)
        self.L.orc_sample_nearest_f(self.g, data.ctypes.data, ijk.ctypes.data, ijk.shape[0], out.ctypes.data)
        return out

    def sample_trilinear_f(self, data, xyz):
        data, xyz = _f32(data), _f32(xyz).reshape(-1, 3)
        out = np.zeros(xyz.shape[0], dtype=np.float32)
        self.L.orc_sample_trilinear_f(self.g, data.ctypes.data, xyz.ctypes.data, xyz.shape[0], out.ctypes.data)
        return out

    def sample_trilinear_v(self, data3, xyz):
        data3, xyz = _f32(data3), _f32(xyz).reshape(-1, 3)
        out = np.zeros((xyz.shape[0], 3), dtype=np.float32)
        self.L.orc_sample_trilinear_v(self.g, data3.ctypes.data, xyz.ctypes.data, xyz.shape[0], out.ctypes.data)
        return out

    # ---- kernels ----
    def advect_vector(self, vel, dt, inv_dx, sdf=None, has_collision=False):
        vel = _f32(vel)
        out = np.zeros_like(vel)
        sdf_ = _f32(sdf) if sdf is not None else None
        self.L.orc_advect_vector(self.g, vel.ctypes.data, out.ctypes.data, sdf_.ctypes.data if sdf_ is not None else None, int(has_collision), dt, inv_dx)
        return out

    def advect_scalar(self, vel, phi, dt, inv_dx, sdf=None, has_collision=False):
        vel, phi = _f32(vel), _f32(phi)
        out = np.zeros_like(phi)
        sdf_ = _f32(sdf) if sdf is not None else None
        self.L.orc_advect_scalar(self.g, vel.ctypes.data, phi.ctypes.data, out.ctypes.data, sdf_.ctypes.data if sdf_ is not None else None,
                                 int(has_collision), dt, inv_dx)
        return out

    def advect_scalars(self, vel, phis, dt, inv_dx, sdf=None, has_collision=False):
        vel = _f32(vel)
        phis = [_f32(p) for p in phis]
        outs = [np.zeros_like(p) for p in phis]
        n = len(phis)
        ins_ = (_vp * max(1, n))(*[p.ctypes.data for p in phis])
        outs_ = (_vp * max(1, n))(*[p.ctypes.data for p in outs])
        sdf_ = _f32(sdf) if sdf is not None else None
        self.L.orc_advect_scalars(self.g, vel.ctypes.data, ins_, outs_, n, sdf_.ctypes.data if sdf_ is not None else None, int(has_collision), dt, inv_dx)
        return outs

    def divergence(self, vel, inv_dx):
        vel = _f32(vel)
        out = np.zeros(self.N, dtype=np.float32)
        self.L.orc_divergence(self.g, vel.ctypes.data, out.ctypes.data, inv_dx)
        return out

    def rbgs(self, div, p, dx, color, omega):
        """in place on p"""
        assert p.dtype == np.float32 and p.flags["C_CONTIGUOUS"]
        div = _f32(div)
        self.L.orc_rbgs(self.g, div.ctypes.data, p.ctypes.data, dx, color, omega)
        return p

    def rbgs_iterations(self, div, dx, omega, iterations, p0=None):
        p = np.zeros(self.N, dtype=np.float32) if p0 is None else np.array(p0, dtype=np.float32)
        for _ in range(iterations):
            self.rbgs(div, p, dx, 0, omega)
            self.rbgs(div, p, dx, 1, omega)
        return p

    def subtract_pressure_gradient(self, vel, p, inv_dx, sdf=None, has_collision=False):
        vel, p = _f32(vel), _f32(p)
        out = np.zeros_like(vel)
        sdf_ = _f32(sdf) if sdf is not None else None
        self.L.orc_subtract_pressure_gradient(self.g, vel.ctypes.data, p.ctypes.data, out.ctypes.data, sdf_.ctypes.data if sdf_ is not None else None,
                                              int(has_collision), inv_dx)
        return out

    def combustion_oxygen(self, fuel, waste, temperature, div, flame, temp_gain, expansion):
        fuel, waste, temperature, flame = _f32(fuel), _f32(waste), _f32(temperature), _f32(flame)
        div = np.array(div, dtype=np.float32)
        outs = [np.zeros_like(fuel) for _ in range(4)]
        self.L.orc_combustion_oxygen(fuel.ctypes.data, waste.ctypes.data, temperature.ctypes.data, div.ctypes.data, flame.ctypes.data,
                                     outs[0].ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data, outs[3].ctypes.data, temp_gain, expansion, fuel.size)
        return outs[0], outs[1], outs[2], outs[3], div

    def temperature_buoyancy(self, vel, temp, dt, ambient, strength):
        vel, temp = _f32(vel), _f32(temp)
        out = np.zeros_like(vel)
        self.L.orc_temperature_buoyancy(vel.ctypes.data, temp.ctypes.data, out.ctypes.data, dt, ambient, strength, temp.size)
        return out

    def vorticity_confinement(self, vel, dt, inv_dx, scale, factor_scale):
        vel = _f32(vel)
        out = np.zeros_like(vel)
        self.L.orc_vorticity_confinement(self.g, vel.ctypes.data, out.ctypes.data, dt, inv_dx, scale, factor_scale)
        return out

    def enforce_collision_boundaries(self, vel, sdf, voxel_size):
        vel = np.array(vel, dtype=np.float32)
        sdf = _f32(sdf)
        self.L.orc_enforce_collision_boundaries(self.g, vel.ctypes.data, sdf.ctypes.data, voxel_size)
        return vel

    # ---- host drivers (results in place, like the reference) ----
    def compute_sim(self, vel, fields: dict, iterations, dt, voxel_size, params, has_collision):
        names = list(fields.keys())
        arr_names = (C.c_char_p * len(names))(*[n.encode() for n in names])
        ptrs = (_vp * len(names))(*[fields[n].ctypes.data for n in names])
        p = orc_combustion_params(params.expansionRate, params.temperatureRelease, params.buoyancyStrength, params.ambientTemp, params.vorticityScale,
                                  params.factorScale)
        return self.L.orc_compute_sim(self.g, vel.ctypes.data, arr_names, ptrs, len(names), iterations, dt, voxel_size, C.byref(p), int(has_collision))

    def project_non_divergent(self, vel, iterations, voxel_size):
        return self.L.orc_project_non_divergent(self.g, vel.ctypes.data, iterations, voxel_size)

    def divergence_op(self, vel, out, voxel_size):
        return self.L.orc_divergence_op(self.g, vel.ctypes.data, out.ctypes.data, voxel_size)

    def advect_index_grid(self, vel, fields: list, dt, voxel_size):
        ptrs = (_vp * len(fields))(*[f.ctypes.data for f in fields])
        return self.L.orc_advect_index_grid(self.g, vel.ctypes.data, ptrs, len(fields), dt, voxel_size)

    def advect_index_grid_velocity(self, vel, dt, voxel_size):
        return self.L.orc_advect_index_grid_velocity(self.g, vel.ctypes.data, dt, voxel_size)
