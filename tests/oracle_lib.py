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


# ---------------------------------------------------------------------------------------------------------------------
# The reference's OWN kernel bodies (src/Cuda/Kernel.cu compiled where it lies, oracle/ref_kernels.cpp is the launch)
# ---------------------------------------------------------------------------------------------------------------------
_REFK = None


def reference_kernels(contracted: bool = False):
    """oracle/_ref/libhns_refk.so, or None when it is absent and cannot be built here (no reference checkout, or no CUDA
    runtime headers in the image). contracted=True: the same reference source built with floating-point contraction on
    (libhns_refk_fma.so), used only to measure the tolerance floor."""
    global _REFK
    if _REFK is not None and not contracted:
        return _REFK
    path = os.path.join(ORACLE_DIR, "_ref", "libhns_refk_fma.so" if contracted else "libhns_refk.so")
    if not os.path.exists(path):
        if not os.path.exists("/root/reference/src/Cuda/Kernel.cu"):
            return None
        subprocess.run(["make", "-C", ORACLE_DIR, "ref_fma" if contracted else "ref"], check=True, capture_output=True)
        if not os.path.exists(path):
            return None
    L = C.CDLL(path)
    u64 = C.c_uint64
    sig = {
        "refk_advect_vector": [_vp, _vp, _vp, _vp, _vp, _i, u64, _f, _f],
        "refk_advect_scalar": [_vp, _vp, _vp, _vp, _vp, _vp, _i, u64, _f, _f],
        "refk_advect_scalars": [_vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, _i, u64, _f, _f],
        "refk_divergence": [_vp, _vp, _vp, _vp, _f, u64],
        "refk_rbgs": [_vp, _vp, _vp, _vp, _f, u64, _i, _f],
        "refk_subtract_pressure_gradient": [_vp, _vp, u64, _vp, _vp, _vp, _vp, _i, _f],
        "refk_temperature_buoyancy": [_vp, _vp, _vp, _f, _f, _f, u64],
        "refk_combustion_oxygen": [_vp] * 9 + [_f, _f, u64],
        "refk_vorticity_confinement": [_vp, _vp, _vp, _vp, _f, _f, _f, _f, u64],
        "refk_enforce_collision_boundaries": [_vp, _vp, _vp, _vp, _f, u64],
        "refk_divergence_opt": [_vp, _vp, _vp, _f, _i],
        "refk_rbgs_opt": [_vp, _vp, _vp, _f, u64, _i, _f, _i],
        "refk_subtract_pressure_gradient_opt": [_vp, _vp, _vp, _vp, _f, u64],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.restype = None
        fn.argtypes = args
    if not contracted:
        _REFK = L
    return L


class RefKernelGrid:
    """The interface of OracleGrid with the REFERENCE's literal kernels as the engine (Kernel.cu built into
    oracle/_ref/libhns_refk.so). The index grid is NanoVDB's own (host builder, as Tests/IndexGrid.cpp:125), d_coords the
    coordinate of every value in value order. The host drivers below are launch sequences only -- each line cites the
    reference's launch it stands for; every number comes out of the reference's own kernel code.

    `leaf_origins` must already be in NanoVDB order (the order the reference's flat arrays have)."""

    def __init__(self, leaf_origins, lib=None):
        self.R = reference_samplers()
        self.K = lib or reference_kernels()
        if self.R is None or self.K is None:
            raise RuntimeError("reference kernel library not available")
        for name, (res, args) in {"ref_grid_nanogrid": (_vp, [_vp]), "ref_coords": (None, [_vp, _vp])}.items():
            fn = getattr(self.R, name)
            fn.restype, fn.argtypes = res, args
        self.origins = np.ascontiguousarray(leaf_origins, dtype=np.int32).reshape(-1, 3)
        self.h = self.R.ref_grid_create(self.origins.ctypes.data, self.origins.shape[0])
        if not self.h:
            raise ValueError("NanoVDB rejected the leaf set")
        self.n_leaves = int(self.R.ref_leaf_count(self.h))
        order = np.zeros((self.n_leaves, 3), dtype=np.int32)
        self.R.ref_leaf_origins(self.h, order.ctypes.data)
        if self.n_leaves != self.origins.shape[0] or not np.array_equal(order, self.origins):
            raise ValueError("leaf origins are not in NanoVDB order")
        self.N = self.n_leaves * 512
        self.g = self.R.ref_grid_nanogrid(self.h)
        self._coords = np.zeros((self.N, 3), dtype=np.int32)
        self.R.ref_coords(self.h, self._coords.ctypes.data)
        self.c = self._coords.ctypes.data
        self._libm = C.CDLL("libm.so.6")
        self._libm.sinf.restype, self._libm.sinf.argtypes = _f, [_f]

    def __del__(self):
        try:
            if self.h:
                self.R.ref_grid_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def coords(self):
        return self._coords.copy()

    @staticmethod
    def _p(a):
        return a.ctypes.data if a is not None else None

    # ---- kernels, one launch each ----
    def advect_vector(self, vel, dt, inv_dx, sdf=None, has_collision=False):
        vel = _f32(vel)
        out = np.zeros_like(vel)
        sdf_ = _f32(sdf) if sdf is not None else None
        self.K.refk_advect_vector(self.g, self.c, vel.ctypes.data, out.ctypes.data, self._p(sdf_), int(has_collision), self.N, dt, inv_dx)
        return out

    def advect_scalar(self, vel, phi, dt, inv_dx, sdf=None, has_collision=False):
        vel, phi = _f32(vel), _f32(phi)
        out = np.zeros_like(phi)
        sdf_ = _f32(sdf) if sdf is not None else None
        self.K.refk_advect_scalar(self.g, self.c, vel.ctypes.data, phi.ctypes.data, out.ctypes.data, self._p(sdf_), int(has_collision), self.N, dt, inv_dx)
        return out

    def advect_scalars(self, vel, phis, dt, inv_dx, sdf=None, has_collision=False):
        vel = _f32(vel)
        phis = [_f32(p) for p in phis]
        outs = [np.zeros_like(p) for p in phis]
        n = len(phis)
        ins_ = (_vp * max(1, n))(*[p.ctypes.data for p in phis])
        outs_ = (_vp * max(1, n))(*[p.ctypes.data for p in outs])
        sdf_ = _f32(sdf) if sdf is not None else None
        self.K.refk_advect_scalars(self.g, self.c, vel.ctypes.data, ins_, outs_, n, self._p(sdf_), int(has_collision), self.N, dt, inv_dx)
        return outs

    def divergence(self, vel, inv_dx):
        vel = _f32(vel)
        out = np.zeros(self.N, dtype=np.float32)
        self.K.refk_divergence(self.g, self.c, vel.ctypes.data, out.ctypes.data, inv_dx, self.N)
        return out

    def divergence_opt(self, vel, inv_dx):
        vel = _f32(vel)
        out = np.zeros(self.N, dtype=np.float32)
        self.K.refk_divergence_opt(self.g, vel.ctypes.data, out.ctypes.data, inv_dx, self.n_leaves)
        return out

    def rbgs(self, div, p, dx, color, omega):
        assert p.dtype == np.float32 and p.flags["C_CONTIGUOUS"]
        div = _f32(div)
        self.K.refk_rbgs(self.g, self.c, div.ctypes.data, p.ctypes.data, dx, self.N, color, omega)
        return p

    def rbgs_opt(self, div, p, dx, color, omega):
        assert p.dtype == np.float32 and p.flags["C_CONTIGUOUS"]
        div = _f32(div)
        self.K.refk_rbgs_opt(self.g, div.ctypes.data, p.ctypes.data, dx, self.N, color, omega, self.n_leaves)
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
        self.K.refk_subtract_pressure_gradient(self.g, self.c, self.N, vel.ctypes.data, p.ctypes.data, out.ctypes.data, self._p(sdf_), int(has_collision), inv_dx)
        return out

    def subtract_pressure_gradient_opt(self, vel, p, inv_dx):
        vel, p = _f32(vel), _f32(p)
        out = np.zeros_like(vel)
        self.K.refk_subtract_pressure_gradient_opt(self.g, vel.ctypes.data, p.ctypes.data, out.ctypes.data, inv_dx, self.n_leaves)
        return out

    def combustion_oxygen(self, fuel, waste, temperature, div, flame, temp_gain, expansion):
        fuel, waste, temperature, flame = _f32(fuel), _f32(waste), _f32(temperature), _f32(flame)
        div = np.array(div, dtype=np.float32)
        outs = [np.zeros_like(fuel) for _ in range(4)]
        self.K.refk_combustion_oxygen(fuel.ctypes.data, waste.ctypes.data, temperature.ctypes.data, div.ctypes.data, flame.ctypes.data,
                                      outs[0].ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data, outs[3].ctypes.data, temp_gain, expansion, fuel.size)
        return outs[0], outs[1], outs[2], outs[3], div

    def temperature_buoyancy(self, vel, temp, dt, ambient, strength):
        vel, temp = _f32(vel), _f32(temp)
        out = np.zeros_like(vel)
        self.K.refk_temperature_buoyancy(vel.ctypes.data, temp.ctypes.data, out.ctypes.data, dt, ambient, strength, temp.size)
        return out

    def vorticity_confinement(self, vel, dt, inv_dx, scale, factor_scale):
        """Out of place. (The reference launches it in place, HNanoSolver.cu:174: a data race on a GPU whenever
        factor_scale >= 1; below 1 the kernel is a copy, so in place = out of place.)"""
        vel = _f32(vel)
        out = np.zeros_like(vel)
        self.K.refk_vorticity_confinement(self.g, self.c, vel.ctypes.data, out.ctypes.data, dt, inv_dx, scale, factor_scale, self.N)
        return out

    def enforce_collision_boundaries(self, vel, sdf, voxel_size):
        vel = np.array(vel, dtype=np.float32)
        sdf = _f32(sdf)
        self.K.refk_enforce_collision_boundaries(self.g, self.c, vel.ctypes.data, sdf.ctypes.data, voxel_size, self.N)
        return vel

    # ---- host drivers: launch sequences of the reference ----
    def project_non_divergent(self, vel, iterations, voxel_size):
        """PressureProjection.cu:43-66: divergence_opt, iterations x (red, black) redBlackGaussSeidelUpdate_opt from p = 0,
        subtractPressureGradient_opt in place; omega evaluated in double (:53)."""
        import math

        vs = float(np.float32(voxel_size))
        inv = float(np.float32(1.0) / np.float32(vs))
        div = self.divergence_opt(vel, inv)  # :48
        omega = float(np.float32(2.0 / (1.0 + math.sin(3.14159 * vs))))  # :53
        p = np.zeros(self.N, dtype=np.float32)  # :30 (cudaMemset)
        for _ in range(int(iterations)):
            self.rbgs_opt(div, p, vs, 0, omega)  # :55
            self.rbgs_opt(div, p, vs, 1, omega)  # :57
        vel[...] = self.subtract_pressure_gradient_opt(vel, p, inv)  # :64
        return 0

    def divergence_op(self, vel, out, voxel_size):
        """PressureProjection.cu:114"""
        vs = np.float32(voxel_size)
        out[...] = self.divergence(vel, float(np.float32(1.0) / vs))
        return 0

    def advect_index_grid(self, vel, fields: list, dt, voxel_size):
        """Advection.cu:88-91: advect_scalar per field, no collision"""
        inv = float(np.float32(1.0) / np.float32(voxel_size))
        for f in fields:
            f[...] = self.advect_scalar(vel, f, dt, inv)
        return 0

    def advect_index_grid_velocity(self, vel, dt, voxel_size):
        """Advection.cu:153"""
        inv = float(np.float32(1.0) / np.float32(voxel_size))
        vel[...] = self.advect_vector(vel, dt, inv)
        return 0

    def compute_sim(self, vel, fields: dict, iterations, dt, voxel_size, params, has_collision):
        """HNanoSolver.cu:150-356, launch for launch."""
        vs = float(np.float32(voxel_size))
        inv = float(np.float32(1.0) / np.float32(vs))  # :38
        names = list(fields.keys())
        for need in ("fuel", "waste", "temperature", "flame"):  # :193-201
            if need not in fields:
                return -1
        coll = bool(has_collision) and "collision_sdf" in fields  # :66-76
        sdf = fields["collision_sdf"].copy() if coll else None
        d_in = {n: fields[n].copy() for n in names}  # :126-131
        u = np.array(vel, dtype=np.float32)
        if coll:
            u = self.enforce_collision_boundaries(u, sdf, vs)  # :154
        ua = self.advect_vector(u, dt, inv, sdf, coll)  # :164
        ua = self.vorticity_confinement(ua, dt, inv, params.vorticityScale, params.factorScale)  # :174
        div = self.divergence(ua, inv)  # :184
        cf, cw, ct, cl, div = self.combustion_oxygen(d_in["fuel"], d_in["waste"], d_in["temperature"], div, d_in["flame"],
                                                     params.temperatureRelease, params.expansionRate)  # :213
        ua = self.temperature_buoyancy(ua, ct, dt, params.ambientTemp, params.buoyancyStrength)  # :228
        d_in.update(fuel=cf, waste=cw, temperature=ct, flame=cl)  # :238-245
        omega = float(self._libm.sinf(C.c_float(float(np.float32(3.14159) * np.float32(vs)))))
        omega = float(np.float32(2.0) / (np.float32(1.0) + np.float32(omega)))  # :257
        p = self.rbgs_iterations(div, vs, omega, int(iterations))  # :260-269, p = 0 (:113)
        u = self.subtract_pressure_gradient(ua, p, inv, sdf, coll)  # :282
        if coll:
            u = self.enforce_collision_boundaries(u, sdf, vs)  # :293
        adv = [n for n in names if n != "collision_sdf"]  # :327
        outs = self.advect_scalars(u, [d_in[n] for n in adv], dt, inv, sdf, coll)  # :346
        vel[...] = u  # :361
        for n, o in zip(adv, outs):  # :364-369
            fields[n][...] = o
        if "collision_sdf" in fields:
            fields["collision_sdf"][...] = 0.0  # its d_outputs buffer is never written after the memset (:108-110)
        return 0
