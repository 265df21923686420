"""ctypes binding of libhns.so (include/hns.h). Fails loudly when the library is missing: no fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HNS_LIBRARY: another build of the same library (A/B measurements of kernel variants, profiles/micro/exp)
_LIB_PATH = os.environ.get("HNS_LIBRARY") or os.path.join(_HERE, "lib", "libhns.so")

HNS_OK = 0
HNS_ERR_INVALID_ARGUMENT = -1
HNS_ERR_RUNTIME = -2
HNS_ERR_HIP = -3
HNS_ERR_NO_DEVICE = -4
HNS_ERR_TOPOLOGY = -5

HNS_GRID_DEFAULT = 0
HNS_GRID_HOST_ONLY = 1
HNS_GRID_SKIP_VALIDATE = 2


class HNSError(RuntimeError):
    """A libhns call returned a negative code. ``code`` is the HNS_ERR_* value."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libhns error {code}: {message}")
        self.code = code
        self.message = message


class hns_field(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ncomp", C.c_int), ("host", C.POINTER(C.c_float))]


class hns_dist_stats(C.Structure):
    _fields_ = [("world", C.c_int), ("rank", C.c_int), ("peers", C.c_int), ("sweeps_per_exchange", C.c_int),
                ("boundary_leaves", C.c_uint64), ("interior_leaves", C.c_uint64), ("ghost_leaves", C.c_uint64),
                ("region_voxels_sent", C.c_uint64 * 4), ("bytes_sent", C.c_uint64 * 4), ("messages_sent", C.c_uint64), ("exchanges", C.c_uint64),
                ("halo_peers", C.c_uint64), ("packed_exchanges", C.c_uint64), ("chained", C.c_uint64)]


HNS_DIST_IPC_BLOB_BYTES = 2048
HNS_DIST_PLAN_ONLY = 1
HNS_DIST_LEAF_ORDER = 2


class hns_combustion_params(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("expansionRate", "temperatureRelease", "buoyancyStrength", "ambientTemp", "vorticityScale", "factorScale")]


_vp, _i, _u64, _f = C.c_void_p, C.c_int, C.c_uint64, C.c_float
_fp = C.c_void_p  # device or host float* passed as an address
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes): every symbol include/hns.h declares
SIGNATURES = {
    "hns_last_error": (C.c_char_p, []),
    "hns_version": (_i, []),
    "hns_device_count": (_i, []),
    "hns_trim_memory": (_i, []),
    "hns_set_option": (_i, [C.c_char_p, C.c_char_p]),
    "hns_get_option": (C.c_char_p, [C.c_char_p]),
    "hns_grid_create": (_vp, [_vp, _u64, _f, C.c_uint, _ip]),
    "hns_grid_create_from_leaves": (_vp, [_vp, _u64, _f, C.c_uint, _ip]),
    "hns_grid_destroy": (None, [_vp]),
    "hns_grid_leaf_count": (_u64, [_vp]),
    "hns_grid_voxel_count": (_u64, [_vp]),
    "hns_grid_voxel_size": (_f, [_vp]),
    "hns_grid_set_active_leaves": (_i, [_vp, _u64]),
    "hns_grid_set_active_range": (_i, [_vp, _u64, _u64]),
    "hns_grid_active_leaves": (_u64, [_vp]),
    "hns_grid_set_outside_element": (_i, [_vp, _u64]),
    "hns_grid_offsets": (_i, [_vp, _vp, _u64, _vp]),
    "hns_grid_neighbor_table": (_i, [_vp, _vp]),
    "hns_grid_coords": (_i, [_vp, _vp]),
    "hns_grid_release_cache": (_i, [_vp]),
    "hns_grid_matches": (_i, [_vp, _vp, _u64, C.c_uint]),
    "hns_grid_export_nanovdb": (_i, [_vp, _vp, _u64, _vp]),
    "hns_grid_launch_tables": (_i, [_vp, _vp]),
    "hns_gather_leaves": (_i, [_vp, _u64, _vp, _u64, _vp, _i, _i, _vp]),
    "hns_scatter_leaves": (_i, [_vp, _u64, _i, C.POINTER(C.c_void_p)]),
    "hns_dilate_leaves": (_i, [_vp, _u64, _vp, _i, _vp, _u64, C.POINTER(C.c_uint64)]),
    "hns_union_leaves": (_i, [_vp, _u64, _vp, _u64, _vp, _u64, C.POINTER(C.c_uint64)]),
    "hns_compute_sim": (_i, [_vp, C.POINTER(hns_field), _i, _i, _f, _f, C.POINTER(hns_combustion_params), _i, _vp]),
    "hns_compute_sim_resident": (_i, [_vp, C.POINTER(hns_field), _i, C.c_char_p, C.POINTER(C.c_int), _i, _f, _f, C.POINTER(hns_combustion_params), _i, _vp]),
    "hns_advect_index_grid": (_i, [_vp, C.POINTER(hns_field), _i, _f, _f, _vp]),
    "hns_advect_index_grid_velocity": (_i, [_vp, C.POINTER(hns_field), _i, _f, _f, _vp]),
    "hns_project_non_divergent": (_i, [_vp, C.POINTER(hns_field), _i, _u64, _f, _vp]),
    "hns_divergence": (_i, [_vp, C.POINTER(hns_field), _i, _f, _vp]),
    "hns_sim_create": (_vp, [_vp, C.POINTER(C.c_char_p), _i, _ip]),
    "hns_sim_destroy": (None, [_vp]),
    "hns_sim_upload": (_i, [_vp, C.POINTER(hns_field), _i, _vp]),
    "hns_sim_download": (_i, [_vp, C.POINTER(hns_field), _i, _vp]),
    "hns_sim_substep": (_i, [_vp, _i, _f, _f, C.POINTER(hns_combustion_params), _i, _vp]),
    "hns_sim_core_substep": (_i, [_vp, _i, _f, _f, _vp]),
    "hns_sim_pressure_solve": (_i, [_vp, _i, _f, _vp]),
    "hns_sim_timing": (_i, [_vp, _i]),
    "hns_sim_pressure_time": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_longlong)]),
    "hns_sim_stage_timing": (_i, [_vp, _i]),
    "hns_sim_stage_times": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_longlong)]),
    "hns_sim_velocity_ptr": (_vp, [_vp]),
    "hns_sim_field_ptr": (_vp, [_vp, C.c_char_p]),
    "hns_sim_divergence_ptr": (_vp, [_vp]),
    "hns_sim_pressure_ptr": (_vp, [_vp]),
    "hns_dev_advect_vector": (_i, [_vp, _fp, _fp, _fp, _i, _f, _f, _vp]),
    "hns_dev_advect_scalar": (_i, [_vp, _fp, _fp, _fp, _fp, _i, _f, _f, _vp]),
    "hns_dev_advect_scalars": (_i, [_vp, _fp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i, _fp, _i, _f, _f, _vp]),
    "hns_dev_divergence": (_i, [_vp, _fp, _fp, _f, _vp]),
    "hns_dev_rbgs_color": (_i, [_vp, _fp, _fp, _f, _f, _i, _vp]),
    "hns_dev_rbgs_iterate": (_i, [_vp, _fp, _fp, _fp, _f, _f, _i, _ip, _vp]),
    "hns_dev_subtract_pressure_gradient": (_i, [_vp, _fp, _fp, _fp, _fp, _i, _f, _vp]),
    "hns_dev_combustion_oxygen": (_i, [_fp] * 9 + [_f, _f, _u64, _vp]),
    "hns_dev_temperature_buoyancy": (_i, [_fp, _fp, _fp, _f, _f, _f, _u64, _vp]),
    "hns_dev_vorticity_confinement": (_i, [_vp, _fp, _fp, _f, _f, _f, _f, _vp]),
    "hns_dev_enforce_collision_boundaries": (_i, [_vp, _fp, _fp, _f, _vp]),
    "hns_dev_pack_leaves": (_i, [_fp, _vp, _u64, _fp, _i, _vp]),
    "hns_dev_unpack_leaves": (_i, [_fp, _vp, _u64, _fp, _i, _vp]),
    "hns_dist_create": (_vp, [_vp, _u64, _i, _i, _f, _i, _i, C.c_uint, _ip]),
    "hns_dist_destroy": (None, [_vp]),
    "hns_dist_unique_id": (_i, [_vp]),
    "hns_dist_connect_rccl": (_i, [_vp, _vp]),
    "hns_dist_connect_local": (_i, [C.POINTER(C.c_void_p), _i]),
    "hns_dist_connect_loopback": (_i, [_vp]),
    "hns_dist_connect_loopback_rccl": (_i, [_vp]),
    "hns_dist_ipc_export": (_i, [_vp, _vp]),
    "hns_dist_connect_ipc": (_i, [_vp, _vp]),
    "hns_dist_owned_leaves": (_u64, [_vp]),
    "hns_dist_first_owned_leaf": (_u64, [_vp]),
    "hns_dist_owned_leaf_ids": (_i, [_vp, _vp]),
    "hns_dist_partition_axis": (_i, [_vp]),
    "hns_dist_one_sided_sweeps": (_i, [_u64, _i]),
    "hns_dist_info": (_i, [_vp, C.POINTER(hns_dist_stats)]),
    "hns_dist_local_leaves": (_i, [_vp, _vp]),
    "hns_dist_peer_rank": (_i, [_vp, _i]),
    "hns_dist_peer_region": (_i, [_vp, _i, _i, _i, _vp, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hns_dist_upload": (_i, [_vp, _vp, C.POINTER(C.c_void_p), _vp]),
    "hns_dist_download": (_i, [_vp, _vp, C.POINTER(C.c_void_p), _vp, _vp]),
    "hns_dist_download_local": (_i, [_vp, _i, _vp, _vp]),
    "hns_dist_core_substep": (_i, [_vp, _i, _f, _vp]),
    "hns_dist_local_core_substep": (_i, [C.POINTER(C.c_void_p), _i, _i, _f, _vp]),
    "hns_dist_sim_substep": (_i, [_vp, _i, _f, _vp, _ip, _i, _vp]),
    "hns_dist_local_sim_substep": (_i, [C.POINTER(C.c_void_p), _i, _i, _f, _vp, _ip, _i, _vp]),
    "hns_dist_timing": (_i, [_vp, _i]),
    "hns_dist_pressure_time": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_longlong)]),
    "hns_dist_synchronize": (_i, [_vp, _vp]),
    "hns_dev_time_rbgs": (_i, [_vp, _fp, _fp, _fp, _f, _f, _i, _i, C.POINTER(C.c_float), _vp]),
    "hns_grid_rbgs_plan": (_i, [_vp, _i, C.c_char_p, C.c_uint64, _ip, _ip]),
}

_lib = None


def library_path() -> str:
    return _LIB_PATH


def load_library() -> C.CDLL:
    """Load libhns.so and declare every signature. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise FileNotFoundError(
            f"{_LIB_PATH} is missing: build it with `make -C hnanosolver_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "hnanosolver_amd has no CPU fallback."
        )
    lib_ = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib_, name)  # AttributeError here means the .so and include/hns.h disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib_
    return lib_


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _LazyLib()


def set_option(name: str, value=None) -> None:
    """hns_set_option: switch an alternative kernel form / data-movement strategy (include/hns.h); value None = default."""
    check(load_library().hns_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name: str):
    v = load_library().hns_get_option(name.encode())
    return None if v is None else v.decode()


def check(code: int) -> None:
    if code < 0:
        raise HNSError(code, load_library().hns_last_error().decode("utf-8", "replace"))
