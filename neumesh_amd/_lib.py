"""ctypes binding of libneumesh_hip.so (C ABI: include/neumesh_hip.h).

There is deliberately NO fallback: if the shared library is missing, cannot be loaded, or no
HIP device is visible, every compute entry point raises.  Tensors are passed as raw device
pointers (``tensor.data_ptr()``) and the current torch stream's handle.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

ABI_VERSION = 11
MAX_K = 32


class GridInfo(C.Structure):
    _fields_ = [("num_vertices", C.c_int64), ("leaf_level", C.c_int32), ("occupied_leaves", C.c_int32),
                ("origin", C.c_float * 3), ("root_size", C.c_float), ("device_bytes", C.c_int64), ("num_nodes", C.c_int64)]


class FieldDesc(C.Structure):
    _fields_ = [("W", C.c_int32), ("D_density", C.c_int32), ("D_color", C.c_int32),
                ("geometry_dim", C.c_int32), ("color_dim", C.c_int32),
                ("multires_d", C.c_int32), ("multires_fg", C.c_int32), ("multires_ft", C.c_int32),
                ("multires_view", C.c_int32), ("enable_nablas_input", C.c_int32), ("use_view_dirs", C.c_int32),
                ("mlp_precision", C.c_int32),
                ("geo_weight", C.c_void_p * 8), ("geo_bias", C.c_void_p * 8),
                ("density_weight", C.c_void_p), ("density_bias", C.c_void_p),
                ("col_weight", C.c_void_p * 8), ("col_bias", C.c_void_p * 8),
                ("rgb_weight", C.c_void_p), ("rgb_bias", C.c_void_p)]


class FieldTables(C.Structure):
    _fields_ = [("geometry_features", C.c_void_p), ("color_features", C.c_void_p),
                ("indicator_vector", C.c_void_p), ("indicator_weight", C.c_float), ("s", C.c_float)]


class RenderCfg(C.Structure):
    _fields_ = [("obj_bounding_radius", C.c_float), ("N_samples", C.c_int32), ("N_importance", C.c_int32),
                ("N_upsample_iters", C.c_int32), ("bounded_near_far", C.c_int32), ("calc_normal", C.c_int32),
                ("white_bkgd", C.c_int32), ("probe_grid", C.c_int32), ("probe_thresh", C.c_float),
                ("near_bypass", C.c_float), ("far_bypass", C.c_float),
                ("flags", C.c_uint32), ("chain_tiles", C.c_int32), ("fine_group_rays", C.c_int32),
                ("mid_group_rays", C.c_int32), ("weight_eps", C.c_float),
                ("n_edit", C.c_int32), ("code_dims", C.c_int32), ("edit_field", C.c_void_p * 4),
                ("edit_mask", C.c_void_p * 4), ("edit_color_features", C.c_void_p),
                ("edit_use_rot", C.c_int32 * 4), ("edit_rot", (C.c_float * 9) * 4), ("u_rand", C.c_void_p), ("mid_passes", C.c_int32)]


GRID_DEFER_BUDGET, GRID_TRIM = 1, 2   # nm_grid_set_option
# nm_render_cfg.flags (include/neumesh_hip.h)
RENDER_FULL_PROBES, RENDER_NO_ZERO_SKIP, RENDER_NO_RAY_SORT, RENDER_NO_MID_ORDER, RENDER_EAGER_NABLAS, RENDER_SAMPLE_ONLY = 1, 2, 4, 8, 16, 32


class Camera(C.Structure):
    _fields_ = [("c2w", C.c_float * 12), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("sk", C.c_float), ("H", C.c_int32), ("W", C.c_int32)]


class SurfaceCfg(C.Structure):
    _fields_ = [("near", C.c_float), ("far", C.c_float), ("N_steps", C.c_int32), ("logit_tau", C.c_float), ("n_secant_steps", C.c_int32),
                ("fill_inf", C.c_int32), ("scene_radius", C.c_float)]


class TrainGrads(C.Structure):
    """nm_train_grads: device pointers the backward pass ADDS into (None = that gradient is not wanted)."""
    _fields_ = [("geo_weight", C.c_void_p * 8), ("geo_bias", C.c_void_p * 8),
                ("density_weight", C.c_void_p), ("density_bias", C.c_void_p),
                ("col_weight", C.c_void_p * 8), ("col_bias", C.c_void_p * 8),
                ("rgb_weight", C.c_void_p), ("rgb_bias", C.c_void_p),
                ("geometry_features", C.c_void_p), ("color_features", C.c_void_p),
                ("indicator_vector", C.c_void_p), ("indicator_weight", C.c_void_p)]


class RenderDebug(C.Structure):
    _fields_ = [("near_far", C.c_void_p), ("d_all", C.c_void_p), ("sdf_all", C.c_void_p),
                ("nablas_all", C.c_void_p), ("radiance", C.c_void_p), ("sdf_coarse", C.c_void_p)]


# name -> (restype, argtypes); exactly the symbols include/neumesh_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "nm_abi_version": (C.c_int, []),
    "nm_last_error": (C.c_char_p, []),
    "nm_device_count": (C.c_int, []),
    "nm_grid_create": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.POINTER(_P)]),
    "nm_grid_destroy": (C.c_int, [_P]),
    "nm_grid_get_info": (C.c_int, [_P, C.POINTER(GridInfo)]),
    "nm_grid_set_option": (C.c_int, [_P, C.c_int, C.c_int64]),
    "nm_knn": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P, _P]),
    "nm_compute_distance": (C.c_int, [_P, _P, C.c_int64, _P, C.c_float, C.c_int, _P, _P, _P, _P, _P]),
    "nm_distance_interpolate": (C.c_int, [_P, _P, C.c_int64, _P, C.c_float, _P, C.c_int, _P, _P, _P, _P, _P]),
    "nm_field_create": (C.c_int, [C.POINTER(FieldDesc), _P, C.POINTER(_P)]),
    "nm_field_update": (C.c_int, [_P, C.POINTER(FieldDesc), _P]),
    "nm_field_destroy": (C.c_int, [_P]),
    "nm_field_overflow": (C.c_int, [_P, C.POINTER(C.c_int), _P]),
    "nm_field_overflow_post": (C.c_int, [_P, _P, _P]),
    "nm_field_scratch_bytes": (C.c_int64, [C.c_int64]),
    "nm_field_density": (C.c_int, [_P, _P, C.POINTER(FieldTables), _P, C.c_int64, _P, _P, _P, _P]),
    "nm_field_forward": (C.c_int, [_P, _P, C.POINTER(FieldTables), _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nm_field_color": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "nm_render_workspace_bytes": (C.c_int64, [C.POINTER(RenderCfg), C.c_int64]),
    "nm_render_rays": (C.c_int, [_P, _P, C.POINTER(FieldTables), _P, _P, C.c_int64, C.POINTER(RenderCfg),
                                 _P, _P, _P, _P, C.POINTER(RenderDebug), _P, _P]),
    "nm_rays_setup": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P, _P, _P]),
    "nm_rays_points": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "nm_rays_bounds": (C.c_int, [_P, C.c_int64, C.c_int, C.c_float, _P, _P, _P]),
    "nm_rays_upsample": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "nm_rays_finalize": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P]),
    "nm_rays_composite": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_float, _P, _P, C.c_int, _P, _P, _P, _P, _P]),
    "nm_make_rays": (C.c_int, [C.POINTER(Camera), C.c_int64, C.c_int64, _P, _P, _P]),
    "nm_make_rays_indexed": (C.c_int, [C.POINTER(Camera), _P, C.c_int64, _P, _P, _P]),
    "nm_train_workspace_bytes": (C.c_int64, [C.POINTER(FieldDesc), C.c_int64]),
    "nm_train_forward": (C.c_int, [C.POINTER(FieldDesc), _P, C.POINTER(FieldTables), _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, _P]),
    "nm_train_backward": (C.c_int, [C.POINTER(FieldDesc), _P, C.POINTER(FieldTables), C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P,
                                    C.POINTER(TrainGrads), _P]),
    "nm_train_composite_forward": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nm_train_composite_backward": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P,
                                              _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nm_surface_workspace_bytes": (C.c_int64, [_P, C.c_int64]),
    "nm_surface_hits": (C.c_int, [_P, _P, C.POINTER(FieldTables), _P, _P, C.c_int64, _P, C.POINTER(SurfaceCfg), _P, _P, _P, _P, _P, _P]),
    "nm_assemble_frame": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, _P]),
    "nm_profile_enable": (C.c_int, [C.c_int]),
    "nm_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "nm_profile_clock": (C.c_int, [C.c_int, C.POINTER(C.c_float), _P]),
    "nm_time_kernel": (C.c_int, [_P, _P, C.POINTER(FieldTables), C.c_int, _P, _P, C.c_int64, _P, C.c_int,
                                 C.POINTER(C.c_float), _P]),
}
# test hooks of the -DNM_TESTING build (tests/_build/libneumesh_hip_testing.so): not in the product library, not in the header
TESTING_SIGNATURES = {
    "nm_debug_wave_log": (C.c_int, [_P]),
    "nm_debug_phase_log": (C.c_int, [_P]),
    "nm_grid_create_host": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.POINTER(_P)]),
    "nm_grid_debug_export": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "nm_selfcheck_field": (C.c_int, [_P, _P, C.POINTER(FieldTables), _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "nm_debug_last_deferred": (C.c_int, [_P, C.POINTER(C.c_int), _P]),
    "nm_debug_gemm": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int64, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
}

_lib = None


class NeuMeshHipError(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get("NEUMESH_HIP_LIB", _build.LIB_PATH)   # override: experiments only


def load(require_device: bool = True):
    """Load (once) and return the ctypes library.  Raises NeuMeshHipError when it is missing;
    with require_device also when no HIP device is visible."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64; it must be the HIP runtime of the process, so make sure
        # it is loaded BEFORE our library pulls in /opt/rocm's copy under the same SONAME
        # (two runtimes in one process: "No HIP GPUs are available").
        import torch  # noqa: F401
        path = lib_path()
        if not os.path.exists(path):
            raise NeuMeshHipError(
                f"{path} not found: build it with `python -m neumesh_amd.build` (needs hipcc). "
                "neumesh_amd has no CPU fallback.")
        try:
            lib = C.CDLL(path)
        except OSError as e:  # e.g. libamdhip64 missing
            raise NeuMeshHipError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("NEUMESH_HIP_LIB"):   # a measurement tool's own build may carry the hooks
            for name, (res, args) in TESTING_SIGNATURES.items():
                if hasattr(lib, name):
                    getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
        if lib.nm_abi_version() != ABI_VERSION:
            raise NeuMeshHipError(f"ABI mismatch: library {lib.nm_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    if require_device and _lib.nm_device_count() < 1:
        raise NeuMeshHipError("no HIP device visible: the NeuMesh HIP path needs an MI355X (gfx950); "
                              "there is no CPU fallback")
    return _lib


def load_testing():
    """The -DNM_TESTING library (product exports + test hooks) as its own ctypes object: for the tests of the hooks and the
    measurement tools.  Handles are NOT interchangeable with those of the product library (two copies of the library state)."""
    import torch  # noqa: F401  (its libamdhip64 first, see load())
    path = os.environ.get("NEUMESH_HIP_TESTING_LIB") or _build.build_testing()   # override: experiments only
    lib = C.CDLL(path)
    for name, (res, args) in {**SIGNATURES, **TESTING_SIGNATURES}.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def check(rc: int, what: str, lib=None):
    if rc != 0:
        src = lib if lib is not None else _lib
        msg = src.nm_last_error().decode("utf-8", "replace") if src is not None else "?"
        raise NeuMeshHipError(f"{what}: {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
