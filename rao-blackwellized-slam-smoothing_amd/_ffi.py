"""ctypes declarations mirroring include/rbpf.h (the C-ABI drop-in boundary)."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

from . import _build

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)

RBPF_OK = 0
RBPF_ERR_INVALID_ARG = 1
RBPF_ERR_UNSUPPORTED = 2
RBPF_ERR_HIP = 3
RBPF_ERR_NO_DEVICE = 4
RBPF_ERR_OUT_OF_MEMORY = 5
RBPF_ERR_CHOL_FAILED = 6
RBPF_ERR_STATE = 7
RBPF_ERR_CALLBACK = 8

RBPF_MODEL_DENSE_MAG_6D = 1
RBPF_MODEL_DENSE_RADIO_2DH = 2
RBPF_MODEL_SPARSE_VISUAL_2D = 3
RBPF_MODEL_GENERIC_DENSE = 4
RBPF_RNG_REPLAY = 0
RBPF_RNG_PHILOX = 1


# host callbacks of the generic family (include/rbpf.h rbpf_callbacks)
DYN_MODEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, c_double_p, c_double_p)
MEAS_MODEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, c_double_p, c_double_p)
DYN_RES_NORM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p)


class rbpf_callbacks(C.Structure):
    _fields_ = [("dyn_model", DYN_MODEL_FN), ("meas_model", MEAS_MODEL_FN), ("dyn_res_norm", DYN_RES_NORM_FN),
                ("user", C.c_void_p)]


class rbpf_model(C.Structure):
    _fields_ = [("kind", C.c_int32), ("m_basis", C.c_int32), ("dim", C.c_int32), ("use_dyn_res_norm", C.c_int32),
                ("NN", c_int32_p), ("L", C.c_double * 3), ("cam", C.c_double * 3),
                ("callbacks", C.POINTER(rbpf_callbacks))]


class rbpf_problem(C.Structure):
    _fields_ = [("N_P", C.c_int32), ("N_T", C.c_int32), ("n_nonlin", C.c_int32), ("n_lin", C.c_int32),
                ("n_y", C.c_int32), ("n_w", C.c_int32), ("n_odo", C.c_int32), ("x0_lin_cols", C.c_int32),
                ("q_pages", C.c_int32), ("dt_len", C.c_int32),
                ("odometry", c_double_p), ("odo_ld", C.c_int32),
                ("y", c_double_p), ("x0_nonlin", c_double_p), ("x0_lin", c_double_p), ("P0_lin", c_double_p),
                ("Q", c_double_p), ("R", c_double_p), ("dt", c_double_p)]


class rbpf_rng(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_iter", C.c_int32), ("U", c_double_p), ("Z", c_double_p),
                ("Ufin", c_double_p), ("seed", C.c_uint64)]


class rbpf_view(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("t", C.c_int32), ("is_smoother", C.c_int32)]


ON_STEP_FN = C.CFUNCTYPE(C.c_int, C.POINTER(rbpf_view), C.c_void_p)


class rbpf_options(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("keep_history", C.c_int32), ("trace", C.c_int32), ("fix_p_mean", C.c_int32),
                ("lazy_depth", C.c_int32), ("jitter", C.c_double), ("inplace", C.c_int32), ("storage", C.c_int32),
                ("chol_variant", C.c_int32), ("chol_refresh", C.c_int32), ("exchange_capacity", C.c_int32), ("on_step", ON_STEP_FN),
                ("on_step_user", C.c_void_p), ("n_devices", C.c_int32), ("device_ids", C.POINTER(C.c_int32)),
                ("info_rebuild", C.c_int32)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if "struct_size" not in kw and not args:
            self.struct_size = C.sizeof(rbpf_options)           # the library refuses options of another layout (ABI 9)


class rbpf_filter_out(C.Structure):
    _fields_ = [("traj_max", c_double_p), ("traj_mean", c_double_p), ("xl_max", c_double_p), ("xl_mean", c_double_p),
                ("P_max", c_double_p), ("P_mean", c_double_p), ("traj_sample_iwmax", c_double_p),
                ("xn_traj", c_double_p), ("trace_logw", c_double_p), ("trace_w", c_double_p),
                ("trace_ai", c_int32_p), ("final_xn", c_double_p), ("final_xl", c_double_p),
                ("final_P", c_double_p), ("iw_max", c_int32_p)]


class rbpf_smoother_out(C.Structure):
    _fields_ = [("XNK", c_double_p), ("XLK", c_double_p), ("PK", c_double_p), ("trace_logw", c_double_p),
                ("trace_w", c_double_p), ("trace_ai", c_int32_p), ("trace_paNt", c_double_p),
                ("trace_ak", c_int32_p)]


class rbpf_timing(C.Structure):
    _fields_ = [("stream_kernel_ms", C.c_double), ("stream_kernel_launches", C.c_int64),
                ("algorithmic_bytes_per_launch", C.c_double), ("scheduled_bytes_per_launch", C.c_double)]


# rbpf_abi_sizeof(which): the mirrors in the order of its `which` argument
ABI_STRUCTS = [rbpf_model, rbpf_problem, rbpf_rng, rbpf_options, rbpf_filter_out, rbpf_smoother_out, rbpf_timing, rbpf_callbacks, rbpf_view]

# every symbol include/rbpf.h declares (tests/test_abi.py checks the library exports all of them)
EXPORTS = [
    "rbpf_abi_version", "rbpf_abi_sizeof", "rbpf_status_string", "rbpf_last_error", "rbpf_device_count",
    "rbpf_particle_filter", "rbpf_particle_smoother",
    "rbpf_filter_create", "rbpf_filter_workspace_bytes", "rbpf_filter_advance", "rbpf_filter_reset", "rbpf_sync",
    "rbpf_filter_finish", "rbpf_filter_tell", "rbpf_filter_schedule", "rbpf_plan_refresh", "rbpf_chol_refresh_resolve", "rbpf_shard_smoother_refresh_reserve", "rbpf_shard_xn_traj", "rbpf_filter_ancestors", "rbpf_filter_step_external", "rbpf_timing_enable", "rbpf_timing_read", "rbpf_destroy",
    "rbpf_philox_fill", "rbpf_meas_model", "rbpf_dyn_model", "rbpf_dyn_res_norm", "rbpf_sample",
    "rbpf_jacobian_phi3d", "rbpf_chol_weights", "rbpf_chol_sweep_probe", "rbpf_quat_helpers", "rbpf_probe_wave_reduce",
    "rbpf_shard_create", "rbpf_shard_views_get", "rbpf_shard_normalise_search", "rbpf_shard_pack", "rbpf_shard_step",
    "rbpf_shard_trajectories", "rbpf_shard_plan", "rbpf_shard_plan_read", "rbpf_shard_normalise_plan",
    "rbpf_stream_get", "rbpf_shard_set_async", "rbpf_shard_finish", "rbpf_shard_set_ancestors",
    "rbpf_shard_smoother_create", "rbpf_shard_smoother_views_get", "rbpf_shard_smoother_begin",
    "rbpf_shard_smoother_normalise", "rbpf_shard_smoother_anc_weights", "rbpf_shard_smoother_anc_sample",
    "rbpf_shard_smoother_refresh_begin", "rbpf_shard_smoother_refresh_pack", "rbpf_shard_smoother_refresh_end",
    "rbpf_shard_smoother_step", "rbpf_shard_smoother_end",
]

ABI_VERSION = 9            # RBPF_ABI_VERSION of include/rbpf.h this mirror was written against
_lib = None


class RBPFError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"rbpf status {status}: {msg}")
        self.status = status


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch wheel carries its own libamdhip64 / libhsa-runtime64 (soname
    libamdhip64.so.7, the soname librbpf_hip.so asks for); if /opt/rocm's copy is mapped first and torch is imported later
    (ShardedFilterSession after a single-GPU call, say) the process holds two runtimes and the second one finds no device
    ("No HIP GPUs are available").  Mapping the wheel's copy first -- without importing torch -- makes the dynamic linker
    hand the same runtime to both; without a torch wheel the library binds /opt/rocm's as usual."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library(build_if_missing: bool = True):
    """dlopen the in-tree HIP library.  Fails loudly when it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    path = os.environ.get("RBPF_LIB_PATH") or _build.LIBPATH     # RBPF_LIB_PATH: tuning variants only
    if not os.path.exists(path):
        if not build_if_missing:
            raise RBPFError(RBPF_ERR_NO_DEVICE, f"{path} not built (run __graft_entry__.build())")
        _build.build()
    lib = C.CDLL(path)
    # the version and struct-size checks come before anything touches a newer symbol: a stale .so then says "rebuild" instead of
    # raising AttributeError on a missing entry point
    if not hasattr(lib, "rbpf_abi_version") or not hasattr(lib, "rbpf_abi_sizeof"):
        raise RBPFError(RBPF_ERR_INVALID_ARG, f"{path} predates rbpf_abi_version / rbpf_abi_sizeof (rebuild)")
    if lib.rbpf_abi_version() != ABI_VERSION:
        raise RBPFError(RBPF_ERR_INVALID_ARG, f"{path} has ABI version {lib.rbpf_abi_version()}, this mirror expects {ABI_VERSION} (rebuild)")
    lib.rbpf_abi_sizeof.argtypes = [C.c_int32]
    for which, mirror in enumerate(ABI_STRUCTS):
        if lib.rbpf_abi_sizeof(which) != C.sizeof(mirror):
            raise RBPFError(RBPF_ERR_INVALID_ARG, f"{path}: sizeof({mirror.__name__}) is {lib.rbpf_abi_sizeof(which)}, this mirror has "
                                                  f"{C.sizeof(mirror)} (include/rbpf.h and _ffi.py disagree)")
    lib.rbpf_status_string.restype = C.c_char_p
    lib.rbpf_status_string.argtypes = [C.c_int]
    lib.rbpf_last_error.restype = C.c_char_p
    lib.rbpf_particle_filter.argtypes = [C.POINTER(rbpf_model), C.POINTER(rbpf_problem), C.POINTER(rbpf_rng),
                                         C.POINTER(rbpf_options), C.POINTER(rbpf_filter_out)]
    lib.rbpf_particle_smoother.argtypes = [C.POINTER(rbpf_model), C.POINTER(rbpf_problem), C.POINTER(rbpf_rng),
                                           C.POINTER(rbpf_options), C.c_int32, C.c_int32,
                                           C.POINTER(rbpf_smoother_out)]
    lib.rbpf_filter_create.argtypes = [C.POINTER(rbpf_model), C.POINTER(rbpf_problem), C.POINTER(rbpf_rng),
                                       C.POINTER(rbpf_options), C.POINTER(C.c_void_p)]
    lib.rbpf_filter_workspace_bytes.argtypes = [C.POINTER(rbpf_model), C.POINTER(rbpf_problem),
                                                C.POINTER(rbpf_options), C.POINTER(C.c_size_t)]
    lib.rbpf_filter_advance.argtypes = [C.c_void_p, C.c_int32]
    lib.rbpf_filter_reset.argtypes = [C.c_void_p]
    lib.rbpf_sync.argtypes = [C.c_void_p]
    lib.rbpf_filter_finish.argtypes = [C.c_void_p, C.POINTER(rbpf_filter_out)]
    lib.rbpf_filter_tell.argtypes = [C.c_void_p, c_int32_p]
    lib.rbpf_filter_schedule.argtypes = [C.c_void_p, c_int32_p, c_int32_p]
    lib.rbpf_filter_ancestors.argtypes = [C.c_void_p, c_int32_p, c_double_p]
    lib.rbpf_filter_step_external.argtypes = [C.c_void_p, c_double_p, c_double_p]
    lib.rbpf_timing_enable.argtypes = [C.c_void_p, C.c_int32]
    lib.rbpf_timing_read.argtypes = [C.c_void_p, C.POINTER(rbpf_timing), C.c_int32]
    lib.rbpf_destroy.argtypes = [C.c_void_p]
    lib.rbpf_philox_fill.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p,
                                     c_double_p]
    lib.rbpf_meas_model.argtypes = [C.POINTER(rbpf_model), C.c_int32, C.c_int32, c_double_p, c_double_p]
    lib.rbpf_dyn_model.argtypes = [C.POINTER(rbpf_model), C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_double_p,
                                   c_double_p, C.c_double, c_double_p, c_double_p, c_double_p]
    lib.rbpf_dyn_res_norm.argtypes = [C.POINTER(rbpf_model), C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_double_p,
                                      c_double_p, c_double_p, C.c_double, c_double_p, c_double_p]
    lib.rbpf_sample.argtypes = [C.c_int32, c_double_p, C.c_int32, c_double_p, c_int32_p]
    lib.rbpf_jacobian_phi3d.argtypes = [C.POINTER(rbpf_model), C.c_int32, c_double_p, c_double_p, c_double_p,
                                        c_double_p]
    lib.rbpf_chol_weights.argtypes = [C.c_int32, C.c_int32, c_double_p, c_double_p, C.c_double, C.c_int32, C.c_int32,
                                      c_double_p, c_int32_p, c_double_p]
    lib.rbpf_chol_sweep_probe.argtypes = [C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int32,
                                          c_double_p, c_double_p, c_int32_p, c_double_p]
    lib.rbpf_quat_helpers.argtypes = [C.c_int32, C.c_int32, c_double_p, c_double_p]
    lib.rbpf_chol_refresh_resolve.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.rbpf_chol_refresh_resolve.restype = C.c_int32
    _lib = lib
    return lib


def check(status: int):
    if status != RBPF_OK:
        lib = load_library()
        msg = (lib.rbpf_last_error() or b"").decode() or lib.rbpf_status_string(status).decode()
        raise RBPFError(status, msg)
