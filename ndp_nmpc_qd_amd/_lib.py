"""ctypes binding of the C-ABI in include/ndp_nmpc.h (libndp_nmpc_hip.so, HIP/gfx950).

There is deliberately no fallback: if the shared library is missing, or no MI355X is
usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NDP_NMPC_LIB: kernel-development override (a library built with other flags, see scripts/dev_kernel.sh); unset in normal use
LIB_PATH = os.environ.get("NDP_NMPC_LIB") or os.path.join(_HERE, "libndp_nmpc_hip.so")
WEIGHTS_PATH = os.path.join(_HERE, "weights", "downwash_sn4.bin")

NX, NU = 10, 4
MLP_NPARAM = 17859
QP_AUTO, QP_IPM_ALWAYS = 0, 1
ABI_VERSION = 7          # include/ndp_nmpc.h: NDP_ABI_VERSION (checked against the loaded library in load())
TICK_ESTIMATE, TICK_WANT_U0, TICK_T_UNIFORM = 1, 2, 4


class NdpCfg(C.Structure):
    """include/ndp_nmpc.h: struct ndp_cfg."""
    _fields_ = [
        ("batch", C.c_int32), ("N", C.c_int32), ("n_rti", C.c_int32), ("use_fd", C.c_int32),
        ("qp_mode", C.c_int32), ("iter_max", C.c_int32), ("device", C.c_int32), ("qp_precision", C.c_int32),
        ("work_queue", C.c_int32), ("ipm_refine", C.c_int32),
        ("dt", C.c_double), ("mass", C.c_double), ("gravity", C.c_double), ("r_horiz", C.c_double),
        ("Qd", C.c_double * 10), ("Rd", C.c_double * 4),
        ("lbu", C.c_double * 4), ("ubu", C.c_double * 4), ("lbv", C.c_double * 3), ("ubv", C.c_double * 3),
        ("mu0", C.c_double), ("thr0", C.c_double), ("tol", C.c_double), ("tau", C.c_double), ("auto_margin", C.c_double),
        ("ts_nmpc", C.c_double), ("mu_floor", C.c_double), ("refine_gamma", C.c_double),
        ("as_iter_max", C.c_int32), ("reserved0", C.c_int32), ("as_gamma", C.c_double),
    ]


EXPORTS = [
    "ndp_default_cfg", "ndp_create", "ndp_destroy", "ndp_last_error", "ndp_set_mlp_weights", "ndp_reset",
    "ndp_reset_device", "ndp_step", "ndp_step_device", "ndp_downwash", "ndp_downwash_device", "ndp_get_iterate",
    "ndp_set_iterate", "ndp_get_status", "ndp_device_iterate_x", "ndp_device_iterate_u", "ndp_device_force",
    "ndp_synchronize", "ndp_timing_enable", "ndp_timing_read", "ndp_debug_lds_doubles", "ndp_step_debug", "ndp_debug_mfma_probe", "ndp_debug_stamps", "ndp_debug_lds_layout", "ndp_throttle_reset", "ndp_throttle_update",
    "ndp_throttle_update_device", "ndp_actuator_cmd", "ndp_actuator_cmd_device", "ndp_throttle_get_state", "ndp_relay_reset", "ndp_relay_formation",
    "ndp_relay_reference", "ndp_relay_reference_device", "ndp_plant_step", "ndp_plant_step_device",
    "ndp_ref_set_trajectory", "ndp_ref_window", "ndp_ref_window_device", "ndp_rollout_device",
    "ndp_step_ex", "ndp_step_device_ex", "ndp_work_queue_enabled", "ndp_ref_list_reset", "ndp_ref_list_fix_pt",
    "ndp_ref_list_window", "ndp_ref_list_advance_device", "ndp_ref_list_window_device", "ndp_debug_mfma_probe_f32",
    "ndp_peer_alloc", "ndp_peer_open", "ndp_peer_close", "ndp_peer_free", "ndp_step_begin", "ndp_step_end",
    "ndp_peer_layout", "ndp_peer_publish_device", "ndp_peer_stats", "ndp_debug_host_timing", "ndp_debug_downwash_stream_device",
    "ndp_downwash_prefetch_device", "ndp_step_device_prefetched", "ndp_prefetch_join", "ndp_prefetch_stats", "ndp_device_force_slot",
    "ndp_track_steps", "ndp_last_step_event",
    "ndp_xchg_unique_id", "ndp_xchg_create", "ndp_xchg_begin", "ndp_xchg_end", "ndp_xchg_tick", "ndp_xchg_last_error", "ndp_xchg_destroy",
    "ndp_abi_version", "ndp_cfg_size",
    "ndp_step_ex_f64", "ndp_refine_active", "ndp_get_active_set", "ndp_set_active_set", "ndp_debug_host_info", "ndp_tick_config", "ndp_tick_reset", "ndp_tick_begin", "ndp_tick_end", "ndp_tick", "ndp_tick_device",
    "ndp_tick_config_remote", "ndp_tick_advance_device", "ndp_tick_window_pv_device", "ndp_tick_step_device", "ndp_xchg_tick_windows", "ndp_xchg_tick_begin", "ndp_xchg_tick_step", "ndp_xchg_tick_async",
]

_lib = None


def load():
    """Loads the HIP library; raises if it has not been built (python -m ndp_nmpc_qd_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libndp_nmpc_hip.so is missing: build it with `python -m ndp_nmpc_qd_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP runtime; if it is going to be used in this process it has to be
    # the first one loaded, otherwise torch later reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
    # A stale library (or one built from another header) would read / write past NdpCfg: refuse it.
    if not hasattr(lib, "ndp_abi_version"):
        raise RuntimeError(f"{LIB_PATH} predates the interface version check (include/ndp_nmpc.h: NDP_ABI_VERSION): rebuild it")
    lib.ndp_cfg_size.restype = C.c_size_t
    if lib.ndp_abi_version() != ABI_VERSION or lib.ndp_cfg_size() != C.sizeof(NdpCfg):
        raise RuntimeError(
            f"{LIB_PATH}: interface version {lib.ndp_abi_version()} / sizeof(ndp_cfg) {lib.ndp_cfg_size()}, this binding expects "
            f"{ABI_VERSION} / {C.sizeof(NdpCfg)}: rebuild the library (python -m ndp_nmpc_qd_amd.build)")
    lib.ndp_default_cfg.argtypes = [C.POINTER(NdpCfg)]
    lib.ndp_create.argtypes = [C.POINTER(NdpCfg), C.POINTER(vp)]
    lib.ndp_destroy.argtypes = [vp]
    lib.ndp_last_error.argtypes = [vp]
    lib.ndp_last_error.restype = C.c_char_p
    lib.ndp_set_mlp_weights.argtypes = [vp, vp, C.c_size_t]
    lib.ndp_reset.argtypes = [vp, vp, vp]
    lib.ndp_reset_device.argtypes = [vp, vp, vp, vp]
    lib.ndp_step.argtypes = [vp] * 8
    lib.ndp_step_device.argtypes = [vp] * 9
    lib.ndp_step_ex.argtypes = [vp] * 12
    lib.ndp_step_ex_f64.argtypes = [vp] * 10
    lib.ndp_step_begin.argtypes = [vp] * 7 + [C.c_int]
    lib.ndp_step_end.argtypes = [vp] * 6
    lib.ndp_debug_host_timing.argtypes = [vp, vp]
    lib.ndp_debug_downwash_stream_device.argtypes = [vp] * 6
    lib.ndp_downwash_prefetch_device.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp]
    lib.ndp_step_device_prefetched.argtypes = [vp] * 6
    lib.ndp_prefetch_join.argtypes = [vp, vp]
    lib.ndp_prefetch_stats.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    lib.ndp_device_force_slot.argtypes = [vp, C.c_int]
    lib.ndp_device_force_slot.restype = vp
    lib.ndp_step_device_ex.argtypes = [vp] * 6 + [C.c_int] + [vp] * 4
    lib.ndp_work_queue_enabled.argtypes = [vp]
    lib.ndp_refine_active.argtypes = [vp]
    lib.ndp_tick_config.argtypes = [vp, vp, C.c_int]
    lib.ndp_tick_reset.argtypes = [vp]
    lib.ndp_tick_begin.argtypes = [vp] * 5 + [C.c_int]
    lib.ndp_tick_end.argtypes = [vp] * 5
    lib.ndp_tick.argtypes = [vp] * 5 + [C.c_int] + [vp] * 4
    lib.ndp_tick_device.argtypes = [vp] * 5 + [C.c_int] + [vp] * 3
    lib.ndp_tick_config_remote.argtypes = [vp, vp, C.c_int, C.c_int64, vp, C.c_int]
    lib.ndp_tick_advance_device.argtypes = [vp] * 5 + [C.c_int, vp]
    lib.ndp_tick_window_pv_device.argtypes = [vp] * 3
    lib.ndp_tick_step_device.argtypes = [vp] * 5
    lib.ndp_xchg_tick_windows.argtypes = [vp] * 4
    lib.ndp_xchg_tick_begin.argtypes = [vp, vp, vp, C.c_int, vp]
    lib.ndp_xchg_tick_async.argtypes = [vp, C.c_int]
    lib.ndp_xchg_tick_step.argtypes = [vp] * 5 + [C.c_int] + [vp] * 4
    lib.ndp_get_active_set.argtypes = [vp] * 3
    lib.ndp_set_active_set.argtypes = [vp] * 2
    lib.ndp_debug_host_info.argtypes = [vp, vp]
    lib.ndp_ref_list_reset.argtypes = [vp]
    lib.ndp_ref_list_fix_pt.argtypes = [vp, vp, C.c_int]
    lib.ndp_ref_list_window.argtypes = [vp] * 4
    lib.ndp_ref_list_advance_device.argtypes = [vp] * 3
    lib.ndp_ref_list_window_device.argtypes = [vp] * 4
    lib.ndp_downwash.argtypes = [vp] * 5
    lib.ndp_downwash_device.argtypes = [vp] * 6
    lib.ndp_get_iterate.argtypes = [vp, vp, vp]
    lib.ndp_set_iterate.argtypes = [vp, vp, vp]
    lib.ndp_get_status.argtypes = [vp, vp, vp]
    for name in ("ndp_device_iterate_x", "ndp_device_iterate_u", "ndp_device_force"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = vp
    lib.ndp_synchronize.argtypes = [vp]
    lib.ndp_timing_enable.argtypes = [vp, C.c_int]
    lib.ndp_timing_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.ndp_debug_lds_doubles.argtypes = [C.c_int]
    lib.ndp_debug_lds_layout.argtypes = [C.c_int, vp]
    lib.ndp_step_debug.argtypes = [vp] * 9
    lib.ndp_debug_mfma_probe.argtypes = [vp] * 4
    lib.ndp_debug_mfma_probe_f32.argtypes = [vp] * 4 + [C.c_int]
    lib.ndp_peer_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp), vp]
    lib.ndp_peer_open.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.ndp_peer_close.argtypes = [C.c_int, vp]
    lib.ndp_peer_free.argtypes = [C.c_int, vp]
    lib.ndp_peer_layout.argtypes = [C.c_size_t] + [C.POINTER(C.c_size_t)] * 3
    lib.ndp_peer_publish_device.argtypes = [C.c_int, vp, C.c_size_t, vp, vp, C.c_int, C.c_uint, vp]
    lib.ndp_peer_stats.argtypes = [C.c_int, vp, C.POINTER(C.c_ulonglong)]
    lib.ndp_xchg_unique_id.argtypes = [C.c_char_p, vp]
    lib.ndp_xchg_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_char_p, C.POINTER(vp)]
    lib.ndp_xchg_begin.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
    lib.ndp_track_steps.argtypes = [vp, C.c_int]
    lib.ndp_last_step_event.argtypes = [vp, C.POINTER(vp)]
    lib.ndp_xchg_end.argtypes = [vp, vp]
    lib.ndp_xchg_tick.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    lib.ndp_xchg_last_error.argtypes = [vp]
    lib.ndp_xchg_last_error.restype = C.c_char_p
    lib.ndp_xchg_destroy.argtypes = [vp]
    lib.ndp_debug_stamps.argtypes = [vp, C.c_int, vp]
    lib.ndp_throttle_reset.argtypes = [vp]
    lib.ndp_throttle_update.argtypes = [vp] * 4
    lib.ndp_throttle_update_device.argtypes = [vp] * 5
    lib.ndp_actuator_cmd.argtypes = [vp] * 4
    lib.ndp_actuator_cmd_device.argtypes = [vp] * 5
    lib.ndp_throttle_get_state.argtypes = [vp, vp]
    lib.ndp_relay_reset.argtypes = [vp]
    lib.ndp_relay_formation.argtypes = [vp, vp, vp]
    lib.ndp_relay_reference.argtypes = [vp, vp, vp]
    lib.ndp_relay_reference_device.argtypes = [vp, vp, vp, vp]
    lib.ndp_ref_set_trajectory.argtypes = [vp, C.c_int] + [vp] * 7
    lib.ndp_ref_window.argtypes = [vp] * 4
    lib.ndp_ref_window_device.argtypes = [vp] * 5
    lib.ndp_rollout_device.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_int, vp, vp, vp]
    lib.ndp_plant_step.argtypes = [vp, vp, vp, vp, C.c_double, C.c_int]
    lib.ndp_plant_step_device.argtypes = [vp, vp, vp, vp, C.c_double, C.c_int, vp]
    _lib = lib
    return lib


def lds_layout(N):
    """Offsets (doubles) inside the debug dump: dict XI, MB, CB, MB_STRIDE, CB_STRIDE, total, stamps."""
    out = (C.c_int * 8)()
    load().ndp_debug_lds_layout(int(N), out)
    return dict(zip(("XI", "MB", "CB", "MB_STRIDE", "CB_STRIDE", "total", "stamps"), list(out)[:7]))


def default_cfg(**kw):
    cfg = NdpCfg()
    load().ndp_default_cfg(C.byref(cfg))
    for k, v in kw.items():
        cur = getattr(cfg, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(cfg, k, v)
    return cfg


def load_weights(path=WEIGHTS_PATH):
    w = np.fromfile(path, dtype="<f4")
    if w.size != MLP_NPARAM:
        raise ValueError(f"{path}: expected {MLP_NPARAM} fp32 values, found {w.size}")
    return w


def ptr(a):
    """numpy array -> void* (None passes NULL)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a
