"""ctypes binding of include/mpc_amd.h (the drop-in C ABI)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

MAX_TABLE = 16
NW = 12
NCOEF = 5
NSTATE = 6
NOUT = 9
MAX_N = 64
ABI_VERSION = 4
PRECISION_F64, PRECISION_F32 = 0, 1

STATUS_NAMES = {0: "success", 1: "maxiter", 2: "linesearch", 3: "infeasible", 4: "numeric", 5: "pending", 6: "acceptable"}
ERR_NAMES = {0: "MPC_OK", -1: "MPC_ERR_INVALID", -2: "MPC_ERR_NO_DEVICE", -3: "MPC_ERR_HIP",
             -4: "MPC_ERR_UNSUPPORTED", -5: "MPC_ERR_IO"}


class MpcError(RuntimeError):
    pass


class MpcParams(C.Structure):
    """Mirror of ``struct MpcParams`` (include/mpc_amd.h); field order is ABI."""
    _fields_ = [
        ("abi_version", C.c_int32), ("N", C.c_int32), ("dt", C.c_double), ("Lf", C.c_double),
        ("weights", C.c_double * NW), ("cte_panic", C.c_double), ("epsi_panic", C.c_double),
        ("max_steering", C.c_double), ("max_acceleration", C.c_double), ("max_deceleration", C.c_double),
        ("max_speed", C.c_double), ("n_steers", C.c_int32), ("n_steer_speeds", C.c_int32),
        ("steers", C.c_double * MAX_TABLE), ("steer_speeds", C.c_double * MAX_TABLE),
        ("n_yaw_changes", C.c_int32), ("n_yaw_change_speeds", C.c_int32),
        ("yaw_changes", C.c_double * MAX_TABLE), ("yaw_change_speeds", C.c_double * MAX_TABLE),
        ("max_fit_order", C.c_int32), ("latency_ms", C.c_int32), ("max_fit_error", C.c_double),
        ("lookahead", C.c_double), ("steer_adj_thresh", C.c_double), ("steer_adj_ratio", C.c_double),
        ("ipopt_timeout", C.c_double), ("branch_mode", C.c_int32), ("precision", C.c_int32),
        ("max_iter", C.c_int32), ("pass_cut", C.c_int32), ("tol", C.c_double),
        ("out_step_tol", C.c_double), ("tol_f32", C.c_double), ("polish", C.c_int32),
        ("pass_cut_next", C.c_int32 * 3), ("honor_original_bounds", C.c_int32), ("bound_relax_factor", C.c_double),
        ("tail_cut", C.c_int32), ("tail_ring", C.c_int32), ("tail_capacity", C.c_int64),
        ("f32_finish", C.c_int32), ("f64_f32_start", C.c_int32), ("mixed_switch_mu", C.c_double),
        ("lane_compact", C.c_int32), ("f32_phase_refill", C.c_int32), ("acceptable_iter", C.c_int32),
        ("dual_inf_tol", C.c_double), ("constr_viol_tol", C.c_double), ("compl_inf_tol", C.c_double),
        ("acceptable_tol", C.c_double), ("acceptable_dual_inf_tol", C.c_double),
        ("acceptable_constr_viol_tol", C.c_double), ("acceptable_compl_inf_tol", C.c_double), ("initial_state_rows", C.c_int32), ("wave_max_batch", C.c_int32),
    ]

    def copy(self):
        q = MpcParams()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(MpcParams))
        return q


class MpcWireTelemetry(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("psi", C.c_double), ("speed", C.c_double),
                ("steering_angle", C.c_double), ("throttle", C.c_double), ("npts", C.c_int32), ("reserved", C.c_int32),
                ("ptsx", C.c_double * 8), ("ptsy", C.c_double * 8)]


class MpcBatchStats(C.Structure):
    _fields_ = [("batch", C.c_int64), ("n_success", C.c_int64), ("n_maxiter", C.c_int64),
                ("n_linesearch", C.c_int64), ("n_infeasible", C.c_int64), ("n_numeric", C.c_int64), ("n_acceptable", C.c_int64),
                ("iter_sum", C.c_int64), ("iter_max", C.c_int32), ("n_pending", C.c_int32),
                ("kernel_ms", C.c_double)]


# every symbol include/mpc_amd.h declares (checked by tests/test_abi.py)
EXPORTS = ["mpc_params_default", "mpc_params_load_json", "mpc_create", "mpc_set_params", "mpc_destroy",
           "mpc_last_error", "mpc_abi_version", "mpc_solve_batch_device", "mpc_solve_batch_host",
           "mpc_synchronize", "mpc_get_stats", "mpc_debug_math", "mpc_run_batch_device",
           "mpc_telemetry_batch_device", "mpc_rollout_batch_device", "mpc_debug_math_ext",
           "mpc_solve_batch_device_f32", "mpc_wire_parse", "mpc_wire_format_steer", "mpc_wire_format_manual",
           "mpc_wire_telemetry_batch_host", "mpc_debug_tile_pool", "mpc_telemetry_batch_host", "mpc_handle_device",
           "mpc_run_batch_host", "mpc_last_batch_id", "mpc_tail_poll", "mpc_tail_wait", "mpc_tail_stream_wait", "mpc_tail_flush", "mpc_tail_pending", "mpc_tail_info", "mpc_solve_batch_host_f32", "mpc_inflight_advice"]

_lib = None


def library_path():
    return os.path.join(HERE, "lib", "libmpc_amd.so")


def build_library(verbose=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return library_path()


def library():
    """Load the product library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise MpcError("HIP extension %s is missing: run __graft_entry__.build() (there is no CPU fallback)" % path)
    # One HIP runtime per process: torch wheels bundle their own libamdhip64 (same soname as /opt/rocm's).  If this
    # library were loaded first it would bind /opt/rocm's copy, and torch, imported later, would then fail to see a
    # device.  Importing torch first makes both use the copy torch ships.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    DP = C.c_void_p
    L.mpc_params_default.argtypes = [C.POINTER(MpcParams)]
    L.mpc_params_load_json.argtypes = [C.c_char_p, C.POINTER(MpcParams)]
    L.mpc_create.argtypes = [C.POINTER(MpcParams), C.c_int, C.c_int64, C.POINTER(C.c_void_p)]
    L.mpc_set_params.argtypes = [C.c_void_p, C.POINTER(MpcParams)]
    L.mpc_destroy.argtypes = [C.c_void_p]
    L.mpc_destroy.restype = None
    L.mpc_last_error.restype = C.c_char_p
    L.mpc_abi_version.restype = C.c_int
    L.mpc_solve_batch_device.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [DP] * 9 + [C.c_void_p]
    L.mpc_solve_batch_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [DP] * 9
    L.mpc_solve_batch_host_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [DP] * 9
    L.mpc_solve_batch_device_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [DP] * 9 + [C.c_void_p]
    L.mpc_synchronize.argtypes = [C.c_void_p]
    L.mpc_get_stats.argtypes = [C.c_void_p, C.POINTER(MpcBatchStats)]
    L.mpc_debug_math.argtypes = [C.c_int, C.c_int64] + [C.c_void_p] * 4
    L.mpc_debug_math_ext.argtypes = [C.c_int, C.c_int64] + [C.c_void_p] * 6
    L.mpc_debug_tile_pool.argtypes = [C.c_void_p, C.c_void_p]
    L.mpc_run_batch_device.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int] + [DP] * 8 + [C.c_void_p]
    L.mpc_run_batch_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int] + [C.c_void_p] * 8
    L.mpc_telemetry_batch_device.argtypes = ([C.c_void_p, C.c_int64, C.c_int64, C.c_int, DP, C.c_double] + [DP] * 5 +
                                             [C.c_void_p])
    L.mpc_rollout_batch_device.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int] + [DP] * 8 + [C.c_void_p]
    L.mpc_telemetry_batch_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, DP, C.c_double] + [DP] * 4
    L.mpc_handle_device.argtypes = [C.c_void_p]
    L.mpc_inflight_advice.argtypes = [C.c_void_p, C.c_int64]
    L.mpc_inflight_advice.restype = C.c_int
    L.mpc_last_batch_id.argtypes = [C.c_void_p]
    L.mpc_last_batch_id.restype = C.c_int64
    L.mpc_tail_wait.argtypes = [C.c_void_p, C.c_int64]
    L.mpc_tail_poll.argtypes = [C.c_void_p, C.c_int64]
    L.mpc_tail_stream_wait.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.mpc_tail_flush.argtypes = [C.c_void_p]
    L.mpc_tail_pending.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.mpc_tail_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.mpc_wire_parse.argtypes = [C.c_char_p, C.c_int64, C.POINTER(MpcWireTelemetry)]
    L.mpc_wire_format_steer.argtypes = [C.c_double, C.c_double, C.c_char_p, C.c_int64]
    L.mpc_wire_format_steer.restype = C.c_int64
    L.mpc_wire_format_manual.argtypes = [C.c_char_p, C.c_int64]
    L.mpc_wire_format_manual.restype = C.c_int64
    L.mpc_wire_telemetry_batch_host.argtypes = [C.c_void_p, C.c_int64, C.POINTER(MpcWireTelemetry), DP, C.c_double, DP, DP]
    if L.mpc_abi_version() != ABI_VERSION:
        raise MpcError("ABI version mismatch between %s and the Python binding" % path)
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = library().mpc_last_error()
        raise MpcError("%s failed: %s (%s)" % (what, ERR_NAMES.get(rc, rc), msg.decode() if msg else ""))


def params_default():
    p = MpcParams()
    check(library().mpc_params_default(C.byref(p)), "mpc_params_default")
    return p


def inflight_advice(params, B):
    """mpc_inflight_advice: batches of B instances to keep in flight for these parameters."""
    return int(library().mpc_inflight_advice(C.byref(params), int(B)))


def params_from_json(path, **overrides):
    """Config::load(path) (src/utils/Config.cpp:31-87) through the C ABI; keyword overrides
    (e.g. N=25, dt=0.05) are applied afterwards like mpc_main.cpp's CLI overrides."""
    p = MpcParams()
    check(library().mpc_params_load_json(os.fspath(path).encode(), C.byref(p)), "mpc_params_load_json(%s)" % path)
    for k, v in overrides.items():
        if k == "weights":
            for i, w in enumerate(v):
                p.weights[i] = w
        else:
            setattr(p, k, v)
    return p
