"""ctypes binding of libaircraft_hip.so (include/aircraft_hip.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible, every
compute call raises.  (`tests -m "not gpu"` only check that the library loads and exports the ABI.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AIRCRAFT_HIP_LIB selects another build flavor of the same ABI (the diagnostic libaircraft_hip_diag.so)
LIB_PATH = os.environ.get("AIRCRAFT_HIP_LIB") or os.path.join(_HERE, "libaircraft_hip.so")

AC_OK = 0
STATUS_NAMES = {0: "AC_OK", -1: "AC_ERR_BAD_ARG", -2: "AC_ERR_HIP", -3: "AC_ERR_UNSUPPORTED",
                -4: "AC_ERR_NO_MODEL", -5: "AC_ERR_NO_DEVICE", -6: "AC_ERR_WORKSPACE"}
MODEL_KINDS = {"default": 0, "linear": 1, "nn": 2, "poly": 3, "quad": 4}
NUM_STATES = 13
NUM_CONTROLS = 7
AERO_ROWS = 22
MAX_LAYERS = 8
MAX_WIDTH = 128


class AcParams(C.Structure):
    """struct ac_params (include/aircraft_hip.h)."""
    _fields_ = [
        ("mass", C.c_float), ("S", C.c_float), ("b", C.c_float), ("c", C.c_float),
        ("inertia", C.c_float * 9),
        ("inertia_inv", C.c_float * 9),
        ("com", C.c_float * 3),
        ("rudder_moment_arm", C.c_float),
        ("epsilon", C.c_float),
        ("gravity", C.c_float * 3),
        ("substeps", C.c_int), ("normalise", C.c_int), ("stall_scaling", C.c_int), ("model_kind", C.c_int),
    ]


class IlqrCost(C.Structure):
    """struct ac_ilqr_cost (include/aircraft_hip.h)."""
    _fields_ = [("q", C.c_float * 13), ("qf", C.c_float * 13), ("r", C.c_float * 7), ("x_ref", C.c_float * 13),
                ("x_goal", C.c_float * 13), ("u_min", C.c_float * 7), ("u_max", C.c_float * 7), ("reg", C.c_float),
                ("u_lin", C.c_float * 7), ("dt_row", C.c_int)]


class GoalLoss(C.Structure):
    """struct ac_goal_loss (include/aircraft_hip.h); defaults are Controller.loss, main/control/control.py:44-68."""
    _fields_ = [("w_goal", C.c_float), ("w_rate", C.c_float), ("eps_rate", C.c_float), ("w_height", C.c_float),
                ("w_speed", C.c_float), ("w_vx", C.c_float), ("w_vyz", C.c_float), ("vx_max", C.c_float), ("w_al", C.c_float),
                ("time_row", C.c_int)]


class EnvelopePenalty(C.Structure):
    """struct ac_envelope_penalty (include/aircraft_hip.h)."""
    _fields_ = [("lo", C.c_float * 4), ("hi", C.c_float * 4), ("weight", C.c_float)]


class MhttWeights(C.Structure):
    """struct ac_mhtt_weights (include/aircraft_hip.h); defaults are moving_horizon.py:47-55."""
    _fields_ = [("w_tracking", C.c_float), ("w_progress", C.c_float), ("w_progress_rate", C.c_float),
                ("w_backward", C.c_float), ("w_terminal_align", C.c_float), ("w_low_velocity", C.c_float),
                ("w_control", C.c_float)]


# every symbol include/aircraft_hip.h declares, with its prototype
_FP = C.POINTER(C.c_float)
_VP = C.c_void_p
PROTOTYPES = {
    "ac_create": (C.c_int, [C.POINTER(AcParams), C.POINTER(_VP)]),
    "ac_destroy": (C.c_int, [_VP]),
    "ac_set_params": (C.c_int, [_VP, C.POINTER(AcParams)]),
    "ac_set_linear": (C.c_int, [_VP, _FP]),
    "ac_set_poly": (C.c_int, [_VP, _FP, _FP]),
    "ac_set_mlp": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_FP), C.POINTER(_FP),
                             _FP, _FP, _FP, _FP, C.c_int]),
    "ac_state_derivative_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, _VP, _VP]),
    "ac_step_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, C.c_long, _VP, _VP]),
    "ac_rollout_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, C.c_long, C.c_long, _VP, _VP]),
    "ac_step_sens_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, C.c_long, _VP, _VP, _VP, _VP, _VP]),
    "ac_shoot_step_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_shoot_sens_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, C.c_long, C.c_long, _VP, _VP, _VP, _VP, _VP]),
    "ac_state_derivative_sens_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, _VP, _VP, _VP, _VP]),
    "ac_shoot_derivative_sens_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP, _VP]),
    "ac_shoot_derivative_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_shoot_defect_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_shoot_implicit_defect_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_shoot_implicit_rows_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, C.c_long, C.c_long, _VP, _VP, _VP, _VP, _VP]),
    "ac_envelope_f32": (C.c_int, [_VP, _VP, C.c_long, _VP, _VP, _VP]),
    "ac_shoot_envelope_f32": (C.c_int, [_VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP]),
    "ac_envelope_cost_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_envelope_model_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP]),
    "ac_envelope_al_cost_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_envelope_al_model_f32": (C.c_int, [_VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP]),
    "ac_envelope_al_update_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP]),
    "ac_quat_rows_f32": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP, _VP]),
    "ac_step_hess_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, _VP, C.c_long, _VP, _VP]),
    "ac_reserve_hess_workspace": (C.c_int, [_VP, C.c_long]),
    "ac_shoot_hess_f32": (C.c_int, [_VP, _VP, _VP, C.c_float, _VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_aero_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, _VP, _VP]),
    "ac_traj_cost_f32": (C.c_int, [_VP, _VP, C.c_long, C.c_long, _FP, C.c_float, C.c_float, _VP, _VP]),
    "ac_best_records_f32": (C.c_int, [_VP, _VP, _VP, _VP, C.c_long, C.c_long, C.c_int, _VP, _VP]),
    "ac_merge_records_f32": (C.c_int, [_VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_ilqr_accept_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int, C.c_long, C.c_long, _VP, _VP, _VP, _VP, _VP]),
    "ac_ilqr_backward_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP, _VP]),
    "ac_ilqr_cost_f32": (C.c_int, [_VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_ilqr_backward_node_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP,
                                            _VP]),
    "ac_ilqr_costate_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_ilqr_backward_newton_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP,
                                              _VP, _VP, _VP]),
    "ac_ilqr_cost_node_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_long, _VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_goal_cost_f32": (C.c_int, [_VP, _VP, _VP, _VP, C.c_long, _VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_goal_model_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ac_goal_multiplier_f32": (C.c_int, [_VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP, _VP]),
    "ac_ilqr_backward_goal_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP,
                                            _VP, _VP]),
    "ac_set_track": (C.c_int, [_VP, C.c_int, _FP, C.c_float]),
    "ac_track_eval_f32": (C.c_int, [_VP, _VP, C.c_long, _VP, _VP, _VP]),
    "ac_track_progress_f32": (C.c_int, [_VP, _VP, _VP, _VP, C.c_float, C.c_long, C.c_long, C.c_int, _VP, _VP, _VP, _VP,
                                        _VP, _VP, _VP]),
    "ac_mhtt_loss_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_long, C.c_long, _VP, _VP]),
    "ac_rollout_policy_f32": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _FP, C.c_int, C.c_float, C.c_long, C.c_long,
                                        _VP, _VP, _VP]),
    "ac_last_error": (C.c_char_p, []),
    "ac_version": (C.c_char_p, []),
    "ac_device_arch": (C.c_int, [C.c_char_p, C.c_size_t]),
    "ac_hess_workspace": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_size_t)]),
    "ac_last_launch": (C.c_int, [_VP, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)]),
}

_lib = None


class AircraftHipError(RuntimeError):
    """Raised for any non-zero ac_status — the reference's failures surface as RuntimeError too
    (control/base.py:471-474)."""


def load():
    """Load libaircraft_hip.so; raise (never fall back) if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AircraftHipError(
                f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). aircraft_amd has no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != AC_OK:
        msg = load().ac_last_error().decode()  # cleared at the top of every entry point: never a stale text
        raise AircraftHipError(f"{what or 'aircraft_hip call'} failed: {STATUS_NAMES.get(rc, rc)} {msg}".strip())
