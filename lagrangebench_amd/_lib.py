"""ctypes binding of liblbhip.so (include/lbhip.h).  There is NO CPU fallback: if the HIP
library is missing or fails to load, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblbhip.so")

LB_OK = 0
LB_FORCE_NONE, LB_FORCE_PIECEWISE, LB_FORCE_BUFFER = 0, 1, 2

D3 = C.c_double * 3


class CaseDesc(C.Structure):
    """lb_case_desc (include/lbhip.h)."""

    _fields_ = [
        ("dim", C.c_int32), ("n_particles", C.c_int32), ("batch", C.c_int32), ("isl", C.c_int32),
        ("periodic", C.c_int32), ("has_bound", C.c_int32), ("has_vel_mag", C.c_int32),
        ("force_kind", C.c_int32), ("force_axis", C.c_int32), ("geometry_f32", C.c_int32),
        ("box", D3), ("r_cutoff", C.c_double), ("capacity_multiplier", C.c_double),
        ("vel_mean", D3), ("vel_std", D3), ("acc_mean", D3), ("acc_std", D3),
        ("bound_lo", D3), ("bound_hi", D3),
        ("force_split", C.c_double), ("force_lo", D3), ("force_hi", D3),
    ]


class GnsDesc(C.Structure):
    """lb_gns_desc (include/lbhip.h)."""

    _fields_ = [
        ("latent_size", C.c_int32), ("blocks_per_step", C.c_int32), ("num_mp_steps", C.c_int32),
        ("embedding_size", C.c_int32), ("num_particle_types", C.c_int32), ("node_in", C.c_int32),
        ("edge_in", C.c_int32), ("out_dim", C.c_int32),
    ]


class SegnnDesc(C.Structure):
    """lb_segnn_desc (include/lbhip.h)."""

    _fields_ = [
        ("hidden", C.c_int32), ("blocks_per_step", C.c_int32), ("num_mp_steps", C.c_int32),
        ("homogeneous", C.c_int32), ("n_vels", C.c_int32), ("velocity_avg", C.c_int32),
        ("lmax_hidden", C.c_int32), ("lmax_attributes", C.c_int32), ("norm", C.c_int32), ("norm_eps", C.c_float),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if "lmax_hidden" not in kw and len(args) < 7:
            self.lmax_hidden = 1
        if "lmax_attributes" not in kw and len(args) < 8:
            self.lmax_attributes = 1


# name -> (restype, argtypes).  Every symbol include/lbhip.h declares must be listed here;
# tests/test_abi.py checks the two against each other.
_P = C.c_void_p
_SIGS = {
    "lb_strerror": (C.c_char_p, [C.c_int]),
    "lb_last_error": (C.c_char_p, []),
    "lb_version": (C.c_int, []),
    "lb_engine_create": (C.c_int, [C.POINTER(CaseDesc), _P, C.POINTER(_P)]),
    "lb_engine_destroy": (None, [_P]),
    "lb_set_particle_type": (C.c_int, [_P, _P]),
    "lb_set_force": (C.c_int, [_P, _P]),
    "lb_load_window": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32]),
    "lb_read_window": (C.c_int, [_P, _P]),
    "lb_nl_allocate": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lb_nl_set_capacity": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "lb_nl_update": (C.c_int, [_P]),
    "lb_nl_read_flags": (C.c_int, [_P, _P]),
    "lb_nl_read_idx": (C.c_int, [_P, _P, _P]),
    "lb_node_features": (C.c_int, [_P, _P, _P, _P, _P]),
    "lb_edge_features": (C.c_int, [_P, _P, _P]),
    "lb_gns_create": (C.c_int, [_P, C.POINTER(GnsDesc), _P, C.c_int64, C.POINTER(_P)]),
    "lb_gns_destroy": (None, [_P]),
    "lb_gns_forward": (C.c_int, [_P, _P, _P]),
    "lb_set_fused_aggregation": (C.c_int, [_P, C.c_int32]),
    "lb_gns_set_tap": (C.c_int, [_P, _P]),
    "lb_math_mode": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lb_math_fallbacks": (C.c_int32, [_P]),
    "lb_debug_inject_guard": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "lb_integrate": (C.c_int, [_P, _P, _P, _P, C.c_int32]),
    "lb_case_integrate": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P]),
    "lb_rollout": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.POINTER(C.c_int32)]),
    "lb_ekin": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_double, C.c_double, _P, C.c_int32]),
    "lb_sinkhorn": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_double, _P, C.c_int32, _P]),
    "lb_sinkhorn_pot": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_double, _P,
                                  C.c_int32, _P]),
    "lb_metrics": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    "lb_timers_enable": (C.c_int, [_P, C.c_int32]),
    "lb_timers_reset": (C.c_int, [_P]),
    "lb_timer_count": (C.c_int32, []),
    "lb_timer_name": (C.c_char_p, [C.c_int32]),
    "lb_timer_get": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "lb_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lb_edge_accounting": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.c_int32]),
    "lb_kernel_names": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "lb_gns_train_create": (C.c_int, [_P, C.POINTER(GnsDesc), C.POINTER(C.c_float), C.c_int64, C.POINTER(_P)]),
    "lb_gns_train_destroy": (None, [_P]),
    "lb_gns_train_loss_grad": (C.c_int, [_P, _P, C.c_float, C.POINTER(C.c_double), _P]),
    "lb_gns_train_zero_grad": (C.c_int, [_P]),
    "lb_adamw_step": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "lb_gns_train_read": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_float), C.c_int64]),
    "lb_gns_train_write": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_float), C.c_int64, C.c_int64]),
    "lb_gns_train_step_count": (C.c_int64, [_P]),
    "lb_gns_train_math_fallbacks": (C.c_int32, [_P]),
    "lb_segnn_train_create": (C.c_int, [_P, C.POINTER(SegnnDesc), C.POINTER(C.c_float), C.c_int64, C.POINTER(_P)]),
    "lb_segnn_train_loss_grad": (C.c_int, [_P, _P, C.c_float, C.POINTER(C.c_double), _P]),
    "lb_segment_sum": (C.c_int, [_P, _P, _P, C.c_int32]),
    "lb_segnn_create": (C.c_int, [_P, C.POINTER(SegnnDesc), _P, C.c_int64, C.POINTER(_P)]),
    "lb_segnn_destroy": (None, [_P]),
    "lb_segnn_forward": (C.c_int, [_P, _P, _P]),
    "lb_segnn_row_floats": (C.c_int32, [_P]),
    "lb_segnn_set_tap": (C.c_int, [_P, _P]),
    "lb_segnn_rollout": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.POINTER(C.c_int32)]),
}

_lib: Optional[C.CDLL] = None


class LbHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load liblbhip.so and bind the prototypes.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LbHipError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run "
            "`python -m lagrangebench_amd.build` (needs hipcc). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != LB_OK:
        lib = load()
        msg = lib.lb_last_error().decode() or lib.lb_strerror(rc).decode()
        raise LbHipError(f"{what or 'liblbhip'} failed ({rc}: {lib.lb_strerror(rc).decode()}): {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (or None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()
