"""ctypes binding of libsac_hip.so (the C ABI of include/sac_hip.h).

There is NO CPU fallback: if the shared library is missing this module raises, and every
handle constructor raises when no MI355X is visible."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsac_hip.so")

SAC_DIAG_N = 32
DIAG_NAMES = [
    "QF1 Loss", "QF2 Loss", "Policy Loss", "Actor Loss",
    "Q1 Predictions Mean", "Q1 Predictions Std", "Q1 Predictions Max", "Q1 Predictions Min",
    "Q2 Predictions Mean", "Q2 Predictions Std", "Q2 Predictions Max", "Q2 Predictions Min",
    "Q Targets Mean", "Q Targets Std", "Q Targets Max", "Q Targets Min",
    "Log Pis Mean", "Log Pis Std", "Log Pis Max", "Log Pis Min",
    "Policy mu Mean", "Policy mu Std", "Policy mu Max", "Policy mu Min",
    "Policy log std Mean", "Policy log std Std", "Policy log std Max", "Policy log std Min",
    "Alpha", "Alpha Loss",
]
NET_IDS = {"policy": 0, "qf1": 1, "qf2": 2, "target_qf1": 3, "target_qf2": 4}
TD3_NET_IDS = dict(NET_IDS, target_policy=5)


class SacConfig(C.Structure):
    _fields_ = [
        ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("hidden", C.c_int32), ("batch", C.c_int32),
        ("discount", C.c_float), ("reward_scale", C.c_float), ("policy_lr", C.c_float), ("qf_lr", C.c_float),
        ("soft_target_tau", C.c_float), ("target_update_period", C.c_int32),
        ("use_automatic_entropy_tuning", C.c_int32), ("target_entropy", C.c_float),
        ("noise_seed", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32),
        ("policy_hidden", C.c_int32 * 2), ("qf_hidden", C.c_int32 * 2),
    ]


class Td3Config(C.Structure):
    _fields_ = [
        ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("hidden", C.c_int32), ("batch", C.c_int32),
        ("discount", C.c_float), ("reward_scale", C.c_float), ("policy_learning_rate", C.c_float),
        ("qf_learning_rate", C.c_float), ("tau", C.c_float), ("target_policy_noise", C.c_float),
        ("target_policy_noise_clip", C.c_float), ("policy_and_target_update_period", C.c_int32),
        ("noise_seed", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32),
        ("policy_hidden", C.c_int32 * 2), ("qf_hidden", C.c_int32 * 2),
    ]


TD3_DIAG_NAMES = DIAG_NAMES[:16] + [f"Bellman Errors {i} {s}" for i in (1, 2) for s in ("Mean", "Std", "Max", "Min")] + [
    f"Policy Action {s}" for s in ("Mean", "Std", "Max", "Min")]

# every symbol include/sac_hip.h declares: (restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
SYMBOLS = {
    "sac_last_error": (C.c_char_p, []),
    "sac_device_count": (C.c_int, []),
    "sac_version": (C.c_char_p, []),
    "sac_buffer_create": (C.c_int, [C.POINTER(_P), C.c_int64, C.c_int, C.c_int, C.c_int]),
    "sac_buffer_destroy": (C.c_int, [_P]),
    "sac_buffer_add": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P]),
    "sac_buffer_add_f64": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P]),
    "sac_buffer_ingest_pending": (C.c_int, [_P]),
    "sac_buffer_ingest_wait": (C.c_int, [_P]),
    "sac_buffer_size": (C.c_int64, [_P]),
    "sac_buffer_top": (C.c_int64, [_P]),
    "sac_buffer_capacity": (C.c_int64, [_P]),
    "sac_buffer_read": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P]),
    "sac_buffer_set_cursor": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "sac_rng_seed": (C.c_int, [_P, C.c_uint32]),
    "sac_rng_get_state": (C.c_int, [_P, _P, C.POINTER(C.c_int32)]),
    "sac_rng_set_state": (C.c_int, [_P, _P, C.c_int32]),
    "sac_rng_bind_host": (C.c_int, [_P, _P, _P]),
    "sac_sample_indices": (C.c_int, [_P, C.c_int, C.c_int64, _P]),
    "sac_random_batch": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "sac_random_batch_device": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "sac_read_batch_device": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "sac_gather": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P]),
    "sac_sample_gather_device": (C.c_int, [_P, C.c_int, C.c_int64, _P]),
    "sac_read_slot": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "sac_trainer_create": (C.c_int, [C.POINTER(_P), C.POINTER(SacConfig)]),
    "sac_trainer_create_mlp": (C.c_int, [C.POINTER(_P), C.POINTER(SacConfig), _P, C.c_int32, _P, C.c_int32]),
    "td3_trainer_create": (C.c_int, [C.POINTER(_P), C.POINTER(Td3Config)]),
    "td3_trainer_create_mlp": (C.c_int, [C.POINTER(_P), C.POINTER(Td3Config), _P, C.c_int32, _P, C.c_int32]),
    "sac_trainer_destroy": (C.c_int, [_P]),
    "sac_param_count": (C.c_int64, [_P, C.c_int]),
    "sac_set_params": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "sac_get_params": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "sac_set_opt_state": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64]),
    "sac_get_opt_state": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64]),
    "sac_set_scalars": (C.c_int, [_P, _P]),
    "sac_get_scalars": (C.c_int, [_P, _P]),
    "sac_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sac_step_device": (C.c_int, [_P, _P, C.c_int64, _P]),
    "sac_train_loop": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "sac_sync": (C.c_int, [_P]),
    "sac_trainer_is_fused": (C.c_int, [_P]),
    "sac_trainer_step_kind": (C.c_int, [_P]),
    "sac_last_loop_ms": (C.c_int, [_P, _F, _F, _F, _F]),
    "sac_profile_loop": (C.c_int, [_P, _P, C.c_int64, _P]),
    "sac_measure_peaks": (C.c_int, [C.c_int, _F]),
    "sac_buffer_set_xcd": (C.c_int, [_P, C.c_int]),
    "sac_trainer_set_xcd": (C.c_int, [_P, C.c_int]),
    "sac_buffer_set_xcd_mask": (C.c_int, [_P, C.c_uint]),
    "sac_trainer_set_xcd_mask": (C.c_int, [_P, C.c_uint]),
    "sac_debug_fetch": (C.c_int64, [_P, C.c_char_p, _P, C.c_int64]),
    "sac_policy_mirror": (C.c_int, [_P]),
    "sac_policy_act": (C.c_int, [_P, _P, C.c_int, _P, _P]),
}

_lib = None


def load():
    """Load libsac_hip.so, binding every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m robosuite_benchmark_amd.build` "
            "(hipcc --offload-arch=gfx950).  robosuite_benchmark_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().sac_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc < 0:
        raise RuntimeError(f"{what} failed: {last_error()}")
    return rc


def device_count() -> int:
    return int(load().sac_device_count())


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
