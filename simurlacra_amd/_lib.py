"""ctypes binding of libvecsim (include/vecsim.h). Fails loudly when the HIP library is missing: no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VS_LIB_PATH") or os.path.join(_HERE, "csrc", "libvecsim.so")  # override: diagnostic builds

VS_OK, VS_ERR_ARG, VS_ERR_HIP, VS_ERR_STATE, VS_ERR_NAN = 0, -1, -2, -3, -4
ENV_TYPES = {"omo": 0, "bob": 1, "qq-su": 2, "qcp-su": 3, "qbb": 4, "qq-st": 5, "qcp-st": 6, "pend": 7, "bob-d": 8}
(VS_STATE, VS_OBS, VS_REW, VS_DONE, VS_HIDDEN, VS_STEPCOUNT, VS_ERRFLAG, VS_RETURNS, VS_PARAMS, VS_CONSTS,
 VS_EP_RETURNS, VS_EP_LENGTHS, VS_EP_ENVIDX, VS_EP_COUNT, VS_TRAJ_REC, _VS_RESERVED_15, _VS_RESERVED_16, VS_TRAJ_DONE,
 VS_FAILED, VS_EPSTAT_COUNT, VS_EPSTAT_RETSUM, VS_EPSTAT_LENSUM, VS_JAC_STATE, VS_JAC_REW, VS_JAC_OBS) = range(25)
VS_FLAG_SIMPLE_DYNAMICS, VS_FLAG_LONG_POLE, VS_FLAG_ACT_NORM, VS_FLAG_FREEZE_DONE, VS_FLAG_LEAN_STEP = 1, 2, 4, 8, 16
RV_PLAIN, RV_WS256, RV_WS64, RV_WS64G, RV_WS256G = 0, 1, 2, 3, 4  # vs_rollout_variant
VS_NL_NONE, VS_NL_TANH, VS_NL_RELU, VS_NL_SIGMOID = 0, 1, 2, 3  # vs_fnn_desc nonlinearities
VS_FNN_MAX_HIDDEN, VS_FNN_MAX_WIDTH = 4, 64
VS_DP_NORMAL, VS_DP_UNIFORM, VS_DP_BERNOULLI = 0, 1, 2
VS_MAX_ACT_DELAY = 64


class TaskCfg(C.Structure):
    _fields_ = [("use_defaults", C.c_int32), ("flags", C.c_int32), ("wild_init", C.c_int32), ("reserved", C.c_int32),
                ("state_des", C.c_float * 8), ("q_diag", C.c_float * 8), ("r_diag", C.c_float * 2),
                ("init_state", C.c_float * 8)]


class DpSpec(C.Structure):
    _fields_ = [("param_index", C.c_int32), ("kind", C.c_int32), ("mean", C.c_float), ("spread", C.c_float),
                ("clip_lo", C.c_float), ("clip_up", C.c_float), ("aux", C.c_float), ("roundint", C.c_int32)]


class FnnDesc(C.Structure):
    _fields_ = [("n_hidden", C.c_int32), ("hidden", C.c_int32 * 4), ("hidden_nonlin", C.c_int32 * 4),
                ("output_nonlin", C.c_int32), ("feat", C.c_int32), ("n_obs", C.c_int32), ("obs_idx", C.c_int32 * 8),
                ("noise_std", C.c_float * 2)]


_P = C.c_void_p
_SIGNATURES = {
    "vs_version": (C.c_int, []),
    "vs_traj_layout": (C.c_int, [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4),
    "vs_env_dims": (C.c_int, [C.c_int] + [C.POINTER(C.c_int)] * 7),
    "vs_env_name": (C.c_char_p, [C.c_int]),
    "vs_param_name": (C.c_char_p, [C.c_int, C.c_int]),
    "vs_nominal_params": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "vs_create": (C.c_int, [C.c_int, C.c_int64, C.c_double, C.c_int64, C.c_int, C.POINTER(TaskCfg), C.POINTER(_P)]),
    "vs_destroy": (C.c_int, [_P]),
    "vs_set_max_steps": (C.c_int, [_P, C.c_int64]),
    "vs_set_dt": (C.c_int, [_P, C.c_double]),
    "vs_set_stream": (C.c_int, [_P, _P]),
    "vs_sync": (C.c_int, [_P]),
    "vs_n_envs": (C.c_int64, [_P]),
    "vs_ld": (C.c_int64, [_P]),
    "vs_last_error": (C.c_char_p, [_P]),
    "vs_set_params": (C.c_int, [_P, _P, C.c_int64, _P]),
    "vs_set_params_uniform": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "vs_sample_params": (C.c_int, [_P, C.POINTER(DpSpec), C.c_int, C.c_uint64, _P]),
    "vs_set_randomizer": (C.c_int, [_P, C.POINTER(DpSpec), C.c_int]),
    "vs_set_param_buffer": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "vs_set_act_norm": (C.c_int, [_P, C.c_int]),
    "vs_set_act_pipeline": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                      C.c_uint64]),
    "vs_set_obs_pipeline": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint64]),
    "vs_reset": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, C.c_uint64]),
    "vs_set_index_offset": (C.c_int, [_P, C.c_uint32]),
    "vs_set_auto_reset": (C.c_int, [_P, C.c_int, C.c_uint64]),
    "vs_step": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "vs_step_jac": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "vs_step_record": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int]),
    "vs_set_record_row": (C.c_int, [_P, C.c_int]),
    "vs_step_random": (C.c_int, [_P, C.c_uint64, C.c_int, C.c_int]),
    "vs_seek_random": (C.c_int, [_P, C.c_uint64]),
    "vs_set_policy_fnn": (C.c_int, [_P, C.POINTER(FnnDesc), _P, C.c_int64]),
    "vs_step_policy": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64]),
    "vs_set_policy_shape": (C.c_int, [_P, C.c_int]),
    "vs_rollout_lengths": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "vs_pack_traj": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    "vs_rollout_variant": (C.c_int, [_P]),
    "vs_set_rollout_variant": (C.c_int, [_P, C.c_int]),
    "vs_set_traj_capacity": (C.c_int, [_P, C.c_int]),
    "vs_set_traj_offset": (C.c_int, [_P, C.c_int]),
    "vs_set_record_mode": (C.c_int, [_P, C.c_int]),
    "vs_record_mode": (C.c_int, [_P]),
    "vs_set_freeze_done": (C.c_int, [_P, C.c_int]),
    "vs_set_lean_step": (C.c_int, [_P, C.c_int]),
    "vs_set_episode_log": (C.c_int, [_P, C.c_int]),
    "vs_clear_episodes": (C.c_int, [_P]),
    "vs_mixed_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(_P)]),
    "vs_mixed_destroy": (C.c_int, [_P]),
    "vs_mixed_last_error": (C.c_char_p, [_P]),
    "vs_mixed_step_random": (C.c_int, [_P, C.c_uint64, C.c_int, C.c_int]),
    "vs_mixed_step": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vs_mixed_time_random": (C.c_int, [_P, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "vs_get": (_P, [_P, C.c_int]),
    "vs_copy_to_host": (C.c_int, [_P, C.c_int, _P]),
    "vs_copy_from_host": (C.c_int, [_P, C.c_int, _P]),
    "vs_error_count": (C.c_int64, [_P]),
    "vs_time_step_kernel": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_float)]),
    "vs_timer_start": (C.c_int, [_P]),
    "vs_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "vs_membw_probe": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_float)]),
    "vs_memwrite_probe": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_float)]),
}

_lib = None


class VecSimLibraryError(ImportError):
    pass


def load():
    """Load libvecsim.so (built in-tree by simurlacra_amd/csrc/build.py). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under the SAME sonames as the system ROCm that
    # libvecsim.so was linked against.  Whichever is loaded first serves both: with torch first, libvecsim shares torch's
    # runtime (one runtime in the process: device pointers, streams and events are interchangeable); the other way round
    # torch would be handed the system runtime it was not built for and reports "No HIP GPUs are available".
    # A plain C host never loads torch and uses the system runtime.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise VecSimLibraryError(
            f"{LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950). simurlacra_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)
