"""ctypes binding of libbbx.so (the C ABI in include/bbx.h).

There is no fallback: if the HIP extension has not been built (see __graft_entry__.build) the
import fails loudly, and every call fails with BBX_E_DEVICE when no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbbx.so")

BBX_GEBAUERMOELLER, BBX_LCM, BBX_NONE = 0, 1, 2
BBX_ADDITIONS, BBX_REDUCTIONS = 0, 1
AGENTS = {"external": 0, "random": 1, "degree": 2, "first": 3, "normal": 4, "sugar": 5,
          "last": 6, "codegree": 7, "strange": 8, "spice": 9, "random_std": 10}
ELIMINATION = {"gebauermoeller": 0, "lcm": 1, "none": 2}
REWARDS = {"additions": 0, "reductions": 1}


class BbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bbx error %d: %s" % (code, msg))
        self.code = code


class Caps(C.Structure):
    _fields_ = [("max_basis", C.c_int32), ("max_pairs", C.c_int32), ("arena_terms", C.c_int32),
                ("max_poly_terms", C.c_int32), ("queue_slots", C.c_int32), ("lds_max_basis", C.c_int32),
                ("wide_waves", C.c_int32), ("general_class", C.c_int32), ("wide_lds_terms", C.c_int32),
                ("no_growth", C.c_int32)]


class TraceRec(C.Structure):
    _fields_ = [("action", C.c_int32), ("rows", C.c_int32), ("basis_size", C.c_int32), ("done", C.c_int32),
                ("reward", C.c_double), ("obs_hash", C.c_uint64), ("pairs_hash", C.c_uint64), ("newpoly_hash", C.c_uint64)]


TRACE_DTYPE = np.dtype([("action", "<i4"), ("rows", "<i4"), ("basis_size", "<i4"), ("done", "<i4"),
                        ("reward", "<f8"), ("obs_hash", "<u8"), ("pairs_hash", "<u8"), ("newpoly_hash", "<u8")])

_vp = C.c_void_p
_i32p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/bbx.h declares
SIGNATURES = {
    "bbx_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Caps), C.POINTER(_vp)]),
    "bbx_create_fixed": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Caps), C.POINTER(_vp)]),
    "bbx_create_ideals": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Caps), C.POINTER(_vp)]),
    "bbx_destroy": (None, [_vp]),
    "bbx_copy": (C.c_int, [_vp, C.POINTER(_vp)]),
    "bbx_clone_envs": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "bbx_seed": (C.c_int, [_vp, _vp]),
    "bbx_seed_agent": (C.c_int, [_vp, _vp]),
    "bbx_seed_strategy": (C.c_int, [_vp, _vp]),
    "bbx_reset": (C.c_int, [_vp, _vp, _vp]),
    "bbx_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "bbx_step_autoreset": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "bbx_step_obs": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "bbx_rollout": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "bbx_obs": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "bbx_cols": (C.c_int, [_vp]),
    "bbx_nvars": (C.c_int, [_vp]),
    "bbx_batch_size": (C.c_int, [_vp]),
    "bbx_value": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_double, C.POINTER(C.c_double)]),
    "bbx_values": (C.c_int, [_vp, C.c_char_p, C.c_double, _vp]),
    "bbx_step_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "bbx_step_device_autoreset": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "bbx_pmlp_prepared_floats": (C.c_int, [C.c_int, C.c_int]),
    "bbx_pmlp_prepare": (C.c_int, [_vp, _vp, _vp, C.c_float, C.c_int, C.c_int, _vp, _vp]),
    "bbx_pmlp_act": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "bbx_pmlp2_prepared_floats": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "bbx_pmlp2_prepare": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "bbx_pmlp2_act": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "bbx_pmlp3_prepared_floats": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "bbx_pmlp3_prepare": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "bbx_pmlp3_act": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "bbx_policy_step_device": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "bbx_policy_rollout_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_longlong, _vp]),
    "bbx_rollout_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "bbx_prefetch": (C.c_int, [_vp]),
    "bbx_accounting": (C.c_int, [_vp, C.c_int]),
    "bbx_timing": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), _i32p]),
    "bbx_sync": (C.c_int, [_vp]),
    "bbx_stats": (C.c_int, [_vp, _vp]),
    "bbx_env_status": (C.c_int, [_vp, _vp]),
    "bbx_capacities": (C.c_int, [_vp, _vp]),
    "bbx_alg_create": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "bbx_alg_destroy": (None, [_vp]),
    "bbx_alg_from_envs": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(_vp)]),
    "bbx_alg_binop": (C.c_int, [_vp, C.c_int, _vp]),
    "bbx_alg_reduce": (C.c_int, [_vp, _vp, _vp]),
    "bbx_alg_update": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    "bbx_alg_minimalize": (C.c_int, [_vp]),
    "bbx_alg_interreduce": (C.c_int, [_vp]),
    "bbx_alg_sizes": (C.c_int, [_vp, _vp, _vp]),
    "bbx_alg_get": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "bbx_values_seeded": (C.c_int, [_vp, C.c_char_p, C.c_double, _vp, _vp]),
    "bbx_persistent": (C.c_int, [_vp, C.c_int]),
    "bbx_join": (C.c_int, [_vp, _vp]),
    "bbx_graph_replayed": (C.c_int, [_vp, _vp]),
    "bbx_session_stats": (C.c_int, [_vp, _vp]),
    "bbx_kernels_launched": (C.c_int, [_vp, _vp]),
    "bbx_state_sizes": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _i32p]),
    "bbx_state_get": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "bbx_reduced_basis": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _vp, _vp, _vp]),
    "bbx_trace_enable": (C.c_int, [_vp, C.c_int]),
    "bbx_trace_read": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "bbx_gen_create": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "bbx_gen_destroy": (None, [_vp]),
    "bbx_gen_seed": (C.c_int, [_vp, C.c_int64]),
    "bbx_gen_nvars": (C.c_int, [_vp]),
    "bbx_gen_next": (C.c_int, [_vp, _i32p, _i32p]),
    "bbx_gen_get": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "bbx_parse_ideal": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "bbx_format_ideal": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_char_p, C.c_int]),
    "bbx_agent_hash": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "bbx_agent_action": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "bbx_last_error": (C.c_char_p, []),
    "bbx_version": (C.c_char_p, []),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libbbx.so has not been built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        dll = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(dll, name)
            fn.restype = res
            fn.argtypes = args
        _lib = dll
    return _lib


def check(rc):
    if rc != 0:
        raise BbxError(rc, lib().bbx_last_error().decode("utf-8", "replace"))


def ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)
