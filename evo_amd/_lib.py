"""ctypes binding of libevo_amd.so (include/evo_amd.h).

This is the thin C-ABI layer BASELINE.json:north_star asks for: Python host code, hand-written
HIP kernels, no PyTorch.  The library must be present and a gfx950 GPU must be visible for any
compute call; there is deliberately NO CPU fallback (the CPU restatement under oracle/ is test
infrastructure and is never imported from here).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libevo_amd.so")

MODEL_BSC, MODEL_SSSC = 0, 1

# kernel-class ids of evoamd_kernel_time_ms (evo_amd.hip: KID_*)
KERNEL_IDS = {
    "lpj_resident": 0, "lpj_candidates": 1, "lpj_overflow": 2, "row_lse": 3, "vary_kn": 4,
    "stats": 5, "stats_overflow": 6, "gemm_f64": 7, "evolve": 8, "misc": 9, "mstep_device": 10,
    "lpj_pass": 11, "stats_pass": 12, "lpj_k3_4": 13, "lpj_k5_8": 14, "lpj_k9plus": 15,
    "stats_k3_4": 16, "stats_k5_8": 17, "stats_k9plus": 18, "allreduce": 19, "estep_fused": 20,
}

_c_dp = ctypes.POINTER(ctypes.c_double)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p
_I, _I64, _U64, _DBL = ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double

# name -> (restype, argtypes); one entry per prototype in include/evo_amd.h
SIGNATURES = {
    "evoamd_abi_version": (_I, []),
    "evoamd_last_error": (ctypes.c_char_p, []),
    "evoamd_device_count": (_I, [ctypes.POINTER(_I)]),
    "evoamd_ctx_create": (_I, [_I, ctypes.POINTER(_vp)]),
    "evoamd_ctx_destroy": (None, [_vp]),
    "evoamd_synchronize": (_I, [_vp]),
    "evoamd_set_option": (_I, [_vp, ctypes.c_char_p, _I]),
    "evoamd_configure": (_I, [_vp, _I, _I64, _I, _I, _I, _I, _I]),
    "evoamd_upload_data": (_I, [_vp, _c_dp]),
    "evoamd_upload_states": (_I, [_vp, _c_u8p]),
    "evoamd_download_states": (_I, [_vp, _c_u8p]),
    "evoamd_upload_states_packed": (_I, [_vp, _c_u8p, _I64, _I64]),
    "evoamd_download_states_packed": (_I, [_vp, _c_u8p, _I64, _I64]),
    "evoamd_upload_lpj": (_I, [_vp, _c_dp]),
    "evoamd_download_lpj": (_I, [_vp, _c_dp]),
    "evoamd_set_params_bsc": (_I, [_vp, _c_dp, _DBL, _DBL, _c_dp]),
    "evoamd_set_params_sssc": (_I, [_vp, _c_dp, _c_dp, _c_dp, _c_dp, _DBL, _c_dp]),
    "evoamd_lpj_resident": (_I, [_vp]),
    "evoamd_lpj_candidates": (_I, [_vp, _c_u8p, _c_i32p, _I, _c_dp]),
    "evoamd_set_candidates": (_I, [_vp, _c_u8p, _c_i32p, _I, _c_dp]),
    "evoamd_lpj_shared": (_I, [_vp, _c_u8p, _I, _c_dp]),
    "evoamd_lpj_single": (_I, [_vp, _c_dp, _c_u8p, _I, _c_dp, _c_i32p]),
    "evoamd_vary_kn": (_I, [_vp, _I, _c_dp]),
    "evoamd_evolve_randflip": (_I, [_vp, _I, _I, _U64, _I]),
    "evoamd_estep": (_I, [_vp, _I, _I, _U64, _I, _I, ctypes.POINTER(_I)]),
    "evoamd_estep_counters": (_I, [_vp, ctypes.POINTER(_I64)]),
    "evoamd_evolve_states": (_I, [_vp, _I, _I, _I, _I, _I, _U64, _DBL, _DBL]),
    "evoamd_download_candidates": (_I, [_vp, _c_u8p, _c_i32p, _c_dp]),
    "evoamd_acc_size": (_I64, [_vp]),
    "evoamd_stats": (_I, [_vp, _c_dp]),
    "evoamd_mstep_device": (_I, [_vp, _I, _c_dp, _c_dp]),
    "evoamd_reconstruct": (_I, [_vp, _c_dp]),
    "evoamd_upload_masks": (_I, [_vp, _c_u8p, _c_u8p]),
    "evoamd_upload_yrec": (_I, [_vp, _c_dp]),
    "evoamd_set_reliable_fraction": (_I, [_vp, ctypes.c_double]),
    "evoamd_lpj_single_masked": (_I, [_vp, _c_dp, _c_u8p, _c_u8p, _I, _c_dp, _c_i32p]),
    "evoamd_inverse": (_I, [_vp, _c_dp, _c_dp, _I, _c_dp]),
    "evoamd_gemm_tn": (_I, [_vp, _c_dp, _c_dp, _c_dp, _I64, _I, _I, _I]),
    "evoamd_get_params_bsc": (_I, [_vp, _c_dp, _c_dp, _c_dp]),
    "evoamd_get_params_sssc": (_I, [_vp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]),
    "evoamd_restore_theta_backup": (_I, [_vp]),
    "evoamd_free_energy": (_I, [_vp, _c_dp, _I64, _I, _c_dp]),
    "evoamd_set_estep_counts": (_I, [_vp, _DBL, _DBL]),
    "evoamd_comm_unique_id": (_I, [_c_u8p]),
    "evoamd_comm_init": (_I, [_vp, _c_u8p, _I, _I]),
    "evoamd_comm_allreduce_host": (_I, [_vp, _c_dp, _I64, _I]),
    "evoamd_comm_destroy": (_I, [_vp]),
    "evoamd_timing_enable": (_I, [_vp, _I]),
    "evoamd_timing_reset": (_I, [_vp]),
    "evoamd_kernel_time_ms": (_I, [_vp, _I, _c_dp, ctypes.POINTER(_I64)]),
    "evoamd_kernel_name": (ctypes.c_char_p, [_I]),
}


class EvoAmdError(RuntimeError):
    pass


_lib = None


def load():
    """Load libevo_amd.so (once).  Raises EvoAmdError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvoAmdError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C evo_amd/csrc`). evo_amd has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise EvoAmdError("libevo_amd error %d: %s" % (rc, load().evoamd_last_error().decode()))


def dptr(a):
    return a.ctypes.data_as(_c_dp)


def u8ptr(a):
    return a.ctypes.data_as(_c_u8p)


def i32ptr(a):
    return a.ctypes.data_as(_c_i32p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_bool_bytes(a):
    """C-contiguous bool array viewed as uint8 (the reference's own K^n layout, 1 byte per latent)."""
    a = np.ascontiguousarray(a, dtype=np.bool_)
    return a.view(np.uint8)
