"""ctypes loader for libgprhip.so (the HIP/CDNA4 kernels + C-ABI, include/gpr_hip.h).

There is no CPU fallback anywhere in this package: if the library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C gpr_calculator_amd/csrc`)
every compute entry point raises.
"""
import ctypes as C
import os

# torch first: its bundled libamdhip64.so.7 must be the HIP runtime the whole process shares
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPR_HIP_LIB") or os.path.join(_HERE, "libgprhip.so")   # GPR_HIP_LIB: A/B builds when tuning
_lib = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class KernelParams(C.Structure):
    """struct gpr_kernel_params (include/gpr_hip.h)."""
    _fields_ = [("kind", C.c_int), ("zeta", C.c_double), ("sigma", C.c_double), ("l", C.c_double),
                ("sigma0", C.c_double), ("tol", C.c_double)]


KERNEL_RBF, KERNEL_DOT = 0, 1
ROLE_STREAM, ROLE_RESIDENT = 0, 1
SET_NUMPY_EPS = 1

_vp, _i, _d, _ll = C.c_void_p, C.c_int, C.c_double, C.c_longlong
_L1_SIGNATURES = {
    "gpr_last_error": ([], C.c_char_p),
    "gpr_build_info": ([], C.c_char_p),
    "gpr_device_count": ([], _i),
    "gpr_l0_last_status": ([], _i),
    "gpr_memcpy_dtod": ([_vp, _vp, _vp, _ll], _i),
    "gpr_set_create": ([C.POINTER(_vp), _vp, _i, _i, _i, _i, _i, _i, c_int_p, _vp, _vp, _vp, _i], _i),
    "gpr_set_destroy": ([_vp, _vp], _i),
    "gpr_set_info": ([_vp, c_int_p, c_int_p, c_int_p, C.POINTER(_ll)], _i),
    "gpr_block_ee": ([_vp, C.POINTER(KernelParams), _vp, _vp, _i, _vp, _ll, _vp, _ll], _i),
    "gpr_block_ef": ([_vp, C.POINTER(KernelParams), _vp, _vp, _i, _vp, _ll, _vp, _ll], _i),
    "gpr_block_ff": ([_vp, C.POINTER(KernelParams), _vp, _vp, _i, _vp, _ll, _vp, _ll], _i),
    "gpr_diag_ff": ([_vp, C.POINTER(KernelParams), _vp, _vp, _vp], _i),
    "gpr_diag_ee": ([_vp, C.POINTER(KernelParams), _vp, _vp, _vp], _i),
    "gpr_potrf_lower": ([_vp, _i, _vp, _ll, _vp], _i),
    "gpr_trsm_lower": ([_vp, _i, _i, _i, _vp, _ll, _vp, _ll], _i),
    "gpr_potrf_lower_dinv": ([_vp, _i, _vp, _ll, _vp, _vp], _i),
    "gpr_trsm_lower_dinv": ([_vp, _i, _i, _i, _vp, _ll, _vp, _vp, _ll], _i),
    "gpr_gemm": ([_vp, _i, _i, _i, _i, _i, _d, _vp, _ll, _vp, _ll, _d, _vp, _ll], _i),
    "gpr_xtx_lower": ([_vp, _i, _vp, _ll, _vp, _ll], _i),
    "gpr_dot2_session_create": ([C.POINTER(_vp), _vp, _i, _i, _d, _d, _i, _i, _vp, _i, _i, c_int_p, _d, _d, _vp, _vp, _vp, _i], _i),
    "gpr_dot2_session_predict": ([_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "gpr_dot2_session_set_hint": ([_vp, _d, _d], _i),
    "gpr_dot2_session_submit": ([_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i)], _i),
    "gpr_dot2_session_wait": ([_vp, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "gpr_dot2_session_set_depth": ([_vp, _i], _i),
    "gpr_dot2_session_destroy": ([_vp], _i),
    "gpr_dot2_session_slot_stream": ([_vp, _i], _vp),
    "gpr_dot2_session_reserve_streams": ([], _i),
    "gpr_dev_cache_count": ([_vp, _i], _i),
    "gpr_trsv_chain_fallbacks": ([], _i),
    "gpr_trsv_debug_max_polls": ([_i], _i),
    "gpr_potrf_set_polling": ([_i], _i),
    "gpr_potrf_polling": ([], _i),
    "gpr_potrf_debug_max_polls": ([_i], _i),
    "gpr_potrf_ops_doubles": ([], _i),
    "gpr_potrf_diag_block": ([_vp, _i, _vp, _ll, _i, _vp, _vp], _i),
    "gpr_potrf_panel_rows": ([_vp, _i, _vp, _vp, _ll, _i], _i),
    "gpr_gemm_nt_cyclic_lower": ([_vp, _i, _i, _i, _d, _vp, _ll, _vp, _ll, _d, _vp, _ll, _i, _i, _i, _i, _i], _i),
    "gpr_potrf_finish": ([_vp, _i, _vp, _ll, _vp], _i),
    "gpr_trtri_lower": ([_vp, _i, _vp, _ll, _vp, _ll], _i),
    "gpr_mirror_lower": ([_vp, _i, _vp, _ll], _i),
    "gpr_trmm_lower": ([_vp, _i, _i, _vp, _ll, _vp, _ll, _vp, _ll], _i),
    "gpr_trmm_lower_sumsq": ([_vp, _i, _i, _vp, _ll, _vp, _ll, _vp], _i),
    "gpr_add_diag": ([_vp, _i, _vp, _ll, _i, _d, _d], _i),
    "gpr_colsumsq": ([_vp, _i, _i, _vp, _ll, _vp], _i),
    "gpr_sum_log_diag": ([_vp, _i, _vp, _ll, _vp], _i),
    "gpr_dot2_features_direct": ([_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, c_int_p, _i, c_int_p, _d, _d, _vp, C.c_longlong, _vp], _i),
    "gpr_set_create_gather": ([C.POINTER(_vp), _vp, _i, _i, _i, _i, _i, _i, c_int_p, _vp, _vp, _vp, _vp, _vp, _i], _i),
    "gpr_nbr_build_device": ([C.POINTER(_vp), _vp, _i, c_int_p, c_double_p, c_double_p, c_int_p, c_int_p, _d, _i], _i),
    "gpr_nbr_device_arrays": ([_vp] + [C.POINTER(_vp)] * 9, _i),
    "gpr_nbr_force_rows": ([_vp, _i, _vp, _vp, _i, _vp, _vp], _i),
    "gpr_nbr_check": ([_vp, c_int_p], _i),
    "gpr_nbr_point_rows": ([_vp, _i, _vp, _vp], _i),
    "gpr_nbr_status_to": ([_vp, _vp], _i),
    "gpr_nbr_info_to": ([_vp, _vp], _i),
    "gpr_nbr_build_device_extra": ([C.POINTER(_vp), _vp, _i, c_int_p, c_double_p, c_double_p, c_int_p, c_int_p, _d, _i, _i, _i, _i, c_int_p], _i),
    "gpr_nbr_extra_ptr": ([_vp], _vp),
    "gpr_nbr_build_device_capped": ([C.POINTER(_vp), _vp, _i, c_int_p, c_double_p, c_double_p, c_int_p, c_int_p, _d, _i, _i, _i], _i),
    "gpr_nbr_build": ([C.POINTER(_vp), _i, c_int_p, c_double_p, c_double_p, c_int_p, c_int_p, _d, _i], _i),
    "gpr_nbr_sizes": ([_vp, c_int_p, c_int_p], _i),
    "gpr_nbr_export": ([_vp, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p], _i),
    "gpr_nbr_destroy": ([_vp], _i),
    "gpr_nbr_build_device_rows": ([C.POINTER(_vp), _vp, _i, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _vp, _vp, C.POINTER(_i)], _i),
    "gpr_nbr_set_fused": ([_i], _i),
    "gpr_nbr_fused_state": ([c_int_p, c_int_p], _i),
    "gpr_nbr_debug_fused_capacity": ([_i], _i),
    "gpr_dot2_features": ([_vp, _vp, _i, c_int_p, _d, _vp, _ll], _i),
    "gpr_set_shard": ([_i, _i], _i),
    "gpr_profile_enable": ([_i], _i),
    "gpr_profile_fetch": ([c_double_p, c_int_p], _i),
    "gpr_profile_counts": ([c_int_p], _i),
    "gpr_so3_power_spectrum": ([_vp, _i, _i, _d, _d, _i, _i, _vp, _i, _i, _i] + [_vp] * 8, _i),
    "gpr_so3_power_spectrum_stress": ([_vp, _i, _i, _d, _d, _i, _i, _vp, _i, _i, _i] + [_vp] * 11, _i),
    "gpr_predict_dot2_batch": ([_vp, _i, _i, _d, _d, _i, _i, _vp, _i, c_int_p, c_double_p, c_double_p, c_int_p, c_int_p, _i,
                               C.POINTER(C.c_ubyte), _i, c_int_p, _d, _d, _vp, _vp, _vp, _i, _i, _i, _vp, _ll], _i),
    "gpr_so3_set_variant": ([_i], _i),
    "gpr_so3_set_prepass_min": ([_i], _i),
    "gpr_so3_table_size": ([_i, _i, _i, _i], _ll),
    "gpr_so3_tables": ([_vp, _i, _i, _d, _d, _i, _vp, _vp, _i, _i, _vp], _i),
    "gpr_so3_radial": ([_vp, _i, _i, _i, _d, _d, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp], _i),
    "gpr_gemv_t": ([_vp, _i, _i, _vp, _ll, _vp, _vp], _i),
    "gpr_ws_finish": ([_vp, _i, _i, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _d], _i),
    "gpr_gemm_nt_lower": ([_vp, _i, _i, _d, _vp, _ll, _vp, _ll, _vp, _ll], _i),
    "gpr_dot2_features_ef": ([_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, c_int_p, _i, c_int_p, _d, _d, _vp, _ll, _vp], _i),
    "gpr_ws_tail": ([_vp, _i, _i, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _d], _i),
    "gpr_ws_tail_status": ([_vp, _i, _i, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _d, _vp, _vp], _i),
    "gpr_nbr_info_ptr": ([_vp], _vp),
    "gpr_rowdot": ([_vp, _i, _i, _vp, _ll, _vp, _ll, _vp], _i),
    "gpr_frob_dot": ([_vp, _i, _i, _vp, _ll, _vp, _ll, _vp], _i),
}


class GprHipError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if it was never built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GprHipError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (args, res) in _L1_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes, fn.restype = args, res
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().gpr_last_error()
        raise GprHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def require_gpu():
    if not torch.cuda.is_available():
        raise GprHipError("no HIP device visible: gpr_calculator_amd computes on the GPU only")
