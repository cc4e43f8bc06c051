"""ctypes binding of libtensoir_hip.so (the C ABI declared in include/tensoir_hip.h).

There is NO fallback: if the shared library is missing or cannot be loaded the import of any
product entry point raises.  Built in-tree by ``tensoir_amd/csrc/build.sh`` (``__graft_entry__.build``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TENSOIR_HIP_LIB", os.path.join(_HERE, "libtensoir_hip.so"))

c_float_p = C.c_void_p   # device pointers are passed as integers
c_int_p = C.c_void_p


class TirField(C.Structure):
    _fields_ = [
        ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("inv_aabb", C.c_float * 3),
        ("grid", C.c_int32 * 3),
        ("step_size", C.c_float), ("distance_scale", C.c_float), ("density_shift", C.c_float),
        ("weight_thres", C.c_float), ("near_", C.c_float), ("far_", C.c_float),
        ("act", C.c_int32), ("n_dcomp", C.c_int32), ("n_acomp", C.c_int32), ("app_dim", C.c_int32),
        ("n_lights", C.c_int32),
        ("dplane", C.c_void_p * 3), ("dline", C.c_void_p * 3),
        ("aplane", C.c_void_p * 3), ("aline", C.c_void_p * 3),
        ("basis_t", C.c_void_p), ("light_line", C.c_void_p), ("light_mean", C.c_void_p),
        ("occ_nbr", C.c_void_p),
        ("occ_dim", C.c_int32 * 3), ("occ_aabb_min", C.c_float * 3), ("occ_inv", C.c_float * 3),
        ("occ_lo", C.c_float * 3), ("occ_hi", C.c_float * 3),
        ("tune_lds_lines", C.c_int32), ("tune_xcd_order", C.c_int32),
    ]


class TirFieldHalf(C.Structure):
    _fields_ = [("aplane", C.c_void_p * 3), ("aline", C.c_void_p * 3)]


class TirFieldGrad(C.Structure):
    _fields_ = [("dplane", C.c_void_p * 3), ("dline", C.c_void_p * 3), ("aplane", C.c_void_p * 3),
                ("aline", C.c_void_p * 3), ("light_line", C.c_void_p), ("light_mean", C.c_void_p)]


class TirMlp(C.Structure):
    _fields_ = [("packed", C.c_void_p), ("feat_dim", C.c_int32), ("pe", C.c_int32),
                ("hidden", C.c_int32), ("out_dim", C.c_int32), ("act", C.c_int32), ("tune_grid", C.c_int32)]


class TirEnvSG(C.Structure):
    _fields_ = [("sgs", C.c_void_p), ("rot", C.c_void_p), ("n_sg", C.c_int32), ("n_lights", C.c_int32)]


P = C.c_void_p
I32 = C.c_int32
I64 = C.c_int64
F32 = C.c_float

# name -> (restype, argtypes); mirrors include/tensoir_hip.h one to one
SIGNATURES = {
    "tir_version": (C.c_int, []),
    "tir_error_string": (C.c_char_p, [C.c_int]),
    "tir_device_check": (C.c_int, []),
    "tir_pack_plane": (C.c_int, [P, P, I32, I32, I32, P]),
    "tir_pack_occupancy": (C.c_int, [P, P, I32, I32, I32, P]),
    "tir_pack_basis": (C.c_int, [P, P, I32, I32, P]),
    "tir_light_mean": (C.c_int, [P, P, I32, I32, P]),
    "tir_mlp_packed_floats": (I64, [I32, I32, I32, I32]),
    "tir_pack_mlp": (C.c_int, [P, P, P, P, P, P, I32, I32, I32, I32, P, P]),
    "tir_vm_density_fwd": (C.c_int, [C.POINTER(TirField), P, P, P, I64, P]),
    "tir_occupancy_query": (C.c_int, [C.POINTER(TirField), P, P, I64, P]),
    "tir_dense_alpha": (C.c_int, [C.POINTER(TirField), P, P, P, I32, I32, I32, F32, P, P]),
    "tir_alpha_pool": (C.c_int, [P, I32, I32, I32, F32, P, P, P]),
    "tir_filter_rays": (C.c_int, [C.POINTER(TirField), P, I64, I32, I32, P, P]),
    "tir_density_grad_fwd": (C.c_int, [C.POINTER(TirField), P, P, P, P, I64, P, P]),
    "tir_vm_app_fwd": (C.c_int, [C.POINTER(TirField), P, P, P, P, P, I32, I32, I64, P, P]),
    "tir_vm_app_fwd_bf16x3": (C.c_int, [C.POINTER(TirField), P, P, P, P, P, I32, I32, I64, P, P]),
    "tir_vm_app_fwd_x3": (C.c_int, [C.POINTER(TirField), P, P, P, P, P, I32, I32, I64, P, P]),
    "tir_pack_half": (C.c_int, [P, P, P, I32, P]),
    "tir_pack_half_checked": (C.c_int, [P, P, P, I32, P, P]),
    "tir_vm_app_fwd_h16": (C.c_int, [C.POINTER(TirField), C.POINTER(TirFieldHalf), P, P, P, P, I32, I32, I64, P, P]),
    "tir_indirect_fused_fwd": (C.c_int, [C.POINTER(TirField), C.POINTER(TirFieldHalf), C.POINTER(TirMlp), P, P, P, I32, I32, P, P, I64, P, P]),
    "tir_indirect_fused_hp_fwd": (C.c_int, [C.POINTER(TirField), C.POINTER(TirMlp), P, P, P, I32, I32, P, P, I64, P, P]),
    "tir_vm_app_fwd_valu": (C.c_int, [C.POINTER(TirField), P, P, P, P, P, I32, I32, I64, P, P]),
    "tir_mlp_fwd": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, I64, P, P]),
    "tir_mlp_fwd_bf16x3": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, I64, P, P]),
    "tir_mlp_fwd_multi_bf16x3": (C.c_int, [P, P, I32, P, P, P, I32, I64, P, P]),
    "tir_mlp_fwd_multi_auxtab_bf16x3": (C.c_int, [P, P, I32, P, P, P, P, I32, I64, P, P]),
    "tir_mlp_aux_table": (C.c_int, [C.POINTER(TirMlp), P, I64, P, P]),
    "tir_mlp_fwd_auxtab_bf16x3": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, I64, P, P]),
    "tir_mlp_fwd_auxtab_f16": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, I64, P, P]),
    "tir_mlp_train_fwd_auxtab_bf16x3": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, P, P, I64, P, P]),
    "tir_mlp_train_fwd_multi_bf16x3": (C.c_int, [P, P, I32, P, P, P, P, P, I32, I64, P, P]),
    "tir_mlp_fwd_bf16": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, I64, P, P]),
    "tir_mlp_fwd_valu": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, I64, P, P]),
    "tir_march_primary_fwd": (C.c_int, [C.POINTER(TirField), P, P, I32, I32, F32, P, P, P, P, P, P, P]),
    "tir_exclusive_scan": (C.c_int, [P, P, I32, P]),
    "tir_exclusive_scan_capped": (C.c_int, [P, P, I32, I32, P, P]),
    "tir_march_primary_fused_fwd": (C.c_int, [C.POINTER(TirField), P, P, I32, I32, F32, P, P, P, P, P, P,
                                              P, P, I32, P, P, I32, P, P]),
    "tir_composite_primary_fused": (C.c_int, [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, F32, P, P, P, P, I64, P]),
    "tir_vm_app_jitter_fwd": (C.c_int, [C.POINTER(TirField), P, I64, P, F32, C.c_uint64, C.c_uint64, P, P, P, I32, P]),
    "tir_vm_app_primary_fwd": (C.c_int, [C.POINTER(TirField), P, P, P, P, P, I32, I64, P, F32, C.c_uint64, C.c_uint64, P, P, P, P]),
    "tir_vm_app_primary_x3_fwd": (C.c_int, [C.POINTER(TirField), P, P, P, P, P, I32, I64, P, F32, C.c_uint64, C.c_uint64, P, P, P, P]),
    "tir_compact_primary": (C.c_int, [C.POINTER(TirField), P, P, P, P, I32, I32, P, P, P, P, P]),
    "tir_composite_primary": (C.c_int, [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, F32, P, P]),
    "tir_march_secondary_fwd": (C.c_int, [C.POINTER(TirField), P, P, P, P, P, I64, I32, I32, P, F32, P, P,
                                          P, I64, P, P, P, P, P, P, P]),
    "tir_accumulate_records": (C.c_int, [P, P, P, P, I64, P, P]),
    "tir_env_sg_fwd": (C.c_int, [C.POINTER(TirEnvSG), P, I32, P, P]),
    "tir_env_pixel_fwd": (C.c_int, [P, I32, I32, P, P, I32, I64, I32, P, P]),
    "tir_env_pixel_bwd": (C.c_int, [P, I32, I32, P, P, I32, I64, P, P, P]),
    "tir_shade_setup": (C.c_int, [P, P, P, I32, I32, F32, P, P, P]),
    "tir_shade_integrate": (C.c_int, [P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, F32, P, P]),
    "tir_shade_integrate_records": (C.c_int, [P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, F32, P, P, P]),
    "tir_shade_setup_compact": (C.c_int, [P, P, P, I32, I32, F32, P, P, P, P, P, P, I32, P]),
    "tir_march_secondary_ids_fwd": (C.c_int, [C.POINTER(TirField), P, P, P, P, P, I64, I32, I32, P, F32, P, P,
                                              P, I64, P, P, P, P, P, P, P, P, P]),
    "tir_relight_importance": (C.c_int, [P, P, P, P, P, P, P, P, P, I32, I32, P, P]),
    "tir_env_sample_setup": (C.c_int, [P, P, I32, I32, P, P, I32, I32, C.c_uint64, C.c_uint64, P, P, P]),
    "tir_env_sample_setup_list": (C.c_int, [P, P, I32, I32, P, I32, P, I32, I32, C.c_uint64, C.c_uint64, I32, I32, I32, P, P, I32, I32, P, P, P, P, P]),
    "tir_relight_importance_cells": (C.c_int, [P, P, P, P, P, P, P, P, P, P, I32, I32, P, P]),
    "tir_relight_importance_cells_packed": (C.c_int, [P, P, P, P, P, P, P, P, I32, I32, P, P]),
    "tir_env_lookup": (C.c_int, [P, I32, I32, P, I64, P, P]),
    "tir_surface_compact": (C.c_int, [P, P, I32, C.c_float, P, P, P, P, P, P, P, P, P]),
    "tir_env_sample_setup_list_n": (C.c_int, [P, P, I32, I32, P, I32, P, I32, I32, C.c_uint64, C.c_uint64, I32, I32, I32, P, P, I32, I32, P, P, P, P, P, P]),
    "tir_relight_importance_cells_packed_n": (C.c_int, [P, P, P, P, P, P, P, P, I32, I32, P, P, P]),
    "tir_env_compose": (C.c_int, [P, I32, I32, P, I32, I64, P, P, P, I32, P]),
    "tir_ggx_specular": (C.c_int, [P, P, P, P, P, I32, I32, P, P]),
    # ---- training (backward) entry points ----
    "tir_march_primary_train_fwd": (C.c_int, [C.POINTER(TirField), P, P, I32, I32, F32, P, P, P, P, P, P, P]),
    "tir_composite_primary_bwd": (C.c_int, [P] * 11 + [I32, I32, I32, I32, F32] + [P] * 9 + [P]),
    "tir_march_primary_bwd": (C.c_int, [C.POINTER(TirField), C.POINTER(TirFieldGrad), P, P, I32, I32, P, P, P, P, P, P, P]),
    "tir_density_grad_bwd": (C.c_int, [C.POINTER(TirField), C.POINTER(TirFieldGrad), P, P, I64, P]),
    "tir_vm_app_bwd": (C.c_int, [C.POINTER(TirField), C.POINTER(TirFieldGrad), P, P, P, P, P, I32, I64, P, P, P]),
    "tir_mlp_train_fwd": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, P, P, I64, P, P]),
    "tir_mlp_train_fwd_bf16x3": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, P, P, I64, P, P]),
    "tir_mlp_inputs": (C.c_int, [C.POINTER(TirMlp), P, I32, P, P, I32, P, I64, P]),
    "tir_mlp_bwd_packed_floats": (I64, [I32, I32, I32, I32]),
    "tir_pack_mlp_bwd": (C.c_int, [P, P, P, I32, I32, I32, I32, P, P]),
    "tir_mlp_bwd": (C.c_int, [C.POINTER(TirMlp), P, P, I32, P, P, P, P, I64, P, P, P, P, P]),
    "tir_mlp_bwd_bf16x3": (C.c_int, [C.POINTER(TirMlp), P, P, I32, P, P, P, P, I64, P, P, P, P, P]),
    "tir_mlp_bwd_multi_bf16x3": (C.c_int, [P, P, P, I32, P, P, P, P, I32, I64, P, P, P, P, P]),
    "tir_gemm_tn_small_bf16x3": (C.c_int, [P, I32, I32, P, I32, I32, I32, I64, P, I32, P]),
    "tir_mlp_wgrad_multi": (C.c_int, [P, P, P, P, P, P, I32, P, P, P, P, P, P, P, P, I32, I64, I32, P]),
    "tir_record_check": (C.c_int, [P, P, I32, P, P, P]),
    "tir_adam_step": (C.c_int, [I32, P, P, P, P, P, P, P, P, F32, F32, F32, P]),
    "tir_gemm_tn": (C.c_int, [P, I32, I32, P, I32, I32, I32, I64, P, I32, P, P]),
    "tir_gemm_tn_bf16x3": (C.c_int, [P, I32, I32, P, I32, I32, I32, I64, P, I32, P, P]),
    "tir_shade_integrate_bwd": (C.c_int, [P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, F32, P, P, P, P]),
    "tir_env_sg_bwd": (C.c_int, [C.POINTER(TirEnvSG), P, I32, P, P, P]),
}

_lib = None


class TensoirHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TensoirHipError(
            f"{LIB_PATH} not found: the HIP library is not built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or tensoir_amd/csrc/build.sh). "
            "tensoir_amd has no CPU / eager fallback by design.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().tir_error_string(int(rc))
        raise TensoirHipError(f"{what or 'libtensoir_hip'} failed: rc={rc} ({msg.decode() if msg else '?'})")
