"""ctypes loader of libpr_amd.so (the C ABI of include/place_recognition.h).

There is no fallback: if the shared library is missing this raises, and every entry point of the library
itself fails with PR_EHIP when no gfx950 device is usable.
"""
from __future__ import annotations

import ctypes as C
import os

PR_OK, PR_EINVAL, PR_ENOMEM, PR_EHIP, PR_EIO, PR_ENAN = 0, -1, -2, -3, -4, -5
TYPE_SC, TYPE_M2DP, TYPE_DELIGHT, TYPE_GIST, TYPE_BOW = 0, 1, 2, 3, 4
SC_ARITH_F16X2, SC_ARITH_F32, SC_ARITH_F16 = 0, 1, 2
NAN_EXCLUDE, NAN_FAIL = 0, 1
WARN_NAN_ROWS, WARN_M2DP_SVD, WARN_F16_FALLBACK, WARN_ORDER_RESOLVED, WARN_ORDER_UNRESOLVED = 1, 2, 4, 8, 16
ROLE_QUERY, ROLE_DB = 0, 1
F64, F32 = 0, 1
HOST, DEVICE = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PR_AMD_LIB") or os.path.join(_HERE, "libpr_amd.so")   # PR_AMD_LIB: experiment builds only

# every symbol include/place_recognition.h declares: (name, restype, argtypes)
_vp, _i32, _dbl = C.c_void_p, C.c_int32, C.c_double
SYMBOLS = {
    "pr_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pr_create_on_stream": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "pr_set_nan_policy": (C.c_int, [_vp, C.c_int]),
    "pr_get_nan_policy": (C.c_int, [_vp]),
    "pr_set_exact_statistics": (C.c_int, [_vp, C.c_int]),
    "pr_take_warnings": (C.c_int, [_vp]),
    "pr_set_sc_binary": (C.c_int, [_vp, C.c_int]),
    "pr_sc_binary_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pr_set_kernel_timing": (C.c_int, [_vp, C.c_int]),
    "pr_last_distance_timing": (C.c_int, [_vp, _vp]),
    "pr_match_topk_f64": (C.c_int, [_vp, C.c_int, _vp, _i32, _vp, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "pr_match_topk_fused_f64": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "pr_rerank_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _i32, _vp,
                                _vp, _i32, _vp, _vp]),
    "pr_rerank_partial_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, _i32,
                                        _vp, _vp, _i32, _vp]),
    "pr_rerank_width": (C.c_int, [_vp, _i32]),
    "pr_f16_margin_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _dbl, _i32, _vp, _i32, _vp, _vp, _vp]),
    "pr_rerank_finish_dev": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _dbl, _vp, _vp]),
    "pr_order_resolve_async_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _i32, _i32, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "pr_order_resolve_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _i32, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp]),
    "pr_order_flagged_count": (C.c_int, [_vp, _i32, _vp]),
    "pr_order_exact_moments_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pr_order_exact_select_dev": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _dbl, C.c_int, C.c_int, _i32, _i32, _vp]),
    "pr_order_exact_merge_dev": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pr_widen_scores_dev": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "pr_merge_topk_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "pr_group_create": (C.c_int, [_vp, _i32, C.POINTER(_vp)]),
    "pr_group_destroy": (None, [_vp]),
    "pr_group_last_error": (C.c_char_p, [_vp]),
    "pr_group_size": (_i32, [_vp]),
    "pr_group_uses_rccl": (C.c_int, [_vp]),
    "pr_group_rccl_ranks": (_i32, [_vp]),
    "pr_group_last_flagged": (_i32, [_vp]),
    "pr_group_set_exact_statistics": (C.c_int, [_vp, C.c_int]),
    "pr_group_set_timing": (C.c_int, [_vp, C.c_int]),
    "pr_group_last_timing": (C.c_int, [_vp, _vp, _i32]),
    "pr_group_set_database": (C.c_int, [_vp, C.c_int, _vp, _i32]),
    "pr_group_set_database_growable": (C.c_int, [_vp, C.c_int, _vp, _i32, _i32]),
    "pr_group_append_database": (C.c_int, [_vp, _vp, _i32]),
    "pr_group_database_rows": (_i32, [_vp]),
    "pr_group_take_warnings": (C.c_int, [_vp]),
    "pr_group_match_topk": (C.c_int, [_vp, _vp, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "pr_destroy": (None, [_vp]),
    "pr_last_error": (C.c_char_p, [_vp]),
    "pr_version": (C.c_char_p, []),
    "pr_set_sc_arith": (C.c_int, [_vp, C.c_int]),
    "pr_get_sc_arith": (C.c_int, [_vp]),
    "pr_sync": (C.c_int, [_vp]),
    "pr_stream": (_vp, [_vp]),
    "pr_sc_generate": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _dbl, _vp]),
    "pr_m2dp_generate": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _dbl, _vp]),
    "pr_m2dp_svd_rows": (C.c_int, [_vp, _vp, _i32, _vp]),
    "pr_delight_generate": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "pr_delight_distance": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp]),
    "pr_gist_distance": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "pr_bow_distance": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "pr_match_topk_fused": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "pr_fuse_select2_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "pr_fuse_select2_f64_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp]),
    "pr_fuse_select_f64_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp]),
    "pr_match_topk_cols": (C.c_int, [_vp, C.c_int, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pr_delight_generate_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "pr_sc_distance": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    "pr_m2dp_distance": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    "pr_match_topk": (C.c_int, [_vp, C.c_int, _vp, _i32, _vp, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "pr_sigset_create": (C.c_int, [_vp, C.c_int, C.c_int, _i32, C.POINTER(_vp)]),
    "pr_sigset_destroy": (None, [_vp, _vp]),
    "pr_sigset_pack": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _i32]),
    "pr_sigset_count": (_i32, [_vp]),
    "pr_sigset_reserve": (C.c_int, [_vp, _vp]),
    "pr_sigset_append": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _i32]),
    "pr_sigset_image": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t), C.POINTER(_i32)]),
    "pr_distances_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "pr_row_moments_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "pr_fuse_select_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _dbl, _i32, _vp, _vp]),
    "pr_sc_generate_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _dbl, _vp]),
    "pr_m2dp_generate_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _dbl, _vp]),
    "pr_cloud_frames_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "pr_sc_generate_frames_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _dbl, _vp, C.c_int, _vp]),
    "pr_m2dp_generate_frames_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _dbl, _vp, C.c_int, _vp]),
    "pr_delight_generate_frames_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "pr_generate_clouds": (C.c_int, [_vp, C.c_int, _vp, _dbl, _vp]),
    "pr_clouds_dev_xyz": (_vp, [_vp]),
    "pr_clouds_dev_inten": (_vp, [_vp]),
    "pr_clouds_dev_offs": (_vp, [_vp]),
    "pr_clouds_dev_frames": (_vp, [_vp]),
    "pr_pts_preprocess": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, _dbl, C.c_int, C.c_int, C.POINTER(_vp)]),
    "pr_pts_preprocess_gpu": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_char_p, _dbl, C.c_int, C.c_int, C.POINTER(_vp)]),
    "pr_hash_order": (C.c_int, [_vp, _i32, _vp]),
    "pr_clouds_avg_ms": (C.c_double, [_vp]),
    "pr_clouds_avg_pts": (C.c_double, [_vp]),
    "pr_clouds_count": (C.c_int64, [_vp]),
    "pr_clouds_offs": (C.POINTER(C.c_int64), [_vp]),
    "pr_clouds_xyz": (C.POINTER(C.c_double), [_vp]),
    "pr_clouds_inten": (C.POINTER(C.c_float), [_vp]),
    "pr_clouds_ids": (C.POINTER(C.c_int32), [_vp]),
    "pr_clouds_free": (None, [_vp]),
    "pr_write_signatures": (C.c_int, [C.c_char_p, _vp, C.c_int64, C.c_int64]),
    "pr_read_signatures": (C.c_int, [C.c_char_p, C.POINTER(_vp), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pr_free": (None, [_vp]),
    "pr_write_signatures_bin": (C.c_int, [C.c_char_p, _vp, C.c_int64, C.c_int64, C.c_int]),
    "pr_read_signatures_bin": (C.c_int, [C.c_char_p, C.POINTER(_vp), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pr_write_poses": (C.c_int, [C.c_char_p, _vp, _vp, C.c_int64]),
    "pr_write_points": (C.c_int, [C.c_char_p, _vp, _vp, _vp, C.c_int64]),
    "pr_precision_recall": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _dbl, _i32, C.POINTER(_dbl), C.POINTER(_dbl), _vp,
                                      C.POINTER(_i32)]),
    "pr_host_last_error": (C.c_char_p, []),
}

_lib = None
HIP_RUNTIME = "system"      # which libamdhip64 this process ended up on (_one_hip_runtime)


class PRError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpr_amd error {code}: {msg}")
        self.code = code


def _one_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's, which
    libpr_amd.so is linked against): whichever copy is loaded first serves both, and torch does not find its GPUs on the other one
    ("No HIP GPUs are available" when libpr_amd.so came first).  So when torch is installed but not imported yet, its copy is loaded
    here - without importing torch - and both end up on it whatever the import order.  This is a process-wide choice made on behalf of a
    caller who may never import torch: PR_AMD_SYSTEM_HIP=1 keeps /opt/rocm's (the runtime the kernels were built against), HIP_RUNTIME
    records which one was taken, PR_AMD_VERBOSE=1 prints it."""
    import importlib.util
    import sys
    global HIP_RUNTIME
    if "torch" in sys.modules:
        HIP_RUNTIME = "torch's (torch was imported first)"
        return
    if os.environ.get("PR_AMD_SYSTEM_HIP"):
        HIP_RUNTIME = "system (PR_AMD_SYSTEM_HIP)"
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for d in (spec.submodule_search_locations or []) if spec else []:
        p = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
            HIP_RUNTIME = p
            if os.environ.get("PR_AMD_VERBOSE"):
                print(f"so_dso_place_recognition_amd: HIP runtime {p} (torch's copy, so that a later `import torch` finds its GPUs; "
                      "PR_AMD_SYSTEM_HIP=1 keeps /opt/rocm's)", file=sys.stderr)
            return


def load() -> C.CDLL:
    """Loads libpr_amd.so and binds every declared symbol (raises if the library or a symbol is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build it with `make -C so_dso_place_recognition_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback")
    _one_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx, rc: int):
    if rc != PR_OK:
        msg = load().pr_last_error(ctx)
        raise PRError(rc, msg.decode() if msg else "")
