"""ctypes binding of the CPU oracle (oracle/libpr_ref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libpr_ref.so")

_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    L = C.CDLL(_SO)
    L.pr_ref_pts_preprocess.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, C.c_int, C.POINTER(C.c_void_p)]
    L.pr_ref_clouds_count.restype = C.c_int64
    L.pr_ref_clouds_count.argtypes = [C.c_void_p]
    for nm, rt in (("offs", C.POINTER(C.c_int64)), ("xyz", C.POINTER(C.c_double)),
                   ("inten", C.POINTER(C.c_float)), ("ids", C.POINTER(C.c_int))):
        f = getattr(L, "pr_ref_clouds_" + nm)
        f.restype = rt
        f.argtypes = [C.c_void_p]
    L.pr_ref_clouds_free.argtypes = [C.c_void_p]
    L.pr_ref_align_pca.argtypes = [_dp, C.c_int64, _dp, _dp]
    L.pr_ref_ave_intensity.restype = C.c_float
    L.pr_ref_ave_intensity.argtypes = [_fp, C.c_int64]
    L.pr_ref_m2dp_plane_table.argtypes = [_dp, _dp]
    L.pr_ref_sc_generate.argtypes = [_dp, _fp, _lp, C.c_int32, C.c_double, _dp]
    L.pr_ref_m2dp_generate.argtypes = [_dp, _fp, _lp, C.c_int32, C.c_double, _dp]
    L.pr_ref_m2dp_matrices.argtypes = [_dp, _fp, C.c_int64, C.c_double, C.c_int, C.c_int, _dp, _dp]
    L.pr_ref_top_singular_pair.argtypes = [_dp, _dp]
    L.pr_ref_sc_distance.argtypes = [_dp, C.c_int32, _dp, C.c_int32, _dp, _dp]
    L.pr_ref_m2dp_distance.argtypes = [_dp, C.c_int32, _dp, C.c_int32, _dp, _dp]
    L.pr_ref_delight_generate.argtypes = [_dp, _fp, _lp, C.c_int32, _dp]
    L.pr_ref_delight_distance.argtypes = [_dp, C.c_int32, _dp, C.c_int32, _dp]
    L.pr_ref_match_topk_fused.argtypes = [_dp, _dp, C.c_int32, _dp, _dp, C.c_int32, C.c_int32, C.c_double, C.c_int32, _ip, _dp]
    L.pr_ref_gist_distance.argtypes = [_dp, C.c_int32, _dp, C.c_int32, C.c_int32, _dp]
    L.pr_ref_bow_distance.argtypes = [_dp, C.c_int32, _dp, C.c_int32, C.c_int32, _dp]
    L.pr_ref_unordered_order.argtypes = [_ip, C.c_int32, _ip]
    L.pr_ref_select_topk.argtypes = [_dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _ip, _dp]
    L.pr_ref_fuse_topk.argtypes = [_dp, _dp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, _ip, _dp]
    L.pr_ref_match_topk.argtypes = [C.c_int, _dp, C.c_int32, _dp, C.c_int32, C.c_int32, C.c_double, C.c_int32, _ip, _dp]
    L.pr_ref_precision_recall.argtypes = [_dp, _ip, C.c_int32, _dp, _dp, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), _ip, C.POINTER(C.c_int32), _ip, C.POINTER(C.c_int32), _dp, _dp]
    _lib = L
    return L


def pts_preprocess(poses_file, pts_file, incoming_id_file, lidar_range=45.0, polar=False):
    h = C.c_void_p()
    L = lib()
    rc = L.pr_ref_pts_preprocess(poses_file.encode(), pts_file.encode(),
                                 incoming_id_file.encode() if incoming_id_file else None,
                                 lidar_range, int(polar), C.byref(h))
    assert rc == 0
    N = L.pr_ref_clouds_count(h)
    offs = np.ctypeslib.as_array(L.pr_ref_clouds_offs(h), (N + 1,)).copy()
    T = int(offs[-1])
    xyz = np.ctypeslib.as_array(L.pr_ref_clouds_xyz(h), (T, 3)).copy() if T else np.zeros((0, 3))
    it = np.ctypeslib.as_array(L.pr_ref_clouds_inten(h), (T,)).copy() if T else np.zeros((0,), np.float32)
    ids = np.ctypeslib.as_array(L.pr_ref_clouds_ids(h), (N,)).copy() if N else np.zeros((0,), np.int32)
    L.pr_ref_clouds_free(h)
    return xyz, it, offs, ids


def align_pca(xyz):
    xyz = np.ascontiguousarray(xyz, np.float64)
    out = np.empty_like(xyz)
    ev = np.empty(9)
    lib().pr_ref_align_pca(xyz, xyz.shape[0], out, ev)
    return out, ev.reshape(3, 3).T  # columns v0|v1|v2


def ave_intensity(inten):
    inten = np.ascontiguousarray(inten, np.float32)
    return np.float32(lib().pr_ref_ave_intensity(inten, inten.shape[0]))


def sc_generate(xyz, inten, offs, max_rho=45.0):
    N = len(offs) - 1
    out = np.empty((N, 2400))
    rc = lib().pr_ref_sc_generate(np.ascontiguousarray(xyz, np.float64), np.ascontiguousarray(inten, np.float32),
                                  np.ascontiguousarray(offs, np.int64), N, max_rho, out)
    assert rc == 0
    return out


def m2dp_generate(xyz, inten, offs, max_rho=45.0):
    N = len(offs) - 1
    out = np.empty((4 * N, 384))
    rc = lib().pr_ref_m2dp_generate(np.ascontiguousarray(xyz, np.float64), np.ascontiguousarray(inten, np.float32),
                                    np.ascontiguousarray(offs, np.int64), N, max_rho, out)
    assert rc == 0
    return out


def m2dp_matrices(aligned, inten, max_rho, dx, dy):
    cm = np.empty((64, 128))
    im = np.empty((64, 128))
    lib().pr_ref_m2dp_matrices(np.ascontiguousarray(aligned, np.float64), np.ascontiguousarray(inten, np.float32),
                               aligned.shape[0], max_rho, dx, dy, cm, im)
    return cm, im


def top_singular_pair(A):
    out = np.empty(192)
    lib().pr_ref_top_singular_pair(np.ascontiguousarray(A, np.float64), out)
    return out


def plane_table():
    x = np.empty((64, 3))
    y = np.empty((64, 3))
    lib().pr_ref_m2dp_plane_table(x, y)
    return x, y


def sc_distance(h1, h2):
    h1 = np.ascontiguousarray(h1, np.float64)
    h2 = np.ascontiguousarray(h2, np.float64)
    m, n = h1.shape[0], h2.shape[0]
    dp = np.empty((m, n))
    di = np.empty((m, n))
    rc = lib().pr_ref_sc_distance(h1, m, h2, n, dp, di)
    return rc, dp, di


def m2dp_distance(h1, h2):
    h1 = np.ascontiguousarray(h1, np.float64)
    h2 = np.ascontiguousarray(h2, np.float64)
    m, n = h1.shape[0] // 4, h2.shape[0] // 4
    dp = np.empty((m, n))
    di = np.empty((m, n))
    rc = lib().pr_ref_m2dp_distance(h1, m, h2, n, dp, di)
    return rc, dp, di


def fuse_topk(dp, di, mask_width, p_weight=2.0, k=1):
    m, n = dp.shape
    idx = np.empty((m, k), np.int32)
    sc = np.empty((m, k))
    rc = lib().pr_ref_fuse_topk(np.ascontiguousarray(dp), np.ascontiguousarray(di), m, n, mask_width, p_weight, k, idx, sc)
    assert rc == 0
    return idx, sc


def match_topk(type_, h1, h2, mask_width, p_weight=2.0, k=1):
    h1 = np.ascontiguousarray(h1, np.float64)
    h2 = np.ascontiguousarray(h2, np.float64)
    div = {0: 1, 1: 4, 2: 16}[type_]
    m, n = h1.shape[0] // div, h2.shape[0] // div
    idx = np.empty((m, k), np.int32)
    sc = np.empty((m, k))
    rc = lib().pr_ref_match_topk(type_, h1, m, h2, n, mask_width, p_weight, k, idx, sc)
    return rc, idx, sc


def match_topk_fused(sc1, m2dp1, sc2, m2dp2, mask_width, p_weight=2.0, k=1):
    """BASELINE config 5 (build-defined): SC and M2DP z-scores of the same (query, entry) pairs added up."""
    sc1 = np.ascontiguousarray(sc1, np.float64); sc2 = np.ascontiguousarray(sc2, np.float64)
    a1 = np.ascontiguousarray(m2dp1, np.float64); a2 = np.ascontiguousarray(m2dp2, np.float64)
    m, n = sc1.shape[0], sc2.shape[0]
    idx = np.empty((m, k), np.int32); sc = np.empty((m, k))
    rc = lib().pr_ref_match_topk_fused(sc1, a1, m, sc2, a2, n, mask_width, p_weight, k, idx, sc)
    return rc, idx, sc


def select_topk(d, mask_width, k=1):
    """run_test.m:47-57 on one distance matrix (no fusion): (rc, idx int32 [m,k], score f64 [m,k])."""
    d = np.ascontiguousarray(d, np.float64)
    m, n = d.shape
    idx = np.empty((m, k), np.int32)
    sc = np.empty((m, k))
    rc = lib().pr_ref_select_topk(d, m, n, mask_width, k, idx, sc)
    return rc, idx, sc


def delight_generate(xyz, inten, offs):
    N = len(offs) - 1
    out = np.empty((16 * N, 256))
    rc = lib().pr_ref_delight_generate(np.ascontiguousarray(xyz, np.float64), np.ascontiguousarray(inten, np.float32),
                                       np.ascontiguousarray(offs, np.int64), N, out)
    assert rc == 0
    return out


def delight_distance(h1, h2):
    h1 = np.ascontiguousarray(h1, np.float64)
    h2 = np.ascontiguousarray(h2, np.float64)
    m, n = h1.shape[0] // 16, h2.shape[0] // 16
    d = np.empty((m, n))
    rc = lib().pr_ref_delight_distance(h1, m, h2, n, d)
    assert rc == 0
    return d


def gist_distance(h1, h2):
    h1 = np.ascontiguousarray(h1, np.float64); h2 = np.ascontiguousarray(h2, np.float64)
    d = np.empty((h1.shape[0], h2.shape[0]))
    assert lib().pr_ref_gist_distance(h1, h1.shape[0], h2, h2.shape[0], h1.shape[1], d) == 0
    return d


def bow_distance(h1, h2):
    h1 = np.ascontiguousarray(h1, np.float64); h2 = np.ascontiguousarray(h2, np.float64)
    m, n = h1.shape[0] // 2, h2.shape[0] // 2
    d = np.empty((m, n))
    assert lib().pr_ref_bow_distance(h1, m, h2, n, h1.shape[1], d) == 0
    return d


def unordered_order(keys):
    k = np.ascontiguousarray(keys, np.int32)
    out = np.empty(len(k), np.int32)
    assert lib().pr_ref_unordered_order(k, len(k), out) == 0
    return out


def precision_recall(diff_v, diff_idx, gt1, gt2, loop_diff, mask_width):
    """run_test.m:3-22, 58-85 -> dict(auc, top_recall, lp_gt [L,2], lp_detected [T,2], precision [m], recall [m]); 0-based indices."""
    diff_v = np.ascontiguousarray(diff_v, np.float64)
    diff_idx = np.ascontiguousarray(diff_idx, np.int32)
    gt1 = np.ascontiguousarray(gt1, np.float64)
    gt2 = np.ascontiguousarray(gt2, np.float64)
    m, n, cols = gt1.shape[0], gt2.shape[0], gt1.shape[1]
    assert diff_v.shape == (m,) and diff_idx.shape == (m,) and gt2.shape[1] == cols
    auc, tr, ng, nd = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
    lp_gt = np.zeros((max(m, 1), 2), np.int32)
    lp_det = np.zeros((max(m, 1), 2), np.int32)
    prec = np.zeros(max(m, 1)); rec = np.zeros(max(m, 1))
    rc = lib().pr_ref_precision_recall(diff_v, diff_idx, m, gt1, gt2, n, cols, float(loop_diff), int(mask_width), C.byref(auc), C.byref(tr),
                                       lp_gt, C.byref(ng), lp_det, C.byref(nd), prec, rec)
    assert rc == 0, rc
    return dict(auc=auc.value, top_recall=tr.value, lp_gt=lp_gt[:ng.value].astype(np.int64), lp_detected=lp_det[:nd.value].astype(np.int64),
                precision=prec[:m], recall=rec[:m])
