"""Evaluation harness of match_signatures/run_test.m: ground-truth loop pairs (:3-22) and the
precision/recall sweep, top recall at 100 % precision and AUC (:58-85).  Host numpy; O(m n) and O(m log m)."""
from __future__ import annotations

import numpy as np


def ground_truth_pairs(gt1, gt2, loop_diff: float, mask_width: int) -> np.ndarray:
    """run_test.m:3-22: for every i the closest j with |i-j| >= mask_width (first minimum), kept when closer than
    loop_diff.  Returns an [L, 2] int array of 0-based (i, j)."""
    gt1 = np.asarray(gt1, np.float64)
    gt2 = np.asarray(gt2, np.float64)
    out = []
    jdx = np.arange(gt2.shape[0])
    for i in range(gt1.shape[0]):
        d = ((gt1[i][None, :] - gt2) ** 2).sum(1)
        d = np.where(np.abs(i - jdx) < mask_width, np.inf, d)
        if d.size == 0:
            continue
        j = int(np.argmin(d))                          # first minimum == strict `min_diff > diff` update
        if d[j] < loop_diff * loop_diff:
            out.append((i, j))
    return np.array(out, np.int64).reshape(-1, 2)


def precision_recall(diff_v, diff_idx, gt1, gt2, loop_diff: float, mask_width: int):
    """run_test.m:57-85 given the per-query best score / index (0-based).  Returns (AUC, top_recall, lp_detected,
    precision, recall); lp_detected is [top_count, 2] 0-based (query, match)."""
    gt1 = np.asarray(gt1, np.float64)
    gt2 = np.asarray(gt2, np.float64)
    diff_v = np.asarray(diff_v, np.float64)
    diff_idx = np.asarray(diff_idx, np.int64)
    # index -1 = no finite candidate (zero-norm signature: every distance NaN; or everything masked): MATLAB's min over an all-NaN /
    # all-Inf row returns index 1 (run_test.m:57) and the sweep pairs the query with gt2(1,:)
    diff_idx = np.where(diff_idx < 0, 0, diff_idx)
    lp_gt = ground_truth_pairs(gt1, gt2, loop_diff, mask_width)
    L = lp_gt.shape[0]
    total_lp = 0 if L == 0 else max(L, 2)              # MATLAB length() of an L x 2 matrix (run_test.m:22)
    rank = np.argsort(diff_v, kind="stable")           # [~, diff_rank] = sort(diff_v)
    m = gt1.shape[0]
    precision = np.zeros(m)
    recall = np.zeros(m)
    tp = fp = 0
    top_recall = 0.0
    top_count = 0
    for i in range(m):
        a = rank[i]
        b = diff_idx[a]
        d = ((gt1[a] - gt2[b]) ** 2).sum()
        if d < loop_diff * loop_diff:
            tp += 1
        else:
            fp += 1
        precision[i] = tp / (tp + fp)
        recall[i] = tp / total_lp if total_lp else (np.inf if tp else np.nan)   # :77 as MATLAB divides (x/0 = Inf, 0/0 = NaN)
        if precision[i] == 1:
            top_count = i + 1
            top_recall = recall[i]
    auc = 0.0                                          # trapz(recall, precision) (:84), summed in sweep order like pr_precision_recall
    with np.errstate(invalid="ignore"):                # (recall is NaN / Inf without ground-truth pairs)
        for i in range(m - 1):
            auc += (recall[i + 1] - recall[i]) * (precision[i] + precision[i + 1]) / 2.0
    lp_detected = np.stack([rank[:top_count], diff_idx[rank[:top_count]]], 1)
    return auc, float(top_recall), lp_detected, precision, recall


# ------------------------------------------------------------------ the drivers: test_kitti.m, test_robotcar.m
KITTI_SEQS = ("seq00", "seq02", "seq05", "seq06", "seq07")                                  # test_kitti.m:5
ROBOTCAR_DATES = ("2014-07-14-14-49-50", "2014-11-28-12-07-13", "2014-12-12-10-45-15", "2015-02-10-11-58-05",
                  "2015-05-19-14-06-38", "2015-05-22-11-14-30", "2015-08-13-16-02-58", "2015-10-30-13-52-14")   # test_robotcar.m:4-6
ROBOTCAR_PAIRS = ((5, 6), (5, 7), (5, 8), (5, 4), (7, 1), (7, 8), (7, 4), (8, 2), (8, 4), (4, 3))              # :9, 1-based
TYPES = ("delight", "m2dp", "sc", "bow", "gist")                                            # :8


def _load(path):
    """MATLAB `load` of a numeric text file (optionally gzip-compressed, as the test fixtures are)."""
    import gzip
    import os
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        path += ".gz"
    with (gzip.open(path, "rt") if path.endswith(".gz") else open(path)) as f:
        return np.loadtxt(f, ndmin=2)


def load_kitti_ground_truth(seq_dir: str) -> np.ndarray:
    """test_kitti.m:23-25: positions of the poses that became clouds - gt.txt rows `incoming_id + 1` (0-based ids in the
    file), columns 4, 8, 12 of the row-major 3 x 4 pose."""
    import os
    ids = _load(os.path.join(seq_dir, "incoming_id_file.txt")).astype(np.int64).ravel()
    return _load(os.path.join(seq_dir, "gt.txt"))[ids][:, [3, 7, 11]]


def load_robotcar_ground_truth(run_dir: str) -> np.ndarray:
    """test_robotcar.m:31-36: gps.txt rows of the poses that became clouds (all three columns)."""
    import os
    ids = _load(os.path.join(run_dir, "incoming_id_file.txt")).astype(np.int64).ravel()
    return _load(os.path.join(run_dir, "gps.txt"))[ids]


def run_kitti(seq_dir: str, type_: str, hist=None, ctx=None):
    """run_kitti of test_kitti.m:17-29: a sequence against itself, mask_width 100, loop_diff 10 m.  `hist` defaults to
    `<seq_dir>/history_<type>.txt` (or .bin).  Returns (AUC, top_recall, lp_detected)."""
    from . import api
    gt = load_kitti_ground_truth(seq_dir)
    h = _history(seq_dir, type_) if hist is None else hist
    return api.run_test(type_, h, h, gt, gt, 10.0, 100, ctx)


def run_robotcar(run1_dir: str, run2_dir: str, type_: str, hist1=None, hist2=None, ctx=None):
    """run_robotcar of test_robotcar.m:26-41: run 1 against run 2, no mask, loop_diff 25 m."""
    from . import api
    gt1, gt2 = load_robotcar_ground_truth(run1_dir), load_robotcar_ground_truth(run2_dir)
    h1 = _history(run1_dir, type_) if hist1 is None else hist1
    h2 = _history(run2_dir, type_) if hist2 is None else hist2
    return api.run_test(type_, h1, h2, gt1, gt2, 25.0, 0, ctx)


def _history(d: str, type_: str):
    import os
    from . import api
    for ext in (".bin", ".txt"):
        p = os.path.join(d, "history_" + type_ + ext)
        if os.path.exists(p):
            return api.read_signatures(p)
    raise FileNotFoundError(os.path.join(d, "history_" + type_ + ".txt"))


def test_kitti(results_dir: str, types=TYPES, seqs=KITTI_SEQS, ctx=None):
    """test_kitti.m:1-15: (AUCs, TRs), each [len(types), len(seqs)]."""
    import os
    auc = np.zeros((len(types), len(seqs))); tr = np.zeros_like(auc)
    for ti, t in enumerate(types):
        for si, s in enumerate(seqs):
            auc[ti, si], tr[ti, si] = run_kitti(os.path.join(results_dir, "KITTI", s), t, ctx=ctx)[:2]
    return auc, tr


def test_robotcar(results_dir: str, types=TYPES, pairs=ROBOTCAR_PAIRS, ctx=None):
    """test_robotcar.m:1-24: (AUCs, TRs), each [len(types), len(pairs)]; pairs are 1-based indices into ROBOTCAR_DATES."""
    import os
    auc = np.zeros((len(types), len(pairs))); tr = np.zeros_like(auc)
    for ti, t in enumerate(types):
        for si, (a, b) in enumerate(pairs):
            auc[ti, si], tr[ti, si] = run_robotcar(os.path.join(results_dir, "RobotCar", ROBOTCAR_DATES[a - 1]),
                                                   os.path.join(results_dir, "RobotCar", ROBOTCAR_DATES[b - 1]), t, ctx=ctx)[:2]
    return auc, tr


test_kitti.__test__ = False       # (names of the reference's scripts, not pytest tests)
test_robotcar.__test__ = False
