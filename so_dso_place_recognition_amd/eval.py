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
        recall[i] = tp / total_lp if total_lp else np.nan
        if precision[i] == 1:
            top_count = i + 1
            top_recall = recall[i]
    trapz = getattr(np, "trapezoid", None) or np.trapz
    auc = float(trapz(precision, recall))              # trapz(recall, precision)
    lp_detected = np.stack([rank[:top_count], diff_idx[rank[:top_count]]], 1)
    return auc, float(top_recall), lp_detected, precision, recall
