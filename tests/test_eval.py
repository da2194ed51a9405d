"""run_test.m evaluation part (ground-truth pairs, PR sweep, AUC) against a loop-by-loop restatement."""
import numpy as np

from so_dso_place_recognition_amd import eval as ev


def _traj(n=120):
    t = np.linspace(0, 4 * np.pi, n)                       # two laps of a circle -> loop closures one lap apart
    return np.stack([30 * np.cos(t), np.zeros(n), 30 * np.sin(t)], 1)


def _gt_loops(gt1, gt2, loop_diff, mask):
    out = []
    for i in range(len(gt1)):                              # run_test.m:4-21 literally
        min_diff, min_j = np.inf, -1
        for j in range(len(gt2)):
            if abs(i - j) < mask:
                continue
            d = gt1[i] - gt2[j]
            d = float(d @ d)
            if min_diff > d:
                min_diff, min_j = d, j
        if min_diff < loop_diff * loop_diff:
            out.append((i, min_j))
    return np.array(out).reshape(-1, 2)


def test_ground_truth_pairs_match_literal_loops():
    gt = _traj()
    for mask, ld in ((10, 3.0), (0, 1.0), (30, 5.0)):
        assert np.array_equal(ev.ground_truth_pairs(gt, gt, ld, mask), _gt_loops(gt, gt, ld, mask))


def test_precision_recall_sweep():
    gt = _traj()
    n = len(gt)
    lp = _gt_loops(gt, gt, 3.0, 10)
    rng = np.random.default_rng(0)
    diff_idx = rng.integers(0, n, n)
    diff_v = rng.uniform(0, 1, n)
    good = lp[: len(lp) // 2]
    diff_idx[good[:, 0]] = good[:, 1]                        # half of the loops are found, with the best scores
    diff_v[good[:, 0]] = -1 - rng.uniform(0, 1, len(good))
    auc, top_recall, det, prec, rec = ev.precision_recall(diff_v, diff_idx, gt, gt, 3.0, 10)
    assert len(det) >= len(good) and (prec[: len(good)] == 1).all()
    assert abs(top_recall - len(det) / max(len(lp), 2)) < 1e-12 or top_recall >= len(good) / max(len(lp), 2)
    tp = 0                                                   # literal restatement of run_test.m:60-78
    order = np.argsort(diff_v, kind="stable")
    p2 = []
    for i, a in enumerate(order):
        d = gt[a] - gt[diff_idx[a]]
        tp += float(d @ d) < 9.0
        p2.append(tp / (i + 1))
    assert np.allclose(prec, p2)
    assert 0 < auc <= 1
