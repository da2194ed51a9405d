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


# ------------------------------------------------------------------ the drivers (test_kitti.m, test_robotcar.m) on reference data
import gzip
import os

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sequences")


def _rff(pos, dim=96, scale=15.0, seed=3):
    """A stand-in descriptor that is a smooth function of position (random Fourier features): near places, near vectors."""
    rng = np.random.default_rng(seed)
    w = rng.normal(0, 1.0 / scale, size=(pos.shape[1], dim)); b = rng.uniform(0, 2 * np.pi, dim)
    return np.cos((pos - pos.mean(0)) @ w + b) * np.sqrt(2.0 / dim)


def test_ground_truth_loaders_on_the_reference_files():
    d = os.path.join(GOLD, "kitti_seq06")
    ids = np.loadtxt(os.path.join(d, "incoming_id_file.txt")).astype(int)
    full = np.loadtxt(gzip.open(os.path.join(d, "gt.txt.gz"), "rt"))
    gt = ev.load_kitti_ground_truth(d)
    assert gt.shape == (880, 3) and np.array_equal(gt, full[ids][:, [3, 7, 11]])      # test_kitti.m:23-25 (1-based there)
    assert np.array_equal(gt[0], full[ids[0], [3, 7, 11]]) and ids[0] >= 30          # 30 warm-up poses after a reset (pts_preprocess.h:203)
    d = os.path.join(GOLD, "robotcar_2015-05-19-14-06-38")
    ids = np.loadtxt(os.path.join(d, "incoming_id_file.txt")).astype(int)
    gps = ev.load_robotcar_ground_truth(d)
    assert gps.shape == (len(ids), 3) and np.array_equal(gps, np.loadtxt(gzip.open(os.path.join(d, "gps.txt.gz"), "rt"))[ids])
    assert ev.ROBOTCAR_DATES[5 - 1] == "2015-05-19-14-06-38" and ev.ROBOTCAR_PAIRS[0] == (5, 6) and len(ev.ROBOTCAR_PAIRS) == 10


@pytest.mark.gpu
def test_drivers_run_kitti_and_run_robotcar():
    """run_kitti / run_robotcar with the reference's ground truth, ids, masks and thresholds; the signatures are stand-ins
    (the reference's own are missing blobs), matched as type 'gist'.  The harness must find the loops the positions imply."""
    d = os.path.join(GOLD, "kitti_seq06")
    gt = ev.load_kitti_ground_truth(d)
    auc, top_recall, det = ev.run_kitti(d, "gist", hist=_rff(gt))
    lp = ev.ground_truth_pairs(gt, gt, 10.0, 100)
    assert len(lp) > 50 and auc > 0.9 and top_recall > 0.5                            # seq06 closes its loop
    assert all(((gt[a] - gt[b]) ** 2).sum() < 100.0 and abs(a - b) >= 100 for a, b in det)
    d1, d2 = (os.path.join(GOLD, "robotcar_" + ev.ROBOTCAR_DATES[i - 1]) for i in ev.ROBOTCAR_PAIRS[0])
    g1, g2 = ev.load_robotcar_ground_truth(d1), ev.load_robotcar_ground_truth(d2)
    both = np.concatenate([g1, g2])
    f = _rff(both, scale=40.0)
    auc, top_recall, det = ev.run_robotcar(d1, d2, "gist", hist1=f[: len(g1)], hist2=f[len(g1):])
    assert auc > 0.9 and top_recall > 0.3 and all(((g1[a] - g2[b]) ** 2).sum() < 625.0 for a, b in det)


def test_c_abi_precision_recall_equals_the_python_restatement():
    """pr_precision_recall (what `match_signatures --gt1 --gt2 --loop_diff` prints) against eval.precision_recall."""
    import ctypes as C
    from so_dso_place_recognition_amd import _lib, eval as E
    lib = _lib.load()
    rng = np.random.default_rng(1)
    for m, n, mask, ld in ((200, 220, 5, 10.0), (50, 50, 0, 3.0), (7, 3, 2, 1e9), (5, 5, 100, 10.0)):
        gt1 = np.cumsum(rng.normal(0, 3, (m, 3)), 0)
        gt2 = np.concatenate([gt1[:n // 2] + rng.normal(0, 1, (n // 2, 3)), rng.normal(0, 100, (n - n // 2, 3))])[:n]
        v = rng.random(m); idx = rng.integers(0, n, m).astype(np.int32)
        idx[:min(m, n) // 2] = np.arange(min(m, n) // 2); v[:min(m, n) // 2] *= 0.3
        if m >= 50:                                  # queries without a finite candidate (index -1, NaN score): MATLAB's min gives index 1
            idx[m - 3:] = -1; v[m - 3:] = np.nan
            gt1[m - 1] = gt2[0]                      # ... so this one is a true positive at the end of the sweep
            idx0 = np.where(idx < 0, 0, idx)
            ref = E.precision_recall(v, idx0, gt1, gt2, ld, mask)
            got = E.precision_recall(v, idx, gt1, gt2, ld, mask)
            assert ref[0] == got[0] and np.array_equal(ref[3], got[3]) and ref[3][-1] > ref[3][-2]
        a = E.precision_recall(v, idx, gt1, gt2, ld, mask)
        auc, tr, nd = C.c_double(), C.c_double(), C.c_int32()
        lp = np.zeros((m, 2), np.int32)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        assert lib.pr_precision_recall(p(v), p(idx), m, p(gt1), p(gt2), n, 3, ld, mask, C.byref(auc), C.byref(tr), p(lp), C.byref(nd)) == 0
        assert (np.isnan(a[0]) and np.isnan(auc.value)) or abs(a[0] - auc.value) < 1e-12
        assert (a[1] == tr.value or (np.isnan(a[1]) and np.isnan(tr.value))) and nd.value == len(a[2]) and np.array_equal(lp[:nd.value], a[2])
