"""run_test.m evaluation part (ground-truth pairs, PR sweep, AUC) against a loop-by-loop restatement."""
import numpy as np

import oracle_lib
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
    o = oracle_lib.precision_recall(diff_v, diff_idx, gt, gt, 3.0, 10)      # the oracle's line-for-line run_test.m:3-22, 58-85
    assert auc == o["auc"] and top_recall == o["top_recall"] and np.array_equal(det, o["lp_detected"])
    assert np.array_equal(prec, o["precision"]) and np.array_equal(rec, o["recall"]) and np.array_equal(lp, o["lp_gt"])
    assert 0 < auc <= 1


def _c_sweep(v, idx, gt1, gt2, ld, mask):
    import ctypes as C
    from so_dso_place_recognition_amd import _lib
    lib = _lib.load()
    v = np.ascontiguousarray(v, np.float64); idx = np.ascontiguousarray(idx, np.int32)
    gt1 = np.ascontiguousarray(gt1, np.float64); gt2 = np.ascontiguousarray(gt2, np.float64)
    m, n = len(gt1), len(gt2)
    auc, tr, nd = C.c_double(), C.c_double(), C.c_int32()
    lp = np.zeros((max(m, 1), 2), np.int32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    assert lib.pr_precision_recall(p(v), p(idx), m, p(gt1), p(gt2), n, gt1.shape[1], ld, mask, C.byref(auc), C.byref(tr), p(lp), C.byref(nd)) == 0
    return auc.value, tr.value, lp[:nd.value].astype(np.int64)


def _same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


def test_auc_top_recall_and_lp_detected_equal_the_oracle_sweep():
    """eval.precision_recall AND pr_precision_recall against oracle/pr_ref.cpp's literal run_test.m:3-22, 58-85 (sort, tp / fp loop,
    top_count, trapz): AUC, top_recall and lp_detected must be EQUAL - also with NaN / +Inf scores, index -1 (no finite candidate:
    a zero-norm query, or a mask that covers the whole DB), ties in diff_v, a single ground-truth pair (length() of a 1 x 2 matrix
    is 2) and no ground-truth pair at all (recall = 0/0)."""
    rng = np.random.default_rng(7)
    cases = []
    for m, n, mask, ld in ((200, 220, 5, 10.0), (50, 50, 0, 3.0), (7, 3, 2, 1e9), (5, 5, 100, 10.0), (64, 64, 8, 4.0), (40, 90, 0, 1e-9), (1, 1, 0, 1.0)):
        gt1 = np.cumsum(rng.normal(0, 3, (m, 3)), 0)
        hn = min(m, n // 2)
        gt2 = np.concatenate([gt1[:hn] + rng.normal(0, 1, (hn, 3)), rng.normal(0, 100, (n - hn, 3))])
        v = rng.random(m); idx = rng.integers(0, n, m).astype(np.int32)
        h = min(m, n) // 2
        idx[:h] = np.arange(h); v[:h] *= 0.3
        cases.append((v, idx, gt1, gt2, ld, mask))
        if m >= 40:
            v2, i2 = v.copy(), idx.copy()
            v2[m - 3:] = np.nan; i2[m - 3:] = -1                  # zero-norm queries: every distance NaN
            v2[m - 6:m - 3] = np.inf                              # MATLAB's all-masked rows: min = +Inf at index 1 ...
            i2[m - 6:m - 3] = 0
            v2[5:9] = v2[5]                                       # ties keep their query order (stable sort)
            g1 = gt1.copy(); g1[m - 1] = gt2[0]                   # ... and the last one is then a true positive at the end of the sweep
            cases.append((v2, i2, g1, gt2, ld, mask))
            cases.append((np.full(m, np.nan), np.full(m, -1, np.int32), gt1, gt2, ld, mask))   # nothing matched at all
    one = np.zeros((6, 3)); one[:, 0] = np.arange(6) * 100.0
    two = one.copy(); two[3] = one[3] + 0.5; two[[0, 1, 2, 4, 5]] += 1e4         # exactly one ground-truth pair
    cases.append((np.array([.5, .4, .3, .1, .2, .6]), np.array([1, 2, 0, 3, 3, 3], np.int32), one, two, 1.0, 0))
    for v, idx, gt1, gt2, ld, mask in cases:
        o = oracle_lib.precision_recall(v, idx, gt1, gt2, ld, mask)
        auc, tr, det, prec, rec = ev.precision_recall(v, idx, gt1, gt2, ld, mask)
        assert _same(auc, o["auc"]) and _same(tr, o["top_recall"]) and np.array_equal(det, o["lp_detected"])
        assert np.array_equal(prec, o["precision"]) and np.array_equal(rec, o["recall"], equal_nan=True)
        assert np.array_equal(ev.ground_truth_pairs(gt1, gt2, ld, mask), o["lp_gt"])
        cauc, ctr, cdet = _c_sweep(v, idx, gt1, gt2, ld, mask)
        assert _same(cauc, o["auc"]) and _same(ctr, o["top_recall"]) and np.array_equal(cdet, o["lp_detected"])


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


@pytest.mark.gpu
def test_run_test_end_to_end_equals_the_oracle_chain():
    """run_test(type, hist1, hist2, gt1, gt2, loop_diff, mask_width) (run_test.m:1-85) on the GPU against the oracle's chain
    pr_ref_match_topk (k = 1) -> pr_ref_precision_recall: AUC, top_recall and lp_detected equal.  A drive that passes every place twice
    (signatures of the second lap = perturbed copies of the first), with two zero-norm queries and a mask wider than the lap gap of the
    first places."""
    from so_dso_place_recognition_amd import api, synth
    n = 360
    base = synth.sc_database(45, n // 2)
    lap2, _ = synth.sc_queries(46, base, n // 2)                      # query t copies a random entry: reorder to entry order
    rng = np.random.default_rng(5)
    hist = np.concatenate([base, base * (1.0 + 0.02 * rng.random(base.shape))])
    hist[7] = 0.0; hist[200] = 0.0                                   # zero-norm rows: NaN distances, index -1 / MATLAB index 1
    t = np.linspace(0, 4 * np.pi, n, endpoint=False)
    gt = np.stack([50 * np.cos(t), np.zeros(n), 50 * np.sin(t)], 1)
    for mask, ld in ((100, 10.0), (0, 5.0), (190, 10.0)):            # (run_test with ground truth answers every query from its exact fp64 row: the
        auc, tr, det = api.run_test("sc", hist, hist, gt, gt, ld, mask)   #  sweep ranks the QUERIES by score, run_test.m:58)
        rc, oidx, osc = oracle_lib.match_topk(0, hist, hist, mask, 2.0, 1)
        o = oracle_lib.precision_recall(osc[:, 0], oidx[:, 0], gt, gt, ld, mask)
        assert _same(auc, o["auc"]) and _same(tr, o["top_recall"]) and np.array_equal(det, o["lp_detected"])
        assert mask == 190 or (len(det) > 50 and tr > 0.2)


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
