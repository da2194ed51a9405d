"""Edge shapes of the top-k calls against the oracle (GPU box): n = 1, k = n, k > n, everything masked, m = 0, one-channel zero rows everywhere."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import helpers
import oracle_lib
from so_dso_place_recognition_amd import api, synth
bad = 0
def run(tag, type_, q, db, mask, k):
    global bad
    t = 0 if type_ == "sc" else 1
    rc, oidx, osc = oracle_lib.match_topk(t, q, db, mask, 2.0, k)
    rc2, odp, odi = oracle_lib.sc_distance(q, db) if type_ == "sc" else oracle_lib.m2dp_distance(q, db)
    with np.errstate(all="ignore"):      # tests/helpers.py: the error model of fp32 row statistics (rows of 2 - 40 entries: its third term)
        tol = np.broadcast_to(helpers.score_tol(np.where(np.isfinite(osc), osc, 0.0), helpers.row_sigmas(odp, odi), eps=1e-7), osc.shape)
    for arith in ("f16x2", "f32", "f16"):
        ctx = api.Context(0, sc_arith=arith)
        try:
            idx, sc = api.match_topk(type_, q, db, mask, 2.0, k, ctx=ctx)
            ok = np.array_equal(idx, oidx) and np.array_equal(np.isfinite(sc), np.isfinite(osc))
            fin = np.isfinite(osc)
            if ok and fin.any() and arith != "f16" and db.shape[0] >= 8 * (1 if type_ == "sc" else 4):   # (a sigma of two samples is no statistic)
                ok = (np.abs(sc - osc)[fin] <= tol[fin]).all()
            if not ok:
                bad += 1
                r = np.argwhere((idx != oidx) | (np.isfinite(sc) != np.isfinite(osc)) | (np.abs(np.where(fin, sc - osc, 0.0)) > np.where(fin, tol, np.inf)))
                print("BAD", tag, type_, arith, r[:4].tolist(), [(idx[tuple(x)], oidx[tuple(x)], sc[tuple(x)], osc[tuple(x)]) for x in r[:4]])
        except Exception as e:
            print("EXC", tag, type_, arith, repr(e)[:200]); bad += 1
        ctx.close()
for type_ in ("sc", "m2dp"):
    mk = (lambda s, n: synth.sc_database(s, n)) if type_ == "sc" else (lambda s, n: synth.m2dp_database(s, n))
    div = 1 if type_ == "sc" else 4
    db = mk(5, 40); q = db[: 7 * div].copy()
    try:                                   # n = 1: no N-1 standard deviation - the library refuses (MATLAB returns NaN scores)
        api.match_topk(type_, q, db[:div], 0, 2.0, 1); print("n=1 accepted?"); bad += 1
    except api.PRError as e:
        assert e.code == -1
    run("n=2,k=2", type_, q, db[: 2 * div], 0, 2)
    run("k=n", type_, q, db, 0, 40)
    run("k>n", type_, q, db[: 5 * div], 0, 9)
    run("all masked", type_, q, db, 100, 3)
    run("mask leaves one", type_, q[:div], db[: 6 * div], 5, 3)
    run("k=120", type_, q, mk(6, 300), 0, 120)
    if type_ == "sc":
        z = db.copy(); z[:, 1200:] = 0.0
        run("intensity channel zero everywhere", type_, q, z, 0, 3)
        z = q.copy(); z[:, :1200] = 0.0
        run("structure channel zero in every query", type_, z, db, 0, 3)
    try:
        idx, sc = api.match_topk(type_, q[:0], db, 0, 2.0, 1)
        assert idx.shape == (0, 1)
    except Exception as e:
        print("EXC m=0", type_, repr(e)[:200]); bad += 1
print("edge cases:", "ok" if not bad else f"{bad} findings")
sys.exit(1 if bad else 0)
