"""Binary intensity channel (SC/SC.cpp:67-72 writes 0 / 1 into channel 1; processSC.m:15-33 on such rows gives
count / sqrt(ones_q ones_d) with an integer count): in split-f16 arithmetic the library runs channel 1 through the single-product kernel on
the hi halves of the packed images and rounds to the integer count when a device-side error bound allows it (include/place_recognition.h:
pr_set_sc_binary), through the split-f16 kernel otherwise.  Either way the distances are the oracle's within 1e-5 (north_star); on the
binary path they are exact up to the fp32 rounding of the final expression."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from so_dso_place_recognition_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from so_dso_place_recognition_amd import api as _api
    return _api


def _binary_state(q, db):
    """What pr_sc_binary_state says about the packed sets of a Matcher over (q, db)."""
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher
    mt = Matcher("sc", q.shape[0], db.shape[0])
    mt.pack_database(torch.from_numpy(db).cuda())
    mt.local_phase1(torch.from_numpy(q).cuda())
    st = C.c_int32(-1)
    mt.ctx.check(mt.lib.pr_sc_binary_state(mt.ctx.h, mt.q, mt.db, C.byref(st)))
    dp, di = (t.cpu().numpy() for t in mt.distances())
    mt.close()
    return st.value, dp, di


def _counts(di, q, db):
    oq = (q[:, 1200:] != 0).sum(1).astype(np.float64)
    od = (db[:, 1200:] != 0).sum(1).astype(np.float64)
    return (1.0 - 2.0 * di.astype(np.float64)) * np.sqrt(np.outer(oq, od))


@pytest.mark.parametrize("m,n", [(37, 101), (70, 1500), (130, 1000), (64, 4096), (1, 300), (5, 77), (8, 5000)])
def test_binary_channel_is_exact_and_matches_the_split_kernel(api, m, n):
    db = synth.sc_database(45, n)
    q, _ = synth.sc_queries(46, db, m)
    state, dp, di = _binary_state(q, db)
    assert state == 1                                               # ~324 ones per signature: inside the bound
    rc, op, oi = oracle_lib.sc_distance(q, db)
    assert np.abs(dp - op).max() < 1e-5 and np.abs(di - oi).max() < 3e-7
    cnt = _counts(di, q, db)
    assert np.abs(cnt - np.rint(cnt)).max() < 2e-4                  # integer counts (what is left is the fp32 rounding of d)
    gp, gi = api.processSC(q, db)                                   # the host call takes the same path
    assert np.array_equal(gp, dp) and np.array_equal(gi, di)
    sp, si = api.processSC(q, db, api.Context(0, sc_binary=False))  # channel 1 in split-f16
    assert np.array_equal(sp, dp)                                   # channel 0: the same kernel, the same bits
    assert np.abs(si - di).max() < 1e-6 and np.abs(si - oi).max() < 1e-5


def test_scaled_binary_rows_and_zero_rows(api):
    """A row whose non-zero entries all equal 3.5 (or 1e-200) normalises to the same 1/sqrt(ones); zero-norm rows stay NaN rows."""
    db = synth.sc_database(45, 400)
    q, _ = synth.sc_queries(46, db, 50)
    db[5, 1200:] *= 3.5
    db[6, 1200:] *= 1e-200
    q[2, 1200:] *= 7.25
    db[9, 1200:] = 0
    q[11, 1200:] = 0
    state, dp, di = _binary_state(q, db)
    assert state == 1
    rc, op, oi = oracle_lib.sc_distance(q, db)
    assert np.array_equal(np.isnan(di), np.isnan(oi)) and np.isnan(di[:, 9]).all() and np.isnan(di[11]).all()
    ok = ~np.isnan(oi)
    assert np.abs(di - oi)[ok].max() < 3e-7 and np.abs(dp - op).max() < 1e-5


@pytest.mark.parametrize("which", ["db", "query", "negative"])
def test_non_binary_rows_take_the_split_kernel(api, which):
    db = synth.sc_database(45, 600)
    q, _ = synth.sc_queries(46, db, 40)
    if which == "db":
        db[123, 1200 + 77] = 0.5 if db[123, 1200 + 77] == 0 else 0.25       # one entry of one row
    elif which == "query":
        q[7, 1200:] = np.where(q[7, 1200:] != 0, synth.uniform(9, 1, 1200) + 0.1, 0.0)
    else:
        db[44, 1200:] *= -1.0                                            # all non-zero entries equal, but negative
    state, dp, di = _binary_state(q, db)
    assert state == 0
    rc, op, oi = oracle_lib.sc_distance(q, db)
    assert np.abs(dp - op).max() < 1e-5 and np.abs(di - oi).max() < 1e-5


def test_dense_signatures_fall_outside_the_bound(api):
    """~1080 ones per signature: a count is 9e-4 of the normalised product, below what one f16 product per term guarantees -
    the gate sends channel 1 through the split-f16 kernel."""
    db = synth.sc_database(45, 300)
    q, _ = synth.sc_queries(46, db, 40)
    u = synth.uniform(7, np.arange(300, dtype=np.uint64), 1200)
    db[:, 1200:] = (u < 0.9).astype(np.float64)
    q[:, 1200:] = np.roll(db[:40, 1200:].reshape(40, 60, 20), 7, axis=1).reshape(40, 1200)
    state, dp, di = _binary_state(q, db)
    assert state == 0
    rc, op, oi = oracle_lib.sc_distance(q, db)
    assert np.abs(di - oi).max() < 1e-5
    # ... and a set that sits inside: ~480 ones (the a-priori bound alone, without the per-pair test, ends at ~400)
    db[:, 1200:] = (u < 0.4).astype(np.float64)
    q[:, 1200:] = np.roll(db[:40, 1200:].reshape(40, 60, 20), 7, axis=1).reshape(40, 1200)
    state, dp, di = _binary_state(q, db)
    rc, op, oi = oracle_lib.sc_distance(q, db)
    assert state == 1 and np.abs(di - oi).max() < 3e-7
    cnt = _counts(di, q, db)
    assert np.abs(cnt - np.rint(cnt)).max() < 2e-4


def test_one_dense_row_switches_the_whole_call(api):
    db = synth.sc_database(45, 500)
    q, _ = synth.sc_queries(46, db, 33)
    db[77, 1200:] = 1.0                                                  # 1200 ones in one signature
    state, dp, di = _binary_state(q, db)
    assert state == 0
    rc, op, oi = oracle_lib.sc_distance(q, db)
    assert np.abs(di - oi).max() < 1e-5


def test_topk_and_timing_hooks(api):
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher
    n, m = 6000, 96
    db = synth.sc_database(45, n)
    q, planted = synth.sc_queries(46, db, m)
    mt = Matcher("sc", m, n)
    mt.ctx.kernel_timing(True)
    mt.pack_database(torch.from_numpy(db).cuda())
    idx, score = mt.match(torch.from_numpy(q).cuda(), mask_width=0, k=3)
    t0, t1, t2 = mt.ctx.last_distance_timing()
    assert t0 > 0 and t1 > t2 >= 0                                       # the single-product launch ran, the split-f16 one behind it left at once
    rc, oidx, osc = oracle_lib.match_topk(0, q, db, 0, 2.0, 3)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert np.abs(score.cpu().numpy() - osc).max() < 1e-4
    mt.close()


def test_both_channel1_passes_at_size(api):
    """A few hundred queries against 20 000 entries (several workgroups per XCD, every query-group slot of the single-product launch in use):
    the binary pass and, with one non-binary row in the DB, the split-f16 pass behind it - each against the oracle on a sample of rows."""
    n, m = 20000, 200
    db = synth.sc_database(45, n)
    q, planted = synth.sc_queries(46, db, m)
    rows = [0, 7, 63, 64, 131, 199]
    for nonbinary in (False, True):
        if nonbinary:
            db[12345, 1200 + 5] = 0.375
        state, dp, di = _binary_state(q, db)
        assert state == (0 if nonbinary else 1)
        rc, op, oi = oracle_lib.sc_distance(q[rows], db)
        assert np.abs(dp[rows] - op).max() < 1e-5 and np.abs(di[rows] - oi).max() < (1e-5 if nonbinary else 3e-7)
        assert (np.argmin(di + 2 * dp, axis=1)[:5] >= 0).all()


@pytest.mark.parametrize("m,n", [(4, 900), (90, 2500)])
def test_a_failed_rounding_test_sends_the_channel_to_the_split_kernel(api, monkeypatch, m, n):
    """PR_SC_BINARY_PAIR_SCALE (a test hook) inflates the bound of the per-pair test only: the single-product pass runs, every pair fails,
    the flag goes up and the split-f16 launch behind it redoes channel 1 - pr_sc_binary_state reports 2, the distances are the split kernel's."""
    db = synth.sc_database(45, n)
    q, _ = synth.sc_queries(46, db, m)
    monkeypatch.setenv("PR_SC_BINARY_PAIR_SCALE", "1000")
    state, dp, di = _binary_state(q, db)
    monkeypatch.delenv("PR_SC_BINARY_PAIR_SCALE")
    assert state == 2
    sp, si = api.processSC(q, db, api.Context(0, sc_binary=False))
    assert np.array_equal(dp, sp) and np.array_equal(di, si)
    state, dp, di = _binary_state(q, db)                                  # ... and the next call starts with a clean flag
    assert state == 1
    rc, op, oi = oracle_lib.sc_distance(q, db)
    assert np.abs(di - oi).max() < 3e-7
