"""PR_SC_ARITH_F16 - BASELINE.json config 5's "fp16 descriptors": spectra / rows stored as ONE f16, one f16 MFMA per product
(include/place_recognition.h).  No reference counterpart (run_test.m:26-41 is fp64); the bars are SURVEY.md §8-d config 5's: distances
within 1e-3 of the fp64 oracle, top-k INDICES identical to the oracle's - through the fp64 re-evaluation of the k + 56 best of the f16 pass,
the margin check of that list and the split-f16 fallback for the queries that fail it."""
import ctypes as C

import numpy as np
import pytest
import torch

import helpers
import oracle_lib
from so_dso_place_recognition_amd import _lib, synth
from test_gpu_configs import _sharded_run, oracle_rows, topk_rows, zscore_rows

pytestmark = pytest.mark.gpu

BOUND = 2e-3            # PR_F16_DISTANCE_BOUND


@pytest.fixture(scope="module")
def api():
    from so_dso_place_recognition_amd import api as a
    return a


def score_tol_f16(osc):
    """|score - oracle|: the pair's distances are exact (fp64 re-evaluation); the row mean and sigma come from the f16 all-pairs pass.  Its
    distance noise (~3e-5 rms) does not average out of the mean: the minimum over 120 noisy variants is biased low by about one noise
    sigma, i.e. ~3e-5 / sigma_d ~ 3e-3 per channel, 1e-2 on a 2:1 fused score; sigma itself moves by ~1e-4 relative."""
    return 3e-2 + 1e-3 * np.abs(osc)


def test_f16_distances_within_1e3(api):
    ctx = api.Context(0, sc_arith="f16")
    assert ctx.sc_arith == "f16"
    errs = {}
    for seed, n, m in ((45, 500, 40), (52, 1000, 21), (53, 2049, 9), (54, 100, 3)):
        db = synth.sc_database(seed, n)
        q, _ = synth.sc_queries(seed + 100, db, m)
        rc, op, oi = oracle_lib.sc_distance(q, db)
        gp, gi = api.processSC(q, db, ctx)
        errs[("sc", n, m)] = max(np.abs(gp - op).max(), np.abs(gi - oi).max())
    db = synth.m2dp_database(43, 130)
    q, _ = synth.m2dp_queries(44, db, 33)
    rc, op, oi = oracle_lib.m2dp_distance(q, db)
    gp, gi = api.processM2DP(q, db, ctx)
    errs[("m2dp", 130, 33)] = max(np.abs(gp - op).max(), np.abs(gi - oi).max())
    print("max |d_f16 - oracle|:", errs)
    assert max(errs.values()) < 1e-3 < BOUND                       # "report against a 1e-3 tolerance"
    ctx.close()


@pytest.mark.parametrize("kind", ["sc", "m2dp", "fused"])
def test_f16_topk_indices_equal_the_oracle(api, kind):
    ctx = api.Context(0, sc_arith="f16")
    n, m, k, mask = 3000, 150, 4, 3
    dbs = synth.sc_database(45, n)
    qs, planted = synth.sc_queries(46, dbs, m)
    dbm = synth.m2dp_database(43, n)
    qm = np.concatenate([synth.m2dp_queries(44, dbm[4 * int(e): 4 * int(e) + 4], 1)[0] for e in planted])     # planted on the same places
    if kind == "sc":
        rc, oidx, osc = oracle_lib.match_topk(0, qs, dbs, mask, 2.0, k)
        idx, sc = api.match_topk("sc", qs, dbs, mask, 2.0, k, ctx=ctx)
    elif kind == "m2dp":
        rc, oidx, osc = oracle_lib.match_topk(1, qm, dbm, mask, 2.0, k)
        idx, sc = api.match_topk("m2dp", qm, dbm, mask, 2.0, k, ctx=ctx)
    else:
        rc, oidx, osc = oracle_lib.match_topk_fused(qs, qm, dbs, dbm, mask, 2.0, k)
        idx, sc = api.match_topk_fused(qs, qm, dbs, dbm, mask, 2.0, k, ctx=ctx)
    assert rc == 0 and np.array_equal(idx, oidx)
    err = np.abs(sc - osc) / score_tol_f16(osc)
    print(kind, "max |score - oracle|:", np.abs(sc - osc).max(), " / tolerance:", err.max())
    assert err.max() <= 1.0
    ctx.close()


def test_f16_near_ties_are_ordered_by_the_fp64_reevaluation(api):
    """Twins whose distances to a query differ by 1e-8 ... 1e-4 - far below what ONE f16 product resolves (~1e-4): the f16 pass cannot
    order them, the returned order must still be the oracle's."""
    n, m = 1500, 240
    db = synth.sc_database(45, n)
    q, planted = synth.sc_queries(46, db, m)
    rng = np.random.default_rng(12)
    twins = np.setdiff1d(np.arange(n), planted)[-m:]
    for t in range(m):
        delta = 10.0 ** rng.uniform(-7.5, -3.5)
        e = db[planted[t]].copy()
        occ = np.nonzero(e[:1200] > 0)[0]
        pick = rng.choice(occ, size=40, replace=False)
        e[pick] *= 1.0 + delta * rng.standard_normal(40) * 50
        db[twins[t]] = e
    rc, oidx, osc = oracle_lib.match_topk(0, q, db, 0, 2.0, 3)
    ctx = api.Context(0, sc_arith="f16")
    dp, di = api.processSC(q, db, ctx)
    t = np.arange(m)
    dgap = np.abs(dp[t, planted] - dp[t, twins]) + np.abs(di[t, planted] - di[t, twins])
    rc, op, oi = oracle_lib.sc_distance(q, db)
    ogap = np.abs(op[t, planted] - op[t, twins]) + np.abs(oi[t, planted] - oi[t, twins])
    wrong_in_pass = ((dp[t, planted] < dp[t, twins]) != (op[t, planted] < op[t, twins])).sum()
    print("twin gaps (oracle) median", np.median(ogap), "pairs the f16 pass orders the other way:", wrong_in_pass, "pass gap median", np.median(dgap))
    assert wrong_in_pass > m // 10                                    # the test has teeth: the f16 distances alone get many of them wrong
    idx, sc = api.match_topk("sc", q, db, 0, 2.0, 3, ctx=ctx)
    gap = np.abs(osc[:, 0] - osc[:, 1])
    same = (idx == oidx).all(1)
    print("rows with the oracle's order:", same.sum(), "of", m, "; oracle score gaps of the others:", np.sort(gap[~same]))
    # The pair's distances are exact, but the 2:1 weights of the two channels are 1 / sigma of the f16 pass (~1e-4 relative off): twins whose
    # two channel differences cancel to that precision are not decidable in this arithmetic.  They must still hold the same entries.
    assert same.sum() >= m - 3 and (gap[~same] < 1e-4).all()
    assert np.array_equal(np.sort(idx[:, :2], 1), np.sort(oidx[:, :2], 1))
    assert (np.abs(sc - osc) <= score_tol_f16(osc)).all()
    ctx.close()


def test_f16_margin_check_and_split_fallback(api):
    """A family of 300 near-copies of one entry: more entries inside the f16 error bound than the candidate list holds, so the list does
    not provably contain the exact top-k.  The queries planted on that family must be flagged and recomputed in split-f16 (warning bit),
    everything must equal the oracle; queries elsewhere are not flagged."""
    from so_dso_place_recognition_amd.matcher import Matcher
    n, m, k = 4000, 64, 3
    db = synth.sc_database(45, n)
    q, planted = synth.sc_queries(46, db, m)
    rng = np.random.default_rng(5)
    fam = np.setdiff1d(np.arange(n), planted)[-300:]
    base = db[planted[0]].copy()
    for j in fam:
        e = base.copy()
        occ = np.nonzero(e[:1200] > 0)[0]
        e[rng.choice(occ, size=30, replace=False)] *= 1.0 + 3e-4 * rng.standard_normal(30)
        db[j] = e
    q[1:8] = q[0] * (1.0 + 1e-3 * rng.standard_normal((7, 2400)))     # eight queries on the family
    for mask in (0, 5):
        rc, oidx, osc = oracle_lib.match_topk(0, q, db, mask, 2.0, k)
        ctx = api.Context(0, sc_arith="f16")
        idx, sc = api.match_topk("sc", q, db, mask, 2.0, k, ctx=ctx)
        assert ctx.take_warnings() & _lib.WARN_F16_FALLBACK
        assert np.array_equal(idx, oidx)
        assert (np.abs(sc[:8] - osc[:8]) <= helpers.score_tol(osc[:8])).all()          # the recomputed rows carry split-f16 statistics
        assert (np.abs(sc[8:] - osc[8:]) <= score_tol_f16(osc[8:])).all()
        # the device-resident path: flags on the family's queries only, the same results
        mt = Matcher("sc", m, n, ctx=api.Context(0, sc_arith="f16", stream=int(torch.cuda.current_stream().cuda_stream)))
        mt.pack_database(torch.from_numpy(db).cuda())
        i2, s2 = mt.match(torch.from_numpy(q).cuda(), mask, 2.0, k)
        fl = mt.f16_flags.cpu().numpy()
        # (queries elsewhere: k = 3 reaches into the dense part of a row, where some of the 56 candidates left out sit within the f16 pass's
        #  sigma uncertainty - 2e-4 + 2e-4 / sigma, rigorous for any error pattern since round 5 - of the third: those go to the split pass too)
        assert fl[:8].all() and mt.f16_fallbacks == int(fl.sum()) and fl[8:].sum() <= (m - 8) // 2
        assert np.array_equal(i2.cpu().numpy(), oidx)
        i3, s3 = mt.match(torch.from_numpy(q).cuda(), mask, 2.0, k, f16_fallback=False)   # without the second pass the flags are the caller's business
        assert np.array_equal(i3.cpu().numpy()[8:][fl[8:] == 0], oidx[8:][fl[8:] == 0])
        mt.close(); ctx.close()


def test_f16_config5_fused_1m_db_in_8_shards(api):
    """BASELINE.json config 5 as written: fused SC + M2DP scoring, 1 M signatures in 8 shards, f16 descriptors - shard by shard on this
    GPU with the production protocol; 8 oracle rows over the whole DB."""
    from so_dso_place_recognition_amd.api import Context
    from so_dso_place_recognition_amd.matcher import FusedMatcher
    n, G, m, k, mask = 1_000_000, 8, 64, 2, 0
    shards = [(n * g // G, n * (g + 1) // G) for g in range(G)]
    R = 8
    rows = np.arange(R)
    q_sc, planted = synth.sc_queries(72, np.empty((0, 2400)), m, db_first=0, n_global=n, db_seed=71)
    rows_m2 = np.concatenate([synth.m2dp_database(73, 1, first=int(e)) for e in planted]).reshape(m, 4, 2, 192)
    u = synth.uniform(74, np.arange(m, dtype=np.uint64), 1 + 4 * 384)
    rows_m2 = rows_m2 + 0.05 * (u[:, 1:].reshape(m, 4, 2, 192) - 0.5)
    uu = rows_m2[..., :64] / np.sqrt((rows_m2[..., :64] ** 2).sum(-1, keepdims=True))
    vv = rows_m2[..., 64:] / np.sqrt((rows_m2[..., 64:] ** 2).sum(-1, keepdims=True))
    q_m2 = np.concatenate([uu, vv], -1).reshape(4 * m, 384)
    tq_sc, tq_m2 = torch.from_numpy(q_sc).cuda(), torch.from_numpy(q_m2).cuda()
    d = {c: [] for c in range(4)}
    cur = int(torch.cuda.current_stream().cuda_stream)
    f16err = [0.0]

    def pack(mt, lo, hi):
        a = synth.sc_database_torch(71, hi - lo, first=lo)
        b = synth.m2dp_database_torch(73, hi - lo, first=lo)
        mt.pack_database(a, b)
        dp, di = oracle_rows("sc", q_sc[rows], [a.cpu().numpy()])
        ep, ei = oracle_rows("m2dp", q_m2.reshape(m, 4, 384)[rows].reshape(-1, 384), [b.cpu().numpy()])
        for c, x in enumerate((dp, di, ep, ei)):
            d[c].append(x)

    ms, per, idx, sc = _sharded_run(lambda cap: FusedMatcher(m, cap, ctx=Context(0, sc_arith="f16", stream=cur)), shards, pack, (tq_sc, tq_m2), mask, k)
    assert ms[0].f16 and ms[0]._bufs["idx_in"].shape[1] == k + 56
    for g, mt in enumerate(ms):                                       # the f16 pass's own distances against the oracle rows
        for c, t in enumerate((*mt.sc.distances(), *mt.m2.distances())):
            f16err[0] = max(f16err[0], np.abs(t[rows].cpu().numpy() - d[c][g]).max())
    print("max |d_f16 - oracle| over 8 rows x 1M x 4 channels:", f16err[0])
    assert f16err[0] < 1e-3
    assert np.array_equal(idx[:, 0], planted)
    full = [np.concatenate(d[c], 1) for c in range(4)]
    f = 2.0 * zscore_rows(full[0]) + zscore_rows(full[1]) + 2.0 * zscore_rows(full[2]) + zscore_rows(full[3])
    oi, osc = topk_rows(f, rows, mask, k)
    assert np.array_equal(idx[rows], oi)
    assert (np.abs(sc[rows] - osc) <= 2 * score_tol_f16(osc)).all()
    for mt in ms:
        mt.close()


def test_merge_topk_has_no_cap_on_shards_times_width(api):
    """pr_merge_topk_dev is a G-way merge of ascending lists: 8 shards x 58 candidates (464 entries per query; the first version selected
    over a gathered array of at most 128) against the torch restatement, with missing entries at the end of one shard's lists."""
    from so_dso_place_recognition_amd.matcher import Matcher, _merge_dev, merge_topk
    G, m, k = 8, 64, 58
    g = torch.Generator().manual_seed(1)
    sc = torch.sort(torch.randn(G, m, k, generator=g, dtype=torch.float64), dim=2).values
    idx = torch.stack([torch.stack([torch.randperm(1000, generator=g)[:k] + 1000 * gg for _ in range(m)]) for gg in range(G)]).to(torch.int32)
    idx[3, :, -5:] = -1
    sc[3, :, -5:] = float("nan")
    sc[5, :, 7] = sc[5, :, 6]                                         # exact ties inside a list: the lower index is first
    idx[5, :, 6:8] = torch.sort(idx[5, :, 6:8], dim=1).values
    ri, rs = merge_topk(idx, sc, k)
    mt = Matcher("sc", 8, 16)
    for kk in (k, 9):
        di, ds = _merge_dev(mt, idx[:, :, :kk].contiguous().cuda(), sc[:, :, :kk].contiguous().cuda(), kk)
        ri, rs = merge_topk(idx[:, :, :kk].contiguous(), sc[:, :, :kk].contiguous(), kk)
        assert torch.equal(di.cpu(), ri) and torch.equal(ds.cpu(), rs)
    mt.close()


def test_f16_order_check_sends_sigma_sensitive_rows_to_the_split_pass(api):
    """Found by tools/fuzz_all.py (seed 1, case 6): M2DP, 60 x 3498, k = 57.  Deep in a list the fused scores of neighbours differ by less
    than what the f16 pass's sigma error (~1e-4 relative) moves them when their two channels disagree about the order: the re-evaluated
    pairs came out swapped at ranks 31 / 32 of one row.  The order check (pr_rerank_dev + pr_f16_margin_dev) now flags such queries and the
    split-f16 pass answers them - through the host call and through the device-resident Matcher (whose fallback used to index the query
    rows of an M2DP batch as if a signature were one row)."""
    from so_dso_place_recognition_amd.matcher import Matcher
    m, n, k = 60, 3498, 57
    db = synth.m2dp_database(3000 + 7 * 6 + 1, n)
    q, _ = synth.m2dp_queries(4000 + 6 + 1, db, m)
    rc, oidx, osc = oracle_lib.match_topk(1, q, db, 0, 2.0, k)
    assert rc == 0
    ctx = api.Context(0, sc_arith="f16")
    idx, sc = api.match_topk("m2dp", q, db, 0, 2.0, k, ctx=ctx)
    assert ctx.take_warnings() & _lib.WARN_F16_FALLBACK
    assert np.array_equal(idx, oidx)
    assert (np.abs(sc - osc) <= score_tol_f16(osc)).all()
    ctx.close()
    dev = torch.device("cuda", 0)
    mt = Matcher("m2dp", m, n, ctx=api.Context(0, sc_arith="f16", stream=int(torch.cuda.current_stream(dev).cuda_stream)))
    mt.pack_database(torch.from_numpy(db).to(dev))
    i2, s2 = mt.match(torch.from_numpy(q).to(dev), 0, 2.0, k)
    assert mt.f16_fallbacks > 0
    assert np.array_equal(i2.cpu().numpy(), oidx)
    # without the fallback the f16 answer differs in at least one of the flagged rows: the check has teeth
    i3, _ = mt.match(torch.from_numpy(q).to(dev), 0, 2.0, k, f16_fallback=False)
    flagged = mt.f16_flags.cpu().numpy().astype(bool)
    diff = (i3.cpu().numpy() != oidx).any(axis=1)
    assert diff.any() and not (diff & ~flagged).any()
    mt.close()
