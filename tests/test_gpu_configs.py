"""BASELINE.json's configurations at their stated sizes on one MI355X, each compared with the CPU oracle on a sample the
oracle finishes in seconds (tests/test_gpu_parity.py holds the small exhaustive cases):

  config 2  5000 clouds x 50 000 points -> SC signatures -> m = n = 5000 match (mask 0 and 100)
  config 3  50 000-signature M2DP DB x 4096 queries
  config 4  200 000-signature SC DB in 8 row shards of 25 000 (every shard as its rank would run it, on this one GPU)
  config 5  fused SC + M2DP scoring, 1 000 000 signatures in 8 shards of 125 000
  + a 2000-frame drive (consecutive, overlapping clouds; second lap = loop closures) with mask 100 as test_kitti.m:19
  + near-ties that fp32 distances cannot order (the fp64 re-evaluation must)

Inputs are drawn in HBM by the torch twins of the synth samplers (bit-identical, tests/test_synth_torch.py).
The oracle's full rows over a 10^5 - 10^6-entry DB are computed with the roles swapped - oracle(DB chunk as hist1, the few
query rows as hist2), transposed - because the oracle parallelises over hist1 rows; processSC's distance is symmetric in
its arguments (a shift of one side is the opposite shift of the other, a mirror stays a mirror), so these ARE the oracle's
distances up to the order of two fp64 additions."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import helpers
import oracle_lib
from so_dso_place_recognition_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from so_dso_place_recognition_amd import api as _api
    return _api


def P(t):
    return C.c_void_p(t.data_ptr())


def oracle_rows(kind, q_rows, db_chunks):
    """Full oracle distance rows of a few queries over a chunked DB: list of (dp [R, n], di [R, n])."""
    fn = oracle_lib.sc_distance if kind == "sc" else oracle_lib.m2dp_distance
    dp, di = [], []
    for chunk in db_chunks:
        rc, a, b = fn(chunk, q_rows)                       # roles swapped: parallel over the chunk's rows
        dp.append(a.T); di.append(b.T)
    return np.concatenate(dp, 1), np.concatenate(di, 1)


def zscore_rows(d):
    mu = d.mean(1, keepdims=True)
    sd = np.sqrt(((d - mu) ** 2).sum(1, keepdims=True) / (d.shape[1] - 1))
    return (d - mu) / sd


def topk_rows(f, rows, mask, k, lo=0):
    """run_test.m:47-57 on fused rows f [R, n] whose queries are global rows `rows`; columns are global lo.."""
    f = f.copy()
    j = lo + np.arange(f.shape[1])[None, :]
    f[np.abs(np.asarray(rows)[:, None] - j) < mask] = np.inf
    order = np.argsort(f, axis=1, kind="stable")[:, :k]
    return (order + lo).astype(np.int32), np.take_along_axis(f, order, 1)


# ------------------------------------------------------------------------------------------------ config 2
def test_config2_generate_5000x50000_then_match(api):
    from so_dso_place_recognition_amd.matcher import Matcher
    N, PTS, B = 5000, 50_000, 500
    ctx = api.Context(0)
    sig = torch.empty((N, 2400), dtype=torch.float64, device="cuda")
    for c0 in range(0, N, B):                                         # 0.7 GB of points per batch
        xyz, it, offs = synth.scene_clouds_torch(42, B, PTS, first=c0)
        torch.cuda.synchronize()
        ctx.check(ctx.lib.pr_sc_generate_dev(ctx.h, P(xyz), P(it), P(offs), B, 45.0, P(sig[c0:c0 + B])))
    sig_h = sig.cpu().numpy()
    sample = np.unique(np.concatenate([np.arange(0, N, 167), [1, 499, 500, N - 1]]))          # 34 clouds
    for c in sample:
        xyz, it, offs = synth.scene_clouds_torch(42, 1, PTS, first=int(c))
        o = oracle_lib.sc_generate(xyz.cpu().numpy(), it.cpu().numpy(), offs.cpu().numpy())[0]
        assert np.array_equal(sig_h[c, 1200:], o[1200:]), c                                     # binary channel exact
        assert np.array_equal(sig_h[c, :1200] > 0, o[:1200] > 0) and np.abs(sig_h[c, :1200] - o[:1200]).max() < 1e-10, c
    mt = Matcher("sc", N, N)
    mt.pack_database(sig)
    rows = np.array([0, 3, 101, 977, 2048, 2500, 3999, 4999])
    dp, di = oracle_rows("sc", sig_h[rows], [sig_h])
    for mask in (0, 100):
        idx, sc = mt.match(sig, mask_width=mask, p_weight=2.0, k=3)
        idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
        gp, gi = mt.distances()
        assert np.abs(gp[rows].cpu().numpy() - dp).max() < 1e-5 and np.abs(gi[rows].cpu().numpy() - di).max() < 1e-5
        oi, osc = topk_rows(2.0 * zscore_rows(dp) + zscore_rows(di), rows, mask, 3)
        assert np.array_equal(idx[rows], oi)
        assert (np.abs(sc[rows] - osc) <= helpers.score_tol(osc)).all()
        flat = np.abs(osc) < 30                                                                  # real matches (|z| < 30) are inside
        assert flat.sum() >= 8                                                                   # BASELINE.json's 1e-5 outright
        assert (np.abs(sc[rows] - osc)[flat] < 1e-5).all(), np.abs(sc[rows] - osc)[flat].max()
        if mask == 0:
            assert np.array_equal(idx[:, 0], np.arange(N))                                       # every cloud finds itself
    mt.close(); ctx.close()


# ------------------------------------------------------------------------------------------------ config 3
def test_config3_m2dp_50k_db_4096_queries(api):
    from so_dso_place_recognition_amd.matcher import Matcher
    n, m = 50_000, 4096
    db = synth.m2dp_database_torch(43, n)
    db_h = db.cpu().numpy()
    q_h, planted = synth.m2dp_queries(44, db_h, m)
    q = torch.from_numpy(q_h).cuda()
    mt = Matcher("m2dp", m, n)
    mt.pack_database(db)
    idx, sc = mt.match(q, 0, 2.0, 2)
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    assert np.array_equal(idx[:, 0], planted)                                                    # planted recall 100 %
    rows = np.array([0, 1, 77, 1000, 2047, 2048, 4000, 4095])
    qr = q_h.reshape(m, 4, 384)[rows].reshape(-1, 384)
    dp, di = oracle_rows("m2dp", qr, [db_h])
    gp, gi = mt.distances()
    assert np.abs(gp[rows].cpu().numpy() - dp).max() < 1e-5 and np.abs(gi[rows].cpu().numpy() - di).max() < 1e-5
    oi, osc = topk_rows(2.0 * zscore_rows(dp) + zscore_rows(di), rows, 0, 2)
    assert np.array_equal(idx[rows], oi) and (np.abs(sc[rows] - osc) <= helpers.score_tol(osc, helpers.row_sigmas(dp, di))).all()
    mt.close()


# ------------------------------------------------------------------------------------------------ configs 4 and 5
def _sharded_run(make_matcher, shards, pack, phase1_args, mask, k):
    """What G ranks compute, one shard after the other on this GPU, with the production protocol (matcher.sharded_topk):
    phase 1 everywhere, moments stacked in rank order; fp32 selection everywhere with GLOBAL row offsets; the lists merged
    into the global top-(k+8); every shard re-evaluates the candidates it owns; owner-wise finish.  Also returns what
    every rank would answer alone (its own top-k under the global statistics)."""
    ms, moms = [], []
    for (lo, hi) in shards:
        mt = make_matcher(hi - lo)
        pack(mt, lo, hi)
        moms.append(mt.local_phase1(*phase1_args).clone())
        ms.append(mt)
    mom_all = torch.stack(moms)
    G = len(shards)
    sel = [mt.local_select(mom_all, G, mask, 2.0, k, lo, 0) for mt, (lo, hi) in zip(ms, shards)]
    sel = [(a.clone(), b.clone()) for a, b in sel]
    kin = sel[0][0].shape[1]
    cand, cand_sc = ms[0].merge(torch.stack([p[0] for p in sel]), torch.stack([p[1] for p in sel]), kin)
    part_all = torch.stack([mt.local_rerank(cand, k, True).clone() for mt in ms])
    # exactly one owner per candidate (masked pairs: +Inf everywhere); PR_SC_ARITH_F16: [G, m, 5, kin] - scores + the four channel parts
    assert (torch.isnan(part_all).sum(0) == G - 1).all() or mask > 0 or part_all.dim() == 4
    assert part_all.dim() == 3 or (torch.isnan(part_all[:, :, 0]).sum(0) == G - 1).all() or mask > 0
    idx, sc = ms[0].finish(cand, cand_sc, part_all, k)
    per = [tuple(t.clone() for t in mt.local_rerank(a, k, False)) for mt, (a, b) in zip(ms, sel)]
    return ms, per, idx.cpu().numpy(), sc.cpu().numpy()


def test_config4_200k_db_in_8_shards(api):
    from so_dso_place_recognition_amd.matcher import Matcher
    n, G, m, k, mask = 200_000, 8, 64, 3, 0
    shards = [(n * g // G, n * (g + 1) // G) for g in range(G)]
    chunks = {}

    def pack(mt, lo, hi):
        chunks[lo] = synth.sc_database_torch(45, hi - lo, first=lo)
        mt.pack_database(chunks[lo])

    # queries planted on the global DB (entries outside the handed shard are re-drawn from the sampler)
    q_h, planted = synth.sc_queries(46, np.empty((0, 2400)), m, db_first=0, n_global=n, db_seed=45)
    q = torch.from_numpy(q_h).cuda()
    ms, per, idx, sc = _sharded_run(lambda cap: Matcher("sc", m, cap), shards, pack, (q,), mask, k)
    assert np.array_equal(idx[:, 0], planted)
    rows = np.arange(16)
    dp, di = oracle_rows("sc", q_h[rows], [chunks[lo].cpu().numpy() for lo, hi in shards])
    f = 2.0 * zscore_rows(dp) + zscore_rows(di)                      # GLOBAL row statistics (run_test.m:40)
    oi, osc = topk_rows(f, rows, mask, k)
    assert np.array_equal(idx[rows], oi) and (np.abs(sc[rows] - osc) <= helpers.score_tol(osc)).all()
    for g in (0, 5, 7):                                              # rank g's own answer: the best of ITS rows under the global statistics
        lo, hi = shards[g]
        si, ss = topk_rows(f[:, lo:hi], rows, mask, k, lo)
        assert np.array_equal(per[g][0].cpu().numpy()[rows], si) and (np.abs(per[g][1].cpu().numpy()[rows] - ss) <= helpers.score_tol(ss)).all()
        gp, gi = ms[g].distances()
        assert np.abs(gp[rows].cpu().numpy() - dp[:, lo:hi]).max() < 1e-5 and np.abs(gi[rows].cpu().numpy() - di[:, lo:hi]).max() < 1e-5
    for mt in ms:
        mt.close()


def test_config5_fused_1m_db_in_8_shards(api):
    from so_dso_place_recognition_amd.matcher import FusedMatcher
    n, G, m, k, mask = 1_000_000, 8, 64, 2, 0
    shards = [(n * g // G, n * (g + 1) // G) for g in range(G)]
    R = 16
    rows = np.arange(R)
    q_sc, planted = synth.sc_queries(72, np.empty((0, 2400)), m, db_first=0, n_global=n, db_seed=71)
    # M2DP queries planted on the SAME entries: rows of m2dp_database(73) + noise, as synth.m2dp_queries does
    ent = synth.m2dp_database(73, 1, first=0)                         # shape probe
    rows_m2 = np.concatenate([synth.m2dp_database(73, 1, first=int(e)) for e in planted]).reshape(m, 4, 2, 192)
    u = synth.uniform(74, np.arange(m, dtype=np.uint64), 1 + 4 * 384)
    rows_m2 = rows_m2 + 0.05 * (u[:, 1:].reshape(m, 4, 2, 192) - 0.5)
    uu = rows_m2[..., :64] / np.sqrt((rows_m2[..., :64] ** 2).sum(-1, keepdims=True))
    vv = rows_m2[..., 64:] / np.sqrt((rows_m2[..., 64:] ** 2).sum(-1, keepdims=True))
    q_m2 = np.concatenate([uu, vv], -1).reshape(4 * m, 384)
    assert ent.shape == (4, 384)
    tq_sc, tq_m2 = torch.from_numpy(q_sc).cuda(), torch.from_numpy(q_m2).cuda()
    d = {c: [] for c in range(4)}                                     # oracle rows per channel, filled shard by shard

    def pack(mt, lo, hi):
        a = synth.sc_database_torch(71, hi - lo, first=lo)
        b = synth.m2dp_database_torch(73, hi - lo, first=lo)
        mt.pack_database(a, b)
        dp, di = oracle_rows("sc", q_sc[rows], [a.cpu().numpy()])
        ep, ei = oracle_rows("m2dp", q_m2.reshape(m, 4, 384)[rows].reshape(-1, 384), [b.cpu().numpy()])
        for c, x in enumerate((dp, di, ep, ei)):
            d[c].append(x)

    ms, per, idx, sc = _sharded_run(lambda cap: FusedMatcher(m, cap), shards, pack, (tq_sc, tq_m2), mask, k)
    assert np.array_equal(idx[:, 0], planted)
    full = [np.concatenate(d[c], 1) for c in range(4)]
    f = 2.0 * zscore_rows(full[0]) + zscore_rows(full[1]) + 2.0 * zscore_rows(full[2]) + zscore_rows(full[3])
    oi, osc = topk_rows(f, rows, mask, k)
    assert np.array_equal(idx[rows], oi) and (np.abs(sc[rows] - osc) <= 2 * helpers.score_tol(osc)).all()    # two descriptor types, each with its own row statistics
    g = 3
    lo, hi = shards[g]
    si, ss = topk_rows(f[:, lo:hi], rows, mask, k, lo)
    assert np.array_equal(per[g][0].cpu().numpy()[rows], si) and (np.abs(per[g][1].cpu().numpy()[rows] - ss) <= 2 * helpers.score_tol(ss)).all()
    for mt in ms:
        mt.close()


# ------------------------------------------------------------------------------------------------ near-ties
def test_near_ties_are_ordered_by_the_fp64_reevaluation(api):
    """Pairs of DB entries whose distances to a query differ by 1e-9 ... 1e-5 - below what the fp32 all-pairs pass resolves
    (its error is ~1e-7): the returned order must still be the oracle's."""
    n, m = 1500, 240
    db = synth.sc_database(45, n)
    q, planted = synth.sc_queries(46, db, m)
    rng = np.random.default_rng(12)
    twins = np.setdiff1d(np.arange(n), planted)[-m:]                 # entry twins[t] becomes a near-copy of query t's planted entry
    for t in range(m):
        delta = 10.0 ** rng.uniform(-8.5, -4.5)
        e = db[planted[t]].copy()
        occ = np.nonzero(e[:1200] > 0)[0]
        pick = rng.choice(occ, size=40, replace=False)
        e[pick] *= 1.0 + delta * rng.standard_normal(40) * 50
        db[twins[t]] = e
    rc, oidx, osc = oracle_lib.match_topk(0, q, db, 0, 2.0, 3)
    gap = np.abs(osc[:, 0] - osc[:, 1])
    assert (gap < 1e-3).sum() > m // 2 and (gap > 0).all()          # the test has teeth: most pairs are closer than fp32 noise / sigma
    idx, sc = api.match_topk("sc", q, db, 0, 2.0, 3)
    assert np.array_equal(idx, oidx)
    assert (np.abs(sc - osc) <= helpers.score_tol(osc)).all()
    # the same through the device-resident path on two shards
    from so_dso_place_recognition_amd.matcher import Matcher
    tq = torch.from_numpy(q).cuda()
    cut = 700
    ms, per, i2, s2 = _sharded_run(lambda cap: Matcher("sc", m, cap), [(0, cut), (cut, n)],
                                   lambda mt, lo, hi: mt.pack_database(torch.from_numpy(db[lo:hi]).cuda()), (tq,), 0, 3)
    assert np.array_equal(i2, oidx) and (np.abs(s2 - osc) <= helpers.score_tol(osc)).all()
    for mt in ms:
        mt.close()


def test_pruned_reevaluation_gives_the_unpruned_result(api):
    """pr_rerank_dev with the candidates' fp32 scores skips candidates that cannot reach the top-k (include/place_recognition.h): the
    result must be bit for bit the one without them - planted matches (eight of nine candidates hopeless), plain random rows
    (candidates ~0.05 apart) and families of near-copies (all candidates inside the margin, nothing may be skipped)."""
    from so_dso_place_recognition_amd.matcher import Matcher
    n, m = 3000, 96
    db = synth.sc_database(45, n)
    q, planted = synth.sc_queries(46, db, m)
    rng = np.random.default_rng(3)
    q[m // 3: 2 * m // 3] = synth.sc_database(77, m // 3)              # a third of the queries have no planted match
    for t in range(2 * m // 3, m):                                     # a third have twelve near-copies of their match in the DB
        for c in range(12):
            e = db[planted[t]].copy()
            occ = np.nonzero(e[:1200] > 0)[0]
            e[rng.choice(occ, size=30, replace=False)] *= 1.0 + 1e-6 * rng.standard_normal(30)
            db[n - 1 - ((t - 2 * m // 3) * 12 + c)] = e               # disjoint slots at the end of the DB
    mt = Matcher("sc", m, n)
    mt.pack_database(torch.from_numpy(db).cuda())
    for k in (1, 3):
        mom = mt.local_phase1(torch.from_numpy(q).cuda())
        idx_in, sc = mt.local_select(mom.unsqueeze(0), 1, 0, 2.0, k, 0, 0)
        a = tuple(x.clone() for x in mt.local_rerank(idx_in, k, False))
        b = tuple(x.clone() for x in mt.local_rerank(idx_in, k, False, sc))
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        pa = mt.local_rerank(idx_in, k, True).clone()                   # the shard's p5 block [m, 5, kin]: scores | four exact channel distances
        pb = mt.local_rerank(idx_in, k, True, sc).clone()
        skipped = (pa[:, 0] != pb[:, 0]).sum().item()
        assert skipped > m * 4                                          # it does skip: most candidates of the planted rows
        assert torch.isnan(pb[:, 1][pa[:, 0] != pb[:, 0]]).all()        # ... and says so: no distances for a pruned candidate
        assert torch.equal(pa[:, :, :k], pb[:, :, :k])
        near = torch.arange(2 * m // 3, m, device="cuda")
        near = near[sc[near, k + 7] - sc[near, 0] < 1e-3]               # rows whose candidates are all copies of the match (a few lost theirs to the copy slots)
        assert len(near) > m // 6 and torch.equal(pa[near], pb[near])   # everything inside the margin is evaluated
    # (no oracle here: families of 13 near-copies are more near-ties than the k + 8 candidates of the fp32 pass can hold; what this test
    #  pins is pruned == unpruned - the oracle comparisons of the near-tie and full-size tests run through the pruned path anyway)
    mt.close()


# ------------------------------------------------------------------------------------------------ a drive
def _drive(frames=2000, per_cloud=6000, seed=5):
    return synth.drive_clouds_torch(frames, per_cloud, seed)


def test_drive_2000_frames_mask_100(api):
    """Consecutive-frame clouds (neighbouring signatures nearly equal, a plateau of near-ties around every true match),
    mask_width 100 as test_kitti.m:19: generation parity on every cloud, and the matcher's top-2 + scores against the
    oracle on 200 rows spread over both laps; the second lap must find the first."""
    xyz, it, offs, lap = _drive()
    N = len(offs) - 1
    assert N == 2000 and (np.diff(offs) > 2000).all()
    g = api.sc_generate(xyz, it, offs)
    o = oracle_lib.sc_generate(xyz, it, offs)
    assert np.array_equal(g[:, 1200:], o[:, 1200:]) and np.abs(g - o).max() < 1e-10
    idx, sc = api.match_topk("sc", g, g, 100, 2.0, 2)
    rows = np.unique(np.concatenate([np.arange(0, N, 11), np.arange(lap - 3, lap + 3)]))
    dp, di = oracle_rows("sc", o[rows], [o])
    oi, osc = topk_rows(2.0 * zscore_rows(dp) + zscore_rows(di), rows, 100, 2)
    assert np.array_equal(idx[rows], oi)
    assert (np.abs(sc[rows] - osc) <= helpers.score_tol(osc, helpers.row_sigmas(dp, di))).all()
    flat = np.abs(osc) < 30                                                      # every compared score of a real sequence: flat 1e-5
    assert flat.sum() >= 100
    assert (np.abs(sc[rows] - osc)[flat] < 1e-5).all(), np.abs(sc[rows] - osc)[flat].max()
    second = np.arange(lap + 50, N - 50)
    assert (np.abs(idx[second, 0] - (second - lap)) <= 3).mean() > 0.7           # loop closures onto the first lap


@pytest.mark.gpu
def test_order_that_hangs_on_the_row_sigmas_is_resolved_with_fp64_statistics(api):
    """Found by tools/fuzz_all.py (seed 1, `match,matcher,fused`, case 1): M2DP, 96 x 1096, k = 35.  Ranks 6 / 7 of one row differ by less
    than what the fp32 pass's sigma error (2.5e-7 relative) moves them, and their two channels disagree about the order: every arithmetic
    returned them swapped.  The order check flags such queries and they are answered with fp64 row statistics on EVERY path - host calls
    (pr_order_resolve_dev), the device-resident Matcher by default (pr_order_resolve_async_dev, stream-ordered), pr_group and the
    torch.distributed Matcher (pr_order_exact_moments_dev, pr_order_exact_select_dev, pr_order_exact_merge_dev around two all-gathers) -
    indices AND scores of that row are then fp64 throughout."""
    import torch
    from so_dso_place_recognition_amd import _lib
    from so_dso_place_recognition_amd.matcher import Matcher
    m, n, k = 96, 1096, 35
    db = synth.m2dp_database(3000 + 7 * 1 + 1, n)
    q, _ = synth.m2dp_queries(4000 + 1 + 1, db, m)
    rc, oidx, osc = oracle_lib.match_topk(1, q, db, 0, 2.0, k)
    assert rc == 0
    for arith in ("f16x2", "f32", "f16"):
        ctx = api.Context(0, sc_arith=arith)
        idx, sc = api.match_topk("m2dp", q, db, 0, 2.0, k, ctx=ctx)
        assert ctx.take_warnings() & _lib.WARN_ORDER_RESOLVED, arith
        assert np.array_equal(idx, oidx), arith
        ctx.close()
    dev = torch.device("cuda", 0)
    mt = Matcher("m2dp", m, n, ctx=api.Context(0, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
    mt.pack_database(torch.from_numpy(db).to(dev))
    i0, s0 = mt.match(torch.from_numpy(q).to(dev), 0, 2.0, k, exact_order=False)   # the order of the fp32-statistics scores
    i0 = i0.cpu().numpy().copy()
    assert not (mt.take_warnings() & _lib.WARN_ORDER_RESOLVED)
    i1, s1 = mt.match(torch.from_numpy(q).to(dev), 0, 2.0, k)                      # the default: stream-ordered fp64-statistics resolution
    assert mt.take_warnings() & _lib.WARN_ORDER_RESOLVED
    assert np.array_equal(i1.cpu().numpy(), oidx) and not np.array_equal(i0, oidx)
    rows = np.flatnonzero((i0 != oidx).any(axis=1))                 # the resolved rows: scores exact to fp64 rounding, not to the fp32 model
    assert len(rows) >= 1 and np.abs(s1.cpu().numpy()[rows] - osc[rows]).max() < 1e-9
    mt.close()
    # the sharded protocol resolves too: pr_group with 2 / 3 / 8 virtual shards on this GPU ...
    for G in (2, 3, 8):
        g = api.Group([0] * G)
        g.set_database("m2dp", db)
        gi, gs = g.match_topk(q, 0, 2.0, k)
        assert g.take_warnings() & _lib.WARN_ORDER_RESOLVED, G
        assert np.array_equal(gi, oidx), G
        assert np.abs(gs[rows] - osc[rows]).max() < 1e-9, G
        g.close()
    # ... and the torch.distributed Matcher with two ranks (gloo) on this GPU
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(root, "tests", "dist_order_case.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["world"] == 2 and d["warnings"] & _lib.WARN_ORDER_RESOLVED
    assert np.array_equal(np.array(d["idx"]), oidx)
    assert np.abs(np.array(d["score"])[rows] - osc[rows]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("type_", ["sc", "m2dp"])
def test_exact_rows_from_float32_signatures(api, type_, monkeypatch):
    """The exact-row kernels read the signatures in the dtype the caller holds them in: float32 tensors (half the HBM of a resident DB) must
    give the oracle's answer for the same rounded numbers, every query through the exact rows, spectral form and direct form alike."""
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher
    m, n, k = 40, 900, 4
    db = (synth.sc_database(61, n) if type_ == "sc" else synth.m2dp_database(61, n)).astype(np.float32)
    q, _ = (synth.sc_queries if type_ == "sc" else synth.m2dp_queries)(62, db.astype(np.float64), m)
    q = q.astype(np.float32)
    rc, oidx, osc = oracle_lib.match_topk(0 if type_ == "sc" else 1, q.astype(np.float64), db.astype(np.float64), 3, 2.0, k)
    dev = torch.device("cuda", 0)
    for form in ("spectral", "direct"):
        if form == "direct":
            monkeypatch.setenv("PR_XROW", "direct")
        mt = Matcher(type_, m, n, ctx=api.Context(0, exact_statistics=True, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
        mt.pack_database(torch.from_numpy(db).to(dev))
        idx, sc = mt.match(torch.from_numpy(q).to(dev), 3, 2.0, k)
        assert np.array_equal(idx.cpu().numpy(), oidx), form
        assert np.abs(sc.cpu().numpy() - osc).max() < 1e-9, form
        mt.close()


@pytest.mark.gpu
def test_group_selftest_and_phase_timing(api):
    """pr_group_create runs every exchange of a call on scratch buffers and checks the rank order of what arrives (here: the copy exchange of
    virtual shards; RCCL on distinct devices); pr_group_set_timing / pr_group_last_timing give every shard's phase times of a call."""
    n, m = 3000, 40
    db = synth.sc_database(45, n)
    q, planted = synth.sc_queries(46, db, m)
    g = api.Group([0, 0, 0])
    g.set_database("sc", db)
    g.set_timing(True)
    idx, sc = g.match_topk(q, 0, 2.0, 1)
    assert np.array_equal(idx[:, 0], planted)
    t = g.last_timing()
    assert len(t) == 3 and all(set(x) == set(api.Group.PHASES) for x in t)
    assert all(v >= 0 for x in t for v in x.values()) and all(x["distances"] > 0 for x in t)
    g.set_timing(False)
    g.match_topk(q, 0, 2.0, 1)
    with pytest.raises(api.PRError):
        g.last_timing()                                        # no timed call since the switch went off
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("type_", ["sc", "m2dp"])
def test_near_copy_clusters_are_answered_from_the_exact_row(api, type_):
    """run_test.m:57 takes the minimum over the WHOLE row; the fp64 re-evaluation sees the k + 8 best of an fp32-grade pass.  Forty
    near-copies of a query's planted entry whose distances differ by 1e-10 ... 1e-7 (helpers.near_copy_clusters) tie in that pass: the
    exact best need not be among the 9 it hands on.  The containment check flags such a query and it is answered from its exact fp64 row -
    on every path: host calls (every arithmetic), the device-resident Matcher (stream-ordered), pr_group with three virtual shards, the
    torch.distributed Matcher with two ranks - with the oracle's indices for k = 1 and k = 5."""
    import json
    import subprocess
    import sys
    import torch
    from so_dso_place_recognition_amd import _lib
    from so_dso_place_recognition_amd.matcher import Matcher
    n, m, rows, copies = helpers.CLUSTER_CASE
    db, q, planted, members = helpers.near_copy_clusters(type_, n, m, rows, copies)
    t_id = 0 if type_ == "sc" else 1
    rc, oidx, osc = oracle_lib.match_topk(t_id, q, db, 0, 2.0, 5)
    assert rc == 0
    for t in rows:                                                    # the test has teeth: the oracle's five best are cluster members, a
        assert set(oidx[t]) <= set(members[t].tolist())              # hair's breadth apart - and distinct
        gaps = np.diff(osc[t])
        assert (gaps > 0).all() and gaps.max() < 1e-4
    clustered = set(int(planted[t]) for t in rows)                   # (two queries may be planted on the same entry)
    others = np.array([t for t in range(m) if int(planted[t]) not in clustered])
    dev = torch.device("cuda", 0)
    for k in (1, 5):
        for arith in ("f16x2", "f32", "f16"):
            ctx = api.Context(0, sc_arith=arith)
            idx, sc = api.match_topk(type_, q, db, 0, 2.0, k, ctx=ctx)
            w = ctx.take_warnings()
            assert np.array_equal(idx, oidx[:, :k]), (k, arith)
            if arith != "f16":                                                             # exact rows: fp64 throughout
                assert np.abs(sc[list(rows)] - osc[list(rows), :k]).max() < 1e-9, (k, arith)
                assert w & _lib.WARN_ORDER_RESOLVED, (k, arith)
            else:     # (its k + 56 candidates hold a cluster of 41: the margin check passes, the scores carry the f16 pass's row statistics)
                assert (np.abs(sc - osc[:, :k]) <= 3e-2 + 1e-3 * np.abs(osc[:, :k])).all(), (k, arith)
            ctx.close()
        mt = Matcher(type_, m, n, ctx=api.Context(0, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
        mt.pack_database(torch.from_numpy(db).to(dev))
        i0, s0 = mt.match(torch.from_numpy(q).to(dev), 0, 2.0, k, exact_order=False)       # what the k + 8 candidates alone give
        i0 = i0.cpu().numpy().copy()
        i1, s1 = mt.match(torch.from_numpy(q).to(dev), 0, 2.0, k)
        assert mt.take_warnings() & _lib.WARN_ORDER_RESOLVED
        i1, s1 = i1.cpu().numpy(), s1.cpu().numpy()
        assert np.array_equal(i1, oidx[:, :k]) and np.abs(s1[list(rows)] - osc[list(rows), :k]).max() < 1e-9
        if k == 1:                                                                      # (deeper lists of OTHER queries may hold members of a cluster too)
            assert np.array_equal(i0[others], oidx[others, :k])
            assert (i0[list(rows), 0] != oidx[list(rows), 0]).sum() >= 3               # ... is wrong for clusters larger than the list
        mt.close()
        g = api.Group([0, 0, 0])
        g.set_database(type_, db)
        gi, gs = g.match_topk(q, 0, 2.0, k)
        assert g.last_flagged >= len(rows) and (g.take_warnings() & _lib.WARN_ORDER_RESOLVED)
        assert np.array_equal(gi, oidx[:, :k]) and np.abs(gs[list(rows)] - osc[list(rows), :k]).max() < 1e-9
        g.close()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + (os.getpid() + 7 * k + t_id) % 90), os.path.join(root, "tests", "dist_order_case.py"), "f16x2",
               "cluster_" + type_, str(k)]
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert d["world"] == 2 and d["warnings"] & _lib.WARN_ORDER_RESOLVED
        assert np.array_equal(np.array(d["idx"]), oidx[:, :k])
        assert np.abs(np.array(d["score"])[list(rows)] - osc[list(rows), :k]).max() < 1e-9


@pytest.mark.gpu
def test_fp64_statistics_for_every_query_give_the_oracles_scores(api, monkeypatch):
    """PR_FORCE_ORDER_FLAGS=1 makes the order check flag every query, so the fp64-statistics resolution - normally a 1-in-10^5 path - answers
    all of them: every returned score must then be the oracle's to fp64 rounding (exact pair distances AND exact row statistics,
    run_test.m:38-57), on every path: the host calls (all queries: several passes of 64), the device-resident call (150 flagged queries: it
    reads the count back and runs three passes; with exact_order="async" the same three passes chained on the stream without a read-back -
    PR_WARN_ORDER_UNRESOLVED never appears; a C caller that stops after one pass of the sharded form and says so gets it), pr_group with virtual shards, two torch.distributed ranks (three passes of the exchange), and the
    fused SC + M2DP form."""
    import json
    import subprocess
    import sys
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher, FusedMatcher
    monkeypatch.setenv("PR_FORCE_ORDER_FLAGS", "1")
    m, n, k = 150, 700, 3
    db = synth.sc_database(81, n)
    q, _ = synth.sc_queries(181, db, m)
    rc, oidx, osc = oracle_lib.match_topk(0, q, db, 5, 2.0, k)
    assert rc == 0
    ctx = api.Context(0)
    idx, sc = api.match_topk("sc", q, db, 5, 2.0, k, ctx=ctx)
    assert ctx.take_warnings() & _lib.WARN_ORDER_RESOLVED
    assert np.array_equal(idx, oidx) and np.abs(sc - osc).max() < 1e-9
    ctx.close()
    monkeypatch.delenv("PR_FORCE_ORDER_FLAGS")                   # the same through the API switch (pr_set_exact_statistics) ...
    ctx = api.Context(0, exact_statistics=True)
    idx, sc = api.match_topk("sc", q[:20], db, 5, 2.0, k, ctx=ctx)
    assert np.array_equal(idx, oidx[:20]) and np.abs(sc - osc[:20]).max() < 1e-9
    ctx.close()
    ctx = api.Context(0)                                          # ... and the default: exact pair distances, fp32-pass row statistics
    idx, sc = api.match_topk("sc", q[:20], db, 5, 2.0, k, ctx=ctx)
    assert np.array_equal(idx, oidx[:20]) and np.abs(sc - osc[:20]).max() > 1e-9
    ctx.close()
    monkeypatch.setenv("PR_FORCE_ORDER_FLAGS", "1")
    dev = torch.device("cuda", 0)
    mt = Matcher("sc", m, n, ctx=api.Context(0, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
    mt.pack_database(torch.from_numpy(db).to(dev))
    i1, s1 = mt.match(torch.from_numpy(q).to(dev), 5, 2.0, k)                            # a call of more than 64 queries: every flagged one
    w = mt.take_warnings()
    assert (w & _lib.WARN_ORDER_RESOLVED) and not (w & _lib.WARN_ORDER_UNRESOLVED) and mt.resolved == m
    assert np.array_equal(i1.cpu().numpy(), oidx) and np.abs(s1.cpu().numpy() - osc).max() < 1e-9
    i1, s1 = mt.match(torch.from_numpy(q).to(dev), 5, 2.0, k, exact_order="async")
    w = mt.take_warnings()
    assert (w & _lib.WARN_ORDER_RESOLVED) and not (w & _lib.WARN_ORDER_UNRESOLVED)      # 150 flagged: three stream-ordered passes, no read-back
    i1, s1 = i1.cpu().numpy(), s1.cpu().numpy()
    assert np.array_equal(i1, oidx) and np.abs(s1 - osc).max() < 1e-9                    # every query: fp64 throughout
    # the sharded form, driven like a C caller that stops after ONE pass and declares it its last: queries 64.. keep the candidate list's
    # answer and the context says so (and a later complete call does not take the bit back before pr_take_warnings has reported it)
    qd = torch.from_numpy(q).to(dev)
    mom = mt.local_phase1(qd)
    ci, cs = mt.local_select(mom.unsqueeze(0), 1, 5, 2.0, k, 0, 0)
    part = mt.local_rerank(ci, k, True, cs)
    i3, s3 = mt.finish(ci, cs, part.unsqueeze(0), k)
    ex = mt.exact_moments(0, last=True)
    sel = mt.exact_select(ex.unsqueeze(0), k, 0)
    i3, s3 = mt.exact_merge(sel.unsqueeze(0), k, i3, s3, 0)
    i3, s3 = i3.cpu().numpy().copy(), s3.cpu().numpy().copy()
    i2, s2 = mt.match(qd[:40].contiguous(), 5, 2.0, k)                                   # (a complete call in between)
    w = mt.take_warnings()
    assert (w & _lib.WARN_ORDER_RESOLVED) and (w & _lib.WARN_ORDER_UNRESOLVED)
    assert np.array_equal(i3, oidx) and np.abs(s3[:64] - osc[:64]).max() < 1e-9          # the resolved ones: fp64 throughout
    rc, odp, odi = oracle_lib.sc_distance(q[64:], db)
    assert (np.abs(s3[64:] - osc[64:]) <= helpers.score_tol(osc[64:], helpers.row_sigmas(odp, odi))).all()   # the others: the fp32-statistics model
    i2, s2 = mt.match(torch.from_numpy(q[:40]).to(dev), 5, 2.0, k)                       # an online-sized call: everything resolved
    assert not (mt.take_warnings() & _lib.WARN_ORDER_UNRESOLVED)
    assert np.array_equal(i2.cpu().numpy(), oidx[:40]) and np.abs(s2.cpu().numpy() - osc[:40]).max() < 1e-9
    mt.close()
    g = api.Group([0, 0, 0])
    g.set_database("sc", db)
    gi, gs = g.match_topk(q[:60], 5, 2.0, k)
    assert np.array_equal(gi, oidx[:60]) and np.abs(gs - osc[:60]).max() < 1e-9
    g.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))               # two ranks: 150 flagged queries = three passes of all-gathers D, E
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + os.getpid() % 90), os.path.join(root, "tests", "dist_order_case.py"), "f16x2", "forced", str(k)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["world"] == 2 and (d["warnings"] & _lib.WARN_ORDER_RESOLVED) and not (d["warnings"] & _lib.WARN_ORDER_UNRESOLVED)
    assert np.array_equal(np.array(d["idx"]), oidx) and np.abs(np.array(d["score"]) - osc).max() < 1e-9
    # fused SC + M2DP (config 5's score): both descriptor types through the resolution
    mdb = synth.m2dp_database(83, n)
    mq, _ = synth.m2dp_queries(183, mdb, 48)
    rc, fidx, fsc = oracle_lib.match_topk_fused(q[:48], mq, db, mdb, 5, 2.0, k)
    assert rc == 0
    fm = FusedMatcher(48, n, ctx=api.Context(0, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
    fm.pack_database(torch.from_numpy(db).to(dev), torch.from_numpy(mdb).to(dev))
    i3, s3 = fm.match(torch.from_numpy(q[:48]).to(dev), torch.from_numpy(mq).to(dev), 5, 2.0, k)
    assert np.array_equal(i3.cpu().numpy(), fidx) and np.abs(s3.cpu().numpy() - fsc).max() < 1e-9
    fm.close()


def test_stream_ordered_resolution_chains_every_pass_of_a_large_call(api, monkeypatch):
    """pr_order_resolve_async_dev above RESOLVE_SMALL_M = 1024 queries: the first pass compacts the flags into a list, the passes behind it
    read the list; every query flagged (PR_FORCE_ORDER_FLAGS) -> ceil(1100 / 64) = 18 passes chained on the stream without a read-back, all
    scores the oracle's doubles, no PR_WARN_ORDER_UNRESOLVED; also as a captured hipGraph (what such a call is for)."""
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher
    monkeypatch.setenv("PR_FORCE_ORDER_FLAGS", "1")
    m, n, k = 1100, 260, 2
    db = synth.sc_database(83, n)
    q = np.concatenate([synth.sc_queries(183 + i, db, 220)[0] for i in range(5)])
    rc, oidx, osc = oracle_lib.match_topk(0, q, db, 3, 2.0, k)
    assert rc == 0
    dev = torch.device("cuda", 0)
    mt = Matcher("sc", m, n, ctx=api.Context(0, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
    mt.pack_database(torch.from_numpy(db).to(dev))
    i1, s1 = mt.match(torch.from_numpy(q).to(dev), 3, 2.0, k, exact_order="async")
    w = mt.take_warnings()
    assert (w & _lib.WARN_ORDER_RESOLVED) and not (w & _lib.WARN_ORDER_UNRESOLVED)
    assert np.array_equal(i1.cpu().numpy(), oidx) and np.abs(s1.cpu().numpy() - osc).max() < 1e-9
    mt.close()
    mg = Matcher.on_new_stream("sc", m, n, device=0)
    with torch.cuda.stream(mg.stream):
        mg.pack_database(torch.from_numpy(db).to(dev))
    cap = mg.capture(torch.from_numpy(q).to(dev), 3, 2.0, k)
    cap.run(); cap.run()
    w = mg.take_warnings()
    assert (w & _lib.WARN_ORDER_RESOLVED) and not (w & _lib.WARN_ORDER_UNRESOLVED)
    assert np.array_equal(cap.idx.cpu().numpy(), oidx) and np.abs(cap.score.cpu().numpy() - osc).max() < 1e-9
    mg.close()
