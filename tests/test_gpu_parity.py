"""GPU parity tests proper: the HIP path through the C ABI vs the CPU oracle on the same seeded inputs.
Tolerances: top-k indices bit-exact; fp32 distances within 1e-5 (BASELINE.json north_star); generated signatures:
bin occupancy / binary channel exact, fp64 structure within 1e-10, M2DP singular vectors within 1e-9."""
import os

import numpy as np
import pytest

import helpers
import oracle_lib
from so_dso_place_recognition_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from so_dso_place_recognition_amd import api as _api
    return _api


# ------------------------------------------------------------------------------------------------ a6
@pytest.mark.parametrize("m,n", [(1, 2), (8, 16), (37, 101), (64, 333)])
def test_sc_distance_vs_oracle(api, m, n):
    db = synth.sc_database(45, n)
    q, _ = synth.sc_queries(46, db, m)
    rc, op, oi = oracle_lib.sc_distance(q, db)
    gp, gi = api.processSC(q, db)
    assert gp.shape == (m, n) and gp.dtype == np.float32
    assert np.abs(gp - op).max() < 1e-5 and np.abs(gi - oi).max() < 1e-5


def test_sc_distance_golden_and_invariants(api, golden_dir):
    g = np.load(os.path.join(golden_dir, "synthetic_v1.npz"))
    db = synth.sc_database(45, 16)
    q, _ = synth.sc_queries(46, db, 8)
    gp, gi = api.processSC(q, db)
    assert np.abs(gp - g["sc_dp"]).max() < 1e-5 and np.abs(gi - g["sc_di"]).max() < 1e-5
    sp, si = api.processSC(db, db)
    assert np.abs(np.diag(sp)).max() < 1e-5 and np.abs(np.diag(si)).max() < 1e-5       # self distance 0
    rot = np.roll(db.reshape(16, 2, 60, 20)[:, :, ::-1], 11, axis=2).reshape(16, 2400)   # mirror + rotate
    rp, ri = api.processSC(rot, db)
    assert np.abs(rp - sp).max() < 2e-6 and np.abs(ri - si).max() < 2e-6


def test_sc_zero_row_is_excluded_like_matlab(api):
    """A zero-norm row is 0/0 = NaN in MATLAB (processSC.m:16,19): NaN distances in its row / column, which normalize(.,2)
    and min (run_test.m:40,57) leave out - the signature never matches and nothing else changes.  Default policy = that
    (with a warning bit); nan_policy="fail" turns it into PR_ENAN."""
    db = synth.sc_database(45, 300)
    db[3, :1200] = 0                                        # structure channel of DB entry 3
    db[7, 1200:] = 0                                        # intensity channel of DB entry 7
    q, _ = synth.sc_queries(46, db, 20)
    q[5, :] = 0                                             # a wholly empty query
    ctx = api.Context(0)
    gp, gi = api.processSC(q, db, ctx)
    rc, op, oi = oracle_lib.sc_distance(q, db)
    assert rc != 0                                          # the oracle reports it too
    assert np.array_equal(np.isnan(gp), np.isnan(op)) and np.array_equal(np.isnan(gi), np.isnan(oi))
    assert np.isnan(gp[:, 3]).all() and np.isnan(gi[:, 7]).all() and np.isnan(gp[5]).all() and not np.isnan(gp[0, 7])
    ok = ~np.isnan(op)
    assert np.abs(gp[ok] - op[ok]).max() < 1e-5
    assert ctx.take_warnings() & 1 and ctx.take_warnings() == 0
    idx, sc = api.match_topk("sc", q, db, 0, 2.0, 3, ctx=ctx)
    rc, oidx, osc = oracle_lib.match_topk(0, q, db, 0, 2.0, 3)
    assert np.array_equal(idx, oidx)
    assert (idx[5] == -1).all() and np.isnan(sc[5]).all() and not np.isin(idx, [3, 7]).any()
    live = idx >= 0
    assert (np.abs(sc[live] - osc[live]) <= helpers.score_tol(osc[live])).all()
    ctx.close()
    strict = api.Context(0, nan_policy="fail")
    with pytest.raises(api.PRError) as e:
        api.processSC(db[:2], db, strict)
    assert e.value.code == -5
    gp, _ = api.processSC(db[:2], db[8:], strict)           # the context stays usable afterwards
    assert np.isfinite(gp).all()
    strict.close()


def test_fuse_select_rows_not_coaligned(api):
    """d_p and d_i carved out of ONE allocation with m*n odd: the rows of the two matrices are not co-aligned mod 16 B, the
    vector body cannot be used and the scalar loop has to cover the whole row (it used to stop after 256 columns)."""
    import ctypes as C
    import torch
    m, n, k = 3, 1001, 4
    rng = np.random.default_rng(3)
    dp = rng.random((m, n)).astype(np.float32); di = rng.random((m, n)).astype(np.float32)
    for r in range(m):
        dp[r, 300 + 200 * r] = -1.0; di[r, 300 + 200 * r] = -1.0          # the winners sit far beyond column 256
    for off in (0, 1):                                                      # both alignments of the block start
        buf = torch.zeros(2 * m * n + 8, dtype=torch.float32, device="cuda")
        a = buf[off:off + m * n]; b = buf[off + m * n:off + 2 * m * n]
        a.copy_(torch.from_numpy(dp.ravel())); b.copy_(torch.from_numpy(di.ravel()))
        ctx = api.Context(0)
        mom = torch.empty((m, 2, 3), dtype=torch.float64, device="cuda")
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); sc = torch.empty((m, k), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        P = lambda t: C.c_void_p(t.data_ptr())
        ctx.check(ctx.lib.pr_row_moments_dev(ctx.h, P(a), P(b), m, n, P(mom)))
        ctx.check(ctx.lib.pr_fuse_select_dev(ctx.h, P(a), P(b), m, n, P(mom), 1, 0, 0, 0, 2.0, k, P(idx), P(sc)))
        ctx.sync()
        o = oracle_lib.fuse_topk(dp.astype(np.float64), di.astype(np.float64), 0, 2.0, k)
        assert np.array_equal(idx.cpu().numpy(), o[0]) and np.abs(sc.cpu().numpy() - o[1]).max() < 1e-4   # fp32 scores of the selection pass
        assert list(idx.cpu().numpy()[:, 0]) == [300, 500, 700]
        ctx.close()


@pytest.mark.parametrize("m,n,k", [(2, 100_000, 57), (1, 100_000, 128), (2, 100_000, 9), (40, 30_000, 57), (3, 5000, 100), (5, 300, 20), (4, 3000, 3),
                                   (2, 2000, 300)])
@pytest.mark.parametrize("flavour", ["distinct", "ties", "mass_ties"])
def test_selection_of_many_candidates_by_radix_select(api, m, n, k, flavour):
    """k > 12: the k best (value, index) pairs of a row come from a radix selection over the LDS list + a 256-entry sort (list_topk) - in the
    row kernel, in the slice kernels of few-row calls (grid m x P) and in their merge.  Plain selection (no fusion) against numpy's stable
    order: distinct values, ties that the index has to break, more than 256 equal values around rank k (the whole-list sort fallback), NaN
    (never wins), +Inf, both zeros."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(m * 1000 + k)
    d = rng.normal(0, 1, (m, n)).astype(np.float32)
    if flavour == "ties":
        d = np.round(d * 200) / 200                                  # ~1400 distinct values, dozens of equal entries each
        d[:, ::7][d[:, ::7] == 0] = -0.0
    if flavour == "mass_ties":
        d = np.round(d * 2) / 2                                      # a dozen distinct values: thousands tie at the k-th
    d[:, 5] = np.nan; d[:, n // 2] = np.inf; d[0, 11] = -np.inf
    if m > 1:
        d[1, : n - 3] = np.nan                                       # a row with fewer candidates than k
    ctx = api.Context(0)
    a = torch.from_numpy(d).cuda()
    idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); sc = torch.empty((m, k), dtype=torch.float32, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    ctx.check(ctx.lib.pr_fuse_select_dev(ctx.h, P(a), None, m, n, None, 1, 0, 0, 0, 2.0, k, P(idx), P(sc)))
    ctx.sync()
    got_i, got_s = idx.cpu().numpy(), sc.cpu().numpy()
    for r in range(m):
        ok = np.nonzero(~np.isnan(d[r]))[0]
        order = ok[np.argsort(d[r, ok], kind="stable")][:k]          # ties -> the lower index (MATLAB min, run_test.m:57)
        want = np.full(k, -1); want[: len(order)] = order
        assert np.array_equal(got_i[r], want), (r, got_i[r][:8], want[:8])
        assert np.array_equal(got_s[r, : len(order)], d[r, order]) and np.isnan(got_s[r, len(order):]).all()
    ctx.close()


# ------------------------------------------------------------------------------------------------ a7
@pytest.mark.parametrize("m,n", [(1, 2), (9, 40), (33, 130)])
def test_m2dp_distance_vs_oracle(api, m, n):
    db = synth.m2dp_database(43, n)
    q, _ = synth.m2dp_queries(44, db, m)
    rc, op, oi = oracle_lib.m2dp_distance(q, db)
    gp, gi = api.processM2DP(q, db)
    assert np.abs(gp - op).max() < 1e-5 and np.abs(gi - oi).max() < 1e-5
    sp, _ = api.processM2DP(db, db)
    assert np.abs(np.diag(sp) + 0.5).max() < 1e-5                                       # self distance -0.5


# ------------------------------------------------------------------------------------------------ a8
@pytest.mark.parametrize("type_,mask,k", [("sc", 0, 1), ("sc", 5, 4), ("m2dp", 0, 1), ("m2dp", 3, 3)])
def test_match_topk_vs_oracle(api, type_, mask, k):
    if type_ == "sc":
        db = synth.sc_database(45, 150); q = db[:60]
        t = 0
    else:
        db = synth.m2dp_database(43, 150); q = db[:240]
        t = 1
    rc, oidx, osc = oracle_lib.match_topk(t, q, db, mask, 2.0, k)
    gidx, gsc = api.match_topk(type_, q, db, mask, 2.0, k)
    assert np.array_equal(gidx, oidx)                       # bit-exact indices
    rc, odp, odi = oracle_lib.sc_distance(q, db) if type_ == "sc" else oracle_lib.m2dp_distance(q, db)
    assert gsc.dtype == np.float64 and (np.abs(gsc - osc) <= helpers.score_tol(osc, helpers.row_sigmas(odp, odi))).all()   # helpers.score_tol: the bound and where it comes from


def test_match_planted_and_ties(api, golden_dir):
    g = np.load(os.path.join(golden_dir, "synthetic_v1.npz"))
    db = synth.sc_database(45, 16); q, et = synth.sc_queries(46, db, 8)
    idx, sc = api.match_topk("sc", q, db)
    assert np.array_equal(idx[:, 0], g["sc_top1"]) and np.array_equal(idx[:, 0], et)
    db = synth.sc_database(45, 40)
    db[27] = db[9]                                          # duplicate -> exact tie -> lower index first
    q, _ = synth.sc_queries(46, db[9:10], 1)
    idx, sc = api.match_topk("sc", q, db, 0, 2.0, 2)
    assert list(idx[0]) == [9, 27] and sc[0, 0] == sc[0, 1]
    # k larger than the unmasked candidates -> -1 / NaN padding after the Inf entries in index order
    idx, sc = api.match_topk("sc", db[:3], db[:6], 100, 2.0, 6)
    assert np.array_equal(idx, np.tile(np.arange(6, dtype=np.int32), (3, 1))) and np.isinf(sc).all()


def test_match_full_size_properties(api):
    """Size-independent properties at a DB size the oracle cannot brute-force in seconds: planted top-1 recall
    = 100 % and distance of the planted pair far below the row mean."""
    n, m = 20000, 256
    db = synth.sc_database(45, n)
    q, et = synth.sc_queries(46, db, m)
    idx, sc = api.match_topk("sc", q, db)
    assert np.array_equal(idx[:, 0], et)
    assert (sc[:, 0] < -8).all()
    rc, oidx, osc = oracle_lib.match_topk(0, q[:2], db, 0)          # two full rows against the oracle
    assert np.array_equal(oidx[:, 0], idx[:2, 0]) and (np.abs(osc[:, 0] - sc[:2, 0]) <= helpers.score_tol(osc[:, 0])).all()


def test_match_at_baseline_size(api):
    """BASELINE.json's metric size (100k-signature DB): planted top-1 recall must be 100 %, and one full row is
    checked against the oracle (index bit-exact, every distance of the row within 1e-5)."""
    n, m = 100_000, 96
    db = synth.sc_database(45, n)
    q, et = synth.sc_queries(46, db, m)
    idx, sc = api.match_topk("sc", q, db)
    assert np.array_equal(idx[:, 0], et)
    rc, oidx, osc = oracle_lib.match_topk(0, q[:1], db, 0)
    assert oidx[0, 0] == idx[0, 0] and abs(osc[0, 0] - sc[0, 0]) <= helpers.score_tol(osc[0, 0])
    gp, gi = api.processSC(q[:1], db)
    rc, op, oi = oracle_lib.sc_distance(q[:1], db)
    assert np.abs(gp - op).max() < 1e-5 and np.abs(gi - oi).max() < 1e-5


# ------------------------------------------------------------------------------------------------ a3 + a4
@pytest.mark.parametrize("P", [1500, 20011])
def test_sc_generate_vs_oracle(api, P):
    xyz, it, offs = synth.scene_clouds(42, 3, P)
    o = oracle_lib.sc_generate(xyz, it, offs)
    g = api.sc_generate(xyz, it, offs)
    assert np.array_equal(g[:, 1200:], o[:, 1200:])                   # binarised intensity exact
    assert np.array_equal(g[:, :1200] > 0, o[:, :1200] > 0)
    assert np.abs(g[:, :1200] - o[:, :1200]).max() < 1e-10


@pytest.mark.parametrize("variant", ["batched", "cluster"])
def test_sc_generate_experiment_paths_vs_oracle(api, monkeypatch, variant):
    """The one-pass generation variants kept in the library (PR_SC_GEN=batched / cluster: several workgroups per cloud with grid-level
    hand-offs) give the oracle's signatures like the default two-pass path - large, ragged, tiny and empty clouds in one call."""
    monkeypatch.setenv("PR_SC_GEN", variant)
    parts = [synth.scene_cloud(42, 1, 30011), synth.scene_cloud(42, 2, 3), synth.scene_cloud(42, 3, 9000),
             synth.scene_cloud(42, 4, 20011)]
    xyz = np.concatenate([p[0] for p in parts]); it = np.concatenate([p[1] for p in parts])
    offs = np.array([0, 30011, 30011, 30014, 39014, 59025], np.int64)              # cloud 1 is empty
    o = oracle_lib.sc_generate(xyz, it, offs)
    g = api.sc_generate(xyz, it, offs)
    assert np.array_equal(g[1], np.zeros(2400))
    for r in (0, 2, 3, 4):
        assert np.array_equal(g[r, 1200:], o[r, 1200:]) and np.abs(g[r] - o[r]).max() < 1e-10
    for nclouds in (40, 300):                                                      # more clouds than clusters: the walk and its double buffers
        xyz, it, offs = synth.scene_clouds(43, nclouds, 9001)
        monkeypatch.setenv("PR_SC_GEN", variant)
        g = api.sc_generate(xyz, it, offs)
        monkeypatch.delenv("PR_SC_GEN")
        d = api.sc_generate(xyz, it, offs)
        assert np.array_equal(g[:, 1200:], d[:, 1200:]) and np.abs(g - d).max() < 1e-10


def test_sc_generate_ragged_and_empty(api, golden_dir):
    a, ia = synth.scene_cloud(42, 5, 700)
    b, ib = synth.scene_cloud(42, 6, 3)
    c, ic = synth.scene_cloud(42, 7, 1300)
    xyz = np.concatenate([a, b, c]); it = np.concatenate([ia, ib, ic])
    offs = np.array([0, 700, 700, 703, 2003], np.int64)               # an empty and a 3-point cloud
    o = oracle_lib.sc_generate(xyz, it, offs)
    g = api.sc_generate(xyz, it, offs)
    assert np.array_equal(g[1], np.zeros(2400))
    for r in (0, 3):
        assert np.array_equal(g[r, 1200:], o[r, 1200:]) and np.abs(g[r] - o[r]).max() < 1e-10
    gold = np.load(os.path.join(golden_dir, "synthetic_v1.npz"))
    xyz, it, offs = synth.scene_clouds(42, 3, int(gold["cloud_P"][0]))
    g = api.sc_generate(xyz, it, offs)
    assert np.array_equal(g[:, 1200:], gold["sc_sig"][:, 1200:]) and np.abs(g - gold["sc_sig"]).max() < 1e-10
    s = api.SC(45.0)
    st, iv = s.getSignature(xyz[:offs[1]], it[:offs[1]])
    assert s.getSignatureSize() == 1200 and np.array_equal(np.concatenate([st, iv]), g[0])


# ------------------------------------------------------------------------------------------------ a5
def test_m2dp_generate_vs_oracle(api, golden_dir):
    xyz, it, offs = synth.scene_clouds(42, 3, 2000)
    o = oracle_lib.m2dp_generate(xyz, it, offs)
    g = api.m2dp_generate(xyz, it, offs)
    assert g.shape == (12, 384)
    assert np.abs(g - o).max() < 1e-9
    gold = np.load(os.path.join(golden_dir, "synthetic_v1.npz"))
    assert np.abs(g - gold["m2dp_sig"]).max() < 1e-9
    # well-separated leading singular values: no row is named (the naming itself: tests/test_cli.py::test_config1_full_kitti_seq00)
    assert len(api.m2dp_svd_rows()) == 0 and not (api.default_context().take_warnings() & 2)


def test_m2dp_generate_intensity_accumulation_modes_vs_oracle(api):
    """m2dp_bin accumulates count and intensity of a projection in ONE 64-bit LDS atomic (fixed point on a per-cloud grid) and reruns a
    workgroup with the exact u32 + f64 accumulation whenever that cannot be vouched for: negative intensities, values far below the
    grid, all-zero clouds, clouds of 2^17 points or more.  Every case must give the oracle's signature."""
    rng = np.random.default_rng(23)
    clouds = []
    def cloud(P, inten):
        xyz = rng.normal(0, [14, 4, 9], (P, 3))
        clouds.append((xyz, np.asarray(inten, np.float32)))
    cloud(3000, rng.random(3000) * 255)                               # the fast mode
    cloud(3000, rng.normal(0, 30, 3000))                              # negative values: exact mode
    cloud(2500, np.zeros(2500))                                       # float average 0
    cloud(2500, np.where(rng.random(2500) < 0.5, 1e-30, 80.0))        # values far below the fixed-point grid
    cloud(2000, np.full(2000, 3e37))                                  # the float running sum overflows to inf: average inf
    cloud((1 << 17) + 77, rng.random((1 << 17) + 77))                 # the count field (17 bits) cannot hold the cloud
    cloud(1, [5.0])
    xyz, it, offs = api._csr(clouds)
    o = oracle_lib.m2dp_generate(xyz, it, offs)
    g = api.m2dp_generate(xyz, it, offs)
    assert g.shape == o.shape and np.abs(g - o).max() < 1e-9


def test_generators_with_caller_frames_equal_the_two_pass_calls(api):
    """pr_cloud_frames_dev + pr_*_generate_frames_dev (binning pass only) give the bits of pr_*_generate_dev, ragged sizes and an empty cloud
    included; the frames themselves hold mean, an orthonormal right-handed basis and the point count."""
    import torch
    rng = np.random.default_rng(11)
    sizes = rng.integers(1, 3000, size=40); sizes[7] = 0; sizes[8] = 1; sizes[9] = 257
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    xyz = rng.normal(0, [12, 3, 9], (offs[-1], 3)); it = rng.random(offs[-1]).astype(np.float32)
    dx, di, do = (torch.from_numpy(a).cuda() for a in (xyz, it, offs))
    N = len(sizes)
    ctx = api.Context(0)
    fr = torch.empty((N, 16), dtype=torch.float64, device="cuda")          # frames alone (slots 14, 15 zero)
    fa = torch.empty((N, 16), dtype=torch.float64, device="cuda")          # frames + the float intensity averages
    ctx.check(ctx.lib.pr_cloud_frames_dev(ctx.h, dx.data_ptr(), None, do.data_ptr(), N, fr.data_ptr()))
    ctx.check(ctx.lib.pr_cloud_frames_dev(ctx.h, dx.data_ptr(), di.data_ptr(), do.data_ptr(), N, fa.data_ptr()))
    ctx.sync()
    f = fr.cpu().numpy(); g = fa.cpu().numpy()
    big = sizes >= 3
    assert np.array_equal(f[:, 13], sizes.astype(np.float64))
    assert np.abs(f[big, :3] - np.stack([xyz[offs[c]:offs[c + 1]].mean(0) for c in np.nonzero(big)[0]])).max() < 1e-12
    R = f[big, 3:12].reshape(-1, 3, 3)
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12 and np.abs(np.linalg.det(R) - 1).max() < 1e-12
    assert np.array_equal(f[:, :14].view(np.uint64), g[:, :14].view(np.uint64)) and not f[:, 14:].any() and (g[:, 15] == 1.0).all()
    for c in np.nonzero(sizes > 0)[0]:                                     # the reference's sequential float sum in input order
        acc = np.float32(0)
        for v in it[offs[c]:offs[c + 1]]:
            acc = np.float32(acc + v)
        assert g[c, 14] == np.float64(np.float32(acc / np.float32(sizes[c]))), c
    for name, rows, cols, rho in (("sc", N, 2400, 45.0), ("m2dp", 4 * N, 384, 45.0), ("delight", 16 * N, 256, None)):
        a = torch.empty((rows, cols), dtype=torch.float64, device="cuda"); b = torch.empty_like(a); b2 = torch.empty_like(a)
        two = getattr(ctx.lib, f"pr_{name}_generate_dev"); one = getattr(ctx.lib, f"pr_{name}_generate_frames_dev")
        if rho is None:
            ctx.check(two(ctx.h, dx.data_ptr(), di.data_ptr(), do.data_ptr(), N, a.data_ptr()))
            ctx.check(one(ctx.h, dx.data_ptr(), di.data_ptr(), do.data_ptr(), N, fr.data_ptr(), b.data_ptr()))
            b2.copy_(b)
        else:
            ctx.check(two(ctx.h, dx.data_ptr(), di.data_ptr(), do.data_ptr(), N, rho, a.data_ptr()))
            ctx.check(one(ctx.h, dx.data_ptr(), di.data_ptr(), do.data_ptr(), N, rho, fr.data_ptr(), 0, b.data_ptr()))    # the call computes the averages
            ctx.check(one(ctx.h, dx.data_ptr(), di.data_ptr(), do.data_ptr(), N, rho, fa.data_ptr(), 1, b2.data_ptr()))   # the binning pass alone
        ctx.sync()
        assert torch.equal(a.view(torch.int64), b.view(torch.int64)) and torch.equal(a.view(torch.int64), b2.view(torch.int64)), name
    assert ctx.lib.pr_sc_generate_frames_dev(ctx.h, dx.data_ptr(), di.data_ptr(), do.data_ptr(), N, 45.0, None, 0, a.data_ptr()) == -1   # PR_EINVAL
    ctx.close()


def test_generators_large_ragged_batch_vs_oracle(api):
    """300 clouds of 1..1100 points (a batch large enough for the 16-plane M2DP workgroups and several rounds of the
    float-average chain, sizes that are not multiples of its 512-float chunks, an empty cloud in the middle)."""
    rng = np.random.default_rng(5)
    sizes = rng.integers(1, 1100, size=300); sizes[17] = 0; sizes[123] = 512; sizes[124] = 513; sizes[299] = 1
    parts = [synth.scene_cloud(77, 10 + i, int(n)) if n else (np.zeros((0, 3)), np.zeros((0,), np.float32)) for i, n in enumerate(sizes)]
    xyz = np.concatenate([p[0] for p in parts]); it = np.concatenate([p[1] for p in parts]).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    live = sizes > 8                                                   # PCA / leading singular pair of a handful of points are degenerate (N3, N6): not compared
    o = oracle_lib.sc_generate(xyz, it, offs); g = api.sc_generate(xyz, it, offs)
    assert np.array_equal(g[live][:, 1200:], o[live][:, 1200:]) and np.abs(g[live] - o[live]).max() < 1e-10
    assert np.array_equal(g[17], np.zeros(2400))
    o = oracle_lib.m2dp_generate(xyz, it, offs); g = api.m2dp_generate(xyz, it, offs)
    rows = np.repeat(live, 4)
    assert np.abs(g[rows] - o[rows]).max() < 1e-9
    o = oracle_lib.delight_generate(xyz, it, offs); g = api.delight_generate(xyz, it, offs)
    rows = np.repeat(live, 16)
    assert np.abs(g[rows] - o[rows]).sum() <= 4 * live.sum()           # float-cast boundary points may move between two bins


def test_generate_then_match_end_to_end(api):
    """config-2 shape at a small size: generate SC signatures on the GPU, match them on the GPU, compare the
    top-1 with the oracle run on the oracle's own signatures."""
    xyz, it, offs = synth.scene_clouds(42, 24, 4000)
    g = api.sc_generate(xyz, it, offs)
    o = oracle_lib.sc_generate(xyz, it, offs)
    gi, gs = api.match_topk("sc", g, g, 2)
    rc, oi, osc = oracle_lib.match_topk(0, o, o, 2)
    assert np.array_equal(gi, oi)


# ------------------------------------------------------------------------------------------------ f3 (DELIGHT)
@pytest.mark.parametrize("P", [1200, 30011])
def test_delight_generate_vs_oracle(api, P):
    xyz, it, offs = synth.scene_clouds(61, 5, P)
    it = (it * 1.5).astype(np.float32)
    offs = np.concatenate([offs[:3], offs[2:]])              # one empty cloud in the middle
    got = api.delight_generate(xyz, it, offs)
    want = oracle_lib.delight_generate(xyz, it, offs)
    assert got.shape == (16 * 6, 256)
    # integer counts; a point within 1 ulp(float) of an octant plane or of the 10 m sphere may land in the sibling histogram
    assert np.abs(got - want).sum() <= 4
    assert np.array_equal(got.reshape(6, 16, 256).sum(1), want.reshape(6, 16, 256).sum(1))


@pytest.mark.parametrize("m,n", [(1, 1), (7, 33), (40, 257)])
def test_delight_distance_vs_oracle(api, m, n):
    db = synth.delight_database(51, n)
    q, _ = synth.delight_queries(52, db, m)
    want = oracle_lib.delight_distance(q, db)
    got = api.processDELIGHT(q, db)
    assert got.shape == (m, n) and got.dtype == np.float32
    assert (np.abs(got - want) <= 1e-5 * np.maximum(1.0, np.abs(want))).all()
    z = np.zeros((16, 256))
    assert np.isinf(api.processDELIGHT(z, np.zeros((32, 256)))).all()


@pytest.mark.parametrize("mask,k", [(0, 1), (4, 3)])
def test_delight_match_topk_vs_oracle(api, mask, k):
    db = synth.delight_database(51, 300)
    q, et = synth.delight_queries(52, db, 64)
    rc, oidx, osc = oracle_lib.match_topk(2, q, db, mask, 2.0, k)
    idx, sc = api.match_topk("delight", q, db, mask_width=mask, k=k)
    assert np.array_equal(idx, oidx)
    assert (np.abs(sc - osc) <= 1e-5 * np.maximum(1.0, np.abs(osc))).all()
    if mask == 0:
        assert np.array_equal(idx[:, 0], et)


def test_delight_matcher_device_path(api):
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher
    db = synth.delight_database(51, 500)
    q, et = synth.delight_queries(52, db, 48)
    mt = Matcher("delight", 48, 500)
    mt.pack_database(torch.from_numpy(db).cuda())
    idx, sc = mt.match(torch.from_numpy(q).cuda(), mask_width=0, k=2)
    rc, oidx, osc = oracle_lib.match_topk(2, q, db, 0, 2.0, 2)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert (np.abs(sc.cpu().numpy() - osc) <= 1e-5 * np.maximum(1.0, np.abs(osc))).all()
    mt.close()


# ------------------------------------------------------------------------------------------------ a6, both arithmetics
def test_sc_both_arithmetics_vs_oracle(api):
    """The SC matcher ships in two arithmetics (include/place_recognition.h PR_SC_ARITH_*): split-f16 MFMA (default) and
    fp32 MFMA.  Both must meet the 1e-5 distance tolerance and give the same top-k as the oracle; measured error of
    either is ~1e-7."""
    db = synth.sc_database(45, 333)
    q, _ = synth.sc_queries(46, db, 64)
    rc, op, oi = oracle_lib.sc_distance(q, db)
    rc, oidx, osc = oracle_lib.match_topk(0, q, db, 3, 2.0, 4)
    errs = {}
    for arith in ("f16x2", "f32"):
        ctx = api.Context(0, sc_arith=arith)
        assert ctx.sc_arith == arith
        gp, gi = api.processSC(q, db, ctx)
        errs[arith] = max(np.abs(gp - op).max(), np.abs(gi - oi).max())
        assert errs[arith] < 2e-6                                   # well inside the 1e-5 of BASELINE.json
        idx, sc = api.match_topk("sc", q, db, 3, 2.0, 4, ctx=ctx)
        assert np.array_equal(idx, oidx)
        ctx.close()
    print("max |d - oracle|:", errs)


@pytest.mark.parametrize("kernel", ["h"])
def test_sc_experiment_kernels_vs_oracle(api, monkeypatch, kernel):
    """The other split-f16 SC matcher kept in the library - sc_match_h.hip (PR_SC_KERNEL=h: round 1's default; still the kernel for m <= 8) -
    must give the oracle's distances and top-k like the default kernel (sc_match_e.hip) when it runs the big shapes too; odd DB group
    counts and a ragged last query group included."""
    monkeypatch.setenv("PR_SC_KERNEL", kernel)
    for seed, n, m in ((51, 333, 64), (52, 1000, 21), (53, 2049, 9)):
        db = synth.sc_database(seed, n)
        q, _ = synth.sc_queries(seed + 100, db, m)
        rc, op, oi = oracle_lib.sc_distance(q, db)
        rc, oidx, osc = oracle_lib.match_topk(0, q, db, 3, 2.0, 4)
        ctx = api.Context(0)
        gp, gi = api.processSC(q, db, ctx)
        assert max(np.abs(gp - op).max(), np.abs(gi - oi).max()) < 2e-6
        idx, sc = api.match_topk("sc", q, db, 3, 2.0, 4, ctx=ctx)
        assert np.array_equal(idx, oidx)
        ctx.close()


@pytest.mark.parametrize("arith", ["f16x2", "f16"])
def test_sc_pack_of_a_few_signatures_is_the_batch_pack_bit_for_bit(api, monkeypatch, arith):
    """Up to 8 signatures are packed by a latency-oriented kernel (sc_pack_h_few_kernel: one workgroup per signature and channel), larger
    sets by the throughput kernel: the operand images must agree bit for bit.  Seen through the distances: rows 0..m-1 of a 20-query
    call == the m-query call, with the matcher pinned to one kernel (PR_SC_KERNEL=h serves any m in split-f16; the single-product
    kernel's two forms walk the same operands in the same order)."""
    monkeypatch.setenv("PR_SC_KERNEL", "h")
    db = synth.sc_database(61, 700)
    q, _ = synth.sc_queries(161, db, 20)
    ctx = api.Context(0, sc_arith=arith)
    big = api.processSC(q, db, ctx)
    for m in (1, 3, 8):
        few = api.processSC(q[:m], db, ctx)
        if arith == "f16x2":
            assert np.array_equal(few[0], big[0][:m]) and np.array_equal(few[1], big[1][:m])
        else:        # (the 8-wave online form of the single-product kernel orders nothing differently, but allow the last bit)
            assert np.abs(few[0] - big[0][:m]).max() < 1e-7 and np.abs(few[1] - big[1][:m]).max() < 1e-7
    ctx.close()


def test_sc_mixed_arithmetic_sets_are_rejected(api):
    import ctypes as C
    from so_dso_place_recognition_amd import _lib
    ctx = api.Context(0, sc_arith="f32")
    q = C.c_void_p(); d = C.c_void_p()
    assert ctx.lib.pr_sigset_create(ctx.h, _lib.TYPE_SC, _lib.ROLE_QUERY, 8, C.byref(q)) == 0
    assert ctx.lib.pr_set_sc_arith(ctx.h, _lib.SC_ARITH_F16X2) == 0
    assert ctx.lib.pr_sigset_create(ctx.h, _lib.TYPE_SC, _lib.ROLE_DB, 16, C.byref(d)) == 0
    dummy = C.c_void_p(16)
    assert ctx.lib.pr_distances_dev(ctx.h, q, d, dummy, dummy) == -1        # PR_EINVAL, nothing launched
    assert b"different arithmetic" in ctx.lib.pr_last_error(ctx.h)
    assert ctx.lib.pr_set_sc_arith(ctx.h, 7) == -1
    ctx.lib.pr_sigset_destroy(ctx.h, q); ctx.lib.pr_sigset_destroy(ctx.h, d)
    ctx.close()


@pytest.mark.parametrize("m,n,cols", [(1, 1, 1), (5, 70, 33), (65, 130, 96), (130, 64, 512)])
def test_gist_distance_vs_oracle(api, m, n, cols):
    a, b = synth.gist_signatures(11, m, cols), synth.gist_signatures(12, n, cols)
    d = api.processGIST(a, b); o = oracle_lib.gist_distance(a, b)
    assert d.shape == (m, n) and np.abs(d - o).max() <= 1e-6 * max(1e-3, o.max())


@pytest.mark.parametrize("m,n,cols,fill", [(1, 1, 8, (1, 8)), (6, 300, 60, (10, 50)), (40, 257, 30, (30, 30)), (3, 9, 4000, (50, 400))])
def test_bow_distance_vs_oracle(api, m, n, cols, fill):
    a = synth.bow_signatures(21, m, cols=cols, vocab=max(500, cols), fill=fill)
    b = synth.bow_signatures(22, n, cols=cols, vocab=max(500, cols), fill=fill)
    d = api.processBoW(a, b); o = oracle_lib.bow_distance(a, b)
    assert d.shape == (m, n) and np.abs(d - o).max() < 2e-7


@pytest.mark.parametrize("type_", ["gist", "bow"])
def test_plain_types_match_topk_vs_oracle(api, type_):
    if type_ == "gist":
        a, b = synth.gist_signatures(31, 37), synth.gist_signatures(32, 120)
        d = oracle_lib.gist_distance(a, b)
    else:
        a, b = synth.bow_signatures(33, 37), synth.bow_signatures(34, 120)
        d = oracle_lib.bow_distance(a, b)
    rc, oidx, osc = oracle_lib.select_topk(d, 4, 3)
    idx, sc = api.match_topk(type_, a, b, mask_width=4, k=3)
    assert np.array_equal(idx, oidx) and np.abs(sc - osc).max() <= 1e-6 * max(1.0, np.abs(osc).max())
    v, i = api.run_test(type_, a, b, mask_width=4)
    assert np.array_equal(i, oidx[:, 0])


def test_fused_sc_m2dp_scoring_vs_oracle(api):
    """BASELINE config 5 at a small size: the build-defined 4-channel score (no reference counterpart) against the fp64
    oracle - indices exact, scores to float accuracy; a planted copy must win."""
    n, m = 230, 41
    sdb = synth.sc_database(71, n); sq, planted = synth.sc_queries(72, sdb, m)
    mdb = synth.m2dp_database(73, n); mq, planted2 = synth.m2dp_queries(74, mdb, m)
    rc, oidx, osc = oracle_lib.match_topk_fused(sq, mq, sdb, mdb, 3, 2.0, 3)
    assert rc == 0
    idx, sc = api.match_topk_fused(sq, mq, sdb, mdb, mask_width=3, p_weight=2.0, k=3)
    sg = [helpers.row_sigmas(*oracle_lib.sc_distance(sq, sdb)[1:]), helpers.row_sigmas(*oracle_lib.m2dp_distance(mq, mdb)[1:])]
    assert np.array_equal(idx, oidx) and (np.abs(sc - osc) <= helpers.score_tol(osc, sg[0]) + helpers.score_tol(osc, sg[1])).all()
    # one channel pair switched off (identical rows give z = NaN there) is not the point; agreement of SC-only with the fused
    # top-1 on queries planted in BOTH databases at the same index is
    same = planted == planted2
    if same.any():
        assert (idx[same, 0] == planted[same]).all()


def test_plain_matchers_size_independent_properties(api):
    """DELIGHT / GIST / BoW at a size the oracle would not finish quickly: self-distance matrices are symmetric (the octant
    permutations are involutions; L2 and L1 scores are symmetric) with a zero diagonal, and every row's top-1 under a
    1-wide mask is never the row itself."""
    rng = np.random.default_rng(9)
    n = 1500
    h = rng.poisson(3.0, size=(16 * n, 256)).astype(np.float64)
    d = api.processDELIGHT(h, h)
    assert d.shape == (n, n) and np.abs(d - d.T).max() < 1e-5 * d.max() and np.abs(np.diag(d)).max() < 1e-6
    g = synth.gist_signatures(10, n, 200)
    d = api.processGIST(g, g)
    assert np.abs(d - d.T).max() <= 1e-6 * d.max() and np.abs(np.diag(d)).max() == 0.0
    b = synth.bow_signatures(11, 600, cols=200, vocab=2000, fill=(40, 150))
    d = api.processBoW(b, b)
    assert np.abs(d - d.T).max() < 1e-6 and np.abs(np.diag(d)).max() < 1e-6
    idx, sc = api.match_topk("bow", b, b, mask_width=1, k=1)
    assert (idx[:, 0] != np.arange(600)).all()


def test_fused_matcher_device_path_and_two_shards(api):
    """FusedMatcher (config 5 on device-resident signatures): one rank == the host API; and the sharded arithmetic - the DB
    split in two, per-shard moments stacked in rank order, per-shard top-k with GLOBAL indices, k-way merge - gives the same
    top-k as the unsharded run (what two ranks would compute, without the processes)."""
    import torch
    from so_dso_place_recognition_amd.matcher import FusedMatcher, merge_topk, _dptr
    n, m, k, mask = 301, 37, 3, 2
    sdb = synth.sc_database(81, n); sq, _ = synth.sc_queries(82, sdb, m)
    mdb = synth.m2dp_database(83, n); mq, _ = synth.m2dp_queries(84, mdb, m)
    want_idx, want_sc = api.match_topk_fused(sq, mq, sdb, mdb, mask_width=mask, k=k)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fm = FusedMatcher(m, n)
    fm.pack_database(t(sdb), t(mdb))
    idx, sc = fm.match(t(sq), t(mq), mask, 2.0, k)
    assert np.array_equal(idx.cpu().numpy(), want_idx) and np.abs(sc.cpu().numpy() - want_sc).max() < 1e-12
    fm.close()
    cut = 160                                                            # shard 0: rows [0, 160), shard 1: [160, 301)
    parts, moms = [], []
    for lo, hi in ((0, cut), (cut, n)):
        f = FusedMatcher(m, hi - lo)
        f.pack_database(t(sdb[lo:hi]), t(mdb[4 * lo:4 * hi]))
        moms.append(f.local_phase1(t(sq), t(mq)).clone())
        parts.append((f, lo))
    mom_all = torch.stack(moms)                                          # [2, m, 4, 3] in rank order
    outs = [tuple(x.clone() for x in f.local_phase2(mom_all, 2, mask, 2.0, k, lo, 0)) for f, lo in parts]   # every shard's own top-k
    idx2, sc2 = parts[0][0].merge(torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), k)
    assert np.array_equal(idx2.cpu().numpy(), want_idx) and np.abs(sc2.cpu().numpy() - want_sc).max() < 1e-9
    i3, s3 = merge_topk(torch.stack([o[0].cpu() for o in outs]), torch.stack([o[1].cpu() for o in outs]), k)   # torch restatement
    assert np.array_equal(i3.numpy(), want_idx)
    # the production protocol: fp32 lists merged first, owners re-evaluate, finish
    sel = [tuple(x.clone() for x in f.local_select(mom_all, 2, mask, 2.0, k, lo, 0)) for f, lo in parts]
    cand, cand_sc = parts[0][0].merge(torch.stack([a for a, b in sel]), torch.stack([b for a, b in sel]), sel[0][0].shape[1])
    part_all = torch.stack([f.local_rerank(cand, k, True).clone() for f, lo in parts])
    idx4, sc4 = parts[0][0].finish(cand, cand_sc, part_all, k)
    assert np.array_equal(idx4.cpu().numpy(), want_idx) and np.abs(sc4.cpu().numpy() - want_sc).max() < 1e-9
    for f, lo in parts:
        f.close()


def test_sigset_capacity_limit_is_an_error_not_an_overflow(api):
    import ctypes as C
    from so_dso_place_recognition_amd import _lib
    ctx = api.Context(0)
    q = C.c_void_p()
    assert ctx.lib.pr_sigset_create(ctx.h, _lib.TYPE_M2DP, _lib.ROLE_DB, 4_000_001, C.byref(q)) == -1   # PR_EINVAL
    assert b"PR_MAX_SIGS" in ctx.lib.pr_last_error(ctx.h) and not q.value
    ctx.close()


def test_m2dp_both_arithmetics_vs_oracle(api):
    """The arithmetic switch covers the M2DP matcher too (split-f16 GEMM by default, fp32 MFMA with PR_SC_ARITH_F32)."""
    db = synth.m2dp_database(43, 130)
    q, _ = synth.m2dp_queries(44, db, 33)
    rc, oc, oi = oracle_lib.m2dp_distance(q, db)
    rc, oidx, osc = oracle_lib.match_topk(1, q, db, 2, 2.0, 3)
    for arith in ("f16x2", "f32"):
        ctx = api.Context(0, sc_arith=arith)
        gc, gi = api.processM2DP(q, db, ctx)
        assert max(np.abs(gc - oc).max(), np.abs(gi - oi).max()) < 2e-6
        idx, sc = api.match_topk("m2dp", q, db, 2, 2.0, 3, ctx=ctx)
        assert np.array_equal(idx, oidx)
        ctx.close()


# ------------------------------------------------------------------------------------------------ e: pr_group (multi-GPU inside the C ABI)
@pytest.mark.parametrize("type_,devices,exchange", [("sc", [0], "rccl"), ("sc", [0, 0, 0], None), ("m2dp", [0, 0], None),
                                                    ("sc", [0] * 8, None)])
def test_group_sharded_match_equals_unsharded(api, type_, devices, exchange, monkeypatch):
    """pr_group on this one-GPU box: a one-rank RCCL communicator (ncclCommInitAll + ncclAllGather on the compute stream really
    run), and 2 / 3 / 8 shards on device 0 (the sharded arithmetic: global offsets, moments of all shards, merge) exchanged by
    device copies.  Either way the answer must be the unsharded one and the oracle's."""
    if exchange:
        monkeypatch.setenv("PR_GROUP_EXCHANGE", exchange)
    n, m, k, mask = 403, 57, 3, 4
    if type_ == "sc":
        db = synth.sc_database(45, n); q, _ = synth.sc_queries(46, db, m); t = 0
    else:
        db = synth.m2dp_database(43, n); q, _ = synth.m2dp_queries(44, db, m); t = 1
    g = api.Group(devices)
    assert g.uses_rccl == (exchange == "rccl")
    g.set_database(type_, db)
    idx, sc = g.match_topk(q, mask, 2.0, k)
    want_idx, want_sc = api.match_topk(type_, q, db, mask, 2.0, k)
    rc, oidx, osc = oracle_lib.match_topk(t, q, db, mask, 2.0, k)
    assert np.array_equal(idx, want_idx) and np.array_equal(idx, oidx)
    rc, odp, odi = oracle_lib.sc_distance(q, db) if t == 0 else oracle_lib.m2dp_distance(q, db)
    assert np.abs(sc - want_sc).max() < 1e-9 and (np.abs(sc - osc) <= helpers.score_tol(osc, helpers.row_sigmas(odp, odi))).all()
    idx2, sc2 = g.match_topk(q[:div_rows(type_) * 5], mask, 2.0, 1)          # a second, smaller call on the same group
    assert np.array_equal(idx2[:, 0], want_idx[:5, 0])
    g.close()


def div_rows(type_):
    return 4 if type_ == "m2dp" else 1


def test_matcher_float32_signatures_and_larger_k(api):
    """Signatures stored as float32 in HBM (half the memory of the reference's doubles): the pack kernels and the fp64
    re-evaluation read them as they are; the oracle gets the same float32-rounded values.  k = 7."""
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher
    n, m, k = 700, 50, 7
    db = synth.sc_database(45, n).astype(np.float32); q, _ = synth.sc_queries(46, db.astype(np.float64), m)
    q = q.astype(np.float32)
    mt = Matcher("sc", m, n)
    mt.pack_database(torch.from_numpy(db).cuda())
    idx, sc = mt.match(torch.from_numpy(q).cuda(), mask_width=2, k=k)
    rc, oidx, osc = oracle_lib.match_topk(0, q.astype(np.float64), db.astype(np.float64), 2, 2.0, k)
    assert np.array_equal(idx.cpu().numpy(), oidx) and (np.abs(sc.cpu().numpy() - osc) <= helpers.score_tol(osc)).all()
    mt.close()


@pytest.mark.parametrize("m", [4, 32, 80])
def test_match_as_hipgraph_replay(api, m):
    """One match() call captured into a hipGraph (online use: a keyframe per call): replays with new signatures in the static
    input give what the eager call gives."""
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher
    n, k = 3000, 2
    db = synth.sc_database(45, n); q, planted = synth.sc_queries(46, db, 3 * m)
    mt = Matcher.on_new_stream("sc", m, n)
    with torch.cuda.stream(mt.stream):
        mt.pack_database(torch.from_numpy(db).cuda())
        qs = torch.from_numpy(q[:m]).cuda()
    cap = mt.capture(qs, 0, 2.0, k)
    for r in range(3):
        idx, sc = cap.run(torch.from_numpy(q[r * m:(r + 1) * m]).cuda())
        want_idx, want_sc = api.match_topk("sc", q[r * m:(r + 1) * m], db, 0, 2.0, k)
        assert np.array_equal(idx.cpu().numpy(), want_idx) and np.abs(sc.cpu().numpy() - want_sc).max() < 1e-9
        assert np.array_equal(idx.cpu().numpy()[:, 0], planted[r * m:(r + 1) * m])
    mt.close()


@pytest.mark.parametrize("m,n", [(1, 1000), (3, 333), (8, 4099), (5, 17)])
def test_sc_small_query_batches_use_the_one_group_kernel(api, m, n):
    """m <= 8 (a keyframe per call): the four waves of a workgroup share one query group and split the DB groups
    (sc_match_h_kernel<true>).  Same distances, same top-k."""
    db = synth.sc_database(45, n)
    q, _ = synth.sc_queries(46, db, m)
    rc, op, oi = oracle_lib.sc_distance(q, db)
    gp, gi = api.processSC(q, db)
    assert np.abs(gp - op).max() < 1e-5 and np.abs(gi - oi).max() < 1e-5
    k = min(3, n)
    idx, sc = api.match_topk("sc", q, db, 0, 2.0, k)
    rc, oidx, osc = oracle_lib.match_topk(0, q, db, 0, 2.0, k)
    assert np.array_equal(idx, oidx)


def test_matcher_called_under_another_torch_stream(api):
    """The library context lives on the stream that was current at construction; calls made later under `with torch.cuda.stream(s)` put
    torch's own work (casts, allocations, zero fills) on ANOTHER stream.  _enter / _leave compare the streams at every call and join
    them with events: the results must be those of the plain call, also when the side stream is busy with unrelated work."""
    import torch
    from so_dso_place_recognition_amd.matcher import Matcher
    n, m, k = 6000, 96, 3
    db = torch.from_numpy(synth.sc_database(45, n)).cuda()
    q_h, planted = synth.sc_queries(46, db.cpu().numpy(), m)
    mt = Matcher("sc", m, n)
    mt.pack_database(db)
    q = torch.from_numpy(q_h).cuda()
    ref_i, ref_s = (t.clone() for t in mt.match(q, 2, 2.0, k))
    side = torch.cuda.Stream()
    junk = torch.empty((4096, 4096), device="cuda")
    for rep in range(3):
        with torch.cuda.stream(side):
            for _ in range(4):
                junk = junk @ junk.T * 1e-4                           # keeps the side stream busy in front of the call
            q2 = torch.from_numpy(q_h).cuda()                        # the queries are produced on the side stream
            mt.pack_database(db)
            i2, s2 = mt.match(q2, 2, 2.0, k)
            i2, s2 = i2.clone(), s2.clone()
        side.synchronize()
        assert torch.equal(i2, ref_i) and torch.equal(s2, ref_s)
    mt.close()
