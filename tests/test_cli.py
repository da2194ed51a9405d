"""The drop-in executables test_sc / test_m2dp / match_signatures (BASELINE.json config 1: real KITTI poses of the
reference + synthetic points, end to end)."""
import os
import subprocess

import numpy as np
import pytest

import helpers
import oracle_lib
from so_dso_place_recognition_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "so_dso_place_recognition_amd", "bin")


def test_missing_params_exit_code_and_message():
    for exe in ("test_sc", "test_m2dp", "test_delight"):
        r = subprocess.run([os.path.join(BIN, exe), "_poses_history_file:=x"], capture_output=True, text=True)
        assert r.returncode == 1 and "Fail to get params, exit." in r.stdout      # test_sc.cpp:19-25
    r = subprocess.run([os.path.join(BIN, "match_signatures"), "--type", "gist"], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stdout                              # --hist1 / --hist2 / --out missing
    r = subprocess.run([os.path.join(BIN, "match_signatures"), "--type", "orb", "--hist1", "a", "--hist2", "b", "--out", "c"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stdout


@pytest.mark.gpu
def test_config1_end_to_end(golden_dir, tmp_path):
    # first 110 poses of the reference's KITTI seq07 file (every cloud then holds >= 2000 points: with near-empty
    # windows the 64x128 M2DP matrices have sigma1 == sigma2 and the leading singular pair is not unique, N6)
    full = open(os.path.join(golden_dir, "kitti_seq07", "poses_history_file.txt")).read().split("\n")
    poses = str(tmp_path / "poses_history_file.txt")
    open(poses, "w").write("\n".join(full[:110]) + "\n")
    pts = str(tmp_path / "pts_history_file.txt")
    helpers.write_synthetic_points(poses, pts, per_pose=80)
    out = {}
    for exe, key, polar in (("test_sc", "sc_file", False), ("test_m2dp", "m2dp_file", True),
                            ("test_delight", "delight_file", True)):
        sig = str(tmp_path / f"history_{exe}.txt"); ids = str(tmp_path / f"ids_{exe}.txt")
        r = subprocess.run([os.path.join(BIN, exe), f"_poses_history_file:={poses}", f"_pts_history_file:={pts}",
                            f"_{key}:={sig}", f"_incoming_id_file:={ids}", "_lidarRange:=45.0"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "average time" in r.stdout and "generate_spherical_points average time" in r.stdout
        x, it, offs, oid = oracle_lib.pts_preprocess(poses, pts, None, 45.0, polar)
        assert [int(v) for v in open(ids).read().split()] == list(oid)
        got = np.loadtxt(sig)
        want = (oracle_lib.delight_generate(x, it, offs) if exe == "test_delight" else
                oracle_lib.m2dp_generate(x, it, offs) if polar else oracle_lib.sc_generate(x, it, offs))
        assert got.shape == want.shape
        assert np.allclose(got, want, rtol=2e-5, atol=1e-12)                   # 6 significant digits in the text
        out[exe] = (sig, got)
    for type_, exe, t in (("sc", "test_sc", 0), ("m2dp", "test_m2dp", 1), ("delight", "test_delight", 2)):
        sig, got = out[exe]
        res = str(tmp_path / f"match_{type_}.txt")
        r = subprocess.run([os.path.join(BIN, "match_signatures"), "--type", type_, "--hist1", sig, "--hist2", sig,
                            "--mask_width", "10", "--topk", "2", "--out", res], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "tm =" in r.stdout
        m = np.loadtxt(res)
        rc, oidx, osc = oracle_lib.match_topk(t, got, got, 10, 2.0, 2)         # same (text-rounded) inputs on both sides
        assert np.array_equal(m[:, [0, 2]].astype(np.int32), oidx)
        if t == 2:
            assert np.abs(m[:, [1, 3]] - osc).max() < 1e-5 * max(1.0, np.abs(osc).max())        # DELIGHT: one fp32 chi-square matrix
        else:       # toy clouds: every row's distances lie within ~1e-3 of each other, so the z-score amplifies the fp32 pass's 2e-8 (helpers.score_tol)
            rc, odp, odi = oracle_lib.sc_distance(got, got) if t == 0 else oracle_lib.m2dp_distance(got, got)
            assert (np.abs(m[:, [1, 3]] - osc) <= helpers.score_tol(osc, helpers.row_sigmas(odp, odi), eps=1e-7)).all()


@pytest.mark.gpu
def test_gpu_prestage_option_gives_the_same_files(golden_dir, tmp_path):
    """`_gpu_prestage:=1` (row f1): same incoming ids and the same signature file as the host pre-stage."""
    full = open(os.path.join(golden_dir, "kitti_seq07", "poses_history_file.txt")).read().split("\n")
    poses = str(tmp_path / "poses_history_file.txt")
    open(poses, "w").write("\n".join(full[:90]) + "\n")
    pts = str(tmp_path / "pts_history_file.txt")
    helpers.write_synthetic_points(poses, pts, per_pose=60)
    outs = []
    for g in (0, 1):
        sig = str(tmp_path / f"sc_{g}.txt"); ids = str(tmp_path / f"ids_{g}.txt")
        r = subprocess.run([os.path.join(BIN, "test_sc"), f"_poses_history_file:={poses}", f"_pts_history_file:={pts}",
                            f"_sc_file:={sig}", f"_incoming_id_file:={ids}", f"_gpu_prestage:={g}"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "generate_spherical_points average time" in r.stdout
        outs.append((open(sig, "rb").read(), open(ids, "rb").read()))
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_match_signatures_gist_and_bow(tmp_path):
    """run_test.m:32-35 through the executable: text files in the reference's formats (BoW: two padded rows per image,
    test_bow.cpp:147-162), plain row minimum with mask."""
    from so_dso_place_recognition_amd import api, synth
    for type_, h1, h2 in (("gist", synth.gist_signatures(5, 23), synth.gist_signatures(6, 57)),
                          ("bow", synth.bow_signatures(7, 19), synth.bow_signatures(8, 41))):
        f1, f2, res = str(tmp_path / f"{type_}1.txt"), str(tmp_path / f"{type_}2.txt"), str(tmp_path / f"{type_}.out")
        if type_ == "bow":                                                        # the reference's writer: "v " per entry, endl per row
            for f, h in ((f1, h1), (f2, h2)):
                open(f, "w").write("".join("".join(("%d " % v) if r % 2 == 0 or v == -1 else ("%.9g " % v) for v in row) + "\n"
                                           for r, row in enumerate(h)))
        else:
            api.write_signatures(f1, h1); api.write_signatures(f2, h2)
        r = subprocess.run([os.path.join(BIN, "match_signatures"), "--type", type_, "--hist1", f1, "--hist2", f2,
                            "--mask_width", "3", "--topk", "2", "--out", res], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        a, b = np.loadtxt(f1), np.loadtxt(f2)                                     # the text-rounded values both sides see
        d = oracle_lib.gist_distance(a, b) if type_ == "gist" else oracle_lib.bow_distance(a, b)
        rc, oidx, osc = oracle_lib.select_topk(d, 3, 2)
        m = np.loadtxt(res)
        assert np.array_equal(m[:, [0, 2]].astype(np.int32), oidx)
        assert np.abs(m[:, [1, 3]] - osc).max() < 1e-5 * max(1.0, np.abs(osc).max())


@pytest.mark.gpu
def test_config1_full_kitti_seq00(ref_sequence, tmp_path):
    """BASELINE.json config 1 at its stated size: all 3505 poses of the reference's KITTI seq00 file x 400 synthetic points per
    pose (the reference's point file is a missing blob) through test_sc / test_m2dp / match_signatures --mask_width 100
    (test_kitti.m:19).  The incoming ids must be byte-equal to the file the reference itself holds
    (results/KITTI/seq00/incoming_id_file.txt); every signature is compared with the oracle's, the matches on a row sample."""
    poses, ref_ids = ref_sequence("kitti_seq00")
    pts = str(tmp_path / "pts_history_file.txt")
    helpers.write_synthetic_points(poses, pts, per_pose=400)
    sigs = {}
    for exe, key, polar in (("test_sc", "sc_file", False), ("test_m2dp", "m2dp_file", True)):
        sig = str(tmp_path / f"history_{exe}.txt"); ids = str(tmp_path / f"ids_{exe}.txt")
        r = subprocess.run([os.path.join(BIN, exe), f"_poses_history_file:={poses}", f"_pts_history_file:={pts}",
                            f"_{key}:={sig}", f"_incoming_id_file:={ids}", "_lidarRange:=45.0"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(ids, "rb").read() == open(ref_ids, "rb").read()                  # the reference's own known answer
        x, it, offs, oid = oracle_lib.pts_preprocess(poses, pts, None, 45.0, polar)
        assert len(oid) == 3475
        got = np.loadtxt(sig)
        want = oracle_lib.m2dp_generate(x, it, offs) if polar else oracle_lib.sc_generate(x, it, offs)
        assert got.shape == want.shape
        bad = ~np.isclose(got, want, rtol=2e-5, atol=1e-12)                          # 6 significant digits in the text
        if polar:   # a cloud whose two leading singular values nearly coincide has no unique leading pair (N6): exactly the rows the
            # library names (pr_m2dp_svd_rows, printed by test_m2dp) may differ, and they are a handful
            line = [l for l in r.stderr.splitlines() if l.startswith("warning: leading singular pair not unique")]
            named = set(int(t) for t in line[0].split(":")[2].split()) if line else set()
            assert set(np.nonzero(bad.any(1))[0].tolist()) <= named, (np.nonzero(bad.any(1))[0], named)
            assert len(named) <= 8, named
            # ... and they do not move the matcher: top-1 of every query (mask 100) from the library's signatures == from the oracle's,
            # except where the query or one of the two answers IS such a cloud
            dev_sig = api.m2dp_generate(x, it, offs)
            i_lib, _ = api.match_topk("m2dp", dev_sig, dev_sig, 100, 2.0, 1)
            i_ora, _ = api.match_topk("m2dp", want, want, 100, 2.0, 1)
            clouds = set(r // 4 for r in named)
            diff = np.flatnonzero(i_lib[:, 0] != i_ora[:, 0])
            assert all((q in clouds) or (int(i_lib[q, 0]) in clouds) or (int(i_ora[q, 0]) in clouds) for q in diff.tolist()), (diff, clouds)
            assert len(diff) <= 8 * 3
        else:
            assert not bad.any()
        sigs[exe] = (sig, got)
    rows = np.arange(0, 3475, 217)
    for type_, exe, t in (("sc", "test_sc", 0), ("m2dp", "test_m2dp", 1)):
        sig, got = sigs[exe]
        res = str(tmp_path / f"match_{type_}.txt")
        r = subprocess.run([os.path.join(BIN, "match_signatures"), "--type", type_, "--hist1", sig, "--hist2", sig,
                            "--mask_width", "100", "--out", res], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        mres = np.loadtxt(res)
        div = 4 if t else 1
        qr = got.reshape(-1, div, got.shape[1])[rows].reshape(-1, got.shape[1])
        fn = oracle_lib.m2dp_distance if t else oracle_lib.sc_distance
        rc, a, b = fn(got, qr)                                                       # roles swapped (symmetric distance), see test_gpu_configs.py
        dp, di = a.T, b.T
        z = lambda d: (d - d.mean(1, keepdims=True)) / np.sqrt(((d - d.mean(1, keepdims=True)) ** 2).sum(1, keepdims=True) / (d.shape[1] - 1))
        f = 2.0 * z(dp) + z(di)
        f[np.abs(rows[:, None] - np.arange(f.shape[1])[None, :]) < 100] = np.inf
        assert np.array_equal(mres[rows, 0].astype(np.int64), f.argmin(1))
        sg = (np.std(dp, axis=1, ddof=1), np.std(di, axis=1, ddof=1), dp.shape[1])
        assert (np.abs(mres[rows, 1] - f.min(1)) <= helpers.score_tol(f.min(1), sg)).all()


@pytest.mark.gpu
def test_match_signatures_devices_and_ground_truth(tmp_path):
    """`match_signatures --devices 0,0` (hist2 row-sharded through pr_group; two shards on the one GPU of the test box) gives
    the file of the single-context run, and `--gt1 --gt2 --loop_diff` prints the AUC / top recall of run_test.m:58-85 as
    eval.precision_recall computes them from the same matches."""
    from so_dso_place_recognition_amd import synth, eval as ev
    n = 260
    sig = synth.sc_database(45, n)
    sig[130:] = synth.sc_queries(46, sig[:130], 130)[0]                      # second half: revisits of (random) places of the first half
    f = str(tmp_path / "history_sc.txt"); api.write_signatures(f, sig)
    rng = np.random.default_rng(4)
    gt = np.cumsum(rng.normal(0, 4, (n, 3)), 0)
    g = str(tmp_path / "gt.txt"); np.savetxt(g, gt)
    outs = []
    for extra in ([], ["--devices", "0,0"]):
        res = str(tmp_path / f"m{len(extra)}.txt")
        r = subprocess.run([os.path.join(BIN, "match_signatures"), "--type", "sc", "--hist1", f, "--hist2", f, "--mask_width", "20",
                            "--out", res, "--gt1", g, "--gt2", g, "--loop_diff", "15"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append((np.loadtxt(res), r.stdout))
    assert np.array_equal(outs[0][0][:, 0], outs[1][0][:, 0]) and np.abs(outs[0][0][:, 1] - outs[1][0][:, 1]).max() < 1e-9
    assert "devices = 2 (copies)" in outs[1][1]
    sig6 = np.loadtxt(f)                                                    # (the 6 significant digits the text file holds)
    rc, oidx, osc = oracle_lib.match_topk(0, sig6, sig6, 20, 2.0, 1)        # ... and both are the oracle's answer (run_test.m:25-57)
    assert rc == 0
    for res, _ in outs:
        assert np.array_equal(res[:, 0].astype(np.int64), oidx[:, 0])
        assert (np.abs(res[:, 1] - osc[:, 0]) <= 1e-5 + 1e-6 * np.abs(osc[:, 0])).all()
    m = outs[0][0]
    auc, tr, det = ev.precision_recall(m[:, 1], m[:, 0].astype(np.int64), np.loadtxt(g), np.loadtxt(g), 15.0, 20)[:3]
    line = {l.split(" = ")[0]: l.split(" = ")[1] for l in outs[0][1].splitlines() if " = " in l}
    assert abs(float(line["AUC"]) - auc) < 1e-9 or (np.isnan(auc) and line["AUC"].strip() == "nan")
    assert abs(float(line["top_recall"]) - tr) < 1e-9 and int(line["lp_detected"]) == len(det)


@pytest.mark.gpu
@pytest.mark.parametrize("type_,mask,devices", [("sc", 12, None), ("sc", 0, "0,0"), ("m2dp", 5, None)])
def test_match_signatures_online_mode(tmp_path, type_, mask, devices):
    """match_signatures --online 1: keyframe t against rows 0 .. t - mask of the same file, the database grown in place on the GPU
    (pr_group_set_database_growable / pr_group_append_database).  Every answered row equals the oracle on the database of that moment (binary file:
    the signatures reach the executable bit for bit)."""
    from so_dso_place_recognition_amd import synth
    N, k = 90, 2
    rows = 1 if type_ == "sc" else 4
    sig = synth.sc_database(301, N) if type_ == "sc" else synth.m2dp_database(302, N)
    for t in (40, 55, 70):                                   # revisits: near-copies of earlier frames
        src = t - 30
        sig[t * rows:(t + 1) * rows] = sig[src * rows:(src + 1) * rows] * (1.0 + 1e-3 * np.random.default_rng(t).standard_normal((rows, sig.shape[1])))
    f = str(tmp_path / "hist.bin")
    from so_dso_place_recognition_amd import _lib
    import ctypes as C
    sig = np.ascontiguousarray(sig, np.float64)
    assert _lib.load().pr_write_signatures_bin(f.encode(), sig.ctypes.data_as(C.c_void_p), sig.shape[0], sig.shape[1], _lib.F64) == 0
    out = str(tmp_path / "online.txt")
    cmd = [os.path.join(BIN, "match_signatures"), "--type", type_, "--hist1", f, "--hist2", f, "--online", "1", "--mask_width", str(mask),
           "--topk", str(k), "--out", out]
    if devices:
        cmd += ["--devices", devices]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lag = max(mask - 1, 0)
    assert f"database rows at the end: {N - 1 - lag}" in r.stdout
    got = [l.split() for l in open(out).read().strip().split("\n")]
    assert len(got) == N
    tcode = 0 if type_ == "sc" else 1
    for t in range(N):
        have = t - lag
        if have < 2:
            assert got[t][0] == "-1"
            continue
        rc, oidx, osc = oracle_lib.match_topk(tcode, sig[t * rows:(t + 1) * rows], sig[:have * rows], 0, 2.0, k)
        assert rc == 0
        gi = [int(got[t][0]), int(got[t][2])]
        assert gi == list(oidx[0]), (t, gi, oidx)
        gs = np.array([float(got[t][1]), float(got[t][3])])
        ok = np.isfinite(osc[0])
        rc, odp, odi = (oracle_lib.sc_distance if type_ == "sc" else oracle_lib.m2dp_distance)(sig[t * rows:(t + 1) * rows], sig[:have * rows])
        tol = helpers.score_tol(osc[0], helpers.row_sigmas(odp, odi), eps=1e-7)[0]          # short rows of a toy database: the fp32-statistics model
        assert (np.abs(gs[ok] - osc[0][ok]) <= tol[ok]).all(), (t, gs, osc[0], tol)
    for t, src in ((40, 10), (55, 25), (70, 40)):            # the revisits find their places
        if src < t - lag:
            assert int(got[t][0]) == src
