"""Rows a1/a2 on the CPU: the product's host pre-stage (libpr_amd.so, host_io.cpp) vs the oracle on config-1 style
input (real KITTI pose file of the reference + synthetic points), and the text formats."""
import os

import numpy as np
import pytest

import helpers
import oracle_lib
from conftest import REF_SEQUENCES
from so_dso_place_recognition_amd import api


@pytest.fixture(scope="module")
def seq07(golden_dir, tmp_path_factory):
    d = tmp_path_factory.mktemp("seq07")
    poses = os.path.join(golden_dir, "kitti_seq07", "poses_history_file.txt")
    pts = str(d / "pts_history_file.txt")
    helpers.write_synthetic_points(poses, pts, per_pose=60, max_poses=140)
    return poses, pts, d


@pytest.mark.parametrize("polar", [False, True])
def test_prestage_matches_oracle_bit_for_bit(seq07, polar):
    poses, pts, d = seq07
    ido = str(d / f"ids_oracle_{polar}.txt"); idp = str(d / f"ids_product_{polar}.txt")
    ox, oi, oo, oid = oracle_lib.pts_preprocess(poses, pts, ido, 45.0, polar)
    px, pi, po, pid = api.pts_preprocess(poses, pts, idp, 45.0, polar)
    assert np.array_equal(po, oo) and np.array_equal(pid, oid)
    assert len(oo) > 100 and oo[-1] > 10000                       # the window really accumulates points
    assert np.array_equal(px.view(np.uint64), ox.view(np.uint64))  # same points, same ORDER, same bits
    assert np.array_equal(pi.view(np.uint32), oi.view(np.uint32))
    assert open(idp, "rb").read() == open(ido, "rb").read()


def test_incoming_ids_byte_identical_to_reference_file(golden_dir, tmp_path):
    for seq in ("kitti_seq06", "kitti_seq07"):
        poses = os.path.join(golden_dir, seq, "poses_history_file.txt")
        empty = tmp_path / "empty.txt"; empty.write_text("")
        out = tmp_path / f"{seq}.txt"
        x, it, offs, ids = api.pts_preprocess(poses, str(empty), str(out))
        assert out.read_bytes() == open(os.path.join(golden_dir, seq, "incoming_id_file.txt"), "rb").read()
        assert offs[-1] == 0 and len(ids) == len(offs) - 1


def test_missing_files_yield_zero_clouds_like_the_reference(tmp_path):
    # ifstream open failures are not checked in the reference (pts_preprocess.h:22,39): zero clouds, no error
    x, it, offs, ids = api.pts_preprocess(str(tmp_path / "nope.txt"), str(tmp_path / "nope2.txt"), None)
    assert len(ids) == 0 and list(offs) == [0]


# `ifstream >> id >> x >> y >> z >> intensity` (pts_preprocess.h:36-47) is a TOKEN stream that ends at the first failed
# extraction; what fails is decided by libstdc++'s num_get grammar, not by strtod.  Each case replaces one line in the
# middle of the points file; the oracle parses with the real ifstream, the product with its sequential parser
# (PR_PARSE_THREADS=1) and with the chunked multi-threaded one (which must notice the odd line and defer to the former).
_ODD_LINES = ["{id} nan {y} {z} {it}", "{id} inf {y} {z} {it}", "{id} 0x10 {y} {z} {it}", "{id} +1.5 {y} {z} {it}",
              "{id} 1e {y} {z} {it}", "{id} 1e+ {y} {z} {it}", "{id} 1ex {y} {z} {it}", "{id} 1e400 {y} {z} {it}",
              "{id} 1e-400 {y} {z} {it}", "{id} 1.2.3 {y} {z} {it}", "{id} {x} {y} {z}", "{id} {x} {y} {z} {it} 7",
              "", "   \t ", "{id}.5 {x} {y} {z} {it}", "99999999999 {x} {y} {z} {it}", "{id} - {y} {z} {it}",
              "{id} 5. .5 -.5e+1 1E0", "{id} {x} {y} {z} 1e39", "{id} {x} {y} {z} {it}\r", "{id}\t{x}  {y}\v{z}\f{it}",
              "-{id} {x} {y} {z} {it}", "{id} {x},{y} {z} {it}", "# comment"]


@pytest.mark.parametrize("case", range(len(_ODD_LINES)))
def test_points_parser_follows_istream_token_rules(seq07, case, monkeypatch, tmp_path):
    poses, pts, _ = seq07
    lines = open(pts).read().split("\n")
    k = len(lines) // 2
    f = lines[k].split()
    lines[k] = _ODD_LINES[case].format(id=f[0], x=f[1], y=f[2], z=f[3], it=f[4])
    odd = str(tmp_path / "odd.txt")
    open(odd, "w").write("\n".join(lines))
    ox, oi, oo, oid = oracle_lib.pts_preprocess(poses, odd, None, 45.0, False)
    assert oo[-1] > 1000
    for env in ({"PR_PARSE_THREADS": "1"}, {"PR_PARSE_THREADS": "4", "PR_PARSE_MIN_BYTES": "0"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        px, pi, po, pid = api.pts_preprocess(poses, odd, None, 45.0, False)
        assert np.array_equal(po, oo) and np.array_equal(pid, oid), env
        assert np.array_equal(px.view(np.uint64), ox.view(np.uint64)), env
        assert np.array_equal(pi.view(np.uint32), oi.view(np.uint32)), env


def test_parallel_points_parser_same_clouds(seq07, monkeypatch):
    poses, pts, _ = seq07
    monkeypatch.setenv("PR_PARSE_THREADS", "1")
    a = api.pts_preprocess(poses, pts, None, 45.0, True)
    for th in ("2", "3", "8", "32"):
        monkeypatch.setenv("PR_PARSE_THREADS", th)
        monkeypatch.setenv("PR_PARSE_MIN_BYTES", "0")
        b = api.pts_preprocess(poses, pts, None, 45.0, True)
        assert all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b))


def test_signature_text_format_roundtrip(tmp_path):
    rng = np.random.default_rng(3)
    m = rng.normal(size=(5, 7)) * np.array([1, 10, 1e-3, 1e5, 1, 1, 1e-7])
    m[1, 2] = 0; m[2, 3] = -12345.678
    p = str(tmp_path / "sig.txt")
    api.write_signatures(p, m)
    txt = open(p).read()
    assert not txt.endswith("\n") and txt.count("\n") == 4                  # Eigen: no trailing newline
    lines = txt.split("\n")
    assert len({len(l) for l in lines}) == 1                                # aligned columns: equal line lengths
    w = max(len("%g" % v) for v in m.ravel())
    assert lines[0] == " ".join(("%g" % v).rjust(w) for v in m[0])
    back = api.read_signatures(p)
    assert back.shape == m.shape
    assert np.allclose(back, m, rtol=1e-5, atol=0)                          # 6 significant digits
    assert np.array_equal(np.loadtxt(p), back)


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 2400), (257, 192), (700, 2400)])
def test_signature_text_threads_do_not_change_a_byte(tmp_path, monkeypatch, rows, cols):
    rng = np.random.default_rng(rows)
    m = rng.normal(size=(rows, cols)) * 10.0 ** rng.integers(-8, 6, size=(rows, cols))
    m[0, 0] = np.nan; m[-1, -1] = -np.inf                                    # Eigen prints them, MATLAB load reads them
    files, backs = [], []
    for th in ("1", "2", "7", "32"):
        monkeypatch.setenv("PR_PARSE_THREADS", th)
        p = str(tmp_path / f"sig_{th}.txt")
        api.write_signatures(p, m)
        files.append(open(p, "rb").read())
        backs.append(api.read_signatures(p))
    assert all(f == files[0] for f in files)
    w = max(len("%g" % v) for v in m.ravel())
    assert files[0].decode().split("\n")[-1] == " ".join(("%g" % v).rjust(w) for v in m[-1])
    assert all(b.shape == (rows, cols) and np.array_equal(b, backs[0], equal_nan=True) for b in backs)
    assert np.array_equal(backs[0], np.array([[float("%g" % v) for v in r] for r in m]), equal_nan=True)


def test_signature_reader_stops_at_junk_like_one_thread(tmp_path, monkeypatch):
    rows = ["%d %d %d" % (3 * i, 3 * i + 1, 3 * i + 2) for i in range(400000)]      # 6 MB: one chunk per MB
    rows[1500] = "4500 oops 4502"
    p = str(tmp_path / "junk.txt"); open(p, "w").write("  \n" + " \n".join(rows) + "\n\n")
    res = []
    for th in ("1", "8"):
        monkeypatch.setenv("PR_PARSE_THREADS", th)
        try:
            res.append(api.read_signatures(p))
        except api.PRError as e:
            res.append(str(e))
    assert isinstance(res[0], str) and "ragged" in res[0] and res[1] == res[0]   # 4501 numbers before the junk: not k * 3
    rows[1500] = "4500 4501 4502 oops"
    open(p, "w").write("\n".join(rows))
    for th in ("1", "8"):
        monkeypatch.setenv("PR_PARSE_THREADS", th)
        b = api.read_signatures(p)
        assert b.shape == (1501, 3) and b[-1, -1] == 4502


def test_posespts_record_format(tmp_path):
    ids = np.array([3, 9], np.int32)
    w = np.arange(24, dtype=np.float64).reshape(2, 12) / 7
    p = str(tmp_path / "poses.txt")
    api.write_poses(p, ids, w)
    l0 = open(p).read().split("\n")[0]
    assert l0 == "3 " + "".join("%g " % v for v in w[0])                    # trailing space (PosesPts.h:14-22)
    q = str(tmp_path / "pts.txt")
    api.write_points(q, ids, np.array([[1.5, -2.25, 1e-9], [1 / 3, 2 / 3, 100.0]]), np.array([127.36, 0.5], np.float32))
    assert open(q).read().split("\n")[0] == "3 1.5 -2.25 1e-09 127.36"


def test_binary_sidecar_roundtrip(tmp_path):
    rng = np.random.default_rng(5)
    m = rng.normal(size=(7, 2400))
    p = str(tmp_path / "sig.bin")
    api.write_signatures(p, m)
    assert np.array_equal(api.read_signatures(p), m)                   # f64 on disk: bit exact
    assert os.path.getsize(p) == 32 + m.size * 8
    api.write_signatures(p, m, dtype=np.float32)
    assert np.array_equal(api.read_signatures(p), m.astype(np.float32).astype(np.float64))
    (tmp_path / "bad.bin").write_bytes(b"nope")
    with pytest.raises(api.PRError):
        api.read_signatures(str(tmp_path / "bad.bin"))


# ------------------------------------------------------------------------------------------------ f1 (GPU pre-stage)
@pytest.mark.parametrize("K", [0, 1, 2, 12, 13, 14, 29, 30, 59, 60, 61, 1000, 5003, 70000])
def test_hash_order_emulation_matches_libstdcxx(K):
    """csrc/hash_order.hpp (what the GPU pre-stage runs per cloud) vs the real std::unordered_map in the oracle: the
    iteration order for distinct keys in a given insertion order, across every rehash threshold."""
    rng = np.random.default_rng(K)
    for trial, hi in enumerate((K + 1, 65341, 450241)):
        keys = rng.permutation(max(hi, K))[:K].astype(np.int32)
        got = api.hash_order(keys)
        want = oracle_lib.unordered_order(keys)
        assert np.array_equal(got, want), (K, trial)
    seq = np.arange(K, dtype=np.int32)[::-1].copy()               # descending keys: collision chains in every bucket
    assert np.array_equal(api.hash_order(seq), oracle_lib.unordered_order(seq))


@pytest.mark.gpu
@pytest.mark.parametrize("polar", [False, True])
def test_gpu_prestage_equals_host_prestage(seq07, polar):
    poses, pts, d = seq07
    idh = str(d / f"ids_host_{polar}.txt"); idg = str(d / f"ids_gpu_{polar}.txt")
    hx, hi, ho, hid = api.pts_preprocess(poses, pts, idh, 45.0, polar)
    gx, gi, go, gid = api.pts_preprocess(poses, pts, idg, 45.0, polar, gpu=True)
    assert np.array_equal(go, ho) and np.array_equal(gid, hid)
    assert ho[-1] > 10000
    assert np.array_equal(gx.view(np.uint64), hx.view(np.uint64))  # same points, same ORDER, same bits
    assert np.array_equal(gi.view(np.uint32), hi.view(np.uint32))
    assert open(idg, "rb").read() == open(idh, "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("polar", [False, True])
def test_gpu_prestage_leaves_clouds_and_frames_in_hbm(seq07, polar):
    """pr_pts_preprocess_gpu keeps the emitted clouds in HBM with the PCA frames its gather pass adds up (pr_clouds_dev_*): the device copies
    are the host arrays, the frames are bit for bit those of a moments pass over them (pr_cloud_frames_dev), and pr_generate_clouds -
    binning pass only, straight from HBM - returns the signatures of the two-pass host-array calls."""
    import ctypes as C
    import torch
    from so_dso_place_recognition_amd import _lib
    poses, pts, d = seq07
    lib = _lib.load()
    ctx = api.Context(0)
    h = C.c_void_p()
    ctx.check(lib.pr_pts_preprocess_gpu(ctx.h, poses.encode(), pts.encode(), None, 45.0, int(polar), 0, C.byref(h)))
    try:
        N = lib.pr_clouds_count(h)
        offs = np.ctypeslib.as_array(lib.pr_clouds_offs(h), (N + 1,)).copy()
        T = int(offs[-1])
        xyz = np.ctypeslib.as_array(lib.pr_clouds_xyz(h), (T, 3)).copy()
        it = np.ctypeslib.as_array(lib.pr_clouds_inten(h), (T,)).copy()
        assert N > 50 and T > 10000 and lib.pr_clouds_dev_frames(h)
        def dev(ptr, shape, dtype):               # a torch view of library-owned device memory
            n = int(np.prod(shape))
            out = torch.empty(n, dtype=dtype, device="cuda")
            torch.cuda.synchronize()
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * out.element_size()), 3) == 0
            return out.reshape(shape)
        dx = dev(lib.pr_clouds_dev_xyz(h), (T, 3), torch.float64)
        di = dev(lib.pr_clouds_dev_inten(h), (T,), torch.float32)
        do = dev(lib.pr_clouds_dev_offs(h), (N + 1,), torch.int64)
        fr = dev(lib.pr_clouds_dev_frames(h), (N, 16), torch.float64)
        assert np.array_equal(dx.cpu().numpy().view(np.uint64), xyz.view(np.uint64)) and np.array_equal(do.cpu().numpy(), offs)
        assert np.array_equal(di.cpu().numpy().view(np.uint32), it.view(np.uint32))
        fr2 = torch.empty_like(fr)
        ctx.check(lib.pr_cloud_frames_dev(ctx.h, dx.data_ptr(), di.data_ptr(), do.data_ptr(), N, fr2.data_ptr()))    # moments pass + average chain
        ctx.sync()
        assert np.array_equal(fr.cpu().numpy().view(np.uint64), fr2.cpu().numpy().view(np.uint64))              # frames AND float averages
        for type_, gen, rows, cols in ((0, api.sc_generate, N, 2400), (1, api.m2dp_generate, 4 * N, 384), (2, api.delight_generate, 16 * N, 256)):
            got = np.empty((rows, cols))
            ctx.check(lib.pr_generate_clouds(ctx.h, type_, h, 45.0, got.ctypes.data))
            want = gen(xyz, it, offs, ctx=ctx) if type_ == 2 else gen(xyz, it, offs, 45.0, ctx)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), type_
    finally:
        lib.pr_clouds_free(h)
        ctx.close()


@pytest.mark.gpu
def test_gpu_prestage_resets_ranges_and_empty_inputs(golden_dir, tmp_path):
    """Two sequences back to back (a second reset in the middle), a shorter lidar range, unsorted point ids, and the
    degenerate inputs."""
    full = open(os.path.join(golden_dir, "kitti_seq07", "poses_history_file.txt")).read().split("\n")
    lines = [l for l in full[:90] if l.strip()]
    second = []
    for k, l in enumerate(lines[:70]):                             # same trajectory again with later ids: |t| < 1 -> reset
        t = l.split(); t[0] = str(int(lines[-1].split()[0]) + 1 + k); second.append(" ".join(t) + " ")
    poses = str(tmp_path / "poses.txt"); open(poses, "w").write("\n".join(lines + second) + "\n")
    pts = str(tmp_path / "pts.txt")
    helpers.write_synthetic_points(poses, pts, per_pose=50)
    rows = open(pts).read().strip().split("\n")
    rows[100], rows[4000] = rows[4000], rows[100]                  # an out-of-order id: the cursor waits behind it
    open(pts, "w").write("\n".join(rows) + "\n")
    for polar, rng_ in ((False, 45.0), (True, 45.0), (False, 20.0), (True, 12.5)):
        h = api.pts_preprocess(poses, pts, None, rng_, polar)
        g = api.pts_preprocess(poses, pts, None, rng_, polar, gpu=True)
        assert np.array_equal(g[2], h[2]) and np.array_equal(g[3], h[3])
        assert np.array_equal(g[0].view(np.uint64), h[0].view(np.uint64)) and np.array_equal(g[1].view(np.uint32), h[1].view(np.uint32))
    empty = str(tmp_path / "empty.txt"); open(empty, "w").write("")
    g = api.pts_preprocess(poses, empty, None, 45.0, False, gpu=True)
    h = api.pts_preprocess(poses, empty, None, 45.0, False)
    assert np.array_equal(g[2], h[2]) and np.array_equal(g[3], h[3]) and g[0].shape == (0, 3)
    g = api.pts_preprocess(empty, empty, None, 45.0, True, gpu=True)
    assert len(g[3]) == 0 and len(g[2]) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_SEQUENCES)
def test_gpu_prestage_reproduces_every_reference_id_file(ref_sequence, name, tmp_path):
    """The device pre-stage on the reference's own pose files (RobotCar: several resets per run) + synthetic points: the
    id file must be the reference's, byte for byte, and the clouds the host pre-stage's."""
    poses, ids_file = ref_sequence(name)
    pts = str(tmp_path / "pts.txt")
    helpers.write_synthetic_points(poses, pts, per_pose=12)
    out = str(tmp_path / "ids.txt")
    g = api.pts_preprocess(poses, pts, out, 45.0, False, gpu=True)
    assert open(out, "rb").read() == open(ids_file, "rb").read()
    h = api.pts_preprocess(poses, pts, None, 45.0, False)
    assert all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(g, h))
