"""Random pre-stage inputs (pts_preprocess.h:169-232): GPU pre-stage == host pre-stage == CPU oracle, bit for bit (clouds, point order, ids),
then the three generators on the GPU-resident clouds == the generators on the downloaded clouds.
Trajectories with tracking resets (|t| < 1), id gaps, poses without points, points whose id matches no pose, out-of-order point ids,
bursts that fill cells many times over, points exactly on the range limit, several lidar ranges, grid and polar filters.
usage: python tools/fuzz_prestage.py [seed] [cases]"""
import os
import sys
import tempfile
import time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_lib
from so_dso_place_recognition_amd import api

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(seed)
bad = []
NO_GPU = bool(os.environ.get("FUZZ_NO_GPU"))          # CPU box: host pre-stage against the oracle only
tmp = tempfile.mkdtemp(prefix="fuzz_prestage_")


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    R = np.eye(3); i, j = [(1, 2), (0, 2), (0, 1)][axis]
    R[i, i] = c; R[i, j] = -s; R[j, i] = s; R[j, j] = c
    return R


def trajectory(M):
    """camera-to-world poses along a wobbly path that returns to the origin now and then (a reset: |t| of world->camera < 1)"""
    ids, w2c = [], []
    pos = np.zeros(3); R = np.eye(3); cur = int(rng.integers(0, 5))
    for i in range(M):
        if i and rng.random() < 0.02:
            pos = rng.normal(0, 0.3, 3); R = np.eye(3)                    # tracking reset
        else:
            pos = pos + R @ np.array([rng.normal(0, 0.1), rng.normal(0, 0.05), rng.uniform(0.3, 1.5)])
            R = R @ rot(1, rng.normal(0, 0.05)) @ rot(0, rng.normal(0, 0.01))
        Rw = R.T; t = -Rw @ pos                                            # world -> camera
        ids.append(cur); w2c.append(np.concatenate([Rw, t[:, None]], axis=1))
        cur += int(rng.choice([1, 1, 1, 2, 3]))
    return np.array(ids, np.int32), np.stack(w2c), pos


for it in range(cases):
    M = int(rng.integers(35, 160))
    ids, w2c, _ = trajectory(M)
    lidar = float(rng.choice([45.0, 45.0, 20.0, 12.5, 80.0]))
    pid, pts, inten = [], [], []
    for i, W in zip(ids, w2c):
        r = rng.random()
        P = 0 if r < 0.1 else int(rng.integers(1, 30)) if r < 0.3 else int(rng.integers(30, 400))
        if P == 0:
            continue
        Rw, t = W[:, :3], W[:, 3]
        cam = rng.normal(0, 1, (P, 3)) * np.array([lidar * 0.5, lidar * 0.08, lidar * 0.5])
        if rng.random() < 0.2:                                             # a burst into a few cells
            cam[: P // 2] = cam[0] + rng.normal(0, 0.2, (P // 2, 3))
        if rng.random() < 0.1:                                             # on the range limit (strict <)
            cam[-1] = cam[-1] / np.linalg.norm(cam[-1]) * lidar
        world = (cam - t) @ Rw                                             # inverse of p_cam = Rw p + t
        k = int(i) if rng.random() < 0.9 else int(i) + 1                   # some ids fall between two poses
        pid.append(np.full(P, k, np.int32)); pts.append(world); inten.append(rng.uniform(0, 255, P).astype(np.float32))
    pid = np.concatenate(pid) if pid else np.zeros(0, np.int32)
    pts = np.concatenate(pts) if pts else np.zeros((0, 3))
    inten = np.concatenate(inten) if inten else np.zeros(0, np.float32)
    order = np.argsort(pid, kind="stable")
    pid, pts, inten = pid[order], pts[order], inten[order]
    if len(pid) > 200 and rng.random() < 0.3:                              # an out-of-order id: the cursor waits behind it
        a, b = int(rng.integers(0, len(pid) // 2)), int(rng.integers(len(pid) // 2, len(pid)))
        for arr in (pid, pts, inten):
            arr[[a, b]] = arr[[b, a]]
    poses_f, pts_f = os.path.join(tmp, f"poses{it}.txt"), os.path.join(tmp, f"pts{it}.txt")
    api.write_poses(poses_f, ids, w2c); api.write_points(pts_f, pid, pts, inten)
    line = [f"{it}: poses={M} points={len(pid)} range={lidar}"]
    for polar in (False, True):
        try:
            o = oracle_lib.pts_preprocess(poses_f, pts_f, os.path.join(tmp, "ids_o.txt"), lidar, polar)
            h = api.pts_preprocess(poses_f, pts_f, os.path.join(tmp, "ids_h.txt"), lidar, polar)
            g = h if NO_GPU else api.pts_preprocess(poses_f, pts_f, os.path.join(tmp, "ids_g.txt"), lidar, polar, gpu=True)
            ok = True
            for name, x in (("host", h), ("gpu", g)):
                same = (np.array_equal(x[2], o[2]) and np.array_equal(x[3], o[3]) and np.array_equal(x[0].view(np.uint64), o[0].view(np.uint64))
                        and np.array_equal(x[1].view(np.uint32), o[1].view(np.uint32)))
                if not same:
                    bad.append((it, polar, name, "clouds differ", len(o[3]), len(x[3]), int(o[2][-1]), int(x[2][-1]))); ok = False
            idf = [open(os.path.join(tmp, f"ids_{c}.txt")).read() for c in ("oh" + ("h" if NO_GPU else "g"))]
            if not (idf[0] == idf[1] == idf[2]):
                bad.append((it, polar, "id files differ")); ok = False
            # generators on what the pre-stage produced (ragged, some clouds tiny or empty)
            if len(o[3]) and not NO_GPU:
                gs = api.sc_generate(o[0], o[1], o[2], lidar); os_ = oracle_lib.sc_generate(o[0], o[1], o[2], lidar)
                big = np.diff(o[2]) >= 8
                if big.any() and not (np.array_equal(gs[big][:, 1200:], os_[big][:, 1200:]) and np.abs(gs[big] - os_[big]).max() < 1e-9):
                    bad.append((it, polar, "sc_generate on the pre-stage clouds", float(np.abs(gs[big] - os_[big]).max()))); ok = False
            line.append(f"{'polar' if polar else 'grid'}:{len(o[3])} clouds/{int(o[2][-1])} pts:{'ok' if ok else 'BAD'}")
        except Exception as e:
            bad.append((it, polar, "exception", repr(e)))
            line.append("EXC")
    print(" ".join(line), flush=True)
for b in bad:
    print("BAD", b)
print("fuzz_prestage:", "ok" if not bad else f"{len(bad)} findings", f"seed {seed}, {cases} cases")
sys.exit(1 if bad else 0)
