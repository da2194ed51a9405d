"""Independent numpy/LAPACK restatement of the hot path (tests only).

Written against the reference files directly, with numpy.linalg.eigh / svd in place of Eigen and
array broadcasting in place of the loops, so that it shares no code with oracle/pr_ref.cpp.
"""
import os

import numpy as np

_TABLE_H = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pr_m2dp_table.h")


def canonical_frame(V):
    """largest-|component| of v0, v1 positive; v2 flipped so det = +1."""
    V = V.copy()
    for j in range(2):
        im = np.argmax(np.abs(V[:, j]))
        if V[im, j] < 0:
            V[:, j] = -V[:, j]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return V


def align_pca(xyz):
    """pts_align.h:7-46"""
    c = xyz - xyz.mean(0)
    w, V = np.linalg.eigh(c.T @ c)           # ascending
    V = canonical_frame(V)
    return c @ V, V


def ave_f32(inten):
    """SC.cpp:60-64 float sequential mean (np.cumsum on float32 is sequential)."""
    inten = np.asarray(inten, np.float32)
    s = np.float32(0) if inten.size == 0 else np.cumsum(inten, dtype=np.float32)[-1]
    return np.float32(s / np.float32(inten.size))


def sc_signature(xyz, inten, max_rho=45.0):
    """SC.cpp:12-76 (bin = sector*20 + ring; drop iff idx >= 1200)."""
    al, _ = align_pca(xyz)
    y, z = al[:, 1], al[:, 2]
    si = np.floor((np.arctan2(z, y) + np.pi) * (60 / (2.0 * np.pi))).astype(np.int64)
    ri = np.floor(np.sqrt(y * y + z * z) * (20 / max_rho)).astype(np.int64)
    idx = si * 20 + ri
    ok = idx < 1200
    idx, x, it = idx[ok], al[ok, 0], np.asarray(inten, np.float32)[ok].astype(np.float64)
    cnt = np.bincount(idx, minlength=1200)
    s = np.bincount(idx, weights=it, minlength=1200)
    lo = np.full(1200, np.inf)
    hi = np.full(1200, -np.inf)
    np.minimum.at(lo, idx, x)
    np.maximum.at(hi, idx, x)
    occ = cnt > 0
    struct = np.where(occ, hi - lo, 0.0)
    ave = np.float64(ave_f32(inten))
    inten_out = np.where(occ, (s / np.maximum(cnt, 1) > ave).astype(np.float64), 0.0)
    return np.concatenate([struct, inten_out])


def plane_table():
    """M2DP.cpp:9-30 from the frozen float normals."""
    rows = []
    for line in open(_TABLE_H):
        line = line.strip()
        if line.startswith("{0x"):
            rows.append([int(t.strip().rstrip("u"), 16) for t in line[1:line.index("}")].split(",")])
    n = np.array(rows, np.uint32).view(np.float32).astype(np.float64)   # [64,3]
    xa = np.array([1.0, 0.0, 0.0])
    xp = xa[None] - n[:, :1] * n
    yp = np.cross(n, xp)
    return xp, yp


def _dot3(a, p):
    return a[0] * p[:, 0] + (a[1] * p[:, 1] + a[2] * p[:, 2])


def m2dp_matrices(aligned, inten, max_rho, dx, dy):
    """M2DP.cpp:47-91 for variant (dx,dy) of test_m2dp.cpp:47-57."""
    xp_t, yp_t = plane_table()
    p = aligned * np.array([dx, dy, dx * dy], np.float64)
    it = np.asarray(inten, np.float32).astype(np.float64)
    cm = np.zeros((64, 128))
    im = np.zeros((64, 128))
    for k in range(64):
        xp = _dot3(xp_t[k], p)
        yp = _dot3(yp_t[k], p)
        si = np.floor((np.arctan2(yp, xp) + np.pi) * (16 / (2.0 * np.pi))).astype(np.int64)
        ri = np.floor(np.sqrt(xp * xp + yp * yp) * (8 / max_rho)).astype(np.int64)
        idx = ri * 16 + si
        ok = idx < 128
        cm[k] = np.bincount(idx[ok], minlength=128)
        im[k] = np.bincount(idx[ok], weights=it[ok], minlength=128)
    ave = np.float64(ave_f32(inten))
    occ = cm > 0
    im = np.where(occ, (im / np.maximum(cm, 1) > ave).astype(np.float64), 0.0)
    return cm, im


def top_pair(A):
    """JacobiSVD U.col(0), V.col(0) (M2DP.cpp:94-103) via LAPACK, sign: sum(u1) >= 0 (N6)."""
    U, s, Vt = np.linalg.svd(A, full_matrices=False)
    u, v = U[:, 0], Vt[0]
    if u.sum() < 0:
        u, v = -u, -v
    return np.concatenate([u, v])


def m2dp_signature(xyz, inten, max_rho=45.0):
    """test_m2dp.cpp:41-68: 4 rows x 384."""
    al, _ = align_pca(xyz)
    rows = []
    for dx in (-1, 1):
        for dy in (-1, 1):
            cm, im = m2dp_matrices(al, inten, max_rho, dx, dy)
            rows.append(np.concatenate([top_pair(cm), top_pair(im)]))
    return np.stack(rows)


def sc_distance_channel(h1, h2):
    """processSC.m:12-34 by explicit variants (fp64)."""
    a = h1 / np.linalg.norm(h1, axis=1, keepdims=True)
    b = h2 / np.linalg.norm(h2, axis=1, keepdims=True)
    m = a.shape[0]
    res = np.empty((m, b.shape[0]))
    cols = np.arange(60)
    for i in range(m):
        img = a[i].reshape(60, 20)                                        # [sector, ring]
        fw = np.stack([img[(k + cols) % 60].reshape(-1) for k in range(60)])
        mi = np.stack([img[(k - cols) % 60].reshape(-1) for k in range(60)])
        sig = np.concatenate([fw, mi])
        res[i] = ((1 - sig @ b.T) / 2).min(0)
    return res


def sc_distance(h1, h2):
    return sc_distance_channel(h1[:, :1200], h2[:, :1200]), sc_distance_channel(h1[:, 1200:], h2[:, 1200:])


def m2dp_distance(h1, h2):
    """processM2DP.m:12-22"""
    out = []
    for ch in range(2):
        a = h1[:, ch * 192:(ch + 1) * 192]
        b = h2[:, ch * 192:(ch + 1) * 192]
        full = (1 - a @ b.T) / 2
        m, n = a.shape[0] // 4, b.shape[0] // 4
        out.append(full.reshape(m, 4, n, 4).min(axis=(1, 3)))
    return out


def fuse_top1(dp, di, mask_width, p_weight=2.0):
    """run_test.m:38-57"""
    def z(d):
        return (d - d.mean(1, keepdims=True)) / d.std(1, ddof=1, keepdims=True)
    f = p_weight * z(dp) + z(di)
    i, j = np.indices(f.shape)
    f = np.where(np.abs(i - j) < mask_width, np.inf, f)
    return f.argmin(1), f.min(1), f


def incoming_ids(poses_file):
    """pts_preprocess.h:187-215 on the pose file alone (the reference-pinned KAT)."""
    ids = []
    frame = 0
    for line in open(poses_file):
        t = line.split()
        if len(t) < 13:
            continue
        w = np.array(t[1:13], np.float64).reshape(3, 4)
        if np.linalg.norm(w[:, 3]) < 1.0:
            frame = 0
        if frame < 30:
            frame += 1
            continue
        ids.append(int(t[0]))
    return ids


DELIGHT_MUT = np.array([[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16],
                        [6, 5, 8, 7, 2, 1, 4, 3, 14, 13, 16, 15, 10, 9, 12, 11],
                        [7, 8, 5, 6, 3, 4, 1, 2, 15, 16, 13, 14, 11, 12, 9, 10],
                        [4, 3, 2, 1, 8, 7, 6, 5, 12, 11, 10, 9, 16, 15, 14, 13]]) - 1   # processDELIGHT.m:2-5


def delight_signature(xyz, inten):
    """DELIGHT.cpp:8-24"""
    al, _ = align_pca(xyz)
    f = al.astype(np.float32)
    d = np.sqrt((al * al).sum(1)).astype(np.float32)
    hist = 8 * (d.astype(np.float64) > 10.0) + 4 * (f[:, 2] > 0) + 2 * (f[:, 1] > 0) + 1 * (f[:, 0] > 0)
    b = np.asarray(inten, np.float32).astype(np.int64)
    ok = (b >= 0) & (b < 256)
    out = np.zeros((16, 256))
    np.add.at(out, (hist[ok], b[ok]), 1)
    return out


def delight_distance(h1, h2):
    """processDELIGHT.m:7-37"""
    m, n = h1.shape[0] // 16, h2.shape[0] // 16
    A = h1.reshape(m, 16, 256)
    B = h2.reshape(n, 16, 256)
    out = np.empty((m, n))
    for i in range(m):
        best = np.full(n, np.inf)
        for k in range(4):
            Bk = B[:, DELIGHT_MUT[k], :]
            s = A[i][None] + Bk
            with np.errstate(divide="ignore", invalid="ignore"):
                t = np.where(s > 0, 2 * (A[i][None] - Bk) ** 2 / s, 0.0).sum((1, 2)) / (s > 0).sum((1, 2))
            best = np.where(best > t, t, best)
        out[i] = best
    return out


def gist_distance(h1, h2):
    """processGIST.m:1-10"""
    h1 = np.asarray(h1, np.float64); h2 = np.asarray(h2, np.float64)
    return ((h1[:, None, :] - h2[None, :, :]) ** 2).sum(2)


def bow_distance(h1, h2):
    """processBoW.m:1-38, literally (1-based cursors; the strict `<` of the guards skips the last column)"""
    h1 = np.asarray(h1, np.float64); h2 = np.asarray(h2, np.float64)
    i1s, v1s, i2s, v2s = h1[0::2], h1[1::2], h2[0::2], h2[1::2]
    out = np.ones((i1s.shape[0], i2s.shape[0]))
    for i in range(i1s.shape[0]):
        for j in range(i2s.shape[0]):
            i1, v1, i2, v2 = i1s[i], v1s[i], i2s[j], v2s[j]
            a = b = 1
            score = 0.0
            while a < len(i1) and i1[a - 1] > -1 and b < len(i2) and i2[b - 1] > -1:
                if i1[a - 1] == i2[b - 1]:
                    score = score + abs(v1[a - 1] - v2[b - 1]) - abs(v1[a - 1]) - abs(v2[b - 1])
                    a += 1; b += 1
                elif i1[a - 1] < i2[b - 1]:
                    a += 1
                else:
                    b += 1
            out[i, j] = 1 - (-score / 2.0)
    return out
