"""Deterministic synthetic inputs for tests and bench.

libm-free: only +, -, *, compares and IEEE sqrt are used, so any re-implementation
(C++/numpy) produces identical bits.  RNG: counter-based splitmix64,
    base(seed, stream) = splitmix64(seed ^ splitmix64(stream))
    U(seed, stream, i) = (splitmix64(base + i) >> 11) * 2**-53        in [0, 1)
There is no reference counterpart: the reference ships no synthetic generator and its real
point files are missing blobs (.MISSING_LARGE_BLOBS:2-14).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _base(seed: int, stream) -> np.ndarray:
    return splitmix64(np.uint64(seed) ^ splitmix64(np.asarray(stream, dtype=np.uint64)))


def uniform(seed: int, stream, count: int, offset: int = 0) -> np.ndarray:
    """U[0,1) doubles.  `stream` scalar -> shape (count,); array of S streams -> shape (S, count)."""
    ctr = np.arange(offset, offset + count, dtype=np.uint64)
    b = _base(seed, stream)
    with np.errstate(over="ignore"):
        h = splitmix64(b[..., None] + ctr) if np.ndim(b) else splitmix64(b + ctr)
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


# ----------------------------------------------------------------------------- point clouds
def scene_cloud(seed: int, cloud: int, P: int, max_rho: float = 45.0):
    """Scene sampler of §8-d config 2: 32 boxes + 30 % ground plane, exactly P accepted points with
    ||p|| < max_rho.  Returns (xyz float64 [P,3] camera frame, intensity float32 [P])."""
    bx = uniform(seed, cloud * 4 + 0, 32 * 7).reshape(32, 7)
    centre = np.stack([-38 + 76 * bx[:, 0], -4 + 5 * bx[:, 1], -24 + 48 * bx[:, 2]], 1)
    half = np.stack([0.5 + 3.5 * bx[:, 3], 0.5 + 2.5 * bx[:, 4], 0.5 + 3.5 * bx[:, 5]], 1)
    base_i = 20 + 215 * bx[:, 6]
    out_xyz = np.empty((0, 3)); out_i = np.empty((0,))
    off = 0
    while out_xyz.shape[0] < P:
        n = int((P - out_xyz.shape[0]) * 1.3) + 64
        u = uniform(seed, cloud * 4 + 1, n * 6, off * 6).reshape(n, 6)
        off += n
        ground = u[:, 0] < 0.3
        box = np.minimum((u[:, 1] * 32).astype(np.int64), 31)
        pg = np.stack([-42 + 84 * u[:, 2], 1.6 + (-0.05 + 0.1 * u[:, 3]), -28 + 56 * u[:, 4]], 1)
        pb = centre[box] + half[box] * (2 * u[:, 2:5] - 1)
        p = np.where(ground[:, None], pg, pb)
        it = np.where(ground, 60.0, base_i[box]) + (-20 + 40 * u[:, 5])
        ok = (p * p).sum(1) < max_rho * max_rho
        out_xyz = np.concatenate([out_xyz, p[ok]]); out_i = np.concatenate([out_i, it[ok]])
    return np.ascontiguousarray(out_xyz[:P]), out_i[:P].astype(np.float32)


def scene_clouds(seed: int, N: int, P: int, max_rho: float = 45.0, first: int = 0):
    """N clouds in CSR layout: (xyz [N*P,3] f64, inten [N*P] f32, offs [N+1] i64)."""
    xyz = np.empty((N * P, 3)); it = np.empty((N * P,), np.float32)
    for c in range(N):
        xyz[c * P:(c + 1) * P], it[c * P:(c + 1) * P] = scene_cloud(seed, first + c, P, max_rho)
    return xyz, it, np.arange(N + 1, dtype=np.int64) * P


# ----------------------------------------------------------------------------- SC signatures
def sc_database(seed: int, n: int, first: int = 0, chunk: int = 4096) -> np.ndarray:
    """SC signature sampler (§8-d metric config): [n, 2400] float64, row = [structure 1200 | intensity 1200],
    bin = sector*20 + ring.  Structure bin occupied w.p. 0.6 with height U[0,8]; intensity bit
    Bernoulli(0.45) on occupied bins."""
    out = np.empty((n, 2400))
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        u = uniform(seed, np.arange(first + s, first + e, dtype=np.uint64), 3600).reshape(e - s, 3, 1200)
        occ = u[:, 0] < 0.6
        out[s:e, :1200] = np.where(occ, 8.0 * u[:, 1], 0.0)
        out[s:e, 1200:] = np.where(occ & (u[:, 2] < 0.45), 1.0, 0.0)
    return out


def sc_queries(seed: int, db: np.ndarray, m: int, db_first: int = 0, n_global: int | None = None,
               db_seed: int | None = None):
    """Queries planted on DB entries: query t copies entry e_t = floor(U*n), rotated by floor(U*60) sectors,
    mirrored w.p. 1/2, 3 % of bins re-drawn.  Returns (queries [m,2400], planted index [m] int64).
    `db` may be a shard starting at global row db_first of an n_global-row DB; only queries whose planted
    entry lies inside the shard are copied from it (the others are re-drawn from sc_database(db_seed), default
    db_seed = seed - 1)."""
    db_seed = seed - 1 if db_seed is None else db_seed
    n = db.shape[0] if n_global is None else n_global
    u = uniform(seed, np.arange(m, dtype=np.uint64), 3 + 3600)
    et = np.minimum((u[:, 0] * n).astype(np.int64), n - 1)
    shift = np.minimum((u[:, 1] * 60).astype(np.int64), 59)
    mirror = u[:, 2] < 0.5
    q = np.empty((m, 2400))
    for t in range(m):
        g = et[t] - db_first
        src = db[g] if 0 <= g < db.shape[0] else sc_database(db_seed, 1, first=int(et[t]))[0]
        for ch in range(2):
            img = src[ch * 1200:(ch + 1) * 1200].reshape(60, 20)
            if mirror[t]:
                img = img[(-np.arange(60)) % 60]
            q[t, ch * 1200:(ch + 1) * 1200] = np.roll(img, shift[t], axis=0).reshape(-1)
        r = u[t, 3:].reshape(3, 1200)
        redo = r[0] < 0.03
        occ = r[1] < 0.6
        q[t, :1200] = np.where(redo, np.where(occ, 8.0 * r[2], 0.0), q[t, :1200])
        q[t, 1200:] = np.where(redo, np.where(occ & (r[2] < 0.45), 1.0, 0.0), q[t, 1200:])
    return q, et


# ----------------------------------------------------------------------------- M2DP signatures
def m2dp_database(seed: int, n: int, first: int = 0, chunk: int = 4096) -> np.ndarray:
    """M2DP signature sampler (§8-d config 3): [4n, 384] float64; per entry/variant/channel
    w = 0.3 + U(entry) + 0.15*(U(variant)-0.5); row = [u|v | u|v] with u = w[:64]/||.||, v = w[64:]/||.||."""
    out = np.empty((4 * n, 384))
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        u = uniform(seed, np.arange(first + s, first + e, dtype=np.uint64), 384 * 5).reshape(e - s, 5, 2, 192)
        w = 0.3 + u[:, :1] + 0.15 * (u[:, 1:] - 0.5)                      # [e, 4 variants, 2 ch, 192]
        uu = w[..., :64] / np.sqrt((w[..., :64] ** 2).sum(-1, keepdims=True))
        vv = w[..., 64:] / np.sqrt((w[..., 64:] ** 2).sum(-1, keepdims=True))
        out[4 * s:4 * e] = np.concatenate([uu, vv], -1).reshape(4 * (e - s), 384)
    return out


def m2dp_queries(seed: int, db: np.ndarray, m: int):
    """Query t copies DB entry e_t (all 4 variant rows), adds 0.05*(U-0.5) per element, renormalises u and v."""
    n = db.shape[0] // 4
    u = uniform(seed, np.arange(m, dtype=np.uint64), 1 + 4 * 384)
    et = np.minimum((u[:, 0] * n).astype(np.int64), n - 1)
    rows = db.reshape(n, 4, 2, 192)[et] + 0.05 * (u[:, 1:].reshape(m, 4, 2, 192) - 0.5)
    uu = rows[..., :64] / np.sqrt((rows[..., :64] ** 2).sum(-1, keepdims=True))
    vv = rows[..., 64:] / np.sqrt((rows[..., 64:] ** 2).sum(-1, keepdims=True))
    return np.concatenate([uu, vv], -1).reshape(4 * m, 384), et


# ----------------------------------------------------------------------------- DELIGHT histograms
def delight_database(seed: int, n: int, first: int = 0) -> np.ndarray:
    """[16n, 256] float64 histograms: per entry 16 x 256 Poisson-like integer counts around a smooth profile."""
    u = uniform(seed, np.arange(first, first + n, dtype=np.uint64), 4096 + 16).reshape(n, 4096 + 16)
    scale = 4.0 + 28.0 * u[:, 4096:]                                          # per-histogram level
    cnt = np.floor(u[:, :4096].reshape(n, 16, 256) * scale[:, :, None] * (u[:, :4096].reshape(n, 16, 256) < 0.5))
    return cnt.reshape(16 * n, 256)


def delight_queries(seed: int, db: np.ndarray, m: int):
    """Query t copies entry e_t, applies one of the 4 octant permutations of processDELIGHT.m and perturbs 2 % of the bins."""
    from numpy import array
    mut = array([[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [5, 4, 7, 6, 1, 0, 3, 2, 13, 12, 15, 14, 9, 8, 11, 10],
                 [6, 7, 4, 5, 2, 3, 0, 1, 14, 15, 12, 13, 10, 11, 8, 9], [3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12]])
    n = db.shape[0] // 16
    u = uniform(seed, np.arange(m, dtype=np.uint64), 2 + 2 * 4096)
    et = np.minimum((u[:, 0] * n).astype(np.int64), n - 1)
    k = np.minimum((u[:, 1] * 4).astype(np.int64), 3)
    q = np.empty((m, 16, 256))
    for t in range(m):
        src = db.reshape(n, 16, 256)[et[t]][mut[k[t]]]
        r = u[t, 2:].reshape(2, 16, 256)
        q[t] = np.where(r[0] < 0.02, np.floor(r[1] * 8), src)
    return q.reshape(16 * m, 256), et


def bow_signatures(seed: int, n: int, cols: int = 120, vocab: int = 400, fill=(20, 90)):
    """[2 n][cols] BoW rows as test_bow.cpp:147-162 writes them: sorted word ids / L1-normalised weights, both padded with -1."""
    rng = np.random.default_rng(seed)
    out = -np.ones((2 * n, cols))
    for i in range(n):
        k = int(rng.integers(fill[0], min(fill[1], cols) + 1))
        ids = np.sort(rng.choice(vocab, size=k, replace=False))
        w = rng.random(k) + 0.05
        out[2 * i, :k] = ids
        out[2 * i + 1, :k] = w / w.sum()
    return out


def gist_signatures(seed: int, n: int, cols: int = 96):
    rng = np.random.default_rng(seed)
    return np.abs(rng.normal(0.1, 0.05, size=(n, cols)))


# ----------------------------------------------------------------------------- the same samplers on a torch device
# Bit-identical to the numpy versions above (uint64 arithmetic carried in int64 two's complement: add / multiply wrap,
# logical shifts by masking), so that full-size inputs (10^5 - 10^6 signatures, 5000 x 50 000 points) are drawn in HBM in
# a fraction of a second instead of minutes of host time.  tests/test_synth_torch.py pins them against the numpy versions.
def _s64(x: int) -> int:
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


def _t_lsr(z, s: int):
    return (z >> s) & ((1 << (64 - s)) - 1)


def _t_splitmix64(x):
    z = x + _s64(0x9E3779B97F4A7C15)
    z = (z ^ _t_lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _t_lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _t_lsr(z, 31)


def uniform_torch(seed: int, streams, count: int, offset: int = 0):
    """streams: int64 tensor [S] (on the target device) -> U[0,1) float64 [S, count], == uniform(seed, streams, count, offset)."""
    import torch
    b = _t_splitmix64(_t_splitmix64(streams) ^ _s64(seed))
    ctr = torch.arange(offset, offset + count, dtype=torch.int64, device=streams.device)
    h = _t_splitmix64(b[:, None] + ctr[None, :])
    return _t_lsr(h, 11).to(torch.float64) * (2.0 ** -53)


def sc_database_torch(seed: int, n: int, first: int = 0, device="cuda", chunk: int = 8192):
    """== sc_database(seed, n, first) as a float64 tensor on `device`."""
    import torch
    out = torch.empty((n, 2400), dtype=torch.float64, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        u = uniform_torch(seed, torch.arange(first + s, first + e, dtype=torch.int64, device=device), 3600).reshape(e - s, 3, 1200)
        occ = u[:, 0] < 0.6
        out[s:e, :1200] = torch.where(occ, 8.0 * u[:, 1], torch.zeros_like(u[:, 1]))
        out[s:e, 1200:] = (occ & (u[:, 2] < 0.45)).to(torch.float64)
    return out


def m2dp_database_torch(seed: int, n: int, first: int = 0, device="cuda", chunk: int = 8192):
    """== m2dp_database(seed, n, first) up to the last ulp of the row norms (torch reduces the 64 / 128 squares in another
    order than numpy); float64 [4n, 384] on `device`."""
    import torch
    out = torch.empty((4 * n, 384), dtype=torch.float64, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        u = uniform_torch(seed, torch.arange(first + s, first + e, dtype=torch.int64, device=device), 384 * 5).reshape(e - s, 5, 2, 192)
        w = 0.3 + u[:, :1] + 0.15 * (u[:, 1:] - 0.5)
        uu = w[..., :64] / torch.sqrt((w[..., :64] ** 2).sum(-1, keepdim=True))
        vv = w[..., 64:] / torch.sqrt((w[..., 64:] ** 2).sum(-1, keepdim=True))
        out[4 * s:4 * e] = torch.cat([uu, vv], -1).reshape(4 * (e - s), 384)
    return out


def scene_clouds_torch(seed: int, N: int, P: int, max_rho: float = 45.0, first: int = 0, device="cuda", chunk: int = 64):
    """== scene_clouds(seed, N, P, max_rho, first) as device tensors (xyz f64 [N*P,3], inten f32 [N*P], offs i64 [N+1]):
    the first P accepted points of the first int(1.3 P) + 64 draws of every cloud (clouds whose first round accepts fewer
    than P points - none with the §8-d scene parameters - are drawn by the numpy sampler)."""
    import torch
    n = int(P * 1.3) + 64
    xyz = torch.empty((N * P, 3), dtype=torch.float64, device=device)
    it = torch.empty((N * P,), dtype=torch.float32, device=device)
    for c0 in range(0, N, chunk):
        c1 = min(N, c0 + chunk)
        C_ = c1 - c0
        cl = torch.arange(first + c0, first + c1, dtype=torch.int64, device=device)
        bx = uniform_torch(seed, cl * 4 + 0, 32 * 7).reshape(C_, 32, 7)
        centre = torch.stack([-38 + 76 * bx[..., 0], -4 + 5 * bx[..., 1], -24 + 48 * bx[..., 2]], -1)
        half = torch.stack([0.5 + 3.5 * bx[..., 3], 0.5 + 2.5 * bx[..., 4], 0.5 + 3.5 * bx[..., 5]], -1)
        base_i = 20 + 215 * bx[..., 6]
        u = uniform_torch(seed, cl * 4 + 1, n * 6).reshape(C_, n, 6)
        ground = u[..., 0] < 0.3
        box = torch.clamp((u[..., 1] * 32).to(torch.int64), max=31)
        pg = torch.stack([-42 + 84 * u[..., 2], 1.6 + (-0.05 + 0.1 * u[..., 3]), -28 + 56 * u[..., 4]], -1)
        gi = box[..., None].expand(-1, -1, 3)
        pb = torch.gather(centre, 1, gi) + torch.gather(half, 1, gi) * (2 * u[..., 2:5] - 1)
        p = torch.where(ground[..., None], pg, pb)
        inten = torch.where(ground, torch.full_like(u[..., 5], 60.0), torch.gather(base_i, 1, box)) + (-20 + 40 * u[..., 5])
        ok = ((p[..., 0] * p[..., 0] + p[..., 1] * p[..., 1]) + p[..., 2] * p[..., 2]) < max_rho * max_rho
        rank = torch.cumsum(ok.to(torch.int64), 1) - 1
        have = rank[:, -1] + 1
        take = ok & (rank < P)
        ci, pi = torch.nonzero(take, as_tuple=True)
        dst = (c0 + ci) * P + rank[ci, pi]
        full = have[ci] >= P
        xyz[dst[full]] = p[ci[full], pi[full]]
        it[dst[full]] = inten[ci[full], pi[full]].to(torch.float32)
        for c in torch.nonzero(have < P).flatten().tolist():          # not seen with the §8-d parameters
            a, b = scene_cloud(seed, first + c0 + c, P, max_rho)
            xyz[(c0 + c) * P:(c0 + c + 1) * P] = torch.from_numpy(a).to(device)
            it[(c0 + c) * P:(c0 + c + 1) * P] = torch.from_numpy(b).to(device)
    return xyz, it, torch.arange(N + 1, dtype=torch.int64, device=device) * P


def drive_clouds_torch(frames=2000, per_cloud=6000, seed=5, stops=(), device="cuda", positions=False):
    """Two laps of a closed circuit through a static world: consecutive clouds overlap almost completely, frame i and
    frame i + frames/2 see the same place from slightly different poses.  Returns CSR clouds in the camera frame
    (x right, y down, z forward) and the lap length.  stops: (first frame, frames) stretches where the vehicle stands still - the pose
    of `first frame` is kept for that many frames (the frame-dependent subsample still differs: near-copies, the clusters the matcher's
    containment check exists for); the laps are then measured in MOVING frames.  positions: also return the sensor positions [frames, 3]
    (the ground truth of run_test.m:62-85).  (tests/test_gpu_configs.py::test_drive_2000_frames_mask_100 and bench.py's
    extra.kitti_shape share this sampler.)"""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    R0 = 160.0                                                        # circuit radius [m]
    # world: ground strip + boxes along the circuit (y down: ground at y = +1.6)
    nb = 900
    ang = torch.rand(nb, generator=g, device=device, dtype=torch.float64) * 2 * np.pi
    rad = R0 + (torch.rand(nb, generator=g, device=device, dtype=torch.float64) - 0.5) * 70
    cx, cz = rad * torch.cos(ang), rad * torch.sin(ang)
    half = 0.5 + 3.5 * torch.rand((nb, 3), generator=g, device=device, dtype=torch.float64)
    base_i = 20 + 215 * torch.rand(nb, generator=g, device=device, dtype=torch.float64)
    ppb = 700
    u = torch.rand((nb, ppb, 3), generator=g, device=device, dtype=torch.float64) * 2 - 1
    bx = torch.stack([cx[:, None] + half[:, None, 0] * u[..., 0], (1.6 - half[:, None, 1]) + half[:, None, 1] * u[..., 1] * 0.999,
                      cz[:, None] + half[:, None, 2] * u[..., 2]], -1).reshape(-1, 3)
    bi = (base_i[:, None] + 40 * (torch.rand((nb, ppb), generator=g, device=device, dtype=torch.float64) - 0.5)).reshape(-1)
    ng = 500_000
    ga = torch.rand(ng, generator=g, device=device, dtype=torch.float64) * 2 * np.pi
    gr = R0 + (torch.rand(ng, generator=g, device=device, dtype=torch.float64) - 0.5) * 100
    gx = torch.stack([gr * torch.cos(ga), 1.6 + 0.1 * (torch.rand(ng, generator=g, device=device, dtype=torch.float64) - 0.5), gr * torch.sin(ga)], -1)
    gi_ = 60 + 40 * (torch.rand(ng, generator=g, device=device, dtype=torch.float64) - 0.5)
    W = torch.cat([bx, gx]); WI = torch.cat([bi, gi_])
    still = np.zeros(frames, bool)
    for f0, cnt in stops:
        still[f0 + 1: f0 + cnt] = True                                                # frames that repeat their predecessor's pose
    step = np.cumsum(~still) - 1                                                      # the pose index of every frame (frame 0 moves)
    moving = int((~still).sum())
    lap = moving // 2
    xyz, inten, offs, poss = [], [], [0], []
    for f in range(frames):
        s = int(step[f])
        th = 2 * np.pi * (s % lap) / lap + (0.002 if s >= lap else 0.0)              # second lap: 0.3 m along-track offset
        r = R0 + (0.4 if s >= lap else 0.0)                                           # ... and 0.4 m lateral
        pos = torch.tensor([r * np.cos(th), 0.0, r * np.sin(th)], dtype=torch.float64, device=device)
        fwd = torch.tensor([-np.sin(th), 0.0, np.cos(th)], dtype=torch.float64, device=device)
        right = torch.tensor([np.cos(th), 0.0, np.sin(th)], dtype=torch.float64, device=device)
        poss.append([r * np.cos(th), 0.0, r * np.sin(th)])
        rel = W - pos
        a, b = rel @ right, rel @ fwd
        near = (a / 26.0) ** 2 + (b / 44.0) ** 2 < 1.0                               # a road corridor: keeps the three PCA eigenvalues apart (N3)
        it = WI[near]
        cam = torch.stack([a[near], rel[near, 1], b[near]], 1)
        if cam.shape[0] > per_cloud:                                                  # frame-dependent subsample (a moving sensor never
            sel = torch.randperm(cam.shape[0], generator=g, device=device)[:per_cloud]   # sees the same points twice)
            cam, it = cam[sel], it[sel]
        xyz.append(cam); inten.append(it.to(torch.float32)); offs.append(offs[-1] + cam.shape[0])
    res = (torch.cat(xyz).cpu().numpy(), torch.cat(inten).cpu().numpy(), np.array(offs, np.int64), lap)
    return res + (np.array(poss),) if positions else res
