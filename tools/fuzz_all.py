"""Wide random parity sweep against the CPU oracle (GPU box): every matcher entry point (host calls, pr_group with virtual shards, the
device-resident Matcher / FusedMatcher), the three SC arithmetics, k up to 60, wide masks, zero-norm rows, duplicated rows (exact ties),
ragged shapes; and the three generators on ragged batches with empty / one-point / collinear clouds.
usage: python tools/fuzz_all.py [seed] [cases] [what: match,group,matcher,fused,gen[,big]]   (big: rows of 16k - 130k entries, 1 - 40 queries)
Prints one line per case; exits 1 if any case differed (indices must be equal, scores inside tests/helpers.score_tol)."""
import sys
import time
import traceback
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import helpers
import oracle_lib
from so_dso_place_recognition_amd import api, synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
what = set((sys.argv[3] if len(sys.argv) > 3 else "match,group,matcher,fused,gen").split(","))
rng = np.random.default_rng(seed)
bad = []


BIG = "big" in what          # long rows, few queries: the sliced moments / selection / merge of an online call, many shards
what.discard("big")


def shapes():
    if BIG:
        m = int(rng.choice([1, 1, 2, rng.integers(3, 9), rng.integers(9, 40)]))
        n = int(rng.choice([rng.integers(16384, 40000), rng.integers(40000, 130000)]))
        while m * n > 600_000:
            m = max(1, m // 2)
    else:
        m = int(rng.choice([rng.integers(1, 9), rng.integers(9, 70), rng.integers(70, 300)]))
        n = int(rng.choice([rng.integers(2, 40), rng.integers(40, 700), rng.integers(700, 5000)]))
        while m * n > 250_000:
            m = max(1, m // 2)
    k = int(min(n, rng.choice([1, 1, rng.integers(2, 6), rng.integers(6, 61)])))
    mask = int(rng.choice([0, 0, rng.integers(1, 6), rng.integers(6, 160)]))
    return m, n, k, mask


def spoil(q, db, div, planted=None):
    """zero-norm rows (NaN distances in MATLAB), duplicated DB signatures (exact ties -> the lower index wins) and clusters of NEAR-copies of
    a query's planted entry (12 - 60 entries whose distances to the query differ by 1e-11 ... 1e-7: more ties than the k + 8 candidates of the
    fp32-grade pass hold - the containment check must send such a query to its exact row)"""
    m, n = q.shape[0] // div, db.shape[0] // div
    q = q.copy(); db = db.copy()
    notes = []
    if div == 1 and rng.random() < 0.3 and n > 4:                           # SC: whole-row or one-channel zero rows
        j = int(rng.integers(0, n)); db[j, :1200 if rng.random() < 0.5 else 2400] = 0.0; notes.append(f"zero db {j}")
    if div == 1 and rng.random() < 0.15 and m > 1:
        i = int(rng.integers(0, m)); q[i, 1200:] = 0.0; notes.append(f"zero q {i}")
    if div == 1:                                                            # SC intensity channel (binary as SC.cpp:67-72 writes it): what the
        r = rng.random()                                                    # binary path's gate sees - scaled rows, real values, dense rows
        if r < 0.15:
            db[:, 1200:] *= rng.uniform(0.1, 9.0, (db.shape[0], 1)); notes.append("scaled db")
        elif r < 0.25:
            j = int(rng.integers(0, n)); db[j, 1200:] = np.where(db[j, 1200:] != 0, rng.uniform(0.2, 1.0, 1200), 0.0); notes.append(f"real db {j}")
        elif r < 0.32:
            i = int(rng.integers(0, m)); q[i, 1200:] = rng.uniform(0.0, 1.0, 1200); notes.append(f"real q {i}")
        elif r < 0.40:
            dens = rng.uniform(0.35, 0.95); db[:, 1200:] = (rng.random((db.shape[0], 1200)) < dens).astype(np.float64); notes.append(f"dense db {dens:.2f}")
        elif r < 0.45:
            j = int(rng.integers(0, n)); db[j, 1200:] = 1.0; notes.append(f"full db {j}")
    if rng.random() < 0.4 and n > 8:
        for _ in range(int(rng.integers(1, 4))):
            a, b = (int(x) for x in rng.integers(0, n, 2))
            db[b * div:(b + 1) * div] = db[a * div:(a + 1) * div]
        notes.append("dups")
    if planted is not None and rng.random() < 0.35 and n > 80:
        for _ in range(int(rng.integers(1, 3))):
            a = int(planted[int(rng.integers(0, m))])
            c = int(rng.integers(12, min(61, n // 2)))
            spots = rng.choice(np.setdiff1d(np.arange(n), [a]), size=c, replace=False)
            delta = 10.0 ** rng.uniform(-9.5, -5.5) * (0.03 if div == 4 else 1.0)
            e = db[a * div:(a + 1) * div].copy()
            cols = rng.choice(np.nonzero(e[0, :1200] > 0)[0] if div == 1 else np.arange(64), size=20, replace=False)
            for j, b in enumerate(spots):
                x = e.copy()
                x[:, cols] *= 1.0 + (j + 1) * delta
                db[b * div:(b + 1) * div] = x
            notes.append(f"cluster {a} x {c} delta {delta:.1e}")
    return q, db, ",".join(notes)


sigma_ties = []


def check(tag, idx, sc, oidx, osc, tol, resolved=True):
    """resolved=False (a path without fp64 row statistics - none since round 4): neighbours whose scores agree to 1e-5 may come out in the
    other order - the limit of fp32 row statistics; such cases are counted apart, not as findings ("order not guaranteed": must stay empty)."""
    okm = oidx >= 0
    if not np.array_equal(idx, oidx):
        r = np.argwhere(idx != oidx)
        if not resolved and all(abs(sc[tuple(x)] - osc[tuple(x)]) < 1e-5 for x in r):
            sigma_ties.append((tag, len(r)))
            return True
        bad.append((tag, "indices", r[:3].tolist(), idx[tuple(r[0])], oidx[tuple(r[0])]))
        return False
    fin = okm & np.isfinite(osc)
    if not (np.abs(sc - osc)[fin] <= tol[fin]).all() or not np.array_equal(np.isfinite(sc), np.isfinite(osc)):
        e = np.where(fin, np.abs(sc - osc) / tol, 0.0)
        bad.append((tag, "scores", float(e.max()), np.unravel_index(e.argmax(), e.shape)))
        return False
    return True


def sigs(type_, it, m, n):
    if type_ == "sc":
        db = synth.sc_database(1000 + 7 * it + seed, n); q, pl = synth.sc_queries(2000 + it + seed, db, m)
        return q, db, 1, 0, pl
    db = synth.m2dp_database(3000 + 7 * it + seed, n); q, pl = synth.m2dp_queries(4000 + it + seed, db, m)
    return q, db, 4, 1, pl


def f16_tol(osc, n=None, note=""):
    """tests/test_gpu_f16.py: score_tol_f16; rows of a handful of entries have no meaningful sigma in the f16 pass (indices still checked).
    The same for the dense intensity channels of spoil(): ~1000 ones per row leave the row's distances a sigma of ~1e-3, which the f16 pass's
    3e-5 of distance noise moves by percents - PR_SC_ARITH_F16 returns exact indices there, its scores only to ~5e-2 (seed 61, case 33)."""
    if "cluster" in note:       # rows dominated by near-copies have sigma ~ 4e-3: the f16 pass's 1e-4 of (shared) distance error moves it by percents
        return 1e-1 + 3e-3 * np.abs(osc)
    return (3e-2 + 1e-3 * np.abs(osc)) * (1.0 if (n is None or n >= 32) and "dense db" not in note and "full db" not in note else np.inf)


import os
ONLY = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None     # re-run ONE case of a (seed, what) sweep: the others only draw their random numbers
DUMP = os.environ.get("FUZZ_DUMP")                                            # ... and leave its inputs + oracle answer in this .npz
t_start = time.time()
for it in range(cases):
    m, n, k, mask = shapes()
    line = [f"{it}: m={m} n={n} k={k} mask={mask}"]
    if ONLY is not None and it != ONLY:
        for type_ in ("sc", "m2dp"):
            q, db, div, t, pl = sigs(type_, it, m, n)
            spoil(q, db, div, pl)
            if "group" in what and n >= 8 * div:
                rng.integers(2, 9 if BIG else 5)
            if "matcher" in what:
                rng.random(); rng.random()
        if "gen" in what:
            raise SystemExit("FUZZ_ONLY does not replay the generator cases")
        continue
    try:
        for type_ in ("sc", "m2dp"):
            q, db, div, t, pl = sigs(type_, it, m, n)
            q, db, note = spoil(q, db, div, pl)
            rc, oidx, osc = oracle_lib.match_topk(t, q, db, mask, 2.0, k)
            assert rc in (0, -5), rc          # -5: zero-norm rows (their NaN distances are in the result, as in MATLAB)
            if DUMP and ONLY is not None:
                np.savez(DUMP + "_" + type_ + ".npz", q=q, db=db, oidx=oidx, osc=osc, mask=mask, k=k, note=note)
            rc, odp, odi = oracle_lib.sc_distance(q, db) if type_ == "sc" else oracle_lib.m2dp_distance(q, db)
            with np.errstate(invalid="ignore", divide="ignore"):
                tol = np.broadcast_to(helpers.score_tol(np.where(np.isfinite(osc), osc, 0.0), helpers.row_sigmas(odp, odi), eps=1e-7), osc.shape)
            near_tol = 3.0 * tol if (type_ == "m2dp" and ("cluster" in note or "dups" in note)) else tol
            if "match" in what:
                ctx = api.Context(0, exact_statistics=True)     # fp64 row statistics: the reference's doubles to rounding, whatever the rows look like
                idx, sc = api.match_topk(type_, q, db, mask, 2.0, k, ctx=ctx)
                ctx.close()
                with np.errstate(invalid="ignore"):
                    okx = check((it, type_, "exact statistics", m, n, k, mask, note), idx, sc, oidx, osc, 1e-9 * (1.0 + np.abs(np.where(np.isfinite(osc), osc, 0.0))))
                line.append(f"{type_}/exact:{'ok' if okx else 'BAD'}")
                for arith in ("f16x2", "f32", "f16"):
                    ctx = api.Context(0, sc_arith=arith)
                    idx, sc = api.match_topk(type_, q, db, mask, 2.0, k, ctx=ctx)
                    # (M2DP rows of near-copies in the fp32-MFMA arithmetic: every dot is ~2 at a 2^16 scaling, the accumulation truncates and the
                    #  whole cluster shares the ~1e-7 shift - the row MEAN moves by up to 3e-7, tests/helpers.score_tol's eps)
                    #  seeds 21, 27, 28 of round 5: the same in the default arithmetic and with exact duplicates, up to 1.42 x tol - three times
                    #  the tolerance for every such M2DP row; the exact-statistics leg below holds the same rows to 1e-9)
                    tol_a = near_tol
                    ok = check((it, type_, arith, "host", m, n, k, mask, note), idx, sc, oidx, osc, f16_tol(osc, n, note) if arith == "f16" else tol_a)
                    if arith != "f16":
                        gp, gi = (api.processSC if type_ == "sc" else api.processM2DP)(q, db, ctx)
                        same_nan = np.array_equal(np.isnan(gp), np.isnan(odp)) and np.array_equal(np.isnan(gi), np.isnan(odi))
                        err = max(np.nanmax(np.abs(gp - odp)), np.nanmax(np.abs(gi - odi))) if np.isfinite(odp).any() else 0.0
                        if not same_nan or err >= 1e-5:
                            bad.append((it, type_, arith, "distances", err, same_nan)); ok = False
                    ctx.close()
                    line.append(f"{type_}/{arith}:{'ok' if ok else 'BAD'}")
            if "group" in what and n >= 8 * div:
                G = int(rng.integers(2, 9 if BIG else 5))
                g = api.Group([0] * G)
                rg = np.random.default_rng([seed, it, 11])
                n0 = int(rg.integers(max(G, 2), n + 1))                # a growable database: rows n0 .. n - 1 appended in place on the last shard
                if n0 < n and rg.random() < 0.5:
                    g.set_database(type_, db[:n0 * div], extra_capacity=n - n0)
                    at = n0
                    while at < n:
                        c = int(min(n - at, rg.choice([1, 2, 9, 40])))
                        g.append_database(db[at * div:(at + c) * div])
                        at += c
                else:
                    g.set_database(type_, db)
                idx, sc = g.match_topk(q, mask, 2.0, k)
                ok = check((it, type_, "group", G, m, n, k, mask, note), idx, sc, oidx, osc, near_tol)   # (since round 4 the sharded protocol resolves too)
                g.close()
                line.append(f"{type_}/group{G}:{'ok' if ok else 'BAD'}")
            if "matcher" in what:
                import torch
                from so_dso_place_recognition_amd.matcher import Matcher
                for arith in ("f16x2", "f16"):
                    dev = torch.device("cuda", 0)
                    mt = Matcher(type_, m, n, ctx=api.Context(0, sc_arith=arith, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
                    dtype = torch.float32 if rng.random() < 0.3 else torch.float64
                    dbt = torch.from_numpy(db).to(dev).to(dtype); qt = torch.from_numpy(q).to(dev).to(dtype)
                    if dtype == torch.float32:      # the oracle sees the same rounded signatures
                        rc, oidx2, osc2 = oracle_lib.match_topk(t, qt.double().cpu().numpy(), dbt.double().cpu().numpy(), mask, 2.0, k)
                    else:
                        oidx2, osc2 = oidx, osc
                    rg = np.random.default_rng([seed, it, 7])        # (its own stream: FUZZ_ONLY replays the main one draw for draw)
                    grown = ""
                    if dtype == torch.float64 and n >= 4 and rg.random() < 0.5:
                        # the same database built IN PLACE: a reserved set, a bulk start and appends of random sizes (pr_sigset_reserve / _append)
                        n0 = int(rg.integers(0, n))
                        mt.reserve_database(dbt[:n0 * div] if n0 else None)
                        at = n0
                        while at < n:
                            c = int(min(n - at, rg.choice([1, 1, 2, 7, 16, 33, 200])))
                            mt.append_database(dbt[at * div:(at + c) * div].contiguous())
                            at += c
                        grown = "+grown"
                    else:
                        mt.pack_database(dbt)
                    idx, sc = mt.match(qt, mask, 2.0, k)
                    ok = check((it, type_, arith, "Matcher", str(dtype), m, n, k, mask, note), idx.cpu().numpy(), sc.cpu().numpy(), oidx2, osc2,
                               f16_tol(osc2, n, note) if arith == "f16" else near_tol)
                    mt.close()
                    line.append(f"{type_}/Matcher/{arith}{grown}:{'ok' if ok else 'BAD'}")
        if "fused" in what and n >= 4:        # (n = 2, 3: the four z-scores are +-0.707 each and sum to EXACT ties, which no arithmetic orders reproducibly)
            sq, sdb, _, _, _ = sigs("sc", it, m, n); mq, mdb, _, _, _ = sigs("m2dp", it, m, n)
            rc, oidx, osc = oracle_lib.match_topk_fused(sq, mq, sdb, mdb, mask, 2.0, k)
            for arith in ("f16x2", "f16"):
                ctx = api.Context(0, sc_arith=arith)
                idx, sc = api.match_topk_fused(sq, mq, sdb, mdb, mask, 2.0, k, ctx=ctx)
                ok = check((it, "fused", arith, m, n, k, mask), idx, sc, oidx, osc,
                           (6e-2 + 2e-3 * np.abs(osc)) * (1.0 if n >= 32 else np.inf) if arith == "f16" else 2 * helpers.score_tol(osc) + 1e-4)
                ctx.close()
                line.append(f"fused/{arith}:{'ok' if ok else 'BAD'}")
        if "gen" in what:
            N = int(rng.integers(1, 12))
            sizes = [int(rng.choice([0, 1, 2, 3, rng.integers(4, 60), rng.integers(60, 3000), rng.integers(3000, 20000)])) for _ in range(N)]
            xs, its = [], []
            for c, P in enumerate(sizes):
                if P == 0:
                    xs.append(np.empty((0, 3))); its.append(np.empty(0, np.float32)); continue
                p, iv = synth.scene_cloud(77 + seed, 100 * it + c, P)
                if rng.random() < 0.1 and P > 3:
                    p[:, 1] = 2.0 * p[:, 0]; p[:, 2] = -0.5 * p[:, 0]           # collinear: the PCA frame is degenerate in two axes
                xs.append(p); its.append(iv)
            xyz = np.concatenate(xs); inten = np.concatenate(its).astype(np.float32)
            offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            degenerate = [c for c, P in enumerate(sizes) if P < 4] + [c for c, x in enumerate(xs) if len(x) > 3 and np.linalg.matrix_rank(x - x.mean(0)) < 3]
            keep = np.array([c not in degenerate for c in range(N)])
            g = api.sc_generate(xyz, inten, offs); o = oracle_lib.sc_generate(xyz, inten, offs)
            ok_sc = np.isfinite(g).all() and (not keep.any() or (np.array_equal(g[keep][:, 1200:], o[keep][:, 1200:]) and np.abs(g[keep] - o[keep]).max() < 1e-9))
            ok_sc = ok_sc and np.array_equal(g[[P == 0 for P in sizes]], o[[P == 0 for P in sizes]])
            if not ok_sc:
                bad.append((it, "sc_generate", sizes))
            g = api.m2dp_generate(xyz, inten, offs); o = oracle_lib.m2dp_generate(xyz, inten, offs)
            ill = set(int(r) for r in api.m2dp_svd_rows())
            for c in range(N):                                  # sigma_1 == sigma_2 (tiny clouds): the leading pair is not unique
                if keep[c] and sizes[c] <= 256:
                    al, _ = oracle_lib.align_pca(xs[c])
                    for v, (dx, dy) in enumerate([(-1, -1), (-1, 1), (1, -1), (1, 1)]):
                        for M in oracle_lib.m2dp_matrices(al, its[c].astype(np.float32), 45.0, dx, dy):
                            sv = np.linalg.svd(M, compute_uv=False)
                            if sv[0] - sv[1] <= 1e-6 * sv[0]:
                                ill.add(4 * c + v)
            rows = np.array([keep[r // 4] and r not in ill for r in range(4 * N)])
            ok_m2 = np.isfinite(g).all() and (not rows.any() or np.abs(g[rows] - o[rows]).max() < 1e-8)
            if not ok_m2:
                bad.append((it, "m2dp_generate", sizes, float(np.abs(g[rows] - o[rows]).max()) if rows.any() else None))
            g = api.delight_generate(xyz, inten, offs); o = oracle_lib.delight_generate(xyz, inten, offs)
            k16 = np.repeat(keep, 16)
            ok_de = not keep.any() or np.abs(g[k16] - o[k16]).sum() <= 2 * keep.sum()      # counts; a float-cast boundary point may move one count
            if not ok_de:
                bad.append((it, "delight_generate", sizes, float(np.abs(g[k16] - o[k16]).sum())))
            line.append(f"gen{sizes}:{'ok' if ok_sc and ok_m2 and ok_de else 'BAD'}")
    except Exception as e:      # an error return is a finding too
        bad.append((it, "exception", m, n, k, mask, repr(e)))
        traceback.print_exc()
    print(" ".join(line), f"[{time.time() - t_start:.0f}s]", flush=True)
for b in bad:
    print("BAD", b)
for t in sigma_ties:
    print("sigma-level tie (sharded call, not resolved):", t)
print("fuzz_all:", "ok" if not bad else f"{len(bad)} findings", f"seed {seed}, {cases} cases")
sys.exit(1 if bad else 0)
