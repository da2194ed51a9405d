#!/usr/bin/env python3
"""Per-stage timings of the hot path on one MI355X (HIP events on the context's stream): SC / M2DP generation
(BASELINE.json config 2 shape, P = 50k points per cloud) and M2DP matching (config 3: 50k-signature DB).
Prints one JSON object.  Usage: python tools/bench_stages.py [--clouds 256] [--m2dp-clouds 32] [--reps 3]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import HipEvents
from so_dso_place_recognition_amd import _lib, synth
from so_dso_place_recognition_amd.api import Context
from so_dso_place_recognition_amd.matcher import Matcher


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=256)
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--m2dp-clouds", type=int, default=32)
    ap.add_argument("--m2dp-db", type=int, default=50000)
    ap.add_argument("--m2dp-queries", type=int, default=4096)
    ap.add_argument("--delight-db", type=int, default=20000)
    ap.add_argument("--delight-queries", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    lib, h = ctx.lib, ctx.h
    ev = HipEvents(); e0, e1 = ev.create(), ev.create()
    out = {}

    def timed(fn):
        ts = []
        for _ in range(args.reps + 1):
            ctx.sync()
            ev.record(e0, ctx.stream); fn(); ev.record(e1, ctx.stream)
            ts.append(ev.elapsed_ms(e0, e1))
        return float(np.min(ts[1:])), float(np.mean(ts[1:]))

    N, P = args.clouds, args.points
    assert args.m2dp_clouds <= N, "--m2dp-clouds uses the first clouds of --clouds: raise --clouds"
    t0 = time.time()
    base = 64                                  # distinct clouds; the rest are rigidly moved copies (cheap to generate)
    xyz, it, offs = synth.scene_clouds(42, min(N, base), P)
    reps = (N + base - 1) // base
    xyz = np.concatenate([xyz + 0.01 * r for r in range(reps)])[: N * P]
    it = np.tile(it, reps)[: N * P]
    offs = np.arange(N + 1, dtype=np.int64) * P
    gen_s = time.time() - t0
    dx = torch.from_numpy(xyz).to(dev); di = torch.from_numpy(it).to(dev); do = torch.from_numpy(offs).to(dev)
    dsc = torch.empty((N, 2400), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    p = lambda t: C.c_void_p(t.data_ptr())
    mn, av = timed(lambda: ctx.check(lib.pr_sc_generate_dev(h, p(dx), p(di), p(do), N, 45.0, p(dsc))))
    bytes_alg = N * (28 * P + 19200)
    out["sc_generate"] = {"clouds": N, "points": P, "ms": mn, "ms_mean": av, "clouds_per_s": N / (mn * 1e-3),
                          "algorithmic_GBps": bytes_alg / (mn * 1e-3) / 1e9, "frac_of_8TBps": bytes_alg / (mn * 1e-3) / 8e12}
    Nm = args.m2dp_clouds
    dm = torch.empty((4 * Nm, 384), dtype=torch.float64, device=dev)
    mn, av = timed(lambda: ctx.check(lib.pr_m2dp_generate_dev(h, p(dx), p(di), p(do), Nm, 45.0, p(dm))))
    out["m2dp_generate"] = {"clouds": Nm, "points": P, "ms": mn, "clouds_per_s": Nm / (mn * 1e-3),
                            "projections_per_s": Nm * 256 * P / (mn * 1e-3)}
    # M2DP match (config 3)
    n, m = args.m2dp_db, args.m2dp_queries
    db = synth.m2dp_database(43, n); q, planted = synth.m2dp_queries(44, db, m)
    mt = Matcher("m2dp", m, n, ctx=ctx)
    ddb = torch.from_numpy(db).to(dev); dq = torch.from_numpy(q).to(dev)
    mt.pack_database(ddb)
    ks = []
    mt.pre_distances = lambda: ev.record(e0, ctx.stream)
    mt.post_distances = lambda: ev.record(e1, ctx.stream)
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        mt.pack_database(ddb)
        idx, sc = mt.match(dq, 0, 2.0, 1)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ks.append((ev.elapsed_ms(e0, e1), wall))
    kms = min(k for k, _ in ks[1:]); wall = min(w for _, w in ks[1:])
    f16 = ctx.sc_arith == "f16x2"      # split-f16 GEMM: 3 f16 products per fp32 product, f16 MFMA peak 2.5 PFLOP/s
    fl = m * n * 12288 * (3 if f16 else 1)
    peak = 2500e12 if f16 else 157.3e12
    out["m2dp_match"] = {"db": n, "queries": m, "arith": ctx.sc_arith, "kernel_ms": kms, "step_ms": wall * 1e3, "queries_per_s": m / wall,
                         "TFLOPs": fl / (kms * 1e-3) / 1e12, "frac_of_peak": fl / (kms * 1e-3) / peak,
                         "fp32_formulation_TFLOPs": m * n * 12288 / (kms * 1e-3) / 1e12,
                         "planted_top1_correct": int((idx.cpu().numpy()[:, 0] == planted).sum())}
    mt.close()
    # DELIGHT (row f3): generation on the same clouds, chi-square match on synthetic histograms
    dd = torch.empty((16 * N, 256), dtype=torch.float64, device=dev)
    mn, av = timed(lambda: ctx.check(lib.pr_delight_generate_dev(h, p(dx), p(di), p(do), N, p(dd))))
    bytes_alg = N * (28 * P + 32768)
    out["delight_generate"] = {"clouds": N, "points": P, "ms": mn, "clouds_per_s": N / (mn * 1e-3),
                               "algorithmic_GBps": bytes_alg / (mn * 1e-3) / 1e9}
    n, m = args.delight_db, args.delight_queries
    db = synth.delight_database(51, n); q, planted = synth.delight_queries(52, db, m)
    mt = Matcher("delight", m, n, ctx=ctx)
    ddb = torch.from_numpy(db).to(dev); dq = torch.from_numpy(q).to(dev)
    mt.pre_distances = lambda: ev.record(e0, ctx.stream)
    mt.post_distances = lambda: ev.record(e1, ctx.stream)
    ks = []
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        mt.pack_database(ddb)
        idx, sc = mt.match(dq, 0, 2.0, 1)
        torch.cuda.synchronize()
        ks.append((ev.elapsed_ms(e0, e1), time.perf_counter() - t0))
    kms = min(k for k, _ in ks[1:]); wall = min(w for _, w in ks[1:])
    out["delight_match"] = {"db": n, "queries": m, "kernel_ms": kms, "step_ms": wall * 1e3, "queries_per_s": m / wall,
                            "Gterms_per_s": m * n * 16384 / (kms * 1e-3) / 1e9,
                            "planted_top1_correct": int((idx.cpu().numpy()[:, 0] == planted).sum())}
    out["setup_s"] = gen_s
    print(json.dumps(out))


if __name__ == "__main__":
    main()
