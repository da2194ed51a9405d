#!/usr/bin/env python3
"""Signed error of the fp32 all-pairs distances against the fp64 oracle: is it noise or a systematic (affine) bias?
Prints mean / std of e = d_gpu - d_oracle per channel and arithmetic, and the least-squares fit e ~ a d + b."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from so_dso_place_recognition_amd import api, synth

n, m = 20000, 48
db = synth.sc_database(45, n)
q, _ = synth.sc_queries(46, db, m)
rc, a, b = oracle_lib.sc_distance(db, q)      # roles swapped (symmetric)
op, oi = a.T, b.T
for arith in ("f16x2", "f32"):
    ctx = api.Context(0, sc_arith=arith)
    gp, gi = api.processSC(q, db, ctx)
    for name, g, o in (("struct", gp, op), ("inten", gi, oi)):
        e = g.astype(np.float64) - o
        A = np.stack([o.ravel(), np.ones(o.size)], 1)
        coef, *_ = np.linalg.lstsq(A, e.ravel(), rcond=None)
        res = e.ravel() - A @ coef
        # error left after rounding the oracle's value to fp32 (what a perfect fp32 kernel would store)
        e32 = g.astype(np.float64) - o.astype(np.float32).astype(np.float64)
        print(f"{arith} {name}: mean e {e.mean():+.3e} std {e.std():.3e} max|e| {np.abs(e).max():.3e} | fit e = {coef[0]:+.3e} d {coef[1]:+.3e}, "
              f"residual std {res.std():.3e} | d mean {o.mean():.4f} sigma {o.std(1).mean():.5f} | vs fp32(oracle): mean {e32.mean():+.3e}")
    ctx.close()
