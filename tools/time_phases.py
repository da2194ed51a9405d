#!/usr/bin/env python3
"""Times the phases of a device-resident match step (pack, distances, moments, select, re-evaluation, margin) with torch events.
usage: python tools/time_phases.py [f16x2|f16] [--fused] [--n 100000] [--m 4096] [--k 1]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from so_dso_place_recognition_amd import api, synth
from so_dso_place_recognition_amd.matcher import Matcher, FusedMatcher

ap = argparse.ArgumentParser()
ap.add_argument("arith", nargs="?", default="f16x2")
ap.add_argument("--fused", action="store_true")
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--k", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cur = int(torch.cuda.current_stream(dev).cuda_stream)
ctx = api.Context(0, sc_arith=a.arith, stream=cur)
db = synth.sc_database_torch(45, a.n, device=dev)
q_h, planted = synth.sc_queries(46, np.empty((0, 2400)), a.m, db_first=0, n_global=a.n, db_seed=45)
q = torch.from_numpy(q_h).to(dev)
if a.fused:
    dbm = synth.m2dp_database_torch(73, a.n, device=dev)
    qm = torch.from_numpy(np.tile(synth.m2dp_queries(74, synth.m2dp_database(73, 64), 64)[0], (a.m // 64, 1))).to(dev)
    mt = FusedMatcher(a.m, a.n, ctx=ctx)
    pack = lambda: mt.pack_database(db, dbm)
    ph1 = lambda: mt.local_phase1(q, qm)
else:
    mt = Matcher("sc", a.m, a.n, ctx=ctx)
    pack = lambda: mt.pack_database(db)
    ph1 = lambda: mt.local_phase1(q)

def timed(name, fn, acc):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
    acc.setdefault(name, []).append(e0.elapsed_time(e1))
    return r

acc = {}
for it in range(4):
    timed("pack_db", pack, acc)
    mom = timed("phase1 (pack q + distances + moments)", ph1, acc)
    sel = timed("select", lambda: mt.local_select(mom.unsqueeze(0), 1, 0, 2.0, a.k, 0, 0), acc)
    idx, sc = timed("re-evaluation", lambda: mt.local_rerank(sel[0], a.k, False, sel[1]), acc)
    if mt.f16:
        if a.fused:
            timed("margin", lambda: mt._margin(mt._m1, mt._m2, 1, 2.0, sel[1], a.k, sc), acc)
        else:
            timed("margin", lambda: mt._margin(mt._mom_all, None, 1, 2.0, sel[1], a.k, sc), acc)
print(a.arith, "fused" if a.fused else "sc", "kin", sel[0].shape[1], {k: round(float(np.mean(v[1:])), 3) for k, v in acc.items()},
      "planted", int((idx.cpu().numpy()[:, 0] == planted).sum()))
