#!/usr/bin/env python3
"""Cost of the exact-row resolution (exact_row.hip): m queries against a resident n-entry DB with EVERY query flagged
(pr_set_exact_statistics), against the same call with nothing flagged.  One JSON line per (type, m): ms per call both ways, the
difference per pass of up to 64 queries, and max |score - oracle| on a sample.  PR_XROW=direct selects the reference's own formulation
(23 ns per pair) for comparison.  usage: python tools/bench_xrow.py [n] [type sc|m2dp|fused]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from so_dso_place_recognition_amd import api, synth
from so_dso_place_recognition_amd.matcher import Matcher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
type_ = sys.argv[2] if len(sys.argv) > 2 else "sc"
dev = torch.device("cuda", 0)
if type_ == "sc":
    db = synth.sc_database_torch(45, n, device=dev)
    q_h, planted = synth.sc_queries(46, np.empty((0, 2400)), 64, db_first=0, n_global=n, db_seed=45)
else:
    db = synth.m2dp_database_torch(43, n, device=dev)
    q_h, planted = synth.m2dp_queries(44, db[: 4 * 4096].cpu().numpy(), 64)
rps = 1 if type_ == "sc" else 4
for m in (1, 8, 64):
    res = {"type": type_, "n": n, "m": m, "xrow": os.environ.get("PR_XROW", "spectral")}
    for exact in (False, True):
        mt = Matcher(type_, 64, n, ctx=api.Context(0, exact_statistics=exact, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
        mt.pack_database(db)
        q = torch.from_numpy(q_h[: rps * m]).to(dev)
        for _ in range(3):
            idx, sc = mt.match(q, 0, 2.0, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            idx, sc = mt.match(q, 0, 2.0, 1)
        torch.cuda.synchronize()
        res["ms_exact" if exact else "ms_default"] = 1e3 * (time.perf_counter() - t0) / reps
        res["idx_exact" if exact else "idx_default"] = idx.cpu().numpy()[:, 0].tolist()[:4]
        if exact:
            res["score_exact"] = sc.cpu().numpy()[:2, 0].tolist()
        else:
            res["score_default"] = sc.cpu().numpy()[:2, 0].tolist()
        mt.close()
    res["ms_per_pass"] = res["ms_exact"] - res["ms_default"]
    print(json.dumps(res), flush=True)
