#!/usr/bin/env python3
"""How many queries of a step the order / containment checks hand to the exact rows, and what that costs: the metric workload (planted
matches: every top-1 stands ~150 sigma clear of the rest) and the opposite - queries that match NOTHING in the DB (the dense part of a
row, where candidates crowd: the worst case for the checks).  One JSON line per (workload, k).  usage: python tools/flag_rate.py [n] [m]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from so_dso_place_recognition_amd import api, synth
from so_dso_place_recognition_amd.matcher import Matcher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
db = synth.sc_database_torch(45, n, device=dev)
q_pl, planted = synth.sc_queries(46, np.empty((0, 2400)), m, db_first=0, n_global=n, db_seed=45)
q_no = synth.sc_database_torch(4545, m, device=dev)                    # unrelated signatures: no match anywhere
mt = Matcher("sc", m, n, ctx=api.Context(0, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
mt.pack_database(db)
for name, q in (("planted", torch.from_numpy(q_pl).to(dev)), ("no_match", q_no)):
    for k in (1, 5):
        mt.match(q, 0, 2.0, k, exact_order=False)
        flagged = mt.flagged_count()
        ts = {}
        for eo in (False, True):
            for _ in range(2):
                mt.match(q, 0, 2.0, k, exact_order=eo)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                mt.match(q, 0, 2.0, k, exact_order=eo)
            torch.cuda.synchronize()
            ts[eo] = 1e3 * (time.perf_counter() - t0) / 5
        print(json.dumps({"workload": name, "n": n, "m": m, "k": k, "flagged": flagged, "ms_match_without_resolution": ts[False],
                          "ms_match": ts[True], "warnings": mt.take_warnings()}), flush=True)
