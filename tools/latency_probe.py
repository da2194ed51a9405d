#!/usr/bin/env python3
"""m queries against a resident 100k DB, 30 calls - for `rocprofv3 --kernel-trace --stats` (which kernels make up an online call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from so_dso_place_recognition_amd import api, synth
from so_dso_place_recognition_amd.matcher import Matcher
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1
arith = sys.argv[2] if len(sys.argv) > 2 else None
n = 100000
dev = torch.device("cuda", 0)
db = synth.sc_database_torch(45, n, device=dev)
q_h, planted = synth.sc_queries(46, np.empty((0, 2400)), max(8, m), db_first=0, n_global=n, db_seed=45)
q = torch.from_numpy(q_h[:m]).to(dev)
mt = Matcher("sc", max(8, m), n, ctx=api.Context(0, sc_arith=arith, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
mt.pack_database(db)
import time
for _ in range(5):
    mt.match(q, 0, 2.0, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    idx, sc = mt.match(q, 0, 2.0, 1)
    torch.cuda.synchronize()
print("ms per call", 1e3 * (time.perf_counter() - t0) / 30, "top1 ok", int((idx.cpu().numpy()[:, 0] == planted[:m]).sum()))
