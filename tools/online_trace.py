"""Kernel sequence of one online call: python tools/online_trace.py <m> [arith]  (under rocprofv3 --kernel-trace; prints nothing itself)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
from so_dso_place_recognition_amd.matcher import Matcher
m = int(sys.argv[1]); arith = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
n = 100_000
dev = torch.device("cuda", 0)
db = synth.sc_database_torch(45, n, device=dev)
q_h, planted = synth.sc_queries(46, np.empty((0, 2400)), 64, db_first=0, n_global=n, db_seed=45)
q = torch.from_numpy(q_h).to(dev)[:m].contiguous()
mt = Matcher("sc", 64, n, ctx=Context(0, sc_arith=arith, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
mt.pack_database(db)
for _ in range(5):
    mt.match(q, 0, 2.0, 1)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    idx, _ = mt.match(q, 0, 2.0, 1); torch.cuda.synchronize()
print("ms per call", 1e3 * (time.perf_counter() - t0) / 20, "top1", int((idx.cpu().numpy()[:, 0] == planted[:m]).sum()))
