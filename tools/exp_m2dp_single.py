"""m2dp_match_h8_kernel<single product> (PR_SC_ARITH_F16) on config 3's shape: launch time by the library's events.  python tools/exp_m2dp_single.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
from so_dso_place_recognition_amd.matcher import Matcher
n, m = 50_000, 4096
dev = torch.device("cuda", 0)
cur = int(torch.cuda.current_stream(dev).cuda_stream)
db = synth.m2dp_database_torch(43, n, device=dev)
q_h, planted = synth.m2dp_queries(44, db.cpu().numpy(), m)
q = torch.from_numpy(q_h).to(dev)
for arith in ("f16", "f16x2"):
    mt = Matcher("m2dp", m, n, ctx=Context(0, sc_arith=arith, stream=cur))
    mt.pack_database(db)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(6):
        mt.pre_distances = lambda: s.record(); mt.post_distances = lambda: e.record()
        idx, _ = mt.match(q, 0, 2.0, 1)
        torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    k = float(np.mean(ts[2:])); fpp = 12288 * (3 if arith == "f16x2" else 1)
    print(arith, "launch ms %.3f" % k, "frac of 2.5 PF %.3f" % (m * n * fpp / (k * 1e-3) / 2.5e15), "top1", int((idx.cpu().numpy()[:, 0] == planted).sum()), flush=True)
    mt.close()
