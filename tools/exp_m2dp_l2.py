"""Does m2dp_match_h_kernel<split-f16> wait for HBM?  The same 4096 queries against DBs of 2 500 ... 50 000 signatures: the small images live in the
L2 of the XCD that sweeps them (a quarter of a channel's tiles per XCD; 2 500 signatures = 1.9 MB of its 4 MB), the large one streams from HBM
behind 31 of 32 workgroups' L2 hits.  python tools/exp_m2dp_l2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
from so_dso_place_recognition_amd.matcher import Matcher
m = 4096
dev = torch.device("cuda", 0)
cur = int(torch.cuda.current_stream(dev).cuda_stream)
big = synth.m2dp_database_torch(43, 50_000, device=dev)
q_h, planted = synth.m2dp_queries(44, big[:4 * 2500].cpu().numpy(), m)
q = torch.from_numpy(q_h).to(dev)
for n in (2500, 5000, 10000, 25000, 50000):
    db = big[:4 * n].contiguous()
    mt = Matcher("m2dp", m, n, ctx=Context(0, stream=cur))
    mt.pack_database(db)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(8):
        mt.pre_distances = lambda: s.record(); mt.post_distances = lambda: e.record()
        mt.local_phase1(q)
        torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    k = float(np.mean(ts[3:]))
    print("n = %6d  launch %.3f ms  frac of 2.5 PF %.3f" % (n, k, m * n * 36864 / (k * 1e-3) / 2.5e15), flush=True)
    mt.close()
