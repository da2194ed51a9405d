"""What the operand DATA costs the SC matcher (power-limited clock): the metric launch on the synthetic DB against the same launch on an all-zero
DB (every signature zero-norm: packed as zeros, same instruction stream).  python tools/exp_zero_data.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
from so_dso_place_recognition_amd.matcher import Matcher
n, m = 100_000, 4096
dev = torch.device("cuda", 0)
cur = int(torch.cuda.current_stream(dev).cuda_stream)
for name in ("synthetic", "zeros", "synthetic"):
    db = synth.sc_database_torch(45, n, device=dev)
    q_h, planted = synth.sc_queries(46, np.empty((0, 2400)), m, db_first=0, n_global=n, db_seed=45)
    q = torch.from_numpy(q_h).to(dev)
    if name == "zeros":
        db.zero_(); q.zero_()
    for binary in (True, False):
        mt = Matcher("sc", m, n, ctx=Context(0, sc_binary=binary, stream=cur))
        mt.pack_database(db)
        mt.ctx.kernel_timing(True)
        acc = []
        for _ in range(6):
            mt.local_phase1(q)
            acc.append(mt.ctx.last_distance_timing())
        mt.ctx.kernel_timing(False)
        a = np.array(acc[2:]).mean(0)
        print(name, "binary path" if binary else "both channels split-f16", "launch ms:", [round(float(x), 3) for x in a], flush=True)
        mt.close()
    del db, q
