"""Per-segment cycle counts of one wave of sc_match_h_kernel (library built with -DPR_SCH_TIMING; results invalid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.matcher import Matcher

n, m = 100000, 4096
db = synth.sc_database(45, n); q, _ = synth.sc_queries(46, db, m)
mt = Matcher("sc", m, n)
mt.pack_database(torch.from_numpy(db).cuda())
dq = torch.from_numpy(q).cuda()
for _ in range(2):
    mt.match(dq, 0, 2.0, 1)
torch.cuda.synchronize()
dp, _ = mt.distances()
t = dp.view(torch.int64).flatten()[:8].cpu().numpy()
names = ["stage1 half0", "tail+swap0 half0", "stage2 half0", "stage1 half1", "tail+swap0 half1", "stage2 half1+epilogue 0-2", "epilogue 3", "loop head"]
for nm, v in zip(names, t):
    print(f"{nm:28s} {v:8d} cycles")
print("total", t.sum())
