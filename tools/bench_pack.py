"""DB pack time of the SC matcher (100 000 signatures, split-f16 images + binary-channel statistics): python tools/bench_pack.py (GPU box)."""
import sys, time, numpy as np
sys.path.insert(0,'.')
import torch
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.matcher import Matcher
n=100000
dev=torch.device("cuda",0)
db = synth.sc_database_torch(45, n, device=dev)
mt = Matcher("sc", 64, n)
for _ in range(3): mt.pack_database(db)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): mt.pack_database(db)
torch.cuda.synchronize(); print("pack ms", 1e3*(time.perf_counter()-t0)/20)
