"""M2DP generation of 1024 x 50 000-point clouds: call time (config 2's second descriptor).  python tools/exp_m2dp_gen.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
dev = torch.device("cuda", 0)
N, PTS = 1024, 50_000
xyz, it, offs = synth.scene_clouds_torch(42, N, PTS, device=dev)
ctx = Context(0, stream=int(torch.cuda.current_stream(dev).cuda_stream))
P = lambda t: C.c_void_p(t.data_ptr())
sig = torch.empty((4 * N, 384), dtype=torch.float64, device=dev)
ts = []
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ctx.check(ctx.lib.pr_m2dp_generate_dev(ctx.h, P(xyz), P(it), P(offs), N, 45.0, P(sig))); e.record()
    torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
print("m2dp generate 1024 clouds: ms", [round(t, 3) for t in ts], "checksum", float(sig.abs().sum().item()))
