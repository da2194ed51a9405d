#!/usr/bin/env python3
"""One SC and one M2DP generation call on synthetic 50 000-point clouds (the bench's config-2 extras alone), for rocprofv3 runs:
   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS ... -- python tools/gen_only.py [--clouds 128] [--reps 2]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context

ap = argparse.ArgumentParser()
ap.add_argument("--clouds", type=int, default=128)
ap.add_argument("--points", type=int, default=50_000)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--sc", action="store_true", help="also the SC generator")
a = ap.parse_args()
dev = torch.device("cuda", 0)
xyz, it, offs = synth.scene_clouds_torch(42, a.clouds, a.points, device=dev)
ctx = Context(0)
P = lambda t: t.data_ptr()
sig = torch.empty((4 * a.clouds, 384), dtype=torch.float64, device=dev)
sc = torch.empty((a.clouds, 2400), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for _ in range(a.reps):
    ctx.check(ctx.lib.pr_m2dp_generate_dev(ctx.h, P(xyz), P(it), P(offs), a.clouds, 45.0, P(sig)))
    if a.sc:
        ctx.check(ctx.lib.pr_sc_generate_dev(ctx.h, P(xyz), P(it), P(offs), a.clouds, 45.0, P(sc)))
ctx.sync()
print("done")
