"""Host vs GPU pre-stage (utils/pts_preprocess.h:135-232) on real KITTI poses of the reference + synthetic points.
usage: python tools/bench_prestage.py [--per-pose 2000] [--seq kitti_seq06]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from so_dso_place_recognition_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--per-pose", type=int, default=2000)
ap.add_argument("--seq", default="kitti_seq06")
ap.add_argument("--dir", default="/tmp/prestage_bench")
a = ap.parse_args()
os.makedirs(a.dir, exist_ok=True)
poses = os.path.join(ROOT, "tests", "golden", a.seq, "poses_history_file.txt")
pts = os.path.join(a.dir, f"pts_{a.seq}_{a.per_pose}.txt")
if not os.path.exists(pts):
    helpers.write_synthetic_points(poses, pts, per_pose=a.per_pose)
out = {"poses": sum(1 for l in open(poses) if l.strip()), "points_per_pose": a.per_pose}
ref = None
for polar in (False, True):
    for gpu in (False, True, True):
        t0 = time.perf_counter()
        r = api.pts_preprocess(poses, pts, None, 45.0, polar, gpu=gpu)
        wall = time.perf_counter() - t0
        key = ("polar" if polar else "grid") + ("_gpu" if gpu else "_host")
        out[key] = {"wall_s_incl_parse": wall, "avg_ms_per_cloud": api.pts_preprocess.last_avg_ms, "clouds": int(len(r[3])),
                    "points_out": int(r[2][-1])}
        if not gpu:
            ref = r
        else:
            out[key]["identical_to_host"] = bool(np.array_equal(r[0].view(np.uint64), ref[0].view(np.uint64)) and
                                                 np.array_equal(r[2], ref[2]) and np.array_equal(r[1].view(np.uint32), ref[1].view(np.uint32)))
print(json.dumps(out))
