"""End-to-end wall time of the drop-in executables (test_sc / test_m2dp / test_delight) on real KITTI poses of the
reference + synthetic points; the phases come from PR_CLI_TIMING (stderr of the executables).
usage: python tools/bench_cli.py [--per-pose 10000]      (GPU box; scratch files under /tmp)"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers

ap = argparse.ArgumentParser()
ap.add_argument("--per-pose", type=int, default=10000)
ap.add_argument("--exes", default="test_sc,test_m2dp,test_delight")
a = ap.parse_args()
D = "/tmp/cli_bench"
os.makedirs(D, exist_ok=True)
poses = os.path.join(ROOT, "tests", "golden", "kitti_seq06", "poses_history_file.txt")
pts = os.path.join(D, f"pts_{a.per_pose}.txt")
if not os.path.exists(pts):
    helpers.write_synthetic_points(poses, pts, per_pose=a.per_pose)
BIN = os.path.join(ROOT, "so_dso_place_recognition_amd", "bin")
OUT = {"test_sc": "sc_file", "test_m2dp": "m2dp_file", "test_delight": "delight_file"}
res = []
for exe in a.exes.split(","):
    for pre in (0, 1):
        for ext in ("txt", "bin"):
            t0 = time.perf_counter()
            r = subprocess.run([os.path.join(BIN, exe), f"_poses_history_file:={poses}", f"_pts_history_file:={pts}",
                                f"_{OUT[exe]}:={D}/{exe}.{ext}", f"_incoming_id_file:={D}/ids.txt", f"_gpu_prestage:={pre}"],
                               capture_output=True, text=True, env=dict(os.environ, PR_CLI_TIMING="1"))
            wall = time.perf_counter() - t0
            ph = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"\[timing\] (.+?)\s+([0-9.]+) s", r.stderr)}
            row = {"exe": exe, "gpu_prestage": pre, "out": ext, "rc": r.returncode, "wall_s": round(wall, 3), "phases": ph}
            if r.returncode:
                row["stderr"] = r.stderr[-300:]; row["stdout"] = r.stdout[-300:]
            res.append(row)
            print(json.dumps(row), flush=True)
