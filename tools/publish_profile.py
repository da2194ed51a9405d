#!/usr/bin/env python3
"""gpurun_out/<tag>.txt (tools/profile_round6.sh) -> the committed artefacts: profiles/<tag>.txt (trace section whole; PMC lines of the library's main
kernels at grids of >= 65536 threads), profiles/<tag>_derived.json (profiles/derive.py on THAT file) and profiles/<round>_traffic.json (HBM bytes per
launch per (kernel, grid, duration cluster), the structure-channel launch of the metric step tagged for bench.py).
usage: python tools/publish_profile.py r06_final r06"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
keep = ("sc_match_e_kernel", "sc_match_kernel", "m2dp_", "sc_bin_kernel", "cloud_frames", "sc_pack_h_col", "sc_pack_kernel", "fuse_select_kernel", "row_moments",
        "rerank_kernel", "xrow_", "ave_chain", "moments_finish", "slice_merge", "sc_pack_h_few", "nan_fixup", "sc_bstat")
out, mode = [], None
for l in open(os.path.join(ROOT, "gpurun_out", tag + ".txt")):
    if l.startswith(("#", "==", "--", "{")) or not l.strip():
        out.append(l)
        if l.startswith("-- PMC"):
            mode = "pmc"
        elif l.startswith("-- kernel trace") or l.startswith("-- duration"):
            mode = "trace"
        continue
    if mode == "pmc":
        m = re.search(r"grid=(\d+)", l)
        if m and int(m.group(1)) >= 65536 and any(k in l for k in keep):
            out.append(l)
    else:
        out.append(l)
dst = os.path.join(ROOT, "profiles", tag + ".txt")
open(dst, "w").write("# (copy of gpurun_out/%s.txt; PMC lines of the library's main kernels at grids of >= 65536 threads only - the full file has one line per "
                     "(kernel, grid, cluster, counter))\n" % tag + "".join(out))
d = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "profiles", "derive.py"), dst]))
json.dump(d, open(os.path.join(ROOT, "profiles", tag + "_derived.json"), "w"), indent=1)
res = {"source": f"profiles/{tag}.txt via profiles/derive.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --steps 3 --warmup 1 "
                 "--no-cpu-baseline --no-kitti-shape` (with the extra workloads); per launch, per (kernel, grid, duration cluster); FETCH_SIZE x2 (gfx950 wide-read "
                 "correction, MI355X_MICROARCH.md)", "workload": {"db": 100000, "queries": 4096, "n_gpus": 1}}
struct = sorted([k for k in d["kernels"] if k.startswith("sc_match_e_kernel<true, 4, 4, false> @ 262144")], key=lambda k: d["kernels"][k]["ms_per_launch"])
struct = [k for k in struct if d["kernels"][k]["ms_per_launch"] > 1.0]
for k, e in d["kernels"].items():
    if "hbm_bytes" not in e:
        continue
    r = {"hbm_bytes_per_launch": e["hbm_bytes"], "hbm_read_bytes": e["hbm_read_bytes"], "hbm_write_bytes": e["hbm_write_bytes"], "ms_per_launch": e["ms_per_launch"],
         "fetch_correction": 2.0}
    if struct and k == struct[0]:
        r["launch"] = "structure channel"
    if "caveat" in e:
        r["caveat"] = e["caveat"]
    res[k] = r
json.dump(res, open(os.path.join(ROOT, "profiles", rnd + "_traffic.json"), "w"), indent=1)
print("wrote", dst, len(d["kernels"]), "kernels;", "structure launch:", struct[0] if struct else None)
