#!/usr/bin/env python3
"""On the GPU box: the issue-bound model of tools/issue_model.py against THIS box's measurements.
  1. tools/ubench/mfma16_fillers (built here)                       -> the per-instruction costs of this box
  2. rocprofv3 --kernel-trace of one bench command per arithmetic    -> ms per launch of the matcher (timed launches)
  3. rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA (its own pass, no trace domain beside it)
                                                                      -> GPU cycles per launch, matrix-pipe busy share, VALU per MFMA
  4. tools/issue_model.py with (1) on the kernel's instruction stream -> model cycles per unit; measured / model
usage: python tools/model_check.py <tag>      -> gpurun_out/<tag>_model.json (+ <tag>_fillers.txt, <tag>_costs.json)"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "model"
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp", PR_SC_BINARY="0")      # both channels through the modelled kernel (the model counts 2 channels of units)


def sh(cmd, **kw):
    return subprocess.run(cmd, shell=True, capture_output=True, text=True, env=env, cwd="/tmp", **kw)


exe = os.path.join(ROOT, "tools", "ubench", "mfma16_fillers")
r = sh(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 {exe}.hip -o {exe} && {exe}")
fillers = os.path.join(out, f"{tag}_fillers.txt")
open(fillers, "w").write(r.stdout + r.stderr[-2000:])
costs = os.path.join(out, f"{tag}_costs.json")
res = {"tag": tag, "host": os.uname().nodename}
for arith, kern, pat in (("f16x2", "split", "sc_match_e_kernel<true"), ("f16", "single", "sc_match_e_kernel<false, 8, 8, false>")):
    bench = f"python {ROOT}/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --sc-arith {arith}"
    d = os.path.join(out, f"{tag}_tr_{arith}")
    sh(f"rm -rf {d}; rocprofv3 --kernel-trace --stats -d {d} -o sc -- {bench}")
    ms = None
    for db in glob.glob(d + "/**/*_results.db", recursive=True):
        c = sqlite3.connect(db)
        rows = [x for x in c.execute("select name, (end-start)/1e6 from kernels") if pat.replace(" ", "") in x[0].replace(" ", "")]
        if rows:
            mx = max(x[1] for x in rows)
            big = [x[1] for x in rows if x[1] > 0.5 * mx]
            ms = sum(big) / len(big)
    d2 = os.path.join(out, f"{tag}_pmc_{arith}")
    sh(f"rm -rf {d2}; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d {d2} -o sc -- "
       f"python {ROOT}/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --sc-arith {arith}")
    pmc = {}
    for db in glob.glob(d2 + "/**/*_results.db", recursive=True):
        c = sqlite3.connect(db)
        for name, ctr, val, n in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if pat.replace(" ", "") in name.replace(" ", ""):
                pmc[ctr] = val / n
    sh(f"rm -rf {d} {d2}")
    e = {"ms_per_launch_trace": ms, "pmc_per_launch": pmc}
    if pmc.get("GRBM_GUI_ACTIVE"):
        cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0                       # the counter sums the 8 XCDs
        e["gpu_cycles_per_launch"] = cyc
        if ms:
            e["sustained_ghz"] = cyc / (ms * 1e-3) / 1e9
        if pmc.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            e["matrix_pipe_busy"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        if pmc.get("SQ_INSTS_MFMA"):
            e["valu_per_mfma"] = pmc.get("SQ_INSTS_VALU", 0.0) / pmc["SQ_INSTS_MFMA"]
        ghz = f" --ghz {e['sustained_ghz']:.4f}" if e.get("sustained_ghz") else ""
        m = sh(f"python {ROOT}/tools/issue_model.py --kernel {kern} --fit {fillers} --costs {costs} --measured-cycles {cyc}{ghz} --json")
        try:
            e["model"] = json.loads([l for l in m.stdout.splitlines() if l.startswith("{")][-1])
        except Exception:
            e["model_error"] = (m.stdout + m.stderr)[-1500:]
    res[arith] = e
json.dump(res, open(os.path.join(out, f"{tag}_model.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
