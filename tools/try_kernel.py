#!/usr/bin/env python3
"""Development harness for SC matcher kernels: for every PR_SC_KERNEL name given, (a) max |d - oracle| of both distance channels on a small
seeded case (oracle = tests/oracle_lib, fp64 dense), (b) ms per launch of the matcher alone on the metric workload (4096 x 100k), planted
top-1 count.  usage: python tools/try_kernel.py d e e1 [--n 100000] [--m 4096] [--reps 5]   (each name runs in its own process: the
library reads PR_SC_KERNEL at context creation)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(name, n, m, reps, small):
    import numpy as np
    import torch
    import oracle_lib
    from bench import HipEvents
    from so_dso_place_recognition_amd import api, synth
    from so_dso_place_recognition_amd.matcher import Matcher
    out = {"kernel": name}
    # accuracy on a small case (m > 8 so that the selected kernel runs); names "f16" / "f32" select the arithmetic instead of a kernel
    arith = name.split(":")[0] if name.split(":")[0] in ("f16", "f32") else None      # "f16:e1" = arithmetic f16, PR_SC_KERNEL=e1
    db = synth.sc_database(45, small)
    q, _ = synth.sc_queries(46, db, 40)
    dp, di = api.processSC(q, db, api.Context(0, sc_arith=arith))
    rc, op, oi = oracle_lib.sc_distance(q, db)
    out["max_abs_err"] = float(max(np.abs(dp - op).max(), np.abs(di - oi).max()))
    dev = torch.device("cuda", 0)
    dbt = synth.sc_database_torch(45, n, device=dev)
    q_h, planted = synth.sc_queries(46, dbt[: min(n, 200000)].cpu().numpy() if n > 200000 else dbt.cpu().numpy(), m)
    qt = torch.from_numpy(q_h).to(dev)
    cur = int(torch.cuda.current_stream(dev).cuda_stream)
    mt = Matcher("sc", m, n, ctx=api.Context(0, sc_arith=arith, stream=cur))
    ev = HipEvents()
    mt.pack_database(dbt)
    pair = [ev.create(), ev.create()]
    mt.pre_distances = lambda: ev.record(pair[0], mt.ctx.stream)
    mt.post_distances = lambda: ev.record(pair[1], mt.ctx.stream)
    ks = []
    for _ in range(reps + 1):
        idx, _sc = mt.match(qt, 0, 2.0, 1)
        ks.append(ev.elapsed_ms(pair[0], pair[1]))
    out["ms_per_launch"] = [round(x, 3) for x in ks[1:]]
    out["planted_top1"] = int((idx.cpu().numpy()[:, 0] == planted).sum())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+")
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--small", type=int, default=500)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        one(a.names[0], a.n, a.m, a.reps, a.small)
    else:
        for nm in a.names:
            env = dict(os.environ, PR_SC_KERNEL=nm.split(":")[-1])
            subprocess.call([sys.executable, os.path.abspath(__file__), nm, "--child", "--n", str(a.n), "--m", str(a.m), "--reps", str(a.reps),
                             "--small", str(a.small)], env=env)
