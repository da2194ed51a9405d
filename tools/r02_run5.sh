mkdir -p gpurun_out
python -m pytest tests -m gpu -q -k "generate or generators or config2 or drive or config1 or cli or config3 or m2dp" 2>&1 | tail -5
python - <<'PY'
import ctypes as C, numpy as np, torch, os, sys, time
sys.path.insert(0,'.')
from bench import HipEvents
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
ev=HipEvents()
P=lambda t: C.c_void_p(t.data_ptr())
for N in (5000, 1024, 64, 1):
    xyz,it,offs=synth.scene_clouds_torch(42,N,50000)
    sig=torch.empty((N,2400),dtype=torch.float64,device='cuda')
    for mode in ("onepass",):
        ctx=Context(0)
        ts=[]
        for r in range(4):
            a,b=ev.create(),ev.create()
            torch.cuda.synchronize(); t0=time.perf_counter()
            ev.record(a,ctx.stream); ctx.check(ctx.lib.pr_sc_generate_dev(ctx.h,P(xyz),P(it),P(offs),N,45.0,P(sig))); ev.record(b,ctx.stream)
            ts.append((ev.elapsed_ms(a,b), 1e3*(time.perf_counter()-t0)))
        by=N*(28*50000+19200)
        print(N, mode, "event ms", [round(t[0],3) for t in ts], "wall ms", [round(t[1],3) for t in ts], "frac of 8TB/s (best event)", by/(min(t[0] for t in ts)*1e-3)/8e12)
        ctx.close()
    del xyz,it,offs,sig
PY
PR_SC_GEN=twopass python - <<'PY'
import ctypes as C, numpy as np, torch, os, sys, time
sys.path.insert(0,'.')
from bench import HipEvents
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
ev=HipEvents()
P=lambda t: C.c_void_p(t.data_ptr())
for N in (5000, 1024, 1):
    xyz,it,offs=synth.scene_clouds_torch(42,N,50000)
    sig=torch.empty((N,2400),dtype=torch.float64,device='cuda')
    ctx=Context(0); ts=[]
    for r in range(4):
        a,b=ev.create(),ev.create()
        torch.cuda.synchronize()
        ev.record(a,ctx.stream); ctx.check(ctx.lib.pr_sc_generate_dev(ctx.h,P(xyz),P(it),P(offs),N,45.0,P(sig))); ev.record(b,ctx.stream)
        ts.append(ev.elapsed_ms(a,b))
    print(N, "twopass event ms", [round(t,3) for t in ts])
    ctx.close(); del xyz,it,offs,sig
PY
python - <<'PY'
# m2dp matcher A/B
import subprocess, os, json
for w in ("8","4"):
    env=dict(os.environ, PR_M2_WAVES=w)
    r=subprocess.run(["python","-c","""
import sys,json; sys.path.insert(0,'.')
import bench, torch, argparse
from bench import HipEvents
class A: db=100000; queries=4096
import numpy as np
ev=HipEvents()
from so_dso_place_recognition_amd import synth
from so_dso_place_recognition_amd.api import Context
from so_dso_place_recognition_amd.matcher import Matcher
dev=torch.device('cuda',0)
n,m=50000,4096
db=synth.m2dp_database_torch(43,n,device=dev); q_h,pl=synth.m2dp_queries(44,db.cpu().numpy(),m); q=torch.from_numpy(q_h).to(dev)
mt=Matcher('m2dp',m,n); mt.pack_database(db)
pair=[ev.create(),ev.create()]; ks=[]
mt.pre_distances=lambda: ev.record(pair[0],mt.ctx.stream); mt.post_distances=lambda: ev.record(pair[1],mt.ctx.stream)
for i in range(6):
    idx,_=mt.match(q,0,2.0,1); ks.append(ev.elapsed_ms(pair[0],pair[1]))
print('kernel ms',[round(k,3) for k in ks],'planted',int((idx.cpu().numpy()[:,0]==pl).sum()))
"""],env=env,capture_output=True,text=True)
    print("PR_M2_WAVES="+w, r.stdout.strip()[-300:], r.stderr.strip()[-300:])
PY
