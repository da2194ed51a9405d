import sys, numpy as np, ctypes as C
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from so_dso_place_recognition_amd import synth, api
from so_dso_place_recognition_amd.matcher import Matcher
m, n, k = 4, 3000, 2
db = synth.sc_database(45, n); q, planted = synth.sc_queries(46, db, 3 * m)
mt = Matcher.on_new_stream("sc", m, n)
with torch.cuda.stream(mt.stream):
    mt.pack_database(torch.from_numpy(db).cuda())
    qs = torch.from_numpy(q[:m]).cuda()
cap = mt.capture(qs, 0, 2.0, k)
def state(mt):
    st = C.c_int32(-1); mt.ctx.check(mt.lib.pr_sc_binary_state(mt.ctx.h, mt.q, mt.db, C.byref(st))); return st.value
for r in range(3):
    qq = q[r*m:(r+1)*m]
    idx, sc = cap.run(torch.from_numpy(qq).cuda())
    dpi = [t.cpu().numpy().copy() for t in mt.distances()]
    s1 = state(mt)
    wi, ws = api.match_topk("sc", qq, db, 0, 2.0, k)
    gp, gi = api.processSC(qq, db)
    print(r, "state", s1, "score diff", np.abs(sc.cpu().numpy()-ws).max(), "d_p diff", np.abs(dpi[0]-gp).max(), "d_i diff", np.abs(dpi[1]-gi).max(), "bad d_i", int((dpi[1]!=gi).sum()))
    with torch.cuda.stream(mt.stream):
        i2, s2 = mt.match(torch.from_numpy(qq).cuda(), 0, 2.0, k)
        mt.stream.synchronize()
    d2 = [t.cpu().numpy().copy() for t in mt.distances()]
    print("   eager matcher: score diff vs host", np.abs(s2.cpu().numpy()-ws).max(), "d_i diff", np.abs(d2[1]-gi).max(), "state", state(mt))
