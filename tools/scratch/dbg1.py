import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib
from so_dso_place_recognition_amd import synth, api
for (m,n) in [(37,101),(64,256),(70,1500)]:
    db = synth.sc_database(45, n); q,_ = synth.sc_queries(46, db, m)
    gp, gi = api.processSC(q, db)
    rc, op, oi = oracle_lib.sc_distance(q, db)
    e = np.abs(gi-oi)
    bad = np.argwhere(e>1e-6)
    print(m,n,"bad count",len(bad), "rows", sorted(set(bad[:,0]))[:20], "cols", sorted(set(bad[:,1]))[:40])
    if len(bad):
        i,j = bad[0]; print(" ex", i,j, gi[i,j], oi[i,j])
        oq=(q[:,1200:]!=0).sum(1); od=(db[:,1200:]!=0).sum(1)
        print(" counts gpu", (1-2*gi[i,j])*np.sqrt(oq[i]*od[j]), "oracle", (1-2*oi[i,j])*np.sqrt(oq[i]*od[j]))
