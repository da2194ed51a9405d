import sys, numpy as np, torch
sys.path.insert(0,"."); sys.path.insert(0,"tests")
import oracle_lib
from so_dso_place_recognition_amd import api, _lib
from so_dso_place_recognition_amd.matcher import Matcher
d=np.load("tools/_case21_sc.npz")
q,db,oidx,osc,mask,k=d["q"],d["db"],d["oidx"],d["osc"],int(d["mask"]),int(d["k"])
m,n=q.shape[0],db.shape[0]
r=166
rc,dp,di=oracle_lib.sc_distance(q[r:r+1],db)
f=2*(dp-dp.mean())/dp.std(ddof=1)+(di-di.mean())/di.std(ddof=1)
o=np.argsort(f[0],kind="stable")[:70]
print("oracle order", o[:12].tolist(), f[0][o[:6]].tolist(), "rank of 93,120:", list(o).index(93), list(o).index(120))
print("score gaps around 58:", f[0][o[40:62]].round(4).tolist())
dev=torch.device("cuda",0)
mt=Matcher("sc",m,n,ctx=api.Context(0,sc_arith="f16",stream=int(torch.cuda.current_stream(dev).cuda_stream)))
mt.pack_database(torch.from_numpy(db).to(dev))
i0,s0=mt.match(torch.from_numpy(q).to(dev),mask,2.0,k,f16_fallback=False)
torch.cuda.synchronize()
cand=mt._last_cand[0].cpu().numpy()[r]; csc=mt._last_cand[1].cpu().numpy()[r]
print("f16 result", i0.cpu().numpy()[r], s0.cpu().numpy()[r], "flag", int(mt.f16_flags.cpu().numpy()[r]), "nflags", int(mt.f16_count.item()))
print("93 in cands", 93 in cand, "120 in cands", 120 in cand, "T", csc[-1], "cand pass scores of 93/120", [csc[list(cand).index(x)] if x in cand else None for x in (93,120)])
i1,s1=mt.match(torch.from_numpy(q).to(dev),mask,2.0,k)
print("with fallback", i1.cpu().numpy()[r], s1.cpu().numpy()[r], "fallbacks", mt.f16_fallbacks, "warn", mt.take_warnings())
tw=mt._split_twin()
i2,s2=tw.match(torch.from_numpy(q[r:r+1]).to(dev),mask,2.0,k,q_row0=r)
print("twin alone", i2.cpu().numpy(), s2.cpu().numpy(), "warn", tw.take_warnings())
print("sigmas", dp.std(ddof=1), di.std(ddof=1), "w", 2/dp.std(ddof=1)+1/di.std(ddof=1))
mt.match(torch.from_numpy(q).to(dev),mask,2.0,k,f16_fallback=False)
ci,cs=mt._last_cand
i3,s3=mt.local_rerank(ci,k,False,None)
torch.cuda.synchronize()
print("rerank without pruning", i3.cpu().numpy()[r], s3.cpu().numpy()[r])
i4,s4=mt.local_rerank(ci,k,False,cs)
torch.cuda.synchronize()
print("rerank with pruning", i4.cpu().numpy()[r], s4.cpu().numpy()[r], "k-th pass score", cs.cpu().numpy()[r][:3])
