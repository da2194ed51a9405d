#!/usr/bin/env python3
"""Experiment: row moments + fused selection of a 4096 x 100k step in row batches small enough for the Infinity Cache
(256 MB), so that the selection's sweep of a batch re-reads what the moments pass just pulled through it, against the two whole-matrix passes.
usage: python tools/exp_mall_batches.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from so_dso_place_recognition_amd import api, synth, _lib
from so_dso_place_recognition_amd.matcher import Matcher, _dptr
n, m = 100000, 4096
dev = torch.device("cuda", 0)
db = synth.sc_database_torch(45, n, device=dev)
q_h, planted = synth.sc_queries(46, np.empty((0, 2400)), m, db_first=0, n_global=n, db_seed=45)
q = torch.from_numpy(q_h).to(dev)
mt = Matcher("sc", m, n, ctx=api.Context(0, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
mt.pack_database(db)
mt.local_phase1(q)
d_p, d_i = mt.distances()
lib, h = mt.lib, mt.ctx.h
kin = 9
mom = torch.empty((m, 2, 3), dtype=torch.float64, device=dev)
idx = torch.empty((m, kin), dtype=torch.int32, device=dev); sc32 = torch.empty((m, kin), dtype=torch.float32, device=dev); sc64 = torch.empty((m, kin), dtype=torch.float64, device=dev)
def run(B):
    for r0 in range(0, m, B):
        r1 = min(m, r0 + B)
        mt.ctx.check(lib.pr_row_moments_dev(h, C.c_void_p(d_p.data_ptr() + r0 * n * 4), C.c_void_p(d_i.data_ptr() + r0 * n * 4), r1 - r0, n, C.c_void_p(mom.data_ptr() + r0 * 48)))
        if B < m:
            mt.ctx.check(lib.pr_fuse_select_f64_dev(h, C.c_void_p(d_p.data_ptr() + r0 * n * 4), C.c_void_p(d_i.data_ptr() + r0 * n * 4), r1 - r0, n, C.c_void_p(mom.data_ptr() + r0 * 48), 1, r0, 0, 0, 2.0, kin,
                                                    C.c_void_p(idx.data_ptr() + r0 * kin * 4), C.c_void_p(sc32.data_ptr() + r0 * kin * 4), C.c_void_p(sc64.data_ptr() + r0 * kin * 8)))
    if B >= m:
        mt.ctx.check(lib.pr_fuse_select_f64_dev(h, _dptr(d_p), _dptr(d_i), m, n, _dptr(mom), 1, 0, 0, 0, 2.0, kin, _dptr(idx), _dptr(sc32), _dptr(sc64)))
ref = None
for B in (4096, 1024, 512, 256, 192, 128, 64):
    for _ in range(3):
        run(B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run(B)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 10
    got = idx.cpu().numpy().copy()
    if ref is None:
        ref = got
    print(f"rows per batch {B:5d}: moments + select {ms:7.3f} ms   same result: {np.array_equal(got, ref)}  top1 ok {int((got[:, 0] == planted).sum())}", flush=True)
