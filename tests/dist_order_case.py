#!/usr/bin/env python3
"""Worker of test_order_that_hangs_on_the_row_sigmas_is_resolved_with_fp64_statistics (case "order": the 96 x 1096 M2DP case), of
test_near_copy_clusters_are_answered_from_the_exact_row (cases "cluster_sc" / "cluster_m2dp": tests/helpers.near_copy_clusters) and of
test_fp64_statistics_for_every_query_give_the_oracles_scores (case "forced") through the
sharded Matcher under torch.distributed (gloo, all ranks on cuda:0).  Rank 0 prints one JSON line: the returned indices / scores of every
query and the PR_WARN_* bits of its context.
usage: python -m torch.distributed.run --nproc-per-node W tests/dist_order_case.py [arith] [case] [k]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

from so_dso_place_recognition_amd import api, synth
from so_dso_place_recognition_amd.matcher import Matcher

arith = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
case = sys.argv[2] if len(sys.argv) > 2 else "order"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mask = 0
if case == "order":
    type_, m, n, k = "m2dp", 96, 1096, 35
    db = synth.m2dp_database(3000 + 7 * 1 + 1, n)
    q, _ = synth.m2dp_queries(4000 + 1 + 1, db, m)
elif case == "forced":           # test_fp64_statistics_for_every_query_give_the_oracles_scores: PR_FORCE_ORDER_FLAGS=1 in the environment, 150 flagged queries = three passes
    type_, m, n, k, mask = "sc", 150, 700, 3, 5
    db = synth.sc_database(81, n)
    q, _ = synth.sc_queries(181, db, m)
else:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    type_ = case.split("_")[1]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    n, m, rows, copies = helpers.CLUSTER_CASE
    db, q, _, _ = helpers.near_copy_clusters(type_, n, m, rows, copies)
rps = 4 if type_ == "m2dp" else 1
lo, hi = n * rank // world, n * (rank + 1) // world
mt = Matcher(type_, m, hi - lo, ctx=api.Context(0, sc_arith=arith, stream=int(torch.cuda.current_stream(dev).cuda_stream)))
mt.pack_database(torch.from_numpy(db[rps * lo: rps * hi]).to(dev))
idx, sc = mt.match(torch.from_numpy(q).to(dev), mask, 2.0, k, db_row0=lo)
torch.cuda.synchronize()
w = mt.take_warnings()
if rank == 0:
    print(json.dumps({"idx": idx.cpu().numpy().tolist(), "score": sc.cpu().numpy().tolist(), "warnings": int(w), "world": world}), flush=True)
dist.barrier()
mt.close()
dist.destroy_process_group()
