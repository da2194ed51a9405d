"""world_size-2 gloo test of the sharded matching protocol on CPU: the two collectives and the
merge of matcher.sharded_topk around a numpy stand-in for the two local HIP steps (tests only), checked against
the oracle run on the unsharded DB."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, m, k, mask, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from so_dso_place_recognition_amd import synth
    from so_dso_place_recognition_amd.matcher import combine_moments, sharded_topk
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = n * rank // world, n * (rank + 1) // world
    db = synth.sc_database(45, hi - lo, first=lo)
    queries, _ = synth.sc_queries(46, db, m, db_first=lo, n_global=n, db_seed=45)
    rc, dp, di = oracle_lib.sc_distance(queries, db)              # stand-in for the local HIP distance kernels

    def local_moments():
        mom = np.empty((m, 2, 3))
        for ch, d in enumerate((dp, di)):
            mean = d.mean(1)
            mom[:, ch, 0] = d.shape[1]; mom[:, ch, 1] = mean; mom[:, ch, 2] = ((d - mean[:, None]) ** 2).sum(1)
        return torch.from_numpy(mom)

    def local_select(mom_all, G):
        ma = mom_all.numpy()
        mean, std = combine_moments(ma)
        f = 2.0 * ((dp - mean[:, :1]) / std[:, :1]) + (di - mean[:, 1:]) / std[:, 1:]
        jg = lo + np.arange(hi - lo)[None, :]
        f = np.where(np.abs(np.arange(m)[:, None] - jg) < mask, np.inf, f)
        order = np.argsort(f, axis=1, kind="stable")[:, :k]
        idx = (order + lo).astype(np.int32)
        sc = np.take_along_axis(f, order, 1)
        return torch.from_numpy(idx), torch.from_numpy(sc)

    idx, sc = sharded_topk(local_moments, local_select, k, None, world)
    # the full protocol (steps 5 - 7 of matcher.py) with stand-ins: owner-wise re-evaluation blocks [m, 5, kin], finish, and the
    # two exchanges of the exact-row resolution - every rank's [m, 4, 3] moments and [64, 2, k] list must arrive at its rank's slot on every rank
    seen = {}

    def rerank(cand_idx, kk, partial, cand_sc):
        assert partial
        ci = cand_idx.numpy()
        own = (ci >= lo) & (ci < hi)
        p5 = np.full((m, 5, ci.shape[1]), np.nan)
        p5[:, 0][own] = cand_sc.numpy()[own]                 # (the stand-in "exact" score = the pass score)
        p5[:, 1:][np.repeat(own[:, None, :], 4, 1)] = 0.25
        return torch.from_numpy(p5)

    def finish(cand_idx, cand_sc, part_all, kk):
        assert cand_sc.shape == cand_idx.shape
        pa = part_all.numpy()
        assert pa.shape == (world, m, 5, cand_idx.shape[1])
        owners = (~np.isnan(pa[:, :, 0])).sum(0)
        assert (owners[cand_idx.numpy() >= 0] == 1).all()    # every candidate has exactly one owner
        sc_ = np.nanmax(np.where(np.isnan(pa[:, :, 0]), -np.inf, pa[:, :, 0]), axis=0)
        return cand_idx[:, :kk].clone(), torch.from_numpy(sc_[:, :kk].copy())

    def exact():
        return torch.full((m, 4, 3), float(rank + 1), dtype=torch.float64)

    def select(exact_all, kk):                                 # this rank's k best of the flagged queries' exact rows: [64, 2, k]
        seen["exact"] = exact_all.numpy().copy()
        return torch.full((64, 2, kk), float(10 * (rank + 1)), dtype=torch.float64)

    def merge(sel_all, kk, i_, s_):
        seen["sel"] = sel_all.numpy().copy()
        return i_, s_

    idx2, sc2 = sharded_topk(local_moments, local_select, k, None, world, rerank=rerank, finish=finish, resolve=(exact, select, merge))
    assert seen["exact"].shape == (world, m, 4, 3) and all((seen["exact"][g] == g + 1).all() for g in range(world))
    assert seen["sel"].shape == (world, 64, 2, k) and all((seen["sel"][g] == 10 * (g + 1)).all() for g in range(world))
    assert torch.equal(idx2, idx) and np.abs(sc2.numpy() - sc.numpy()).max() == 0.0
    # the Matcher's form of step 7: a fourth entry gives the number of passes (a stream-ordered call: ceil(m / 64), whatever is flagged); every
    # pass runs both exchanges with its offset, and only the LAST one is declared last (what may raise PR_WARN_ORDER_UNRESOLVED in the library)
    calls = []

    def exact_p(off, last):
        calls.append(("exact", off, last))
        return torch.full((m, 4, 3), float(rank + 1), dtype=torch.float64)

    def select_p(exact_all, kk, off=0):
        calls.append(("select", off))
        return torch.full((64, 2, kk), float(off), dtype=torch.float64)

    def merge_p(sel_all, kk, i_, s_, off=0):
        assert (sel_all.numpy() == float(off)).all() and sel_all.shape[0] == world
        calls.append(("merge", off))
        return i_, s_

    idx3, sc3 = sharded_topk(local_moments, local_select, k, None, world, rerank=rerank, finish=finish, resolve=(exact_p, select_p, merge_p, lambda: 3))
    assert calls == [("exact", 0, False), ("select", 0), ("merge", 0), ("exact", 64, False), ("select", 64), ("merge", 64),
                     ("exact", 128, True), ("select", 128), ("merge", 128)]
    assert torch.equal(idx3, idx)
    if rank == 0:
        q.put((idx.numpy(), sc.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mask,k", [(0, 1), (4, 3)])
def test_sharded_topk_matches_unsharded_oracle(mask, k):
    import oracle_lib
    from so_dso_place_recognition_amd import synth
    n, m, world = 90, 12, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + mask
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, m, k, mask, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _q
    res = None
    for _ in range(120):
        try:
            res = q.get(timeout=1)
            break
        except _q.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    assert res is not None, 'a rank failed'
    idx, sc = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    db = synth.sc_database(45, n)
    queries, _ = synth.sc_queries(46, db, m)
    rc, oidx, osc = oracle_lib.match_topk(0, queries, db, mask, 2.0, k)
    assert np.array_equal(idx, oidx)
    assert np.abs(sc - osc).max() < 1e-9


def test_merge_topk_ties_and_padding():
    from so_dso_place_recognition_amd.matcher import merge_topk
    idx = torch.tensor([[[5, 9, -1]], [[2, 7, 11]]], dtype=torch.int32)           # [G=2, m=1, k=3]
    sc = torch.tensor([[[0.5, 1.0, float("nan")]], [[0.5, 1.0, 3.0]]], dtype=torch.float64)
    i, s = merge_topk(idx, sc, 3)
    assert i.tolist() == [[2, 5, 7]] and s.tolist() == [[0.5, 0.5, 1.0]]       # ties -> lower global index
    i, s = merge_topk(idx, sc, 6)
    assert i.tolist()[0][:5] == [2, 5, 7, 9, 11] and i.tolist()[0][5] == -1


def test_bench_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's command) must start two ranks itself.  Without a GPU
    every rank refuses loudly (there is no CPU path): the refusal - not an assertion about WORLD_SIZE - is what comes back."""
    import subprocess
    import torch as _t
    if _t.cuda.is_available():
        pytest.skip("CPU-only check; the GPU form is test_bench_two_ranks_on_one_gpu")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo",
                        "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    # (the launcher ends the other rank as soon as one has failed: one or two refusals)
    assert (r.stdout + r.stderr).count("bench.py needs an MI355X") >= 1 and "AssertionError" not in r.stderr
    assert "local_rank" in r.stderr                                  # the launcher's failure report: ranks were started


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["f16x2", "f16"])
def test_bench_two_ranks_on_one_gpu(tmp_path, arith):
    """The real sharded GPU path (Matcher + libpr_amd.so) with 2 ranks sharing cuda:0 over gloo: the merged top-1 of the
    row-sharded DB must equal the planted ground truth, exactly like the single-rank run.  f16: the single-product arithmetic through the
    same exchanges (k + 56 candidates per shard, margin check on the merged list)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # the driver's plain command: no launcher around it, bench.py starts its own ranks (bench.py::self_launch)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "0", "--db", "6001", "--queries", "130", "--backend", "gloo", "--no-cpu-baseline", "--no-extra", "--sc-arith", arith]
    if arith == "f16":   # and the driver's launcher form (python -m torch.distributed.run ... bench.py --gpus 2) for the other arithmetic
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29600 + os.getpid() % 300)] + cmd[1:]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["parity"]["planted_top1_correct"] == 130 and d["scaling"] == "strong"
    # the per-rank breakdown that makes a multi-GPU curve readable: one entry per rank, every phase of the protocol with its own time,
    # the all-gathers separated from the compute, the shard's share of the DB and the matcher's time per pair
    pr_ = d["per_rank"]
    assert [r["rank"] for r in pr_] == [0, 1] and sum(r["db_rows"] for r in pr_) == 6001
    for r in pr_:
        ph = r["phases_ms"]
        assert {"pack(db)", "pack+distances+moments", "all_gather A (moments)", "select", "all_gather B (candidates)", "merge+rerank",
                "all_gather C (evaluations)", "finish+checks"} <= set(ph)
        if arith != "f16":      # (the single-product arithmetic hands its flagged queries to a split-f16 twin instead of the exact-row exchange)
            assert "flagged count" in ph       # 130 queries > 64: the count is read back; nothing flagged - no exact-row pass, no all-gather D / E
            assert not ({"exact rows", "all_gather D (exact moments)", "exact select", "all_gather E (exact lists)", "exact merge"} & set(ph))
        assert all(v >= 0 for v in ph.values()) and abs(r["collective_ms"] + r["compute_ms"] - sum(ph.values())) < 1e-6
        assert 0.49 < r["shard_fraction"] < 0.51 and r["matcher_ns_per_pair"] > 0 and r["step_ms"] >= sum(ph.values()) * 0.99
    assert d["collective"]["world"] == 2
    if arith != "f16":          # every query flagged (PR_FORCE_ORDER_FLAGS): three passes of the exact-row exchange, each phase on the line
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, PR_FORCE_ORDER_FLAGS="1"), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        df = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert df["parity"]["planted_top1_correct"] == 130
        for r_ in df["per_rank"]:
            assert {"flagged count", "exact rows", "all_gather D (exact moments)", "exact select", "all_gather E (exact lists)",
                    "exact merge"} <= set(r_["phases_ms"])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--db", "6001",
                          "--queries", "130", "--no-cpu-baseline", "--no-extra", "--sc-arith", arith], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d1["parity"]["planted_top1_correct"] == 130
