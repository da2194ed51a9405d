"""Device-resident matcher: run_test.m:26-57 on signatures that already live in HBM, optionally with the
database row-sharded over the ranks of a torch.distributed group (one process per GPU, RCCL over xGMI).

PyTorch is plumbing here (device buffers, streams, the two small collectives); all arithmetic is in
libpr_amd.so through the C ABI (plain device pointers).

Sharding: every rank holds the DB rows [db_row0, db_row0 + n_local) and ALL queries.
  1. local: pack, distances (m x n_local), per-row fp64 moments (count, mean, M2) per channel   [HIP]
  2. all_gather of the moments  (m x 2 x 3 f64 per rank = 48 B per query)                             [RCCL]
  3. local: Chan-combine in rank order -> global mean/std, fused score, mask on GLOBAL indices,
     per-shard top-k (ties -> lower global index)                                                    [HIP]
  4. all_gather of (idx, score) (8k B per query per rank), k-way merge by (score, idx)               [RCCL + tiny sort]
With one rank steps 2 and 4 are skipped.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .api import Context


def _dptr(t: torch.Tensor):
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


class Matcher:
    def __init__(self, type_: str, max_queries: int, max_db: int, ctx: Context | None = None, device: int | None = None):
        self.type = {"sc": _lib.TYPE_SC, "m2dp": _lib.TYPE_M2DP, "delight": _lib.TYPE_DELIGHT}[type_]
        self.rows_per_sig, self.sig_len = {_lib.TYPE_SC: (1, 2400), _lib.TYPE_M2DP: (4, 384), _lib.TYPE_DELIGHT: (16, 256)}[self.type]
        self.plain = self.type == _lib.TYPE_DELIGHT      # one distance matrix, no z-score fusion (run_test.m:26-36)
        if device is None:
            device = torch.cuda.current_device()
        self.ctx = ctx or Context(device)
        self.dev = torch.device("cuda", self.ctx.device)
        self.lib = self.ctx.lib
        self.q = C.c_void_p()
        self.db = C.c_void_p()
        self.ctx.check(self.lib.pr_sigset_create(self.ctx.h, self.type, _lib.ROLE_QUERY, max_queries, C.byref(self.q)))
        self.ctx.check(self.lib.pr_sigset_create(self.ctx.h, self.type, _lib.ROLE_DB, max_db, C.byref(self.db)))
        self.max_queries, self.max_db = max_queries, max_db
        self.n = 0
        self._bufs = {}
        self.pre_distances = None      # optional callables (e.g. HIP event records) around the distance launch
        self.post_distances = None

    def close(self):
        if self.q:
            self.lib.pr_sigset_destroy(self.ctx.h, self.q)
            self.lib.pr_sigset_destroy(self.ctx.h, self.db)
            self.q = self.db = None

    def _buf(self, name, shape, dtype):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.dev)
            self._bufs[name] = t
        return t

    def _pack(self, handle, sig: torch.Tensor):
        assert sig.is_cuda and sig.is_contiguous() and sig.dim() == 2 and sig.shape[1] == self.sig_len
        assert sig.shape[0] % self.rows_per_sig == 0
        dt = {torch.float64: _lib.F64, torch.float32: _lib.F32}[sig.dtype]
        n = sig.shape[0] // self.rows_per_sig
        self.ctx.check(self.lib.pr_sigset_pack(self.ctx.h, handle, _dptr(sig), dt, _lib.DEVICE, n))
        return n

    def pack_database(self, sig: torch.Tensor):
        """processSC.m:18-20 (normalise hist2) + operand layout; sig: device [n(*4), sig_len] f64/f32."""
        torch.cuda.current_stream(self.dev).synchronize()
        self.n = self._pack(self.db, sig)

    def match(self, queries: torch.Tensor, mask_width: int = 0, p_weight: float = 2.0, k: int = 1,
              db_row0: int = 0, q_row0: int = 0, group=None):
        """Returns (idx int32 [m,k] GLOBAL DB row indices, score float32 [m,k]) as device tensors."""
        import torch.distributed as dist
        torch.cuda.current_stream(self.dev).synchronize()
        m = self._pack(self.q, queries)
        n = self.n
        G = dist.get_world_size(group) if (group is not None or (dist.is_available() and dist.is_initialized())) else 1
        d_p = self._buf("d_p", (m, n), torch.float32)
        d_i = None if self.plain else self._buf("d_i", (m, n), torch.float32)
        mom = self._buf("mom", (m, 2, 3), torch.float64)
        idx = self._buf("idx", (m, k), torch.int32)
        score = self._buf("score", (m, k), torch.float32)
        lib, h = self.lib, self.ctx.h
        if self.pre_distances:
            self.pre_distances()
        p_i = None if self.plain else _dptr(d_i)
        self.ctx.check(lib.pr_distances_dev(h, self.q, self.db, _dptr(d_p), p_i))
        if self.post_distances:
            self.post_distances()

        def local_moments():
            if self.plain:
                mom.zero_()
                torch.cuda.current_stream(self.dev).synchronize()
                return mom
            self.ctx.check(lib.pr_row_moments_dev(h, _dptr(d_p), _dptr(d_i), m, n, _dptr(mom)))
            if G > 1:
                self.ctx.sync()
            return mom

        def local_select(mom_all, G_):
            if G_ > 1:
                torch.cuda.current_stream(self.dev).synchronize()
            self.ctx.check(lib.pr_fuse_select_dev(h, _dptr(d_p), p_i, m, n, _dptr(mom_all), G_, q_row0, db_row0,
                                                  int(mask_width), float(p_weight), int(k), _dptr(idx), _dptr(score)))
            self.ctx.sync()
            return idx, score

        return sharded_topk(local_moments, local_select, k, group if G > 1 else None, G)

    def distances(self):
        """The last distance matrices (device, float32 [m, n_local])."""
        return self._bufs["d_p"], self._bufs.get("d_i")


class FusedMatcher:
    """BASELINE.json config 5 ("fused SC + M2DP scoring", build-defined: DESIGN.md §7): an SC and an M2DP matcher over the
    same places; the four row z-scores are added (weights p, 1, p, 1) in ONE top-k pass (pr_fuse_select2_dev).  Shards like
    Matcher: the moments of both descriptor types travel in the same all_gather ([m, 4, 3] f64 per rank)."""

    def __init__(self, max_queries: int, max_db: int, ctx: Context | None = None, device: int | None = None):
        self.sc = Matcher("sc", max_queries, max_db, ctx, device)
        self.m2 = Matcher("m2dp", max_queries, max_db, self.sc.ctx)
        self.ctx, self.dev, self.lib = self.sc.ctx, self.sc.dev, self.sc.lib

    def close(self):
        self.sc.close(); self.m2.close()

    def pack_database(self, sc_sig: torch.Tensor, m2dp_sig: torch.Tensor):
        self.sc.pack_database(sc_sig); self.m2.pack_database(m2dp_sig)
        assert self.sc.n == self.m2.n, "the two databases must describe the same places"

    def match(self, sc_queries: torch.Tensor, m2dp_queries: torch.Tensor, mask_width: int = 0, p_weight: float = 2.0, k: int = 1,
              db_row0: int = 0, q_row0: int = 0, group=None):
        import torch.distributed as dist
        torch.cuda.current_stream(self.dev).synchronize()
        m = self.sc._pack(self.sc.q, sc_queries)
        assert self.m2._pack(self.m2.q, m2dp_queries) == m
        n = self.sc.n
        G = dist.get_world_size(group) if (group is not None or (dist.is_available() and dist.is_initialized())) else 1
        lib, h = self.lib, self.ctx.h
        d = [self.sc._buf("d_p", (m, n), torch.float32), self.sc._buf("d_i", (m, n), torch.float32),
             self.m2._buf("d_p", (m, n), torch.float32), self.m2._buf("d_i", (m, n), torch.float32)]
        mom = [self.sc._buf("mom", (m, 2, 3), torch.float64), self.m2._buf("mom", (m, 2, 3), torch.float64)]
        idx = self.sc._buf("idx", (m, k), torch.int32)
        score = self.sc._buf("score", (m, k), torch.float32)
        self.ctx.check(lib.pr_distances_dev(h, self.sc.q, self.sc.db, _dptr(d[0]), _dptr(d[1])))
        self.ctx.check(lib.pr_distances_dev(h, self.m2.q, self.m2.db, _dptr(d[2]), _dptr(d[3])))

        def local_moments():
            self.ctx.check(lib.pr_row_moments_dev(h, _dptr(d[0]), _dptr(d[1]), m, n, _dptr(mom[0])))
            self.ctx.check(lib.pr_row_moments_dev(h, _dptr(d[2]), _dptr(d[3]), m, n, _dptr(mom[1])))
            self.ctx.sync()
            return torch.cat(mom, dim=1)                                   # [m, 4, 3]

        def local_select(mom_all, G_):
            mom_all = mom_all.reshape(G_, m, 4, 3)
            m1, m2 = mom_all[:, :, :2].contiguous(), mom_all[:, :, 2:].contiguous()
            torch.cuda.current_stream(self.dev).synchronize()
            self.ctx.check(lib.pr_fuse_select2_dev(h, _dptr(d[0]), _dptr(d[1]), _dptr(d[2]), _dptr(d[3]), m, n, _dptr(m1), _dptr(m2), G_,
                                                   q_row0, db_row0, int(mask_width), float(p_weight), int(k), _dptr(idx), _dptr(score)))
            self.ctx.sync()
            return idx, score

        return sharded_topk(local_moments, local_select, k, group if G > 1 else None, G)


def sharded_topk(local_moments, local_select, k: int, group, G: int):
    """The exchange protocol of SURVEY.md §8-e around two local callables (HIP in production; a numpy stand-in in
    the gloo CPU tests): moments -> all_gather -> select with the moments of all shards -> all_gather -> merge."""
    import torch.distributed as dist
    mom = local_moments()
    if G == 1:
        return local_select(mom, 1)
    stage_on_host = dist.get_backend(group) == "gloo"   # gloo has no device all_gather: used by the single-GPU tests

    def gather(t):   # list form: identical semantics on nccl (RCCL) and gloo
        src = t.contiguous()
        if stage_on_host and src.is_cuda:
            src = src.cpu()
        outs = [torch.empty_like(src) for _ in range(G)]
        dist.all_gather(outs, src, group=group)
        return torch.stack(outs).to(t.device)

    mom_all = gather(mom)
    idx, score = local_select(mom_all, G)
    idx_all, sc_all = gather(idx), gather(score)
    return merge_topk(idx_all, sc_all, k)


def merge_topk(idx_all: torch.Tensor, sc_all: torch.Tensor, k: int):
    """k-way merge of per-shard top-k lists [G, m, k] by (score, index) ascending; -1 / NaN entries sort last."""
    G, m, kk = idx_all.shape
    idx = idx_all.permute(1, 0, 2).reshape(m, G * kk).to(torch.int64)
    sc = sc_all.permute(1, 0, 2).reshape(m, G * kk).to(torch.float64)
    bad = (idx < 0) | torch.isnan(sc)
    sc = torch.where(bad, torch.full_like(sc, float("inf")), sc)
    idx_key = torch.where(bad, torch.full_like(idx, 2 ** 62), idx)
    o1 = torch.argsort(idx_key, dim=1, stable=True)
    sc1 = torch.gather(sc, 1, o1)
    o2 = torch.argsort(sc1, dim=1, stable=True)
    order = torch.gather(o1, 1, o2)[:, :k]
    out_idx = torch.gather(idx, 1, order)
    out_sc = torch.gather(sc_all.permute(1, 0, 2).reshape(m, G * kk), 1, order)
    out_bad = torch.gather(bad, 1, order)
    out_idx = torch.where(out_bad, torch.full_like(out_idx, -1), out_idx)
    return out_idx.to(torch.int32), out_sc


def combine_moments(mom_all: np.ndarray):
    """Host restatement of the rank-order Chan combination done inside fuse_select (for tests): [G,m,2,3] -> mean,std."""
    G = mom_all.shape[0]
    cn = np.zeros(mom_all.shape[1:3]); mean = np.zeros_like(cn); m2 = np.zeros_like(cn)
    for g in range(G):
        nb, mb, m2b = mom_all[g, ..., 0], mom_all[g, ..., 1], mom_all[g, ..., 2]
        tot = cn + nb
        delta = mb - mean
        mean = mean + delta * (nb / tot)
        m2 = m2 + m2b + delta * delta * (cn * nb / tot)
        cn = tot
    return mean, np.sqrt(m2 / (cn - 1.0))
