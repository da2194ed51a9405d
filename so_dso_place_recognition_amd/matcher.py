"""Device-resident matcher: run_test.m:26-57 on signatures that already live in HBM, optionally with the
database row-sharded over the ranks of a torch.distributed group (one process per GPU, RCCL over xGMI).

PyTorch is plumbing here (device buffers, streams, the two small collectives); all arithmetic is in
libpr_amd.so through the C ABI (plain device pointers).  The library context is created ON torch's current
stream (pr_create_on_stream), so the library's kernels, torch's allocations and RCCL's collectives are ordered
by the stream itself: a step has no host synchronisation at all.

Sharding: every rank holds the DB rows [db_row0, db_row0 + n_local) and ALL queries.
  1. local: pack, distances (m x n_local), per-row fp64 moments (count, mean, M2) per channel          [HIP]
  2. all_gather_into_tensor of the moments  (m x 2 x 3 f64 per rank = 48 B per query)                        [RCCL]
     (all-gather + Chan combination in rank order instead of §8-e's all-reduce: same bytes at this size, and the
     result does not depend on the reduction order RCCL happens to pick - every rank computes the same bits)
  3. local: Chan-combine -> global mean/std, fused fp32 score, mask on GLOBAL indices, per-shard top-(k+8)
     (ties -> lower global index)                                                                      [HIP]
  4. all_gather_into_tensor of the per-shard (idx i32, fp32 score) lists, k-way merge on the device by
     (score, idx) (pr_merge_topk_dev) -> the GLOBAL top-(k+8) candidates, the same list an unsharded run
     selects                                                                                           [RCCL + HIP]
  5. local: fp64 re-evaluation, from the raw signatures, of the candidates that lie in this shard
     (pr_rerank_partial_dev; on average (k+8)/G pairs per query: the cost does not grow with G)          [HIP]
  6. all_gather_into_tensor of the shards' evaluations (p5 blocks: score + 4 exact channel distances per candidate, 40 (k+8) B per
     query per rank), every candidate's score taken from its owner, the k best by (score, idx) and the order check
     (pr_rerank_finish_dev)                                                                             [RCCL + HIP]
  7. queries the re-evaluated candidates cannot answer for certain - their order hangs on the fp32 pass's sigmas, or the k + 8 candidates do
     not provably hold the top-k (near-copies of one place) - none, as a rule: the kernels leave at once.  Such a query is answered from
     its EXACT ROW (run_test.m:38-57 are fp64 over the whole row): this shard's fp64 distances to all of its entries and their moments
     (pr_order_exact_moments_dev), all_gather_into_tensor (96 B per query per rank), the shard's k best under the statistics of the whole
     row (pr_order_exact_select_dev), all_gather_into_tensor (64 x 16 k B per rank), merge on every rank (pr_order_exact_merge_dev)
                                                                                                [HIP + RCCL + HIP + RCCL + HIP]
With one rank steps 2, 4, 6 and 7's gathers are skipped (pr_rerank_dev does 5 + the selection, pr_order_resolve_async_dev does 7).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .api import Context


def _dptr(t: torch.Tensor | None):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _dptr_mom(mt, sc: bool):
    """The combined moments tensor of a Matcher as the (mom_sc | mom_m2) argument it belongs to (the other one is None)."""
    return mt._mom_all if (mt.type == _lib.TYPE_SC) == sc else None


def _torch_dt(t: torch.Tensor) -> int:
    return {torch.float64: _lib.F64, torch.float32: _lib.F32}[t.dtype]


def _stream_context(device: int, **kw) -> Context:
    """A library context whose kernels run on torch's current stream of that device."""
    return Context(device, stream=int(torch.cuda.current_stream(device).cuda_stream), **kw)


class _Base:
    resolved = None       # queries the last single-shard match() of more than 64 queries answered from their exact rows (it reads the count back)

    def _init_ctx(self, ctx, device):
        if device is None:
            device = ctx.device if ctx is not None else torch.cuda.current_device()
        self.ctx = ctx or _stream_context(device)
        self.dev = torch.device("cuda", self.ctx.device)
        self.lib = self.ctx.lib
        self._lib_stream = None if self.ctx.stream == 0 else torch.cuda.ExternalStream(self.ctx.stream, device=self.dev)

    # The library's kernels run on self.ctx.stream; torch work (casts, zero_(), all_gather_into_tensor, output allocations) runs on torch's
    # CURRENT stream, which may differ from the one the context was created on (`with torch.cuda.stream(s)`, DDP side streams).  The two
    # are compared at every call: the same stream needs nothing, different streams are joined by events (no host wait).
    def _same_stream(self) -> bool:
        return self.ctx.stream == int(torch.cuda.current_stream(self.dev).cuda_stream)

    def _enter(self):      # torch work issued so far must be visible to the library's stream
        if self._same_stream():
            return
        if self._lib_stream is not None:
            self._lib_stream.wait_stream(torch.cuda.current_stream(self.dev))
        else:
            torch.cuda.current_stream(self.dev).synchronize()

    def _leave(self):      # ... and the library's work to torch's
        if self._same_stream():
            return
        if self._lib_stream is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self._lib_stream)
        else:
            self.ctx.sync()

    # ---- PR_SC_ARITH_F16 (single f16 product per term): exact indices need a margin check of the candidate list and, for the
    # queries that fail it, a second pass in split-f16 (include/place_recognition.h)
    @property
    def f16(self) -> bool:
        return self.ctx.sc_arith == "f16"

    def _margin(self, mom_sc, mom_m2, G, p_weight, cand_sc: torch.Tensor, k: int, score: torch.Tensor):
        """pr_f16_margin_dev -> (flags int32 [m], count int32 [1]) device tensors; no host synchronisation."""
        m, kin = cand_sc.shape
        flags = torch.empty((m,), dtype=torch.int32, device=self.dev)
        count = torch.empty((1,), dtype=torch.int32, device=self.dev)
        self._enter()
        self.ctx.check(self.lib.pr_f16_margin_dev(self.ctx.h, _dptr(mom_sc), _dptr(mom_m2), m, G, float(p_weight), kin, _dptr(cand_sc.contiguous()),
                                                  int(k), _dptr(score.contiguous()), _dptr(flags), _dptr(count)))
        self._leave()
        self.f16_flags, self.f16_count = flags, count
        return flags, count

    def _resolve_order(self, raw6, mom_sc, mom_m2, m, n, q_row0, mask_width, p_weight, k, idx, score, exact_order=True):
        """After a single-shard pr_rerank_dev: queries whose re-evaluated order hangs on the fp32 pass's sigmas, or whose candidate list does
        not provably hold the top-k, are answered from their exact fp64 rows; idx / score (and the moments rows) are patched in place.
        One pass takes RESOLVE_SLOTS flagged queries.  Calls of up to that many queries (and exact_order == "async"): pr_order_resolve_async_dev,
        ceil(m / 64) stream-ordered passes without host synchronisation - all flagged queries, empty passes leave at once
        (PR_WARN_ORDER_RESOLVED at ctx.take_warnings() tells whether it happened).  Larger calls: pr_order_resolve_dev - reads the number of
        flagged queries back (ONE synchronisation of the stream per call) and runs as many passes as it takes."""
        self._enter()
        if m <= RESOLVE_SLOTS or exact_order == "async":
            self.ctx.check(self.lib.pr_order_resolve_async_dev(self.ctx.h, *raw6, _dptr(mom_sc), _dptr(mom_m2), m, n, int(q_row0), int(mask_width),
                                                               float(p_weight), int(k), _dptr(idx), _dptr(score)))
        else:
            cnt = C.c_int32(0)
            self.ctx.check(self.lib.pr_order_resolve_dev(self.ctx.h, *raw6, _dptr(mom_sc), _dptr(mom_m2), m, n, int(q_row0), int(mask_width),
                                                         float(p_weight), int(k), _dptr(idx), _dptr(score), C.byref(cnt)))
            self.resolved = int(cnt.value)
        self._leave()

    def _exact_passes(self, m, exact_order=True):
        """Passes of RESOLVE_SLOTS flagged queries step 7 of the sharded protocol needs (the same number on every rank: the flags are a
        function of the gathered evaluations).  Up to RESOLVE_SLOTS queries: one, unconditionally and without synchronisation; exact_order ==
        "async" (a captured graph): ceil(m / 64), every one of them with its two all-gathers, whatever was flagged - empty passes leave at once;
        otherwise the flagged count is read back (one synchronisation) - none flagged, no pass and no all-gather."""
        if m <= RESOLVE_SLOTS:
            return 1
        if exact_order == "async":
            return (m + RESOLVE_SLOTS - 1) // RESOLVE_SLOTS
        return (self.flagged_count() + RESOLVE_SLOTS - 1) // RESOLVE_SLOTS

    def _fallback_rows(self, run_rows, idx, score, mask_width, q_row0):
        """Recomputes the flagged queries through `run_rows(rows tensor, q_row0 or None)` (a split-f16 matcher over the same DB) and
        patches idx / score.  Reads the flag count back: one host synchronisation per call, only in the f16 arithmetic."""
        cnt = int(self.f16_count.item())
        self.f16_fallbacks = cnt
        if cnt == 0:
            return idx, score
        rows = torch.nonzero(self.f16_flags, as_tuple=False).flatten()
        idx, score = idx.clone(), score.clone()
        if mask_width <= 0:                       # the mask does not look at the query's row number: one batch
            i2, s2 = run_rows(rows, None)
            idx[rows], score[rows] = i2, s2
        else:
            for r in rows.tolist():
                i2, s2 = run_rows(rows.new_tensor([r]), q_row0 + r)
                idx[r], score[r] = i2[0], s2[0]
        return idx, score


class Matcher(_Base):
    def __init__(self, type_: str, max_queries: int, max_db: int, ctx: Context | None = None, device: int | None = None):
        self.type = {"sc": _lib.TYPE_SC, "m2dp": _lib.TYPE_M2DP, "delight": _lib.TYPE_DELIGHT}[type_]
        self.rows_per_sig, self.sig_len = {_lib.TYPE_SC: (1, 2400), _lib.TYPE_M2DP: (4, 384), _lib.TYPE_DELIGHT: (16, 256)}[self.type]
        self.plain = self.type == _lib.TYPE_DELIGHT      # one distance matrix, no z-score fusion (run_test.m:26-36)
        self._init_ctx(ctx, device)
        self._max_q, self._max_db = max_queries, max_db
        self.q = C.c_void_p()
        self.db = C.c_void_p()
        self.ctx.check(self.lib.pr_sigset_create(self.ctx.h, self.type, _lib.ROLE_QUERY, max_queries, C.byref(self.q)))
        self.ctx.check(self.lib.pr_sigset_create(self.ctx.h, self.type, _lib.ROLE_DB, max_db, C.byref(self.db)))
        self.max_queries, self.max_db = max_queries, max_db
        self.n = 0
        self.db_sig = None             # the raw DB shard (the fp64 re-evaluation reads it)
        self._bufs = {}
        self._flat = {}
        self.pre_distances = None      # optional callables (e.g. HIP event records) around the distance launch
        self.post_distances = None

    def close(self):
        if getattr(self, "_twin", None) is not None:
            self._twin.close()
            self._twin = None
        if self.q:
            self.lib.pr_sigset_destroy(self.ctx.h, self.q)
            self.lib.pr_sigset_destroy(self.ctx.h, self.db)
            self.q = self.db = None

    def _buf(self, name, shape, dtype):
        """The call's working tensors, kept between calls.  A shape that grows a little per call (a DB that gains a row per keyframe: the
        [m, n] distance matrices) is served as a view of flat storage with an eighth of headroom instead of a new allocation per call."""
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            need = 1
            for d in shape:
                need *= int(d)
            flat = self._flat.get(name)
            if flat is None or flat.dtype != dtype or flat.numel() < need:
                flat = torch.empty(need + (need >> 3 if getattr(self, "_raw", None) is not None else 0), dtype=dtype, device=self.dev)
                self._flat[name] = flat
            t = flat[:need].view(shape)
            self._bufs[name] = t
        return t

    def _pack(self, handle, sig: torch.Tensor):
        assert sig.is_cuda and sig.is_contiguous() and sig.dim() == 2 and sig.shape[1] == self.sig_len
        assert sig.shape[0] % self.rows_per_sig == 0
        n = sig.shape[0] // self.rows_per_sig
        self.ctx.check(self.lib.pr_sigset_pack(self.ctx.h, handle, _dptr(sig), _torch_dt(sig), _lib.DEVICE, n))
        return n

    def pack_database(self, sig: torch.Tensor):
        """processSC.m:18-20 (normalise hist2) + operand layout; sig: device [n(*4), sig_len] f64/f32 (kept: the
        re-evaluation of the selected pairs reads the raw rows)."""
        self._enter()
        self.n = self._pack(self.db, sig)
        self.db_sig = sig
        self._raw = None

    def reserve_database(self, sig: torch.Tensor | None = None):
        """A DB that grows (SC/test_sc.cpp:40-56: one signature per keyframe): the operand image is laid out for the matcher's CAPACITY
        (pr_sigset_reserve) and the raw rows live in a capacity-sized buffer of the matcher, so append_database() adds rows in place - no
        re-pack, no re-upload.  sig (optional): the rows to start with, device [n(*4), sig_len] f64."""
        assert self.type in (_lib.TYPE_SC, _lib.TYPE_M2DP)
        self._enter()
        self.ctx.check(self.lib.pr_sigset_reserve(self.ctx.h, self.db))
        self._raw = torch.empty((self._max_db * self.rows_per_sig, self.sig_len), dtype=torch.float64, device=self.dev)
        self.n = 0
        if sig is not None and sig.shape[0]:
            assert sig.dtype == torch.float64
            r = sig.shape[0]
            self._raw[:r].copy_(sig)
            self._enter()
            self.n = self._pack(self.db, self._raw[:r])
        self.db_sig = self._raw[:self.n * self.rows_per_sig]       # (a view: the same storage as the rows appended later)

    def append_database(self, sig: torch.Tensor):
        """Rows [n, n + n_new) of the growing DB: the raw rows into the matcher's buffer, their operand rows into the image
        (pr_sigset_append: one kernel, bit for bit what a pack of all rows would write there)."""
        assert getattr(self, "_raw", None) is not None, "reserve_database() first"
        assert sig.is_cuda and sig.dim() == 2 and sig.shape[1] == self.sig_len and sig.dtype == torch.float64 and sig.shape[0] % self.rows_per_sig == 0
        k, r0 = sig.shape[0] // self.rows_per_sig, self.n * self.rows_per_sig
        assert self.n + k <= self._max_db
        dst = self._raw[r0:r0 + sig.shape[0]]
        dst.copy_(sig)
        self._enter()                                              # (the copy above is torch's: ordered in front of the library's kernel)
        self.ctx.check(self.lib.pr_sigset_append(self.ctx.h, self.db, _dptr(dst), 0, _lib.DEVICE, k))
        self.n += k
        self.db_sig = self._raw[:self.n * self.rows_per_sig]       # (a new view object: a split-f16 twin of the f16 arithmetic is re-packed when next needed)

    def local_phase1(self, queries: torch.Tensor):
        """pack(q) + distances + row moments of this shard -> moments [m, 2, 3] f64 (zeros for DELIGHT)."""
        self._enter()
        m = self._pack(self.q, queries)
        n = self.n
        self._q_sig, self._m = queries, m
        d_p = self._buf("d_p", (m, n), torch.float32)
        d_i = None if self.plain else self._buf("d_i", (m, n), torch.float32)
        mom = self._buf("mom", (m, 2, 3), torch.float64)
        lib, h = self.lib, self.ctx.h
        if self.pre_distances:
            self.pre_distances()
        self.ctx.check(lib.pr_distances_dev(h, self.q, self.db, _dptr(d_p), _dptr(d_i)))
        if self.post_distances:
            self.post_distances()
        if self.plain:
            self._leave()
            mom.zero_()
        else:
            self.ctx.check(lib.pr_row_moments_dev(h, _dptr(d_p), _dptr(d_i), m, n, _dptr(mom)))
            self._leave()
        return mom

    def _kin(self, k):
        return k if self.plain else int(self.lib.pr_rerank_width(self.ctx.h, int(k)))

    def local_select(self, mom_all: torch.Tensor, G: int, mask_width, p_weight, k, db_row0, q_row0):
        """fp32 selection of this shard's k + 8 best with the moments of all shards -> (idx_in i32 [m,kin], score f64 [m,kin])."""
        m, n = self._m, self.n
        d_p, d_i = self._bufs["d_p"], self._bufs.get("d_i")
        kin = self._kin(k)
        idx_in = self._buf("idx_in", (m, kin), torch.int32)
        sc32 = self._buf("sc32", (m, kin), torch.float32)
        sc64 = self._buf("sc64", (m, kin), torch.float64)
        self._mom_all = mom_all.contiguous()
        self._args = (G, q_row0, db_row0, int(mask_width), float(p_weight))
        self._enter()
        self.ctx.check(self.lib.pr_fuse_select_f64_dev(self.ctx.h, _dptr(d_p), None if self.plain else _dptr(d_i), m, n, _dptr(self._mom_all), G,
                                                       q_row0, db_row0, int(mask_width), float(p_weight), int(kin), _dptr(idx_in), _dptr(sc32),
                                                       _dptr(sc64)))
        self._leave()
        return idx_in, sc64

    def _raw_args(self):
        sc = self.type == _lib.TYPE_SC
        assert self._q_sig.dtype == self.db_sig.dtype
        raw = (_dptr(self._q_sig), _dptr(self.db_sig), _torch_dt(self.db_sig))
        none = (None, None, 0)
        return (*(raw if sc else none), *(none if sc else raw), _dptr(self._mom_all) if sc else None, None if sc else _dptr(self._mom_all))

    def local_rerank(self, cand_idx: torch.Tensor, k: int, partial: bool, cand_sc: torch.Tensor = None):
        """fp64 re-evaluation of the candidates [m,kin]: partial=False -> (idx [m,k], score [m,k]) (all candidates are this
        shard's: the one-rank path); partial=True -> scores [m,kin], NaN for candidates outside this shard.  cand_sc: the
        candidates' fp32-pass scores as f64 [m,kin] (ascending) - candidates that cannot reach the top-k are then not evaluated."""
        m, n = self._m, self.n
        G, q_row0, db_row0, mask_width, p_weight = self._args
        kin = cand_idx.shape[1]
        cand_idx = cand_idx.contiguous()
        cand_sc = None if cand_sc is None else cand_sc.contiguous()
        csc = None if cand_sc is None else _dptr(cand_sc)
        self._last_cand = (cand_idx, cand_sc)
        self._enter()
        if partial:
            part = self._buf("part", (m, 5, kin), torch.float64)           # the shard's p5 block (include/place_recognition.h)
            self.ctx.check(self.lib.pr_rerank_partial_dev(self.ctx.h, *self._raw_args(), m, n, G, q_row0, db_row0, mask_width, p_weight, kin,
                                                          _dptr(cand_idx), csc, int(k), _dptr(part)))
            self._leave()
            return part
        idx = self._buf("idx", (m, k), torch.int32)
        score = self._buf("score", (m, k), torch.float64)
        self.ctx.check(self.lib.pr_rerank_dev(self.ctx.h, *self._raw_args(), m, n, G, q_row0, db_row0, mask_width, p_weight, kin,
                                              _dptr(cand_idx), csc, int(k), _dptr(idx), _dptr(score)))
        self._leave()
        return idx, score

    def _moms(self):
        sc = self.type == _lib.TYPE_SC
        return (self._mom_all if sc else None, None if sc else self._mom_all)

    def finish(self, cand_idx: torch.Tensor, cand_sc: torch.Tensor, part_all: torch.Tensor, k: int):
        return _finish_dev(self, cand_idx, cand_sc, part_all, k, (*self._moms(), self._args[0]), self._args[4])

    def exact_moments(self, offset: int = 0, last: bool = True):
        """Step 7, first local part: this shard's exact rows of flagged queries offset .. offset + 63 of the last finish() (kept in the
        context) and their moments -> [m, 4, 3] f64.  last: no pass follows (flagged queries behind it raise WARN_ORDER_UNRESOLVED)."""
        return _exact_moments_dev(self, self._raw_args()[:6], *self._moms(), self._args[0], self._m, self.n, offset, last)

    def exact_select(self, exact_all: torch.Tensor, k: int, offset: int = 0):
        """Step 7, second local part: this shard's k best of the flagged queries' exact rows under the statistics of all shards
        -> [64, 2, k] f64 (scores | global indices)."""
        sc = self.type == _lib.TYPE_SC
        G, q_row0, db_row0, mask_width, p_weight = self._args
        return _exact_select_dev(self, exact_all, self._m, self.n, q_row0, db_row0, mask_width, p_weight, sc, not sc, k, offset)

    def exact_merge(self, sel_all: torch.Tensor, k: int, idx: torch.Tensor, score: torch.Tensor, offset: int = 0):
        return _exact_merge_dev(self, sel_all, self._m, k, idx, score, offset)

    def local_phase2(self, mom_all: torch.Tensor, G: int, mask_width, p_weight, k, db_row0, q_row0):
        """Selection + re-evaluation of this shard alone -> its own top-k (what rank g would answer by itself)."""
        idx_in, sc = self.local_select(mom_all, G, mask_width, p_weight, k, db_row0, q_row0)
        if self.plain:
            return idx_in, sc
        return self.local_rerank(idx_in, k, partial=False, cand_sc=sc)

    def merge(self, idx_all: torch.Tensor, sc_all: torch.Tensor, k: int):
        return _merge_dev(self, idx_all, sc_all, k)

    def match(self, queries: torch.Tensor, mask_width: int = 0, p_weight: float = 2.0, k: int = 1,
              db_row0: int = 0, q_row0: int = 0, group=None, force_exchange: bool = False, f16_fallback: bool = True,
              exact_order: bool = True, mark=None):
        """Returns (idx int32 [m,k] GLOBAL DB row indices, score float64 [m,k]) as device tensors.
        force_exchange: run the all-gathers and the merge even with one rank (measures the protocol's overhead).
        exact_order (default): queries whose re-evaluated order is not certain under the fp32 pass's row sigmas, or whose k + 8 candidates
        do not provably hold the top-k of the whole row, are answered from their exact fp64 rows, sharded or not (run_test.m:38-57 are fp64
        over the whole row), 64 flagged queries per pass: a call of up to 64 queries runs one pass of stream-ordered kernels that leave at
        once when nothing is flagged (no host synchronisation: such a call can be captured in a hipGraph); a larger call reads the
        number of flagged queries back (one synchronisation) and runs the passes it takes.  "async": ceil(m / 64) such passes chained on the
        stream whatever was flagged (no read-back: what a captured graph of a large call uses; empty passes leave at once);
        False skips the resolution (the answer of the re-evaluated candidate list; the flags are simply dropped)."""
        G = _world(group)
        f16 = self.f16 and not self.plain
        resolve = ((self.exact_moments, self.exact_select, self.exact_merge, lambda: self._exact_passes(self._m, exact_order))
                   if (exact_order and not self.plain and not f16) else None)
        post = (lambda cand_sc, idx, score: self._margin(_dptr_mom(self, True), _dptr_mom(self, False), self._args[0], p_weight, cand_sc, k,
                                                          score)) if f16 else None
        idx, score = sharded_topk(lambda: self.local_phase1(queries),
                                  lambda mom_all, G_: self.local_select(mom_all, G_, mask_width, p_weight, k, db_row0, q_row0),
                                  k, group if (G > 1 or force_exchange) else None, G, merge=self.merge, force_exchange=force_exchange,
                                  rerank=None if self.plain else self.local_rerank, finish=self.finish, post=post, resolve=resolve, mark=mark)
        if f16 and f16_fallback:
            def run_rows(rows, qr0):
                fb = self._split_twin()
                qsel = queries.view(-1, self.rows_per_sig, self.sig_len)[rows].reshape(-1, self.sig_len).contiguous()   # M2DP: 4 rows per query
                return fb.match(qsel, mask_width, p_weight, k, db_row0, q_row0 if qr0 is None else qr0, group, force_exchange,
                                exact_order=exact_order)
            idx, score = self._fallback_rows(run_rows, idx, score, mask_width, q_row0)
        elif resolve is not None and G == 1 and not force_exchange:
            self._resolve_order(self._raw_args()[:6], *self._moms(), self._m, self.n, q_row0, mask_width, p_weight, k, idx, score, exact_order)
            if mark is not None:
                mark("exact rows (one shard)")
        return idx, score

    def flagged_count(self) -> int:
        """Queries the last match(..., exact_order=False) of ONE rank left flagged by the order / containment checks (the ones the default
        match() answers from their exact rows).  Synchronises (pr_order_flagged_count); 0 once a resolving call has taken the flags."""
        cnt = C.c_int32(0)
        self._enter()
        self.ctx.check(self.lib.pr_order_flagged_count(self.ctx.h, int(self._m if hasattr(self, "_m") else self.sc._m), C.byref(cnt)))
        self._leave()
        return int(cnt.value)

    def take_warnings(self) -> int:
        """PR_WARN_* bits of the context since the last call (synchronises its stream): WARN_ORDER_RESOLVED after a match() whose order
        needed fp64 row statistics (WARN_ORDER_UNRESOLVED: only when exact_moments(.., last=True) was called with flagged queries left)."""
        return self.ctx.take_warnings()

    def _split_twin(self):
        """The same matcher in split-f16 over the same (already resident) raw DB, created and packed on first use."""
        if getattr(self, "_twin", None) is None or self._twin_of is not self.db_sig:
            if getattr(self, "_twin", None) is not None:
                self._twin.close()
            tw = Matcher({_lib.TYPE_SC: "sc", _lib.TYPE_M2DP: "m2dp"}[self.type], self._max_q, self._max_db,
                         ctx=Context(self.ctx.device, sc_arith="f16x2", stream=self.ctx.stream))
            tw.pack_database(self.db_sig)
            self._twin, self._twin_of = tw, self.db_sig
        return self._twin

    def distances(self):
        """The last distance matrices (device, float32 [m, n_local])."""
        return self._bufs["d_p"], self._bufs.get("d_i")

    @classmethod
    def on_new_stream(cls, type_: str, max_queries: int, max_db: int, device: int | None = None):
        """A matcher whose library context lives on a stream of its own (`.stream`): what hipGraph capture needs (the null
        stream cannot be captured).  Use it under `with torch.cuda.stream(mt.stream):`."""
        device = torch.cuda.current_device() if device is None else device
        st = torch.cuda.Stream(device)
        with torch.cuda.stream(st):
            mt = cls(type_, max_queries, max_db, ctx=_stream_context(device))
        mt.stream = st
        return mt

    def capture(self, queries: torch.Tensor, mask_width: int = 0, p_weight: float = 2.0, k: int = 1):
        """Captures one single-rank match() of the STATIC tensor `queries` against the resident, packed DB into a hipGraph
        (online use: one keyframe per call - the ~10 kernel launches of a call replay as one graph launch).  Returns a
        CapturedMatch: `.run(new_queries)` copies them into the static input and replays ON THE MATCHER'S STREAM (a replay
        on another stream would not be ordered with the copy), `.idx` / `.score` are the static outputs.  The matcher must
        come from on_new_stream().  A graph cannot read a count back: calls above 64 queries are captured with exact_order="async" (ceil(m / 64)
        chained passes: every flagged query of every replay is resolved)."""
        st = self.stream
        eo = True if queries.shape[0] // self.rows_per_sig <= RESOLVE_SLOTS else "async"
        with torch.cuda.stream(st):
            assert self.ctx.stream == int(st.cuda_stream)
            self.match(queries, mask_width, p_weight, k, exact_order=eo)            # warm-up: allocates every buffer the call uses
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                idx, score = self.match(queries, mask_width, p_weight, k, exact_order=eo)
        return CapturedMatch(g, st, queries, idx, score)


class CapturedMatch:
    """A match() call as a hipGraph (Matcher.capture)."""

    def __init__(self, graph, stream, queries, idx, score):
        self.graph, self.stream, self.queries, self.idx, self.score = graph, stream, queries, idx, score

    def run(self, new_queries: torch.Tensor | None = None, sync: bool = True):
        if new_queries is not None:
            self.stream.wait_stream(torch.cuda.current_stream(new_queries.device))   # whoever produced new_queries did it there
            new_queries.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            if new_queries is not None:
                self.queries.copy_(new_queries, non_blocking=True)
            self.graph.replay()
        if sync:
            self.stream.synchronize()
        return self.idx, self.score


class FusedMatcher(_Base):
    """BASELINE.json config 5 ("fused SC + M2DP scoring", build-defined: DESIGN.md §7): an SC and an M2DP matcher over the
    same places; the four row z-scores are added (weights p, 1, p, 1) in ONE top-k pass (pr_fuse_select2_dev).  Shards like
    Matcher: the moments of both descriptor types travel in the same all_gather ([m, 4, 3] f64 per rank)."""

    def __init__(self, max_queries: int, max_db: int, ctx: Context | None = None, device: int | None = None):
        self._init_ctx(ctx, device)
        self.sc = Matcher("sc", max_queries, max_db, self.ctx)
        self.m2 = Matcher("m2dp", max_queries, max_db, self.ctx)
        self._bufs = {}
        self._flat = {}

    _buf = Matcher._buf

    def close(self):
        if getattr(self, "_twin", None) is not None:
            self._twin.close()
            self._twin = None
        self.sc.close(); self.m2.close()

    def pack_database(self, sc_sig: torch.Tensor, m2dp_sig: torch.Tensor):
        self.sc.pack_database(sc_sig); self.m2.pack_database(m2dp_sig)
        assert self.sc.n == self.m2.n, "the two databases must describe the same places"

    def local_phase1(self, sc_queries, m2dp_queries):
        a = self.sc.local_phase1(sc_queries)
        b = self.m2.local_phase1(m2dp_queries)
        assert self.sc._m == self.m2._m
        return torch.cat([a, b], dim=1)                                    # [m, 4, 3]

    def local_select(self, mom_all, G, mask_width, p_weight, k, db_row0, q_row0):
        m, n = self.sc._m, self.sc.n
        lib, h = self.lib, self.ctx.h
        mom_all = mom_all.reshape(G, m, 4, 3)
        self._m1, self._m2 = mom_all[:, :, :2].contiguous(), mom_all[:, :, 2:].contiguous()
        self._args = (G, q_row0, db_row0, int(mask_width), float(p_weight))
        d = [self.sc._bufs["d_p"], self.sc._bufs["d_i"], self.m2._bufs["d_p"], self.m2._bufs["d_i"]]
        kin = int(self.lib.pr_rerank_width(self.ctx.h, int(k)))
        idx_in = self._buf("idx_in", (m, kin), torch.int32)
        sc32 = self._buf("sc32", (m, kin), torch.float32)
        sc64 = self._buf("sc64", (m, kin), torch.float64)
        self._enter()
        self.ctx.check(lib.pr_fuse_select2_f64_dev(h, _dptr(d[0]), _dptr(d[1]), _dptr(d[2]), _dptr(d[3]), m, n, _dptr(self._m1), _dptr(self._m2), G,
                                                   q_row0, db_row0, int(mask_width), float(p_weight), int(kin), _dptr(idx_in), _dptr(sc32),
                                                   _dptr(sc64)))
        self._leave()
        return idx_in, sc64

    def local_rerank(self, cand_idx, k, partial, cand_sc=None):
        m, n = self.sc._m, self.sc.n
        G, q_row0, db_row0, mask_width, p_weight = self._args
        kin = cand_idx.shape[1]
        cand_idx = cand_idx.contiguous()
        raw = (_dptr(self.sc._q_sig), _dptr(self.sc.db_sig), _torch_dt(self.sc.db_sig), _dptr(self.m2._q_sig), _dptr(self.m2.db_sig),
               _torch_dt(self.m2.db_sig), _dptr(self._m1), _dptr(self._m2))
        cand_sc = None if cand_sc is None else cand_sc.contiguous()
        csc = None if cand_sc is None else _dptr(cand_sc)
        self._last_cand = (cand_idx, cand_sc)
        self._enter()
        if partial:
            part = self._buf("part", (m, 5, kin), torch.float64)
            self.ctx.check(self.lib.pr_rerank_partial_dev(self.ctx.h, *raw, m, n, G, q_row0, db_row0, mask_width, p_weight, kin, _dptr(cand_idx),
                                                          csc, int(k), _dptr(part)))
            self._leave()
            return part
        idx = self._buf("idx", (m, k), torch.int32)
        score = self._buf("score", (m, k), torch.float64)
        self.ctx.check(self.lib.pr_rerank_dev(self.ctx.h, *raw, m, n, G, q_row0, db_row0, mask_width, p_weight, kin, _dptr(cand_idx), csc, int(k),
                                              _dptr(idx), _dptr(score)))
        self._leave()
        return idx, score

    def finish(self, cand_idx, cand_sc, part_all, k):
        return _finish_dev(self, cand_idx, cand_sc, part_all, k, (self._m1, self._m2, self._args[0]), self._args[4])

    def _raw6(self):
        return (_dptr(self.sc._q_sig), _dptr(self.sc.db_sig), _torch_dt(self.sc.db_sig), _dptr(self.m2._q_sig), _dptr(self.m2.db_sig),
                _torch_dt(self.m2.db_sig))

    def exact_moments(self, offset=0, last=True):
        return _exact_moments_dev(self, self._raw6(), self._m1, self._m2, self._args[0], self.sc._m, self.sc.n, offset, last)

    def exact_select(self, exact_all, k, offset=0):
        G, q_row0, db_row0, mask_width, p_weight = self._args
        return _exact_select_dev(self, exact_all, self.sc._m, self.sc.n, q_row0, db_row0, mask_width, p_weight, True, True, k, offset)

    def exact_merge(self, sel_all, k, idx, score, offset=0):
        return _exact_merge_dev(self, sel_all, self.sc._m, k, idx, score, offset)

    def local_phase2(self, mom_all, G, mask_width, p_weight, k, db_row0, q_row0):
        idx_in, sc = self.local_select(mom_all, G, mask_width, p_weight, k, db_row0, q_row0)
        return self.local_rerank(idx_in, k, partial=False, cand_sc=sc)

    def merge(self, idx_all, sc_all, k):
        return _merge_dev(self, idx_all, sc_all, k)

    def match(self, sc_queries: torch.Tensor, m2dp_queries: torch.Tensor, mask_width: int = 0, p_weight: float = 2.0, k: int = 1,
              db_row0: int = 0, q_row0: int = 0, group=None, f16_fallback: bool = True, exact_order: bool = True):
        G = _world(group)
        post = (lambda cand_sc, idx, score: self._margin(self._m1, self._m2, self._args[0], p_weight, cand_sc, k, score)) if self.f16 else None
        resolve = ((self.exact_moments, self.exact_select, self.exact_merge, lambda: self._exact_passes(self.sc._m, exact_order))
                   if (exact_order and not self.f16) else None)
        idx, score = sharded_topk(lambda: self.local_phase1(sc_queries, m2dp_queries),
                                  lambda mom_all, G_: self.local_select(mom_all, G_, mask_width, p_weight, k, db_row0, q_row0),
                                  k, group if G > 1 else None, G, merge=self.merge, rerank=self.local_rerank, finish=self.finish, post=post,
                                  resolve=resolve)
        if self.f16 and f16_fallback:
            def run_rows(rows, qr0):
                fb = self._split_twin()
                return fb.match(sc_queries[rows].contiguous(), m2dp_queries.view(-1, 4, 384)[rows].reshape(-1, 384).contiguous(), mask_width,
                                p_weight, k, db_row0, q_row0 if qr0 is None else qr0, group, exact_order=exact_order)
            idx, score = self._fallback_rows(run_rows, idx, score, mask_width, q_row0)
        elif resolve is not None and G == 1:
            self._resolve_order(self._raw6(), self._m1, self._m2, self.sc._m, self.sc.n, q_row0, mask_width, p_weight, k, idx, score, exact_order)
        return idx, score

    take_warnings = Matcher.take_warnings
    flagged_count = Matcher.flagged_count

    def _split_twin(self):
        if getattr(self, "_twin", None) is None or self._twin_of is not self.sc.db_sig:
            if getattr(self, "_twin", None) is not None:
                self._twin.close()
            tw = FusedMatcher(self.sc._max_q, self.sc._max_db, ctx=Context(self.ctx.device, sc_arith="f16x2", stream=self.ctx.stream))
            tw.pack_database(self.sc.db_sig, self.m2.db_sig)
            self._twin, self._twin_of = tw, self.sc.db_sig
        return self._twin


def _world(group) -> int:
    import torch.distributed as dist
    return dist.get_world_size(group) if (group is not None or (dist.is_available() and dist.is_initialized())) else 1


def _merge_dev(owner, idx_all: torch.Tensor, sc_all: torch.Tensor, k: int):
    """pr_merge_topk_dev on [G, m, k] device tensors."""
    G, m, kk = idx_all.shape
    assert kk == k
    idx = torch.empty((m, k), dtype=torch.int32, device=idx_all.device)
    score = torch.empty((m, k), dtype=torch.float64, device=idx_all.device)
    owner._enter()
    owner.ctx.check(owner.lib.pr_merge_topk_dev(owner.ctx.h, _dptr(idx_all.contiguous()), _dptr(sc_all.contiguous()), G, m, k,
                                                _dptr(idx), _dptr(score)))
    owner._leave()
    return idx, score


def _finish_dev(owner, cand_idx: torch.Tensor, cand_sc: torch.Tensor | None, part_all: torch.Tensor, k: int, moms, p_weight: float):
    """pr_rerank_finish_dev: candidates [m, kin] (+ their merged pass scores) + the shards' p5 blocks [G, m, 5, kin] -> (idx [m,k], score [m,k]);
    the order and containment checks of the result (statistics moms = (mom_sc | None, mom_m2 | None, shards in them)) stay in the context:
    pr_f16_margin_dev (PR_SC_ARITH_F16) or exact_moments() / exact_select() / exact_merge() take them."""
    G, m, five, kin = part_all.shape
    assert five == 5
    idx = torch.empty((m, k), dtype=torch.int32, device=cand_idx.device)
    score = torch.empty((m, k), dtype=torch.float64, device=cand_idx.device)
    cand_idx = cand_idx.contiguous()
    cand_sc = None if cand_sc is None else cand_sc.contiguous()
    part_all = part_all.contiguous()
    mom_sc, mom_m2, g_mom = moms
    owner._enter()
    owner.ctx.check(owner.lib.pr_rerank_finish_dev(owner.ctx.h, _dptr(mom_sc), _dptr(mom_m2), int(g_mom), _dptr(cand_idx), _dptr(cand_sc),
                                                   _dptr(part_all), G, m, kin, k, float(p_weight), _dptr(idx), _dptr(score)))
    owner._leave()
    return idx, score


RESOLVE_SLOTS = 64      # flagged queries one pass of the exact-row resolution serves (kernels.hpp)


def _exact_moments_dev(owner, raw6, mom_sc, mom_m2, g_mom: int, m: int, n_local: int, offset: int = 0, last: bool = True):
    exact = torch.empty((m, 4, 3), dtype=torch.float64, device=owner.dev)
    owner._enter()
    owner.ctx.check(owner.lib.pr_order_exact_moments_dev(owner.ctx.h, *raw6, _dptr(mom_sc), _dptr(mom_m2), int(g_mom), m, n_local, int(offset),
                                                         int(bool(last)), _dptr(exact)))
    owner._leave()
    return exact


def _exact_select_dev(owner, exact_all: torch.Tensor, m: int, n_local: int, q_row0: int, db_row0: int, mask_width: int, p_weight: float,
                      has_sc: bool, has_m2: bool, k: int, offset: int = 0):
    G = exact_all.shape[0]
    exact_all = exact_all.contiguous()
    sel = torch.empty((RESOLVE_SLOTS, 2, k), dtype=torch.float64, device=owner.dev)
    owner._enter()
    owner.ctx.check(owner.lib.pr_order_exact_select_dev(owner.ctx.h, _dptr(exact_all), G, m, n_local, int(q_row0), int(db_row0), int(mask_width),
                                                        float(p_weight), int(has_sc), int(has_m2), int(k), int(offset), _dptr(sel)))
    owner._leave()
    return sel


def _exact_merge_dev(owner, sel_all: torch.Tensor, m: int, k: int, idx, score, offset: int = 0):
    G = sel_all.shape[0]
    sel_all = sel_all.contiguous()
    owner._enter()
    owner.ctx.check(owner.lib.pr_order_exact_merge_dev(owner.ctx.h, _dptr(sel_all), G, m, int(k), int(offset), _dptr(idx), _dptr(score)))
    owner._leave()
    return idx, score


def sharded_topk(local_moments, local_select, k: int, group, G: int, merge=None, force_exchange: bool = False, rerank=None, finish=None,
                 post=None, resolve=None, mark=None):
    """The exchange protocol of SURVEY.md §8-e around two local callables (HIP in production; a numpy stand-in in
    the gloo CPU tests): moments -> all_gather -> select with the moments of all shards -> all_gather -> merge.
    mark (optional): called with a phase name after every phase of the protocol has been ENQUEUED (bench.py records an event on the
    stream there: the per-rank phase times of a step)."""
    import torch.distributed as dist
    mark = mark or (lambda name: None)
    mom = local_moments()
    mark("pack+distances+moments")
    if G == 1 and not force_exchange:
        idx_in, sc = local_select(mom.unsqueeze(0) if mom.dim() == 3 else mom, 1)
        mark("select")
        if rerank is None:
            return idx_in, sc
        idx, score = rerank(idx_in, k, False, sc)
        mark("rerank")
        if post is not None:                             # PR_SC_ARITH_F16: margin flags of the candidate list (no synchronisation)
            post(sc, idx, score)
        return idx, score
    stage_on_host = dist.get_backend(group) == "gloo"   # gloo has no device all_gather: used by the single-GPU tests

    def gather(t):   # output = the ranks' tensors concatenated along dim 0, viewed as [G, ...]
        src = t.contiguous()
        if stage_on_host and src.is_cuda:
            src = src.cpu()
        out = torch.empty((G * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=group)   # nccl: RCCL, on the stream the library's kernels run on
        return out.view((G,) + tuple(src.shape)).to(t.device)

    mom_all = gather(mom)
    mark("all_gather A (moments)")
    idx_in, sc = local_select(mom_all, G)
    mark("select")
    kin = idx_in.shape[1]
    idx_all, sc_all = gather(idx_in), gather(sc)
    mark("all_gather B (candidates)")
    do_merge = merge if (merge is not None and idx_all.is_cuda) else merge_topk
    if rerank is None:                                   # nothing to re-evaluate (numpy stand-ins of the gloo tests, DELIGHT)
        return do_merge(idx_all, sc_all, k)
    cand_idx, cand_sc = do_merge(idx_all, sc_all, kin)   # the global top-(k+8) of the fp32 pass, identical on every rank
    part = rerank(cand_idx, k, True, cand_sc)
    mark("merge+rerank")
    part_all = gather(part)
    mark("all_gather C (evaluations)")
    idx, score = finish(cand_idx, cand_sc, part_all, k)
    mark("finish+checks")
    if resolve is not None:                              # step 7: the flagged queries from their exact rows (moments, per-shard k best, merge),
        passes = resolve[3]() if len(resolve) > 3 else 1   # 64 per pass; the same number of passes on every rank
        if len(resolve) > 3:
            mark("flagged count")                        # (calls above 64 queries: one read-back of the count; none flagged - no pass)
        for p in range(passes):
            off = p * RESOLVE_SLOTS
            ex = resolve[0](off, p == passes - 1) if len(resolve) > 3 else (resolve[0](off) if p else resolve[0]())
            mark("exact rows")
            exact_all = gather(ex)
            mark("all_gather D (exact moments)")
            sel = resolve[1](exact_all, k, off) if p else resolve[1](exact_all, k)
            mark("exact select")
            sel_all = gather(sel)
            mark("all_gather E (exact lists)")
            idx, score = resolve[2](sel_all, k, idx, score, off) if p else resolve[2](sel_all, k, idx, score)
            mark("exact merge")
    if post is not None:
        post(cand_sc, idx, score)
    return idx, score


def merge_topk(idx_all: torch.Tensor, sc_all: torch.Tensor, k: int):
    """k-way merge of per-shard top-k lists [G, m, k] by (score, index) ascending; -1 / NaN entries sort last.
    (torch restatement of pr_merge_topk_dev for host tensors: the gloo tests)"""
    G, m, kk = idx_all.shape
    idx = idx_all.permute(1, 0, 2).reshape(m, G * kk).to(torch.int64)
    sc = sc_all.permute(1, 0, 2).reshape(m, G * kk).to(torch.float64)
    bad = (idx < 0) | torch.isnan(sc)
    sc = torch.where(bad, torch.full_like(sc, float("inf")), sc)
    idx_key = torch.where(bad, torch.full_like(idx, 2 ** 62), idx)
    o1 = torch.argsort(idx_key, dim=1, stable=True)
    sc1 = torch.gather(sc, 1, o1)
    bad1 = torch.gather(bad, 1, o1)
    # bad entries after every good one, also after good +Inf scores
    key2 = torch.where(bad1, torch.full_like(sc1, float("inf")), sc1)
    o2 = torch.argsort(key2 + 0.0, dim=1, stable=True)
    o2b = torch.argsort(torch.gather(bad1, 1, o2).to(torch.int8), dim=1, stable=True)
    order = torch.gather(o1, 1, torch.gather(o2, 1, o2b))[:, :k]
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
