"""Host-side mirror of the reference's interface for the hot path, on top of the C ABI (libpr_amd.so).

Same names, argument meaning and error behaviour as the reference:
  SC / M2DP classes            SC/SC.h:10-23, M2DP/M2DP.h:12-30  (getSignatureSize / getSignature)
  processSC / processM2DP      match_signatures/processSC.m:1, processM2DP.m:1
  run_test                     match_signatures/run_test.m:1  (fusion, mask, top-1; PR/AUC evaluation in eval.py)
Arrays are numpy on the host; the device-resident path used by bench.py / dist.py is `Matcher`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import PRError, TYPE_BOW, TYPE_DELIGHT, TYPE_GIST, TYPE_M2DP, TYPE_SC  # noqa: F401


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One HIP device + stream (pr_ctx)."""

    def __init__(self, device: int = 0, sc_arith: str | None = None, stream: int | None = None, nan_policy: str | None = None,
                 exact_statistics: bool | None = None, sc_binary: bool | None = None):
        """sc_arith: None (library default: "f16x2", or PR_SC_MATCH=f32 from the environment), "f16x2" or "f32".
        exact_statistics: True = every query of a top-k call gets fp64 row statistics (pr_set_exact_statistics: scores are the reference's
        doubles to rounding, at 2.3 ms per query and 100k entries).
        stream: a hipStream_t the caller owns (e.g. torch.cuda.current_stream().cuda_stream; 0 = the null stream): the
        context's kernels are enqueued there (pr_create_on_stream).  nan_policy: "exclude" (default, MATLAB's behaviour for
        zero-norm SC rows) or "fail" (PR_ENAN)."""
        self.lib = _lib.load()
        h = C.c_void_p()
        if stream is None:
            rc = self.lib.pr_create(device, C.byref(h))
        else:
            rc = self.lib.pr_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if rc != 0:
            raise PRError(rc, self.lib.pr_last_error(None).decode())
        self.h = h
        self.device = device
        if sc_arith is not None:
            self.check(self.lib.pr_set_sc_arith(h, {"f16x2": _lib.SC_ARITH_F16X2, "f32": _lib.SC_ARITH_F32, "f16": _lib.SC_ARITH_F16}[sc_arith]))
        if nan_policy is not None:
            self.check(self.lib.pr_set_nan_policy(h, {"exclude": _lib.NAN_EXCLUDE, "fail": _lib.NAN_FAIL}[nan_policy]))
        self.exact_statistics = bool(exact_statistics) if exact_statistics is not None else os.environ.get("PR_FORCE_ORDER_FLAGS") == "1"
        if exact_statistics is not None:
            self.check(self.lib.pr_set_exact_statistics(h, int(bool(exact_statistics))))
        if sc_binary is not None:      # False: a binary intensity channel goes through the split-f16 kernel like any other (pr_set_sc_binary)
            self.check(self.lib.pr_set_sc_binary(h, int(bool(sc_binary))))

    def kernel_timing(self, on: bool):
        """HIP events around the matcher launches of pr_distances_dev (pr_set_kernel_timing)."""
        self.check(self.lib.pr_set_kernel_timing(self.h, int(bool(on))))

    def last_distance_timing(self):
        """(ms channel-0 or only launch, ms channel-1 single-product launch, ms channel-1 split-f16 launch) of the last timed call."""
        ms = (C.c_float * 3)()
        self.check(self.lib.pr_last_distance_timing(self.h, ms))
        return float(ms[0]), float(ms[1]), float(ms[2])

    def take_warnings(self) -> int:
        """PR_WARN_* bits raised since the last call (1: zero-norm SC rows excluded, 2: an M2DP singular pair did not converge)."""
        return int(self.lib.pr_take_warnings(self.h))

    @property
    def sc_arith(self) -> str:
        return {_lib.SC_ARITH_F32: "f32", _lib.SC_ARITH_F16: "f16"}.get(self.lib.pr_get_sc_arith(self.h), "f16x2")

    def close(self):
        if getattr(self, "h", None):
            self.lib.pr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        _lib.check(self.h, rc)

    def sync(self):
        self.check(self.lib.pr_sync(self.h))

    @property
    def stream(self) -> int:
        return int(self.lib.pr_stream(self.h) or 0)


class Group:
    """pr_group: hist2 row-sharded over the GPUs `devices` of this node inside the C ABI (RCCL all-gathers between the kernels;
    repeated device ids = several shards on one GPU, exchanged by device copies)."""

    def __init__(self, devices):
        self.lib = _lib.load()
        d = np.ascontiguousarray(devices, np.int32)
        h = C.c_void_p()
        rc = self.lib.pr_group_create(_ptr(d), len(d), C.byref(h))
        if rc != 0:
            raise PRError(rc, self.lib.pr_group_last_error(None).decode())
        self.h = h
        self.div, self.width = None, None

    @property
    def uses_rccl(self) -> bool:
        return bool(self.lib.pr_group_uses_rccl(self.h))

    @property
    def rccl_ranks(self) -> int:
        """Ranks the group's RCCL communicator reports (ncclCommCount); 0 when the group exchanges by device copies."""
        return int(self.lib.pr_group_rccl_ranks(self.h))

    PHASES = ("upload+pack(q)", "distances", "moments", "all_gather A (moments)", "select", "all_gather B (candidates)", "merge+rerank",
              "all_gather C (evaluations)", "finish+checks", "exact rows (flagged queries)")

    def set_timing(self, on: bool):
        """HIP events between the phases of every following match_topk, on every shard's stream (pr_group_set_timing)."""
        self._check(self.lib.pr_group_set_timing(self.h, int(bool(on))))

    def last_timing(self):
        """[{phase: ms}] per shard for the last timed match_topk (pr_group_last_timing)."""
        G = int(self.lib.pr_group_size(self.h))
        ms = (C.c_float * (G * len(self.PHASES)))()
        self._check(self.lib.pr_group_last_timing(self.h, ms, G * len(self.PHASES)))
        return [{name: float(ms[r * len(self.PHASES) + p]) for p, name in enumerate(self.PHASES)} for r in range(G)]

    @property
    def last_flagged(self) -> int:
        """Queries of the last match_topk that were answered from their exact fp64 rows (order / containment checks)."""
        return int(self.lib.pr_group_last_flagged(self.h))

    def _check(self, rc):
        if rc != 0:
            raise PRError(rc, self.lib.pr_group_last_error(self.h).decode())

    def set_database(self, type_: str, hist2, extra_capacity: int = 0):
        """extra_capacity > 0: room for that many more signatures, added later in place by append_database (the online loop of
        SC/test_sc.cpp:40-56 + run_test.m:57 through host buffers)."""
        t = {"sc": TYPE_SC, "m2dp": TYPE_M2DP}[type_]
        self.div, self.width = {TYPE_SC: (1, 2400), TYPE_M2DP: (4, 384)}[t]
        h2 = np.ascontiguousarray(hist2, np.float64)
        if h2.ndim != 2 or h2.shape[1] != self.width or h2.shape[0] % self.div:
            raise ValueError(f"expected a [{self.div}*n, {self.width}] signature matrix")
        if extra_capacity:
            self._check(self.lib.pr_group_set_database_growable(self.h, t, _ptr(h2), h2.shape[0] // self.div, int(extra_capacity)))
        else:
            self._check(self.lib.pr_group_set_database(self.h, t, _ptr(h2), h2.shape[0] // self.div))

    def append_database(self, hist_new):
        h = np.ascontiguousarray(hist_new, np.float64)
        if h.ndim != 2 or h.shape[1] != self.width or h.shape[0] % self.div:
            raise ValueError(f"expected a [{self.div}*n_new, {self.width}] signature matrix")
        self._check(self.lib.pr_group_append_database(self.h, _ptr(h), h.shape[0] // self.div))

    @property
    def database_rows(self) -> int:
        return int(self.lib.pr_group_database_rows(self.h))

    def match_topk(self, hist1, mask_width=0, p_weight=2.0, k=1):
        """run_test.m:26-57 against the sharded database -> (idx int32 [m,k] global rows, score float64 [m,k])."""
        h1 = np.ascontiguousarray(hist1, np.float64)
        if h1.ndim != 2 or h1.shape[1] != self.width or h1.shape[0] % self.div:
            raise ValueError(f"expected a [{self.div}*m, {self.width}] signature matrix")
        m = h1.shape[0] // self.div
        idx = np.empty((m, k), np.int32); sc = np.empty((m, k), np.float64)
        self._check(self.lib.pr_group_match_topk(self.h, _ptr(h1), m, int(mask_width), float(p_weight), int(k), _ptr(idx), _ptr(sc)))
        return idx, sc

    def take_warnings(self) -> int:
        """PR_WARN_* bits raised on any shard since the last call."""
        return int(self.lib.pr_group_take_warnings(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.pr_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class _exact_statistics:
    """`with _exact_statistics(ctx):` - pr_set_exact_statistics(ctx, 1) around a call, the context's own setting back afterwards."""

    def __init__(self, ctx: Context):
        self.ctx, self.prev = ctx, ctx.exact_statistics

    def __enter__(self):
        self.ctx.check(self.ctx.lib.pr_set_exact_statistics(self.ctx.h, 1))
        return self.ctx

    def __exit__(self, *exc):
        self.ctx.check(self.ctx.lib.pr_set_exact_statistics(self.ctx.h, int(self.prev)))
        return False


def _csr(pts_list):
    """list of (xyz [P,3], intensity [P]) -> CSR arrays."""
    offs = np.zeros(len(pts_list) + 1, np.int64)
    for i, (p, _) in enumerate(pts_list):
        offs[i + 1] = offs[i] + len(p)
    xyz = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float64).reshape(-1, 3) for p, _ in pts_list])
                               if pts_list else np.zeros((0, 3)))
    it = np.ascontiguousarray(np.concatenate([np.asarray(i, np.float32).reshape(-1) for _, i in pts_list])
                              if pts_list else np.zeros((0,), np.float32))
    return xyz, it, offs


def sc_generate(xyz, inten, offs, max_rho=45.0, ctx: Context | None = None) -> np.ndarray:
    """test_sc.cpp:40-56 over clouds in CSR layout -> [N, 2400]."""
    ctx = ctx or default_context()
    xyz = np.ascontiguousarray(xyz, np.float64)
    inten = np.ascontiguousarray(inten, np.float32)
    offs = np.ascontiguousarray(offs, np.int64)
    N = len(offs) - 1
    out = np.empty((N, 2400))
    ctx.check(ctx.lib.pr_sc_generate(ctx.h, _ptr(xyz), _ptr(inten), _ptr(offs), N, float(max_rho), _ptr(out)))
    return out


def m2dp_generate(xyz, inten, offs, max_rho=45.0, ctx: Context | None = None) -> np.ndarray:
    """test_m2dp.cpp:41-68 over clouds in CSR layout -> [4N, 384]."""
    ctx = ctx or default_context()
    xyz = np.ascontiguousarray(xyz, np.float64)
    inten = np.ascontiguousarray(inten, np.float32)
    offs = np.ascontiguousarray(offs, np.int64)
    N = len(offs) - 1
    out = np.empty((4 * N, 384))
    ctx.check(ctx.lib.pr_m2dp_generate(ctx.h, _ptr(xyz), _ptr(inten), _ptr(offs), N, float(max_rho), _ptr(out)))
    return out


def m2dp_svd_rows(ctx: Context | None = None) -> np.ndarray:
    """Rows (cloud * 4 + variant) of the context's last m2dp_generate call whose leading singular pair is not unique (pr_m2dp_svd_rows):
    the rows PR_WARN_M2DP_SVD is about."""
    ctx = ctx or default_context()
    rows = np.empty(1024, np.int32)
    cnt = np.zeros(1, np.int32)
    ctx.check(ctx.lib.pr_m2dp_svd_rows(ctx.h, _ptr(rows), len(rows), _ptr(cnt)))
    return rows[: int(cnt[0])].copy()


def delight_generate(xyz, inten, offs, ctx: Context | None = None) -> np.ndarray:
    """test_delight.cpp:41-56 over clouds in CSR layout -> [16N, 256]."""
    ctx = ctx or default_context()
    xyz = np.ascontiguousarray(xyz, np.float64)
    inten = np.ascontiguousarray(inten, np.float32)
    offs = np.ascontiguousarray(offs, np.int64)
    N = len(offs) - 1
    out = np.empty((16 * N, 256))
    ctx.check(ctx.lib.pr_delight_generate(ctx.h, _ptr(xyz), _ptr(inten), _ptr(offs), N, _ptr(out)))
    return out


class DELIGHT:
    """DELIGHT/DELIGHT.h:11-18."""

    def __init__(self, ctx: Context | None = None):
        self.ctx = ctx

    def getSignatureSize(self) -> int:
        return 256

    def getSignature(self, pts, intensity):
        xyz, it, offs = _csr([(pts, intensity)])
        return delight_generate(xyz, it, offs, self.ctx)


class SC:
    """SC/SC.h:10-23."""

    def __init__(self, max_rho: float, ctx: Context | None = None):
        self.max_rho = float(max_rho)
        self.ctx = ctx

    def getSignatureSize(self) -> int:
        return 1200

    def getSignature(self, pts, intensity):
        """pts [P,3] camera frame (PCA alignment happens inside, SC.cpp:17) -> (structure[1200], intensity[1200])."""
        xyz, it, offs = _csr([(pts, intensity)])
        row = sc_generate(xyz, it, offs, self.max_rho, self.ctx)[0]
        return row[:1200].copy(), row[1200:].copy()


class M2DP:
    """M2DP/M2DP.h:12-30.  The reference driver aligns once and loops the 4 sign variants (test_m2dp.cpp:44-68);
    getSignatures returns all 4 rows of that loop."""

    def __init__(self, max_rho: float, ctx: Context | None = None):
        self.max_rho = float(max_rho)
        self.ctx = ctx

    def getSignatureSize(self) -> int:
        return 192

    def getSignatures(self, pts, intensity):
        xyz, it, offs = _csr([(pts, intensity)])
        return m2dp_generate(xyz, it, offs, self.max_rho, self.ctx)


def _distance(fn_name, div, width, hist1, hist2, ctx):
    ctx = ctx or default_context()
    h1 = np.ascontiguousarray(hist1, np.float64)
    h2 = np.ascontiguousarray(hist2, np.float64)
    if h1.ndim != 2 or h2.ndim != 2 or h1.shape[1] != width or h2.shape[1] != width or h1.shape[0] % div or h2.shape[0] % div:
        raise ValueError(f"expected [{div}*m, {width}] and [{div}*n, {width}] signature matrices")
    m, n = h1.shape[0] // div, h2.shape[0] // div
    dp = np.empty((m, n), np.float32)
    di = np.empty((m, n), np.float32)
    ctx.check(getattr(ctx.lib, fn_name)(ctx.h, _ptr(h1), m, _ptr(h2), n, _ptr(dp), _ptr(di)))
    return dp, di


def processSC(hist1, hist2, ctx: Context | None = None):
    """[diff_m_p, diff_m_i] = processSC(hist1, hist2)  (processSC.m:1); float32 [m, n] each."""
    return _distance("pr_sc_distance", 1, 2400, hist1, hist2, ctx)


def processM2DP(hist1, hist2, ctx: Context | None = None):
    """[diff_m_p, diff_m_i] = processM2DP(hist1, hist2)  (processM2DP.m:1); hist rows come in groups of 4 variants."""
    return _distance("pr_m2dp_distance", 4, 384, hist1, hist2, ctx)


def processDELIGHT(hist1, hist2, ctx: Context | None = None):
    """dist = processDELIGHT(hist1, hist2)  (processDELIGHT.m:1); 16 rows per signature; float32 [m, n]."""
    ctx = ctx or default_context()
    h1 = np.ascontiguousarray(hist1, np.float64)
    h2 = np.ascontiguousarray(hist2, np.float64)
    if h1.ndim != 2 or h2.ndim != 2 or h1.shape[1] != 256 or h2.shape[1] != 256 or h1.shape[0] % 16 or h2.shape[0] % 16:
        raise ValueError("expected [16*m, 256] and [16*n, 256] histogram matrices")
    m, n = h1.shape[0] // 16, h2.shape[0] // 16
    d = np.empty((m, n), np.float32)
    ctx.check(ctx.lib.pr_delight_distance(ctx.h, _ptr(h1), m, _ptr(h2), n, _ptr(d)))
    return d


def processGIST(hist1, hist2, ctx: Context | None = None):
    """diff_m = processGIST(hist1, hist2)  (processGIST.m:1): squared Euclidean distances, float32 [m, n]."""
    ctx = ctx or default_context()
    h1 = np.ascontiguousarray(hist1, np.float64); h2 = np.ascontiguousarray(hist2, np.float64)
    if h1.shape[1] != h2.shape[1]:
        raise ValueError("hist1 and hist2 must have the same number of columns")
    d = np.empty((h1.shape[0], h2.shape[0]), np.float32)
    ctx.check(ctx.lib.pr_gist_distance(ctx.h, _ptr(h1), h1.shape[0], _ptr(h2), h2.shape[0], h1.shape[1], _ptr(d)))
    return d


def processBoW(hist1, hist2, ctx: Context | None = None):
    """diff_m = processBoW(hist1, hist2)  (processBoW.m:1): rows alternate word ids / weights (padded with -1); float32 [m, n]."""
    ctx = ctx or default_context()
    h1 = np.ascontiguousarray(hist1, np.float64); h2 = np.ascontiguousarray(hist2, np.float64)
    if h1.shape[1] != h2.shape[1] or h1.shape[0] % 2 or h2.shape[0] % 2:
        raise ValueError("BoW files hold two rows of the same width per image")
    m, n = h1.shape[0] // 2, h2.shape[0] // 2
    d = np.empty((m, n), np.float32)
    ctx.check(ctx.lib.pr_bow_distance(ctx.h, _ptr(h1), m, _ptr(h2), n, h1.shape[1], _ptr(d)))
    return d


def match_topk(type_, hist1, hist2, mask_width=0, p_weight=2.0, k=1, ctx: Context | None = None):
    """run_test.m:26-57 generalised to top-k: returns (idx int32 [m,k] 0-based, score float64 [m,k] as MATLAB holds it;
    float32 for gist / bow, whose distances are a single fp32 matrix)."""
    ctx = ctx or default_context()
    t = {"sc": TYPE_SC, "m2dp": TYPE_M2DP, "delight": TYPE_DELIGHT, "gist": TYPE_GIST, "bow": TYPE_BOW}.get(type_, type_)
    if t in (TYPE_GIST, TYPE_BOW):
        h1 = np.ascontiguousarray(hist1, np.float64); h2 = np.ascontiguousarray(hist2, np.float64)
        div = 2 if t == TYPE_BOW else 1
        if h1.shape[1] != h2.shape[1] or h1.shape[0] % div or h2.shape[0] % div:
            raise ValueError("hist1 / hist2 shapes do not fit the type")
        m, n = h1.shape[0] // div, h2.shape[0] // div
        idx = np.empty((m, k), np.int32); sc = np.empty((m, k), np.float32)
        ctx.check(ctx.lib.pr_match_topk_cols(ctx.h, t, _ptr(h1), m, _ptr(h2), n, h1.shape[1], int(mask_width), int(k), _ptr(idx), _ptr(sc)))
        return idx, sc
    if t not in (TYPE_SC, TYPE_M2DP, TYPE_DELIGHT):
        raise ValueError("type must be 'sc', 'm2dp', 'delight', 'gist' or 'bow'")
    div, width = {TYPE_SC: (1, 2400), TYPE_M2DP: (4, 384), TYPE_DELIGHT: (16, 256)}[t]
    h1 = np.ascontiguousarray(hist1, np.float64)
    h2 = np.ascontiguousarray(hist2, np.float64)
    if h1.ndim != 2 or h2.ndim != 2 or h1.shape[1] != width or h2.shape[1] != width:
        raise ValueError(f"signature width must be {width}")
    if h1.shape[0] % div or h2.shape[0] % div:
        raise ValueError(f"signature matrices of this type hold {div} rows per signature")
    m, n = h1.shape[0] // div, h2.shape[0] // div
    idx = np.empty((m, k), np.int32)
    sc = np.empty((m, k), np.float64)
    ctx.check(ctx.lib.pr_match_topk_f64(ctx.h, t, _ptr(h1), m, _ptr(h2), n, int(mask_width), float(p_weight), int(k),
                                        _ptr(idx), _ptr(sc)))
    return idx, sc


def match_topk_fused(sc1, m2dp1, sc2, m2dp2, mask_width=0, p_weight=2.0, k=1, ctx: Context | None = None):
    """BASELINE config 5 (build-defined, no reference counterpart): SC [m, 2400] and M2DP [4 m, 384] signatures of the same
    places scored together - the four row z-scores added with weights p, 1, p, 1.  Returns (idx int32 [m,k], score float64)."""
    ctx = ctx or default_context()
    a1 = np.ascontiguousarray(sc1, np.float64); a2 = np.ascontiguousarray(sc2, np.float64)
    b1 = np.ascontiguousarray(m2dp1, np.float64); b2 = np.ascontiguousarray(m2dp2, np.float64)
    m, n = a1.shape[0], a2.shape[0]
    if b1.shape != (4 * m, 384) or b2.shape != (4 * n, 384) or a1.shape[1] != 2400 or a2.shape[1] != 2400:
        raise ValueError("need SC [m, 2400] and M2DP [4 m, 384] signatures of the same m (n) places")
    idx = np.empty((m, k), np.int32); sc = np.empty((m, k), np.float64)
    ctx.check(ctx.lib.pr_match_topk_fused_f64(ctx.h, _ptr(a1), _ptr(b1), m, _ptr(a2), _ptr(b2), n, int(mask_width), float(p_weight),
                                          int(k), _ptr(idx), _ptr(sc)))
    return idx, sc


def run_test(type_, hist1, hist2, gt1=None, gt2=None, loop_diff=None, mask_width=0, ctx: Context | None = None):
    """run_test.m:1.  Without ground truth: returns (diff_v, diff_idx) of run_test.m:57 (0-based indices).
    With gt1/gt2/loop_diff: returns (AUC, top_recall, lp_detected) through eval.precision_recall; the sweep ranks the QUERIES by their
    best score (run_test.m:58), so every query is then answered from its exact fp64 row unless the caller brings a context of its own
    (pr_set_exact_statistics: scores are the reference's doubles to rounding, two queries whose scores agree to 1e-5 keep their places)."""
    if gt1 is not None and ctx is None and type_ in ("sc", "m2dp", TYPE_SC, TYPE_M2DP):
        with _exact_statistics(default_context()) as c:            # the default context (its device, its arithmetic), exact statistics for this call
            idx, sc = match_topk(type_, hist1, hist2, mask_width, 2.0, 1, c)
    else:
        idx, sc = match_topk(type_, hist1, hist2, mask_width, 2.0, 1, ctx)
    if gt1 is None:
        return sc[:, 0], idx[:, 0]
    from . import eval as _eval
    return _eval.precision_recall(sc[:, 0], idx[:, 0], np.asarray(gt1), np.asarray(gt2), loop_diff, mask_width)[:3]


# ------------------------------------------------------------------------------- host-side rows a1 / a2
def pts_preprocess(poses_file: str, pts_file: str, incoming_id_file: str | None, lidarRange: float = 45.0,
                   polar_filter: bool = False, verbose: bool = False, gpu: bool = False, ctx: Context | None = None):
    """pts_preprocess(...) of utils/pts_preprocess.h:169-232 -> (xyz [T,3] f64, inten [T] f32, offs [N+1] i64, ids [N] i32).
    gpu=True runs the sliding-window / best-point-per-cell work on the device (same clouds, same point order)."""
    lib = _lib.load()
    h = C.c_void_p()
    args = (poses_file.encode(), pts_file.encode(), incoming_id_file.encode() if incoming_id_file else None,
            float(lidarRange), int(polar_filter), int(verbose), C.byref(h))
    if gpu:
        ctx = ctx or default_context()
        ctx.check(lib.pr_pts_preprocess_gpu(ctx.h, *args))
    else:
        rc = lib.pr_pts_preprocess(*args)
        if rc != 0:
            raise PRError(rc, lib.pr_host_last_error().decode())
    try:
        N = lib.pr_clouds_count(h)
        offs = np.ctypeslib.as_array(lib.pr_clouds_offs(h), (N + 1,)).copy()
        T = int(offs[-1])
        xyz = np.ctypeslib.as_array(lib.pr_clouds_xyz(h), (T, 3)).copy() if T else np.zeros((0, 3))
        it = np.ctypeslib.as_array(lib.pr_clouds_inten(h), (T,)).copy() if T else np.zeros((0,), np.float32)
        ids = np.ctypeslib.as_array(lib.pr_clouds_ids(h), (N,)).copy() if N else np.zeros((0,), np.int32)
        pts_preprocess.last_avg_ms = float(lib.pr_clouds_avg_ms(h))     # "generate_spherical_points average time"
    finally:
        lib.pr_clouds_free(h)
    return xyz, it, offs, ids


def hash_order(keys) -> np.ndarray:
    """Iteration order of a libstdc++ unordered_map<int,...> after inserting these distinct keys in this order."""
    k = np.ascontiguousarray(keys, np.int32)
    out = np.empty(len(k), np.int32)
    rc = _lib.load().pr_hash_order(_ptr(k), len(k), _ptr(out))
    if rc != 0:
        raise PRError(rc, "pr_hash_order")
    return out


def write_signatures(path: str, sig, dtype=np.float64) -> None:
    """`ofstream << Eigen::MatrixXd` text (test_sc.cpp:63-66); a path ending in .bin writes the binary side-car."""
    sig = np.ascontiguousarray(sig, np.float64)
    lib = _lib.load()
    if path.endswith(".bin"):
        rc = lib.pr_write_signatures_bin(path.encode(), _ptr(sig), sig.shape[0], sig.shape[1],
                                         _lib.F32 if dtype == np.float32 else _lib.F64)
    else:
        rc = lib.pr_write_signatures(path.encode(), _ptr(sig), sig.shape[0], sig.shape[1])
    if rc != 0:
        raise PRError(rc, lib.pr_host_last_error().decode())


def read_signatures(path: str) -> np.ndarray:
    lib = _lib.load()
    p = C.c_void_p(); r = C.c_int64(); c = C.c_int64()
    fn = lib.pr_read_signatures_bin if path.endswith(".bin") else lib.pr_read_signatures
    rc = fn(path.encode(), C.byref(p), C.byref(r), C.byref(c))
    if rc != 0:
        raise PRError(rc, lib.pr_host_last_error().decode())
    try:
        out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (r.value, c.value)).copy()
    finally:
        lib.pr_free(p)
    return out


def write_poses(path: str, ids, w2c) -> None:
    ids = np.ascontiguousarray(ids, np.int32); w2c = np.ascontiguousarray(w2c, np.float64).reshape(len(ids), 12)
    _lib.load().pr_write_poses(path.encode(), _ptr(ids), _ptr(w2c), len(ids))


def write_points(path: str, ids, xyz, inten) -> None:
    ids = np.ascontiguousarray(ids, np.int32); xyz = np.ascontiguousarray(xyz, np.float64)
    inten = np.ascontiguousarray(inten, np.float32)
    _lib.load().pr_write_points(path.encode(), _ptr(ids), _ptr(xyz), _ptr(inten), len(ids))
