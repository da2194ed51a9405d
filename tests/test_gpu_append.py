"""pr_sigset_reserve / pr_sigset_append (SC/test_sc.cpp:40-56 produces one signature per keyframe; run_test.m:57 matches against all of
them): a DB that grows in place.  The appended operand image must be the bulk pack's bit for bit, the matcher's answers the oracle's."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib
from so_dso_place_recognition_amd import _lib, api, synth
from so_dso_place_recognition_amd.matcher import Matcher

pytestmark = pytest.mark.gpu


def _image(ctx, s):
    p, nb, st = C.c_void_p(), C.c_size_t(), C.c_int32()
    assert ctx.lib.pr_sigset_image(s, C.byref(p), C.byref(nb), C.byref(st)) == 0
    ctx.sync()
    out = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), p, nb.value, 3) == 0
    return out.cpu().numpy(), st.value


@pytest.mark.parametrize("arith,type_", [("f16x2", "sc"), ("f16", "sc"), ("f16x2", "m2dp")])
def test_appended_image_is_the_bulk_pack_bit_for_bit(arith, type_):
    cap, n0, steps = 1000, 403, (1, 1, 7, 30, 1, 170)          # appends of one row, a few, many; across group boundaries (16 rows / 8 signatures)
    rows = 1 if type_ == "sc" else 4
    full = (synth.sc_database(61, n0 + sum(steps)) if type_ == "sc" else synth.m2dp_database(62, n0 + sum(steps)))
    ctx = api.Context(0, sc_arith=arith)
    tcode = _lib.TYPE_SC if type_ == "sc" else _lib.TYPE_M2DP
    a, b = C.c_void_p(), C.c_void_p()
    assert ctx.lib.pr_sigset_create(ctx.h, tcode, _lib.ROLE_DB, cap, C.byref(a)) == 0
    assert ctx.lib.pr_sigset_create(ctx.h, tcode, _lib.ROLE_DB, cap, C.byref(b)) == 0
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    ctx.check(ctx.lib.pr_sigset_reserve(ctx.h, a))
    ctx.check(ctx.lib.pr_sigset_pack(ctx.h, a, P(full), _lib.F64, _lib.HOST, n0))
    n = n0
    for k in steps:
        chunk = np.ascontiguousarray(full[n * rows:(n + k) * rows])
        ctx.check(ctx.lib.pr_sigset_append(ctx.h, a, P(chunk), _lib.F64, _lib.HOST, k))
        n += k
        assert ctx.lib.pr_sigset_count(a) == n
    ctx.check(ctx.lib.pr_sigset_reserve(ctx.h, b))
    ctx.check(ctx.lib.pr_sigset_pack(ctx.h, b, P(full), _lib.F64, _lib.HOST, n))
    ia, sa = _image(ctx, a)
    ib, sb = _image(ctx, b)
    assert sa == sb and sa == ((cap + 15) // 16 if type_ == "sc" else (cap + 7) // 8)    # the capacity's stride, not the count's
    assert np.array_equal(ia, ib)
    # an empty set reserves by itself on its first append; a set packed in the count's geometry refuses; the capacity is enforced
    c = C.c_void_p()
    assert ctx.lib.pr_sigset_create(ctx.h, tcode, _lib.ROLE_DB, 40, C.byref(c)) == 0
    ctx.check(ctx.lib.pr_sigset_append(ctx.h, c, P(full), _lib.F64, _lib.HOST, 33))
    assert ctx.lib.pr_sigset_count(c) == 33
    assert ctx.lib.pr_sigset_append(ctx.h, c, P(full), _lib.F64, _lib.HOST, 8) == -1 and b"capacity" in ctx.lib.pr_last_error(ctx.h)
    d = C.c_void_p()
    assert ctx.lib.pr_sigset_create(ctx.h, tcode, _lib.ROLE_DB, 40, C.byref(d)) == 0
    ctx.check(ctx.lib.pr_sigset_pack(ctx.h, d, P(full), _lib.F64, _lib.HOST, 20))
    assert ctx.lib.pr_sigset_append(ctx.h, d, P(full), _lib.F64, _lib.HOST, 1) == -1 and b"pr_sigset_reserve" in ctx.lib.pr_last_error(ctx.h)
    for s in (a, b, c, d):
        ctx.lib.pr_sigset_destroy(ctx.h, s)
    ctx.close()


@pytest.mark.parametrize("arith", ["f16x2", "f16"])
def test_growing_database_answers_like_the_oracle(arith):
    """An online loop: every keyframe is matched against the DB so far (mask 0: the synthetic queries are planted copies, not the frames
    themselves), then appended.  Indices bit-exact against the oracle on the DB of that moment; the binary intensity channel keeps its
    single-product pass (statistics folded in by the appends)."""
    cap, n0 = 900, 300
    db = synth.sc_database(71, cap)
    mt = Matcher("sc", 8, cap, ctx=api.Context(0, sc_arith=arith))
    dbd = torch.from_numpy(db).cuda()
    mt.reserve_database(dbd[:n0])
    n = n0
    for step, k in enumerate((1, 1, 1, 5, 16, 1, 100, 1)):
        q, planted = synth.sc_queries(100 + step, db[:n], 8 if step % 2 else 1)
        idx, sc = mt.match(torch.from_numpy(q).cuda(), 0, 2.0, 2)
        rc, oidx, osc = oracle_lib.match_topk(0, q, db[:n], 0, 2.0, 2)
        assert rc == 0 and np.array_equal(idx.cpu().numpy(), oidx), (step, n)
        mt.append_database(dbd[n:n + k])
        n += k
        assert mt.n == n
    if arith == "f16x2":
        st = C.c_int32(-1)
        mt.ctx.check(mt.lib.pr_sc_binary_state(mt.ctx.h, mt.q, mt.db, C.byref(st)))
        assert st.value == 1                                          # the appended set is still recognised as binary in channel 1
    # a non-binary row appended: the statistics say so at once and the channel goes back to split-f16 (same answers)
    odd = db[:1].copy()
    odd[0, 1200:] *= np.linspace(0.5, 1.5, 1200)
    mt.append_database(torch.from_numpy(odd).cuda())
    dbx = np.concatenate([db[:n], odd])
    q, _ = synth.sc_queries(300, dbx, 8)
    idx, sc = mt.match(torch.from_numpy(q).cuda(), 0, 2.0, 1)
    rc, oidx, osc = oracle_lib.match_topk(0, q, dbx, 0, 2.0, 1)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    if arith == "f16x2":
        st = C.c_int32(-1)
        mt.ctx.check(mt.lib.pr_sc_binary_state(mt.ctx.h, mt.q, mt.db, C.byref(st)))
        assert st.value != 1
    mt.close()


def test_appendable_set_needs_the_default_matcher(monkeypatch):
    monkeypatch.setenv("PR_SC_KERNEL", "h")
    ctx = api.Context(0)
    q, d = C.c_void_p(), C.c_void_p()
    assert ctx.lib.pr_sigset_create(ctx.h, _lib.TYPE_SC, _lib.ROLE_QUERY, 8, C.byref(q)) == 0
    assert ctx.lib.pr_sigset_create(ctx.h, _lib.TYPE_SC, _lib.ROLE_DB, 64, C.byref(d)) == 0
    ctx.check(ctx.lib.pr_sigset_reserve(ctx.h, d))
    assert ctx.lib.pr_sigset_reserve(ctx.h, q) == -1                 # DB sets only
    dummy = C.c_void_p(16)
    assert ctx.lib.pr_distances_dev(ctx.h, q, d, dummy, dummy) == -1 and b"default matcher" in ctx.lib.pr_last_error(ctx.h)
    ctx.lib.pr_sigset_destroy(ctx.h, q); ctx.lib.pr_sigset_destroy(ctx.h, d)
    ctx.close()


@pytest.mark.parametrize("shards,type_", [(1, "sc"), (3, "sc"), (2, "m2dp")])
def test_pr_group_database_grows_in_place(shards, type_):
    """The host-buffer form of the online loop (pr_group: what `match_signatures` and a MATLAB caller see): a database with room to grow on its
    last shard, every keyframe matched against the rows so far (mask 3, k 2) and then appended; answers = the oracle's on the database of that moment."""
    n0, extra = 250, 120
    tcode = 0 if type_ == "sc" else 1
    rows = 1 if type_ == "sc" else 4
    full = synth.sc_database(91, n0 + extra) if type_ == "sc" else synth.m2dp_database(92, n0 + extra)
    g = api.Group([0] * shards)
    g.set_database(type_, full[:n0 * rows], extra_capacity=extra)
    n = n0
    for step, k_new in enumerate((1, 1, 2, 17, 1, 60)):
        q = (synth.sc_queries(200 + step, full[:n], 6)[0] if type_ == "sc" else synth.m2dp_queries(200 + step, full[:n * rows], 6)[0])
        idx, sc = g.match_topk(q, 3, 2.0, 2)
        rc, oidx, osc = oracle_lib.match_topk(tcode, q, full[:n * rows], 3, 2.0, 2)
        assert rc == 0 and np.array_equal(idx, oidx), (step, n)
        g.append_database(full[n * rows:(n + k_new) * rows])
        n += k_new
        assert g.database_rows == n
    with pytest.raises(api.PRError):
        g.append_database(full[:(extra + 1) * rows])                 # past the reserve
    g.close()
    g = api.Group([0])
    g.set_database(type_, full[:n0 * rows])                          # not growable: no room
    with pytest.raises(api.PRError):
        g.append_database(full[:rows])
    g.close()
