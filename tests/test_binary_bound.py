"""The error bound behind the binary-channel path (csrc/kernels.hpp: sc_bin_bound, csrc/sc_match_e.hip: ep_store_round),
checked on the CPU against a numpy emulation of the single-product pass's arithmetic: spectra of the normalised rows (processSC.m:15-20 +
the sector DFT) rounded to f16, S_f = sum_r Q conj(D) accumulated in fp32 and rounded to f16, stage-2 constants w_f (cos, -sin) 2^10 rounded
to f16, c_k in fp32, count = max_k c_k sqrt(ones_q ones_d).  For every pair: |computed count - true integer count| <= b, with
b = sqrt(ones_q ones_d) (eq + ed + eq ed + bconst (1 + eq)(1 + ed)), eq / ed the measured residual norms of the rounded spectra - and the
per-pair test |x - rint(x)| + b < 1 then makes rint(x) the exact count.  No GPU, no oracle: this is host logic (the bound's constants)."""
import numpy as np
import pytest

U = 2.0 ** -11
W = np.full(31, 2.0); W[0] = 1.0; W[30] = 1.0


def _f16(x):
    return x.astype(np.float32).astype(np.float16).astype(np.float64)


def _spectra(rows, scale):
    """rows [n, 1200] (bin = sector * 20 + ring) -> exact and f16-rounded spectra [n, 31, 20] of the normalised rows, residual norms, ones."""
    x = rows.reshape(-1, 60, 20)
    nr = np.sqrt((x * x).sum((1, 2)))
    X = np.fft.rfft(x, axis=1) / np.sqrt(60.0) / nr[:, None, None]
    Xh = (_f16(X.real * scale) + 1j * _f16(X.imag * scale)) / scale
    eps = np.sqrt((W[None, :, None] * np.abs(X - Xh) ** 2).sum((1, 2)))
    return X, Xh, eps, (x != 0).sum((1, 2))


def _constants():
    k = np.arange(31); f = np.arange(31)
    th = 2 * np.pi * np.outer(k, f) / 60
    C, S = W * np.cos(th) * 1024, -W * np.sin(th) * 1024
    Ch, Sh = _f16(C), _f16(S)
    gamma = (np.sqrt((C - Ch) ** 2 + (S - Sh) ** 2) / (W[None, :] * 1024)).max()
    return C, S, Ch, Sh, gamma


def _max_corr(Q, D, C, S, emulate):
    """max over the 120 variants of the normalised product, exact (fp64) or as the single-product pass forms it"""
    out = []
    for Z in (np.einsum('qfr,efr->qef', Q, D.conj()), np.einsum('qfr,efr->qef', Q, D)):      # forward | mirror
        re, im = Z.real, Z.imag
        if emulate:
            re, im = _f16(re.astype(np.float32).astype(np.float64) * 2 ** 15) / 2 ** 15, _f16(im.astype(np.float32).astype(np.float64) * 2 ** 15) / 2 ** 15
        E = np.einsum('kf,qef->qek', C, re); O = np.einsum('kf,qef->qek', S, im)
        if emulate:
            E, O = E.astype(np.float32).astype(np.float64), O.astype(np.float32).astype(np.float64)
        out.append((E + np.abs(O)).max(-1))                                                   # shifts k and 60 - k
    return np.maximum(out[0], out[1]) / 1024


def _check(q, d):
    Q, Qh, eq, oq = _spectra(q, 256.0)
    D, Dh, ed, od = _spectra(d, 128.0)
    C, S, Ch, Sh, gamma = _constants()
    N = np.sqrt(np.outer(oq, od))
    exact = _max_corr(Q, D, C, S, False) * N
    assert np.abs(exact - np.rint(exact)).max() < 1e-9                 # processSC.m:30 on binary rows: integer counts
    x = _max_corr(Qh, Dh, Ch, Sh, True) * N
    bconst = (U + gamma) * (1 + U) + 7e-5                              # pr_api.cpp: create_common
    Eq, Ed = eq[:, None], ed[None, :]
    b = N * (Eq + Ed + Eq * Ed + bconst * (1 + Eq) * (1 + Ed))
    err = np.abs(x - exact)
    assert (err <= b).all(), (err / b).max()
    ok = np.abs(x - np.rint(x)) + b < 0.98                             # ep_store_round's test ...
    assert (np.rint(x)[ok] == np.rint(exact)[ok]).all()                # ... certifies the rounded count
    return err.max(), (err / b).max(), ok.mean(), gamma, eq.max(), ed.max()


def test_stage2_constants_residual():
    gamma = _constants()[4]
    assert 0.2 * U < gamma < U                                         # (the library computes the same number at context creation)


@pytest.mark.parametrize("density", [0.05, 0.27, 0.45])
def test_bound_holds_for_random_binary_rows(density):
    rng = np.random.default_rng(int(density * 100))
    d = (rng.random((60, 1200)) < density).astype(np.float64)
    q = np.roll(d[:12].reshape(12, 60, 20), 13, axis=1).reshape(12, 1200).copy()      # rotated copies (planted matches) ...
    flip = rng.random(q.shape) < 0.03
    q = np.where(flip, 1.0 - q, q)                                                   # ... with 3 % of the bins re-drawn
    emax, ratio, okfrac, gamma, eq, ed = _check(q, d)
    assert ratio < 0.5                                                               # the bound is not tight: ~6x the observed error
    if density <= 0.27:
        assert okfrac == 1.0                                                         # every pair passes the rounding test at these sizes


def test_bound_holds_for_adversarial_rows():
    """Rows built to align all their energy (sum_f w_f |S_f| at its maximum 1: identical rows, single rings, periodic patterns),
    scaled rows (any positive constant value normalises to the same 1/sqrt(ones))."""
    rows = []
    x = np.zeros((60, 20)); x[:, 3] = 1; rows.append(x)                              # one full ring
    x = np.zeros((60, 20)); x[::2, :] = 1; rows.append(x)                            # every second sector
    x = np.zeros((60, 20)); x[::3, ::2] = 1; rows.append(x)
    x = np.zeros((60, 20)); x[7, :] = 1; rows.append(x)                              # one sector
    x = np.zeros((60, 20)); x[:30, :10] = 1; rows.append(x)                          # a block
    x = np.zeros((60, 20)); x[5, 5] = 1; rows.append(x)                              # a single bin
    rng = np.random.default_rng(5)
    rows.append((rng.random((60, 20)) < 0.3).astype(float))
    d = np.stack([r.reshape(1200) for r in rows])
    q = np.concatenate([d, np.roll(d.reshape(-1, 60, 20), 1, axis=1).reshape(-1, 1200), 3.5 * d[:3]])
    _check(q, d)
