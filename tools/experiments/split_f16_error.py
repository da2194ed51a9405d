"""Numerical experiment: error of the SC ring-correlation max when both stages run as split-f16 (hi+lo, 3 products,
fp32 accumulate) instead of fp32.  Not shipped; numpy only."""
import sys
import numpy as np
sys.path.insert(0, ".")
from so_dso_place_recognition_amd import synth


def split(x, scale):
    xs = (x * scale).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def prod3(ah, al, bh, bl, sub):
    """sum over `sub` of a*b with 3 split products, fp32 result"""
    r = np.einsum(sub, ah, bh) + np.einsum(sub, ah, bl) + np.einsum(sub, al, bh)
    return r.astype(np.float32).astype(np.float64)


def main(n=64, m=16, four=False):
    db = synth.sc_database(45, n)
    q, _ = synth.sc_queries(46, db, m)
    worst = 0.0
    worst32 = 0.0
    for ch in range(2):
        a = q[:, ch * 1200:(ch + 1) * 1200]; b = db[:, ch * 1200:(ch + 1) * 1200]
        a = a / np.linalg.norm(a, axis=1, keepdims=True); b = b / np.linalg.norm(b, axis=1, keepdims=True)
        A = np.fft.rfft(a.reshape(m, 60, 20), axis=1) / np.sqrt(60)     # [m, 31, 20]
        B = np.fft.rfft(b.reshape(n, 60, 20), axis=1) / np.sqrt(60)
        w = np.full(31, 2.0); w[0] = w[30] = 1.0
        k = np.arange(60)
        ang = 2 * np.pi * np.outer(np.arange(31), k) / 60
        Cc, Cs = w[:, None] * np.cos(ang), w[:, None] * np.sin(ang)     # [31, 60]
        # exact
        S = np.einsum("qfr,dfr->qdf", A, B.conj()); P = np.einsum("qfr,dfr->qdf", A, B)
        def corr(Z):
            return np.einsum("qdf,fk->qdk", Z.real, Cc) - np.einsum("qdf,fk->qdk", Z.imag, Cs)
        exact = np.maximum(corr(S).max(2), corr(P).max(2))
        # fp32 emulation (everything rounded to fp32 at the same places)
        A32r, A32i, B32r, B32i = [x.astype(np.float32).astype(np.float64) for x in (A.real, A.imag, B.real, B.imag)]
        def f32(x): return x.astype(np.float32).astype(np.float64)
        Sr = f32(np.einsum("qfr,dfr->qdf", A32r, B32r) + np.einsum("qfr,dfr->qdf", A32i, B32i))
        Si = f32(np.einsum("qfr,dfr->qdf", A32i, B32r) - np.einsum("qfr,dfr->qdf", A32r, B32i))
        Pr = f32(np.einsum("qfr,dfr->qdf", A32r, B32r) - np.einsum("qfr,dfr->qdf", A32i, B32i))
        Pi = f32(np.einsum("qfr,dfr->qdf", A32i, B32r) + np.einsum("qfr,dfr->qdf", A32r, B32i))
        c32, s32 = f32(Cc), f32(Cs)
        e32 = np.maximum(f32(np.einsum("qdf,fk->qdk", Sr, c32) - np.einsum("qdf,fk->qdk", Si, s32)).max(2),
                         f32(np.einsum("qdf,fk->qdk", Pr, c32) - np.einsum("qdf,fk->qdk", Pi, s32)).max(2))
        worst32 = max(worst32, np.abs(e32 - exact).max() / 2)
        # split-f16
        sa, sb, sc = 2.0 ** 8, 2.0 ** 7, 2.0 ** 10
        Arh, Arl = split(A.real, sa); Aih, Ail = split(A.imag, sa)
        Brh, Brl = split(B.real, sb); Bih, Bil = split(B.imag, sb)
        sub = "qfr,dfr->qdf"
        def p3(xh, xl, yh, yl):
            r = np.einsum(sub, xh, yh) + np.einsum(sub, xh, yl) + np.einsum(sub, xl, yh)
            if four: r = r + np.einsum(sub, xl, yl)
            return r
        rr, ii = p3(Arh, Arl, Brh, Brl), p3(Aih, Ail, Bih, Bil)
        ir, ri = p3(Aih, Ail, Brh, Brl), p3(Arh, Arl, Bih, Bil)
        Sr, Si, Pr, Pi = f32(rr + ii), f32(ir - ri), f32(rr - ii), f32(ir + ri)           # scaled by 2^15
        Cch, Ccl = split(Cc, sc); Csh, Csl = split(Cs, sc)
        def st2(Zr, Zi):
            zrh, zrl = split(Zr, 1.0); zih, zil = split(Zi, 1.0)
            s2 = "qdf,fk->qdk"
            def p(xh, xl, ch_, cl_):
                r = np.einsum(s2, xh, ch_) + np.einsum(s2, xl, ch_) + np.einsum(s2, xh, cl_)
                if four: r = r + np.einsum(s2, xl, cl_)
                return r
            return f32(p(zrh, zrl, Cch, Ccl) - p(zih, zil, Csh, Csl)) / (sa * sb * sc)
        got = np.maximum(st2(Sr, Si).max(2), st2(Pr, Pi).max(2))
        worst = max(worst, np.abs(got - exact).max() / 2)
    print(f"distance error: fp32 emulation {worst32:.3e}, split-f16 ({'4' if four else '3'} products) {worst:.3e}")


if __name__ == "__main__":
    main()
    main(four=True)
