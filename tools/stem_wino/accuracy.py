"""Accuracy of a Winograd form of the 7x7 / stride 2 stem (fp32), against the fp64 direct sum.

The stride-2 correlation splits into four stride-1 phase correlations (taps 4x4, 4x3, 3x4, 3x3 on the even/odd
sub-lattices); each is computed as F(2x2, r x s): 25 + 20 + 20 + 16 = 81 multiplications per 2x2 output tile and (cin, cout)
instead of 196.  The 4-tap factor needs five interpolation points; this script measures what the choice costs.
"""
import itertools
import sys
from fractions import Fraction as Fr

import numpy as np


def toom_cook(m, r, pts):
    """A^T (m x n), G (n x r), B^T (n x n) of F(m, r) with n = m + r - 1 points `pts` (n - 1 finite ones + infinity):
    y = A^T [(G g) * (B^T d)].  Exact rationals."""
    n = m + r - 1
    assert len(pts) == n - 1
    pts = [Fr(p) for p in pts]
    # polynomial M(x) = prod (x - p_i); N_i = M / (x - p_i)
    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    M = [Fr(1)]
    for p in pts:
        M = polymul(M, [-p, Fr(1)])
    AT = [[Fr(0)] * n for _ in range(m)]
    G = [[Fr(0)] * r for _ in range(n)]
    BT = [[Fr(0)] * n for _ in range(n)]
    for i, p in enumerate(pts):
        Ni = [Fr(1)]
        for j, q in enumerate(pts):
            if j != i:
                Ni = polymul(Ni, [-q, Fr(1)])
        scale = Fr(1)
        for j, q in enumerate(pts):
            if j != i:
                scale *= (p - q)
        for k in range(m):
            AT[k][i] = p ** k
        for k in range(r):
            G[i][k] = p ** k / scale
        for k in range(n - 1):
            BT[i][k] = Ni[k]
    # the point at infinity
    AT[m - 1][n - 1] = Fr(1)
    G[n - 1][r - 1] = Fr(1)
    for k in range(n):
        BT[n - 1][k] = M[k]
    return AT, G, BT


def check(m, r, pts):
    AT, G, BT = toom_cook(m, r, pts)
    n = m + r - 1
    rng = np.random.default_rng(0)
    g = rng.standard_normal(r); d = rng.standard_normal(n)
    f = lambda Mx: np.array([[float(v) for v in row] for row in Mx])
    y = f(AT) @ ((f(G) @ g) * (f(BT) @ d))
    want = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    assert np.abs(y - want).max() < 1e-9, (y, want)
    return f(AT), f(G), f(BT)


def rescale(AT, G, BT, s):
    """Move per-point scale factors between G and B^T: (G_i / s_i) * (s_i B^T_i)."""
    s = np.asarray(s, dtype=np.float64)
    return AT, G / s[:, None], BT * s[:, None]


def stem_winograd(x, w, T4, T3, dtype):
    """x (C, 229 + ..) padded input as (C, Hp, Wp) with Hp = 2 * Ho + 5; w (Co, C, 7, 7).  Returns (Co, Ho, Wo)."""
    C, Hp, Wp = x.shape
    Co = w.shape[0]
    Ho, Wo = (Hp - 5) // 2, (Wp - 5) // 2
    y = np.zeros((Co, Ho, Wo), dtype=dtype)
    for ry, rx in itertools.product((0, 1), (0, 1)):
        ATy, Gy, BTy = T4 if ry == 0 else T3
        ATx, Gx, BTx = T4 if rx == 0 else T3
        ATy, Gy, BTy, ATx, Gx, BTx = [a.astype(dtype) for a in (ATy, Gy, BTy, ATx, Gx, BTx)]
        wp = w[:, :, ry::2, rx::2].astype(np.float64)                       # (Co, C, ty, tx)
        U = np.einsum("ia,ocab,jb->ijco", Gy.astype(np.float64), wp, Gx.astype(np.float64)).astype(dtype)   # host side, fp64 then rounded
        xp = x[:, ry::2, rx::2].astype(dtype)                               # phase image
        ny, nx = BTy.shape[0], BTx.shape[0]
        ty, tx = Ho // 2, Wo // 2
        # patches: tile (p, q) reads phase rows 2p .. 2p + ny - 1
        idx_y = (2 * np.arange(ty))[:, None] + np.arange(ny)[None, :]
        idx_x = (2 * np.arange(tx))[:, None] + np.arange(nx)[None, :]
        d = xp[:, idx_y][:, :, :, idx_x]                                    # (C, ty, ny, tx, nx)
        t = np.einsum("ia,cpaqb->cpiqb", BTy, d).astype(dtype)
        V = np.einsum("jb,cpiqb->ijpqc", BTx, t).astype(dtype)
        Mm = np.zeros((ny, nx, ty, tx, Co), dtype=dtype)
        for c in range(C):                                                  # sequential fp32 accumulation over channels (as an MFMA chain)
            Mm += V[:, :, :, :, c, None] * U[:, :, None, None, c, :]
        s = np.einsum("ai,ijpqo->ajpqo", ATy, Mm).astype(dtype)
        Y = np.einsum("bj,ajpqo->opaqb", ATx, s).astype(dtype)              # (Co, ty, 2, tx, 2)
        y += Y.reshape(Co, Ho, Wo)
    return y


def main():
    rng = np.random.default_rng(1)
    C, Co, H = 18, 64, 64
    Ho = H // 2
    # stem-like input: edge map (0/1), heat maps in [0, 1]; and a Gaussian case
    cases = {}
    x = np.zeros((C, H, H))
    x[0] = (rng.random((H, H)) < 0.1).astype(np.float64)
    yy, xx = np.mgrid[0:H, 0:H]
    for j in range(1, C):
        cy, cx = rng.uniform(8, H - 8, 2)
        x[j] = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 4.0 ** 2))
    cases["proxy"] = x
    cases["gauss"] = rng.standard_normal((C, H, H))
    cases["uniform+"] = rng.random((C, H, H))
    w = rng.standard_normal((Co, C, 7, 7)) * np.sqrt(2.0 / (C * 49))       # kaiming
    variants = {
        "pts4 (0,1,-1,2)": ((0, 1, -1, 2), None),
        "pts4 (0,1,-1,1/2)": ((0, 1, -1, Fr(1, 2)), None),
        "pts4 (0,1,-1,-1/2)": ((0, 1, -1, Fr(-1, 2)), None),
        "pts4 (0,1/2,-1/2,1)": ((0, Fr(1, 2), Fr(-1, 2), 1), None),
        "pts4 (0,1,-1,-2)": ((0, 1, -1, -2), None),
    }
    T3 = check(2, 3, (0, 1, -1))
    for name, (pts, _) in variants.items():
        T4 = check(2, 4, pts)
        print("==", name)
        print(" B^T =", np.array2string(T4[2], precision=3).replace("\n", ""))
        print(" A^T =", np.array2string(T4[0], precision=3).replace("\n", ""))
        for cname, xin in cases.items():
            xpad = np.zeros((C, H + 5, H + 5))               # pad 3 before, 2 after (the last tap row of an even kernel phase)
            xpad[:, 3:3 + H, 3:3 + H] = xin
            # fp64 direct
            ref = np.zeros((Co, Ho, Ho))
            for ky in range(7):
                for kx in range(7):
                    ref += np.einsum("oc,chw->ohw", w[:, :, ky, kx], xpad[:, ky:ky + 2 * Ho:2, kx:kx + 2 * Ho:2])
            # fp32 direct (sequential over taps)
            d32 = np.zeros((Co, Ho, Ho), dtype=np.float32)
            for ky in range(7):
                for kx in range(7):
                    for c in range(C):
                        d32 += w[:, c, ky, kx].astype(np.float32)[:, None, None] * xpad[c, ky:ky + 2 * Ho:2, kx:kx + 2 * Ho:2].astype(np.float32)[None]
            w64 = stem_winograd(xpad, w, T4, T3, np.float64)
            assert np.abs(w64 - ref).max() < 1e-10 * max(1, np.abs(ref).max()), np.abs(w64 - ref).max()
            w32 = stem_winograd(xpad, w, T4, T3, np.float32)
            sc = np.abs(ref).max()
            print("   %-9s scale %.3f  direct fp32 err %.2e   winograd fp32 err %.2e (of scale), rms %.2e" %
                  (cname, sc, np.abs(d32 - ref).max() / sc, np.abs(w32 - ref).max() / sc, np.sqrt(((w32 - ref) ** 2).mean()) / sc))


if __name__ == "__main__":
    main()
