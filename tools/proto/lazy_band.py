#!/usr/bin/env python
"""NumPy twin of the deferred-update form of stage 1 (csrc/herm_band.h, `ml_reduce` = 0): the two-sided updates of up
to NB panels stay pending -- the stored matrix is touched by every NB-th sweep only, the sweeps between read it and
the panel kernel corrects what they produce:

    Z_k = A_stored V_k - sum_{p pending} [ V_p (X_p^H V_k) + X_p (V_p^H V_k) ]
    M_k = V_k^H Z_k    = M_raw  - sum_p [ S2_p^H S1_p + S1_p^H S2_p ],   S1_p = X_p^H V_k,  S2_p = V_p^H V_k
    P_k = A_stored[:, cols_k] - sum_{p pending} [ X_p V_p[cols]^H + V_p X_p[cols]^H ]

Written in the kernels' order of events (panel kernel k finishes update k-1 and starts panel k; sweep k is a flush
when NB updates are pending).  Checked against the eager form of tools/proto/twostage.py.

    python tools/proto/lazy_band.py [n] [NB]
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from twostage import larfg, stage1  # noqa: E402


def panel_qr(P, b):
    npr = P.shape[0]
    V = np.zeros((npr, b), complex)
    taus = np.zeros(b, complex)
    for c in range(min(b, npr)):
        v, tau, beta = larfg(P[c:, c].copy())
        V[c:, c] = v
        taus[c] = tau
        P[c:, c] = 0
        P[c, c] = beta
        w = np.conj(tau) * (v.conj() @ P[c:, c + 1 :])
        P[c:, c + 1 :] -= np.outer(v, w)
    T = np.zeros((b, b), complex)
    G = V.conj().T @ V
    for c in range(b):
        T[c, c] = taus[c]
        T[:c, c] = -taus[c] * (T[:c, :c] @ G[:c, c])
    return P, V, T


def stage1_lazy(A, b, NB):
    A = A.copy()
    n = A.shape[0]
    K = n // b - 1
    Vg = {}  # panel -> V indexed by GLOBAL row (zero outside the support)
    Xg = {}
    Tk = {}
    p0 = 0  # first pending panel
    Zraw = Mraw = None
    diag = A.diagonal().real.copy()
    for k in range(K + 1):
        j0, o = b * k, b * k + b
        # ---- panel kernel k: finish update k-1
        if k > 0:
            V1 = Vg[k - 1]
            Z = Zraw.copy()
            M = Mraw.copy()
            for p in range(p0, k - 1):
                S1 = Xg[p].conj().T @ V1
                S2 = Vg[p].conj().T @ V1
                Z -= Vg[p] @ S1 + Xg[p] @ S2
                M -= S2.conj().T @ S1 + S1.conj().T @ S2
            assert np.allclose(M, V1.conj().T @ Z, atol=1e-9 * np.abs(A).max())
            T = Tk[k - 1]
            X = Z @ T - 0.5 * V1 @ (T.conj().T @ M @ T)
            X[:j0] = 0
            Xg[k - 1] = X
            diag -= 2.0 * np.real(np.sum(X * V1.conj(), axis=1))
        # ---- the panel's columns with every pending update applied
        P = A[:, j0:o].copy()
        for p in range(p0, k):
            P -= Xg[p] @ Vg[p][j0:o].conj().T + Vg[p] @ Xg[p][j0:o].conj().T
        A[j0:o, j0:o] = P[j0:o]
        for p in range(p0, k):  # rows [j0, o) of the pending operands: zero from here on
            Vg[p][j0:o] = 0
            Xg[p][j0:o] = 0
        if k == K:
            break
        R, V, T = panel_qr(P[o:].copy(), b)
        A[o:, j0:o] = R
        A[j0:o, o:] = R.conj().T
        Vg[k] = np.zeros((n, b), complex)
        Vg[k][o:] = V
        Tk[k] = T
        # ---- sweep k: a flush when NB updates are pending
        if k - p0 == NB:
            for p in range(p0, k):
                A[o:, o:] -= Xg[p][o:] @ Vg[p][o:].conj().T + Vg[p][o:] @ Xg[p][o:].conj().T
            p0 = k
        Zraw = np.zeros((n, b), complex)
        Zraw[o:] = A[o:, o:] @ V
        Mraw = V.conj().T @ Zraw[o:]
    refl = [(b * k + b, Vg[k][b * k + b :], Tk[k]) for k in range(K)]
    return A, refl, diag


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    b = 8
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, n + 10)) + 1j * rng.standard_normal((n, n + 10))
    G = X @ X.conj().T
    Ab, refl = stage1(G, b)
    for nb in (1, 2, NB, 8):
        Al, refl_l, diag = stage1_lazy(G, b, nb)
        band = lambda M: np.tril(np.triu(M, -b), b)  # noqa: E731
        err = np.abs(band(Al) - band(Ab)).max() / np.abs(G).max()
        print(f"NB = {nb}: band vs eager form {err:.2e}; eigenvalues {np.abs(np.linalg.eigvalsh(band(Al)) - np.linalg.eigvalsh(G)).max() / np.abs(G).max():.2e}")
        assert err < 1e-12


if __name__ == "__main__":
    main()
