#!/usr/bin/env python
"""NumPy prototype of the two-stage Hermitian tridiagonalisation the ML eigen path uses (csrc/herm_band.h), written the
way the kernels work: stage 1 = panels of b columns (QR of the sub-panel, two-sided update A -= X V^H + V X^H with
X = Z T - V (T^H M T) / 2, Z = A V), stage 2 = bulge chasing on b x b blocks with the leftover bulge triangles kept
apart from the band.  Checks T's eigenvalues and x = Q f(T) Q^H rhs against numpy.linalg.eigh.

    python tools/proto/twostage.py [n] [b]
"""
import sys

import numpy as np


def larfg(x):
    """LAPACK zlarfg: H^H x = beta e_1, H = I - tau v v^H, v[0] = 1, beta real."""
    alpha = x[0]
    xn = np.linalg.norm(x[1:])
    if xn == 0.0 and alpha.imag == 0.0:
        return np.concatenate([[1.0], np.zeros(len(x) - 1)]).astype(complex), 0.0, alpha.real
    beta = -np.copysign(np.sqrt(abs(alpha) ** 2 + xn**2), alpha.real)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def stage1(A, b):
    """Dense -> band (lower bandwidth b).  Returns the band matrix (dense storage) and the block reflectors."""
    A = A.copy()
    n = A.shape[0]
    refl = []
    j0 = 0
    while n - (j0 + b) >= 2:
        o = j0 + b
        P = A[o:, j0 : j0 + b].copy()
        npr = P.shape[0]
        V = np.zeros((npr, b), complex)
        taus = np.zeros(b, complex)
        for c in range(min(b, npr)):
            v, tau, beta = larfg(P[c:, c].copy())
            V[c:, c] = v
            taus[c] = tau
            P[c:, c] = 0
            P[c, c] = beta
            w = np.conj(tau) * (v.conj() @ P[c:, c + 1 :])  # H^H = I - conj(tau) v v^H
            P[c:, c + 1 :] -= np.outer(v, w)
        T = np.zeros((b, b), complex)
        G = V.conj().T @ V
        for c in range(b):
            T[c, c] = taus[c]
            T[:c, c] = -taus[c] * (T[:c, :c] @ G[:c, c])
        A22 = A[o:, o:]
        Z = A22 @ V
        M = V.conj().T @ Z
        X = Z @ T - 0.5 * V @ (T.conj().T @ M @ T)
        A22 -= X @ V.conj().T + V @ X.conj().T
        A[o:, j0 : j0 + b] = P
        A[j0 : j0 + b, o:] = P.conj().T
        refl.append((o, V, T))
        j0 += b
    return A, refl


def apply_q1(refl, x, adjoint):
    """x <- Q1^H x (adjoint) or Q1 x, Q1 = prod_k (I - V_k T_k V_k^H)."""
    x = x.copy()
    for o, V, T in refl if adjoint else reversed(refl):
        s = V.conj().T @ x[o:]
        x[o:] -= V @ ((T.conj().T if adjoint else T) @ s)
    return x


def stage2(Ab, b):
    """Band -> real tridiagonal by bulge chasing with length-b reflectors.  Blocks of sweep j: diagonal D_s on
    R_s = [j+1+s b, j+1+(s+1) b), off-diagonal O_s = rows R_{s+1} x cols R_s.  Returns d, e and the reflector log
    [(row0, v, tau)] in generation order."""
    A = Ab.copy()
    n = A.shape[0]
    log = []
    for j in range(n - 1):
        r0 = j + 1
        r1 = min(r0 + b, n)
        v, tau, beta = larfg(A[r0:r1, j].copy())
        A[r0:r1, j] = 0
        A[r0, j] = beta
        A[j, r0:r1] = np.conj(A[r0:r1, j])
        log.append((j, r0, v, tau))
        while True:
            # two-sided on the diagonal block of the reflector's rows
            H = np.eye(r1 - r0) - tau * np.outer(v, v.conj())
            A[r0:r1, r0:r1] = H.conj().T @ A[r0:r1, r0:r1] @ H
            q0, q1 = r1, min(r1 + b, n)
            if q0 >= n:
                break
            O = A[q0:q1, r0:r1] @ H  # right-multiplication creates the bulge
            v2, tau2, beta2 = larfg(O[:, 0].copy())
            H2 = np.eye(q1 - q0) - tau2 * np.outer(v2, v2.conj())
            O = H2.conj().T @ O
            O[1:, 0] = 0
            O[0, 0] = beta2
            A[q0:q1, r0:r1] = O
            A[r0:r1, q0:q1] = O.conj().T
            log.append((j, q0, v2, tau2))
            r0, r1, v, tau = q0, q1, v2, tau2
    d = A.diagonal().real.copy()
    e = np.array([A[i + 1, i] for i in range(n - 1)])
    assert np.abs(e.imag).max() < 1e-12 * max(1.0, np.abs(e).max())
    off = A - np.diag(A.diagonal()) - np.diag(e, -1) - np.diag(e.conj(), 1)
    assert np.abs(off).max() < 1e-10 * np.abs(Ab).max(), np.abs(off).max()
    return d, e.real, log


def apply_q2(log, x, adjoint):
    x = x.copy()
    for _, r0, v, tau in log if adjoint else reversed(log):
        seg = x[r0 : r0 + len(v)]
        t = (np.conj(tau) if adjoint else tau) * (v.conj() @ seg)
        seg -= t * v
    return x


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, n + 10)) + 1j * rng.standard_normal((n, n + 10))
    G = X @ X.conj().T
    rhs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    Ab, refl = stage1(G, b)
    low = np.tril(Ab, -b - 1)
    assert np.abs(low).max() < 1e-10 * np.abs(G).max(), "stage 1 left entries below the band"
    d, e, log = stage2(Ab, b)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    lam, S = np.linalg.eigh(T)
    ref = np.linalg.eigvalsh(G)
    print("eigenvalues rel err", np.abs(lam - ref).max() / np.abs(ref).max())
    z = apply_q2(log, apply_q1(refl, rhs, True), True)
    y = S @ ((S.T @ z) / lam)
    x = apply_q1(refl, apply_q2(log, y, False), False)
    xr = np.linalg.solve(G, rhs)
    print("solve rel err", np.abs(x - xr).max() / np.abs(xr).max(), "steps logged", len(log), "~ n^2/2b =", n * n // (2 * b))


if __name__ == "__main__":
    main()
