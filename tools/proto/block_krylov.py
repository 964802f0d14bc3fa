#!/usr/bin/env python
"""NumPy model of a block-Krylov (block Lanczos with full reorthogonalisation + explicit Rayleigh-Ritz) front stage for the ML
eigen path: for a numerically low-rank Gram matrix G = C C^H (C = D B, n x K) find an orthonormal Q (n x 64 J) whose span holds
the eigenvectors above pinv_svd's cut, T = Q^H G Q, and solve on T.  Start block = the first 64 unit vectors (W_0 = the first
64 columns of G: free).  Orthonormalisation of a residual block by Cholesky-QR twice with a tiny shift -- GEMMs and 64 x 64
factorisations only, which is what the GPU has fast kernels for.

    python tools/proto/block_krylov.py [ntrial]

Prints, per synthetic tile: numerical rank, steps, kept rank (oracle SVD of C vs this route), error of the telescope-side
solution and of a = C^H-side solution against the oracle's SVD solution.
"""
import sys

import numpy as np

RCOND, ACOND = 1e-3, 1e-4


def make_tile(rng, n=758, K=1500, decade_cols=12.0, floor=1e-10, scale=0.1, zero_frac=0.02):
    """C = D B with B = U diag(s) V^H, s falling a decade per `decade_cols` columns to a floor (relative)."""
    r = min(n, K)
    U, _ = np.linalg.qr(rng.standard_normal((n, r)) + 1j * rng.standard_normal((n, r)))
    V, _ = np.linalg.qr(rng.standard_normal((K, r)) + 1j * rng.standard_normal((K, r)))
    s = scale * np.maximum(10.0 ** (-np.arange(r) / decade_cols), floor)
    B = (U * s) @ V.conj().T
    w = (rng.uniform(0.5, 1.5, n)) * 20.0 * 1024
    w[rng.uniform(size=n) < zero_frac] = 0.0
    return B, w


def oracle(B, w, v):
    D = np.sqrt(w)
    C = B * D[:, None]
    u, sig, vh = np.linalg.svd(C, full_matrices=False)
    rank = int(np.sum((sig > RCOND * sig.max()) & (sig > ACOND)))
    a = (vh[:rank].conj().T * (1.0 / sig[:rank])) @ (u[:, :rank].conj().T @ (D * v))
    return a, rank, sig


def cholqr2(W, passes=3):
    """SVQB (Stathopoulos & Wu): W <- W Z max(Theta, eps theta_max)^-1/2 with W^H W = Z Theta Z^H, repeated -- each pass brings
    the condition number from kappa to ~ max(1, kappa sqrt(eps)); directions that are pure rounding noise come out as SOME unit
    vectors, which a Rayleigh-Ritz stage tolerates.  On the GPU: a 64 x 64 Hermitian Jacobi (exists) and two skinny GEMMs."""
    for it in range(passes):
        S = W.conj().T @ W
        S = 0.5 * (S + S.conj().T)
        th, Z = np.linalg.eigh(S)
        th = np.maximum(th, 2e-16 * th.max())
        W = W @ (Z / np.sqrt(th))
    return W


def block_krylov(G, b=64, tol=1e-13, jmax=12):
    n = G.shape[0]
    Q = np.zeros((n, 0), dtype=complex)
    d = np.diag(G).real
    # start block: the unit vectors of the b largest diagonal entries (never a zero-weight row, whose column of G is empty)
    Qj = np.zeros((n, b), dtype=complex)
    Qj[np.argsort(-d)[:b], np.arange(b)] = 1.0
    lb = d.max()
    steps = 0
    while True:
        Q = np.concatenate([Q, Qj], axis=1)
        steps += 1
        W = G @ Qj
        for _ in range(2):
            W = W - Q @ (Q.conj().T @ W)
        res = np.linalg.norm(W)
        if res <= tol * lb or steps >= jmax or Q.shape[1] + b > n:
            break
        # orthonormal basis of the residual block: SVQB, then out of span(Q) again (a direction that was rounding noise comes
        # back from the normalisation with components along Q), until it stays out
        Qj = W
        for _ in range(4):
            Qj = cholqr2(Qj)
            c = Q.conj().T @ Qj
            if np.abs(c).max() < 1e-14:
                break
            Qj = Qj - Q @ c
    T = Q.conj().T @ (G @ Q)
    T = 0.5 * (T + T.conj().T)
    return Q, T, steps, res / lb


def main():
    ntrial = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rng = np.random.default_rng(7)
    for trial in range(ntrial):
        dec = [8.0, 12.0, 16.0, 20.0, 30.0, 40.0, 55.0, 70.0][trial % 8]
        B, w = make_tile(rng, decade_cols=dec)
        n, K = B.shape
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        a_ref, rank_ref, sig = oracle(B, w, v)
        D = np.sqrt(w)
        C = B * D[:, None]
        G = C @ C.conj().T
        numrank = int(np.sum(sig > 1e-7 * sig.max()))  # lambda > 1e-14 lambda_max
        Q, T, steps, res = block_krylov(G)
        lam, Y = np.linalg.eigh(T)
        s_est = np.sqrt(np.maximum(lam, 0.0))
        keep = (s_est > RCOND * s_est.max()) & (s_est > ACOND)
        rank = int(keep.sum())
        y = Y[:, keep] @ ((Y[:, keep].conj().T @ (Q.conj().T @ (D * v))) / lam[keep])
        x = Q @ y
        a = C.conj().T @ x
        err = np.abs(a - a_ref).max() / np.abs(a_ref).max()
        smin_ref, smin = sig[rank_ref - 1], np.sort(s_est[keep])[0]
        print(f"decade per {dec:4.0f} columns: rank(1e-14) {numrank:3d}  steps {steps} (order {Q.shape[1]}, residual {res:.1e})  kept {rank} (oracle {rank_ref})  "
              f"smallest kept sigma rel err {abs(smin / smin_ref - 1):.1e}  solution err {err:.1e}")


if __name__ == "__main__":
    main()
