#!/usr/bin/env python
"""NumPy twin of k_pchol (lowrank_chol.h): 8 candidate pivots per step, in-block Cholesky in selection order with the 1 %
put-back rule, columns formed from G every step.  Accuracy of the eigenvalues of L^H L against eigvalsh(G)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def blocked(G, tol, nb=8, putback=0.01):
    n = G.shape[0]
    d = np.real(np.diag(G)).copy()
    lb = d.max()
    L = np.zeros((n, 0), dtype=G.dtype)
    while True:
        if np.maximum(d, 0).sum() <= tol * lb:
            break
        P = [int(i) for i in np.argsort(-d)[:nb] if d[i] > 0]
        C = G[:, P] - L @ L[P].conj().T
        blk = C[P].copy()
        acc, Lb = [], np.zeros((len(P), len(P)), dtype=G.dtype)
        a = blk.copy()
        for c in range(len(P)):
            pv = a[c, c].real
            if pv > putback * d[P[c]] and pv > 0:
                l = a[:, c] / np.sqrt(pv)
                l[:c] = 0
                Lb[:, c] = l
                a -= np.outer(l, l.conj())
                acc.append(c)
        Y = np.zeros((n, len(acc)), dtype=G.dtype)
        Cw = C.copy()
        Yfull = np.zeros((n, len(P)), dtype=G.dtype)
        for c in range(len(P)):
            if c in acc:
                y = Cw[:, c].copy()
                for c2 in range(c):
                    y -= Yfull[:, c2] * np.conj(Lb[c, c2])
                Yfull[:, c] = y / Lb[c, c].real
        Y = Yfull[:, acc]
        d -= (np.abs(Y) ** 2).sum(axis=1)
        for c in acc:
            d[P[c]] = 0.0
        L = np.concatenate([L, Y], axis=1)
        if not acc:
            break
    return L


def main():
    from draco_amd import workloads as wl
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context
    Context.get()
    c = wl.CONFIGS[3]
    tel = TransitTelescope(wl.frequencies(c["nfreq"])[:1], lmax=c["lmax"], ncyl=c["ncyl"], nfeed_cyl=c["nfeed_cyl"])
    bt = BeamScreenProvider(tel, seed=3003)
    rng = np.random.default_rng(5)
    for m in (15, 86):
        B = np.asarray(bt.beam_m(m, fi=0))[..., m:].reshape(2 * tel.npairs, -1)
        ni = rng.uniform(0.5, 1.5, B.shape[0]) * 20.0 * 1024
        ni[rng.uniform(size=ni.size) < 0.02] = 0.0
        DB = np.sqrt(ni)[:, None] * B
        G = DB @ DB.conj().T
        sv = np.linalg.svd(DB, compute_uv=False)
        lam = sv ** 2
        kept = int(np.sum(sv > 1e-3 * sv[0]))
        for tol in (1e-13, 1e-14, 1e-15):
            L = blocked(G, tol)
            mu = np.linalg.eigvalsh(L.conj().T @ L)[::-1]
            E = np.linalg.norm(G - L @ L.conj().T, 2) / lam[0]
            print(json.dumps({"m": m, "tol": tol, "columns": L.shape[1], "kept": kept, "resid_norm_over_lam_max": E,
                              "rel_err_smallest_kept_sigma": float(abs(np.sqrt(mu[kept - 1]) / sv[kept - 1] - 1)),
                              "worst_rel_err_kept_sigma": float(np.abs(np.sqrt(mu[:kept]) / sv[:kept] - 1).max()),
                              "eigvalsh_G_itself": float(abs(np.sqrt(np.linalg.eigvalsh(G)[::-1][kept - 1]) / sv[kept - 1] - 1))}), flush=True)


if __name__ == "__main__":
    main()
