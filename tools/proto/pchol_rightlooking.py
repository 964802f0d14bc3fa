#!/usr/bin/env python
"""Would an UPDATING (right-looking) blocked pivoted Cholesky keep the accuracy the lazy one loses?  NumPy model: blocks
of `bs` columns; inside a block the columns are formed lazily from the CURRENT residual matrix S (at most bs - 8 columns
of cancellation, against a matrix whose size is the residual's), after the block S -= Y Y^H (one read-and-write pass
over the matrix per bs columns: bs / 8 times fewer than the band reduction's sweeps).  Accuracy of the eigenvalues of
L^H L at pinv_svd's cut against the SVD of D B (mapmaker.py:287-300)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def right_looking(G, tol, bs=32, nb=8, putback=0.01):
    n = G.shape[0]
    S = G.copy()
    d = np.real(np.diag(S)).copy()
    lb = d.max()
    L = np.zeros((n, 0), dtype=G.dtype)
    done = False
    while not done:
        Yb = np.zeros((n, 0), dtype=G.dtype)
        while Yb.shape[1] + nb <= bs:
            if np.maximum(d, 0).sum() <= tol * lb:
                done = True
                break
            P = [int(i) for i in np.argsort(-d)[:nb] if d[i] > 0]
            C = S[:, P] - Yb @ Yb[P].conj().T
            a = C[P].copy()
            Lb = np.zeros((len(P), len(P)), dtype=G.dtype)
            acc = []
            for c in range(len(P)):
                pv = a[c, c].real
                if pv > putback * d[P[c]] and pv > 0:
                    l = a[:, c] / np.sqrt(pv)
                    l[:c] = 0
                    Lb[:, c] = l
                    a -= np.outer(l, l.conj())
                    acc.append(c)
            Yf = np.zeros((n, len(P)), dtype=G.dtype)
            for c in acc:
                y = C[:, c].copy()
                for c2 in range(c):
                    y -= Yf[:, c2] * np.conj(Lb[c, c2])
                Yf[:, c] = y / Lb[c, c].real
            Y = Yf[:, acc]
            d -= (np.abs(Y) ** 2).sum(axis=1)
            for c in acc:
                d[P[c]] = 0.0
            Yb = np.concatenate([Yb, Y], axis=1)
            if not acc:
                done = True
                break
        S -= Yb @ Yb.conj().T
        L = np.concatenate([L, Yb], axis=1)
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
    out = []
    for m in (15, 86, 113, 24):
        B = np.asarray(bt.beam_m(m, fi=0))[..., m:].reshape(2 * tel.npairs, -1)
        ni = rng.uniform(0.5, 1.5, B.shape[0]) * 20.0 * 1024
        ni[rng.uniform(size=ni.size) < 0.02] = 0.0
        DB = np.sqrt(ni)[:, None] * B
        G = DB @ DB.conj().T
        sv = np.linalg.svd(DB, compute_uv=False)
        kept = int(np.sum(sv > 1e-3 * sv[0]))
        for bs in (8, 32, 64):
            for tol in (1e-13, 1e-14):
                L = right_looking(G, tol, bs=bs)
                mu = np.linalg.eigvalsh(L.conj().T @ L)[::-1]
                rec = {"m": m, "block": bs, "tol": tol, "columns": L.shape[1], "kept": kept,
                       "resid_norm_over_lam_max": float(np.linalg.norm(G - L @ L.conj().T, 2) / sv[0] ** 2),
                       "worst_rel_err_kept_sigma": float(np.abs(np.sqrt(np.maximum(mu[:kept], 0)) / sv[:kept] - 1).max())}
                out.append(rec)
                print(json.dumps(rec), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/pchol_rightlooking.json", "w"), indent=1)


if __name__ == "__main__":
    main()
