#!/usr/bin/env python
"""Would a rank-revealing (diagonally pivoted) Cholesky of the ML Gram matrix do instead of the band reduction's sweeps?
It reads only the pivot columns of G -- but its truncation is FIRST order: G = L L^H + R, R the positive semi-definite
residual, and the eigenvalues of L^H L are those of G - R.  NumPy model on tiles read back from BeamScreenProvider:
for residual tolerances 1e-12 ... 1e-16 of max diag, the number of columns taken and the relative error of the
eigenvalues pinv_svd's cut keeps (lambda > 1e-6 lambda_max, mapmaker.py:296), smallest kept one first.

    python tools/proto/pivoted_cholesky.py            (needs the GPU only to generate the tiles)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def pivoted_cholesky(G, tol):
    n = G.shape[0]
    d = np.real(np.diag(G)).copy()
    L = np.zeros((n, 0), dtype=G.dtype)
    thr = tol * d.max()
    while L.shape[1] < n:
        p = int(np.argmax(d))
        if d[p] <= thr:
            break
        col = G[:, p] - L @ L[p].conj()
        col /= np.sqrt(d[p])
        L = np.concatenate([L, col[:, None]], axis=1)
        d -= np.abs(col) ** 2
        d[p] = 0.0
    return L, float(max(d.max(), 0.0))


def main():
    from draco_amd import workloads as wl
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context

    Context.get()
    c = wl.CONFIGS[3]
    tel = TransitTelescope(wl.frequencies(c["nfreq"])[:: c["nfreq"] // 4][:4], lmax=c["lmax"], ncyl=c["ncyl"], nfeed_cyl=c["nfeed_cyl"])
    bt = BeamScreenProvider(tel, seed=3003)
    rng = np.random.default_rng(5)
    out = []
    for f, m in ((0, 40), (0, 180), (3, 80), (3, 240)):
        B = np.asarray(bt.beam_m(m, fi=f))[..., m:].reshape(2 * tel.npairs, -1)
        ni = rng.uniform(0.5, 1.5, B.shape[0]) * 20.0 * 1024
        ni[rng.uniform(size=ni.size) < 0.02] = 0.0
        DB = np.sqrt(ni)[:, None] * B
        G = DB @ DB.conj().T
        lam = np.linalg.eigvalsh(G)[::-1]
        kept = int(np.sum(lam > 1e-6 * lam[0]))
        rec = {"f": f, "m": m, "kept_rank": kept, "arms": []}
        for tol in (1e-12, 1e-13, 1e-14, 1e-15, 1e-16):
            L, resid = pivoted_cholesky(G, tol)
            mu = np.linalg.eigvalsh(L.conj().T @ L)[::-1]
            k = min(kept, mu.size)
            rel = np.abs(mu[:k] - lam[:k]) / lam[:k]
            rec["arms"].append({"tol_of_max_diag": tol, "columns": int(L.shape[1]), "residual_max_diag_over_lam_max": resid / lam[0],
                                "rel_err_smallest_kept_eigenvalue": float(rel[-1]) if k else None, "worst_rel_err_kept": float(rel.max()) if k else None,
                                "kept_rank_from_L": int(np.sum(mu > 1e-6 * mu[0]))})
        out.append(rec)
        print(json.dumps(rec), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/pivoted_cholesky.json", "w"), indent=1)


if __name__ == "__main__":
    main()
