#!/usr/bin/env python
"""How fast does the trailing matrix of the band reduction (block Householder, band 8) of an ML Gram matrix become
rounding dust on the bench's structured tiles?  NumPy model of stage 1 on tiles read back from BeamScreenProvider.

    python tools/proto/trailing_trace.py            (needs the GPU only to generate the tiles)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def band_reduce_trace(G, b=8):
    """trace of the trailing matrix before every panel, and the running max of the reduced part's diagonal."""
    A = G.copy()
    n = A.shape[0]
    tr, lb = [], []
    seen = float(np.max(np.real(np.diag(A))))
    for k in range(0, n - b, b):
        tr.append(float(np.real(np.trace(A[k:, k:]))))
        lb.append(seen)
        X = A[k + b:, k:k + b]
        Q, R = np.linalg.qr(X, mode="complete")
        A[k + b:, :] = Q.conj().T @ A[k + b:, :]
        A[:, k + b:] = A[:, k + b:] @ Q
        seen = max(seen, float(np.max(np.real(np.diag(A[k:k + b, k:k + b])))))
    return np.array(tr), np.array(lb)


def main():
    import torch

    from draco_amd import workloads as wl
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context

    Context.get()
    c = wl.CONFIGS[3]
    tel = TransitTelescope(wl.frequencies(c["nfreq"])[:: c["nfreq"] // 4][:4], lmax=c["lmax"], ncyl=c["ncyl"], nfeed_cyl=c["nfeed_cyl"])
    bt = BeamScreenProvider(tel, seed=3003)
    rng = np.random.default_rng(5)
    out = []
    for f in (0, 3):
        for m in (0, 20, 40, 80, 120, 160, 200, 240, 260, 280, 290):
            B = np.asarray(bt.beam_m(m, fi=f))  # [2, npairs, npol, lmax+1]
            B = B[..., m:].reshape(2 * tel.npairs, -1)
            ni = (rng.uniform(0.5, 1.5, B.shape[0]) * 20.0 * 1024)
            ni[rng.uniform(size=ni.size) < 0.02] = 0.0
            DB = np.sqrt(ni)[:, None] * B
            G = DB @ DB.conj().T
            lam = np.linalg.eigvalsh(G)[::-1]
            lmax_ = lam[0]
            kept = int(np.sum(np.sqrt(np.maximum(lam, 0)) > max(1e-3 * np.sqrt(lmax_), 1e-4)))
            tr, lb = band_reduce_trace(G)
            rec = {"f": f, "m": m, "n": G.shape[0], "kept_rank": kept, "lam_max": lmax_, "maxdiag_over_lam_max": float(np.max(np.real(np.diag(G))) / lmax_),
                   "rank_1e-10": int(np.sum(lam > 1e-10 * lmax_)), "rank_1e-12": int(np.sum(lam > 1e-12 * lmax_)), "rank_1e-14": int(np.sum(lam > 1e-14 * lmax_)),
                   "dust_floor_trace_over_lam_max": float(np.min(np.abs(tr)) / lmax_)}
            for thr in (1e-9, 1e-10, 1e-11, 1e-12):
                hit = np.nonzero(tr <= thr * lb)[0]
                rec[f"first_col_trace<={thr:g}*lb"] = int(8 * hit[0]) if hit.size else None
            rec["lb_over_lam_max_at_panel_4"] = float(lb[min(4, lb.size - 1)] / lmax_)
            out.append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/trailing_trace.json", "w"), indent=1)


if __name__ == "__main__":
    main()
