"""probe: ranks of structured tiles for a small telescope (ntel 566, Np 576) -- to choose the parameters of the ADVICE r4 test"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
from draco_amd import _lib
from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
from draco_amd.core import containers
from draco_amd.core.products import BeamScreenProvider, TransitTelescope
from draco_amd.device import Context, ptr
ctx = Context.get()
out = {}
for name, freq, lmax, kw in (("e", 798.0, 512, {"feed_sep": 0.8}), ("f", 798.0, 512, {"feed_sep": 1.0}), ("g", 798.0, 512, {"feed_sep": 1.3}), ("h", 798.0, 512, {"feed_sep": 1.0, "sigma_n": 1.2})):
    tel = TransitTelescope(np.array([freq]), lmax=lmax, ncyl=2, nfeed_cyl=24)
    bt = BeamScreenProvider(tel, seed=3003, **kw)
    gen = torch.Generator(device=ctx.device).manual_seed(3)
    shape = (lmax + 1, 2, 1, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = (torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5) * 20.0 * 1024
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
    mm.attach("vis", mv); mm.attach("vis_weight", mw)
    per_f = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16
    t = MaximumLikelihoodMapMaker(nside=64, pool_bytes=per_f + (1 << 20))
    t.setup(bt)
    diag = torch.full((1, lmax + 1, 4), -1.0, dtype=torch.float64, device=ctx.device)
    _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, ptr(diag)))
    t.make_alm(mm); ctx.sync()
    _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, None))
    r = diag.cpu().numpy()[0, :, 0]
    out[name] = {"npairs": tel.npairs, "ranks_every_16": r[::16].tolist(), "max": float(r.max())}
    print(name, tel.npairs, r.max(), r[::32], flush=True)
    del t
    from draco_amd.analysis import _solve
    _solve.release_pools()
json.dump(out, open("gpurun_out/probe_small_tel.json", "w"))
