#!/usr/bin/env python
"""`dmm_mfft_pack` (sidereal-time -> m FFT + +/-m pack) at four shapes: nra 1024 / 2048, complex128 / complex64 output
(the `HybridVisStream` form): HIP-event time and fraction of the HBM peak on the algorithmic bytes.  The output's
(m, +/-) slots receive rows-per-block x element-size bytes: 128 / 64 bytes at nra 1024, 64 / 32 at nra 2048.

    python tools/mfft_shapes.py
"""
import os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, json
from draco_amd import _lib
from draco_amd.device import Context, ptr
ctx=Context.get()
for nra,nrow,out in ((2048,65536,_lib.DMM_C64),(2048,65536,_lib.DMM_C128),(1024,97024,_lib.DMM_C128),(1024,97024,_lib.DMM_C64)):
    mmax=nra//2
    gen=torch.Generator(device=ctx.device).manual_seed(1)
    vis=torch.randn((nrow,nra),dtype=torch.complex64,device=ctx.device,generator=gen)
    es=8 if out==_lib.DMM_C64 else 16
    mv=torch.empty((mmax+1,2,nrow),dtype=torch.complex64 if es==8 else torch.complex128,device=ctx.device)
    f=lambda: _lib.check(_lib.lib.dmm_mfft_pack(ctx.handle,ptr(vis),nrow,nra,ptr(mv),mmax,out,None))
    f(); ctx.sync(); best=1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(10): f()
        best=min(best,ctx.timer_stop()/10)
    by=nrow*nra*8+(mmax+1)*2*nrow*es
    print(json.dumps({"nra":nra,"nrow":nrow,"out":"c64" if es==8 else "c128","ms":best,"GBs":by/best/1e6,"frac":by/best/1e6/8000}))
