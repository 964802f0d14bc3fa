#!/usr/bin/env python
"""What plain streaming kernels reach on this GPU (torch's own fill / copy / sum / add over 1.6 GB): the yardstick for the
HBM fractions quoted for write-heavy stages.

    python tools/stream_bw.py > gpurun_out/stream_bw.json
"""
import torch, json
dev='cuda:0'
n=1600*1024*1024//8
a=torch.empty(n,dtype=torch.float64,device=dev); b=torch.empty(n,dtype=torch.float64,device=dev)
def t(fn,reps=10):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
out={}
ms=t(lambda: a.fill_(1.0)); out['fill_1.6GB_write_only']={'ms':ms,'TBs':n*8/ms/1e9}
ms=t(lambda: b.copy_(a)); out['copy_1.6GB_read+write']={'ms':ms,'TBs':2*n*8/ms/1e9}
ms=t(lambda: a.sum()); out['sum_1.6GB_read_only']={'ms':ms,'TBs':n*8/ms/1e9}
c=torch.empty(n//2,dtype=torch.float64,device=dev)
ms=t(lambda: torch.add(c,1.0,out=a[:n//2]));  out['read0.8+write0.8']={'ms':ms,'TBs':n*8/ms/1e9}
print(json.dumps(out))
