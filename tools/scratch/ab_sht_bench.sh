python -m pytest tests/test_gpu_sht.py -x -q 2>&1 | tail -2
python - <<'PY'
import numpy as np, torch, sys
sys.path.insert(0,'.')
from draco_amd import _lib
from draco_amd.device import Context, ptr
ctx=Context.get()
gen=torch.Generator(device=ctx.device).manual_seed(1)
alm=torch.randn((5,4,201,201),dtype=torch.complex128,device=ctx.device,generator=gen)
outs=[]
for v in (64, 64+1024):
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle,b"sht_variant",v))
    m=ctx.empty((5,4,12*128*128),np.float64)
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle,ptr(alm),5,4,200,200,128,ptr(m)))
    ctx.sync(); outs.append(m.cpu().numpy())
print("staged == direct:", np.array_equal(outs[0],outs[1]))
PY
for v in 64 1088 64 1088; do
  DMM_OPTS=sht_variant=$v python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab.json
  python - $v <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print("sht_variant", sys.argv[1], "value %.1f frac %.3f alone %.3f ms/day %.1f"%(d["value"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["ms_per_step"]), d["stages_alone_ms"])
PY
done
