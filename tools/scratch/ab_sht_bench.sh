python -m pytest tests/test_gpu_sht.py -x -q 2>&1 | tail -2
for v in 64 320 576 64; do
  DMM_OPTS=sht_variant=$v python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab.json
  python - $v <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print("sht_variant", sys.argv[1], "value %.1f frac %.3f alone %.3f ms/day %.1f"%(d["value"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["ms_per_step"]), d["stages_alone_ms"])
PY
done
