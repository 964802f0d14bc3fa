python -m pytest tests/test_gpu_sht.py tests/test_gpu_process_many.py tests/test_gpu_process_golden.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab.json"))
print("value %.1f frac %.3f alone %.3f ms/day %.1f"%(d["value"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["ms_per_step"]), d["stages_alone_ms"], {k:d.get(k) for k in ("ml_day_s","wiener_day_s")}, d.get("secondary"))
PY
