for v in 64 128 0; do
  DMM_OPTS=sht_variant=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab.json
  python - $v <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
x=json.load(open("gpurun_out/bench_extra.json"))
print("sht_variant", sys.argv[1], "headline %.1f"%d["value"], "many_days", d["secondary"]["many_days_D1"], d["secondary"]["many_days_D8"], "c64", d["secondary"]["b_complex64"])
PY
done
