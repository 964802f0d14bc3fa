for k in 0 8 6 0; do
  DMM_ML_CU_SPLIT=$k python bench.py --maker ml --freqs 32 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/abml.json
  python - $k <<'PY'
import json,sys
d=json.load(open("gpurun_out/abml.json"))
x=json.load(open("gpurun_out/bench_extra.json"))
k=x["kernel_classes_ms_per_day_timed"]
print("ml_cu_split", sys.argv[1], "day_s %.2f gram_frac %.3f"%(d["ms_per_step"]/1e3, d["roofline"]["frac"]), {c:round(k[c]["ms"]) for c in ("gram","band","chase","ql")})
PY
done
