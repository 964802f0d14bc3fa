# the other standing records of a round, on the final tree: the same path at the other single-GPU sizes, 25 days back to back, D days per pass
set -e
mkdir -p gpurun_out
python bench.py --config 2 --steps 20 --warmup 3 --no-extra > gpurun_out/r06_bench_cfg2.out 2> /dev/null; tail -n 1 gpurun_out/r06_bench_cfg2.out > gpurun_out/r06_bench_cfg2.json
python bench.py --config 4 --steps 1 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r06_bench_cfg4.out 2> /dev/null; tail -n 1 gpurun_out/r06_bench_cfg4.out > gpurun_out/r06_bench_cfg4_1step.json
python tools/soak_days.py > gpurun_out/r06_soak_25days.json 2> /dev/null
python - <<'PY'
import json
for f in ("r06_bench_cfg2.json", "r06_bench_cfg4_1step.json"):
    d = json.loads(open("gpurun_out/" + f).read())
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], (d["roofline"].get("alone") or {}).get("frac"))
d = json.loads(open("gpurun_out/r06_soak_25days.json").read().strip().splitlines()[-1])
print("soak", {k: d[k] for k in list(d)[:8]})
PY
