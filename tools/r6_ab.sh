set -e
mkdir -p gpurun_out
for r in 0 3 2; do
DMM_ML_REDUCE=$r python bench.py --maker ml --freqs 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6_ml32_red$r.json 2> gpurun_out/r6_ml32_red$r.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r6_ml32_red$r.json').read().strip().splitlines()[-1])
print('ml_reduce=$r', d['ms_per_step'], [(x['kernel'][:40], x.get('ms_per_day'), x.get('frac')) for x in d['roofline_secondary']])
PY
done
