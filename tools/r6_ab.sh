# whole-pass A/B of "ml_reduce" (stage 1 of the two-stage reduction): 32-frequency structured cfg-3 ML day, interleaved, twice
set -e
mkdir -p gpurun_out
for rep in 1 2; do
for r in 2 0; do
DMM_ML_REDUCE=$r python bench.py --maker ml --freqs 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6_ml32_red${r}_$rep.json 2> gpurun_out/r6_ml32_red${r}_$rep.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r6_ml32_red${r}_$rep.json').read().strip().splitlines()[-1])
print('ml_reduce=$r rep $rep: day', d['ms_per_step'], 'ms;', [(x['kernel'][26:52], x.get('ms_per_day'), x.get('frac')) for x in d['roofline_secondary']])
PY
done
done
