# round-6 records that are not part of tools/round_records.sh: the stand-alone sweep probe, one rank's share of cfg 4 / cfg 5,
# the full dense days (256 frequencies) and the interleaved A/B of the stage-1 form
set -e
mkdir -p gpurun_out
timeout -k 10 300 tools/probe/build/sweep_probe 1185 768 > gpurun_out/r06_sweep_probe.txt
echo probe done
timeout -k 10 900 python tools/rank_share.py > gpurun_out/r06_rank_share.json 2> gpurun_out/r06_rank_share.err
echo rank share done
for MK in ml wiener; do
  python bench.py --maker $MK --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_${MK}_cfg3_day.json 2> gpurun_out/r06_bench_${MK}_cfg3_day.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r06_bench_${MK}_cfg3_day.json').read().strip().splitlines()[-1])
print('$MK day ms', d['ms_per_step'], d['roofline'].get('frac'), [(x['kernel'][:40], x.get('ms_per_day'), x.get('frac')) for x in d.get('roofline_secondary') or []])
PY
done
DMM_ML_REDUCE=2 python bench.py --maker ml --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r06_bench_ml_cfg3_day_undeferred.json 2> /dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r06_bench_ml_cfg3_day_undeferred.json').read().strip().splitlines()[-1])
print('ml undeferred day ms', d['ms_per_step'], [(x['kernel'][:40], x.get('ms_per_day'), x.get('frac')) for x in d.get('roofline_secondary') or []])
PY
