#!/bin/bash
# The round's records on the final tree, in one GPU call (≈ 8 min): GPU tests, the driver's bench command twice (the second run
# shows what the CPU baseline does from run to run), the SHT alone with counters, the SimulateSidereal day.
#   bash tools/round_records.sh r05      -> gpurun_out/r05_*; copy what is to be judged into profiles/
TAG=${1:-rXX}
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputests.log 2>&1; tail -2 gpurun_out/${TAG}_gputests.log
for run in "" _run2; do
  ( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_cfg3$run.out 2> gpurun_out/${TAG}_bench_cfg3$run.err ) 2> gpurun_out/${TAG}_bench_cfg3$run.time
  tail -c 4096 gpurun_out/${TAG}_bench_cfg3$run.out | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('parsed', len(json.dumps(d)), d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['values'])"
  cp bench_extra.json gpurun_out/${TAG}_bench_cfg3${run}_extra.json
done
PMC=1 bash tools/prof_sht_ab.sh 0 ${TAG}new 3 > /dev/null 2>&1; head -8 gpurun_out/sht_stats_${TAG}new.txt
python3 tools/simulate_day.py > gpurun_out/${TAG}_simulate_day.json 2>/dev/null; tail -c 300 gpurun_out/${TAG}_simulate_day.json
