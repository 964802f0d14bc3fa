python -m pytest tests -m gpu -q > gpurun_out/r05_gputests.log 2>&1; tail -2 gpurun_out/r05_gputests.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_cfg3.out 2> gpurun_out/r05_bench_cfg3.err ) 2> gpurun_out/r05_bench_cfg3.time
tail -c 4096 gpurun_out/r05_bench_cfg3.out | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('parsed', len(json.dumps(d)), d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['values'])"
cp bench_extra.json gpurun_out/r05_bench_cfg3_extra.json
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_cfg3_run2.out 2>/dev/null ) 2> gpurun_out/r05_bench_cfg3_run2.time
tail -n 1 gpurun_out/r05_bench_cfg3_run2.out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('run2', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['values'])"
cat gpurun_out/r05_bench_cfg3.time
PMC=1 bash tools/prof_sht_ab.sh 0 r05new 3 > /dev/null 2>&1; head -8 gpurun_out/sht_stats_r05new.txt; grep alm2map gpurun_out/sht_stats_r05new.txt
python tools/simulate_day.py > gpurun_out/r05_simulate_day.json 2>/dev/null; tail -c 300 gpurun_out/r05_simulate_day.json
