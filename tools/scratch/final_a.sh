set -x
python -m pytest tests -m gpu -q > gpurun_out/r05_gputests.log 2>&1; tail -3 gpurun_out/r05_gputests.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_cfg3.out 2> gpurun_out/r05_bench_cfg3.err ) 2> gpurun_out/r05_bench_cfg3.time
tail -c 4096 gpurun_out/r05_bench_cfg3.out | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('parsed', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['spread'])"
cp bench_extra.json gpurun_out/r05_bench_cfg3_extra.json
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_cfg3_run2.out 2>/dev/null ) 2> gpurun_out/r05_bench_cfg3_run2.time
tail -n 1 gpurun_out/r05_bench_cfg3_run2.out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('run2', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['spread'])"
cat gpurun_out/r05_bench_cfg3.time
