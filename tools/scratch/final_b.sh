python -m pytest tests/test_gpu_edges.py tests/test_gpu_sht.py tests/test_gpu_bench_ranks.py -q 2>&1 | tail -2
bash tools/prof_round.sh r05 > gpurun_out/r05_prof_round.log 2>&1; tail -12 gpurun_out/r05_prof_round.log
PMC=1 bash tools/prof_sht_ab.sh 0 r05new 3 > /dev/null 2>&1; PMC=1 bash tools/prof_sht_ab.sh 64 r05first 3 > /dev/null 2>&1
head -12 gpurun_out/sht_stats_r05new.txt
python bench.py --maker ml --steps 1 --warmup 1 > gpurun_out/r05_bench_ml_cfg3_day.out 2>/dev/null; cp bench_extra.json gpurun_out/r05_bench_ml_cfg3_day.json; tail -c 600 gpurun_out/r05_bench_ml_cfg3_day.out
python bench.py --maker wiener --steps 1 --warmup 1 > gpurun_out/r05_bench_wiener_cfg3_day.out 2>/dev/null; cp bench_extra.json gpurun_out/r05_bench_wiener_cfg3_day.json; tail -c 400 gpurun_out/r05_bench_wiener_cfg3_day.out
python tools/simulate_day.py > gpurun_out/r05_simulate_day.json 2>/dev/null; tail -c 600 gpurun_out/r05_simulate_day.json
