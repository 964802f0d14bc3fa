set -e
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o kt -- python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > "$REPO/gpurun_out/r03_bench_cfg3_under_rocprof.json" 2> /tmp/pb.err || { tail -5 /tmp/pb.err; exit 1; }
cd "$REPO"
python tools/prof_db_summary.py "$(find /tmp/prof_b -name '*.db' | head -1)" 14 > gpurun_out/r03_bench_kernel_stats_new.txt
head -16 gpurun_out/r03_bench_kernel_stats_new.txt
tail -c 400 gpurun_out/r03_bench_cfg3_under_rocprof.json
