#!/bin/bash
# kernel-trace profile of the 16-frequency ML bench; summary into gpurun_out/$1
# usage (on the GPU box, from the repo root): bash tools/prof_ml.sh ml_stats.txt [extra bench args]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift || true
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ml
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_ml -o ml -- python3 "$REPO/bench.py" --maker ${MAKER:-ml} --steps 1 --warmup 0 --freqs 16 --no-cpu-baseline "$@" > /tmp/ml.log 2>&1 || { tail -5 /tmp/ml.log; exit 1; }
cd "$REPO"
python tools/prof_db_summary.py "$(find /tmp/prof_ml -name '*.db' | head -1)" > "gpurun_out/$OUT"
head -24 "gpurun_out/$OUT"; if [ -n "$LAUNCHES" ]; then python tools/prof_db_launches.py "$(find /tmp/prof_ml -name "*.db" | head -1)" "$LAUNCHES" | tee "gpurun_out/launches_$OUT"; fi
