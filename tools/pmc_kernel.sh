#!/bin/bash
# PMC counters of one kernel of the 4-frequency ML bench: bash tools/pmc_kernel.sh <kernel substring> <out file> COUNTER...
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
KERN=$1; OUT=$2; shift 2
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_ml
timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_ml -o ml -- python3 "$REPO/bench.py" --maker ${MAKER:-ml} --steps 1 --warmup 0 --freqs 4 --no-cpu-baseline > /tmp/pmc.log 2>&1 || { tail -5 /tmp/pmc.log; exit 1; }
cd "$REPO"
python - "$KERN" $(find /tmp/pmc_ml -name '*counter_collection.csv' | head -1) > "gpurun_out/$OUT" <<'PY'
import csv, sys, collections
kern, path = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for row in csv.DictReader(open(path)):
    if kern in row["Kernel_Name"]:
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:32s} launches {n[k]:4d}  total {tot[k]:.4g}  per launch {tot[k]/n[k]:.4g}")
PY
cat "gpurun_out/$OUT"
