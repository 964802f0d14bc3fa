#!/bin/bash
# per-kernel durations (and, with PMC=1, counters of the Legendre kernels) of the SHT alone: bash tools/prof_sht_ab.sh <sht_variant> <tag> [niter]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
V=${1:-0}; TAG=${2:-v$V}; NITER=${3:-0}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sht_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sht_$TAG -o s -- python3 "$REPO/tools/sht_prof.py" --nfreq 32 --reps 3 --niter $NITER --variant $V > /tmp/sht_$TAG.log 2>&1 || { tail -5 /tmp/sht_$TAG.log; exit 1; }
cd "$REPO"
python tools/prof_db_summary.py "$(find /tmp/prof_sht_$TAG -name '*.db' | head -1)" 14 > gpurun_out/sht_stats_$TAG.txt
tail -1 /tmp/sht_$TAG.log >> gpurun_out/sht_stats_$TAG.txt
if [ -n "$PMC" ]; then
  for SET in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVES SQ_INSTS_BRANCH"; do
    cd /tmp; rm -rf /tmp/pmc_sht
    timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_sht -o p -- python3 "$REPO/tools/sht_prof.py" --nfreq 8 --reps 1 --niter $NITER --variant $V > /tmp/pmc_sht.log 2>&1 || { tail -5 /tmp/pmc_sht.log; exit 1; }
    cd "$REPO"
    python - $(find /tmp/pmc_sht -name '*counter_collection.csv' | head -1) >> gpurun_out/sht_stats_$TAG.txt <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    if "k_leg_" in row["Kernel_Name"]:
        key = (row["Kernel_Name"].split("(")[0][-28:], row["Counter_Name"])
        tot[key] += float(row["Counter_Value"]); n[key] += 1
for k in sorted(tot): print(f"{k[0]:30s} {k[1]:28s} launches {n[k]:3d}  per launch {tot[k]/n[k]:.5g}")
PY
  done
fi
cat gpurun_out/sht_stats_$TAG.txt
