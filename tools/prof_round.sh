#!/bin/bash
# The round's profile set of the headline command: rocprofv3 kernel statistics of `bench.py --gpus 1 --steps 20 --warmup 5`
# and, in their OWN passes (never together with a trace domain), the PMC counters FETCH_SIZE and WRITE_SIZE of k_dirty.
#   bash tools/prof_round.sh r04
set -e
TAG=${1:-rXX}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b /tmp/pmc_f /tmp/pmc_w
if [ -z "$PMC_ONLY" ]; then
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o kt -- python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > "$REPO/gpurun_out/${TAG}_bench_cfg3_under_rocprof.json" 2> /tmp/pb.err || { tail -5 /tmp/pb.err; exit 1; }
python3 "$REPO/tools/prof_db_summary.py" "$(find /tmp/prof_b -name '*.db' | head -1)" 14 > "$REPO/gpurun_out/${TAG}_bench_kernel_stats.txt"
echo "kernel stats done"
fi
for C in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/pmc_$C
  rm -rf $D
  timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $D -o p -- python3 "$REPO/bench.py" --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extra > /tmp/pmc_$C.log 2>&1 || { tail -5 /tmp/pmc_$C.log; exit 1; }
  python3 - "$C" "$(find $D -name '*counter_collection.csv' | head -1)" >> "$REPO/gpurun_out/${TAG}_bench_pmc.txt" <<'PY'
import csv, sys, collections
cname, path = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for row in csv.DictReader(open(path)):
    if row["Counter_Name"] == cname:
        import re
        m = re.search(r"\b(k_[a-z0-9_]+)", row["Kernel_Name"])
        k = m.group(1) if m else row["Kernel_Name"][:60]
        tot[k] += float(row["Counter_Value"]); n[k] += 1
for k in sorted(tot, key=lambda k: -tot[k])[:6]:
    print(f"{cname:12s} {k:62s} dispatches {n[k]:4d}  mean per dispatch {tot[k]/n[k]:.1f}")
PY
  echo "$C done"
done
cat "$REPO/gpurun_out/${TAG}_bench_pmc.txt"
head -16 "$REPO/gpurun_out/${TAG}_bench_kernel_stats.txt"
