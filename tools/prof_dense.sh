#!/bin/bash
# rocprofv3 evidence for the dense days, each in its OWN pass (never --pmc together with a trace domain):
#   kernel statistics of the default ML day and Wiener day (32 of 256 frequencies, band-spread structured tiles), and
#   FETCH_SIZE / WRITE_SIZE of every kernel of an 8-frequency ML day, beside that day's own bench record (its
#   ml_band_bytes / ml_gram_flops counters): the traffic ratio of stage 1 is (2 x FETCH + WRITE) / ml_band_bytes.
#   bash tools/prof_dense.sh r06     -> gpurun_out/r06_*
set -e
TAG=${1:-rXX}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for MK in ml wiener; do
  rm -rf /tmp/prof_$MK
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$MK -o kt -- python3 "$REPO/bench.py" --maker $MK --steps 1 --warmup 1 --freqs 32 --no-cpu-baseline > "$REPO/gpurun_out/${TAG}_bench_${MK}_cfg3_32freq_under_rocprof.json" 2> /tmp/p_$MK.err || { tail -5 /tmp/p_$MK.err; exit 1; }
  python3 "$REPO/tools/prof_db_summary.py" "$(find /tmp/prof_$MK -name '*.db' | head -1)" 30 > "$REPO/gpurun_out/${TAG}_${MK}_cfg3_32freq_kernel_stats.txt"
  echo "$MK kernel stats done"
done
rm -f "$REPO/gpurun_out/${TAG}_ml_cfg3_8freq_pmc.txt"
for C in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/pmcd_$C
  rm -rf $D
  timeout -k 10 900 rocprofv3 --pmc $C --output-format csv -d $D -o p -- python3 "$REPO/bench.py" --maker ml --steps 1 --warmup 0 --freqs 8 --no-cpu-baseline > "$REPO/gpurun_out/${TAG}_bench_ml_cfg3_8freq_under_pmc_$C.json" 2> /tmp/pmcd_$C.log || { tail -5 /tmp/pmcd_$C.log; exit 1; }
  python3 - "$C" "$(find $D -name '*counter_collection.csv' | head -1)" >> "$REPO/gpurun_out/${TAG}_ml_cfg3_8freq_pmc.txt" <<'PY'
import csv, sys, collections, re
cname, path = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for row in csv.DictReader(open(path)):
    if row["Counter_Name"] == cname:
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.search(r"\b(k_[a-z0-9_]+(<[^>(]*>)?)", name)
        k = m.group(1) if m else name[:60]
        tot[k] += float(row["Counter_Value"]); n[k] += 1
for k in sorted(tot, key=lambda k: -tot[k])[:14]:
    print(f"{cname:12s} {k:44s} dispatches {n[k]:5d}  total {tot[k]:.6g}  mean per dispatch {tot[k]/n[k]:.1f}   (KiB as rocprofv3 reports them)")
PY
  echo "$C done"
done
cat "$REPO/gpurun_out/${TAG}_ml_cfg3_8freq_pmc.txt"
head -20 "$REPO/gpurun_out/${TAG}_ml_cfg3_32freq_kernel_stats.txt"
