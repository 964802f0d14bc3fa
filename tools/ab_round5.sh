#!/bin/bash
# The A/B runs behind profiles/r05_cu_split_ab.txt (one MI355X box; each line = one bench.py run):
#   bash tools/ab_round5.sh headline    # the headline day with the side stream / the solve kernel on CU subsets, the Legendre
#                                       # synthesis forms (sht_variant 64 = first MFMA form, 128 / 0 = pipelined, 4 / 8 frequencies
#                                       # per block) and `sht_grid` walking blocks
#   bash tools/ab_round5.sh ml          # the ML day with the side streams on every K-th CU (DMM_ML_CU_SPLIT)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"; mkdir -p gpurun_out
show() { python3 - "$@" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(" ".join(sys.argv[1:]), "value %.1f frac %.3f alone %.3f ms/day %.1f T_sht %.1f" % (d["value"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["ms_per_step"], d["stages_alone_ms"]["T_sht"]))
PY
}
if [ "$1" = "ml" ]; then
  for k in 0 8 6 0; do
    DMM_ML_CU_SPLIT=$k python3 bench.py --maker ml --freqs 32 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab.json
    python3 - $k <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json")); x = json.load(open("bench_extra.json")); k = x["kernel_classes_ms_per_day_timed"]
print("ml_cu_split", sys.argv[1], "day_s %.2f gram_frac %.3f" % (d["ms_per_step"] / 1e3, d["roofline"]["frac"]), {c: round(k[c]["ms"]) for c in ("gram", "band", "chase", "ql")})
PY
  done
  exit 0
fi
# side_cu_every dirty_cu_split sht_variant sht_grid
for cfg in "0 0 64 0" "0 0 128 0" "0 0 0 0" "8 0 0 0" "8 0 64 0" "4 0 128 0" "6 0 0 0" "8 8 0 0" "8 8 64 0" "16 16 0 0" "12 12 0 0" "0 0 64 32" "0 0 64 64" "0 0 64 128"; do
  set -- $cfg
  DRACO_AMD_SIDE_CU_EVERY=$1 DMM_OPTS=sht_variant=$3,dirty_cu_split=$2,sht_grid=$4 python3 bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab.json
  show side_cu_every $1 dirty_cu_split $2 sht_variant $3 sht_grid $4
done
