for cfg in "0 0 64" "8 8 0" "8 8 128" "8 8 64" "16 16 0" "16 16 128" "12 12 0"; do
  set -- $cfg
  DRACO_AMD_SIDE_CU_EVERY=$1 DMM_OPTS=sht_variant=$3,dirty_cu_split=$2 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab.json
  python - $1 $2 $3 <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print("side_cu_every", sys.argv[1], "dirty_cu_split", sys.argv[2], "sht_variant", sys.argv[3], "value %.1f frac %.3f alone %.3f ms/day %.1f"%(d["value"], d["roofline"]["frac"], d["roofline"]["alone"]["frac"], d["ms_per_step"]))
PY
done
