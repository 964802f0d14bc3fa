# usage: bash tools/r6_prof.sh "ENV1=..,ENV2=.." ... (one rocprof run per argument; prints the stage-1 kernels)
set -e
i=0
for cfg in "$@"; do
  i=$((i+1))
  (
    IFS=','; for kv in $cfg; do export "$kv"; done
    bash tools/prof_ml.sh r6_prof_$i.txt > /dev/null
  )
  echo "== $cfg"
  grep -E "all kernels|k_sb_|k_nt<0>" gpurun_out/r6_prof_$i.txt
  python - <<PY
import re
t=0
for l in open('gpurun_out/r6_prof_$i.txt'):
    if 'k_sb_sweep' in l or 'k_sb_panel' in l or 'k_sb_zero' in l: t+=float(l.split()[-3])
print('stage-1 kernels total ms', round(t,1))
PY
done
