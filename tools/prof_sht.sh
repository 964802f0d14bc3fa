set -e
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sht
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sht -o s -- python3 "$REPO/tools/sht_prof.py" --nfreq 32 --reps 3 > /tmp/sht.log 2>&1 || { tail -5 /tmp/sht.log; exit 1; }
cd "$REPO"
python tools/prof_db_summary.py "$(find /tmp/prof_sht -name '*.db' | head -1)" 14
tail -1 /tmp/sht.log
