#!/usr/bin/env python
"""Every launch of the kernels whose name contains a substring, from a rocprofv3 rocpd database: start (ms from the
first launch listed), duration, grid and workgroup size.

    python tools/prof_db_launches.py <.db> <substring>
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
if not cols:
    print("views/tables:", [r[0] for r in db.execute("select name from sqlite_master")])
    sys.exit(1)
pick = [c for c in ("name", "start", "end", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "stream_id", "queue_id") if c in cols]
rows = [r for r in db.execute(f"select {', '.join(pick)} from kernels order by start") if sub in r[0]]
if not rows:
    print("columns:", cols)
    sys.exit(0)
t0 = rows[0][1]
print("# " + " ".join(pick[1:]))
for r in rows:
    d = dict(zip(pick, r))
    rest = " ".join(f"{k}={d[k]}" for k in pick[3:]) + " " + d["name"][:60]
    print(f"{(d['start'] - t0) / 1e6:10.3f} ms  {(d['end'] - d['start']) / 1e3:10.1f} us  {rest}")
