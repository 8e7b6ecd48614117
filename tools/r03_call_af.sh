#!/bin/bash
# Round 3, GPU call AF: kernel trace of the sweeps with the group inverses on (durations of k_solve_group_inv by grid)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03af_${1:-lap3d}; rm -rf $O; mkdir -p $O
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O -o s -- python3 $R/bench.py --problem ${1:-lap3d} --grid ${2:-150} --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/bench.json 2> $O/err.txt || { tail $O/err.txt; exit 1; }
python3 - $O/s_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for pat in ['%k_solve_group_inv%', '%k_solve_group_mma%', '%k_ginv%']:
    rows = db.execute("select duration, grid_x, workgroup_x, grid_y, name from kernels where name like ? order by start", (pat,)).fetchall()
    b = {}
    for d, g, w, gy, n in rows:
        wg = g // w
        k = (n.split('(')[0][-44:], 1 if wg <= 1 else 4 if wg <= 4 else 16 if wg <= 16 else 64 if wg <= 64 else 1024 if wg <= 1024 else 10**6)
        b.setdefault(k, []).append(d)
    for k in sorted(b):
        v = sorted(b[k]); print(k, len(v), "min %.1f med %.1f max %.1f us sum %.2f ms" % (v[0]/1e3, v[len(v)//2]/1e3, v[-1]/1e3, sum(v)/1e6))
PY
