#!/bin/bash
# kernel trace of one forward+backward solve at 150^3 (nrhs = $1, default 1)
set -e
NR=${1:-1}
PROB=${2:-lap3d}
GRID=${3:-150}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_solve_${2:-lap3d}_$NR
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O -o s -- python3 $R/bench.py --problem $PROB --grid $GRID --steps 1 --warmup 0 --no-cpu --solve --nrhs $NR > $O/bench.json 2> $O/err.txt
python3 - $O/s_results.db <<'PY'
# per kernel: launches, total, mean, and the same by grid size; then the gaps between consecutive solve kernels
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
names = [n for (n,) in db.execute("select distinct name from kernels where name like '%k_solve%'")]
tot_all = 0
print("(the bench runs every sweep twice: once untimed, once timed)")
for n in sorted(names):
    rows = db.execute("select duration, grid_x, workgroup_x from kernels where name = ?", (n,)).fetchall()
    tot = sum(r[0] for r in rows); tot_all += tot
    print(f"{n.split('(')[0][:64]:64s} {len(rows):6d} launches {tot/1e6:9.2f} ms  mean {tot/len(rows)/1e3:8.1f} us")
    b = {}
    for d, g, w in rows:
        wg = g // max(w, 1)
        key = 1 if wg <= 1 else 64 if wg <= 64 else 1024 if wg <= 1024 else 16384 if wg <= 16384 else 10**6
        b.setdefault(key, [0, 0]); b[key][0] += 1; b[key][1] += d
    for k in sorted(b):
        print(f"        workgroups <= {k:7d}: {b[k][0]:6d} launches {b[k][1]/1e6:9.2f} ms  mean {b[k][1]/b[k][0]/1e3:8.1f} us")
rows = db.execute("select start, end from kernels where name like '%k_solve%' order by start").fetchall()
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
gaps = [g for g in gaps if g < 1e6]
print(f"solve kernels: {len(rows)}, busy {tot_all/1e6:.2f} ms, gaps {sum(gaps)/1e6:.2f} ms (mean {sum(gaps)/max(len(gaps),1)/1e3:.2f} us)")
PY
cat $O/bench.json
