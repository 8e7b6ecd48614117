#!/bin/bash
# kernel trace of one forward+backward solve at 150^3 (nrhs = $1, default 1)
set -e
NR=${1:-1}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_solve_$NR
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O -o s -- python3 $R/bench.py --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs $NR > $O/bench.json 2> $O/err.txt
python3 - $O/s_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for kn in ("k_solve_groupILb0", "k_solve_groupILb1", "k_solve_fwd_update", "k_solve_bwd_update", "k_solve_diag", "k_solve_fwd_mma", "k_solve_bwd_mma", "k_solve_transpose"):
    rows = db.execute("select duration, grid_x, workgroup_x from kernels where name like ? order by duration desc", (f"%{kn}%",)).fetchall()
    tot = sum(r[0] for r in rows)
    print(f"{kn}: {len(rows)} launches, total {tot/1e6:.2f} ms")
    for d, g, w in rows[:8]:
        print(f"   {d/1e3:9.1f} us  workgroups {g//max(w,1)}")
    if not rows:
        continue
    small = [r for r in rows if r[1] // max(r[2], 1) <= 64]
    print(f"   launches with <= 64 workgroups: {len(small)}, total {sum(r[0] for r in small)/1e6:.2f} ms, mean {sum(r[0] for r in small)/max(len(small),1)/1e3:.1f} us")
PY
python3 - $O/s_results.db <<'PY'
# gaps between consecutive solve kernels (launch latency of the dependent chain)
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels where name like '%k_solve%' order by start").fetchall()
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
gaps = [g for g in gaps if g < 1e6]
busy = sum(r[1] - r[0] for r in rows)
print(f"solve kernels: {len(rows)}, busy {busy/1e6:.2f} ms, gaps {sum(gaps)/1e6:.2f} ms (mean {sum(gaps)/max(len(gaps),1)/1e3:.2f} us)")
PY
