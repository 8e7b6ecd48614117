#!/bin/bash
# Round 3, GPU call AB: kernel trace of the sweeps with the look-ahead on -- do the far parts (second stream) overlap the
# next group's kernels?  Prints the kernels around the top supernode's chain with their queues.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ab; rm -rf $O; mkdir -p $O
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O -o s -- python3 $R/bench.py --problem lap3d --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 --tune 30=12 > $O/bench.json 2> $O/err.txt || { tail $O/err.txt; exit 1; }
python3 - $O/s_results.db > $O/trace_window.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
q = "queue_id" if "queue_id" in cols else None
s = "stream_id" if "stream_id" in cols else None
sel = "start, end, name, grid_x, workgroup_x" + (", " + q if q else ", 0") + (", " + s if s else ", 0")
rows = db.execute(f"select {sel} from kernels where name like '%k_solve%' order by start").fetchall()
# the last forward sweep = the kernels before the last k_solve_bwd_wide run starts ... simply: find the last fwd_wide kernel and print the 90 kernels before it
idx = [i for i, r in enumerate(rows) if "k_solve_fwd_wide" in r[2]]
last = idx[-1]
t0 = rows[max(0, last - 120)][0]
for r in rows[max(0, last - 120): last + 3]:
    nm = r[2].split("(")[0].replace("mi355::", "").replace("void ", "")[:40]
    print(f"{(r[0]-t0)/1e3:10.1f} us  +{(r[1]-r[0])/1e3:8.1f} us  wgs {r[3]//max(r[4],1):5d}  q {r[5]} s {r[6]}  {nm}")
idx = [i for i, r in enumerate(rows) if "k_solve_bwd_wide" in r[2]]
last = idx[-1]
print("---- backward: the first 100 kernels of the last backward sweep's chain")
# the backward sweep starts with the top supernode: the first bwd_wide of the last sweep = after the last fwd_wide
first = [i for i in idx if i > [j for j, r in enumerate(rows) if 'k_solve_fwd_wide' in r[2]][-1]][0]
t0 = rows[first][0]
for r in rows[first: first + 100]:
    nm = r[2].split("(")[0].replace("mi355::", "").replace("void ", "")[:40]
    print(f"{(r[0]-t0)/1e3:10.1f} us  +{(r[1]-r[0])/1e3:8.1f} us  wgs {r[3]//max(r[4],1):5d}  q {r[5]} s {r[6]}  {nm}")
PY
head -60 $O/trace_window.txt; cat $O/bench.json | head -c 600
