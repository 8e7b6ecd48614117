#!/bin/bash
# Round 3, GPU call G: kernel trace of the nrhs = 64 solve at 150^3 next to the bytes every wide launch reads (MI355_SOLVE_LOG)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
export MI355_SOLVE_LOG=$O/solve_launches.log
bash tools/solve_profile.sh 64 > $O/solve_kernels_nrhs64.txt 2> $O/prof.err || { tail $O/prof.err; exit 1; }
unset MI355_SOLVE_LOG
tail -2 $O/solve_kernels_nrhs64.txt | cut -c1-300
python3 - $R/gpurun_out/prof_solve_64/s_results.db $O/solve_launches.log > $O/wide_launch_rates.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
log = [l.split() for l in open(sys.argv[2])]
for kind, pat in (("fwd", "%k_solve_fwd_wide%"), ("bwd", "%k_solve_bwd_wide%")):
    rows = db.execute("select duration, name from kernels where name like ? order by start", (pat,)).fetchall()
    ll = [l for l in log if l[0] == kind]
    assert len(rows) == len(ll), (len(rows), len(ll))
    n = len(rows) // 2
    tot_b = tot_t = 0
    print(kind, "timed sweep: us, workgroups, MB, TB/s, tasks, maxrows")
    for (d, name), l in list(zip(rows, ll))[n:]:
        b = float(l[4]); tot_b += b; tot_t += d
        print(f"  {d/1e3:8.1f} {int(l[2]):5d} {b/1e6:9.1f} {b/d*1e-3:6.2f} {l[6]:>5s} {l[8]:>6s}")
    print(kind, f"total {tot_t/1e6:.2f} ms, {tot_b/1e9:.2f} GB, {tot_b/tot_t*1e-3:.2f} TB/s")
PY
tail -1 $O/wide_launch_rates.txt
