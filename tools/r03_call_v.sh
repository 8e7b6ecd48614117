#!/bin/bash
# Round 3, GPU call V: kernel trace of one factorization + solves (how long k_panel_trsm_reg / k_panel_diag / the wide sweeps take now)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; rm -rf $O; mkdir -p $O
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/bench.py --grid 150 --steps 2 --warmup 1 --no-cpu --solve --nrhs 64 > $O/bench.json 2> $O/trace.err || { tail $O/trace.err; exit 1; }
python3 $R/tools/rocpd_summary.py $(find $O/trace -name '*.db' | head -1) > $O/kernel_stats.txt
rm -rf $O/trace
head -16 $O/kernel_stats.txt | cut -c1-150
python3 -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['solve'])"
