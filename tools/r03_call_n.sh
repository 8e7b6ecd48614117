#!/bin/bash
# Round 3, GPU call N: kernel traces of the few-RHS sweeps (helm 100^3 nrhs = 1, lap3d 150^3 nrhs = 8)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
bash tools/solve_profile.sh 1 helm 100 > $O/solve_kernels_helm100_nrhs1.txt 2> $O/prof.err || { tail $O/prof.err; exit 1; }
grep -v "workgroups <" $O/solve_kernels_helm100_nrhs1.txt | cut -c1-140
cd $R
bash tools/solve_profile.sh 8 lap3d 150 > $O/solve_kernels_lap3d150_nrhs8.txt 2> $O/prof.err || { tail $O/prof.err; exit 1; }
grep -v "workgroups <" $O/solve_kernels_lap3d150_nrhs8.txt | cut -c1-140
