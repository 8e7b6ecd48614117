#!/bin/bash
# Round 3, GPU call J: records of the wide sweeps -- HBM traffic per solve kernel at 150^3 nrhs = 64 (two PMC passes), kernel
# trace, and the complete bench lines of configs 3 and 5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
bash tools/pmc_solve_traffic.sh > $O/pmc_solve.out 2>&1 || { tail $O/pmc_solve.out; exit 1; }
cp $R/gpurun_out/pmc_solve/pmc_traffic_solve64.txt $R/gpurun_out/pmc_solve/pmc_traffic_solve64.json $O/
cat $O/pmc_traffic_solve64.txt
cd $R
bash tools/solve_profile.sh 64 > $O/solve_kernels_nrhs64.txt 2> $O/prof.err || { tail $O/prof.err; exit 1; }
cd $R
timeout -k 10 600 python -u bench.py --problem helm --steps 5 --warmup 1 > $O/bench_helm.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; d=json.load(open('$O/bench_helm.json')); print('helm100', d['value'], d['config']['solve'])"
