#!/bin/bash
# Round 3, GPU call AD: look-ahead of the sweeps, the near part issued before the event (experiment, forward sweep only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ad; rm -rf $O; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
for P in "lap3d 150 64" "helm 100 64" "elast 80 64"; do
  set -- $P
  for NF in 0 1; do
    if [ $NF = 1 ]; then export MI355_NEAR_FIRST=1; else unset MI355_NEAR_FIRST; fi
    timeout -k 10 300 python -u bench.py --problem $1 --grid $2 --steps 2 --warmup 1 --no-cpu --solve --nrhs $3 --tune 30=12 > $O/bench_$1_$3_$NF.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
    python3 -c "import json; d=json.load(open('$O/bench_$1_$3_$NF.json')); print('$1 nrhs $3 near_first=$NF', json.dumps(d['config']['solve']))"
  done
done
