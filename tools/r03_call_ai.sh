#!/bin/bash
# Round 3, GPU call AI: the look-ahead (key 30) again, now that the group step is short (key 31 = 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ai; rm -rf $O; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
for P in "lap3d 150 64" "elast 80 64" "helm 100 64" "helm 100 1" "lap3d 150 8"; do
  set -- $P
  for T in 0 12; do
    timeout -k 10 300 python -u bench.py --problem $1 --grid $2 --steps 2 --warmup 1 --no-cpu --solve --nrhs $3 --tune 30=$T > $O/bench_$1_$3_$T.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
    python3 -c "import json; d=json.load(open('$O/bench_$1_$3_$T.json')); print('$1 nrhs $3 key30=$T', json.dumps(d['config']['solve']))"
  done
done
