#!/bin/bash
# Round 3, GPU call Y: the tile-order and tile-class knobs again on the final kernels (keys 4 = super-tile, 5 = XCD dealing, 0 / 1 = big-tile thresholds)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03y; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
for T in "4=8" "4=4" "4=16" "5=1" "0=128" "0=256" "1=64" "1=128" "4=8"; do
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --tune $T > $O/bench_$T.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_$T.json')); print('$T', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['roofline']['frac'], d['roofline']['launches'])"
done
