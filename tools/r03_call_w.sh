#!/bin/bash
# Round 3, GPU call W: nested block widths again, now that the 64-column level is inside the panel groups (keys 7 = second width, 8 = third)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03w; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
for T in "7=1024" "7=1024,8=4096" "7=1024,8=3072" "7=768" "7=1280" "7=1536" "7=2048" "7=512,8=2048"; do
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --tune $T > $O/bench_$T.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_$T.json')); print('$T', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['roofline']['frac'], d['roofline']['launches'])"
done
