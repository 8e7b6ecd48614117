#!/bin/bash
# Round 3, GPU call O: the round's records on the default tuning -- rocprofv3 kernel trace + PMC traffic of the default bench
# (tools/profile_round.sh), the driver's bench command, and the bench lines of configs 2, 4 and 5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
bash tools/profile_round.sh r03 > $O/profile_round.out 2>&1 || { tail -20 $O/profile_round.out; exit 1; }
head -12 $R/gpurun_out/prof_r03/kernel_stats.txt | cut -c1-160
cd $R
timeout -k 10 900 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_lap3d150.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_lap3d150.json')); print('lap3d150', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['roofline'], d['config']['solve'])"
for P in lap2d elast helm; do
  timeout -k 10 600 python -u bench.py --problem $P --steps 5 --warmup 1 > $O/bench_$P.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_$P.json')); print('$P', d['value'], d['ms_per_step'], d['roofline'], d['config']['solve'])"
done
