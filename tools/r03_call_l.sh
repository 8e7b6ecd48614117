#!/bin/bash
# Round 3, GPU call L: the round's records of the factorization -- rocprofv3 kernel trace + PMC traffic of the default bench
# (tools/profile_round.sh), the driver's bench command, the bench lines of configs 2 and 4, and the per-launch step log of config 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
bash tools/profile_round.sh r03 > $O/profile_round.out 2>&1 || { tail -20 $O/profile_round.out; exit 1; }
head -12 $R/gpurun_out/prof_r03/kernel_stats.txt | cut -c1-160
cd $R
timeout -k 10 900 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_lap3d150.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_lap3d150.json')); print('lap3d150', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['roofline'], d['config']['solve'])"
timeout -k 10 600 python -u bench.py --problem lap2d --steps 10 --warmup 2 > $O/bench_lap2d.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_lap2d.json')); print('lap2d', d['value'], d['ms_per_step'], d['roofline'], d['config']['solve'])"
timeout -k 10 600 python -u bench.py --problem elast --steps 5 --warmup 1 > $O/bench_elast.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_elast.json')); print('elast', d['value'], d['ms_per_step'], d['roofline'], d['config']['solve'])"
MI355_STEP_LOG=$O/steps_lap2d.log timeout -k 10 300 python -u bench.py --problem lap2d --steps 1 --warmup 0 --no-cpu > $O/bench_steplog_lap2d.json 2> $O/steplog.err || { tail $O/steplog.err; exit 1; }
python3 tools/step_log_levels.py $O/steps_lap2d.log > $O/step_levels_lap2d.txt; python3 tools/step_log_stats.py $O/steps_lap2d.log > $O/step_breakdown_lap2d.txt
cat $O/step_levels_lap2d.txt | head -30
