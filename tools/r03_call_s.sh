#!/bin/bash
# Round 3, GPU call S: per-level step logs (synchronised launches) of configs 2, 3 and 4 with panel groups on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
for P in lap2d lap3d elast; do
  MI355_STEP_LOG=$O/steps_$P.log timeout -k 10 300 python -u bench.py --problem $P --steps 1 --warmup 0 --no-cpu > $O/bench_steplog_$P.json 2> $O/steplog.err || { tail $O/steplog.err; exit 1; }
  python3 tools/step_log_levels.py $O/steps_$P.log > $O/step_levels_$P.txt
  python3 tools/step_log_stats.py $O/steps_$P.log > $O/step_breakdown_$P.txt 2>/dev/null || true
done
cat $O/step_levels_lap2d.txt
