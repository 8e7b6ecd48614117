#!/bin/bash
# Round 2, GPU call I (measurement only): kernel traces of the solve at 150^3 for nrhs = 64 and 1 on the final tree, and
# the per-launch step log of one factorization grouped by kernel / mode / K.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O
cd $R
bash tools/solve_profile.sh 64 > $O/solve_kernels_nrhs64.txt 2> $O/solve64.err || { tail $O/solve64.err; exit 1; }
head -12 $O/solve_kernels_nrhs64.txt
bash tools/solve_profile.sh 1 > $O/solve_kernels_nrhs1.txt 2> $O/solve1.err || { tail $O/solve1.err; exit 1; }
cd $R
MI355_STEP_LOG=$O/steps.log timeout -k 10 300 python -u bench.py --grid 150 --steps 1 --warmup 0 --no-cpu > $O/bench_steplog.json 2> $O/steplog.err || { tail $O/steplog.err; exit 1; }
python3 tools/step_log_stats.py $O/steps.log > $O/step_breakdown.txt && rm -f $O/steps.log
cat $O/step_breakdown.txt
