#!/bin/bash
# Round 2, GPU call C: the other BASELINE configs (bench lines with roofline + cpu_baseline), the solve kernel traces
# at nrhs 1 / 64, the parked tiny-class tests and their sweep on the 2D problem.
# usage: gpurun --timeout 1200 -- 'bash tools/r02_call_c.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
for P in elast helm lap2d; do
    timeout -k 10 300 python -u bench.py --problem $P --steps 5 --warmup 2 > $O/bench_$P.json 2> $O/bench_$P.err || exit $?
    tail -c 1200 $O/bench_$P.json
done
timeout -k 10 240 python -u tools/time_host_inclusive.py 100 $O/host_inclusive_lap3d100.json > $O/host_inclusive.log 2>&1 || { tail -5 $O/host_inclusive.log; exit 1; }
tail -1 $O/host_inclusive.log
timeout -k 10 200 bash tools/solve_profile.sh 64 > $O/solve_nrhs64.txt 2>&1 || exit $?
timeout -k 10 200 python -u -m pytest tests/test_gpu_tiny_tasks.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/wip_tiny.log 2>&1 || { tail -20 $O/wip_tiny.log; exit 1; }
for K in 0 256 2048 16384; do
    timeout -k 10 120 python -u bench.py --problem lap2d --steps 5 --warmup 2 --no-cpu --tune 14=$K > $O/bench_lap2d_tiny$K.json 2> $O/bench_lap2d_tiny$K.err || exit $?
    python3 -c "import json,sys; d=json.load(open('$O/bench_lap2d_tiny$K.json')); print('tiny_max', $K, d['ms_per_step'], 'ms', d['value'], 'GFLOP/s')"
done
