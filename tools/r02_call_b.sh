#!/bin/bash
# Round 2, GPU call B: smoke, the driver's exact bench command, then the rocprofv3 kernel summary and the two PMC
# traffic passes of the same workload (tools/profile_round.sh).  Results: gpurun_out/r02/ -> copy into profiles/r02_*.
# usage: gpurun --timeout 1200 -- 'bash tools/r02_call_b.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && \
timeout -k 10 500 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_lap3d150.json 2> $O/bench_lap3d150.err && \
tail -c 2500 $O/bench_lap3d150.json && \
timeout -k 10 420 bash tools/profile_round.sh r02 > $O/profile_round.log 2>&1
rc=$?
tail -30 $O/profile_round.log
exit $rc
