#!/bin/bash
# Round 3, GPU call A: what round 2 wrote but never ran -- the parked distributed-top cases with the default block
# widths, and the bench lines on the reference's own METIS ordering (configs 2 and 3).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 400 python -u -m pytest tests/wip_distributed_default_widths.py -m gpu -x -v --timeout 300 --timeout-method thread > $O/wip_dist.log 2>&1 || { tail -40 $O/wip_dist.log; exit 1; }
tail -5 $O/wip_dist.log
timeout -k 10 500 python -u bench.py --problem lap2d --ordering metis --steps 10 --warmup 2 > $O/bench_lap2d_metis.json 2> $O/bench_lap2d_metis.err || { tail -20 $O/bench_lap2d_metis.err; exit 1; }
cat $O/bench_lap2d_metis.json
timeout -k 10 900 python -u bench.py --problem lap3d --ordering metis --steps 5 --warmup 1 > $O/bench_lap3d150_metis.json 2> $O/bench_lap3d150_metis.err || { tail -20 $O/bench_lap3d150_metis.err; exit 1; }
cat $O/bench_lap3d150_metis.json
