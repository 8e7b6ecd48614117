#!/bin/bash
# Round profile of the default bench workload (lap3d 150^3): kernel trace + the two PMC traffic passes.
# usage (on the GPU box): bash tools/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/bench.py --grid 150 --steps 3 --warmup 1 --no-cpu > $O/bench_under_rocprof.json 2> $O/trace.err
python3 $R/tools/rocpd_summary.py $(find $O/trace -name '*.db' | head -1) > $O/kernel_stats.txt
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- python3 $R/bench.py --grid 150 --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- python3 $R/bench.py --grid 150 --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/write.err
python3 $R/tools/pmc_traffic.py $(find $O/fetch -name '*.db' | head -1) $(find $O/write -name '*.db' | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -rf $O/trace $O/fetch $O/write
cat $O/kernel_stats.txt $O/pmc_traffic.txt
