#!/bin/bash
# HBM traffic per solve kernel at 150^3, nrhs = 64 (two PMC passes, one counter each, kernel trace only).
# usage (GPU box): bash tools/pmc_solve_traffic.sh     -> gpurun_out/pmc_solve/pmc_traffic_solve64.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_solve; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- python3 $R/bench.py --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/bench_fetch.json 2> $O/fetch.err || { tail $O/fetch.err; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- python3 $R/bench.py --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/bench_write.json 2> $O/write.err || { tail $O/write.err; exit 1; }
python3 $R/tools/pmc_traffic.py $(find $O/fetch -name '*.db' | head -1) $(find $O/write -name '*.db' | head -1) $O/pmc_traffic_solve64.json > $O/pmc_traffic_solve64.txt
rm -rf $O/fetch $O/write
cat $O/pmc_traffic_solve64.txt
