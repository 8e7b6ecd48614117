#!/bin/bash
# Round 3, GPU call B: HBM traffic (two PMC passes, one counter each) of configs 2, 4, 5 so that roofline.traffic is not
# null there.  Output: gpurun_out/r03b/pmc_traffic_<problem><grid>.{json,txt}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
for pg in "lap2d 2000" "helm 100" "elast 80"; do
  set -- $pg; P=$1; G=$2
  timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$P -o f -- python3 $R/bench.py --problem $P --grid $G --steps 1 --warmup 0 --no-cpu > $O/bench_fetch_$P.json 2> $O/fetch_$P.err || { tail $O/fetch_$P.err; exit 1; }
  timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$P -o w -- python3 $R/bench.py --problem $P --grid $G --steps 1 --warmup 0 --no-cpu > $O/bench_write_$P.json 2> $O/write_$P.err || { tail $O/write_$P.err; exit 1; }
  python3 $R/tools/pmc_traffic.py $(find $O/fetch_$P -name '*.db' | head -1) $(find $O/write_$P -name '*.db' | head -1) $O/pmc_traffic_$P$G.json > $O/pmc_traffic_$P$G.txt || exit 1
  rm -rf $O/fetch_$P $O/write_$P
  head -6 $O/pmc_traffic_$P$G.txt
done
