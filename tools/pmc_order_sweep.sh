#!/bin/bash
# FETCH_SIZE of the GEMM kernels at 100^3 for three tile orders (one --pmc pass each)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for t in "4=1,5=0" "4=8,5=0" "4=8,5=1"; do
  d=$R/gpurun_out/pmc_order_${t//[=,]/_}
  rm -rf $d
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $d -o f -- python3 $R/bench.py --grid 100 --steps 1 --warmup 0 --no-cpu --tune "$t" > $d.log 2>&1
  db=$(find $d -name '*.db' | head -1)
  echo "== $t"
  python3 $R/tools/pmc_traffic.py $db $db | head -4
done
