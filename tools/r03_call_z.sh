#!/bin/bash
# Round 3, GPU call Z: the group kernels of the MFMA sweeps with unconditional block loads -- parity, then solve timings and a kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; rm -rf $O; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "solve or rhs or wide" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for PN in "lap3d 64" "lap3d 8" "helm 1" "helm 64" "elast 64"; do
  set -- $PN
  timeout -k 10 300 python -u bench.py --problem $1 --steps 1 --warmup 0 --no-cpu --solve --nrhs $2 > $O/bench_$1_$2.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_$1_$2.json')); print('$1', d['config']['solve'])"
done
bash tools/solve_profile.sh 64 lap3d 150 > $O/solve_kernels_lap3d150_nrhs64.txt 2>&1 || { tail $O/solve_kernels_lap3d150_nrhs64.txt; exit 1; }
grep "group_mma" -A3 $O/solve_kernels_lap3d150_nrhs64.txt | head -40
