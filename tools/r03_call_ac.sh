#!/bin/bash
# Round 3, GPU call AC: look-ahead of the sweeps, second form (near tiles on the per-block lists, far rows on the wide lists on the
# second stream): parity of the solve tests, sweeps of configs 3 / 5 with key 30 = 0 / 12, and the trace window of the chain
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ac; rm -rf $O; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solve or sweep or rhs" --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for P in "lap3d 150 64" "helm 100 64" "helm 100 1" "lap3d 150 8" "elast 80 64"; do
  set -- $P
  for T in 0 12; do
    timeout -k 10 300 python -u bench.py --problem $1 --grid $2 --steps 2 --warmup 1 --no-cpu --solve --nrhs $3 --tune 30=$T > $O/bench_$1_$3_$T.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
    python3 -c "import json; d=json.load(open('$O/bench_$1_$3_$T.json')); print('$1 nrhs $3 key30=$T', json.dumps(d['config']['solve']))"
  done
done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O -o s -- python3 $R/bench.py --problem lap3d --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/bench.json 2> $O/err.txt || { tail $O/err.txt; exit 1; }
echo traced
