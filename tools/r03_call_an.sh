#!/bin/bash
# Round 3, GPU call AN: small wide steps cut along the columns too (key 33): parity, then the sweeps with 0 / 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03an; rm -rf $O; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solve or sweep or rhs or golden or inverses" --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for P in "lap3d 150 64" "helm 100 64" "elast 80 64" "lap3d 150 8" "helm 100 1" "lap2d 2000 64"; do
  set -- $P
  for T in 0 1; do
    timeout -k 10 300 python -u bench.py --problem $1 --grid $2 --steps 2 --warmup 1 --no-cpu --solve --nrhs $3 --tune 33=$T > $O/bench_$1_$3_$T.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
    python3 -c "import json; d=json.load(open('$O/bench_$1_$3_$T.json')); print('$1 nrhs $3 key33=$T', json.dumps(d['config']['solve']))"
  done
done
