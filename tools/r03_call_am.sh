#!/bin/bash
# Round 3, GPU call AM: 2 .. 4 column-major right-hand sides through the inverse group step: parity, then the crossover sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03am; rm -rf $O; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solve or sweep or rhs or golden or inverses" --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 400 python -u tools/solve_mma_min_sweep.py lap3d 150 1,2,3,4 > $O/sweep.txt 2>&1 || { tail $O/sweep.txt; exit 1; }
grep nrhs $O/sweep.txt
