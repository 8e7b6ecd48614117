#!/bin/bash
# Round 3, GPU call I: forward wide kernel with 16 waves (key 25) on / off, leaf group kernels with 64 columns per workgroup
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "wide_multi or every_solve_path or multi_rhs_solve or complex_hermitian or solve_with_permutation" > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -3 $O/parity.log
for w in 1 0; do
timeout -k 10 400 python -u bench.py --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 --tune 25=$w > $O/solve64_lap3d150_k25_$w.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; print('lap3d150 key25=$w', json.load(open('$O/solve64_lap3d150_k25_$w.json'))['config']['solve'])"
done
timeout -k 10 400 python -u bench.py --problem helm --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/solve64_helm100.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; print('helm100', json.load(open('$O/solve64_helm100.json'))['config']['solve'])"
