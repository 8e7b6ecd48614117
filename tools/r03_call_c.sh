#!/bin/bash
# Round 3, GPU call C: the wide multi-RHS sweeps (tuning key 23) -- parity first, then nrhs = 64 solve times at 150^3 and
# config 5 with the key on and off.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "wide_multi or every_solve_path or multi_rhs_solve or complex_hermitian or solve_with_permutation" > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -3 $O/parity.log
for w in 1 0; do
timeout -k 10 400 python -u bench.py --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 --tune 23=$w > $O/solve64_lap3d150_wide$w.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; print('lap3d150 wide=$w', json.load(open('$O/solve64_lap3d150_wide$w.json'))['config']['solve'])"
timeout -k 10 400 python -u bench.py --problem helm --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 --tune 23=$w > $O/solve64_helm100_wide$w.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; print('helm100 wide=$w', json.load(open('$O/solve64_helm100_wide$w.json'))['config']['solve'])"
done
