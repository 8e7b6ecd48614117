#!/bin/bash
# Round 3, GPU call H: wide sweeps with the tiles of a step dealt contiguously to <= 256 workgroups: parity, times, per-launch rates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "wide_multi or every_solve_path or multi_rhs_solve or complex_hermitian or solve_with_permutation" > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -3 $O/parity.log
timeout -k 10 400 python -u bench.py --problem helm --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/solve64_helm100.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; print('helm100', json.load(open('$O/solve64_helm100.json'))['config']['solve'])"
export MI355_SOLVE_LOG=$O/solve_launches.log
bash tools/solve_profile.sh 64 > $O/solve_kernels_nrhs64.txt 2> $O/prof.err || { tail $O/prof.err; exit 1; }
unset MI355_SOLVE_LOG
tail -2 $O/solve_kernels_nrhs64.txt | cut -c1-100
python3 -c "import json,sys; print('lap3d150', json.load(open('$R/gpurun_out/prof_solve_64/bench.json'))['config']['solve'])"
python3 tools/wide_launch_rates.py $R/gpurun_out/prof_solve_64/s_results.db $O/solve_launches.log > $O/wide_launch_rates.txt
tail -2 $O/wide_launch_rates.txt
