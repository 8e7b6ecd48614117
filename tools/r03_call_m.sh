#!/bin/bash
# Round 3, GPU call M: 16-column variants of the wide sweeps (few right-hand sides; a complex factor from nrhs = 1), the
# deterministic mode (key 27): parity, then times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_cholmod_dropin.py -m gpu -x -q --timeout 300 --timeout-method thread -k "wide_multi or every_solve_path or multi_rhs_solve or complex or solve_with_permutation or deterministic or solve2 or soft_fallback or memory_knobs" > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -3 $O/parity.log
for nr in 1 2 8; do
timeout -k 10 400 python -u bench.py --problem helm --steps 1 --warmup 0 --no-cpu --solve --nrhs $nr > $O/solve_helm100_nrhs$nr.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; print('helm100 nrhs=$nr', json.load(open('$O/solve_helm100_nrhs$nr.json'))['config']['solve'])"
done
for nr in 4 5 8 16; do
timeout -k 10 400 python -u bench.py --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs $nr > $O/solve_lap3d150_nrhs$nr.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; print('lap3d150 nrhs=$nr', json.load(open('$O/solve_lap3d150_nrhs$nr.json'))['config']['solve'])"
done
timeout -k 10 400 python -u bench.py --grid 150 --steps 3 --warmup 1 --no-cpu --tune 27=1 > $O/bench_lap3d150_deterministic.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_lap3d150_deterministic.json')); print('lap3d150 deterministic', d['value'], d['ms_per_step'], d['config']['kernel_ms'])"
timeout -k 10 400 python -u bench.py --problem lap2d --steps 3 --warmup 1 --no-cpu --tune 27=1 > $O/bench_lap2d_deterministic.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_lap2d_deterministic.json')); print('lap2d deterministic', d['value'], d['ms_per_step'], d['config']['kernel_ms'])"
