#!/bin/bash
# Round 3, GPU call K: POTRF fused into the trailing update (tuning key 26): the whole GPU suite, then configs 3 / 2 / 4 with the key on and off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
for w in 1 0; do
timeout -k 10 400 python -u bench.py --grid 150 --steps 5 --warmup 1 --no-cpu --tune 26=$w > $O/bench_lap3d150_fuse$w.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_lap3d150_fuse$w.json')); print('lap3d150 fuse=$w', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['roofline']['frac'], d['config']['kernel_ms'])"
timeout -k 10 400 python -u bench.py --problem lap2d --steps 10 --warmup 2 --no-cpu --tune 26=$w > $O/bench_lap2d_fuse$w.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_lap2d_fuse$w.json')); print('lap2d fuse=$w', d['value'], d['ms_per_step'])"
timeout -k 10 400 python -u bench.py --problem elast --steps 5 --warmup 1 --no-cpu --tune 26=$w > $O/bench_elast_fuse$w.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_elast_fuse$w.json')); print('elast fuse=$w', d['value'], d['ms_per_step'])"
done
