#!/bin/bash
# Round 3, GPU call Q: panel groups (tuning key 28) -- parity cases, then the bench with the key off / 64 / 32 / 16 rows per workgroup
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "panel_groups" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for K in 0 64 32 16; do
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --tune 28=$K > $O/bench_lap3d150_k$K.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_lap3d150_k$K.json')); print('lap3d150 key28=$K', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['roofline']['frac'], d['config']['kernel_ms'])"
done
for P in lap2d elast; do for K in 0 64 32; do
  timeout -k 10 300 python -u bench.py --problem $P --steps 3 --warmup 1 --no-cpu --tune 28=$K > $O/bench_${P}_k$K.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_${P}_k$K.json')); print('$P key28=$K', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_ms'])"
done; done
