#!/bin/bash
# Round 3, GPU call U: forward wide kernel with two / three register sets in flight (key 29), and the factorization with the final k_panel_diag
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03u; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
true
true
for K in 2 3; do
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --solve --nrhs 64 --tune 29=$K > $O/bench_lap3d150_ring$K.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_lap3d150_ring$K.json')); print('lap3d150 key29=$K', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['config']['kernel_ms'], d['config']['solve'])"
done
for P in elast helm; do for K in 2 3; do
  timeout -k 10 300 python -u bench.py --problem $P --steps 2 --warmup 1 --no-cpu --solve --nrhs 64 --tune 29=$K > $O/bench_${P}_ring$K.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_${P}_ring$K.json')); print('$P key29=$K', d['value'], d['ms_per_step'], d['config']['solve'])"
done; done
