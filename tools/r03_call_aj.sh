#!/bin/bash
# Round 3, GPU call AJ: PTRSM with two row tiles per wave (key 32 = 2, rows per workgroup 128): parity of the panel-group tests
# (the variant was taken back after this call -- profiles/r03_tile_knobs_final_kernels.txt; key 32 no longer exists, the script is kept as the record of what ran)
# with it, then the 150^3 factorization with key 28 = 64 (default) / 128 / 128 + key 32 = 2, and the kernel's time from a trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03aj; rm -rf $O; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
for T in "28=64" "28=128" "28=128,32=2" "28=64"; do
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --tune $T > $O/bench_$T.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_$T.json')); print('$T', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'])"
done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $O/tr -o t -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --tune 28=128,32=2 > $O/bench_tr.json 2> $O/err.txt || { tail $O/err.txt; exit 1; }
python3 $R/tools/rocpd_summary.py $(find $O/tr -name '*.db' | head -1) | head -8 | cut -c1-150
rm -rf $O/tr
