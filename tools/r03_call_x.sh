#!/bin/bash
# Round 3, GPU call X: the DIRECT_SUB epilogue of k_gemm128 with 1 / 2 / 4 row blocks per read-modify-write batch (G128_RMW), three builds of the library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03x; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
cp suitesparse_amd/libmi355_cholmod.so $O/lib_keep.so
for V in 4 1 4 1; do
  cp build_exp/lib_rmw$V.so suitesparse_amd/libmi355_cholmod.so
  for P in lap3d elast; do
  timeout -k 10 300 python -u bench.py --problem $P --steps 3 --warmup 1 --no-cpu > $O/bench_${P}_rmw$V.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_${P}_rmw$V.json')); print('$P G128_RMW=$V', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['roofline']['frac'])"
  done
done
cp $O/lib_keep.so suitesparse_amd/libmi355_cholmod.so; rm $O/lib_keep.so
