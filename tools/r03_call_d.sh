#!/bin/bash
# Round 3, GPU call D: kernel trace of the nrhs = 64 solve at 150^3 with the wide sweeps, and the run-length knob (key 24)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
bash tools/solve_profile.sh 64 > $O/solve_kernels_nrhs64.txt 2> $O/prof.err || { tail $O/prof.err; exit 1; }
cat $O/solve_kernels_nrhs64.txt | cut -c1-150
cd $R
for w in 128 512 1024; do
timeout -k 10 400 python -u bench.py --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 --tune 24=$w > $O/solve64_wgs$w.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 -c "import json,sys; print('lap3d150 wgs=$w', json.load(open('$O/solve64_wgs$w.json'))['config']['solve'])"
done
