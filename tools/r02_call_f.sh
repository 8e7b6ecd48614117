#!/bin/bash
# Round 2, GPU call F: single-knob sweeps of the factorization schedule at 150^3 (--no-cpu, 3 steps each).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
for T in "4=8" "4=4" "4=16" "5=1" "7=2048" "7=512" "2=128" "2=512" "0=128,1=64" "0=256,1=128" "8=4096"; do
    F=$O/sweep_$(echo $T | tr '=,' '__').json
    timeout -k 10 200 python -u bench.py --problem lap3d --steps 3 --warmup 1 --no-cpu --tune $T > $F 2> $F.err || { echo "$T failed"; tail -3 $F.err; continue; }
    python3 -c "import json; d=json.load(open('$F')); print('$T', d['ms_per_step'], 'ms', round(d['value']), 'GFLOP/s', 'frac', d['roofline']['frac'])"
done
