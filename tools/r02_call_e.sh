#!/bin/bash
# Round 2, GPU call E: sweep of the two row-division knobs of the multi-RHS sweeps (tuning keys 13 backward, 15 forward)
# at nrhs = 64, 150^3.  One factorization per line (--steps 1), the solve is timed as in the default bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
for K in 512 1024 2048 4096 8192 16384; do
    timeout -k 10 200 python -u bench.py --problem lap3d --steps 1 --warmup 1 --no-cpu --solve --nrhs 64 --tune 13=$K,15=$K > $O/solve64_div$K.json 2> $O/solve64_div$K.err || exit $?
    python3 -c "import json; d=json.load(open('$O/solve64_div$K.json')); print('div', $K, d['config']['solve'])"
done
