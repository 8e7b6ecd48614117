#!/bin/bash
# Round 2, GPU call D: does tuning key 14 (tiny launch classes) pay on the 3D configs too?  --no-cpu bench lines.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
for P in lap3d elast; do
  for K in 0 2048; do
    timeout -k 10 300 python -u bench.py --problem $P --steps 5 --warmup 2 --no-cpu --tune 14=$K > $O/bench_${P}_tiny$K.json 2> $O/bench_${P}_tiny$K.err || exit $?
    python3 -c "import json; d=json.load(open('$O/bench_${P}_tiny$K.json')); print('$P tiny_max', $K, d['ms_per_step'], 'ms', d['value'], 'GFLOP/s', d['config'].get('kernel_ms'))"
  done
done
