#!/bin/bash
# TIMING-ONLY experiment (the variant's results are wrong on purpose; nothing here is shipped or tested): how much of
# k_solve_bwd_mma_run<1> at nrhs = 64 is the gather of the rows of Y?  Builds one variant of the library in a scratch
# copy of the tree on the GPU box (the gathered value replaced by a function of the row index) and prints the sweeps.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp_bwd; mkdir -p $O
K=suitesparse_amd/csrc/solve_kernels.hip.hpp
W=/tmp/exp_nogather; rm -rf $W; mkdir -p $W; cp -r $R/suitesparse_amd $R/include $R/oracle $R/bench.py $R/BASELINE.json $W/ 2>/dev/null
cd $W
sed -i 's/                v = Y\[row \* ldr + j0 + lane\];/                v = (double)(row \& 7);/' $K
if cmp -s $K $R/$K; then echo "pattern not found"; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-result \
    suitesparse_amd/csrc/engine.hip suitesparse_amd/csrc/schedule.cpp suitesparse_amd/csrc/symbolic.cpp \
    -o suitesparse_amd/libmi355_cholmod.so 2> $O/build.err || { tail $O/build.err; exit 1; }
timeout -k 10 200 python -u bench.py --problem lap3d --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/nogather.json 2> $O/nogather.err || { tail $O/nogather.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/nogather.json')); print('nogather', d['config']['solve'])" | tee $O/summary.txt
