#!/bin/bash
# TIMING-ONLY experiment (results of the variants are wrong on purpose; nothing here is shipped or tested): what bounds
# k_solve_fwd_mma<1> at nrhs = 64?  Builds three variants of the library in a scratch copy of the tree on the GPU box
# and prints the forward / backward sweep times of each at 150^3:
#   plain    every result row written by plain read-modify-write instead of the FP64 atomic
#   nostore  results not written at all (one guarded store keeps the products alive)
#   nomfma   the products skipped, the atomics kept
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp_fwd; mkdir -p $O
K=suitesparse_amd/csrc/solve_kernels.hip.hpp
LINE='if (own) yp\[16 \* jb\] -= acc\[jb\]\[g\]; else atomic_add_f64(yp + 16 \* jb, -acc\[jb\]\[g\]);'
MF='acc\[jb\] = __builtin_amdgcn_mfma_f64_16x16x4f64(al\[q\], Xs\[16 \* jb + l15\]\[4 \* q + l4\], acc\[jb\], 0, 0, 0);'
for V in plain nostore nomfma; do
    W=/tmp/exp_$V; rm -rf $W; mkdir -p $W; cp -r $R/suitesparse_amd $R/include $R/oracle $R/bench.py $R/BASELINE.json $W/ 2>/dev/null
    cd $W
    case $V in
      plain)   sed -i "s/$LINE/yp[16 * jb] -= acc[jb][g];/" $K ;;
      nostore) sed -i "s/$LINE/if (acc[jb][g] == 1.2345e300) yp[16 * jb] = 0.0;/" $K ;;
      nomfma)  sed -i "s/$MF/acc[jb][0] += al[q];/" $K ;;
    esac
    if cmp -s $K $R/$K; then echo "$V: pattern not found"; exit 1; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-result \
        suitesparse_amd/csrc/engine.hip suitesparse_amd/csrc/schedule.cpp suitesparse_amd/csrc/symbolic.cpp \
        -o suitesparse_amd/libmi355_cholmod.so 2> $O/build_$V.err || { tail $O/build_$V.err; exit 1; }
    timeout -k 10 200 python -u bench.py --problem lap3d --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/$V.json 2> $O/$V.err || { tail $O/$V.err; exit 1; }
    python3 -c "import json; d=json.load(open('$O/$V.json')); print('$V', d['config']['solve'])" | tee -a $O/summary.txt
done
