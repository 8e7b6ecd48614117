#!/bin/bash
# Round 3, GPU call T: SQ counters of the solve kernels at 150^3, nrhs = 64 (what the wide sweeps wait for), one --pmc pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; rm -rf $O; mkdir -p $O
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/sq -o f -- python3 $R/bench.py --grid 150 --steps 1 --warmup 0 --no-cpu --solve --nrhs 64 > $O/bench.json 2> $O/err.txt || { tail $O/err.txt; exit 1; }
python3 $R/tools/pmc_counters.py $(find $O/sq -name '*.db' | head -1) | grep -i "wide\|gemm128\|panel" > $O/sq_counters.txt
rm -rf $O/sq
cat $O/sq_counters.txt
