#!/bin/bash
# SQ stall breakdown of k_gemm128 on one 16384^2 x K trailing update (K = $1, default 512)
set -e
K=${1:-512}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/pmc_sq_gemm; rm -rf $d
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $d -o f -- python3 $R/tools/gemm_one.py 1 16384 16384 $K 5 2>&1 | grep -v "simple_timer\|generateRocpd\|tool.cpp"
python3 $R/tools/pmc_counters.py $(find $d -name '*.db' | head -1) | grep -v rocclr
python3 $R/tools/rocpd_summary.py $(find $d -name '*.db' | head -1) | grep -v rocclr
