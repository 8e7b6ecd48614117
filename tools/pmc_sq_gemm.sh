#!/bin/bash
# SQ stall breakdown of k_gemm128<16x16x4> on one 16384^2 x 512 trailing update
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/pmc_sq_gemm; rm -rf $d
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $d -o f -- python3 $R/tools/gemm_one.py 2 16384 16384 512 5
python3 $R/tools/pmc_counters.py $(find $d -name '*.db' | head -1)
python3 $R/tools/rocpd_summary.py $(find $d -name '*.db' | head -1)
d=$R/gpurun_out/pmc_sq_gemm2; rm -rf $d
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $d -o f -- python3 $R/tools/gemm_one.py 2 16384 16384 512 5
python3 $R/tools/pmc_counters.py $(find $d -name '*.db' | head -1)
