#!/bin/bash
# Round 2, GPU call A: the whole -m gpu suite (every step under its own time limit; nothing follows a failure).
# usage: gpurun --timeout 1200 -- 'bash tools/r02_call_a.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout -k 10 1100 python -u -m pytest tests -m gpu -v --timeout 600 --timeout-method thread > $O/gpu_tests.log 2>&1
rc=$?
tail -15 $O/gpu_tests.log
exit $rc
