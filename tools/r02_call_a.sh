#!/bin/bash
# Round 2, GPU call A: the -m gpu suite without its three heaviest groups (full-size properties, live reference at
# scale), every step under its own time limit; nothing follows a failure of the process itself.  Call A2 runs the rest.
# usage: gpurun --timeout 1200 -- 'bash tools/r02_call_a.sh [heavy]'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
if [ "$1" = "heavy" ]; then
    timeout -k 10 1100 python -u -m pytest tests -m gpu -v --timeout 900 --timeout-method thread \
        -k "full_size or test_vs_live_reference or not_posdef_minor_at_scale" > $O/gpu_tests_heavy.log 2>&1
    rc=$?
    tail -25 $O/gpu_tests_heavy.log
else
    timeout -k 10 1100 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread \
        -k "not full_size and not test_vs_live_reference and not not_posdef_minor_at_scale" > $O/gpu_tests.log 2>&1
    rc=$?
    tail -25 $O/gpu_tests.log
fi
exit $rc
