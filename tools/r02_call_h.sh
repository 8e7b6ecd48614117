#!/bin/bash
# Round 2, GPU call H (run twice: prefetch for all blocks, then for complex blocks only): the forward multi-RHS kernel with the next tile's operand requested before the results of the
# current one go out.  The driver's test command first (parity gate), then the two nrhs = 64 solve lines, then the
# driver's bench command.  Steps chained: nothing follows a failure.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 400 --timeout-method thread > $O/gpu_tests.log 2>&1 \
  || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 200 python -u bench.py --problem lap3d --steps 2 --warmup 1 --no-cpu --solve --nrhs 64 > $O/bench_solve64.json 2> $O/solve64.err || { tail $O/solve64.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_solve64.json')); print('lap3d', d['config']['solve'])"
timeout -k 10 200 python -u bench.py --problem helm --steps 2 --warmup 1 --no-cpu --solve --nrhs 64 > $O/bench_helm_solve64.json 2> $O/helm64.err || { tail $O/helm64.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench_helm_solve64.json')); print('helm', d['config']['solve'])"
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_lap3d150.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cat $O/bench_lap3d150.json
