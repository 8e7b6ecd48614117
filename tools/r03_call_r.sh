#!/bin/bash
# Round 3, GPU call R: panel groups with the register-resident TRSM (key 28 = 128 / 64) against the LDS-image one (16)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O
cd $R
( while true; do sleep 45; echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "panel_groups" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for K in ${KEYS:-0 128 64}; do
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --tune 28=$K > $O/bench_lap3d150_k$K.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
  python3 -c "import json; d=json.load(open('$O/bench_lap3d150_k$K.json')); print('lap3d150 key28=$K', d['value'], d['ms_per_step'], d['config']['frac_fp64_mfma_peak'], d['roofline']['frac'], d['config']['kernel_ms'])"
done
MI355_STEP_LOG=$O/steps_lap3d150.log timeout -k 10 300 python -u bench.py --steps 1 --warmup 0 --no-cpu --tune 28=${BEST:-128} > $O/bench_steplog.json 2> $O/steplog.err || { tail $O/steplog.err; exit 1; }
grep -c . $O/steps_lap3d150.log
python3 - <<PY
import re,collections
t=collections.defaultdict(lambda:[0,0.0])
for l in open('$O/steps_lap3d150.log'):
    k=l.split()[0]; ms=float(l.split()[-1])
    if k=='gemm':
        m=re.search(r'big (\d) mode (\d).*kmax (\d+)',l); k='gemm big%s mode%s K%s'%(m.group(1),m.group(2),('64' if int(m.group(3))<=64 else '256' if int(m.group(3))<=256 else 'big'))
    t[k][0]+=1; t[k][1]+=ms
for k,v in sorted(t.items(), key=lambda x:-x[1][1]): print('%-28s %6d launches %9.2f ms  %8.1f us each'%(k,v[0],v[1],1e3*v[1]/v[0]))
PY
