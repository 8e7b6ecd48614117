#!/usr/bin/env python3
"""Phase statistics of k_gemm128 from the MI355_GEMM_STAMPS dump (tile tm tn t0 t1 t2 t3; shader clocks)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.uint64)
ok = d[:, 6] > 0                       # waves that skipped (upper quadrant) or wrote nothing have zeros
d = d[ok]
t0, t1, t2, t3 = (d[:, i].astype(np.int64) for i in (3, 4, 5, 6))
base = t0.min()
print(f"tiles with stamps {len(d)}   launch span {(t3.max() - base)/1e3:.1f} kcycles")
for name, v in (("prologue", t1 - t0), ("main loop", t2 - t1), ("epilogue", t3 - t2), ("total", t3 - t0)):
    print(f"{name:10s} mean {v.mean()/1e3:8.1f}  p10 {np.percentile(v,10)/1e3:8.1f}  p50 {np.percentile(v,50)/1e3:8.1f}  p90 {np.percentile(v,90)/1e3:8.1f}  max {v.max()/1e3:8.1f} kcycles")
start = np.sort(t0 - base)
print("start times (kcycles) of tiles 0,256,511,512,768,1024:", [float(start[i]) / 1e3 for i in (0, 256, 511, 512, 768, 1024) if i < len(start)])
# busy fraction in epilogue/prologue across time
edges = np.linspace(0, float(t3.max() - base), 41)
mid = 0.5 * (edges[1:] + edges[:-1])
inmain = [np.mean(((t1 - base) <= m) & ((t2 - base) > m)) * len(d) for m in mid]
inepi = [np.mean(((t2 - base) <= m) & ((t3 - base) > m)) * len(d) for m in mid]
inpro = [np.mean(((t0 - base) <= m) & ((t1 - base) > m)) * len(d) for m in mid]
print("time-slice census (workgroups in prologue/main/epilogue):")
for m, a, b, c in zip(mid, inpro, inmain, inepi):
    print(f"  t={m/1e3:9.1f}k  pro {a:6.0f}  main {b:6.0f}  epi {c:6.0f}")
