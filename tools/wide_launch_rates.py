#!/usr/bin/env python3
"""per launch of the wide solve kernels: duration (rocprofv3 kernel trace, rocpd sqlite) next to the bytes of L it reads
(MI355_SOLVE_LOG of the same run; both in launch order).  usage: wide_launch_rates.py <results.db> <solve_launches.log>"""
import math
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
log = [l.split() for l in open(sys.argv[2])]
for kind, pat in (("fwd", "%k_solve_fwd_wide%"), ("bwd", "%k_solve_bwd_wide%")):
    rows = db.execute("select duration, name from kernels where name like ? order by start", (pat,)).fetchall()
    ll = [l for l in log if l[0] == kind]
    assert len(rows) == len(ll), (len(rows), len(ll))
    n = len(rows) // 2          # the bench runs every sweep twice: untimed, timed
    tot_b = tot_t = 0
    buckets = {}
    print(kind, "timed sweep: us, workgroups, MB, TB/s, tasks, maxrows")
    for (d, name), l in list(zip(rows, ll))[n:]:
        b = float(l[4]); tot_b += b; tot_t += d
        k = int(math.log2(max(b / 1e6, 1)))
        buckets.setdefault(k, [0, 0.0, 0.0]); buckets[k][0] += 1; buckets[k][1] += d; buckets[k][2] += b
        print(f"  {d/1e3:8.1f} {int(l[2]):5d} {b/1e6:9.1f} {b/d*1e-3:6.2f} {l[6]:>5s} {l[8]:>6s}")
    for k in sorted(buckets):
        n_, t_, b_ = buckets[k]
        print(f"{kind} launches of ~2^{k} MB: {n_:4d}, {t_/1e6:7.2f} ms, {b_/1e9:7.2f} GB, {b_/t_*1e-3:5.2f} TB/s")
    print(kind, f"total {tot_t/1e6:.2f} ms, {tot_b/1e9:.2f} GB, {tot_b/tot_t*1e-3:.2f} TB/s")
