#!/usr/bin/env python3
"""Where the factorization time goes: groups the MI355_STEP_LOG lines by kernel, mode and K bucket."""
import sys, collections
g = collections.defaultdict(lambda: [0, 0.0, 0.0, 0, 0.0])
tot = 0.0
for line in open(sys.argv[1]):
    w = line.split()
    d = dict(zip(w[1::2], w[2::2]))
    ms = float(d["ms"]); tot += ms
    if w[0] == "potrf":
        key = ("potrf", "-", "-")
        fl, nt, xfl = 0.0, int(d["count"]), 0.0
    else:
        k = int(d["kmax"])
        kb = "K<=64" if k <= 64 else "K<=128" if k <= 128 else "K<=256" if k <= 256 else "K<=512" if k <= 512 else "K>512"
        key = ("gemm128" if d["big"] == "1" else "gemm64", {"0": "scatter(update)", "1": "direct_sub(trail)", "2": "overwrite(trsm)"}[d["mode"]], kb)
        fl, nt, xfl = float(d["flops"]), int(d["ntiles"]), float(d.get("xflops", 0))
    e = g[key]; e[0] += 1; e[1] += ms; e[2] += fl; e[3] += nt; e[4] += xfl
print(f"{'kernel':8s} {'mode':18s} {'K':7s} {'launches':>8s} {'ms':>9s} {'%':>6s} {'TFLOP':>8s} {'TF/s':>7s} {'tiles/launch':>12s} {'tile TFLOP':>10s} {'tile TF/s':>9s}")
for key, e in sorted(g.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:8s} {key[1]:18s} {key[2]:7s} {e[0]:8d} {e[1]:9.2f} {100*e[1]/tot:6.2f} {e[2]/1e12:8.3f} {e[2]/max(e[1],1e-9)/1e9:7.2f} {e[3]/e[0]:12.0f} {e[4]/1e12:10.3f} {e[4]/max(e[1],1e-9)/1e9:9.2f}")
print(f"total {tot:.2f} ms (launches synchronised one by one)")
