#!/usr/bin/env python3
"""MI355_STEP_LOG grouped by etree level: launches and time per kind."""
import sys, collections
lv = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0]))
for line in open(sys.argv[1]):
    w = line.split()
    d = dict(zip(w[1::2], w[2::2]))
    ms = float(d["ms"]); L = int(d["level"])
    if w[0] == "potrf":
        k, n = "potrf", int(d["count"])
    elif w[0] in ("pdiag", "ptrsm"):
        k, n = w[0], int(d["ntiles"])
    else:
        k = ("g128" if d["big"] == "1" else "g64") + {"0": "_upd", "1": "_trail", "2": "_trsm"}[d["mode"]]
        n = int(d["ntiles"])
    e = lv[L][k]; e[0] += 1; e[1] += ms; e[2] = max(e[2], n)
kinds = ["potrf", "pdiag", "ptrsm", "g64_trsm", "g64_trail", "g128_trail", "g64_upd", "g128_upd", "g128_trsm"]
print("level " + " ".join(f"{k:>22s}" for k in kinds) + "   total ms")
tot = 0
for L in sorted(lv):
    row = []
    s = 0
    for k in kinds:
        e = lv[L].get(k, [0, 0.0, 0]); s += e[1]
        row.append(f"{e[0]:5d} {e[1]:8.2f} {e[2]:7d}")
    tot += s
    print(f"{L:5d} " + " ".join(f"{r:>22s}" for r in row) + f"   {s:8.2f}")
print("total", round(tot, 2), "(launches, ms, max tiles per launch)")
