#!/usr/bin/env python3
"""Times the supernodal forward/backward solves on a resident factor (GPU box)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--problem", default="lap3d"); ap.add_argument("--grid", type=int, default=100)
ap.add_argument("--nrhs", type=int, default=1); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
import torch
from suitesparse_amd import engine
p, name = bench.make_problem(a.problem, a.grid)
sym, (Sp, Si, Sx), _ = bench.analyze(p)
cplx = np.iscomplexobj(p["Ax"]); E = 2 if cplx else 1
L = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
L.set_matrix(Sp, Si, Sx); assert L.numeric() == 0
X = torch.ones(p["n"] * a.nrhs * E, dtype=torch.float64, device="cuda")
best = [1e9, 1e9]
for _ in range(a.reps):
    X.fill_(1.0)
    L.solve_dev(X.data_ptr(), a.nrhs, p["n"], lower=True); tf = L.stats()["solve_ms"]
    L.solve_dev(X.data_ptr(), a.nrhs, p["n"], lower=False); tb = L.stats()["solve_ms"]
    best = [min(best[0], tf), min(best[1], tb)]
bytes_alg = (2 * sym["xsize"] * 8 + 4 * p["n"] * a.nrhs * 8) * E
print(f"{name} nrhs={a.nrhs}: forward {best[0]:.3f} ms backward {best[1]:.3f} ms  -> {bytes_alg / ((best[0] + best[1]) * 1e-3) * 1e-9:.1f} GB/s algorithmic")
