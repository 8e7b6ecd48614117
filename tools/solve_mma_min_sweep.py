"""Solve time vs tuning key 9 (fewest right-hand sides on the MFMA kernels)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from suitesparse_amd import engine, problems, symbolic
kind, G = sys.argv[1], int(sys.argv[2])
p = {"lap3d": problems.laplacian_3d, "helm": problems.helmholtz_3d}[kind](G)
sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
n = p["n"]
cplx = np.iscomplexobj(Sx)
E = 2 if cplx else 1
L = engine.SupernodalFactor(sym, xtype=E)
L.set_matrix(Sp, Si, Sx)
assert L.numeric() == 0
if len(sys.argv) > 4:
    engine.set_tuning(int(sys.argv[5]) if len(sys.argv) > 5 else 13, int(sys.argv[4]))
for nrhs in [int(a) for a in sys.argv[3].split(",")]:
    out = []
    ref = None
    for mm in (1, 1000):
        engine.set_tuning(9, mm)
        X = torch.ones(n * nrhs * E, dtype=torch.float64, device="cuda")
        for rep in range(2):
            X.fill_(1.0); torch.cuda.synchronize()
            L.solve_dev(X.data_ptr(), nrhs, n, lower=True); tf = L.stats()["solve_ms"]
            L.solve_dev(X.data_ptr(), nrhs, n, lower=False); tb = L.stats()["solve_ms"]
        x = X.cpu().numpy()
        if ref is None:
            ref = x
        out.append(f"{'mfma' if mm == 1 else 'scalar'}: fwd {tf:.2f} bwd {tb:.2f} sum {tf + tb:.2f}")
    print(f"{kind} {G} nrhs {nrhs}: " + " | ".join(out) + f" | rel diff {np.abs(x - ref).max() / np.abs(ref).max():.1e}", flush=True)
