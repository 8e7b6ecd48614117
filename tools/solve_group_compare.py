"""Grouped vs plain solve sweeps (tuning key 10): times and the error of x against a known solution."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, scipy.sparse as sp
from suitesparse_amd import engine, problems, symbolic
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if len(sys.argv) > 2:
    engine.set_tuning(11, int(sys.argv[2]))
p = problems.laplacian_3d(G)
sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
n = p["n"]
S = sp.csc_matrix((Sx, Si, Sp), shape=(n, n))
S = S + sp.tril(S, -1).T
L = engine.SupernodalFactor(sym)
L.set_matrix(Sp, Si, Sx)
assert L.numeric() == 0
for nrhs in ((1, 4) if len(sys.argv) > 2 else (1, 2, 4)):
    want = np.random.default_rng(nrhs).standard_normal((nrhs, n))
    B = np.stack([S @ w for w in want])
    for grp in (0, 1):
        engine.set_tuning(10, grp)
        X = torch.from_numpy(B.copy()).cuda().reshape(-1)
        for rep in range(2):
            X.copy_(torch.from_numpy(B).reshape(-1))
            L.solve_dev(X.data_ptr(), nrhs, n, lower=True); tf = L.stats()["solve_ms"]
            if rep == 1:
                Y = X.cpu().numpy().copy()
            L.solve_dev(X.data_ptr(), nrhs, n, lower=False); tb = L.stats()["solve_ms"]
        x = X.cpu().numpy().reshape(nrhs, n)
        err = [float(np.abs(x[j] - want[j]).max()) for j in range(nrhs)]
        if grp == 0:
            Y0 = Y
        else:
            print("  forward diff vs plain per rhs:", [float(np.abs(Y.reshape(nrhs, n)[j] - Y0.reshape(nrhs, n)[j]).max()) for j in range(nrhs)])
        print(f"grid {G} nrhs {nrhs} grouped {grp}: fwd {tf:.2f} bwd {tb:.2f} ms, max |x - want| per rhs {err}", flush=True)
if len(sys.argv) > 2:
    sys.exit(0)
# identical right-hand sides (all ones): residual of both modes
for nrhs in (4,):
    for grp in (0, 1, 0, 1):
        engine.set_tuning(10, grp)
        X = torch.ones(n * nrhs, dtype=torch.float64, device="cuda")
        L.solve_dev(X.data_ptr(), nrhs, n, lower=True)
        L.solve_dev(X.data_ptr(), nrhs, n, lower=False)
        x = X.cpu().numpy().reshape(nrhs, n)
        res = [float(np.abs(S @ x[j] - 1.0).max()) for j in range(nrhs)]
        print(f"ones: grid {G} nrhs {nrhs} grouped {grp}: max residual per rhs {res} max|x| {np.abs(x).max():.3e}", flush=True)
