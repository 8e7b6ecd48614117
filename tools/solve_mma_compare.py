"""Grouped vs plain MFMA solve sweeps (tuning key 10) for many right-hand sides: times and error vs a known solution."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, scipy.sparse as sp
from suitesparse_amd import engine, problems, symbolic
kind = sys.argv[1] if len(sys.argv) > 1 else "lap3d"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nrhs = int(sys.argv[3]) if len(sys.argv) > 3 else 64
p = {"lap3d": problems.laplacian_3d, "helm": problems.helmholtz_3d}[kind](G)
sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
n = p["n"]
cplx = np.iscomplexobj(Sx)
S = sp.csc_matrix((Sx, Si, Sp), shape=(n, n))
S = S + sp.tril(S, -1).conj().T
L = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
L.set_matrix(Sp, Si, Sx)
assert L.numeric() == 0
rng = np.random.default_rng(1)
want = rng.standard_normal((nrhs, n)) + (1j * rng.standard_normal((nrhs, n)) if cplx else 0)
B = np.stack([S @ w for w in want])
E = 2 if cplx else 1
for grp in (0, 1):
    engine.set_tuning(10, grp)
    Bt = torch.from_numpy(np.ascontiguousarray(B).view(np.float64).reshape(-1))
    X = Bt.cuda()
    for rep in range(2):
        X.copy_(Bt)
        torch.cuda.synchronize()
        L.solve_dev(X.data_ptr(), nrhs, n, lower=True); tf = L.stats()["solve_ms"]
        L.solve_dev(X.data_ptr(), nrhs, n, lower=False); tb = L.stats()["solve_ms"]
    x = X.cpu().numpy()
    x = x.view(np.complex128).reshape(nrhs, n) if cplx else x.reshape(nrhs, n)
    err = float(np.abs(x - want).max())
    print(f"{kind} {G} nrhs {nrhs} grouped {grp}: fwd {tf:.2f} bwd {tb:.2f} ms, max |x - want| {err:.3e}", flush=True)
