"""N ranks on ONE GPU over gloo (rehearsal of the multi-GPU path without a second GPU): every rank factorizes its
part (owner-computes top by default), then the partitioned solves recover a known solution.
  MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nproc-per-node N tools/rehearse_ranks.py [grid] [kind]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist
from suitesparse_amd import engine, multigpu, problems, symbolic

G = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kind = sys.argv[2] if len(sys.argv) > 2 else "lap3d"
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
p = {"lap3d": problems.laplacian_3d, "helm": problems.helmholtz_3d, "elast": problems.elasticity27}[kind](G)
sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
n = p["n"]
cplx = np.iscomplexobj(Sx)
L = engine.SupernodalFactor(sym, device=0, xtype=2 if cplx else 1)
L.set_matrix(Sp, Si, Sx)
run = multigpu.PartitionedFactor(L, sym, rank, world)
st = run.numeric()
S = sp.csc_matrix((Sx, Si, Sp), shape=(n, n))
S = S + sp.tril(S, -1).conj().T
nrhs = 3
rng = np.random.default_rng(0)
want = rng.standard_normal((nrhs, n)) + (1j * rng.standard_normal((nrhs, n)) if cplx else 0)
B = np.stack([S @ w for w in want])
X = torch.from_numpy(np.ascontiguousarray(B).view(np.float64).reshape(-1).copy()).cuda()
torch.cuda.synchronize()
run.lsolve(X, nrhs)
run.ltsolve(X, nrhs)
x = X.cpu().numpy()
x = x.view(np.complex128).reshape(nrhs, n) if cplx else x.reshape(nrhs, n)
err = float(np.abs(x - want).max())
ntop = int((run.part["owner"] < 0).sum())
print(f"RANK {rank}/{world}: status {st} distributed_top {run.dist} top supernodes {ntop} "
      f"broadcasts {len(run.plan) if run.dist else 0} max|x - want| {err:.2e}", flush=True)
assert st == 0 and err < 1e-9
dist.destroy_process_group()
