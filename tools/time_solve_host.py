"""cholmod_solve2-shaped call with HOST right-hand sides on the GPU box (config 5: complex 100^3, 64 right-hand sides =
1 GB down and 1 GB up): mi355_solve_x through one plain copy each way (tuning key 22 huge) and through the lanes
(default), best of 3; the device-resident sweeps beside it.   python tools/time_solve_host.py [grid=100] [nrhs=64] [out.json]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suitesparse_amd import engine, problems, symbolic  # noqa: E402


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    out = sys.argv[3] if len(sys.argv) > 3 else None
    p = problems.helmholtz_3d(grid)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    n = int(sym["n"])
    L = engine.SupernodalFactor(sym, xtype=2)
    L.set_matrix_perm(p["Ap"], p["Ai"], p["Ax"], sym["Perm"])
    assert L.numeric() == 0
    L.set_perm(sym["Perm"])
    rng = np.random.default_rng(0)
    B = (rng.standard_normal((nrhs, n)) + 1j * rng.standard_normal((nrhs, n)))
    X = np.zeros_like(B)
    X.real.fill(1.0)                                           # touched once, like a caller's array
    f = engine.lib().mi355_solve_x
    bf, xf = B.view(np.float64).reshape(-1), X.view(np.float64).reshape(-1)

    def call():
        rc = f(L._h, 0, nrhs, 2, bf, n, xf, n)
        assert rc == 0, rc
    res = {}
    for name, kb in (("plain_copy_s", 1 << 30), ("lanes_s", 4096)):
        engine.set_tuning(22, kb)
        call()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
        res[name] = min(ts)
        res[name.replace("_s", "_X")] = X.copy() if name == "plain_copy_s" else None
    ref = res.pop("plain_copy_X"); res.pop("lanes_X")
    assert float(np.max(np.abs(X - ref))) <= 1e-12 * float(np.max(np.abs(ref)))
    st = L.stats()
    res.update(problem=f"helm {grid}", n=n, nrhs=nrhs, rhs_GB=B.nbytes * 1e-9, sweeps_ms_last_call=st["solve_ms"])
    print(json.dumps(res))
    if out:
        with open(out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
