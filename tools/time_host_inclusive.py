"""PCIe-inclusive factorization times on the GPU box: mi355_factor_numeric (factor stays in HBM), the same followed
by mi355_factor_get_Lx (serial copy), and mi355_factor_numeric_host (copy under the factorization), best of 3 each.
python tools/time_host_inclusive.py [grid=100] [out.json|-] [lanes] [MB per lane]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suitesparse_amd import engine, problems, symbolic  # noqa: E402


def best(f, reps=3):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t)
    return min(ts)


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    out = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    if len(sys.argv) > 3:
        engine.set_tuning(20, int(sys.argv[3]))              # lanes of the copy
    if len(sys.argv) > 4:
        engine.set_tuning(21, int(sys.argv[4]))              # MB per staging buffer
    p = problems.laplacian_3d(grid)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    L = engine.SupernodalFactor(sym)
    L.set_matrix_perm(p["Ap"], p["Ai"], p["Ax"], sym["Perm"])
    assert L.numeric() == 0                                  # warm-up
    ref = L.Lx().copy()
    buf = np.zeros(L.xsize * L.E)                            # the caller's L->x: allocated and touched once, as CHOLMOD's is
    buf.fill(1.0)
    t_res = best(lambda: L.numeric())
    t_ser = best(lambda: (L.numeric(), L.Lx(out=buf)))
    got = {}

    def host():
        buf.fill(-1.0) if False else None
        got["st"], got["Lx"] = L.numeric_host(out=buf)
    t_host = best(host)
    assert got["st"] == 0 and float(np.max(np.abs(got["Lx"] - ref))) <= 1e-12 * float(np.max(np.abs(ref)))
    early = L.stats()["copy_early_entries"]
    res = dict(problem=f"lap3d {grid}", n=int(sym["n"]), fl=float(sym["fl"]), factor_GB=sym["xsize"] * 8e-9,
               resident_s=t_res, numeric_then_get_Lx_s=t_ser, numeric_host_s=t_host,
               gflops_resident=sym["fl"] / t_res * 1e-9, gflops_host_inclusive=sym["fl"] / t_host * 1e-9,
               chunks=int(len(L.host_copy_plan())), early_fraction=early / max(sym["xsize"], 1),
               lanes=int(sys.argv[3]) if len(sys.argv) > 3 else 4, lane_mb=int(sys.argv[4]) if len(sys.argv) > 4 else 32)
    print(json.dumps(res))
    if out:
        with open(out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
