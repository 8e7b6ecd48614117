"""Planning figures of the multi-GPU partition (CPU only, no numeric work): per-rank flops of phase 0 and of the
owner-computes top, panel-chain lengths, broadcast volume -- and a simple time model from the rates measured on one
MI355X.  python tools/partition_plan.py [grid] [problem]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suitesparse_amd import multigpu, problems, symbolic  # noqa: E402

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

GEMM_TFLOPS = 58.0        # big-update rate on one GPU (55-65 TFLOP/s at K >= 600)
STEP_US = 135.0           # one 64-column panel step (POTRF + TRSM + inner update), latency-bound
XGMI_GBS = 300.0          # what one rank moves per second in a ring all-reduce / a broadcast (a few links in use)


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    kind = sys.argv[2] if len(sys.argv) > 2 else "lap3d"
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d, "elast": problems.elasticity27}[kind](grid)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    emu = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    emu.emu_partition_plan.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i8p, i8p,
                                       C.c_int, C.c_int, f64p]
    a = [sym["n"], sym["nsuper"], sym["Super"], sym["Lpi"], sym["Lpx"], sym["Ls"], len(sym["Ls"]), sym["xsize"]]
    print(f"{kind} {grid}: n {sym['n']}  fl {sym['fl']:.4g}  xsize {sym['xsize']:.4g}")
    t1 = None
    for world in (1, 2, 4, 8):
        part = multigpu.partition(sym, world)
        owner = part["owner"]
        rows = []
        for r in range(world):
            out = np.zeros(16)
            loc = (owner == r).astype(np.int8) if world > 1 else np.ones(sym["nsuper"], dtype=np.int8)
            top = (owner < 0).astype(np.int8)
            emu.emu_partition_plan(*a, loc, top, r, world, out)
            rows.append(out.copy())
        R = np.array(rows)
        top_entries = sum(hi - lo for lo, hi in part["top_ranges"])
        ph0 = R[:, 0].max() / (GEMM_TFLOPS * 1e12) + R[:, 3].max() * 0 + 0.0
        chain0 = R[:, 3].max() * STEP_US * 1e-6 * 0.25          # low levels: steps are short launches, not full chains
        # reduce of every top block to its owner: a rank sends each block at most once (the all-reduce it replaced:
        # 2 (N-1)/N times the volume)
        allred = top_entries * 8 * (world - 1) / max(world, 1) / (XGMI_GBS * 1e9) if world > 1 else 0.0
        topw = R[:, 1].max() / (GEMM_TFLOPS * 1e12)
        chain1 = R[0, 4] * STEP_US * 1e-6 if world > 1 else 0.0
        bcast = R[0, 6] * 8 / (XGMI_GBS * 1e9) if world > 1 else 0.0
        total = ph0 + chain0 + allred + topw + chain1 + bcast
        if world == 1:
            t1 = total
        print(f"N={world}: phase0 flops max/mean {R[:, 0].max() / max(R[:, 0].mean(), 1):.3f} ({R[:, 0].max():.3g})  "
              f"top share max/mean {R[:, 1].max() / max(R[:, 1].mean(), 1):.3f} ({R[:, 1].max():.3g} of {R[0, 2]:.3g})  "
              f"top chain {int(R[0, 4])} steps  bcasts {int(R[0, 5])} = {R[0, 6] * 8e-9:.1f} GB  reduced to owners {top_entries * 8e-9:.1f} GB")
        print(f"      model: phase0 {ph0 * 1e3:.0f} + {chain0 * 1e3:.0f} ms, reduce {allred * 1e3:.0f} ms, top gemm {topw * 1e3:.0f} ms, "
              f"top chain {chain1 * 1e3:.0f} ms, broadcasts {bcast * 1e3:.0f} ms -> {total * 1e3:.0f} ms"
              + (f"  speedup {t1 / total:.2f}, efficiency {t1 / total / world:.2f}" if world > 1 else ""))


if __name__ == "__main__":
    main()
