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
STEP_US = 77.0            # one 64-column panel step, latency-bound: a quarter of a panel group (k_panel_diag 154 us + k_panel_trsm_reg
                          # 152 us per 256 columns, profiles/r03_lap3d150_kernel_stats.txt; 135 us with the per-block chain of rounds 1 and 2)
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
    # a handle of the CPU build of the engine (test infrastructure; no numeric work is done with it): the real schedules,
    # reduce plan and look-ahead counts of mi355_factor_partition
    handle = None
    try:
        from suitesparse_amd import engine
        engine.LIB_PATH, engine._LIB = os.path.join(ROOT, "tests", "helpers", "libmi355_cpuemu.so"), None
        symbolic._ready = False
        handle = engine.SupernodalFactor(sym)
    except Exception as e:      # noqa: BLE001
        print(f"(no CPU build of the engine: {e}; round-2 lines left out)")
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
        print(f"      model (round 1 schedule, everything serial): phase0 {ph0 * 1e3:.0f} + {chain0 * 1e3:.0f} ms, reduce {allred * 1e3:.0f} ms, top gemm {topw * 1e3:.0f} ms, "
              f"top chain {chain1 * 1e3:.0f} ms, broadcasts {bcast * 1e3:.0f} ms -> {total * 1e3:.0f} ms"
              + (f"  speedup {t1 / total:.2f}, efficiency {t1 / total / world:.2f}" if world > 1 else ""))
        if world > 1 and handle is not None:
            # round 2 schedule: (i) the reductions are issued as soon as a rank's last contribution to a block is
            # enqueued -- the part of the volume whose ready point lies before the end of the own-subtree schedule can
            # hide behind the subtree work that follows it (steps weighted equally: a crude proxy for time);
            # (ii) look-ahead in the top: per block, chain + max(broadcast, far update) instead of their sum
            owner_, cnt = handle.partition(0, world, 2, True)
            plan = handle.reduce_plan()
            own = max(cnt["own_steps"], 1)
            vol = plan[:, 2].astype(float)
            hidden = float((vol * (1.0 - plan[:, 0] / own)).sum() / max(vol.sum(), 1.0))
            red2 = max(0.0, allred - hidden * (ph0 + chain0))
            top2 = chain1 + max(bcast, topw)
            total2 = ph0 + chain0 + red2 + top2
            print(f"      model (round 2 schedule): {cnt['reductions']} reductions, {100 * float(vol[plan[:, 0] == 0].sum() / max(vol.sum(), 1)):.0f} % of their volume "
                  f"ready at once, {100 * float(vol[plan[:, 0] >= own].sum() / max(vol.sum(), 1)):.0f} % only after the last subtree step; reduce not hidden {red2 * 1e3:.0f} ms; "
                  f"top with look-ahead ({cnt['lookahead_bcasts']} of {cnt['top_blocks']} broadcasts have launches behind them) {top2 * 1e3:.0f} ms "
                  f"-> {total2 * 1e3:.0f} ms  speedup {t1 / total2:.2f}, efficiency {t1 / total2 / world:.2f}, "
                  f"{sym['fl'] / total2 / world / 78.6e12 * 100:.0f} % of {world} x 78.6 TF")
            # (ii b) the reduce channel (tuning key 19, built in round 2, default off until it has run on xGMI): the sums
            # of the top supernodes ABOVE the lowest top level travel on a communicator and stream of their own while
            # the lowest top level -- whose ranges must have arrived first -- runs its panel phase (chains + inner and
            # outer trailing updates of those supernodes; their scatter updates into the higher top supernodes come at
            # the end of the level and wait for the sums).  Hidden = min(time of that panel phase, volume above / rate).
            lev = np.zeros(sym["nsuper"], dtype=np.int64)
            Super, Lpi, Ls = sym["Super"], sym["Lpi"], sym["Ls"]
            colmap = np.repeat(np.arange(sym["nsuper"]), np.diff(Super))
            nscol = np.diff(Super).astype(float); nsrow = np.diff(Lpi).astype(float)
            for s_ in range(sym["nsuper"]):
                if nsrow[s_] > nscol[s_]:
                    par = colmap[Ls[Lpi[s_] + int(nscol[s_])]]
                    lev[par] = max(lev[par], lev[s_] + 1)
            top_s = np.nonzero(owner < 0)[0]
            if len(top_s):
                lo_lev = lev[top_s].min()
                low = top_s[lev[top_s] == lo_lev]
                high = top_s[lev[top_s] > lo_lev]
                vol_high = float((nsrow[high] * nscol[high]).sum()) * 8
                below = nsrow[low] - nscol[low]
                # panel phase of the lowest top level: dense partial factorization of its supernodes, shared by the owners
                fl_low = float((nscol[low] ** 3 / 3 + below * nscol[low] ** 2).sum())
                chain_low = float((np.ceil(nscol[low] / 64)).max()) * STEP_US * 1e-6
                t_low = fl_low / world / (GEMM_TFLOPS * 1e12) + chain_low
                t_high = vol_high * (world - 1) / world / (XGMI_GBS * 1e9)
                hid = min(t_low, t_high, red2)
                total3 = total2 - hid
                print(f"      model (round 2 schedule + reduce channel, key 19): lowest top level = {len(low)} supernodes (panel phase {t_low * 1e3:.0f} ms), "
                      f"{vol_high * 1e-9:.1f} of {top_entries * 8e-9:.1f} GB of sums belong to the levels above it ({t_high * 1e3:.0f} ms) -> {hid * 1e3:.0f} ms of the "
                      f"reduction hidden -> {total3 * 1e3:.0f} ms  speedup {t1 / total3:.2f}, efficiency {t1 / total3 / world:.2f}, "
                      f"{sym['fl'] / total3 / world / 78.6e12 * 100:.0f} % of {world} x 78.6 TF")
            # (iii) pieces (tuning key 17): what still has to travel when a block's chain ends is its last piece only; the
            # chain -> broadcast -> update of the next block -> next chain path of the top carries that, not the whole block
            tail = bcast * (cnt["top_blocks"] - cnt["top_pieces"]) / max(cnt["top_blocks"], 1)
            print(f"      pieces: {cnt['top_pieces']} of {cnt['top_blocks']} ranges leave before their block is complete; broadcast left on the "
                  f"chain -> broadcast -> next update path {bcast * 1e3:.0f} -> about {tail * 1e3:.0f} ms (equal-sized pieces assumed)")


if __name__ == "__main__":
    main()
