#!/usr/bin/env python3
"""bench.py -- numeric supernodal Cholesky factorization throughput on MI355X.

Metric (BASELINE.json): numeric factorization GFLOP/s (CHOLMOD's own flop count
Common->fl = sum_j ColCount[j]^2) of the 3D 7-point Laplacian, inputs (symbolic factor,
schedule, relative maps, the permuted matrix S) resident in HBM when the timed region
starts, the factor left resident in HBM when it ends.  A "step" = one complete numeric
factorization (zero Lx, assemble S, every level's POTRF/TRSM/trailing/update launches).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--grid G] [--problem lap3d|lap2d|elast]

N > 1 (launched by torch.distributed.run, one rank per GPU): disjoint etree subtrees are
mapped to ranks, their contributions to the shared top supernodes are summed with one RCCL
all-reduce, the top is factorized owner-computes over 1024-column blocks with one broadcast
per finished block (MI355_TOP=replicated: on every rank) -- "scaling": "strong", total work fixed.

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel (k_gemm128, FP64 MFMA);
`cpu_baseline` is the unchanged reference CPU worker (oracle/_ref) timed on the host cores
on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X vendor figure, dense FP64 matrix: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0


def make_problem(kind, grid):
    from suitesparse_amd import problems
    if kind == "lap3d":
        return problems.laplacian_3d(grid), f"3D 7-point Laplacian {grid}^3"
    if kind == "lap2d":
        return problems.laplacian_2d(grid), f"2D 5-point Laplacian {grid}^2"
    if kind == "elast":
        return problems.elasticity27(grid), f"3D 27-point 3-dof elasticity-like {grid}^3"
    if kind == "helm":
        return problems.helmholtz_3d(grid), f"complex Hermitian 3D Helmholtz-shifted Laplacian {grid}^3"
    raise ValueError(kind)


def metis_perm(p):
    """the permutation the REFERENCE's analysis picks with METIS (cholmod_l_analyze, CHOLMOD_METIS,
    CHOLMOD/Cholesky/cholmod_analyze.c:629-684) -- input data for --ordering metis, computed by oracle/_ref before
    anything is timed"""
    from oracle import refcholmod
    if not refcholmod.available():
        raise SystemExit("--ordering metis needs oracle/_ref (the reference's METIS)")
    r = refcholmod.RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze(ordering="metis")
    perm = r.symbolic()["Perm"]
    r.close()
    return perm


def analyze(p, ordering="nd"):
    from suitesparse_amd import symbolic
    t = time.time()
    perm = metis_perm(p) if ordering == "metis" else symbolic.nd_grid_order(p["shape"], p["dof"])
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], perm)
    Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
    return sym, (Sp, Si, Sx), time.time() - t


def cpu_baseline(kind, grid, ordering="nd", full_grid=None, full_fl=None, full_xsize=None):
    """reference CPU supernodal numeric (cholmod_l_super_numeric, host BLAS) on a bounded sample"""
    try:
        from oracle import refcholmod
        if not refcholmod.available():
            raise RuntimeError("oracle/_ref not present")
        p, name = make_problem(kind, grid)
        sym, (Sp, Si, Sx), _ = analyze(p, ordering)
        r = refcholmod.RefCholmod()
        r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
        inf = r.analyze(perm=sym["Perm"])           # CHOLMOD_GIVEN: same ordering as the GPU leg
        r.permuted()
        best = None
        reps = 1 if inf["fl"] > 2e12 else 2         # the default sample (lap3d 100^3) is ~30-50 s per factorization
        for _ in range(reps):
            st = r.factorize()
            assert st == 0
            t = r.info()["t_factor"]
            best = t if best is None else min(best, t)
        # the reference's supernodal solve (cholmod_l_super_lsolve + _ltsolve) on the same sample, nrhs = 1
        import numpy as np
        b = np.ones(p["n"], dtype=np.complex128 if np.iscomplexobj(p["Ax"]) else np.float64)
        tsolve = None
        for _ in range(2):
            t0 = time.time()
            y = r.super_solve(b, lower=True)
            r.super_solve(y, lower=False)
            dt = time.time() - t0
            tsolve = dt if tsolve is None else min(tsolve, dt)
        try:
            affinity = len(os.sched_getaffinity(0))
        except AttributeError:
            affinity = os.cpu_count()
        return {"value": round(inf["fl"] / best * 1e-9, 2), "unit": "GFLOP/s", "cores": int(r.nthreads()),
                "kind": "reference", "solve_ms_nrhs1": round(tsolve * 1e3, 2),
                "threads": {"omp_get_max_threads": int(r.nthreads()), "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
                            "MKL_NUM_THREADS": os.environ.get("MKL_NUM_THREADS"),
                            "MKL_THREADING_LAYER": os.environ.get("MKL_THREADING_LAYER"),
                            "cpus_in_affinity_mask": affinity, "os_cpu_count": os.cpu_count()},
                "L_ordering": inf["ordering_name"], "fl": inf["fl"], "factor_s": round(best, 3),
                "sample": f"{name}, same {'METIS' if ordering == 'metis' else 'nested-dissection'} ordering, fl={inf['fl']:.4g}, "
                          f"reference cholmod_l_super_numeric + MKL, best of {reps} = {best:.3f} s; "
                          f"cholmod_l_super_lsolve + _ltsolve, nrhs 1, best of 2 = {tsolve * 1e3:.1f} ms"
                          + (f"; a bounded sample, not the bench workload (grid {full_grid}): at this rate the reference "
                             f"would need about {full_fl / (inf['fl'] / best):.0f} s per factorization of the full size on these "
                             f"cores (fl {full_fl:.3g}, a factor of {full_xsize * 8e-9:.0f} GB on the host), beyond the few "
                             f"minutes a default run may take"
                             if full_grid is not None and full_grid != grid and full_fl and full_xsize else "")}
    except Exception as e:      # noqa: BLE001
        return {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "reference", "sample": f"unavailable: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--problem", default="lap3d")
    ap.add_argument("--grid", type=int, default=0, help="grid size (default: the largest named config that fits)")
    ap.add_argument("--cpu-grid", type=int, default=0, help="grid of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--solve", action="store_true", help="also time the supernodal solve")
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--tune", default="", help="experiments: key=value,... (keys of mi355_set_tuning, include/mi355_cholmod.h)")
    ap.add_argument("--ordering", default="nd", choices=["nd", "metis"],
                    help="nd: the engine's geometric nested dissection; metis: the permutation the reference's own "
                         "analysis picks (computed by oracle/_ref before the timed region, passed in as data)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # rehearsal switches (a 1-GPU box can run the N>1 path with every rank on cuda:0 and gloo in place of RCCL)
    if os.environ.get("MI355_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(os.environ.get("MI355_BENCH_BACKEND", "nccl"))
    from suitesparse_amd import engine
    tune_now = {}
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        engine.set_tuning(int(k), int(v))
        tune_now[int(k)] = int(v)

    default_grid = {"lap3d": 150, "lap2d": 2000, "elast": 80, "helm": 100}[args.problem]
    grid = args.grid or default_grid
    p, name = make_problem(args.problem, grid)
    sym, (Sp, Si, Sx), t_analyze = analyze(p, args.ordering)

    t0 = time.time()
    cplx = np.iscomplexobj(p["Ax"])
    L = engine.SupernodalFactor(sym, device=local_rank, xtype=2 if cplx else 1)
    L.set_matrix(Sp, Si, Sx)
    t_setup = time.time() - t0
    if world > 1:
        from suitesparse_amd import multigpu
        # the whole N > 1 driver runs behind the C ABI (mi355_factor_numeric_multi) on an RCCL communicator of the
        # engine's own; MI355_MULTI=python keeps the older torch.distributed driver of multigpu.py.  Whether the
        # communicator could be created is agreed on by all ranks, so that they take the same path.
        # MI355_MULTI=python (explicit request only) keeps the older torch.distributed driver of multigpu.py.  There is NO
        # silent fallback: a rank that cannot create the engine's communicator makes every rank stop with an error, so
        # that a scaling number is never a number of the other algorithm.
        comm = None
        if os.environ.get("MI355_MULTI", "c") != "python":
            if dist.get_backend() != "nccl":
                raise SystemExit("bench.py --gpus N runs the C driver on RCCL (backend nccl); for the torch.distributed "
                                 "rehearsal on another backend set MI355_MULTI=python")
            err = ""
            try:
                comm = engine.Comm.from_torch_distributed(local_rank)
            except Exception as e:      # noqa: BLE001
                err = f"[rank {rank}] RCCL communicator of the engine not created: {e}"
                print(err, file=sys.stderr, flush=True)
            ok = torch.tensor([1 if comm is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                dist.destroy_process_group()
                raise SystemExit(err or f"[rank {rank}] another rank could not create the engine's RCCL communicator")
        if comm is not None:
            runner = multigpu.PartitionedFactor(L, sym, rank, world, comm=comm)
        else:
            runner = multigpu.PartitionedFactor(L, sym, rank, world, allreduce=multigpu.PartitionedFactor._dist_allreduce,
                                                reduce=multigpu.PartitionedFactor._dist_reduce)

        def step():
            st = runner.numeric()
            assert st == 0, f"factorization failed: status {st}"
    else:
        def step():
            st = L.numeric()
            assert st == 0, f"factorization failed: status {st}"

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.time()
    for k in range(args.steps):
        # per-launch HIP events (for the roofline of k_gemm128) bracket the launches of the LAST timed step
        # only: they cost ~2 % of a step, and they are read back after the timed region
        if k == args.steps - 1:
            L.set_profile(True)
        step()
    sync()
    elapsed = time.time() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    st = L.stats()
    gemm_ms, gemm_launches = st["gemm_ms"], st["gemm_launches"]
    ms_per_step = elapsed / args.steps * 1e3
    gflops = sym["fl"] / (elapsed / args.steps) * 1e-9

    # roofline of the dominant kernel, k_gemm128 (FP64 MFMA): ALGORITHMIC flops per launch = Common->fl
    # x (share of the executed useful flops that k_gemm128's tasks carry) / launches, over the mean launch
    # duration measured with HIP events on the engine's stream during the last timed step
    f128, f64_, fpotrf = st["flops_gemm128"], st["flops_gemm64"], sym_potrf_flops(sym)
    exec_total = f128 + f64_ + fpotrf
    share = f128 / exec_total if exec_total > 0 else 0.0
    big_ms, big_n = st["gemm128_ms"], st["gemm128_launches"]
    roof = {"bound": "mfma", "kernel": "k_gemm128", "achieved": None, "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": None, "traffic": None}
    if big_ms > 0 and big_n > 0:
        # Common->fl counts a complex multiply-add as 2 flops; the MFMA pipe runs it as 8 real ones
        # (4 real products through the real view), so the roofline of a complex factor is priced in real flops
        # N > 1: this rank's launches carry its share of those flops (own subtrees + its blocks of the top)
        alg_per_launch = sym["fl"] * share * (st["flops_gemm128_here"] / f128) / big_n * (4 if cplx else 1)
        mean_s = big_ms * 1e-3 / big_n
        ach = alg_per_launch / mean_s * 1e-12
        roof.update(achieved=round(ach, 3), frac=round(ach / FP64_MFMA_PEAK_TFLOPS, 4),
                    launches=int(big_n), mean_launch_ms=round(mean_s * 1e3, 4),
                    algorithmic_flops_per_launch=alg_per_launch, share_of_fl=round(share, 4))
        if cplx:
            roof["flops_counted_as"] = "real FP64 flops = 4 x Common->fl (complex arithmetic)"
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload
        # (FETCH_SIZE x2 on gfx950, WRITE_SIZE; tools/pmc_traffic.py), when one exists
        pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.problem}{grid}.json")
        if world == 1 and args.ordering == "nd" and os.path.exists(pmc):
            try:
                t = json.load(open(pmc))
                key = [k for k in t if "k_gemm128" in k][0]
                roof["traffic"] = t[key]["bytes_per_launch"]
                roof["traffic_source"] = os.path.relpath(pmc, ROOT)
            except Exception:   # noqa: BLE001
                pass
    out = {
        "metric": "numeric factorization GFLOP/s (CHOLMOD flop count)", "value": round(gflops, 2),
        "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": name + (", the reference's METIS ordering" if args.ordering == "metis"
                                       else ", geometric nested dissection") + ", supernodal LL' (int64, double)",
                   "n": p["n"], "fl": sym["fl"], "lnz": sym["lnz"], "nsuper": sym["nsuper"],
                   "xsize": sym["xsize"], "levels": int(st["nlevels"]), "update_pairs": int(st["npairs"]),
                   "exec_flops": {"update": st["flops_update"], "panel": st["flops_panel"]},
                   "parallelism": "1 GPU" if world == 1 else
                   f"subtree-to-GPU x{world} + " + ("owner-computes top (1024-column blocks, broadcasts)"
                                                    if runner.dist else "replicated top") +
                   (", C driver on the engine's RCCL communicator" if runner.comm is not None
                    else ", torch.distributed driver (MI355_MULTI=python)"),
                   "multi_driver": None if world == 1 else
                   {"driver": "C (mi355_factor_numeric_multi)" if runner.comm is not None else "python (torch.distributed)",
                    "tuning": {"16_dist_lookahead": tune_now.get(16, 1), "17_dist_pieces": tune_now.get(17, 1),
                               "19_reduce_channel": tune_now.get(19, 0)},
                    "validated_on": "CPU stand-in for RCCL and gloo only (N > 1 has not run on xGMI before this line)"},
                   "analyze_s": round(t_analyze, 2), "setup_s": round(t_setup, 2),
                   "frac_fp64_mfma_peak": round(gflops * 1e-3 * (4 if cplx else 1)
                                                / (FP64_MFMA_PEAK_TFLOPS * world), 4),
                   "kernel_ms": {"gemm": round(gemm_ms, 3), "potrf": round(st["potrf_ms"], 3),
                                 "assembly": round(st["assembly_ms"], 3), "device_total": round(st["total_ms"], 3)}},
        "roofline": roof,
    }
    # the solve is outside the timed region: forward + backward sweep with X resident in HBM.  Default run
    # (N = 1): nrhs = 1 and 64, the two cases BASELINE.md names; --solve --nrhs K times one case.
    if world == 1 and (args.solve or not args.no_cpu):
        E = 2 if cplx else 1
        sol = []
        for nrhs in ([args.nrhs] if args.solve else [1, 64]):
            X = torch.ones(p["n"] * nrhs * E, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()            # the engine runs on its own stream
            L.solve_dev(X.data_ptr(), nrhs, p["n"], lower=True)     # untimed: builds the sweep's work lists
            L.solve_dev(X.data_ptr(), nrhs, p["n"], lower=False)
            X.fill_(1.0)
            torch.cuda.synchronize()
            L.solve_dev(X.data_ptr(), nrhs, p["n"], lower=True)
            t_f = L.stats()["solve_ms"]
            L.solve_dev(X.data_ptr(), nrhs, p["n"], lower=False)
            t_b = L.stats()["solve_ms"]
            # both sweeps in one call (what cholmod_solve2 with CHOLMOD_LDLt / CHOLMOD_A runs): the row-major image of many
            # right-hand sides stays in place between the sweeps -- one transpose in, one out
            B = torch.ones(p["n"] * nrhs * E, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            L.solve2_dev(1, B.data_ptr(), p["n"], X.data_ptr(), p["n"], nrhs)       # 1 = MI355_SYS_LDLt
            t_both = L.stats()["solve_ms"]
            del X, B
            bytes_alg = (2 * sym["xsize"] * 8 + 4 * p["n"] * nrhs * 8) * E
            sol.append({"nrhs": nrhs, "forward_ms": round(t_f, 3), "backward_ms": round(t_b, 3), "both_in_one_call_ms": round(t_both, 3),
                        "algorithmic_GBs": round(bytes_alg / ((t_f + t_b) * 1e-3) * 1e-9, 1),
                        "frac_hbm_peak": round(bytes_alg / ((t_f + t_b) * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4)})
        out["config"]["solve"] = sol[0] if args.solve else sol
    elif world > 1 and not args.no_cpu:
        # N > 1: sweeps over the own subtrees + ONE all-reduce of X per sweep + replicated top sweeps
        # (multigpu.PartitionedFactor.lsolve / ltsolve); wall clock, barrier on both sides, max over the ranks
        E = 2 if cplx else 1
        sol = []
        for nrhs in ([args.nrhs] if args.solve else [1, 64]):
            X = torch.ones(p["n"] * nrhs * E, dtype=torch.float64, device="cuda")
            runner.lsolve(X, nrhs)          # untimed: workspaces, communicator warm-up
            runner.ltsolve(X, nrhs)
            X.fill_(1.0)
            sync()
            t0 = time.time()
            runner.lsolve(X, nrhs)
            sync()
            t1 = time.time()
            runner.ltsolve(X, nrhs)
            sync()
            t2 = time.time()
            tt = torch.tensor([t1 - t0, t2 - t1], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            del X
            sol.append({"nrhs": nrhs, "forward_ms": round(float(tt[0]) * 1e3, 3),
                        "backward_ms": round(float(tt[1]) * 1e3, 3)})
        out["config"]["solve"] = sol[0] if args.solve else sol
    if rank == 0:
        if world == 1 and not args.no_cpu:
            try:
                out["config"]["mfma_f64_measured_tflops"] = round(engine.mfma_f64_rate(local_rank), 2)
            except Exception:   # noqa: BLE001
                pass
            # bounded sample, ~10-40 s of host work: lap3d 100^3 is fl 6.3e12 (8 GB of host factor)
            cg = args.cpu_grid or {"lap3d": 100, "lap2d": 2000, "elast": 48, "helm": 64}[args.problem]
            out["cpu_baseline"] = cpu_baseline(args.problem, min(cg, grid), args.ordering, full_grid=grid, full_fl=float(sym["fl"]),
                                              full_xsize=int(sym["xsize"]) * (2 if np.iscomplexobj(p["Ax"]) else 1))
        else:
            out["cpu_baseline"] = {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "reference",
                                   "sample": "measured at N=1 only"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def sym_potrf_flops(sym):
    """flops of the 64-block diagonal Cholesky steps (the part of the panel work not done by k_gemm64)"""
    nscol = np.diff(sym["Super"]).astype(np.float64)
    full = np.floor(nscol / 64)
    rem = nscol - 64 * full
    return float((full * 64.0 ** 3 / 3 + rem ** 3 / 3).sum())


if __name__ == "__main__":
    main()
