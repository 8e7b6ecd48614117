"""CPU tests of the multi-GPU host logic (suitesparse_amd/multigpu.py): the subtree partition,
the two-phase schedule and the top-panel reduction, executed with the CPU schedule emulator --
sequentially for 1..8 simulated ranks, and for real with torch.distributed (gloo, world_size 2)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, sym_of, relerr
from suitesparse_amd import multigpu

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")


class EmuFactor:
    """same interface as engine.SupernodalFactor's multi-GPU part, on the CPU emulator"""

    def __init__(self, g):
        self.g = g
        self.E = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
        self.E.emu_numeric_phase.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64,
                                             i64p, i64p, f64p, C.c_double, i8p, i8p, C.c_int, f64p,
                                             C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64, f64p,
                                             C.POINTER(C.c_int64)]
        self.E.emu_solve_phase.argtypes = [C.c_int, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64,
                                           C.c_int64, f64p, f64p, C.c_int64, C.c_void_p, C.c_int64, i8p, C.c_int64]
        self.E.emu_dinv_size.restype = C.c_int64
        self.E.emu_dinv_size.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64]
        self.E.emu_top_plan.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i8p,
                                        C.c_int, C.c_int, C.c_int64, C.c_int64, i64p, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self.E.emu_top_run.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i8p,
                                       C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, f64p, f64p,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self.Lx = np.zeros(g["xsize"])
        self.Dinv = np.zeros(self.E.emu_dinv_size(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"],
                                                  g["ssize"], g["xsize"]) + 1)
        self.dinv_local_end = 0

    def set_partition(self, local, top, asm):
        self.local, self.top, self.asm = (np.ascontiguousarray(m, dtype=np.int8) for m in (local, top, asm))

    def numeric_phase(self, phase, beta=0.0):
        g = self.g
        bad, info, end = C.c_int64(-1), C.c_int64(0), C.c_int64(0)
        rc = self.E.emu_numeric_phase(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"],
                                      g["xsize"], g["Sp"], g["Si"], np.ascontiguousarray(g["Sx"], dtype=np.float64),
                                      beta, self.local if phase == 0 else self.top, self.asm, int(phase == 0),
                                      self.Lx, C.byref(bad), C.byref(info),
                                      0 if phase == 0 else self.dinv_local_end, self.Dinv, C.byref(end))
        assert rc == 0
        if phase == 0:
            self.dinv_local_end = int(end.value)
        return int(bad.value), int(info.value)

    def numeric(self, beta=0.0, quick=False):
        """mi355_factor_numeric: the whole factor with the undistributed schedule, incl. the not-posdef protocol"""
        g = self.g
        self.E.emu_super_numeric.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i64p,
                                             i64p, f64p, C.c_double, C.c_int, f64p, f64p, i64p, C.POINTER(C.c_int), i64p]
        minor, st, stats = np.zeros(1, dtype=np.int64), C.c_int(0), np.zeros(8, dtype=np.int64)
        D = np.zeros(self.Dinv.size + 1)
        rc = self.E.emu_super_numeric(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"],
                                      g["xsize"], g["Sp"], g["Si"], np.ascontiguousarray(g["Sx"], dtype=np.float64),
                                      beta, int(bool(quick)), self.Lx, D, minor, C.byref(st), stats)
        assert rc == 0
        self.minor = int(minor[0])
        return st.value

    def set_top_distribution(self, rank, world):
        g = self.g
        self.rank, self.world = rank, world
        # the Dinv slots of the own subtrees end where phase 0's schedule ends: known before phase 0 runs
        loc = np.ascontiguousarray(self.local)
        self.E.emu_dinv_size_masked.restype = C.c_int64
        self.E.emu_dinv_size_masked.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i8p]
        self.dinv_local_end = int(self.E.emu_dinv_size_masked(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"],
                                                              g["Ls"], g["ssize"], g["xsize"], loc))
        plan = np.zeros((4096, 4), dtype=np.int64)
        cnt, nsteps, dend = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = self.E.emu_top_plan(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"],
                                 self.top, rank, world, self.dinv_local_end, 4096, plan.reshape(-1), C.byref(cnt),
                                 C.byref(nsteps), C.byref(dend))
        assert rc == 0, rc
        self.plan, self.nsteps = plan[:cnt.value].copy(), int(nsteps.value)
        self.first_fail = (-1, 0)
        return self.plan, (self.dinv_local_end, int(dend.value))

    def top_run(self, seg):
        g = self.g
        nb = len(self.plan)
        if seg == 0:
            self.first_fail = (-1, 0)
            self.Dinv[self.dinv_local_end:] = 0.0
        step0 = 0 if seg == 0 else int(self.plan[seg - 1, 0])
        step1 = self.nsteps if seg == nb else int(self.plan[seg, 0])
        bad, info = C.c_int64(-1), C.c_int64(0)
        rc = self.E.emu_top_run(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"],
                                self.top, self.rank, self.world, self.dinv_local_end, step0, step1, self.Lx, self.Dinv,
                                C.byref(bad), C.byref(info))
        assert rc == 0, rc
        if bad.value >= 0 and (self.first_fail[0] < 0 or bad.value < self.first_fail[0]):
            self.first_fail = (int(bad.value), int(info.value))
        return self.first_fail if seg == nb else (-1, 0)

    def dinv_tensors(self, ranges):
        import torch
        t = torch.from_numpy(self.Dinv)
        return [t[lo:hi] for lo, hi in ranges]

    def solve_phase_dev(self, part, lower, X_ptr, nrhs, d):
        g = self.g
        rc = self.E.emu_solve_phase(int(lower), g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"],
                                    g["ssize"], g["xsize"], self.Lx, self.Dinv, nrhs, C.c_void_p(X_ptr), d,
                                    self.local if part == 0 else self.top, 0 if part == 0 else self.dinv_local_end)
        assert rc == 0

    def keep_rows_dev(self, X_ptr, nrhs, d):
        g = self.g
        X = np.ctypeslib.as_array(C.cast(C.c_void_p(X_ptr), C.POINTER(C.c_double)), shape=(nrhs, d))
        supermap = np.repeat(np.arange(g["nsuper"]), np.diff(g["Super"]))
        X[:, :g["n"]][:, self.asm[supermap] == 0] = 0.0

    def top_tensors(self, ranges):
        import torch
        t = torch.from_numpy(self.Lx)
        return [t[lo:hi] for lo, hi in ranges]


def problem(kind="lap3d", size=12):
    from suitesparse_amd import problems, symbolic
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d, "elast": problems.elasticity27,
         "helm": problems.helmholtz_3d}[kind](size)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
    g = dict(sym, Sp=Sp, Si=Si, Sx=Sx, beta=0.0)
    g["ssize"] = len(sym["Ls"])
    return g


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_partition_is_a_partition(world):
    g = problem("lap3d", 14)
    part = multigpu.partition(g, world)
    owner = part["owner"]
    parent, work = multigpu.supernode_tree(g)
    assert owner.shape == (g["nsuper"],) and owner.max() < world
    # top is closed under "parent": an owned supernode's ancestors are owned by the same rank or top
    for s in range(g["nsuper"]):
        p = parent[s]
        if p >= 0:
            assert owner[p] == -1 or owner[p] == owner[s]
            if owner[s] == -1:
                assert owner[p] == -1
    assert abs(part["top_work"] + sum(part["rank_work"]) - part["total_work"]) <= 1e-6 * part["total_work"]
    if world > 1:
        assert max(part["rank_work"]) <= 0.75 * part["total_work"]
        for lo, hi in part["top_ranges"]:
            assert 0 <= lo < hi <= g["xsize"]


@pytest.mark.parametrize("kind,size,world", [("lap3d", 12, 2), ("lap3d", 12, 4), ("lap2d", 40, 8), ("elast", 5, 3)])
def test_phases_with_simulated_ranks_match_oracle(kind, size, world):
    from oracle import port
    g = problem(kind, size)
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert so == 0
    facs = [EmuFactor(g) for _ in range(world)]

    class Shared:            # the all-reduce of the simulated ranks: sum over all copies, once
        done = False

    runners = []

    def allreduce_factory(r):
        def ar(tensors):
            pass             # replaced below: the sum needs every rank's phase 0 first
        return ar
    runners = [multigpu.PartitionedFactor(f, g, r, world, allreduce=allreduce_factory(r)) for r, f in enumerate(facs)]
    # phase 0 everywhere
    for rn in runners:
        assert rn.f.numeric_phase(0) == (-1, 0)
    # reduction
    ranges = runners[0].part["top_ranges"]
    for lo, hi in ranges:
        tot = sum(f.Lx[lo:hi] for f in facs)
        for f in facs:
            f.Lx[lo:hi] = tot
    # phase 1 everywhere
    for rn in runners:
        assert rn.f.numeric_phase(1) == (-1, 0)
    owner = runners[0].part["owner"]
    Lpx = g["Lpx"]
    for r, f in enumerate(facs):
        for s in range(g["nsuper"]):
            if owner[s] == r or owner[s] == -1:
                a, b = int(Lpx[s]), int(Lpx[s + 1])
                assert relerr(f.Lx[a:b], Lo[a:b]) < 1e-11, (r, s)


@pytest.mark.parametrize("kind,size,world", [("lap3d", 30, 4), ("lap3d", 24, 3), ("helm", 14, 2), ("lap2d", 120, 8)])
def test_every_task_of_a_partition_stays_inside_its_arrays(kind, size, world):
    """check_schedule_bounds (what the engine runs before any schedule reaches the device) on the four schedules
    of every rank -- whole factor, own subtrees, top, owner-computes top -- and on their solve work lists"""
    from suitesparse_amd import problems, symbolic
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d, "helm": problems.helmholtz_3d}[kind](size)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    emu = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    emu.emu_validate_partition.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, C.c_int,
                                           i8p, i8p, C.c_int, C.c_int, C.c_char_p, C.c_int]
    owner = multigpu.partition(sym, world)["owner"]
    for r in range(world):
        msg = C.create_string_buffer(300)
        rc = emu.emu_validate_partition(sym["n"], sym["nsuper"], sym["Super"], sym["Lpi"], sym["Lpx"], sym["Ls"],
                                        len(sym["Ls"]), sym["xsize"], int(kind == "helm"),
                                        (owner == r).astype(np.int8), (owner < 0).astype(np.int8), r, world, msg, 300)
        assert rc == 0, (r, rc, msg.value.decode())


@pytest.mark.parametrize("where", ["subtree", "top", "top_and_later_subtree"])
@pytest.mark.parametrize("quick", [False, True])
def test_not_posdef_on_a_partition_replays_the_reference_protocol(where, quick):
    """a bad pivot on N ranks: status, L->minor AND the factor (zero tail, refactorized leading columns of the
    failing supernode, t_cholmod_super_numeric_worker.c:1084-1146) are the oracle's on every rank -- also when
    a subtree with a larger index fails in phase 0 and a top supernode with a smaller index fails as well"""
    from oracle import port
    g = dict(problem("lap3d", 12))
    world = 3
    owner = multigpu.partition(g, world)["owner"]
    Super, Sp = g["Super"], g["Sp"]
    tops = np.nonzero(owner < 0)[0]
    assert tops.size >= 2
    s_top = int(tops[0])                                  # the lowest top supernode
    later = [s for s in range(s_top + 1, g["nsuper"]) if owner[s] >= 0]
    Sx = g["Sx"].copy()

    def spoil(s, col_in):
        col = int(Super[s]) + min(col_in, int(Super[s + 1] - Super[s]) - 1)
        Sx[Sp[col]] = -5.0                               # the diagonal entry comes first in a column of tril(S)
    if where == "subtree":
        spoil(int(np.nonzero(owner == 1)[0][3]), 0)
    elif where == "top":
        spoil(s_top, 2)
    else:
        if not later:
            pytest.skip("no subtree after the first top supernode in this partition")
        spoil(s_top, 2); spoil(later[-1], 0)
    g["Sx"] = Sx
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"], quick_return=quick)
    assert so == 1
    facs = [EmuFactor(g) for _ in range(world)]
    # sequential simulation: the ranks cannot exchange (bad, info) mid-call, so drive the phases by hand as
    # PartitionedFactor.numeric does, with the agreement computed over all ranks
    runners = [multigpu.PartitionedFactor(f, g, r, world, allreduce=lambda t: None, distribute_top=False)
               for r, f in enumerate(facs)]
    res0 = [rn.f.numeric_phase(0) for rn in runners]
    bad0 = min([b for b, _ in res0 if b >= 0], default=-1)
    if bad0 < 0:
        for lo, hi in runners[0].part["top_ranges"]:
            tot = sum(f.Lx[lo:hi] for f in facs)
            for f in facs:
                f.Lx[lo:hi] = tot
        res1 = [rn.f.numeric_phase(1) for rn in runners]
        assert min([b for b, _ in res1 if b >= 0], default=-1) >= 0
    for rn in runners:
        st = rn._replay(0.0, quick)
        assert (st, rn.minor) == (so, mo)
        assert relerr(rn.f.Lx, Lo) < 1e-11
        assert ((rn.f.Lx == 0) == (Lo == 0)).all()


@pytest.mark.parametrize("kind,size,world,ob2,tiny_max", [("lap3d", 14, 4, 128, 0), ("lap3d", 12, 3, 1024, 0),
                                                          ("lap2d", 40, 8, 64, 4096), ("elast", 5, 2, 64, 0),
                                                          ("lap3d", 12, 2, 64, 1 << 16)])
def test_reduce_to_owner_covers_the_top_panels(kind, size, world, ob2, tiny_max):
    """the blocks of the broadcast plan tile the top panels exactly, so reducing every block to its owner replaces the
    all-reduce: simulated ranks, the reduction done by hand block by block (non-owners keep their partial sums, as
    after a real reduce), then the distributed top; every rank must end with the oracle's factor"""
    from oracle import port
    g = problem(kind, size)
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    emu = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    emu.emu_set_tuning(2, min(64, ob2)); emu.emu_set_tuning(7, ob2)
    emu.emu_set_tuning(14, tiny_max)            # the tiny launch classes (order lists, fused TRSM) under a partition too
    try:
        facs = [EmuFactor(g) for _ in range(world)]
        calls = []
        runners = [multigpu.PartitionedFactor(f, g, r, world, distribute_top=True, min_subtrees_per_rank=1,
                                              reduce=lambda ts, roots: calls.append(len(ts)))
                   for r, f in enumerate(facs)]
        assert all(rn.dist and rn.reduce is not None for rn in runners)
        covered = sum(int(c) for c in runners[0].plan[:, 2])
        assert covered == sum(hi - lo for lo, hi in runners[0].part["top_ranges"])
        for rn in runners:
            assert rn.f.numeric_phase(0) == (-1, 0)
        for k, (step, lo, cnt, root) in enumerate(runners[0].plan):      # what dist.reduce(dst=root) leaves behind
            tot = sum(rn.bviews[k].clone() for rn in runners)
            runners[int(root)].bviews[k][...] = tot
        res = run_distributed_lockstep(runners, facs)
        assert all(r == (-1, 0) for r in res)
        owner = runners[0].part["owner"]
        for r, f in enumerate(facs):
            for s in range(g["nsuper"]):
                if owner[s] == r or owner[s] == -1:
                    a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
                    nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
                    got, want = f.Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
                    low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
                    assert relerr(got[low], want[low]) < 1e-11, (r, s)
    finally:
        emu.emu_set_tuning(2, 256); emu.emu_set_tuning(7, 1024); emu.emu_set_tuning(14, 2048)


def run_distributed_lockstep(runners, facs):
    """the owner-computes top with simulated ranks: every rank runs segment k, then block k travels"""
    plans = [rn.plan for rn in runners]
    for pl in plans[1:]:                                  # same broadcasts in the same order on every rank
        assert pl.shape == plans[0].shape and (pl[:, 1:] == plans[0][:, 1:]).all()
    for k in range(len(plans[0])):
        for rn in runners:
            rn.f.top_run(k)
        root = int(plans[0][k, 3])
        for rn in runners:
            if rn.rank != root:
                rn.bviews[k][...] = runners[root].bviews[k]
    res = [rn.f.top_run(len(plans[0])) for rn in runners]
    for rn in runners:                                    # block inverses: sum over the ranks
        if rn.dinv_views:
            tot = sum(r2.dinv_views[0].clone() for r2 in runners)
            break
    for rn in runners:
        if rn.dinv_views:
            rn.dinv_views[0][...] = tot
    return res


@pytest.mark.parametrize("kind,size,world,ob2", [("lap3d", 12, 2, 64), ("lap3d", 14, 4, 128), ("lap3d", 12, 3, 1024),
                                                 ("lap2d", 40, 8, 64), ("elast", 5, 3, 64)])
def test_distributed_top_with_simulated_ranks(kind, size, world, ob2):
    """owner-computes top: every rank ends with the oracle's factor in its own subtrees and in the whole top,
    and the solves on the partitioned factor work from the summed block inverses"""
    import torch
    from oracle import port
    g = problem(kind, size)
    n = g["n"]
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    emu = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    emu.emu_set_tuning(2, min(64, ob2)); emu.emu_set_tuning(7, ob2)      # small outer blocks: many owners
    try:
        facs = [EmuFactor(g) for _ in range(world)]
        runners = [multigpu.PartitionedFactor(f, g, r, world, allreduce=lambda t: None, distribute_top=True,
                                              min_subtrees_per_rank=1) for r, f in enumerate(facs)]
        assert all(rn.dist for rn in runners)
        roots = set(int(r) for r in runners[0].plan[:, 3])
        if ob2 <= 128:
            assert len(roots) > 1                           # the top really is spread over ranks
        for rn in runners:
            assert rn.f.numeric_phase(0) == (-1, 0)
        for lo, hi in runners[0].part["top_ranges"]:
            tot = sum(f.Lx[lo:hi] for f in facs)
            for f in facs:
                f.Lx[lo:hi] = tot
        res = run_distributed_lockstep(runners, facs)
        assert all(r == (-1, 0) for r in res)
        owner = runners[0].part["owner"]
        Lpx = g["Lpx"]
        for r, f in enumerate(facs):
            for s in range(g["nsuper"]):
                if owner[s] == r or owner[s] == -1:
                    a, b = int(Lpx[s]), int(Lpx[s + 1])
                    nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
                    got, want = f.Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
                    low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T   # entries on or below the diagonal
                    assert relerr(got[low], want[low]) < 1e-11, (r, s)
        # solves
        B = np.random.default_rng(1).standard_normal((2, n))
        want_l = port.super_solve(g, Lo, B.T.copy(), lower=True).T
        want = port.super_solve(g, Lo, want_l.T.copy(), lower=False).T
        Xs = [torch.from_numpy(B.copy()) for _ in range(world)]

        def summed():
            tot = sum(x.numpy() for x in Xs)
            for x in Xs:
                x.numpy()[...] = tot
        for rn, x in zip(runners, Xs):
            rn.f.keep_rows_dev(x.data_ptr(), 2, n)
            rn.f.solve_phase_dev(0, True, x.data_ptr(), 2, n)
        summed()
        for rn, x in zip(runners, Xs):
            rn.f.solve_phase_dev(1, True, x.data_ptr(), 2, n)
            rn.f.solve_phase_dev(1, False, x.data_ptr(), 2, n)
            rn.f.solve_phase_dev(0, False, x.data_ptr(), 2, n)
            rn.f.keep_rows_dev(x.data_ptr(), 2, n)
        summed()
        for x in Xs:
            assert relerr(x.numpy(), want) < 1e-11
    finally:
        emu.emu_set_tuning(2, 256); emu.emu_set_tuning(7, 1024)


@pytest.mark.parametrize("kind,size,world,nrhs", [("lap3d", 12, 2, 1), ("lap3d", 12, 4, 3), ("lap2d", 40, 8, 2)])
def test_solve_phases_with_simulated_ranks(kind, size, world, nrhs):
    """lsolve / ltsolve on the partitioned factor: local sweeps + sum over the ranks + replicated top
    sweeps give L^-1 b and L^-T b of the oracle's factor on every rank"""
    import torch
    from oracle import port
    g = problem(kind, size)
    n = g["n"]
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    facs = [EmuFactor(g) for _ in range(world)]
    X = [None] * world

    def allreduce(tensors):          # simulated ranks run one after the other: the sum is taken below
        pass
    runners = [multigpu.PartitionedFactor(f, g, r, world, allreduce=allreduce) for r, f in enumerate(facs)]
    for rn in runners:
        assert rn.f.numeric_phase(0) == (-1, 0)
    for lo, hi in runners[0].part["top_ranges"]:
        tot = sum(f.Lx[lo:hi] for f in facs)
        for f in facs:
            f.Lx[lo:hi] = tot
    for rn in runners:
        assert rn.f.numeric_phase(1) == (-1, 0)
    rng = np.random.default_rng(5)
    B = rng.standard_normal((nrhs, n))
    want_l = port.super_solve(g, Lo, B.T.copy(), lower=True).T
    want_lt = port.super_solve(g, Lo, want_l.T.copy(), lower=False).T
    Xs = [torch.from_numpy(B.copy()) for _ in range(world)]

    def summed():
        tot = sum(x.numpy() for x in Xs)
        for x in Xs:
            x.numpy()[...] = tot
    # forward: keep_rows + local sweep, sum, top sweep
    for rn, x in zip(runners, Xs):
        rn.f.keep_rows_dev(x.data_ptr(), nrhs, n)
        rn.f.solve_phase_dev(0, True, x.data_ptr(), nrhs, n)
    summed()
    for rn, x in zip(runners, Xs):
        rn.f.solve_phase_dev(1, True, x.data_ptr(), nrhs, n)
        assert relerr(x.numpy(), want_l) < 1e-11
    # backward: top sweep, local sweep, keep_rows, sum
    for rn, x in zip(runners, Xs):
        rn.f.solve_phase_dev(1, False, x.data_ptr(), nrhs, n)
        rn.f.solve_phase_dev(0, False, x.data_ptr(), nrhs, n)
        rn.f.keep_rows_dev(x.data_ptr(), nrhs, n)
    summed()
    for x in Xs:
        assert relerr(x.numpy(), want_lt) < 1e-11


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import test_multigpu as tm
from suitesparse_amd import multigpu
from oracle import port
g = tm.problem("lap3d", 10)
f = tm.EmuFactor(g)
f.E.emu_set_tuning(2, 64); f.E.emu_set_tuning(7, 64)     # 64-column outer blocks: the top spreads over both ranks
run = multigpu.PartitionedFactor(f, g, rank, world, min_subtrees_per_rank=1)
assert run.dist and len(set(int(r) for r in run.plan[:, 3])) == 2
assert run.reduce is not None          # the top panels go to their owners block by block (dist.reduce), no all-reduce
st = run.numeric()
so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
owner = run.part["owner"]
worst = 0.0
for s in range(g["nsuper"]):
    if owner[s] in (rank, -1):
        a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
        worst = max(worst, float(np.abs(f.Lx[a:b] - Lo[a:b]).max() / np.abs(Lo).max()))
# solves on the partitioned factor: every rank ends with the oracle's L^-T L^-1 b
rng = np.random.default_rng(3)
b = rng.standard_normal((2, g["n"]))
X = torch.from_numpy(b.copy())
run.lsolve(X, 2)
run.ltsolve(X, 2)
want = port.super_solve(g, Lo, port.super_solve(g, Lo, b.T.copy(), lower=True), lower=False).T
worst = max(worst, float(np.abs(X.numpy() - want).max() / np.abs(want).max()))
# not positive definite: every rank must agree on status and minor
g2 = dict(g); Sx = g["Sx"].copy(); col = int(g["n"] // 2); Sx[g["Sp"][col]] = -5.0; g2["Sx"] = Sx
f2 = tm.EmuFactor(g2)
run2 = multigpu.PartitionedFactor(f2, g2, rank, world)
st2 = run2.numeric()
so2, mo2, _ = port.super_numeric(g2, g2["Sp"], g2["Si"], g2["Sx"])
print("RESULT", rank, st, worst, st2, run2.minor, so2, mo2, flush=True)
with open(os.path.join({out!r}, f"result_{{rank}}.txt"), "w") as fh:      # stdout of two ranks can interleave
    fh.write(f"RESULT {{rank}} {{st}} {{worst}} {{st2}} {{run2.minor}} {{so2}} {{mo2}}\n")
dist.destroy_process_group()
'''


def test_two_process_gloo_run(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    import socket
    for attempt in range(2):                         # the rendezvous port can be taken between probe and use: once more
        with socket.socket() as sk:                  # a free port: a fixed one may still be held by an earlier run
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                             capture_output=True, text=True, env=env, timeout=300)
        if out.returncode == 0 or "RESULT" in out.stdout:
            break
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [open(tmp_path / f"result_{r}.txt").read().split() for r in range(2)]
    assert len(lines) == 2
    for l in lines:
        st, worst, st2, minor2, so2, mo2 = int(l[2]), float(l[3]), int(l[4]), int(l[5]), int(l[6]), int(l[7])
        assert st == 0 and worst < 1e-11
        assert st2 == so2 == 1 and minor2 == mo2


WORKER_DEVICE_SOURCE = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import test_multigpu as tm
from suitesparse_amd import multigpu
from oracle import port
g = tm.problem("lap3d", 6)
f = tm.EmuFactor(g)
f.E.emu_set_tuning(100, 3)              # every factor kernel from the product's own source (one fiber per lane)
run = multigpu.PartitionedFactor(f, g, rank, world, min_subtrees_per_rank=1)
assert run.dist and run.reduce is not None
st = run.numeric()
so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
owner = run.part["owner"]
worst = 0.0
for s in range(g["nsuper"]):
    if owner[s] in (rank, -1):
        a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
        nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
        got, want = f.Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
        low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
        worst = max(worst, float(np.abs(got[low] - want[low]).max() / np.abs(Lo).max()))
with open(os.path.join({out!r}, f"result_{{rank}}.txt"), "w") as fh:
    fh.write(f"RESULT {{rank}} {{st}} {{worst}} {{len(run.plan)}}\n")
dist.destroy_process_group()
'''


def test_two_process_gloo_run_with_the_device_kernels(tmp_path):
    """the N = 2 flow end to end -- subtrees, reduce to the owners, owner-computes top with broadcasts -- with every
    factor kernel executed from the product's own source on the CPU (emulator key 100 = 3) and gloo as the transport"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER_DEVICE_SOURCE.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    import socket
    for attempt in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                             capture_output=True, text=True, env=env, timeout=900)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stdout + out.stderr
    for r in range(2):
        l = open(tmp_path / f"result_{r}.txt").read().split()
        assert int(l[2]) == 0 and float(l[3]) < 1e-10 and int(l[4]) >= 2
