"""The GPU test bodies, on the CPU.  build() compiles the engine ITSELF -- engine.hip's host code and every kernel, the
whole C ABI of include/mi355_cholmod.h -- for the host against tests/helpers/hipcpu/hip/hip_runtime.h (device memory =
host memory, a launch = one fiber per lane) into tests/helpers/libmi355_cpuemu.so.  Here the Python binding is
pointed at that library (from the test only: the product knows nothing of it and has no CPU path) and the bodies of
the GPU tests run on small cases.  What this covers that the schedule emulator cannot: the engine's host logic --
staging and validation, the launch sequences of run_schedule and solve_dev, the not-positive-definite protocol, the
permutations of mi355_solve, rcond, the simplicial conversion -- together with the kernels, through the C ABI."""
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden, sym_of, relerr

CPUEMU = os.path.join(ROOT, "tests", "helpers", "libmi355_cpuemu.so")


@pytest.fixture(scope="module")
def engine():
    if not os.path.exists(CPUEMU):
        pytest.skip("libmi355_cpuemu.so not built (needs /opt/rocm's clang++)")
    from suitesparse_amd import engine as e, symbolic
    saved = (e.LIB_PATH, e._LIB)
    e.LIB_PATH, e._LIB = CPUEMU, None
    symbolic._ready = False              # its argtypes are set once per library object
    try:
        assert e.probe() == 1
        yield e
    finally:
        e.LIB_PATH, e._LIB = saved
        symbolic._ready = False


@pytest.mark.parametrize("name", ["demo_bcsstk01", "demo_bcsstk02", "demo_c", "demo_can24", "tcov_c30lo_singular", "tcov_r5lo",
                                  "tcov_z5up", "tcov_ex5lo"])
def test_golden_factor_and_solve(engine, name):
    import test_gpu_parity as t
    t.test_golden_factor_and_solve(engine, name)


def test_quick_return_if_not_posdef(engine):
    import test_gpu_parity as t
    for name in ("elast27_4_npd100", "lap2d_16_npd0"):
        t.test_quick_return_if_not_posdef(engine, name)


def test_error_paths_and_empty_factor(engine):
    import test_gpu_parity as t
    t.test_error_paths(engine)
    t.test_empty_factor(engine)


@pytest.mark.parametrize("n", [1, 5, 64, 65, 130])
def test_dense_single_supernode(engine, n):
    import test_gpu_parity as t
    t.test_dense_single_supernode(engine, n)


def test_zomplex_input_gives_complex_factor(engine):
    import test_gpu_parity as t
    t.test_zomplex_input_gives_complex_factor(engine)


def test_solve2_with_permutation_all_systems(engine):
    """mi355_solve (cholmod_solve2's supernodal branch): Y = P b, sweeps, x = P' y on the 'device', every sys value,
    against dense algebra on the golden factor; real L with a complex right-hand side (the dual path)"""
    g = load_golden("demo_bcsstk01")
    n = g["n"]
    L = engine.SupernodalFactor(sym_of(g))
    assert L.super_numeric(g["Sp"], g["Si"], g["Sx"]) == 0
    L.set_perm(g["Perm"])
    import test_unsym as tu
    Ld = tu.dense_L(g, g["Lx"])
    P = np.zeros((n, n))
    P[np.arange(n), g["Perm"]] = 1.0                         # (P b)[k] = b[Perm[k]]
    rng = np.random.default_rng(0)
    for B in (rng.standard_normal((n, 2)), rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))):
        want = {0: P.T @ np.linalg.solve(Ld @ Ld.T, P @ B), 1: np.linalg.solve(Ld @ Ld.T, B),
                2: np.linalg.solve(Ld, B), 3: np.linalg.solve(Ld.T, B), 4: np.linalg.solve(Ld, B),
                5: np.linalg.solve(Ld.T, B), 6: B, 7: P @ B, 8: P.T @ B}
        for sys_, w in want.items():
            assert relerr(L.solve(B, sys=sys_), w) < 1e-9, sys_
    L.close()


def test_device_side_permutation_rcond_and_simplicial(engine):
    """mi355_factor_set_matrix_perm (A + Perm instead of S), mi355_factor_rcond, mi355_factor_to_simplicial"""
    import scipy.sparse as sp
    g = load_golden("lap3d_8_amd") if False else load_golden("demo_bcsstk01")
    n = g["n"]
    A = sp.csc_matrix((g["Ax"], g["Ai"], g["Ap"]), shape=(n, n))
    if g["stype"] > 0:
        A = A.T.tocsc()
    A = sp.tril(A, format="csc")
    A.sort_indices()
    L = engine.SupernodalFactor(sym_of(g))
    L.set_matrix_perm(A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data, g["Perm"])
    assert L.numeric() == 0
    assert relerr(L.Lx(), g["Lx"]) < 1e-10
    import test_unsym as tu
    Ld = tu.dense_L(g, g["Lx"])
    d = np.diag(Ld)
    assert abs(L.rcond() - (d.min() / d.max()) ** 2) < 1e-12
    Lp, Li, Lnz, Lv = L.to_simplicial(True)
    got = sp.csc_matrix((Lv, Li, Lp), shape=(n, n)).toarray()
    assert relerr(got, Ld) < 1e-10
    Lp, Li, Lnz, Lv = L.to_simplicial(False)                 # LDL': D on the diagonal, unit-scaled columns
    got = sp.csc_matrix((Lv, Li, Lp), shape=(n, n)).toarray()
    assert relerr(np.diag(got), d ** 2) < 1e-10
    assert relerr(np.tril(got, -1), np.tril(Ld / d[None, :], -1)) < 1e-10
    L.close()


@pytest.mark.parametrize("name", ["elast27_4_npd100", "lap2d_16_npd0", "tcov_c30lo_singular"])
def test_device_side_permutation_not_posdef_through_the_abi(engine, name):
    import test_gpu_parity as tp
    tp.test_device_side_permutation_of_a_matrix_that_is_not_posdef(engine, name)


# ---- this session's additions, which no GPU has run yet

def test_unsym_through_the_abi(engine):
    """mi355_factor_set_matrix_unsym + k_assemble_aat: every unsymmetric fixture (real, complex, beta), zomplex and
    unpacked inputs, the argument checks"""
    import test_unsym as tu
    for name in tu.unsym_names():
        tu.test_gpu_golden_unsym(engine, name)
    tu.test_gpu_unsym_zomplex_unpacked_and_errors(engine)


@pytest.mark.parametrize("case", [0, 2, 4])
def test_unsym_live_through_the_abi(engine, have_ref, case):
    import test_unsym as tu
    tu.test_gpu_live_unsym(engine, have_ref, *tu.LIVE[case])


def test_unsym_fset_through_the_abi(engine, have_ref):
    import test_unsym as tu
    tu.test_gpu_column_subset_fset(engine, have_ref, *tu.FSET[0])


@pytest.mark.parametrize("name,tiny_max", [("demo_bcsstk01", 1 << 18), ("tcov_ex5lo", 64), ("demo_bcsstk02", 4096)])
def test_tiny_launch_classes_through_the_abi(engine, name, tiny_max):
    """tuning key 14 through the engine: the potrf order list upload, k_potrf16 (+ fused TRSM) / k_potrf64 with the order,
    k_gemm_tiny launches"""
    import test_gpu_tiny_tasks as w
    w.test_tiny_launch_class_golden(engine, name, tiny_max)


@pytest.mark.parametrize("n", [1, 5, 16, 17, 80])
def test_small_dense_supernode_through_the_wave_potrf(engine, n):
    import test_gpu_tiny_tasks as w
    w.test_small_dense_supernode_through_the_wave_potrf(engine, n)


def test_wave_potrf_reports_the_first_bad_pivot(engine):
    import test_gpu_tiny_tasks as w
    w.test_wave_potrf_reports_the_first_bad_pivot(engine)


# ---- the multi-GPU entry points of the C ABI (set_partition, numeric_phase, set_top_distribution, top_plan, top_run,
# solve_phase_dev, keep_rows_dev) with several handles in one process; "device" memory is host memory here, so the
# views RCCL would work on are numpy / torch views of the same buffers

class EngineFactor:
    """the interface multigpu.PartitionedFactor expects, on a real engine handle of the CPU build"""

    def __init__(self, engine, g):
        import ctypes as C
        self.e, self.g = engine, g
        self.L = engine.SupernodalFactor(g)
        self.L.set_matrix(g["Sp"], g["Si"], g["Sx"])
        self.Lx = np.ctypeslib.as_array((C.c_double * int(g["xsize"])).from_address(self.L.device_Lx_ptr()))
        nd = int(self.L.stats()["dinv_entries"])
        self.Dinv = np.ctypeslib.as_array((C.c_double * max(nd, 1)).from_address(
            engine.lib().mi355_factor_device_Dinv(self.L._h)))
        self.minor = g["n"]

    def set_partition(self, *m): self.L.set_partition(*m)
    def numeric_phase(self, phase, beta=0.0): return self.L.numeric_phase(phase, beta)
    def set_top_distribution(self, rank, world): return self.L.set_top_distribution(rank, world)
    def top_run(self, seg): return self.L.top_run(seg)
    def solve_phase_dev(self, part, lower, X_ptr, nrhs, d): self.L.solve_phase_dev(part, lower, X_ptr, nrhs, d)
    def keep_rows_dev(self, X_ptr, nrhs, d): self.L.keep_rows_dev(X_ptr, nrhs, d)

    def numeric(self, beta=0.0, quick=False):
        st = self.L.numeric(beta, quick)
        self.minor = self.L.minor
        return st

    def top_tensors(self, ranges):
        import torch
        t = torch.from_numpy(self.Lx)
        return [t[lo:hi] for lo, hi in ranges]

    def dinv_tensors(self, ranges):
        import torch
        t = torch.from_numpy(self.Dinv)
        return [t[lo:hi] for lo, hi in ranges]


@pytest.mark.parametrize("world,spoil,tiny_max", [(3, False, 0), (2, True, 0), (2, False, 4096)])
def test_partitioned_factor_through_the_abi(engine, world, spoil, tiny_max):
    """N handles = N simulated ranks: subtrees, reduce to the owners (partial sums left elsewhere, as after a real
    reduce), owner-computes top segment by segment with the broadcasts done by hand, block inverses summed, the solves on
    the partitioned factor; spoil: a negative pivot, every rank must end with the reference's minor and factor"""
    import torch
    from oracle import port
    from suitesparse_amd import multigpu
    from test_multigpu import problem
    g = dict(problem("lap3d", 5 if spoil else 7))
    n = g["n"]
    if spoil:
        Sx = g["Sx"].copy()
        Sx[g["Sp"][n // 2]] = -3.0
        g["Sx"] = Sx
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    engine.set_tuning(2, 64); engine.set_tuning(7, 64)            # 64-column outer blocks: the top spreads over the ranks
    engine.set_tuning(14, tiny_max)
    try:
        facs = [EngineFactor(engine, g) for _ in range(world)]
        runs = [multigpu.PartitionedFactor(f, g, r, world, distribute_top=True, min_subtrees_per_rank=1,
                                           reduce=lambda ts, roots: None) for r, f in enumerate(facs)]
        assert all(rn.dist and rn.reduce is not None for rn in runs)
        plan = runs[0].plan
        for rn in runs[1:]:
            assert (rn.plan[:, 1:] == plan[:, 1:]).all()
        res0 = [rn.f.numeric_phase(0) for rn in runs]
        failed = any(b >= 0 for b, _ in res0)
        if not failed:
            for k, (step, lo, cnt, root) in enumerate(plan):
                tot = sum(rn.bviews[k].clone() for rn in runs)
                runs[int(root)].bviews[k][...] = tot
            for k in range(len(plan)):
                for rn in runs:
                    rn.f.top_run(k)
                root = int(plan[k, 3])
                for rn in runs:
                    if rn.rank != root:
                        rn.bviews[k][...] = runs[root].bviews[k]
            res1 = [rn.f.top_run(len(plan)) for rn in runs]
            failed = any(b >= 0 for b, _ in res1)
        assert failed == spoil
        if spoil:
            for rn in runs:
                st = rn._replay(0.0, False)
                assert (st, rn.minor) == (so, mo) == (1, mo)
                assert relerr(rn.f.Lx, Lo) < 1e-10 and np.array_equal(rn.f.Lx == 0, Lo == 0)
            return
        if runs[0].dinv_views:
            tot = sum(rn.dinv_views[0].clone() for rn in runs)
            for rn in runs:
                rn.dinv_views[0][...] = tot
        owner = runs[0].part["owner"]
        for r, f in enumerate(facs):
            for s in range(g["nsuper"]):
                if owner[s] in (r, -1):
                    a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
                    nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
                    got, want = f.Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
                    low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
                    assert relerr(got[low], want[low]) < 1e-10, (r, s)
        for rn in runs:                      # the engine only solves on a factor it saw complete
            rn.status = 0
        B = np.random.default_rng(1).standard_normal((2, n))
        want = port.super_solve(g, Lo, port.super_solve(g, Lo, B.T.copy(), lower=True), lower=False).T
        Xs = [torch.from_numpy(B.copy()) for _ in range(world)]

        def summed():
            tot = sum(x.numpy().copy() for x in Xs)
            for x in Xs:
                x.numpy()[...] = tot
        for rn, x in zip(runs, Xs):
            rn.f.keep_rows_dev(x.data_ptr(), 2, n)
            rn.f.solve_phase_dev(0, True, x.data_ptr(), 2, n)
        summed()
        for rn, x in zip(runs, Xs):
            rn.f.solve_phase_dev(1, True, x.data_ptr(), 2, n)
            rn.f.solve_phase_dev(1, False, x.data_ptr(), 2, n)
            rn.f.solve_phase_dev(0, False, x.data_ptr(), 2, n)
            rn.f.keep_rows_dev(x.data_ptr(), 2, n)
        summed()
        for x in Xs:
            assert relerr(x.numpy(), want) < 1e-9
    finally:
        engine.set_tuning(2, 256); engine.set_tuning(7, 1024); engine.set_tuning(14, 2048)


@pytest.mark.parametrize("kind,size", [("lap3d", 4), ("helm", 3)])
def test_every_solve_path_through_the_abi(engine, kind, size):
    """solve_dev's dispatch (tuning keys 9 / 10: scalar or MFMA kernels, grouped or per block; real and complex) through
    mi355_super_lsolve / _ltsolve, for 1, 3, 7 and 20 right-hand sides"""
    import test_gpu_parity as t
    t.test_every_solve_path_against_the_oracle(engine, kind, size)


@pytest.mark.parametrize("cplx,n0,n1,nrhs,wgs", [(False, 300, 100, 20, 256), (False, 130, 70, 70, 256), (True, 140, 50, 7, 256),
                                                 (True, 70, 60, 70, 256), (False, 600, 130, 6, 3), (True, 300, 90, 5, 2)])
def test_wide_multi_rhs_sweeps_through_the_abi(engine, cplx, n0, n1, nrhs, wgs):
    """k_solve_fwd_wide / k_solve_bwd_wide (tuning key 23) from their own source on the CPU: contraction over a whole solve
    group, real (256 columns) and complex (128), ragged blocks and row tiles, more than one pass over the right-hand sides"""
    import test_gpu_parity as t
    t.test_wide_multi_rhs_sweeps(engine, cplx, n0, n1, nrhs, wgs)


@pytest.mark.parametrize("kind,n,seed,tiny_max", [("arrow", 150, 32, 0), ("sparse", 60, 35, 2048)])
def test_random_structure_end_to_end_through_the_abi(engine, kind, n, seed, tiny_max):
    """random SPD structure under a random permutation, the way a CHOLMOD caller would use the library: our symbolic
    analysis, A + Perm staged (assembly through the inverse permutation), numeric, mi355_solve with CHOLMOD_A --
    residual of A x = b in the ORIGINAL ordering, and the factor against the oracle's"""
    import scipy.sparse as sp
    import test_fuzz_schedule as tf
    from oracle import port
    from suitesparse_amd import symbolic
    rng = np.random.default_rng(seed)
    Ap, Ai, Ax = tf.random_spd(kind, n, seed)
    sym = symbolic.analyze(n, Ap, Ai, rng.permutation(n))
    Sp, Si, Sx = symbolic.permute(sym, Ap, Ai, Ax)
    so, mo, Lo = port.super_numeric(sym, Sp, Si, Sx)
    engine.set_tuning(14, tiny_max)
    try:
        L = engine.SupernodalFactor(sym)
        L.set_matrix_perm(Ap, Ai, Ax, sym["Perm"])
        assert L.numeric() == so == 0
        Lx = L.Lx()
        L.set_perm(sym["Perm"])
        A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
        A = A + sp.tril(A, -1).T
        B = rng.standard_normal((n, 3))
        X = L.solve(B, sys=0)
        L.close()
    finally:
        engine.set_tuning(14, 2048)
    assert np.abs(A @ X - B).max() <= 1e-12 * np.abs(B).max() * n
    for s in range(sym["nsuper"]):
        a, b = int(sym["Lpx"][s]), int(sym["Lpx"][s + 1])
        nsrow = int(sym["Lpi"][s + 1] - sym["Lpi"][s])
        got, want = Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
        low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
        assert np.abs(got[low] - want[low]).max() <= 1e-10 * np.abs(Lo).max()


def test_two_supernodes_scatter_update(engine):
    import test_gpu_parity as t
    t.test_two_supernodes_scatter_update(engine)


def test_complex_solve2_with_permutation(engine):
    """mi355_solve on a complex factor with L->Perm: complex B, and a real B under a complex L (zero imaginary part on the
    way in), against dense algebra on the golden factor"""
    import test_unsym as tu
    g = load_golden("tcov_c30up")
    n = g["n"]
    L = engine.SupernodalFactor(sym_of(g), xtype=2)
    assert L.super_numeric(g["Sp"], g["Si"], g["Sx"]) == 0
    L.set_perm(g["Perm"])
    Ld = tu.dense_L(g, g["Lx"])
    P = np.zeros((n, n))
    P[np.arange(n), g["Perm"]] = 1.0
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    want = P.T @ np.linalg.solve(Ld @ Ld.conj().T, P @ B)
    assert relerr(L.solve(B, 0), want) < 1e-9
    Br = rng.standard_normal((n, 2))
    assert relerr(L.solve(Br, 0), L.solve(Br + 0j, 0)) < 1e-14
    assert relerr(L.solve(B, 4), np.linalg.solve(Ld, B)) < 1e-9           # CHOLMOD_L
    assert relerr(L.solve(B, 5), np.linalg.solve(Ld.conj().T, B)) < 1e-9  # CHOLMOD_Lt
    L.close()


def test_too_large_is_reported_not_wrapped(engine):
    """SURVEY 8 row a9 (BLAS int narrowing): the reference guards every narrowing of its int64 sizes to the BLAS integer
    and reports CHOLMOD_TOO_LARGE (t_cholmod_super_numeric_worker.c:183-188, cholmod_super_numeric.c:202).  Here the
    narrow types are the kernels' int32 row indices and launch counts: n >= 2^31 - 1 is refused by mi355_factor_create
    before any array of that length is read, and more right-hand sides than one launch can index by the solves"""
    import ctypes as C
    CHOLMOD_TOO_LARGE = -3
    one = np.zeros(2, dtype=np.int64)
    h = C.c_void_p()
    n = (1 << 31) - 1
    Super = np.array([0, n], dtype=np.int64)
    rc = engine.lib().mi355_factor_create(1, n, 1, Super, Super, np.array([0, n * n], dtype=np.int64), one, n, n * n, 0,
                                          C.byref(h))
    assert rc == CHOLMOD_TOO_LARGE and not h.value
    assert b"too large" in engine.lib().mi355_last_error()
    g = load_golden("demo_bcsstk01")
    L = engine.SupernodalFactor(sym_of(g))
    assert L.super_numeric(g["Sp"], g["Si"], g["Sx"]) == 0
    x = np.zeros(g["n"])
    for fn in (engine.lib().mi355_super_lsolve_dev, engine.lib().mi355_super_ltsolve_dev):
        assert fn(L._h, (1 << 30), C.c_void_p(x.ctypes.data), g["n"]) == CHOLMOD_TOO_LARGE
    assert engine.lib().mi355_solve_dev(L._h, 0, (1 << 30), C.c_void_p(x.ctypes.data), g["n"],
                                        C.c_void_p(x.ctypes.data), g["n"]) == CHOLMOD_TOO_LARGE
    L.close()


def test_blas_timer_statistics(engine):
    """SURVEY 8 row a10 (the -DBLAS_TIMER counters, worker :246-266,:763-811): with profiling on, every launch is
    bracketed; the launch counts equal the schedule's steps, the kernel times are positive and add up to no more than
    the call, and a call with profiling off reports counts but no kernel times"""
    from test_multigpu import problem
    g = dict(problem("lap3d", 5))
    L = engine.SupernodalFactor(g)
    L.set_matrix(g["Sp"], g["Si"], g["Sx"])
    L.set_profile(True)
    assert L.numeric() == 0
    st = L.stats()
    assert st["gemm_launches"] > 0 and st["potrf_launches"] > 0
    assert st["gemm_ms"] > 0 and st["potrf_ms"] > 0 and st["assembly_ms"] > 0
    assert st["gemm128_ms"] <= st["gemm_ms"] and st["gemm128_launches"] <= st["gemm_launches"]
    assert st["gemm_ms"] + st["potrf_ms"] + st["assembly_ms"] <= st["total_ms"] * 1.001
    assert st["gemm_tiles"] == st["tiles_gemm128"] + st["tiles_gemm64"]
    assert st["flops_update"] > 0 and st["flops_panel"] > 0 and st["npairs"] > 0 and st["nlevels"] > 0
    n1 = (st["gemm_launches"], st["potrf_launches"])
    assert L.numeric() == 0                             # counts are per call, not cumulative
    st2 = L.stats()
    assert (st2["gemm_launches"], st2["potrf_launches"]) == n1
    L.set_profile(False)
    assert L.numeric() == 0
    st3 = L.stats()
    assert (st3["gemm_launches"], st3["potrf_launches"]) == n1 and st3["total_ms"] > 0
    assert st3["gemm_ms"] == st2["gemm_ms"]             # nothing new was measured: the last profiled call's times stay
    L.close()


@pytest.mark.parametrize("name,chunk", [("lap3d_8_metis", 2000), ("helm3d_6_metis", 500), ("elast27_4_metis", 4000),
                                         ("lap2d_16_metis", 300), ("demo_bcsstk02", 1 << 25),
                                         ("lap2d_16_npd0", 300), ("tcov_c30lo_singular", 50)])
def test_numeric_host_copies_under_the_factorization(engine, name, chunk):
    import test_gpu_parity as tp
    tp.test_numeric_host_copies_under_the_factorization(engine, name, chunk)


def test_host_solve_through_the_lanes(engine):
    """mi355_solve_x with host B / X moved through the lanes (tuning key 22 = 0: every size; key 21 = 1 MB staging buffers
    so that a column is cut into pieces): real and complex right-hand sides, twice (the device
    images of B and X are kept from call to call), against the plain path"""
    g = load_golden("lap3d_8_metis")
    n = g["n"]
    rng = np.random.default_rng(2)
    L = engine.SupernodalFactor(sym_of(g))
    L.set_matrix(g["Sp"], g["Si"], g["Sx"])
    assert L.numeric(beta=g["beta"]) == 0
    L.set_perm(g["Perm"])
    want = {}
    for cplx in (False, True):
        B = rng.standard_normal((n, 5)) + (1j * rng.standard_normal((n, 5)) if cplx else 0)
        want[cplx] = (B, L.solve(B, sys=0))
    L.close()
    engine.set_tuning(22, 0); engine.set_tuning(21, 1); engine.set_tuning(20, 3)
    try:
        L = engine.SupernodalFactor(sym_of(g))
        L.set_matrix(g["Sp"], g["Si"], g["Sx"])
        assert L.numeric(beta=g["beta"]) == 0
        L.set_perm(g["Perm"])
        for _ in range(2):
            for cplx in (False, True):
                B, X = want[cplx]
                assert relerr(L.solve(B, sys=0), X) < 1e-13
        L.close()
    finally:
        engine.set_tuning(22, 4096); engine.set_tuning(21, 32); engine.set_tuning(20, 4)


def test_solve_recovers_after_a_failed_allocation(engine):
    """mi355_solve_x keeps its device images of B and X from call to call; an allocation that fails
    while they grow must not leave the old capacity on record with the buffer gone -- the natural recovery, the same
    solve with fewer right-hand sides, has to work.  Every allocation of a 40-column solve is failed in turn (the CPU
    stand-in of the HIP runtime injects the failure); after each failure a 3-column solve must succeed and be right."""
    import ctypes as C
    emu = engine.lib()
    emu.hipcpu_alloc_calls.restype = C.c_long
    emu.hipcpu_live_allocs.restype = C.c_long
    emu.hipcpu_fail_alloc_at.argtypes = [C.c_long]
    g = load_golden("lap3d_8_metis")
    n = g["n"]
    rng = np.random.default_rng(5)
    base_live = emu.hipcpu_live_allocs()
    L = engine.SupernodalFactor(sym_of(g))
    L.set_matrix(g["Sp"], g["Si"], g["Sx"])
    assert L.numeric(beta=g["beta"]) == 0
    L.set_perm(g["Perm"])
    Bs, Bl = rng.standard_normal((n, 3)), rng.standard_normal((n, 40))
    Xs = L.solve(Bs, sys=0)             # (buffers for 3 columns now exist)
    failed = 0
    try:
        for k in range(1, 40):
            emu.hipcpu_fail_alloc_at(k)
            try:
                Xl = L.solve(Bl, sys=0)
                injected = emu.hipcpu_alloc_calls() >= k
            except engine.Mi355Error as e:
                assert e.code == -2, e               # CHOLMOD_OUT_OF_MEMORY
                injected, Xl = True, None
                failed += 1
            emu.hipcpu_fail_alloc_at(0)
            assert relerr(L.solve(Bs, sys=0), Xs) < 1e-13, k        # the smaller solve works whatever happened before
            if Xl is not None:
                assert relerr(L.solve(Bl, sys=0), Xl) < 1e-13
            if not injected:
                break
            # shrink back so that the next round has to grow the buffers again
            L.close()
            L = engine.SupernodalFactor(sym_of(g))
            L.set_matrix(g["Sp"], g["Si"], g["Sx"])
            assert L.numeric(beta=g["beta"]) == 0
            L.set_perm(g["Perm"])
            assert relerr(L.solve(Bs, sys=0), Xs) < 1e-13
    finally:
        emu.hipcpu_fail_alloc_at(0)
    assert failed >= 2
    L.close()
    assert emu.hipcpu_live_allocs() == base_live


def test_copy_lanes_are_sized_by_the_factor(engine):
    """every factor that is copied to the host (mi355_factor_numeric_host: each cholmod_factor of the
    drop-in) used to pin 4 x 32 MB and start four threads whatever its size.  The lanes are now sized by the factor: a
    factor of a few KB pins 64 KB in one lane; a larger one gets more lanes, never more than its own size rounded up."""
    import ctypes as C
    emu = engine.lib()
    emu.hipcpu_pinned_bytes.restype = C.c_longlong
    base = emu.hipcpu_pinned_bytes()
    for name, most in (("demo_bcsstk01", 1 << 16), ("lap3d_8_metis", 1 << 20)):
        g = load_golden(name)
        L = engine.SupernodalFactor(sym_of(g), xtype=2 if g["xtype"] == 2 else 1)
        L.set_matrix(g["Sp"], g["Si"], g["Sx"])
        assert L.numeric_host(beta=g["beta"])[0] == g["status"]
        pinned = emu.hipcpu_pinned_bytes() - base
        fbytes = int(g["xsize"]) * 8
        assert 0 < pinned <= max(most, 2 * fbytes), (name, pinned, fbytes)
        L.close()
        assert emu.hipcpu_pinned_bytes() == base


@pytest.mark.parametrize("name", ["lap3d_8_metis", "helm3d_6_metis", "tcov_c30lo_singular"])
def test_deterministic_mode_through_the_abi(engine, name):
    """tuning key 27 (rounds of scatter updates): the schedule it builds gives the reference's factor (the CPU build runs
    workgroups one after the other, so identical bits are no news here; the GPU test of the same name is the real one)"""
    import test_gpu_parity as tp
    tp.test_deterministic_mode_gives_identical_bits(engine, name)


@pytest.mark.parametrize("case,rows", [("dense130", 64), ("dense300", 32), ("dense700", 128), ("lap3d_8_metis", 128), ("lap3d_8_npd200", 16), ("arrow800", 64)])
def test_panel_groups_through_the_abi(engine, case, rows):
    """tuning key 28: the engine's own launch code for STEP_PDIAG / STEP_PTRSM (k_panel_diag, k_panel_trsm from their
    source) on small cases -- the body of the GPU test of the same name"""
    import test_gpu_parity as tp
    tp.test_panel_groups(engine, case, rows)


@pytest.mark.parametrize("n0,n1,nrhs,wgs", [(300, 100, 20, 256), (600, 130, 33, 3)])
def test_wide_forward_sweep_with_three_register_sets_through_the_abi(engine, n0, n1, nrhs, wgs):
    """tuning key 29 = 3 (k_solve_fwd_wide with a ring of three register sets) from its own source on the CPU"""
    import test_gpu_parity as t
    t.test_wide_forward_sweep_with_three_register_sets(engine, n0, n1, nrhs, wgs)


@pytest.mark.parametrize("cplx,n0,n1,nrhs,keep", [(False, 1100, 200, 9, 12), (True, 600, 130, 7, 12)])
def test_sweeps_with_look_ahead_through_the_abi(engine, cplx, n0, n1, nrhs, keep):
    """tuning key 30: the split of a step's wide launch into its first workgroups and the rest, on two streams (the stand-in
    runs a stream's work at once, so this checks the split itself -- every workgroup exactly once -- and the event calls)"""
    import test_gpu_parity as t
    t.test_sweeps_with_look_ahead(engine, cplx, n0, n1, nrhs, keep)


@pytest.mark.parametrize("cplx,n0,n1,nrhs,most", [(False, 600, 150, 20, 4), (True, 300, 90, 7, 4), (False, 330, 100, 9, 0), (False, 600, 150, 3, 4),
                                                  (False, 330, 90, 1, 4)])
def test_group_steps_with_explicit_inverses_through_the_abi(engine, cplx, n0, n1, nrhs, most):
    """tuning key 31: k_solve_group_inv, k_ginv_identity and the rebuild after a second factorization, from their own source"""
    import test_gpu_parity as t
    t.test_group_steps_with_explicit_inverses(engine, cplx, n0, n1, nrhs, most)
