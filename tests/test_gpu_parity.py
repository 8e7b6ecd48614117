"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the C ABI
(include/mi355_cholmod.h), against the reference's golden outputs and the CPU oracle.

Bars (BASELINE.json north_star): structure is an input (bit-exact by construction);
max|L_gpu - L_ref| / max|L_ref| <= 1e-10; status / minor / zeroing pattern identical;
backward error of solves <= 10x the reference's.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden, sym_of, relerr
from test_schedule import backward_error_S

pytestmark = pytest.mark.gpu

TOL_L = 1e-10


@pytest.fixture(scope="module")
def engine():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from suitesparse_amd import engine as e
    assert e.probe() >= 1
    return e


@pytest.mark.parametrize("name", golden_names())
def test_golden_factor_and_solve(engine, name):
    """every committed fixture, real and complex (the cd_/zd_ worker instantiations)"""
    g = load_golden(name)
    L = engine.SupernodalFactor(sym_of(g), xtype=g["xtype"])
    st = L.super_numeric(g["Sp"], g["Si"], g["Sx"], beta=(g["beta"], 0.0))
    Lx = L.Lx()
    assert st == g["status"]
    assert L.minor == g["minor"]
    assert relerr(Lx, g["Lx"]) < TOL_L
    if st == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)      # reference zeroing pattern
        with pytest.raises(engine.Mi355Error):
            L.super_lsolve(g["B"])
        return
    Y = L.super_lsolve(g["B"])
    assert relerr(Y, g["Y"]) < 1e-8
    Z = L.super_ltsolve(Y)
    assert relerr(Z, g["Z"]) < 1e-8
    assert backward_error_S(g, Z) <= 10 * backward_error_S(g, g["Z"]) + 1e-18
    # single right-hand side takes the nrhs == 1 shape of the reference (worker :63-172)
    y1 = L.super_lsolve(g["B"][:, 0].copy())
    assert relerr(y1, g["Y"][:, 0]) < 1e-8
    L.close()


@pytest.mark.parametrize("name", ["lap3d_8_npd200", "elast27_4_npd100", "lap2d_16_npd0"])
def test_quick_return_if_not_posdef(engine, name):
    from oracle import port
    g = load_golden(name)
    sym = sym_of(g)
    L = engine.SupernodalFactor(sym)
    st = L.super_numeric(g["Sp"], g["Si"], g["Sx"], quick_return_if_not_posdef=True)
    so, mo, Lo = port.super_numeric(sym, g["Sp"], g["Si"], g["Sx"], quick_return=True)
    Lx = L.Lx()
    assert (st, L.minor) == (so, mo) == (1, g["minor"])
    assert relerr(Lx, Lo) < TOL_L
    assert np.array_equal(Lx == 0, Lo == 0)


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 130, 256, 257, 300, 700, 1100, 1733])
def test_dense_single_supernode(engine, n):
    """one dense supernode: exercises POTRF blocks, TRSM-by-inverse, inner (K=64) and outer
    (K=512) trailing updates with ragged edges; from 1025 columns on also the column slices of the outer
    update that ride with the next block's inner launches."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    sym = dict(n=n, nsuper=1, ssize=n, xsize=n * n, Super=np.array([0, n]), Lpi=np.array([0, n]),
               Lpx=np.array([0, n * n]), Ls=np.arange(n))
    Sp = np.arange(n + 1, dtype=np.int64) * n
    Si = np.tile(np.arange(n, dtype=np.int64), n)
    Sx = np.ascontiguousarray(A.T).reshape(-1)              # full columns; upper entries are ignored
    L = engine.SupernodalFactor(sym)
    assert L.super_numeric(Sp, Si, Sx) == 0
    Lg = L.Lx().reshape(n, n).T
    Lr = np.linalg.cholesky(A)
    assert np.all(np.triu(Lg, 1) == 0)                      # strictly upper part stays zero (as in the reference)
    assert relerr(np.tril(Lg), Lr) < TOL_L
    B = rng.standard_normal((n, 5))
    X = L.super_ltsolve(L.super_lsolve(B))
    assert relerr(X, np.linalg.solve(A, B)) < 1e-9


def test_two_supernodes_scatter_update(engine):
    """hand-built arrow structure: supernode 0 = columns 0..39 with rows {0..39} + scattered rows
    of supernode 1 (columns 40..139): exercises the relative-map scatter with gaps."""
    from oracle import port
    n0, n1 = 40, 100
    n = n0 + n1
    extra = np.array(sorted(set(range(n0, n, 3)) | {n - 1}), dtype=np.int64)
    rows0 = np.concatenate([np.arange(n0), extra])
    rows1 = np.arange(n0, n)
    sym = dict(n=n, nsuper=2, Super=np.array([0, n0, n]), Lpi=np.array([0, len(rows0), len(rows0) + n1]),
               Lpx=np.array([0, len(rows0) * n0, len(rows0) * n0 + n1 * n1]), Ls=np.concatenate([rows0, rows1]))
    sym["ssize"] = len(sym["Ls"])
    sym["xsize"] = int(sym["Lpx"][-1])
    sym["maxcsize"] = len(extra) ** 2
    rng = np.random.default_rng(7)
    A = np.zeros((n, n))
    A[np.ix_(rows0, np.arange(n0))] = rng.standard_normal((len(rows0), n0))
    A[n0:, n0:] = rng.standard_normal((n1, n1))
    A = np.tril(A)
    A = A + A.T
    A += np.eye(n) * (np.abs(A).sum(1).max() + 1)
    import scipy.sparse as sp
    S = sp.csc_matrix(np.tril(A))
    S.sort_indices()
    Sp, Si, Sx = S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data
    L = engine.SupernodalFactor(sym)
    st = L.super_numeric(Sp, Si, Sx)
    so, mo, Lo = port.super_numeric(sym, Sp, Si, Sx)
    assert st == so == 0
    assert relerr(L.Lx(), Lo) < TOL_L


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,size,ordering", [("lap3d", 24, "metis"), ("lap2d", 150, "metis"),
                                                ("elast", 10, "metis"), ("lap3d", 20, "amd"), ("lap3d", 64, "metis"),
                                                ("lap3d", 100, "metis")])
def test_vs_live_reference(engine, kind, size, ordering, have_ref):
    """mid-size problems (supernodes > 256 wide, ~12 levels) and two large ones -- 64^3 and 100^3 (n = 10^6, an 8 GB
    factor, supernodes up to ~15 000 columns, fl 5.4e12: half a minute of the reference on the host cores) -- against
    the reference CPU worker in the same run, on the REFERENCE's METIS ordering: the 1e-10 bar on L anchored on the
    oracle at a size where thousands of descendants add into one ancestor with FP64 atomics in no fixed order"""
    if not have_ref:
        pytest.skip("oracle/_ref not available on this box")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d,
         "elast": problems.elasticity27}[kind](size)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze(ordering)
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    sym = r.symbolic()
    L = engine.SupernodalFactor(sym)
    assert L.super_numeric(Sp, Si, Sx) == 0 and L.minor == p["n"]
    Lx = L.Lx()
    assert relerr(Lx, r.Lx()) < TOL_L
    # refactorize with new values on the same handle (cholmod_super_numeric.c:27-31)
    assert L.super_numeric(Sp, Si, 2.0 * Sx) == 0
    assert relerr(L.Lx(), np.sqrt(2.0) * r.Lx()) < TOL_L
    assert L.super_numeric(Sp, Si, Sx) == 0
    B = problems.demo_rhs(p["n"], 3)
    Bp = B[sym["Perm"]]
    Xg = L.super_ltsolve(L.super_lsolve(Bp))
    Xr = r.super_solve(r.super_solve(Bp, True), False)
    assert relerr(Xg, Xr) < 1e-9
    X = np.empty_like(Xg)
    X[sym["Perm"]] = Xg
    bg = problems.backward_error(p, X, B)
    br = problems.backward_error(p, r.solve(B), B)
    assert bg <= 10 * br + 1e-18
    # drop-in: the reference's own cholmod_l_solve runs on the GPU-computed factor
    r.set_Lx(Lx)
    assert problems.backward_error(p, r.solve(B), B) <= 10 * br + 1e-18


def test_error_paths(engine):
    """argument checking in the spirit of CHOLMOD/Tcov/t_raw_factor.c:155-346"""
    g = load_golden("demo_bcsstk01")
    sym = sym_of(g)
    bad = dict(sym)
    bad["Lpx"] = sym["Lpx"].copy()
    bad["Lpx"][3] += 1
    with pytest.raises(engine.Mi355Error) as ei:
        engine.SupernodalFactor(bad)
    assert ei.value.code == engine.CHOLMOD_INVALID
    bad = dict(sym)
    bad["Ls"] = sym["Ls"].copy()
    bad["Ls"][5], bad["Ls"][6] = bad["Ls"][6], bad["Ls"][5]          # unsorted rows
    with pytest.raises(engine.Mi355Error):
        engine.SupernodalFactor(bad)
    with pytest.raises(engine.Mi355Error) as ei:
        engine.SupernodalFactor(sym, xtype=3)                        # L is never zomplex
    assert ei.value.code == engine.CHOLMOD_INVALID
    L = engine.SupernodalFactor(sym)
    with pytest.raises(engine.Mi355Error):
        L.set_matrix(g["Sp"], g["Si"], g["Sx"] + 0j)                 # complex A, real L: type mismatch
    with pytest.raises(engine.Mi355Error):
        L.numeric()                                                  # no matrix staged
    with pytest.raises(engine.Mi355Error):
        L.super_lsolve(g["B"])                                       # no numeric factor yet
    Si = g["Si"].copy()
    Si[0] = g["n"] + 3
    with pytest.raises(engine.Mi355Error):
        L.set_matrix(g["Sp"], Si, g["Sx"])                           # row index out of range
    assert L.super_numeric(g["Sp"], g["Si"], g["Sx"]) == 0
    x = np.zeros(g["n"])
    with pytest.raises(engine.Mi355Error):                           # d < n: "X and L dimensions must match"
        engine._check(engine.lib().mi355_super_lsolve(L._h, 1, x, g["n"] - 1))


def test_empty_factor(engine):
    sym = dict(n=0, nsuper=0, ssize=0, xsize=0, Super=np.zeros(1, np.int64), Lpi=np.zeros(1, np.int64),
               Lpx=np.zeros(1, np.int64), Ls=np.zeros(0, np.int64))
    L = engine.SupernodalFactor(sym)
    assert L.super_numeric(np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(0)) == 0
    assert L.minor == 0


@pytest.mark.parametrize("world", [2, 4])
def test_multigpu_phases_simulated_on_one_gpu(engine, world):
    """the two-phase multi-GPU path (mi355_factor_set_partition / _numeric_phase) with `world` handles on
    one GPU and the top-panel reduction done by hand on the device views RCCL would reduce"""
    import torch
    from oracle import port
    from suitesparse_amd import multigpu
    from test_multigpu import problem
    g = problem("lap3d", 16)
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    facs = []
    for r in range(world):
        L = engine.SupernodalFactor(g)
        L.set_matrix(g["Sp"], g["Si"], g["Sx"])
        facs.append(multigpu.PartitionedFactor(L, g, r, world, allreduce=lambda t: None))
    for f in facs:
        assert f.f.numeric_phase(0) == (-1, 0)
    torch.cuda.synchronize()
    for k in range(len(facs[0].tensors)):
        tot = sum(f.tensors[k] for f in facs)
        for f in facs:
            f.tensors[k].copy_(tot)
    torch.cuda.synchronize()
    for f in facs:
        assert f.f.numeric_phase(1) == (-1, 0)
    owner = facs[0].part["owner"]
    for r, f in enumerate(facs):
        Lx = f.f.Lx()
        for s in range(g["nsuper"]):
            if owner[s] in (r, -1):
                a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
                assert relerr(Lx[a:b], Lo[a:b]) < TOL_L
    # solves on the partitioned factor (mi355_super_solve_phase_dev / mi355_solve_keep_rows_dev): scalar
    # kernels (nrhs = 3) and MFMA sweeps (nrhs = 16), the sums over the ranks done by hand
    n = g["n"]
    for f in facs:
        f.status = 0
    for nrhs in (3, 16):
        B = np.random.default_rng(nrhs).standard_normal((nrhs, n))
        want_l = port.super_solve(g, Lo, B.T.copy(), lower=True).T
        want_lt = port.super_solve(g, Lo, want_l.T.copy(), lower=False).T
        Xs = [torch.from_numpy(B.copy()).cuda() for _ in range(world)]

        def summed():
            torch.cuda.synchronize()
            tot = sum(Xs)
            for x in Xs:
                x.copy_(tot)
            torch.cuda.synchronize()
        for f, x in zip(facs, Xs):
            f.f.keep_rows_dev(x.data_ptr(), nrhs, n)
            f.f.solve_phase_dev(0, True, x.data_ptr(), nrhs, n)
        summed()
        for f, x in zip(facs, Xs):
            f.f.solve_phase_dev(1, True, x.data_ptr(), nrhs, n)
            assert relerr(x.cpu().numpy(), want_l) < 1e-10
        for f, x in zip(facs, Xs):
            f.f.solve_phase_dev(1, False, x.data_ptr(), nrhs, n)
            f.f.solve_phase_dev(0, False, x.data_ptr(), nrhs, n)
            f.f.keep_rows_dev(x.data_ptr(), nrhs, n)
        summed()
        for x in Xs:
            assert relerr(x.cpu().numpy(), want_lt) < 1e-10
    # world = 1 through the same driver: lsolve + ltsolve of PartitionedFactor
    if world == 2:
        L1 = engine.SupernodalFactor(g)
        L1.set_matrix(g["Sp"], g["Si"], g["Sx"])
        run1 = multigpu.PartitionedFactor(L1, g, 0, 1)
        assert run1.numeric() == 0
        b = np.random.default_rng(9).standard_normal((2, n))
        X = torch.from_numpy(b.copy()).cuda()
        run1.lsolve(X, 2)
        run1.ltsolve(X, 2)
        want = port.super_solve(g, Lo, port.super_solve(g, Lo, b.T.copy(), lower=True), lower=False).T
        assert relerr(X.cpu().numpy(), want) < 1e-10


@pytest.mark.parametrize("world,ob,ob2,size,kind", [(2, 64, 128, 16, "lap3d"), (4, 64, 64, 16, "lap3d"),
                                                     (3, 256, 1024, 24, "lap3d"), (3, 64, 128, 12, "helm"),
                                                     # default block widths, top supernodes of several 1024-column blocks
                                                     # (root 2531 / 1048 columns): the shape of the first 8-GPU run
                                                     (4, 256, 1024, 40, "lap3d"), (2, 256, 1024, 26, "helm")])
def test_distributed_top_simulated_on_one_gpu(engine, world, ob, ob2, size, kind):
    """owner-computes top (mi355_factor_set_top_distribution / _top_plan / _top_run) with `world` handles on
    one GPU; the broadcasts and the sums are done by hand on the device views RCCL would use"""
    import torch
    from oracle import port
    from suitesparse_amd import multigpu
    from test_multigpu import problem
    g = problem(kind, size)
    n = g["n"]
    cplx = np.iscomplexobj(g["Sx"])
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    engine.set_tuning(2, ob); engine.set_tuning(7, ob2)
    try:
        runs = []
        for r in range(world):
            L = engine.SupernodalFactor(g, xtype=2 if cplx else 1)
            L.set_matrix(g["Sp"], g["Si"], g["Sx"])
            runs.append(multigpu.PartitionedFactor(L, g, r, world, allreduce=lambda t: None, distribute_top=True,
                                                   min_subtrees_per_rank=1))
        assert all(rn.dist for rn in runs)
        plan = runs[0].plan
        for rn in runs[1:]:
            assert (rn.plan[:, 1:] == plan[:, 1:]).all()
        if ob2 <= 128:
            assert len(set(int(r) for r in plan[:, 3])) > 1
        for rn in runs:
            assert rn.f.numeric_phase(0) == (-1, 0)
        torch.cuda.synchronize()
        for k in range(len(runs[0].tensors)):
            tot = sum(rn.tensors[k] for rn in runs)
            for rn in runs:
                rn.tensors[k].copy_(tot)
        torch.cuda.synchronize()
        for k in range(len(plan)):
            for rn in runs:
                rn.f.top_run(k)
            root = int(plan[k, 3])
            for rn in runs:
                if rn.rank != root:
                    rn.bviews[k].copy_(runs[root].bviews[k])
            torch.cuda.synchronize()
        for rn in runs:
            assert rn.f.top_run(len(plan)) == (-1, 0)
        if runs[0].dinv_views:
            tot = sum(rn.dinv_views[0] for rn in runs)
            for rn in runs:
                rn.dinv_views[0].copy_(tot)
            torch.cuda.synchronize()
        owner = runs[0].part["owner"]
        for r, rn in enumerate(runs):
            Lx = rn.f.Lx()
            for s in range(g["nsuper"]):
                if owner[s] in (r, -1):
                    a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
                    assert relerr(Lx[a:b], Lo[a:b]) < TOL_L, (r, s)
        B = np.random.default_rng(2).standard_normal((3, n))
        if cplx:
            B = B + 1j * np.random.default_rng(3).standard_normal((3, n))
        want = port.super_solve(g, Lo, port.super_solve(g, Lo, B.T.copy(), lower=True), lower=False).T
        Xs = [torch.from_numpy(np.ascontiguousarray(B).view(np.float64).copy()).cuda() for _ in range(world)]

        def summed():
            torch.cuda.synchronize()
            tot = sum(Xs)
            for x in Xs:
                x.copy_(tot)
            torch.cuda.synchronize()
        for rn, x in zip(runs, Xs):
            rn.f.keep_rows_dev(x.data_ptr(), 3, n)
            rn.f.solve_phase_dev(0, True, x.data_ptr(), 3, n)
        summed()
        for rn, x in zip(runs, Xs):
            rn.f.solve_phase_dev(1, True, x.data_ptr(), 3, n)
            rn.f.solve_phase_dev(1, False, x.data_ptr(), 3, n)
            rn.f.solve_phase_dev(0, False, x.data_ptr(), 3, n)
            rn.f.keep_rows_dev(x.data_ptr(), 3, n)
        summed()
        for x in Xs:
            got = x.cpu().numpy()
            got = got.view(np.complex128) if cplx else got
            assert relerr(got.reshape(3, n), want) < 1e-10
    finally:
        engine.set_tuning(2, 256); engine.set_tuning(7, 1024)


def test_zomplex_input_gives_complex_factor(engine):
    """A zomplex (split re/im), L complex: the zd_ instantiation (worker :69-83)"""
    g = load_golden("tcov_z30lo")
    L = engine.SupernodalFactor(sym_of(g), xtype=2)
    Sx = np.asarray(g["Sx"])
    st = L.super_numeric(g["Sp"], g["Si"], np.ascontiguousarray(Sx.real), Sz=np.ascontiguousarray(Sx.imag))
    assert st == 0 and relerr(L.Lx(), g["Lx"]) < TOL_L


@pytest.mark.parametrize("size,nrhs", [(12, 1), (16, 5), (14, 20), (12, 64), (10, 70)])
def test_complex_hermitian_vs_live_reference(engine, size, nrhs, have_ref):
    """config-5 shaped problem (complex Hermitian shifted Laplacian) incl. multi-RHS solve"""
    if not have_ref:
        pytest.skip("oracle/_ref not available on this box")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = problems.helmholtz_3d(size)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze("metis")
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    sym = r.symbolic()
    L = engine.SupernodalFactor(sym, xtype=2)
    assert L.super_numeric(Sp, Si, Sx) == 0
    Lx = L.Lx()
    assert relerr(Lx, r.Lx()) < TOL_L
    B = problems.demo_rhs(p["n"], nrhs, True)
    Bp = B[sym["Perm"]]
    Xg = L.super_ltsolve(L.super_lsolve(Bp))
    Xr = r.super_solve(r.super_solve(Bp, True), False)
    assert relerr(Xg, Xr) < 1e-9
    X = np.empty_like(Xg)
    X[sym["Perm"]] = Xg
    assert problems.backward_error(p, X, B) <= 10 * problems.backward_error(p, r.solve(B), B) + 1e-18


def _grid_problem(grid):
    from suitesparse_amd import problems, symbolic
    p = problems.laplacian_3d(grid)
    perm = symbolic.nd_grid_order((grid, grid, grid), 1)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], perm)
    Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
    return p, sym, Sp, Si, Sx


@pytest.mark.parametrize("grid", [60, 150])
def test_full_size_properties(engine, grid):
    """BASELINE-size runs (150^3 = config 3, the bench workload) checked through size-independent properties:
    b = S*1 solves back to 1 with a backward error at rounding level, the solve is linear, and a second
    factorization of the same matrix reproduces the first one to the atomics' summation-order noise."""
    import scipy.sparse as sp
    import torch
    p, sym, Sp, Si, Sx = _grid_problem(grid)
    n = p["n"]
    S = sp.csc_matrix((Sx, Si, Sp), shape=(n, n))
    A = (S + S.T - sp.diags(S.diagonal())).tocsr()
    L = engine.SupernodalFactor(sym)
    L.set_matrix(Sp, Si, Sx)
    assert L.numeric() == 0 and L.minor == n
    ones = np.ones(n)
    b = A @ ones
    rng = np.random.default_rng(grid)
    c = rng.standard_normal(n)
    B = np.stack([b, c, 2.0 * b - 3.0 * c], axis=1)            # n x 3, third column = combination of the others
    X = torch.from_numpy(np.asfortranarray(B).T.copy()).cuda()  # column-major n x 3 on the device
    L.solve_dev(X.data_ptr(), 3, n, lower=True)
    L.solve_dev(X.data_ptr(), 3, n, lower=False)
    Xh = X.cpu().numpy().T                                       # n x 3
    anorm = abs(A).sum(axis=1).max()
    for j in range(3):
        r = A @ Xh[:, j] - B[:, j]
        assert np.abs(r).max() / (anorm * np.abs(Xh[:, j]).max() + np.abs(B[:, j]).max()) < 1e-14
    assert np.abs(Xh[:, 0] - 1.0).max() < 1e-9
    assert np.abs(Xh[:, 2] - (2.0 * Xh[:, 0] - 3.0 * Xh[:, 1])).max() < 1e-9 * np.abs(Xh).max()
    # refactorization: same structure, values equal up to summation order
    Y0 = X.clone()
    assert L.numeric() == 0
    X2 = torch.from_numpy(np.asfortranarray(B).T.copy()).cuda()
    L.solve_dev(X2.data_ptr(), 3, n, lower=True)
    L.solve_dev(X2.data_ptr(), 3, n, lower=False)
    assert float((X2 - Y0).abs().max()) < 1e-9 * float(Y0.abs().max())


def _named_problem(kind, grid, ordering="nd"):
    from suitesparse_amd import problems, symbolic
    p = {"lap2d": problems.laplacian_2d, "lap3d": problems.laplacian_3d, "elast": problems.elasticity27,
         "helm": problems.helmholtz_3d}[kind](grid)
    if ordering == "metis":
        # the permutation the reference's own analysis picks (cholmod_l_analyze with CHOLMOD_METIS,
        # CHOLMOD/Cholesky/cholmod_analyze.c:629-684): the structure CHOLMOD would hand to the engine
        from oracle import refcholmod
        r = refcholmod.RefCholmod()
        r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
        r.analyze(ordering="metis")
        perm = r.symbolic()["Perm"]
        r.close()
    else:
        perm = symbolic.nd_grid_order(p["shape"], p["dof"])
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], perm)
    Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
    return p, sym, Sp, Si, Sx


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,grid,nrhs,ordering", [("lap2d", 2000, 3, "nd"), ("elast", 80, 3, "nd"), ("helm", 100, 64, "nd"),
                                                     ("lap2d", 2000, 3, "metis")])
def test_full_size_properties_of_the_other_configs(engine, kind, grid, nrhs, ordering, have_ref):
    """BASELINE configs 2, 4 and 5 at their stated sizes (2D 2000^2 -- on the geometric ordering and, as the config
    states it, on the reference's METIS ordering; 27-point 3-dof 80^3; complex Hermitian Helmholtz
    100^3 with 64 right-hand sides), checked through size-independent properties: the backward error
    ||A x - b||_inf / (||A||_inf ||x||_inf + ||b||_inf) of every column at rounding level, b = A*1 solves back to 1,
    the solve is linear (last column = a combination of the first two), and a refactorization reproduces the
    solution to the atomics' summation-order noise."""
    import scipy.sparse as sp
    import torch
    if ordering == "metis" and not have_ref:
        pytest.skip("oracle/_ref (the reference's METIS) not built")
    p, sym, Sp, Si, Sx = _named_problem(kind, grid, ordering)
    n = p["n"]
    cplx = np.iscomplexobj(Sx)
    S = sp.csc_matrix((Sx, Si, Sp), shape=(n, n))
    A = (S + S.conj().T - sp.diags(S.diagonal().real)).tocsr()
    L = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
    L.set_matrix(Sp, Si, Sx)
    assert L.numeric() == 0 and L.minor == n
    rng = np.random.default_rng(grid)
    dt = np.complex128 if cplx else np.float64
    B = np.empty((n, nrhs), dtype=dt)
    B[:, 0] = A @ np.ones(n)
    for j in range(1, nrhs - 1):
        B[:, j] = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    B[:, nrhs - 1] = 2.0 * B[:, 0] - 3.0 * B[:, 1]

    def solve():
        X = torch.from_numpy(np.ascontiguousarray(B.T).view(np.float64).copy()).cuda()   # column-major n x nrhs
        L.solve_dev(X.data_ptr(), nrhs, n, lower=True)
        L.solve_dev(X.data_ptr(), nrhs, n, lower=False)
        return X
    X = solve()
    Xh = X.cpu().numpy().reshape(nrhs, -1)
    Xh = (Xh.view(np.complex128) if cplx else Xh).T                                      # n x nrhs
    anorm = abs(A).sum(axis=1).max()
    R = A @ Xh - B
    for j in range(nrhs):
        assert np.abs(R[:, j]).max() / (anorm * np.abs(Xh[:, j]).max() + np.abs(B[:, j]).max()) < 1e-13, j
    assert np.abs(Xh[:, 0] - 1.0).max() < 1e-8
    assert np.abs(Xh[:, -1] - (2.0 * Xh[:, 0] - 3.0 * Xh[:, 1])).max() < 1e-9 * np.abs(Xh).max()
    assert L.numeric() == 0
    X2 = solve()
    assert float((X2 - X).abs().max()) < 1e-9 * float(X.abs().max())


def test_not_posdef_minor_at_scale(engine):
    """one negated diagonal entry at a known permuted column k of a 60^3 Laplacian: the pivot of column k is
    the first that fails, so L->minor = k exactly and the columns from k on are zero (worker :1084-1146)."""
    p, sym, Sp, Si, Sx = _grid_problem(60)
    n = p["n"]
    k = int(sym["Super"][sym["nsuper"] * 2 // 3]) + 1
    Sx = Sx.copy()
    assert Si[Sp[k]] == k
    Sx[Sp[k]] = -Sx[Sp[k]]
    L = engine.SupernodalFactor(sym)
    st = L.super_numeric(Sp, Si, Sx)
    assert st == 1 and L.minor == k
    Lx = L.Lx()
    s = int(np.searchsorted(sym["Super"], k, side="right") - 1)
    nsrow = int(sym["Lpi"][s + 1] - sym["Lpi"][s])
    start = int(sym["Lpx"][s]) + nsrow * (k - int(sym["Super"][s]))
    assert not Lx[start:].any()                                  # column k of s and everything after it
    assert Lx[:start].any()


@pytest.mark.parametrize("kind,size,nrhs", [("lap3d", 16, 1), ("lap2d", 60, 4), ("elast", 6, 2)])
def test_solve_with_permutation_vs_reference(engine, kind, size, nrhs, have_ref):
    """mi355_solve (Y = P*B, both sweeps, X = P'*Y on the device) against the reference's cholmod_l_solve
    (CHOLMOD/Cholesky/cholmod_solve.c:656-717), plus every other `sys` against its definition."""
    if not have_ref:
        pytest.skip("oracle/_ref not available on this box")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d,
         "elast": problems.elasticity27}[kind](size)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze("metis")
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    sym = r.symbolic()
    perm = np.asarray(sym["Perm"], dtype=np.int64)
    L = engine.SupernodalFactor(sym)
    assert L.super_numeric(Sp, Si, Sx) == 0
    L.set_perm(perm)
    B = problems.demo_rhs(p["n"], nrhs)
    if B.ndim == 1:
        B = B[:, None]
    Xr = r.solve(B)
    Xg = L.solve(B, 0)                                   # CHOLMOD_A
    assert relerr(Xg, Xr) < 1e-9
    assert problems.backward_error(p, Xg, B) <= 10 * problems.backward_error(p, Xr, B) + 1e-18
    # the other systems, from their definitions (cholmod_solve.c:14-24)
    Yl = L.super_lsolve(B)
    assert relerr(L.solve(B, 4), Yl) < 1e-12             # L   (no permutation)
    assert relerr(L.solve(B, 2), Yl) < 1e-12             # LD  (D = I for LL')
    Yt = L.super_ltsolve(B)
    assert relerr(L.solve(B, 5), Yt) < 1e-12             # Lt
    assert relerr(L.solve(B, 3), Yt) < 1e-12             # DLt
    assert relerr(L.solve(B, 1), L.super_ltsolve(Yl)) < 1e-12   # LDLt = both sweeps, no permutation
    assert np.array_equal(L.solve(B, 6), B)              # D
    assert np.array_equal(L.solve(B, 7), B[perm])        # P:  x = P b
    Z = np.empty_like(B)
    Z[perm] = B
    assert np.array_equal(L.solve(B, 8), Z)              # Pt: x = P' b
    # real L, complex B: the reference's "dual" path (cholmod_solve.c:668) = two real solves
    Bc = B + 1j * B[::-1]
    Xc = L.solve(Bc, 0)
    assert np.iscomplexobj(Xc)
    assert relerr(Xc, L.solve(Bc.real.copy(), 0) + 1j * L.solve(Bc.imag.copy(), 0)) < 1e-13
    assert np.array_equal(L.solve(Bc, 7), Bc[perm])
    # natural ordering after set_perm(None)
    L.set_perm(None)
    assert relerr(L.solve(B, 0), L.super_ltsolve(Yl)) < 1e-12
    with pytest.raises(engine.Mi355Error):
        L.solve(B, 9)
    with pytest.raises(engine.Mi355Error):
        L.set_perm(np.zeros(p["n"], dtype=np.int64))


def test_solve_complex_with_permutation(engine, have_ref):
    if not have_ref:
        pytest.skip("oracle/_ref not available on this box")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = problems.helmholtz_3d(10)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze("metis")
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    sym = r.symbolic()
    L = engine.SupernodalFactor(sym, xtype=2)
    assert L.super_numeric(Sp, Si, Sx) == 0
    L.set_perm(sym["Perm"])
    rng = np.random.default_rng(3)
    B = rng.standard_normal((p["n"], 3)) + 1j * rng.standard_normal((p["n"], 3))
    assert relerr(L.solve(B, 0), r.solve(B)) < 1e-9
    Br = rng.standard_normal((p["n"], 2))                      # complex L, real B: zero imaginary part on the way in
    assert relerr(L.solve(Br, 0), L.solve(Br + 0j, 0)) < 1e-14


@pytest.mark.parametrize("kind,size,beta", [("lap3d", 14, 0.0), ("helm", 10, 0.5), ("elast", 5, 0.0), ("lap2d", 50, 1.0)])
def test_device_side_permutation_matches_host_permutation(engine, kind, size, beta):
    """mi355_factor_set_matrix_perm (A + Perm, S = tril(P A P') placed by the assembly kernel)
    against mi355_factor_set_matrix on the host-permuted S, and against the oracle"""
    from oracle import port
    from suitesparse_amd import problems, symbolic
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d, "helm": problems.helmholtz_3d,
         "elast": problems.elasticity27}[kind](size)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
    cplx = np.iscomplexobj(Sx)
    so, mo, Lo = port.super_numeric(sym, Sp, Si, Sx, beta=beta)
    assert so == 0
    L1 = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
    L1.set_matrix(Sp, Si, Sx)
    assert L1.numeric(beta=beta) == 0
    L2 = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
    L2.set_matrix_perm(p["Ap"], p["Ai"], p["Ax"], sym["Perm"])
    assert L2.numeric(beta=beta) == 0
    assert relerr(L2.Lx(), L1.Lx()) < 1e-13
    assert relerr(L2.Lx(), Lo) < TOL_L
    # refactorization from the staged A; identity permutation; a bad permutation
    assert L2.numeric(beta=beta) == 0 and relerr(L2.Lx(), Lo) < TOL_L
    L3 = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
    L3.set_matrix_perm(Sp, Si, Sx, None)
    assert L3.numeric(beta=beta) == 0 and relerr(L3.Lx(), Lo) < TOL_L
    bad = np.array(sym["Perm"]).copy(); bad[0] = bad[1]
    with pytest.raises(engine.Mi355Error):
        L3.set_matrix_perm(p["Ap"], p["Ai"], p["Ax"], bad)




@pytest.mark.parametrize("kind,size", [("lap3d", 11), ("helm", 8), ("lap2d", 40)])
def test_supernodal_to_simplicial_on_the_device(engine, kind, size, have_ref):
    """mi355_factor_to_simplicial against cholmod_l_change_factor of the reference applied to
    the SAME supernodal values (the GPU's Lx handed to the reference factor): pattern and values bit-exact,
    LL' and LDL'"""
    if not have_ref:
        pytest.skip("oracle/_ref not available on this box")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = {"lap3d": problems.laplacian_3d, "helm": problems.helmholtz_3d, "lap2d": problems.laplacian_2d}[kind](size)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze("metis")
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    cplx = np.iscomplexobj(Sx)
    L = engine.SupernodalFactor(r.symbolic(), xtype=2 if cplx else 1)
    assert L.super_numeric(Sp, Si, Sx) == 0
    r.set_Lx(L.Lx())
    for to_ll in (True, False):
        got, want = L.to_simplicial(to_ll), r.to_simplicial(to_ll)
        for a, b, name in zip(got, want, ("Lp", "Li", "Lnz", "Lx")):
            assert a.shape == b.shape and np.array_equal(a, b), (to_ll, name)


@pytest.mark.parametrize("kind,size", [("lap3d", 12), ("helm", 9), ("elast", 5)])
def test_rcond_on_the_device_factor(engine, kind, size, have_ref):
    """mi355_factor_rcond against cholmod_l_rcond of the reference's factor, the
    not-positive-definite and empty cases as cholmod_rcond.c:73-80"""
    if not have_ref:
        pytest.skip("oracle/_ref not available on this box")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = {"lap3d": problems.laplacian_3d, "helm": problems.helmholtz_3d, "elast": problems.elasticity27}[kind](size)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze("metis")
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    cplx = np.iscomplexobj(Sx)
    L = engine.SupernodalFactor(r.symbolic(), xtype=2 if cplx else 1)
    with pytest.raises(engine.Mi355Error):
        L.rcond()                                   # nothing factorized yet
    assert L.super_numeric(Sp, Si, Sx) == 0
    want = r.rcond()
    assert 0 < want < 1 and abs(L.rcond() - want) <= 1e-12 * want
    # not positive definite: 0
    Sx2 = np.array(Sx).copy()
    Sx2[Sp[p["n"] // 2]] = -3.0
    assert L.super_numeric(Sp, Si, Sx2) == 1
    assert L.rcond() == 0.0
    empty = dict(n=0, nsuper=0, ssize=0, xsize=0, Super=np.zeros(1, np.int64), Lpi=np.zeros(1, np.int64),
                 Lpx=np.zeros(1, np.int64), Ls=np.zeros(0, np.int64))
    assert engine.SupernodalFactor(empty).rcond() == 1.0


@pytest.mark.parametrize("kind,size", [("lap3d", 18), ("helm", 14), ("lap2d", 70)])
def test_every_solve_path_against_the_oracle(engine, kind, size):
    """the sweeps have four code paths -- scalar or MFMA kernels (tuning key 9), one launch pair per 64-column
    block or solve groups of four blocks (key 10) -- and defaults that pick by nrhs and xtype: every path, for
    few and many right-hand sides, against the oracle's sweeps on the oracle's factor"""
    from oracle import port
    from suitesparse_amd import problems, symbolic
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d, "helm": problems.helmholtz_3d}[kind](size)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
    cplx = np.iscomplexobj(Sx)
    so, mo, Lo = port.super_numeric(sym, Sp, Si, Sx)
    assert so == 0
    L = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
    assert L.super_numeric(Sp, Si, Sx) == 0
    rng = np.random.default_rng(7)
    try:
        for nrhs in (1, 3, 7, 20):
            B = rng.standard_normal((p["n"], nrhs)) + (1j * rng.standard_normal((p["n"], nrhs)) if cplx else 0)
            Yo = port.super_solve(sym, Lo, B, lower=True)
            Xo = port.super_solve(sym, Lo, Yo, lower=False)
            for mma_min in (1, 1000):
                for groups in (0, 1):
                    engine.set_tuning(9, mma_min); engine.set_tuning(10, groups)
                    assert relerr(L.super_lsolve(B), Yo) < 1e-10, (nrhs, mma_min, groups)
                    assert relerr(L.super_ltsolve(Yo), Xo) < 1e-10, (nrhs, mma_min, groups)
    finally:
        engine.set_tuning(9, 5); engine.set_tuning(10, 1)


@pytest.mark.parametrize("cplx,n0,n1,nrhs,wgs", [(False, 300, 200, 20, 256), (False, 600, 150, 70, 256), (False, 100, 330, 5, 256),
                                                 (True, 300, 90, 7, 256), (True, 130, 200, 40, 256), (True, 70, 100, 70, 256),
                                                 (False, 600, 150, 70, 3), (True, 300, 90, 7, 2), (False, 1500, 400, 9, 5)])
def test_wide_multi_rhs_sweeps(engine, cplx, n0, n1, nrhs, wgs):
    """the wide sweeps (tuning key 23; k_solve_fwd_wide / k_solve_bwd_wide): the rows below a solve group of more than
    64 columns take all its columns in one pass -- contraction depth 256 real / 128 complex, ragged last block and last
    row tile, several passes over the right-hand sides (more than 64 real, more than 64 complex) -- on a hand-built
    two-supernode structure whose first supernode has n0 columns and scattered rows of the second: against the
    oracle's sweeps, and against the per-block kernels (key 23 off) on the same factor.  wgs (key 24): workgroups the tiles
    of a step are dealt to -- few of them give shares of several segments that run over task boundaries"""
    from oracle import port
    engine.set_tuning(24, wgs)
    n = n0 + n1
    extra = np.array(sorted(set(range(n0, n, 3)) | set(range(n0 + 1, n, 7)) | {n - 1}), dtype=np.int64)
    rows0 = np.concatenate([np.arange(n0), extra])
    rows1 = np.arange(n0, n)
    sym = dict(n=n, nsuper=2, Super=np.array([0, n0, n]), Lpi=np.array([0, len(rows0), len(rows0) + n1]),
               Lpx=np.array([0, len(rows0) * n0, len(rows0) * n0 + n1 * n1]), Ls=np.concatenate([rows0, rows1]))
    sym["ssize"] = len(sym["Ls"])
    sym["xsize"] = int(sym["Lpx"][-1])
    sym["maxcsize"] = len(extra) ** 2
    rng = np.random.default_rng(n0 + nrhs)
    rnd = (lambda *sh: rng.standard_normal(sh) + 1j * rng.standard_normal(sh)) if cplx else (lambda *sh: rng.standard_normal(sh))
    A = np.zeros((n, n), dtype=np.complex128 if cplx else np.float64)
    A[np.ix_(rows0, np.arange(n0))] = rnd(len(rows0), n0)
    A[n0:, n0:] = rnd(n1, n1)
    A = np.tril(A, -1)
    A = A + A.conj().T
    A += np.eye(n) * (np.abs(A).sum(1).max() + 1)
    import scipy.sparse as sp
    S = sp.csc_matrix(np.tril(A))
    S.sort_indices()
    Sp, Si, Sx = S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data
    L = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
    assert L.super_numeric(Sp, Si, Sx) == 0
    so, mo, Lo = port.super_numeric(sym, Sp, Si, Sx)
    assert so == 0
    B = rnd(n, nrhs)
    Yo = port.super_solve(sym, Lo, B, lower=True)
    Xo = port.super_solve(sym, Lo, Yo, lower=False)
    try:
        for wide in (1, 0):
            engine.set_tuning(23, wide)
            assert relerr(L.super_lsolve(B), Yo) < 1e-10, wide
            assert relerr(L.super_ltsolve(Yo), Xo) < 1e-10, wide
    finally:
        engine.set_tuning(23, 1); engine.set_tuning(24, 256)


@pytest.mark.parametrize("kind,size,nrhs", [("lap3d", 20, 8), ("lap3d", 20, 23), ("lap2d", 90, 64), ("elast", 8, 70),
                                            ("lap3d", 28, 130)])
def test_multi_rhs_solve_on_mfma(engine, kind, size, nrhs, have_ref):
    """nrhs >= 8 routes the off-diagonal parts of both sweeps to the MFMA kernels (k_solve_fwd_mma /
    k_solve_bwd_mma; chunks of 64 right-hand sides, ragged last chunk): against the reference's sweeps."""
    if not have_ref:
        pytest.skip("oracle/_ref not available on this box")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d,
         "elast": problems.elasticity27}[kind](size)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze("metis")
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    sym = r.symbolic()
    L = engine.SupernodalFactor(sym)
    assert L.super_numeric(Sp, Si, Sx) == 0
    rng = np.random.default_rng(nrhs)
    B = rng.standard_normal((p["n"], nrhs))
    Yg, Yr = L.super_lsolve(B), r.super_solve(B, True)
    assert relerr(Yg, Yr) < 1e-9
    Xg, Xr = L.super_ltsolve(Yr), r.super_solve(Yr, False)
    assert relerr(Xg, Xr) < 1e-9
    # column j of a multi-RHS solve equals the single-RHS solve of that column (same kernels or not)
    j = nrhs // 2
    assert relerr(L.super_ltsolve(L.super_lsolve(B[:, j])), L.super_ltsolve(L.super_lsolve(B))[:, j]) < 1e-10


def test_too_large_is_reported_not_wrapped(engine):
    """SURVEY 8 row a9 on the device library (body shared with the CPU build's test)"""
    import test_abi_on_cpu as t
    t.test_too_large_is_reported_not_wrapped(engine)


def test_blas_timer_statistics(engine):
    """SURVEY 8 row a10 on the device library: per-launch HIP events, counts and times"""
    import test_abi_on_cpu as t
    t.test_blas_timer_statistics(engine)


# ---- written while GPU use was closed (round 2): kept at the end of the file so that a first-run surprise hides nothing


@pytest.mark.parametrize("name", ["lap3d_8_npd200", "elast27_4_npd100", "lap2d_16_npd0", "tcov_c30lo_singular"])
def test_device_side_permutation_of_a_matrix_that_is_not_posdef(engine, name):
    """mi355_factor_set_matrix_perm + a bad pivot: the replay of the failing supernode assembles the columns of S below
    the limit from ALL columns of A (the limit is on the permuted target column) -- status, L->minor, zero pattern and
    values as the reference left them (fixture), same as with the host-permuted S"""
    import scipy.sparse as sp
    from conftest import load_golden, sym_of
    g = load_golden(name)
    n = g["n"]
    A = sp.csc_matrix((g["Ax"], g["Ai"], g["Ap"]), shape=(n, n))
    if g["stype"] > 0:
        A = A.conj().T.tocsc()
    A = sp.tril(A, format="csc")
    A.sort_indices()
    cplx = g["xtype"] == 2
    L = engine.SupernodalFactor(sym_of(g), xtype=2 if cplx else 1)
    L.set_matrix_perm(A.indptr.astype(np.int64), A.indices.astype(np.int64), np.ascontiguousarray(A.data), g["Perm"])
    for _ in range(2):
        assert L.numeric(beta=g["beta"]) == g["status"] == 1
        assert L.minor == g["minor"]
        Lx = L.Lx()
        assert np.array_equal(Lx == 0, g["Lx"] == 0)
        assert relerr(Lx, g["Lx"]) < TOL_L
    L.close()


@pytest.mark.parametrize("name,chunk", [("lap3d_8_metis", 2000), ("helm3d_6_metis", 500), ("elast27_4_metis", 4000),
                                         ("lap2d_16_metis", 300), ("demo_bcsstk02", 1 << 25), ("lap3d_8_npd200", 2000),
                                         ("lap2d_16_npd0", 300), ("tcov_c30lo_singular", 50)])
def test_numeric_host_copies_under_the_factorization(engine, name, chunk):
    """mi355_factor_numeric_host: L->x on the host filled chunk by chunk while the factorization runs (tuning key 18 =
    entries per chunk, small here so that a fixture has many chunks) must be what mi355_factor_numeric +
    mi355_factor_get_Lx give (to rounding of the atomically added updates) -- positive definite or not (the failing case gets one plain copy after the
    replay) -- and on a refactorization; the plan tiles Lx, is ordered, and most chunks are ready before the last launch"""
    from conftest import load_golden, sym_of
    import scipy.sparse as sp
    g = load_golden(name)
    n = g["n"]
    cplx = g["xtype"] == 2
    A = sp.csc_matrix((g["Ax"], g["Ai"], g["Ap"]), shape=(n, n))
    if g["stype"] > 0:
        A = A.conj().T.tocsc()
    A = sp.tril(A, format="csc")
    A.sort_indices()
    engine.set_tuning(18, chunk)
    try:
        L = engine.SupernodalFactor(sym_of(g), xtype=2 if cplx else 1)
        L.set_matrix_perm(A.indptr.astype(np.int64), A.indices.astype(np.int64), np.ascontiguousarray(A.data), g["Perm"])
        plan = L.host_copy_plan()
        xsize = int(g["Lpx"][-1]) if len(g["Lpx"]) else 0
        assert plan[0, 0] == 0 and plan[-1, 1] == L.xsize and np.array_equal(plan[1:, 0], plan[:-1, 1])
        assert np.all(plan[:, 1] > plan[:, 0]) and np.all(plan[:, 2] >= 0)
        starts = set(int(v) for v in g["Lpx"]) | {L.xsize}
        assert all(int(v) in starts for v in plan[:, 0])            # whole supernodes
        if chunk < L.xsize // 8:
            assert len(plan) >= 4 and plan[:, 2].min() < plan[:, 2].max()
        for _ in range(2):
            st = L.numeric(beta=g["beta"])
            want = L.Lx().copy()
            # leave ANOTHER factor on the device: a chunk that left before it was recomputed must not look right
            L.numeric(beta=g["beta"] + 0.75)
            assert not np.array_equal(L.Lx().view(np.float64), want.view(np.float64))
            st2, got = L.numeric_host(beta=g["beta"])
            assert st == st2 == g["status"] and L.minor == g["minor"]
            # (not bit for bit: the scatter updates of one level add with atomics, whose order differs from run to run on
            # the device; a chunk that left too early would hold the other factor, which differs in the first digit)
            assert np.array_equal(got == 0, want == 0) and relerr(got, want) < 1e-12
            assert relerr(got, g["Lx"]) < TOL_L
            early = L.stats()["copy_early_entries"]
            if st == 0 and len(plan) >= 4:
                assert 0 < early < L.xsize
            if st != 0:
                assert early == 0
        L.close()
    finally:
        engine.set_tuning(18, 32 << 20)


@pytest.mark.parametrize("name", ["lap3d_8_metis", "elast27_4_metis", "lap2d_16_metis", "helm3d_6_metis", "tcov_c30lo_singular"])
def test_deterministic_mode_gives_identical_bits(engine, name):
    """tuning key 27: the scatter updates of a level run in rounds of at most one descendant per ancestor, so every entry
    of L receives its additions in one fixed order -- three factorizations of the same matrix (plain, again, and with the
    copy to the host under the factorization) are equal BIT FOR BIT, as two runs of the reference are; and the factor is
    still the reference's to 1e-10.  (Without the key the FP64 atomics of concurrent descendants arrive in any order.)"""
    from conftest import load_golden, sym_of
    g = load_golden(name)
    cplx = g["xtype"] == 2
    engine.set_tuning(27, 1)
    try:
        L = engine.SupernodalFactor(sym_of(g), xtype=2 if cplx else 1)
        L.set_matrix(g["Sp"], g["Si"], g["Sx"])
        st = L.numeric(beta=g["beta"])
        first = L.Lx().copy()
        assert st == g["status"] and L.minor == g["minor"]
        L.numeric(beta=g["beta"] + 0.5)                     # something else in between
        assert L.numeric(beta=g["beta"]) == st
        second = L.Lx().copy()
        st3, third = L.numeric_host(beta=g["beta"])
        assert st3 == st
        assert np.array_equal(first.view(np.uint64), second.view(np.uint64))
        assert np.array_equal(first.view(np.uint64), np.ascontiguousarray(third).view(np.uint64))
        assert np.array_equal(first == 0, g["Lx"] == 0) and relerr(first, g["Lx"]) < TOL_L
        L.close()
    finally:
        engine.set_tuning(27, 0)


@pytest.mark.parametrize("rows", [128, 64, 32, 16])
@pytest.mark.parametrize("case", ["dense130", "dense300", "dense700", "dense1100", "dense1733", "lap3d_8_metis", "elast27_4_metis",
                                  "demo_bcsstk02", "lap3d_8_npd200", "tcov_r5lo", "arrow800", "blocks1100"])
def test_panel_groups(engine, case, rows):
    """tuning key 28 (schedule.hpp, STEP_PDIAG / STEP_PTRSM): the panel chain 256 columns at a time -- k_panel_diag on the
    diagonal window, k_panel_trsm<rows> on the rows below it -- gives the factor of the per-block chain: dense supernodes
    with ragged windows against numpy, fixtures (incl. one that is not positive definite: minor, status, zero pattern)
    and random ragged structures against the reference's values / the oracle"""
    from conftest import load_golden, sym_of
    engine.set_tuning(28, rows)
    try:
        if case.startswith("dense"):
            n = int(case[5:])
            rng = np.random.default_rng(n)
            M = rng.standard_normal((n, n))
            A = M @ M.T + n * np.eye(n)
            sym = dict(n=n, nsuper=1, ssize=n, xsize=n * n, Super=np.array([0, n]), Lpi=np.array([0, n]),
                       Lpx=np.array([0, n * n]), Ls=np.arange(n))
            Sp = np.arange(n + 1, dtype=np.int64) * n
            Si = np.tile(np.arange(n, dtype=np.int64), n)
            Sx = np.ascontiguousarray(A.T).reshape(-1)
            L = engine.SupernodalFactor(sym)
            assert L.super_numeric(Sp, Si, Sx) == 0
            Lg = L.Lx().reshape(n, n).T
            assert np.all(np.triu(Lg, 1) == 0)
            assert relerr(np.tril(Lg), np.linalg.cholesky(A)) < TOL_L
            B = rng.standard_normal((n, 5))
            assert relerr(L.super_ltsolve(L.super_lsolve(B)), np.linalg.solve(A, B)) < 1e-9
            # a bad pivot inside a window: LAPACK's info, the reference's minor and zero tail
            k = (2 * n) // 3
            Sx2 = Sx.copy()
            Sx2[k * n + k] = -1.0
            assert L.super_numeric(Sp, Si, Sx2) == 1 and L.minor == k
            Lb = L.Lx().reshape(n, n).T
            assert np.all(Lb[:, k:] == 0)
            assert relerr(np.tril(Lb[:, :k]), np.linalg.cholesky(A)[:, :k]) < TOL_L
            L.close()
        elif case[:5] in ("arrow", "block"):
            from oracle import port
            import test_fuzz_schedule as tf
            kind, n = ("arrow", int(case[5:])) if case.startswith("arrow") else ("blocks", int(case[6:]))
            g, _ = tf.build(kind, n, 6 if kind == "arrow" else 8)
            so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
            L = engine.SupernodalFactor(g)
            assert L.super_numeric(g["Sp"], g["Si"], g["Sx"]) == so == 0 and L.minor == mo
            Lx = L.Lx()
            for s in range(g["nsuper"]):
                a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
                nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
                got, want = Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
                low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
                assert np.abs(got[low] - want[low]).max() <= 1e-10 * np.abs(Lo).max(), s
            L.close()
        else:
            g = load_golden(case)
            L = engine.SupernodalFactor(sym_of(g))
            L.set_matrix(g["Sp"], g["Si"], g["Sx"])
            st = L.numeric(beta=g["beta"])
            assert st == g["status"] and L.minor == g["minor"]
            Lx = L.Lx()
            assert np.array_equal(Lx == 0, g["Lx"] == 0) and relerr(Lx, g["Lx"]) < TOL_L
            st2, Lh = L.numeric_host(beta=g["beta"])              # the copy under the factorization sees the group steps too
            assert st2 == st and relerr(np.ascontiguousarray(Lh), g["Lx"]) < TOL_L
            L.close()
    finally:
        engine.set_tuning(28, 64)


@pytest.mark.parametrize("n0,n1,nrhs,wgs", [(600, 150, 70, 256), (300, 200, 20, 3), (1500, 400, 33, 5)])
def test_wide_forward_sweep_with_three_register_sets(engine, n0, n1, nrhs, wgs):
    """tuning key 29 = 3: the forward wide kernel keeps three chunks of L in flight per wave instead of two (a deeper ring
    for the memory latency under load); same cases, same answers"""
    engine.set_tuning(29, 3)
    try:
        test_wide_multi_rhs_sweeps(engine, False, n0, n1, nrhs, wgs)
    finally:
        engine.set_tuning(29, 2)


@pytest.mark.parametrize("cplx,n0,n1,nrhs,keep", [(False, 1500, 400, 9, 12), (False, 2700, 300, 70, 12), (True, 1100, 200, 7, 12), (True, 900, 330, 40, 40),
                                                  (False, 1500, 400, 33, 0)])
def test_sweeps_with_look_ahead(engine, cplx, n0, n1, nrhs, keep):
    """tuning key 30 (12 = on; off by default): the steps of a supernode of several solve groups run on two streams -- the group kernels and the
    rows of the next group on one, the rows beyond on the other, ordered by events (engine.hip, solve_dev).  A first supernode
    of 5 .. 11 groups (more than the ring of four events holds) with scattered rows of a second one below it; `keep` = CUs
    kept free (0 = off: the lists are not split); same answers as the oracle's sweeps, on the factor the same test builds"""
    engine.set_tuning(30, keep)
    try:
        test_wide_multi_rhs_sweeps(engine, cplx, n0, n1, nrhs, 256)
    finally:
        engine.set_tuning(30, 0)


@pytest.mark.parametrize("cplx,n0,n1,nrhs,most", [(False, 600, 150, 70, 4), (False, 1500, 400, 9, 64), (True, 300, 90, 7, 4), (True, 700, 100, 40, 1),
                                                  (False, 330, 200, 20, 0), (False, 150, 100, 33, 4), (False, 600, 150, 3, 4), (False, 330, 90, 1, 4),
                                                  (False, 700, 100, 4, 2)])
def test_group_steps_with_explicit_inverses(engine, cplx, n0, n1, nrhs, most):
    """tuning key 31 (default 4): the steps of at most `most` solve groups run k_solve_group_inv -- ten independent products with
    the explicit inverse of the group's 256 x 256 diagonal block, combined through a ticketed scratch tile -- instead of the chain
    of k_solve_group_mma.  Ragged groups (330 = 256 + 74, 150 = 2 blocks + 22), real and complex, more right-hand sides than one
    slab, 1 .. 4 column-major right-hand sides (the scalar sweeps: one slab per column), the key off; and a SECOND factorization of other values on the same handle: the inverses are rebuilt (a stale image
    would give the first matrix's answers)"""
    from oracle import port
    engine.set_tuning(31, most)
    try:
        test_wide_multi_rhs_sweeps(engine, cplx, n0, n1, nrhs, 256)
        # the same structure, two sets of values on one handle
        n = n0 + n1
        extra = np.array(sorted(set(range(n0, n, 3)) | {n - 1}), dtype=np.int64)
        rows0 = np.concatenate([np.arange(n0), extra])
        sym = dict(n=n, nsuper=2, Super=np.array([0, n0, n]), Lpi=np.array([0, len(rows0), len(rows0) + n1]),
                   Lpx=np.array([0, len(rows0) * n0, len(rows0) * n0 + n1 * n1]), Ls=np.concatenate([rows0, np.arange(n0, n)]))
        sym["ssize"] = len(sym["Ls"]); sym["xsize"] = int(sym["Lpx"][-1]); sym["maxcsize"] = len(extra) ** 2
        rng = np.random.default_rng(7 * n0 + nrhs)
        rnd = (lambda *sh: rng.standard_normal(sh) + 1j * rng.standard_normal(sh)) if cplx else (lambda *sh: rng.standard_normal(sh))
        import scipy.sparse as sp
        L = engine.SupernodalFactor(sym, xtype=2 if cplx else 1)
        B = rnd(n, nrhs)
        for scale in (1.0, 3.0):
            A = np.zeros((n, n), dtype=np.complex128 if cplx else np.float64)
            A[np.ix_(rows0, np.arange(n0))] = rnd(len(rows0), n0) * scale
            A[n0:, n0:] = rnd(n1, n1)
            A = np.tril(A, -1)
            A = A + A.conj().T
            A += np.eye(n) * (np.abs(A).sum(1).max() + 1)
            S = sp.csc_matrix(np.tril(A)); S.sort_indices()
            Sp, Si, Sx = S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data
            assert L.super_numeric(Sp, Si, Sx) == 0
            so, mo, Lo = port.super_numeric(sym, Sp, Si, Sx)
            Yo = port.super_solve(sym, Lo, B, lower=True)
            assert relerr(L.super_lsolve(B), Yo) < 1e-10, scale
            assert relerr(L.super_ltsolve(Yo), port.super_solve(sym, Lo, Yo, lower=False)) < 1e-10, scale
        L.close()
    finally:
        engine.set_tuning(31, 4)
