"""GPU cases of the tiny launch classes (tuning key 14: k_gemm_tiny, k_potrf16 with the fused TRSM, the POTRF order
lists) at widths below, at and far above the default.  Written in round 1 without a GPU, green on the MI355X in round 2
(33 cases), after which the classes were measured (profiles/r02_tiny_classes.txt) and switched on by default."""
import numpy as np
import pytest

from conftest import load_golden, sym_of, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import torch
    assert torch.cuda.is_available()
    from suitesparse_amd import engine as e
    return e


@pytest.mark.parametrize("tiny_max", [64, 4096, 1 << 18])
@pytest.mark.parametrize("name", ["lap3d_8_metis", "lap2d_16_metis", "elast27_4_metis", "demo_bcsstk01", "tcov_ex5lo",
                                  "lap3d_8_npd200", "nd_lap2d_20", "lap3d_8_beta"])
def test_tiny_launch_class_golden(engine, name, tiny_max):
    g = load_golden(name)
    engine.set_tuning(14, tiny_max)
    try:
        L = engine.SupernodalFactor(sym_of(g), xtype=g["xtype"])
        st = L.super_numeric(g["Sp"], g["Si"], g["Sx"], beta=(g["beta"], 0.0))
        Lx = L.Lx()
        L.close()
    finally:
        engine.set_tuning(14, 2048)
    assert (st, L.minor) == (g["status"], g["minor"])
    assert relerr(Lx, g["Lx"]) < 1e-10
    if st == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)


@pytest.mark.parametrize("kind,size", [("lap2d", 300), ("lap3d", 30)])
def test_tiny_launch_class_same_factor_as_default(engine, kind, size):
    from suitesparse_amd import problems, symbolic
    p = {"lap2d": problems.laplacian_2d, "lap3d": problems.laplacian_3d}[kind](size)
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], symbolic.nd_grid_order(p["shape"], p["dof"]))
    Sp, Si, Sx = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
    out = []
    for tiny_max in (0, 2048):
        engine.set_tuning(14, tiny_max)
        try:
            L = engine.SupernodalFactor(sym)
            assert L.super_numeric(Sp, Si, Sx) == 0
            out.append(L.Lx())
            L.close()
        finally:
            engine.set_tuning(14, 2048)
    assert relerr(out[1], out[0]) < 1e-12


@pytest.mark.parametrize("n", [1, 5, 16, 17, 33, 80])
def test_small_dense_supernode_through_the_wave_potrf(engine, n):
    """blocks of <= 16 columns go to k_potrf16 (one wave per block): n <= 16 is such a block without rows below it,
    n = 17, 33, 80 end in one after full 64-column (or wider) blocks"""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    sym = dict(n=n, nsuper=1, ssize=n, xsize=n * n, Super=np.array([0, n]), Lpi=np.array([0, n]),
               Lpx=np.array([0, n * n]), Ls=np.arange(n))
    Sp = np.arange(n + 1, dtype=np.int64) * n
    Si = np.tile(np.arange(n, dtype=np.int64), n)
    Sx = np.ascontiguousarray(A.T).reshape(-1)
    engine.set_tuning(14, 4096)
    try:
        L = engine.SupernodalFactor(sym)
        assert L.super_numeric(Sp, Si, Sx) == 0
        Lg = L.Lx().reshape(n, n).T
        b = rng.standard_normal(n)
        x = L.super_ltsolve(L.super_lsolve(b))
        L.close()
    finally:
        engine.set_tuning(14, 2048)
    assert relerr(np.tril(Lg), np.linalg.cholesky(A)) < 1e-11
    assert relerr(x, np.linalg.solve(A, b)) < 1e-9


def test_wave_potrf_reports_the_first_bad_pivot(engine):
    """info from k_potrf16: a negative pivot inside a block of <= 16 columns (status, minor, zero pattern)"""
    from oracle import port
    g = load_golden("lap2d_16_npd0")
    engine.set_tuning(14, 4096)
    try:
        L = engine.SupernodalFactor(sym_of(g))
        st = L.super_numeric(g["Sp"], g["Si"], g["Sx"])
        Lx = L.Lx()
        L.close()
    finally:
        engine.set_tuning(14, 2048)
    assert (st, L.minor) == (g["status"], g["minor"])
    assert relerr(Lx, g["Lx"]) < 1e-10 and np.array_equal(Lx == 0, g["Lx"] == 0)
