"""CPU tests of the HOST logic of the product: the level schedule (suitesparse_amd/csrc/
schedule.cpp) executed by the CPU emulator tests/helpers/sched_emu.cpp must reproduce the
reference's factor, status, minor, zeroing pattern and solves on the golden fixtures."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, golden_names, load_golden, sym_of, relerr

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def emu():
    E = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    E.emu_super_numeric.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i64p,
                                    i64p, f64p, C.c_double, C.c_int, f64p, f64p, i64p, C.POINTER(C.c_int), i64p]
    E.emu_dinv_size.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64]
    E.emu_dinv_size.restype = C.c_int64
    E.emu_super_solve.argtypes = [C.c_int, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64,
                                  f64p, f64p, C.c_int64, f64p, C.c_int64]
    E.emu_super_solve_grouped.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64,
                                          C.c_int64, f64p, f64p, C.c_int64, f64p, C.c_int64]
    return E


def run_emu(E, g, quick=0):
    a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
    ds = E.emu_dinv_size(*a)
    Lx = np.zeros(g["xsize"])
    D = np.zeros(ds + 1)
    minor = np.zeros(1, dtype=np.int64)
    st = C.c_int(0)
    stats = np.zeros(8, dtype=np.int64)
    rc = E.emu_super_numeric(*a, g["Sp"], g["Si"], np.ascontiguousarray(g["Sx"], dtype=np.float64), g["beta"],
                             quick, Lx, D, minor, C.byref(st), stats)
    assert rc == 0
    return st.value, int(minor[0]), Lx, D


def backward_error_S(g, X):
    """||S x - b|| / (||S|| ||x|| + ||b||) for the permuted system (cholmod_dl_demo.c:614-617)"""
    import scipy.sparse as sp
    n = g["n"]
    S = sp.csc_matrix((g["Sx"], g["Si"], g["Sp"]), shape=(n, n))
    A = (S + S.conj().T - sp.diags(S.diagonal())).tocsr()
    if g["beta"]:
        A = A + g["beta"] * sp.identity(n)
    R = A @ X - g["B"]
    return float(np.abs(R).max() / (abs(A).sum(axis=1).max() * np.abs(X).max() + np.abs(g["B"]).max()))


REAL = [n for n in golden_names() if load_golden(n)["xtype"] == 1]


@pytest.mark.parametrize("cplx", [0, 1])
@pytest.mark.parametrize("n", [100, 200, 321, 700])
def test_tile_tables_rectangular_tiles(emu, n, cplx):
    """tile tables of dense supernodes wide enough for the big-tile kernel, for a real factor (128 x 128 tiles)
    and a complex one (64 x 128 entries per tile): every tile on or below the diagonal exactly once."""
    emu.emu_check_tile_tables.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64,
                                          C.c_int, i64p]
    nb = np.zeros(1, dtype=np.int64)
    emu.emu_set_tuning(28, 0)        # the per-block panel chain (with panel groups a real 321-column supernode has no big update)
    try:
        bad = emu.emu_check_tile_tables(n, 1, np.array([0, n], dtype=np.int64), np.array([0, n], dtype=np.int64),
                                        np.array([0, n * n], dtype=np.int64), np.arange(n, dtype=np.int64), n, n * n,
                                        cplx, nb)
    finally:
        emu.emu_set_tuning(28, 64)
    assert bad == 0
    assert (nb[0] > 0) == (n >= 192 // (2 if cplx else 1) + 64)      # wide enough for a big trailing update


@pytest.mark.parametrize("n,ob,ob2", [(700, 128, 0), (700, 128, 256), (1000, 192, 576), (333, 64, 128), (1300, 512, 1024),
                                      (900, 128, 512), (130, 512, 2048)])
def test_dense_supernode_blocking(emu, n, ob, ob2):
    """one dense supernode through the emulator with small nested block widths (64 < ob < ob2): every entry
    must be updated exactly once by every earlier column (compared with numpy's Cholesky)."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    g = dict(n=n, nsuper=1, ssize=n, xsize=n * n, Super=np.array([0, n], dtype=np.int64),
             Lpi=np.array([0, n], dtype=np.int64), Lpx=np.array([0, n * n], dtype=np.int64),
             Ls=np.arange(n, dtype=np.int64), Sp=np.arange(n + 1, dtype=np.int64) * n,
             Si=np.tile(np.arange(n, dtype=np.int64), n), Sx=np.ascontiguousarray(A.T).reshape(-1), beta=0.0)
    emu.emu_set_tuning(2, ob)
    emu.emu_set_tuning(7, ob2)
    try:
        st, minor, Lx, D = run_emu(emu, g)
    finally:
        emu.emu_set_tuning(2, 256)
        emu.emu_set_tuning(7, 1024)
    assert st == 0 and minor == n
    Lg = Lx.reshape(n, n).T
    assert np.all(np.triu(Lg, 1) == 0)
    assert relerr(np.tril(Lg), np.linalg.cholesky(A)) < 1e-11


@pytest.mark.parametrize("tiny_max", [1, 64, 4096, 1 << 20])
@pytest.mark.parametrize("name", ["lap3d_8_metis", "lap2d_16_metis", "elast27_4_metis", "demo_bcsstk01", "tcov_ex5lo",
                                  "lap3d_8_npd200", "nd_lap2d_20"])
def test_tiny_launch_class(emu, name, tiny_max):
    """tuning key 14 (STEP_TINY, k_gemm_tiny): small update tasks move to launches of their own; the schedule still
    reproduces the reference (status, minor, zero pattern), every task stays inside its arrays (the emulator runs
    check_schedule_bounds, which also holds a tiny step to one tile per task and update modes only)"""
    g = load_golden(name)
    emu.emu_set_tuning(14, tiny_max)
    try:
        a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
        Lx, D = np.zeros(g["xsize"]), np.zeros(emu.emu_dinv_size(*a) + 1)
        minor, st, stats = np.zeros(1, dtype=np.int64), C.c_int(0), np.zeros(8, dtype=np.int64)
        rc = emu.emu_super_numeric(*a, g["Sp"], g["Si"], np.ascontiguousarray(g["Sx"], dtype=np.float64), g["beta"],
                                   0, Lx, D, minor, C.byref(st), stats)
    finally:
        emu.emu_set_tuning(14, 2048)
    assert rc == 0
    assert st.value == g["status"] and int(minor[0]) == g["minor"]
    assert relerr(Lx, g["Lx"]) < 1e-12
    if st.value == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)
    cls = np.zeros(8, dtype=np.int64)
    emu.emu_schedule_classes.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i64p]
    emu.emu_set_tuning(14, tiny_max)
    emu.emu_set_tuning(28, 0)        # count the classes of the per-block chain (a panel group takes its blocks out of them)
    try:
        assert emu.emu_schedule_classes(*a, cls) == 0
    finally:
        emu.emu_set_tuning(14, 2048)
        emu.emu_set_tuning(28, 64)
    assert cls[4] == 1
    nb = np.concatenate([np.minimum(64, w - np.arange(0, w, 64)) for w in np.diff(g["Super"])])
    if st.value == 0:
        assert cls[3] == nb.size and cls[2] == (nb <= 16).sum()      # every block of <= 16 columns is in the wave class
    if tiny_max >= 4096 and g["nsuper"] > 4:
        assert stats[6] > 0 and stats[7] > 0            # tasks really went to tiny launches
    if tiny_max == 1 << 20 and g["nsuper"] > 4:
        assert stats[6] >= stats[2] // 2                # ... most of the update pairs of these small problems


@pytest.mark.parametrize("name", REAL)
def test_schedule_reproduces_reference(emu, name):
    g = load_golden(name)
    st, minor, Lx, D = run_emu(emu, g)
    assert st == g["status"] and minor == g["minor"]
    assert relerr(Lx, g["Lx"]) < 1e-12
    if st == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)
        return
    a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
    n, nrhs = g["B"].shape
    X = np.ascontiguousarray(g["B"].T).reshape(-1).copy()
    assert emu.emu_super_solve(1, *a, Lx, D, nrhs, X, n) == 0
    # the block-inverse diagonal solves differ from substitution by cond(L11)*eps in the
    # forward error (tcov_ex5lo: cond(A) = 6.6e7); the backward error is the contract
    assert relerr(X.reshape(nrhs, n).T, g["Y"]) < 1e-8
    assert emu.emu_super_solve(0, *a, Lx, D, nrhs, X, n) == 0
    Z = X.reshape(nrhs, n).T
    assert relerr(Z, g["Z"]) < 1e-8
    assert backward_error_S(g, Z) <= 10 * backward_error_S(g, g["Z"]) + 1e-18


@pytest.mark.parametrize("kind,size,bwd_div", [("lap3d", 16, 256), ("lap3d", 12, 1), ("lap2d", 60, 4), ("elast", 6, 16),
                                               ("dense", 700, 2)])
def test_solve_groups_cover_every_block_and_tile(emu, kind, size, bwd_div):
    """the grouped sweeps (build_solve_groups: groups of four blocks, per-block update tasks below the group,
    tile tables, backward runs for the scalar and the MFMA kernel) through the emulator against the oracle"""
    from oracle import port
    if kind == "dense":
        n = size
        rng = np.random.default_rng(n)
        M = rng.standard_normal((n, n))
        A = M @ M.T + n * np.eye(n)
        g = dict(n=n, nsuper=1, ssize=n, xsize=n * n, Super=np.array([0, n], dtype=np.int64),
                 Lpi=np.array([0, n], dtype=np.int64), Lpx=np.array([0, n * n], dtype=np.int64),
                 Ls=np.arange(n, dtype=np.int64), Sp=np.arange(n + 1, dtype=np.int64) * n,
                 Si=np.tile(np.arange(n, dtype=np.int64), n), Sx=np.ascontiguousarray(A.T).reshape(-1), beta=0.0,
                 maxcsize=1)
    else:
        from test_multigpu import problem
        g = problem(kind, size)
    st, minor, Lx, D = run_emu(emu, g)
    assert st == 0
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
    n, nrhs = g["n"], 3
    B = np.random.default_rng(1).standard_normal((n, nrhs))
    Yo = port.super_solve(g, Lo, B, lower=True)
    Xo = port.super_solve(g, Lo, Yo, lower=False)
    emu.emu_set_tuning(11, bwd_div); emu.emu_set_tuning(13, bwd_div)
    try:
        for mma in (0, 1):
            X = np.ascontiguousarray(B.T).reshape(-1).copy()
            assert emu.emu_super_solve_grouped(1, mma, *a, Lx, D, nrhs, X, n) == 0
            assert relerr(X.reshape(nrhs, n).T, Yo) < 1e-9
            assert emu.emu_super_solve_grouped(0, mma, *a, Lx, D, nrhs, X, n) == 0
            assert relerr(X.reshape(nrhs, n).T, Xo) < 1e-9
    finally:
        emu.emu_set_tuning(11, 256); emu.emu_set_tuning(13, 4096)


def test_schedule_large_supernodes_vs_oracle(emu, have_ref):
    """two-level blocking (nscol > 256) and many levels: 3D 24^3 through the live reference"""
    if not have_ref:
        pytest.skip("oracle/_ref not available")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = problems.laplacian_3d(24)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze("metis")
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    sym = r.symbolic()
    assert np.diff(sym["Super"]).max() > 256
    g = dict(sym, Sp=Sp, Si=Si, Sx=Sx, beta=0.0)
    st, minor, Lx, _ = run_emu(emu, g)
    assert st == 0 and minor == p["n"]
    assert relerr(Lx, r.Lx()) < 1e-12


def test_task_wider_than_the_tile_fields_is_rejected(emu):
    """TileEntry packs tile row and column into 16 bits each: a task with more than 65536 tile rows (4.19 M rows of
    64, 8.39 M of 128) must be refused (MI355_TOO_LARGE through upload_schedule), never wrapped"""
    emu.emu_check_wide_task.argtypes = [C.c_int64, C.c_int64, C.c_int]
    assert emu.emu_check_wide_task(65536 * 64, 64, 0) == 0
    assert emu.emu_check_wide_task(65536 * 64 + 1, 64, 0) == 1
    assert emu.emu_check_wide_task(65536 * 128, 128, 1) == 0
    assert emu.emu_check_wide_task(65536 * 128 + 1, 128, 1) == 1
