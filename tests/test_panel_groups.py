"""Panel groups (tuning key 28, schedule.hpp: STEP_PDIAG / STEP_PTRSM): the panel chain of a supernode taken 256
columns at a time -- one workgroup factorizes the diagonal window (k_panel_diag), one launch solves the rows below it
against the whole window (k_panel_trsm = the forward chain of the solve groups on the panel's rows).  Here on the CPU:
the schedule with the key on, run by the emulator's plain loops (host logic) and by the two kernels from their own
source (emulator key 100 = 3), against the oracle and the reference's fixtures."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, relerr
from test_schedule import emu, run_emu  # noqa: F401
from test_device_source_on_cpu import dense_problem

REAL = [n for n in golden_names() if load_golden(n)["xtype"] == 1]


def _lower_panels_close(g, Lx, Lo, tol=1e-10):
    for s in range(g["nsuper"]):
        a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
        nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
        got, want = Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
        low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
        assert np.abs(got[low] - want[low]).max() <= tol * np.abs(Lo).max(), s


def _with_keys(E, keys, fn):
    defaults = {100: 0, 28: 64, 2: 256, 7: 1024, 14: 2048}
    for k, v in keys.items():
        E.emu_set_tuning(k, v)
    try:
        return fn()
    finally:
        for k in keys:
            E.emu_set_tuning(k, defaults[k])


@pytest.mark.parametrize("name", REAL)
def test_goldens_with_panel_groups_host_loops(emu, name):  # noqa: F811
    """every real fixture with the key on (most supernodes stay below two blocks and keep the per-block steps; the
    wide ones go through the groups): status, minor, L and the zero pattern of the failing cases as the reference"""
    g = load_golden(name)
    st, minor, Lx, D = _with_keys(emu, {28: 64}, lambda: run_emu(emu, g))
    assert (st, minor) == (g["status"], g["minor"])
    assert relerr(Lx, g["Lx"]) < 1e-10
    if st == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)


@pytest.mark.parametrize("n,negate", [(100, None), (129, None), (256, None), (300, None), (330, 200), (520, None), (700, 650),
                                      (1100, None)])
@pytest.mark.parametrize("rows", [128, 16])
def test_dense_supernode_with_panel_groups_host_loops(emu, n, negate, rows):  # noqa: F811
    """one dense supernode of n columns: windows of 256 columns (the last one partial: one block = per-block steps, more =
    a group), the rows below a window, the K = 256 / 1024 updates that follow; a bad pivot inside a window"""
    from oracle import port
    g, A = dense_problem(n)
    if negate is not None:
        g["Sx"] = g["Sx"].copy()
        g["Sx"][negate * n + negate] = -1.0
    g["maxcsize"] = 1
    st, minor, Lx, D = _with_keys(emu, {28: rows}, lambda: run_emu(emu, g))
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert (st, minor) == (so, mo)
    low = np.tril(np.ones((n, n), dtype=bool)).T.reshape(-1)
    assert relerr(Lx[low], Lo[low]) < 1e-10
    if st == 1:
        assert np.array_equal(Lx[low] == 0, Lo[low] == 0)


@pytest.mark.parametrize("n,rows,negate", [(150, 64, None), (200, 32, 131), (330, 64, None), (330, 16, 40), (300, 32, None), (330, 128, None), (460, 128, 300),
                                           (600, 64, None)])
def test_dense_supernode_through_the_panel_group_kernels(emu, n, rows, negate):  # noqa: F811
    """the same through k_panel_diag and k_panel_trsm<rows> from their own source (and k_potrf64 / k_trsm64 / k_gemm64 /
    k_gemm128 for the rest): ragged windows (150 = 2 blocks + 22 columns, 330 = a full window + a second one of 74),
    LAPACK's info from inside a window"""
    from oracle import port
    g, A = dense_problem(n)
    if negate is not None:
        g["Sx"] = g["Sx"].copy()
        g["Sx"][negate * n + negate] = -1.0
    g["maxcsize"] = 1
    st, minor, Lx, D = _with_keys(emu, {100: 3, 28: rows}, lambda: run_emu(emu, g))
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert (st, minor) == (so, mo)
    low = np.tril(np.ones((n, n), dtype=bool)).T.reshape(-1)
    assert relerr(Lx[low], Lo[low]) < 1e-10
    if st == 1:
        assert np.array_equal(Lx[low] == 0, Lo[low] == 0)


@pytest.mark.parametrize("kind,n,seed,dev", [("arrow", 130, 12, 3), ("banded", 400, 3, 3), ("arrow", 800, 6, 0), ("blocks", 1100, 8, 0),
                                             ("blocks", 500, 7, 0), ("arrow", 350, 5, 3)])
def test_random_structures_with_panel_groups(emu, kind, n, seed, dev):  # noqa: F811
    """ragged supernodes, scattered relative maps, wide fronts (tests/test_fuzz_schedule.py's structures) with the key
    on: host loops, and for two of them the kernels from their own source"""
    from oracle import port
    import test_fuzz_schedule as tf
    g, rng = tf.build(kind, n, seed)
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    st, minor, Lx, D = _with_keys(emu, {100: dev, 28: 128 if seed % 2 else 64}, lambda: run_emu(emu, g))
    assert (st, minor) == (so, mo) == (0, n)
    _lower_panels_close(g, Lx, Lo)


def test_the_schedule_has_three_launches_per_window(emu):  # noqa: F811
    """what the key is for: a dense 1024-column supernode takes 16 POTRF + 15 TRSM + 12 K = 64 updates + 3 K = 256 updates
    without it, and 4 PDIAG + 3 PTRSM + 3 K = 256 updates with it (emu_schedule_counts: steps by kind)"""
    g, A = dense_problem(1024)
    a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
    import ctypes as C
    i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
    emu.emu_schedule_counts.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i64p]
    out = np.zeros(8, dtype=np.int64)
    assert _with_keys(emu, {28: 0}, lambda: emu.emu_schedule_counts(*a, out)) == 0
    off = out.copy()
    assert _with_keys(emu, {28: 64}, lambda: emu.emu_schedule_counts(*a, out)) == 0
    # [launched POTRF steps, GEMM steps, PDIAG steps, PTRSM steps, bounds check ok]
    assert list(off[:5]) == [16, 15 + 12 + 3, 0, 0, 1]
    assert list(out[:5]) == [0, 3, 4, 3, 1]
