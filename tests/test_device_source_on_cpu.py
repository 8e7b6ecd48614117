"""The product's own kernel source, executed on the CPU.  build() cuts the marked host-runnable region out of
suitesparse_amd/csrc/kernels.hip.hpp (assembly kernels, k_relmap, k_gemm_tiny: no barrier, no LDS, no wave intrinsic)
and the emulator runs those kernels thread by thread with the engine's launch geometry and arguments (emulator key 100)
in place of its own plain loops.  Checked against the reference's golden outputs / the oracle -- a GPU-less check of the
device code itself, not only of the host logic around it."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, golden_names, load_golden, relerr, sym_of
from test_schedule import emu, run_emu  # noqa: F401
import test_unsym as tu

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
REAL = [n for n in golden_names() if load_golden(n)["xtype"] == 1]


@pytest.mark.parametrize("tiny_max", [0, 4096, 1 << 20])
@pytest.mark.parametrize("name", REAL)
def test_factorization_through_the_device_source(emu, name, tiny_max):  # noqa: F811
    """k_assemble (+ beta), k_relmap and -- with the tiny launch class on -- k_gemm_tiny from their own source; the
    other steps by the emulator's loops: status, minor, factor and zero pattern of the reference"""
    g = load_golden(name)
    emu.emu_set_tuning(100, 1); emu.emu_set_tuning(14, tiny_max)
    try:
        st, minor, Lx, _ = run_emu(emu, g)
    finally:
        emu.emu_set_tuning(100, 0); emu.emu_set_tuning(14, 2048)
    assert (st, minor) == (g["status"], g["minor"])
    assert relerr(Lx, g["Lx"]) < 1e-12
    if st == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)


@pytest.mark.parametrize("name", ["demo_bcsstk01", "tcov_r5lo", "elast27_4_npd100"])
def test_wave_potrf_source_one_host_thread_per_lane(emu, name):  # noqa: F811
    """k_potrf16 (one wave per diagonal block of <= 16 columns, registers + v_readlane only) from its own source with
    one host thread per lane: factor, block inverse, the fused TRSM of the rows below a supernode of <= 16 columns
    (PotrfTask::fuse_trsm) and the first bad pivot of the not-positive-definite fixture.  Slow (a readlane is two
    64-thread barriers), hence few fixtures."""
    g = load_golden(name)
    emu.emu_set_tuning(100, 2); emu.emu_set_tuning(14, 4096)
    try:
        st, minor, Lx, D = run_emu(emu, g)
        if st == 0:
            a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
            n, nrhs = g["B"].shape
            X = np.ascontiguousarray(g["B"].T).reshape(-1).copy()
            assert emu.emu_super_solve(1, *a, Lx, D, nrhs, X, n) == 0      # the solves run on the kernel's inverses
            assert emu.emu_super_solve(0, *a, Lx, D, nrhs, X, n) == 0
            assert relerr(X.reshape(nrhs, n).T, g["Z"]) < 1e-8
    finally:
        emu.emu_set_tuning(100, 0); emu.emu_set_tuning(14, 2048)
    assert (st, minor) == (g["status"], g["minor"])
    assert relerr(Lx, g["Lx"]) < 1e-12
    if st == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)


@pytest.mark.parametrize("name", [n for n in tu.unsym_names() if "c10" not in n])
def test_unsym_factorization_through_the_device_source(emu, name):  # noqa: F811
    """k_assemble_aat<1> + k_add_beta<1> (stype == 0) from their own source"""
    g = tu.load_unsym(name)
    emu.emu_set_tuning(100, 1); emu.emu_set_tuning(14, 4096)
    try:
        st, minor, Lx = tu.emu_numeric_f(g)
    finally:
        emu.emu_set_tuning(100, 0); emu.emu_set_tuning(14, 2048)
    assert (st, minor) == (g["status"], g["minor"])
    assert relerr(Lx, g["Lx"]) < 1e-11


def device_assemble(kind, g, Ap, Ai, Ax, F=None, iperm=None, beta=0.0, Anz=None, Fnz=None):
    E = 2 if np.iscomplexobj(Ax) else 1
    lib = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    lib.emu_device_assemble.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, i64p,
                                        i64p, C.c_void_p, f64p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_double, f64p]
    flat = lambda x: np.ascontiguousarray(x, dtype=np.complex128).view(np.float64) if E == 2 else \
        np.ascontiguousarray(x, dtype=np.float64)      # noqa: E731
    keep = []

    def ptr(a, dt=np.int64):
        if a is None:
            return None
        keep.append(np.ascontiguousarray(a, dtype=dt))
        return keep[-1].ctypes.data_as(C.c_void_p)
    Lx = np.zeros(g["xsize"] * E)
    fx = None
    if F is not None:
        keep.append(flat(F[2]))
        fx = keep[-1].ctypes.data_as(C.c_void_p)
    rc = lib.emu_device_assemble(kind, E, g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["xsize"],
                                 np.ascontiguousarray(Ap, dtype=np.int64), np.ascontiguousarray(Ai, dtype=np.int64),
                                 ptr(Anz), flat(Ax), ptr(F[0]) if F else None, ptr(F[1]) if F else None, ptr(Fnz), fx,
                                 ptr(iperm), beta, Lx)
    assert rc == 0
    return Lx.view(np.complex128) if E == 2 else Lx


def expected_panels(g, M):
    """the supernodal layout filled from the dense matrix M (entries on or below the diagonal, pattern of L)"""
    out = np.zeros(g["xsize"], dtype=M.dtype)
    for s in range(g["nsuper"]):
        k1, k2 = int(g["Super"][s]), int(g["Super"][s + 1])
        rows = g["Ls"][int(g["Lpi"][s]):int(g["Lpi"][s + 1])]
        P = M[np.ix_(rows, np.arange(k1, k2))].copy()
        P[rows[:, None] < np.arange(k1, k2)[None, :]] = 0
        out[int(g["Lpx"][s]):int(g["Lpx"][s + 1])] = P.T.reshape(-1)
    return out


@pytest.mark.parametrize("name", tu.unsym_names())
@pytest.mark.parametrize("beta", [0.0, 0.75])
def test_assemble_aat_source_real_and_complex(name, beta):
    """k_assemble_aat<1> / <2> + k_add_beta: tril(A*F) + beta*I in the supernodal layout, against numpy; packed and
    unpacked (Anz / Fnz with slack entries that must not be read)"""
    import scipy.sparse as sp
    g = tu.load_unsym(name)
    S = sp.csc_matrix((g["Sx"], g["Si"], g["Sp"]), shape=(g["n"], g["ncol"]))
    F = sp.csc_matrix((g["Fx"], g["Fi"], g["Fp"]), shape=(g["ncol"], g["n"]))
    want = expected_panels(g, (S @ F).toarray() + beta * np.eye(g["n"]))
    got = device_assemble(2, g, g["Sp"], g["Si"], g["Sx"], F=(g["Fp"], g["Fi"], g["Fx"]), beta=beta)
    assert relerr(got, want) < 1e-14

    def unpack(p, i, x):
        cnt = np.diff(p)
        p2 = np.concatenate([[0], np.cumsum(cnt + 2)]).astype(np.int64)
        i2 = np.zeros(p2[-1], dtype=np.int64)
        x2 = np.full(p2[-1], 1e30, dtype=x.dtype)
        for k in range(len(cnt)):
            i2[p2[k]:p2[k] + cnt[k]] = i[p[k]:p[k + 1]]
            x2[p2[k]:p2[k] + cnt[k]] = x[p[k]:p[k + 1]]
        return p2, i2, x2, cnt.astype(np.int64)
    Sp, Si, Sx, Snz = unpack(g["Sp"], g["Si"], g["Sx"])
    Fp, Fi, Fx, Fnz = unpack(g["Fp"], g["Fi"], g["Fx"])
    got = device_assemble(2, g, Sp, Si, Sx, F=(Fp, Fi, Fx), beta=beta, Anz=Snz, Fnz=Fnz)
    assert relerr(got, want) < 1e-14


@pytest.mark.parametrize("name", ["demo_c", "tcov_c30up", "tcov_z30lo", "helm3d_6_metis", "lap3d_8_metis", "tcov_ex5lo"])
def test_assemble_sources_symmetric_real_and_complex(name):
    """k_assemble / k_assemble_c on S = tril(PAP'), and k_assemble_perm<E> on A + the inverse permutation (entries that
    cross the diagonal conjugated): the same panels"""
    import scipy.sparse as sp
    g = load_golden(name)
    n = g["n"]
    S = sp.csc_matrix((g["Sx"], g["Si"], g["Sp"]), shape=(n, n)).toarray()
    beta = 0.5
    want = expected_panels(g, np.tril(S) + beta * np.eye(n))
    got = device_assemble(0, g, g["Sp"], g["Si"], g["Sx"], beta=beta)
    assert np.array_equal(got, want)
    # A itself (lower or upper stored) through the inverse permutation
    A = sp.csc_matrix((g["Ax"], g["Ai"], g["Ap"]), shape=(n, n))
    if g["stype"] > 0:
        A = A.conj().T.tocsc()          # mi355_factor_set_matrix_perm takes the lower part
    A = sp.tril(A, format="csc")
    A.sort_indices()
    iperm = np.empty(n, dtype=np.int64)
    iperm[g["Perm"]] = np.arange(n)
    got = device_assemble(1, g, A.indptr.astype(np.int64), A.indices.astype(np.int64),
                          np.asarray(A.data).astype(g["Sx"].dtype), iperm=iperm, beta=beta)
    assert relerr(got, want) < 1e-15


# ---------------------------------------------------------------------------------------------------------------
# Every kernel of the real factorization from its own source (emulator key 100 = 3): k_potrf64, k_trsm64, k_gemm64,
# k_gemm128 compiled by g++ against tests/helpers/hipcpu/hip/hip_runtime.h (one fiber per lane; __syncthreads,
# v_readlane and v_mfma_f64_16x16x4_f64 as barriers around exchange arrays) and launched with the engine's geometry.
# Slow (a 256-thread barrier per __syncthreads), so the cases are small -- but they are the MFMA / LDS kernels.

def dense_problem(n, seed=None):
    rng = np.random.default_rng(n if seed is None else seed)
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    g = dict(n=n, nsuper=1, ssize=n, xsize=n * n, Super=np.array([0, n], dtype=np.int64),
             Lpi=np.array([0, n], dtype=np.int64), Lpx=np.array([0, n * n], dtype=np.int64),
             Ls=np.arange(n, dtype=np.int64), Sp=np.arange(n + 1, dtype=np.int64) * n,
             Si=np.tile(np.arange(n, dtype=np.int64), n), Sx=np.ascontiguousarray(A.T).reshape(-1), beta=0.0)
    return g, A


@pytest.mark.parametrize("name", ["demo_bcsstk01", "tcov_r5lo", "tcov_ex5lo", "tcov_20lo", "demo_bcsstk02"])
def test_all_factor_kernels_from_source_golden(emu, name):  # noqa: F811
    g = load_golden(name)
    emu.emu_set_tuning(100, 3)
    try:
        st, minor, Lx, D = run_emu(emu, g)
    finally:
        emu.emu_set_tuning(100, 0)
    assert (st, minor) == (g["status"], g["minor"])
    assert relerr(Lx, g["Lx"]) < 1e-10                 # the GPU parity bar (TRSM by explicit block inverses)
    if st == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)


@pytest.mark.parametrize("n,ob,ob2,negate", [(330, 128, 0, None), (200, 64, 128, None), (150, 64, 0, 97)])
def test_dense_supernode_through_the_mfma_kernels(emu, n, ob, ob2, negate):  # noqa: F811
    """one dense supernode: k_potrf64 (all four 16-column blocks, the MFMA assembly of the inverse), k_trsm64, the
    K = 64 updates in k_gemm64 and -- n = 330: M = 202 >= 192, N >= 96 -- the outer update in k_gemm128 (DIRECT_SUB
    epilogue, ragged edges); negate: a bad pivot in the middle, LAPACK's info from k_potrf64"""
    from oracle import port
    g, A = dense_problem(n)
    if negate is not None:
        g["Sx"] = g["Sx"].copy()
        g["Sx"][negate * n + negate] = -1.0
    g["maxcsize"] = 1
    emu.emu_set_tuning(100, 3); emu.emu_set_tuning(2, ob); emu.emu_set_tuning(7, ob2)
    try:
        st, minor, Lx, D = run_emu(emu, g)
    finally:
        emu.emu_set_tuning(100, 0); emu.emu_set_tuning(2, 256); emu.emu_set_tuning(7, 1024)
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert (st, minor) == (so, mo)
    low = np.tril(np.ones((n, n), dtype=bool)).T.reshape(-1)       # column-major: entries on or below the diagonal
    assert relerr(Lx[low], Lo[low]) < 1e-10
    if st == 1:
        assert np.array_equal(Lx[low] == 0, Lo[low] == 0)


def test_scatter_update_through_k_gemm128(emu):  # noqa: F811
    """two supernodes, the first one's rows scattered with gaps over the second: the update task is big enough for
    k_gemm128 (M = 230, N = 230, K = 96), i.e. its SCATTER_SUB epilogue -- the relative map staged in LDS, FP64
    atomics -- plus k_relmap, from their own source"""
    from oracle import port
    n0, n1 = 96, 340
    n = n0 + n1
    extra = np.array(sorted(set(range(n0, n, 3)) | set(range(n0 + 1, n, 3)) | {n - 1}), dtype=np.int64)
    rows0 = np.concatenate([np.arange(n0), extra])
    g = dict(n=n, nsuper=2, Super=np.array([0, n0, n], dtype=np.int64),
             Lpi=np.array([0, len(rows0), len(rows0) + n1], dtype=np.int64),
             Lpx=np.array([0, len(rows0) * n0, len(rows0) * n0 + n1 * n1], dtype=np.int64),
             Ls=np.concatenate([rows0, np.arange(n0, n)]).astype(np.int64), beta=0.0)
    g["ssize"], g["xsize"], g["maxcsize"] = len(g["Ls"]), int(g["Lpx"][-1]), len(extra) ** 2
    assert len(extra) >= 192
    rng = np.random.default_rng(7)
    A = np.zeros((n, n))
    A[np.ix_(rows0, np.arange(n0))] = rng.standard_normal((len(rows0), n0))
    A[n0:, n0:] = rng.standard_normal((n1, n1))
    A = np.tril(A)
    A = A + A.T + np.eye(n) * (np.abs(A + A.T).sum(1).max() + 1)
    import scipy.sparse as sp
    S = sp.csc_matrix(np.tril(A))
    S.sort_indices()
    g.update(Sp=S.indptr.astype(np.int64), Si=S.indices.astype(np.int64), Sx=S.data)
    cls = np.zeros(8, dtype=np.int64)
    emu.emu_set_tuning(100, 3)
    try:
        st, minor, Lx, D = run_emu(emu, g)
    finally:
        emu.emu_set_tuning(100, 0)
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert (st, minor) == (so, mo) == (0, n)
    # compare on the lower trapezoid of each panel (the upper corner of a panel's diagonal block is not defined)
    for s in range(2):
        a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
        nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
        got, want = Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
        low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
        assert relerr(got[low], want[low]) < 1e-10


# ---------------------------------------------------------------------------------------------------------------
# The solve kernels from their own source: emu_device_solve issues the launch sequence of the engine's solve_dev
# (grouped scalar sweeps, grouped MFMA sweeps over the row-major image, one launch pair per block) on the CPU.

def device_solve(emu, g, Lx, D, B, mma, group):  # noqa: F811
    emu.emu_device_solve.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64,
                                     C.c_int64, f64p, f64p, C.c_int64, f64p, C.c_int64]
    a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
    n, nrhs = B.shape
    X = np.ascontiguousarray(B.T).reshape(-1).copy()
    assert emu.emu_device_solve(1, mma, group, *a, Lx, D, nrhs, X, n) == 0
    Y = X.reshape(nrhs, n).T.copy()
    assert emu.emu_device_solve(0, mma, group, *a, Lx, D, nrhs, X, n) == 0
    return Y, X.reshape(nrhs, n).T.copy()


PATHS = [(0, 1, 1), (0, 1, 4), (0, 0, 1), (0, 0, 6), (1, 1, 5), (1, 0, 9), (1, 1, 70)]      # (mma, grouped, nrhs)


@pytest.mark.parametrize("name,mma,group,nrhs",
                         [(nm, *pth) for nm in ("demo_bcsstk01", "demo_bcsstk02", "tcov_r5lo") for pth in PATHS]
                         + [("lap2d_16_metis", 0, 1, 1), ("lap2d_16_metis", 1, 1, 5)])
def test_solve_kernels_from_source_golden(emu, name, mma, group, nrhs):  # noqa: F811
    from oracle import port
    g = load_golden(name)
    st, minor, Lx, D = run_emu(emu, g)
    B = np.random.default_rng(nrhs).standard_normal((g["n"], nrhs))
    Yo = port.super_solve(g, g["Lx"], B, lower=True)
    Xo = port.super_solve(g, g["Lx"], Yo, lower=False)
    Y, X = device_solve(emu, g, Lx, D, B, mma, group)
    assert relerr(Y, Yo) < 1e-9 and relerr(X, Xo) < 1e-9


@pytest.mark.parametrize("mma,group,nrhs", [(0, 1, 1), (0, 1, 3), (1, 1, 8), (0, 0, 2), (1, 0, 5)])
def test_solve_kernels_from_source_wide_supernode(emu, mma, group, nrhs):  # noqa: F811
    """one dense supernode of 330 columns: a full solve group of four 64-column blocks plus a second group, row tiles
    below the groups, the backward runs"""
    g, A = dense_problem(330)
    st, minor, Lx, D = run_emu(emu, g)
    assert st == 0
    B = np.random.default_rng(nrhs).standard_normal((330, nrhs))
    Y, X = device_solve(emu, g, Lx, D, B, mma, group)
    Lc = np.linalg.cholesky(A)
    assert relerr(Y, np.linalg.solve(Lc, B)) < 1e-9
    assert relerr(X, np.linalg.solve(A, B)) < 1e-9


# ---------------------------------------------------------------------------------------------------------------
# The complex kernels (cd_ / zd_ instantiations of the reference) from their own source: k_assemble_c, k_potrf64c,
# k_gemm64c and k_gemm128<true> (big tasks through the real view).  Needs the emulator built with clang++ (the
# complex kernels use its vector swizzles); build() does that where /opt/rocm's clang++ exists.

def device_numeric_c(g, quick=0, want_dinv=False):
    lib = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    if not hasattr(lib, "emu_device_numeric_c"):
        pytest.skip("emulator built without clang++")
    lib.emu_device_numeric_c.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, i64p, i64p,
                                         f64p, C.c_double, C.c_int, f64p, i64p, C.POINTER(C.c_int), i64p, C.c_void_p]
    lib.emu_dinv_size.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64]
    lib.emu_dinv_size.restype = C.c_int64
    Lx = np.zeros(2 * g["xsize"])
    minor, st, nbig = np.zeros(1, dtype=np.int64), C.c_int(0), np.zeros(1, dtype=np.int64)
    D = np.zeros(2 * lib.emu_dinv_size(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"],
                                       g["xsize"]) + 2) if want_dinv else None
    rc = lib.emu_device_numeric_c(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"],
                                  g["Sp"], g["Si"], np.ascontiguousarray(g["Sx"], dtype=np.complex128).view(np.float64),
                                  g["beta"], quick, Lx, minor, C.byref(st), nbig,
                                  D.ctypes.data_as(C.c_void_p) if want_dinv else None)
    assert rc == 0
    if want_dinv:
        return st.value, int(minor[0]), Lx.view(np.complex128), D
    return st.value, int(minor[0]), Lx.view(np.complex128), int(nbig[0])


@pytest.mark.parametrize("name", [n for n in golden_names() if load_golden(n)["xtype"] == 2 and n != "helm3d_6_metis"])
def test_complex_kernels_from_source_golden(name):
    g = load_golden(name)
    st, minor, Lx, _ = device_numeric_c(g)
    assert (st, minor) == (g["status"], g["minor"])
    assert relerr(Lx, g["Lx"]) < 1e-10
    if st == 1:
        assert np.array_equal(Lx == 0, g["Lx"] == 0)


def test_complex_big_tile_kernel_from_source():
    """a dense Hermitian supernode of 250 columns with 128-wide outer blocks: the outer update (M = N = 122 complex
    rows = 244 real ones, K = 128) runs in k_gemm128<true> -- the interleaved real view, the pair images of B built at
    the LDS write, the rectangular 64 x 128 tile table"""
    from oracle import port
    n = 250
    rng = np.random.default_rng(3)
    M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = M @ M.conj().T + n * np.eye(n)
    g = dict(n=n, nsuper=1, ssize=n, xsize=n * n, Super=np.array([0, n], dtype=np.int64),
             Lpi=np.array([0, n], dtype=np.int64), Lpx=np.array([0, n * n], dtype=np.int64),
             Ls=np.arange(n, dtype=np.int64), Sp=np.arange(n + 1, dtype=np.int64) * n,
             Si=np.tile(np.arange(n, dtype=np.int64), n), Sx=np.ascontiguousarray(A.T).reshape(-1), beta=0.0, maxcsize=1)
    lib = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    lib.emu_set_tuning(2, 128); lib.emu_set_tuning(7, 0)
    try:
        st, minor, Lx, nbig = device_numeric_c(g)
    finally:
        lib.emu_set_tuning(2, 256); lib.emu_set_tuning(7, 1024)
    assert (st, minor) == (0, n) and nbig >= 1
    Lg = Lx.reshape(n, n).T
    assert relerr(np.tril(Lg), np.linalg.cholesky(A)) < 1e-11


@pytest.mark.parametrize("kind,n,seed,tiny_max", [("arrow", 130, 12, 4096), ("banded", 400, 3, 0)])
def test_random_structures_through_the_device_kernels(emu, kind, n, seed, tiny_max):  # noqa: F811
    """the structures of tests/test_fuzz_schedule.py (random graphs, chains, arrowheads: ragged supernodes, scattered
    relative maps, a 350-column front through k_gemm128) with every factor kernel from its own source"""
    from oracle import port
    import test_fuzz_schedule as tf
    g, rng = tf.build(kind, n, seed)
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    emu.emu_set_tuning(100, 3); emu.emu_set_tuning(14, tiny_max)
    try:
        st, minor, Lx, D = run_emu(emu, g)
    finally:
        emu.emu_set_tuning(100, 0); emu.emu_set_tuning(14, 2048)
    assert (st, minor) == (so, mo) == (0, n)
    for s in range(g["nsuper"]):
        a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
        nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
        got, want = Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
        low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
        assert np.abs(got[low] - want[low]).max() <= 1e-10 * np.abs(Lo).max(), s


_CFACTOR = {}


@pytest.mark.parametrize("mma,group,nrhs", [(0, 0, 1), (0, 0, 3), (1, 1, 4), (1, 0, 3), (1, 1, 40)])
@pytest.mark.parametrize("name", ["demo_c", "tcov_c30up", "tcov_z30lo"])
def test_complex_solve_kernels_from_source(name, mma, group, nrhs):
    """the complex sweeps: k_solve_diag_c / k_solve_fwd_update_c / k_solve_bwd_update_c per block, and the MFMA sweeps of
    a complex factor on the real / imaginary row-major image (E = 2 instantiations, k_solve_transpose<double2>), on the
    factor and the block inverses the complex factor kernels produced -- everything from the product's own source"""
    from oracle import port
    g = load_golden(name)
    if name not in _CFACTOR:
        _CFACTOR[name] = device_numeric_c(g, want_dinv=True)
    st, minor, Lx, D = _CFACTOR[name]
    assert st == 0
    lib = C.CDLL(os.path.join(ROOT, "tests", "helpers", "libsched_emu.so"))
    lib.emu_device_solve_c.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64,
                                       C.c_int64, f64p, f64p, C.c_int64, f64p, C.c_int64]
    a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
    n = g["n"]
    rng = np.random.default_rng(nrhs)
    B = rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs))
    Yo = port.super_solve(g, g["Lx"], B, lower=True)
    Xo = port.super_solve(g, g["Lx"], Yo, lower=False)
    X = np.ascontiguousarray(B.T).reshape(-1).copy().view(np.float64)
    Lxf = np.ascontiguousarray(Lx).view(np.float64)
    assert lib.emu_device_solve_c(1, mma, group, *a, Lxf, D, nrhs, X, n) == 0
    assert relerr(X.view(np.complex128).reshape(nrhs, n).T, Yo) < 1e-9
    assert lib.emu_device_solve_c(0, mma, group, *a, Lxf, D, nrhs, X, n) == 0
    assert relerr(X.view(np.complex128).reshape(nrhs, n).T, Xo) < 1e-9
