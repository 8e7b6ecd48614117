"""Randomised structures through the host logic of the product (CPU, seeded): random sparse symmetric positive
definite matrices under random permutations and relaxed-amalgamation settings -> our symbolic analysis -> the level
schedule with randomly narrowed block widths -> the CPU emulator of the schedule (factor, not-posdef rerun, plain /
grouped / MFMA-shaped solve sweeps) against the oracle restatement of the reference's worker, plus
check_schedule_bounds on every schedule of random partitions.  What this guards: a task descriptor outside its
arrays is a GPU memory fault on the device, so every shape of elimination tree the builder can meet is worth a run."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, relerr
from test_schedule import emu, run_emu  # noqa: F401  (fixture)

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")


def random_spd(kind, n, seed):
    """lower-stored CSC of a random SPD matrix with a structure of the given kind"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    if kind == "sparse":                        # random graph, a few entries per row
        M = sp.random(n, n, density=min(1.0, 3.0 / n), random_state=rng, format="csr")
    elif kind == "banded":                      # a chain: deep etree, one child per node
        w = int(rng.integers(1, 9))
        M = sp.diags([rng.standard_normal(n - k) for k in range(1, w + 1)], list(range(1, w + 1)), format="csr")
    elif kind == "arrow":                       # dense last rows: one wide supernode on top of many leaves
        k = int(rng.integers(5, max(6, n // 3)))
        M = sp.lil_matrix((n, n))
        M[n - k:, :] = rng.standard_normal((k, n))
        M = M.tocsr() + sp.random(n, n, density=1.0 / n, random_state=rng, format="csr")
    elif kind == "blocks":                      # dense diagonal blocks of random width joined by a sparse graph
        M = sp.lil_matrix((n, n))
        at = 0
        while at < n:
            w = int(min(n - at, rng.integers(1, 200)))
            M[at:at + w, at:at + w] = rng.standard_normal((w, w))
            at += w
        M = M.tocsr() + sp.random(n, n, density=0.5 / n, random_state=rng, format="csr")
    else:
        raise ValueError(kind)
    A = M + M.T
    d = np.asarray(abs(A).sum(axis=1)).ravel() + 1.0
    A = (A + sp.diags(d)).tocsc()
    L = sp.tril(A, format="csc")
    L.sort_indices()
    return L.indptr.astype(np.int64), L.indices.astype(np.int64), np.ascontiguousarray(L.data)


CASES = [("sparse", 300, 1), ("sparse", 900, 2), ("banded", 400, 3), ("banded", 70, 4), ("arrow", 350, 5),
         ("arrow", 800, 6), ("blocks", 500, 7), ("blocks", 1100, 8), ("sparse", 1, 9), ("sparse", 2, 10),
         ("blocks", 65, 11), ("arrow", 130, 12)]


def build(kind, n, seed):
    from suitesparse_amd import symbolic
    rng = np.random.default_rng(1000 + seed)
    Ap, Ai, Ax = random_spd(kind, n, seed)
    perm = rng.permutation(n) if seed % 3 else None
    # relaxed amalgamation: the reference's defaults, none at all, or aggressive (cholmod.h nrelax / zrelax)
    relax = [(None, None), ([0, 0, 0], [0.0, 0.0, 0.0]), ([8, 32, 96], [0.9, 0.5, 0.2])][seed % 3]
    sym = symbolic.analyze(n, Ap, Ai, perm, nrelax=relax[0], zrelax=relax[1])
    Sp, Si, Sx = symbolic.permute(sym, Ap, Ai, Ax)
    g = dict(sym, Sp=Sp, Si=Si, Sx=Sx, beta=0.0)
    g["ssize"] = len(sym["Ls"])
    return g, rng


@pytest.mark.parametrize("kind,n,seed", CASES)
def test_random_structure_factor_and_solves(emu, kind, n, seed):  # noqa: F811
    from oracle import port
    g, rng = build(kind, n, seed)
    ob = int(rng.choice([64, 128, 256]))
    ob2 = int(rng.choice([0, 2 * ob, 4 * ob]))
    emu.emu_set_tuning(2, ob); emu.emu_set_tuning(7, ob2)
    emu.emu_set_tuning(0, int(rng.choice([64, 192]))); emu.emu_set_tuning(1, int(rng.choice([64, 96])))
    div = int(rng.choice([1, 4, 256]))
    emu.emu_set_tuning(11, div); emu.emu_set_tuning(13, div)
    emu.emu_set_tuning(14, int(rng.choice([0, 256, 1 << 16])))          # tiny launch class off / narrow / wide
    try:
        so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
        st, minor, Lx, D = run_emu(emu, g)
        assert (st, minor) == (so, mo) == (0, n)
        assert relerr(Lx, Lo) < 1e-11
        a = [g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"], g["ssize"], g["xsize"]]
        nrhs = int(rng.integers(1, 5))
        B = rng.standard_normal((n, nrhs))
        Yo = port.super_solve(g, Lo, B, lower=True)
        Xo = port.super_solve(g, Lo, Yo, lower=False)
        for path in ("plain", 0, 1):
            X = np.ascontiguousarray(B.T).reshape(-1).copy()
            for lower, want in ((1, Yo), (0, Xo)):
                rc = emu.emu_super_solve(lower, *a, Lx, D, nrhs, X, n) if path == "plain" else \
                    emu.emu_super_solve_grouped(lower, path, *a, Lx, D, nrhs, X, n)
                assert rc == 0
                assert relerr(X.reshape(nrhs, n).T, want) < 1e-8, (path, lower)
        # a bad pivot at a random column: status, minor, zero pattern and the refactorized leading columns
        g2 = dict(g)
        Sx = g["Sx"].copy()
        col = int(rng.integers(0, n))
        Sx[g["Sp"][col]] = -abs(Sx[g["Sp"][col]])
        g2["Sx"] = Sx
        for quick in (0, 1):
            so2, mo2, Lo2 = port.super_numeric(g2, g2["Sp"], g2["Si"], g2["Sx"], quick_return=bool(quick))
            st2, minor2, Lx2, _ = run_emu(emu, g2, quick=quick)
            assert (st2, minor2) == (so2, mo2) and so2 == 1
            assert relerr(Lx2, Lo2) < 1e-11
            assert np.array_equal(Lx2 == 0, Lo2 == 0)
    finally:
        for k, v in ((2, 256), (7, 1024), (0, 192), (1, 96), (11, 256), (13, 4096), (14, 0)):
            emu.emu_set_tuning(k, v)


@pytest.mark.parametrize("kind,n,seed,world", [("sparse", 900, 2, 3), ("arrow", 800, 6, 2), ("blocks", 1100, 8, 4),
                                                ("banded", 400, 3, 8), ("sparse", 300, 1, 5)])
def test_random_structure_partition_bounds(emu, kind, n, seed, world):  # noqa: F811
    """every schedule and solve work list of every rank of a partition of a random tree stays inside its arrays"""
    from suitesparse_amd import multigpu
    g, rng = build(kind, n, seed)
    emu.emu_validate_partition.argtypes = [C.c_int64, C.c_int64, i64p, i64p, i64p, i64p, C.c_int64, C.c_int64, C.c_int,
                                           i8p, i8p, C.c_int, C.c_int, C.c_char_p, C.c_int]
    ob = int(rng.choice([64, 128]))
    emu.emu_set_tuning(2, ob); emu.emu_set_tuning(7, 2 * ob); emu.emu_set_tuning(14, int(rng.choice([0, 1 << 12])))
    try:
        owner = multigpu.partition(g, world, min_subtrees_per_rank=1)["owner"]
        for cplx in (0, 1):
            for r in range(world):
                msg = C.create_string_buffer(300)
                rc = emu.emu_validate_partition(g["n"], g["nsuper"], g["Super"], g["Lpi"], g["Lpx"], g["Ls"],
                                                g["ssize"], g["xsize"], cplx, (owner == r).astype(np.int8),
                                                (owner < 0).astype(np.int8), r, world, msg, 300)
                assert rc == 0, (r, rc, msg.value.decode())
    finally:
        emu.emu_set_tuning(2, 256); emu.emu_set_tuning(7, 1024); emu.emu_set_tuning(14, 2048)
