"""CPU tests of the host symbolic analysis (suitesparse_amd/csrc/symbolic.cpp): for a given
permutation it must reproduce the reference's cholmod_l_analyze (CHOLMOD_GIVEN) bit for bit --
Perm after the weighted postorder, ColCount, Super/Lpi/Lpx/Ls, maxcsize/maxesize, fl/lnz and
S = tril(P A P')."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from suitesparse_amd import problems, symbolic

KEYS = ["Perm", "ColCount", "Super", "Lpi", "Lpx", "Ls"]
SIZES = ["nsuper", "ssize", "xsize", "maxcsize", "maxesize"]


def lower_of(g):
    """lower-triangular CSC of the golden matrix whatever its stored stype"""
    import scipy.sparse as sp
    n = g["n"]
    A = sp.csc_matrix((g["Ax"], g["Ai"], g["Ap"]), shape=(n, n))
    if g["stype"] > 0:
        A = sp.csc_matrix(A.T)
    A = sp.tril(A, format="csc")
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("nd_")])
def test_symbolic_matches_reference_given_perm_golden(name):
    g = load_golden(name)
    Ap, Ai, Ax = lower_of(g)
    sym = symbolic.analyze(g["n"], Ap, Ai, g["UserPerm"])
    for k in KEYS:
        assert np.array_equal(sym[k], g[k]), k
    for k in SIZES:
        assert sym[k] == g[k], k
    assert sym["fl"] == float(g["fl"]) and sym["lnz"] == float(g["lnz"])
    Sp, Si, Sx = symbolic.permute(sym, Ap, Ai, Ax)
    assert np.array_equal(Sp, g["Sp"]) and np.array_equal(Si, g["Si"]) and np.array_equal(Sx, g["Sx"])


@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith("nd_")])
def test_symbolic_with_reference_perm_is_consistent(name):
    """feeding the reference's final (already postordered) Perm back in: same fl/lnz and a valid
    factor structure of the same size class (the postorder of a postordered tree may relabel)."""
    g = load_golden(name)
    if g["xtype"] != 1 and g["stype"] > 0:
        pytest.skip("complex upper-stored fixture: pattern-only check not needed")
    Ap, Ai, _ = lower_of(g)
    sym = symbolic.analyze(g["n"], Ap, Ai, g["Perm"])
    assert sym["fl"] == float(g["fl"]) and sym["lnz"] == float(g["lnz"])
    assert sorted(sym["Perm"]) == list(range(g["n"]))


@pytest.mark.parametrize("shape,dof", [((7, 5, 3), 1), ((16, 16), 1), ((5, 5, 5), 3), ((33, 2, 1), 2), ((1, 1, 1), 1)])
def test_nd_order_is_a_permutation(shape, dof):
    perm = symbolic.nd_grid_order(shape, dof)
    n = int(np.prod(shape)) * dof
    assert perm.shape == (n,) and np.array_equal(np.sort(perm), np.arange(n))


@pytest.mark.parametrize("kind,size", [("lap3d", 17), ("lap2d", 90), ("elast", 6), ("helm", 9)])
def test_symbolic_vs_live_reference(kind, size, have_ref):
    """structure bit-exact; S = tril(P A P') equal to the reference's, complex values included (an entry that
    moves across the diagonal is conjugated)"""
    if not have_ref:
        pytest.skip("oracle/_ref not available")
    from oracle.refcholmod import RefCholmod
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d, "elast": problems.elasticity27,
         "helm": problems.helmholtz_3d}[kind](size)
    perm = symbolic.nd_grid_order(p["shape"], p["dof"])
    sym = symbolic.analyze(p["n"], p["Ap"], p["Ai"], perm)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    inf = r.analyze(perm=perm)
    rs = r.symbolic()
    for k in KEYS:
        assert np.array_equal(sym[k], rs[k]), k
    for k in SIZES:
        assert sym[k] == rs[k], k
    assert sym["fl"] == inf["fl"] and sym["lnz"] == inf["lnz"]
    Sp, Si, Sx = r.permuted()
    Sp2, Si2, Sx2 = symbolic.permute(sym, p["Ap"], p["Ai"], p["Ax"])
    assert np.array_equal(Sp, Sp2) and np.array_equal(Si, Si2) and np.array_equal(Sx, Sx2)


def test_invalid_inputs():
    from suitesparse_amd import engine
    p = problems.laplacian_2d(4)
    with pytest.raises(engine.Mi355Error):
        symbolic.analyze(p["n"], p["Ap"], p["Ai"], np.zeros(p["n"], dtype=np.int64))      # not a permutation
    Ai = p["Ai"].copy()
    Ai[1] = 0                                                                             # upper entry in column 0? (row < col elsewhere)
    Ai[p["Ap"][3]] = 0
    with pytest.raises(engine.Mi355Error):
        symbolic.analyze(p["n"], p["Ap"], Ai, None)


@pytest.mark.parametrize("seed", range(8))
def test_symbolic_vs_live_reference_on_irregular_graphs(seed, have_ref):
    """random sparse symmetric patterns (no grid regularity: forests, dense rows, isolated vertices) under random
    permutations: the chain / contiguous-range amalgamation / bottom-up row structures of symbolic.cpp against
    cholmod_l_analyze (CHOLMOD_GIVEN), bit for bit"""
    if not have_ref:
        pytest.skip("oracle/_ref not available")
    import scipy.sparse as sp
    from oracle.refcholmod import RefCholmod
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 400))
    dens = [0.002, 0.01, 0.03, 0.1][seed % 4]
    M = sp.random(n, n, density=dens, random_state=int(rng.integers(1 << 30)), format="csc")
    if seed % 3 == 0 and n > 10:                        # a few dense rows/columns
        M = sp.lil_matrix(M)
        for r in rng.integers(0, n, 2):
            M[r, rng.integers(0, n, n // 3)] = 1.0
        M = sp.csc_matrix(M)
    A = sp.tril(M + M.T, -1) + sp.identity(n) * (n + 1.0)
    A = sp.csc_matrix(A)
    A.sort_indices()
    Ap, Ai, Ax = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)
    perm = rng.permutation(n).astype(np.int64)
    sym = symbolic.analyze(n, Ap, Ai, perm)
    r = RefCholmod()
    r.load(n, Ap, Ai, Ax, -1)
    inf = r.analyze(perm=perm)
    rs = r.symbolic()
    for k in KEYS:
        assert np.array_equal(sym[k], rs[k]), k
    for k in SIZES:
        assert sym[k] == rs[k], k
    assert sym["fl"] == inf["fl"] and sym["lnz"] == inf["lnz"]
