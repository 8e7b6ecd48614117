"""CPU tests: the plain-C oracle (oracle/super_oracle.c) against the committed golden
fixtures produced by the unchanged reference (tests/golden/make_golden.py), and -- where
oracle/_ref is present -- against the reference run live on other sizes."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, sym_of, relerr
from oracle import port

TOL_L = 1e-12      # oracle vs reference L: both are CPU double, differ only by BLAS summation order


@pytest.mark.parametrize("name", golden_names())
def test_oracle_numeric_matches_reference_golden(name):
    g = load_golden(name)
    st, minor, Lx = port.super_numeric(sym_of(g), g["Sp"], g["Si"], g["Sx"], beta=g["beta"])
    assert st == g["status"]
    assert minor == g["minor"]
    assert relerr(Lx, g["Lx"]) < TOL_L
    if g["status"] == 1:
        # reference zeroing pattern: failing supernode beyond the good columns + everything after
        assert np.array_equal(Lx == 0, g["Lx"] == 0)


@pytest.mark.parametrize("name", [n for n in golden_names()])
def test_oracle_solve_matches_reference_golden(name):
    g = load_golden(name)
    if g["status"] != 0:
        pytest.skip("not positive definite")
    sym = sym_of(g)
    Y = port.super_solve(sym, g["Lx"], g["B"], lower=True)
    Z = port.super_solve(sym, g["Lx"], Y, lower=False)
    assert relerr(Y, g["Y"]) < 1e-11
    assert relerr(Z, g["Z"]) < 1e-11


def test_golden_structure_matches_survey_table():
    # SURVEY.md Appendix A small-fixture table (bcsstk01 forced supernodal)
    g = load_golden("demo_bcsstk01")
    assert g["n"] == 48 and g["nsuper"] == 7 and g["ssize"] == 101 and g["xsize"] == 1064
    assert g["maxcsize"] == 169 and g["maxesize"] == 13
    assert list(g["Super"]) == [0, 1, 13, 21, 22, 23, 24, 48]
    assert list(g["Lpi"]) == [0, 8, 32, 53, 61, 69, 77, 101]
    assert list(g["Lpx"]) == [0, 8, 296, 464, 472, 480, 488, 1064]
    g = load_golden("demo_bcsstk02")
    assert g["nsuper"] == 1 and g["xsize"] == 4356


def test_oracle_quick_return_zeroes_whole_supernode():
    g = load_golden("lap3d_8_npd200")
    sym = sym_of(g)
    st, minor, Lx = port.super_numeric(sym, g["Sp"], g["Si"], g["Sx"], quick_return=True)
    assert st == 1 and minor == g["minor"]
    s = int(np.searchsorted(g["Super"], minor, side="right") - 1)
    assert np.all(Lx[g["Lpx"][s]:] == 0)
    assert relerr(Lx[:g["Lpx"][s]], g["Lx"][:g["Lpx"][s]]) < TOL_L


@pytest.mark.parametrize("kind,size,ordering", [("lap3d", 11, "metis"), ("lap2d", 40, "amd"),
                                                ("elast", 5, "nesdis"), ("helm", 7, "metis")])
def test_oracle_vs_live_reference(kind, size, ordering, have_ref):
    if not have_ref:
        pytest.skip("oracle/_ref not available")
    from oracle.refcholmod import RefCholmod
    from suitesparse_amd import problems
    p = {"lap3d": problems.laplacian_3d, "lap2d": problems.laplacian_2d,
         "elast": problems.elasticity27, "helm": problems.helmholtz_3d}[kind](size)
    r = RefCholmod()
    r.load(p["n"], p["Ap"], p["Ai"], p["Ax"], -1)
    r.analyze(ordering)
    Sp, Si, Sx = r.permuted()
    assert r.factorize() == 0
    sym = r.symbolic()
    st, minor, Lx = port.super_numeric(sym, Sp, Si, Sx)
    assert st == 0 and minor == p["n"]
    assert relerr(Lx, r.Lx()) < TOL_L
    B = problems.demo_rhs(p["n"], 2, np.iscomplexobj(p["Ax"]))
    X = port.super_solve(sym, Lx, port.super_solve(sym, Lx, B, True), False)
    Xr = r.super_solve(r.super_solve(B, True), False)
    assert relerr(X, Xr) < 1e-11
