"""The drop-in boundary compiled for real: the reference CHOLMOD with the insertions of INTEGRATION.md section 2
(oracle/apply_dispatch.py applies them to copies, oracle/build_ref_mi355.sh compiles them together with OUR glue
suitesparse_amd/csrc/cholmod_mi355.c into oracle/_ref/libcholmod_mi355.so) running

        cholmod_l_analyze  ->  cholmod_l_factorize  ->  cholmod_l_solve

through the public API, with the supernodal numeric factorization and the supernodal triangular solves taken over
by the engine behind include/mi355_cholmod.h.  Pass criterion = the reference demo's own:  maxresid < 1e-7 with
resid = ||A x - b||_inf / (||A||_inf ||x||_inf + ||b||_inf)   (CHOLMOD/Demo/cholmod_dl_demo.c:614-617,769-777), on the
demo's matrices bcsstk01, bcsstk02 and c.tri (held as data in tests/golden/demo_*.npz), plus agreement of L->x,
L->minor and Common->status with the unchanged reference's outputs stored in the same fixtures.

Each case exists twice: [cpu] binds the patched CHOLMOD to the CPU build of the engine (tests/helpers/
libmi355_cpuemu.so, test infrastructure), [gpu] (-m gpu) to the real suitesparse_amd/libmi355_cholmod.so.  One process
binds one engine: the two sets run in the two separate pytest invocations (-m "not gpu" / -m gpu)."""
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden, relerr

CPUEMU = os.path.join(ROOT, "tests", "helpers", "libmi355_cpuemu.so")
REAL = os.path.join(ROOT, "suitesparse_amd", "libmi355_cholmod.so")
NOT_POSDEF = 1              # CHOLMOD_NOT_POSDEF (cholmod.h)

BOTH = [pytest.param("cpu", id="cpu"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]


def cases(cpu, gpu_only=()):
    """(which, name) pairs; gpu_only: fixtures that take the CPU build of the engine (an emulation, lane by lane) more
    than a few seconds"""
    out = [pytest.param("cpu", n, id=f"{n}-cpu") for n in cpu]
    out += [pytest.param("gpu", n, id=f"{n}-gpu", marks=pytest.mark.gpu) for n in list(cpu) + list(gpu_only)]
    return out


def _cholmod(which, **kw):
    from oracle import refcholmod
    path = REAL if which == "gpu" else CPUEMU
    present = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_helper_mi355.so")) and os.path.exists(path)
    if not present:
        if which == "gpu" or os.path.isdir("/root/reference/CHOLMOD"):
            pytest.fail("the drop-in build (oracle/build_ref_mi355.sh) or the engine library is missing: "
                        "run __graft_entry__.build()")
        pytest.skip("no reference here to build the patched CHOLMOD from")
    try:
        return refcholmod.RefCholmod(dropin=path, **kw)
    except RuntimeError as e:               # the other engine is already bound in this process
        pytest.skip(str(e))


def _dense_A(g):
    """the full Hermitian matrix of a golden fixture (stored as one triangle, stype +-1)"""
    import scipy.sparse as sp
    n = g["n"]
    T = sp.csc_matrix((g["Ax"], g["Ai"], g["Ap"]), shape=(n, n)).toarray()
    if g["stype"] < 0:
        T = np.tril(T)
    else:
        T = np.triu(T)
    return T + T.conj().T - np.diag(np.diag(T).real)


def _demo_resid(A, X, B):
    inf = lambda M: float(np.abs(M).sum(axis=1).max()) if M.size else 0.0
    return inf(A @ X - B) / (inf(A) * inf(X) + inf(B))


@pytest.mark.parametrize("which", BOTH)
@pytest.mark.parametrize("name", ["demo_bcsstk01", "demo_bcsstk02", "demo_c", "demo_can24"])
def test_demo_chain_residual(which, name):
    """analyze (the demo's default ordering strategy) -> factorize -> solve, forced supernodal, the demo's criterion"""
    g = load_golden(name)
    n = g["n"]
    r = _cholmod(which)
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    inf = r.analyze(ordering=-1, supernodal=2)          # nmethods = 0: the default strategy of cholmod_analyze
    assert inf["is_super"] == 1
    c0 = r.counters()
    assert c0["useGPU"] == 1 and c0["L_useGPU"] == 1, c0
    assert r.factorize_full() == 0
    c1 = r.counters()
    assert c1["gpu_gemm_calls"] + c1["gpu_potrf_calls"] > 0 and c1["cpu_blas_calls"] == 0, c1
    assert r.live_handles() >= 1
    A = _dense_A(g)
    rng = np.random.default_rng(1)
    for B in (np.ones((n, 1)), rng.standard_normal((n, 4))):
        if g["xtype"] == 2:
            B = B + 1j * rng.standard_normal(B.shape)
        t0 = r.calls_taken()
        X = r.solve(B)
        assert _demo_resid(A, X, B) < 1e-7
        t1 = r.calls_taken()
        assert t1[2] - t0[2] == 1 and t1[1] == t0[1], (t0, t1)     # cholmod_l_solve: one trip to the device, not two
    r.close()


@pytest.mark.parametrize("which,name", cases(["demo_bcsstk01", "demo_bcsstk02", "demo_c", "demo_can24", "elast27_4_metis",
                                               "tcov_z30lo", "tcov_c30up", "lap3d_8_metis", "helm3d_6_metis",
                                               "lap2d_16_metis"]))
def test_same_factor_as_unchanged_reference(which, name):
    """on the fixture's permutation the patched build must hand back the SAME cholmod_factor: structure bit-exact,
    L->x within 1e-10 of what the unchanged reference produced (fixture), solution of cholmod_l_solve likewise"""
    g = load_golden(name)
    n = g["n"]
    r = _cholmod(which)
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    s = r.symbolic()
    for k in ("Super", "Lpi", "Lpx", "Ls", "Perm"):
        assert np.array_equal(s[k], g[k]), k
    assert r.factorize_full(beta=g["beta"]) == g["status"] == 0
    assert r.info()["minor"] == g["minor"] == n
    assert relerr(r.Lx(), g["Lx"]) < 1e-10
    # cholmod_l_solve: P, the two supernodal sweeps on the device, P'
    X = r.solve(g["B"])
    A = _dense_A(g) + g["beta"] * np.eye(n)
    want = np.linalg.solve(A, g["B"])
    assert relerr(X, want) < 1e-9
    assert _demo_resid(A, X, g["B"]) < 1e-12
    # the two sweeps one at a time (cholmod_l_super_lsolve / _ltsolve), against the fixture's Y and Z
    Y = r.super_solve(g["B"], lower=True)
    assert relerr(Y, g["Y"]) < 1e-10
    Z = r.super_solve(Y, lower=False)
    assert relerr(Z, g["Z"]) < 1e-10
    r.close()


@pytest.mark.parametrize("which", BOTH)
def test_factorize_hands_over_the_unpermuted_matrix(which):
    """cholmod_l_factorize on a lower-stored A with a fill-reducing ordering: the two cholmod_ptranspose calls that build
    S = tril (P A P') (cholmod_factorize.c:228-238) are skipped, the device places A through L->Perm -- counted by
    cholmod_mi355_calls_taken [3]; an upper-stored or zomplex A, the natural ordering and a conversion that needs S
    afterwards (final_resymbol) keep the host copy.  The factor is the fixture's either way."""
    def run(name, expect_deferred, **options):
        g = load_golden(name)
        n = g["n"]
        r = _cholmod(which)
        if options:
            r.set_option(**options)
        r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
        r.analyze(perm=g["Perm"], supernodal=2)
        for _ in range(2):                                  # the second call refactorizes the same L
            t0 = r.calls_taken()
            assert r.factorize_full(beta=g["beta"]) == g["status"]
            t1 = r.calls_taken()
            assert t1[0] - t0[0] == 1
            assert t1[3] - t0[3] == (1 if expect_deferred else 0), (name, options, t0, t1)
            if not options:
                assert r.info()["minor"] == g["minor"]
                Lx = r.Lx()
                assert np.array_equal(Lx == 0, g["Lx"] == 0) and relerr(Lx, g["Lx"]) < 1e-10
            if g["status"] == 0:
                A = _dense_A(g) + g["beta"] * np.eye(n)
                X = r.solve(g["B"])
                assert relerr(X, np.linalg.solve(A, g["B"])) < 1e-9
            if options:
                break                                       # L was converted: the next factorize is not supernodal
        r.close()
        return g
    stypes = set()
    names = ("helm3d_6_metis", "elast27_4_metis", "lap2d_16_metis", "lap2d_16_npd0")
    for name in names + (("lap3d_8_metis", "lap3d_8_npd200") if which == "gpu" else ()):    # (seconds each on the CPU build)
        stypes.add(int(run(name, True)["stype"]))
    assert stypes == {-1}
    assert run("demo_bcsstk01", False)["stype"] > 0         # upper-stored: one ptranspose on the host, as before
    assert run("tcov_z30lo", False)["xtype"] == 2           # zomplex A: the host copy
    # final form simplicial with resymbol: cholmod_resymbol_noperm reads S after the numeric phase
    run("lap3d_8_metis", False, final_asis=0, final_super=0, final_ll=1, final_resymbol=1)
    run("lap3d_8_metis", True, final_asis=0, final_super=0, final_ll=1, final_resymbol=0)


@pytest.mark.parametrize("which", BOTH)
def test_hand_edited_factor_after_forget(which):
    """INTEGRATION.md, "what the device image assumes": a caller who rewrites L->x by hand calls cholmod_mi355_forget (L);
    the solves then run on the CPU workers from L->x (here L scaled by 2, so the solution shrinks by 4)"""
    g = load_golden("demo_bcsstk02")
    n = g["n"]
    r = _cholmod(which)
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.factorize_full() == 0
    base = r.live_handles()
    X = r.solve(g["B"])
    r.set_Lx(2.0 * r.Lx())
    r.forget()
    assert r.live_handles() == base - 1
    t0 = r.calls_taken()
    X2 = r.solve(g["B"])
    assert r.calls_taken() == t0                    # nothing was taken over: the CPU workers answered
    assert relerr(4.0 * X2, X) < 1e-12
    r.close()


@pytest.mark.parametrize("which,name", cases(["demo_bcsstk02", "lap3d_8_metis", "helm3d_6_metis"]))
def test_hand_edited_factor_is_noticed_without_forget(which, name):
    """the guard behind that rule (round 3): the glue keeps a fingerprint of L->x as the device left it (a few thousand
    sampled entries) and every solve hook compares it first, so a caller who scales L->x and solves WITHOUT calling
    cholmod_mi355_forget gets what the reference gives -- the solution of the edited factor, from the CPU code -- and the
    stale device image is dropped; an untouched factor keeps its device solves"""
    g = load_golden(name)
    n = g["n"]
    r = _cholmod(which)
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.factorize_full() == 0
    base = r.live_handles()
    t0 = r.calls_taken()
    X = r.solve(g["B"])
    assert r.calls_taken() != t0 and r.live_handles() == base       # the device answered, the image stays
    r.set_Lx(2.0 * r.Lx())                                             # ... and now the caller edits L->x and does not say so
    t1 = r.calls_taken()
    X2 = r.solve(g["B"])
    assert r.calls_taken() == t1                                       # the CPU code answered
    assert r.live_handles() == base - 1                                # the stale image is gone
    assert relerr(4.0 * X2, X) < 1e-12
    assert r.factorize_full() == 0                                     # a new factorization brings the device back
    t2 = r.calls_taken()
    X3 = r.solve(g["B"])
    assert r.calls_taken() != t2 and relerr(X3, X) < 1e-12
    r.close()


@pytest.mark.parametrize("which,name", cases(["demo_bcsstk02", "demo_c", "lap3d_8_metis", "helm3d_6_metis", "tcov_z30lo"]))
def test_solve2_branch_on_the_device(which, name):
    """cholmod_l_solve2's supernodal branch (cholmod_solve.c:656-717) taken over as a whole: B down once, P, both sweeps
    and P' on the device, X back once -- for every system of an LL' factor, B real or complex under a real or complex
    L (the reference's "dual" case), a padded B (d > n), reused X / Y / E handles; zomplex B or X and CHOLMOD_D keep
    the host branch (whose sweeps are taken over one by one).  Expected values: dense algebra on P A P'."""
    g = load_golden(name)
    n = g["n"]
    r = _cholmod(which)
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.factorize_full(beta=g["beta"]) == 0
    A = _dense_A(g) + g["beta"] * np.eye(n)
    P = np.asarray(g["Perm"])
    S = A[np.ix_(P, P)]
    Ld = np.linalg.cholesky(S)
    rng = np.random.default_rng(3)
    Lcplx = g["xtype"] == 2
    want = {"A": lambda B: np.linalg.solve(A, B), "LDLt": lambda B: np.linalg.solve(S, B),
            "LD": lambda B: np.linalg.solve(Ld, B), "L": lambda B: np.linalg.solve(Ld, B),
            "DLt": lambda B: np.linalg.solve(Ld.conj().T, B), "Lt": lambda B: np.linalg.solve(Ld.conj().T, B),
            "D": lambda B: B}
    for bc in (False, True):
        for nrhs, ldb in ((1, n), (5, n + 3)):
            B = rng.standard_normal((n, nrhs)) + (1j * rng.standard_normal((n, nrhs)) if bc else 0)
            for sys in ("A", "LDLt", "LD", "DLt", "L", "Lt", "D"):
                t0 = r.calls_taken()
                X = r.solve2(sys, B, ldb=ldb, reps=2)
                t1 = r.calls_taken()
                assert np.iscomplexobj(X) == (bc or Lcplx)
                assert relerr(X, want[sys](B)) < 1e-9, (sys, bc, nrhs)
                if sys == "D":
                    assert t1 == t0
                else:
                    assert t1[2] - t0[2] == 2 and t1[1] == t0[1], (sys, t0, t1)
    # zomplex B, or a zomplex X asked for: the host branch, its sweeps on the device one at a time
    B = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
    for kw in (dict(zomplex_b=True), dict(prefer_zomplex=True)):
        t0 = r.calls_taken()
        X = r.solve2("A", B, **kw)
        t1 = r.calls_taken()
        assert relerr(X, np.linalg.solve(A, B)) < 1e-9
        assert t1[2] == t0[2] and t1[1] - t0[1] == 2, (kw, t0, t1)
    # after a refactorization the same device image (and its Perm) is used again
    assert r.factorize_full(beta=g["beta"]) == 0
    B = rng.standard_normal((n, 3))
    assert relerr(r.solve2("A", B), np.linalg.solve(A, B)) < 1e-9
    r.close()


@pytest.mark.parametrize("which,name", cases(["tcov_c30lo_singular", "elast27_4_npd100", "lap3d_8_npd200",
                                              "lap2d_16_npd0"]))
def test_not_posdef_through_the_api(which, name):
    """a warning, not an error: Common->status = CHOLMOD_NOT_POSDEF, L->minor = the failing column, the columns
    before it as the reference leaves them, the rest zero (t_cholmod_super_numeric_worker.c:1084-1146)"""
    g = load_golden(name)
    n = g["n"]
    r = _cholmod(which)
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.factorize_full(beta=g["beta"]) == NOT_POSDEF == g["status"]
    assert r.info()["minor"] == g["minor"] < n
    Lx = r.Lx()
    assert np.array_equal(Lx == 0, g["Lx"] == 0)
    assert relerr(Lx, g["Lx"]) < 1e-10
    # cholmod_l_solve on such a factor stays on the CPU worker (it sweeps what is in L->x) and must not fail
    X = r.solve(g["B"])
    assert X.shape == g["B"].shape
    r.close()


@pytest.mark.parametrize("which,name", cases(["demo_bcsstk02", "lap3d_8_metis"]))
def test_refactorize_reuses_the_device_image_and_teardown(which, name):
    """second cholmod_l_factorize on the same L: same handle, new values; cholmod_l_free_factor and
    cholmod_l_change_factor (supernodal -> simplicial) destroy the device image; the converted factor still solves"""
    g = load_golden(name)
    n = g["n"]
    r = _cholmod(which)
    base = r.live_handles()
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.live_handles() == base                 # analysis reserves nothing on the device
    assert r.factorize_full() == 0
    assert r.live_handles() == base + 1
    L1 = r.Lx()
    shift = 0.5 * float(np.abs(g["Ax"]).max())
    assert r.factorize_full(beta=shift) == 0        # A + shift I
    assert r.live_handles() == base + 1
    A = _dense_A(g) + shift * np.eye(n)
    X = r.solve(g["B"])
    assert _demo_resid(A, X, g["B"]) < 1e-12
    assert relerr(r.Lx(), L1) > 1e-3
    assert r.factorize_full() == 0
    assert relerr(r.Lx(), g["Lx"]) < 1e-10
    # supernodal -> simplicial LL': the device image goes, the values stay, the simplicial solve works
    assert r.make_simplicial(True) == 0
    assert r.live_handles() == base
    assert r.info()["is_super"] == 0
    X = r.solve(g["B"])
    assert _demo_resid(_dense_A(g), X, g["B"]) < 1e-12
    # a new factor, freed while it has an image
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.factorize_full() == 0
    assert r.live_handles() == base + 1
    r.drop_factor()
    assert r.live_handles() == base
    r.close()


@pytest.mark.parametrize("which", BOTH)
def test_use_gpu_off_runs_the_cpu_workers(which):
    """Common->useGPU = 0: the hooks step aside, the unchanged CPU workers run, no device image appears"""
    g = load_golden("lap3d_8_metis")
    r = _cholmod(which)
    base = r.live_handles()
    r.set_use_gpu(0)
    r.load(g["n"], g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.factorize_full() == 0
    c = r.counters()
    assert c["L_useGPU"] == 0 and c["gpu_gemm_calls"] == 0 and c["gpu_potrf_calls"] == 0, c
    assert r.live_handles() == base
    assert relerr(r.Lx(), g["Lx"]) < 1e-12
    assert relerr(r.solve(g["B"]), np.linalg.solve(_dense_A(g), g["B"])) < 1e-9
    r.close()


@pytest.mark.parametrize("which", BOTH)
def test_cpu_refactorization_drops_a_stale_image(which):
    """factorize on the device, then switch Common->useGPU off and factorize again: the CPU worker rewrites L->x, so
    the device image must go, and the following solve must use the NEW values"""
    g = load_golden("demo_bcsstk01")
    n = g["n"]
    r = _cholmod(which)
    base = r.live_handles()
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.factorize_full() == 0
    assert r.live_handles() == base + 1
    r.set_use_gpu(0)
    shift = float(np.abs(g["Ax"]).max())
    assert r.factorize_full(beta=shift) == 0
    assert r.live_handles() == base
    A = _dense_A(g) + shift * np.eye(n)
    assert _demo_resid(A, r.solve(g["B"]), g["B"]) < 1e-12
    r.close()


@pytest.mark.parametrize("which", BOTH)
@pytest.mark.parametrize("name", ["tcov_5by50", "tcov_5by50_beta", "tcov_c10", "lp_afiro"])
def test_unsymmetric_chain(which, name):
    """stype == 0: cholmod_l_factorize forms F = A(p,:)' and hands (A, F) to super_numeric, which factorizes
    A F + beta I (cholmod_factorize.c:201-250); here the hook passes both to mi355_factor_set_matrix_unsym"""
    import test_unsym as tu
    g = tu.load_unsym(name)
    r = _cholmod(which)
    r.load_rect(g["n"], g["ncol"], g["Ap"], g["Ai"], g["Ax"])
    r.analyze(perm=g["Perm"], supernodal=2)
    s = r.symbolic()
    for k in ("Super", "Lpi", "Lpx", "Ls"):
        assert np.array_equal(s[k], g[k]), k
    assert r.factorize_full(beta=g["beta"]) == g["status"]
    assert r.info()["minor"] == g["minor"]
    c = r.counters()
    assert c["L_useGPU"] == 1 and c["cpu_blas_calls"] == 0, c
    assert np.array_equal(r.Lx() == 0, g["Lx"] == 0)
    assert relerr(r.Lx(), g["Lx"]) < 1e-10
    r.close()


@pytest.mark.parametrize("which", BOTH)
def test_blas_statistics_are_filled(which):
    """Common->cholmod_gpu_*_calls/_time (cholmod.h:966-986; cholmod_l_gpu_stats prints them,
    CHOLMOD/Check/cholmod_check.c:608-672): the patched build is compiled with BLAS_TIMER as the reference's GPU
    build is, so both calls and times must move with every factorization"""
    g = load_golden("lap3d_8_metis")      # (several supernodes: update launches as well as panel launches)
    r = _cholmod(which)
    r.load(g["n"], g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.factorize_full() == 0
    c1 = r.counters()
    assert c1["gpu_gemm_calls"] > 0 and c1["gpu_potrf_calls"] > 0, c1
    assert c1["gpu_gemm_time"] > 0 and c1["gpu_potrf_time"] > 0, c1
    assert r.factorize_full() == 0
    c2 = r.counters()
    assert c2["gpu_gemm_calls"] == 2 * c1["gpu_gemm_calls"] and c2["gpu_potrf_calls"] == 2 * c1["gpu_potrf_calls"]
    assert c2["gpu_gemm_time"] > c1["gpu_gemm_time"]
    r.close()


def test_integration_md_lists_what_is_compiled():
    """every line oracle/apply_dispatch.py inserts into the reference's files appears verbatim in INTEGRATION.md"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("apply_dispatch", os.path.join(ROOT, "oracle", "apply_dispatch.py"))
    ad = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ad)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for ln in ad.INCLUDE:
        assert ln in doc, ln
    for rel, edits in ad.EDITS.items():
        assert rel in doc, rel
        for _, _, ins in edits:
            for ln in ins or ():
                assert ln.strip() in doc, (rel, ln)


def _both_builds(which, g):
    """(drop-in build, unchanged reference), the same matrix loaded into each"""
    from oracle import refcholmod
    r = _cholmod(which)
    u = refcholmod.RefCholmod()
    for x in (r, u):
        x.load(g["n"], g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    return r, u


@pytest.mark.parametrize("which,name", cases(["demo_bcsstk01", "demo_bcsstk02", "demo_c", "demo_can24",
                                              "tcov_c30lo_singular", "tcov_z30lo", "lap3d_8_metis", "lap2d_16_npd0"]))
def test_raw_factor_expert_sequence(which, name):
    """The reference's own raw_factor test (CHOLMOD/Tcov/t_raw_factor.c:109-340) through the patched build: no
    ordering, alloc_factor -> etree -> postorder -> rowcolcounts -> super_symbolic -> super_numeric -> super_lsolve ->
    super_ltsolve called one by one, plus every wrong-argument call of that test -- each must be refused exactly as
    the unchanged build refuses it (the hooks sit behind the wrappers' own checks) -- and the same residual, status
    and minor as the unchanged build on the same matrix."""
    g = load_golden(name)
    r, u = _both_builds(which, g)
    got, want = r.raw_factor(True), u.raw_factor(True)
    assert want["ok"] and want["refused"] == want["expected"] >= 9
    assert got["ok"] and (got["refused"], got["expected"]) == (want["refused"], want["expected"])
    assert (got["status"], got["minor"]) == (want["status"], want["minor"])
    c = r.counters()
    assert c["L_useGPU"] == 1 and c["gpu_gemm_calls"] + c["gpu_potrf_calls"] > 0 and c["cpu_blas_calls"] == 0, c
    if want["status"] == 0:
        assert want["resid"] < 1e-12 and got["resid"] < max(10 * want["resid"], 1e-14)
        assert relerr(r.Lx(), u.Lx()) < 1e-10
    else:
        assert got["status"] == NOT_POSDEF
        assert np.array_equal(r.Lx() == 0, u.Lx() == 0) and relerr(r.Lx(), u.Lx()) < 1e-10
    r.close(); u.close()


SOLVER_SETTINGS = {      # CHOLMOD/Tcov/t_solve.c:61-124, the supernodal runs of test_solver
    "supernodal": dict(nthreads_max=-1),
    "no_final_resymbol": dict(final_resymbol=0),
    "final_super_false": dict(final_asis=0, final_resymbol=1, final_super=0),
    "final_super_true": dict(final_asis=0, final_resymbol=1, final_super=1),
    "no_relaxed_supernodes": dict(final_asis=1, nrelax0=0, nrelax1=0, nrelax2=0, zrelax0=0, zrelax1=0, zrelax2=0,
                                  grow0=1, grow1=1, grow2=0),
}


@pytest.mark.parametrize("which", BOTH)
@pytest.mark.parametrize("setting", list(SOLVER_SETTINGS))
def test_solver_settings_of_the_reference(which, setting):
    """test_solver's supernodal runs: cm->supernodal = CHOLMOD_SUPERNODAL under the final_* / relaxation settings the
    reference varies; analyze -> factorize -> solve must give the unchanged build's answer, whatever form the factor
    is left in (final_super = FALSE converts it to simplicial right after the numeric phase: the device image must
    be gone and the solve runs on the CPU)"""
    for name in ("demo_bcsstk02", "demo_c"):
        g = load_golden(name)
        r, u = _both_builds(which, g)
        base = r.live_handles()
        for x in (r, u):
            x.set_option(**SOLVER_SETTINGS[setting])
            x.analyze(perm=g["Perm"], supernodal=2)
            assert x.factorize_full() == 0
        ir, iu = r.info(), u.info()
        assert (ir["is_super"], ir["is_ll"], ir["minor"]) == (iu["is_super"], iu["is_ll"], iu["minor"])
        assert r.counters()["gpu_gemm_calls"] + r.counters()["gpu_potrf_calls"] > 0
        assert r.live_handles() == base + (1 if ir["is_super"] else 0)
        if ir["is_super"]:
            for k in ("Super", "Lpi", "Lpx", "Ls"):
                assert np.array_equal(r.symbolic()[k], u.symbolic()[k]), k
            assert relerr(r.Lx(), u.Lx()) < 1e-10
        X, Xu = r.solve(g["B"]), u.solve(g["B"])
        assert relerr(X, Xu) < 1e-9
        assert _demo_resid(_dense_A(g), X, g["B"]) < 1e-7
        r.close(); u.close()


@pytest.mark.parametrize("which", BOTH)
@pytest.mark.parametrize("seed", range(10))
def test_fuzz_public_api_against_the_unchanged_build(which, seed):
    """random SPD / Hermitian positive definite structures (sparse graphs, chains, arrows, dense blocks), lower-stored,
    CHOLMOD's own ordering strategy, through analyze -> factorize -> solve2 of both builds: the same Perm and supernodal
    structure (the hooks change nothing symbolic), L->x within 1e-10, the same solutions for A, L and Lt systems;
    every third seed gets a negative pivot (status, minor and zero pattern as the unchanged build)"""
    import scipy.sparse as sp
    import test_fuzz_schedule as tf
    rng = np.random.default_rng(500 + seed)
    kind = ["sparse", "banded", "arrow", "blocks"][seed % 4]
    n = int(rng.integers(60, 400))
    Ap, Ai, Ax = tf.random_spd(kind, n, 40 + seed)
    cplx = seed % 2 == 1
    if cplx:                                    # Hermitian: imaginary parts off the diagonal only, magnitudes unchanged
        ph = np.exp(1j * rng.uniform(0, 2 * np.pi, len(Ax)))
        col = np.repeat(np.arange(n), np.diff(Ap))
        ph[Ai == col] = 1.0
        Ax = Ax * ph
    spoil = seed % 3 == 2
    if spoil:
        k = int(rng.integers(0, n))
        assert Ai[Ap[k]] == k
        Ax = Ax.copy(); Ax[Ap[k]] = -abs(Ax[Ap[k]])
    g = dict(n=n, Ap=Ap, Ai=Ai, Ax=Ax, stype=-1)
    r, u = _both_builds(which, g)
    for x in (r, u):
        inf = x.analyze(ordering=-1, supernodal=2)
        assert inf["is_super"] == 1
    sr, su = r.symbolic(), u.symbolic()
    for k in ("Perm", "Super", "Lpi", "Lpx", "Ls"):
        assert np.array_equal(sr[k], su[k]), k
    t0 = r.calls_taken()
    st_r, st_u = r.factorize_full(), u.factorize_full()
    t1 = r.calls_taken()
    assert st_r == st_u == (NOT_POSDEF if spoil else 0)
    assert t1[0] - t0[0] == 1 and r.counters()["cpu_blas_calls"] == 0
    assert r.info()["minor"] == u.info()["minor"]
    Lr, Lu = r.Lx(), u.Lx()
    assert np.array_equal(Lr == 0, Lu == 0) and relerr(Lr, Lu) < 1e-10
    if not spoil:
        B = rng.standard_normal((n, 3)) + (1j * rng.standard_normal((n, 3)) if seed % 4 == 3 else 0)
        for sys in ("A", "L", "Lt"):
            assert relerr(r.solve2(sys, B), u.solve2(sys, B)) < 1e-9, sys
        A = _dense_A(g)
        assert _demo_resid(A, r.solve2("A", B), B) < 1e-12
    r.close(); u.close()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_full_size_dropin_factorize_and_solve():
    """cholmod_l_analyze -> cholmod_l_factorize -> cholmod_l_solve through the patched CHOLMOD on the real library at a
    size where time matters (3D 7-point 64^3, lower-stored A, the engine's nested-dissection permutation given to
    analyze), beside the unchanged build on the host cores: the same factor to 1e-10, the demo's residual criterion,
    and the wall-clock times of the PUBLIC calls -- what a CHOLMOD user sees, host copies of L->x and of the right-hand
    sides included -- written to gpurun_out/r02/dropin_timing.json (the first factorize call also creates the handle)."""
    import json
    import time
    from suitesparse_amd import problems, symbolic
    p = problems.laplacian_3d(64)
    perm = symbolic.nd_grid_order(p["shape"], p["dof"])
    n = p["n"]
    g = dict(n=n, Ap=p["Ap"], Ai=p["Ai"], Ax=p["Ax"], stype=-1)
    r, u = _both_builds("gpu", g)
    t = {}
    for name, x in (("dropin", r), ("reference_cpu", u)):
        t0 = time.perf_counter(); inf = x.analyze(perm=perm, supernodal=2); t[name + "_analyze_s"] = time.perf_counter() - t0
        assert inf["is_super"] == 1
        for k in ("first", "second"):
            t0 = time.perf_counter(); st = x.factorize_full(); t[f"{name}_factorize_{k}_s"] = time.perf_counter() - t0
            assert st == 0
    assert r.calls_taken()[3] >= 2                      # the unpermuted A went to the device both times
    fl = float(r.info().get("fl", 0.0)) or None
    Lr, Lu = r.Lx(), u.Lx()
    assert relerr(Lr, Lu) < 1e-10
    rng = np.random.default_rng(0)
    for nrhs in (1, 16):
        B = rng.standard_normal((n, nrhs))
        for name, x in (("dropin", r), ("reference_cpu", u)):
            x.solve2("A", B)                            # warm-up (workspaces, device images)
            t0 = time.perf_counter(); X = x.solve2("A", B); t[f"{name}_solve_nrhs{nrhs}_s"] = time.perf_counter() - t0
            if name == "dropin":
                Xr = X
        assert relerr(Xr, X) < 1e-9
    t.update(problem="lap3d 64", n=n, xsize=int(r.symbolic()["Lpx"][-1]), fl=fl, threads_reference=u.nthreads())
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r02"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r02", "dropin_timing.json"), "w") as fh:
        json.dump(t, fh, indent=1)
    print(json.dumps(t))
    assert t["dropin_factorize_second_s"] < t["reference_cpu_factorize_second_s"]
    r.close(); u.close()


# ---- soft fallback, memory limits, allocation failures (VERDICT r02 "missing" 4) ---------------------------------------

def _factor_and_check(r, g, expect_device):
    """analyze on the fixture's permutation, cholmod_l_factorize, cholmod_l_solve; the factor must be the fixture's
    whoever computed it; returns the counters after the factorization"""
    n = g["n"]
    r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    t0 = r.calls_taken()
    c0 = r.counters()
    assert r.factorize_full(beta=g["beta"]) == g["status"] == 0
    t1 = r.calls_taken()
    c = r.counters()
    assert relerr(r.Lx(), g["Lx"]) < 1e-10
    assert r.info()["minor"] == n
    if expect_device is not None:
        assert (t1[0] - t0[0] == 1) == expect_device, (t0, t1)
        # the CPU worker zeroes its BLAS counters when it starts (t_cholmod_super_numeric_worker.c:246-266) and counts its
        # calls; a device run leaves them as they were
        if expect_device:
            assert c["cpu_blas_calls"] == c0["cpu_blas_calls"] and c["cpu_blas_time"] == c0["cpu_blas_time"], (c0, c)
        else:
            assert c["cpu_blas_calls"] > 0 and c["cpu_blas_time"] != c0["cpu_blas_time"], (c0, c)
        assert c["L_useGPU"] == (1 if expect_device else 0), c
    A = _dense_A(g) + g["beta"] * np.eye(n)
    X = r.solve(g["B"])
    assert relerr(X, np.linalg.solve(A, g["B"])) < 1e-9
    return c


@pytest.mark.parametrize("which", BOTH)
@pytest.mark.parametrize("name", ["lap2d_16_metis", "demo_bcsstk01", "helm3d_6_metis"])
def test_soft_fallback_when_the_factor_exceeds_the_memory_limit(which, name):
    """Common->maxGpuMemBytes / maxGpuMemFraction (CHOLMOD/GPU/cholmod_gpu.c:446-489) cap what a device image may take; a
    factor that needs more is factorized by the unchanged CPU worker -- silently, status CHOLMOD_OK, L->useGPU = 0 -- as
    the reference does when gpu_init cannot set itself up (t_cholmod_super_numeric_worker.c:233-244).  lap2d_16 and
    helm3d_6 are lower-stored with a permutation: cholmod_factorize_p hands A itself over (deferred permutation), so the
    fallback first builds S = tril (P A P') the way cholmod_factorize_p would have."""
    g = load_golden(name)
    r = _cholmod(which)
    base = r.live_handles()
    f0, _ = r.fallbacks()
    r.set_gpu_mem(max_bytes=4096)                        # nothing fits into 4 KB
    _factor_and_check(r, g, expect_device=False)
    f1, why = r.fallbacks()
    assert f1 == f0 + 1 and "device memory" in why, (f0, f1, why)
    assert r.live_handles() == base
    # a fraction works the same way
    r.set_gpu_mem(max_fraction=1e-9)
    _factor_and_check(r, g, expect_device=False)
    assert r.fallbacks()[0] == f1 + 1
    # the limit lifted: the next analysis asks for the device again and gets it
    r.set_gpu_mem(0, 0.0)
    _factor_and_check(r, g, expect_device=True)
    assert r.fallbacks()[0] == f1 + 1
    assert r.live_handles() == base + 1
    r.close()
    assert r.L.ref_live_handles() == base


@pytest.mark.parametrize("which", BOTH)
def test_memory_knobs_from_the_environment(which, monkeypatch):
    """Common->useGPU = EMPTY: CHOLMOD_USE_GPU decides, and CHOLMOD_GPU_MEM_BYTES / CHOLMOD_GPU_MEM_FRACTION set
    Common->maxGpuMemBytes / maxGpuMemFraction (fraction clamped to [0, 1]) -- cholmod_super_symbolic.c:250-289"""
    g = load_golden("demo_bcsstk02")
    r = _cholmod(which)
    r.set_use_gpu(-1)
    monkeypatch.setenv("CHOLMOD_USE_GPU", "1")
    monkeypatch.setenv("CHOLMOD_GPU_MEM_BYTES", "123456789")
    monkeypatch.setenv("CHOLMOD_GPU_MEM_FRACTION", "7.5")
    r.load(g["n"], g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.counters()["useGPU"] == 1
    assert r.gpu_mem() == (123456789.0, 1.0)
    assert r.factorize_full() == 0                       # 123 MB are plenty for bcsstk02
    assert r.counters()["cpu_blas_calls"] == 0
    r.close()
    # CHOLMOD_USE_GPU=0: the limits are not even read
    r = _cholmod(which)
    r.set_use_gpu(-1)
    monkeypatch.setenv("CHOLMOD_USE_GPU", "0")
    monkeypatch.setenv("CHOLMOD_GPU_MEM_BYTES", "5")
    r.load(g["n"], g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
    r.analyze(perm=g["Perm"], supernodal=2)
    assert r.counters()["useGPU"] == 0 and r.gpu_mem()[0] == 0.0
    r.close()


@pytest.mark.parametrize("name", ["demo_bcsstk02", "lap2d_16_metis"])
def test_every_allocation_failure_falls_back_to_the_cpu(name):
    """the reference's Tcov suite fails every malloc in turn (CHOLMOD/Tcov/t_memory.c, t_raw_factor.c:343-347); here the
    k-th device / pinned allocation of analyze -> factorize -> solve fails, for every k until a run needs fewer than k:
    each run must still return the reference's factor and solution (the CPU workers take over what the device could not
    do), leave no device image behind when the numeric phase fell back, and leak nothing.  CPU build of the engine only:
    its stand-in for the HIP runtime carries the injection (tests/helpers/hipcpu/hip/hip_runtime.h)."""
    import ctypes as C
    g = load_golden(name)
    r = _cholmod("cpu")
    emu = C.CDLL(CPUEMU)
    emu.hipcpu_alloc_calls.restype = C.c_long
    emu.hipcpu_live_allocs.restype = C.c_long
    emu.hipcpu_fail_alloc_at.argtypes = [C.c_long]
    r.close()
    base_handles = _cholmod("cpu").L.ref_live_handles()
    base_live = emu.hipcpu_live_allocs()
    k, ran_clean, fell_back = 1, False, 0
    try:
        while not ran_clean:
            assert k < 400, "allocation count does not converge"
            r = _cholmod("cpu")
            f0 = r.fallbacks()[0]
            emu.hipcpu_fail_alloc_at(k)
            _factor_and_check(r, g, expect_device=None)
            used = emu.hipcpu_alloc_calls()
            emu.hipcpu_fail_alloc_at(0)
            ran_clean = used < k                        # the run made fewer than k allocations: nothing was injected
            fell_back += r.fallbacks()[0] - f0
            r.close()
            assert r.L.ref_live_handles() == base_handles, k
            assert emu.hipcpu_live_allocs() == base_live, (k, emu.hipcpu_live_allocs(), base_live)
            k += 1
    finally:
        emu.hipcpu_fail_alloc_at(0)
    # every injected failure ended in a fallback, none in an error -- except the three allocations of the solve that are optional
    # (the explicit inverses of the solve groups, their scratch tiles and tickets: engine.hip, ensure_group_inverses): without them
    # the sweeps run the chain kernels and the run above still had to give the reference's solution
    assert k > 10 and fell_back >= k - 2 - 3, (k, fell_back)


@pytest.mark.parametrize("name,world", [("lap3d_8_metis", 2), ("helm3d_6_metis", 3), ("lap2d_16_npd0", 2)])
def test_n_ranks_through_cholmod(name, world):
    """cholmod_mi355_register_comm (VERDICT r02 "missing" 5): one THREAD per rank, each with its own cholmod_common, the
    same matrix and analysis, a blocking transport of Python callbacks as the engine communicator (tests/test_multi_abi.py's
    Bus); cholmod_l_factorize on every rank = mi355_factor_partition + mi355_factor_numeric_multi + mi355_factor_gather_multi,
    and every rank's L->x is the complete factor of the fixture (incl. the not-positive-definite protocol: same status,
    minor and zero pattern on every rank); cholmod_l_solve then runs on each rank's host copy.  CPU build of the engine."""
    from suitesparse_amd import engine as e, symbolic
    import test_multi_abi as tm
    g = load_golden(name)
    n = g["n"]
    _cholmod("cpu").close()                     # (skips here when the drop-in build is absent)
    saved = (e.LIB_PATH, e._LIB)
    e.LIB_PATH, e._LIB = CPUEMU, None
    symbolic._ready = False
    bus = tm.Bus(world)
    out = [None] * world
    try:
        def body(rank):
            r = _cholmod("cpu")
            comm = bus.comm(e, rank)
            assert r.register_comm(comm._h, rank, world) == 1
            r.load(n, g["Ap"], g["Ai"], g["Ax"], stype=g["stype"])
            r.analyze(perm=g["Perm"], supernodal=2)
            t0 = r.calls_taken()
            for _ in range(2):                  # the second call refactorizes on the same partition
                st = r.factorize_full(beta=g["beta"])
                Lx = r.Lx()
                X = r.solve(g["B"]) if g["status"] == 0 else None
            out[rank] = (st, r.info()["minor"], Lx, X, r.calls_taken()[0] - t0[0], r.counters()["cpu_blas_calls"])
            assert r.register_comm(None, 0, 0) == 1
            r.close()
            comm.close()
        tm.run_ranks(world, body)
    finally:
        e.LIB_PATH, e._LIB = saved
        symbolic._ready = False
    for rank in range(world):
        st, minor, Lx, X, taken, cpu_calls = out[rank]
        assert st == g["status"] and minor == g["minor"], (rank, st, minor)
        assert taken >= 2 and cpu_calls == 0, (rank, taken, cpu_calls)        # (the counter is the process's: all ranks add to it)
        assert np.array_equal(Lx == 0, g["Lx"] == 0) and relerr(Lx, g["Lx"]) < 1e-10, rank
        if g["status"] == 0:
            A = _dense_A(g) + g["beta"] * np.eye(n)
            assert relerr(X, np.linalg.solve(A, g["B"])) < 1e-9
    # every rank made the same collectives in the same order
    assert all(bus.log[r] == bus.log[0] for r in range(world)) and len(bus.log[0]) > 0
