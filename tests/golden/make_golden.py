"""tests/golden/make_golden.py -- regenerates the committed golden fixtures.

Runs ONLY in the build container (needs /root/reference and oracle/_ref built by
oracle/build_ref.sh).  For each input matrix it records what the UNCHANGED reference
CHOLMOD 5.3.0 CPU path produces:
  A (lower/upper CSC as read), ordering, Perm, ColCount, Super/Lpi/Lpx/Ls, sizes,
  S = tril(P A P') (cholmod_factorize.c:181-251), Lx = cholmod_l_super_numeric(S),
  status, minor, and B -> Y = L\\B -> Z = L'\\Y from cholmod_l_super_lsolve/ltsolve.
Inputs: the reference's own fixture files (CHOLMOD/Demo/Matrix, CHOLMOD/Tcov/Matrix;
SURVEY.md section 8c) and small instances of the synthetic configs.
The fixtures are DATA (inputs + expected outputs), never reference source text.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.refcholmod import RefCholmod  # noqa: E402
from suitesparse_amd import problems  # noqa: E402

REF = "/root/reference/CHOLMOD"
OUT = os.path.dirname(os.path.abspath(__file__))


def record(name, r, ordering, note, beta=0.0, user_perm=None):
    m = r.matrix()
    inf = r.analyze(ordering, perm=user_perm)
    if not inf["is_super"]:
        raise RuntimeError("not supernodal")
    sym = r.symbolic()
    Sp, Si, Sx = r.permuted()
    status = r.factorize(beta=beta)
    inf = r.info()
    Lx = r.Lx()
    n = inf["n"]
    cplx = inf["xtype"] == 2
    B = problems.demo_rhs(n, 3, cplx)
    out = dict(name=name, note=note, n=n, stype=m["stype"], xtype=inf["xtype"], beta=beta,
               Ap=m["Ap"], Ai=m["Ai"], Ax=m["Ax"], ordering=inf["ordering"],
               fl=inf["fl"], lnz=inf["lnz"], nsuper=inf["nsuper"], ssize=inf["ssize"],
               xsize=inf["xsize"], maxcsize=inf["maxcsize"], maxesize=inf["maxesize"],
               Perm=sym["Perm"], ColCount=sym["ColCount"], Super=sym["Super"], Lpi=sym["Lpi"],
               Lpx=sym["Lpx"], Ls=sym["Ls"], Sp=Sp, Si=Si, Sx=Sx, Lx=Lx, status=status,
               minor=inf["minor"], B=B)
    if user_perm is not None:
        out["UserPerm"] = np.asarray(user_perm, dtype=np.int64)
    if status == 0:
        Y = r.super_solve(B, True)
        Z = r.super_solve(Y, False)
        out.update(Y=Y, Z=Z)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name:22s} n={n:6d} xtype={inf['xtype']} ordering={inf['ordering_name']:8s} nsuper={inf['nsuper']:5d} "
          f"xsize={inf['xsize']:8d} status={status} minor={inf['minor']}")


def from_file(name, path, ordering="default", note=""):
    r = RefCholmod()
    r.read(path)
    m = r.matrix()
    if m["stype"] == 0 or m["n"] != m["ncol"]:
        print(f"{name}: unsymmetric, skipped")
        return
    record(name, r, ordering, note or f"reference fixture {os.path.relpath(path, '/root/reference')}")


def record_unsym(name, path, ordering="default", beta=0.0):
    """stype == 0 fixtures (tests/golden/unsym/): the reference factorizes A*A' + beta*I from the pair
    (S, F) = (A(p,:), A(p,:)') that cholmod_factorize hands to cholmod_super_numeric
    (cholmod_factorize.c:201-210,240-250; worker :471-506)."""
    r = RefCholmod()
    m = r.read(path)
    assert m["stype"] == 0
    inf = r.analyze(ordering)
    if not inf["is_super"]:
        raise RuntimeError("not supernodal")
    sym = r.symbolic()
    (Sp, Si, Sx), (Fp, Fi, Fx) = r.permuted_unsym()
    status = r.factorize(beta=beta)
    inf = r.info()
    out = dict(name=name, note=f"reference fixture {os.path.relpath(path, '/root/reference')}, A*A' + {beta}*I",
               n=inf["n"], ncol=m["ncol"], stype=0, xtype=inf["xtype"], beta=beta, Ap=m["Ap"], Ai=m["Ai"], Ax=m["Ax"],
               ordering=inf["ordering"], fl=inf["fl"], lnz=inf["lnz"], nsuper=inf["nsuper"], ssize=inf["ssize"],
               xsize=inf["xsize"], maxcsize=inf["maxcsize"], maxesize=inf["maxesize"], Perm=sym["Perm"],
               ColCount=sym["ColCount"], Super=sym["Super"], Lpi=sym["Lpi"], Lpx=sym["Lpx"], Ls=sym["Ls"],
               Sp=Sp, Si=Si, Sx=Sx, Fp=Fp, Fi=Fi, Fx=Fx, Lx=r.Lx(), status=status, minor=inf["minor"])
    os.makedirs(os.path.join(OUT, "unsym"), exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "unsym", name + ".npz"), **out)
    print(f"unsym/{name:16s} {inf['n']:4d} x {m['ncol']:4d} xtype={inf['xtype']} ordering={inf['ordering_name']:8s} "
          f"nsuper={inf['nsuper']:4d} xsize={inf['xsize']:6d} status={status} minor={inf['minor']}")


def from_problem(name, p, ordering, note, negate_col=None, beta=0.0, user_perm=None):
    r = RefCholmod()
    Ax = p["Ax"].copy()
    if negate_col is not None:
        d = p["Ap"][negate_col]
        assert p["Ai"][d] == negate_col
        Ax[d] = -abs(Ax[d])
    r.load(p["n"], p["Ap"], p["Ai"], Ax, -1)
    record(name, r, ordering, note, beta=beta, user_perm=user_perm)


if __name__ == "__main__":
    for nm, path, kw in (("lp_afiro", f"{REF}/Demo/Matrix/lp_afiro.tri", {}),
                         ("pts5ldd03", f"{REF}/Demo/Matrix/pts5ldd03.mtx", {}),
                         ("tcov_5by50", f"{REF}/Tcov/Matrix/5by50", {}),
                         ("tcov_5by50_beta", f"{REF}/Tcov/Matrix/5by50", {"beta": 0.25}),
                         ("tcov_c10", f"{REF}/Tcov/Matrix/c10", {}),
                         ("tcov_c10_natural", f"{REF}/Tcov/Matrix/c10", {"ordering": "natural"}),
                         ("tcov_galenet", f"{REF}/Tcov/Matrix/galenet", {}),
                         ("tcov_ibm32", f"{REF}/Tcov/Matrix/ibm32", {})):
        # (Tcov/Matrix/3_2 is left out: 3 x 2, A*A' is exactly singular and whether the last pivot comes out as
        # +1e-17 or -1e-17 is rounding noise of the BLAS in use)
        try:
            record_unsym(nm, path, **kw)
        except Exception as e:   # noqa: BLE001
            print(f"unsym/{nm}: skipped ({e})")
    if "--unsym-only" in sys.argv:
        sys.exit(0)
    from_file("demo_bcsstk01", f"{REF}/Demo/Matrix/bcsstk01.tri")
    from_file("demo_bcsstk02", f"{REF}/Demo/Matrix/bcsstk02.tri")
    from_file("demo_c", f"{REF}/Demo/Matrix/c.tri")
    from_file("demo_can24", f"{REF}/Demo/Matrix/can___24.mtx")     # pattern-only: cholmod_read_sparse supplies the values
    for t in ["20lo", "k01up", "ex5lo", "r5lo", "r5lo2", "c10", "c30up", "c30lo_singular", "z30lo", "c5up2", "z5up"]:
        try:
            from_file("tcov_" + t, f"{REF}/Tcov/Matrix/{t}")
        except Exception as e:   # noqa: BLE001
            print(f"tcov_{t}: skipped ({e})")
    from_problem("lap3d_8_metis", problems.laplacian_3d(8), "metis", "3D 7-pt 8^3, METIS")
    from_problem("lap3d_8_amd", problems.laplacian_3d(8), "amd", "3D 7-pt 8^3, AMD")
    from_problem("lap2d_16_metis", problems.laplacian_2d(16), "metis", "2D 5-pt 16^2, METIS")
    from_problem("lap2d_16_nesdis", problems.laplacian_2d(16), "nesdis", "2D 5-pt 16^2, NESDIS")
    from_problem("elast27_4_metis", problems.elasticity27(4), "metis", "27-pt 3-dof 4^3, METIS")
    from_problem("helm3d_6_metis", problems.helmholtz_3d(6), "metis", "complex Hermitian 6^3, METIS")
    from suitesparse_amd import symbolic
    for nm, p in (("nd_lap3d_9", problems.laplacian((9, 8, 7))), ("nd_lap2d_20", problems.laplacian((20, 17))),
                  ("nd_elast27_4", problems.elasticity27(4))):
        from_problem(nm, p, "given", nm + ": CHOLMOD_GIVEN with the geometric ND permutation (UserPerm)",
                     user_perm=symbolic.nd_grid_order(p["shape"], p["dof"]))
    from_problem("lap3d_8_natural", problems.laplacian_3d(8), "natural", "3D 7-pt 8^3, natural ordering")
    from_problem("lap3d_8_beta", problems.laplacian_3d(8), "metis", "3D 7-pt 8^3, METIS, beta=0.5", beta=0.5)
    from_problem("lap3d_8_npd200", problems.laplacian_3d(8), "metis",
                 "3D 7-pt 8^3 with A(200,200) negated: not positive definite", negate_col=200)
    from_problem("lap2d_16_npd0", problems.laplacian_2d(16), "metis",
                 "2D 16^2 with A(0,0) negated: not positive definite", negate_col=0)
    from_problem("elast27_4_npd100", problems.elasticity27(4), "metis",
                 "27-pt 3-dof 4^3 with A(100,100) negated", negate_col=100)
