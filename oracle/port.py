"""oracle/port.py -- TEST INFRASTRUCTURE: ctypes binding of oracle/liboracle_port.so
(the plain-C restatement in oracle/super_oracle.c).  Used only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(_HERE, "liboracle_port.so"))
        L.oracle_super_numeric.argtypes = [C.c_int, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p,
                                           C.c_int64, C.c_int64, i64p, i64p, C.c_void_p, f64p,
                                           C.c_double, C.c_int, f64p, i64p]
        L.oracle_super_numeric.restype = C.c_int
        L.oracle_super_numeric_f.argtypes = [C.c_int, C.c_int64, C.c_int64, i64p, i64p, i64p, i64p,
                                             C.c_int64, C.c_int64, i64p, i64p, C.c_void_p, f64p,
                                             i64p, i64p, C.c_void_p, f64p,
                                             C.c_double, C.c_int, f64p, i64p]
        L.oracle_super_numeric_f.restype = C.c_int
        for f in (L.oracle_super_lsolve, L.oracle_super_ltsolve):
            f.argtypes = [C.c_int, C.c_int64, i64p, i64p, i64p, i64p, f64p, C.c_int64, f64p, C.c_int64]
            f.restype = C.c_int
        _LIB = L
    return _LIB


def _flat(x):
    x = np.ascontiguousarray(x)
    if np.iscomplexobj(x):
        return np.ascontiguousarray(x, dtype=np.complex128).view(np.float64), 2
    return np.ascontiguousarray(x, dtype=np.float64), 1


def super_numeric(sym, Sp, Si, Sx, beta=0.0, quick_return=False):
    """sym: dict with n nsuper Super Lpi Lpx Ls xsize maxcsize.  Returns (status, minor, Lx)."""
    Sxf, E = _flat(Sx)
    if Sxf.size == 0:
        Sxf = np.zeros(2)
    Lx = np.zeros(max(int(sym["xsize"]), 1) * E)
    minor = np.zeros(1, dtype=np.int64)
    Ls = np.ascontiguousarray(sym["Ls"], dtype=np.int64)
    if Ls.size == 0:
        Ls = np.zeros(1, dtype=np.int64)
    Si = np.ascontiguousarray(Si, dtype=np.int64)
    if Si.size == 0:
        Si = np.zeros(1, dtype=np.int64)
    st = lib().oracle_super_numeric(E, int(sym["n"]), int(sym["nsuper"]),
                                    np.ascontiguousarray(sym["Super"], dtype=np.int64),
                                    np.ascontiguousarray(sym["Lpi"], dtype=np.int64),
                                    np.ascontiguousarray(sym["Lpx"], dtype=np.int64), Ls,
                                    int(sym["xsize"]), int(sym["maxcsize"]),
                                    np.ascontiguousarray(Sp, dtype=np.int64), Si, None, Sxf,
                                    float(beta), int(bool(quick_return)), Lx, minor)
    if st < 0:
        raise RuntimeError(f"oracle_super_numeric failed: {st}")
    return st, int(minor[0]), (Lx.view(np.complex128) if E == 2 else Lx)


def super_numeric_unsym(sym, Ap, Ai, Ax, Fp, Fi, Fx, beta=0.0, quick_return=False):
    """stype == 0: chol(A*F + beta*I), A n x ncol, F = A' (conjugate transpose) ncol x n, both packed CSC
    (cholmod_super_numeric(A, F, ...), worker :471-506).  Returns (status, minor, Lx)."""
    Axf, E = _flat(Ax)
    Fxf, E2 = _flat(Fx)
    assert E == E2
    pad = lambda a: a if a.size else np.zeros(2, dtype=a.dtype)
    Lx = np.zeros(max(int(sym["xsize"]), 1) * E)
    minor = np.zeros(1, dtype=np.int64)
    i64 = lambda a: pad(np.ascontiguousarray(a, dtype=np.int64))
    st = lib().oracle_super_numeric_f(E, int(sym["n"]), int(sym["nsuper"]), i64(sym["Super"]), i64(sym["Lpi"]),
                                      i64(sym["Lpx"]), i64(sym["Ls"]), int(sym["xsize"]), int(sym["maxcsize"]),
                                      i64(Ap), i64(Ai), None, pad(Axf), i64(Fp), i64(Fi), None, pad(Fxf),
                                      float(beta), int(bool(quick_return)), Lx, minor)
    if st < 0:
        raise RuntimeError(f"oracle_super_numeric_f failed: {st}")
    return st, int(minor[0]), (Lx.view(np.complex128) if E == 2 else Lx)


def super_solve(sym, Lx, X, lower=True):
    """X: n x nrhs array; returns L\\X (lower) or L^H\\X."""
    Lxf, E = _flat(Lx)
    X = np.asarray(X)
    if X.ndim == 1:
        X = X[:, None]
    n, nrhs = X.shape
    dt = np.complex128 if E == 2 else np.float64
    Xt = np.array(np.asarray(X, dtype=dt).T, order="C", copy=True)   # nrhs x n == col-major n x nrhs (own copy)
    flat = Xt.view(np.float64).reshape(-1)
    f = lib().oracle_super_lsolve if lower else lib().oracle_super_ltsolve
    Ls = np.ascontiguousarray(sym["Ls"], dtype=np.int64)
    if Ls.size == 0:
        Ls = np.zeros(1, dtype=np.int64)
    f(E, int(sym["nsuper"]), np.ascontiguousarray(sym["Super"], dtype=np.int64),
      np.ascontiguousarray(sym["Lpi"], dtype=np.int64),
      np.ascontiguousarray(sym["Lpx"], dtype=np.int64), Ls, Lxf, nrhs, flat, n)
    out = flat.view(dt).reshape(nrhs, n).T.copy()
    return out
