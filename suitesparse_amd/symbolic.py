"""ctypes binding of include/mi355_symbolic.h: host symbolic analysis (etree, postorder,
column counts, relaxed supernodes -- restating cholmod_analyze / cholmod_super_symbolic2 for
a given permutation) and the geometric nested-dissection ordering for grid problems."""
import ctypes as C

import numpy as np

from . import engine

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ready = False

EXPORTS = ["mi355_nd_grid_order", "mi355_symbolic_analyze", "mi355_symbolic_free", "mi355_symbolic_info",
           "mi355_symbolic_get", "mi355_symbolic_permute"]


def _lib():
    global _ready
    L = engine.lib()
    if not _ready:
        L.mi355_nd_grid_order.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i64p]
        L.mi355_symbolic_analyze.argtypes = [C.c_int64, _i64p, _i64p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_void_p)]
        L.mi355_symbolic_free.argtypes = [C.c_void_p]
        L.mi355_symbolic_info.argtypes = [C.c_void_p, _i64p, _f64p]
        L.mi355_symbolic_get.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.mi355_symbolic_permute.argtypes = [C.c_void_p, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p]
        _ready = True
    return L


def nd_grid_order(shape, dof=1):
    """Nested-dissection permutation of a regular grid (shape = (nx, ny[, nz]), x fastest)."""
    nx, ny = int(shape[0]), int(shape[1])
    nz = int(shape[2]) if len(shape) > 2 else 1
    perm = np.zeros(nx * ny * nz * dof, dtype=np.int64)
    rc = _lib().mi355_nd_grid_order(nx, ny, nz, int(dof), perm)
    if rc != 0:
        raise engine.Mi355Error(rc, "mi355_nd_grid_order failed")
    return perm


def analyze(n, Ap, Ai, perm=None, nrelax=None, zrelax=None):
    """Symbolic analysis of the lower-stored symmetric pattern (Ap, Ai) under `perm`.
    Returns the dict every other layer uses (same keys as cholmod_factor's fields)."""
    L = _lib()
    Ap = np.ascontiguousarray(Ap, dtype=np.int64)
    Ai = np.ascontiguousarray(Ai, dtype=np.int64)
    if Ai.size == 0:
        Ai = np.zeros(1, dtype=np.int64)
    keep = []

    def opt(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)
    h = C.c_void_p()
    rc = L.mi355_symbolic_analyze(int(n), Ap, Ai, opt(perm, np.int64), opt(nrelax, np.int64),
                                  opt(zrelax, np.float64), C.byref(h))
    if rc != 0:
        raise engine.Mi355Error(rc, "mi355_symbolic_analyze failed (invalid matrix or permutation)")
    try:
        io = np.zeros(8, dtype=np.int64)
        do = np.zeros(2)
        L.mi355_symbolic_info(h, io, do)
        n_, ns, ss, xs, mc, me, snnz = (int(v) for v in io[:7])
        out = dict(n=n_, nsuper=ns, ssize=ss, xsize=xs, maxcsize=mc, maxesize=me, snnz=snnz,
                   fl=float(do[0]), lnz=float(do[1]),
                   Perm=np.zeros(n_, dtype=np.int64), ColCount=np.zeros(n_, dtype=np.int64),
                   Parent=np.zeros(n_, dtype=np.int64), Super=np.zeros(ns + 1, dtype=np.int64),
                   Lpi=np.zeros(ns + 1, dtype=np.int64), Lpx=np.zeros(ns + 1, dtype=np.int64),
                   Ls=np.zeros(ss, dtype=np.int64))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        L.mi355_symbolic_get(h, ptr(out["Perm"]), ptr(out["ColCount"]), ptr(out["Parent"]), ptr(out["Super"]),
                             ptr(out["Lpi"]), ptr(out["Lpx"]), ptr(out["Ls"]))
        out["_handle_perm"] = None
    finally:
        L.mi355_symbolic_free(h)
    return out


def permute(sym, Ap, Ai, Ax):
    """S = tril(P A P') with sorted columns (what cholmod_factorize hands to cholmod_super_numeric)."""
    import scipy.sparse as sp
    n = sym["n"]
    Perm = sym["Perm"]
    pinv = np.empty(n, dtype=np.int64)
    pinv[Perm] = np.arange(n, dtype=np.int64)
    Ap = np.asarray(Ap)
    cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap))
    a, b = pinv[np.asarray(Ai)], pinv[cols]
    r, c = np.maximum(a, b), np.minimum(a, b)
    v = np.asarray(Ax)
    if np.iscomplexobj(v):
        v = np.where(a < b, v.conj(), v)        # A(i,j) moves to the upper part: its lower mirror is the conjugate
    S = sp.csc_matrix((v, (r, c)), shape=(n, n))
    S.sort_indices()
    return S.indptr.astype(np.int64), S.indices.astype(np.int64), np.ascontiguousarray(S.data)
