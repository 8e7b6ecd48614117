"""oracle/refcholmod.py -- TEST INFRASTRUCTURE, not product code.

ctypes binding of oracle/_ref/libref_helper.so (the unchanged reference CHOLMOD
5.3.0 CPU path built by oracle/build_ref.sh).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

ORDERINGS = {"natural": 0, "given": 1, "amd": 2, "metis": 3, "nesdis": 4, "default": -1}
ORDERING_NAMES = {0: "natural", 1: "given", 2: "amd", 3: "metis", 4: "nesdis", 5: "colamd", 6: "postordered"}


def available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_helper.so")) and \
        os.path.exists("/opt/conda/lib/libmkl_rt.so")


def lib():
    global _LIB
    if _LIB is None:
        # MKL's default threading layer is Intel OpenMP; CHOLMOD is built with libgomp
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_helper.so"), mode=C.RTLD_GLOBAL)
        _prototypes(L)
        _LIB = L
    return _LIB


_LIB_DROPIN = None


def dropin_available():
    return available() and os.path.exists(os.path.join(_HERE, "_ref", "libref_helper_mi355.so"))


def lib_dropin(engine_path):
    """The reference CHOLMOD with the dispatch of INTEGRATION.md compiled in (oracle/build_ref_mi355.sh).  Its mi355_*
    symbols are undefined: the engine library at `engine_path` is loaded first, globally.  The helper itself is loaded
    RTLD_LOCAL | RTLD_DEEPBIND (and libcholmod_mi355.so is linked -Bsymbolic), so that its cholmod_l_* calls reach the
    patched library even when the unchanged reference is loaded in the same process."""
    global _LIB_DROPIN
    if _LIB_DROPIN is None:
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        C.CDLL(engine_path, mode=C.RTLD_GLOBAL)
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_helper_mi355.so"), mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND | os.RTLD_NOW)
        _prototypes(L)
        L.engine_path = engine_path
        _LIB_DROPIN = L
    if _LIB_DROPIN.engine_path != engine_path:
        raise RuntimeError("the drop-in build is already bound to " + _LIB_DROPIN.engine_path)
    return _LIB_DROPIN


def _prototypes(L):
    L.ref_new.restype = C.c_void_p
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_nthreads.argtypes = [C.c_void_p]
    L.ref_set_nthreads.argtypes = [C.c_void_p, C.c_int]
    L.ref_load_matrix.argtypes = [C.c_void_p, C.c_int64, i64p, i64p, f64p, C.c_int, C.c_int]
    L.ref_read_matrix.argtypes = [C.c_void_p, C.c_char_p]
    L.ref_load_matrix_rect.argtypes = [C.c_void_p, C.c_int64, C.c_int64, i64p, i64p, f64p, C.c_int]
    L.ref_set_fset.argtypes = [C.c_void_p, C.c_int64]
    L.ref_F_info.argtypes = [C.c_void_p, i64p]
    L.ref_get_F.argtypes = [C.c_void_p, i64p, i64p, f64p]
    L.ref_matrix_info.argtypes = [C.c_void_p, i64p]
    L.ref_get_matrix.argtypes = [C.c_void_p, i64p, i64p, f64p]
    L.ref_analyze.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.ref_info.argtypes = [C.c_void_p, i64p, f64p]
    L.ref_get_symbolic.argtypes = [C.c_void_p, i64p, i64p, i64p, i64p, i64p, i64p]
    L.ref_permute.argtypes = [C.c_void_p]
    L.ref_permute.restype = C.c_int64
    L.ref_get_S.argtypes = [C.c_void_p, i64p, i64p, f64p]
    L.ref_factorize.argtypes = [C.c_void_p, C.c_double, C.c_int]
    L.ref_factorize_full.argtypes = [C.c_void_p, C.c_double]
    L.ref_get_Lx.argtypes = [C.c_void_p, f64p]
    L.ref_set_Lx.argtypes = [C.c_void_p, f64p, C.c_int64]
    L.ref_super_solve.argtypes = [C.c_void_p, C.c_int, C.c_int64, f64p, C.c_int64]
    L.ref_solve.argtypes = [C.c_void_p, C.c_int64, f64p, f64p]
    L.ref_set_use_gpu.argtypes = [C.c_void_p, C.c_int]
    L.ref_set_gpu_mem.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.ref_get_gpu_mem.argtypes = [C.c_void_p, f64p]
    L.ref_register_comm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.ref_fallbacks.argtypes = [C.c_char_p, C.c_int64]
    L.ref_fallbacks.restype = C.c_int64
    L.ref_counters.argtypes = [C.c_void_p, f64p]
    L.ref_drop_factor.argtypes = [C.c_void_p]
    L.ref_make_simplicial.argtypes = [C.c_void_p, C.c_int]
    L.ref_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.ref_raw_factor.argtypes = [C.c_void_p, C.c_int, f64p]


class RefCholmod:
    """One cholmod_common + matrix + factor of the reference library."""

    def __init__(self, nthreads=None, dropin=None):
        """dropin: path of the engine library -> the build with the dispatch compiled in, Common->useGPU = 1"""
        self.L = lib_dropin(dropin) if dropin else lib()
        self.h = self.L.ref_new()
        if dropin:
            self.L.ref_set_use_gpu(self.h, 1)
        if nthreads is not None:
            self.L.ref_set_nthreads(self.h, int(nthreads))
        self.n = 0
        self.xtype = 1

    def close(self):
        if self.h:
            self.L.ref_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def nthreads(self):
        return self.L.ref_nthreads(self.h)

    # ---- drop-in build ----
    def set_use_gpu(self, use):
        self.L.ref_set_use_gpu(self.h, int(use))

    def set_gpu_mem(self, max_bytes=0, max_fraction=0.0):
        """Common->maxGpuMemBytes / maxGpuMemFraction"""
        self.L.ref_set_gpu_mem(self.h, float(max_bytes), float(max_fraction))

    def gpu_mem(self):
        out = np.zeros(2)
        self.L.ref_get_gpu_mem(self.h, out)
        return float(out[0]), float(out[1])

    def register_comm(self, comm, rank, world):
        """cholmod_mi355_register_comm: comm = the mi355_comm* of this rank (engine.Comm._h), None to remove it"""
        return int(self.L.ref_register_comm(self.h, comm, int(rank), int(world)))

    def fallbacks(self):
        """(count, reason of the last one) of the drop-in's soft fallbacks to the CPU worker"""
        buf = C.create_string_buffer(512)
        k = int(self.L.ref_fallbacks(buf, 512))
        return k, buf.value.decode(errors="replace")

    def counters(self):
        out = np.zeros(10)
        self.L.ref_counters(self.h, out)
        keys = ("useGPU", "L_useGPU", "gpu_gemm_calls", "gpu_potrf_calls", "gpu_gemm_time", "gpu_potrf_time",
                "assemble_time", "cpu_blas_calls", "cpu_blas_time")
        return dict(zip(keys, out))

    def live_handles(self):
        return int(self.L.ref_live_handles())

    def calls_taken(self):
        """(numeric, single sweeps, whole solve2 branches, numeric calls that got the unpermuted A) the drop-in took
        over since the library was loaded"""
        out = np.zeros(4, dtype=np.int64)
        self.L.ref_calls_taken.argtypes = [i64p]
        self.L.ref_calls_taken(out)
        return tuple(int(v) for v in out)

    SYS = {"A": 0, "LDLt": 1, "LD": 2, "DLt": 3, "L": 4, "Lt": 5, "D": 6, "P": 7, "Pt": 8}

    def solve2(self, sys, B, ldb=None, zomplex_b=False, prefer_zomplex=False, reps=1):
        """cholmod_l_solve2 (sys, L, B) with reused X / Y / E workspaces; B is n x nrhs, real or complex (any
        combination with L's xtype: cholmod_solve.c:228-235,668); ldb > n pads the columns of B; zomplex_b hands a
        complex B over as separate real and imaginary arrays.  Returns X (n x nrhs)."""
        B = np.asarray(B)
        if B.ndim == 1:
            B = B[:, None]
        n, nrhs = B.shape
        ldb = int(ldb or n)
        cplx = np.iscomplexobj(B)
        pad = np.full((ldb, nrhs), 7.5, dtype=B.dtype, order="F")      # the padding rows must not be read
        pad[:n] = B
        Bz = None
        if cplx and zomplex_b:
            Bx = np.ascontiguousarray(pad.real.T).reshape(-1)
            Bz = np.ascontiguousarray(pad.imag.T).reshape(-1)
            bxtype = 3
        elif cplx:
            Bx = np.ascontiguousarray(pad.T).view(np.float64).reshape(-1)
            bxtype = 2
        else:
            Bx = np.ascontiguousarray(pad.T, dtype=np.float64).reshape(-1)
            bxtype = 1
        out = np.zeros(2 * n * nrhs)
        xx = C.c_int(0)
        f = self.L.ref_solve2
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, f64p, C.c_void_p, C.c_int64, C.c_int, C.c_int, f64p,
                      C.POINTER(C.c_int)]
        rc = f(self.h, self.SYS[sys] if isinstance(sys, str) else int(sys), bxtype, nrhs, Bx,
               Bz.ctypes.data_as(C.c_void_p) if Bz is not None else None, ldb, int(bool(prefer_zomplex)), int(reps), out,
               C.byref(xx))
        if rc != 0:
            raise RuntimeError("ref_solve2 failed")
        if xx.value == 1:
            return out[:n * nrhs].reshape(nrhs, n).T.copy()
        if xx.value == 2:
            return out.view(np.complex128).reshape(nrhs, n).T.copy()
        return (out[:n * nrhs] + 1j * out[n * nrhs:]).reshape(nrhs, n).T.copy()

    def forget(self):
        """cholmod_mi355_forget (L): drop the device image of the factor (drop-in build only)"""
        self.L.ref_forget.argtypes = [C.c_void_p]
        self.L.ref_forget(self.h)

    def drop_factor(self):
        self.L.ref_drop_factor(self.h)

    def make_simplicial(self, to_ll=True):
        return int(self.L.ref_make_simplicial(self.h, 1 if to_ll else 0))

    def set_option(self, **kw):
        """Common-> options the reference's test_solver varies: final_asis, final_super, final_ll, final_pack,
        final_monotonic, final_resymbol, nrelax0..2, zrelax0..2, grow0..2, nthreads_max"""
        for k, v in kw.items():
            if self.L.ref_set_option(self.h, k.encode(), float(v)) != 0:
                raise KeyError(k)

    def raw_factor(self, check_errors=True):
        """the expert sequence of CHOLMOD/Tcov/t_raw_factor.c on the loaded matrix (natural ordering); the factor
        becomes this object's factor"""
        out = np.zeros(6)
        if self.L.ref_raw_factor(self.h, int(bool(check_errors)), out) != 0:
            raise RuntimeError("ref_raw_factor: no symmetric matrix loaded")
        return dict(resid=float(out[0]), refused=int(out[1]), expected=int(out[2]), status=int(out[3]),
                    minor=int(out[4]), ok=bool(out[5]))

    # ---- matrix ----
    def load(self, n, Ap, Ai, Ax, stype=-1):
        Ap = np.ascontiguousarray(Ap, dtype=np.int64)
        Ai = np.ascontiguousarray(Ai, dtype=np.int64)
        cplx = np.iscomplexobj(Ax)
        Axf = np.ascontiguousarray(Ax, dtype=np.complex128).view(np.float64) if cplx \
            else np.ascontiguousarray(Ax, dtype=np.float64)
        if Axf.size == 0:
            Axf = np.zeros(2)
        rc = self.L.ref_load_matrix(self.h, n, Ap, Ai, Axf, 2 if cplx else 1, stype)
        if rc != 0:
            raise RuntimeError("ref_load_matrix failed")
        self.n = n
        self.xtype = 2 if cplx else 1

    def load_rect(self, nrow, ncol, Ap, Ai, Ax):
        """unsymmetric nrow x ncol matrix (stype 0): analyze/factorize then work on A*A'"""
        Ap = np.ascontiguousarray(Ap, dtype=np.int64)
        Ai = np.ascontiguousarray(Ai, dtype=np.int64)
        cplx = np.iscomplexobj(Ax)
        Axf = np.ascontiguousarray(Ax, dtype=np.complex128).view(np.float64) if cplx \
            else np.ascontiguousarray(Ax, dtype=np.float64)
        if Axf.size == 0:
            Axf = np.zeros(2)
        if Ai.size == 0:
            Ai = np.zeros(1, dtype=np.int64)
        if self.L.ref_load_matrix_rect(self.h, nrow, ncol, Ap, Ai, Axf, 2 if cplx else 1) != 0:
            raise RuntimeError("ref_load_matrix_rect failed")
        self.n = nrow
        self.xtype = 2 if cplx else 1

    def set_fset(self, fset=None):
        """column subset of an unsymmetric matrix: A(:,f)*A(:,f)' is analyzed and factorized (None: all columns)"""
        if fset is None:
            self.L.ref_set_fset(None, 0)
        else:
            f = np.ascontiguousarray(fset, dtype=np.int64)
            self.L.ref_set_fset(f.ctypes.data_as(C.c_void_p), f.size)

    def read(self, path):
        rc = self.L.ref_read_matrix(self.h, path.encode())
        if rc != 0:
            raise RuntimeError(f"ref_read_matrix({path}) -> {rc}")
        info = np.zeros(8, dtype=np.int64)
        self.L.ref_matrix_info(self.h, info)
        self.n = int(info[0])
        self.xtype = int(info[2])
        return self.matrix()

    def matrix(self):
        info = np.zeros(8, dtype=np.int64)
        self.L.ref_matrix_info(self.h, info)
        n, nnz, xtype, stype, ncol = (int(v) for v in info[:5])
        Ap = np.zeros(ncol + 1, dtype=np.int64)
        Ai = np.zeros(max(nnz, 1), dtype=np.int64)
        Ax = np.zeros(max(nnz, 1) * (2 if xtype == 2 else 1))
        self.L.ref_get_matrix(self.h, Ap, Ai, Ax)
        Ai = Ai[:nnz]
        Ax = Ax[:nnz * (2 if xtype == 2 else 1)]
        if xtype == 2:
            Ax = Ax.view(np.complex128)
        return dict(n=n, ncol=ncol, Ap=Ap, Ai=Ai, Ax=Ax, stype=stype, xtype=xtype)

    # ---- analyze ----
    def analyze(self, ordering="metis", perm=None, supernodal=2):
        code = ORDERINGS[ordering] if isinstance(ordering, str) else int(ordering)
        p = None
        if perm is not None:
            self._perm = np.ascontiguousarray(perm, dtype=np.int64)
            p = self._perm.ctypes.data_as(C.c_void_p)
            code = 1
        rc = self.L.ref_analyze(self.h, code, p, supernodal)
        if rc < 0:
            raise RuntimeError(f"ref_analyze failed: {rc}")
        return self.info()

    def info(self):
        io = np.zeros(16, dtype=np.int64)
        do = np.zeros(8)
        self.L.ref_info(self.h, io, do)
        keys = ["n", "nsuper", "ssize", "xsize", "maxcsize", "maxesize", "ordering", "is_super",
                "xtype", "minor", "is_ll", "status", "nthreads"]
        d = {k: int(v) for k, v in zip(keys, io)}
        d.update(fl=do[0], lnz=do[1], anz=do[2], t_analyze=do[3], t_factor=do[4], t_solve=do[5])
        d["ordering_name"] = ORDERING_NAMES.get(d["ordering"], "?")
        return d

    def symbolic(self):
        inf = self.info()
        n, ns, ss = inf["n"], inf["nsuper"], inf["ssize"]
        Perm = np.zeros(n, dtype=np.int64)
        ColCount = np.zeros(n, dtype=np.int64)
        Super = np.zeros(ns + 1, dtype=np.int64)
        Lpi = np.zeros(ns + 1, dtype=np.int64)
        Lpx = np.zeros(ns + 1, dtype=np.int64)
        Ls = np.zeros(max(ss, 1), dtype=np.int64)
        self.L.ref_get_symbolic(self.h, Perm, ColCount, Super, Lpi, Lpx, Ls)
        return dict(n=n, nsuper=ns, ssize=ss, xsize=inf["xsize"], maxcsize=inf["maxcsize"],
                    maxesize=inf["maxesize"], Perm=Perm, ColCount=ColCount, Super=Super,
                    Lpi=Lpi, Lpx=Lpx, Ls=Ls[:ss])

    def permuted_unsym(self):
        """stype == 0: (S, F) = (A(p,:), A(p,:)') exactly as cholmod_factorize hands them to super_numeric
        (cholmod_factorize.c:201-210,240-250).  Returns ((Sp, Si, Sx), (Fp, Fi, Fx)); S is n x ncol, F ncol x n."""
        S = self.permuted()
        info = np.zeros(8, dtype=np.int64)
        self.L.ref_F_info(self.h, info)
        frow, fcol, fnz = (int(v) for v in info[:3])
        assert fcol == self.n
        E = 2 if self.xtype == 2 else 1
        Fp = np.zeros(fcol + 1, dtype=np.int64)
        Fi = np.zeros(max(fnz, 1), dtype=np.int64)
        Fx = np.zeros(max(fnz, 1) * E)
        self.L.ref_get_F(self.h, Fp, Fi, Fx)
        Fi, Fx = Fi[:fnz], Fx[:fnz * E]
        return S, (Fp, Fi, Fx.view(np.complex128) if E == 2 else Fx)

    def permuted(self):
        """S = tril(P A P') exactly as cholmod_factorize builds it."""
        nnz = self.L.ref_permute(self.h)
        if nnz < 0:
            raise RuntimeError("ref_permute failed")
        E = 2 if self.xtype == 2 else 1
        info = np.zeros(8, dtype=np.int64)
        self.L.ref_matrix_info(self.h, info)
        Sp = np.zeros((int(info[4]) if int(info[3]) == 0 else self.n) + 1, dtype=np.int64)
        Si = np.zeros(max(nnz, 1), dtype=np.int64)
        Sx = np.zeros(max(nnz, 1) * E)
        self.L.ref_get_S(self.h, Sp, Si, Sx)
        Si = Si[:nnz]
        Sx = Sx[:nnz * E]
        if E == 2:
            Sx = Sx.view(np.complex128)
        return Sp, Si, Sx

    # ---- numeric ----
    def factorize(self, beta=0.0, quick_return=False):
        """reference cholmod_l_super_numeric on S; returns Common->status."""
        return self.L.ref_factorize(self.h, float(beta), int(bool(quick_return)))

    def factorize_full(self, beta=0.0):
        return self.L.ref_factorize_full(self.h, float(beta))

    def Lx(self):
        inf = self.info()
        E = 2 if inf["xtype"] == 2 else 1
        out = np.zeros(inf["xsize"] * E)
        self.L.ref_get_Lx(self.h, out)
        return out.view(np.complex128) if E == 2 else out

    def set_Lx(self, Lx, minor=None):
        inf = self.info()
        Lxf = np.ascontiguousarray(Lx)
        if np.iscomplexobj(Lxf):
            Lxf = Lxf.astype(np.complex128).view(np.float64)
        Lxf = np.ascontiguousarray(Lxf, dtype=np.float64)
        rc = self.L.ref_set_Lx(self.h, Lxf, inf["n"] if minor is None else int(minor))
        if rc != 0:
            raise RuntimeError("ref_set_Lx failed")

    def _as_f64(self, X):
        inf = self.info()
        if inf["xtype"] == 2:
            Xc = np.asfortranarray(X, dtype=np.complex128)
            return Xc, np.ascontiguousarray(Xc.T).view(np.float64).reshape(-1)
        Xr = np.asfortranarray(X, dtype=np.float64)
        return Xr, np.ascontiguousarray(Xr.T).reshape(-1)

    def super_solve(self, X, lower=True):
        """L\\X (lower) or L^H\\X; X is n x nrhs; returns the solution (same shape)."""
        X = np.atleast_2d(np.asarray(X).T).T if np.asarray(X).ndim == 1 else np.asarray(X)
        n, nrhs = X.shape
        _, flat = self._as_f64(X)
        flat = flat.copy()
        rc = self.L.ref_super_solve(self.h, 1 if lower else 0, nrhs, flat, n)
        if rc != 0:
            raise RuntimeError("ref_super_solve failed")
        if self.info()["xtype"] == 2:
            return flat.view(np.complex128).reshape(nrhs, n).T.copy()
        return flat.reshape(nrhs, n).T.copy()

    def to_simplicial(self, to_ll=True):
        """(Lp, Li, Lnz, Lx) after cholmod_l_change_factor (simplicial, packed, monotonic) on a copy of L"""
        f = self.L.ref_to_simplicial
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        lnz = C.c_int64(0)
        if f(self.h, int(bool(to_ll)), None, None, None, None, C.byref(lnz)) != 0:
            raise RuntimeError("ref_to_simplicial failed")
        inf = self.info()
        n, E = int(inf["n"]), (2 if inf["xtype"] == 2 else 1)
        Lp = np.zeros(n + 1, dtype=np.int64); Li = np.zeros(max(lnz.value, 1), dtype=np.int64)
        Lnz = np.zeros(max(n, 1), dtype=np.int64); Lx = np.zeros(max(lnz.value, 1) * E)
        if f(self.h, int(bool(to_ll)), Lp.ctypes.data_as(C.c_void_p), Li.ctypes.data_as(C.c_void_p),
             Lnz.ctypes.data_as(C.c_void_p), Lx.ctypes.data_as(C.c_void_p), C.byref(lnz)) != 0:
            raise RuntimeError("ref_to_simplicial failed")
        Lx = Lx[:lnz.value * E]
        return Lp, Li[:lnz.value], Lnz[:n], (Lx.view(np.complex128) if E == 2 else Lx)

    def rcond(self):
        self.L.ref_rcond.restype = C.c_double
        self.L.ref_rcond.argtypes = [C.c_void_p]
        return float(self.L.ref_rcond(self.h))

    def solve(self, B):
        B = np.asarray(B)
        if B.ndim == 1:
            B = B[:, None]
        n, nrhs = B.shape
        _, flat = self._as_f64(B)
        out = np.zeros_like(flat)
        rc = self.L.ref_solve(self.h, nrhs, flat, out)
        if rc != 0:
            raise RuntimeError("ref_solve failed")
        if self.info()["xtype"] == 2:
            return out.view(np.complex128).reshape(nrhs, n).T.copy()
        return out.reshape(nrhs, n).T.copy()
