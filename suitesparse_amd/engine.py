"""ctypes binding of the C ABI in include/mi355_cholmod.h (libmi355_cholmod.so).

The names mirror the reference's Supernodal module so that tests read like the
reference's own (CHOLMOD/Tcov/t_raw_factor.c:186-280):

    L = SupernodalFactor(sym)            # a supernodal symbolic cholmod_factor
    status = L.super_numeric(S, beta)    # cholmod_super_numeric (cholmod_super_numeric.c:100)
    L.super_lsolve(X); L.super_ltsolve(X)   # cholmod_super_lsolve / _ltsolve (cholmod_super_solve.c:52,153)

There is no CPU path: if the shared library or a gfx950 device is missing, every
call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355_cholmod.so")

CHOLMOD_OK = 0
CHOLMOD_NOT_POSDEF = 1
CHOLMOD_INVALID = -4
CHOLMOD_GPU_PROBLEM = -5
CHOLMOD_NOT_INSTALLED = -1

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

EXPORTS = [
    "mi355_probe", "mi355_last_error", "mi355_factor_create", "mi355_factor_free",
    "mi355_factor_set_matrix", "mi355_factor_numeric", "mi355_factor_get_Lx",
    "mi355_factor_numeric_host", "mi355_factor_host_copy_plan", "mi355_factor_rcond_noalloc",
    "mi355_super_numeric", "mi355_super_lsolve", "mi355_super_ltsolve",
    "mi355_super_lsolve_dev", "mi355_super_ltsolve_dev", "mi355_factor_device_Lx",
    "mi355_factor_set_profile", "mi355_factor_get_stats", "mi355_debug_mfma_f64_rate",
    "mi355_factor_set_partition", "mi355_factor_numeric_phase",
    "mi355_super_solve_phase_dev", "mi355_solve_keep_rows_dev",
    "mi355_factor_set_matrix_perm", "mi355_factor_rcond", "mi355_factor_to_simplicial",
    "mi355_factor_set_top_distribution", "mi355_factor_top_plan", "mi355_factor_top_run", "mi355_factor_device_Dinv",
    "mi355_debug_gemm_rate", "mi355_set_tuning", "mi355_factor_set_matrix_zomplex",
    "mi355_factor_set_matrix_unsym", "mi355_factor_set_perm", "mi355_solve", "mi355_solve_dev", "mi355_solve_x", "mi355_solve_x_dev",
    "mi355_partition", "mi355_factor_partition", "mi355_factor_partition_info", "mi355_comm_create", "mi355_rccl_unique_id",
    "mi355_comm_create_rccl", "mi355_comm_adopt_rccl", "mi355_comm_free", "mi355_factor_numeric_multi",
    "mi355_super_solve_multi_dev", "mi355_factor_gather_multi", "mi355_factor_reduce_plan",
    "mi355_factor_create_limited", "mi355_device_memory", "mi355_current_device",
]

# mi355_transport (include/mi355_cholmod.h): blocking collectives of the caller's
_RED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
_BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
_AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
_AMIN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64))


class _Transport(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("ctx", C.c_void_p), ("reduce_sum", _RED), ("bcast", _BC),
                ("allreduce_sum", _AR), ("allreduce_min_i64", _AMIN), ("flags", C.c_int)]

_LIB = None


class Mi355Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mi355 error {code}: {msg}")
        self.code = code


def lib():
    """Load libmi355_cholmod.so (fails loudly if it was not built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355Error(CHOLMOD_NOT_INSTALLED,
                             f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.mi355_probe.restype = C.c_int
        L.mi355_last_error.restype = C.c_char_p
        L.mi355_factor_create.argtypes = [C.c_int, C.c_int64, C.c_int64, _i64p, _i64p, _i64p, _i64p,
                                          C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        L.mi355_factor_create_limited.argtypes = [C.c_int, C.c_int64, C.c_int64, _i64p, _i64p, _i64p, _i64p,
                                                  C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_double,
                                                  C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]
        L.mi355_device_memory.argtypes = [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mi355_current_device.argtypes = [C.POINTER(C.c_int)]
        L.mi355_factor_free.argtypes = [C.c_void_p]
        L.mi355_factor_set_matrix.argtypes = [C.c_void_p, _i64p, _i64p, C.c_void_p, _f64p]
        L.mi355_factor_set_matrix_zomplex.argtypes = [C.c_void_p, _i64p, _i64p, C.c_void_p, _f64p, _f64p]
        L.mi355_factor_set_matrix_unsym.argtypes = [C.c_void_p, C.c_int64, _i64p, _i64p, C.c_void_p, _f64p,
                                                    C.c_void_p, _i64p, _i64p, C.c_void_p, _f64p, C.c_void_p]
        L.mi355_factor_numeric.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int)]
        L.mi355_factor_get_Lx.argtypes = [C.c_void_p, _f64p]
        L.mi355_super_numeric.argtypes = [C.c_void_p, _i64p, _i64p, C.c_void_p, _f64p, C.c_double, C.c_int,
                                          _f64p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        for f in (L.mi355_super_lsolve, L.mi355_super_ltsolve):
            f.argtypes = [C.c_void_p, C.c_int64, _f64p, C.c_int64]
        for f in (L.mi355_super_lsolve_dev, L.mi355_super_ltsolve_dev):
            f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.mi355_factor_set_perm.argtypes = [C.c_void_p, C.c_void_p]
        L.mi355_solve.argtypes = [C.c_void_p, C.c_int, C.c_int64, _f64p, C.c_int64, _f64p, C.c_int64]
        L.mi355_solve_x.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, _f64p, C.c_int64, _f64p, C.c_int64]
        L.mi355_solve_x_dev.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int64]
        L.mi355_solve_dev.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.mi355_factor_device_Lx.argtypes = [C.c_void_p]
        L.mi355_factor_device_Lx.restype = C.c_void_p
        L.mi355_factor_set_profile.argtypes = [C.c_void_p, C.c_int]
        L.mi355_factor_get_stats.argtypes = [C.c_void_p, _f64p]
        _i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
        L.mi355_factor_set_partition.argtypes = [C.c_void_p, _i8p, _i8p, _i8p]
        L.mi355_factor_numeric_phase.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_int64),
                                                 C.POINTER(C.c_int64)]
        L.mi355_super_solve_phase_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int64]
        L.mi355_solve_keep_rows_dev.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.mi355_factor_set_matrix_perm.argtypes = [C.c_void_p, _i64p, _i64p, C.c_void_p, _f64p, C.c_void_p]
        L.mi355_factor_set_top_distribution.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mi355_factor_top_plan.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mi355_factor_top_run.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mi355_factor_device_Dinv.argtypes = [C.c_void_p]
        L.mi355_factor_device_Dinv.restype = C.c_void_p
        _i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.mi355_partition.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, _i64p, C.c_int, C.c_int, _i32p, C.c_void_p]
        L.mi355_factor_partition.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mi355_factor_partition_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi355_comm_create.argtypes = [C.POINTER(_Transport), C.POINTER(C.c_void_p)]
        L.mi355_rccl_unique_id.argtypes = [C.c_char_p]
        L.mi355_comm_create_rccl.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mi355_comm_adopt_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mi355_comm_free.argtypes = [C.c_void_p]
        L.mi355_factor_numeric_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_int64),
                                                 C.POINTER(C.c_int)]
        L.mi355_super_solve_multi_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64]
        L.mi355_factor_gather_multi.argtypes = [C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def last_error():
    return lib().mi355_last_error().decode(errors="replace")


def _check(rc):
    if rc != 0:
        raise Mi355Error(rc, last_error())


def probe():
    return lib().mi355_probe()


def mfma_f64_rate(device=0):
    """measured TFLOP/s of bare v_mfma_f64_16x16x4_f64 issue on `device`"""
    return mfma_f64_probe(device)[0]


def mfma_f64_probe(device=0):
    """(TFLOP/s, shader cycles per MFMA for one of 2 waves on a SIMD, shader clock MHz)"""
    out = np.zeros(3)
    lib().mi355_debug_mfma_f64_rate.argtypes = [C.c_int, _f64p]
    _check(lib().mi355_debug_mfma_f64_rate(int(device), out))
    return tuple(out.tolist())


def gemm_rate(kernel, M, N, K, iters=10, device=0):
    """TFLOP/s of one lower-trapezoid trailing-update task on kernel 0 (64x64 tile) / 1 (128x128)"""
    out = C.c_double(0.0)
    f = lib().mi355_debug_gemm_rate
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    _check(f(int(device), int(kernel), int(M), int(N), int(K), int(iters), C.byref(out)))
    return out.value


def set_tuning(key, value):
    _check(lib().mi355_set_tuning(int(key), int(value)))


def partition(sym, world, min_subtrees_per_rank=2):
    """mi355_partition: (owner[nsuper] int32 (-1 = top), top flops, total flops, per-rank loads).  Host work only."""
    ns = int(sym["nsuper"])

    def arr(a):
        a = np.ascontiguousarray(a, dtype=np.int64)
        return a if a.size else np.zeros(1, dtype=np.int64)
    owner = np.zeros(max(ns, 1), dtype=np.int32)
    stats = np.zeros(2 + int(world))
    _check(lib().mi355_partition(int(sym["n"]), ns, arr(sym["Super"]), arr(sym["Lpi"]), arr(sym["Ls"]), int(world),
                                 int(min_subtrees_per_rank), owner, stats.ctypes.data_as(C.c_void_p)))
    return owner[:ns], float(stats[0]), float(stats[1]), stats[2:].tolist()


class Comm:
    """mi355_comm: the collectives of mi355_factor_numeric_multi / mi355_super_solve_multi_dev."""

    def __init__(self, handle, rank, world, keep=None):
        self._h, self.rank, self.world, self._keep = handle, rank, world, keep

    @classmethod
    def custom(cls, rank, world, reduce_sum, bcast, allreduce_sum, allreduce_min, defer=False):
        """blocking collectives of the caller's: reduce_sum(ptr, count, root), bcast(ptr, count, root),
        allreduce_sum(ptr, count) on DEVICE pointers (counts in doubles), allreduce_min(value) -> value on the host;
        defer: MI355_TRANSPORT_DEFER (collectives made as late as the engine's ordering allows; a testing aid)"""
        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except Exception:          # an exception cannot cross the C frames
                    import traceback
                    traceback.print_exc()
                    return -1
            return run

        def amin(_, p):
            p[0] = int(allreduce_min(int(p[0])))
        t = _Transport(int(rank), int(world), None,
                       _RED(guard(lambda _, p, n, r: reduce_sum(p, n, r))), _BC(guard(lambda _, p, n, r: bcast(p, n, r))),
                       _AR(guard(lambda _, p, n: allreduce_sum(p, n))), _AMIN(guard(amin)), 1 if defer else 0)
        h = C.c_void_p()
        _check(lib().mi355_comm_create(C.byref(t), C.byref(h)))
        return cls(h, int(rank), int(world), keep=t)

    @classmethod
    def rccl(cls, rank, world, device, exchange):
        """RCCL communicator of its own: exchange(id_or_None) hands rank 0's 128-byte id to every rank (any side
        channel: MPI_Bcast, a file, torch.distributed) and returns it"""
        # rank 0 ALWAYS takes part in the exchange (a collective of the caller's): when it cannot make an id it sends the
        # error instead, and every rank raises after the exchange -- nobody is left waiting in it
        idb = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            rc = lib().mi355_rccl_unique_id(buf)
            idb = buf.raw if rc == 0 else ("error", int(rc), last_error())
        idb = exchange(idb)
        if isinstance(idb, tuple):
            raise Mi355Error(idb[1], "rank 0 could not create the RCCL id: " + idb[2])
        if not isinstance(idb, (bytes, bytearray)) or len(idb) != 128:
            raise Mi355Error(CHOLMOD_INVALID, "the RCCL id did not arrive")
        h = C.c_void_p()
        _check(lib().mi355_comm_create_rccl(idb, int(rank), int(world), int(device), C.byref(h)))
        return cls(h, int(rank), int(world))

    @classmethod
    def from_torch_distributed(cls, device):
        """the ranks of the initialised torch.distributed process group (backend nccl = RCCL): the id travels through
        it, the communicator is the engine's own"""
        import torch.distributed as dist

        def exchange(idb):
            box = [idb]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return cls.rccl(dist.get_rank(), dist.get_world_size(), device, exchange)

    def close(self):
        if self._h:
            lib().mi355_comm_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


STAT_KEYS = ["total_ms", "gemm_ms", "gemm_launches", "potrf_ms", "potrf_launches", "assembly_ms",
             "flops_update", "flops_panel", "bytes_assembly", "npairs", "nlevels", "relmap_entries",
             "dinv_entries", "solve_ms", "gemm_tiles", "copy_early_entries", "gemm128_ms", "gemm128_launches",
             "flops_gemm128", "flops_gemm64", "tiles_gemm128", "tiles_gemm64", "flops_gemm128_here", "reduce_overlap_launches"]


def device_memory(device=0):
    """(total, free) bytes of a device (cholmod_l_gpu_memorysize, CHOLMOD/GPU/cholmod_gpu.c:63-131)"""
    t, f = C.c_int64(0), C.c_int64(0)
    _check(lib().mi355_device_memory(int(device), C.byref(t), C.byref(f)))
    return int(t.value), int(f.value)


def current_device():
    d = C.c_int(0)
    _check(lib().mi355_current_device(C.byref(d)))
    return int(d.value)


class SupernodalFactor:
    """Device-resident supernodal factor (cholmod_factor with is_super, CHOLMOD/Include/cholmod.h:1584-1651)."""

    def __init__(self, sym, device=0, xtype=1, max_bytes=0, max_fraction=0.0):
        """xtype: 1 = real L, 2 = complex L (cholmod.h:134-137); max_bytes / max_fraction: cap on the device memory of the
        handle (Common->maxGpuMemBytes / maxGpuMemFraction, mi355_factor_create_limited); self.needed_bytes = what it takes"""
        self.E = 2 if xtype == 2 else 1
        self.n = int(sym["n"])
        self.nsuper = int(sym["nsuper"])
        self.xsize = int(sym["xsize"])
        self.ssize = int(sym.get("ssize", len(sym["Ls"])))
        self.minor = self.n
        self.status = CHOLMOD_OK
        self.xtype = xtype
        self._h = C.c_void_p()
        Ls = np.ascontiguousarray(sym["Ls"], dtype=np.int64)
        if Ls.size == 0:
            Ls = np.zeros(1, dtype=np.int64)

        def arr(a):
            a = np.ascontiguousarray(a, dtype=np.int64)
            return a if a.size else np.zeros(1, dtype=np.int64)
        need = C.c_int64(0)
        rc = lib().mi355_factor_create_limited(xtype, self.n, self.nsuper, arr(sym["Super"]), arr(sym["Lpi"]),
                                               arr(sym["Lpx"]), Ls, self.ssize, self.xsize, int(device),
                                               int(max_bytes), float(max_fraction), C.byref(need), C.byref(self._h))
        self.needed_bytes = int(need.value)
        _check(rc)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().mi355_factor_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- numeric ----
    def to_simplicial(self, to_ll=True):
        """(Lp, Li, Lnz, Lx) of the packed simplicial LL' (or LDL') form of the resident factor"""
        f = lib().mi355_factor_to_simplicial
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        lnz = C.c_int64(0)
        _check(f(self._h, int(bool(to_ll)), None, None, None, None, C.byref(lnz)))
        Lp = np.zeros(self.n + 1, dtype=np.int64)
        Li = np.zeros(max(lnz.value, 1), dtype=np.int64)
        Lnz = np.zeros(max(self.n, 1), dtype=np.int64)
        Lx = np.zeros(max(lnz.value, 1) * self.E)
        _check(f(self._h, int(bool(to_ll)), Lp.ctypes.data_as(C.c_void_p), Li.ctypes.data_as(C.c_void_p),
                 Lnz.ctypes.data_as(C.c_void_p), Lx.ctypes.data_as(C.c_void_p), C.byref(lnz)))
        Lx = Lx[:lnz.value * self.E]
        return Lp, Li[:lnz.value], Lnz[:self.n], (Lx.view(np.complex128) if self.E == 2 else Lx)

    def rcond(self):
        """cholmod_rcond of the resident factor"""
        out = C.c_double(0.0)
        f = lib().mi355_factor_rcond
        f.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        _check(f(self._h, C.byref(out)))
        return out.value

    def rcond_noalloc(self, out_dev_ptr):
        """mi355_factor_rcond_noalloc: the result goes to the caller's device double at out_dev_ptr"""
        f = lib().mi355_factor_rcond_noalloc
        f.argtypes = [C.c_void_p, C.c_void_p]
        _check(f(self._h, C.c_void_p(int(out_dev_ptr))))

    def set_matrix_perm(self, Ap, Ai, Ax, Perm=None, Anz=None):
        """stage A itself (lower part, stype < 0) + L->Perm: S = tril(P A P') is placed by the assembly kernel"""
        Ap = np.ascontiguousarray(Ap, dtype=np.int64)
        Ai = np.ascontiguousarray(Ai, dtype=np.int64)
        if self.E == 2:
            Ax = np.ascontiguousarray(Ax, dtype=np.complex128).view(np.float64)
        else:
            if np.iscomplexobj(Ax):
                raise Mi355Error(CHOLMOD_INVALID, "complex type mismatch: complex A with a real L")
            Ax = np.ascontiguousarray(Ax, dtype=np.float64)
        if Ai.size == 0:
            Ai = np.zeros(1, dtype=np.int64)
            Ax = np.zeros(2)
        nz = None
        if Anz is not None:
            self._nz = np.ascontiguousarray(Anz, dtype=np.int64)
            nz = self._nz.ctypes.data_as(C.c_void_p)
        pp = None
        if Perm is not None:
            self._perm_in = np.ascontiguousarray(Perm, dtype=np.int64)
            pp = self._perm_in.ctypes.data_as(C.c_void_p)
        _check(lib().mi355_factor_set_matrix_perm(self._h, Ap, Ai, nz, Ax, pp))

    def set_matrix(self, Sp, Si, Sx, Snz=None, Sz=None):
        """Sx: real array, or complex array (interleaved on the way in) for a complex L;
        Sz: imaginary parts of a zomplex matrix (then Sx holds the real parts)."""
        Sp = np.ascontiguousarray(Sp, dtype=np.int64)
        Si = np.ascontiguousarray(Si, dtype=np.int64)
        if self.E == 2 and Sz is None:
            Sx = np.ascontiguousarray(Sx, dtype=np.complex128).view(np.float64)
        else:
            if np.iscomplexobj(Sx):
                raise Mi355Error(CHOLMOD_INVALID, "complex type mismatch: complex A with a real L")
            Sx = np.ascontiguousarray(Sx, dtype=np.float64)
        if Si.size == 0:
            Si = np.zeros(1, dtype=np.int64)
            Sx = np.zeros(2)
        nz = None
        if Snz is not None:
            self._nz = np.ascontiguousarray(Snz, dtype=np.int64)
            nz = self._nz.ctypes.data_as(C.c_void_p)
        if Sz is not None:
            Sz = np.ascontiguousarray(Sz, dtype=np.float64)
            _check(lib().mi355_factor_set_matrix_zomplex(self._h, Sp, Si, nz, Sx, Sz))
        else:
            _check(lib().mi355_factor_set_matrix(self._h, Sp, Si, nz, Sx))

    def set_matrix_unsym(self, ncol, Ap, Ai, Ax, Fp, Fi, Fx, Anz=None, Fnz=None, Az=None, Fz=None):
        """stype == 0: the pair (A, F) of cholmod_super_numeric(A, F, beta, L); A is n x ncol, F = A' (conjugate
        transpose) is ncol x n.  Ax/Fx complex arrays for a complex L, or real parts with Az/Fz (zomplex)."""
        def vals(x, z):
            if self.E == 2 and z is None:
                return np.ascontiguousarray(x, dtype=np.complex128).view(np.float64)
            if np.iscomplexobj(x):
                raise Mi355Error(CHOLMOD_INVALID, "complex type mismatch: complex A with a real L")
            return np.ascontiguousarray(x, dtype=np.float64)

        def pad(a, k=1):
            return a if a.size else np.zeros(k, dtype=a.dtype)
        Ap, Ai, Fp, Fi = (np.ascontiguousarray(a, dtype=np.int64) for a in (Ap, Ai, Fp, Fi))
        keep = []

        def opt(a, dt):
            if a is None:
                return None
            keep.append(np.ascontiguousarray(a, dtype=dt))
            return keep[-1].ctypes.data_as(C.c_void_p)
        _check(lib().mi355_factor_set_matrix_unsym(self._h, int(ncol), Ap, pad(Ai), opt(Anz, np.int64),
                                                   pad(vals(Ax, Az), 2), opt(Az, np.float64), Fp, pad(Fi),
                                                   opt(Fnz, np.int64), pad(vals(Fx, Fz), 2), opt(Fz, np.float64)))

    def numeric(self, beta=0.0, quick_return_if_not_posdef=False):
        """Factorize the staged matrix; everything stays in HBM.  Returns CHOLMOD status."""
        minor = C.c_int64(0)
        status = C.c_int(0)
        _check(lib().mi355_factor_numeric(self._h, float(beta), int(bool(quick_return_if_not_posdef)),
                                          C.byref(minor), C.byref(status)))
        self.minor, self.status = int(minor.value), int(status.value)
        return self.status

    def numeric_host(self, beta=0.0, quick_return_if_not_posdef=False, out=None):
        """mi355_factor_numeric_host: factorize and fill a host L->x while the factorization runs; returns
        (status, Lx); out: a float64 array of at least E * xsize entries to fill instead of a new one"""
        minor = C.c_int64(0)
        status = C.c_int(0)
        if out is None:
            out = np.full(max(self.xsize, 1) * self.E, np.nan)
        f = lib().mi355_factor_numeric_host
        f.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        _check(f(self._h, float(beta), int(bool(quick_return_if_not_posdef)), out.ctypes.data_as(C.c_void_p),
                 C.byref(minor), C.byref(status)))
        self.minor, self.status = int(minor.value), int(status.value)
        out = out[:self.xsize * self.E]
        return self.status, (out.view(np.complex128) if self.E == 2 else out)

    def host_copy_plan(self):
        """rows (lo, hi, ready_step, flops executed before ready_step) of the chunks of Lx that
        mi355_factor_numeric_host copies early"""
        f = lib().mi355_factor_host_copy_plan
        f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
        cnt = C.c_int64(0)
        _check(f(self._h, 0, None, C.byref(cnt)))
        plan = np.zeros((cnt.value, 4), dtype=np.int64)
        if cnt.value:
            _check(f(self._h, cnt.value, plan.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return plan

    def super_numeric(self, Sp, Si, Sx, beta=(0.0, 0.0), quick_return_if_not_posdef=False, Snz=None, Sz=None,
                      F=None):
        """cholmod_super_numeric(S, F, beta, L): S = tril(P A P') and F = None (stype < 0), or S = A(p,f) with
        F = (Fp, Fi, Fx) = S' (stype == 0: A*F is factorized); only beta[0] is used
        (t_cholmod_super_numeric_worker.c:510-520).  Returns Common->status; sets self.minor."""
        b = beta[0] if isinstance(beta, (tuple, list, np.ndarray)) else beta
        if F is not None:
            self.set_matrix_unsym(len(Sp) - 1, Sp, Si, Sx, F[0], F[1], F[2], Anz=Snz)
            return self.numeric(b, quick_return_if_not_posdef)
        self.set_matrix(Sp, Si, Sx, Snz, Sz)
        return self.numeric(b, quick_return_if_not_posdef)

    def Lx(self, out=None):
        """L->x on the host; out: a float64 array of at least E * xsize entries to fill instead of a new one"""
        if out is None:
            out = np.zeros(max(self.xsize, 1) * self.E)
        _check(lib().mi355_factor_get_Lx(self._h, out))
        out = out[:self.xsize * self.E]
        return out.view(np.complex128) if self.E == 2 else out

    # ---- solves ----
    def _solve(self, fn, X):
        dt = np.complex128 if self.E == 2 else np.float64
        if self.E == 1 and np.iscomplexobj(X):
            raise Mi355Error(CHOLMOD_INVALID, "L and X must have the same xtype")
        X = np.asarray(X, dtype=dt)
        one_d = X.ndim == 1
        if one_d:
            X = X[:, None]
        n, nrhs = X.shape
        buf = np.array(X.T, dtype=dt, order="C", copy=True)     # column-major n x nrhs, ld n
        if buf.size == 0:
            return X.copy()
        _check(fn(self._h, nrhs, buf.view(np.float64).reshape(-1), n))
        out = buf.T.copy()
        return out[:, 0] if one_d else out

    def super_lsolve(self, X):
        """x = L \\ b (cholmod_super_lsolve)."""
        return self._solve(lib().mi355_super_lsolve, X)

    def super_ltsolve(self, X):
        """x = L' \\ b (cholmod_super_ltsolve)."""
        return self._solve(lib().mi355_super_ltsolve, X)

    def set_perm(self, Perm):
        """L->Perm for solve() (None: natural ordering)."""
        if Perm is None:
            _check(lib().mi355_factor_set_perm(self._h, None))
        else:
            P = np.ascontiguousarray(Perm, dtype=np.int64)
            _check(lib().mi355_factor_set_perm(self._h, P.ctypes.data_as(C.c_void_p)))

    def solve(self, B, sys=0):
        """cholmod_solve(sys, L, B) for the supernodal factor: sys = CHOLMOD_A (0) ... CHOLMOD_Pt (8);
        permutations and sweeps run on the device (cholmod_solve.c:656-717).  B may be real or complex
        whatever L is; X is complex if either is (real L + complex B: the reference's "dual" path)."""
        B = np.asarray(B)
        bc = np.iscomplexobj(B)
        xc = bc or self.E == 2
        B = B.astype(np.complex128 if bc else np.float64)
        one_d = B.ndim == 1
        if one_d:
            B = B[:, None]
        n, nrhs = B.shape
        bb = np.array(B.T, order="C", copy=True)
        xx = np.zeros((nrhs, n), dtype=np.complex128 if xc else np.float64)
        if bb.size:
            _check(lib().mi355_solve_x(self._h, int(sys), nrhs, 2 if bc else 1, bb.view(np.float64).reshape(-1), n,
                                       xx.view(np.float64).reshape(-1), n))
        out = xx.T.copy()
        return out[:, 0] if one_d else out

    def solve2_dev(self, sys, B_ptr, ldb, X_ptr, ldx, nrhs):
        _check(lib().mi355_solve_dev(self._h, int(sys), int(nrhs), C.c_void_p(int(B_ptr)), int(ldb),
                                     C.c_void_p(int(X_ptr)), int(ldx)))

    def solve_dev(self, X_ptr, nrhs, d, lower=True):
        fn = lib().mi355_super_lsolve_dev if lower else lib().mi355_super_ltsolve_dev
        _check(fn(self._h, int(nrhs), C.c_void_p(int(X_ptr)), int(d)))

    # ---- multi-GPU phases (see suitesparse_amd/multigpu.py) ----
    def set_partition(self, local_mask, top_mask, asm_mask):
        m = [np.ascontiguousarray(a, dtype=np.int8) for a in (local_mask, top_mask, asm_mask)]
        _check(lib().mi355_factor_set_partition(self._h, *m))

    def numeric_phase(self, phase, beta=0.0):
        """returns (first_bad_supernode or -1, info)"""
        bad = C.c_int64(-1)
        info = C.c_int64(0)
        _check(lib().mi355_factor_numeric_phase(self._h, int(phase), float(beta), C.byref(bad), C.byref(info)))
        return int(bad.value), int(info.value)

    def solve_phase_dev(self, part, lower, X_ptr, nrhs, d):
        """one sweep over the supernodes of part 0 (own subtrees) / 1 (top) of a partitioned factor"""
        _check(lib().mi355_super_solve_phase_dev(self._h, int(part), int(bool(lower)), int(nrhs),
                                                 C.c_void_p(int(X_ptr)), int(d)))

    def keep_rows_dev(self, X_ptr, nrhs, d):
        """zero the rows of X outside this rank's asm_mask (before a sum over the ranks)"""
        _check(lib().mi355_solve_keep_rows_dev(self._h, int(nrhs), C.c_void_p(int(X_ptr)), int(d)))

    def set_top_distribution(self, rank, world):
        """owner-computes top: returns (plan, (dinv_lo, dinv_hi)); plan rows = (step, lo, n, root)"""
        _check(lib().mi355_factor_set_top_distribution(self._h, int(rank), int(world)))
        cnt, lo, hi = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(lib().mi355_factor_top_plan(self._h, 0, None, C.byref(cnt), C.byref(lo), C.byref(hi)))
        plan = np.zeros((cnt.value, 4), dtype=np.int64)
        _check(lib().mi355_factor_top_plan(self._h, cnt.value, plan.ctypes.data_as(C.c_void_p), C.byref(cnt),
                                           C.byref(lo), C.byref(hi)))
        return plan, (int(lo.value), int(hi.value))

    def top_run(self, seg):
        """launches of the distributed top up to broadcast `seg`; (first_bad, info) after the last segment"""
        bad, info = C.c_int64(-1), C.c_int64(0)
        _check(lib().mi355_factor_top_run(self._h, int(seg), C.byref(bad), C.byref(info)))
        return int(bad.value), int(info.value)

    def dinv_tensors(self, ranges):
        return self.top_tensors(ranges, base=lib().mi355_factor_device_Dinv(self._h))

    def top_tensors(self, ranges, base=None):
        """torch views of Lx[lo:hi) for each (lo, hi) (device memory, no copy)"""
        import torch
        base = self.device_Lx_ptr() if base is None else base
        out = []
        for lo, hi in ranges:
            class _View:
                pass
            v = _View()
            v.__cuda_array_interface__ = {"shape": (int(hi - lo) * self.E,), "typestr": "<f8", "data": (base + 8 * self.E * int(lo), False),
                                          "version": 2}
            out.append(torch.as_tensor(v, device="cuda"))
        return out

    # ---- more than one GPU behind the C ABI ----
    def partition(self, rank, world, min_subtrees_per_rank=2, distribute_top=True):
        """mi355_factor_partition; returns (owner[nsuper], counts dict)"""
        _check(lib().mi355_factor_partition(self._h, int(rank), int(world), int(min_subtrees_per_rank),
                                            int(bool(distribute_top))))
        return self.partition_info()

    def partition_info(self):
        owner = np.zeros(max(self.nsuper, 1), dtype=np.int32)
        cnt = np.zeros(11, dtype=np.int64)
        _check(lib().mi355_factor_partition_info(self._h, owner.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)))
        keys = ("world", "rank", "top_blocks", "top_ranges", "reductions", "steps_before_first_reduce",
                "steps_before_last_reduce", "own_steps", "lookahead_bcasts", "lookahead_launches", "top_pieces")
        return owner[:self.nsuper], dict(zip(keys, cnt.tolist()))

    def reduce_plan(self):
        """rows (ready_step, lo, n, root) of the reductions of the top's blocks, in issue order"""
        f = lib().mi355_factor_reduce_plan
        f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
        cnt = C.c_int64(0)
        _check(f(self._h, 0, None, C.byref(cnt)))
        plan = np.zeros((cnt.value, 4), dtype=np.int64)
        if cnt.value:
            _check(f(self._h, cnt.value, plan.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return plan

    def numeric_multi(self, comm, beta=0.0, quick_return_if_not_posdef=False):
        minor, status = C.c_int64(0), C.c_int(0)
        _check(lib().mi355_factor_numeric_multi(self._h, comm._h, float(beta), int(bool(quick_return_if_not_posdef)),
                                                C.byref(minor), C.byref(status)))
        self.minor, self.status = int(minor.value), int(status.value)
        return self.status

    def gather_multi(self, comm):
        """every rank receives the other ranks' subtree columns: Lx() is then the whole factor everywhere"""
        _check(lib().mi355_factor_gather_multi(self._h, comm._h))

    def solve_multi_dev(self, comm, lower, X_ptr, nrhs, d):
        _check(lib().mi355_super_solve_multi_dev(self._h, comm._h, int(bool(lower)), int(nrhs), C.c_void_p(int(X_ptr)),
                                                 int(d)))

    # ---- instrumentation ----
    def set_profile(self, on=True):
        _check(lib().mi355_factor_set_profile(self._h, int(bool(on))))

    def stats(self):
        out = np.zeros(24)
        _check(lib().mi355_factor_get_stats(self._h, out))
        return dict(zip(STAT_KEYS, out.tolist()))

    def device_Lx_ptr(self):
        return lib().mi355_factor_device_Lx(self._h)
