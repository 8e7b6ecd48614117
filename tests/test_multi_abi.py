"""More than one GPU behind the C ABI (csrc/multi.hip.hpp): mi355_partition, mi355_factor_partition,
mi355_factor_numeric_multi, mi355_super_solve_multi_dev with a mi355_comm.

CPU part: the CPU build of the engine (tests/helpers/libmi355_cpuemu.so, test infrastructure: "device" memory is host
memory there) with one THREAD per rank and a blocking transport made of Python callbacks over a barrier -- the
caller-supplied transport of include/mi355_cholmod.h -- and once with two PROCESSES over gloo.  What is checked: every
rank ends with the oracle's columns of L for its subtrees and the top, the solves on the partitioned factor, the
not-positive-definite protocol (same status, minor and zero pattern as the reference on every rank), a factor whose
subtrees are not contiguous index ranges.
GPU part (-m gpu): the RCCL transport bound at run time, with the one rank a one-GPU box has."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from conftest import ROOT, relerr

CPUEMU = os.path.join(ROOT, "tests", "helpers", "libmi355_cpuemu.so")


@pytest.fixture(scope="module")
def engine():
    if not os.path.exists(CPUEMU):
        pytest.skip("libmi355_cpuemu.so not built (needs /opt/rocm's clang++)")
    from suitesparse_amd import engine as e, symbolic
    saved = (e.LIB_PATH, e._LIB)
    e.LIB_PATH, e._LIB = CPUEMU, None
    symbolic._ready = False
    # the engine's RCCL loader binds the stand-in that the CPU build exports (collectives between threads, as stream
    # operations of the emulated runtime): test_stream_ordered_driver_with_the_rccl_stand_in
    saved_env = os.environ.get("MI355_RCCL_LIB")
    os.environ["MI355_RCCL_LIB"] = CPUEMU
    try:
        assert e.probe() == 1
        yield e
    finally:
        e.LIB_PATH, e._LIB = saved
        symbolic._ready = False
        if saved_env is None:
            os.environ.pop("MI355_RCCL_LIB", None)
        else:
            os.environ["MI355_RCCL_LIB"] = saved_env


def host_view(ptr, count):
    return np.ctypeslib.as_array((C.c_double * int(count)).from_address(int(ptr)))


class Bus:
    """blocking collectives among the threads of one process, on host memory; a reduce leaves the other ranks'
    buffers as they are (partial sums), as a real one does"""

    def __init__(self, world, defer=False):
        self.world = world
        self.defer = defer          # MI355_TRANSPORT_DEFER: the engine makes the calls as late as its ordering allows
        self.bar = threading.Barrier(world, timeout=120)
        self.slot = [None] * world
        self.log = [[] for _ in range(world)]

    def comm(self, engine, rank):
        def reduce_sum(p, n, root):
            self.log[rank].append(("reduce", n, root))
            self.slot[rank] = host_view(p, n)
            self.bar.wait()
            if rank == root:
                tot = np.sum([self.slot[r] for r in range(self.world)], axis=0)
                self.slot[rank][...] = tot
            self.bar.wait()

        def bcast(p, n, root):
            self.log[rank].append(("bcast", n, root))
            self.slot[rank] = host_view(p, n)
            self.bar.wait()
            if rank != root:
                self.slot[rank][...] = self.slot[root]
            self.bar.wait()

        def allreduce_sum(p, n):
            self.log[rank].append(("allreduce", n, -1))
            self.slot[rank] = host_view(p, n)
            self.bar.wait()
            tot = np.sum([self.slot[r] for r in range(self.world)], axis=0)
            self.bar.wait()
            self.slot[rank][...] = tot
            self.bar.wait()

        def allreduce_min(v):
            self.slot[rank] = v
            self.bar.wait()
            m = min(self.slot)
            self.bar.wait()
            return m
        return engine.Comm.custom(rank, self.world, reduce_sum, bcast, allreduce_sum, allreduce_min, defer=self.defer)


def run_ranks(world, body):
    """body(rank) on one thread per rank; re-raises the first failure"""
    errs = [None] * world

    def wrap(r):
        try:
            body(r)
        except BaseException as e:          # noqa: BLE001 -- reported below
            errs[r] = e
    ts = [threading.Thread(target=wrap, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not any(t.is_alive() for t in ts), "a rank is stuck"
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errs:
        if e is not None:
            raise e


def check_own_columns(g, owner, rank, Lx, Lo, tol=1e-10):
    for s in range(g["nsuper"]):
        if owner[s] in (rank, -1):
            a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
            nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
            got, want = Lx[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
            low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
            assert relerr(got[low], want[low]) < tol, (rank, s)


@pytest.mark.parametrize("kind,size,world,dist_top,spoil,ob,defer,lookahead", [
    ("lap3d", 6, 3, True, False, 64, True, 1), ("lap3d", 4, 2, True, True, 64, False, 1),
    ("lap3d", 5, 2, False, False, 256, True, 1), ("lap2d", 16, 4, True, False, 64, False, 1),
    ("helm", 4, 2, True, False, 64, True, 1), ("elast", 3, 3, True, False, 64, False, 0),
    ("lap3d", 6, 2, True, False, 64, True, 0), ("lap3d", 10, 2, True, False, 64, True, 1),
    ("lap3d", 12, 3, True, False, 64, True, 1), ("lap2d", 8, 2, False, True, 256, False, 1),
    ("helm", 8, 2, True, False, 64, True, 1),
    # ob = (first nested width, block width): blocks of the top travel in pieces (tuning key 17) while their chain runs
    ("lap3d", 12, 2, True, False, (128, 256), True, 1), ("lap3d", 14, 3, True, False, (128, 256), True, 1),
    ("lap3d", 12, 2, True, False, (128, 256), False, 0), ("lap3d", 12, 2, True, True, (128, 256), True, 1),
    # the default first width (256): the owner of a block runs its windows as panel groups (k_panel_diag, k_panel_trsm_reg;
    # tuning key 28) and a piece leaves when its window is done -- top separators of 324 and 400 columns
    ("lap3d", 18, 2, True, False, (256, 512), True, 1), ("lap3d", 20, 3, True, False, (256, 1024), False, 1),
    ("lap3d", 18, 2, True, True, (256, 512), True, 1), ("lap3d", 18, 3, True, False, (256, 512), True, 0),
    ])
def test_numeric_and_solve_multi_with_threads_as_ranks(engine, kind, size, world, dist_top, spoil, ob, defer, lookahead):
    """defer: the transport completes every collective at the LATEST point the engine's ordering allows (a blocking one
    shows the earliest) -- between the two, a broadcast or a reduction that is read too early or written too late
    gives a wrong factor; lookahead: tuning key 16 (far trailing updates behind the next block's broadcast)"""
    from oracle import port
    from test_multigpu import problem
    g = dict(problem(kind, size))
    n = g["n"]
    cplx = np.iscomplexobj(g["Sx"])
    if spoil:
        Sx = g["Sx"].copy()
        Sx[g["Sp"][n // 2]] = -3.0
        g["Sx"] = Sx
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert (so != 0) == spoil
    ob1, ob2 = ob if isinstance(ob, tuple) else (ob, ob)
    engine.set_tuning(2, ob1); engine.set_tuning(7, ob2); engine.set_tuning(16, lookahead)
    bus = Bus(world, defer)
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, 3)) + (1j * rng.standard_normal((n, 3)) if cplx else 0)
    want = None if spoil else port.super_solve(g, Lo, port.super_solve(g, Lo, B, lower=True), lower=False)
    infos = [None] * world
    try:
        def body(rank):
            L = engine.SupernodalFactor(g, xtype=2 if cplx else 1)
            comm = bus.comm(engine, rank)
            owner, cnt = L.partition(rank, world, 1, dist_top)
            infos[rank] = (owner.copy(), cnt)
            assert cnt["world"] == world and cnt["rank"] == rank
            assert (cnt["top_blocks"] > 0) == dist_top and cnt["reductions"] == cnt["top_blocks"]
            L.set_matrix(g["Sp"], g["Si"], g["Sx"])
            for _ in range(1 if size >= 10 else 2):             # a refactorization reuses everything
                st = L.numeric_multi(comm, 0.0, False)
                assert (st, L.minor) == (so, mo)
                E = 2 if cplx else 1
                Lx = host_view(L.device_Lx_ptr(), g["xsize"] * E)
                Lx = Lx.view(np.complex128) if cplx else Lx
                if spoil:
                    assert np.array_equal(Lx == 0, Lo == 0) and relerr(Lx, Lo) < 1e-10
                else:
                    check_own_columns(g, owner, rank, Lx, Lo)
            if not spoil:
                X = np.array(B.T, order="C", copy=True)        # column-major n x 3
                Xf = X.view(np.float64).reshape(-1)
                L.solve_multi_dev(comm, True, Xf.ctypes.data, 3, n)
                L.solve_multi_dev(comm, False, Xf.ctypes.data, 3, n)
                assert relerr(X.T, want) < 1e-9
            comm.close()
            L.close()
        run_ranks(world, body)
    finally:
        engine.set_tuning(2, 256); engine.set_tuning(7, 1024); engine.set_tuning(16, 1)
    if not lookahead and not isinstance(ob, tuple):
        assert all(i[1]["lookahead_bcasts"] == 0 for i in infos)
    if kind == "lap3d" and size >= 10:      # a top supernode of three or more blocks: the look-ahead really reorders launches
        assert sum(i[1]["lookahead_bcasts"] for i in infos) > 0
    # every rank decided the same partition, and issued the same collectives in the same order
    for r in range(1, world):
        assert np.array_equal(infos[r][0], infos[0][0])
        assert bus.log[r] == bus.log[0]
    kinds = [k for k, _, _ in bus.log[0]]
    if dist_top and not spoil:
        reps = 1 if size >= 10 else 2
        assert kinds.count("reduce") == reps * infos[0][1]["top_blocks"] == kinds.count("bcast")
    if isinstance(ob, tuple):
        # pieces: a piece that is not the last of its block is issued with chain launches between it and its wait point
        assert infos[0][1]["top_pieces"] > 0


def test_reductions_start_before_the_subtrees_are_done(engine):
    """the order of the reductions: a block of the top that only low levels of the subtrees update is summed while the
    remaining levels are still to be enqueued (mi355_factor_partition_info counts[5..7]); and the look-ahead of the
    distributed top: with three or more blocks in a top supernode some broadcasts have launches between their issue
    and their wait point (counts[8..9]), none with tuning key 16 off"""
    from test_multigpu import problem
    engine.set_tuning(2, 64); engine.set_tuning(7, 64)
    try:
        g = dict(problem("lap3d", 9))
        L = engine.SupernodalFactor(g)
        owner, cnt = L.partition(0, 4, 1, True)
        assert cnt["top_blocks"] > 0 and cnt["reductions"] == cnt["top_blocks"]
        assert cnt["steps_before_first_reduce"] <= cnt["steps_before_last_reduce"] <= cnt["own_steps"]
        assert cnt["steps_before_first_reduce"] < cnt["own_steps"]
        L.close()
        g = dict(problem("lap3d", 12))
        L = engine.SupernodalFactor(g)
        owner, cnt = L.partition(0, 2, 1, True)
        assert cnt["lookahead_bcasts"] > 0 and cnt["lookahead_launches"] >= cnt["lookahead_bcasts"]
        L.close()
        engine.set_tuning(16, 0)
        L = engine.SupernodalFactor(g)
        owner0, cnt0 = L.partition(0, 2, 1, True)
        assert np.array_equal(owner0, owner) and cnt0["lookahead_bcasts"] == 0 and cnt0["top_blocks"] == cnt["top_blocks"]
        L.close()
    finally:
        engine.set_tuning(2, 256); engine.set_tuning(7, 1024); engine.set_tuning(16, 1)


def bordered(nleaf, wleaf, border, seed):
    """block diagonal (nleaf dense blocks of wleaf columns) with `border` dense last rows: independent leaves under one
    wide top supernode, in the natural order"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = nleaf * wleaf + border
    M = sp.lil_matrix((n, n))
    for b in range(nleaf):
        M[b * wleaf:(b + 1) * wleaf, b * wleaf:(b + 1) * wleaf] = rng.standard_normal((wleaf, wleaf))
    M[n - border:, :] = rng.standard_normal((border, n))
    A = M.tocsr() + M.tocsr().T
    d = np.asarray(abs(A).sum(axis=1)).ravel() + 1.0
    Lw = sp.tril((A + sp.diags(d)).tocsc(), format="csc")
    Lw.sort_indices()
    return n, Lw.indptr.astype(np.int64), Lw.indices.astype(np.int64), np.ascontiguousarray(Lw.data)


@pytest.mark.parametrize("world,defer,lookahead,pieces,spoil", [
    (2, True, 1, 1, False), (3, False, 1, 1, False), (2, True, 0, 1, False), (4, True, 1, 1, False),
    (2, True, 1, 0, False)])
def test_a_wide_top_supernode_travels_in_pieces(engine, world, defer, lookahead, pieces, spoil):
    """a top supernode of 420 columns with blocks of 256 and a first nested width of 128: four ranges travel (columns
    0-127 and 128-255 of block 0, 256-383 and 384-419 of block 1), the first of each block while its chain is still
    running (tuning key 17); block 1 is updated by block 0 on its owner, so a piece that is read before it has
    arrived, or overwritten while in flight, shows in L (deferred transport: latest completion)"""
    from oracle import port
    from suitesparse_amd import symbolic
    n, Ap, Ai, Ax = bordered(8, 24, 420, 7)
    g = symbolic.analyze(n, Ap, Ai, None)
    Sp, Si, Sx = symbolic.permute(g, Ap, Ai, Ax)
    widest = int(np.max(np.diff(g["Super"])))
    assert widest >= 420
    if spoil:
        Sx = Sx.copy()
        Sx[Sp[n - 100]] = -1.0
    so, mo, Lo = port.super_numeric(g, Sp, Si, Sx)
    assert (so != 0) == spoil
    engine.set_tuning(2, 128); engine.set_tuning(7, 256); engine.set_tuning(16, lookahead); engine.set_tuning(17, pieces)
    bus = Bus(world, defer)
    infos = [None] * world
    try:
        def body(rank):
            L = engine.SupernodalFactor(g)
            comm = bus.comm(engine, rank)
            owner, cnt = L.partition(rank, world, 1, True)
            infos[rank] = cnt
            L.set_matrix(Sp, Si, Sx)
            for _ in range(2):
                st = L.numeric_multi(comm, 0.0, False)
                assert (st, L.minor) == (so, mo)
                Lx = host_view(L.device_Lx_ptr(), g["xsize"])
                if spoil:
                    assert np.array_equal(Lx == 0, Lo == 0) and relerr(Lx, Lo) < 1e-10
                else:
                    check_own_columns(g, owner, rank, Lx, Lo)
            if not spoil:
                L.gather_multi(comm)
                full = L.Lx()
                for s in range(g["nsuper"]):
                    a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
                    nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
                    got, ref = full[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
                    low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
                    assert relerr(got[low], ref[low]) < 1e-10, (rank, s)
            comm.close()
            L.close()
        run_ranks(world, body)
    finally:
        engine.set_tuning(2, 256); engine.set_tuning(7, 1024); engine.set_tuning(16, 1); engine.set_tuning(17, 1)
    for r in range(1, world):
        assert bus.log[r] == bus.log[0]
    for cnt in infos:
        assert cnt["top_pieces"] == (2 if pieces else 0) and cnt["top_blocks"] - cnt["top_pieces"] >= 2
    if pieces:      # on the owner's schedule the chain's launches lie between a piece's issue and its wait point
        assert sum(cnt["lookahead_bcasts"] for cnt in infos) >= 2


class IdBox:
    """rank 0's 128-byte communicator id for the other threads (the side channel of Comm.rccl)"""

    def __init__(self, world):
        self.bar = threading.Barrier(world, timeout=120)
        self.id = None

    def exchange(self, idb):
        if idb is not None:
            self.id = idb
        self.bar.wait()
        return self.id


@pytest.mark.parametrize("kind,size,world,dist_top,spoil,ob,channel", [
    ("lap3d", 6, 3, True, False, 64, 0), ("lap3d", 4, 2, True, True, 64, 0), ("lap3d", 5, 2, False, False, 256, 0),
    ("helm", 4, 2, True, False, 64, 0), ("lap3d", 12, 2, True, False, (128, 256), 0), ("lap2d", 16, 4, True, False, 64, 0),
    ("bordered", 0, 3, True, False, (128, 256), 0),
    ("lap3d", 6, 3, True, False, 64, 1), ("lap3d", 4, 2, True, True, 64, 1), ("helm", 4, 2, True, False, 64, 1),
    ("lap3d", 12, 2, True, False, (128, 256), 1), ("lap3d", 9, 4, True, False, 64, 1),
    ("lap3d", 18, 2, True, False, (256, 512), 1), ("lap3d", 18, 3, True, False, (256, 512), 0)])
def test_stream_ordered_driver_with_the_rccl_stand_in(engine, kind, size, world, dist_top, spoil, ob, channel):
    """The RCCL path of the driver -- mi355_comm_create_rccl, collectives ENQUEUED on the communicator's stream, compute
    and communication ordered by events only, the status agreement through a device buffer -- with the stand-in for
    librccl that the CPU build exports (MI355_RCCL_LIB; tests/helpers/hipcpu/hip/hip_runtime.h): collectives between
    threads, run as operations of the emulated streams, aborting on calls that do not match between ranks.  With
    HIPCPU_STREAMS=lazy / greedy (tests/test_stream_order_emulation.py) a missing fence between the two streams gives a
    wrong factor here.  channel = 1: tuning key 19, the sums of the top's ranges into their owners on a second
    communicator (ncclCommSplit) and stream, waited for supernode by supernode where the top first touches them."""
    from oracle import port
    from suitesparse_amd import symbolic
    from test_multigpu import problem
    if kind == "bordered":
        n, Ap, Ai, Ax = bordered(8, 24, 420, 7)
        g = symbolic.analyze(n, Ap, Ai, None)
        g["Sp"], g["Si"], g["Sx"] = symbolic.permute(g, Ap, Ai, Ax)
    else:
        g = dict(problem(kind, size))
    n = g["n"]
    cplx = np.iscomplexobj(g["Sx"])
    if spoil:
        Sx = g["Sx"].copy()
        Sx[g["Sp"][n // 2 if kind != "bordered" else n - 100]] = -3.0
        g["Sx"] = Sx
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert (so != 0) == spoil
    ob1, ob2 = ob if isinstance(ob, tuple) else (ob, ob)
    engine.set_tuning(2, ob1); engine.set_tuning(7, ob2); engine.set_tuning(19, channel)
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, 3)) + (1j * rng.standard_normal((n, 3)) if cplx else 0)
    want = None if spoil else port.super_solve(g, Lo, port.super_solve(g, Lo, B, lower=True), lower=False)
    box = IdBox(world)
    overlap = [None] * world
    try:
        def body(rank):
            L = engine.SupernodalFactor(g, xtype=2 if cplx else 1)
            comm = engine.Comm.rccl(rank, world, 0, box.exchange)
            owner, cnt = L.partition(rank, world, 1, dist_top)
            L.set_matrix(g["Sp"], g["Si"], g["Sx"])
            for _ in range(2):
                st = L.numeric_multi(comm, 0.0, False)
                assert (st, L.minor) == (so, mo)
                E = 2 if cplx else 1
                Lx = host_view(L.device_Lx_ptr(), g["xsize"] * E)
                Lx = Lx.view(np.complex128) if cplx else Lx
                if spoil:
                    assert np.array_equal(Lx == 0, Lo == 0) and relerr(Lx, Lo) < 1e-10
                else:
                    check_own_columns(g, owner, rank, Lx, Lo)
            overlap[rank] = L.stats()["reduce_overlap_launches"]
            if not spoil:
                X = np.array(B.T, order="C", copy=True)
                Xf = X.view(np.float64).reshape(-1)
                L.solve_multi_dev(comm, True, Xf.ctypes.data, 3, n)
                L.solve_multi_dev(comm, False, Xf.ctypes.data, 3, n)
                assert relerr(X.T, want) < 1e-9
                L.gather_multi(comm)
                check_own_columns(g, np.full(g["nsuper"], rank), rank, L.Lx(), Lo)     # now every supernode
            comm.close()
            L.close()
        run_ranks(world, body)
    finally:
        engine.set_tuning(2, 256); engine.set_tuning(7, 1024); engine.set_tuning(19, 0)
    if not channel or not dist_top:
        assert all(v == 0 for v in overlap)
    elif not spoil and kind == "lap3d" and size > 4:
        # a top of several levels: some rank has top launches enqueued before its last reduction is awaited
        assert max(overlap) > 0, overlap


@pytest.mark.parametrize("seed,transport", [(k, "bus") for k in range(12)] + [(k, "rccl") for k in (1, 2, 3, 4, 6, 11)])
def test_fuzz_of_the_c_driver_on_random_structures(engine, seed, transport):
    """random SPD structures (sparse graphs, chains, arrows, dense blocks) under a random permutation, 2-5 ranks, random
    top mode, block width, look-ahead switch and transport (transport = "rccl": the stream-ordered path with the stand-in
    for librccl instead of the blocking callbacks) (blocking = earliest, deferred = latest completion of every
    collective), sometimes a negative pivot: every rank must end with the oracle's columns of L (or its truncated
    factor, minor and status), and the partitioned solves with its solution"""
    import test_fuzz_schedule as tf
    from oracle import port
    from suitesparse_amd import symbolic
    rng = np.random.default_rng(1000 + seed)
    kind = ["sparse", "banded", "arrow", "blocks"][seed % 4]
    n = int(rng.integers(150, 420))
    world = int(rng.integers(2, 6))
    dist_top, defer, lookahead = bool(rng.integers(0, 4)), bool(rng.integers(0, 2)), int(rng.integers(0, 3) > 0)
    ob = int(rng.choice([64, 64, 128, 256]))
    ob1 = 128 if ob == 256 and seed % 2 else min(ob, 256)      # (128, 256): the top's blocks travel in pieces (key 17)
    spoil = seed % 5 == 4
    Ap, Ai, Ax = tf.random_spd(kind, n, seed)
    g = symbolic.analyze(n, Ap, Ai, rng.permutation(n))
    Sp, Si, Sx = symbolic.permute(g, Ap, Ai, Ax)
    if spoil:
        Sx = Sx.copy()
        k = int(rng.integers(0, n))
        assert Si[Sp[k]] == k
        Sx[Sp[k]] = -abs(Sx[Sp[k]])
    so, mo, Lo = port.super_numeric(g, Sp, Si, Sx)
    assert (so != 0) == spoil
    B = rng.standard_normal((n, 2))
    want = None if spoil else port.super_solve(g, Lo, port.super_solve(g, Lo, B, lower=True), lower=False)
    engine.set_tuning(2, ob1); engine.set_tuning(7, ob); engine.set_tuning(16, lookahead)
    engine.set_tuning(19, 1 if transport == "rccl" and seed % 2 else 0)       # the reduce channel on the odd seeds
    bus = Bus(world, defer)
    box = IdBox(world)
    try:
        def body(rank):
            L = engine.SupernodalFactor(g)
            # "rccl": the stream-ordered path with the stand-in for librccl (collectives as stream operations)
            comm = bus.comm(engine, rank) if transport == "bus" else engine.Comm.rccl(rank, world, 0, box.exchange)
            owner, cnt = L.partition(rank, world, 1, dist_top)
            L.set_matrix(Sp, Si, Sx)
            st = L.numeric_multi(comm, 0.0, False)
            assert (st, L.minor) == (so, mo)
            Lx = host_view(L.device_Lx_ptr(), g["xsize"])
            if spoil:
                assert np.array_equal(Lx == 0, Lo == 0) and relerr(Lx, Lo) < 1e-10
            else:
                check_own_columns(g, owner, rank, Lx, Lo)
                X = np.array(B.T, order="C", copy=True)
                Xf = X.reshape(-1)
                L.solve_multi_dev(comm, True, Xf.ctypes.data, 2, n)
                L.solve_multi_dev(comm, False, Xf.ctypes.data, 2, n)
                assert relerr(X.T, want) < 1e-9
                # the whole factor on every rank, for a caller that promised its own caller a complete L->x
                L.gather_multi(comm)
                full = L.Lx()
                for s in range(g["nsuper"]):
                    a, b = int(g["Lpx"][s]), int(g["Lpx"][s + 1])
                    nsrow = int(g["Lpi"][s + 1] - g["Lpi"][s])
                    got, ref = full[a:b].reshape(-1, nsrow), Lo[a:b].reshape(-1, nsrow)
                    low = np.tril(np.ones((nsrow, got.shape[0]), dtype=bool)).T
                    assert relerr(got[low], ref[low]) < 1e-10, (rank, s)
            comm.close()
            L.close()
        run_ranks(world, body)
    finally:
        engine.set_tuning(2, 256); engine.set_tuning(7, 1024); engine.set_tuning(16, 1); engine.set_tuning(19, 0)
    for r in range(1, world):
        assert bus.log[r] == bus.log[0]


def interleaved_forest():
    """thirteen one-column supernodes whose subtrees are NOT contiguous index ranges (an analysis without
    postordering): leaves 0..7, leaf i -> 8 + i % 4, the four middle nodes -> 12"""
    n = 13
    rows = {j: [j, 8 + j % 4, 12] for j in range(8)}
    rows.update({j: [j, 12] for j in range(8, 12)})
    rows[12] = [12]
    Super = np.arange(n + 1, dtype=np.int64)
    Lpi = np.zeros(n + 1, dtype=np.int64)
    Ls = []
    for j in range(n):
        Ls += rows[j]
        Lpi[j + 1] = len(Ls)
    Ls = np.array(Ls, dtype=np.int64)
    g = dict(n=n, nsuper=n, ssize=len(Ls), xsize=len(Ls), maxcsize=4, maxesize=2, Super=Super, Lpi=Lpi, Lpx=Lpi.copy(), Ls=Ls)
    # S = lower triangle with that pattern, diagonally dominant
    Sp, Si, Sx = Lpi.copy(), Ls.copy(), np.zeros(len(Ls))
    for j in range(n):
        for p in range(int(Lpi[j]), int(Lpi[j + 1])):
            Sx[p] = 10.0 + j if Ls[p] == j else -1.0 - 0.1 * p
    g.update(Sp=Sp, Si=Si, Sx=Sx)
    return g


def test_partition_of_a_factor_that_is_not_postordered(engine):
    from oracle import port
    g = interleaved_forest()
    n = g["n"]
    owner, top_work, total, loads = engine.partition(g, 2, 1)
    assert owner[12] == -1 and (owner[:12] >= 0).all()
    for j in range(8):                                  # a subtree stays whole although its indices are scattered
        assert owner[j] == owner[8 + j % 4]
    assert sorted(np.bincount(owner[8:12], minlength=2).tolist()) == [2, 2]
    assert abs(sum(loads) + top_work - total) < 1e-9
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert so == 0
    bus = Bus(2)

    def body(rank):
        L = engine.SupernodalFactor(g)
        comm = bus.comm(engine, rank)
        own, _ = L.partition(rank, 2, 1, True)
        assert np.array_equal(own, owner)
        L.set_matrix(g["Sp"], g["Si"], g["Sx"])
        assert L.numeric_multi(comm) == 0
        check_own_columns(g, own, rank, host_view(L.device_Lx_ptr(), g["xsize"]), Lo)
        b = np.arange(1.0, n + 1.0)
        x = b.copy()
        L.solve_multi_dev(comm, True, x.ctypes.data, 1, n)
        L.solve_multi_dev(comm, False, x.ctypes.data, 1, n)
        assert relerr(x, port.super_solve(g, Lo, port.super_solve(g, Lo, b[:, None], True), False)[:, 0]) < 1e-12
        comm.close(); L.close()
    run_ranks(2, body)


def test_partition_rejects_a_parent_that_does_not_come_later(engine):
    g = interleaved_forest()
    Ls = g["Ls"].copy()
    # supernode 9's first row below its column becomes row 2: its parent would be supernode 2 < 9
    Ls[int(g["Lpi"][9]) + 1] = 2
    bad = dict(g, Ls=Ls)
    with pytest.raises(engine.Mi355Error) as ei:
        engine.partition(bad, 2, 1)
    assert ei.value.code == engine.CHOLMOD_INVALID


def test_argument_checks(engine):
    from test_multigpu import problem
    g = dict(problem("lap3d", 4))
    L = engine.SupernodalFactor(g)
    bus = Bus(1)
    c1 = bus.comm(engine, 0)
    with pytest.raises(engine.Mi355Error):             # no partition yet
        L.numeric_multi(c1)
    L.partition(0, 2, 1, True)
    L.set_matrix(g["Sp"], g["Si"], g["Sx"])
    with pytest.raises(engine.Mi355Error) as ei:       # the communicator's world differs from the partition's
        L.numeric_multi(c1)
    assert "rank/world" in str(ei.value)
    with pytest.raises(engine.Mi355Error):
        L.partition(2, 2, 1, True)
    # world 1: the plain factorization
    L.partition(0, 1, 1, True)
    assert L.numeric_multi(c1) == 0
    from oracle import port
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert relerr(L.Lx(), Lo) < 1e-10
    # a transport without its callbacks is refused for world > 1
    t = engine._Transport(0, 2, None)
    h = C.c_void_p()
    assert engine.lib().mi355_comm_create(C.byref(t), C.byref(h)) == engine.CHOLMOD_INVALID
    c1.close(); L.close()


WORKER = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from suitesparse_amd import engine, multigpu, symbolic
engine.LIB_PATH, engine._LIB = {lib!r}, None          # the CPU build of the engine (test infrastructure)
symbolic._ready = False
from test_multigpu import problem
from test_multi_abi import host_view, check_own_columns
from oracle import port
g = dict(problem("lap3d", 5))
engine.set_tuning(2, 64); engine.set_tuning(7, 64)
view = lambda p, n: torch.from_numpy(host_view(p, n))
def amin(v):
    t = torch.tensor([v], dtype=torch.int64); dist.all_reduce(t, op=dist.ReduceOp.MIN); return int(t.item())
comm = engine.Comm.custom(rank, world,
                          lambda p, n, root: dist.reduce(view(p, n), dst=root),
                          lambda p, n, root: dist.broadcast(view(p, n), src=root),
                          lambda p, n: dist.all_reduce(view(p, n)), amin)
L = engine.SupernodalFactor(g)
run = multigpu.PartitionedFactor(L, g, rank, world, min_subtrees_per_rank=1, comm=comm)   # forwards to the C driver
L.set_matrix(g["Sp"], g["Si"], g["Sx"])
st = run.numeric()
so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
check_own_columns(g, run.owner, rank, host_view(L.device_Lx_ptr(), g["xsize"]), Lo)
b = np.random.default_rng(3).standard_normal((2, g["n"]))
X = torch.from_numpy(b.copy())
run.lsolve(X, 2)
run.ltsolve(X, 2)
want = port.super_solve(g, Lo, port.super_solve(g, Lo, b.T.copy(), lower=True), lower=False).T
worst = float(np.abs(X.numpy() - want).max() / np.abs(want).max())
Sx = g["Sx"].copy(); Sx[g["Sp"][g["n"] // 2]] = -4.0
L.set_matrix(g["Sp"], g["Si"], Sx)
st2 = run.numeric()
so2, mo2, Lo2 = port.super_numeric(g, g["Sp"], g["Si"], Sx)
Lx = host_view(L.device_Lx_ptr(), g["xsize"])
same = bool(np.array_equal(Lx == 0, Lo2 == 0)) and float(np.abs(Lx - Lo2).max()) < 1e-10 * float(np.abs(Lo2).max())
with open(os.path.join({out!r}, f"result_{{rank}}.txt"), "w") as fh:
    fh.write(f"RESULT {{rank}} {{st}} {{worst}} {{st2}} {{run.minor}} {{so2}} {{mo2}} {{int(same)}} {{run.counts['top_blocks']}}\n")
comm.close()
dist.destroy_process_group()
'''


def test_two_process_gloo_run_of_the_c_driver(engine, tmp_path):
    """two processes, the transport callbacks over gloo: mi355_factor_numeric_multi, the solves, a bad pivot"""
    import socket
    import subprocess
    import sys
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, lib=CPUEMU, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    for attempt in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                             capture_output=True, text=True, env=env, timeout=900)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for r in range(2):
        l = open(tmp_path / f"result_{r}.txt").read().split()
        st, worst, st2, minor2, so2, mo2, same, blocks = int(l[2]), float(l[3]), int(l[4]), int(l[5]), int(l[6]), \
            int(l[7]), int(l[8]), int(l[9])
        assert st == 0 and worst < 1e-9 and blocks > 0
        assert st2 == so2 == 1 and minor2 == mo2 and same == 1


@pytest.mark.gpu
def test_rccl_transport_binds_and_runs_with_one_rank():
    """the RCCL transport on the one GPU of the box: librccl is found and bound at run time (ncclGetUniqueId,
    ncclCommInitRank with the 128-byte id by value), the communicator's stream and key buffer exist, and
    mi355_factor_numeric_multi / _solve_multi_dev with world 1 are the plain calls.  (More ranks need more GPUs: the
    N > 1 order of operations is what the CPU tests above cover.)"""
    from suitesparse_amd import engine
    from oracle import port
    from test_multigpu import problem
    import torch
    g = dict(problem("lap3d", 10))
    comm = engine.Comm.rccl(0, 1, 0, lambda idb: idb)
    L = engine.SupernodalFactor(g)
    L.partition(0, 1, 2, True)
    L.set_matrix(g["Sp"], g["Si"], g["Sx"])
    assert L.numeric_multi(comm) == 0
    so, mo, Lo = port.super_numeric(g, g["Sp"], g["Si"], g["Sx"])
    assert relerr(L.Lx(), Lo) < 1e-10
    b = np.random.default_rng(0).standard_normal((2, g["n"]))
    X = torch.from_numpy(b.copy()).cuda()
    L.solve_multi_dev(comm, True, X.data_ptr(), 2, g["n"])
    L.solve_multi_dev(comm, False, X.data_ptr(), 2, g["n"])
    want = port.super_solve(g, Lo, port.super_solve(g, Lo, b.T.copy(), lower=True), lower=False).T
    assert relerr(X.cpu().numpy(), want) < 1e-9
    comm.close()
    L.close()
