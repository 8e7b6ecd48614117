"""Subtree-to-GPU partitioning of the supernodal factorization.

The reference has no multi-device path (CHOLMOD/GPU/cholmod_gpu.c:138-142 uses one GPU);
this is new work.  One process per GPU:

  1. partition(): supernodes are split into a shared TOP set (ancestors of the cut) and
     disjoint subtrees below it; subtrees are assigned to ranks by LPT on their flop
     weight.  Postordering makes every subtree a contiguous supernode range and a
     contiguous slab of Lx.
  2. phase 0 (per rank): assemble + factorize the own subtrees; their Schur-complement
     contributions to top supernodes accumulate in this rank's copy of the top panels
     (rank 0's copy also holds the entries of A).
  3. ONE sum all-reduce over the Lx ranges of the top supernodes (RCCL over xGMI; adjacent
     top supernodes are coalesced into a single range).
  4. the top: owner-computes by default -- the columns of each top supernode are cut into outer blocks
     (1024 wide), block (s, jb) belongs to one rank, which alone factorizes it and applies the updates
     that target it; a finished block is broadcast before it is read (mi355_factor_top_plan / _top_run).
     The block inverses are summed over the ranks afterwards, so that every rank can run the top sweeps
     of the solves.  (distribute_top=False or MI355_TOP=replicated: every rank factorizes the whole top.)

Solves on the partitioned factor (PartitionedFactor.lsolve / ltsolve): the sweeps over the own subtrees run
locally, the top sweeps replicated; ONE sum all-reduce of X joins the ranks' rows per sweep.

No collective sits inside the data path of a subtree.  """
import os

import numpy as np


def supernode_tree(sym):
    """parent[] of the supernodal etree (parent = SuperMap[Ls[psi+nscol]],
    t_cholmod_super_numeric_worker.c:1226) and the per-supernode flop weight
    nscol^3/3 + nsrow2*nscol^2 (POTRF+TRSM) + nscol*nsrow2^2 (all updates it pushes)."""
    Super, Lpi, Ls = sym["Super"], sym["Lpi"], sym["Ls"]
    ns = int(sym["nsuper"])
    n = int(sym["n"])
    nscol = np.diff(Super).astype(np.int64)
    nsrow = np.diff(Lpi).astype(np.int64)
    supermap = np.repeat(np.arange(ns, dtype=np.int64), nscol)
    assert supermap.size == n
    parent = np.full(ns, -1, dtype=np.int64)
    has = nsrow > nscol
    first_below = Ls[(Lpi[:-1] + nscol)[has]]
    parent[has] = supermap[first_below]
    c = nscol.astype(np.float64)
    r2 = (nsrow - nscol).astype(np.float64)
    work = c ** 3 / 3.0 + r2 * c * c + c * r2 * r2
    return parent, work


def partition(sym, world, min_subtrees_per_rank=2):
    """Returns dict(owner[nsuper] (-1 = top), top_ranges [(lo,hi) Lx offsets], stats).  The partition itself is
    mi355_partition (C ABI, csrc/multi.hip.hpp): one implementation for C callers and for this module."""
    from . import engine
    owner32, top_work, total, loads = engine.partition(sym, world, min_subtrees_per_rank)
    owner = owner32.astype(np.int64)
    if world <= 1:
        return dict(owner=owner, top_ranges=[], top_work=0.0, rank_work=[total], total_work=total)
    Lpx = sym["Lpx"]
    ranges = []
    for s in np.nonzero(owner < 0)[0]:
        lo, hi = int(Lpx[s]), int(Lpx[s + 1])
        if ranges and ranges[-1][1] == lo:
            ranges[-1] = (ranges[-1][0], hi)
        else:
            ranges.append((lo, hi))
    return dict(owner=owner, top_ranges=ranges, top_work=top_work, rank_work=loads, total_work=total)


class PartitionedFactor:
    """Drives the two phases + the reduction.  `factor` is a suitesparse_amd.engine.SupernodalFactor
    (or any object with set_partition / numeric_phase / top_tensors); `allreduce(tensors)` sums
    every tensor over the ranks (default: torch.distributed all_reduce = RCCL on GPUs)."""

    def __init__(self, factor, sym, rank, world, allreduce=None, min_subtrees_per_rank=2, distribute_top=None,
                 bcast=None, reduce=None, comm=None):
        """comm (an engine.Comm): the whole driver runs behind the C ABI -- mi355_factor_partition,
        mi355_factor_numeric_multi, mi355_super_solve_multi_dev -- and this class only forwards.  Without comm and
        without caller-supplied collectives, on an initialised torch.distributed group with the nccl (= RCCL) backend,
        such a communicator is created here.  The Python driver below remains for callers that bring their own
        collectives as Python callables (the lockstep harnesses of the tests)."""
        self.f, self.rank, self.world = factor, rank, world
        self.n = int(sym["n"])
        self.Super = sym["Super"]
        if distribute_top is None:
            distribute_top = os.environ.get("MI355_TOP", "distributed") != "replicated"
        if comm is None and allreduce is None and bcast is None and reduce is None and world > 1 \
                and hasattr(factor, "numeric_multi"):
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
                import torch
                from . import engine
                comm = engine.Comm.from_torch_distributed(torch.cuda.current_device())
        self.comm = comm
        if comm is not None:
            self.owner, self.counts = factor.partition(rank, world, min_subtrees_per_rank, distribute_top)
            self.status, self.minor = 0, self.n
            self.dist = self.counts["top_blocks"] > 0
            return
        self.part = partition(sym, world, min_subtrees_per_rank)
        owner = self.part["owner"]
        local = (owner == rank).astype(np.int8)
        top = (owner < 0).astype(np.int8)
        asm = local.copy()
        if rank == 0:
            asm |= top
        factor.set_partition(local, top, asm)
        self.tensors = factor.top_tensors(self.part["top_ranges"])
        self.allreduce = allreduce or self._dist_allreduce
        self.bcast = bcast or self._dist_bcast
        # owner-computes top: only the owner of a block needs the sum of the ranks' contributions to it, everybody
        # else gets the finished block by broadcast -- so the top panels are REDUCED block by block to their owners
        # (half the volume of the all-reduce per rank; MI355_TOP_REDUCE=allreduce keeps the all-reduce).  A caller
        # that brings its own all-reduce and no reduce keeps the all-reduce.
        self.reduce = reduce or (self._dist_reduce if allreduce is None else None)
        if os.environ.get("MI355_TOP_REDUCE", "owner") == "allreduce":
            self.reduce = None
        self.status, self.minor = 0, self.n
        self.dist = bool(distribute_top) and world > 1 and bool(self.part["top_ranges"])
        if self.dist:
            self.plan, (dlo, dhi) = factor.set_top_distribution(rank, world)
            self.bviews = factor.top_tensors([(int(lo), int(lo + cnt)) for _, lo, cnt, _ in self.plan])
            # the blocks of the plan tile the top panels exactly (each column of a top supernode in one block):
            # what lets the per-block reduce stand in for the all-reduce of the ranges
            blocks = sorted((int(lo), int(lo + cnt)) for _, lo, cnt, _ in self.plan)
            merged = []
            for lo, hi in blocks:
                if merged and merged[-1][1] == lo:
                    merged[-1] = (merged[-1][0], hi)
                else:
                    merged.append((lo, hi))
            if merged != [(int(a), int(b)) for a, b in self.part["top_ranges"]]:
                self.reduce = None        # never expected; the all-reduce is always right
            self.dinv_views = factor.dinv_tensors([(dlo, dhi)]) if dhi > dlo else []

    @staticmethod
    def _dist_allreduce(tensors):
        import torch.distributed as dist
        work = [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in tensors]
        for w in work:
            w.wait()
        if tensors and tensors[0].is_cuda:
            # the engine launches on its own (non-blocking) HIP stream: phase 1 must not start
            # before the collective has finished on the communication stream
            import torch
            torch.cuda.synchronize()

    @staticmethod
    def _dist_reduce(tensors, roots):
        """sum tensors[k] over the ranks into rank roots[k]; the other ranks' copies are overwritten by the block's
        broadcast before anything reads them"""
        import torch.distributed as dist
        work = [dist.reduce(t, dst=int(r), op=dist.ReduceOp.SUM, async_op=True) for t, r in zip(tensors, roots)]
        for w in work:
            w.wait()
        if tensors and tensors[0].is_cuda:
            import torch
            torch.cuda.synchronize()      # the engine's stream is non-blocking: nothing of it may start before the sums are in

    @staticmethod
    def _dist_bcast(tensor, root):
        """enqueue the broadcast of one finished block from its owner (RCCL: returns once enqueued)"""
        import torch.distributed as dist
        dist.broadcast(tensor, src=int(root))

    @staticmethod
    def _comm_done(tensor):
        if tensor.is_cuda:
            # the engine launches on its own (non-blocking) HIP stream: the next segment must not start
            # before the broadcasts have finished on the communication stream
            import torch
            torch.cuda.synchronize()

    def _top_distributed(self):
        """segments of launches (engine stream, finished when top_run returns) between the broadcasts"""
        prev, pending = 0, None
        for k, (step, lo, cnt, root) in enumerate(self.plan):
            if k == 0 or step > prev:
                if pending is not None:
                    self._comm_done(pending)
                    pending = None
                self.f.top_run(k)
            self.bcast(self.bviews[k], root)
            pending, prev = self.bviews[k], step
        if pending is not None:
            self._comm_done(pending)
        return self.f.top_run(len(self.plan))

    def _agree(self, bad, info):
        """smallest failing supernode over the ranks -> (status, minor)"""
        if self.world > 1:
            import torch
            import torch.distributed as dist
            dev = self.tensors[0].device if self.tensors else "cpu"
            key = torch.tensor([bad if bad >= 0 else 2 ** 62], dtype=torch.int64, device=dev)
            dist.all_reduce(key, op=dist.ReduceOp.MIN)
            gbad = int(key.item())
            inf = torch.tensor([info if (bad == gbad) else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(inf, op=dist.ReduceOp.MAX)
            bad, info = (gbad if gbad < 2 ** 62 else -1), int(inf.item())
        if bad >= 0:
            self.status, self.minor = 1, int(self.Super[bad]) + info - 1
        return bad

    def _replay(self, beta, quick):
        """Not positive definite: the reference stops at the SMALLEST failing supernode, zeroes everything after
        the failing column and refactorizes the leading columns of that supernode
        (t_cholmod_super_numeric_worker.c:1084-1146).  A phase only sees the supernodes it processes (a subtree with
        a larger index may fail in phase 0 while a top supernode with a smaller index would fail later), so the
        rare failing case is replayed by every rank on its own with the undistributed schedule
        (mi355_factor_numeric ignores the partition): same L->minor, status and zero pattern as the reference."""
        full = getattr(self.f, "numeric", None)
        if full is None:
            return self.status
        self.status = int(full(beta, quick))
        self.minor = int(self.f.minor)
        return self.status

    def numeric(self, beta=0.0, quick_return_if_not_posdef=False):
        if self.comm is not None:
            self.status = self.f.numeric_multi(self.comm, beta, quick_return_if_not_posdef)
            self.minor = int(self.f.minor)
            return self.status
        self.status, self.minor = 0, self.n
        bad, info = self.f.numeric_phase(0, beta)
        if self._agree(bad, info) >= 0:
            return self._replay(beta, quick_return_if_not_posdef)
        if self.world > 1 and self.tensors:
            if self.dist and self.reduce is not None:
                self.reduce(self.bviews, [int(r) for r in self.plan[:, 3]])
            else:
                self.allreduce(self.tensors)
        if self.dist:
            bad, info = self._top_distributed()
            if self._agree(bad, info) < 0 and self.dinv_views:
                self.allreduce(self.dinv_views)      # every rank gets every block inverse (solves)
        else:
            bad, info = self.f.numeric_phase(1, beta)
            self._agree(bad, info)
        if self.status != 0:
            return self._replay(beta, quick_return_if_not_posdef)
        return self.status

    # ---- solves: X is a torch tensor holding n x nrhs entries column-major (leading dimension n), the
    # same right-hand sides on every rank on entry, the same solution on every rank on return
    def _sum(self, X):
        if self.world > 1:
            self.allreduce([X])

    def lsolve(self, X, nrhs):
        """L x = b (cholmod_super_lsolve): own subtrees, sum over the ranks, top"""
        assert self.status == 0, "no positive-definite factor"
        if self.comm is not None:
            return self.f.solve_multi_dev(self.comm, True, X.data_ptr(), nrhs, self.n)
        self.f.keep_rows_dev(X.data_ptr(), nrhs, self.n)
        self.f.solve_phase_dev(0, True, X.data_ptr(), nrhs, self.n)
        self._sum(X)
        self.f.solve_phase_dev(1, True, X.data_ptr(), nrhs, self.n)

    def ltsolve(self, X, nrhs):
        """L' x = b (cholmod_super_ltsolve): top, own subtrees, union of the ranks' rows"""
        assert self.status == 0, "no positive-definite factor"
        if self.comm is not None:
            return self.f.solve_multi_dev(self.comm, False, X.data_ptr(), nrhs, self.n)
        self.f.solve_phase_dev(1, False, X.data_ptr(), nrhs, self.n)
        self.f.solve_phase_dev(0, False, X.data_ptr(), nrhs, self.n)
        self.f.keep_rows_dev(X.data_ptr(), nrhs, self.n)
        self._sum(X)
