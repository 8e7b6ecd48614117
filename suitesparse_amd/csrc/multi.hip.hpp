// multi.hip.hpp -- more than one GPU behind the C ABI (included at the end of engine.hip: it drives the handle's
// partition schedules directly).  No counterpart in the reference, which uses one device
// (CHOLMOD/GPU/cholmod_gpu.c:138-142); SURVEY.md section 8(e).
//
// One process per GPU, every rank with its own handle on the same symbolic factor and the same staged matrix:
//   mi355_partition               whole etree subtrees -> ranks (LPT on their flops), the ancestors of the cut = the top
//   mi355_factor_partition        the masks + the owner-computes distribution of the top + the order of the reductions
//   mi355_factor_numeric_multi    own subtrees | per-block reduce of the top panels to their owners, issued on the
//                                 communication stream as soon as THIS rank's last contribution to a block is enqueued
//                                 (it overlaps the rest of the subtree work) | the top, owner-computes, a finished block
//                                 broadcast before it is read | sum of the block inverses | one agreement on the status
//   mi355_super_solve_multi_dev   own sweeps, ONE sum of X over the ranks per sweep, top sweeps
// The collectives come from a mi355_comm: RCCL (xGMI), bound at run time from the process's librccl so that the engine
// has no link-time dependency on it -- stream-ordered: the compute and the communication streams order each other
// with events, the host never waits inside a factorization -- or caller-supplied blocking callbacks (MPI, tests).
#include <dlfcn.h>

namespace {

// ---- RCCL, bound at run time.  Signatures of rccl.h (ROCm 7): ncclUniqueId is 128 opaque bytes passed BY VALUE,
// ncclFloat64 = 8, ncclInt64 = 4, ncclSum = 0, ncclMin = 3.
struct RcclId { char internal[128]; };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommSplit)(void *, int, int, void **, void *) = nullptr;     // optional (RCCL >= 2.18): the reduce channel
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
constexpr int RCCL_F64 = 8, RCCL_I64 = 4, RCCL_SUM = 0, RCCL_MIN = 3;

Rccl &rccl()
{
    static Rccl R;
    if (R.lib || !R.why.empty()) return R;
    // MI355_RCCL_LIB: the library to bind, by path (a site's own build of RCCL; the tests' stand-in)
    if (const char *path = getenv("MI355_RCCL_LIB")) {
        R.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!R.lib) { R.why = std::string("MI355_RCCL_LIB: ") + (dlerror() ? dlerror() : "?"); return R; }
    }
    // the copy this process already holds first (PyTorch ships its own): one RCCL per process
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *nm : names) if (!R.lib) R.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char *nm : names) if (!R.lib) R.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!R.lib) R.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!R.lib) { R.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return R; }
    bool ok = true;
    auto sym = [&](const char *nm) { void *p = dlsym(R.lib, nm); if (!p) { ok = false; R.why = std::string("librccl lacks ") + nm; } return p; };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.Reduce = (decltype(R.Reduce))sym("ncclReduce");
    R.Broadcast = (decltype(R.Broadcast))sym("ncclBroadcast");
    R.AllReduce = (decltype(R.AllReduce))sym("ncclAllReduce");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    if (ok) R.CommSplit = (decltype(R.CommSplit))dlsym(R.lib, "ncclCommSplit");
    if (!ok) { R.lib = nullptr; }
    return R;
}

}  // namespace

struct mi355_comm_s {
    int rank = 0, world = 1, device = 0;
    bool stream_ordered = false;        // RCCL: collectives are enqueued; custom: they block the host
    mi355_transport t{};                // custom
    void *nccl = nullptr;               // RCCL communicator
    bool own_nccl = false;
    hipStream_t cs = nullptr;           // communication stream
    std::vector<hipEvent_t> ev;         // fences between the streams, reused round-robin
    size_t ev_next = 0;
    // the reduce channel (tuning key 19): a second communicator (ncclCommSplit) on a stream of its own for the sums of the
    // top's blocks into their owners, so that they run BESIDE the first broadcasts of the top instead of in front of them
    void *nccl2 = nullptr;
    hipStream_t cs2 = nullptr;
    std::vector<hipEvent_t> red_ev;     // red_ev[g]: the reductions of issue group g (and all before it) are complete
    int64_t *dkey = nullptr;            // one int64 on the device (the agreement on the status)
    int grouped = 0;
    // MI355_TRANSPORT_DEFER (blocking transport): collectives are queued and run at the next point where the engine
    // waits for them -- the LATEST completion the ordering allows, where a plain blocking transport shows the earliest
    struct Queued { int kind; double *p; int64_t count; int root; };
    std::vector<Queued> queue;
};

namespace {

#define RCCL_TRY(expr)                                                                    \
    do {                                                                                  \
        int r_ = (expr);                                                                  \
        if (r_ != 0) {                                                                    \
            char b_[512];                                                                 \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr,                       \
                     rccl().GetErrorString ? rccl().GetErrorString(r_) : "?", __FILE__, __LINE__); \
            return fail(MI355_GPU_PROBLEM, b_);                                           \
        }                                                                                 \
    } while (0)
#define CB_TRY(expr, what)                                                                \
    do { if ((expr) != 0) return fail(MI355_GPU_PROBLEM, std::string("transport callback failed: ") + what); } while (0)

// a fence event (64 in rotation: a wait captures the record it follows, so a later re-record does not disturb it)
int comm_event(mi355_comm_s *c, hipEvent_t *out)
{
    if (c->ev.size() < 64) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev.push_back(e);
        *out = e;
        return MI355_OK;
    }
    *out = c->ev[c->ev_next++ % c->ev.size()];
    return MI355_OK;
}

// the next collectives come after everything enqueued on the engine's streams so far
int fence_to_comm(mi355_factor_s *F, mi355_comm_s *c)
{
    if (c->stream_ordered) {
        hipEvent_t e;
        int rc = comm_event(c, &e);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e, F->stream));
        HIP_TRY(hipStreamWaitEvent(c->cs, e, 0));
    } else {
        HIP_TRY(hipStreamSynchronize(F->stream));
    }
    return MI355_OK;
}

int run_queued(mi355_comm_s *c)
{
    for (const mi355_comm_s::Queued &q : c->queue) {
        if (q.kind == 0) CB_TRY(c->t.reduce_sum(c->t.ctx, q.p, q.count, q.root), "reduce_sum");
        else if (q.kind == 1) CB_TRY(c->t.bcast(c->t.ctx, q.p, q.count, q.root), "bcast");
        else CB_TRY(c->t.allreduce_sum(c->t.ctx, q.p, q.count), "allreduce_sum");
    }
    c->queue.clear();
    return MI355_OK;
}

// the next launches on the engine's stream come after the collectives enqueued so far
int fence_to_compute(mi355_factor_s *F, mi355_comm_s *c)
{
    if (!c->stream_ordered && !c->queue.empty()) {
        HIP_TRY(hipStreamSynchronize(F->stream));       // what was launched since the collectives were queued
        return run_queued(c);
    }
    if (c->stream_ordered) {
        hipEvent_t e;
        int rc = comm_event(c, &e);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e, c->cs));
        HIP_TRY(hipStreamWaitEvent(F->stream, e, 0));
    }
    return MI355_OK;        // blocking transport: the callbacks have returned
}

int c_group_begin(mi355_comm_s *c)
{
    if (c->stream_ordered && c->grouped++ == 0) RCCL_TRY(rccl().GroupStart());
    return MI355_OK;
}
int c_group_end(mi355_comm_s *c)
{
    if (c->stream_ordered && --c->grouped == 0) RCCL_TRY(rccl().GroupEnd());
    return MI355_OK;
}
int c_reduce(mi355_comm_s *c, double *p, int64_t count, int root)
{
    if (count <= 0) return MI355_OK;
    if (c->stream_ordered) RCCL_TRY(rccl().Reduce(p, p, (size_t)count, RCCL_F64, RCCL_SUM, root, c->nccl, c->cs));
    else if (c->t.flags & MI355_TRANSPORT_DEFER) c->queue.push_back({0, p, count, root});
    else CB_TRY(c->t.reduce_sum(c->t.ctx, p, count, root), "reduce_sum");
    return MI355_OK;
}
int c_bcast(mi355_comm_s *c, double *p, int64_t count, int root)
{
    if (count <= 0) return MI355_OK;
    if (c->stream_ordered) RCCL_TRY(rccl().Broadcast(p, p, (size_t)count, RCCL_F64, root, c->nccl, c->cs));
    else if (c->t.flags & MI355_TRANSPORT_DEFER) c->queue.push_back({1, p, count, root});
    else CB_TRY(c->t.bcast(c->t.ctx, p, count, root), "bcast");
    return MI355_OK;
}
int c_allreduce(mi355_comm_s *c, double *p, int64_t count)
{
    if (count <= 0) return MI355_OK;
    if (c->stream_ordered) RCCL_TRY(rccl().AllReduce(p, p, (size_t)count, RCCL_F64, RCCL_SUM, c->nccl, c->cs));
    else if (c->t.flags & MI355_TRANSPORT_DEFER) c->queue.push_back({2, p, count, -1});
    else CB_TRY(c->t.allreduce_sum(c->t.ctx, p, count), "allreduce_sum");
    return MI355_OK;
}
// smallest key over the ranks, on the host of every rank (waits for the communication stream)
int c_agree_min(mi355_comm_s *c, int64_t *key)
{
    if (c->stream_ordered) {
        HIP_TRY(hipMemcpyAsync(c->dkey, key, sizeof(int64_t), hipMemcpyHostToDevice, c->cs));
        RCCL_TRY(rccl().AllReduce(c->dkey, c->dkey, 1, RCCL_I64, RCCL_MIN, c->nccl, c->cs));
        HIP_TRY(hipMemcpyAsync(key, c->dkey, sizeof(int64_t), hipMemcpyDeviceToHost, c->cs));
        HIP_TRY(hipStreamSynchronize(c->cs));
    } else {
        if (!c->queue.empty()) return fail(MI355_GPU_PROBLEM, "internal: collectives still queued at the agreement");
        CB_TRY(c->t.allreduce_min_i64(c->t.ctx, key), "allreduce_min_i64");
    }
    return MI355_OK;
}

// ---- subtree-to-rank partition ------------------------------------------------------------------------------------
// supernodal etree (parent = SuperMap[Ls[psi + nscol]], worker :1226) and per-supernode work
// nscol^3/3 + nsrow2 nscol^2 (POTRF + TRSM) + nscol nsrow2^2 (the updates it pushes)
struct Tree { std::vector<int32_t> parent; std::vector<double> work; };

std::string build_tree(int64_t n, int64_t ns, const int64_t *Super, const int64_t *Lpi, const int64_t *Ls, Tree &T)
{
    std::vector<int32_t> supermap((size_t)n);
    for (int64_t s = 0; s < ns; s++) {
        if (Super[s] < 0 || Super[s + 1] > n || Super[s + 1] < Super[s]) return "Super invalid";
        for (int64_t k = Super[s]; k < Super[s + 1]; k++) supermap[(size_t)k] = (int32_t)s;
    }
    if (ns > 0 && (Super[0] != 0 || Super[ns] != n)) return "Super does not cover 0..n";
    T.parent.assign((size_t)ns, -1);
    T.work.assign((size_t)ns, 0.0);
    for (int64_t s = 0; s < ns; s++) {
        const int64_t nscol = Super[s + 1] - Super[s], nsrow = Lpi[s + 1] - Lpi[s];
        if (nsrow < nscol) return "a supernode has fewer rows than columns";
        if (nsrow > nscol) {
            const int64_t i = Ls[Lpi[s] + nscol];
            if (i < 0 || i >= n) return "row index out of range";
            T.parent[(size_t)s] = supermap[(size_t)i];
            if (T.parent[(size_t)s] <= s) return "the parent of a supernode does not come after it";
        }
        const double c = (double)nscol, r2 = (double)(nsrow - nscol);
        T.work[(size_t)s] = c * c * c / 3.0 + r2 * c * c + c * r2 * r2;
    }
    return "";
}

// owner[s] = rank of the subtree s lies in, -1 = top.  The heaviest subtree is split (its root joins the top, its
// children become subtrees) until it weighs at most total / (2 world) and there are at least min_sub * world
// subtrees; the subtrees then go to the least loaded rank, heaviest first.  Membership is explicit (a walk over the
// children lists), so the supernodes of a subtree need not be a contiguous index range (Common->postorder = FALSE).
void partition_tree(const Tree &T, int world, int min_sub, std::vector<int32_t> &owner, std::vector<double> &loads,
                    double &top_work, double &total)
{
    const int64_t ns = (int64_t)T.parent.size();
    std::vector<double> sub(T.work);
    std::vector<int64_t> child_ptr((size_t)ns + 1, 0);
    for (int64_t s = 0; s < ns; s++) {                    // parent > child: one ascending pass accumulates
        const int32_t p = T.parent[(size_t)s];
        if (p >= 0) { sub[(size_t)p] += sub[(size_t)s]; child_ptr[(size_t)p + 1]++; }
    }
    for (int64_t s = 0; s < ns; s++) child_ptr[(size_t)s + 1] += child_ptr[(size_t)s];
    std::vector<int32_t> child((size_t)child_ptr[(size_t)ns]);
    {
        std::vector<int64_t> fill(child_ptr.begin(), child_ptr.end() - 1);
        for (int64_t s = 0; s < ns; s++) {
            const int32_t p = T.parent[(size_t)s];
            if (p >= 0) child[(size_t)fill[(size_t)p]++] = (int32_t)s;
        }
    }
    total = 0.0;
    for (double w : T.work) total += w;
    owner.assign((size_t)ns, -1);
    loads.assign((size_t)std::max(world, 1), 0.0);
    top_work = 0.0;
    if (world <= 1) {
        std::fill(owner.begin(), owner.end(), 0);
        loads[0] = total;
        return;
    }
    typedef std::pair<double, int64_t> Item;            // (-weight, root): the smallest is the heaviest, ties by index
    std::vector<Item> heap;
    auto cmp = [](const Item &a, const Item &b) { return a > b; };    // min-heap on (-weight, root)
    for (int64_t s = 0; s < ns; s++) if (T.parent[(size_t)s] < 0) heap.push_back({-sub[(size_t)s], s});
    std::make_heap(heap.begin(), heap.end(), cmp);
    const double target = total / (2.0 * world);
    std::vector<int64_t> top;
    while (!heap.empty()) {
        const Item head = heap.front();
        const int64_t s = head.second;
        const bool enough = (int64_t)heap.size() >= (int64_t)min_sub * world;
        const bool leaf = child_ptr[(size_t)s] == child_ptr[(size_t)s + 1];
        if ((-head.first <= target && enough) || leaf) break;
        std::pop_heap(heap.begin(), heap.end(), cmp);
        heap.pop_back();
        top.push_back(s);
        for (int64_t q = child_ptr[(size_t)s]; q < child_ptr[(size_t)s + 1]; q++) {
            heap.push_back({-sub[(size_t)child[(size_t)q]], (int64_t)child[(size_t)q]});
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    std::sort(heap.begin(), heap.end());
    std::vector<int64_t> stack;
    for (const Item &it : heap) {
        int r = 0;
        for (int k = 1; k < world; k++) if (loads[(size_t)k] < loads[(size_t)r]) r = k;
        loads[(size_t)r] += -it.first;
        stack.assign(1, it.second);
        while (!stack.empty()) {
            const int64_t s = stack.back();
            stack.pop_back();
            owner[(size_t)s] = r;
            for (int64_t q = child_ptr[(size_t)s]; q < child_ptr[(size_t)s + 1]; q++) stack.push_back(child[(size_t)q]);
        }
    }
    for (int64_t s : top) { owner[(size_t)s] = -1; top_work += T.work[(size_t)s]; }
}

// Order of the per-block reductions of the top panels (the same on every rank) and, for this rank, the step of the
// own-subtree schedule after which its last contribution to each block has been enqueued.  Block k (a BcastPoint of
// the distributed top: columns [c0, c1) of top supernode s) receives updates from the subtree supernodes d that have
// a row in those columns; they are applied by the update step of level(d).  Global order: by the highest such level
// over ALL ranks (a block nobody updates late is summed early), then by k.
int build_reduce_plan(mi355_factor_s *F)
{
    F->reduce_plan.clear();
    if (!F->part_dist) return MI355_OK;
    const Symbolic &S = F->sym;
    const Schedule &dist = F->part_dist->sched, &loc = F->part_local->sched, &full = F->full.sched;
    const int64_t nb = (int64_t)dist.bcasts.size();
    if (nb == 0) return MI355_OK;
    // blocks of each top supernode, by first column
    std::vector<std::vector<std::pair<int64_t, int64_t>>> blocks_of((size_t)S.nsuper);
    for (int64_t k = 0; k < nb; k++) {
        const BcastPoint &b = dist.bcasts[(size_t)k];
        const int64_t nsrow = S.Lpi[b.s + 1] - S.Lpi[b.s];
        blocks_of[(size_t)b.s].push_back({(b.lo - S.Lpx[b.s]) / nsrow, k});
    }
    for (auto &v : blocks_of) std::sort(v.begin(), v.end());
    std::vector<int32_t> glob((size_t)nb, -1), mine((size_t)nb, -1);
    for (int64_t d = 0; d < S.nsuper; d++) {
        const int32_t r = F->part_owner[(size_t)d];
        if (r < 0) continue;
        const int32_t lev = full.level[(size_t)d];
        const int64_t nscol = S.Super[d + 1] - S.Super[d];
        int64_t last_k = -1;
        for (int64_t p = S.Lpi[d] + nscol; p < S.Lpi[d + 1]; p++) {
            const int64_t i = S.Ls[p];
            const int32_t anc = full.supermap[(size_t)i];
            if (F->part_owner[(size_t)anc] >= 0) continue;
            const auto &v = blocks_of[(size_t)anc];
            if (v.empty()) return fail(MI355_GPU_PROBLEM, "internal: a top supernode has no block in the plan");
            const int64_t c = i - S.Super[anc];
            auto it = std::upper_bound(v.begin(), v.end(), std::make_pair(c, (int64_t)INT64_MAX));
            const int64_t k = (it - 1)->second;
            if (k == last_k) continue;
            last_k = k;
            glob[(size_t)k] = std::max(glob[(size_t)k], lev);
            if (r == F->part_rank) mine[(size_t)k] = std::max(mine[(size_t)k], lev);
        }
    }
    // first step of the own schedule beyond level l (steps are in level order)
    auto step_after_level = [&](int32_t l) {
        int64_t i = 0;
        const int64_t nst = (int64_t)loc.steps.size();
        while (i < nst && loc.steps[(size_t)i].level <= l) i++;
        return i;
    };
    std::vector<int64_t> order((size_t)nb);
    for (int64_t k = 0; k < nb; k++) order[(size_t)k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return glob[(size_t)a] < glob[(size_t)b]; });
    int64_t floor_step = 0;
    for (int64_t k : order) {
        mi355_factor_s::ReduceItem it;
        it.k = k;
        it.ready_step = std::max(floor_step, mine[(size_t)k] < 0 ? (int64_t)0 : step_after_level(mine[(size_t)k]));
        floor_step = it.ready_step;         // issue order = list order
        F->reduce_plan.push_back(it);
    }
    return MI355_OK;
}

int check_multi(mi355_factor_s *F, mi355_comm_s *c)
{
    int rc = check_handle(F);
    if (rc) return rc;
    if (!c) return fail(MI355_INVALID, "null communicator");
    if (F->part_world < 1) return fail(MI355_INVALID, "mi355_factor_partition was not called");
    if (c->world != F->part_world || c->rank != F->part_rank)
        return fail(MI355_INVALID, "the communicator's rank/world differ from the partition's");
    if (c->stream_ordered && c->device != F->device)
        return fail(MI355_INVALID, "the communicator was created for another device");
    return MI355_OK;
}

int first_failure(mi355_factor_s *F, int64_t *key)
{
    const Symbolic &S = F->sym;
    int *first = F->info.p + S.nsuper;
    const int big = 0x7f7f7f7f;
    HIP_TRY(hipMemsetAsync(first, 0x7f, sizeof(int), F->stream));
    hipLaunchKernelGGL(k_first_info, dim3((unsigned)((S.nsuper + 255) / 256)), dim3(256), 0, F->stream,
                       S.nsuper, F->info.p, first);
    HIP_TRY(hipGetLastError());
    int sf = big;
    HIP_TRY(hipMemcpyAsync(&sf, first, sizeof(int), hipMemcpyDeviceToHost, F->stream));
    HIP_TRY(hipStreamSynchronize(F->stream));
    *key = INT64_MAX;
    if (sf != big) {
        int info = 0;
        HIP_TRY(hipMemcpy(&info, F->info.p + sf, sizeof(int), hipMemcpyDeviceToHost));
        *key = ((int64_t)sf << 32) | (int64_t)(uint32_t)info;       // smallest supernode, then smallest column
    }
    return MI355_OK;
}

}  // namespace

extern "C" {

int mi355_partition(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Ls,
                    int world, int min_subtrees_per_rank, int32_t *owner, double *stats)
{
    g_err.clear();
    if (n < 0 || nsuper < 0 || world < 1 || min_subtrees_per_rank < 1) return fail(MI355_INVALID, "bad argument");
    if (!owner || (nsuper > 0 && (!Super || !Lpi || !Ls))) return fail(MI355_INVALID, "null argument");
    Tree T;
    if (nsuper > 0) {
        const std::string bad = build_tree(n, nsuper, Super, Lpi, Ls, T);
        if (!bad.empty()) return fail(MI355_INVALID, "invalid supernodal symbolic factor: " + bad);
    }
    std::vector<int32_t> own;
    std::vector<double> loads;
    double top_work = 0, total = 0;
    partition_tree(T, world, min_subtrees_per_rank, own, loads, top_work, total);
    for (int64_t s = 0; s < nsuper; s++) owner[s] = own[(size_t)s];
    if (stats) {
        stats[0] = top_work; stats[1] = total;
        for (int r = 0; r < world; r++) stats[2 + r] = loads[(size_t)r];
    }
    return MI355_OK;
}

int mi355_factor_partition(mi355_factor *F, int rank, int world, int min_subtrees_per_rank, int distribute_top)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (world < 1 || rank < 0 || rank >= world) return fail(MI355_INVALID, "rank/world out of range");
    const Symbolic &S = F->sym;
    F->part_world = 0;
    F->reduce_plan.clear();
    std::vector<int32_t> owner((size_t)S.nsuper);
    rc = mi355_partition(S.n, S.nsuper, S.Super, S.Lpi, S.Ls, world, std::max(1, min_subtrees_per_rank),
                         owner.data(), nullptr);
    if (rc) return rc;
    std::vector<int8_t> local((size_t)S.nsuper), top((size_t)S.nsuper), asmm((size_t)S.nsuper);
    bool any_top = false;
    for (int64_t s = 0; s < S.nsuper; s++) {
        local[(size_t)s] = owner[(size_t)s] == rank;
        top[(size_t)s] = owner[(size_t)s] < 0;
        asmm[(size_t)s] = local[(size_t)s] || (rank == 0 && top[(size_t)s]);       // rank 0's copy of the top holds A
        any_top = any_top || top[(size_t)s];
    }
    rc = mi355_factor_set_partition(F, local.data(), top.data(), asmm.data());
    if (rc) return rc;
    F->part_owner = owner;
    F->part_rank = rank;
    if (distribute_top && world > 1 && any_top) {
        rc = mi355_factor_set_top_distribution(F, rank, world);
        if (rc) return rc;
        F->part_world = world;          // build_reduce_plan reads part_rank / part_owner
        rc = build_reduce_plan(F);
        if (rc) { F->part_world = 0; return rc; }
    }
    // replicated top: the Lx ranges of the top supernodes, adjacent ones merged (one all-reduce each)
    F->top_ranges.clear();
    for (int64_t s = 0; s < S.nsuper; s++) {
        if (!top[(size_t)s]) continue;
        const int64_t lo = S.Lpx[s], hi = S.Lpx[s + 1];
        if (!F->top_ranges.empty() && F->top_ranges.back().second == lo) F->top_ranges.back().second = hi;
        else F->top_ranges.push_back({lo, hi});
    }
    F->part_world = world;
    return MI355_OK;
}

int mi355_factor_partition_info(mi355_factor *F, int32_t *owner, int64_t *counts)
{
    g_err.clear();
    if (!F) return fail(MI355_INVALID, "null factor handle");
    if (F->part_world < 1) return fail(MI355_INVALID, "mi355_factor_partition was not called");
    if (owner) for (size_t s = 0; s < F->part_owner.size(); s++) owner[s] = F->part_owner[s];
    if (counts) {
        counts[0] = F->part_world; counts[1] = F->part_rank;
        counts[2] = F->part_dist ? (int64_t)F->part_dist->sched.bcasts.size() : 0;
        counts[3] = (int64_t)F->top_ranges.size();
        counts[4] = (int64_t)F->reduce_plan.size();
        // how many of this rank's own-subtree steps are enqueued before the first / the last reduction may start
        counts[5] = F->reduce_plan.empty() ? 0 : F->reduce_plan.front().ready_step;
        counts[6] = F->reduce_plan.empty() ? 0 : F->reduce_plan.back().ready_step;
        counts[7] = F->part_local ? (int64_t)F->part_local->sched.steps.size() : 0;
        // look-ahead of the distributed top: broadcasts whose wait point lies behind their issue point, and the launches
        // that run in between (summed over those broadcasts)
        // pieces: entries of the plan that are not the last of their 1024-column block (tuning key 17)
        counts[8] = counts[9] = counts[10] = 0;
        if (F->part_dist)
            for (const BcastPoint &b : F->part_dist->sched.bcasts) {
                if (b.wait_step > b.step) { counts[8]++; counts[9] += b.wait_step - b.step; }
                if (!b.last) counts[10]++;
            }
    }
    return MI355_OK;
}

int mi355_factor_reduce_plan(mi355_factor *F, int64_t cap, int64_t *plan, int64_t *count)
{
    g_err.clear();
    if (!F || !count) return fail(MI355_INVALID, "null argument");
    if (F->part_world < 1) return fail(MI355_INVALID, "mi355_factor_partition was not called");
    *count = (int64_t)F->reduce_plan.size();
    if (!plan) return MI355_OK;
    if (cap < *count) return fail(MI355_INVALID, "plan buffer too small");
    if (!F->part_dist) return MI355_OK;             // replicated top: no per-block reductions
    const Schedule &dsc = F->part_dist->sched;
    for (size_t i = 0; i < F->reduce_plan.size(); i++) {
        const BcastPoint &b = dsc.bcasts[(size_t)F->reduce_plan[i].k];
        plan[4 * i] = F->reduce_plan[i].ready_step; plan[4 * i + 1] = b.lo; plan[4 * i + 2] = b.count; plan[4 * i + 3] = b.root;
    }
    return MI355_OK;
}

int mi355_comm_create(const mi355_transport *t, mi355_comm **out)
{
    g_err.clear();
    if (!t || !out) return fail(MI355_INVALID, "null argument");
    *out = nullptr;
    if (t->world < 1 || t->rank < 0 || t->rank >= t->world) return fail(MI355_INVALID, "rank/world out of range");
    if (t->world > 1 && (!t->reduce_sum || !t->bcast || !t->allreduce_sum || !t->allreduce_min_i64))
        return fail(MI355_INVALID, "a transport callback is missing");
    mi355_comm_s *c = new (std::nothrow) mi355_comm_s;
    if (!c) return fail(MI355_OUT_OF_MEMORY, "host out of memory");
    c->rank = t->rank; c->world = t->world; c->t = *t; c->stream_ordered = false;
    *out = c;
    return MI355_OK;
}

int mi355_rccl_unique_id(void *id128)
{
    g_err.clear();
    if (!id128) return fail(MI355_INVALID, "null argument");
    Rccl &R = rccl();
    if (!R.lib) return fail(MI355_NOT_INSTALLED, R.why);
    RcclId id;
    RCCL_TRY(R.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof id.internal);
    return MI355_OK;
}

static int comm_rccl_finish(mi355_comm_s *c, mi355_comm **out)
{
    HIP_TRY(hipSetDevice(c->device));
    int plo = 0, phi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&plo, &phi));
    HIP_TRY(hipStreamCreateWithPriority(&c->cs, hipStreamNonBlocking, phi));
    HIP_TRY(hipMalloc((void **)&c->dkey, sizeof(int64_t)));
    if (tuning().dist_reduce_channel && c->world > 1) {
        // collective: every rank makes the same call here (the tuning is process-wide and set before the communicator)
        if (!rccl().CommSplit) return fail(MI355_NOT_INSTALLED, "tuning key 19 asks for a reduce channel, but this librccl has no ncclCommSplit");
        RCCL_TRY(rccl().CommSplit(c->nccl, 0, c->rank, &c->nccl2, nullptr));
        HIP_TRY(hipStreamCreateWithPriority(&c->cs2, hipStreamNonBlocking, phi));
    }
    *out = c;
    return MI355_OK;
}

int mi355_comm_create_rccl(const void *id128, int rank, int world, int device, mi355_comm **out)
{
    g_err.clear();
    if (!id128 || !out) return fail(MI355_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(MI355_INVALID, "rank/world out of range");
    Rccl &R = rccl();
    if (!R.lib) return fail(MI355_NOT_INSTALLED, R.why);
    HIP_TRY(hipSetDevice(device));
    mi355_comm_s *c = new (std::nothrow) mi355_comm_s;
    if (!c) return fail(MI355_OUT_OF_MEMORY, "host out of memory");
    struct Guard { mi355_comm_s *c; ~Guard() { if (c) mi355_comm_free(c); } } guard{c};
    c->rank = rank; c->world = world; c->device = device; c->stream_ordered = true;
    RcclId id;
    memcpy(id.internal, id128, sizeof id.internal);
    RCCL_TRY(R.CommInitRank(&c->nccl, world, id, rank));
    c->own_nccl = true;
    int rc = comm_rccl_finish(c, out);
    if (rc) return rc;
    guard.c = nullptr;
    return MI355_OK;
}

int mi355_comm_adopt_rccl(void *nccl_comm, int rank, int world, int device, mi355_comm **out)
{
    g_err.clear();
    if (!nccl_comm || !out) return fail(MI355_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(MI355_INVALID, "rank/world out of range");
    Rccl &R = rccl();
    if (!R.lib) return fail(MI355_NOT_INSTALLED, R.why);
    mi355_comm_s *c = new (std::nothrow) mi355_comm_s;
    if (!c) return fail(MI355_OUT_OF_MEMORY, "host out of memory");
    struct Guard { mi355_comm_s *c; ~Guard() { if (c) mi355_comm_free(c); } } guard{c};
    c->rank = rank; c->world = world; c->device = device; c->stream_ordered = true;
    c->nccl = nccl_comm; c->own_nccl = false;
    int rc = comm_rccl_finish(c, out);
    if (rc) return rc;
    guard.c = nullptr;
    return MI355_OK;
}

int mi355_comm_free(mi355_comm *c)
{
    if (!c) return MI355_OK;
    if (c->stream_ordered) {
        (void)hipSetDevice(c->device);
        if (c->cs) (void)hipStreamSynchronize(c->cs);
        if (c->cs2) (void)hipStreamSynchronize(c->cs2);
        if (c->nccl2 && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl2);
        if (c->own_nccl && c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
        if (c->cs2) (void)hipStreamDestroy(c->cs2);
        if (c->cs) (void)hipStreamDestroy(c->cs);
        if (c->dkey) (void)hipFree(c->dkey);
    }
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->red_ev) (void)hipEventDestroy(e);
    delete c;
    return MI355_OK;
}

int mi355_factor_numeric_multi(mi355_factor *F, mi355_comm *c, double beta_re, int quick_return_if_not_posdef,
                               int64_t *minor, int *status)
{
    g_err.clear();
    int rc = check_multi(F, c);
    if (rc) return rc;
    if (!F->have_A) return fail(MI355_INVALID, "mi355_factor_set_matrix was not called");
    const Symbolic &S = F->sym;
    if (c->world == 1 || S.nsuper == 0) {
        rc = mi355_factor_numeric(F, beta_re, quick_return_if_not_posdef, minor, status);
        return rc;
    }
    F->numeric_ok = false;
    F->part_ok = false; F->lx_epoch++;
    F->not_posdef = false;
    F->st_ngemm = F->st_npotrf = F->st_tiles = F->st_nbig = 0;
    if (!F->timer) F->timer = new EventTimer;
    EventTimer &tm = *F->timer;
    tm.drop();
    tm.on = F->profile != 0;
    tm.st = F->stream;
    HIP_TRY(hipEventRecord(F->ev0, F->stream));
    const int E = F->E;
    DeviceSchedule &loc = *F->part_local;
    const int64_t nloc = (int64_t)loc.sched.steps.size();

    if (F->part_dist) {
        const Schedule &dsc = F->part_dist->sched;
        const int64_t nb = (int64_t)dsc.bcasts.size();
        // ---- own subtrees; every block of the top is summed into its owner as soon as this rank's last
        //      contribution to it is enqueued (the reductions run beside the rest of the subtree work).
        //      With a reduce channel (tuning key 19) they go to the second communicator on its own stream, in issue
        //      groups that also end where the next block belongs to another top supernode, each followed by an event
        const bool chan2 = c->stream_ordered && c->nccl2 != nullptr;
        F->st_red_overlap = 0;
        void *rcomm = chan2 ? c->nccl2 : c->nccl;
        hipStream_t rstream = chan2 ? c->cs2 : c->cs;
        std::vector<int64_t> group_of((size_t)nb, -1);      // issue group of the reduction of each range of the plan
        int64_t ngroups = 0;
        int64_t pos = 0;
        bool init = true;
        size_t i = 0;
        while (i < F->reduce_plan.size()) {
            const int64_t need = F->reduce_plan[i].ready_step;
            if (need > pos || init) {
                rc = run_schedule(F, loc, S.n, beta_re, tm, F->asm_mask.p, init, pos, need);
                if (rc) return rc;
                pos = need;
                init = false;
            }
            if (chan2) {
                hipEvent_t e;
                rc = comm_event(c, &e);
                if (rc) return rc;
                HIP_TRY(hipEventRecord(e, F->stream));
                HIP_TRY(hipStreamWaitEvent(c->cs2, e, 0));
            } else {
                rc = fence_to_comm(F, c);
                if (rc) return rc;
            }
            while (i < F->reduce_plan.size() && F->reduce_plan[i].ready_step == need) {
                const int32_t s0 = dsc.bcasts[(size_t)F->reduce_plan[i].k].s;
                rc = c_group_begin(c);
                if (rc) return rc;
                for (; i < F->reduce_plan.size() && F->reduce_plan[i].ready_step == need &&
                       (!chan2 || dsc.bcasts[(size_t)F->reduce_plan[i].k].s == s0); i++) {
                    const BcastPoint &b = dsc.bcasts[(size_t)F->reduce_plan[i].k];
                    group_of[(size_t)F->reduce_plan[i].k] = ngroups;
                    if (c->stream_ordered) {
                        if (b.count > 0) {
                            const int r_ = rccl().Reduce(F->Lx.p + b.lo * E, F->Lx.p + b.lo * E, (size_t)(b.count * E), RCCL_F64,
                                                         RCCL_SUM, b.root, rcomm, rstream);
                            if (r_ != 0) { (void)c_group_end(c); return fail(MI355_GPU_PROBLEM, "ncclReduce failed"); }
                        }
                    } else {
                        rc = c_reduce(c, F->Lx.p + b.lo * E, b.count * E, b.root);
                        if (rc) { (void)c_group_end(c); return rc; }
                    }
                }
                rc = c_group_end(c);
                if (rc) return rc;
                if (chan2) {
                    while ((int64_t)c->red_ev.size() <= ngroups) {
                        hipEvent_t e;
                        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                        c->red_ev.push_back(e);
                    }
                    HIP_TRY(hipEventRecord(c->red_ev[(size_t)ngroups], c->cs2));
                }
                ngroups++;
            }
        }
        if (pos < nloc || init) {
            rc = run_schedule(F, loc, S.n, beta_re, tm, F->asm_mask.p, init, pos, nloc);
            if (rc) return rc;
        }
        // when must which reductions be complete?  Without a reduce channel: all of them, before the top starts.  With
        // one: group g before the first launch of THIS rank's top schedule that touches a top supernode with a range in
        // a group >= ... <= g, i.e. for every top supernode s the last group that holds one of its ranges, before the
        // first launch that reads or writes s (the in-order channel completes the earlier groups with it)
        std::vector<std::pair<int64_t, int64_t>> red_wait;      // (step of the top schedule, group), by step
        if (chan2) {
            std::vector<int64_t> gmax((size_t)S.nsuper, -1), touch((size_t)S.nsuper, INT64_MAX);
            for (int64_t k = 0; k < nb; k++)
                if (group_of[(size_t)k] >= 0) gmax[(size_t)dsc.bcasts[(size_t)k].s] = std::max(gmax[(size_t)dsc.bcasts[(size_t)k].s], group_of[(size_t)k]);
            auto super_of = [&](int64_t off) { return (int64_t)(std::upper_bound(S.Lpx, S.Lpx + S.nsuper + 1, off) - S.Lpx) - 1; };
            auto mark = [&](int64_t off, int64_t si) {
                const int64_t s_ = super_of(off);
                if (s_ >= 0 && s_ < S.nsuper) touch[(size_t)s_] = std::min(touch[(size_t)s_], si);
            };
            for (size_t si = 0; si < dsc.steps.size(); si++) {
                const Step &st = dsc.steps[si];
                for (int64_t ti = st.first; ti < st.first + st.count; ti++) {
                    if (st.kind == STEP_POTRF) { mark(dsc.potrf[(size_t)ti].a_off, (int64_t)si); continue; }
                    if (st.kind == STEP_PDIAG || st.kind == STEP_PTRSM) { mark(dsc.pgroups[(size_t)ti].a_off, (int64_t)si); continue; }
                    const GemmTask &t = dsc.gemm[(size_t)ti];
                    mark(t.a_off, (int64_t)si);
                    if (t.mode != GEMM_OVERWRITE) mark(t.b_off, (int64_t)si);
                    mark(t.c_off, (int64_t)si);
                }
            }
            for (int64_t s_ = 0; s_ < S.nsuper; s_++)
                if (gmax[(size_t)s_] >= 0)      // (a top supernode no launch of this rank touches: waited for at the end)
                    red_wait.push_back({touch[(size_t)s_] == INT64_MAX ? (int64_t)dsc.steps.size() : touch[(size_t)s_], gmax[(size_t)s_]});
            std::sort(red_wait.begin(), red_wait.end());
        } else {
            rc = fence_to_compute(F, c);
            if (rc) return rc;
        }
        // ---- the top, owner-computes: the launches between two broadcast points, then the finished blocks
        {
            const int64_t lo = F->part_local->sched.dinv_size, hi = dsc.dinv_size;
            if (hi > lo) HIP_TRY(hipMemsetAsync(F->Dinv.p + lo * E, 0, (size_t)(hi - lo) * E * sizeof(double), F->stream));
        }
        // every broadcast has an issue point (its block is finished) and a wait point (the first launch that reads the
        // block); between the two lie launches that do not need it (BcastPoint::wait_step: look-ahead)
        int64_t step0 = 0, k = 0, wait_at = -1;
        size_t rw = 0;
        int64_t red_done = -1;          // groups <= red_done have been waited for by the compute stream
        // the compute stream waits for the reductions that the launches up to (not including) step `upto` need
        auto wait_reductions = [&](int64_t upto) -> int {
            int64_t g = red_done;
            for (; rw < red_wait.size() && red_wait[rw].first <= upto; rw++) g = std::max(g, red_wait[rw].second);
            if (g > red_done) {
                HIP_TRY(hipStreamWaitEvent(F->stream, c->red_ev[(size_t)g], 0));
                red_done = g;
                F->st_red_overlap = (double)std::min<int64_t>(upto, (int64_t)dsc.steps.size());
            }
            return MI355_OK;
        };
        // run the top's launches [step0, upto), stopping where a reduction has to be waited for first
        auto run_top = [&](int64_t upto) -> int {
            while (step0 < upto) {
                int r = wait_reductions(step0);
                if (r) return r;
                const int64_t stop = rw < red_wait.size() ? std::min(upto, std::max(red_wait[rw].first, step0 + 1)) : upto;
                r = run_schedule(F, *F->part_dist, 0, 0.0, tm, nullptr, false, step0, stop);
                if (r) return r;
                step0 = stop;
            }
            return MI355_OK;
        };
        while (k < nb || wait_at >= 0) {
            const int64_t issue_at = k < nb ? dsc.bcasts[(size_t)k].step : INT64_MAX;
            if (wait_at >= 0 && wait_at <= issue_at) {
                rc = run_top(wait_at);
                if (rc) return rc;
                wait_at = -1;
                rc = fence_to_compute(F, c);            // everything issued so far has arrived
                if (rc) return rc;
                continue;
            }
            rc = run_top(issue_at);
            if (rc) return rc;
            rc = fence_to_comm(F, c);
            if (rc) return rc;
            if (chan2) {
                // a range is broadcast into the copies that its reduction read: the broadcast channel waits for it
                int64_t g = -1;
                for (int64_t kk = k; kk < nb && dsc.bcasts[(size_t)kk].step == issue_at; kk++) g = std::max(g, group_of[(size_t)kk]);
                if (g >= 0) HIP_TRY(hipStreamWaitEvent(c->cs, c->red_ev[(size_t)g], 0));
            }
            rc = c_group_begin(c);
            if (rc) return rc;
            for (; k < nb && dsc.bcasts[(size_t)k].step == issue_at; k++) {
                const BcastPoint &b = dsc.bcasts[(size_t)k];
                if (b.wait_step < b.step) { (void)c_group_end(c); return fail(MI355_GPU_PROBLEM, "internal: a broadcast is awaited before it is issued"); }
                wait_at = wait_at < 0 ? b.wait_step : std::min(wait_at, b.wait_step);
                rc = c_bcast(c, F->Lx.p + b.lo * E, b.count * E, b.root);
                if (rc) { (void)c_group_end(c); return rc; }
            }
            rc = c_group_end(c);
            if (rc) return rc;
        }
        rc = run_top((int64_t)dsc.steps.size());
        if (rc) return rc;
        rc = wait_reductions(INT64_MAX);        // (ranges of top supernodes this rank never touches)
        if (rc) return rc;
    } else {
        // ---- replicated top: own subtrees, ONE sum of the top panels over the ranks, the whole top on every rank
        rc = run_schedule(F, loc, S.n, beta_re, tm, F->asm_mask.p, true);
        if (rc) return rc;
        if (!F->top_ranges.empty()) {
            rc = fence_to_comm(F, c);
            if (rc) return rc;
            rc = c_group_begin(c);
            if (rc) return rc;
            for (const auto &r : F->top_ranges) {
                rc = c_allreduce(c, F->Lx.p + r.first * E, (r.second - r.first) * E);
                if (rc) { (void)c_group_end(c); return rc; }
            }
            rc = c_group_end(c);
            if (rc) return rc;
            rc = fence_to_compute(F, c);
            if (rc) return rc;
        }
        rc = run_schedule(F, *F->part_top, 0, 0.0, tm, nullptr, false);
        if (rc) return rc;
    }

    // ---- one agreement: the smallest failing supernode over the ranks (and the smallest column inside it)
    int64_t key = INT64_MAX;
    rc = first_failure(F, &key);
    if (rc) return rc;
    rc = c_agree_min(c, &key);
    if (rc) return rc;
    if (key != INT64_MAX) {
        // Not positive definite: the reference stops at the SMALLEST failing supernode, zeroes what follows the failing
        // column and refactorizes the leading columns of that supernode (worker :1084-1146).  The rare failing case
        // is replayed by every rank on its own with the undistributed schedule: same L->minor, status, zero pattern.
        rc = mi355_factor_numeric(F, beta_re, quick_return_if_not_posdef, minor, status);
        if (rc) return rc;
        if (status && *status == MI355_STATUS_OK)
            return fail(MI355_GPU_PROBLEM, "internal: the partitioned run met a bad pivot that the replay did not");
        return MI355_OK;
    }
    if (F->part_dist) {
        // every rank gets every block inverse of the top (the solves sweep the top replicated)
        const int64_t lo = F->part_local->sched.dinv_size, hi = F->part_dist->sched.dinv_size;
        if (hi > lo) {
            rc = fence_to_comm(F, c);
            if (rc) return rc;
            rc = c_allreduce(c, F->Dinv.p + lo * E, (hi - lo) * E);
            if (rc) return rc;
            rc = fence_to_compute(F, c);
            if (rc) return rc;
        }
    }
    HIP_TRY(hipEventRecord(F->ev1, F->stream));
    HIP_TRY(hipEventSynchronize(F->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, F->ev0, F->ev1));
    F->st_total = ms;
    tm.pending = tm.on;
    F->minor = S.n;
    F->part_ok = true; F->lx_epoch++;
    if (minor) *minor = S.n;
    if (status) *status = MI355_STATUS_OK;
    return MI355_OK;
}

int mi355_factor_gather_multi(mi355_factor *F, mi355_comm *c)
{
    g_err.clear();
    int rc = check_multi(F, c);
    if (rc) return rc;
    if (c->world == 1) return MI355_OK;
    if (!F->part_ok) return fail(MI355_INVALID, "no positive-definite partitioned factor is resident");
    const Symbolic &S = F->sym;
    const int E = F->E;
    rc = fence_to_comm(F, c);
    if (rc) return rc;
    rc = c_group_begin(c);
    if (rc) return rc;
    // every maximal run of consecutive supernodes of one owner is one contiguous range of Lx
    for (int64_t s = 0; s < S.nsuper;) {
        const int32_t r = F->part_owner[(size_t)s];
        int64_t e = s + 1;
        while (e < S.nsuper && F->part_owner[(size_t)e] == r) e++;
        if (r >= 0) {
            rc = c_bcast(c, F->Lx.p + S.Lpx[s] * E, (S.Lpx[e] - S.Lpx[s]) * E, r);
            if (rc) { (void)c_group_end(c); return rc; }
        }
        s = e;
    }
    rc = c_group_end(c);
    if (rc) return rc;
    rc = fence_to_compute(F, c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(F->stream));
    return MI355_OK;
}

int mi355_super_solve_multi_dev(mi355_factor *F, mi355_comm *c, int lower, int64_t nrhs, double *X, int64_t d)
{
    g_err.clear();
    int rc = check_multi(F, c);
    if (rc) return rc;
    const int64_t n = F->sym.n;
    if (nrhs < 0 || d < n) return fail(MI355_INVALID, "X and L dimensions must match");
    if (n == 0 || nrhs == 0) return MI355_OK;
    if (!X) return fail(MI355_INVALID, "null X");
    if (c->world == 1) return solve_dev(F, lower != 0, nrhs, X, d);
    if (!F->part_ok) return fail(MI355_INVALID, "no positive-definite partitioned factor is resident");
    const int E = F->E;
    auto sum_X = [&]() -> int {
        int r = fence_to_comm(F, c);
        if (r) return r;
        if (d == n) r = c_allreduce(c, X, n * nrhs * E);
        else {
            r = c_group_begin(c);
            for (int64_t j = 0; j < nrhs && !r; j++) r = c_allreduce(c, X + j * d * E, n * E);
            const int r2 = c_group_end(c);
            if (!r) r = r2;
        }
        if (r) return r;
        r = fence_to_compute(F, c);
        if (r) return r;
        if (c->stream_ordered) HIP_TRY(hipStreamSynchronize(F->stream));    // X is the caller's: complete on return
        return MI355_OK;
    };
    double ms = 0;
    if (lower) {
        // L x = b: own subtrees, sum over the ranks, top (replicated)
        rc = mi355_solve_keep_rows_dev(F, nrhs, X, d);
        if (rc) return rc;
        rc = solve_dev(F, true, nrhs, X, d, F->part_local);
        if (rc) return rc;
        ms += F->st_solve;
        rc = sum_X();
        if (rc) return rc;
        rc = solve_dev(F, true, nrhs, X, d, F->part_top);
        if (rc) return rc;
        ms += F->st_solve;
    } else {
        // L' x = b: top, own subtrees, union of the ranks' rows
        rc = solve_dev(F, false, nrhs, X, d, F->part_top);
        if (rc) return rc;
        ms += F->st_solve;
        rc = solve_dev(F, false, nrhs, X, d, F->part_local);
        if (rc) return rc;
        ms += F->st_solve;
        rc = mi355_solve_keep_rows_dev(F, nrhs, X, d);
        if (rc) return rc;
        rc = sum_X();
        if (rc) return rc;
    }
    F->st_solve = ms;
    return MI355_OK;
}

}  // extern "C"
