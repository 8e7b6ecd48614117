// engine.hip -- device engine + C ABI (include/mi355_cholmod.h) of the MI355X-native
// supernodal numeric factorization / solve.  gfx950 only.
//
// Host side mirrors the reference wrapper cholmod_super_numeric
// (CHOLMOD/Supernodal/cholmod_super_numeric.c:100-335): validate, build SuperMap,
// run the worker, report status/minor -- except that the worker is the level-scheduled
// device pipeline of schedule.hpp instead of the sequential CPU loop
// (t_cholmod_super_numeric_worker.c:335-1266).  There is NO CPU fallback in this
// library: without a usable device every entry point fails with MI355_GPU_PROBLEM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/mi355_cholmod.h"
#include "schedule.hpp"
#include "kernels.hip.hpp"
#include "solve_kernels.hip.hpp"
#include "kernels_complex.hip.hpp"

using namespace mi355;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            char b_[512];                                                                 \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                 \
            return fail(e_ == hipErrorOutOfMemory ? MI355_OUT_OF_MEMORY : MI355_GPU_PROBLEM, b_); \
        }                                                                                 \
    } while (0)

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        return hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
    }
    hipError_t upload(const T *src, size_t count, hipStream_t st)
    {
        hipError_t e = hipSuccess;
        if (!p || n != count) e = alloc(count);            // same size (refactorization): keep the buffer
        if (e != hipSuccess || count == 0) return e;
        e = hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(st);
    }
};

// device image of one Schedule
struct DeviceSchedule {
    Schedule sched;
    DevBuf<GemmTask> gemm;
    DevBuf<PotrfTask> potrf;
    DevBuf<int32_t> potrf_order;        // Schedule::potrf_order (empty unless the tiny classes are on)
    DevBuf<PanelGroup> pgroups;         // Schedule::pgroups / pg_wg_task (panel groups, Tuning::panel_group)
    DevBuf<int32_t> pg_wg_task;
    DevBuf<int32_t> relmap, solve_tmap;
    DevBuf<Schedule::TileEntry> gemm_tmap;
    // grouped solve sweep (built on first use)
    SolveSchedule ssched;
    DevBuf<SolveGroupTask> sgroups;
    DevBuf<PotrfTask> sutasks;
    DevBuf<int32_t> stmap;
    DevBuf<SolveRange> sranges, sranges_mma;
    // ... its wide path (many right-hand sides): one-block groups on lists of their own + SolveGroupUpd tasks
    DevBuf<PotrfTask> snutasks;
    DevBuf<int32_t> sntmap;
    DevBuf<SolveRange> snranges_mma, swranges;
    DevBuf<SolveGroupUpd> swtasks;
    DevBuf<int32_t> swwg;
    bool solve_groups_ready = false;
    // ... explicit inverses of the groups' diagonal blocks for the steps of few groups (tuning key 31; k_solve_group_inv)
    DevBuf<double> ginv;                // images of inv(L_g): row r of group g at (ginv_off[g] + r * 256) entries; the images of
                                        // inv(L_g)^H (backward sweep) follow, ginv_rows * 256 entries further
    DevBuf<int64_t> ginv_off;           // per group of ssched.groups; -1: none
    DevBuf<SolveGroupTask> ginv_tasks;  // the groups that have one, x0 = first row of their image (for the chain that builds it)
    DevBuf<double> ginv_scratch;        // k_solve_group_inv: a 256 x 16 tile per (group of a step, slab of 16 real columns), zero between launches
    DevBuf<int> ginv_tickets;           // ... and its ticket
    int64_t ginv_slots = 0;
    std::vector<uint8_t> step_inv;      // per step of ssched.steps: every group of the step has an inverse
    int64_t ginv_rows = 0;
    uint64_t ginv_epoch = ~(uint64_t)0; // F->lx_epoch the images were built from
    bool ginv_ready = false, ginv_off_for_good = false;
};

}  // namespace

namespace { struct EventTimer; }
struct mi355_factor_s {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;             // look-ahead: the far part of the outermost trailing updates (STEP_SIDE)
    std::vector<hipEvent_t> side_ev;        // fork/join events, reused by every factorization
    std::vector<hipEvent_t> solve_ev;       // look-ahead of the sweeps (tuning key 30): 4 + 4 + 1 events, reused by every solve
    int xtype = MI355_REAL;
    int E = 1;                  // doubles per entry: 1 real, 2 complex (interleaved)
    // host copy of the symbolic factor
    std::vector<int64_t> Super, Lpi, Lpx, Ls;
    Symbolic sym;
    // device
    DevBuf<double> Lx, Dinv;
    DevBuf<int64_t> dPerm;                 // L->Perm for mi355_solve (empty: identity)
    DevBuf<double> Ywork;                  // n x nrhs workspace of mi355_solve (grown on demand)
    size_t ywork_cap = 0;
    DevBuf<double> Yrow;                   // row-major image of the right-hand sides for the MFMA sweeps
    size_t yrow_cap = 0;
    DevBuf<int32_t> dLs, dSupermap;
    DevBuf<int64_t> dSuper, dLpi, dLpx;
    DevBuf<int> info;           // nsuper + 2 (first-failure scratch at [nsuper], [nsuper+1])
    DeviceSchedule full;
    // multi-GPU partition: own subtrees / shared top, see mi355_factor_set_partition
    DeviceSchedule *part_local = nullptr, *part_top = nullptr;
    DeviceSchedule *part_dist = nullptr;   // the top, owner-computes over the ranks (mi355_factor_set_top_distribution)
    std::vector<int8_t> top_mask_host;
    DevBuf<int8_t> asm_mask;
    // mi355_factor_partition (multi.hip.hpp): who owns what, and the order of the reductions of the top's blocks
    int part_world = 0, part_rank = 0;      // part_world == 0: not called (or undone by a direct set_partition)
    std::vector<int32_t> part_owner;        // rank of each supernode's subtree, -1 = top
    struct ReduceItem { int64_t k; int64_t ready_step; };   // block k of the plan; own-subtree steps enqueued before it
    std::vector<ReduceItem> reduce_plan;
    std::vector<std::pair<int64_t, int64_t>> top_ranges;    // Lx ranges of the top supernodes, adjacent ones merged
    // staged matrix
    DevBuf<int64_t> Ap, Ai, Anz;
    DevBuf<double> Ax;
    bool have_A = false, A_unpacked = false;
    bool A_perm = false;        // the staged matrix is A itself; dIperm places it (mi355_factor_set_matrix_perm)
    DevBuf<int64_t> dIperm;
    // stype == 0: the staged pair is (A, F) and the matrix factorized is A*F (mi355_factor_set_matrix_unsym)
    bool A_unsym = false, F_unpacked = false;
    DevBuf<int64_t> Fp, Fi, Fnz;
    DevBuf<double> Fx;
    bool numeric_ok = false;    // a positive-definite factor is resident
    uint64_t lx_epoch = 0;      // bumped whenever numeric_ok or part_ok is assigned: what was derived from Lx / Dinv (the groups' inverses) is stale
    bool part_ok = false;       // ... through the two partition phases (own subtrees + top)
    bool not_posdef = false;    // the last mi355_factor_numeric stopped at a bad pivot (L->minor < n)
    int64_t minor = 0;
    // instrumentation
    int profile = 0;
    double st_total = 0, st_gemm = 0, st_potrf = 0, st_asm = 0, st_solve = 0;
    double st_ngemm = 0, st_npotrf = 0, st_tiles = 0, st_nbig = 0, st_big = 0;
    EventTimer *timer = nullptr;   // events of the last profiled numeric call, collected lazily
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // bracket of one numeric / solve call (created once: no leak on error paths)
    // mi355_factor_numeric_host: chunks of Lx go to the host as soon as they are final
    std::vector<HostCopyChunk> copy_plan;
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> copy_ev;            // one per chunk, created on first use
    // lanes of the copy: a pinned staging buffer + a stream each (the DMA runs at PCIe rate into pinned memory; the lanes'
    // host threads then move their pieces into the caller's pageable L->x in parallel)
    std::vector<void *> lane_pin;
    std::vector<hipStream_t> lane_stream;
    size_t lane_bytes = 0;
    DevBuf<double> solve_b, solve_x;            // device images of B and X of mi355_solve_x, kept from call to call
    size_t solve_b_cap = 0, solve_x_cap = 0;
    double st_copy_early = 0;                   // entries copied before the last step had been enqueued
    double st_red_overlap = 0;                  // reduce channel: launches of the top enqueued before the last reduction was awaited
};

namespace {

__global__ void k_first_info(int64_t nsuper, const int *__restrict__ info, int *__restrict__ out)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nsuper && info[s] != 0) atomicMin(out, (int)s);
}

// MI355_CREATE_LOG=1: host seconds of the phases of mi355_factor_create / upload_schedule on stderr
struct PhaseClock {
    bool on = getenv("MI355_CREATE_LOG") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mi355 create] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

int upload_schedule(mi355_factor_s *F, DeviceSchedule &D)
{
    Schedule &S = D.sched;
    PhaseClock pc;
    {
        // no task may point outside the arrays the kernels index with it (a GPU memory fault takes the machine down)
        // (Dinv is sized by the full schedule: all blocks)
        const std::string bad = check_schedule_bounds(F->sym, S, nullptr, std::max<int64_t>(S.dinv_size, F->full.sched.dinv_size));
        if (bad.compare(0, 10, "too large:") == 0) return fail(MI355_TOO_LARGE, bad);
        if (!bad.empty()) return fail(MI355_GPU_PROBLEM, "internal: schedule rejected before upload: " + bad);
    }
    pc.lap("check_schedule_bounds");
    HIP_TRY(D.gemm.upload(S.gemm.data(), S.gemm.size(), F->stream));
    HIP_TRY(D.potrf.upload(S.potrf.data(), S.potrf.size(), F->stream));
    if (!S.potrf_order.empty()) HIP_TRY(D.potrf_order.upload(S.potrf_order.data(), S.potrf_order.size(), F->stream));
    if (!S.pgroups.empty()) HIP_TRY(D.pgroups.upload(S.pgroups.data(), S.pgroups.size(), F->stream));
    if (!S.pg_wg_task.empty()) HIP_TRY(D.pg_wg_task.upload(S.pg_wg_task.data(), S.pg_wg_task.size(), F->stream));
    HIP_TRY(D.gemm_tmap.upload(S.gemm_tile_tab.data(), S.gemm_tile_tab.size(), F->stream));
    HIP_TRY(D.solve_tmap.upload(S.solve_tile_task.data(), S.solve_tile_task.size(), F->stream));
    HIP_TRY(D.relmap.alloc((size_t)S.relmap_size));
    pc.lap("upload task lists");
    if (S.relmap_size > 0) {
        std::vector<RelmapPair> rp(S.pairs.size());
        for (size_t i = 0; i < rp.size(); i++) {
            rp[i].ls_src = S.pairs[i].ls_src; rp[i].ls_dst = S.pairs[i].ls_dst;
            rp[i].rm_off = S.pairs[i].rm_off; rp[i].nsrow = S.pairs[i].nsrow; rp[i].pad = 0;
        }
        DevBuf<RelmapPair> drp;
        HIP_TRY(drp.upload(rp.data(), rp.size(), F->stream));
        DevBuf<int> derr;
        HIP_TRY(derr.alloc(1));
        HIP_TRY(hipMemsetAsync(derr.p, 0, sizeof(int), F->stream));
        const int64_t total = S.relmap_size;
        const unsigned grid = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(k_relmap, dim3(grid), dim3(256), 0, F->stream, total, (int64_t)rp.size(),
                           drp.p, F->dLs.p, D.relmap.p, derr.p);
        HIP_TRY(hipGetLastError());
        int herr = 0;
        HIP_TRY(hipMemcpyAsync(&herr, derr.p, sizeof(int), hipMemcpyDeviceToHost, F->stream));
        HIP_TRY(hipStreamSynchronize(F->stream));
        if (herr) return fail(MI355_INVALID, "symbolic factor invalid: a descendant row is missing from its ancestor");
    }
    pc.lap("relmap (device)");
    return MI355_OK;
}

struct EventTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[4];   // 0 gemm64 1 potrf 2 asm 3 gemm128
    std::vector<hipEvent_t> pool;                           // events of earlier calls, reused: none is created
                                                            // inside a timed step once the first profiled call has run
    bool on = false;
    hipStream_t st = nullptr;
    hipEvent_t get()
    {
        hipEvent_t x = nullptr;
        if (!pool.empty()) { x = pool.back(); pool.pop_back(); }
        else (void)hipEventCreate(&x);
        return x;
    }
    void begin(int kind, hipStream_t on_stream = nullptr)
    {
        if (!on) return;
        hipEvent_t x = get(), y = get();
        (void)hipEventRecord(x, on_stream ? on_stream : st);
        ev[kind].push_back({x, y});
    }
    void end(int kind, hipStream_t on_stream = nullptr)
    {
        if (!on) return;
        (void)hipEventRecord(ev[kind].back().second, on_stream ? on_stream : st);
    }
    bool pending = false;
    void drop()
    {
        for (int k = 0; k < 4; k++) {
            for (auto &p : ev[k]) { pool.push_back(p.first); pool.push_back(p.second); }
            ev[k].clear();
        }
        pending = false;
    }
    void destroy_pool()
    {
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
    double collect(int kind)
    {
        double ms = 0;
        for (auto &p : ev[kind]) {
            float t = 0;
            (void)hipEventSynchronize(p.second);
            (void)hipEventElapsedTime(&t, p.first, p.second);
            ms += t;
            pool.push_back(p.first); pool.push_back(p.second);
        }
        ev[kind].clear();
        return ms;
    }
};

// Copies chunks of the device factor to the host on a thread of its own while the main thread goes on enqueueing the
// factorization: the thread waits for a chunk's event (recorded on the compute stream after the chunk's last writer),
// then copies on the copy stream.  A blocking copy into pageable memory (L->x is cholmod_malloc'ed) therefore never
// holds up the enqueueing of kernels.
struct HostCopier {
    struct Item { hipEvent_t ev; int64_t lo, hi; };     // entries [lo, hi) of Lx, final once ev has completed
    mi355_factor_s *F;
    double *dst;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Item> q;
    bool closed = false;
    hipError_t err = hipSuccess;
    size_t next = 0;            // next chunk of F->copy_plan to hand over (main thread)
    // lanes: threads that each own a pinned staging buffer and a stream of the handle (make_lanes)
    bool started = true;        // false: a lane's thread could not be created (the caller reports out of memory)
    HostCopier(mi355_factor_s *F_, double *dst_) : F(F_), dst(dst_)
    {
        // a std::vector of joinable threads must never be destroyed by an exception: the threads already running are
        // told to end and joined here, and the caller sees `started == false`
        try {
            for (size_t l = 0; l < F->lane_stream.size(); l++) th.emplace_back([this, l] { run(l); });
        } catch (const std::exception &) {
            started = false;
            { std::lock_guard<std::mutex> lk(mu); closed = true; }
            cv.notify_all();
            for (std::thread &t : th) if (t.joinable()) t.join();
            th.clear();
        }
    }
    ~HostCopier() { (void)finish(); }
    void run(size_t lane)
    {
        hipError_t e = hipSetDevice(F->device);
        const size_t E = (size_t)F->E;
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return closed || !q.empty(); });
                if (q.empty()) break;
                it = q.front();
                q.pop_front();
            }
            if (e != hipSuccess) continue;      // keep draining so that finish() returns
            e = hipEventSynchronize(it.ev);
            const size_t bytes = (size_t)(it.hi - it.lo) * E * sizeof(double);
            if (e == hipSuccess)
                e = hipMemcpyAsync(F->lane_pin[lane], F->Lx.p + (size_t)it.lo * E, bytes, hipMemcpyDeviceToHost, F->lane_stream[lane]);
            if (e == hipSuccess) e = hipStreamSynchronize(F->lane_stream[lane]);
            if (e == hipSuccess) memcpy(dst + (size_t)it.lo * E, F->lane_pin[lane], bytes);
        }
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess) err = e;
    }
    // main thread: entries [lo, hi) are final on stream `st` from here on; they go to the lanes in pieces that fit a
    // staging buffer
    int hand_over_range(hipEvent_t ev, int64_t lo, int64_t hi)
    {
        const int64_t piece = std::max<int64_t>(1, (int64_t)(F->lane_bytes / (sizeof(double) * (size_t)F->E)));
        {
            std::lock_guard<std::mutex> lk(mu);
            for (int64_t a = lo; a < hi; a += piece) q.push_back({ev, a, std::min(hi, a + piece)});
        }
        cv.notify_all();
        return MI355_OK;
    }
    int hand_over(size_t k, hipStream_t st)
    {
        while (F->copy_ev.size() <= k) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            F->copy_ev.push_back(e);
        }
        HIP_TRY(hipEventRecord(F->copy_ev[k], st));
        return hand_over_range(F->copy_ev[k], F->copy_plan[k].lo, F->copy_plan[k].hi);
    }
    // wait for every copy handed over so far; the threads end
    int finish()
    {
        { std::lock_guard<std::mutex> lk(mu); closed = true; }
        cv.notify_all();
        for (std::thread &t : th) if (t.joinable()) t.join();
        th.clear();
        if (err != hipSuccess) return fail(MI355_GPU_PROBLEM, std::string("copy of the factor to the host failed: ") + hipGetErrorString(err));
        return MI355_OK;
    }
};

// the lanes of the handle (tuning key 20 = how many, key 21 = MB per staging buffer), made on first use
int make_lanes(mi355_factor_s *F)
{
    if (!F->lane_stream.empty()) return MI355_OK;
    // sized by the factor, the largest thing a handle moves: one lane per MB of it up to the tuned number, staging buffers
    // no larger than a lane's share (a factor of a few KB, as most of a CHOLMOD application's are, pins 64 KB and starts
    // one thread, not 4 x 32 MB and four)
    const size_t fbytes = (size_t)F->sym.xsize * (size_t)F->E * sizeof(double);
    const int lanes = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, std::min(32, tuning().host_lanes)), (fbytes + (1u << 20) - 1) >> 20));
    const size_t share = ((fbytes / (size_t)lanes + 65535) >> 16) << 16;
    F->lane_bytes = std::max<size_t>(65536, std::min<size_t>((size_t)std::max(1, tuning().host_lane_mb) << 20, share));
    for (int l = 0; l < lanes; l++) {
        void *p = nullptr;
        HIP_TRY(hipHostMalloc(&p, F->lane_bytes, hipHostMallocDefault));
        F->lane_pin.push_back(p);
        hipStream_t st = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        F->lane_stream.push_back(st);
    }
    return MI355_OK;
}

// Host <-> device through the lanes: pieces (each at most one staging buffer) are taken by the lanes' threads in turn.
// To the device: memcpy into the lane's pinned buffer, DMA, wait -- the data is on the device when the call returns.
// To the host: every lane first waits for `after` (recorded on the stream that produces the data), then DMA into the
// pinned buffer and memcpy out.  Pageable memory on the host side at PCIe rate either way.
struct LanePiece { char *host; char *dev; size_t bytes; };
int lanes_transfer(mi355_factor_s *F, const std::vector<LanePiece> &pieces, bool to_device, hipEvent_t after)
{
    int rc = make_lanes(F);
    if (rc) return rc;
    std::atomic<size_t> next{0};
    std::mutex mu;
    hipError_t err = hipSuccess;
    auto work = [&](size_t lane) {
        hipError_t e = hipSetDevice(F->device);
        if (e == hipSuccess && after) e = hipEventSynchronize(after);
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= pieces.size() || e != hipSuccess) break;
            const LanePiece &p = pieces[k];
            if (to_device) {
                memcpy(F->lane_pin[lane], p.host, p.bytes);
                e = hipMemcpyAsync(p.dev, F->lane_pin[lane], p.bytes, hipMemcpyHostToDevice, F->lane_stream[lane]);
                if (e == hipSuccess) e = hipStreamSynchronize(F->lane_stream[lane]);
            } else {
                e = hipMemcpyAsync(F->lane_pin[lane], p.dev, p.bytes, hipMemcpyDeviceToHost, F->lane_stream[lane]);
                if (e == hipSuccess) e = hipStreamSynchronize(F->lane_stream[lane]);
                if (e == hipSuccess) memcpy(p.host, F->lane_pin[lane], p.bytes);
            }
        }
        if (e != hipSuccess) { std::lock_guard<std::mutex> lk(mu); err = e; }
    };
    const size_t nl = std::min(F->lane_stream.size(), pieces.size());
    {
        // (a thread that cannot be created must not take the vector of running threads down with it: the lanes that did
        // start, and the calling thread, do all the pieces)
        std::vector<std::thread> th;
        for (size_t l = 1; l < nl; l++) {
            try { th.emplace_back(work, l); } catch (const std::exception &) { break; }
        }
        work(0);                                   // the calling thread is lane 0
        for (std::thread &t : th) t.join();
    }
    if (err != hipSuccess) return fail(MI355_GPU_PROBLEM, std::string("staged copy failed: ") + hipGetErrorString(err));
    return MI355_OK;
}

// the pieces of a matrix of `cols` columns of `width` bytes (host pitch hp, device pitch dp)
void lane_pieces(mi355_factor_s *F, char *host, size_t hp, char *dev, size_t dp, size_t width, size_t cols,
                 std::vector<LanePiece> &out)
{
    const size_t cap = F->lane_bytes;
    if (hp == width && dp == width) { width *= cols; cols = 1; }       // contiguous on both sides: one long run
    for (size_t j = 0; j < cols; j++)
        for (size_t a = 0; a < width; a += cap)
            out.push_back({host + j * hp + a, dev + j * dp + a, std::min(cap, width - a)});
}

// run one schedule: zero Lx, assemble the first ncols_asm columns of A, then the steps
int run_schedule(mi355_factor_s *F, DeviceSchedule &D, int64_t ncols_asm, double beta, EventTimer &tm,
                 const int8_t *smask = nullptr, bool init = true, int64_t step0 = 0, int64_t step1 = -1,
                 HostCopier *hc = nullptr)
{
    const Symbolic &S = F->sym;
    tm.begin(2);
    if (init) {
        HIP_TRY(hipMemsetAsync(F->Lx.p, 0, (size_t)S.xsize * F->E * sizeof(double), F->stream));
        HIP_TRY(hipMemsetAsync(F->info.p, 0, (size_t)(S.nsuper + 2) * sizeof(int), F->stream));
    }
    const bool cplx = F->E == 2;
    if (init && ncols_asm > 0) {
        const unsigned grid = (unsigned)((ncols_asm + 255) / 256);
        if (F->A_unsym) {
            hipLaunchKernelGGL(cplx ? k_assemble_aat<2> : k_assemble_aat<1>, dim3(grid), dim3(256), 0, F->stream, ncols_asm,
                               F->Ap.p, F->Ai.p, F->A_unpacked ? F->Anz.p : (const int64_t *)nullptr, F->Ax.p,
                               F->Fp.p, F->Fi.p, F->F_unpacked ? F->Fnz.p : (const int64_t *)nullptr, F->Fx.p,
                               F->dSupermap.p, F->dSuper.p, F->dLpi.p, F->dLpx.p, F->dLs.p, F->Lx.p, smask);
            if (beta != 0.0)
                hipLaunchKernelGGL(cplx ? k_add_beta<2> : k_add_beta<1>, dim3(grid), dim3(256), 0, F->stream, ncols_asm, beta,
                                   F->dSupermap.p, F->dSuper.p, F->dLpi.p, F->dLpx.p, F->Lx.p, smask);
        } else if (F->A_perm) {
            // every column of A is walked; ncols_asm limits the columns of S that receive entries
            hipLaunchKernelGGL(cplx ? k_assemble_perm<2> : k_assemble_perm<1>, dim3((unsigned)((S.n + 255) / 256)), dim3(256), 0,
                               F->stream, S.n, ncols_asm, F->Ap.p, F->Ai.p, F->A_unpacked ? F->Anz.p : (const int64_t *)nullptr, F->Ax.p, F->dIperm.p,
                               F->dSupermap.p, F->dSuper.p, F->dLpi.p, F->dLpx.p, F->dLs.p, F->Lx.p, smask);
            if (beta != 0.0)
                hipLaunchKernelGGL(cplx ? k_add_beta<2> : k_add_beta<1>, dim3(grid), dim3(256), 0, F->stream, ncols_asm, beta,
                                   F->dSupermap.p, F->dSuper.p, F->dLpi.p, F->dLpx.p, F->Lx.p, smask);
        } else {
            hipLaunchKernelGGL(cplx ? k_assemble_c : k_assemble, dim3(grid), dim3(256), 0, F->stream, ncols_asm,
                               F->Ap.p, F->Ai.p, F->A_unpacked ? F->Anz.p : (const int64_t *)nullptr, F->Ax.p,
                               F->dSupermap.p, F->dSuper.p, F->dLpi.p, F->dLpx.p, F->dLs.p, F->Lx.p, beta, smask);
        }
        HIP_TRY(hipGetLastError());
    }
    tm.end(2);
    // MI355_STEP_LOG=file: one line per launch (synchronising; analysis runs only)
    FILE *steplog = nullptr;
    if (const char *fn = getenv("MI355_STEP_LOG")) steplog = fopen(fn, "w");
    hipEvent_t le0 = nullptr, le1 = nullptr;
    if (steplog) { (void)hipEventCreate(&le0); (void)hipEventCreate(&le1); }
    struct LogCloser { FILE *&f; hipEvent_t &a, &b; ~LogCloser() { if (f) { fclose(f); (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } } closer{steplog, le0, le1};
    if (step1 < 0) step1 = (int64_t)D.sched.steps.size();
    bool side_pending = false;
    // chunks of Lx that are final once the first si steps have run go to the host now
    auto hand_over_ready = [&](int64_t si) -> int {
        while (hc && hc->next < F->copy_plan.size() && F->copy_plan[hc->next].ready_step <= si) {
            if (side_pending) {         // a far update on the side stream may still be writing
                HIP_TRY(hipStreamWaitEvent(F->stream, F->side_ev[1], 0));
                side_pending = false;
            }
            int rc = hc->hand_over(hc->next, F->stream);
            if (rc) return rc;
            if (si < step1) F->st_copy_early += (double)(F->copy_plan[hc->next].hi - F->copy_plan[hc->next].lo);
            hc->next++;
        }
        return MI355_OK;
    };
    for (int64_t si = step0; si < step1; si++) {
        { int rc = hand_over_ready(si); if (rc) return rc; }
        const Step &st = D.sched.steps[(size_t)si];
        if (st.count <= 0) continue;
        // look-ahead fork/join (schedule.hpp): `q` is the stream this step launches on
        hipStream_t q = F->stream;
        if ((st.flags & STEP_JOIN) && side_pending) {
            HIP_TRY(hipStreamWaitEvent(F->stream, F->side_ev[1], 0));
            side_pending = false;
        }
        if (st.flags & STEP_SIDE) {
            while (F->side_ev.size() < 2) {
                hipEvent_t ev;
                HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                F->side_ev.push_back(ev);
            }
            HIP_TRY(hipEventRecord(F->side_ev[0], F->stream));
            HIP_TRY(hipStreamWaitEvent(F->side, F->side_ev[0], 0));
            q = F->side;
        }
        if (steplog) (void)hipEventRecord(le0, q);
        if (st.kind == STEP_POTRF) {
            if (st.count > 0x7fffffffLL) return fail(MI355_TOO_LARGE, "too many POTRF tasks in one step");
            tm.begin(1);
            if (cplx)
                hipLaunchKernelGGL(k_potrf64c, dim3((unsigned)st.count), dim3(256), 0, F->stream,
                                   D.potrf.p + st.first, F->Lx.p, F->Dinv.p, F->info.p);
            else if (!D.sched.potrf_order.empty()) {
                // blocks of at most 16 columns first (one wave each), the others one workgroup each
                const int32_t *ord = D.potrf_order.p + st.first;
                if (st.nsmall > 0)
                    hipLaunchKernelGGL(k_potrf16, dim3((unsigned)((st.nsmall + POTRF16_WAVES - 1) / POTRF16_WAVES)),
                                       dim3(64 * POTRF16_WAVES), 0, F->stream, D.potrf.p + st.first, ord, (int64_t)st.nsmall,
                                       F->Lx.p, F->Dinv.p, F->info.p);
                // (the last nfused tasks of the step were factorized by the trailing update that completed them)
                if (st.count - st.nfused > st.nsmall)
                    hipLaunchKernelGGL(k_potrf64, dim3((unsigned)(st.count - st.nfused - st.nsmall)), dim3(256), 0, F->stream,
                                       D.potrf.p + st.first, F->Lx.p, F->Dinv.p, F->info.p, ord + st.nsmall);
            } else if (st.count > st.nfused)
                hipLaunchKernelGGL(k_potrf64, dim3((unsigned)(st.count - st.nfused)), dim3(256), 0, F->stream,
                                   D.potrf.p + st.first, F->Lx.p, F->Dinv.p, F->info.p, (const int32_t *)nullptr);
            tm.end(1);
            if (st.count > st.nfused) F->st_npotrf += 1;
        } else if (st.kind == STEP_PDIAG) {
            // panel groups: the diagonal windows (<= 256 columns each), one workgroup per window
            if (cplx || st.count > 0x7fffffffLL) return fail(MI355_GPU_PROBLEM, "internal: panel group step");
            tm.begin(1);
            hipLaunchKernelGGL(k_panel_diag, dim3((unsigned)st.count), dim3(256), 0, F->stream,
                               (const PanelGroup *)(D.pgroups.p + st.first), F->Lx.p, F->Dinv.p, F->info.p);
            tm.end(1);
            F->st_npotrf += 1;
        } else if (st.kind == STEP_PTRSM) {
            // ... and the rows below them, ROWS rows of a panel per workgroup through the whole window
            if (cplx || st.ntiles > 0x7fffffffLL) return fail(MI355_GPU_PROBLEM, "internal: panel group step");
            tm.begin(0);
            const PanelGroup *pg = D.pgroups.p + st.first;
            const int32_t *wt = D.pg_wg_task.p + st.tmap_off;
            if (st.ngrouped == 128)
                hipLaunchKernelGGL(k_panel_trsm_reg<8>, dim3((unsigned)st.ntiles), dim3(512), 0, F->stream, pg, wt, (const double *)F->Dinv.p, F->Lx.p);
            else if (st.ngrouped == 64)
                hipLaunchKernelGGL(k_panel_trsm_reg<4>, dim3((unsigned)st.ntiles), dim3(256), 0, F->stream, pg, wt, (const double *)F->Dinv.p, F->Lx.p);
            else if (st.ngrouped == 32)
                hipLaunchKernelGGL(k_panel_trsm<32>, dim3((unsigned)st.ntiles), dim3(256), 0, F->stream, pg, wt, (const double *)F->Dinv.p, F->Lx.p);
            else
                hipLaunchKernelGGL(k_panel_trsm<16>, dim3((unsigned)st.ntiles), dim3(256), 0, F->stream, pg, wt, (const double *)F->Dinv.p, F->Lx.p);
            tm.end(0);
            F->st_ngemm += 1;
        } else {
            if (st.ntiles > 0x7fffffffLL || st.count > 0x7fffffffLL)
                return fail(MI355_TOO_LARGE, "too many GEMM tiles in one step");
            const int tk = st.big ? 3 : 0;
            tm.begin(tk, q);
            if (st.big)
                hipLaunchKernelGGL(cplx ? k_gemm128<true> : k_gemm128<false>, dim3((unsigned)st.ntiles),
                                   dim3(256), 0, q,
                                   D.gemm.p + st.first, D.gemm_tmap.p + st.tmap_off, F->Lx.p, F->Dinv.p, D.relmap.p,
                                   (unsigned long long *)nullptr, (const PotrfTask *)D.potrf.p, F->Dinv.p, F->info.p);
            else if (cplx)
                hipLaunchKernelGGL(k_gemm64c, dim3((unsigned)st.ntiles), dim3(256), 0, q,
                                   D.gemm.p + st.first, D.gemm_tmap.p + st.tmap_off, F->Lx.p, F->Dinv.p, D.relmap.p);
            else if (st.flags & STEP_TINY)
                hipLaunchKernelGGL(k_gemm_tiny, dim3((unsigned)((st.ntiles + TINY_WAVES - 1) / TINY_WAVES)), dim3(64 * TINY_WAVES), 0, q,
                                   D.gemm.p + st.first, D.gemm_tmap.p + st.tmap_off, st.ntiles, F->Lx.p, D.relmap.p);
            else if (st.trsm && !st.big)
                hipLaunchKernelGGL(k_trsm64, dim3((unsigned)st.ntiles), dim3(256), 0, q,
                                   D.gemm.p + st.first, D.gemm_tmap.p + st.tmap_off, F->Lx.p, F->Dinv.p);
            else
                hipLaunchKernelGGL(k_gemm64, dim3((unsigned)st.ntiles), dim3(256), 0, q,
                                   D.gemm.p + st.first, D.gemm_tmap.p + st.tmap_off, F->Lx.p, F->Dinv.p, D.relmap.p);
            tm.end(tk, q);
            F->st_ngemm += 1;
            if (tk == 3) F->st_nbig += 1;
            F->st_tiles += (double)st.ntiles;
            if (st.flags & STEP_SIDE) {
                HIP_TRY(hipEventRecord(F->side_ev[1], F->side));
                side_pending = true;
            }
        }
        HIP_TRY(hipGetLastError());
        if (steplog) {
            (void)hipEventRecord(le1, q);
            (void)hipEventSynchronize(le1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, le0, le1);
            if (st.kind == STEP_POTRF) {
                fprintf(steplog, "potrf level %d count %lld ms %.4f\n", st.level, (long long)st.count, ms);
            } else if (st.kind == STEP_PDIAG || st.kind == STEP_PTRSM) {
                fprintf(steplog, "%s level %d count %lld ntiles %lld ms %.4f\n", st.kind == STEP_PDIAG ? "pdiag" : "ptrsm", st.level,
                        (long long)st.count, (long long)st.ntiles, ms);
            } else {
                double fl = 0, xfl = 0; int kmin = 1 << 30, kmax = 0; int64_t mmax = 0, nmax = 0;
                for (int64_t i = 0; i < st.count; i++) {
                    const GemmTask &t = D.sched.gemm[(size_t)(st.first + i)];
                    const double Mm = t.M, Nn = t.N, Kk = t.K;
                    fl += (t.mode == GEMM_OVERWRITE) ? 2 * Mm * Nn * Kk : (Nn * Nn + 2.0 * (Mm - Nn) * Nn) * Kk;
                    // flops the launched tiles execute (full tiles; complex counted like fl, once per entry)
                    xfl += (double)gemm_tiles(t.M, t.N, t.mode, tile_rows(t.big), tile_cols(t.big)) *
                           tile_rows(t.big) * tile_cols(t.big) * 2.0 * Kk;
                    kmin = std::min(kmin, t.K); kmax = std::max(kmax, t.K);
                    mmax = std::max<int64_t>(mmax, t.M); nmax = std::max<int64_t>(nmax, t.N);
                }
                fprintf(steplog, "gemm level %d big %d mode %d count %lld ntiles %lld flops %.6g xflops %.6g kmin %d kmax %d mmax %lld nmax %lld ms %.4f\n",
                        st.level, st.big, D.sched.gemm[(size_t)st.first].mode, (long long)st.count, (long long)st.ntiles, fl, xfl,
                        kmin, kmax, (long long)mmax, (long long)nmax, ms);
            }
        }
    }
    if (side_pending) { HIP_TRY(hipStreamWaitEvent(F->stream, F->side_ev[1], 0)); side_pending = false; }
    return hand_over_ready(step1);
}

int check_handle(mi355_factor_s *F)
{
    if (!F) return fail(MI355_INVALID, "null factor handle");
    HIP_TRY(hipSetDevice(F->device));
    return MI355_OK;
}

}  // namespace

extern "C" {

const char *mi355_last_error(void) { return g_err.c_str(); }

int mi355_probe(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int usable = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) usable++;
    }
    return usable;
}

int mi355_current_device(int *device)
{
    g_err.clear();
    if (!device) return fail(MI355_INVALID, "null argument");
    *device = 0;
    HIP_TRY(hipGetDevice(device));
    return MI355_OK;
}

int mi355_device_memory(int device, int64_t *total_bytes, int64_t *free_bytes)
{
    g_err.clear();
    if (!total_bytes || !free_bytes) return fail(MI355_INVALID, "null argument");
    *total_bytes = *free_bytes = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MI355_GPU_PROBLEM, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(MI355_INVALID, "device index out of range");
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    HIP_TRY(hipSetDevice(device));
    size_t fr = 0, tot = 0;
    const hipError_t e = hipMemGetInfo(&fr, &tot);
    (void)hipSetDevice(cur);
    HIP_TRY(e);
    *total_bytes = (int64_t)tot; *free_bytes = (int64_t)fr;
    return MI355_OK;
}

int mi355_factor_create(int xtype, int64_t n, int64_t nsuper, const int64_t *Super,
                        const int64_t *Lpi, const int64_t *Lpx, const int64_t *Ls,
                        int64_t ssize, int64_t xsize, int device, mi355_factor **handle)
{
    return mi355_factor_create_limited(xtype, n, nsuper, Super, Lpi, Lpx, Ls, ssize, xsize, device, 0, 0.0, nullptr, handle);
}

int mi355_factor_create_limited(int xtype, int64_t n, int64_t nsuper, const int64_t *Super,
                                const int64_t *Lpi, const int64_t *Lpx, const int64_t *Ls,
                                int64_t ssize, int64_t xsize, int device, int64_t max_bytes, double max_fraction,
                                int64_t *needed_bytes, mi355_factor **handle)
{
    g_err.clear();
    if (needed_bytes) *needed_bytes = 0;
    if (!handle) return fail(MI355_INVALID, "null handle pointer");
    *handle = nullptr;
    if (max_bytes < 0 || !(max_fraction >= 0.0 && max_fraction <= 1.0)) return fail(MI355_INVALID, "memory limit out of range");
    if (xtype != MI355_REAL && xtype != MI355_COMPLEX)
        return fail(MI355_INVALID, "factor xtype must be MI355_REAL or MI355_COMPLEX (a zomplex A gives a complex L)");
    if (n < 0 || nsuper < 0 || ssize < 0 || xsize < 0) return fail(MI355_INVALID, "negative size");
    if (nsuper > 0 && (!Super || !Lpi || !Lpx || !Ls)) return fail(MI355_INVALID, "null symbolic array");
    // before anything of length n is read: the reference reports an index that overflows the BLAS integer as
    // CHOLMOD_TOO_LARGE (t_cholmod_super_numeric_worker.c:183-188, cholmod_super_numeric.c:202); here the narrow type is
    // the kernels' int32 row index
    if (n >= (int64_t(1) << 31) - 1)
        return fail(MI355_TOO_LARGE, "too large: n does not fit the 32-bit row indices of the device kernels");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MI355_GPU_PROBLEM, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(MI355_INVALID, "device index out of range");

    mi355_factor_s *F = new (std::nothrow) mi355_factor_s;
    if (!F) return fail(MI355_OUT_OF_MEMORY, "host out of memory");
    struct Guard { mi355_factor_s *f; ~Guard() { if (f) mi355_factor_free(f); } } guard{F};
    F->device = device;
    F->xtype = xtype;
    F->E = (xtype == MI355_COMPLEX) ? 2 : 1;
    HIP_TRY(hipSetDevice(device));
    {
        // the panel chain (main stream) must get its few workgroups in while a trailing update fills the chip
        int plo = 0, phi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&plo, &phi));
        HIP_TRY(hipStreamCreateWithPriority(&F->stream, hipStreamNonBlocking, phi));
        HIP_TRY(hipStreamCreateWithPriority(&F->side, hipStreamNonBlocking, plo));
        HIP_TRY(hipEventCreate(&F->ev0));
        HIP_TRY(hipEventCreate(&F->ev1));
    }

    PhaseClock pc;
    F->Super.assign(Super, Super + (nsuper > 0 ? nsuper + 1 : 0));
    F->Lpi.assign(Lpi, Lpi + (nsuper > 0 ? nsuper + 1 : 0));
    F->Lpx.assign(Lpx, Lpx + (nsuper > 0 ? nsuper + 1 : 0));
    const int64_t nls = nsuper > 0 ? Lpi[nsuper] : 0;
    if (nls < 0 || nls > ssize) return fail(MI355_INVALID, "Lpi[nsuper] exceeds ssize");
    F->Ls.assign(Ls, Ls + nls);
    F->sym.n = n; F->sym.nsuper = nsuper; F->sym.ssize = ssize; F->sym.xsize = xsize;
    F->sym.Super = F->Super.data(); F->sym.Lpi = F->Lpi.data();
    F->sym.Lpx = F->Lpx.data(); F->sym.Ls = F->Ls.data();
    const std::string bad = validate_symbolic(F->sym);
    if (bad.compare(0, 10, "too large:") == 0) return fail(MI355_TOO_LARGE, bad);
    if (!bad.empty()) return fail(MI355_INVALID, "invalid supernodal symbolic factor: " + bad);
    pc.lap("copy + validate_symbolic");

    F->sym.cplx = (F->E == 2);          // the big-tile kernel then covers 64 x 128 complex entries per tile
    build_schedule(F->sym, nsuper, 0, F->full.sched);
    pc.lap("build_schedule");
    {
        // Device memory of this handle, known before anything is allocated: the factor, the block inverses, the symbolic
        // arrays and the schedule's tables (+ the transient pair list of k_relmap).  The reference sizes its pool from
        // Common->maxGpuMemBytes / maxGpuMemFraction and leaves 50 MB to the driver (CHOLMOD/GPU/cholmod_gpu.c:446-489);
        // here the same knobs cap what a handle may take, and a factor that does not fit is refused as a whole instead
        // of failing half-way through its allocations.
        const Schedule &sc = F->full.sched;
        const double need = 8.0 * F->E * ((double)xsize + (double)sc.dinv_size) + 4.0 * (double)nls + 4.0 * (double)n
            + 24.0 * (double)(nsuper + 1) + 4.0 * (double)(nsuper + 2)
            + (double)sizeof(GemmTask) * sc.gemm.size() + (double)sizeof(PotrfTask) * sc.potrf.size()
            + 4.0 * sc.potrf_order.size() + (double)sizeof(Schedule::TileEntry) * sc.gemm_tile_tab.size()
            + (double)sizeof(PanelGroup) * sc.pgroups.size() + 4.0 * sc.pg_wg_task.size()
            + 4.0 * sc.solve_tile_task.size() + 4.0 * (double)sc.relmap_size + (double)sizeof(RelmapPair) * sc.pairs.size();
        if (needed_bytes) *needed_bytes = (int64_t)need;
        size_t fr = 0, tot = 0;
        HIP_TRY(hipMemGetInfo(&fr, &tot));
        double limit = (double)fr - 50.0 * 1024 * 1024;
        if (max_bytes > 0) limit = std::min(limit, (double)max_bytes);
        if (max_fraction > 0.0) limit = std::min(limit, max_fraction * (double)tot);
        if (need > limit) {
            char b[256];
            snprintf(b, sizeof b, "device memory: the factor needs %.0f MB, %.0f MB may be used (free %.0f MB, limits %lld bytes / fraction %g)",
                     need / 1048576.0, std::max(0.0, limit) / 1048576.0, (double)fr / 1048576.0, (long long)max_bytes, max_fraction);
            return fail(MI355_OUT_OF_MEMORY, b);
        }
    }

    // device copies of the symbolic arrays (row indices narrowed to int32)
    std::vector<int32_t> ls32((size_t)nls);
    for (int64_t i = 0; i < nls; i++) ls32[i] = (int32_t)F->Ls[i];
    HIP_TRY(F->dLs.upload(ls32.data(), ls32.size(), F->stream));
    HIP_TRY(F->dSupermap.upload(F->full.sched.supermap.data(), F->full.sched.supermap.size(), F->stream));
    HIP_TRY(F->dSuper.upload(F->Super.data(), F->Super.size(), F->stream));
    HIP_TRY(F->dLpi.upload(F->Lpi.data(), F->Lpi.size(), F->stream));
    HIP_TRY(F->dLpx.upload(F->Lpx.data(), F->Lpx.size(), F->stream));
    HIP_TRY(F->Lx.alloc((size_t)xsize * F->E));
    HIP_TRY(F->Dinv.alloc((size_t)F->full.sched.dinv_size * F->E));
    HIP_TRY(F->info.alloc((size_t)nsuper + 2));
    pc.lap("device arrays");
    int rc = upload_schedule(F, F->full);
    if (rc != MI355_OK) return rc;
    guard.f = nullptr;
    *handle = F;
    return MI355_OK;
}

int mi355_factor_free(mi355_factor *F)
{
    if (!F) return MI355_OK;
    (void)hipSetDevice(F->device);
    if (F->side) { (void)hipStreamSynchronize(F->side); (void)hipStreamDestroy(F->side); }
    if (F->stream) { (void)hipStreamSynchronize(F->stream); (void)hipStreamDestroy(F->stream); }
    for (hipEvent_t ev : F->side_ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : F->solve_ev) (void)hipEventDestroy(ev);
    if (F->copy_stream) { (void)hipStreamSynchronize(F->copy_stream); (void)hipStreamDestroy(F->copy_stream); }
    for (hipStream_t st : F->lane_stream) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (void *p : F->lane_pin) (void)hipHostFree(p);
    for (hipEvent_t ev : F->copy_ev) (void)hipEventDestroy(ev);
    if (F->ev0) (void)hipEventDestroy(F->ev0);
    if (F->ev1) (void)hipEventDestroy(F->ev1);
    if (F->timer) { F->timer->drop(); F->timer->destroy_pool(); delete F->timer; }
    delete F->part_local;
    delete F->part_top;
    delete F->part_dist;
    delete F;
    return MI355_OK;
}

// one cholmod_sparse (ncols columns, row indices below nrows; packed when nz == NULL; z != NULL: zomplex,
// interleaved on the way in as the zd_ worker does on the fly, worker :69-83) into device buffers
static int stage_csc(mi355_factor *F, int64_t ncols, int64_t nrows, const int64_t *p_, const int64_t *i_,
                     const int64_t *nz_, const double *x_, const double *z_, DevBuf<int64_t> &dp, DevBuf<int64_t> &di,
                     DevBuf<int64_t> &dnz, DevBuf<double> &dx, bool &unpacked)
{
    if (!p_ || !i_ || !x_) return fail(MI355_INVALID, "null matrix array");
    // extent of i/x actually referenced
    int64_t nzmax = 0;
    for (int64_t k = 0; k < ncols; k++) {
        const int64_t p = p_[k], pend = nz_ ? p + nz_[k] : p_[k + 1];
        if (p < 0 || pend < p) return fail(MI355_INVALID, "matrix column pointers invalid");
        nzmax = std::max(nzmax, pend);
    }
    if (!nz_) nzmax = std::max(nzmax, p_[ncols]);
    for (int64_t p = 0; p < nzmax; p++)
        if (i_[p] < 0 || i_[p] >= nrows) return fail(MI355_INVALID, "matrix row index out of range");
    HIP_TRY(dp.upload(p_, (size_t)ncols + 1, F->stream));
    HIP_TRY(di.upload(i_, (size_t)nzmax, F->stream));
    if (F->E == 2 && z_) {
        std::vector<double> z((size_t)nzmax * 2);
        for (int64_t p = 0; p < nzmax; p++) { z[2 * p] = x_[p]; z[2 * p + 1] = z_[p]; }
        HIP_TRY(dx.upload(z.data(), z.size(), F->stream));
    } else {
        if (F->E == 1 && z_) return fail(MI355_INVALID, "complex type mismatch: zomplex A with a real L");
        HIP_TRY(dx.upload(x_, (size_t)nzmax * F->E, F->stream));
    }
    unpacked = (nz_ != nullptr);
    if (nz_) HIP_TRY(dnz.upload(nz_, (size_t)ncols, F->stream));
    return MI355_OK;
}

static int set_matrix_impl(mi355_factor *F, const int64_t *Ap, const int64_t *Ai,
                           const int64_t *Anz, const double *Ax, const double *Az)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    const int64_t n = F->sym.n;
    F->have_A = false;
    F->not_posdef = false;
    F->A_unsym = false;
    if (n == 0) { F->have_A = true; return MI355_OK; }
    rc = stage_csc(F, n, n, Ap, Ai, Anz, Ax, Az, F->Ap, F->Ai, F->Anz, F->Ax, F->A_unpacked);
    if (rc) return rc;
    F->have_A = true;
    F->A_perm = false;
    return MI355_OK;
}

int mi355_factor_set_matrix_unsym(mi355_factor *F, int64_t ncol, const int64_t *Ap, const int64_t *Ai,
                                  const int64_t *Anz, const double *Ax, const double *Az,
                                  const int64_t *Fp, const int64_t *Fi, const int64_t *Fnz,
                                  const double *Fx, const double *Fz)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    const int64_t n = F->sym.n;
    F->have_A = false;
    F->not_posdef = false;
    if (ncol < 0) return fail(MI355_INVALID, "invalid dimensions");
    if ((Az == nullptr) != (Fz == nullptr)) return fail(MI355_INVALID, "A and F must have same xtype and dtype");
    if (n == 0) { F->have_A = true; F->A_unsym = false; return MI355_OK; }
    if (!Fp || !Fi || !Fx) return fail(MI355_INVALID, "F invalid");
    rc = stage_csc(F, ncol, n, Ap, Ai, Anz, Ax, Az, F->Ap, F->Ai, F->Anz, F->Ax, F->A_unpacked);
    if (rc) return rc;
    rc = stage_csc(F, n, ncol, Fp, Fi, Fnz, Fx, Fz, F->Fp, F->Fi, F->Fnz, F->Fx, F->F_unpacked);
    if (rc) return rc;
    F->A_unsym = true;
    F->A_perm = false;
    F->have_A = true;
    return MI355_OK;
}

int mi355_factor_set_matrix_perm(mi355_factor *F, const int64_t *Ap, const int64_t *Ai, const int64_t *Anz,
                                 const double *Ax, const int64_t *Perm)
{
    int rc = set_matrix_impl(F, Ap, Ai, Anz, Ax, nullptr);
    if (rc) return rc;
    F->have_A = false;
    const int64_t n = F->sym.n;
    std::vector<int64_t> iperm((size_t)n, -1);
    for (int64_t k = 0; k < n; k++) {
        const int64_t pk = Perm ? Perm[k] : k;
        if (pk < 0 || pk >= n || iperm[(size_t)pk] >= 0) return fail(MI355_INVALID, "Perm is not a permutation of 0..n-1");
        iperm[(size_t)pk] = k;
    }
    HIP_TRY(F->dIperm.upload(iperm.data(), (size_t)n, F->stream));
    F->A_perm = true;
    F->have_A = true;
    return MI355_OK;
}

int mi355_factor_set_matrix(mi355_factor *F, const int64_t *Ap, const int64_t *Ai,
                            const int64_t *Anz, const double *Ax)
{
    return set_matrix_impl(F, Ap, Ai, Anz, Ax, nullptr);
}

int mi355_factor_set_matrix_zomplex(mi355_factor *F, const int64_t *Ap, const int64_t *Ai,
                                    const int64_t *Anz, const double *Ax, const double *Az)
{
    if (!Az) return fail(MI355_INVALID, "null Az");
    return set_matrix_impl(F, Ap, Ai, Anz, Ax, Az);
}

}  // extern "C"

namespace {

// all of Lx to the host through the lanes (everything enqueued on the compute stream so far is waited for)
int copy_all_to_host(mi355_factor_s *F, double *host_Lx)
{
    int rc = make_lanes(F);
    if (rc) return rc;
    if (F->copy_ev.empty()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        F->copy_ev.push_back(e);
    }
    HIP_TRY(hipEventRecord(F->copy_ev[0], F->stream));
    try {
        HostCopier hc(F, host_Lx);
        if (!hc.started) return fail(MI355_OUT_OF_MEMORY, "cannot start the copy threads");
        rc = hc.hand_over_range(F->copy_ev[0], 0, F->sym.xsize);
        const int rc2 = hc.finish();
        return rc ? rc : rc2;
    } catch (const std::exception &e) {
        return fail(MI355_OUT_OF_MEMORY, std::string("cannot start the copy threads: ") + e.what());
    }
}

// the numeric factorization; host_Lx != NULL: L->x on the host is filled as well, chunk by chunk while the
// factorization runs (mi355_factor_numeric_host)
int numeric_impl(mi355_factor *F, double beta_re, int quick_return_if_not_posdef, int64_t *minor, int *status,
                 double *host_Lx)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!F->have_A) return fail(MI355_INVALID, "mi355_factor_set_matrix was not called");
    const Symbolic &S = F->sym;
    F->st_copy_early = 0;
    std::unique_ptr<HostCopier> hc;
    if (host_Lx && S.xsize > 0 && S.nsuper > 0) {
        rc = make_lanes(F);
        if (rc) return rc;
        if (F->copy_plan.empty()) {
            // chunks of 32 M entries (256 MB real): large enough for PCIe to run at its rate, small enough that the
            // bottom of the tree leaves while the top is still being factorized
            build_host_copy_plan(S, F->full.sched, std::max(1, tuning().host_chunk), F->copy_plan);
            const std::string bad = check_host_copy_plan(S, F->full.sched, F->copy_plan);
            if (!bad.empty()) { F->copy_plan.clear(); return fail(MI355_GPU_PROBLEM, "internal: " + bad); }
        }
        try {
            hc.reset(new HostCopier(F, host_Lx));
        } catch (const std::exception &e) {       // no thread to be had: nothing was started
            return fail(MI355_OUT_OF_MEMORY, std::string("cannot start the copy thread: ") + e.what());
        }
        if (!hc->started) { hc.reset(); return fail(MI355_OUT_OF_MEMORY, "cannot start the copy threads"); }
    }
    F->numeric_ok = false;
    F->part_ok = false; F->lx_epoch++;         // whatever a partitioned run left in Lx is overwritten from here on
    F->st_ngemm = F->st_npotrf = F->st_tiles = F->st_nbig = 0;
    if (S.n == 0 || S.nsuper == 0) {
        F->minor = S.n; F->numeric_ok = true; F->lx_epoch++;
        if (minor) *minor = S.n;
        if (status) *status = MI355_STATUS_OK;
        return MI355_OK;
    }
    if (!F->timer) F->timer = new EventTimer;
    EventTimer &tm = *F->timer;
    tm.drop();
    tm.on = F->profile != 0;
    tm.st = F->stream;
    const hipEvent_t e0 = F->ev0, e1 = F->ev1;
    HIP_TRY(hipEventRecord(e0, F->stream));

    rc = run_schedule(F, F->full, S.n, beta_re, tm, nullptr, true, 0, -1, hc.get());
    if (rc) return rc;
    int *first = F->info.p + S.nsuper;
    const int big = 0x7f7f7f7f;
    HIP_TRY(hipMemsetAsync(first, 0x7f, sizeof(int), F->stream));
    hipLaunchKernelGGL(k_first_info, dim3((unsigned)((S.nsuper + 255) / 256)), dim3(256), 0, F->stream,
                       S.nsuper, F->info.p, first);
    HIP_TRY(hipGetLastError());
    int sf = big;
    HIP_TRY(hipMemcpyAsync(&sf, first, sizeof(int), hipMemcpyDeviceToHost, F->stream));
    HIP_TRY(hipStreamSynchronize(F->stream));

    int st_out = MI355_STATUS_OK;
    int64_t minor_out = S.n;
    if (sf != big) {
        // not positive definite: worker :1084-1146.  Supernode sf holds the first bad pivot.
        int info = 0;
        HIP_TRY(hipMemcpy(&info, F->info.p + sf, sizeof(int), hipMemcpyDeviceToHost));
        st_out = MI355_STATUS_NOT_POSDEF;
        minor_out = S.Super[sf] + info - 1;
        const int64_t nscol2 = (info == 1 || quick_return_if_not_posdef) ? 0 : info - 1;
        DeviceSchedule trunc;
        build_schedule(S, sf, nscol2, trunc.sched);
        rc = upload_schedule(F, trunc);
        if (rc) return rc;
        const int64_t ncols_asm = (nscol2 > 0) ? S.Super[sf + 1] : S.Super[sf];
        rc = run_schedule(F, trunc, ncols_asm, beta_re, tm);
        if (rc) return rc;
        const int64_t nsrow = S.Lpi[sf + 1] - S.Lpi[sf];
        const int64_t tail = S.Lpx[sf] + nsrow * nscol2;
        if (tail < S.xsize)
            HIP_TRY(hipMemsetAsync(F->Lx.p + tail * F->E, 0, (size_t)(S.xsize - tail) * F->E * sizeof(double), F->stream));
        HIP_TRY(hipStreamSynchronize(F->stream));
    }
    if (hc) {
        rc = hc->finish();
        if (rc) return rc;
        if (sf != big) {
            // the rare failing case: what left early is not what the replay produced -- one plain copy of everything
            F->st_copy_early = 0;
            rc = copy_all_to_host(F, host_Lx);
            if (rc) return rc;
        }
    }
    HIP_TRY(hipEventRecord(e1, F->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    F->st_total = ms;
    tm.pending = tm.on;
    F->minor = minor_out;
    F->numeric_ok = (st_out == MI355_STATUS_OK); F->lx_epoch++;
    F->not_posdef = (st_out == MI355_STATUS_NOT_POSDEF);
    if (minor) *minor = minor_out;
    if (status) *status = st_out;
    return MI355_OK;
}

}  // namespace

extern "C" {

int mi355_factor_numeric(mi355_factor *F, double beta_re, int quick_return_if_not_posdef,
                         int64_t *minor, int *status)
{
    return numeric_impl(F, beta_re, quick_return_if_not_posdef, minor, status, nullptr);
}

int mi355_factor_numeric_host(mi355_factor *F, double beta_re, int quick_return_if_not_posdef,
                              double *Lx, int64_t *minor, int *status)
{
    if (!Lx && F && F->sym.xsize > 0) return fail(MI355_INVALID, "null Lx");
    return numeric_impl(F, beta_re, quick_return_if_not_posdef, minor, status, Lx);
}

int mi355_factor_host_copy_plan(mi355_factor *F, int64_t cap, int64_t *plan, int64_t *count)
{
    g_err.clear();
    if (!F || !count) return fail(MI355_INVALID, "null argument");
    if (F->copy_plan.empty() && F->sym.xsize > 0 && F->sym.nsuper > 0) {
        build_host_copy_plan(F->sym, F->full.sched, std::max(1, tuning().host_chunk), F->copy_plan);
        const std::string bad = check_host_copy_plan(F->sym, F->full.sched, F->copy_plan);
        if (!bad.empty()) { F->copy_plan.clear(); return fail(MI355_GPU_PROBLEM, "internal: " + bad); }
    }
    *count = (int64_t)F->copy_plan.size();
    if (!plan) return MI355_OK;
    if (cap < *count) return fail(MI355_INVALID, "plan buffer too small");
    // flops of the launches that precede each step (what the device has done when a chunk may leave)
    const Schedule &sc = F->full.sched;
    std::vector<double> before(sc.steps.size() + 1, 0.0);
    for (size_t si = 0; si < sc.steps.size(); si++) {
        const Step &st = sc.steps[si];
        double fl = 0;
        for (int64_t ti = st.first; ti < st.first + st.count; ti++) {
            if (st.kind == STEP_POTRF) {
                const double nb = sc.potrf[(size_t)ti].nb;
                fl += nb * nb * nb / 3.0;
            } else if (st.kind == STEP_PDIAG || st.kind == STEP_PTRSM) {
                const PanelGroup &g = sc.pgroups[(size_t)ti];
                const double nc = g.ncols;
                fl += st.kind == STEP_PDIAG ? nc * nc * nc / 3.0 : (double)g.below * nc * nc;
            } else {
                const GemmTask &t = sc.gemm[(size_t)ti];
                const double Mm = t.M, Nn = t.N, Kk = t.K;
                fl += (t.mode == GEMM_OVERWRITE) ? Mm * Nn * Kk : (Nn * Nn + 2.0 * (Mm - Nn) * Nn) * Kk;
            }
        }
        before[si + 1] = before[si] + fl;
    }
    for (size_t k = 0; k < F->copy_plan.size(); k++) {
        plan[4 * k] = F->copy_plan[k].lo; plan[4 * k + 1] = F->copy_plan[k].hi; plan[4 * k + 2] = F->copy_plan[k].ready_step;
        plan[4 * k + 3] = (int64_t)before[(size_t)F->copy_plan[k].ready_step];
    }
    return MI355_OK;
}

int mi355_factor_get_Lx(mi355_factor *F, double *Lx)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!Lx) return fail(MI355_INVALID, "null Lx");
    if (F->sym.xsize > 0) return copy_all_to_host(F, Lx);
    return MI355_OK;
}

// supernodal -> packed simplicial values (super_num_to_simplicial_num, CHOLMOD/Utility/t_cholmod_change_factor.c:875
// and t_cholmod_change_factor_3_template.c:13-112): one workgroup per column j = k1 + jj copies rows jj .. nsrow-1 of
// supernode column jj to out[Lp[j] ...]; LDL' (to_ll == 0): D(j,j) = L(j,j)^2 with a zero imaginary part and
// L(i,j) / L(j,j) below it, a column whose pivot is <= 0 left as it is (:54-72).
__global__ __launch_bounds__(256) void k_to_simplicial(int64_t n, int E, int to_ll, const int32_t *__restrict__ supermap,
                                                       const int64_t *__restrict__ Super, const int64_t *__restrict__ Lpi,
                                                       const int64_t *__restrict__ Lpx, const int64_t *__restrict__ Lp,
                                                       const double *__restrict__ Lx, double *__restrict__ out)
{
    const int64_t j = blockIdx.x;
    if (j >= n) return;
    const int s = supermap[j];
    const int64_t jj = j - Super[s], nsrow = Lpi[s + 1] - Lpi[s];
    const double *__restrict__ src = Lx + E * (Lpx[s] + jj + jj * nsrow);       // L(j,j)
    double *__restrict__ dst = out + E * Lp[j];
    const int64_t len = nsrow - jj;
    double ljj = 1.0;
    int64_t first = 0;
    if (!to_ll) {
        const double d = src[0];
        if (threadIdx.x == 0) {
            dst[0] = (d <= 0.0) ? d : d * d;
            if (E == 2) dst[1] = 0.0;
        }
        ljj = (d <= 0.0) ? 1.0 : d;
        first = 1;
    }
    for (int64_t i = first + threadIdx.x; i < len; i += 256) {
        if (E == 2) {
            dst[2 * i] = to_ll ? src[2 * i] : src[2 * i] / ljj;
            dst[2 * i + 1] = to_ll ? src[2 * i + 1] : src[2 * i + 1] / ljj;
        } else {
            dst[i] = to_ll ? src[i] : src[i] / ljj;
        }
    }
}

int mi355_factor_to_simplicial(mi355_factor *F, int to_ll, int64_t *Lp, int64_t *Li, int64_t *Lnz, double *Lx,
                               int64_t *lnz_out)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    const Symbolic &S = F->sym;
    const int64_t n = S.n;
    // pattern: pure symbolic work on the host copy of the supernodal structure
    std::vector<int64_t> lp((size_t)n + 1, 0);
    int64_t p = 0;
    for (int64_t s = 0; s < S.nsuper; s++) {
        const int64_t k1 = S.Super[s], nscol = S.Super[s + 1] - k1, psi = S.Lpi[s], nsrow = S.Lpi[s + 1] - psi;
        for (int64_t jj = 0; jj < nscol; jj++) {
            lp[(size_t)(k1 + jj)] = p;
            if (Lnz) Lnz[k1 + jj] = nsrow - jj;
            if (Li) for (int64_t ii = jj; ii < nsrow; ii++) Li[p + ii - jj] = S.Ls[psi + ii];
            p += nsrow - jj;
        }
    }
    lp[(size_t)n] = p;
    if (lnz_out) *lnz_out = p;
    if (Lp) memcpy(Lp, lp.data(), ((size_t)n + 1) * sizeof(int64_t));
    if (!Lx || n == 0) return MI355_OK;
    if (!F->numeric_ok) return fail(MI355_INVALID, "no positive-definite numeric factor is resident");
    if (n > 0x7fffffffLL) return fail(MI355_TOO_LARGE, "too many columns for one launch");
    DevBuf<int64_t> dLp;
    DevBuf<double> dOut;
    HIP_TRY(dLp.upload(lp.data(), (size_t)n + 1, F->stream));
    HIP_TRY(dOut.alloc((size_t)p * F->E));
    hipLaunchKernelGGL(k_to_simplicial, dim3((unsigned)n), dim3(256), 0, F->stream, n, F->E, to_ll, F->dSupermap.p,
                       F->dSuper.p, F->dLpi.p, F->dLpx.p, dLp.p, F->Lx.p, dOut.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(Lx, dOut.p, (size_t)p * F->E * sizeof(double), hipMemcpyDeviceToHost, F->stream));
    HIP_TRY(hipStreamSynchronize(F->stream));
    return MI355_OK;
}

// min / max / any-NaN of the diagonal of L, one partial triple per workgroup (cholmod_rcond,
// CHOLMOD/Cholesky/t_cholmod_rcond_worker.c:76-99: the real part of L(j,j) of every supernode column)
__global__ __launch_bounds__(256) void k_diag_minmax(int64_t n, int E, const int32_t *__restrict__ supermap,
                                                     const int64_t *__restrict__ Super, const int64_t *__restrict__ Lpi,
                                                     const int64_t *__restrict__ Lpx, const double *__restrict__ Lx,
                                                     double *__restrict__ part)
{
    __shared__ double smin[256], smax[256], snan[256];
    double lo = INFINITY, hi = -INFINITY, bad = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int s = supermap[k];
        const int64_t jj = k - Super[s], nsrow = Lpi[s + 1] - Lpi[s];
        const double v = Lx[E * (Lpx[s] + jj + jj * nsrow)];
        if (v != v) bad = 1.0;
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
    smin[threadIdx.x] = lo; smax[threadIdx.x] = hi; snan[threadIdx.x] = bad;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + h]);
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + h]);
            snan[threadIdx.x] = fmax(snan[threadIdx.x], snan[threadIdx.x + h]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = smin[0]; part[3 * blockIdx.x + 1] = smax[0]; part[3 * blockIdx.x + 2] = snan[0]; }
}

int mi355_factor_rcond(mi355_factor *F, double *rcond)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!rcond) return fail(MI355_INVALID, "null rcond");
    const Symbolic &S = F->sym;
    if (S.n == 0) { *rcond = 1.0; return MI355_OK; }                      // cholmod_rcond.c:73-76
    if (!F->numeric_ok) {
        if (F->not_posdef) { *rcond = 0.0; return MI355_OK; }             // L->minor < n, :77-80
        return fail(MI355_INVALID, "no numeric factor is resident");
    }
    const unsigned grid = (unsigned)std::min<int64_t>(1024, (S.n + 255) / 256);
    DevBuf<double> part;
    HIP_TRY(part.alloc((size_t)grid * 3));
    hipLaunchKernelGGL(k_diag_minmax, dim3(grid), dim3(256), 0, F->stream, S.n, F->E, F->dSupermap.p, F->dSuper.p,
                       F->dLpi.p, F->dLpx.p, F->Lx.p, part.p);
    HIP_TRY(hipGetLastError());
    std::vector<double> h((size_t)grid * 3);
    HIP_TRY(hipMemcpyAsync(h.data(), part.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, F->stream));
    HIP_TRY(hipStreamSynchronize(F->stream));
    double lo = INFINITY, hi = -INFINITY;
    for (unsigned b = 0; b < grid; b++) {
        if (h[3 * b + 2] != 0.0) { *rcond = 0.0; return MI355_OK; }       // a NaN on the diagonal
        lo = std::min(lo, h[3 * b]); hi = std::max(hi, h[3 * b + 1]);
    }
    const double r = lo / hi;
    *rcond = r * r;                                                        // LL': (lmin / lmax)^2
    return MI355_OK;
}

}  // extern "C"

// ---- mi355_factor_rcond_noalloc: the scratch of k_diag_minmax (one min / max / NaN triple per workgroup) lives in a
// file-scope device array instead of a hipMalloc'ed buffer.  Every GPU holds its own copy of the array at its own address,
// until the process ends.  Host code never uses the array's NAME as a pointer (on the host it is a stand-in, not a device
// address): the address on each device comes from hipGetSymbolAddress, once per device, and `ptr` is what the kernels get.
#define MI355_RCOND_SCRATCH_BLOCKS 1024
static __device__ __align__(256) double scratch[3 * MI355_RCOND_SCRATCH_BLOCKS];       // 24 KB per GPU

namespace {

constexpr int MAX_SCRATCH_DEVICES = 64;
double *g_scratch_ptr[MAX_SCRATCH_DEVICES] = {};
std::mutex g_scratch_mu;

// the address of `scratch` on `device` (which is current)
int scratch_on_device(int device, double **out)
{
    if (device < 0 || device >= MAX_SCRATCH_DEVICES) return fail(MI355_INVALID, "device index beyond the scratch table");
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        double *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(MI355_GPU_PROBLEM, "hipGetSymbolAddress returned no address for the scratch array");
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return MI355_OK;
}

// the partial triples of `blocks` workgroups -> (min / max)^2, 0 if a NaN was seen; fixed != 0: that value as it is
__global__ __launch_bounds__(256) void k_rcond_finish(int blocks, const double *__restrict__ part, double fixed, int use_fixed,
                                                      double *__restrict__ out)
{
    __shared__ double smin[256], smax[256], snan[256];
    double lo = INFINITY, hi = -INFINITY, bad = 0.0;
    for (int b = (int)threadIdx.x; b < blocks; b += 256) {
        lo = fmin(lo, part[3 * b]); hi = fmax(hi, part[3 * b + 1]); bad = fmax(bad, part[3 * b + 2]);
    }
    smin[threadIdx.x] = lo; smax[threadIdx.x] = hi; snan[threadIdx.x] = bad;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + h]);
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + h]);
            snan[threadIdx.x] = fmax(snan[threadIdx.x], snan[threadIdx.x + h]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double r = smin[0] / smax[0];
        out[0] = use_fixed ? fixed : (snan[0] != 0.0 ? 0.0 : r * r);
    }
}

}  // namespace

extern "C" {

int mi355_factor_rcond_noalloc(mi355_factor *F, double *rcond_dev)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!rcond_dev) return fail(MI355_INVALID, "null rcond");
    const Symbolic &S = F->sym;
    if (S.n > 0 && !F->numeric_ok && !F->not_posdef) return fail(MI355_INVALID, "no numeric factor is resident");
    double *ptr = nullptr;
    rc = scratch_on_device(F->device, &ptr);
    if (rc) return rc;
    const int64_t want = S.n > 0 ? (S.n + 255) / 256 : 1;
    const int blocks = (int)std::min<int64_t>(MI355_RCOND_SCRATCH_BLOCKS, want);        // a workgroup strides over more columns
    if ((size_t)blocks * 3 * sizeof(double) > sizeof(double) * 3 * MI355_RCOND_SCRATCH_BLOCKS)
        return fail(MI355_TOO_LARGE, "too large: the diagonal needs more partial results than the scratch array holds");
    if (S.n == 0 || !F->numeric_ok) {
        // cholmod_rcond.c:73-80: 1 for an empty matrix, 0 when the factorization stopped at a bad pivot
        hipLaunchKernelGGL(k_rcond_finish, dim3(1), dim3(256), 0, F->stream, 0, ptr, S.n == 0 ? 1.0 : 0.0, 1, rcond_dev);
    } else {
        // every slot the second kernel reads is written by the first (blocks triples): nothing of an earlier call is seen
        hipLaunchKernelGGL(k_diag_minmax, dim3((unsigned)blocks), dim3(256), 0, F->stream, S.n, F->E, F->dSupermap.p,
                           F->dSuper.p, F->dLpi.p, F->dLpx.p, F->Lx.p, ptr);
        hipLaunchKernelGGL(k_rcond_finish, dim3(1), dim3(256), 0, F->stream, blocks, ptr, 0.0, 0, rcond_dev);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(F->stream));         // rcond_dev is the caller's: complete on return
    return MI355_OK;
}

int mi355_super_numeric(mi355_factor *F, const int64_t *Ap, const int64_t *Ai, const int64_t *Anz,
                        const double *Ax, double beta_re, int quick_return_if_not_posdef,
                        double *Lx, int64_t *minor, int *status)
{
    int rc = mi355_factor_set_matrix(F, Ap, Ai, Anz, Ax);
    if (rc) return rc;
    return mi355_factor_numeric_host(F, beta_re, quick_return_if_not_posdef, Lx, minor, status);
}

int mi355_factor_set_partition(mi355_factor *F, const int8_t *local_mask, const int8_t *top_mask,
                               const int8_t *asm_mask)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!local_mask || !top_mask || !asm_mask) return fail(MI355_INVALID, "null mask");
    const Symbolic &S = F->sym;
    for (int64_t s = 0; s < S.nsuper; s++)
        if (local_mask[s] && top_mask[s]) return fail(MI355_INVALID, "a supernode is both local and top");
    delete F->part_local; F->part_local = nullptr;
    delete F->part_top; F->part_top = nullptr;
    delete F->part_dist; F->part_dist = nullptr;
    F->part_world = 0;
    F->reduce_plan.clear();
    F->top_mask_host.assign(top_mask, top_mask + S.nsuper);
    F->part_local = new DeviceSchedule;
    F->part_top = new DeviceSchedule;
    build_schedule(S, S.nsuper, 0, F->part_local->sched, local_mask);
    // the top's block inverses sit behind the own subtrees' in Dinv: the solve phases need both
    build_schedule(S, S.nsuper, 0, F->part_top->sched, top_mask, F->part_local->sched.dinv_size);
    F->part_ok = false; F->lx_epoch++;
    rc = upload_schedule(F, *F->part_local);
    if (rc) return rc;
    rc = upload_schedule(F, *F->part_top);
    if (rc) return rc;
    HIP_TRY(F->asm_mask.upload(asm_mask, (size_t)S.nsuper, F->stream));
    return MI355_OK;
}

int mi355_factor_numeric_phase(mi355_factor *F, int phase, double beta_re, int64_t *first_bad, int64_t *info_out)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!F->part_local || !F->part_top) return fail(MI355_INVALID, "mi355_factor_set_partition was not called");
    if (!F->have_A) return fail(MI355_INVALID, "mi355_factor_set_matrix was not called");
    if (phase != 0 && phase != 1) return fail(MI355_INVALID, "phase must be 0 (own subtrees) or 1 (top)");
    const Symbolic &S = F->sym;
    F->numeric_ok = false;
    F->part_ok = false; F->lx_epoch++;
    if (first_bad) *first_bad = -1;
    if (info_out) *info_out = 0;
    if (S.nsuper == 0) return MI355_OK;
    if (!F->timer) F->timer = new EventTimer;
    EventTimer &tm = *F->timer;
    if (phase == 0) { tm.drop(); F->st_ngemm = F->st_npotrf = F->st_tiles = F->st_nbig = 0; }
    tm.on = F->profile != 0;
    tm.st = F->stream;
    const hipEvent_t e0 = F->ev0, e1 = F->ev1;
    HIP_TRY(hipEventRecord(e0, F->stream));
    rc = run_schedule(F, phase == 0 ? *F->part_local : *F->part_top, S.n, beta_re, tm, F->asm_mask.p, phase == 0);
    if (rc) return rc;
    int *first = F->info.p + S.nsuper;
    const int big = 0x7f7f7f7f;
    HIP_TRY(hipMemsetAsync(first, 0x7f, sizeof(int), F->stream));
    hipLaunchKernelGGL(k_first_info, dim3((unsigned)((S.nsuper + 255) / 256)), dim3(256), 0, F->stream,
                       S.nsuper, F->info.p, first);
    HIP_TRY(hipGetLastError());
    int sf = big;
    HIP_TRY(hipMemcpyAsync(&sf, first, sizeof(int), hipMemcpyDeviceToHost, F->stream));
    HIP_TRY(hipEventRecord(e1, F->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    F->st_total = (phase == 0 ? 0.0 : F->st_total) + ms;
    tm.pending = tm.on;
    if (sf != big) {
        int info = 0;
        HIP_TRY(hipMemcpy(&info, F->info.p + sf, sizeof(int), hipMemcpyDeviceToHost));
        if (first_bad) *first_bad = sf;
        if (info_out) *info_out = info;
    } else if (phase == 1) {
        F->part_ok = true; F->lx_epoch++;      // the caller vouches for phase 0 (on every rank) having succeeded
    }
    return MI355_OK;
}

int mi355_factor_set_top_distribution(mi355_factor *F, int rank, int world)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!F->part_local || !F->part_top) return fail(MI355_INVALID, "mi355_factor_set_partition was not called");
    if (world < 1 || rank < 0 || rank >= world) return fail(MI355_INVALID, "rank/world out of range");
    const Symbolic &S = F->sym;
    delete F->part_dist; F->part_dist = nullptr;
    F->part_dist = new DeviceSchedule;
    DistTop dt; dt.rank = rank; dt.world = world;
    build_schedule(S, S.nsuper, 0, F->part_dist->sched, F->top_mask_host.data(), F->part_local->sched.dinv_size, &dt);
    if (F->part_dist->sched.dinv_size != F->part_top->sched.dinv_size)
        return fail(MI355_GPU_PROBLEM, "internal: Dinv slots of the distributed top differ from the replicated top's");
    F->part_ok = false; F->lx_epoch++;
    return upload_schedule(F, *F->part_dist);
}

int mi355_factor_top_plan(mi355_factor *F, int64_t cap, int64_t *plan, int64_t *count, int64_t *dinv_lo, int64_t *dinv_hi)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!F->part_dist) return fail(MI355_INVALID, "mi355_factor_set_top_distribution was not called");
    if (!count) return fail(MI355_INVALID, "null count");
    const Schedule &sc = F->part_dist->sched;
    *count = (int64_t)sc.bcasts.size();
    if (dinv_lo) *dinv_lo = F->part_local->sched.dinv_size;
    if (dinv_hi) *dinv_hi = sc.dinv_size;
    if (!plan) return MI355_OK;
    if (cap < *count) return fail(MI355_INVALID, "plan buffer too small");
    for (size_t i = 0; i < sc.bcasts.size(); i++) {
        plan[4 * i] = sc.bcasts[i].step; plan[4 * i + 1] = sc.bcasts[i].lo;
        plan[4 * i + 2] = sc.bcasts[i].count; plan[4 * i + 3] = sc.bcasts[i].root;
    }
    return MI355_OK;
}

int mi355_factor_top_run(mi355_factor *F, int64_t seg, int64_t *first_bad, int64_t *info_out)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!F->part_dist) return fail(MI355_INVALID, "mi355_factor_set_top_distribution was not called");
    if (!F->have_A) return fail(MI355_INVALID, "mi355_factor_set_matrix was not called");
    const Symbolic &S = F->sym;
    const Schedule &sc = F->part_dist->sched;
    const int64_t nb = (int64_t)sc.bcasts.size();
    if (seg < 0 || seg > nb) return fail(MI355_INVALID, "segment out of range");
    if (first_bad) *first_bad = -1;
    if (info_out) *info_out = 0;
    F->part_ok = false; F->lx_epoch++;
    if (S.nsuper == 0) return MI355_OK;
    if (!F->timer) F->timer = new EventTimer;
    EventTimer &tm = *F->timer;
    tm.on = F->profile != 0;
    tm.st = F->stream;
    const hipEvent_t e0 = F->ev0, e1 = F->ev1;
    HIP_TRY(hipEventRecord(e0, F->stream));
    if (seg == 0) {
        // the block inverses of the blocks other ranks own stay zero: a sum over the ranks completes them
        const int64_t lo = F->part_local->sched.dinv_size, hi = sc.dinv_size;
        if (hi > lo) HIP_TRY(hipMemsetAsync(F->Dinv.p + lo * F->E, 0, (size_t)(hi - lo) * F->E * sizeof(double), F->stream));
    }
    const int64_t step0 = seg == 0 ? 0 : sc.bcasts[(size_t)seg - 1].step;
    const int64_t step1 = seg == nb ? (int64_t)sc.steps.size() : sc.bcasts[(size_t)seg].step;
    rc = run_schedule(F, *F->part_dist, 0, 0.0, tm, nullptr, false, step0, step1);
    if (rc) return rc;
    int sf = 0x7f7f7f7f;
    const int big = 0x7f7f7f7f;
    if (seg == nb) {
        int *first = F->info.p + S.nsuper;
        HIP_TRY(hipMemsetAsync(first, 0x7f, sizeof(int), F->stream));
        hipLaunchKernelGGL(k_first_info, dim3((unsigned)((S.nsuper + 255) / 256)), dim3(256), 0, F->stream,
                           S.nsuper, F->info.p, first);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&sf, first, sizeof(int), hipMemcpyDeviceToHost, F->stream));
    }
    HIP_TRY(hipEventRecord(e1, F->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    F->st_total += ms;
    tm.pending = tm.on;
    if (seg == nb) {
        if (sf != big) {
            int info = 0;
            HIP_TRY(hipMemcpy(&info, F->info.p + sf, sizeof(int), hipMemcpyDeviceToHost));
            if (first_bad) *first_bad = sf;
            if (info_out) *info_out = info;
        } else {
            F->part_ok = true; F->lx_epoch++;  // the caller vouches for every rank having succeeded (and for the Dinv sum)
        }
    }
    return MI355_OK;
}

double *mi355_factor_device_Dinv(mi355_factor *F) { return F ? F->Dinv.p : nullptr; }

// the grouped sweep's work lists (schedule.hpp: SolveSchedule) of one schedule, built and uploaded on first use
static int ensure_solve_groups(mi355_factor *F, DeviceSchedule &DS)
{
    if (DS.solve_groups_ready) return MI355_OK;
    build_solve_groups(DS.sched, DS.ssched, F->E == 2 ? 128 : 256);
    {
        const Schedule none;                           // tasks already checked at upload: only the solve work lists here
        const std::string bad = check_schedule_bounds(F->sym, none, &DS.ssched,
                                                      std::max<int64_t>(DS.sched.dinv_size, F->full.sched.dinv_size));
        if (!bad.empty()) return fail(MI355_GPU_PROBLEM, "internal: solve work lists rejected before upload: " + bad);
    }
    HIP_TRY(DS.sgroups.upload(DS.ssched.groups.data(), DS.ssched.groups.size(), F->stream));
    HIP_TRY(DS.sutasks.upload(DS.ssched.utasks.data(), DS.ssched.utasks.size(), F->stream));
    HIP_TRY(DS.stmap.upload(DS.ssched.tile_task.data(), DS.ssched.tile_task.size(), F->stream));
    HIP_TRY(DS.sranges.upload(DS.ssched.ranges.data(), DS.ssched.ranges.size(), F->stream));
    HIP_TRY(DS.sranges_mma.upload(DS.ssched.ranges_mma.data(), DS.ssched.ranges_mma.size(), F->stream));
    HIP_TRY(DS.snutasks.upload(DS.ssched.nutasks.data(), DS.ssched.nutasks.size(), F->stream));
    HIP_TRY(DS.sntmap.upload(DS.ssched.ntile_task.data(), DS.ssched.ntile_task.size(), F->stream));
    HIP_TRY(DS.snranges_mma.upload(DS.ssched.nranges_mma.data(), DS.ssched.nranges_mma.size(), F->stream));
    HIP_TRY(DS.swtasks.upload(DS.ssched.wtasks.data(), DS.ssched.wtasks.size(), F->stream));
    HIP_TRY(DS.swranges.upload(DS.ssched.wranges.data(), DS.ssched.wranges.size(), F->stream));
    HIP_TRY(DS.swwg.upload(DS.ssched.wwg.data(), DS.ssched.wwg.size(), F->stream));
    DS.solve_groups_ready = true;
    return MI355_OK;
}

// Tuning key 31: the groups of the steps with at most that many groups get the explicit inverse of their diagonal block, so that
// their step of the sweeps is one product (k_solve_group_inv) instead of a chain of ten.  Lists and memory once per schedule
// (out of memory: the chain kernels stay, silently); the images again after every factorization, by the forward chain itself on
// the identity as right-hand sides.
static int ensure_group_inverses(mi355_factor *F, DeviceSchedule &DS, int nrr)
{
    constexpr int64_t W = SOLVE_GROUP * NB;
    if (DS.ginv_off_for_good) return MI355_OK;
    {
        // scratch tiles and tickets for the widest step this call can launch: solve_group_inv groups x slabs of nrr
        const int64_t slots = (int64_t)tuning().solve_group_inv * ((nrr + SG_CW - 1) / SG_CW);
        if (slots > DS.ginv_slots) {
            DS.ginv_slots = 0;
            if (DS.ginv_scratch.alloc((size_t)slots * W * SG_CW) != hipSuccess || DS.ginv_tickets.alloc((size_t)slots) != hipSuccess) {
                (void)hipGetLastError();
                DS.ginv_scratch.release(); DS.ginv_tickets.release();
                DS.ginv_off_for_good = true;
                return MI355_OK;
            }
            HIP_TRY(hipMemsetAsync(DS.ginv_scratch.p, 0, (size_t)slots * W * SG_CW * sizeof(double), F->stream));
            HIP_TRY(hipMemsetAsync(DS.ginv_tickets.p, 0, (size_t)slots * sizeof(int), F->stream));
            DS.ginv_slots = slots;
        }
    }
    if (!DS.ginv_ready) {
        const SolveSchedule &ss = DS.ssched;
        std::vector<int64_t> off(ss.groups.size(), -1);
        std::vector<SolveGroupTask> tasks;
        DS.step_inv.assign(ss.steps.size(), 0);
        int64_t rows = 0;
        for (size_t i = 0; i < ss.steps.size(); i++) {
            const SolveStep &st = ss.steps[i];
            if (st.gcount <= 0 || st.gcount > tuning().solve_group_inv || st.maxblk < 2) continue;
            DS.step_inv[i] = 1;
            for (int64_t g = st.gfirst; g < st.gfirst + st.gcount; g++) {
                SolveGroupTask t = ss.groups[(size_t)g];
                off[(size_t)g] = rows * W;
                t.x0 = (int32_t)rows;
                rows += t.ncols;
                tasks.push_back(t);
            }
        }
        if (tasks.empty() || rows > 0x7ffffff0LL) { DS.ginv_off_for_good = true; return MI355_OK; }
        if (DS.ginv.alloc((size_t)2 * rows * W * F->E) != hipSuccess) {
            (void)hipGetLastError();
            DS.ginv_off_for_good = true;
            return MI355_OK;
        }
        HIP_TRY(DS.ginv_off.upload(off.data(), off.size(), F->stream));
        HIP_TRY(DS.ginv_tasks.upload(tasks.data(), tasks.size(), F->stream));
        DS.ginv_rows = rows;
        DS.ginv_ready = true;
    }
    if (DS.ginv_epoch != F->lx_epoch) {
        const int64_t nt = (int64_t)DS.ginv_tasks.n;
        const size_t half = (size_t)DS.ginv_rows * W * F->E;        // doubles of one set of images
        HIP_TRY(hipMemsetAsync(DS.ginv.p, 0, 2 * half * sizeof(double), F->stream));
        const dim3 g((unsigned)nt, (unsigned)(W * F->E / SG_CW));
        if (F->E == 2) {
            hipLaunchKernelGGL(k_ginv_identity<2>, dim3((unsigned)nt), dim3(256), 0, F->stream, DS.ginv_tasks.p, nt, DS.ginv.p);
            hipLaunchKernelGGL(k_ginv_identity<2>, dim3((unsigned)nt), dim3(256), 0, F->stream, DS.ginv_tasks.p, nt, DS.ginv.p + half);
            hipLaunchKernelGGL((k_solve_group_mma<false, 2, SOLVE_GROUP>), g, dim3(256), 0, F->stream,
                               DS.ginv_tasks.p, F->Dinv.p, F->Lx.p, DS.ginv.p, W * 2, (int)(W * 2));
            hipLaunchKernelGGL((k_solve_group_mma<true, 2, SOLVE_GROUP>), g, dim3(256), 0, F->stream,
                               DS.ginv_tasks.p, F->Dinv.p, F->Lx.p, DS.ginv.p + half, W * 2, (int)(W * 2));
        } else {
            hipLaunchKernelGGL(k_ginv_identity<1>, dim3((unsigned)nt), dim3(256), 0, F->stream, DS.ginv_tasks.p, nt, DS.ginv.p);
            hipLaunchKernelGGL(k_ginv_identity<1>, dim3((unsigned)nt), dim3(256), 0, F->stream, DS.ginv_tasks.p, nt, DS.ginv.p + half);
            hipLaunchKernelGGL((k_solve_group_mma<false, 1, SOLVE_GROUP>), g, dim3(256), 0, F->stream,
                               DS.ginv_tasks.p, F->Dinv.p, F->Lx.p, DS.ginv.p, W, (int)W);
            hipLaunchKernelGGL((k_solve_group_mma<true, 1, SOLVE_GROUP>), g, dim3(256), 0, F->stream,
                               DS.ginv_tasks.p, F->Dinv.p, F->Lx.p, DS.ginv.p + half, W, (int)W);
        }
        HIP_TRY(hipGetLastError());
        DS.ginv_epoch = F->lx_epoch;
    }
    return MI355_OK;
}

// part == nullptr: the whole factor; else the sweeps over one masked schedule of a partitioned factor
// image (many right-hand sides only): 1 = the row-major image in F->Yrow already holds the right-hand sides (the sweep before left
// it there), 2 = leave the result in the image -- a forward + backward solve (sys A, LDLt) transposes once in and once out
static int solve_dev(mi355_factor *F, bool lower, int64_t nrhs, double *X, int64_t d, DeviceSchedule *part = nullptr, int image = 0)
{
    const Symbolic &S = F->sym;
    if (nrhs < 0 || d < S.n) return fail(MI355_INVALID, "X and L dimensions must match");
    if (S.n == 0 || nrhs == 0) return MI355_OK;
    if (!(part ? F->part_ok : F->numeric_ok))
        return fail(MI355_INVALID, "no positive-definite numeric factor is resident");
    if (nrhs > 0x3fffffff) return fail(MI355_TOO_LARGE, "nrhs too large");
    DeviceSchedule &DS = part ? *part : F->full;
    const std::vector<Step> &steps = DS.sched.steps;
    const PotrfTask *P = DS.potrf.p;
    const hipEvent_t e0 = F->ev0, e1 = F->ev1;
    HIP_TRY(hipEventRecord(e0, F->stream));
    const int nr = (int)nrhs;
    const bool cplx = F->E == 2;
    // many right-hand sides: the MFMA sweeps run on a row-major image (solve_kernels.hip.hpp); X and d then
    // name that image and its leading dimension until the transpose back
    double *const Xcol = X;
    const int64_t dcol = d;
    // measured crossover of the grouped scalar and MFMA sweeps: 5 right-hand sides (real), 3 (complex)
    // (with the wide sweeps a complex factor takes the MFMA path from one right-hand side on: their 16-column variants read
    // L at the rate of the scalar kernels of a real factor, which a complex factor has no grouped version of)
    const bool mma = nr >= (cplx ? std::min(tuning().solve_mma_min, (tuning().solve_wide && tuning().solve_group) ? 1 : tuning().solve_mma_min_c)
                                 : tuning().solve_mma_min);
    const int nrr = nr * F->E;                          // real columns of the row-major image
    if (!cplx && !mma && nr <= SOLVE_NR && tuning().solve_group) {
        // few right-hand sides: the sweep is a chain of dependent launches -- two per SOLVE_GROUP blocks
        { const int rcg = ensure_solve_groups(F, DS); if (rcg) return rcg; }
        const std::vector<SolveStep> &gs = DS.ssched.steps;
        // a column of X is its own row-major image (ldr = 1), so the steps of few groups take the explicit inverses here too
        // (k_solve_group_inv with one slab per column)
        const bool inv = tuning().solve_group_inv > 0;
        if (inv) { const int rci = ensure_group_inverses(F, DS, SG_CW * nr); if (rci) return rci; }      // (nr slabs of one column)
        for (size_t ii = 0; ii < gs.size(); ii++) {
            const SolveStep &st = gs[lower ? ii : gs.size() - 1 - ii];
            if (st.gcount <= 0) continue;
            if (st.gcount > 0x7fffffffLL || st.stiles > 0x7fffffffLL) return fail(MI355_TOO_LARGE, "too many solve tiles in one step");
            const size_t step_at = lower ? ii : gs.size() - 1 - ii;
            auto group_kernel = [&]() {
                const dim3 g((unsigned)st.gcount), b(256);
                const SolveGroupTask *gp = DS.sgroups.p + st.gfirst;
                if (inv && DS.ginv_ready && !DS.ginv_off_for_good && DS.step_inv[step_at] && st.gcount * nr <= DS.ginv_slots) {
                    const dim3 gi((unsigned)st.gcount, (unsigned)(10 * nr));
                    if (lower)
                        hipLaunchKernelGGL((k_solve_group_inv<false, 1>), gi, b, 0, F->stream, gp, DS.ginv_off.p + st.gfirst, DS.ginv.p,
                                           X, (int64_t)1, nr, DS.ginv_scratch.p, DS.ginv_tickets.p, 1, d);
                    else
                        hipLaunchKernelGGL((k_solve_group_inv<true, 1>), gi, b, 0, F->stream, gp, DS.ginv_off.p + st.gfirst,
                                           DS.ginv.p + (size_t)DS.ginv_rows * SOLVE_GROUP * NB, X, (int64_t)1, nr, DS.ginv_scratch.p, DS.ginv_tickets.p, 1, d);
                    return;
                }
#define MI355_SG(T, NRT, MB) hipLaunchKernelGGL((k_solve_group<T, NRT, MB>), g, b, 0, F->stream, gp, F->Dinv.p, F->Lx.p, X, d, nr)
#define MI355_SG_MB(T, NRT) do { if (st.maxblk <= 1) MI355_SG(T, NRT, 1); else if (st.maxblk <= 2) MI355_SG(T, NRT, 2); \
                                 else MI355_SG(T, NRT, SOLVE_GROUP); } while (0)
                if (lower) { if (nr == 1) MI355_SG_MB(false, 1); else MI355_SG_MB(false, SOLVE_NR); }
                else       { if (nr == 1) MI355_SG_MB(true, 1); else MI355_SG_MB(true, SOLVE_NR); }
#undef MI355_SG_MB
#undef MI355_SG
            };
            if (lower) {
                group_kernel();
                if (st.stiles > 0)
                    hipLaunchKernelGGL(k_solve_fwd_update, dim3((unsigned)st.stiles), dim3(256), 0, F->stream,
                                       DS.sutasks.p + st.ufirst, DS.stmap.p + st.tmap_off, F->Lx.p, F->dLs.p, X, d, nr);
            } else {
                if (st.rcount > 0) {
                    if (st.rcount > 0x7fffffffLL) return fail(MI355_TOO_LARGE, "too many solve runs in one step");
                    if (nr == 1)
                        hipLaunchKernelGGL(k_solve_bwd_range<1>, dim3((unsigned)st.rcount), dim3(256), 0, F->stream,
                                           DS.sutasks.p + st.ufirst, DS.sranges.p + st.rfirst, F->Lx.p, F->dLs.p, X, d, nr);
                    else
                        hipLaunchKernelGGL(k_solve_bwd_range<SOLVE_NR>, dim3((unsigned)st.rcount), dim3(256), 0, F->stream,
                                           DS.sutasks.p + st.ufirst, DS.sranges.p + st.rfirst, F->Lx.p, F->dLs.p, X, d, nr);
                }
                group_kernel();
            }
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(e1, F->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float gms = 0;
        HIP_TRY(hipEventElapsedTime(&gms, e0, e1));
        F->st_solve = gms;
        return MI355_OK;
    }
    if (mma) {
        const int64_t ldr = ((int64_t)nrr + 15) & ~(int64_t)15;
        const size_t need = (size_t)S.n * (size_t)ldr;
        if (need > F->yrow_cap) {
            if (image & 1) return fail(MI355_GPU_PROBLEM, "internal: no image of the right-hand sides to continue on");
            F->Yrow.release();
            F->yrow_cap = 0;
            HIP_TRY(F->Yrow.alloc(need));
            F->yrow_cap = need;
        }
        const unsigned tg = (unsigned)((S.n + 63) / 64);
        if (image & 1) {
            // the forward sweep left its result in the image
        } else if (cplx)
            hipLaunchKernelGGL(k_solve_transpose<double2>, dim3(tg), dim3(256), 0, F->stream, S.n, nr,
                               reinterpret_cast<double2 *>(Xcol), dcol, reinterpret_cast<double2 *>(F->Yrow.p), ldr / 2, 1);
        else
            hipLaunchKernelGGL(k_solve_transpose<double>, dim3(tg), dim3(256), 0, F->stream, S.n, nr,
                               Xcol, dcol, F->Yrow.p, ldr, 1);
        HIP_TRY(hipGetLastError());
        X = F->Yrow.p;
        d = ldr;
    }
    if (mma && tuning().solve_group) {
        // many right-hand sides, grouped: k_solve_group_mma per SOLVE_GROUP blocks + the MFMA updates of the rows below
        { const int rcg = ensure_solve_groups(F, DS); if (rcg) return rcg; }
        const std::vector<SolveStep> &gs = DS.ssched.steps;
        // (a partitioned factor too: part_ok says that every piece of Lx and Dinv this schedule reads is final, and whatever
        // assigns it bumps lx_epoch; the images belong to the schedule -- own subtrees and top have their own)
        const bool inv = tuning().solve_group_inv > 0;
        if (inv) { const int rci = ensure_group_inverses(F, DS, nrr); if (rci) return rci; }
        // Look-ahead (tuning key 30; off by default -- measured below).  The top supernodes are chains of single-group steps: group kernel (a handful of
        // workgroups, ~20 us of dependent products) -> update of every row below (the whole chip) -> next group ...  Only the
        // first SOLVE_GROUP tiles of that update are rows of the NEXT group; the rest is needed two groups later at the
        // earliest.  The schedule splits such a step (st.wsplit): the near tiles on the per-block lists (~10 us on a handful of
        // tiles), the rows beyond on the wide lists.  Within a run of consecutive groups of one supernode the near part stays
        // on F->stream (high priority) with the group kernels, and the far part goes to F->side, ordered by events:
        //   forward   G(p) -> near(p) -> G(p+1) ...   far(p) after G(p);  G(p+2) after far(p)
        //   backward  near(p) -> G(p) -> near(p+1) .. far(p) after G(p-2) (the run's start for p = 1);  G(p) after far(p)
        // (p = position in the run; backward runs go up the supernode, and their first step stays on one stream).  Every add
        // into X is atomic in these kernels, so the two parts of a step may overlap; a step outside the run joins the streams.
        // Measured (profiles/r03_solve_look_ahead.txt): the far parts do run beside the next group, but every link of the chain
        // -- group kernel ~20 us, near part ~15 us, whole update 20 .. 45 us -- is a latency floor of its own, and the events
        // add ~10 us between a group kernel and its near part: 150^3 with 64 right-hand sides 74.2 -> 72.3 ms, elasticity
        // 51.0 -> 49.0 ms, but Helmholtz 54.1 -> 57.0 ms and 8 right-hand sides 46.7 -> 48.6 ms.  Hence off.
        const bool look = tuning().solve_lookahead > 0 && tuning().solve_wide && DS.ssched.kmax > 0 && F->side != nullptr;
        if (look)
            while (F->solve_ev.size() < 9) {
                hipEvent_t ev;
                HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                F->solve_ev.push_back(ev);
            }
        hipEvent_t *const evG = look ? F->solve_ev.data() : nullptr, *const evU = look ? evG + 4 : nullptr;
        int run = -1;                       // position of the previous step in its run; -1: it was not a split step
        int64_t run_next = -1;              // a_off of the group that continues the run
        int32_t run_lda = 0;
        bool side_busy = false;             // forward: far parts on F->side that F->stream has not waited for
        int side_last = 0;
        for (size_t ii = 0; ii < gs.size(); ii++) {
            const SolveStep &st = gs[lower ? ii : gs.size() - 1 - ii];
            if (st.gcount <= 0) continue;
            if (st.gcount > 0x7fffffffLL || st.stiles > 0x7fffffffLL) return fail(MI355_TOO_LARGE, "too many solve tiles in one step");
            int pos = -1;
            if (look && st.wsplit > 0) {
                // consecutive groups of one supernode: same leading dimension, diagonal entries ncols * (lda + 1) apart
                const SolveGroupTask &hg = DS.ssched.groups[(size_t)st.gfirst];
                const int64_t len = (int64_t)hg.ncols * ((int64_t)hg.lda + 1);
                const bool cont = run >= 0 && hg.lda == run_lda && (lower ? hg.a_off : hg.a_off + len) == run_next;
                pos = cont ? run + 1 : 0;
                run_lda = hg.lda;
                run_next = lower ? hg.a_off + len : hg.a_off;
            }
            if (pos <= 0 && side_busy) {
                HIP_TRY(hipStreamWaitEvent(F->stream, evU[side_last], 0));
                side_busy = false;
            }
            run = pos;
            const size_t step_at = lower ? ii : gs.size() - 1 - ii;
            auto group_kernel = [&]() {
                if (inv && DS.ginv_ready && !DS.ginv_off_for_good && DS.step_inv[step_at] &&
                    st.gcount * ((nrr + SG_CW - 1) / SG_CW) <= DS.ginv_slots) {
                    const dim3 g((unsigned)st.gcount, (unsigned)(10 * ((nrr + SG_CW - 1) / SG_CW))), b(256);
#define MI355_SGI(T, EE) hipLaunchKernelGGL((k_solve_group_inv<T, EE>), g, b, 0, F->stream, DS.sgroups.p + st.gfirst, \
                                            DS.ginv_off.p + st.gfirst, DS.ginv.p + (T ? (size_t)DS.ginv_rows * SOLVE_GROUP * NB * F->E : 0), X, d, nrr, \
                                            DS.ginv_scratch.p, DS.ginv_tickets.p, SG_CW, (int64_t)SG_CW)
                    if (lower) { if (cplx) MI355_SGI(false, 2); else MI355_SGI(false, 1); }
                    else       { if (cplx) MI355_SGI(true, 2); else MI355_SGI(true, 1); }
#undef MI355_SGI
                    return;
                }
                // a step of one-block groups only (the leaf levels: 10^5 groups of a few columns) takes 64 real columns of Y per
                // workgroup instead of 16: a quarter of the workgroups, each with four times the products of its one step
                const int cw = (st.maxblk <= 1 && st.gcount >= 1024) ? 64 : SG_CW;
                const dim3 g((unsigned)st.gcount, (unsigned)((nrr + cw - 1) / cw)), b(256);
                const SolveGroupTask *gp = DS.sgroups.p + st.gfirst;
#define MI355_SGM(T, EE, MB) hipLaunchKernelGGL((k_solve_group_mma<T, EE, MB>), g, b, 0, F->stream, gp, F->Dinv.p, F->Lx.p, X, d, nrr)
#define MI355_SGM_MB(T, EE) do { if (cw == 64) hipLaunchKernelGGL((k_solve_group_mma<T, EE, 1, 64>), g, b, 0, F->stream, gp, F->Dinv.p, F->Lx.p, X, d, nrr); \
                                 else if (st.maxblk <= 1) MI355_SGM(T, EE, 1); else if (st.maxblk <= 2) MI355_SGM(T, EE, 2); \
                                 else MI355_SGM(T, EE, SOLVE_GROUP); } while (0)
                if (lower) { if (cplx) MI355_SGM_MB(false, 2); else MI355_SGM_MB(false, 1); }
                else       { if (cplx) MI355_SGM_MB(true, 2); else MI355_SGM_MB(true, 1); }
#undef MI355_SGM_MB
#undef MI355_SGM
            };
            // wide path (tuning key 23, on): groups of one block keep the per-block kernels (lists of their own), the rows
            // below a wider group take all its columns in one pass (k_solve_fwd_wide / k_solve_bwd_wide)
            const bool wide = tuning().solve_wide && DS.ssched.kmax > 0;
            if (wide && st.wrcount > 0) {
                // MI355_SOLVE_LOG=file: one line per wide launch -- sweep, workgroups, bytes of L it reads, tasks, widest task
                // (analysis runs: the lines are in launch order, as the kernel trace is)
                static FILE *slog = [] { const char *fn = getenv("MI355_SOLVE_LOG"); return fn ? fopen(fn, "w") : (FILE *)nullptr; }();
                if (slog) {
                    double bytes = 0; int64_t rows = 0;
                    for (int64_t u = 0; u < st.wcount; u++) {
                        const SolveGroupUpd &wt = DS.ssched.wtasks[(size_t)(st.wfirst + u)];
                        bytes += 8.0 * F->E * (double)wt.nrows * wt.ncols; rows = std::max<int64_t>(rows, wt.nrows);
                    }
                    fprintf(slog, "%s wgs %lld bytes %.0f tasks %lld maxrows %lld maxblk %d\n", lower ? "fwd" : "bwd",
                            (long long)st.wgcount, bytes, (long long)st.wcount, (long long)rows, st.wmaxblk);
                    fflush(slog);
                }
            }
            const int64_t ptiles = wide ? st.nstiles : st.stiles;
            const PotrfTask *ptasks = wide ? DS.snutasks.p + st.nufirst : DS.sutasks.p + st.ufirst;
            const int32_t *ptmap = wide ? DS.sntmap.p + st.ntmap_off : DS.stmap.p + st.tmap_off;
            const int group = (int)std::min<int64_t>(16, std::max<int64_t>(1, ptiles / std::max(1, tuning().solve_fwd_div_mma)));
            const unsigned grid = (unsigned)((ptiles + group - 1) / group);
            const bool few = nrr <= 16;                 // a few right-hand sides: the wide kernels' 16-column variants
            const int wnj = few ? 16 : 64 * F->E;       // real columns of Y per pass of the wide kernels
            int64_t wt64 = 0;                           // 64-row tiles of the step's wide tasks
            if (wide) for (int64_t u = 0; u < st.wcount; u++) wt64 += (DS.ssched.wtasks[(size_t)(st.wfirst + u)].nrows + 63) / 64;
            if (wide && st.wrcount > 0x7fffffffLL) return fail(MI355_TOO_LARGE, "too many solve runs in one step");
            // workgroups [w0, w0 + wn) of the step's wide launch on stream wq
#define MI355_WIDE_T(K, NT) hipLaunchKernelGGL(K, dim3((unsigned)wn), dim3(NT), 0, wq, \
                                         DS.swtasks.p + st.wfirst, DS.swranges.p + st.wrfirst, DS.swwg.p + st.wgfirst + w0, \
                                         F->Lx.p, F->dLs.p, X, d, nrr, j0)
#define MI355_WIDE(K) MI355_WIDE_T(K, WIDE_THREADS)
            auto wide_fwd = [&](hipStream_t wq, int64_t w0, int64_t wn) {
                if (wide && st.wrcount > 0 && wn > 0)
                    for (int j0 = 0; j0 < nrr; j0 += wnj) {
                        // real factor: 16 waves per workgroup (4 per SIMD: the kernel fits 128 registers) once the shares are
                        // long enough to give every wave a slab (tuning key 25)
                        const bool big = tuning().solve_wide_1024 && st.wgcount > 0 && wt64 >= 4 * st.wgcount;
                        if (few) {
                            // at most 16 real columns: one j-block per workgroup, an eighth of the products
                            if (cplx) MI355_WIDE((k_solve_fwd_wide<2, 2, WIDE_THREADS, 1>));
                            else if (st.wmaxblk <= 2) MI355_WIDE((k_solve_fwd_wide<1, 2, WIDE_THREADS, 1>));
                            else MI355_WIDE((k_solve_fwd_wide<1, 4, WIDE_THREADS, 1>));
                        }
                        else if (cplx) MI355_WIDE((k_solve_fwd_wide<2, 2, WIDE_THREADS>));
                        else if (st.wmaxblk <= 2) MI355_WIDE((k_solve_fwd_wide<1, 2, WIDE_THREADS>));
                        else if (big) MI355_WIDE_T((k_solve_fwd_wide<1, 4, 1024>), 1024);
                        else if (tuning().solve_wide_ring >= 3) MI355_WIDE((k_solve_fwd_wide<1, 4, WIDE_THREADS, 4, 3>));
                        else MI355_WIDE((k_solve_fwd_wide<1, 4, WIDE_THREADS>));
                    }
            };
            auto wide_bwd = [&](hipStream_t wq, int64_t w0, int64_t wn) {
                if (wide && st.wrcount > 0 && wn > 0)
                    for (int j0 = 0; j0 < nrr; j0 += wnj) {
                        if (few) {
                            if (cplx) MI355_WIDE((k_solve_bwd_wide<2, 2, 1>));
                            else if (st.wmaxblk <= 2) MI355_WIDE((k_solve_bwd_wide<1, 2, 1>));
                            else MI355_WIDE((k_solve_bwd_wide<1, 4, 1>));
                        }
                        else if (cplx) MI355_WIDE((k_solve_bwd_wide<2, 2>));
                        else if (st.wmaxblk <= 2) MI355_WIDE((k_solve_bwd_wide<1, 2>));
                        else MI355_WIDE((k_solve_bwd_wide<1, 4>));
                    }
            };
#undef MI355_WIDE
#undef MI355_WIDE_T
            if (lower) {
                if (pos >= 2) HIP_TRY(hipStreamWaitEvent(F->stream, evU[(pos - 2) & 3], 0));
                group_kernel();
                if (pos >= 0) {
                    HIP_TRY(hipEventRecord(evG[pos & 3], F->stream));
                    HIP_TRY(hipStreamWaitEvent(F->side, evG[pos & 3], 0));
                    wide_fwd(F->side, 0, st.wgcount);
                    HIP_TRY(hipEventRecord(evU[pos & 3], F->side));
                    side_busy = true; side_last = pos & 3;
                }
                if (ptiles > 0)
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipLaunchKernelGGL(cplx ? k_solve_fwd_mma<2> : k_solve_fwd_mma<1>, dim3(grid), dim3(256), 0, F->stream,
                                           ptasks, ptmap, ptiles, group, F->Lx.p, F->dLs.p, X, d, nrr, j0);
                if (pos < 0) wide_fwd(F->stream, 0, st.wgcount);
            } else {
                const int64_t prcount = wide ? st.nrmcount : st.rmcount;
                const SolveRange *pruns = wide ? DS.snranges_mma.p + st.nrmfirst : DS.sranges_mma.p + st.rmfirst;
                if (pos >= 1) {
                    HIP_TRY(hipStreamWaitEvent(F->side, pos >= 2 ? evG[(pos - 2) & 3] : F->solve_ev[8], 0));
                    wide_bwd(F->side, 0, st.wgcount);
                    HIP_TRY(hipEventRecord(evU[pos & 3], F->side));
                } else {
                    if (pos == 0) HIP_TRY(hipEventRecord(F->solve_ev[8], F->stream));
                    wide_bwd(F->stream, 0, st.wgcount);
                }
                if (prcount > 0) {
                    if (prcount > 0x7fffffffLL) return fail(MI355_TOO_LARGE, "too many solve runs in one step");
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipLaunchKernelGGL(cplx ? k_solve_bwd_mma_run<2> : k_solve_bwd_mma_run<1>, dim3((unsigned)prcount),
                                           dim3(256), 0, F->stream, ptasks, pruns, F->Lx.p, F->dLs.p, X, d, nrr, j0);
                }
                if (pos >= 1) HIP_TRY(hipStreamWaitEvent(F->stream, evU[pos & 3], 0));
                group_kernel();
                if (pos >= 0) HIP_TRY(hipEventRecord(evG[pos & 3], F->stream));
            }
            HIP_TRY(hipGetLastError());
        }
        if (side_busy) HIP_TRY(hipStreamWaitEvent(F->stream, evU[side_last], 0));
    } else if (lower) {
        for (size_t i = 0; i < steps.size(); i++) {
            const Step &st = steps[i];
            if (st.kind != STEP_POTRF || st.count <= 0) continue;
            if (mma && cplx)
                hipLaunchKernelGGL((k_solve_diag_mma<false, 2>), dim3((unsigned)st.count), dim3(256),
                                   0, F->stream, P + st.first, F->Dinv.p, X, d, nrr);
            else if (mma)
                hipLaunchKernelGGL((k_solve_diag_mma<false, 1>), dim3((unsigned)st.count), dim3(256),
                                   0, F->stream, P + st.first, F->Dinv.p, X, d, nrr);
            else
                hipLaunchKernelGGL(cplx ? k_solve_diag_c<false> : k_solve_diag<false>, dim3((unsigned)st.count), dim3(256),
                                   0, F->stream, P + st.first, F->Dinv.p, X, d, nr);
            if (st.stiles > 0) {
                if (mma) {
                    const int group = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / 1024));
                    const unsigned grid = (unsigned)((st.stiles + group - 1) / group);
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipLaunchKernelGGL(cplx ? k_solve_fwd_mma<2> : k_solve_fwd_mma<1>, dim3(grid), dim3(256), 0, F->stream,
                                           P + st.first, DS.solve_tmap.p + st.tmap_off, st.stiles, group,
                                           F->Lx.p, F->dLs.p, X, d, nrr, j0);
                } else {
                    hipLaunchKernelGGL(cplx ? k_solve_fwd_update_c : k_solve_fwd_update, dim3((unsigned)st.stiles),
                                       dim3(cplx ? SOLVE_ROWS : 256), 0, F->stream,
                                       P + st.first, DS.solve_tmap.p + st.tmap_off, F->Lx.p, F->dLs.p, X, d, nr);
                }
            }
            HIP_TRY(hipGetLastError());
        }
    } else {
        for (size_t i = steps.size(); i-- > 0;) {
            const Step &st = steps[i];
            if (st.kind != STEP_POTRF || st.count <= 0) continue;
            if (st.stiles > 0) {
                if (mma) {
                    const int group = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / 1024));
                    const unsigned grid = (unsigned)((st.stiles + group - 1) / group);
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipLaunchKernelGGL(cplx ? k_solve_bwd_mma<2> : k_solve_bwd_mma<1>, dim3(grid), dim3(256), 0, F->stream,
                                           P + st.first, DS.solve_tmap.p + st.tmap_off, st.stiles, group,
                                           F->Lx.p, F->dLs.p, X, d, nrr, j0);
                } else if (cplx) {
                    hipLaunchKernelGGL(k_solve_bwd_update_c, dim3((unsigned)st.stiles), dim3(256), 0, F->stream,
                                       P + st.first, DS.solve_tmap.p + st.tmap_off, F->Lx.p, F->dLs.p, X, d, nr);
                } else if (nr <= SOLVE_NR) {
                    // a workgroup walks `group` table entries: fewer atomics into the same 64 x nrhs entries of a
                    // tall block, while at least ~128 workgroups stay per launch
                    const int group = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / 128));
                    const unsigned grid = (unsigned)((st.stiles + group - 1) / group);
                    if (nr == 1)
                        hipLaunchKernelGGL(k_solve_bwd_update<1>, dim3(grid), dim3(256), 0, F->stream,
                                           P + st.first, DS.solve_tmap.p + st.tmap_off, st.stiles, group,
                                           F->Lx.p, F->dLs.p, X, d, nr);
                    else
                        hipLaunchKernelGGL(k_solve_bwd_update<SOLVE_NR>, dim3(grid), dim3(256), 0, F->stream,
                                           P + st.first, DS.solve_tmap.p + st.tmap_off, st.stiles, group,
                                           F->Lx.p, F->dLs.p, X, d, nr);
                } else {
                    hipLaunchKernelGGL(k_solve_bwd_update<0>, dim3((unsigned)st.stiles), dim3(256), 0, F->stream,
                                       P + st.first, DS.solve_tmap.p + st.tmap_off, st.stiles, 1,
                                       F->Lx.p, F->dLs.p, X, d, nr);
                }
            }
            if (mma && cplx)
                hipLaunchKernelGGL((k_solve_diag_mma<true, 2>), dim3((unsigned)st.count), dim3(256),
                                   0, F->stream, P + st.first, F->Dinv.p, X, d, nrr);
            else if (mma)
                hipLaunchKernelGGL((k_solve_diag_mma<true, 1>), dim3((unsigned)st.count), dim3(256),
                                   0, F->stream, P + st.first, F->Dinv.p, X, d, nrr);
            else
                hipLaunchKernelGGL(cplx ? k_solve_diag_c<true> : k_solve_diag<true>, dim3((unsigned)st.count), dim3(256),
                                   0, F->stream, P + st.first, F->Dinv.p, X, d, nr);
            HIP_TRY(hipGetLastError());
        }
    }
    if (mma && !(image & 2)) {
        const unsigned tg = (unsigned)((S.n + 63) / 64);
        if (cplx)
            hipLaunchKernelGGL(k_solve_transpose<double2>, dim3(tg), dim3(256), 0, F->stream, S.n, nr,
                               reinterpret_cast<double2 *>(Xcol), dcol, reinterpret_cast<double2 *>(F->Yrow.p), d / 2, 0);
        else
            hipLaunchKernelGGL(k_solve_transpose<double>, dim3(tg), dim3(256), 0, F->stream, S.n, nr,
                               Xcol, dcol, F->Yrow.p, d, 0);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(e1, F->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    F->st_solve = ms;
    return MI355_OK;
}

static int solve_host(mi355_factor *F, bool lower, int64_t nrhs, double *X, int64_t d)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    const int64_t n = F->sym.n;
    if (nrhs < 0 || d < n) return fail(MI355_INVALID, "X and L dimensions must match");
    if (n == 0 || nrhs == 0) return MI355_OK;
    if (!X) return fail(MI355_INVALID, "null X");
    DevBuf<double> dX;
    const size_t es = (size_t)F->E * sizeof(double);        // bytes per entry
    HIP_TRY(dX.alloc((size_t)n * nrhs * F->E));
    HIP_TRY(hipMemcpy2DAsync(dX.p, n * es, X, d * es, n * es,
                             (size_t)nrhs, hipMemcpyHostToDevice, F->stream));
    rc = solve_dev(F, lower, nrhs, dX.p, n);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(X, d * es, dX.p, n * es, n * es,
                             (size_t)nrhs, hipMemcpyDeviceToHost, F->stream));
    HIP_TRY(hipStreamSynchronize(F->stream));
    return MI355_OK;
}

int mi355_super_lsolve(mi355_factor *F, int64_t nrhs, double *X, int64_t d) { return solve_host(F, true, nrhs, X, d); }
int mi355_super_ltsolve(mi355_factor *F, int64_t nrhs, double *X, int64_t d) { return solve_host(F, false, nrhs, X, d); }

int mi355_super_lsolve_dev(mi355_factor *F, int64_t nrhs, double *X, int64_t d)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    return solve_dev(F, true, nrhs, X, d);
}

int mi355_super_ltsolve_dev(mi355_factor *F, int64_t nrhs, double *X, int64_t d)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    return solve_dev(F, false, nrhs, X, d);
}

// zero the rows of X (n x nrhs entries of E doubles, leading dimension d) whose supernode is not kept
__global__ __launch_bounds__(256) void k_keep_rows(int64_t n, int nrhs, int E, const int32_t *__restrict__ supermap,
                                                   const int8_t *__restrict__ keep, double *__restrict__ X, int64_t d)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || keep[supermap[k]]) return;
    for (int j = 0; j < nrhs; j++)
        for (int e = 0; e < E; e++) X[(k + (int64_t)j * d) * E + e] = 0.0;
}

int mi355_super_solve_phase_dev(mi355_factor *F, int part, int lower, int64_t nrhs, double *X, int64_t d)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!F->part_local || !F->part_top) return fail(MI355_INVALID, "mi355_factor_set_partition was not called");
    if (part != 0 && part != 1) return fail(MI355_INVALID, "part must be 0 (own subtrees) or 1 (top)");
    if (!X && F->sym.n > 0 && nrhs > 0) return fail(MI355_INVALID, "null X");
    return solve_dev(F, lower != 0, nrhs, X, d, part == 0 ? F->part_local : F->part_top);
}

int mi355_solve_keep_rows_dev(mi355_factor *F, int64_t nrhs, double *X, int64_t d)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    if (!F->part_local || !F->part_top) return fail(MI355_INVALID, "mi355_factor_set_partition was not called");
    const int64_t n = F->sym.n;
    if (nrhs < 0 || d < n) return fail(MI355_INVALID, "X and L dimensions must match");
    if (n == 0 || nrhs == 0) return MI355_OK;
    if (!X) return fail(MI355_INVALID, "null X");
    if (nrhs > 0x3fffffff) return fail(MI355_TOO_LARGE, "nrhs too large");
    hipLaunchKernelGGL(k_keep_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, F->stream, n, (int)nrhs, F->E,
                       F->dSupermap.p, F->asm_mask.p, X, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(F->stream));
    return MI355_OK;
}

// Y(k,:) = B(Perm[k],:)  (gather: d_perm, CHOLMOD/Cholesky/cholmod_solve.c:689,
// t_cholmod_psolve_worker.c:40-240)   or   X(Perm[k],:) = Y(k,:)  (scatter: d_iperm, :712, worker :251-420).
// The permuted side P (B or X) has Ep doubles per entry and leading dimension ldp; the workspace side Y has Ey
// doubles per entry, ld ldy, or -- dual: a real L with a complex right-hand side -- two real columns per
// right-hand side (real part in column 2j, imaginary part in column 2j+1, worker :102-115).  A real B under a
// complex L gets a zero imaginary part.  perm == nullptr: identity.
__global__ __launch_bounds__(256) void k_permute_rows(int64_t n, int nrhs, const int64_t *__restrict__ perm,
                                                      const double *__restrict__ src, double *__restrict__ dst,
                                                      int64_t ldp, int Ep, int64_t ldy, int Ey, int dual, int gather)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t pk = perm ? perm[k] : k;
    for (int j = 0; j < nrhs; j++) {
        if (gather) {
            const double *b = src + (pk + (int64_t)j * ldp) * Ep;
            const double re = b[0], im = (Ep == 2) ? b[1] : 0.0;
            if (dual) {
                dst[k + (int64_t)(2 * j) * ldy] = re;
                dst[k + (int64_t)(2 * j + 1) * ldy] = im;
            } else {
                double *y = dst + (k + (int64_t)j * ldy) * Ey;
                y[0] = re;
                if (Ey == 2) y[1] = im;
            }
        } else {
            double re, im;
            if (dual) {
                re = src[k + (int64_t)(2 * j) * ldy];
                im = src[k + (int64_t)(2 * j + 1) * ldy];
            } else {
                const double *y = src + (k + (int64_t)j * ldy) * Ey;
                re = y[0];
                im = (Ey == 2) ? y[1] : 0.0;
            }
            double *x = dst + (pk + (int64_t)j * ldp) * Ep;
            x[0] = re;
            if (Ep == 2) x[1] = im;
        }
    }
}

int mi355_factor_set_perm(mi355_factor *F, const int64_t *Perm)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    F->dPerm.release();
    if (!Perm) return MI355_OK;
    const int64_t n = F->sym.n;
    std::vector<uint8_t> seen((size_t)n, 0);
    for (int64_t k = 0; k < n; k++) {
        const int64_t p = Perm[k];
        if (p < 0 || p >= n || seen[(size_t)p]) return fail(MI355_INVALID, "Perm is not a permutation of 0..n-1");
        seen[(size_t)p] = 1;
    }
    HIP_TRY(F->dPerm.upload(Perm, (size_t)n, F->stream));
    HIP_TRY(hipStreamSynchronize(F->stream));
    return MI355_OK;
}

int mi355_solve_x_dev(mi355_factor *F, int sys, int64_t nrhs, int bxtype, const double *B, int64_t ldb,
                      double *X, int64_t ldx)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    const int64_t n = F->sym.n;
    if (sys < MI355_SYS_A || sys > MI355_SYS_Pt) return fail(MI355_INVALID, "invalid system");
    if (bxtype != MI355_REAL && bxtype != MI355_COMPLEX) return fail(MI355_INVALID, "B must be real or complex (interleaved)");
    if (nrhs < 0 || ldb < n || ldx < n) return fail(MI355_INVALID, "dimensions of L, X and B do not match");
    if (n == 0 || nrhs == 0) return MI355_OK;
    if (!B || !X) return fail(MI355_INVALID, "null argument");
    if (nrhs > 0x3fffffff) return fail(MI355_TOO_LARGE, "nrhs too large");
    const int E = F->E, nr = (int)nrhs;
    const int Eb = (bxtype == MI355_COMPLEX) ? 2 : 1;
    const int Ex = (Eb == 2 || E == 2) ? 2 : 1;                 // xtype of X (cholmod_solve.c:228-235)
    const int dual = (E == 1 && Eb == 2) ? 1 : 0;               // real L, complex B (cholmod_solve.c:668)
    const unsigned grid = (unsigned)((n + 255) / 256);
    // Perm is used for A, P and Pt only (cholmod_solve.c:262-270); the other systems run unpermuted
    const bool use_perm = (sys == MI355_SYS_A || sys == MI355_SYS_P || sys == MI355_SYS_Pt);
    const int64_t *perm = use_perm ? F->dPerm.p : nullptr;
    if (B == X && (sys == MI355_SYS_P || sys == MI355_SYS_Pt || Ex != Eb))
        return fail(MI355_INVALID, "X must not alias B for this system");
    const int64_t nry = dual ? 2 * nrhs : nrhs;                 // columns of Y
    const size_t need = (size_t)n * (size_t)nry * (size_t)E;
    if (need > F->ywork_cap) {
        F->Ywork.release();
        F->ywork_cap = 0;
        HIP_TRY(F->Ywork.alloc(need));
        F->ywork_cap = need;
    }
    double *Y = F->Ywork.p;
    if (sys == MI355_SYS_Pt) {
        // x = P'*b: no gather; b goes through Y unpermuted, then the scatter
        hipLaunchKernelGGL(k_permute_rows, dim3(grid), dim3(256), 0, F->stream, n, nr, (const int64_t *)nullptr,
                           B, Y, ldb, Eb, n, E, dual, 1);
    } else {
        hipLaunchKernelGGL(k_permute_rows, dim3(grid), dim3(256), 0, F->stream, n, nr, perm,
                           B, Y, ldb, Eb, n, E, dual, 1);                                   // Y = P*B
    }
    HIP_TRY(hipGetLastError());
    double ms = 0;
    const bool fwd = (sys == MI355_SYS_A || sys == MI355_SYS_LDLt || sys == MI355_SYS_L || sys == MI355_SYS_LD);
    const bool bwd = (sys == MI355_SYS_A || sys == MI355_SYS_LDLt || sys == MI355_SYS_Lt || sys == MI355_SYS_DLt);
    // (both sweeps: the row-major image of the many-RHS path stays in place between them; the flags mean nothing to the other paths)
    if (fwd) { rc = solve_dev(F, true, nry, Y, n, nullptr, bwd ? 2 : 0); if (rc) return rc; ms += F->st_solve; }
    if (bwd) { rc = solve_dev(F, false, nry, Y, n, nullptr, fwd ? 1 : 0); if (rc) return rc; ms += F->st_solve; }
    F->st_solve = ms;
    const int64_t *sperm = (sys == MI355_SYS_A || sys == MI355_SYS_Pt) ? perm : nullptr;     // X = P'*Y, or X = Y
    hipLaunchKernelGGL(k_permute_rows, dim3(grid), dim3(256), 0, F->stream, n, nr, sperm,
                       (const double *)Y, X, ldx, Ex, n, E, dual, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(F->stream));
    return MI355_OK;
}

int mi355_solve_x(mi355_factor *F, int sys, int64_t nrhs, int bxtype, const double *B, int64_t ldb,
                  double *X, int64_t ldx)
{
    g_err.clear();
    int rc = check_handle(F);
    if (rc) return rc;
    const int64_t n = F->sym.n;
    if (bxtype != MI355_REAL && bxtype != MI355_COMPLEX) return fail(MI355_INVALID, "B must be real or complex (interleaved)");
    if (nrhs < 0 || ldb < n || ldx < n) return fail(MI355_INVALID, "dimensions of L, X and B do not match");
    if (n == 0 || nrhs == 0) return MI355_OK;
    if (!B || !X) return fail(MI355_INVALID, "null argument");
    const int Eb = (bxtype == MI355_COMPLEX) ? 2 : 1, Ex = (Eb == 2 || F->E == 2) ? 2 : 1;
    const size_t bs = (size_t)Eb * sizeof(double), xs = (size_t)Ex * sizeof(double);
    // device images of B and X, kept by the handle from call to call (cholmod_solve2's Y / E workspaces play that part
    // in the reference, cholmod_solve.c:119-153)
    const size_t need_b = (size_t)n * nrhs * Eb, need_x = (size_t)n * nrhs * Ex;
    // alloc() releases the old buffer first: the recorded capacity goes to zero before the attempt and is set again only
    // on success, so that after an out-of-memory failure a later, smaller request allocates afresh (as Ywork does)
    if (F->solve_b_cap < need_b) { F->solve_b_cap = 0; HIP_TRY(F->solve_b.alloc(need_b)); F->solve_b_cap = need_b; }
    if (F->solve_x_cap < need_x) { F->solve_x_cap = 0; HIP_TRY(F->solve_x.alloc(need_x)); F->solve_x_cap = need_x; }
    double *dB = F->solve_b.p, *dX = F->solve_x.p;
    const bool staged = need_b * sizeof(double) >= ((size_t)std::max(0, tuning().host_staged_min_kb) << 10);   // small ones: not worth the threads
    if (staged) {
        // B and X are pageable host memory: through the lanes (pinned staging, parallel memcpy) at PCIe rate
        rc = make_lanes(F);                         // (sets the staging size the pieces are cut to)
        if (rc) return rc;
        std::vector<LanePiece> pieces;
        lane_pieces(F, (char *)const_cast<double *>(B), ldb * bs, (char *)dB, n * bs, n * bs, (size_t)nrhs, pieces);
        rc = lanes_transfer(F, pieces, true, nullptr);
        if (rc) return rc;
    } else {
        HIP_TRY(hipMemcpy2DAsync(dB, n * bs, B, ldb * bs, n * bs, (size_t)nrhs, hipMemcpyHostToDevice, F->stream));
    }
    rc = mi355_solve_x_dev(F, sys, nrhs, bxtype, dB, n, dX, n);
    if (rc) return rc;
    if (staged) {
        HIP_TRY(hipEventRecord(F->ev1, F->stream));
        std::vector<LanePiece> pieces;
        lane_pieces(F, (char *)X, ldx * xs, (char *)dX, n * xs, n * xs, (size_t)nrhs, pieces);
        return lanes_transfer(F, pieces, false, F->ev1);
    }
    HIP_TRY(hipMemcpy2DAsync(X, ldx * xs, dX, n * xs, n * xs, (size_t)nrhs, hipMemcpyDeviceToHost, F->stream));
    HIP_TRY(hipStreamSynchronize(F->stream));
    return MI355_OK;
}

int mi355_solve_dev(mi355_factor *F, int sys, int64_t nrhs, const double *B, int64_t ldb, double *X, int64_t ldx)
{
    if (!F) return fail(MI355_INVALID, "null factor handle");
    return mi355_solve_x_dev(F, sys, nrhs, F->E == 2 ? MI355_COMPLEX : MI355_REAL, B, ldb, X, ldx);
}

int mi355_solve(mi355_factor *F, int sys, int64_t nrhs, const double *B, int64_t ldb, double *X, int64_t ldx)
{
    if (!F) return fail(MI355_INVALID, "null factor handle");
    return mi355_solve_x(F, sys, nrhs, F->E == 2 ? MI355_COMPLEX : MI355_REAL, B, ldb, X, ldx);
}

double *mi355_factor_device_Lx(mi355_factor *F) { return F ? F->Lx.p : nullptr; }

int mi355_factor_set_profile(mi355_factor *F, int profile)
{
    if (!F) return fail(MI355_INVALID, "null factor handle");
    F->profile = profile;
    return MI355_OK;
}

// bare v_mfma_f64_16x16x4_f64 issue-rate probe: 4 independent accumulators per wave, operands in
// registers, 4 waves per workgroup (one per SIMD), 2 workgroups per CU
__global__ __launch_bounds__(256, 2) void k_mfma_rate(int iters, double *out)
{
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    v4d s = a0 + a1 + a2 + a3;
    if (s[0] + s[1] + s[2] + s[3] == 123.456) out[0] = s[0];
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[1] = (double)(c1 - c0) / (4.0 * iters);                 // shader cycles per MFMA (this wave)
        out[2] = (double)(c1 - c0) / (double)(r1 - r0) * 100.0;     // shader clock in MHz (realtime = 100 MHz)
    }
}

int mi355_set_tuning(int key, int value)
{
    if (key == 0) tuning().big_min_m = value;
    else if (key == 1) tuning().big_min_n = value;
    else if (key == 2) tuning().ob = value;
    else if (key == 3) (void)value;                 // retired (4x4x4 MFMA variant)
    else if (key == 4) tuning().supertile = value;
    else if (key == 5) tuning().xcd_deal = value;
    else if (key == 6) (void)value;                 // retired (start stagger experiment)
    else if (key == 7) tuning().ob2 = value;
    else if (key == 8) tuning().ob3 = value;
    else if (key == 9) tuning().solve_mma_min = value;
    else if (key == 10) tuning().solve_group = value;
    else if (key == 11) tuning().solve_bwd_div = std::max(1, value);
    else if (key == 12) tuning().lookahead = value;
    else if (key == 13) tuning().solve_bwd_div_mma = std::max(1, value);
    else if (key == 14) tuning().tiny_max = std::max(0, value);
    else if (key == 15) tuning().solve_fwd_div_mma = std::max(1, value);
    else if (key == 16) tuning().dist_lookahead = value ? 1 : 0;
    else if (key == 17) tuning().dist_pieces = value ? 1 : 0;
    else if (key == 18) tuning().host_chunk = std::max(1, value);
    else if (key == 19) tuning().dist_reduce_channel = value ? 1 : 0;
    else if (key == 20) tuning().host_lanes = std::max(1, value);
    else if (key == 21) tuning().host_lane_mb = std::max(1, value);
    else if (key == 22) tuning().host_staged_min_kb = std::max(0, value);
    else if (key == 23) tuning().solve_wide = value ? 1 : 0;
    else if (key == 24) tuning().solve_wide_wgs = std::max(1, value);
    else if (key == 25) tuning().solve_wide_1024 = value ? 1 : 0;
    else if (key == 26) tuning().fuse_potrf = value ? 1 : 0;
    else if (key == 27) tuning().deterministic = value ? 1 : 0;
    else if (key == 28) tuning().panel_group = std::max(0, value);
    else if (key == 29) tuning().solve_wide_ring = value >= 3 ? 3 : 2;
    else if (key == 30) tuning().solve_lookahead = std::max(0, std::min(value, 128));
    else if (key == 31) tuning().solve_group_inv = std::max(0, std::min(value, 4096));
    else if (key == 33) tuning().solve_wide_ksplit = value ? 1 : 0;
    else return fail(MI355_INVALID, "unknown tuning key");
    return MI355_OK;
}

// one GEMM_DIRECT_SUB task C(MxN lower trapezoid) -= A(MxK) A(0:N,:)^T on random data, `iters` launches;
// kernel: 0 = k_gemm64, 1 = k_gemm128.  Reports TFLOP/s of the useful (lower-trapezoid) flops.
int mi355_debug_gemm_rate(int device, int kernel, int M, int N, int K, int iters, double *tflops)
{
    g_err.clear();
    if (!tflops || M < N || N <= 0 || K <= 0 || iters <= 0) return fail(MI355_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const int64_t lda = M;
    std::vector<double> hA((size_t)lda * K);
    uint64_t x = 88172645463325252ull;
    for (auto &v : hA) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 - 0.5; }
    DevBuf<double> buf;
    HIP_TRY(buf.alloc((size_t)lda * K + (size_t)M * N));
    HIP_TRY(hipMemcpy(buf.p, hA.data(), hA.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(buf.p + (size_t)lda * K, 0, (size_t)M * N * sizeof(double)));
    GemmTask t;
    t.a_off = 0; t.b_off = 0; t.c_off = (int64_t)lda * K; t.rm_off = -1;
    t.lda = (int32_t)lda; t.ldb = (int32_t)lda; t.ldc = M; t.M = M; t.N = N; t.K = K;
    t.mode = GEMM_DIRECT_SUB; t.big = kernel; t.tile0 = 0; t.fuse_potrf = -1; t.pad = 0;
    DevBuf<GemmTask> dt;
    HIP_TRY(dt.upload(&t, 1, 0));
    const int64_t ntiles = gemm_tiles(M, N, t.mode, kernel ? BTILE : TILE);
    std::vector<Schedule::TileEntry> tab;
    {
        // same table builder as the factorization (super-tiles, XCD chunks): a one-task step
        std::vector<Schedule::TileEntry> seq;
        append_tiles(seq, 0, t, kernel ? BTILE : TILE, kernel ? BTILE : TILE);
        deal_tiles(seq, tab);
        const int64_t n = (int64_t)tab.size();
        if (n != ntiles) return fail(MI355_INVALID, "tile table size mismatch");
    }
    DevBuf<Schedule::TileEntry> dmap;
    HIP_TRY(dmap.upload(tab.data(), tab.size(), 0));
    DevBuf<unsigned long long> dstamps;                 // MI355_GEMM_STAMPS=file: per-tile phase clocks of the last launch
    const char *stamp_file = getenv("MI355_GEMM_STAMPS");
    if (stamp_file && kernel >= 1) HIP_TRY(dstamps.alloc((size_t)ntiles * 4));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int it = -2; it < iters; it++) {
        if (it == 0) HIP_TRY(hipEventRecord(e0, 0));
        if (kernel >= 1)
            hipLaunchKernelGGL(k_gemm128<false>, dim3((unsigned)ntiles), dim3(256), 0, 0, dt.p, dmap.p, buf.p, buf.p, (const int32_t *)nullptr,
                               dstamps.p, (const PotrfTask *)nullptr, (double *)nullptr, (int *)nullptr);
        else
            hipLaunchKernelGGL(k_gemm64, dim3((unsigned)ntiles), dim3(256), 0, 0, dt.p, dmap.p, buf.p, buf.p, (const int32_t *)nullptr);
    }
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const double flops = 2.0 * K * ((double)N * (N + 1) / 2 + (double)(M - N) * N) * iters;
    *tflops = flops / (ms * 1e-3) * 1e-12;
    if (dstamps.p) {
        std::vector<unsigned long long> hs((size_t)ntiles * 4);
        HIP_TRY(hipMemcpy(hs.data(), dstamps.p, hs.size() * sizeof(hs[0]), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(stamp_file, "w")) {
            for (int64_t b = 0; b < ntiles; b++)
                fprintf(f, "%lld %u %u %llu %llu %llu %llu\n", (long long)b, tab[(size_t)b].tmtn & 0xffffu, tab[(size_t)b].tmtn >> 16,
                        hs[4 * b], hs[4 * b + 1], hs[4 * b + 2], hs[4 * b + 3]);
            fclose(f);
        }
    }
    return MI355_OK;
}

int mi355_debug_mfma_f64_rate(int device, double *tflops)
{
    g_err.clear();
    if (!tflops) return fail(MI355_INVALID, "null argument");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    DevBuf<double> d;
    HIP_TRY(d.alloc(8));
    const int iters = 20000, grid = prop.multiProcessorCount * 2;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mfma_rate, dim3(grid), dim3(256), 0, 0, 100, d.p);       // warm-up
    HIP_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_mfma_rate, dim3(grid), dim3(256), 0, 0, iters, d.p);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 /*mfma per iter*/ * 2048.0;
    *tflops = flops / (ms * 1e-3) * 1e-12;
    double h[3] = {0, 0, 0};
    HIP_TRY(hipMemcpy(h, d.p, sizeof h, hipMemcpyDeviceToHost));
    tflops[1] = h[1];
    tflops[2] = h[2];
    return MI355_OK;
}

int mi355_factor_get_stats(mi355_factor *F, double *out)
{
    if (!F || !out) return fail(MI355_INVALID, "null argument");
    const Schedule &S = F->full.sched;
    if (F->timer && F->timer->pending) {
        (void)hipSetDevice(F->device);
        F->st_big = F->timer->collect(3);
        F->st_gemm = F->timer->collect(0) + F->st_big; F->st_potrf = F->timer->collect(1); F->st_asm = F->timer->collect(2);
        F->timer->pending = false;
    }
    out[0] = F->st_total; out[1] = F->st_gemm; out[2] = F->st_ngemm; out[3] = F->st_potrf;
    out[4] = F->st_npotrf; out[5] = F->st_asm; out[6] = S.flops_update; out[7] = S.flops_panel;
    out[8] = S.bytes_assembly; out[9] = (double)S.pairs.size(); out[10] = (double)S.nlevels;
    out[11] = (double)S.relmap_size; out[12] = (double)S.dinv_size; out[13] = F->st_solve;
    out[14] = F->st_tiles; out[15] = F->st_copy_early;
    out[16] = F->st_big; out[17] = F->st_nbig; out[18] = S.flops_big; out[19] = S.flops_small;
    out[20] = (double)S.tiles_big; out[21] = (double)S.tiles_small; out[23] = F->st_red_overlap;
    // useful k_gemm128 flops THIS handle executes per factorization: all of them, or its share of a partition
    out[22] = S.flops_big;
    if (F->part_local && F->part_top)
        out[22] = F->part_local->sched.flops_big + (F->part_dist ? F->part_dist->sched.flops_big : F->part_top->sched.flops_big);
    return MI355_OK;
}

}  // extern "C"

#include "multi.hip.hpp"
