// tests/helpers/hipcpu/hip/hip_runtime.h -- TEST INFRASTRUCTURE.
// A stand-in for <hip/hip_runtime.h> that lets g++ compile the product's kernel sources
// (suitesparse_amd/csrc/*.hip.hpp) UNCHANGED for execution on the CPU: one fiber of the launching thread per lane, 64 lanes per
// wave, blockDim.x / 64 waves per workgroup.  Only what those kernels use exists here:
//   threadIdx / blockIdx / blockDim, __shared__ (a function-local static: workgroups run one after the other),
//   __syncthreads (a barrier over the workgroup's threads), v_readlane / readfirstlane / __shfl / __shfl_xor and
//   v_mfma_f64_16x16x4_f64 (an exchange through per-wave arrays between two wave barriers), FP64 atomic add;
//   and the slice of the HIP runtime API the engine's host code calls (device memory = host memory, streams and
//   events = nothing, hipLaunchKernelGGL = hipcpu::launch), so that engine.hip itself builds as a CPU library with the
//   same C ABI (tests/helpers/libmi355_cpuemu.so, loaded by tests only).
// MFMA operand and result layout as the kernels assume it and the GPU runs confirmed it: lane l supplies
// A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; lane l, register r holds D[i = (l >> 4) + 4 r][j = l & 15].
// Never part of the product; the emulator (tests/helpers/sched_emu.cpp) is its only user.
#pragma once
#include <pthread.h>
#include <ucontext.h>
#include <stdint.h>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <vector>

namespace hipcpu {
struct Dim { unsigned x = 1, y = 1, z = 1; };
// The lanes of a workgroup are FIBERS (ucontext) of the launching host thread, resumed round-robin: a barrier is a
// counter and a generation number, a lane that waits yields to the next lane.  No kernel-level context switch and
// no thread creation is involved, which is what made one-host-thread-per-lane slow (a 256-lane barrier cost 256
// futex round trips on 8 cores).  Deterministic, too: the lanes always interleave in the same order.
struct Barrier { unsigned n = 0, arrived = 0, gen = 0; };
struct Wave { Barrier bar; int64_t slot[64]; double a[64], b[64]; };
struct Block { Barrier bar; Wave wave[16]; };
inline thread_local Dim threadIdx, blockIdx, blockDim, gridDim;
inline thread_local Block *blk = nullptr;   // null: the thread-by-thread mode (kernels without any cross-lane operation)
inline thread_local Wave *wav = nullptr;
inline thread_local int lane = 0;
inline void yield();                        // to the next lane of the launch (defined with the scheduler below)
inline void barrier_wait(Barrier *b)
{
    const unsigned g = b->gen;
    if (++b->arrived == b->n) { b->arrived = 0; b->gen = g + 1; return; }
    while (b->gen == g) yield();
}

inline void syncthreads() { barrier_wait(&blk->bar); }
inline int64_t exchange(int64_t v, int src)
{
    wav->slot[lane] = v;
    barrier_wait(&wav->bar);
    const int64_t r = wav->slot[src & 63];
    barrier_wait(&wav->bar);
    return r;
}
inline int readlane(int v, int src) { return (int)exchange(v, src); }
inline int readfirstlane(int v) { return (int)exchange(v, 0); }           // full waves only
template <class T> inline T shfl(T v, int src, int width = 64)
{
    static_assert(sizeof(T) <= 8, "shfl of at most 8 bytes");
    int64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    const int base = lane & ~(width - 1);
    bits = exchange(bits, base + (src & (width - 1)));
    T out;
    std::memcpy(&out, &bits, sizeof(T));
    return out;
}
template <class T> inline T shfl_xor(T v, int mask, int width = 64) { return shfl(v, (lane ^ mask), width); }

#ifdef __clang__
typedef double v4d_t __attribute__((ext_vector_type(4)));      // clang: the kernels' own vector types, swizzles included
#else
typedef double v4d_t __attribute__((vector_size(32)));
#endif
inline v4d_t mfma_f64_16x16x4(double a, double b, v4d_t c)
{
    wav->a[lane] = a; wav->b[lane] = b;
    barrier_wait(&wav->bar);
    const int j = lane & 15, l4 = lane >> 4;
    for (int r = 0; r < 4; r++) {
        const int i = l4 + 4 * r;
        double s = c[r];
        for (int k = 0; k < 4; k++) s += wav->a[k * 16 + i] * wav->b[k * 16 + j];
        c[r] = s;
    }
    barrier_wait(&wav->bar);
    return c;
}
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline double unsafeAtomicAdd(double *p, double v)
{
    // workgroups run one after the other, the threads of one concurrently: a CAS loop keeps the add atomic
    uint64_t *q = reinterpret_cast<uint64_t *>(p);
    uint64_t old = __atomic_load_n(q, __ATOMIC_RELAXED), want;
    double o;
    do { std::memcpy(&o, &old, 8); const double n = o + v; std::memcpy(&want, &n, 8); }
    while (!__atomic_compare_exchange_n(q, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return o;
}
inline int atomicMin(int *p, int v)
{
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline int d2lo(double d) { int64_t b; std::memcpy(&b, &d, 8); return (int)(b & 0xffffffffLL); }
inline int d2hi(double d) { int64_t b; std::memcpy(&b, &d, 8); return (int)(b >> 32); }
inline double hilo2d(int hi, int lo) { const int64_t b = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); double d; std::memcpy(&d, &b, 8); return d; }
inline double rsqrt(double d) { return 1.0 / std::sqrt(d); }

// kernels without any cross-lane operation or barrier: thread after thread, no host threads needed
template <class F> void launch_serial(int64_t grid, unsigned block, F call)
{
    blk = nullptr; wav = nullptr;
    blockDim.x = block; gridDim.x = (unsigned)grid;
    for (int64_t b = 0; b < grid; b++)
        for (unsigned t = 0; t < block; t++) { blockIdx.x = (unsigned)b; threadIdx.x = t; lane = (int)(t & 63); call(); }
}
// any kernel: `block` lanes (a multiple of 64) as fibers of the calling thread, each running its lane of workgroup
// 0, 1, 2, ... in turn.  One launch at a time per process: __shared__ arrays are function statics here, so launches
// from several host threads (one per simulated rank) take turns.
// AddressSanitizer must be told about every stack switch (tools/asan_cpuemu.sh builds the engine's CPU build with it:
// an out-of-bounds access of a kernel to "device" memory, which is malloc'ed host memory here, is then reported)
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPCPU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#endif
#endif
#if defined(__x86_64__)
// switch stacks: callee-saved registers pushed on the old stack, popped from the new one (System V x86-64); no system
// call, unlike swapcontext, which saves and restores the signal mask on every switch
extern "C" void hipcpu_switch(void **save_sp, void *new_sp);
__asm__(".text\n.weak hipcpu_switch\n.type hipcpu_switch,@function\nhipcpu_switch:\n"
        "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
        "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
        "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
        ".size hipcpu_switch, .-hipcpu_switch\n");
struct Context {
    void *sp = nullptr;
    void *fake = nullptr; const void *bottom = nullptr; size_t size = 0;       // AddressSanitizer's view of this stack
    void prepare(char *stack, size_t sz, void (*entry)())
    {
        bottom = stack; size = sz;
        uintptr_t top = ((uintptr_t)stack + sz) & ~(uintptr_t)15;
        void **w = (void **)top;
        *--w = nullptr;                 // where a caller's return address would be: entry() sees rsp = 8 mod 16
        *--w = (void *)entry;           // popped by the ret of hipcpu_switch
        for (int i = 0; i < 6; i++) *--w = nullptr;     // rbp rbx r12 r13 r14 r15
        sp = w;
    }
};
inline void raw_switch(Context &from, Context &to) { hipcpu_switch(&from.sp, to.sp); }
#else
struct Context {
    ucontext_t uc;
    void *fake = nullptr; const void *bottom = nullptr; size_t size = 0;
    void prepare(char *stack, size_t sz, void (*entry)())
    {
        bottom = stack; size = sz;
        getcontext(&uc);
        uc.uc_stack.ss_sp = stack; uc.uc_stack.ss_size = sz; uc.uc_link = nullptr;
        makecontext(&uc, entry, 0);
    }
};
inline void raw_switch(Context &from, Context &to) { swapcontext(&from.uc, &to.uc); }
#endif
inline void switch_to(Context &from, Context &to)
{
#ifdef HIPCPU_ASAN
    __sanitizer_start_switch_fiber(&from.fake, to.bottom, to.size);
    raw_switch(from, to);
    __sanitizer_finish_switch_fiber(from.fake, nullptr, nullptr);      // back on `from`'s stack
#else
    raw_switch(from, to);
#endif
}
struct Fiber {
    Context ctx;
    char *stack = nullptr;
    bool done = false;
    // the lane's view of the "registers" every kernel reads
    Dim tIdx, bIdx;
    Wave *wv = nullptr;
    int ln = 0;
};
struct Launch {
    std::vector<Fiber> fibers;
    Context home;
    int cur = -1;
    std::function<void()> body;
};
inline thread_local Launch *running = nullptr;
inline void yield()
{
    Launch *L = running;
    Fiber &f = L->fibers[(size_t)L->cur];
    f.tIdx = threadIdx; f.bIdx = blockIdx; f.wv = wav; f.ln = lane;
    switch_to(f.ctx, L->home);
    threadIdx = f.tIdx; blockIdx = f.bIdx; wav = f.wv; lane = f.ln;
}
inline void fiber_main()
{
    Launch *L = running;
#ifdef HIPCPU_ASAN
    // first time on this stack: tell the sanitizer, and learn the bounds of the launching thread's stack from it
    __sanitizer_finish_switch_fiber(nullptr, &L->home.bottom, &L->home.size);
#endif
    L->body();
    L->fibers[(size_t)L->cur].done = true;
#ifdef HIPCPU_ASAN
    __sanitizer_start_switch_fiber(nullptr, L->home.bottom, L->home.size);      // nullptr: this stack is finished
    raw_switch(L->fibers[(size_t)L->cur].ctx, L->home);
#else
    switch_to(L->fibers[(size_t)L->cur].ctx, L->home);              // never resumed
#endif
    __builtin_trap();
}
inline std::vector<char *> &stack_cache() { static thread_local std::vector<char *> c; return c; }
inline std::mutex launch_mu;
template <class F> void launch(int64_t grid, unsigned block, F call, unsigned grid_y = 1)
{
    std::lock_guard<std::mutex> one_at_a_time(launch_mu);
    constexpr size_t STACK = 256 * 1024;
    Block B;
    const unsigned nw = block / 64;
    B.bar.n = block;
    for (unsigned w = 0; w < nw; w++) B.wave[w].bar.n = 64;
    Launch L;
    L.fibers.resize(block);
    std::vector<char *> &cache = stack_cache();
    while (cache.size() < block) cache.push_back((char *)std::malloc(STACK));
    blk = &B; blockDim.x = block; gridDim.x = (unsigned)grid; gridDim.y = grid_y;
    L.body = [&] {
        for (unsigned by = 0; by < grid_y; by++)
            for (int64_t b = 0; b < grid; b++) {
                blockIdx.x = (unsigned)b; blockIdx.y = by;
                call();
                barrier_wait(&B.bar);   // the next workgroup reuses the __shared__ statics
            }
        blockIdx.y = 0;
    };
    for (unsigned t = 0; t < block; t++) {
        Fiber &f = L.fibers[t];
        f.stack = cache[t];
        f.ctx.prepare(f.stack, STACK, fiber_main);
        f.tIdx.x = t; f.wv = &B.wave[t / 64]; f.ln = (int)(t & 63);
    }
    Launch *outer = running;
    running = &L;
    // HIPCPU_LANE_ORDER: the order in which the lanes are resumed in every round -- 0 (default) ascending, 1 descending,
    // 2 a new random order every round.  A kernel that lacks a barrier between a write of one lane and a read of
    // another gives different results under different orders (the real machine runs the waves in any order)
    static const int order_mode = [] { const char *e = std::getenv("HIPCPU_LANE_ORDER"); return e ? std::atoi(e) : 0; }();
    std::vector<unsigned> order(block);
    for (unsigned t = 0; t < block; t++) order[t] = order_mode == 1 ? block - 1 - t : t;
    static thread_local uint64_t rng_state = 0x9e3779b97f4a7c15ull;
    unsigned left = block;
    while (left > 0) {
        if (order_mode == 2)
            for (unsigned t = block - 1; t > 0; t--) {
                rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
                std::swap(order[t], order[(size_t)(rng_state % (t + 1))]);
            }
        for (unsigned oi = 0; oi < block; oi++) {
            const unsigned t = order[oi];
            Fiber &f = L.fibers[t];
            if (f.done) continue;
            L.cur = (int)t;
            threadIdx = f.tIdx; blockIdx = f.bIdx; wav = f.wv; lane = f.ln;
            switch_to(L.home, f.ctx);
            if (f.done) left--;
        }
    }
    running = outer;
    blk = nullptr; wav = nullptr;
}
}  // namespace hipcpu

// HIP's global min / max overloads and double2
#include <type_traits>
template <class A, class B> inline typename std::common_type<A, B>::type min(A a, B b) { return b < a ? b : a; }
template <class A, class B> inline typename std::common_type<A, B>::type max(A a, B b) { return a < b ? b : a; }
struct double2 { double x, y; };
inline double2 make_double2(double x, double y) { double2 v; v.x = x; v.y = y; return v; }

using hipcpu::threadIdx; using hipcpu::blockIdx; using hipcpu::blockDim; using hipcpu::gridDim;
using hipcpu::unsafeAtomicAdd; using hipcpu::atomicAdd; using hipcpu::atomicMin; using hipcpu::rsqrt;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define MI355_WAVES_PER_EU(n)
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) ((const void *)&(x))
#define __shared__ static
#ifndef __clang__
#define ext_vector_type(n) vector_size((n) * 8)                // g++: no .x/.y swizzles (kernels_complex needs clang)
#endif
#define __syncthreads() hipcpu::syncthreads()
#define __builtin_amdgcn_readlane(v, l) hipcpu::readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) hipcpu::readfirstlane((v))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __threadfence() ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_memtime() (0ULL)
#define __builtin_amdgcn_s_memrealtime() (0ULL)
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) hipcpu::mfma_f64_16x16x4((a), (b), (c))
#define __shfl(...) hipcpu::shfl(__VA_ARGS__)
#define __shfl_xor(...) hipcpu::shfl_xor(__VA_ARGS__)
#define __double2loint(d) hipcpu::d2lo(d)
#define __double2hiint(d) hipcpu::d2hi(d)
#define __hiloint2double(hi, lo) hipcpu::hilo2d((hi), (lo))

// ---- the HIP runtime API the engine's host code uses.
// HIPCPU_STREAMS (environment, read once):
//   unset / "sync"   every call completes before it returns (the default: fastest, what the test suite runs)
//   "lazy"           streams are queues: work runs only when somebody waits for it (hipStreamSynchronize,
//                    hipEventSynchronize, hipFree; a hipStreamWaitEvent in a queue pulls the other stream up to the record
//                    it captured).  A missing fence or synchronisation reads memory nobody has written yet.
//   "greedy"         the same queues, but every wait first runs ALL other streams as far as they can go: work that a
//                    reader should have been protected from (write-after-read across streams) has run before it reads.
// Host memory is treated as pageable: a host-to-device copy takes its snapshot at the call, as the runtime does.
#include <chrono>
#include <cstdlib>
#include <atomic>
#include <map>
#include <deque>
#include <memory>
#include <tuple>
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
namespace hipcpu {
struct Stream;
struct StreamOp { std::function<void()> fn; Stream *dep = nullptr; uint64_t dep_seq = 0; std::function<bool()> ready; };
struct Stream { std::deque<StreamOp> q; uint64_t enq = 0, done = 0; bool running = false; };
inline int stream_mode()
{
    static const int m = [] {
        const char *e = std::getenv("HIPCPU_STREAMS");
        if (!e || !*e || !std::strcmp(e, "sync")) return 0;
        if (!std::strcmp(e, "lazy")) return 1;
        if (!std::strcmp(e, "greedy")) return 2;
        std::abort();
    }();
    return m;
}
inline std::recursive_mutex &rt_mu() { static std::recursive_mutex m; return m; }
// threads that are trying to get the lock: a thread that works through a queue lets them in between two operations (a
// plain unlock/lock pair would win the mutex back before a sleeping waiter has woken up)
inline std::atomic<int> &rt_waiters() { static std::atomic<int> n{0}; return n; }
struct RtLock {
    RtLock() { rt_waiters()++; rt_mu().lock(); rt_waiters()--; }
    ~RtLock() { rt_mu().unlock(); }
    RtLock(const RtLock &) = delete;
};
inline std::vector<Stream *> &all_streams() { static std::vector<Stream *> v; return v; }    // never shrinks: an event may name a destroyed stream
// run s until `seq` of its operations are done (lock held).  false: it cannot get there yet -- an operation waits for
// something another host thread has still to enqueue (a collective whose peers have not issued it); the caller lets go
// of the lock and tries again
inline bool run_until(Stream *s, uint64_t seq)
{
    while (s->done < seq && !s->q.empty()) {
        if (s->running) return false;       // somebody is inside this stream's head operation (a collective polling its peers)
        s->running = true;
        bool ok = true;
        {
            StreamOp &op = s->q.front();
            if (op.dep) ok = run_until(op.dep, op.dep_seq);
            if (ok && op.ready) ok = op.ready();
        }
        if (!ok) { s->running = false; return false; }
        StreamOp op = std::move(s->q.front());
        s->q.pop_front();
        if (op.fn) op.fn();
        s->running = false;
        s->done++;
    }
    return true;
}
inline void run_all_but(Stream *keep)
{
    for (size_t i = 0; i < all_streams().size(); i++) {
        Stream *s = all_streams()[i];
        if (s != keep && !s->running) (void)run_until(s, s->enq);
    }
}
inline void let_others_in()
{
    for (int spin = 0; spin < 200 && rt_waiters().load() > 0; spin++) std::this_thread::sleep_for(std::chrono::microseconds(20));
}
// called WITHOUT the lock: one queued operation at a time under it, so that another host thread (a copy thread, another
// rank) gets in between two operations -- as it would while a device works through its queue
inline void wait_for(Stream *s, uint64_t seq)
{
    if (!s) return;
    if (stream_mode() == 2) { RtLock lk; run_all_but(s); }
    for (;;) {
        bool moved;
        {
            RtLock lk;
            if (s->done >= seq || s->q.empty()) return;
            moved = run_until(s, s->done + 1);
        }
        if (moved) let_others_in();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
// hipDeviceSynchronize, hipFree: every stream of THIS device empty.  With one host thread per rank (tests) all ranks
// share the emulated device, so "this device" = the streams the calling thread created; waiting for a peer's streams
// would wait for collectives the caller itself has still to issue
inline std::vector<Stream *> &my_streams() { static thread_local std::vector<Stream *> v; return v; }
inline void wait_for_all()
{
    for (;;) {
        bool empty = true;
        {
            RtLock lk;
            if (stream_mode() == 2) run_all_but(nullptr);
            for (Stream *s : my_streams()) {
                if (!s->running) (void)run_until(s, s->enq);
                empty = empty && s->q.empty();
            }
        }
        if (empty) return;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
inline void push(Stream *s, std::function<void()> fn, Stream *dep = nullptr, uint64_t dep_seq = 0,
                 std::function<bool()> ready = nullptr)
{
    StreamOp op; op.fn = std::move(fn); op.dep = dep; op.dep_seq = dep_seq; op.ready = std::move(ready);
    s->q.push_back(std::move(op));
    s->enq++;
}
}  // namespace hipcpu
typedef hipcpu::Stream *hipStream_t;
struct hipcpuEvent { std::chrono::steady_clock::time_point t; hipcpu::Stream *s = nullptr; uint64_t seq = 0; };
typedef hipcpuEvent *hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount = 1; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
#define HIPCPU_LOCK hipcpu::RtLock hipcpu_lk_
// a null stream handle means "right now" (the engine creates its streams non-blocking: nothing orders them against it)
#define HIPCPU_DEFER(s) (hipcpu::stream_mode() != 0 && (s) != nullptr)
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "hipcpu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { std::strcpy(p->gcnArchName, "gfx950:cpu-emulation"); return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int)
{
    HIPCPU_LOCK;
    *s = new hipcpu::Stream;
    hipcpu::all_streams().push_back(*s);
    hipcpu::my_streams().push_back(*s);
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreateWithPriority(s, 0, 0); }
inline hipError_t hipStreamSynchronize(hipStream_t s)
{
    if (hipcpu::stream_mode() == 0 || !s) return hipSuccess;
    uint64_t upto;
    { HIPCPU_LOCK; upto = s->enq; }
    hipcpu::wait_for(s, upto);
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize()
{
    if (hipcpu::stream_mode() == 0) return hipSuccess;
    hipcpu::wait_for_all();
    return hipSuccess;
}
// the stream's work completes, the object stays (an event may still name it)
inline hipError_t hipStreamDestroy(hipStream_t s) { return hipStreamSynchronize(s); }
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    if (!HIPCPU_DEFER(s)) {
        if (hipcpu::stream_mode() != 0 && e->s) hipcpu::wait_for(e->s, e->seq);
        return hipSuccess;
    }
    HIPCPU_LOCK;
    if (e->s) hipcpu::push(s, nullptr, e->s, e->seq);       // the record made so far; a never-recorded event is complete
    return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipcpuEvent; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventSynchronize(hipEvent_t e)
{
    if (hipcpu::stream_mode() == 0 || !e->s) return hipSuccess;
    hipcpu::wait_for(e->s, e->seq);
    return hipSuccess;
}
inline hipError_t hipEventDestroy(hipEvent_t e) { (void)hipEventSynchronize(e); delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr)
{
    if (!HIPCPU_DEFER(s)) { e->t = std::chrono::steady_clock::now(); e->s = nullptr; return hipSuccess; }
    HIPCPU_LOCK;
    hipcpu::push(s, [e] { e->t = std::chrono::steady_clock::now(); });
    e->s = s;
    e->seq = s->enq;
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    (void)hipEventSynchronize(a);
    (void)hipEventSynchronize(b);
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline hipError_t hipGetSymbolAddress(void **p, const void *symbol) { *p = const_cast<void *>(symbol); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
// Allocation-failure injection and leak accounting for the tests (the reference's Tcov suite fails every malloc in turn
// through SuiteSparse_config's allocator hooks, CHOLMOD/Tcov/t_memory.c; this is the device-side counterpart):
//   hipcpu_fail_alloc_at(k)  the k-th allocation from now on (k = 1: the next one; hipMalloc and hipHostMalloc count
//                            alike) fails with hipErrorOutOfMemory; 0 switches the injection off
//   hipcpu_alloc_calls()     allocations attempted since the last hipcpu_fail_alloc_at
//   hipcpu_live_allocs()     device + pinned allocations that have not been freed
//   hipcpu_set_free_memory(b) what hipMemGetInfo reports as free (total stays 288 GB); < 0: back to the default
namespace hipcpu {
inline std::atomic<long> alloc_fail_at{0}, alloc_calls{0}, alloc_live{0};
inline std::atomic<long long> mem_free_override{-1};
inline bool alloc_should_fail()
{
    const long k = ++alloc_calls, f = alloc_fail_at.load();
    return f > 0 && k == f;
}
}  // namespace hipcpu
extern "C" {
__attribute__((visibility("default"), used)) inline void hipcpu_fail_alloc_at(long k) { hipcpu::alloc_calls = 0; hipcpu::alloc_fail_at = k; }
__attribute__((visibility("default"), used)) inline long hipcpu_alloc_calls() { return hipcpu::alloc_calls.load(); }
__attribute__((visibility("default"), used)) inline long hipcpu_live_allocs() { return hipcpu::alloc_live.load(); }
__attribute__((visibility("default"), used)) inline void hipcpu_set_free_memory(long long b) { hipcpu::mem_free_override = b; }
}
namespace hipcpu {
inline std::mutex pinned_mu;
inline std::map<void *, size_t> pinned_sizes;           // pinned allocations and their sizes (hipcpu_pinned_bytes)
}
extern "C" __attribute__((visibility("default"), used)) inline long long hipcpu_pinned_bytes()
{
    std::lock_guard<std::mutex> lk(hipcpu::pinned_mu);
    long long t = 0;
    for (auto &kv : hipcpu::pinned_sizes) t += (long long)kv.second;
    return t;
}
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned)
{
    *p = hipcpu::alloc_should_fail() ? nullptr : std::malloc(bytes ? bytes : 1);
    if (*p) { hipcpu::alloc_live++; std::lock_guard<std::mutex> lk(hipcpu::pinned_mu); hipcpu::pinned_sizes[*p] = bytes; }
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipHostFree(void *p)
{
    if (p) { hipcpu::alloc_live--; std::lock_guard<std::mutex> lk(hipcpu::pinned_mu); hipcpu::pinned_sizes.erase(p); }
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t bytes)
{
    *p = hipcpu::alloc_should_fail() ? nullptr : std::malloc(bytes ? bytes : 1);
    if (*p) hipcpu::alloc_live++;
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void *p) { (void)hipDeviceSynchronize(); if (p) hipcpu::alloc_live--; std::free(p); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b)
{
    *total_b = (size_t)288 << 30;
    const long long o = hipcpu::mem_free_override.load();
    *free_b = o >= 0 ? (size_t)o : *total_b - ((size_t)1 << 30);
    return hipSuccess;
}
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st)
{
    if (!HIPCPU_DEFER(st)) return hipMemcpy(d, s, n, k);
    HIPCPU_LOCK;
    if (k == hipMemcpyHostToDevice) {
        auto snap = std::make_shared<std::vector<char>>((const char *)s, (const char *)s + n);
        hipcpu::push(st, [d, snap] { std::memcpy(d, snap->data(), snap->size()); });
    } else {
        hipcpu::push(st, [d, s, n] { std::memcpy(d, s, n); });
    }
    return hipSuccess;
}
inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st)
{
    auto rows = [=](const char *src, size_t spitch) { for (size_t r = 0; r < h; r++) std::memcpy((char *)d + r * dp, src + r * spitch, w); };
    if (!HIPCPU_DEFER(st)) { rows((const char *)s, sp); return hipSuccess; }
    HIPCPU_LOCK;
    if (k == hipMemcpyHostToDevice) {
        auto snap = std::make_shared<std::vector<char>>(w * h);
        for (size_t r = 0; r < h; r++) std::memcpy(snap->data() + r * w, (const char *)s + r * sp, w);
        hipcpu::push(st, [rows, snap, w] { rows(snap->data(), w); });
    } else {
        hipcpu::push(st, [rows, s, sp] { rows((const char *)s, sp); });
    }
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st)
{
    if (!HIPCPU_DEFER(st)) return hipMemset(d, v, n);
    HIPCPU_LOCK;
    hipcpu::push(st, [d, v, n] { std::memset(d, v, n); });
    return hipSuccess;
}
namespace hipcpu {
template <class F> void launch_dims(dim3 g, dim3 b, F call)
{
    // blocks of any width: pad to whole waves (the extra lanes must not exist for the kernel, so only multiples of 64
    // are accepted -- every kernel of the product is launched with 64 or 256 threads)
    if (b.x % 64 != 0 || b.y != 1 || b.z != 1 || g.z != 1) std::abort();
    if (g.x == 0 || g.y == 0) return;
    launch((int64_t)g.x, b.x, call, g.y);
}
// the arguments are evaluated at the launch, as on the device; the kernel runs now (sync) or when the stream gets there
template <class K, class... A> void enqueue_kernel(Stream *st, dim3 g, dim3 b, K kernel, A... args)
{
    if (!HIPCPU_DEFER(st)) { launch_dims(g, b, [&] { kernel(args...); }); return; }
    HIPCPU_LOCK;
    auto tup = std::make_tuple(args...);
    push(st, [g, b, kernel, tup] { launch_dims(g, b, [&] { std::apply(kernel, tup); }); });
}
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipcpu::enqueue_kernel((hipStream_t)(stream), dim3(grid), dim3(block), (kernel), __VA_ARGS__)

#ifdef HIPCPU_FAKE_RCCL
// ---- a stand-in for librccl between THREADS of one process (one thread = one rank, tests/test_multi_abi.py): the nine
// entry points the engine binds (suitesparse_amd/csrc/multi.hip.hpp), exported from the CPU build of the engine so that
// MI355_RCCL_LIB can point the engine's loader at it.  A collective is an operation in the stream it was issued to; it
// can run when every rank has ISSUED its matching call and every peer's stream has done everything in front of it; the
// first rank whose stream gets there moves the data for all.  Calls that do not match (kind, count, type, root) abort:
// on the device they would hang or corrupt.
#include <map>
#include <string>
namespace hipcpu {
struct FakeCall { bool have = false; int kind = 0; const void *send = nullptr; void *recv = nullptr; size_t count = 0;
                  int dtype = 0, op = 0, root = 0; Stream *st = nullptr; uint64_t seq = 0;
                  bool snapped = false, executed = false; std::vector<char> snap; };
struct FakeGroup { int world = 0, joined = 0; std::vector<std::vector<FakeCall>> slot;
                   std::vector<FakeGroup *> children; };         // ncclCommSplit: the n-th split of every rank meets in children[n]
struct FakeComm { FakeGroup *g; int rank; size_t issued = 0; size_t splits = 0; };
inline std::map<std::string, FakeGroup *> &fake_groups() { static std::map<std::string, FakeGroup *> m; return m; }
inline size_t fake_width(int dtype) { if (dtype != 8 && dtype != 4) std::abort(); return 8; }
// does rank r's call k need rank p's data?
inline bool fake_needs(const FakeCall &me, int r, int p)
{
    if (me.kind == 0) return r == me.root;              // reduce: the root needs everybody's
    if (me.kind == 1) return p == me.root;              // broadcast: everybody needs the root's
    return true;                                        // all-reduce
}
// A rank's send buffer is READ when its stream stands in front of its call (by whoever gets there first: nothing of
// that stream runs in between, so it is the data the rank itself would send); a rank's receive buffer is WRITTEN only
// when its OWN stream runs the call -- the latest moment its ordering allows, so a launch that should have waited for the
// collective and does not sees the old contents, whatever the other ranks' threads are doing.
inline bool fake_ready(FakeGroup *g, size_t k, int r)
{
    for (int p = 0; p < g->world; p++) if (!g->slot[k][(size_t)p].have) return false;
    const FakeCall &me = g->slot[k][(size_t)r];
    bool all = true;
    for (int p = 0; p < g->world; p++) {
        FakeCall &c = g->slot[k][(size_t)p];
        if (c.kind != me.kind || c.count != me.count || c.dtype != me.dtype || c.op != me.op || c.root != me.root) {
            fprintf(stderr, "fake rccl: collective %zu does not match between ranks %d and %d\n", k, r, p);
            std::abort();
        }
        if (c.snapped || !(p == r || fake_needs(me, r, p))) continue;
        const bool there = p == r || (c.st && stream_mode() != 0 && run_until(c.st, c.seq - 1));
        if (!there) { all = false; continue; }
        c.snap.assign((const char *)c.send, (const char *)c.send + c.count * fake_width(c.dtype));
        c.snapped = true;
    }
    return all;
}
template <class T> void fake_result(FakeGroup *g, size_t k, int r)
{
    FakeCall &me = g->slot[k][(size_t)r];
    if (me.kind == 0 && r != me.root) return;           // a reduce leaves the other ranks' buffers as they are
    std::vector<T> acc(me.count);
    if (me.kind == 1) {
        std::memcpy(acc.data(), g->slot[k][(size_t)me.root].snap.data(), me.count * sizeof(T));
    } else {
        for (int p = 0; p < g->world; p++) {
            const T *src = (const T *)g->slot[k][(size_t)p].snap.data();
            for (size_t i = 0; i < me.count; i++)
                acc[i] = p == 0 ? src[i] : (me.op == 3 ? std::min(acc[i], src[i]) : acc[i] + src[i]);
        }
    }
    std::memcpy(me.recv, acc.data(), me.count * sizeof(T));
}
inline void fake_run(FakeGroup *g, size_t k, int r)
{
    FakeCall &me = g->slot[k][(size_t)r];
    if (me.dtype == 8) fake_result<double>(g, k, r); else fake_result<int64_t>(g, k, r);
    me.executed = true;
    bool all = true;
    for (int p = 0; p < g->world; p++) all = all && g->slot[k][(size_t)p].executed;
    if (all) for (int p = 0; p < g->world; p++) std::vector<char>().swap(g->slot[k][(size_t)p].snap);
}
inline int fake_collective(int kind, const void *send, void *recv, size_t count, int dtype, int op, int root, void *comm, Stream *st)
{
    FakeComm *c = (FakeComm *)comm;
    FakeGroup *g = c->g;
    size_t k;
    const int r = c->rank;
    {
        RtLock lk;
        k = c->issued++;
        if (g->slot.size() <= k) g->slot.resize(k + 1, std::vector<FakeCall>((size_t)g->world));
        FakeCall &f = g->slot[k][(size_t)r];
        f.have = true; f.kind = kind; f.send = send; f.recv = recv; f.count = count; f.dtype = dtype; f.op = op; f.root = root;
        if (HIPCPU_DEFER(st)) {
            f.st = st; f.seq = st->enq + 1;
            push(st, [g, k, r] { fake_run(g, k, r); }, nullptr, 0, [g, k, r] { return fake_ready(g, k, r); });
            return 0;
        }
    }
    for (;;) {                                   // synchronous streams: the call returns when the collective is done
        {
            RtLock lk;
            if (fake_ready(g, k, r)) { fake_run(g, k, r); return 0; }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
}  // namespace hipcpu
struct hipcpuRcclId { char internal[128]; };
extern "C" {
inline __attribute__((used, visibility("default"))) int ncclGetUniqueId(hipcpuRcclId *id)
{
    static std::atomic<uint64_t> next{1};
    std::memset(id->internal, 0, sizeof id->internal);
    const uint64_t v = next++;
    std::memcpy(id->internal, &v, sizeof v);
    return 0;
}
inline __attribute__((used, visibility("default"))) int ncclCommInitRank(void **comm, int nranks, hipcpuRcclId id, int rank)
{
    hipcpu::FakeGroup *g;
    {
        hipcpu::RtLock lk;
        hipcpu::FakeGroup *&slot = hipcpu::fake_groups()[std::string(id.internal, sizeof id.internal)];
        if (!slot) { slot = new hipcpu::FakeGroup; slot->world = nranks; }
        g = slot;
        if (g->world != nranks || rank < 0 || rank >= nranks) return 4;     // ncclInvalidArgument
        g->joined++;
    }
    for (;;) {                                   // like the real call: returns when every rank has joined
        { hipcpu::RtLock lk; if (g->joined >= g->world) break; }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    *comm = new hipcpu::FakeComm{g, rank, 0, 0};
    return 0;
}
// (all ranks in one colour, key = rank: what the engine asks for)
inline __attribute__((used, visibility("default"))) int ncclCommSplit(void *comm, int color, int key, void **newcomm, void *)
{
    hipcpu::FakeComm *c = (hipcpu::FakeComm *)comm;
    if (color != 0 || key != c->rank) return 4;
    hipcpu::FakeGroup *g;
    {
        hipcpu::RtLock lk;
        const size_t n = c->splits++;
        if (c->g->children.size() <= n) c->g->children.resize(n + 1, nullptr);
        if (!c->g->children[n]) { c->g->children[n] = new hipcpu::FakeGroup; c->g->children[n]->world = c->g->world; }
        g = c->g->children[n];
        g->joined++;
    }
    for (;;) {
        { hipcpu::RtLock lk; if (g->joined >= g->world) break; }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    *newcomm = new hipcpu::FakeComm{g, c->rank, 0, 0};
    return 0;
}
inline __attribute__((used, visibility("default"))) int ncclCommDestroy(void *comm) { delete (hipcpu::FakeComm *)comm; return 0; }
inline __attribute__((used, visibility("default"))) int ncclReduce(const void *s, void *r, size_t n, int dt, int op, int root, void *comm, hipStream_t st)
{ return hipcpu::fake_collective(0, s, r, n, dt, op, root, comm, st); }
inline __attribute__((used, visibility("default"))) int ncclBroadcast(const void *s, void *r, size_t n, int dt, int root, void *comm, hipStream_t st)
{ return hipcpu::fake_collective(1, s, r, n, dt, 0, root, comm, st); }
inline __attribute__((used, visibility("default"))) int ncclAllReduce(const void *s, void *r, size_t n, int dt, int op, void *comm, hipStream_t st)
{ return hipcpu::fake_collective(2, s, r, n, dt, op, 0, comm, st); }
inline __attribute__((used, visibility("default"))) int ncclGroupStart() { return 0; }
inline __attribute__((used, visibility("default"))) int ncclGroupEnd() { return 0; }
inline __attribute__((used, visibility("default"))) const char *ncclGetErrorString(int) { return "fake rccl error"; }
}
#endif  // HIPCPU_FAKE_RCCL
