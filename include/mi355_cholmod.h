/* mi355_cholmod.h -- C ABI of the MI355X-native supernodal numeric engine.
 *
 * Drop-in boundary for the numeric phase of CHOLMOD's supernodal Cholesky
 * (int64 indices, double precision).  Plain pointers and sizes only; every array
 * is laid out exactly like the corresponding cholmod_factor / cholmod_sparse field
 * (CHOLMOD/Include/cholmod.h:1584-1651, :1168-1206), so a CHOLMOD build can pass its
 * own arrays straight through (INTEGRATION.md shows the dispatch a maintainer adds
 * at CHOLMOD/Supernodal/cholmod_super_numeric.c:295-322 and
 * CHOLMOD/Supernodal/cholmod_super_solve.c:114-131,215-232).
 *
 * All functions return MI355_OK (0) or a negative MI355_* code; the CHOLMOD-style
 * status (Common->status, cholmod.h:833-840) is reported separately where the
 * reference would set it.  mi355_last_error() gives a message for the calling thread.
 * A factor handle is not re-entrant (like a cholmod_common); distinct handles may be
 * used from distinct threads.
 */
#ifndef MI355_CHOLMOD_H
#define MI355_CHOLMOD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes */
#define MI355_OK              0
#define MI355_INVALID        (-4)   /* = CHOLMOD_INVALID       */
#define MI355_OUT_OF_MEMORY  (-2)   /* = CHOLMOD_OUT_OF_MEMORY */
#define MI355_TOO_LARGE      (-3)   /* = CHOLMOD_TOO_LARGE     */
#define MI355_GPU_PROBLEM    (-5)   /* = CHOLMOD_GPU_PROBLEM   */
#define MI355_NOT_INSTALLED  (-1)   /* = CHOLMOD_NOT_INSTALLED (xtype/stype routed to the CPU worker) */

/* CHOLMOD status values reported through *status */
#define MI355_STATUS_OK          0  /* CHOLMOD_OK         */
#define MI355_STATUS_NOT_POSDEF  1  /* CHOLMOD_NOT_POSDEF */

/* xtype values (cholmod.h:134-137) */
#define MI355_REAL     1
#define MI355_COMPLEX  2
#define MI355_ZOMPLEX  3

typedef struct mi355_factor_s mi355_factor;   /* opaque: device-resident factor + schedule */

/* ---- lifecycle: replaces cholmod_l_gpu_probe / _allocate / _deallocate / _start / _end
 *      (CHOLMOD/Include/cholmod.h:4148-4175; CHOLMOD/GPU/cholmod_gpu.c:148,426,187,234,319) */

/* number of usable MI355X-class devices visible to this process (0: none).
 * replaces cholmod_l_gpu_probe (cholmod_gpu.c:148). */
int mi355_probe(void);

/* message of the last failing call on this thread ("" if none) */
const char *mi355_last_error(void);

/* Build the device-side image of a supernodal SYMBOLIC factor: uploads
 * Super/Lpi/Lpx/Ls, replays the descendant lists into the level schedule, builds
 * all relative maps on the device, allocates Lx (xsize entries) in HBM.
 * Arrays are cholmod_factor's super/pi/px/s (cholmod.h:1604-1651); they are copied.
 * xtype: the xtype of L -- MI355_REAL, or MI355_COMPLEX (interleaved re,im; also what a
 * zomplex A produces, cholmod_super_numeric.c:223-225).
 * replaces gpu_init + cholmod_l_gpu_allocate (CHOLMOD/GPU/t_cholmod_gpu.c:84-160,
 * cholmod_gpu.c:426-543) and the per-supernode Map/RelativeMap kernels. */
int mi355_factor_create(int xtype, int64_t n, int64_t nsuper,
                        const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                        const int64_t *Ls, int64_t ssize, int64_t xsize,
                        int device, mi355_factor **handle);

/* mi355_factor_create with a cap on the device memory the handle may take.  Replaces the sizing of the reference's device
 * pool from Common->maxGpuMemBytes / Common->maxGpuMemFraction (CHOLMOD/GPU/cholmod_gpu.c:446-489; environment
 * CHOLMOD_GPU_MEM_BYTES / CHOLMOD_GPU_MEM_FRACTION, CHOLMOD/Supernodal/cholmod_super_symbolic.c:266-279): max_bytes > 0
 * and / or 0 < max_fraction <= 1 of the device's total memory, the smaller one counts; both 0: whatever is free less the
 * 50 MB the reference leaves to the driver (cholmod_gpu.c:464-471).  What the handle needs -- the factor, the block
 * inverses, the symbolic arrays and the schedule's tables -- is computed from the schedule BEFORE anything is allocated:
 * if it exceeds the cap nothing is allocated and MI355_OUT_OF_MEMORY is returned (a caller with a CPU path, such as
 * CHOLMOD, then takes it: the soft fallback of t_cholmod_super_numeric_worker.c:236-237).  *needed_bytes (may be NULL)
 * receives the figure either way. */
int mi355_factor_create_limited(int xtype, int64_t n, int64_t nsuper, const int64_t *Super,
                                const int64_t *Lpi, const int64_t *Lpx, const int64_t *Ls,
                                int64_t ssize, int64_t xsize, int device, int64_t max_bytes, double max_fraction,
                                int64_t *needed_bytes, mi355_factor **handle);

/* Total and free memory of a device in bytes (cholmod_l_gpu_memorysize, CHOLMOD/GPU/cholmod_gpu.c:63-131, without its
 * trial allocations). */
int mi355_device_memory(int device, int64_t *total_bytes, int64_t *free_bytes);

/* The calling thread's current HIP device: the reference works on the current device (cudaGetDevice,
 * CHOLMOD/GPU/cholmod_gpu.c:165), so the CHOLMOD glue passes this to mi355_factor_create. */
int mi355_current_device(int *device);

/* replaces cholmod_l_gpu_deallocate / gpu_end (cholmod_gpu.c:187,319) */
int mi355_factor_free(mi355_factor *handle);

/* ---- numeric factorization: replaces {rd}_cholmod_super_numeric_worker
 *      (CHOLMOD/Supernodal/t_cholmod_super_numeric_worker.c:133-1279) */

/* Stage the matrix S = tril(P A P') (stype < 0; cholmod_sparse p/i/nz/x, packed when
 * Anz == NULL) in HBM.  Host pointers; copied.  Ax holds E doubles per entry (E = 2,
 * interleaved, for a complex L). */
int mi355_factor_set_matrix(mi355_factor *handle, const int64_t *Ap, const int64_t *Ai,
                            const int64_t *Anz, const double *Ax);

/* zomplex A (cholmod_sparse x and z arrays) for a complex L: interleaved on the way in, as the
 * zd_ worker does (t_cholmod_super_numeric_worker.c:69-83). */
int mi355_factor_set_matrix_zomplex(mi355_factor *handle, const int64_t *Ap, const int64_t *Ai,
                                    const int64_t *Anz, const double *Ax, const double *Az);

/* Stage A ITSELF (stype < 0: its lower part) together with L->Perm (NULL: identity): the assembly kernel places
 * A(i,j) at S(max, min) of the permuted indices, conjugated when they swap, so the copy S = tril(P A P') that
 * cholmod_factorize builds with two cholmod_ptranspose calls (CHOLMOD/Cholesky/cholmod_factorize.c:225-236)
 * is never formed.  Same result as mi355_factor_set_matrix on that S. */
int mi355_factor_set_matrix_perm(mi355_factor *handle, const int64_t *Ap, const int64_t *Ai,
                                 const int64_t *Anz, const double *Ax, const int64_t *Perm);

/* stype == 0 (cholmod_super_numeric(A, F, beta, L), CHOLMOD/Supernodal/cholmod_super_numeric.c:135-154): the pair
 * A (n x ncol; Ap has ncol+1 entries, row indices < n) and F = A' or A(:,f)' (ncol x n; Fp has n+1 entries, row
 * indices < ncol; the conjugate transpose for complex values) exactly as cholmod_factorize builds them
 * (CHOLMOD/Cholesky/cholmod_factorize.c:201-210,240-250).  The matrix factorized is A*F + beta*I: the assembly kernel
 * accumulates column k as the sum over F(j,k) of A(:,j)*F(j,k), rows i >= k (worker
 * t_cholmod_super_numeric_worker.c:471-506); A*F is never formed.  Anz / Fnz: NULL when packed.  Az, Fz: both NULL
 * (real, or complex interleaved in Ax / Fx) or both given (zomplex, complex L). */
int mi355_factor_set_matrix_unsym(mi355_factor *handle, int64_t ncol,
                                  const int64_t *Ap, const int64_t *Ai, const int64_t *Anz,
                                  const double *Ax, const double *Az,
                                  const int64_t *Fp, const int64_t *Fi, const int64_t *Fnz,
                                  const double *Fx, const double *Fz);

/* Factorize the staged matrix: Lx (device) <- chol(S + beta*I).  *status receives
 * CHOLMOD_OK or CHOLMOD_NOT_POSDEF and *minor the reference's L->minor (n if positive
 * definite), with the reference's zeroing/repeat-supernode semantics
 * (worker :1084-1146).  Inputs and outputs stay resident in HBM. */
int mi355_factor_numeric(mi355_factor *handle, double beta_re,
                         int quick_return_if_not_posdef, int64_t *minor, int *status);

/* Copy the numeric factor to the host array L->x (E*xsize doubles). */
int mi355_factor_get_Lx(mi355_factor *handle, double *Lx);

/* mi355_factor_numeric + mi355_factor_get_Lx in one, with the copy UNDER the factorization: L->x on the host is part
 * of cholmod_super_numeric's contract (cholmod_solve, cholmod_change_factor, cholmod_rcond and the user read it), and at
 * 150^3 it is 56 GB over PCIe -- more time than the factorization itself.  Lx is cut into chunks of whole supernodes
 * (about 256 MB); a chunk leaves for the host as soon as the levels of its supernodes are complete, copied by a thread
 * of its own on a stream of its own while the main thread goes on enqueueing kernels (Lx may be pageable memory).  The
 * bottom of the elimination tree -- most supernodes, about half the bytes -- is on the host before the top is
 * factorized.  Same results as the two calls; a matrix that is not positive definite gets one plain copy after the
 * reference's repeat-supernode protocol has run.  mi355_factor_get_stats reports how much left early. */
int mi355_factor_numeric_host(mi355_factor *handle, double beta_re, int quick_return_if_not_posdef,
                              double *Lx, int64_t *minor, int *status);

/* The chunks: `count` rows of 4 int64 (lo, hi, ready, flops) -- entries Lx[lo, hi) may be copied once the first `ready`
 * launches of the factorization have run, which execute `flops` of its flops (a multiply and an add counted once each,
 * like Common->fl).  Diagnostic; plan may be NULL to ask for count only. */
int mi355_factor_host_copy_plan(mi355_factor *handle, int64_t cap, int64_t *plan, int64_t *count);

/* cholmod_rcond on the device-resident factor (CHOLMOD/Cholesky/cholmod_rcond.c:51-110, supernodal LL' branch of
 * t_cholmod_rcond_worker.c:76-99,132-137): (min L(j,j) / max L(j,j))^2 over the real parts of the diagonal, 0 if one
 * of them is NaN or the last factorization was not positive definite, 1 for n = 0.  Reads n doubles of HBM; the
 * factor need not be copied to the host for it. */
int mi355_factor_rcond(mi355_factor *handle, double *rcond);

/* mi355_factor_rcond for callers that cannot allocate (code under HIP graph capture set-up, tight serving loops,
 * harnesses that forbid hipMalloc): NO counterpart in the reference.  The same two-step reduction, but the per-workgroup
 * partial results live in a 24 KB array that the library holds on every GPU for the life of the process, the result goes
 * to a DEVICE double of the caller's (complete when the call returns), and the call allocates nothing on the host or the
 * device.  Limit: 1024 partial triples -- always enough, a workgroup strides over more columns for n > 262 144 (a
 * request beyond it would be refused with MI355_TOO_LARGE).  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU race (mi355_factor_rcond does not have this restriction). */
int mi355_factor_rcond_noalloc(mi355_factor *handle, double *rcond_dev);

/* The resident supernodal factor as a packed, monotonic simplicial one (cholmod_change_factor to simplicial numeric:
 * super_num_to_simplicial_num, CHOLMOD/Utility/t_cholmod_change_factor.c:875-1065 and
 * t_cholmod_change_factor_3_template.c): Lp[n+1], Li[lnz], Lnz[n] from the symbolic structure,
 * Lx[E*lnz] packed on the device (to_ll != 0: LL'; 0: LDL' with D(j,j) = L(j,j)^2 and unit-scaled columns) and
 * copied to the host -- lnz instead of xsize entries cross PCIe.  Any of Lp, Li, Lnz may be NULL; Lx == NULL
 * returns the pattern and *lnz only. */
int mi355_factor_to_simplicial(mi355_factor *handle, int to_ll, int64_t *Lp, int64_t *Li, int64_t *Lnz,
                               double *Lx, int64_t *lnz);

/* One call = cholmod_super_numeric's numeric part for (int64, double, stype<0; stype == 0: the split form above):
 * set_matrix + numeric + get_Lx on an existing handle (create it once per analyze). */
int mi355_super_numeric(mi355_factor *handle, const int64_t *Ap, const int64_t *Ai,
                        const int64_t *Anz, const double *Ax, double beta_re,
                        int quick_return_if_not_posdef, double *Lx,
                        int64_t *minor, int *status);

/* ---- multi-GPU phases (no counterpart in the reference, which uses one device:
 *      CHOLMOD/GPU/cholmod_gpu.c:138-142).  Masks have nsuper entries.
 *      local_mask: supernodes of the subtrees this rank owns; top_mask: the shared top
 *      supernodes (every rank factorizes them after the reduction); asm_mask: supernodes whose
 *      columns of S this rank assembles (own subtrees, plus the top on exactly one rank).
 *      phase 0: zero Lx, assemble, factorize the own subtrees and accumulate their updates
 *               (also into this rank's copy of the top panels);
 *      [caller sums the top panels' Lx ranges over the ranks: RCCL all-reduce]
 *      phase 1: factorize the top supernodes.
 *      *first_bad = first supernode whose pivot failed in that phase (-1: none), *info = LAPACK
 *      info inside it.  A phase sees only its own supernodes, so on a failure the caller replays the
 *      reference's protocol (smallest failing supernode, zero tail, refactorized leading columns) with
 *      mi355_factor_numeric, which ignores the partition (suitesparse_amd/multigpu.py, _replay). */
int mi355_factor_set_partition(mi355_factor *handle, const int8_t *local_mask,
                               const int8_t *top_mask, const int8_t *asm_mask);
int mi355_factor_numeric_phase(mi355_factor *handle, int phase, double beta_re,
                               int64_t *first_bad, int64_t *info);

/* The top, owner-computes instead of replicated (no counterpart in the reference).  After set_partition:
 *   mi355_factor_set_top_distribution(h, rank, world): the columns of every top supernode are cut into outer
 *     blocks (tuning key 7 wide); block (s, jb) has one owner, who alone factorizes it and applies the updates
 *     that target it; a finished block is broadcast from its owner before anybody reads it.
 *   mi355_factor_top_plan: the broadcasts, in the order every rank issues them -- `count` rows of 4 int64
 *     (step, lo, n, root): entries Lx[lo, lo+n) of mi355_factor_device_Lx from rank `root`; and the range
 *     [dinv_lo, dinv_hi) of block inverses (entries of mi355_factor_device_Dinv) to SUM over the ranks after
 *     the last segment, for the solves.  plan may be NULL to ask for count only.  `step` is where a broadcast is
 *     ISSUED; the launches right behind it need not wait for it (tuning key 16) -- mi355_factor_numeric_multi knows the
 *     wait points, a caller of this plan may simply wait at once.
 *   mi355_factor_top_run(h, seg): the launches between broadcast seg-1 and broadcast seg (seg = 0 .. count;
 *     returns when they have finished).  seg = count also reports the first failed pivot like numeric_phase.
 * Per factorization: numeric_phase(0); all-reduce the top panels; for seg in 0..count-1: top_run(seg),
 * broadcast seg; top_run(count); all-reduce Dinv[dinv_lo, dinv_hi).  (suitesparse_amd/multigpu.py) */
int mi355_factor_set_top_distribution(mi355_factor *handle, int rank, int world);
int mi355_factor_top_plan(mi355_factor *handle, int64_t cap, int64_t *plan, int64_t *count,
                          int64_t *dinv_lo, int64_t *dinv_hi);
int mi355_factor_top_run(mi355_factor *handle, int64_t seg, int64_t *first_bad, int64_t *info);
double *mi355_factor_device_Dinv(mi355_factor *handle);

/* Solves on a partitioned factor (after both phases succeeded on every rank; no counterpart in the
 * reference).  X is a device pointer, n x nrhs entries, leading dimension d, as in mi355_super_lsolve_dev.
 *   mi355_super_solve_phase_dev: the forward (lower != 0) or backward sweep over the supernodes of one
 *     part only -- 0: this rank's subtrees, 1: the top (replicated on every rank).
 *   mi355_solve_keep_rows_dev: zeroes the rows of X that belong to supernodes outside asm_mask, so that a
 *     sum over the ranks (RCCL all-reduce of X) is the union of the ranks' rows.
 * L x = b on N ranks: keep_rows; phase(0, lower); all-reduce X; phase(1, lower).
 * L' x = b: phase(1, upper); phase(0, upper); keep_rows; all-reduce X.   (suitesparse_amd/multigpu.py) */
int mi355_super_solve_phase_dev(mi355_factor *handle, int part, int lower, int64_t nrhs, double *X, int64_t d);
int mi355_solve_keep_rows_dev(mi355_factor *handle, int64_t nrhs, double *X, int64_t d);

/* ---- more than one GPU behind one call (no counterpart in the reference: one device, cholmod_gpu.c:138-142).
 *      One process per GPU; every rank creates its handle on the SAME symbolic factor, stages the SAME matrix and makes
 *      the same calls.  What suitesparse_amd/multigpu.py used to drive from Python lives here, so that a C caller
 *      (CHOLMOD is one) can use N > 1. */

/* Whole etree subtrees -> ranks.  owner[nsuper] receives the rank of the subtree each supernode lies in, or -1 for
 * the shared top (the ancestors of the cut).  The heaviest subtree is split until it weighs at most
 * total / (2 world) and at least min_subtrees_per_rank * world subtrees exist; they go to the least loaded rank,
 * heaviest first (weights: nscol^3/3 + nsrow2 nscol^2 + nscol nsrow2^2 flops per supernode).  Pure host work, no
 * device needed.  The supernodes of a subtree need not be contiguous (a factor analysed without postordering is
 * fine); a supernode whose parent does not come after it is rejected (MI355_INVALID).
 * stats (may be NULL): [0] flops of the top, [1] total, [2 + r] load of rank r. */
int mi355_partition(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Ls,
                    int world, int min_subtrees_per_rank, int32_t *owner, double *stats);

/* mi355_partition on the handle's symbolic factor + mi355_factor_set_partition with this rank's masks (rank 0
 * assembles the top's entries of A) + (distribute_top != 0) mi355_factor_set_top_distribution + the order of the
 * per-block reductions.  distribute_top == 0: the top is summed once over the ranks and factorized by every rank. */
int mi355_factor_partition(mi355_factor *handle, int rank, int world, int min_subtrees_per_rank,
                           int distribute_top);

/* what mi355_factor_partition decided: owner[nsuper] (may be NULL) and counts[11] (may be NULL): world, rank, column
 * ranges of the distributed top that travel on their own (= broadcasts = reductions: the blocks, or with tuning key 17
 * their pieces), merged top ranges, reductions planned, own-subtree steps enqueued before the first / before the last
 * reduction may start, own-subtree steps in all, broadcasts of THIS rank's schedule with launches between their issue
 * and their wait point (look-ahead, tuning key 16; pieces, key 17), those launches in all, and how many of the ranges
 * are pieces sent before their block was complete. */
int mi355_factor_partition_info(mi355_factor *handle, int32_t *owner, int64_t *counts);

/* The reductions of the top's blocks in the order every rank issues them: `count` rows of 4 int64 (ready, lo, n, root) --
 * entries Lx[lo, lo+n) are summed into rank `root` once THIS rank has enqueued `ready` launches of its own-subtree
 * schedule (mi355_factor_partition_info counts[7] in all).  Diagnostic; plan may be NULL to ask for count only. */
int mi355_factor_reduce_plan(mi355_factor *handle, int64_t cap, int64_t *plan, int64_t *count);

/* The collectives.  Either RCCL over xGMI -- stream-ordered: the engine's compute stream and the communicator's own
 * stream order each other with events, the host does not wait inside a factorization; librccl is bound at run time
 * (the copy the process already holds, else the system's; the environment variable MI355_RCCL_LIB names another by
 * path), the engine has no link-time dependency on it --
 *   mi355_rccl_unique_id(id)               on one rank: 128 bytes (ncclGetUniqueId) for the caller to hand to the others
 *   mi355_comm_create_rccl(id, rank, world, device, &c)     ncclCommInitRank on `device`
 *   mi355_comm_adopt_rccl(ncclComm, rank, world, device, &c)   an ncclComm_t the caller already has (not destroyed)
 * or blocking callbacks of the caller's (MPI, a test harness): buffers are DEVICE pointers of the calling rank, counts
 * are in doubles, the engine has finished its own work on a buffer before it calls and uses it after the callback
 * returns; every rank calls the same callbacks in the same order; 0 = success. */
typedef struct mi355_comm_s mi355_comm;
typedef struct {
    int rank, world;
    void *ctx;
    int (*reduce_sum)(void *ctx, double *dev, int64_t count, int root);      /* sum over the ranks into rank root */
    int (*bcast)(void *ctx, double *dev, int64_t count, int root);
    int (*allreduce_sum)(void *ctx, double *dev, int64_t count);
    int (*allreduce_min_i64)(void *ctx, int64_t *host_value);                 /* one value, on the host */
    int flags;      /* MI355_TRANSPORT_DEFER: the engine queues reduce / bcast / allreduce calls and makes them at the next
                     * point where it waits for them -- the LATEST completion its ordering allows (a plain blocking
                     * transport shows the earliest); a testing aid for the ordering of communication and compute */
} mi355_transport;
#define MI355_TRANSPORT_DEFER 1
int mi355_comm_create(const mi355_transport *t, mi355_comm **comm);
int mi355_rccl_unique_id(void *id128);
int mi355_comm_create_rccl(const void *id128, int rank, int world, int device, mi355_comm **comm);
int mi355_comm_adopt_rccl(void *nccl_comm, int rank, int world, int device, mi355_comm **comm);
int mi355_comm_free(mi355_comm *comm);

/* mi355_factor_numeric on N ranks (after mi355_factor_partition and a set_matrix call on every rank):
 *   own subtrees; each block of the top panels is summed into its owner (reduce) as soon as this rank's last
 *   contribution to it is enqueued, beside the remaining subtree work; the top, owner-computes, every finished block
 *   broadcast before it is read; the block inverses of the top summed over the ranks (for the solves); ONE agreement
 *   on the status.  *minor / *status as mi355_factor_numeric, identical on every rank.  Not positive definite: every
 *   rank replays the reference's protocol on its own with the undistributed schedule (then the WHOLE truncated factor
 *   is resident on every rank, as after mi355_factor_numeric).  After success each rank holds its subtrees' and the
 *   top's columns of L; mi355_super_solve_multi_dev solves with them.  world == 1: mi355_factor_numeric. */
int mi355_factor_numeric_multi(mi355_factor *handle, mi355_comm *comm, double beta_re,
                               int quick_return_if_not_posdef, int64_t *minor, int *status);

/* After a successful mi355_factor_numeric_multi: every rank sends the columns of its subtrees to all others (one
 * broadcast per run of consecutive supernodes of one owner), so that every rank holds ALL of L in HBM and
 * mi355_factor_get_Lx returns the complete L->x that cholmod_factorize promises its caller.  Volume: the subtrees'
 * part of the factor crosses xGMI once.  The solves on N ranks do not need it. */
int mi355_factor_gather_multi(mi355_factor *handle, mi355_comm *comm);

/* L x = b (lower != 0) or L^H x = b on the partitioned factor: X (device, n x nrhs entries, ld d) holds the same
 * right-hand sides on every rank on entry and the same solution on every rank on return; one sum of X over the ranks
 * per sweep is the only exchange. */
int mi355_super_solve_multi_dev(mi355_factor *handle, mi355_comm *comm, int lower, int64_t nrhs, double *X_dev,
                                int64_t d);

/* ---- supernodal triangular solves: replace cholmod_l_super_lsolve / _ltsolve
 *      (CHOLMOD/Supernodal/cholmod_super_solve.c:52,153; workers
 *      t_cholmod_super_solve_worker.c:20-303,309-580).  X is n x nrhs, column-major,
 *      leading dimension d >= n (in entries; E doubles per entry, the xtype of L), host
 *      memory, overwritten with L\X or L^H\X.
 *      No workspace E is needed. */
int mi355_super_lsolve(mi355_factor *handle, int64_t nrhs, double *X, int64_t d);
int mi355_super_ltsolve(mi355_factor *handle, int64_t nrhs, double *X, int64_t d);

/* device-resident variants (X already in HBM, ld d): used when the caller keeps the
 * right-hand sides on the device between the forward and the backward sweep. */
int mi355_super_lsolve_dev(mi355_factor *handle, int64_t nrhs, double *X_dev, int64_t d);
int mi355_super_ltsolve_dev(mi355_factor *handle, int64_t nrhs, double *X_dev, int64_t d);

/* ---- solve with the permutation on the device: replaces the supernodal branch of cholmod_solve2
 *      (CHOLMOD/Cholesky/cholmod_solve.c:656-717: Y = P*B, super_lsolve, super_ltsolve, X = P'*Y) and its
 *      P / Pt systems (:622-655).  sys takes CHOLMOD's values (cholmod.h, CHOLMOD_A ... CHOLMOD_Pt); for an
 *      LL' factor D is the identity; exactly as the reference, A and LDLt run both
 *      sweeps, L/LD the forward one, Lt/DLt the backward one, D copies; Perm is applied for A, P and Pt only
 *      (cholmod_solve.c:262-270).  B and X are n x nrhs, column-major, E doubles per entry. */
#define MI355_SYS_A    0
#define MI355_SYS_LDLt 1
#define MI355_SYS_LD   2
#define MI355_SYS_DLt  3
#define MI355_SYS_L    4
#define MI355_SYS_Lt   5
#define MI355_SYS_D    6
#define MI355_SYS_P    7
#define MI355_SYS_Pt   8

/* L->Perm (n entries, copied to the device; NULL = natural ordering).  Checked to be a permutation. */
int mi355_factor_set_perm(mi355_factor *handle, const int64_t *Perm);

/* host B (ld ldb) -> host X (ld ldx) */
int mi355_solve(mi355_factor *handle, int sys, int64_t nrhs, const double *B, int64_t ldb,
                double *X, int64_t ldx);

/* device B -> device X; everything (both permutations, both sweeps) stays in HBM.  X may alias B except
 * for the P and Pt systems. */
int mi355_solve_dev(mi355_factor *handle, int sys, int64_t nrhs, const double *B_dev, int64_t ldb,
                    double *X_dev, int64_t ldx);

/* The same with a right-hand side whose xtype differs from L's (cholmod_solve.c:228-235,668;
 * t_cholmod_psolve_worker.c:83-240): bxtype = MI355_REAL or MI355_COMPLEX (interleaved).  X is complex if
 * either L or B is.  A complex B under a real L is solved as 2*nrhs real right-hand sides ("dual"); a real
 * B under a complex L gets a zero imaginary part.  (Zomplex B/X: interleave on the host first.) */
int mi355_solve_x(mi355_factor *handle, int sys, int64_t nrhs, int bxtype, const double *B, int64_t ldb,
                  double *X, int64_t ldx);
int mi355_solve_x_dev(mi355_factor *handle, int sys, int64_t nrhs, int bxtype, const double *B_dev,
                      int64_t ldb, double *X_dev, int64_t ldx);

/* ---- instrumentation: replaces the -DBLAS_TIMER counters
 *      (worker :246-266,:763-811; Common->cholmod_gpu_*_time, cholmod.h:966-986) */

/* Device pointer of Lx (for callers that consume the factor on the device). */
double *mi355_factor_device_Lx(mi355_factor *handle);

/* profile != 0: the next numeric calls bracket every launch with HIP events on the
 * engine's stream. */
int mi355_factor_set_profile(mi355_factor *handle, int profile);

/* out[0..23]: 0 total ms of the last numeric call (events on the engine stream)
 *             1 ms in update/trailing/trsm GEMM launches   2 number of GEMM launches
 *             3 ms in POTRF launches                       4 number of POTRF launches
 *             5 ms in assembly (memset + A scatter)        6 executed GEMM flops
 *             7 executed panel flops                       8 assembly bytes (sum ndrow1*ndrow2*8)
 *             9 number of (d,s) update pairs              10 number of levels
 *            11 relative-map entries                      12 block-inverse entries
 *            13 ms of the last solve call                 14 GEMM tiles launched
 *            15 entries of Lx that the last mi355_factor_numeric_host handed to the host copy before its last launch
 *            16 ms in k_gemm128 launches (part of 1)      17 number of k_gemm128 launches
 *            18 useful flops of the tasks in k_gemm128    19 useful flops of the tasks in k_gemm64
 *            20 k_gemm128 tiles per factorization         21 k_gemm64 tiles
 *            22 useful k_gemm128 flops this handle executes (its share of a partition)
 *            23 reduce channel (tuning key 19): launches of the top that the last mi355_factor_numeric_multi enqueued
 *               before it had to wait for the last reduction (0: all reductions were awaited before the top) */
int mi355_factor_get_stats(mi355_factor *handle, double *out24);

/* Measured issue rate of bare v_mfma_f64_16x16x4_f64 on `device` (no counterpart in the reference;
 * used by bench.py to state the FP64 matrix peak next to the vendor figure).
 * out3[0] TFLOP/s chip-wide, out3[1] shader cycles per MFMA seen by one wave (2 waves per SIMD),
 * out3[2] shader clock in MHz during the loop. */
int mi355_debug_mfma_f64_rate(int device, double *out3);

/* Rate of ONE trailing-update task (lower trapezoid M x N, depth K) on random data with kernel 0
 * (k_gemm64) or 1 (k_gemm128); kernel micro-benchmark for profiles/, no reference counterpart. */
int mi355_debug_gemm_rate(int device, int kernel, int M, int N, int K, int iters, double *tflops);

/* Experiment switches (defaults are the measured best; they affect factors created afterwards):
 * key 0 / 1: minimum M / N of a task routed to the 128x128-tile kernel; key 2, 7, 8: nested block widths of
 * the panel factorization above the 64-column panels (defaults 256, 1024, none); key 4: super-tile edge of the
 * tile order (1 = plain row-major); key 5: 1 = deal the tile sequence to the 8 XCDs in contiguous chunks;
 * key 9: fewest right-hand sides that run the solves on the MFMA kernels (default 5; a complex factor: min(value, 3)); key 10: 0 = no solve groups (one
 * diagonal + one update launch per 64-column block instead of two per 256 columns); key 11 / 13: tiles per run of the
 * backward solve updates (scalar / MFMA kernels) = step tiles / value; key 12: 1 = look-ahead (far part of the outermost trailing updates on
 * a second stream, measured +1.6 % at 150^3, off by default); key 14: update tasks of at most one 64 x 64 tile with
 * M*N*K <= value run one wave per task (k_gemm_tiny) instead of one workgroup per task (0 = off, the default: not
 * measured on the device yet); key 15: forward MFMA solve updates walk step tiles / value table entries per
 * workgroup (default 4096); key 16: look-ahead in the distributed top (default 1: of a finished block's trailing update only
 * the part that targets the NEXT block runs before that block's panel chain, the rest after the chain's broadcast has been
 * issued, so chain and broadcast overlap it; 0: the whole update first); key 17: a block of the distributed top is
 * broadcast in finished pieces of the first nested width (key 2: 256 columns) while its panel chain is still running
 * (default 1; 0: in one piece when it is complete); key 18: entries of Lx per chunk of mi355_factor_numeric_host
 * (default 32 Mi = 256 MB real; read when a handle copies for the first time); key 19: the reduce channel of the
 * N-rank driver (read by mi355_comm_create_rccl / _adopt_rccl: a second communicator made with ncclCommSplit and a
 * stream of its own for the sums of the top's ranges into their owners, awaited supernode by supernode where the top
 * first touches them instead of all before the top; default 0 until it has been measured on xGMI); keys 20 / 21: lanes of the
 * copy of the factor to the host (threads, each with a pinned staging buffer and a stream; default 4: they saturate PCIe Gen5 on the MI355X box) and MB per
 * staging buffer (default 32), read when a handle copies for the first time; key 22: smallest B (KB) that mi355_solve /
 * mi355_solve_x move through the lanes (default 4096; smaller ones take one plain copy); key 23: the wide multi-RHS
 * sweeps (default 1: the rows below a solve group of more than 64 columns take all its columns in one pass -- contraction
 * depth 256, 128 for a complex factor -- instead of one pass per 64-column block; 0: per-block kernels everywhere); key 24:
 * workgroups the tiles of a step are dealt to for the wide kernels (default 256 = one per CU; read when a handle builds its
 * solve lists); key 25: 1 = the forward wide kernel of a real factor with 16 waves per workgroup on long shares (measured
 * 3 % slower at 150^3, default 0); key 26: 1 = the trailing update of a panel step also factorizes the diagonal block it
 * completes, in the workgroup of its tile (0, 0) (two dependent launches per 64 columns instead of three; measured + 0.4 %
 * at 150^3 and 2D 2000^2 while the POTRF time moves into the k_gemm128 launches the roofline is built from, so default 0);
 * key 27: deterministic mode (default 0): the scatter updates of an etree level are launched in rounds that hold at most
 * one descendant per ancestor, so every entry of L receives its additions in one fixed order and two factorizations of the
 * same matrix give identical bits, as the reference's sequential loop over the descendants does (the default adds with FP64
 * atomics in whatever order the workgroups arrive: last bits differ from run to run); costs as many launch groups per level
 * as the busiest ancestor has descendants on it; key 28: panel groups (default 64; 0 = off): a real factor's panel chain is
 * taken 256 columns at a time -- one workgroup factorizes the diagonal window (k_panel_diag), one launch solves the rows
 * below it against the whole window -- three launches per 256 columns instead of thirteen (150^3: - 2.8 % of the time);
 * the value is the rows of a panel per workgroup of that second launch: 64 or 128 = k_panel_trsm_reg (the rows in
 * registers), 16 or 32 = k_panel_trsm (the rows in LDS; slower, kept for comparison); key 29: register sets of L in flight
 * per wave of the forward wide sweep, 2 (default) or 3 (measured the same); key 30: look-ahead of the sweeps with several
 * right-hand sides (default 0 = off; 12 = on): inside a supernode of several solve groups the update of the rows beyond the
 * next group runs on a second stream beside the next group's kernel; the value is the number of CUs such a step leaves
 * to that kernel (read when the solve lists are built, at a factor's first solve).  Measured: - 2 .. 4 % for a real factor
 * with 64 right-hand sides, + 4 .. 5 % for a complex one or for 8 right-hand sides; key 31: the explicit inverse of a solve
 * group's whole diagonal block (up to 256 x 256) for the steps of at most that many groups (default 4; 0 = off): such a step of the
 * multi-RHS sweeps becomes ten independent 64 x 64 products on ten workgroups instead of a chain of ten dependent block steps on
 * one (9 us instead of 20 .. 28 us per step of the top supernodes' chains); the inverses are built at the first solve after each
 * factorization, by the chains themselves on the identity, and take 1 MB per group (2 MB complex); more than 4 groups per step
 * measured slower than the chain; key 33 (default 1; 0 = off): a step of the wide sweeps whose
 * tiles, cut to 64 (or 128) columns per task, still fit one round of workgroups is cut that way -- a workgroup then fetches a
 * quarter of the x image and of L (measured - 0.5 .. 1.7 % of a 64-RHS solve); keys 3 and 6: retired.  The values are
 * process-wide and read when a factor is created. */
int mi355_set_tuning(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* MI355_CHOLMOD_H */
