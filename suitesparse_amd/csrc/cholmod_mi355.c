// cholmod_mi355.c -- the CHOLMOD side of the drop-in (include/cholmod_mi355.h): plain C, compiled INTO a CHOLMOD
// build with -DCHOLMOD_HAS_MI355 (it needs cholmod.h), calling libmi355_cholmod.so through its C ABI.
//
// What it holds: the side table  cholmod_factor* -> mi355_factor*  (sizeof (cholmod_factor) stays what it is), the
// selection logic of Common->useGPU / L->useGPU, and the translation of cholmod_sparse / cholmod_factor /
// cholmod_dense fields into the plain arrays of the C ABI and of its results back into L->x, L->minor,
// Common->status and the Common->cholmod_gpu_* counters.  Everything numeric happens behind the ABI.
//
// Reference interfaces mirrored (file:line under /root/reference):
//   CHOLMOD/Supernodal/cholmod_super_symbolic.c:239-320   useGPU selection (env CHOLMOD_USE_GPU, probe, L->useGPU)
//   CHOLMOD/Supernodal/cholmod_super_numeric.c:100-335     wrapper whose worker switch (:295-322) is taken over
//   CHOLMOD/Supernodal/t_cholmod_super_numeric_worker.c:233  "Common->useGPU == 1 && L->useGPU" gate
//   CHOLMOD/Supernodal/cholmod_super_solve.c:52-139,153-240  lsolve / ltsolve wrappers
//   CHOLMOD/Cholesky/cholmod_solve.c:656-724                supernodal branch of cholmod_solve2 (Y = P*B, sweeps, X = P'*Y)
//   CHOLMOD/Cholesky/cholmod_factorize.c:181-262            S = tril (P*A*P') by two cholmod_ptranspose calls, then super_numeric
//   CHOLMOD/Utility/t_cholmod_free_factor.c:22-90          where a factor loses its supernodal arrays
//   CHOLMOD/GPU/t_cholmod_gpu.c:74-82                       error policy (report CHOLMOD_GPU_PROBLEM)
//   CHOLMOD/Include/cholmod.h:966-986                       per-kernel time / call counters
#include "cholmod_mi355.h"
#include "mi355_cholmod.h"

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

// cholmod.h is the public header: these three live in cholmod_internal.h
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#ifndef EMPTY
#define EMPTY (-1)
#endif

// ---- side table ------------------------------------------------------------------------------------------------
// One entry per factor with a device image.  The signature guards against a recycled address: a factor whose
// supernodal arrays were replaced (or whose xtype changed) never meets a stale handle.
typedef struct
{
    const cholmod_factor *L ;
    mi355_factor *h ;
    const void *super, *s ;         // the arrays the handle was built from
    size_t n, nsuper, ssize, xsize ;
    int lx ;                        // MI355_REAL / MI355_COMPLEX
    int perm_set ;                  // L->Perm is on the device (cholmod_mi355_solve2)
    int multi_world ;               // > 1: the handle is partitioned over that many ranks (cholmod_mi355_register_comm)
    int xprint_set ;                // xprint is the fingerprint of L->x as the device left it
    uint64_t xprint ;
} entry_t ;

static entry_t *table = NULL ;
static int table_len = 0, table_cap = 0 ;
static pthread_mutex_t table_lock = PTHREAD_MUTEX_INITIALIZER ;
static int64_t taken [4] = {0, 0, 0, 0} ;  // calls taken over: numeric, single sweeps, whole solve2 branches;
                                            // [3]: numeric calls that placed A through L->Perm on the device

// cholmod_factorize_p told us that the matrix it is about to hand to cholmod_super_numeric for this factor is A
// ITSELF, not S = tril (P*A*P') (cholmod_mi355_defer_permutation).  Per thread: CHOLMOD is not re-entrant per Common,
// distinct Commons may run on distinct threads.
static __thread const cholmod_sparse *deferred_A = NULL ;
static __thread const cholmod_factor *deferred_L = NULL ;

static int find_locked (const cholmod_factor *L)
{
    for (int k = 0 ; k < table_len ; k++) if (table [k].L == L) return (k) ;
    return (-1) ;
}

static int matches (const entry_t *e, const cholmod_factor *L, int lx)
{
    return (e->super == L->super && e->s == L->s && e->n == L->n && e->nsuper == L->nsuper &&
            e->ssize == L->ssize && e->xsize == L->xsize && e->lx == lx) ;
}

// the handle of L if it is still the image of L's present structure, else NULL (a stale one is destroyed)
static mi355_factor *handle_of (const cholmod_factor *L, int lx)
{
    mi355_factor *h = NULL, *stale = NULL ;
    pthread_mutex_lock (&table_lock) ;
    int k = find_locked (L) ;
    if (k >= 0)
    {
        if (matches (&table [k], L, lx)) h = table [k].h ;
        else
        {
            stale = table [k].h ;
            table [k] = table [--table_len] ;
        }
    }
    pthread_mutex_unlock (&table_lock) ;
    if (stale) mi355_factor_free (stale) ;
    return (h) ;
}

static int remember_handle (const cholmod_factor *L, mi355_factor *h, int lx)
{
    pthread_mutex_lock (&table_lock) ;
    if (table_len == table_cap)
    {
        int cap = table_cap ? 2 * table_cap : 8 ;
        entry_t *t = realloc (table, (size_t) cap * sizeof (entry_t)) ;
        if (!t) { pthread_mutex_unlock (&table_lock) ; return (0) ; }
        table = t ; table_cap = cap ;
    }
    entry_t *e = &table [table_len++] ;
    e->L = L ; e->h = h ; e->super = L->super ; e->s = L->s ;
    e->n = L->n ; e->nsuper = L->nsuper ; e->ssize = L->ssize ; e->xsize = L->xsize ; e->lx = lx ;
    e->perm_set = 0 ;
    e->multi_world = 0 ;
    e->xprint_set = 0 ; e->xprint = 0 ;
    pthread_mutex_unlock (&table_lock) ;
    return (1) ;
}

// ---- is L->x still what the device computed? --------------------------------------------------------------------
// The solves use the factor that is resident on the device, not L->x.  CHOLMOD's own code changes the values of a
// supernodal L only through the hooked functions, but L->x is a public array: a caller may scale or patch it and solve.
// The reference's GPU path has no such trap (it keeps nothing between calls).  So after every device factorization a
// fingerprint of L->x is kept with the handle -- the bit patterns of a few thousand entries spread evenly over the array
// plus the first and the last entry of up to 1024 supernodes -- and every solve hook compares it first: a mismatch drops
// the device image and the CPU code solves from L->x, as it would in a build without the device.  A sample, not a proof:
// an edit that misses every sampled entry still needs cholmod_mi355_forget (L), as INTEGRATION.md says.
#define XPRINT_SAMPLES 4096
static uint64_t lx_fingerprint (const cholmod_factor *L)
{
    const size_t E = (L->xtype == CHOLMOD_REAL) ? 1 : 2 ;
    const size_t len = L->xsize * E ;
    const uint64_t *x = (const uint64_t *) L->x ;
    uint64_t hsh = 1469598103934665603ull ;
    if (x == NULL || len == 0) return (hsh) ;
    const size_t step = (len + XPRINT_SAMPLES - 1) / XPRINT_SAMPLES ;
    for (size_t k = 0 ; k < len ; k += step) { hsh ^= x [k] ; hsh *= 1099511628211ull ; }
    hsh ^= x [len - 1] ; hsh *= 1099511628211ull ;
    const int64_t *px = (const int64_t *) L->px ;
    if (px != NULL && L->nsuper > 0)
    {
        const size_t sstep = (L->nsuper + 1023) / 1024 ;
        for (size_t sn = 0 ; sn < L->nsuper ; sn += sstep)
        {
            hsh ^= x [(size_t) px [sn] * E] ; hsh *= 1099511628211ull ;
            hsh ^= x [(size_t) px [sn + 1] * E - 1] ; hsh *= 1099511628211ull ;
        }
    }
    return (hsh) ;
}

// remember the fingerprint of L->x (set != 0, after the device filled it) or compare it (set == 0): TRUE if L->x is
// still the factor the device holds (or nothing is known about it)
static int lx_unchanged (const cholmod_factor *L, int set)
{
    const uint64_t now = lx_fingerprint (L) ;
    int same = 1 ;
    pthread_mutex_lock (&table_lock) ;
    int k = find_locked (L) ;
    if (k >= 0)
    {
        if (set) { table [k].xprint = now ; table [k].xprint_set = 1 ; }
        else if (table [k].xprint_set) same = (table [k].xprint == now) ;
    }
    pthread_mutex_unlock (&table_lock) ;
    return (same) ;
}

// multi_world of L's entry: set (world > 0) or read (world == 0)
static int multi_flag (const cholmod_factor *L, int world)
{
    int was = 0 ;
    pthread_mutex_lock (&table_lock) ;
    int k = find_locked (L) ;
    if (k >= 0)
    {
        was = table [k].multi_world ;
        if (world > 0) table [k].multi_world = world ;
    }
    pthread_mutex_unlock (&table_lock) ;
    return (was) ;
}

// ---- N ranks ------------------------------------------------------------------------------------------------------
// One process (or thread) per GPU, each with its own cholmod_common, the same matrix and the same analysis.  The
// application registers its rank's engine communicator (mi355_comm_create_rccl / _adopt_rccl / mi355_comm_create) for
// its Common; cholmod_l_factorize then runs the subtree-to-GPU factorization behind mi355_factor_numeric_multi and
// gathers the factor, so that every rank's L->x is the complete factor cholmod_factorize promises.  No counterpart in
// the reference (one device only, CHOLMOD/GPU/cholmod_gpu.c:138-142).
typedef struct { const cholmod_common *cm ; mi355_comm *comm ; int rank, world ; } comm_entry_t ;
static comm_entry_t *comms = NULL ;
static int comms_len = 0, comms_cap = 0 ;

int cholmod_mi355_register_comm (cholmod_common *Common, void *comm, int rank, int world)
{
    if (!Common || (comm && (world < 1 || rank < 0 || rank >= world))) return (FALSE) ;
    int ok = TRUE ;
    pthread_mutex_lock (&table_lock) ;
    int k ;
    for (k = 0 ; k < comms_len ; k++) if (comms [k].cm == Common) break ;
    if (!comm)
    {
        if (k < comms_len) comms [k] = comms [--comms_len] ;
    }
    else
    {
        if (k == comms_len && comms_len == comms_cap)
        {
            int cap = comms_cap ? 2 * comms_cap : 4 ;
            comm_entry_t *t = realloc (comms, (size_t) cap * sizeof (comm_entry_t)) ;
            if (t) { comms = t ; comms_cap = cap ; } else ok = FALSE ;
        }
        if (ok)
        {
            if (k == comms_len) comms_len++ ;
            comms [k].cm = Common ; comms [k].comm = (mi355_comm *) comm ; comms [k].rank = rank ; comms [k].world = world ;
        }
    }
    pthread_mutex_unlock (&table_lock) ;
    return (ok) ;
}

static mi355_comm *comm_of (const cholmod_common *Common, int *rank, int *world)
{
    mi355_comm *c = NULL ;
    pthread_mutex_lock (&table_lock) ;
    for (int k = 0 ; k < comms_len ; k++)
    {
        if (comms [k].cm == Common) { c = comms [k].comm ; *rank = comms [k].rank ; *world = comms [k].world ; break ; }
    }
    pthread_mutex_unlock (&table_lock) ;
    return (c) ;
}

static int perm_flag (const cholmod_factor *L, int set)
{
    int was = 0 ;
    pthread_mutex_lock (&table_lock) ;
    int k = find_locked (L) ;
    if (k >= 0)
    {
        was = table [k].perm_set ;
        if (set) table [k].perm_set = 1 ;
    }
    pthread_mutex_unlock (&table_lock) ;
    return (was) ;
}

static void count_taken (int what)
{
    pthread_mutex_lock (&table_lock) ;
    taken [what]++ ;
    pthread_mutex_unlock (&table_lock) ;
}

void cholmod_mi355_calls_taken (int64_t out [4])
{
    pthread_mutex_lock (&table_lock) ;
    for (int k = 0 ; k < 4 ; k++) out [k] = taken [k] ;
    pthread_mutex_unlock (&table_lock) ;
}

void cholmod_mi355_forget (cholmod_factor *L)
{
    if (!L) return ;
    mi355_factor *h = NULL ;
    pthread_mutex_lock (&table_lock) ;
    int k = find_locked (L) ;
    if (k >= 0)
    {
        h = table [k].h ;
        table [k] = table [--table_len] ;
    }
    pthread_mutex_unlock (&table_lock) ;
    if (h) mi355_factor_free (h) ;
}

int cholmod_mi355_live_handles (void)
{
    pthread_mutex_lock (&table_lock) ;
    int k = table_len ;
    pthread_mutex_unlock (&table_lock) ;
    return (k) ;
}

// ---- selection ---------------------------------------------------------------------------------------------------
void cholmod_mi355_select (cholmod_sparse *A, int for_whom, cholmod_factor *L, cholmod_common *Common)
{
    if (!(for_whom == CHOLMOD_ANALYZE_FOR_CHOLESKY && A->dtype == CHOLMOD_DOUBLE))
    {
        // as in the reference: the request is left as it is for other callers (SPQR) and single precision,
        // which the numeric gate refuses anyway through L->useGPU == 0
        L->useGPU = 0 ;
        return ;
    }
    if (Common->useGPU == EMPTY)
    {
        // not asked for, not forbidden: the environment decides, and with it come the memory limits
        // (cholmod_super_symbolic.c:250-289)
        const char *env = getenv ("CHOLMOD_USE_GPU") ;
        Common->useGPU = (env && atoi (env) != 0) ? 1 : 0 ;
        if (Common->useGPU == 1)
        {
            const char *env_bytes = getenv ("CHOLMOD_GPU_MEM_BYTES") ;
            const char *env_fraction = getenv ("CHOLMOD_GPU_MEM_FRACTION") ;
            if (env_bytes)
            {
                const long long b = atoll (env_bytes) ;
                Common->maxGpuMemBytes = (b > 0) ? (size_t) b : 0 ;
            }
            if (env_fraction)
            {
                double f = atof (env_fraction) ;
                if (f < 0) f = 0 ;
                if (f > 1) f = 1 ;
                Common->maxGpuMemFraction = f ;
            }
        }
    }
    if (Common->useGPU == 1) Common->useGPU = (mi355_probe ( ) > 0) ? 1 : 0 ;
    L->useGPU = Common->useGPU ;
}

// ---- numeric factorization -------------------------------------------------------------------------------------
// the gate of t_cholmod_super_numeric_worker.c:233 ("Common->useGPU == 1 && L->useGPU") plus what this device path
// covers: int64, double, stype <= 0
static int numeric_gate (const cholmod_sparse *A, const cholmod_factor *L, const cholmod_common *Common)
{
    return (Common->useGPU == 1 && L->useGPU && A->dtype == CHOLMOD_DOUBLE && L->dtype == CHOLMOD_DOUBLE &&
            L->itype == CHOLMOD_LONG && A->stype <= 0) ;
}

int cholmod_mi355_defer_permutation (cholmod_sparse *A, cholmod_factor *L, cholmod_common *Common)
{
    deferred_A = NULL ; deferred_L = NULL ;
    // exactly the calls the numeric hook below will take over (else the CPU worker would meet an unpermuted matrix):
    // a symbolic L gets A's dtype from the wrapper (cholmod_super_numeric.c:222), a numeric one must have it already
    if (!(Common->useGPU == 1 && L->useGPU) || !L->is_super || L->itype != CHOLMOD_LONG) return (FALSE) ;
    if (A->dtype != CHOLMOD_DOUBLE || (L->xtype != CHOLMOD_PATTERN && L->dtype != CHOLMOD_DOUBLE)) return (FALSE) ;
    if (A->stype >= 0 || L->ordering == CHOLMOD_NATURAL || L->Perm == NULL) return (FALSE) ;
    if (A->xtype != CHOLMOD_REAL && A->xtype != CHOLMOD_COMPLEX) return (FALSE) ;      // zomplex: the host copy
    if (A->nrow != A->ncol || A->nrow != L->n) return (FALSE) ;
    // cholmod_factorize_p hands S to cholmod_resymbol_noperm afterwards when it converts the result to a simplicial
    // factor with final_resymbol (cholmod_factorize.c:275-282): it needs the permuted copy then
    if (!Common->final_asis && Common->final_resymbol && !Common->final_super) return (FALSE) ;
    deferred_A = A ; deferred_L = L ;
    return (TRUE) ;
}

void cholmod_mi355_defer_clear (void)
{
    deferred_A = NULL ; deferred_L = NULL ;
}

static void gpu_problem (cholmod_common *Common, const char *file, int line)
{
    cholmod_l_error (CHOLMOD_GPU_PROBLEM, file, line, mi355_last_error ( ), Common) ;
}

// ---- soft fallback -----------------------------------------------------------------------------------------------
// The reference never fails a factorization because of its GPU: gpu_init that cannot set itself up turns the GPU off for
// the call and the CPU worker runs (t_cholmod_super_numeric_worker.c:233-244, CHOLMOD/GPU/t_cholmod_gpu.c:107-118), any
// later gpu_* step that fails returns 0 and the CPU BLAS does that step (:755-759, :1007-1011, :1163-1166).  Here the
// device takes a factorization whole or not at all: when the handle cannot be created (a factor larger than the memory
// the device has free or Common->maxGpuMemBytes / maxGpuMemFraction allow, no device, an allocation that fails) or the
// device run fails, L->useGPU goes to 0 for this factor and the unchanged CPU worker factorizes it.
static int64_t fallbacks = 0 ;
static char fallback_reason [512] = "" ;

int64_t cholmod_mi355_fallbacks (char *reason, size_t len)
{
    pthread_mutex_lock (&table_lock) ;
    const int64_t k = fallbacks ;
    if (reason && len > 0)
    {
        strncpy (reason, fallback_reason, len - 1) ;
        reason [len - 1] = '\0' ;
    }
    pthread_mutex_unlock (&table_lock) ;
    return (k) ;
}

static int soft_fallback (cholmod_sparse *A, double beta [2], cholmod_factor *L, cholmod_common *Common,
    int unpermuted, int *ok)
{
    pthread_mutex_lock (&table_lock) ;
    fallbacks++ ;
    strncpy (fallback_reason, mi355_last_error ( ), sizeof (fallback_reason) - 1) ;
    fallback_reason [sizeof (fallback_reason) - 1] = '\0' ;
    pthread_mutex_unlock (&table_lock) ;
    cholmod_mi355_forget (L) ;
    L->useGPU = 0 ;
    if (!unpermuted) return (FALSE) ;       // the wrapper's own switch runs the CPU worker on the S it was given
    // cholmod_factorize_p handed over A itself (cholmod_mi355_defer_permutation): build S = tril (P*A*P') the way it
    // would have (cholmod_factorize.c:228-238) and run the wrapper on it; its hook declines now (L->useGPU == 0)
    cholmod_sparse *A2 = cholmod_l_ptranspose (A, 2, (int64_t *) L->Perm, NULL, 0, Common) ;
    cholmod_sparse *S = (Common->status >= CHOLMOD_OK) ? cholmod_l_ptranspose (A2, 2, NULL, NULL, 0, Common) : NULL ;
    cholmod_l_free_sparse (&A2, Common) ;
    *ok = (S != NULL && Common->status >= CHOLMOD_OK) ? cholmod_l_super_numeric (S, NULL, beta, L, Common) : FALSE ;
    cholmod_l_free_sparse (&S, Common) ;
    return (TRUE) ;
}

int cholmod_mi355_super_numeric (cholmod_sparse *A, cholmod_sparse *F, double beta [2], cholmod_factor *L,
    cholmod_common *Common, int *ok)
{
    const int unpermuted = (deferred_A == A && deferred_L == L) ;
    deferred_A = NULL ; deferred_L = NULL ;
    if (!numeric_gate (A, L, Common))
    {
        // the CPU worker is about to rewrite L->x: a device image of L would be stale afterwards
        cholmod_mi355_forget (L) ;
        if (unpermuted)
        {
            // cannot happen (cholmod_mi355_defer_permutation applies the same gate); never let the CPU worker
            // factorize a matrix that was not permuted for it
            cholmod_l_error (CHOLMOD_GPU_PROBLEM, __FILE__, __LINE__, "internal: unpermuted matrix at the CPU worker", Common) ;
            *ok = FALSE ;
            return (TRUE) ;
        }
        return (FALSE) ;
    }
    // L is complex for a complex or zomplex A (cholmod_super_numeric.c:223-225)
    const int lx = (A->xtype == CHOLMOD_REAL) ? MI355_REAL : MI355_COMPLEX ;
    int rank = 0, world = 1 ;
    mi355_comm *comm = comm_of (Common, &rank, &world) ;
    if (comm != NULL && world <= 1) comm = NULL ;
    mi355_factor *h = handle_of (L, lx) ;
    int rc = MI355_OK ;
    if (h == NULL)
    {
        // on the thread's current device, as the reference (cudaGetDevice, CHOLMOD/GPU/cholmod_gpu.c:165), within the
        // memory Common allows (cholmod_gpu.c:446-489)
        int device = 0 ;
        rc = mi355_current_device (&device) ;
        if (rc == MI355_OK)
        {
            rc = mi355_factor_create_limited (lx, (int64_t) L->n, (int64_t) L->nsuper, (const int64_t *) L->super,
                    (const int64_t *) L->pi, (const int64_t *) L->px, (const int64_t *) L->s,
                    (int64_t) L->ssize, (int64_t) L->xsize, device, (int64_t) Common->maxGpuMemBytes,
                    (Common->maxGpuMemFraction > 0 && Common->maxGpuMemFraction <= 1) ? Common->maxGpuMemFraction : 0.0,
                    NULL, &h) ;
        }
        if (rc == MI355_OK && !remember_handle (L, h, lx))
        {
            mi355_factor_free (h) ;
            cholmod_l_error (CHOLMOD_OUT_OF_MEMORY, __FILE__, __LINE__, "out of memory", Common) ;
            *ok = FALSE ;
            return (TRUE) ;
        }
    }
    int64_t minor = (int64_t) L->n ;
    int status = MI355_STATUS_OK ;
    if (rc == MI355_OK)
    {
#ifdef BLAS_TIMER
        mi355_factor_set_profile (h, 1) ;
#endif
        const int64_t *Anz = A->packed ? NULL : (const int64_t *) A->nz ;
        const int zomplex = (A->xtype == CHOLMOD_ZOMPLEX) ;
        if (A->stype == 0)
        {
            // A*F + beta*I with F = A' or A(:,f)' (checked by the wrapper, cholmod_super_numeric.c:135-154)
            rc = mi355_factor_set_matrix_unsym (h, (int64_t) A->ncol, (const int64_t *) A->p, (const int64_t *) A->i,
                    Anz, (const double *) A->x, zomplex ? (const double *) A->z : NULL,
                    (const int64_t *) F->p, (const int64_t *) F->i, F->packed ? NULL : (const int64_t *) F->nz,
                    (const double *) F->x, zomplex ? (const double *) F->z : NULL) ;
        }
        else if (unpermuted)
        {
            // A itself: the assembly kernel places A (i,j) at S (max, min) of the permuted indices, so the copy
            // S = tril (P*A*P') of cholmod_factorize.c:228-238 is never built
            rc = mi355_factor_set_matrix_perm (h, (const int64_t *) A->p, (const int64_t *) A->i, Anz,
                    (const double *) A->x, (const int64_t *) L->Perm) ;
        }
        else if (zomplex)
        {
            rc = mi355_factor_set_matrix_zomplex (h, (const int64_t *) A->p, (const int64_t *) A->i, Anz,
                    (const double *) A->x, (const double *) A->z) ;
        }
        else
        {
            rc = mi355_factor_set_matrix (h, (const int64_t *) A->p, (const int64_t *) A->i, Anz,
                    (const double *) A->x) ;
        }
        // only beta [0] is used (t_cholmod_super_numeric_worker.c:510-520)
        if (rc == MI355_OK && comm != NULL)
        {
            // N ranks: subtrees to GPUs, the top distributed, then every rank receives the other ranks' columns and
            // copies the complete factor to its L->x
            if (multi_flag (L, 0) != world)
            {
                rc = mi355_factor_partition (h, rank, world, 2, 1) ;
                if (rc == MI355_OK) (void) multi_flag (L, world) ;
            }
            if (rc == MI355_OK) rc = mi355_factor_numeric_multi (h, comm, beta [0], Common->quick_return_if_not_posdef,
                    &minor, &status) ;
            if (rc == MI355_OK && status == MI355_STATUS_OK) rc = mi355_factor_gather_multi (h, comm) ;
            if (rc == MI355_OK) rc = mi355_factor_get_Lx (h, (double *) L->x) ;
        }
        // L->x on the host is part of the contract (cholmod_solve, change_factor, rcond, factor_to_sparse and the user
        // read it): it is filled chunk by chunk WHILE the factorization runs, the bottom of the tree first
        else if (rc == MI355_OK) rc = mi355_factor_numeric_host (h, beta [0], Common->quick_return_if_not_posdef,
                (double *) L->x, &minor, &status) ;
    }
    if (rc == MI355_INVALID)
    {
        // the arrays of A or L are not what the wrapper's checks promised: nothing a CPU run would cure
        cholmod_l_error (CHOLMOD_INVALID, __FILE__, __LINE__, mi355_last_error ( ), Common) ;
        cholmod_mi355_forget (L) ;
        *ok = FALSE ;
        return (TRUE) ;
    }
    if (rc != MI355_OK && comm != NULL)
    {
        // N ranks: a rank that left for its CPU worker would leave the others waiting in their collectives
        gpu_problem (Common, __FILE__, __LINE__) ;
        cholmod_mi355_forget (L) ;
        *ok = FALSE ;
        return (TRUE) ;
    }
    if (rc != MI355_OK) return (soft_fallback (A, beta, L, Common, unpermuted, ok)) ;
    {
        // BLAS statistics (cholmod.h:966-986, printed by cholmod_l_gpu_stats): here one GEMM launch is a batch of
        // fused SYRK + GEMM + assembly (or TRSM) tasks, so they are all booked under GEMM; times only with BLAS_TIMER
        double st [24] ;
        if (mi355_factor_get_stats (h, st) == MI355_OK)
        {
            Common->cholmod_gpu_gemm_calls += (size_t) st [2] ;
            Common->cholmod_gpu_potrf_calls += (size_t) st [4] ;
#ifdef BLAS_TIMER
            Common->cholmod_gpu_gemm_time += st [1] * 1e-3 ;
            Common->cholmod_gpu_potrf_time += st [3] * 1e-3 ;
            Common->cholmod_assemble_time += st [5] * 1e-3 ;
#endif
        }
    }
    count_taken (0) ;
    if (unpermuted) count_taken (3) ;
    (void) lx_unchanged (L, 1) ;            // L->x as the device left it
    L->minor = (size_t) minor ;
    if (status == MI355_STATUS_NOT_POSDEF)
    {
        // a warning, not an error: super_numeric still returns TRUE (cholmod_super_numeric.c:37-41)
        cholmod_l_error (CHOLMOD_NOT_POSDEF, __FILE__, __LINE__, "matrix not positive definite", Common) ;
    }
    *ok = TRUE ;
    return (TRUE) ;
}

// ---- triangular solves -------------------------------------------------------------------------------------------
// a device sweep that cannot run (no memory for the right-hand sides, a failed launch) leaves the call to the host code,
// whose input -- L->x -- is complete; the device image stays (the next call may fit)
static int solve_fallback (void)
{
    pthread_mutex_lock (&table_lock) ;
    fallbacks++ ;
    strncpy (fallback_reason, mi355_last_error ( ), sizeof (fallback_reason) - 1) ;
    fallback_reason [sizeof (fallback_reason) - 1] = '\0' ;
    pthread_mutex_unlock (&table_lock) ;
    return (FALSE) ;
}

int cholmod_mi355_super_solve (cholmod_factor *L, cholmod_dense *X, int lower, cholmod_common *Common, int *ok)
{
    if (L->dtype != CHOLMOD_DOUBLE || L->itype != CHOLMOD_LONG || X->dtype != CHOLMOD_DOUBLE) return (FALSE) ;
    if (L->xtype != CHOLMOD_REAL && L->xtype != CHOLMOD_COMPLEX) return (FALSE) ;
    if (L->minor < L->n) return (FALSE) ;           // not positive definite: the CPU worker sweeps what is in L->x
    mi355_factor *h = handle_of (L, L->xtype == CHOLMOD_REAL ? MI355_REAL : MI355_COMPLEX) ;
    if (h == NULL || multi_flag (L, 0) > 1) return (FALSE) ;     // (N ranks: every rank solves on its complete L->x)
    if (!lx_unchanged (L, 0))
    {
        // somebody rewrote L->x since the factorization: the device image is not this factor any more
        cholmod_mi355_forget (L) ;
        return (FALSE) ;
    }
    const int rc = lower ? mi355_super_lsolve (h, (int64_t) X->ncol, (double *) X->x, (int64_t) X->d)
                         : mi355_super_ltsolve (h, (int64_t) X->ncol, (double *) X->x, (int64_t) X->d) ;
    if (rc == MI355_INVALID)
    {
        gpu_problem (Common, __FILE__, __LINE__) ;
        *ok = FALSE ;
        return (TRUE) ;
    }
    if (rc != MI355_OK) return (solve_fallback ( )) ;    // L->x on the host is complete: the CPU worker sweeps it
    count_taken (1) ;
    *ok = TRUE ;
    return (TRUE) ;
}

// The whole supernodal branch of cholmod_solve2: B crosses to the device once, both permutations and both sweeps run
// there, X comes back once (the two hooks above would move Y down and up once per sweep).
int cholmod_mi355_solve2 (int sys, cholmod_factor *L, cholmod_dense *B, cholmod_dense *X, cholmod_common *Common,
    int *ok)
{
    if (!L->is_super || L->dtype != CHOLMOD_DOUBLE || L->itype != CHOLMOD_LONG) return (FALSE) ;
    if (B->dtype != CHOLMOD_DOUBLE || X->dtype != CHOLMOD_DOUBLE) return (FALSE) ;
    if (L->xtype != CHOLMOD_REAL && L->xtype != CHOLMOD_COMPLEX) return (FALSE) ;
    // zomplex right-hand sides and Common->prefer_zomplex results keep the host path (whose sweeps are still taken
    // over one by one); D alone is a copy for an LL' factor
    if (B->xtype != CHOLMOD_REAL && B->xtype != CHOLMOD_COMPLEX) return (FALSE) ;
    const int cplx = (L->xtype == CHOLMOD_COMPLEX || B->xtype == CHOLMOD_COMPLEX) ;
    if (X->xtype != (cplx ? CHOLMOD_COMPLEX : CHOLMOD_REAL)) return (FALSE) ;
    if (sys < CHOLMOD_A || sys > CHOLMOD_Lt) return (FALSE) ;
    if (L->minor < L->n) return (FALSE) ;
    if (X->nrow != L->n || X->ncol != B->ncol || X->d < L->n) return (FALSE) ;
    mi355_factor *h = handle_of (L, L->xtype == CHOLMOD_REAL ? MI355_REAL : MI355_COMPLEX) ;
    if (h == NULL || multi_flag (L, 0) > 1) return (FALSE) ;
    if (!lx_unchanged (L, 0))
    {
        cholmod_mi355_forget (L) ;              // L->x was rewritten by hand: solve from it on the host
        return (FALSE) ;
    }
    int rc = MI355_OK ;
    if (!perm_flag (L, 0))
    {
        // once per device image; the engine applies it for CHOLMOD_A only, as cholmod_solve.c:262-267 does
        rc = mi355_factor_set_perm (h, L->ordering == CHOLMOD_NATURAL ? NULL : (const int64_t *) L->Perm) ;
        if (rc == MI355_OK) (void) perm_flag (L, 1) ;
    }
    if (rc == MI355_OK)
    {
        rc = mi355_solve_x (h, sys, (int64_t) B->ncol, B->xtype == CHOLMOD_REAL ? MI355_REAL : MI355_COMPLEX,
                (const double *) B->x, (int64_t) B->d, (double *) X->x, (int64_t) X->d) ;
    }
    if (rc == MI355_INVALID)
    {
        gpu_problem (Common, __FILE__, __LINE__) ;
        *ok = FALSE ;
        return (TRUE) ;
    }
    if (rc != MI355_OK) return (solve_fallback ( )) ;    // the host branch of cholmod_solve2 runs on L->x
    count_taken (2) ;
    *ok = TRUE ;
    return (TRUE) ;
}
