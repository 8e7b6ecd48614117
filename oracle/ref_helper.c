// oracle/ref_helper.c -- TEST INFRASTRUCTURE, not product code.
//
// Flat-array (ctypes-friendly) wrapper around the UNCHANGED reference CHOLMOD
// built by oracle/build_ref.sh into oracle/_ref/libcholmod_ref.so.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
// It drives the reference exactly the way its own callers do:
//   analyze    : cholmod_l_analyze_p            (CHOLMOD/Cholesky/cholmod_analyze.c:121,384)
//   permute    : the ptranspose sequence of     CHOLMOD/Cholesky/cholmod_factorize.c:181-251
//   factorize  : cholmod_l_super_numeric(S,..)  (CHOLMOD/Supernodal/cholmod_super_numeric.c:100,
//                called as in CHOLMOD/Tcov/t_raw_factor.c:239)
//   solve      : cholmod_l_super_lsolve/ltsolve (CHOLMOD/Supernodal/cholmod_super_solve.c:52,153)
//                and cholmod_l_solve            (CHOLMOD/Cholesky/cholmod_solve.c:99)

#include "cholmod.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef struct
{
    cholmod_common cm ;
    cholmod_sparse *A ;     // user matrix (any stype)
    cholmod_sparse *S ;     // tril (P A P'), what super_numeric receives (stype == 0: A(p,f))
    cholmod_sparse *F ;     // stype == 0 only: F = A(p,f)', the second argument of super_numeric
    cholmod_factor *L ;
    double t_analyze, t_factor, t_solve ;
    int last_status ;
} ref_ctx ;

static void quiet_handler (int status, const char *file, int line, const char *msg)
{
    (void) status ; (void) file ; (void) line ; (void) msg ;
}

// column subset f for the unsymmetric case (cholmod_analyze_p / cholmod_factorize_p's fset): A(:,f)*A(:,f)' is
// analyzed and factorized.  NULL / 0 = all columns.  Kept until the next call; used by ref_analyze and ref_permute.
static int64_t *g_fset = NULL ;
static size_t g_fsize = 0 ;
void ref_set_fset (const int64_t *fset, int64_t fsize)
{
    free (g_fset) ; g_fset = NULL ; g_fsize = 0 ;
    if (fset && fsize > 0)
    {
        g_fset = malloc (fsize * sizeof (int64_t)) ;
        memcpy (g_fset, fset, fsize * sizeof (int64_t)) ;
        g_fsize = (size_t) fsize ;
    }
}

void *ref_new (void)
{
    ref_ctx *h = calloc (1, sizeof (ref_ctx)) ;
    cholmod_l_start (&h->cm) ;
    h->cm.useGPU = 0 ;
    h->cm.error_handler = quiet_handler ;
    h->cm.supernodal = CHOLMOD_SUPERNODAL ;
    return h ;
}

void ref_free (void *hv)
{
    ref_ctx *h = hv ;
    if (!h) return ;
    if (h->S && h->S != h->A) cholmod_l_free_sparse (&h->S, &h->cm) ;
    cholmod_l_free_sparse (&h->F, &h->cm) ;
    cholmod_l_free_sparse (&h->A, &h->cm) ;
    cholmod_l_free_factor (&h->L, &h->cm) ;
    cholmod_l_finish (&h->cm) ;
    free (h) ;
}

int ref_nthreads (void *hv) { return ((ref_ctx *) hv)->cm.nthreads_max ; }
void ref_set_nthreads (void *hv, int nt) { ((ref_ctx *) hv)->cm.nthreads_max = nt ; }

// xtype: 1 real, 2 complex (interleaved re,im).  Ax may be NULL only if nnz==0.
int ref_load_matrix (void *hv, int64_t n, const int64_t *Ap, const int64_t *Ai,
    const double *Ax, int xtype, int stype)
{
    ref_ctx *h = hv ;
    if (h->S && h->S != h->A) cholmod_l_free_sparse (&h->S, &h->cm) ;
    h->S = NULL ;
    cholmod_l_free_sparse (&h->A, &h->cm) ;
    int64_t nz = Ap [n] ;
    h->A = cholmod_l_allocate_sparse (n, n, nz > 0 ? nz : 1, /*sorted*/ 1,
        /*packed*/ 1, stype, xtype + CHOLMOD_DOUBLE, &h->cm) ;
    if (!h->A) return -1 ;
    memcpy (h->A->p, Ap, (n+1) * sizeof (int64_t)) ;
    memcpy (h->A->i, Ai, nz * sizeof (int64_t)) ;
    memcpy (h->A->x, Ax, nz * sizeof (double) * (xtype == CHOLMOD_COMPLEX ? 2 : 1)) ;
    return 0 ;
}

// unsymmetric n x ncol matrix (stype 0): the reference then factorizes A*A' (cholmod_factorize.c:201-210,240-250)
int ref_load_matrix_rect (void *hv, int64_t nrow, int64_t ncol, const int64_t *Ap, const int64_t *Ai,
    const double *Ax, int xtype)
{
    ref_ctx *h = hv ;
    if (h->S && h->S != h->A) cholmod_l_free_sparse (&h->S, &h->cm) ;
    h->S = NULL ;
    cholmod_l_free_sparse (&h->F, &h->cm) ;
    cholmod_l_free_sparse (&h->A, &h->cm) ;
    int64_t nz = Ap [ncol] ;
    h->A = cholmod_l_allocate_sparse (nrow, ncol, nz > 0 ? nz : 1, 1, 1, 0, xtype + CHOLMOD_DOUBLE, &h->cm) ;
    if (!h->A) return -1 ;
    memcpy (h->A->p, Ap, (ncol+1) * sizeof (int64_t)) ;
    memcpy (h->A->i, Ai, nz * sizeof (int64_t)) ;
    memcpy (h->A->x, Ax, nz * sizeof (double) * (xtype == CHOLMOD_COMPLEX ? 2 : 1)) ;
    return 0 ;
}

// read one of the reference's own fixture files (.tri / .mtx); returns 0 on success
int ref_read_matrix (void *hv, const char *path)
{
    ref_ctx *h = hv ;
    FILE *f = fopen (path, "r") ;
    if (!f) return -1 ;
    if (h->S && h->S != h->A) cholmod_l_free_sparse (&h->S, &h->cm) ;
    h->S = NULL ;
    cholmod_l_free_sparse (&h->A, &h->cm) ;
    h->A = cholmod_l_read_sparse2 (f, CHOLMOD_DOUBLE, &h->cm) ;
    fclose (f) ;
    if (!h->A) return -2 ;
    if (h->A->xtype == CHOLMOD_PATTERN)
    {
        // same convention as the demo (cholmod_dl_demo.c: pattern -> values)
        return -3 ;
    }
    if (h->A->xtype == CHOLMOD_ZOMPLEX)
    {
        cholmod_l_sparse_xtype (CHOLMOD_COMPLEX + CHOLMOD_DOUBLE, h->A, &h->cm) ;
    }
    cholmod_l_sort (h->A, &h->cm) ;
    return 0 ;
}

// iout: n, nnz, xtype, stype
void ref_matrix_info (void *hv, int64_t *iout)
{
    ref_ctx *h = hv ;
    iout [0] = h->A->nrow ;
    iout [1] = cholmod_l_nnz (h->A, &h->cm) ;
    iout [2] = h->A->xtype ;
    iout [3] = h->A->stype ;
    iout [4] = h->A->ncol ;
}

void ref_get_matrix (void *hv, int64_t *Ap, int64_t *Ai, double *Ax)
{
    ref_ctx *h = hv ;
    cholmod_sparse *A = h->A ;
    int64_t n = A->ncol, *p = A->p, *ii = A->i, *nzv = A->nz ;
    double *x = A->x ;
    int E = (A->xtype == CHOLMOD_COMPLEX) ? 2 : 1 ;
    int64_t q = 0 ;
    for (int64_t j = 0 ; j < n ; j++)
    {
        Ap [j] = q ;
        int64_t pend = A->packed ? p [j+1] : p [j] + nzv [j] ;
        for (int64_t k = p [j] ; k < pend ; k++)
        {
            Ai [q] = ii [k] ;
            for (int e = 0 ; e < E ; e++) Ax [E*q+e] = x [E*k+e] ;
            q++ ;
        }
    }
    Ap [n] = q ;
}

// ordering: CHOLMOD_NATURAL 0, GIVEN 1, AMD 2, METIS 3, NESDIS 4; -1 = default strategy
// supernodal: CHOLMOD_SIMPLICIAL 0 / AUTO 1 / SUPERNODAL 2
int ref_analyze (void *hv, int ordering, const int64_t *perm, int supernodal)
{
    ref_ctx *h = hv ;
    cholmod_l_free_factor (&h->L, &h->cm) ;
    h->cm.supernodal = supernodal ;
    if (ordering >= 0)
    {
        h->cm.nmethods = 1 ;
        h->cm.method [0].ordering = ordering ;
        h->cm.postorder = 1 ;
    }
    else
    {
        h->cm.nmethods = 0 ;
    }
    double t = SuiteSparse_time ( ) ;
    const int un = (h->A->stype == 0) ;
    h->L = cholmod_l_analyze_p (h->A, (int64_t *) perm, un ? g_fset : NULL, un ? g_fsize : 0, &h->cm) ;
    h->t_analyze = SuiteSparse_time ( ) - t ;
    h->last_status = h->cm.status ;
    return (h->L == NULL) ? -1 : h->cm.status ;
}

// iout[16]: n nsuper ssize xsize maxcsize maxesize ordering is_super xtype minor is_ll status nthreads
// dout[8]:  fl lnz anz t_analyze t_factor t_solve
void ref_info (void *hv, int64_t *iout, double *dout)
{
    ref_ctx *h = hv ;
    cholmod_factor *L = h->L ;
    memset (iout, 0, 16 * sizeof (int64_t)) ;
    if (L)
    {
        iout [0] = L->n ; iout [1] = L->nsuper ; iout [2] = L->ssize ;
        iout [3] = L->xsize ; iout [4] = L->maxcsize ; iout [5] = L->maxesize ;
        iout [6] = L->ordering ; iout [7] = L->is_super ; iout [8] = L->xtype ;
        iout [9] = L->minor ; iout [10] = L->is_ll ;
    }
    iout [11] = h->last_status ;
    iout [12] = h->cm.nthreads_max ;
    dout [0] = h->cm.fl ; dout [1] = h->cm.lnz ; dout [2] = h->cm.anz ;
    dout [3] = h->t_analyze ; dout [4] = h->t_factor ; dout [5] = h->t_solve ;
}

void ref_get_symbolic (void *hv, int64_t *Perm, int64_t *ColCount,
    int64_t *Super, int64_t *Lpi, int64_t *Lpx, int64_t *Ls)
{
    ref_ctx *h = hv ;
    cholmod_factor *L = h->L ;
    memcpy (Perm, L->Perm, L->n * sizeof (int64_t)) ;
    memcpy (ColCount, L->ColCount, L->n * sizeof (int64_t)) ;
    if (L->is_super)
    {
        memcpy (Super, L->super, (L->nsuper+1) * sizeof (int64_t)) ;
        memcpy (Lpi, L->pi, (L->nsuper+1) * sizeof (int64_t)) ;
        memcpy (Lpx, L->px, (L->nsuper+1) * sizeof (int64_t)) ;
        memcpy (Ls, L->s, L->ssize * sizeof (int64_t)) ;
    }
}

// S = tril (P A P'), the ptranspose sequence of cholmod_factorize.c:181-251
int64_t ref_permute (void *hv)
{
    ref_ctx *h = hv ;
    cholmod_sparse *A = h->A, *A1 = NULL, *A2 = NULL ;
    cholmod_factor *L = h->L ;
    if (h->S && h->S != h->A) cholmod_l_free_sparse (&h->S, &h->cm) ;
    h->S = NULL ;
    cholmod_l_free_sparse (&h->F, &h->cm) ;
    if (A->stype == 0)
    {
        // F = A(p,:)' ; S = F'  (cholmod_factorize.c:201-210 natural, :240-250 permuted; no fset)
        if (L->ordering == CHOLMOD_NATURAL)
        {
            h->F = cholmod_l_ptranspose (A, 2, NULL, g_fset, g_fsize, &h->cm) ;
            h->S = A ;
        }
        else
        {
            h->F = cholmod_l_ptranspose (A, 2, L->Perm, g_fset, g_fsize, &h->cm) ;
            h->S = h->F ? cholmod_l_ptranspose (h->F, 2, NULL, NULL, 0, &h->cm) : NULL ;
        }
        if (!h->S || !h->F) return -2 ;
        return cholmod_l_nnz (h->S, &h->cm) ;
    }
    if (L->ordering == CHOLMOD_NATURAL)
    {
        if (A->stype > 0) { A1 = cholmod_l_ptranspose (A, 2, NULL, NULL, 0, &h->cm) ; h->S = A1 ; }
        else h->S = A ;
    }
    else if (A->stype > 0)
    {
        A1 = cholmod_l_ptranspose (A, 2, L->Perm, NULL, 0, &h->cm) ;
        h->S = A1 ;
    }
    else
    {
        A2 = cholmod_l_ptranspose (A, 2, L->Perm, NULL, 0, &h->cm) ;
        A1 = cholmod_l_ptranspose (A2, 2, NULL, NULL, 0, &h->cm) ;
        cholmod_l_free_sparse (&A2, &h->cm) ;
        h->S = A1 ;
    }
    if (!h->S) return -2 ;
    return cholmod_l_nnz (h->S, &h->cm) ;
}

void ref_get_S (void *hv, int64_t *Sp, int64_t *Si, double *Sx)
{
    ref_ctx *h = hv ;
    cholmod_sparse *Asave = h->A ;
    h->A = h->S ;
    ref_get_matrix (hv, Sp, Si, Sx) ;
    h->A = Asave ;
}

// F of the last ref_permute of an unsymmetric matrix (ncol x n); sizes via ref_F_info: nrow, ncol, nnz
void ref_F_info (void *hv, int64_t *iout)
{
    ref_ctx *h = hv ;
    iout [0] = h->F ? (int64_t) h->F->nrow : 0 ;
    iout [1] = h->F ? (int64_t) h->F->ncol : 0 ;
    iout [2] = h->F ? cholmod_l_nnz (h->F, &h->cm) : 0 ;
}

void ref_get_F (void *hv, int64_t *Fp, int64_t *Fi, double *Fx)
{
    ref_ctx *h = hv ;
    cholmod_sparse *Asave = h->A ;
    h->A = h->F ;
    ref_get_matrix (hv, Fp, Fi, Fx) ;
    h->A = Asave ;
}

// numeric factorization through the reference CPU worker; returns Common->status
// (0 OK, 1 NOT_POSDEF, <0 error).  Needs ref_permute first.
int ref_factorize (void *hv, double beta_re, int quick_return_if_not_posdef)
{
    ref_ctx *h = hv ;
    double beta [2] = { beta_re, 0 } ;
    h->cm.quick_return_if_not_posdef = quick_return_if_not_posdef ;
    double t = SuiteSparse_time ( ) ;
    int ok = cholmod_l_super_numeric (h->S, h->S->stype == 0 ? h->F : NULL, beta, h->L, &h->cm) ;
    h->t_factor = SuiteSparse_time ( ) - t ;
    h->last_status = h->cm.status ;
    return ok ? h->cm.status : (h->cm.status < 0 ? h->cm.status : -99) ;
}

// whole cholmod_l_factorize (permute + numeric), as the demo times it
int ref_factorize_full (void *hv, double beta_re)
{
    ref_ctx *h = hv ;
    double beta [2] = { beta_re, 0 } ;
    double t = SuiteSparse_time ( ) ;
    cholmod_l_factorize_p (h->A, beta, NULL, 0, h->L, &h->cm) ;
    h->t_factor = SuiteSparse_time ( ) - t ;
    h->last_status = h->cm.status ;
    return h->cm.status ;
}

void ref_get_Lx (void *hv, double *Lx)
{
    ref_ctx *h = hv ;
    cholmod_factor *L = h->L ;
    int E = (L->xtype == CHOLMOD_COMPLEX) ? 2 : 1 ;
    memcpy (Lx, L->x, L->xsize * E * sizeof (double)) ;
}

// overwrite the numeric values of the reference factor (drop-in check: the
// reference's own solve runs on a factor computed elsewhere)
int ref_set_Lx (void *hv, const double *Lx, int64_t minor)
{
    ref_ctx *h = hv ;
    cholmod_factor *L = h->L ;
    if (L->xtype == CHOLMOD_PATTERN) return -1 ;
    int E = (L->xtype == CHOLMOD_COMPLEX) ? 2 : 1 ;
    memcpy (L->x, Lx, L->xsize * E * sizeof (double)) ;
    L->minor = minor ;
    return 0 ;
}

// X (n x nrhs, ld d, xtype of L) is overwritten: lower=1 -> L\X, lower=0 -> L'\X
int ref_super_solve (void *hv, int lower, int64_t nrhs, double *Xx, int64_t d)
{
    ref_ctx *h = hv ;
    cholmod_factor *L = h->L ;
    int E = (L->xtype == CHOLMOD_COMPLEX) ? 2 : 1 ;
    cholmod_dense X ;
    memset (&X, 0, sizeof (X)) ;
    X.nrow = L->n ; X.ncol = nrhs ; X.d = d ; X.nzmax = d * nrhs ;
    X.x = Xx ; X.xtype = L->xtype ; X.dtype = CHOLMOD_DOUBLE ;
    cholmod_dense *Ew = cholmod_l_allocate_dense (nrhs, L->maxesize, nrhs,
        L->xtype + CHOLMOD_DOUBLE, &h->cm) ;
    if (!Ew) return -1 ;
    (void) E ;
    double t = SuiteSparse_time ( ) ;
    int ok = lower ? cholmod_l_super_lsolve (L, &X, Ew, &h->cm)
                   : cholmod_l_super_ltsolve (L, &X, Ew, &h->cm) ;
    h->t_solve = SuiteSparse_time ( ) - t ;
    cholmod_l_free_dense (&Ew, &h->cm) ;
    return ok ? 0 : -2 ;
}

// packed monotonic simplicial copy of the current factor (cholmod_l_change_factor on a copy): sizes first
// (Lx == NULL), then the arrays
int ref_to_simplicial (void *hv, int to_ll, int64_t *Lp, int64_t *Li, int64_t *Lnz, double *Lx, int64_t *lnz)
{
    ref_ctx *h = hv ;
    cholmod_factor *L2 = cholmod_l_copy_factor (h->L, &h->cm) ;
    if (!L2) return -1 ;
    if (!cholmod_l_change_factor (L2->xtype, to_ll, 0, 1, 1, L2, &h->cm)) { cholmod_l_free_factor (&L2, &h->cm) ; return -2 ; }
    int64_t n = L2->n ;
    int64_t nz = ((int64_t *) L2->p) [n] ;
    int E = (L2->xtype == CHOLMOD_COMPLEX) ? 2 : 1 ;
    *lnz = nz ;
    if (Lx)
    {
        memcpy (Lp, L2->p, (n+1) * sizeof (int64_t)) ;
        memcpy (Lnz, L2->nz, n * sizeof (int64_t)) ;
        memcpy (Li, L2->i, nz * sizeof (int64_t)) ;
        memcpy (Lx, L2->x, nz * E * sizeof (double)) ;
    }
    cholmod_l_free_factor (&L2, &h->cm) ;
    return 0 ;
}

// cholmod_l_rcond of the current factor
double ref_rcond (void *hv)
{
    ref_ctx *h = hv ;
    return cholmod_l_rcond (h->L, &h->cm) ;
}

// X = A\B through cholmod_l_solve (CHOLMOD_A); B, X: n x nrhs, ld n, xtype of L
int ref_solve (void *hv, int64_t nrhs, const double *Bx, double *Xx)
{
    ref_ctx *h = hv ;
    cholmod_factor *L = h->L ;
    int E = (L->xtype == CHOLMOD_COMPLEX) ? 2 : 1 ;
    cholmod_dense B ;
    memset (&B, 0, sizeof (B)) ;
    B.nrow = L->n ; B.ncol = nrhs ; B.d = L->n ; B.nzmax = L->n * nrhs ;
    B.x = (void *) Bx ; B.xtype = L->xtype ; B.dtype = CHOLMOD_DOUBLE ;
    double t = SuiteSparse_time ( ) ;
    cholmod_dense *X = cholmod_l_solve (CHOLMOD_A, L, &B, &h->cm) ;
    h->t_solve = SuiteSparse_time ( ) - t ;
    if (!X) return -1 ;
    memcpy (Xx, X->x, L->n * nrhs * E * sizeof (double)) ;
    cholmod_l_free_dense (&X, &h->cm) ;
    return 0 ;
}

// ---- drop-in build only (oracle/build_ref_mi355.sh: the same helper linked to libcholmod_mi355.so) --------------------
// Common->useGPU before the next analyze: 1 request the device, 0 CPU workers, -1 ask the environment (CHOLMOD_USE_GPU)
void ref_set_use_gpu (void *hv, int use) { ((ref_ctx *) hv)->cm.useGPU = use ; }
// Common->maxGpuMemBytes / maxGpuMemFraction (cholmod.h; read by cholmod_l_gpu_allocate in the reference)
void ref_set_gpu_mem (void *hv, double bytes, double fraction)
{
    ((ref_ctx *) hv)->cm.maxGpuMemBytes = (size_t) bytes ;
    ((ref_ctx *) hv)->cm.maxGpuMemFraction = fraction ;
}
void ref_get_gpu_mem (void *hv, double *out)
{
    out [0] = (double) ((ref_ctx *) hv)->cm.maxGpuMemBytes ;
    out [1] = ((ref_ctx *) hv)->cm.maxGpuMemFraction ;
}

// out[10]: Common->useGPU, L->useGPU, then the BLAS statistics of cholmod.h:966-986:
// gpu gemm calls, gpu potrf calls, gpu gemm time, gpu potrf time, assemble time,
// cpu (syrk + gemm + trsm + potrf) calls, cpu time of the same, reserved
void ref_counters (void *hv, double *out)
{
    ref_ctx *h = hv ;
    cholmod_common *cm = &h->cm ;
    out [0] = cm->useGPU ;
    out [1] = h->L ? h->L->useGPU : -1 ;
    out [2] = (double) cm->cholmod_gpu_gemm_calls ;
    out [3] = (double) cm->cholmod_gpu_potrf_calls ;
    out [4] = cm->cholmod_gpu_gemm_time ;
    out [5] = cm->cholmod_gpu_potrf_time ;
    out [6] = cm->cholmod_assemble_time ;
    out [7] = (double) (cm->cholmod_cpu_syrk_calls + cm->cholmod_cpu_gemm_calls + cm->cholmod_cpu_trsm_calls +
                        cm->cholmod_cpu_potrf_calls) ;
    out [8] = cm->cholmod_cpu_syrk_time + cm->cholmod_cpu_gemm_time + cm->cholmod_cpu_trsm_time +
              cm->cholmod_cpu_potrf_time ;
    out [9] = 0 ;
}

// free the factor now (the handle table must shrink with it)
void ref_drop_factor (void *hv)
{
    ref_ctx *h = hv ;
    cholmod_l_free_factor (&h->L, &h->cm) ;
}

// convert the factor IN PLACE to a packed monotonic simplicial one (cholmod_l_change_factor): it loses its
// supernodal arrays, so a device image must go with them
int ref_make_simplicial (void *hv, int to_ll)
{
    ref_ctx *h = hv ;
    return cholmod_l_change_factor (h->L->xtype, to_ll, 0, 1, 1, h->L, &h->cm) ? 0 : -1 ;
}

#ifdef REF_HELPER_MI355
int cholmod_mi355_live_handles (void) ;
void cholmod_mi355_calls_taken (int64_t out [4]) ;
int ref_live_handles (void) { return cholmod_mi355_live_handles ( ) ; }
void ref_calls_taken (int64_t *out) { cholmod_mi355_calls_taken (out) ; }
void cholmod_mi355_forget (cholmod_factor *L) ;
void ref_forget (void *hv) { cholmod_mi355_forget (((ref_ctx *) hv)->L) ; }
int cholmod_mi355_register_comm (cholmod_common *Common, void *comm, int rank, int world) ;
int ref_register_comm (void *hv, void *comm, int rank, int world) { return cholmod_mi355_register_comm (&((ref_ctx *) hv)->cm, comm, rank, world) ; }
int64_t cholmod_mi355_fallbacks (char *reason, size_t len) ;
int64_t ref_fallbacks (char *reason, int64_t len) { return cholmod_mi355_fallbacks (reason, (size_t) len) ; }
#else
int64_t ref_fallbacks (char *reason, int64_t len) { if (reason && len > 0) reason [0] = 0 ; return -1 ; }
int ref_register_comm (void *hv, void *comm, int rank, int world) { (void) hv ; (void) comm ; (void) rank ; (void) world ; return 0 ; }
void ref_forget (void *hv) { (void) hv ; }
int ref_live_handles (void) { return -1 ; }
void ref_calls_taken (int64_t *out) { out [0] = out [1] = out [2] = out [3] = -1 ; }
#endif

// cholmod_l_solve2 (sys, L, B) with the caller-kept X / Y / E of cholmod_solve.c:119-153 (reused from call to call and
// freed at the end of `reps` calls).  B: n x nrhs, leading dimension ldb, bxtype 1 real / 2 complex (interleaved) /
// 3 zomplex (Bx then Bz); prefer_zomplex as Common->prefer_zomplex.  Xx receives X (n x nrhs, ld n; a zomplex X as
// real part then imaginary part); *xxtype its xtype.
int ref_solve2 (void *hv, int sys, int bxtype, int64_t nrhs, const double *Bx, const double *Bz, int64_t ldb,
    int prefer_zomplex, int reps, double *Xx, int *xxtype)
{
    ref_ctx *h = hv ;
    cholmod_factor *L = h->L ;
    cholmod_dense B ;
    memset (&B, 0, sizeof (B)) ;
    B.nrow = L->n ; B.ncol = nrhs ; B.d = ldb ; B.nzmax = ldb * nrhs ;
    B.x = (void *) Bx ; B.z = (void *) Bz ; B.xtype = bxtype ; B.dtype = CHOLMOD_DOUBLE ;
    cholmod_dense *X = NULL, *Y = NULL, *E = NULL ;
    const int save = h->cm.prefer_zomplex ;
    h->cm.prefer_zomplex = prefer_zomplex ;
    int ok = 1 ;
    for (int r = 0 ; r < reps && ok ; r++)
    {
        ok = cholmod_l_solve2 (sys, L, &B, NULL, &X, NULL, &Y, &E, &h->cm) ;
    }
    h->cm.prefer_zomplex = save ;
    if (ok && X)
    {
        const size_t cnt = L->n * nrhs ;
        *xxtype = X->xtype ;
        if (X->xtype == CHOLMOD_ZOMPLEX)
        {
            memcpy (Xx, X->x, cnt * sizeof (double)) ;
            memcpy (Xx + cnt, X->z, cnt * sizeof (double)) ;
        }
        else memcpy (Xx, X->x, cnt * (X->xtype == CHOLMOD_COMPLEX ? 2 : 1) * sizeof (double)) ;
    }
    cholmod_l_free_dense (&X, &h->cm) ;
    cholmod_l_free_dense (&Y, &h->cm) ;
    cholmod_l_free_dense (&E, &h->cm) ;
    return ok ? 0 : -1 ;
}

// Common-> options the reference's test_solver varies (CHOLMOD/Tcov/t_solve.c:61-124)
int ref_set_option (void *hv, const char *name, double v)
{
    cholmod_common *cm = &((ref_ctx *) hv)->cm ;
    if (!strcmp (name, "final_asis")) cm->final_asis = (int) v ;
    else if (!strcmp (name, "final_super")) cm->final_super = (int) v ;
    else if (!strcmp (name, "final_ll")) cm->final_ll = (int) v ;
    else if (!strcmp (name, "final_pack")) cm->final_pack = (int) v ;
    else if (!strcmp (name, "final_monotonic")) cm->final_monotonic = (int) v ;
    else if (!strcmp (name, "final_resymbol")) cm->final_resymbol = (int) v ;
    else if (!strcmp (name, "nrelax0")) cm->nrelax [0] = (size_t) v ;
    else if (!strcmp (name, "nrelax1")) cm->nrelax [1] = (size_t) v ;
    else if (!strcmp (name, "nrelax2")) cm->nrelax [2] = (size_t) v ;
    else if (!strcmp (name, "zrelax0")) cm->zrelax [0] = v ;
    else if (!strcmp (name, "zrelax1")) cm->zrelax [1] = v ;
    else if (!strcmp (name, "zrelax2")) cm->zrelax [2] = v ;
    else if (!strcmp (name, "grow0")) cm->grow0 = v ;
    else if (!strcmp (name, "grow1")) cm->grow1 = v ;
    else if (!strcmp (name, "grow2")) cm->grow2 = (size_t) v ;
    else if (!strcmp (name, "nthreads_max")) cm->nthreads_max = (int) v ;
    else return -1 ;
    return 0 ;
}

// The expert sequence of the reference's raw_factor test (CHOLMOD/Tcov/t_raw_factor.c:109-340), through the public
// API: no fill-reducing ordering, alloc_factor -> etree -> postorder -> rowcolcounts -> super_symbolic ->
// super_numeric -> super_lsolve -> super_ltsolve, and -- check_errors -- every wrong-argument call of that test,
// each of which must be refused.  The matrix is the loaded one (one triangle stored, either).  The factor is kept as
// the context's factor.  out[6]: residual ||Ax-b||_inf / (||A||_inf ||x||_inf + ||b||_inf) with b = 1 + i/n,
// refusals seen, refusals expected, Common->status after super_numeric, L->minor, 1 if every step that must succeed did.
int ref_raw_factor (void *hv, int check_errors, double *out)
{
    ref_ctx *h = hv ;
    cholmod_common *cm = &h->cm ;
    if (!h->A || h->A->stype == 0) return -1 ;
    const int64_t n = h->A->nrow ;
    double zero [2] = { 0, 0 }, one [2] = { 1, 0 }, minusone [2] = { -1, 0 } ;
    int refused = 0, expected = 0, good = 1 ;
    #define MUST(e) do { if (!(e)) good = 0 ; } while (0)
    #define MUSTNOT(e) do { expected++ ; if (!(e)) refused++ ; } while (0)
    cholmod_sparse *AU = (h->A->stype > 0) ? cholmod_l_copy_sparse (h->A, cm) : cholmod_l_transpose (h->A, 2, cm) ;
    cholmod_sparse *AL = cholmod_l_transpose (AU, 2, cm) ;
    int64_t *Parent = malloc ((n + 1) * sizeof (int64_t)), *Post = malloc ((n + 1) * sizeof (int64_t)),
            *First = malloc ((n + 1) * sizeof (int64_t)), *Level = malloc ((n + 1) * sizeof (int64_t)) ;
    cholmod_l_free_factor (&h->L, cm) ;
    cholmod_factor *L = cholmod_l_allocate_factor (n, cm) ;
    h->L = L ;
    MUST (AU && AL && L && AL->stype == -1) ;
    MUST (cholmod_l_etree (AU, Parent, cm)) ;
    MUST (cholmod_l_postorder (Parent, n, NULL, Post, cm) >= 0) ;
    MUST (cholmod_l_rowcolcounts (AL, NULL, 0, Parent, Post, NULL, L->ColCount, First, Level, cm)) ;
    if (check_errors) MUSTNOT (cholmod_l_rowcolcounts (AU, NULL, 0, Parent, Post, NULL, L->ColCount, First, Level, cm)) ;
    MUST (cholmod_l_super_symbolic (AU, NULL, Parent, L, cm)) ;
    if (check_errors)
    {
        MUSTNOT (cholmod_l_super_symbolic (AU, NULL, Parent, L, cm)) ;          // L is already supernodal
        MUSTNOT (cholmod_l_super_symbolic (AL, NULL, Parent, L, cm)) ;
        MUSTNOT (cholmod_l_super_symbolic (NULL, NULL, Parent, L, cm)) ;
        MUSTNOT (cholmod_l_super_symbolic (AU, NULL, NULL, L, cm)) ;
        MUSTNOT (cholmod_l_super_symbolic (AU, NULL, Parent, NULL, cm)) ;
    }
    cm->status = CHOLMOD_OK ;
    MUST (cholmod_l_super_numeric (AL, NULL, zero, L, cm)) ;
    const int st = cm->status ;
    if (check_errors)
    {
        MUSTNOT (cholmod_l_super_numeric (AU, NULL, zero, L, cm)) ;            // upper form
        MUSTNOT (cholmod_l_super_numeric (NULL, NULL, zero, L, cm)) ;
        MUSTNOT (cholmod_l_super_numeric (AL, NULL, zero, NULL, cm)) ;
        // the refused calls must not have disturbed the factor: factorize again
        cm->status = CHOLMOD_OK ;
        MUST (cholmod_l_super_numeric (AL, NULL, zero, L, cm)) ;
        MUST (cm->status == st) ;
    }
    const int lxt = L->xtype ;
    const int E = (lxt == CHOLMOD_COMPLEX) ? 2 : 1 ;
    cholmod_dense *B = cholmod_l_zeros (n, 1, lxt + CHOLMOD_DOUBLE, cm) ;
    for (int64_t i = 0 ; i < n ; i++) ((double *) B->x) [E * i] = 1 + (double) i / (double) (n > 0 ? n : 1) ;
    cholmod_dense *W = cholmod_l_zeros (n, 1, lxt + CHOLMOD_DOUBLE, cm) ;
    cholmod_dense *X = cholmod_l_copy_dense (B, cm) ;
    double r = -1 ;
    if (st == CHOLMOD_OK)
    {
        MUST (cholmod_l_super_lsolve (L, X, W, cm)) ;
        MUST (cholmod_l_super_ltsolve (L, X, W, cm)) ;
        // R = B - A X
        cholmod_dense *R = cholmod_l_copy_dense (B, cm) ;
        cholmod_l_sdmult (AU, 0, minusone, one, X, R, cm) ;
        const double rn = cholmod_l_norm_dense (R, 0, cm), xn = cholmod_l_norm_dense (X, 0, cm),
                     bn = cholmod_l_norm_dense (B, 0, cm), an = cholmod_l_norm_sparse (AU, 0, cm) ;
        r = rn / (an * xn + bn + (n == 0 ? 1 : 0)) ;
        cholmod_l_free_dense (&R, cm) ;
        if (check_errors)
        {
            MUSTNOT (cholmod_l_super_lsolve (NULL, X, W, cm)) ;
            MUSTNOT (cholmod_l_super_ltsolve (NULL, X, W, cm)) ;
            MUSTNOT (cholmod_l_super_lsolve (L, NULL, W, cm)) ;
            MUSTNOT (cholmod_l_super_ltsolve (L, NULL, W, cm)) ;
            MUSTNOT (cholmod_l_super_lsolve (L, X, NULL, cm)) ;
            MUSTNOT (cholmod_l_super_ltsolve (L, X, NULL, cm)) ;
            if (L->maxesize > 1)
            {
                cholmod_dense *W1 = cholmod_l_zeros (1, 1, lxt + CHOLMOD_DOUBLE, cm) ;      // W is too small
                MUSTNOT (cholmod_l_super_lsolve (L, X, W1, cm)) ;
                MUSTNOT (cholmod_l_super_ltsolve (L, X, W1, cm)) ;
                cholmod_l_free_dense (&W1, cm) ;
            }
            cholmod_dense *X2 = cholmod_l_zeros (n + 1, 1, lxt + CHOLMOD_DOUBLE, cm) ;      // wrong dimensions
            MUSTNOT (cholmod_l_super_lsolve (L, X2, W, cm)) ;
            MUSTNOT (cholmod_l_super_ltsolve (L, X2, W, cm)) ;
            cholmod_l_free_dense (&X2, cm) ;
        }
        cholmod_dense *X0 = cholmod_l_zeros (n, 0, lxt + CHOLMOD_DOUBLE, cm) ;              // n-by-0 is fine
        MUST (cholmod_l_super_lsolve (L, X0, W, cm)) ;
        MUST (cholmod_l_super_ltsolve (L, X0, W, cm)) ;
        cholmod_l_free_dense (&X0, cm) ;
    }
    out [0] = r ; out [1] = refused ; out [2] = expected ; out [3] = st ; out [4] = (double) L->minor ; out [5] = good ;
    cholmod_l_free_dense (&X, cm) ; cholmod_l_free_dense (&W, cm) ; cholmod_l_free_dense (&B, cm) ;
    cholmod_l_free_sparse (&AU, cm) ; cholmod_l_free_sparse (&AL, cm) ;
    free (Parent) ; free (Post) ; free (First) ; free (Level) ;
    h->last_status = cm->status = st ;
    #undef MUST
    #undef MUSTNOT
    return 0 ;
}
