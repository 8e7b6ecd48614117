// oracle/super_oracle.c -- TEST INFRASTRUCTURE (the parity oracle), not product code.
//
// Plain-C CPU restatement of the reference's supernodal numeric hot path:
//   * left-looking supernodal Cholesky   CHOLMOD/Supernodal/t_cholmod_super_numeric_worker.c:133-1279
//   * supernodal forward / backward solve CHOLMOD/Supernodal/t_cholmod_super_solve_worker.c:20-303,309-580
// for int64 indices, double precision, real (E=1) and complex-interleaved (E=2)
// factors.  The BLAS/LAPACK calls of the reference (dsyrk/zherk, dgemm/zgemm,
// dpotrf/zpotrf, dtrsm/ztrsm, dtrsv/dtrsm, dgemv/dgemm) are restated as plain loops
// with the documented BLAS semantics (an external Fortran BLAS is a dependency of the
// reference, not vendored in it: SURVEY.md section 8c).
//
// Pinned (tests/test_oracle.py) against the reference itself built from its own
// sources (oracle/_ref, oracle/build_ref.sh) on the reference's demo/Tcov fixtures and
// on the synthetic configs, and against committed fixtures in tests/golden/.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef int64_t Int ;
#define EMPTY (-1)

#define ORACLE_OK 0
#define ORACLE_NOT_POSDEF 1
#define ORACLE_OUT_OF_MEMORY (-2)
#define ORACLE_INVALID (-4)

//------------------------------------------------------------------------------
// dense kernels (column-major, BLAS semantics)
//------------------------------------------------------------------------------

// C (n x n, lower, ldc) = A * A^H, A is n x k (lda).  dsyrk/zherk ("L","N"), alpha 1 beta 0
// (t_cholmod_super_numeric_worker.c:769-793).  zherk: diagonal imaginary part set to 0.
static void x_syrk (int E, Int n, Int k, const double *A, Int lda, double *C, Int ldc)
{
    for (Int j = 0 ; j < n ; j++)
        for (Int i = j ; i < n ; i++)
            for (int e = 0 ; e < E ; e++) C [E*(i + j*ldc) + e] = 0 ;
    for (Int p = 0 ; p < k ; p++)
    {
        const double *a = A + E*p*lda ;
        for (Int j = 0 ; j < n ; j++)
        {
            if (E == 1)
            {
                double ajp = a [j] ;
                double *c = C + j*ldc ;
                for (Int i = j ; i < n ; i++) c [i] += a [i] * ajp ;
            }
            else
            {
                double br = a [2*j], bi = -a [2*j+1] ;   // conj (A(j,p))
                double *c = C + 2*j*ldc ;
                for (Int i = j ; i < n ; i++)
                {
                    double ar = a [2*i], ai = a [2*i+1] ;
                    c [2*i]   += ar*br - ai*bi ;
                    c [2*i+1] += ar*bi + ai*br ;
                }
            }
        }
    }
    if (E == 2) for (Int j = 0 ; j < n ; j++) C [2*(j + j*ldc) + 1] = 0 ;
}

// C (m x n, ldc) = A (m x k, lda) * B (n x k, ldb)^H.  dgemm/zgemm ("N","C"), alpha 1 beta 0
// (t_cholmod_super_numeric_worker.c:824-873)
static void x_gemm_nc (int E, Int m, Int n, Int k, const double *A, Int lda,
    const double *B, Int ldb, double *C, Int ldc)
{
    for (Int j = 0 ; j < n ; j++)
        for (Int i = 0 ; i < m ; i++)
            for (int e = 0 ; e < E ; e++) C [E*(i + j*ldc) + e] = 0 ;
    for (Int p = 0 ; p < k ; p++)
    {
        const double *a = A + E*p*lda ;
        const double *b = B + E*p*ldb ;
        for (Int j = 0 ; j < n ; j++)
        {
            if (E == 1)
            {
                double bjp = b [j] ;
                double *c = C + j*ldc ;
                for (Int i = 0 ; i < m ; i++) c [i] += a [i] * bjp ;
            }
            else
            {
                double br = b [2*j], bi = -b [2*j+1] ;
                double *c = C + 2*j*ldc ;
                for (Int i = 0 ; i < m ; i++)
                {
                    double ar = a [2*i], ai = a [2*i+1] ;
                    c [2*i]   += ar*br - ai*bi ;
                    c [2*i+1] += ar*bi + ai*br ;
                }
            }
        }
    }
}

// in-place lower Cholesky of the n x n block A (lda): dpotrf/zpotrf ("L")
// (t_cholmod_super_numeric_worker.c:1023-1048).  Returns LAPACK info: 0, or the
// 1-based index of the first pivot that is <= 0 or NaN (dpotf2/zpotf2 semantics).
// Right-looking column algorithm; the strictly upper part is not referenced.  For
// complex input the imaginary part of the diagonal is ignored and set to zero.
static Int x_potrf (int E, Int n, double *A, Int lda)
{
    for (Int j = 0 ; j < n ; j++)
    {
        double ajj = A [E*(j + j*lda)] ;
        if (!(ajj > 0.0))        // catches <= 0 and NaN
        {
            return (j+1) ;
        }
        ajj = sqrt (ajj) ;
        A [E*(j + j*lda)] = ajj ;
        if (E == 2) A [2*(j + j*lda) + 1] = 0 ;
        double *cj = A + E*j*lda ;
        for (Int i = j+1 ; i < n ; i++)
        {
            cj [E*i] /= ajj ;
            if (E == 2) cj [2*i+1] /= ajj ;
        }
        // trailing update: A (i,k) -= L (i,j) * conj (L (k,j)) for j < k <= i
        for (Int k = j+1 ; k < n ; k++)
        {
            double *ck = A + E*k*lda ;
            if (E == 1)
            {
                double lkj = cj [k] ;
                for (Int i = k ; i < n ; i++) ck [i] -= cj [i] * lkj ;
            }
            else
            {
                double br = cj [2*k], bi = -cj [2*k+1] ;
                for (Int i = k ; i < n ; i++)
                {
                    double ar = cj [2*i], ai = cj [2*i+1] ;
                    ck [2*i]   -= ar*br - ai*bi ;
                    ck [2*i+1] -= ar*bi + ai*br ;
                }
                ck [2*k+1] = 0 ;     // diagonal stays real
            }
        }
    }
    return (0) ;
}

// B (m x n, ldb) = B * inv (A)^H, A n x n lower non-unit (lda): dtrsm/ztrsm ("R","L","C","N")
// (t_cholmod_super_numeric_worker.c:1175-1208)
static void x_trsm_rlcn (int E, Int m, Int n, const double *A, Int lda, double *B, Int ldb)
{
    // X * L^H = B  =>  for j = 0..n-1: X(:,j) = (B(:,j) - sum_{k<j} X(:,k) conj(L(j,k))) / conj(L(j,j))
    for (Int j = 0 ; j < n ; j++)
    {
        double *bj = B + E*j*ldb ;
        for (Int k = 0 ; k < j ; k++)
        {
            const double *bk = B + E*k*ldb ;
            if (E == 1)
            {
                double ljk = A [j + k*lda] ;
                if (ljk != 0) for (Int i = 0 ; i < m ; i++) bj [i] -= bk [i] * ljk ;
            }
            else
            {
                double br = A [2*(j + k*lda)], bi = -A [2*(j + k*lda)+1] ;
                for (Int i = 0 ; i < m ; i++)
                {
                    double ar = bk [2*i], ai = bk [2*i+1] ;
                    bj [2*i]   -= ar*br - ai*bi ;
                    bj [2*i+1] -= ar*bi + ai*br ;
                }
            }
        }
        if (E == 1)
        {
            double d = A [j + j*lda] ;
            for (Int i = 0 ; i < m ; i++) bj [i] /= d ;
        }
        else
        {
            // divide by conj (L(j,j)); general complex division
            double dr = A [2*(j + j*lda)], di = -A [2*(j + j*lda)+1] ;
            double den = dr*dr + di*di ;
            for (Int i = 0 ; i < m ; i++)
            {
                double ar = bj [2*i], ai = bj [2*i+1] ;
                bj [2*i]   = (ar*dr + ai*di) / den ;
                bj [2*i+1] = (ai*dr - ar*di) / den ;
            }
        }
    }
}

//------------------------------------------------------------------------------
// oracle_super_numeric
//------------------------------------------------------------------------------

// A = tril (P A P') in CSC form (stype < 0), packed (Anz NULL) or unpacked; complex
// values interleaved in Ax (a zomplex A must be interleaved by the caller, which is
// what the zd_ instantiation does on the fly, t_cholmod_super_numeric_worker.c:69-83).
// Lx (E*xsize doubles) is overwritten.  Returns ORACLE_OK / ORACLE_NOT_POSDEF / <0.
// With F != NULL (stype == 0): A is n x ncol, F = A' or A(:,f)' is ncol x n (conjugate transpose for complex
// values, cholmod_factorize.c:203-210,242-250), and the matrix factorized is A*F + beta*I (worker :471-506).
int oracle_super_numeric_f
(
    int E,                      // 1 real, 2 complex
    Int n, Int nsuper,
    const Int *Super, const Int *Lpi, const Int *Lpx, const Int *Ls,
    Int xsize, Int maxcsize,
    const Int *Ap, const Int *Ai, const Int *Anz, const double *Ax,
    const Int *Fp, const Int *Fi, const Int *Fnz, const double *Fx,
    double beta_re, int quick_return_if_not_posdef,
    double *Lx, Int *minor_out
)
{
    if (E != 1 && E != 2) return ORACLE_INVALID ;
    int status = ORACLE_OK ;
    Int *SuperMap = malloc ((n+1) * sizeof (Int)) ;
    Int *RelativeMap = malloc ((n+1) * sizeof (Int)) ;
    Int *Map = malloc ((n+1) * sizeof (Int)) ;
    Int *Head = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Next = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Lpos = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Next_save = malloc ((nsuper+1) * sizeof (Int)) ;
    Int *Lpos_save = malloc ((nsuper+1) * sizeof (Int)) ;
    double *C = malloc ((size_t) (maxcsize > 0 ? maxcsize : 1) * E * sizeof (double)) ;
    if (!SuperMap || !RelativeMap || !Map || !Head || !Next || !Lpos || !Next_save
        || !Lpos_save || !C)
    {
        free (SuperMap) ; free (RelativeMap) ; free (Map) ; free (Head) ; free (Next) ;
        free (Lpos) ; free (Next_save) ; free (Lpos_save) ; free (C) ;
        return ORACLE_OUT_OF_MEMORY ;
    }

    // SuperMap (cholmod_super_numeric.c:276-285), Map cleared (:264-269), Head empty
    for (Int s = 0 ; s < nsuper ; s++)
        for (Int k = Super [s] ; k < Super [s+1] ; k++) SuperMap [k] = s ;
    for (Int i = 0 ; i < n ; i++) Map [i] = EMPTY ;
    for (Int s = 0 ; s <= nsuper ; s++) Head [s] = EMPTY ;

    int repeat_supernode = 0 ;
    Int nscol_new = 0 ;
    Int minor = n ;

    for (Int s = 0 ; s < nsuper ; s++)          // worker :335
    {
        Int k1 = Super [s], k2 = Super [s+1] ;
        Int nscol = k2 - k1 ;
        Int psi = Lpi [s], psx = Lpx [s], psend = Lpi [s+1] ;
        Int nsrow = psend - psi ;

        // zero the supernode (:360-379)
        memset (Lx + E*psx, 0, (size_t) (nsrow*nscol) * E * sizeof (double)) ;

        // Map (:392-399)
        for (Int k = 0 ; k < nsrow ; k++) Map [Ls [psi + k]] = k ;

        // copy A*F into the supernode (stype == 0, :471-506): column k = sum over j in F(:,k) of A(:,j)*F(j,k)
        for (Int k = k1 ; Fp != NULL && k < k2 ; k++)
        {
            Int pf = Fp [k] ;
            Int pfend = (Fnz == NULL) ? Fp [k+1] : pf + Fnz [k] ;
            for ( ; pf < pfend ; pf++)
            {
                Int j = Fi [pf] ;
                double fr = Fx [E*pf], fi = (E == 2) ? Fx [2*pf+1] : 0 ;
                Int p = Ap [j] ;
                Int pend = (Anz == NULL) ? Ap [j+1] : p + Anz [j] ;
                for ( ; p < pend ; p++)
                {
                    Int i = Ai [p] ;
                    if (i < k) continue ;
                    Int imap = Map [i] ;
                    if (imap < 0 || imap >= nsrow) continue ;
                    Int q = imap + (psx + (k-k1)*nsrow) ;
                    if (E == 1) Lx [q] += Ax [p] * fr ;
                    else
                    {
                        // L_MULTADD: Lx [q] += Ax [p] * fjk (complex product, no conjugate)
                        Lx [2*q]   += Ax [2*p] * fr - Ax [2*p+1] * fi ;
                        Lx [2*q+1] += Ax [2*p+1] * fr + Ax [2*p] * fi ;
                    }
                }
            }
        }

        // copy A into the supernode (:438-470)
        for (Int k = k1 ; Fp == NULL && k < k2 ; k++)
        {
            Int p = Ap [k] ;
            Int pend = (Anz == NULL) ? Ap [k+1] : p + Anz [k] ;
            for ( ; p < pend ; p++)
            {
                Int i = Ai [p] ;
                if (i >= k)
                {
                    Int imap = Map [i] ;
                    if (imap >= 0 && imap < nsrow)
                    {
                        Int q = imap + (psx + (k-k1)*nsrow) ;
                        for (int e = 0 ; e < E ; e++) Lx [E*q+e] = Ax [E*p+e] ;
                    }
                }
            }
        }

        // add beta to the diagonal (:510-520)
        if (beta_re != 0.0)
        {
            Int pk = psx ;
            for (Int k = k1 ; k < k2 ; k++) { Lx [E*pk] += beta_re ; pk += nsrow + 1 ; }
        }

        // save / restore the descendant list (:531-551)
        if (!repeat_supernode)
        {
            for (Int d = Head [s] ; d != EMPTY ; d = Next [d])
            { Lpos_save [d] = Lpos [d] ; Next_save [d] = Next [d] ; }
        }
        else
        {
            for (Int d = Head [s] ; d != EMPTY ; d = Next [d])
            { Lpos [d] = Lpos_save [d] ; Next [d] = Next_save [d] ; }
        }

        // update s with each pending descendant d (:585-969)
        Int dnext = Head [s] ;
        while (dnext != EMPTY)
        {
            Int d = dnext ;
            Int kd1 = Super [d], kd2 = Super [d+1] ;
            Int ndcol = kd2 - kd1 ;
            Int pdi = Lpi [d], pdx = Lpx [d], pdend = Lpi [d+1] ;
            Int ndrow = pdend - pdi ;

            Int p = Lpos [d] ;
            Int pdi1 = pdi + p, pdx1 = pdx + p ;
            Int pdi2 ;
            for (pdi2 = pdi1 ; pdi2 < pdend && Ls [pdi2] < k2 ; pdi2++) ;
            Int ndrow1 = pdi2 - pdi1 ;
            Int ndrow2 = pdend - pdi1 ;
            Int ndrow3 = ndrow2 - ndrow1 ;

            // C1 = L1*L1' (lower), C2 = L2*L1' (:769-873)
            x_syrk (E, ndrow1, ndcol, Lx + E*pdx1, ndrow, C, ndrow2) ;
            if (ndrow3 > 0)
            {
                x_gemm_nc (E, ndrow3, ndrow1, ndcol, Lx + E*(pdx1 + ndrow1), ndrow,
                    Lx + E*pdx1, ndrow, C + E*ndrow1, ndrow2) ;
            }

            // relative map (:899-903)
            for (Int i = 0 ; i < ndrow2 ; i++) RelativeMap [i] = Map [Ls [pdi1 + i]] ;

            // assemble (:917-930)
            for (Int j = 0 ; j < ndrow1 ; j++)
            {
                Int px = psx + RelativeMap [j] * nsrow ;
                for (Int i = j ; i < ndrow2 ; i++)
                {
                    Int q = px + RelativeMap [i] ;
                    for (int e = 0 ; e < E ; e++) Lx [E*q+e] -= C [E*(i + ndrow2*j)+e] ;
                }
            }

            // next ancestor of d (:950-967)
            dnext = Next [d] ;
            if (!repeat_supernode)
            {
                Lpos [d] = pdi2 - pdi ;
                if (Lpos [d] < ndrow)
                {
                    Int dancestor = SuperMap [Ls [pdi2]] ;
                    Next [d] = Head [dancestor] ;
                    Head [dancestor] = d ;
                }
            }
        }

        // factorize the diagonal block (:1004-1058)
        Int nscol2 = repeat_supernode ? nscol_new : nscol ;
        Int info = x_potrf (E, nscol2, Lx + E*psx, nsrow) ;

        if (repeat_supernode)                       // (:1064-1077)
        {
            info = 0 ;
            memset (Lx + E*(psx + nsrow*nscol_new), 0,
                (size_t) (nsrow*(nscol - nscol_new)) * E * sizeof (double)) ;
        }

        if (info != 0)                              // (:1084-1146)
        {
            status = ORACLE_NOT_POSDEF ;
            minor = k1 + info - 1 ;
            for (Int ss = s+1 ; ss < nsuper ; ss++) Head [ss] = EMPTY ;
            memset (Lx + E*psx, 0, (size_t) (xsize - psx) * E * sizeof (double)) ;
            if (info == 1 || quick_return_if_not_posdef)
            {
                Head [s] = EMPTY ;
                goto done ;
            }
            repeat_supernode = 1 ;
            s-- ;
            nscol_new = info - 1 ;
            continue ;
        }

        // sub-diagonal block and parent link (:1152-1236)
        Int nsrow2 = nsrow - nscol2 ;
        if (nsrow2 > 0)
        {
            x_trsm_rlcn (E, nsrow2, nscol2, Lx + E*psx, nsrow, Lx + E*(psx + nscol2), nsrow) ;
            if (!repeat_supernode)
            {
                Lpos [s] = nscol ;
                Int sparent = SuperMap [Ls [psi + nscol]] ;
                Next [s] = Head [sparent] ;
                Head [sparent] = s ;
            }
        }
        Head [s] = EMPTY ;
        if (repeat_supernode) goto done ;           // (:1253-1265)
    }
    minor = n ;     // (:1269) only reached when positive definite

done:
    *minor_out = minor ;
    free (SuperMap) ; free (RelativeMap) ; free (Map) ; free (Head) ; free (Next) ;
    free (Lpos) ; free (Next_save) ; free (Lpos_save) ; free (C) ;
    return status ;
}

//------------------------------------------------------------------------------
// oracle_super_lsolve / oracle_super_ltsolve
//------------------------------------------------------------------------------

// X (n x nrhs, ld d, E doubles per entry) <- L \ X  (t_cholmod_super_solve_worker.c:20-303)
int oracle_super_lsolve (int E, Int nsuper, const Int *Super, const Int *Lpi,
    const Int *Lpx, const Int *Ls, const double *Lx, Int nrhs, double *X, Int d)
{
    for (Int s = 0 ; s < nsuper ; s++)
    {
        Int k1 = Super [s], k2 = Super [s+1] ;
        Int psi = Lpi [s], psend = Lpi [s+1], psx = Lpx [s] ;
        Int nsrow = psend - psi, nscol = k2 - k1 ;
        Int ps2 = psi + nscol ;
        const double *L1 = Lx + E*psx ;
        for (Int j = 0 ; j < nrhs ; j++)
        {
            double *x = X + E*j*d ;
            // x1 = L1 \ x1  (dtrsv/dtrsm "L","N","N", :93,:204)
            for (Int c = 0 ; c < nscol ; c++)
            {
                double *xc = x + E*(k1 + c) ;
                const double *col = L1 + E*c*nsrow ;
                if (E == 1)
                {
                    xc [0] /= col [c] ;
                    double v = xc [0] ;
                    for (Int r = c+1 ; r < nscol ; r++) x [k1 + r] -= col [r] * v ;
                    // E = E - L2*x1, scattered straight back (:99,:213,:165,:292)
                    for (Int r = nscol ; r < nsrow ; r++) x [Ls [psi + r]] -= col [r] * v ;
                }
                else
                {
                    double dr = col [2*c], di = col [2*c+1] ;
                    double den = dr*dr + di*di ;
                    double ar = xc [0], ai = xc [1] ;
                    double vr = (ar*dr + ai*di) / den, vi = (ai*dr - ar*di) / den ;
                    xc [0] = vr ; xc [1] = vi ;
                    for (Int r = c+1 ; r < nsrow ; r++)
                    {
                        Int i = (r < nscol) ? (k1 + r) : Ls [psi + r] ;
                        double lr = col [2*r], li = col [2*r+1] ;
                        x [2*i]   -= lr*vr - li*vi ;
                        x [2*i+1] -= lr*vi + li*vr ;
                    }
                }
            }
        }
        (void) ps2 ;
    }
    return 0 ;
}

// X <- L^H \ X  (t_cholmod_super_solve_worker.c:309-580)
int oracle_super_ltsolve (int E, Int nsuper, const Int *Super, const Int *Lpi,
    const Int *Lpx, const Int *Ls, const double *Lx, Int nrhs, double *X, Int d)
{
    for (Int s = nsuper-1 ; s >= 0 ; s--)
    {
        Int k1 = Super [s], k2 = Super [s+1] ;
        Int psi = Lpi [s], psend = Lpi [s+1], psx = Lpx [s] ;
        Int nsrow = psend - psi, nscol = k2 - k1 ;
        const double *L1 = Lx + E*psx ;
        for (Int j = 0 ; j < nrhs ; j++)
        {
            double *x = X + E*j*d ;
            // x1 = x1 - L2^H * E ; then x1 = L1^H \ x1 : column c from last to first
            for (Int c = nscol-1 ; c >= 0 ; c--)
            {
                const double *col = L1 + E*c*nsrow ;
                double *xc = x + E*(k1 + c) ;
                if (E == 1)
                {
                    double acc = xc [0] ;
                    for (Int r = c+1 ; r < nsrow ; r++)
                    {
                        Int i = (r < nscol) ? (k1 + r) : Ls [psi + r] ;
                        acc -= col [r] * x [i] ;
                    }
                    xc [0] = acc / col [c] ;
                }
                else
                {
                    double ar = xc [0], ai = xc [1] ;
                    for (Int r = c+1 ; r < nsrow ; r++)
                    {
                        Int i = (r < nscol) ? (k1 + r) : Ls [psi + r] ;
                        double lr = col [2*r], li = -col [2*r+1] ;   // conj
                        ar -= lr * x [2*i] - li * x [2*i+1] ;
                        ai -= lr * x [2*i+1] + li * x [2*i] ;
                    }
                    double dr = col [2*c], di = -col [2*c+1] ;       // conj (L(c,c))
                    double den = dr*dr + di*di ;
                    xc [0] = (ar*dr + ai*di) / den ;
                    xc [1] = (ai*dr - ar*di) / den ;
                }
            }
        }
    }
    return 0 ;
}

// stype < 0: A = tril (P A P'), F not used
int oracle_super_numeric
(
    int E, Int n, Int nsuper,
    const Int *Super, const Int *Lpi, const Int *Lpx, const Int *Ls,
    Int xsize, Int maxcsize,
    const Int *Ap, const Int *Ai, const Int *Anz, const double *Ax,
    double beta_re, int quick_return_if_not_posdef,
    double *Lx, Int *minor_out
)
{
    return oracle_super_numeric_f (E, n, nsuper, Super, Lpi, Lpx, Ls, xsize, maxcsize, Ap, Ai, Anz, Ax,
        NULL, NULL, NULL, NULL, beta_re, quick_return_if_not_posdef, Lx, minor_out) ;
}
