/* mi355_symbolic.h -- host-side symbolic analysis for stand-alone use of the engine.
 *
 * Inside CHOLMOD the symbolic factor comes from the UNCHANGED reference
 * (cholmod_analyze -> cholmod_super_symbolic2) and is handed to mi355_factor_create.
 * Callers without CHOLMOD (bench.py, the multi-GPU driver) get the same structure from
 * these functions, which restate -- pure integer host C++, bit-exact for a given
 * permutation --
 *   etree            CHOLMOD/Cholesky/cholmod_etree.c           (cholmod_analyze.c:333)
 *   postorder        CHOLMOD/Cholesky/cholmod_postorder.c:131-300, combined with the
 *                    fill-reducing ordering as cholmod_analyze.c:828-880
 *   column counts    CHOLMOD/Cholesky/cholmod_rowcolcounts.c (fl, lnz at :511-524)
 *   relaxed supernodes + Ls + maxcsize/maxesize
 *                    CHOLMOD/Supernodal/cholmod_super_symbolic.c:391-928
 *   S = tril(P A P') CHOLMOD/Cholesky/cholmod_factorize.c:225-236
 * plus a geometric nested-dissection ordering for the regular-grid benchmark problems
 * (the reference gets its orderings from METIS/NESDIS; any permutation may be passed in).
 */
#ifndef MI355_SYMBOLIC_H
#define MI355_SYMBOLIC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355_symbolic_s mi355_symbolic;

/* Nested-dissection permutation of an nx x ny x nz grid (x fastest) with `dof`
 * unknowns per node: perm[k] = original index of the k-th pivot.  nz = 1 for 2D. */
int mi355_nd_grid_order(int64_t nx, int64_t ny, int64_t nz, int64_t dof, int64_t *perm);

/* Analyze the symmetric matrix whose LOWER triangle is (Ap, Ai) [CSC, stype = -1,
 * sorted or not] under the fill-reducing permutation perm (NULL = natural), then
 * postorder as cholmod_analyze does.  nrelax/zrelax: Common->nrelax / zrelax
 * (NULL = the defaults {4,16,48} / {0.8,0.1,0.05}, t_cholmod_defaults.c). */
int mi355_symbolic_analyze(int64_t n, const int64_t *Ap, const int64_t *Ai,
                           const int64_t *perm, const int64_t *nrelax, const double *zrelax,
                           mi355_symbolic **out);
int mi355_symbolic_free(mi355_symbolic *S);

/* iout[8]: n nsuper ssize xsize maxcsize maxesize nnz(S) reserved ; dout[2]: fl lnz */
int mi355_symbolic_info(const mi355_symbolic *S, int64_t *iout, double *dout);

/* copy out Perm[n], ColCount[n], Parent[n] (etree of the postordered matrix),
 * Super[nsuper+1], Lpi[nsuper+1], Lpx[nsuper+1], Ls[ssize]; NULL pointers are skipped */
int mi355_symbolic_get(const mi355_symbolic *S, int64_t *Perm, int64_t *ColCount, int64_t *Parent,
                       int64_t *Super, int64_t *Lpi, int64_t *Lpx, int64_t *Ls);

/* S = tril(P A P') for numeric values Ax (real) aligned with (Ap, Ai): fills
 * Sp[n+1], Si[nnz], Sx[nnz], sorted columns. */
int mi355_symbolic_permute(const mi355_symbolic *S, const int64_t *Ap, const int64_t *Ai,
                           const double *Ax, int64_t *Sp, int64_t *Si, double *Sx);

#ifdef __cplusplus
}
#endif
#endif
