/* cholmod_mi355.h -- the CHOLMOD side of the drop-in: the calls a CHOLMOD build adds to route its supernodal
 * numeric factorization and supernodal triangular solves to libmi355_cholmod.so (include/mi355_cholmod.h).
 *
 * Implemented in suitesparse_amd/csrc/cholmod_mi355.c, which is compiled INTO CHOLMOD (it includes cholmod.h) with
 * -DCHOLMOD_HAS_MI355; INTEGRATION.md lists the insertion points, oracle/build_ref_mi355.sh applies exactly those
 * insertions to a build of the reference and tests/test_cholmod_dropin.py runs cholmod_l_analyze /
 * cholmod_l_factorize / cholmod_l_solve through it.  int64 (cholmod_l_*) and double only, like the reference's own
 * GPU path (CHOLMOD/Supernodal/t_cholmod_super_numeric_worker.c:174); everything else stays on the CPU workers.
 */
#ifndef CHOLMOD_MI355_H
#define CHOLMOD_MI355_H

#include "cholmod.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Inserted in cholmod_l_super_symbolic2 where a build without CUDA sets Common->useGPU = 0
 * (CHOLMOD/Supernodal/cholmod_super_symbolic.c:317-320): the reference's selection logic of :239-315 for this
 * device -- Common->useGPU == EMPTY asks the environment (CHOLMOD_USE_GPU), a request is confirmed by
 * mi355_probe() (replaces cholmod_l_gpu_probe), the outcome is cached in L->useGPU.  No device memory is
 * reserved here (replaces cholmod_l_gpu_allocate: the factor handle allocates what it needs). */
void cholmod_mi355_select (cholmod_sparse *A, int for_whom, cholmod_factor *L, cholmod_common *Common) ;

/* Inserted in cholmod_l_super_numeric before the worker switch (CHOLMOD/Supernodal/cholmod_super_numeric.c:295):
 * returns TRUE if the call was taken over -- *ok is then cholmod_super_numeric's return value, L->x, L->minor and
 * Common->status are set as the worker would (t_cholmod_super_numeric_worker.c:1084-1146,1269) -- and FALSE if the
 * CPU worker must run (GPU not selected, single precision). */
int cholmod_mi355_super_numeric (cholmod_sparse *A, cholmod_sparse *F, double beta [2], cholmod_factor *L,
    cholmod_common *Common, int *ok) ;

/* Inserted in cholmod_l_factorize_p before it builds S = tril (P*A*P') (CHOLMOD/Cholesky/cholmod_factorize.c:181-251;
 * for a lower-stored A that is two cholmod_ptranspose calls, :228-238): TRUE = hand A itself to cholmod_super_numeric,
 * the device places its entries through L->Perm (mi355_factor_set_matrix_perm) -- granted exactly when
 * cholmod_mi355_super_numeric will take the call over (same gate), A is stored lower, real or complex, L is permuted,
 * and nobody needs S afterwards (cholmod_resymbol_noperm, :275-282).  cholmod_mi355_defer_clear, inserted after the
 * cholmod_super_numeric call, drops a grant the wrapper never consumed (it returned early on invalid input). */
int cholmod_mi355_defer_permutation (cholmod_sparse *A, cholmod_factor *L, cholmod_common *Common) ;
void cholmod_mi355_defer_clear (void) ;

/* Inserted in cholmod_l_super_lsolve / _ltsolve before their worker switches
 * (CHOLMOD/Supernodal/cholmod_super_solve.c:114, :215): lower != 0 solves L x = b, else L' x = b, in X.
 * Returns TRUE if taken over (*ok = the return value), FALSE if L has no device image. */
int cholmod_mi355_super_solve (cholmod_factor *L, cholmod_dense *X, int lower, cholmod_common *Common, int *ok) ;

/* Inserted at the head of the supernodal branch of cholmod_l_solve2 (CHOLMOD/Cholesky/cholmod_solve.c:656-717: Y = P*B,
 * cholmod_super_lsolve, cholmod_super_ltsolve, X = P'*Y): the whole branch on the device -- B goes down once, X comes
 * back once, Y and E are not needed.  sys = CHOLMOD_A ... CHOLMOD_Lt; B real or complex under a real or complex L (a
 * complex B under a real L is the reference's "dual" case, :668).  Returns TRUE if taken over (*ok = solve2's return
 * value, X filled), FALSE if the host branch must run (no device image, zomplex B or X, L->minor < n) -- its two
 * sweeps are then still taken over one by one by cholmod_mi355_super_solve. */
int cholmod_mi355_solve2 (int sys, cholmod_factor *L, cholmod_dense *B, cholmod_dense *X, cholmod_common *Common,
    int *ok) ;

/* Inserted where a factor loses its supernodal arrays: cholmod_l_to_simplicial_sym
 * (CHOLMOD/Utility/t_cholmod_free_factor.c:22, reached from cholmod_l_free_factor and cholmod_l_change_factor) and
 * super_num_to_simplicial_num (CHOLMOD/Utility/t_cholmod_change_factor.c:1046).  Destroys the device image of L,
 * if any (replaces cholmod_l_gpu_deallocate for this factor). */
void cholmod_mi355_forget (cholmod_factor *L) ;

/* number of factors that currently have a device image (tests: teardown really happens) */
int cholmod_mi355_live_handles (void) ;

/* calls taken over since the library was loaded: [0] cholmod_super_numeric, [1] single sweeps (cholmod_super_lsolve /
 * _ltsolve), [2] whole cholmod_solve2 branches, [3] those of [0] that got A itself and placed it through L->Perm on the
 * device (tests: the device path, not a silent CPU one, produced a result) */
void cholmod_mi355_calls_taken (int64_t out [4]) ;

/* Soft fallback (the reference never fails a factorization because of its GPU: gpu_init failing turns the GPU off for the
 * call, CHOLMOD/Supernodal/t_cholmod_super_numeric_worker.c:233-244, CHOLMOD/GPU/t_cholmod_gpu.c:107-118): when the device
 * cannot take a factorization -- the factor does not fit the memory the device has free or Common->maxGpuMemBytes /
 * maxGpuMemFraction allow (environment CHOLMOD_GPU_MEM_BYTES / _FRACTION read with CHOLMOD_USE_GPU, as
 * CHOLMOD/Supernodal/cholmod_super_symbolic.c:250-289 does), an allocation fails, the device run fails -- L->useGPU is
 * set to 0 and the unchanged CPU worker factorizes, silently as in the reference.  Returns how often that has happened
 * since the library was loaded; reason (may be NULL): the engine's message of the last time. */
int64_t cholmod_mi355_fallbacks (char *reason, size_t len) ;

/* N ranks (no counterpart in the reference, which knows one device: CHOLMOD/GPU/cholmod_gpu.c:138-142).  The application
 * runs one process or thread per GPU, each with its own cholmod_common, the same matrix and the same analysis, and
 * registers its rank's engine communicator (a mi355_comm* from mi355_comm_create_rccl, mi355_comm_adopt_rccl or
 * mi355_comm_create, include/mi355_cholmod.h) for its Common.  From then on cholmod_l_factorize with that Common runs
 * mi355_factor_partition (once per factor) + mi355_factor_numeric_multi + mi355_factor_gather_multi: disjoint subtrees on
 * the ranks, the top supernodes distributed, and every rank's L->x = the complete factor.  Every rank must make the same
 * CHOLMOD calls in the same order; a device failure is CHOLMOD_GPU_PROBLEM on the rank that has it (no soft fallback:
 * the other ranks wait in their collectives).  cholmod_l_solve afterwards runs on each rank's host copy of L.
 * comm == NULL removes the registration.  Returns TRUE / FALSE (bad arguments, out of memory). */
int cholmod_mi355_register_comm (cholmod_common *Common, void *comm, int rank, int world) ;

#ifdef __cplusplus
}
#endif
#endif
