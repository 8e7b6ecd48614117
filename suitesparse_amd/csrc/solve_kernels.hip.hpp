// solve_kernels.hip.hpp -- supernodal forward/backward triangular solves on the
// device-resident factor (real double).  Replaces the gather -> TRSV/TRSM ->
// GEMV/GEMM -> scatter sweep of CHOLMOD/Supernodal/t_cholmod_super_solve_worker.c:
// 20-303 (forward) and :309-580 (backward).  HBM-bound: every entry of Lx is read once
// per sweep (2*E*xsize*8 bytes per solve).
//
// The sweep is level-scheduled with the same 64-column blocks as the factorization
// (one PotrfTask per block).  The diagonal-block solves use the block inverses kept
// from the factorization (x1 <- inv(L11) x1 is a dense mat-vec, no dependent chain);
// the off-diagonal part is a tall GEMV/GEMM per block, one workgroup per SOLVE_ROWS rows:
//   forward :  X[rows below] -= L[rows below, block] * x1        (scatter, atomic)
//   backward:  x1 -= L[rows below, block]^T * X[rows below]      (gather, atomic on x1)
// No workspace E (cholmod_super_solve.c:92) is needed: gather/scatter go straight
// through the row list Ls.
#pragma once
#include "kernels.hip.hpp"

namespace mi355 {

constexpr int SOLVE_NR = 4;      // right-hand sides per register pass

// x1 <- inv(L11) x1  (TRANS = false)   or   x1 <- inv(L11)^T x1  (TRANS = true)
template <bool TRANS>
__global__ __launch_bounds__(256) void k_solve_diag(
    const PotrfTask *__restrict__ tasks, const double *__restrict__ Dinv,
    double *__restrict__ X, int64_t ldx, int nrhs)
{
    __shared__ double Ds[NB][NB + 1];
    __shared__ double xs[NB][16 + 1];
    const PotrfTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, nb = t.nb;
    const double *__restrict__ D = Dinv + t.dinv_off;
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int i = idx % nb, j = idx / nb;
        Ds[i][j] = D[idx];
    }
    double *__restrict__ x1 = X + t.k1 + t.col0;
    for (int j0 = 0; j0 < nrhs; j0 += 16) {
        const int nj = min(16, nrhs - j0);
        __syncthreads();
        for (int idx = tid; idx < nb * nj; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            xs[i][j] = x1[i + (int64_t)(j0 + j) * ldx];
        }
        __syncthreads();
        for (int idx = tid; idx < nb * nj; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            double acc = 0.0;
            if (!TRANS) { for (int k = 0; k <= i; k++) acc += Ds[i][k] * xs[k][j]; }
            else        { for (int k = i; k < nb; k++) acc += Ds[k][i] * xs[k][j]; }
            x1[i + (int64_t)(j0 + j) * ldx] = acc;
        }
    }
}

// locate the block (task) of a row tile
__device__ __forceinline__ int find_solve_task(const PotrfTask *__restrict__ tasks, int ntasks, int64_t tile)
{
    int lo = 0, hi = ntasks - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (tasks[mid].stile0 <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// forward: X[row(r)] -= sum_c L[r][col0+c] * x1[c]   for SOLVE_ROWS rows r below the block
__global__ __launch_bounds__(SOLVE_ROWS) void k_solve_fwd_update(
    const PotrfTask *__restrict__ tasks, int ntasks, const double *__restrict__ Lx,
    const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    __shared__ double ys[NB][SOLVE_NR];
    const int ti = find_solve_task(tasks, ntasks, blockIdx.x);
    const PotrfTask t = tasks[ti];
    const int tid = threadIdx.x, nb = t.nb;
    const int r = t.col0 + nb + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS + tid;   // row inside s
    const bool ok = r < t.lda;
    const double *__restrict__ Lr = Lx + t.a_off + (r - t.col0);                   // L[r][col0]
    int64_t row = 0;
    if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
    const double *__restrict__ x1 = X + t.k1 + t.col0;
    for (int j0 = 0; j0 < nrhs; j0 += SOLVE_NR) {
        const int nj = min(SOLVE_NR, nrhs - j0);
        __syncthreads();
        for (int idx = tid; idx < nb * SOLVE_NR; idx += SOLVE_ROWS) {
            const int c = idx % nb, j = idx / nb;
            ys[c][j] = (j < nj) ? x1[c + (int64_t)(j0 + j) * ldx] : 0.0;
        }
        __syncthreads();
        if (ok) {
            double acc[SOLVE_NR];
#pragma unroll
            for (int j = 0; j < SOLVE_NR; j++) acc[j] = 0.0;
            for (int c = 0; c < nb; c++) {
                const double l = Lr[(int64_t)c * t.lda];
#pragma unroll
                for (int j = 0; j < SOLVE_NR; j++) acc[j] += l * ys[c][j];
            }
#pragma unroll
            for (int j = 0; j < SOLVE_NR; j++)
                if (j < nj) atomic_add_f64(X + row + (int64_t)(j0 + j) * ldx, -acc[j]);
        }
    }
}

// backward: x1[c] -= sum_r L[r][col0+c] * X[row(r)]   over SOLVE_ROWS rows r below the block
__global__ __launch_bounds__(SOLVE_ROWS) void k_solve_bwd_update(
    const PotrfTask *__restrict__ tasks, int ntasks, const double *__restrict__ Lx,
    const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    __shared__ double T[64][NB + 1];            // 64 rows x nb columns of L
    __shared__ double xs[64][SOLVE_NR];
    __shared__ double red[4][NB][SOLVE_NR];
    const int ti = find_solve_task(tasks, ntasks, blockIdx.x);
    const PotrfTask t = tasks[ti];
    const int tid = threadIdx.x, nb = t.nb;
    const int rbase = t.col0 + nb + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS;
    const int c = tid & 63, part = tid >> 6;
    double *__restrict__ x1 = X + t.k1 + t.col0;
    for (int j0 = 0; j0 < nrhs; j0 += SOLVE_NR) {
        const int nj = min(SOLVE_NR, nrhs - j0);
        double acc[SOLVE_NR];
#pragma unroll
        for (int j = 0; j < SOLVE_NR; j++) acc[j] = 0.0;
        for (int sub = 0; sub < SOLVE_ROWS / 64; sub++) {
            const int r0 = rbase + sub * 64;
            if (r0 >= t.lda) break;                                 // uniform
            __syncthreads();
            {   // stage 64 rows x nb cols of L (lanes along rows: coalesced) and the X rows
                const int rl = tid & 63, r = r0 + rl;
                const bool ok = r < t.lda;
                const double *__restrict__ Lr = Lx + t.a_off + (r - t.col0);
                for (int cc = (tid >> 6); cc < nb; cc += 4)
                    T[rl][cc] = ok ? Lr[(int64_t)cc * t.lda] : 0.0;
                if (tid < 64) {
                    int64_t row = 0;
                    if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
#pragma unroll
                    for (int j = 0; j < SOLVE_NR; j++)
                        xs[rl][j] = (ok && j < nj) ? X[row + (int64_t)(j0 + j) * ldx] : 0.0;
                }
            }
            __syncthreads();
            if (c < nb) {
                for (int rr = part * 16; rr < part * 16 + 16; rr++) {
                    const double l = T[rr][c];
#pragma unroll
                    for (int j = 0; j < SOLVE_NR; j++) acc[j] += l * xs[rr][j];
                }
            }
        }
        __syncthreads();
        if (c < nb) {
#pragma unroll
            for (int j = 0; j < SOLVE_NR; j++) red[part][c][j] = acc[j];
        }
        __syncthreads();
        if (part == 0 && c < nb) {
#pragma unroll
            for (int j = 0; j < SOLVE_NR; j++) {
                if (j < nj) {
                    const double v = red[0][c][j] + red[1][c][j] + red[2][c][j] + red[3][c][j];
                    atomic_add_f64(x1 + c + (int64_t)(j0 + j) * ldx, -v);
                }
            }
        }
    }
}

}  // namespace mi355
