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
// One workgroup per block.  Dinv is read straight from global memory (column-major, ld nb): for a
// fixed k the 64 lanes of a wave read 64 consecutive doubles; the k range is split over the 4 waves
// and reduced through LDS, so no lane runs a 64-long dependent chain.
template <bool TRANS>
__global__ __launch_bounds__(256) void k_solve_diag(
    const PotrfTask *__restrict__ tasks, const double *__restrict__ Dinv,
    double *__restrict__ X, int64_t ldx, int nrhs)
{
    __shared__ double xs[NB][SOLVE_NR];
    __shared__ double red[4][NB][SOLVE_NR];
    const PotrfTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, nb = t.nb;
    const int i = tid & 63, part = tid >> 6;
    const double *__restrict__ D = Dinv + t.dinv_off;
    double *__restrict__ x1 = X + t.k1 + t.col0;
    for (int j0 = 0; j0 < nrhs; j0 += SOLVE_NR) {
        const int nj = min(SOLVE_NR, nrhs - j0);
        __syncthreads();
        for (int idx = tid; idx < nb * SOLVE_NR; idx += 256) {
            const int r = idx % nb, j = idx / nb;
            xs[r][j] = (j < nj) ? x1[r + (int64_t)(j0 + j) * ldx] : 0.0;
        }
        __syncthreads();
        double acc[SOLVE_NR];
#pragma unroll
        for (int j = 0; j < SOLVE_NR; j++) acc[j] = 0.0;
        if (i < nb) {
            // y_i = sum_k D(i,k) x_k (k <= i)   or   sum_k D(k,i) x_k (k >= i)
            const int klo = TRANS ? i : 0, khi = TRANS ? nb : i + 1;
            for (int k = klo + part; k < khi; k += 4) {
                const double d = TRANS ? D[k + i * nb] : D[i + k * nb];
#pragma unroll
                for (int j = 0; j < SOLVE_NR; j++) acc[j] += d * xs[k][j];
            }
        }
#pragma unroll
        for (int j = 0; j < SOLVE_NR; j++) red[part][i][j] = acc[j];
        __syncthreads();
        if (part == 0 && i < nb) {
#pragma unroll
            for (int j = 0; j < SOLVE_NR; j++)
                if (j < nj) x1[i + (int64_t)(j0 + j) * ldx] = red[0][i][j] + red[1][i][j] + red[2][i][j] + red[3][i][j];
        }
    }
}

// forward: X[row(r)] -= sum_c L[r][col0+c] * x1[c]  for the SOLVE_ROWS (64) rows r of one tile.
// 256 threads: lane = row, wave w owns columns [16w, 16w+16): 16 independent coalesced loads per
// lane in flight, partial sums reduced over the 4 waves through LDS, one atomic per row and rhs.
__global__ __launch_bounds__(256) void k_solve_fwd_update(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ tile_task, const double *__restrict__ Lx,
    const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    __shared__ double ys[NB][SOLVE_NR];
    __shared__ double red[4][SOLVE_ROWS][SOLVE_NR];
    const PotrfTask t = tasks[tile_task[blockIdx.x]];
    const int tid = threadIdx.x, nb = t.nb;
    const int rl = tid & 63, w = tid >> 6;
    const int r = t.col0 + nb + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS + rl;     // row inside s
    const bool ok = r < t.lda;
    const double *__restrict__ Lr = Lx + t.a_off + ((ok ? r : t.col0) - t.col0);     // L[r][col0]
    const double *__restrict__ x1 = X + t.k1 + t.col0;
    double l[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int c = w * 16 + q;
        l[q] = (ok && c < nb) ? Lr[(int64_t)(c < nb ? c : 0) * t.lda] : 0.0;
    }
    for (int j0 = 0; j0 < nrhs; j0 += SOLVE_NR) {
        const int nj = min(SOLVE_NR, nrhs - j0);
        __syncthreads();
        for (int idx = tid; idx < nb * SOLVE_NR; idx += 256) {
            const int c = idx % nb, j = idx / nb;
            ys[c][j] = (j < nj) ? x1[c + (int64_t)(j0 + j) * ldx] : 0.0;
        }
        __syncthreads();
        double acc[SOLVE_NR];
#pragma unroll
        for (int j = 0; j < SOLVE_NR; j++) acc[j] = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = w * 16 + q;
            if (c < nb) {
#pragma unroll
                for (int j = 0; j < SOLVE_NR; j++) acc[j] += l[q] * ys[c][j];
            }
        }
#pragma unroll
        for (int j = 0; j < SOLVE_NR; j++) red[w][rl][j] = acc[j];
        __syncthreads();
        if (w == 0 && ok) {
            const int64_t row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
#pragma unroll
            for (int j = 0; j < SOLVE_NR; j++)
                if (j < nj)
                    atomic_add_f64(X + row + (int64_t)(j0 + j) * ldx,
                                   -(red[0][rl][j] + red[1][rl][j] + red[2][rl][j] + red[3][rl][j]));
        }
    }
}

// Sums, over the 64 lanes of the wave, 8 values per lane at once ("transpose-reduce"): three
// halving exchanges (lane bits 5,4,3) leave one value per lane -- the partial sum of column
// b5*4 + b4*2 + b3 over 8 lanes -- and three plain exchanges (bits 2,1,0) finish it: 10 shuffles
// instead of 8 x 6.  On return lane q (q < 8) holds the complete sum of column q.
__device__ __forceinline__ double wave_colsum8(double (&v)[8], int lane)
{
#pragma unroll
    for (int h = 4, bit = 32; h >= 1; h >>= 1, bit >>= 1) {
        const bool upper = (lane & bit) != 0;
#pragma unroll
        for (int q = 0; q < h; q++) {
            const double keep = upper ? v[q + h] : v[q];
            const double send = upper ? v[q] : v[q + h];
            v[q] = keep + __shfl_xor(send, bit, 64);
        }
    }
    double x = v[0];
    x += __shfl_xor(x, 4, 64);
    x += __shfl_xor(x, 2, 64);
    x += __shfl_xor(x, 1, 64);
    // lane q (< 8) fetches column q from the lane whose bits 5,4,3 spell q
    const int src = ((lane >> 2) & 1) * 32 + ((lane >> 1) & 1) * 16 + (lane & 1) * 8;
    return __shfl(x, src, 64);
}

// backward: x1[c] -= sum_r L[r][col0+c] * X[row(r)]  over the 64 rows r of one tile.
// lane = row, wave w owns columns [16w, 16w+16) (a wave without columns leaves at once: most
// supernodes are narrower than 16); the 64-lane sums are formed with two transpose-reduces per
// rhs, lanes 0..7 issue the atomics.
// NRA > 0 (nrhs <= NRA <= SOLVE_NR): one workgroup walks `group` consecutive entries of the tile table and keeps
// its partial sums in registers while the entries belong to the same 64-column block, so a tall block
// (tens of thousands of rows under 64 columns, hundreds of tiles that all add into the same 64 x nrhs
// entries of X) sends one set of atomics per workgroup instead of one per tile.
template <int NRA>
__global__ __launch_bounds__(256) void k_solve_bwd_update(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ tile_task, int64_t ntiles, int group,
    const double *__restrict__ Lx, const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    if constexpr (NRA == 0) {
        const PotrfTask t = tasks[tile_task[blockIdx.x]];
        const int nb = t.nb;
        if (w * 16 >= nb) return;                       // no barrier in this kernel
        const int r = t.col0 + nb + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS + lane;
        const bool ok = r < t.lda;
        const double *__restrict__ Lr = Lx + t.a_off + ((ok ? r : t.col0) - t.col0);
        double *__restrict__ x1 = X + t.k1 + t.col0;
        double l[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = w * 16 + q;
            l[q] = (ok && c < nb) ? Lr[(int64_t)(c < nb ? c : 0) * t.lda] : 0.0;
        }
        int64_t row = 0;
        if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
        for (int j = 0; j < nrhs; j++) {
            const double xr = ok ? X[row + (int64_t)j * ldx] : 0.0;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                if (w * 16 + half * 8 >= nb) break;     // wave-uniform
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = l[half * 8 + q] * xr;
                const double s = wave_colsum8(v, lane);
                const int c = w * 16 + half * 8 + lane;
                if (lane < 8 && c < nb) atomic_add_f64(x1 + c + (int64_t)j * ldx, -s);
            }
        }
    } else {
        const int64_t e0 = (int64_t)blockIdx.x * group, e1 = min(e0 + (int64_t)group, ntiles);
        double acc[NRA][16];
#pragma unroll
        for (int j = 0; j < NRA; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[j][q] = 0.0;
        int cur = -1;
        PotrfTask t = tasks[tile_task[e0]];
        auto flush = [&]() {
            const int nb = t.nb;
            if (w * 16 >= nb) return;
            double *__restrict__ x1 = X + t.k1 + t.col0;
#pragma unroll
            for (int j = 0; j < NRA; j++) {
                if (j >= nrhs) break;
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    if (w * 16 + half * 8 >= nb) break;
                    double v[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) { v[q] = acc[j][half * 8 + q]; acc[j][half * 8 + q] = 0.0; }
                    const double s = wave_colsum8(v, lane);
                    const int c = w * 16 + half * 8 + lane;
                    if (lane < 8 && c < nb) atomic_add_f64(x1 + c + (int64_t)j * ldx, -s);
                }
            }
        };
        for (int64_t e = e0; e < e1; e++) {
            const int ti = tile_task[e];
            if (ti != cur) {
                if (cur >= 0) flush();
                cur = ti;
                t = tasks[ti];
            }
            const int nb = t.nb;
            if (w * 16 >= nb) continue;                 // wave-uniform
            const int r = t.col0 + nb + (int)(e - t.stile0) * SOLVE_ROWS + lane;
            const bool ok = r < t.lda;
            const double *__restrict__ Lr = Lx + t.a_off + ((ok ? r : t.col0) - t.col0);
            int64_t row = 0;
            if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
            double xr[NRA];
#pragma unroll
            for (int j = 0; j < NRA; j++) xr[j] = (ok && j < nrhs) ? X[row + (int64_t)j * ldx] : 0.0;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int c = w * 16 + q;
                const double l = (ok && c < nb) ? Lr[(int64_t)(c < nb ? c : 0) * t.lda] : 0.0;
#pragma unroll
                for (int j = 0; j < NRA; j++) acc[j][q] += l * xr[j];
            }
        }
        if (cur >= 0) flush();
    }
}

}  // namespace mi355
