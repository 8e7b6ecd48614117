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

// occupancy target of a kernel (waves per SIMD): the register allocator may use 512 / n VGPRs
#ifndef MI355_WAVES_PER_EU          // (the CPU stand-in of the HIP header, tests only, defines it away)
#define MI355_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif
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
    const int r = t.urow0 + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS + rl;         // row inside s
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
            // rows inside the supernode's own column range are written by this block alone in this launch
            // (any other writer would be a descendant, i.e. on a lower level): plain read-modify-write;
            // rows of ancestors can be hit by several supernodes of the level at once: atomics
            const bool own = r < t.nscol;
            const int64_t row = own ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
#pragma unroll
            for (int j = 0; j < SOLVE_NR; j++)
                if (j < nj) {
                    const double v = red[0][rl][j] + red[1][rl][j] + red[2][rl][j] + red[3][rl][j];
                    double *xp = X + row + (int64_t)(j0 + j) * ldx;
                    if (own) *xp -= v; else atomic_add_f64(xp, -v);
                }
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
        const int r = t.urow0 + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS + lane;
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
            const int r = t.urow0 + (int)(e - t.stile0) * SOLVE_ROWS + lane;
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

// backward update over a work list: entry = a run of consecutive row tiles of ONE block task (the host cuts
// the tall blocks into runs, schedule.cpp), summed in registers and flushed once -- one transpose-reduce and
// one set of atomics per run.  Two tiles are in flight per pass (32 loads of L per lane), so a run costs
// about half a memory latency per tile instead of a whole one.
constexpr int BR_U = 2;     // row tiles in flight per pass of k_solve_bwd_range
template <int NRA>
__global__ __launch_bounds__(256) void k_solve_bwd_range(
    const PotrfTask *__restrict__ tasks, const SolveRange *__restrict__ work,
    const double *__restrict__ Lx, const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const SolveRange e = work[blockIdx.x];
    const PotrfTask t = tasks[e.task];
    const int nb = t.nb;
    if (w * 16 >= nb) return;                           // no barrier in this kernel
    double acc[NRA][16];
#pragma unroll
    for (int j = 0; j < NRA; j++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[j][q] = 0.0;
    for (int tt = 0; tt < e.ntiles; tt += BR_U) {
        double l[BR_U][16], xr[BR_U][NRA];
#pragma unroll
        for (int u = 0; u < BR_U; u++) {
            const int r = t.urow0 + (e.tile0 + tt + u) * SOLVE_ROWS + lane;
            const bool ok = (tt + u < e.ntiles) && r < t.lda;
            const double *__restrict__ Lr = Lx + t.a_off + ((ok ? r : t.col0) - t.col0);
            int64_t row = 0;
            if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
#pragma unroll
            for (int j = 0; j < NRA; j++) xr[u][j] = (ok && j < nrhs) ? X[row + (int64_t)j * ldx] : 0.0;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int c = w * 16 + q;
                l[u][q] = (ok && c < nb) ? Lr[(int64_t)(c < nb ? c : 0) * t.lda] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < BR_U; u++)
#pragma unroll
            for (int q = 0; q < 16; q++)
#pragma unroll
                for (int j = 0; j < NRA; j++) acc[j][q] += l[u][q] * xr[u][j];
    }
    double *__restrict__ x1 = X + t.k1 + t.col0;
#pragma unroll
    for (int j = 0; j < NRA; j++) {
        if (j >= nrhs) break;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (w * 16 + half * 8 >= nb) break;
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = acc[j][half * 8 + q];
            const double s = wave_colsum8(v, lane);
            const int c = w * 16 + half * 8 + lane;
            if (lane < 8 && c < nb) atomic_add_f64(x1 + c + (int64_t)j * ldx, -s);
        }
    }
}

//------------------------------------------------------------------------------
// k_solve_group: the part of a sweep that lies inside one solve group (SOLVE_GROUP consecutive 64-column
// blocks of a supernode, schedule.hpp): the diagonal solves x_c <- inv(L_cc) x_c and the updates between the
// group's blocks, one workgroup per group, x of the group in LDS.  The work is a chain of small mat-vecs,
// step s = c(c+1)/2 + b  <->  block (c, b), b <= c < nblk:
//   forward  (s ascending) : (c,b), b < c:  x_c -= L(c,b)   x_b ;   (c,c):  x_c <- inv(L_cc)   x_c
//   backward (s descending): (c,c):  x_c <- inv(L_cc)^T x_c ;   (c,b), b < c:  x_b -= L(c,b)^T x_c
// each at most 64 x 64, 16 entries per thread, lanes along the rows of the block; forward the column range is
// split over the 4 waves and reduced through LDS, backward every wave owns 16 columns and the sums over the
// rows go through LDS as 16 partial sums per column.  A step computes in a fraction of a memory latency, so ALL entries of ALL
// steps are loaded into registers before the first step (MAXBLK(MAXBLK+1)/2 x 16 doubles per thread): the
// chain pays one memory latency, not one per step (18 -> 9 us per group of four).  MAXBLK = most blocks of
// any group in the launch (1, 2 or SOLVE_GROUP): the wide leaf levels keep a small register footprint.
//------------------------------------------------------------------------------
__host__ __device__ constexpr int sg_c(int s) { return s >= 6 ? 3 : s >= 3 ? 2 : s >= 1 ? 1 : 0; }
__host__ __device__ constexpr int sg_b(int s) { return s - sg_c(s) * (sg_c(s) + 1) / 2; }
static_assert(SOLVE_GROUP == 4, "sg_c/sg_b enumerate the blocks of a group of four");

template <bool TRANS, int NR, int MAXBLK>
__global__ __launch_bounds__(256) void k_solve_group(
    const SolveGroupTask *__restrict__ groups, const double *__restrict__ Dinv, const double *__restrict__ Lx,
    double *__restrict__ X, int64_t ldx, int nrhs)
{
    constexpr int NS = MAXBLK * (MAXBLK + 1) / 2;
    __shared__ double xs[MAXBLK * NB][NR];
    __shared__ double red[TRANS ? 1 : 4][NB][NR];
    __shared__ double redT[TRANS ? NB : 1][17][NR];
    const SolveGroupTask t = groups[blockIdx.x];
    const int tid = threadIdx.x, i = tid & 63, part = tid >> 6;
    const int ncols = t.ncols, nblk = (ncols + NB - 1) / NB;
    double *__restrict__ x0 = X + t.x0;
    // entries of block (c, b) for this thread; everything outside the block (and the strict upper part of an
    // inverse) reads as zero
    // forward: entries (row i, columns part + 4q) of block (c, b); backward: wave `part` owns columns
    // [16 part, 16 part + 16) and holds entries (row 16 rc + lane%16, column 16 part + 4 cc + lane/16) at
    // [4 cc + rc] -- either way the lanes run along the rows (coalesced; column-strided lanes took 64 cache
    // lines per load instruction and 17 us per group)
    const int r16 = i & 15, c4 = i >> 4;
    auto load = [&](int b, int c, double (&m)[16]) {
        const int nbb = min(NB, ncols - NB * b), nbc = min(NB, ncols - NB * c);
        const double *__restrict__ D = Dinv + t.dinv_off[b];
        const double *__restrict__ Lb = Lx + t.a_off + (int64_t)NB * c + (int64_t)NB * b * t.lda;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = TRANS ? 16 * (q & 3) + r16 : i;
            const int col = TRANS ? 16 * part + 4 * (q >> 2) + c4 : part + 4 * q;
            if (c == b) m[q] = (row < nbb && col <= row) ? D[row + col * nbb] : 0.0;
            else        m[q] = (row < nbc && col < nbb) ? Lb[row + (int64_t)col * t.lda] : 0.0;
        }
    };
    double m[NS][16];
#pragma unroll
    for (int u = 0; u < NS; u++) {
        const int s = TRANS ? NS - 1 - u : u;                // in the order of use
        if (sg_c(s) < nblk) load(sg_b(s), sg_c(s), m[s]);
    }
    for (int idx = tid; idx < MAXBLK * NB * NR; idx += 256) {
        const int r = idx % (MAXBLK * NB), j = idx / (MAXBLK * NB);
        xs[r][j] = (r < ncols && j < nrhs) ? x0[r + (int64_t)j * ldx] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NS; u++) {
        const int s = TRANS ? NS - 1 - u : u;
        const int b = sg_b(s), c = sg_c(s);
        if (c >= nblk) continue;                              // uniform over the workgroup
        if constexpr (TRANS) {
            // y_col = sum_row M[row][col] x_c[row]: 16 partial sums per column (one per lane%16) through LDS.
            // Target block: b (== c for the diagonal solve, which reads x_c before the first barrier and
            // writes it after)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                double acc[NR];
#pragma unroll
                for (int j = 0; j < NR; j++) acc[j] = 0.0;
#pragma unroll
                for (int rc = 0; rc < 4; rc++) {
#pragma unroll
                    for (int j = 0; j < NR; j++) acc[j] += m[s][4 * cc + rc] * xs[NB * c + 16 * rc + r16][j];
                }
#pragma unroll
                for (int j = 0; j < NR; j++) redT[16 * part + 4 * cc + c4][r16][j] = acc[j];
            }
            __syncthreads();
            if (tid < NB) {
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    double v = 0.0;
#pragma unroll
                    for (int p = 0; p < 16; p++) v += redT[tid][p][j];
                    if (c == b) xs[NB * b + tid][j] = v; else xs[NB * b + tid][j] -= v;
                }
            }
            __syncthreads();
        } else {
            const int src = b, dst = c;
            double acc[NR];
#pragma unroll
            for (int j = 0; j < NR; j++) acc[j] = 0.0;
#pragma unroll
            for (int q = 0; q < 16; q++) {
#pragma unroll
                for (int j = 0; j < NR; j++) acc[j] += m[s][q] * xs[NB * src + part + 4 * q][j];
            }
#pragma unroll
            for (int j = 0; j < NR; j++) red[part][i][j] = acc[j];
            __syncthreads();
            if (part == 0) {
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const double v = red[0][i][j] + red[1][i][j] + red[2][i][j] + red[3][i][j];
                    if (c == b) xs[NB * dst + i][j] = v; else xs[NB * dst + i][j] -= v;
                }
            }
            __syncthreads();
        }
    }
    for (int idx = tid; idx < ncols * NR; idx += 256) {
        const int r = idx % ncols, j = idx / ncols;
        if (j < nrhs) x0[r + (int64_t)j * ldx] = xs[r][j];
    }
}

//------------------------------------------------------------------------------
// Multi-right-hand-side sweeps on the FP64 MFMA pipe (real L, nrhs >= Tuning::solve_mma_min).
// With many right-hand sides the off-diagonal parts are GEMMs (64 columns of L against 64 x nrhs of X):
// 16 flop per byte of L at nrhs = 64, i.e. past the HBM roofline and onto the matrix pipe.
// These kernels work on a ROW-MAJOR image Y of the right-hand sides (entry (row, j) at Y[row*ldr + j]): the
// 64 x nrhs block of X that a tile touches is then 64 runs of nrhs consecutive doubles instead of
// 64 x nrhs doubles spread over nrhs columns that lie n*8 bytes apart (27 MB at 150^3: one TLB entry and
// one DRAM page per right-hand side and tile).  The engine transposes on the way in and out.
// Each kernel walks `group` consecutive entries of the tile table (64 rows of one 64-column block each)
// and handles up to 64 right-hand sides [j0, j0+64) per launch; the right-hand sides run along lane%16 of
// the MFMA result, so stores and atomics touch runs of 16 consecutive doubles.
//------------------------------------------------------------------------------
constexpr int SM_LD = 66;            // LDS row stride in doubles: 16 rows x 4 k land on disjoint banks

// Complex L (E = 2): Y is still handled as a REAL row-major matrix -- the interleaved (re, im) pairs of the
// nrhs complex right-hand sides are 2*nrhs real columns -- and every complex product becomes two real ones:
//   L x     : columns (re, im) receive  Lr*(xr, xi) + Li*(-xi, xr)
//   L^H x   : columns (re, im) receive  Lr^T*(xr, xi) + Li^T*(xi, -xr)
// i.e. the same MFMA loops run twice, once on the real part of L against the image of X as it is, once on
// the imaginary part against the image with the columns of each pair swapped and one sign flipped.

// the partner column of real column j in its (re, im) pair, with the sign of the second product
// CONJ = false (L x): (re, im) <- (-xi, +xr);  CONJ = true (L^H x): (re, im) <- (+xi, -xr)
template <bool CONJ>
__device__ __forceinline__ double pair_image(const double *__restrict__ rowp, int j)
{
    const bool re_slot = (j & 1) == 0;
    const double v = rowp[j ^ 1];
    return (re_slot != CONJ) ? -v : v;
}

// Y (row-major, ld ldr) <- X (column-major, ld ldx) or back; 64 x 64 tiles through LDS, both sides coalesced.
// T = double (real) or double2 (complex entries move as units).
template <typename T>
__global__ __launch_bounds__(256) void k_solve_transpose(int64_t n, int nrhs, T *__restrict__ Xc, int64_t ldx,
                                                         T *__restrict__ Yr, int64_t ldr, int to_rowmajor)
{
    __shared__ T tile[64][65];
    const int tid = threadIdx.x, a = tid & 63, b4 = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    for (int j0 = 0; j0 < nrhs; j0 += 64) {
        __syncthreads();
        for (int b = b4; b < 64; b += 4) {
            if (to_rowmajor) { if (r0 + a < n && j0 + b < nrhs) tile[b][a] = Xc[r0 + a + (int64_t)(j0 + b) * ldx]; }   // tile[j][r]
            else             { if (r0 + b < n && j0 + a < nrhs) tile[a][b] = Yr[(r0 + b) * ldr + j0 + a]; }            // tile[j][r]
        }
        __syncthreads();
        for (int b = b4; b < 64; b += 4) {
            if (to_rowmajor) { if (r0 + b < n && j0 + a < nrhs) Yr[(r0 + b) * ldr + j0 + a] = tile[a][b]; }
            else             { if (r0 + a < n && j0 + b < nrhs) Xc[r0 + a + (int64_t)(j0 + b) * ldx] = tile[b][a]; }
        }
    }
}

// forward: Y[row(r)][j] -= sum_c L[r][col0+c] * x1[c][j].  Contraction over the 64 columns: L is read straight
// from global memory in MFMA operand layout (16 consecutive rows per k: 128-byte runs, every entry of the
// tile read once by exactly one wave); x1 (shared by all tiles of the block) is staged in LDS once per block.
// ldr, nrhs, j0 count REAL columns of Y (2 per complex right-hand side).
template <int E>
__global__ __launch_bounds__(256) void k_solve_fwd_mma(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ tile_task, int64_t ntiles, int group,
    const double *__restrict__ Lx, const int32_t *__restrict__ Ls, double *__restrict__ Y, int64_t ldr,
    int nrhs, int j0)
{
    __shared__ double Xs[64][SM_LD];                     // Xs[j][c] = x1[c][j0 + j]
    __shared__ double Xs2[E == 2 ? 64 : 1][SM_LD];       // complex: the pair-swapped image
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nj = min(64, nrhs - j0);
    const int64_t e0 = (int64_t)blockIdx.x * group, e1 = min(e0 + (int64_t)group, ntiles);
    int cur = -1;
    PotrfTask t = tasks[tile_task[e0]];
    double al[16], ai[E == 2 ? 16 : 1];
    // the operand of tile e of block tt: this lane's row, 16 k-steps of 4 columns
    auto load_tile = [&](const PotrfTask &tt, int64_t e) {
        const int nq = (tt.nb + 3) >> 2;                 // k-steps of 4 columns that hold anything
        const int r = tt.urow0 + (int)(e - tt.stile0) * SOLVE_ROWS + w * 16 + l15;
        const int64_t roff = tt.a_off + ((r < tt.lda ? r : tt.col0) - tt.col0);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = 4 * q + l4;
            const int64_t off = roff + (int64_t)(c < tt.nb ? c : 0) * tt.lda;   // c >= nb: finite value against a zero of Xs
            if constexpr (E == 2) {
                const double2 v = (q < nq) ? reinterpret_cast<const double2 *>(Lx)[off] : make_double2(0.0, 0.0);
                al[q] = v.x; ai[q] = v.y;
            } else {
                al[q] = (q < nq) ? Lx[off] : 0.0;
            }
        }
    };
    // complex blocks (twice the bytes and twice the products per tile) request the next tile's operand before the
    // results of the current one go out: 42.1 -> 39.5 ms at helm 100^3, nrhs = 64; real blocks measured 3 % slower
    // that way (53.6 -> 55.2 ms at 150^3) and load where they use
    constexpr bool AHEAD = (E == 2);
    if constexpr (AHEAD) load_tile(t, e0);
    for (int64_t e = e0; e < e1; e++) {
        const int ti = tile_task[e];
        if (ti != cur) {
            cur = ti;
            t = tasks[ti];
            __syncthreads();                             // readers of the previous block's x1 are done
            const double *__restrict__ x1 = Y + (int64_t)(t.k1 + t.col0) * ldr + j0;
            const int nbp = (t.nb + 3) & ~3;             // k padded to whole MFMA steps (zeros)
            for (int idx = tid; idx < 64 * nbp; idx += 256) {
                const int j = idx & 63, c = idx >> 6;    // consecutive threads: consecutive right-hand sides
                const bool in = c < t.nb && j < nj;
                Xs[j][c] = in ? x1[(int64_t)c * ldr + j] : 0.0;
                if constexpr (E == 2) Xs2[j][c] = in ? pair_image<false>(x1 + (int64_t)c * ldr, j) : 0.0;
            }
            __syncthreads();
        }
        const int nq = (t.nb + 3) >> 2;
        const int rbase = t.urow0 + (int)(e - t.stile0) * SOLVE_ROWS + w * 16;       // first row of this wave
        if constexpr (!AHEAD) load_tile(t, e);
        v4d acc[4];
#pragma unroll
        for (int jb = 0; jb < 4; jb++) acc[jb] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (q >= nq) break;                                      // uniform
#pragma unroll
            for (int jb = 0; jb < 4; jb++) {
                if (16 * jb >= nj) break;                            // uniform
                acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(al[q], Xs[16 * jb + l15][4 * q + l4], acc[jb], 0, 0, 0);
                if constexpr (E == 2)
                    acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[q], Xs2[16 * jb + l15][4 * q + l4], acc[jb], 0, 0, 0);
            }
        }
        if constexpr (AHEAD) {
            if (e + 1 < e1) load_tile(tasks[tile_task[e + 1]], e + 1);       // its latency runs under the atomics
        }
        // result: rows rbase + lane/16 + 4g (g = 0..3), right-hand side 16 jb + lane%16
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int rr = rbase + l4 + 4 * g;
            if (rr >= t.lda) continue;
            // rows inside the supernode's own column range are written by this block alone in this launch
            // (any other writer would be a descendant, i.e. on a lower level): plain read-modify-write;
            // rows of ancestors can be hit by several supernodes of the level at once: atomics
            const bool own = rr < t.nscol;
            const int64_t row = own ? (int64_t)t.k1 + rr : (int64_t)Ls[t.ls_off + rr];
            double *__restrict__ yp = Y + row * ldr + j0 + l15;
#pragma unroll
            for (int jb = 0; jb < 4; jb++) {
                if (16 * jb + l15 < nj) {
                    if (own) yp[16 * jb] -= acc[jb][g]; else atomic_add_f64(yp + 16 * jb, -acc[jb][g]);
                }
            }
        }
    }
}

// backward: x1[c][j] -= sum_r conj(L[r][col0+c]) * Y[row(r)][j].  Contraction over the rows: the L tile and the
// gathered rows of Y are staged in LDS and read back in operand layout; the products of all the tiles a
// workgroup walks accumulate in the MFMA registers while the block stays the same, so a tall block sends
// one set of atomics per workgroup.  Complex: a second pass through the same LDS buffers with Im(L) and the
// pair-swapped image of Y.
template <int E>
__global__ __launch_bounds__(256) void k_solve_bwd_mma(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ tile_task, int64_t ntiles, int group,
    const double *__restrict__ Lx, const int32_t *__restrict__ Ls, double *__restrict__ Y, int64_t ldr,
    int nrhs, int j0)
{
    __shared__ double Lt[64][SM_LD];                     // Lt[c][r]
    __shared__ double Xt[64][SM_LD];                     // Xt[j][r]
    __shared__ int64_t rowof[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nj = min(64, nrhs - j0);
    const int64_t e0 = (int64_t)blockIdx.x * group, e1 = min(e0 + (int64_t)group, ntiles);
    v4d acc[4];                                          // wave w: columns c = 16w + lane/16 + 4g, rhs 16 jb + lane%16
#pragma unroll
    for (int jb = 0; jb < 4; jb++) acc[jb] = (v4d){0.0, 0.0, 0.0, 0.0};
    int cur = -1;
    PotrfTask t = tasks[tile_task[e0]];
    auto flush = [&]() {
        double *__restrict__ x1 = Y + (int64_t)(t.k1 + t.col0) * ldr + j0 + l15;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int c = 16 * w + l4 + 4 * g;
#pragma unroll
            for (int jb = 0; jb < 4; jb++) {
                if (c < t.nb && 16 * jb + l15 < nj) atomic_add_f64(x1 + (int64_t)c * ldr + 16 * jb, -acc[jb][g]);
                acc[jb][g] = 0.0;
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
        const int r0 = t.urow0 + (int)(e - t.stile0) * SOLVE_ROWS;
        const int rows_here = min(SOLVE_ROWS, t.lda - r0);
        const int nq = (rows_here + 3) >> 2;             // k-steps of 4 rows that hold anything (the rest is zero)
        const int r = r0 + lane;
        const bool ok = lane < rows_here;
        __syncthreads();                                 // the previous tile's operand reads are done
        if (w == 0) rowof[lane] = !ok ? 0 : (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
        const int64_t roff = t.a_off + ((ok ? r : t.col0) - t.col0);
        double li[E == 2 ? 16 : 1];
#pragma unroll 4
        for (int q = 0; q < 16; q++) {
            const int c = w * 16 + q;
            if constexpr (E == 2) {
                const double2 v = (ok && c < nb) ? reinterpret_cast<const double2 *>(Lx)[roff + (int64_t)c * t.lda]
                                                 : make_double2(0.0, 0.0);
                Lt[c][lane] = v.x; li[q] = v.y;
            } else {
                Lt[c][lane] = (ok && c < nb) ? Lx[roff + (int64_t)c * t.lda] : 0.0;
            }
        }
        __syncthreads();                                 // rowof visible
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int j = idx & 63, rr = idx >> 6;       // consecutive threads: consecutive right-hand sides
            Xt[j][rr] = (rr < rows_here && j < nj) ? Y[rowof[rr] * ldr + j0 + j] : 0.0;
        }
        __syncthreads();
        auto products = [&]() {
            if (16 * w < nb) {                           // wave-uniform; no barrier inside
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    if (q >= nq) break;
                    const double a = Lt[16 * w + l15][4 * q + l4];
#pragma unroll
                    for (int jb = 0; jb < 4; jb++) {
                        if (16 * jb >= nj) break;
                        acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xt[16 * jb + l15][4 * q + l4], acc[jb], 0, 0, 0);
                    }
                }
            }
        };
        products();
        if constexpr (E == 2) {
            __syncthreads();                             // pass 0 has read Lt and Xt
#pragma unroll 4
            for (int q = 0; q < 16; q++) Lt[w * 16 + q][lane] = li[q];
            for (int idx = tid; idx < 64 * 64; idx += 256) {
                const int j = idx & 63, rr = idx >> 6;
                Xt[j][rr] = (rr < rows_here && j < nj) ? pair_image<true>(Y + rowof[rr] * ldr + j0, j) : 0.0;
            }
            __syncthreads();
            products();
        }
    }
    if (cur >= 0) flush();
}

// k_solve_bwd_mma over a work list (grouped sweeps): entry = a run of consecutive row tiles of ONE block task, so
// the loop has no task switches and is software-pipelined -- while the MFMAs of tile t run, the L entries and
// the gathered rows of Y of tile t+1 are already in flight into registers; they go to LDS between two barriers
// at the top of the next pass.  Complex: Im(L) stays in registers and the second product reads the
// pair-swapped image of Y from the same LDS rows.  One set of atomics per run.
template <int E>
__global__ __launch_bounds__(256) MI355_WAVES_PER_EU(E == 1 ? 3 : 2) void k_solve_bwd_mma_run(
    const PotrfTask *__restrict__ tasks, const SolveRange *__restrict__ work,
    const double *__restrict__ Lx, const int32_t *__restrict__ Ls, double *__restrict__ Y, int64_t ldr,
    int nrhs, int j0)
{
    __shared__ double Xt[64][SM_LD];                     // Xt[j][r]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nj = min(64, nrhs - j0);
    const SolveRange e = work[blockIdx.x];
    const PotrfTask t = tasks[e.task];
    const int nb = t.nb;
    v4d acc[4];                                          // wave w: columns c = 16w + lane/16 + 4g, rhs 16 jb + lane%16
#pragma unroll
    for (int jb = 0; jb < 4; jb++) acc[jb] = (v4d){0.0, 0.0, 0.0, 0.0};
    double lr[16], li[E == 2 ? 16 : 1], xv[16];
    // thread's share of tile tt: the MFMA A operand L[r0 + 4q + lane/16][16w + lane%16] (q = 0..15), straight from
    // global memory (runs of 4 rows; the 4 k-steps that share a cache line are issued back to back), and
    // Y[row(r0 + w + 4i)][j0 + lane] (i = 0..15) for the LDS image
    auto fetch = [&](int tt) {
        const int r0 = t.urow0 + (e.tile0 + tt) * SOLVE_ROWS;
        const int rows_here = min(SOLVE_ROWS, t.lda - r0);
        const int c = 16 * w + l15;
        const int64_t coff = t.a_off + (r0 - t.col0) + (int64_t)(c < nb ? c : 0) * t.lda;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int rr = 4 * q + l4;
            const bool ok = rr < rows_here && c < nb;
            if constexpr (E == 2) {
                const double2 v = ok ? reinterpret_cast<const double2 *>(Lx)[coff + rr] : make_double2(0.0, 0.0);
                lr[q] = v.x; li[q] = v.y;
            } else {
                lr[q] = ok ? Lx[coff + rr] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int rr = w + 4 * i, r = r0 + rr;
            double v = 0.0;
            if (rr < rows_here && lane < nj) {
                const int64_t row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
                v = Y[row * ldr + j0 + lane];
            }
            xv[i] = v;
        }
    };
    fetch(0);
    for (int tt = 0; tt < e.ntiles; tt++) {
        const int r0 = t.urow0 + (e.tile0 + tt) * SOLVE_ROWS;
        const int nq = (min(SOLVE_ROWS, t.lda - r0) + 3) >> 2;    // k-steps of 4 rows that hold anything
        __syncthreads();                                 // the previous tile's operand reads are done
#pragma unroll
        for (int i = 0; i < 16; i++) Xt[lane][w + 4 * i] = xv[i];
        double a[16], a2[E == 2 ? 16 : 1];
#pragma unroll
        for (int q = 0; q < 16; q++) { a[q] = lr[q]; if constexpr (E == 2) a2[q] = li[q]; }
        __syncthreads();
        if (tt + 1 < e.ntiles) fetch(tt + 1);            // in flight during the products below
        if (16 * w < nb) {                               // wave-uniform; no barrier inside
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if (q >= nq) break;
#pragma unroll
                for (int jb = 0; jb < 4; jb++) {
                    if (16 * jb >= nj) break;
                    const int j = 16 * jb + l15;
                    acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], Xt[j][4 * q + l4], acc[jb], 0, 0, 0);
                    if constexpr (E == 2) {
                        // conj(L)^T y: (re, im) <- Li * (+yi, -yr)
                        const double v = Xt[j ^ 1][4 * q + l4];
                        acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[q], (j & 1) ? -v : v, acc[jb], 0, 0, 0);
                    }
                }
            }
        }
    }
    double *__restrict__ x1 = Y + (int64_t)(t.k1 + t.col0) * ldr + j0 + l15;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int c = 16 * w + l4 + 4 * g;
#pragma unroll
        for (int jb = 0; jb < 4; jb++)
            if (c < nb && 16 * jb + l15 < nj) atomic_add_f64(x1 + (int64_t)c * ldr + 16 * jb, -acc[jb][g]);
    }
}

// x1 <- inv(L11) x1 (TRANS = false) or inv(L11)^H x1 (TRANS = true) for many right-hand sides: a 64 x 64 x nrhs
// product per block on the MFMA pipe.  One workgroup per block; wave w owns rows [16w, 16w+16) of the result.
template <bool TRANS, int E>
__global__ __launch_bounds__(256) void k_solve_diag_mma(
    const PotrfTask *__restrict__ tasks, const double *__restrict__ Dinv,
    double *__restrict__ Y, int64_t ldr, int nrhs)
{
    __shared__ double Ds[64][SM_LD];                     // Ds[i][k] = Re D(i,k), or Re D(k,i) when TRANS
    __shared__ double Xs[64][SM_LD];                     // Xs[j][k] = x1[k][j0 + j]
    __shared__ double Ds2[E == 2 ? 64 : 1][SM_LD];       // complex: Im D the same way
    __shared__ double Xs2[E == 2 ? 64 : 1][SM_LD];       // complex: pair-swapped image of x1
    const PotrfTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nb = t.nb;
    double *__restrict__ x1 = Y + (int64_t)(t.k1 + t.col0) * ldr;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx & 63, k = idx >> 6;            // D is column-major, ld nb: consecutive threads read a column
        double vr = 0.0, vi = 0.0;
        if (i < nb && k < nb) {
            if constexpr (E == 2) {
                const double2 v = reinterpret_cast<const double2 *>(Dinv)[t.dinv_off + i + k * nb];
                vr = v.x; vi = v.y;
            } else {
                vr = Dinv[t.dinv_off + i + k * nb];
            }
        }
        if (TRANS) Ds[k][i] = vr; else Ds[i][k] = vr;
        if constexpr (E == 2) { if (TRANS) Ds2[k][i] = vi; else Ds2[i][k] = vi; }
    }
    const int nq = (nb + 3) >> 2;
    for (int j0 = 0; j0 < nrhs; j0 += 64) {
        const int nj = min(64, nrhs - j0);
        __syncthreads();                                 // Ds filled / previous chunk's reads and writes done
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int j = idx & 63, k = idx >> 6;
            const bool in = k < nb && j < nj;
            Xs[j][k] = in ? x1[(int64_t)k * ldr + j0 + j] : 0.0;
            if constexpr (E == 2) Xs2[j][k] = in ? pair_image<TRANS>(x1 + (int64_t)k * ldr + j0, j) : 0.0;
        }
        __syncthreads();
        if (16 * w < nb) {
            v4d acc[4];
#pragma unroll
            for (int jb = 0; jb < 4; jb++) acc[jb] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if (q >= nq) break;
                const double a = Ds[16 * w + l15][4 * q + l4];
                double a2 = 0.0;
                if constexpr (E == 2) a2 = Ds2[16 * w + l15][4 * q + l4];
#pragma unroll
                for (int jb = 0; jb < 4; jb++) {
                    if (16 * jb >= nj) break;
                    acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xs[16 * jb + l15][4 * q + l4], acc[jb], 0, 0, 0);
                    if constexpr (E == 2)
                        acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Xs2[16 * jb + l15][4 * q + l4], acc[jb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int i = 16 * w + l4 + 4 * g;       // result row = lane/16 + 4g, right-hand side = lane%16
#pragma unroll
                for (int jb = 0; jb < 4; jb++)
                    if (i < nb && 16 * jb + l15 < nj) x1[(int64_t)i * ldr + j0 + 16 * jb + l15] = acc[jb][g];
            }
        }
    }
}

//------------------------------------------------------------------------------
// k_solve_group_mma: k_solve_group for many right-hand sides (row-major image Y, 64 real columns per pass, the
// group's rows of Y in LDS).  Same chain of steps (c, b); each step is a 64 x 64 x 64 product on the MFMA pipe:
// the block of L (or the block inverse) is the A operand, read from global memory straight in operand layout
// (wave w owns result rows [16w, 16w+16) of the target block), the source rows of Y the B operand from LDS.
// The operands of the next NSETS steps are in flight in a ring of register sets (all ten steps for a real
// factor, five for a complex one, whose steps take Re and Im of the block).  Complex: second product with
// Im(block) against the pair-swapped image of Y, as in k_solve_fwd_mma / k_solve_bwd_mma.
//------------------------------------------------------------------------------
constexpr int SG_CW = 16;      // real columns of Y per workgroup of k_solve_group_mma
// The chain of one group on the CW columns [j0, j0 + nj) of the image y0 (row r of the group at y0 + r * ldr), Yg: the
// workgroup's LDS image of those rows.  Shared by the solves (k_solve_group_mma: y0 = the group's rows of Y) and the
// factorization (k_panel_trsm: y0 = the group's columns of the PANEL, whose rows are the right-hand sides).
template <bool TRANS, int E, int MAXBLK, int CW, int RING = 0 /* register sets of L's blocks in flight; 0: as many as there are steps, up to 8 (4 complex) */>
__device__ __forceinline__ void solve_group_mma_body(
    const SolveGroupTask &t, const double *__restrict__ Dinv, const double *__restrict__ Lx,
    double *__restrict__ y0, int64_t ldr, int j0, int nj, double (*Yg)[CW + 2])
{
    constexpr int NS = MAXBLK * (MAXBLK + 1) / 2;
    constexpr int NSETS = RING > 0 ? (RING < NS ? RING : NS) : (E == 1) ? (NS < 8 ? NS : 8) : (NS < 4 ? NS : 4);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ncols = t.ncols, nblk = (ncols + NB - 1) / NB;
    // A operand of step (c, b): A[i][k], i = 16w + lane%16, k = 4q + lane/16 -- entry (i, k) of the block, or
    // the conjugate of entry (k, i) when TRANS; [q] real parts, [16 + q] imaginary parts
    auto load = [&](int b, int c, double (&m)[16 * E]) {
        const int nbb = min(NB, ncols - NB * b), nbc = min(NB, ncols - NB * c);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = TRANS ? 4 * q + l4 : 16 * w + l15;
            const int col = TRANS ? 16 * w + l15 : 4 * q + l4;
            bool ok;
            int64_t off;
            const double *__restrict__ base;
            if (c == b) { ok = row < nbb && col <= row; off = t.dinv_off[b] + row + col * nbb; base = Dinv; }
            else { ok = row < nbc && col < nbb; off = t.a_off + NB * c + row + (int64_t)(NB * b + col) * t.lda; base = Lx; }
            if constexpr (E == 2) {
                const double2 v = ok ? reinterpret_cast<const double2 *>(base)[off] : make_double2(0.0, 0.0);
                m[q] = v.x;
                m[16 + q] = TRANS ? -v.y : v.y;
            } else {
                m[q] = ok ? base[off] : 0.0;
            }
        }
    };
    {
        double m[NSETS][16 * E];
#pragma unroll
        for (int u = 0; u < NSETS; u++) {
            const int s = TRANS ? NS - 1 - u : u;
            if (sg_c(s) < nblk) load(sg_b(s), sg_c(s), m[u]);
        }
        // rows of Y the group owns -> LDS (thread: column tid % CW, rows tid / CW + (256 / CW) i)
        {
            constexpr int RS = 256 / CW;
            const int j = tid % CW, rq = tid / CW;
#pragma unroll
            for (int i = 0; i < MAXBLK * NB / RS; i++) {
                const int r = rq + RS * i;
                Yg[r][j] = (r < ncols && j < nj) ? y0[(int64_t)r * ldr + j0 + j] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NS; u++) {
            const int s = TRANS ? NS - 1 - u : u;
            const int b = sg_b(s), c = sg_c(s);
            if (c < nblk) {                                  // uniform over the workgroup
                const int src = (TRANS || c == b) ? c : b, dst = (TRANS && c != b) ? b : c;
                const double (&ms)[16 * E] = m[u % NSETS];
                constexpr int NJB = CW / 16;
                v4d acc[NJB];
#pragma unroll
                for (int jb = 0; jb < NJB; jb++) acc[jb] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 16; q++) {
#pragma unroll
                    for (int jb = 0; jb < NJB; jb++) {
                        if (16 * jb >= nj) break;
                        const double *__restrict__ yr = &Yg[NB * src + 4 * q + l4][0];
                        acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ms[q], yr[16 * jb + l15], acc[jb], 0, 0, 0);
                        if constexpr (E == 2) {
                            // Im(block) against the pair image: (re, im) <- (-xi, +xr); for the conjugate transpose the
                            // sign already sits in ms[16 + q]
                            const int j = 16 * jb + l15;
                            const double v = yr[j ^ 1];
                            acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ms[16 + q], (j & 1) ? v : -v, acc[jb], 0, 0, 0);
                        }
                    }
                }
                if (c == b) __syncthreads();                 // the diagonal solve overwrites the rows it has just read
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    double *__restrict__ yd = &Yg[NB * dst + 16 * w + l4 + 4 * g][0];
#pragma unroll
                    for (int jb = 0; jb < NJB; jb++) {
                        if (16 * jb >= nj) break;
                        if (c == b) yd[16 * jb + l15] = acc[jb][g]; else yd[16 * jb + l15] -= acc[jb][g];
                    }
                }
                __syncthreads();
            }
            if (u + NSETS < NS) {
                const int s2 = TRANS ? NS - 1 - (u + NSETS) : u + NSETS;
                if (sg_c(s2) < nblk) load(sg_b(s2), sg_c(s2), m[u % NSETS]);
            }
        }
        {
            constexpr int RS = 256 / CW;
            const int j = tid % CW, rq = tid / CW;
#pragma unroll 16
            for (int i = 0; i < MAXBLK * NB / RS; i++) {
                const int r = rq + RS * i;
                if (r < ncols && j < nj) y0[(int64_t)r * ldr + j0 + j] = Yg[r][j];
            }
        }
    }
}

template <bool TRANS, int E, int MAXBLK, int CW = SG_CW>
__global__ __launch_bounds__(256) void k_solve_group_mma(
    const SolveGroupTask *__restrict__ groups, const double *__restrict__ Dinv, const double *__restrict__ Lx,
    double *__restrict__ Y, int64_t ldr, int nrhs)
{
    __shared__ double Yg[MAXBLK * NB][CW + 2];
    const SolveGroupTask t = groups[blockIdx.x];
    // blockIdx.y: which CW real columns of Y this workgroup sweeps (a group's chain is MFMA-bound on one
    // CU -- 10 steps x 64 x 64 x nrhs -- so the right-hand sides are spread over workgroups)
    const int j0 = blockIdx.y * CW;
    solve_group_mma_body<TRANS, E, MAXBLK, CW>(t, Dinv, Lx, Y + (int64_t)t.x0 * ldr, ldr, j0, min(CW, nrhs - j0), Yg);
}

//------------------------------------------------------------------------------
// k_solve_group_inv (tuning key 31): the group's step of the sweeps as products with the explicit inverse of the group's whole
// diagonal block, x1 <- inv(L_g) x1 or inv(L_g)^H x1, spread over ten workgroups.
// Why: the group step is what the chain of the top supernodes is made of, and its time is the time ONE CU needs to fetch the
// group's ten 64 x 64 blocks -- 320 KB at the ~25 GB/s a single CU draws = 13 of the ~19 us of k_solve_group_mma, whose ten block
// steps depend on each other and so cannot leave their CU.  (Measured: the same products with the explicit inverse in ONE workgroup,
// all loads independent, still took 18 us.)  With the inverse every product reads the ORIGINAL rows of Y, so each of the ten
// (target block, source block) pairs goes to a workgroup of its own: 32 KB each.  The partial results meet in a zeroed scratch
// tile per (group, slab of 16 real columns) by FP64 atomics; the workgroup that takes the last ticket of its tile copies the tile
// over the group's rows of Y -- nobody reads those rows any more by then -- and leaves tile and ticket zeroed for the next launch.
// No workgroup waits for another one.
// The inverses are built once per factorization, at the first solve after it, by k_solve_group_mma itself on the identity as
// right-hand sides (engine.hip, ensure_group_inverses): row-major images M[r * W + j], W = 256 entries per row (complex: (re, im)
// pairs), rows r < ncols only, columns j >= ncols zero -- one of inv(L_g) for the forward sweep (lower triangular) and one of
// inv(L_g)^H for the backward sweep (UPPER), so that both sweeps read their A operand along the rows of an image: a lane takes
// 16 consecutive entries (the contraction index of a block is dealt k = 16 (lane / 16) + q, on both operands).
// grid: (groups of the step, 10 block pairs per slab of SG_CW real columns); wave w owns rows [16w, 16w + 16) of the target block.
//------------------------------------------------------------------------------
template <bool UPPER, int E>
__global__ __launch_bounds__(256) void k_solve_group_inv(
    const SolveGroupTask *__restrict__ groups, const int64_t *__restrict__ ginv_off, const double *__restrict__ Ginv,
    double *__restrict__ Y, int64_t ldr, int nrhs, double *__restrict__ scratch, int *__restrict__ tickets,
    int slab_cols, int64_t slab_stride)
{
    // slab_cols / slab_stride: columns of Y per slab and the distance between two slabs -- SG_CW and SG_CW on the row-major image;
    // 1 and the leading dimension for a few COLUMN-major right-hand sides (ldr = 1: a slab is one column, a sixteenth of the products)
    constexpr int W = SOLVE_GROUP * NB, CW = SG_CW;
    __shared__ double Ys[NB][CW + 2];
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ncols = groups[blockIdx.x].ncols, nblk = (ncols + NB - 1) / NB;
    const int slab = blockIdx.y / 10, pair = blockIdx.y % 10;
    const int c = sg_c(pair), b = sg_b(pair);                               // pair (c, b), b <= c < nblk
    if (c >= nblk) return;                                                  // (uniform: the whole workgroup leaves, no ticket)
    const int tt = UPPER ? b : c, ss = UPPER ? c : b;
    double *__restrict__ y0 = Y + (int64_t)groups[blockIdx.x].x0 * ldr;
    const double *__restrict__ G = Ginv + ginv_off[blockIdx.x] * E;
    const int64_t j0 = (int64_t)slab * slab_stride;
    const int nj = min(slab_cols, nrhs - slab * slab_cols);
    const int64_t slot = (int64_t)blockIdx.x * (gridDim.y / 10) + slab;
    double *__restrict__ S = scratch + slot * (W * CW);
    // source rows of Y (block ss) -> LDS
    {
        constexpr int RS = 256 / CW;
        const int j = tid % CW, rq = tid / CW;
#pragma unroll
        for (int i = 0; i < NB / RS; i++) {
            const int r = NB * ss + rq + RS * i;
            Ys[rq + RS * i][j] = (r < ncols && j < nj) ? y0[(int64_t)r * ldr + j0 + j] : 0.0;
        }
    }
    // A operand: A[i][k], i = 16w + lane%16, k = 16 (lane/16) + q: entry (64 tt + i, 64 ss + k) of the image; rows of the image
    // beyond ncols do not exist (address clamped, value zeroed)
    double m[16 * E];
    {
        const int r = NB * tt + 16 * w + l15;
        const double2 *__restrict__ p2 = reinterpret_cast<const double2 *>(G + ((int64_t)min(r, ncols - 1) * W + NB * ss + 16 * l4) * E);
        const bool ok = r < ncols;
        if constexpr (E == 2) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const double2 v = p2[q];
                m[q] = ok ? v.x : 0.0;
                m[16 + q] = ok ? v.y : 0.0;
            }
        } else {
#pragma unroll
            for (int q2 = 0; q2 < 8; q2++) {
                const double2 v = p2[q2];
                m[2 * q2] = ok ? v.x : 0.0;
                m[2 * q2 + 1] = ok ? v.y : 0.0;
            }
        }
    }
    __syncthreads();
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const double *__restrict__ yr = &Ys[16 * l4 + q][0];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m[q], yr[l15], acc, 0, 0, 0);
        if constexpr (E == 2) {
            const double v = yr[l15 ^ 1];                // Im(block) against the pair image, as in k_solve_group_mma
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m[16 + q], (l15 & 1) ? v : -v, acc, 0, 0, 0);
        }
    }
    // partial result: rows 64 tt + 16 w + lane/16 + 4 g of the tile, column lane%16
#pragma unroll
    for (int g = 0; g < 4; g++) atomic_add_f64(&S[(NB * tt + 16 * w + l4 + 4 * g) * CW + l15], acc[g]);
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(&tickets[slot], 1) == nblk * (nblk + 1) / 2 - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // (plain loads: the tile was written by atomics only, which work at the L2; no line of it is in this CU's vector cache --
    // the cache starts a launch empty and no workgroup reads the tile before this point.  All loads first: one round trip)
    double2 v[W * CW / 512];
#pragma unroll
    for (int i = 0; i < W * CW / 512; i++) v[i] = reinterpret_cast<const double2 *>(S)[tid + 256 * i];
#pragma unroll
    for (int i = 0; i < W * CW / 512; i++) {
        const int idx = 2 * (tid + 256 * i), r = idx / CW, j = idx % CW;
        reinterpret_cast<double2 *>(S)[tid + 256 * i] = make_double2(0.0, 0.0);
        if (r < ncols && j < nj) y0[(int64_t)r * ldr + j0 + j] = v[i].x;
        if (r < ncols && j + 1 < nj) y0[(int64_t)r * ldr + j0 + j + 1] = v[i].y;
    }
    if (tid == 0) tickets[slot] = 0;
}

// the identity as right-hand sides of the group chains that build the inverses: entry (r, r) of every image
template <int E>
__global__ __launch_bounds__(256) void k_ginv_identity(
    const SolveGroupTask *__restrict__ tasks, int64_t ntasks, double *__restrict__ Ginv)
{
    constexpr int W = SOLVE_GROUP * NB;
    const int64_t gi = blockIdx.x;
    if (gi >= ntasks) return;
    const int r = threadIdx.x;
    if (r < tasks[gi].ncols) Ginv[(((int64_t)tasks[gi].x0 + r) * W + r) * E] = 1.0;
}

//------------------------------------------------------------------------------
// k_panel_trsm (factorization, STEP_PTRSM): the rows below a panel group's diagonal window, X * L_w^T = A for the window's
// <= 256 columns, in place.  Transposed, that is L_w * X^T = A^T: a forward solve with the window as the group and the
// ROWS of the panel as right-hand sides -- and the column-major panel (column c of the window at c * nsrow, row r at + r)
// is exactly the row-major image the group chain works on (row c of the image at c * ldr, right-hand side r at + r, ldr =
// nsrow).  So one workgroup takes ROWS rows of the panel through the whole window: the four products with the block
// inverses and the six updates between the blocks, ten 64 x 64 x ROWS MFMA steps on an LDS image of its rows, L's blocks
// in a ring of register sets -- where the per-block schedule needed four TRSM and three trailing launches, each reading
// and writing the window's columns again.
//------------------------------------------------------------------------------
template <int ROWS>
__global__ __launch_bounds__(256, ROWS <= 32 ? 2 : 1) void k_panel_trsm(
    const PanelGroup *__restrict__ groups, const int32_t *__restrict__ wg_task, const double *__restrict__ Dinv, double *Lx)
{
    __shared__ double Yg[SOLVE_GROUP * NB][ROWS + 2];
    const PanelGroup g = groups[wg_task[blockIdx.x]];
    SolveGroupTask t;
    t.a_off = g.a_off;
#pragma unroll
    for (int b = 0; b < SOLVE_GROUP; b++) t.dinv_off[b] = g.dinv_off[b];
    t.lda = g.lda; t.ncols = g.ncols; t.x0 = 0; t.pad = 0;
    const int r0 = g.ncols + (int)((int64_t)blockIdx.x - g.wg0) * ROWS;      // first row of this workgroup below the window's origin
    // image origin: the window's first column, at the row of the window's first row (so that "right-hand side" r0 + j
    // is the row r0 + j below the diagonal block's origin)
    // (three register sets of L's blocks in flight: the chain of a workgroup is short, what counts is how many workgroups a CU holds)
    solve_group_mma_body<false, 1, SOLVE_GROUP, ROWS, 3>(t, Dinv, Lx, Lx + g.a_off, (int64_t)g.lda, r0, min(ROWS, g.ncols + g.below - r0), Yg);
}

//------------------------------------------------------------------------------
// Wide sweeps (round 3): the rows below a solve group take the update of ALL the group's columns in one pass.
// The per-block kernels above touch the 64 x nrhs rows of Y of a row tile once per 64 columns of L: at nrhs = 64 a
// 32 KB tile of L moves 32 KB of Y in and 32 KB out, and the counters showed twice the algorithmic bytes per sweep
// (profiles/r02_pmc_traffic_solve64_lap3d150.txt).  Here the contraction runs over up to 256 columns (real; 128
// complex, all 2 x 64 real columns of Y in one pass, so a complex L is also read once instead of twice):
//   forward  k_solve_fwd_wide: x1 of the whole task (ncols x NJ) staged in LDS once per workgroup; every wave
//            streams 16-row slabs of L straight into MFMA operand layout through a ring of two register sets (the
//            next chunk of 64 columns is in flight while the current one is multiplied) and sends ONE set of
//            atomics per slab;
//   backward k_solve_bwd_wide: the products of a run of row tiles accumulate in registers for all the columns of the
//            task; L goes through LDS (global loads with lanes along the rows: whole cache lines, where the
//            per-block kernel read 32-byte runs), double-buffered per 64-column block, the gathered rows of Y are
//            fetched once per row tile and serve all the blocks.
// Complex L: P = Re(L) Y and Q = Im(L) Y accumulate separately against the SAME image of Y (Y as 2 nrhs real
// columns) and are combined between neighbouring lanes at the end (no pair-swapped second image):
//   L x   : (re, im) = (P_re - Q_im, P_im + Q_re);    L^H x : (re, im) = (P_re + Q_im, P_im - Q_re).
// 512 threads, one workgroup per CU (LDS), two waves per SIMD.
//------------------------------------------------------------------------------
constexpr int WIDE_THREADS = 512;
constexpr int WIDE_MAX_RUN = 64;          // most 64-row tiles of a run (build_solve_groups cuts the runs accordingly)

// The hot loops of both kernels are free of per-entry conditions (a guarded load or MFMA becomes a branch of its own
// and the compiler then waits for every load right where it is issued): addresses are clamped into the task (rows past
// its end, columns past its width, right-hand sides past nrr read valid entries), what must not count is multiplied
// by zeros of the LDS images or left out of the atomics.
// v or +0.0, without a select on the loaded value: the compiler turns "cond ? load : 0" back into a guarded load (a branch
// and a full wait per entry), a bit mask keeps the load unconditional
__device__ __forceinline__ double keep_if(double v, bool keep)
{
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    u &= keep ? ~0ull : 0ull;
    __builtin_memcpy(&v, &u, 8);
    return v;
}

template <int E, int KB, int NT, int NJB = 4 * E, int RING = 2 /* register sets of L in flight per wave: 2 or 3 */>
__global__ __launch_bounds__(NT) void k_solve_fwd_wide(
    const SolveGroupUpd *__restrict__ tasks, const SolveRange *__restrict__ work, const int32_t *__restrict__ wg_first,
    const double *__restrict__ Lx, const int32_t *__restrict__ Ls, double *__restrict__ Y, int64_t ldr,
    int nrr, int j0)
{
    constexpr int NJ = 16 * NJB;          // real columns of Y per pass: 64 E (all of 64 right-hand sides), or 16 for a few
                                          // right-hand sides (NJB = 1: an eighth of the products and of the LDS image)
    constexpr int LD = KB * 64 + 2;       // LDS row stride: 16 j x 2 k of a half-wave on 32 disjoint bank pairs
    constexpr int QC = 16 / E;            // k-steps (of 4 columns) per chunk of the operand ring
    constexpr int NW = NT / 64;
    __shared__ double Xs[NJ][LD];         // Xs[j][c] = x1[c][j0 + j]
    __shared__ int32_t rows_s[WIDE_MAX_RUN * 64];   // rows of Y the run updates (a load from Ls inside the loop would make
                                                    // the wave wait for every operand request that is in flight)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nj = min(NJ, nrr - j0);
    // the workgroup's share of the step: consecutive segments (runs of row tiles of one task) of the work list
    for (int seg = wg_first[blockIdx.x]; seg < wg_first[blockIdx.x + 1]; seg++) {
    const SolveRange e = work[seg];
    const SolveGroupUpd t = tasks[e.task];
    const int ncols = t.ncols, nq = (ncols + 3) >> 2;            // k-steps that hold anything
    const int ncq = (nq + QC - 1) / QC;                           // chunks per slab
    const int row0 = e.tile0 * 64;                                // first row (below the group) of the run
    const int rows_run = min(64 * min(e.ntiles, WIDE_MAX_RUN), t.nrows - row0);
    for (int i = tid; i < rows_run; i += NT) rows_s[i] = Ls[t.ls_off + row0 + i];
    const int nslab = (rows_run + 15) >> 4;
    const int nmine = nslab > w ? (nslab - w + NW - 1) / NW : 0;  // slabs w, w + NW, ...
    const int nch = nmine * ncq;
    double a0[QC * E], a1[QC * E], a2[QC * E];
    // chunk ch = (slab si, k-steps [QC cc, QC cc + QC)): this lane's entries L[row][4 q + lane/16]
    auto chunk_ptr = [&](int ch, int &q0) -> const double * {
        const int si = ch / ncq;
        q0 = (ch - si * ncq) * QC;
        const int r = min(row0 + (w + NW * si) * 16 + l15, t.nrows - 1);
        return Lx + (t.a_off + r) * E;
    };
    auto load1 = [&](double (&a)[QC * E], const double *__restrict__ base, int q0, int qq) {
        const int c = min(4 * (q0 + qq) + l4, ncols - 1);
        if constexpr (E == 2) {
            const double2 v = *reinterpret_cast<const double2 *>(base + (int64_t)c * t.lda * 2);
            a[2 * qq] = v.x; a[2 * qq + 1] = v.y;
        } else {
            a[qq] = base[(int64_t)c * t.lda];
        }
    };
    // the first two chunks are requested before x1 is staged: their latency runs under the staging
    {
        int q0; const double *b0 = chunk_ptr(0, q0);
#pragma unroll
        for (int qq = 0; qq < QC; qq++) load1(a0, b0, q0, qq);
        const double *b1 = chunk_ptr(nch > 1 ? 1 : 0, q0);
#pragma unroll
        for (int qq = 0; qq < QC; qq++) load1(a1, b1, q0, qq);
        if constexpr (RING == 3) {
            const double *b2 = chunk_ptr(nch > 2 ? 2 : 0, q0);
#pragma unroll
            for (int qq = 0; qq < QC; qq++) load1(a2, b2, q0, qq);
        }
    }
    {
        // x1 of the task, zero beyond its columns (up to the end of the last chunk) and beyond the right-hand sides;
        // batches of 8 independent requests per lane (one request per pass of a plain loop costs a memory latency each)
        const double *__restrict__ x1 = Y + (int64_t)t.x0 * ldr + j0;
        const int units = NJB * ncq * QC;
        for (int u0 = w; u0 < units; u0 += 8 * NW) {
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int u = min(u0 + NW * i, units - 1);
                const int jb = u % NJB, c = 4 * (u / NJB) + l4, j = 16 * jb + l15;
                v[i] = x1[(int64_t)min(c, ncols - 1) * ldr + min(j, nj - 1)];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int u = u0 + NW * i;
                const int jb = u % NJB, c = 4 * (u / NJB) + l4, j = 16 * jb + l15;
                if (u < units) Xs[j][c] = keep_if(v[i], c < ncols && j < nj);
            }
        }
    }
    __syncthreads();
    v4d P[NJB], Q[E == 2 ? NJB : 1];
#pragma unroll
    for (int jb = 0; jb < NJB; jb++) { P[jb] = (v4d){0.0, 0.0, 0.0, 0.0}; if constexpr (E == 2) Q[jb] = (v4d){0.0, 0.0, 0.0, 0.0}; }
    // chunk ch from register set a; with AGAIN its registers are re-loaded for chunk ch + RING as they are consumed
    auto run_chunk = [&](double (&a)[QC * E], int ch, auto again_tag) {
        constexpr bool AGAIN = decltype(again_tag)::value;
        const int si = ch / ncq, cc = ch - si * ncq;
        int q2 = 0;
        const double *__restrict__ p2 = AGAIN ? chunk_ptr(ch + RING, q2) : nullptr;
        const double *__restrict__ xs = &Xs[l15][4 * cc * QC + l4];
        // 8 waves: the x1 operands of k-step qq + 1 are read from LDS while the products of k-step qq run (16 waves have
        // no registers to spare for that, and four waves per SIMD to cover the read)
        constexpr bool AHEAD = (NT == 512);
        double bc[NJB], bn[NJB];
#pragma unroll
        for (int jb = 0; jb < NJB; jb++) bc[jb] = xs[16 * jb * LD];
#pragma unroll
        for (int qq = 0; qq < QC; qq++) {
            if (qq + 1 < QC && AHEAD) {
#pragma unroll
                for (int jb = 0; jb < NJB; jb++) bn[jb] = xs[16 * jb * LD + 4 * (qq + 1)];
            }
#pragma unroll
            for (int jb = 0; jb < NJB; jb++) {
                if constexpr (E == 2) {
                    P[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2 * qq], bc[jb], P[jb], 0, 0, 0);
                    Q[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2 * qq + 1], bc[jb], Q[jb], 0, 0, 0);
                } else {
                    P[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qq], bc[jb], P[jb], 0, 0, 0);
                }
            }
            if constexpr (AGAIN) load1(a, p2, q2, qq);
            if (qq + 1 < QC) {
#pragma unroll
                for (int jb = 0; jb < NJB; jb++) bc[jb] = AHEAD ? bn[jb] : xs[16 * jb * LD + 4 * (qq + 1)];
            }
        }
        if (cc == ncq - 1) {
            // the slab is complete: rows rbase + lane/16 + 4g (g = 0..3), real column 16 jb + lane%16
            const int rbase = row0 + (w + NW * si) * 16;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int rr = rbase + l4 + 4 * g;
                const bool okr = rr < t.nrows;
                const int64_t row = (int64_t)rows_s[min(rr - row0, rows_run - 1)];
                double *__restrict__ yp = Y + row * ldr + j0 + l15;
#pragma unroll
                for (int jb = 0; jb < NJB; jb++) {
                    double v = P[jb][g];
                    if constexpr (E == 2) {
                        const double qx = __shfl_xor(Q[jb][g], 1, 64);       // every lane takes part
                        v += (l15 & 1) ? qx : -qx;
                        Q[jb][g] = 0.0;
                    }
                    P[jb][g] = 0.0;
                    if (okr && 16 * jb + l15 < nj) atomic_add_f64(yp + 16 * jb, -v);
                }
            }
        }
    };
    for (int ch = 0; ch < nch; ch += RING) {
        if (ch + RING < nch) run_chunk(a0, ch, std::true_type{}); else run_chunk(a0, ch, std::false_type{});
        if (ch + 1 < nch) {
            if (ch + 1 + RING < nch) run_chunk(a1, ch + 1, std::true_type{}); else run_chunk(a1, ch + 1, std::false_type{});
        }
        if constexpr (RING == 3) {
            if (ch + 2 < nch) {
                if (ch + 2 + RING < nch) run_chunk(a2, ch + 2, std::true_type{}); else run_chunk(a2, ch + 2, std::false_type{});
            }
        }
    }
    __syncthreads();                                              // the next segment overwrites Xs and rows_s
    }
}

template <int E, int KB, int NJB = 4 * E>
__global__ __launch_bounds__(WIDE_THREADS) void k_solve_bwd_wide(
    const SolveGroupUpd *__restrict__ tasks, const SolveRange *__restrict__ work, const int32_t *__restrict__ wg_first,
    const double *__restrict__ Lx, const int32_t *__restrict__ Ls, double *__restrict__ Y, int64_t ldr,
    int nrr, int j0)
{
    constexpr int NJ = 16 * NJB;          // real columns of Y per pass: 64 E, or 16 for a few right-hand sides (NJB = 1)
    constexpr int RT = 64 / E;            // rows per tile step (the same bytes of L per step, real and complex)
    constexpr int LDT = RT + 2;
    constexpr int NQ = RT / 4;            // k-steps per tile
    constexpr int NW = WIDE_THREADS / 64;
    constexpr int JW = NJB >= 2 ? NJB / 2 : 1;       // j-blocks per wave
    // the 8 waves: 4 column blocks x 2 halves of the j-blocks -- or, with one j-block, x 2 halves of the tile's rows (both
    // halves add their partial sums into x1 at the end)
    constexpr int QH = NJB >= 2 ? NQ : NQ / 2;       // k-steps per wave and tile
    constexpr int NYT = (NJ * RT + WIDE_THREADS - 1) / WIDE_THREADS;
    constexpr int NL = 64 * RT * E / WIDE_THREADS;   // doubles of an L block per thread (8)
    constexpr int NY = NYT;                          // entries of the gathered rows of Y per thread (8; 2 with NJB = 1)
    __shared__ double Xt[2][NJ][LDT];     // Xt[.][j][r]
    __shared__ double Lt[2][E][64][LDT];  // Lt[.][re/im][c][r]
    __shared__ int32_t rows_s[WIDE_MAX_RUN * 64];   // rows of Y the run gathers
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nj = min(NJ, nrr - j0);
    const int cb = w & 3, jh = w >> 2;    // this wave: columns 16 cb + [0, 16) of every block, j-blocks JW jh + [0, JW)
    // the workgroup's share of the step: consecutive segments (runs of row tiles of one task) of the work list
    for (int seg = wg_first[blockIdx.x]; seg < wg_first[blockIdx.x + 1]; seg++) {
    const SolveRange e = work[seg];
    const SolveGroupUpd t = tasks[e.task];
    const int ncols = t.ncols;
    const int row0 = e.tile0 * 64;
    const int rows_run = min(64 * min(e.ntiles, WIDE_MAX_RUN), t.nrows - row0);
    const int ntile = (rows_run + RT - 1) / RT;
    v4d P[KB][JW], Q[E == 2 ? KB : 1][JW];
#pragma unroll
    for (int b = 0; b < KB; b++)
#pragma unroll
        for (int jj = 0; jj < JW; jj++) { P[b][jj] = (v4d){0.0, 0.0, 0.0, 0.0}; if constexpr (E == 2) Q[b][jj] = (v4d){0.0, 0.0, 0.0, 0.0}; }
    double lreg[2][NL], yreg[NY];
    const double *__restrict__ Lbase = Lx + (t.a_off + row0) * E;
    // block b of tile ti -> registers: real: lane = row, 8 columns per wave; complex: 32 rows x 2 column quads per wave.
    // Rows past the end of the run meet zero rows of Xt, columns past the task are never flushed: both read clamped.
    auto load_L = [&](double (&a)[NL], int ti, int b) {
        if constexpr (E == 1) {
            const int r = min(ti * RT + lane, rows_run - 1);
#pragma unroll
            for (int i = 0; i < NL; i++) {
                const int c = min(64 * b + 8 * w + i, ncols - 1);
                a[i] = Lbase[r + (int64_t)c * t.lda];
            }
        } else {
            const int r = min(ti * RT + (lane & 31), rows_run - 1);
#pragma unroll
            for (int i = 0; i < NL / 2; i++) {
                const int c = min(64 * b + 4 * (2 * w + (lane >> 5)) + i, ncols - 1);
                const double2 v = *reinterpret_cast<const double2 *>(Lbase + 2 * (r + (int64_t)c * t.lda));
                a[2 * i] = v.x; a[2 * i + 1] = v.y;
            }
        }
    };
    auto store_L = [&](const double (&a)[NL], int buf) {
        if constexpr (E == 1) {
#pragma unroll
            for (int i = 0; i < NL; i++) Lt[buf][0][8 * w + i][lane] = a[i];
        } else {
#pragma unroll
            for (int i = 0; i < NL / 2; i++) {
                const int c = 4 * (2 * w + (lane >> 5)) + i;
                Lt[buf][0][c][lane & 31] = a[2 * i];
                Lt[buf][1][c][lane & 31] = a[2 * i + 1];
            }
        }
    };
    // gathered rows of Y of tile ti: unit u = w + NW i -> rows 4 (u / NJB) + lane/16, real columns 16 (u % NJB) + lane%16
    auto load_Y = [&](int ti) {
#pragma unroll
        for (int i = 0; i < NY; i++) {
            const int u = w + NW * i, j = 16 * (u % NJB) + l15, r = ti * RT + 4 * (u / NJB) + l4;
            static_assert(NW * NY == (RT / 4) * NJB, "the units of a tile's rows of Y divide evenly over the waves");
            yreg[i] = Y[(int64_t)rows_s[min(r, rows_run - 1)] * ldr + j0 + min(j, nj - 1)];
        }
    };
    // (the mask is applied here, not at the load: the wave would wait for the loads where they are issued)
    auto store_Y = [&](int slot, int ti) {       // tile ti into buffer slot & 1
#pragma unroll
        for (int i = 0; i < NY; i++) {
            const int u = w + NW * i, j = 16 * (u % NJB) + l15, r = ti * RT + 4 * (u / NJB) + l4;
            Xt[slot & 1][j][4 * (u / NJB) + l4] = keep_if(yreg[i], r < rows_run && j < nj);
        }
    };
    // the first two blocks of L are requested before the row list is staged (their latency runs under it)
    load_L(lreg[0], 0, 0);
    load_L(lreg[1], 0, 1);                                        // KB >= 2: phase 1 of tile 0 exists
    for (int i = tid; i < rows_run; i += WIDE_THREADS) rows_s[i] = Ls[t.ls_off + row0 + i];
    __syncthreads();
    load_Y(0);
    store_L(lreg[0], 0);
    store_Y(0, 0);
    __syncthreads();
    for (int ti = 0; ti < ntile; ti++) {
        // Every phase issues the same loads and stores whether or not a next tile or block exists (past the end they
        // repeat the last tile / a clamped column and land in buffers nobody reads): with requests that are issued
        // on some paths only, the compiler can no longer count how many younger ones may stay in flight and waits
        // for all of them at every use.
        const int tn = min(ti + 1, ntile - 1);
#pragma unroll
        for (int b = 0; b < KB; b++) {
            // phase p = KB ti + b works on Lt[b & 1] (KB is even); block p + 1 is in lreg[(b + 1) & 1], block p + 2 is
            // requested now into lreg[b & 1] (written to LDS at the end of the previous phase)
            if (b == 0) load_Y(tn);
            load_L(lreg[b & 1], min(ti + (b + 2) / KB, ntile - 1), (b + 2) % KB);
            if (64 * b < ncols) {                                 // uniform; no memory requests inside
                const int qoff = (NJB >= 2) ? 0 : 4 * QH * jh;        // one j-block: wave half jh takes the k-steps [QH jh, QH jh + QH)
                const double *__restrict__ la = &Lt[b & 1][0][16 * cb + l15][l4 + qoff];
                const double *__restrict__ xa = &Xt[ti & 1][(NJB >= 2 ? 16 * JW * jh : 0) + l15][l4 + qoff];
#pragma unroll
                for (int q = 0; q < QH; q++) {
                    const double ar = la[4 * q];
                    double ai = 0.0;
                    if constexpr (E == 2) ai = la[64 * LDT + 4 * q];
#pragma unroll
                    for (int jj = 0; jj < JW; jj++) {
                        const double y = xa[16 * jj * LDT + 4 * q];
                        P[b][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, y, P[b][jj], 0, 0, 0);
                        if constexpr (E == 2) Q[b][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, y, Q[b][jj], 0, 0, 0);
                    }
                }
            }
            store_L(lreg[(b + 1) & 1], (b + 1) & 1);
            if (b == KB - 1) store_Y(ti + 1, tn);
            __syncthreads();
        }
    }
    // x1[c][j] -= sum: lane holds columns 64 b + 16 cb + lane/16 + 4g, real column 16 (JW jh + jj) + lane%16
    double *__restrict__ x1 = Y + (int64_t)t.x0 * ldr + j0 + l15;
#pragma unroll
    for (int b = 0; b < KB; b++)
#pragma unroll
        for (int jj = 0; jj < JW; jj++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int c = 64 * b + 16 * cb + l4 + 4 * g, jo = (NJB >= 2) ? 16 * (JW * jh + jj) : 0;
                double v = P[b][jj][g];
                if constexpr (E == 2) {
                    const double qx = __shfl_xor(Q[b][jj][g], 1, 64);        // every lane takes part
                    v += (l15 & 1) ? -qx : qx;
                }
                if (c < ncols && jo + l15 < nj) atomic_add_f64(x1 + (int64_t)c * ldr + jo, -v);
            }
    __syncthreads();                                              // the next segment overwrites the LDS images
    }
}

//------------------------------------------------------------------------------
// k_panel_trsm_reg (STEP_PTRSM, 64 rows per workgroup): the same solve with the roles turned round.  k_panel_trsm above
// keeps the panel's rows in LDS and L's blocks in registers, so every workgroup fetches the window's ten blocks (320 KB)
// for ROWS rows of the panel: ten times the panel's own bytes out of L2 at 16 rows per workgroup, which is what bounds
// it.  Here every wave keeps ITS 16 rows of the panel -- all 256 columns, 64 doubles per lane -- in registers from the
// first load to the last store, and the blocks of the window go through LDS once per workgroup, double-buffered, one
// barrier per block.  The result registers of one product (16 x 64, rows along lane % 16) ARE the operand registers of
// the next (k_trsm64's layout), so X_b never leaves the register file between the product with the inverse that makes
// it and the updates that use it: T_c -= X_b L(c,b)^T for b < c, then X_c = T_c W_c^T, block after block.
//------------------------------------------------------------------------------
template <int WAVES>      // 4: 64 rows per workgroup, two workgroups per CU; 8: 128 rows, one workgroup (the blocks of the window are fetched half as often)
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void k_panel_trsm_reg(
    const PanelGroup *__restrict__ groups, const int32_t *__restrict__ wg_task, const double *__restrict__ Dinv, double *Lx)
{
    __shared__ double Bs[2][NB][NB + 2];                 // Bs[.][n][k]: block (c, b) of the window as L(64c + n, 64b + k), or W_c(n, k)
    const int gi = wg_task[blockIdx.x];
    const PanelGroup g = groups[gi];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ncols = g.ncols, nblk = (ncols + NB - 1) / NB;
    const int64_t lda = g.lda;
    const int nrows = ncols + g.below;                   // rows of the panel from the window's first row down
    const int r = ncols + (int)((int64_t)blockIdx.x - g.wg0) * (16 * WAVES) + 16 * w + l15;
    const bool ok = r < nrows;
    double *Ar = Lx + g.a_off + (ok ? r : nrows - 1);    // this lane's row (clamped: loaded, never stored)
    const uint32_t roff = (uint32_t)(ok ? r : nrows - 1) + (uint32_t)l4 * (uint32_t)lda;   // ... and its lane offset from a column's top
    // the stages: for c = 0 .. nblk - 1: blocks (c, 0) .. (c, c - 1), then the inverse of block c
    auto stage_bc = [](int s, int &b, int &c) { c = 0; while (s > c) { s -= c + 1; c++; } b = s; };   // s -> (b <= c)
    const int nstages = nblk * (nblk + 1) / 2;
    constexpr int NS = 16 * 4 / WAVES;                   // entries of a 64 x 64 block per thread
    constexpr int KS = WAVES;                            // columns of a block between two entries of a thread
    double sreg[NS];
    // (addresses are kept affine -- one pointer and a uniform stride per batch of loads -- wherever the block is known to
    // be full: with a clamp or a select per load the compiler computes every address of a batch up front, 64-bit each)
    auto fetch = [&](int s) {                            // the block of stage s -> sreg (thread: n = idx % 64, k = idx / 64)
        int b, c;
        stage_bc(s, b, c);
        const int nbc = min(NB, ncols - NB * c);
        const int n = tid & 63, k0 = tid >> 6;
        if (b == c) {
            // the inverse of block c: nbc x nbc, ld = nbc (the last block of a window may be narrower than 64)
            const double *p = Dinv + groups[gi].dinv_off[c] + min(n, nbc - 1);   // (read in place: indexing a private copy of g by c would put it in scratch)
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const int k = k0 + KS * i;
                sreg[i] = keep_if(p[min(k, nbc - 1) * nbc], n < nbc && k < nbc);
            }
        } else {
            // block (c, b), b < c: all 64 columns of block b exist
            // (uniform base + 32-bit lane offset: the saddr form of global_load, no 64-bit address per load in VGPRs;
            // lda < 2^27 for a panel group, schedule.cpp)
            const uint32_t voff = (uint32_t)min(n, nbc - 1) + (uint32_t)k0 * (uint32_t)lda;
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const double *pu = Lx + g.a_off + NB * c + (int64_t)(NB * b + KS * i) * lda;
                sreg[i] = keep_if(pu[voff], n < nbc);
            }
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const int idx = tid + 64 * WAVES * i;
            Bs[buf][idx & 63][idx >> 6] = sreg[i];
        }
    };
    fetch(0);
    // this wave's rows: t[c][q] = entry (r, 64 c + 4 q + lane / 16), the operand layout of the products
    double t[SOLVE_GROUP][16];
#pragma unroll
    for (int c = 0; c < SOLVE_GROUP; c++) {
        if (NB * (c + 1) <= ncols) {                     // a full block (uniform)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const double *pu = Lx + g.a_off + (int64_t)(NB * c + 4 * q) * lda;     // uniform
                t[c][q] = pu[roff];
            }
        } else if (NB * c < ncols) {                     // the last, narrower block of a window
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int col = NB * c + 4 * q + l4;
                t[c][q] = keep_if(Ar[(int64_t)min(col, ncols - 1) * lda], col < ncols);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) t[c][q] = 0.0;
        }
    }
    put(0);
    __syncthreads();
    if (nstages > 1) fetch(1);
#pragma unroll
    for (int c = 0; c < SOLVE_GROUP; c++) {
        if (c >= nblk) break;                            // uniform
#pragma unroll
        for (int b = 0; b <= c; b++) {
            const int s = c * (c + 1) / 2 + b;           // (a constant after unrolling)
            const double (*B)[NB + 2] = Bs[s & 1];
            // two of the four 16-column tiles of the result at a time (16 accumulator registers instead of 32: the rows
            // of the panel take 128 of the 256 a lane may have at two waves per SIMD).  The product with the inverse goes
            // from the last tile pair to the first and skips what lies above W's diagonal: tile cb needs the operand
            // registers q <= 4 cb + 3 only, so X_c can replace T_c in place, pair after pair.
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int cb0 = (b == c) ? 2 - 2 * h : 2 * h;
                v4d acc[2];
                acc[0] = (v4d){0.0, 0.0, 0.0, 0.0};
                acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 16; q++)
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int cb = cb0 + u;
                        if (b == c && q > 4 * cb + 3) continue;          // above the diagonal of W: zeros
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(B[16 * cb + l15][4 * q + l4], t[b][q], acc[u], 0, 0, 0);
                    }
                // result (r, 16 cb + lane / 16 + 4 gq) = operand register 4 cb + gq
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const int cb = cb0 + u;
                        if (b == c) t[c][4 * cb + gq] = acc[u][gq]; else t[c][4 * cb + gq] -= acc[u][gq];
                    }
            }
            if (b == c && ok) {
                // X_c is final.  (Tried: all stores at the end of the kernel, so that no wait in the chain has to cover
                // them -- 152 -> 178 us per launch at 150^3, the registers it costs spill.)
                const int nbc = min(NB, ncols - NB * c);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    double *pu = Lx + g.a_off + (int64_t)(NB * c + 4 * q) * lda;          // uniform
                    if (4 * q + l4 < nbc) pu[roff] = t[c][q];
                }
            }
            if (s + 1 < nstages) {
                put((s + 1) & 1);                        // (fetched while this stage was multiplied)
                __syncthreads();
                if (s + 2 < nstages) fetch(s + 2);
            }
        }
    }
}

}  // namespace mi355
