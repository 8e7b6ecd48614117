// kernels_complex.hip.hpp -- complex (Hermitian, LL^H) variants of the numeric and solve
// kernels: the device side of the cd_/zd_ instantiations of the reference worker
// (CHOLMOD/Supernodal/t_cholmod_super_numeric_worker.c:35-83: L is stored interleaved
// re,im; zherk/zgemm "N","C"; zpotrf; ztrsm "R","L","C","N") and of the complex solve
// workers (t_cholmod_super_solve_worker.c:127-161,246-266,420-448,520-560).
//
// Same schedule, same task structs, same tiling as the real kernels; every offset in a task
// is an ENTRY offset (16 bytes per entry here).  A complex product A*B^H is four real MFMA
// products on de-interleaved LDS images:  Cr = Ar Br^T + Ai Bi^T,  Ci = Ai Br^T - Ar Bi^T.
// As in LAPACK's zpotrf/zherk the imaginary part of a diagonal entry is never read and is
// written as exactly zero.
#pragma once
#include "kernels.hip.hpp"

namespace mi355 {

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_assemble_c(
    int64_t ncols, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
    const int64_t *__restrict__ Anz, const double *__restrict__ Ax,
    const int32_t *__restrict__ supermap, const int64_t *__restrict__ Super,
    const int64_t *__restrict__ Lpi, const int64_t *__restrict__ Lpx,
    const int32_t *__restrict__ Ls, double *__restrict__ Lx, double beta,
    const int8_t *__restrict__ smask)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    const int s = supermap[k];
    if (smask && !smask[s]) return;
    const int64_t k1 = Super[s], psi = Lpi[s], psx = Lpx[s];
    const int nsrow = (int)(Lpi[s + 1] - psi);
    const int64_t colbase = psx + (k - k1) * (int64_t)nsrow;
    int64_t p = Ap[k];
    const int64_t pend = Anz ? p + Anz[k] : Ap[k + 1];
    for (; p < pend; p++) {
        const int64_t i = Ai[p];
        if (i >= k) {
            const int imap = find_row(Ls + psi, nsrow, (int)i);
            if (imap >= 0) {
                Lx[2 * (colbase + imap)] = Ax[2 * p];
                Lx[2 * (colbase + imap) + 1] = Ax[2 * p + 1];
            }
        }
    }
    if (beta != 0.0) Lx[2 * (colbase + (k - k1))] += beta;      // only the real part of beta is used (worker :510-520)
}

//------------------------------------------------------------------------------
// k_gemm64c: complex C = A * B^H, 64x64 tile, 4 waves (2x2), 2x2 MFMA blocks x {re,im}
//------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gemm64c(
    const GemmTask *__restrict__ tasks, const Schedule::TileEntry *__restrict__ tile_tab, double *__restrict__ Lx,
    const double *__restrict__ Dinv, const int32_t *__restrict__ relmap)
{
    __shared__ double Ar[G64_KS][G64_LD], Ai_[G64_KS][G64_LD];
    __shared__ double Br[G64_KS][G64_LD], Bi_[G64_KS][G64_LD];
    __shared__ int rmM[TILE], rmN[TILE];

    const int tid = threadIdx.x;
    const int64_t bid = blockIdx.x;
    const Schedule::TileEntry te = tile_tab[bid];
    const GemmTask t = tasks[te.task];
    const int M = t.M, N = t.N, K = t.K, mode = t.mode;
    const int tm = (int)(te.tmtn & 0xffffu), tn = (int)(te.tmtn >> 16);
    const int m0 = tm * TILE, n0 = tn * TILE;

    const v2d *__restrict__ A = (const v2d *)Lx + t.a_off;
    const v2d *__restrict__ B = (const v2d *)(mode == GEMM_OVERWRITE ? Dinv : Lx) + t.b_off;
    const int64_t lda = t.lda, ldb = t.ldb;

    if (mode == GEMM_SCATTER_SUB) {
        if (tid < TILE) rmM[tid] = (m0 + tid < M) ? relmap[t.rm_off + m0 + tid] : 0;
        else if (tid < 2 * TILE) { int j = tid - TILE; rmN[j] = (n0 + j < N) ? relmap[t.rm_off + n0 + j] : 0; }
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    v4d accr[2][2], acci[2][2];                  // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) { accr[a][b] = (v4d){0.0, 0.0, 0.0, 0.0}; acci[a][b] = accr[a][b]; }

    const int lrow = tid & 63, lk = tid >> 6;
    const bool arow_ok = (m0 + lrow) < M, brow_ok = (n0 + lrow) < N;
    const v2d *Ap = A + (m0 + lrow);
    const v2d *Bp = B + (n0 + lrow);
    const v2d zero = {0.0, 0.0};

    for (int k0 = 0; k0 < K; k0 += G64_KS) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = lk + 4 * q, kk = k0 + k;
            const bool kok = kk < K;
            const v2d av = (arow_ok && kok) ? Ap[(int64_t)kk * lda] : zero;
            const v2d bv = (brow_ok && kok) ? Bp[(int64_t)kk * ldb] : zero;
            Ar[k][lrow] = av.x; Ai_[k][lrow] = av.y;
            Br[k][lrow] = bv.x; Bi_[k][lrow] = bv.y;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kk = ks * 4 + l4;
            double ar[2], ai[2], br[2], bi[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                ar[i] = Ar[kk][wm * 32 + i * 16 + l15];
                ai[i] = Ai_[kk][wm * 32 + i * 16 + l15];
                br[i] = Br[kk][wn * 32 + i * 16 + l15];
                bi[i] = Bi_[kk][wn * 32 + i * 16 + l15];
            }
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++) {
                    accr[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[ni], ar[mi], accr[ni][mi], 0, 0, 0);
                    accr[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[ni], ai[mi], accr[ni][mi], 0, 0, 0);
                    acci[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[ni], ai[mi], acci[ni][mi], 0, 0, 0);
                    acci[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bi[ni], ar[mi], acci[ni][mi], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    const int64_t ldc = t.ldc;
    double *__restrict__ C = Lx + 2 * t.c_off;
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
            const int ml = wm * 32 + mi * 16 + l15;
            const int m = m0 + ml;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nl = wn * 32 + ni * 16 + l4 + 4 * r;
                const int n = n0 + nl;
                if (m < M && n < N) {
                    const double vr = accr[ni][mi][r], vi = acci[ni][mi][r];
                    if (mode == GEMM_SCATTER_SUB) {
                        if (m >= n) {
                            double *q = C + 2 * (rmM[ml] + (int64_t)rmN[nl] * ldc);
                            atomic_add_f64(q, -vr);
                            if (m != n) atomic_add_f64(q + 1, -vi);     // zherk: diagonal of C is real
                        }
                    } else if (mode == GEMM_DIRECT_SUB) {
                        if (m >= n) {
                            double *q = C + 2 * (m + (int64_t)n * ldc);
                            q[0] -= vr;
                            if (m != n) q[1] -= vi;
                        }
                    } else {
                        double *q = C + 2 * (m + (int64_t)n * ldc);
                        q[0] = vr; q[1] = vi;
                    }
                }
            }
        }
    }
}

//------------------------------------------------------------------------------
// k_potrf64c: zpotrf of one nb x nb (nb <= 64) diagonal block + its inverse, all in LDS
//------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf64c(
    const PotrfTask *__restrict__ tasks, double *__restrict__ Lx,
    double *__restrict__ Dinv, int *__restrict__ info)
{
    __shared__ double Lr[NB][NB + 1], Li[NB][NB + 1];
    __shared__ double Xr[NB][NB + 1], Xi[NB][NB + 1];
    const PotrfTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x;
    const int nb = t.nb;
    const int64_t lda = t.lda;
    v2d *__restrict__ A = (v2d *)Lx + t.a_off;

    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx & (NB - 1), j = idx >> 6;
        double vr = (i == j) ? 1.0 : 0.0, vi = 0.0;             // identity padding
        if (i < nb && j < nb) {
            if (i >= j) { const v2d v = A[i + (int64_t)j * lda]; vr = v.x; vi = (i == j) ? 0.0 : v.y; }
            else vr = 0.0;
        }
        Lr[i][j] = vr; Li[i][j] = vi;
        Xr[i][j] = (i == j) ? 1.0 : 0.0; Xi[i][j] = 0.0;
    }
    __syncthreads();

    bool failed = false;
    for (int j = 0; j < nb; j++) {
        const double d = Lr[j][j];
        if (!(d > 0.0) && !failed) {
            failed = true;
            if (tid == 0 && info[t.s] == 0) info[t.s] = t.col0 + j + 1;
        }
        const double ljj = sqrt(d);
        __syncthreads();
        if (tid == 0) { Lr[j][j] = ljj; Li[j][j] = 0.0; }
        for (int i = j + 1 + tid; i < nb; i += 256) { Lr[i][j] /= ljj; Li[i][j] /= ljj; }
        __syncthreads();
        const int i = j + 1 + (tid & 63);
        if (i < nb) {
            const double ar = Lr[i][j], ai = Li[i][j];
            for (int k = j + 1 + (tid >> 6); k <= i; k += 4) {
                const double br = Lr[k][j], bi = Li[k][j];          // L(i,k) -= L(i,j) * conj(L(k,j))
                Lr[i][k] -= ar * br + ai * bi;
                Li[i][k] -= ai * br - ar * bi;
            }
        }
        __syncthreads();
    }

    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx & (NB - 1), j = idx >> 6;
        if (i < nb && j < nb && i >= j) { v2d v = {Lr[i][j], (i == j) ? 0.0 : Li[i][j]}; A[i + (int64_t)j * lda] = v; }
    }

    // X = inv(L): column sweep in LDS; thread (c, part) owns rows i = part (mod 4) of column c
    {
        const int c = tid & 63, part = tid >> 6;
        for (int k = 0; k < NB; k++) {
            const double dk = Lr[k][k];                               // real, > 0 (1 in the padding)
            const double xr = Xr[k][c] / dk, xi = Xi[k][c] / dk;
            __syncthreads();
            if (part == 0) { Xr[k][c] = xr; Xi[k][c] = xi; }
            for (int i = k + 1 + ((part - (k + 1)) & 3); i < NB; i += 4) {
                const double lr = Lr[i][k], li = Li[i][k];
                Xr[i][c] -= lr * xr - li * xi;
                Xi[i][c] -= lr * xi + li * xr;
            }
            __syncthreads();
        }
    }
    if (t.dinv_off >= 0) {
        v2d *__restrict__ D = (v2d *)Dinv + t.dinv_off;
        for (int idx = tid; idx < nb * nb; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            v2d v = {0.0, 0.0};
            if (i >= j) { v.x = Xr[i][j]; v.y = Xi[i][j]; }
            D[idx] = v;
        }
    }
}

//------------------------------------------------------------------------------
// complex solves
//------------------------------------------------------------------------------
// x1 <- inv(L11) x1 (TRANS = false)  or  x1 <- inv(L11)^H x1 (TRANS = true)
template <bool TRANS>
__global__ __launch_bounds__(256) void k_solve_diag_c(
    const PotrfTask *__restrict__ tasks, const double *__restrict__ Dinv,
    double *__restrict__ X, int64_t ldx, int nrhs)
{
    __shared__ double Dr[NB][NB + 1], Di[NB][NB + 1];
    __shared__ double xr[NB][8 + 1], xi[NB][8 + 1];
    const PotrfTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, nb = t.nb;
    const v2d *__restrict__ D = (const v2d *)Dinv + t.dinv_off;
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int i = idx % nb, j = idx / nb;
        const v2d v = D[idx];
        Dr[i][j] = v.x; Di[i][j] = v.y;
    }
    v2d *__restrict__ x1 = (v2d *)X + t.k1 + t.col0;
    for (int j0 = 0; j0 < nrhs; j0 += 8) {
        const int nj = min(8, nrhs - j0);
        __syncthreads();
        for (int idx = tid; idx < nb * nj; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            const v2d v = x1[i + (int64_t)(j0 + j) * ldx];
            xr[i][j] = v.x; xi[i][j] = v.y;
        }
        __syncthreads();
        for (int idx = tid; idx < nb * nj; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            double ar = 0.0, ai = 0.0;
            if (!TRANS) {
                for (int k = 0; k <= i; k++) {
                    const double dr = Dr[i][k], di = Di[i][k];
                    ar += dr * xr[k][j] - di * xi[k][j];
                    ai += dr * xi[k][j] + di * xr[k][j];
                }
            } else {
                for (int k = i; k < nb; k++) {
                    const double dr = Dr[k][i], di = -Di[k][i];       // conj(D(k,i))
                    ar += dr * xr[k][j] - di * xi[k][j];
                    ai += dr * xi[k][j] + di * xr[k][j];
                }
            }
            v2d v = {ar, ai};
            x1[i + (int64_t)(j0 + j) * ldx] = v;
        }
    }
}

__global__ __launch_bounds__(SOLVE_ROWS) void k_solve_fwd_update_c(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ tile_task, const double *__restrict__ Lx,
    const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    constexpr int NR = 2;
    __shared__ double yr[NB][NR], yi[NB][NR];
    const PotrfTask t = tasks[tile_task[blockIdx.x]];
    const int tid = threadIdx.x, nb = t.nb;
    const int r = t.col0 + nb + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS + tid;
    const bool ok = r < t.lda;
    const v2d *__restrict__ Lr = (const v2d *)Lx + t.a_off + (r - t.col0);
    int64_t row = 0;
    if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
    const v2d *__restrict__ x1 = (const v2d *)X + t.k1 + t.col0;
    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        const int nj = min(NR, nrhs - j0);
        __syncthreads();
        for (int idx = tid; idx < nb * NR; idx += SOLVE_ROWS) {
            const int c = idx % nb, j = idx / nb;
            v2d v = {0.0, 0.0};
            if (j < nj) v = x1[c + (int64_t)(j0 + j) * ldx];
            yr[c][j] = v.x; yi[c][j] = v.y;
        }
        __syncthreads();
        if (ok) {
            double ar[NR], ai[NR];
#pragma unroll
            for (int j = 0; j < NR; j++) { ar[j] = 0.0; ai[j] = 0.0; }
            for (int c = 0; c < nb; c++) {
                const v2d l = Lr[(int64_t)c * t.lda];
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    ar[j] += l.x * yr[c][j] - l.y * yi[c][j];
                    ai[j] += l.x * yi[c][j] + l.y * yr[c][j];
                }
            }
#pragma unroll
            for (int j = 0; j < NR; j++)
                if (j < nj) {
                    double *q = X + 2 * (row + (int64_t)(j0 + j) * ldx);
                    atomic_add_f64(q, -ar[j]);
                    atomic_add_f64(q + 1, -ai[j]);
                }
        }
    }
}

// backward: x1[c] -= sum_r conj(L[r][col0+c]) * X[row(r)]
__global__ __launch_bounds__(256) void k_solve_bwd_update_c(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ tile_task, const double *__restrict__ Lx,
    const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    constexpr int NR = 2;
    __shared__ double Tr[64][NB + 1], Ti[64][NB + 1];
    __shared__ double xr[64][NR], xi[64][NR];
    __shared__ double redr[4][NB][NR], redi[4][NB][NR];
    const PotrfTask t = tasks[tile_task[blockIdx.x]];
    const int tid = threadIdx.x, nb = t.nb;
    const int rbase = t.col0 + nb + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS;
    const int c = tid & 63, part = tid >> 6;
    double *__restrict__ x1 = X + 2 * (int64_t)(t.k1 + t.col0);
    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        const int nj = min(NR, nrhs - j0);
        double ar[NR], ai[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) { ar[j] = 0.0; ai[j] = 0.0; }
        for (int sub = 0; sub < SOLVE_ROWS / 64; sub++) {
            const int r0 = rbase + sub * 64;
            if (r0 >= t.lda) break;
            __syncthreads();
            {
                const int rl = tid & 63, r = r0 + rl;
                const bool ok = r < t.lda;
                const v2d *__restrict__ Lrow = (const v2d *)Lx + t.a_off + (r - t.col0);
                for (int cc = (tid >> 6); cc < nb; cc += 4) {
                    v2d v = {0.0, 0.0};
                    if (ok) v = Lrow[(int64_t)cc * t.lda];
                    Tr[rl][cc] = v.x; Ti[rl][cc] = v.y;
                }
                if (tid < 64) {
                    int64_t row = 0;
                    if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        v2d v = {0.0, 0.0};
                        if (ok && j < nj) v = ((const v2d *)X)[row + (int64_t)(j0 + j) * ldx];
                        xr[rl][j] = v.x; xi[rl][j] = v.y;
                    }
                }
            }
            __syncthreads();
            if (c < nb) {
                for (int rr = part * 16; rr < part * 16 + 16; rr++) {
                    const double lr = Tr[rr][c], li = -Ti[rr][c];       // conj
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        ar[j] += lr * xr[rr][j] - li * xi[rr][j];
                        ai[j] += lr * xi[rr][j] + li * xr[rr][j];
                    }
                }
            }
        }
        __syncthreads();
        if (c < nb) {
#pragma unroll
            for (int j = 0; j < NR; j++) { redr[part][c][j] = ar[j]; redi[part][c][j] = ai[j]; }
        }
        __syncthreads();
        if (part == 0 && c < nb) {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                if (j < nj) {
                    const double vr = redr[0][c][j] + redr[1][c][j] + redr[2][c][j] + redr[3][c][j];
                    const double vi = redi[0][c][j] + redi[1][c][j] + redi[2][c][j] + redi[3][c][j];
                    double *q = x1 + 2 * (c + (int64_t)(j0 + j) * ldx);
                    atomic_add_f64(q, -vr);
                    atomic_add_f64(q + 1, -vi);
                }
            }
        }
    }
}

}  // namespace mi355
